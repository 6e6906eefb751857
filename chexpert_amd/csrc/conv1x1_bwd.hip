// Fused input gradient + weight gradient of the dense-layer bottleneck 1x1 convolution.
//
//   dX[m][c] (+)= e_scale[c] * mask(x[m][c]) * sum_n dZ[m][n] * W[c][n]          (conv1x1_dgrad.hip)
//   dW[n][c]  += sum_m dZ[m][n] * relu(x[m][c]*e_sc[c] + e_sh[c])                 (conv_wgrad.hip, pw_wgrad_kernel)
//
// Run separately the two kernels each read dZ (two tensors under AFFINE2) and the activation slice x: 20 of the 60 GB they
// move per DenseNet121 step are that second pass.  Here one workgroup owns a 64-channel tile of x / dX / dW and a range of
// pixels and walks it in 128-pixel tiles:
//   * W[64 c][128 n] sits in LDS for the whole workgroup; the dZ tile (128 px x 128 n) is normalised into LDS once per tile
//     and serves both products -- as the B operand of the swapped input-gradient MFMA (rows of 272 B, ds_read_b128) and, read
//     with the transposing ds_read_b64_tr_b16, as the A operand of the weight-gradient MFMA (contraction over pixels);
//   * the mask epilogue runs on the accumulators as in conv1x1_dgrad.hip (lane = pixel, 8 consecutive channels after
//     v_permlane32_swap, 16-B loads / stores) and hands relu(bn(x)) -- already in registers for the mask -- to LDS as two
//     32-channel pixel-major images (64-B rows) for the second product;
//   * dW (128 x 64 fp32) and the channel sums S1, S2 live in registers across all tiles of the workgroup: one burst of
//     atomics at the end.  The channel tiles of one pixel range are neighbours in the grid and share dZ in L2.
#include <cstdlib>
#include <type_traits>
#include "common.h"

namespace {

#ifndef CX_PW_NT
#define CX_PW_NT 7                      // bit 0: x loads, bit 1: old-dX loads, bit 2: new-dX stores are non-temporal (pw_bwd2)
#endif
constexpr int PW_NT = CX_PW_NT;
constexpr int KD = 128;                 // dZ channels (n)
constexpr int BM = 128;                 // pixels per tile
constexpr int PITCH = KD * 2 + 16;      // 272 B
constexpr int A_BYTES = BM * PITCH;
constexpr int XH_PITCH = 64;            // 32 channels per image row
constexpr int XH_BYTES = BM * XH_PITCH; // one 32-channel image

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float half_sum(float v) {
  v = dpp_add<0xB1>(v);
  v = dpp_add<0x4E>(v);
  v = dpp_add<0x141>(v);
  v = dpp_add<0x140>(v);
  return v + __shfl_xor(v, 16);
}

// fragment of the 32x32x16 MFMA from a pixel-major image: this lane gets channel ch0 + (lane&31), pixels k0 + 8*(lane>>5) + 0..7
__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int pitch, int k0, int ch0, int lane) {
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const char* base = tile + (k0 + 8 * (g >> 1) + q) * pitch + (ch0 + 16 * (g & 1) + 4 * pp) * 2;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  // (joined by a shuffle + bit cast: assembled element by element the compiler emits a v_bfi per dword on the loaded registers and
  // waits for the read right where it is issued, not where the MFMA uses it -- common.h cx_join_tr)
  return cx_join_tr(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base)), __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + 4 * pitch)));
}

// BC = buffer channels per workgroup: 64 (256 threads, two workgroups per CU) or 128 (512 threads, one per CU: dZ is re-read by
// half as many channel tiles, for the layers with many input channels)
template <int PRO, bool ACC, int BC>
__global__ __launch_bounds__(BC * 4, BC == 64 ? 2 : 1) void pw_bwd_kernel(const CxConv p, float* __restrict__ dw, const int M,
                                                                         const int c_tiles, const int tiles_per_split,
                                                                         float* __restrict__ slab, const int dw_ld) {
  constexpr int NT = BC * 4;
  constexpr int W_BYTES = BC * PITCH;
  constexpr int COEF_BYTES = (5 * BC + 3 * KD) * 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ecoef = reinterpret_cast<float*>(smem);               // e_sc, e_sh, e_mu, e_r, e_scale [64] each, then pa, pb, pc [128]
  char* Wt = smem + COEF_BYTES;                                 // [64 c][272 B]
  char* At = Wt + W_BYTES;                                      // [128 px][272 B]
  char* Xh = At + A_BYTES;                                      // [BC/32][128 px][64 B]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 31, lh = lane >> 5;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int ct = id % c_tiles, split = id / c_tiles;
  const int c0 = ct * BC;
  const int m_tiles = (M + BM - 1) / BM;
  const int t0 = split * tiles_per_split;
  const int t1 = (t0 + tiles_per_split < m_tiles) ? t0 + tiles_per_split : m_tiles;

  const bf16* __restrict__ X = reinterpret_cast<const bf16*>(p.x);
  const bf16* __restrict__ X2 = reinterpret_cast<const bf16*>(p.x2);
  const bf16* __restrict__ Wp = reinterpret_cast<const bf16*>(p.w);
  const bf16* __restrict__ EX = reinterpret_cast<const bf16*>(p.ex);
  bf16* __restrict__ Y = reinterpret_cast<bf16*>(p.y);

  // ---- once per workgroup: epilogue vectors and the weight tile of this channel range
  if (tid < BC) {
    const int n = c0 + tid;
    const bool ok = n < p.N;
    ecoef[tid] = ok ? p.e_sc[n] : 0.f;
    ecoef[BC + tid] = ok ? p.e_sh[n] : 0.f;
    ecoef[2 * BC + tid] = ok ? p.e_mu[n] : 0.f;
    ecoef[3 * BC + tid] = ok ? p.e_r[n] : 0.f;
    ecoef[4 * BC + tid] = ok ? p.e_scale[n] : 0.f;
  }
  const int q = tid & 15, r0 = tid >> 4;
  constexpr int RS = NT / 16;                                   // rows covered by one pass of the workgroup
#pragma unroll
  for (int i = 0; i < BC / RS; ++i) {
    const int n = c0 + r0 + RS * i;
    const uint4 v = *reinterpret_cast<const uint4*>(Wp + (size_t)(n < p.N ? n : 0) * KD + q * 8);
    *reinterpret_cast<uint4*>(Wt + (r0 + RS * i) * PITCH + q * 16) = n < p.N ? v : make_uint4(0, 0, 0, 0);
  }
  if (PRO == CX_PRO_AFFINE2 && tid < KD) {
    ecoef[5 * BC + tid] = p.pa[tid];
    ecoef[5 * BC + KD + tid] = p.pb[tid];
    ecoef[5 * BC + 2 * KD + tid] = p.pc[tid];
  }
  const float* aco = ecoef + 5 * BC + q * 8;                    // AFFINE2 vectors of this thread's dZ channel chunk (LDS)
  __syncthreads();                                              // the dZ staging of the first tile reads them

  // input gradient: wave = (pixel sub-tile pw, 64-channel half ch); weight gradient: two 32x32 tiles of dW per wave
  const int pw = wave & 3, ch = wave >> 2;
  const int wn_[2] = {BC == 64 ? (wave >> 1) * 2 : (wave & 3), BC == 64 ? (wave >> 1) * 2 + 1 : (wave & 3)};      // n sub-tiles
  const int wc_[2] = {BC == 64 ? (wave & 1) : (wave >> 2) * 2, BC == 64 ? (wave & 1) : (wave >> 2) * 2 + 1};      // c sub-tiles
  f32x16 accw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[i][r] = 0.f;
  float s1[2][2][8], s2[2][2][8];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int e = 0; e < 8; ++e) s1[j][cc][e] = s2[j][cc][e] = 0.f;

  for (int mt = t0; mt < t1; ++mt) {
    const int m0 = mt * BM;
    // ---- read-modify-write operands of this lane's pixel (requested first: the longest round trip)
    const int m = m0 + pw * 32 + lrow;
    const bool pok = m < M;
    const int mc = pok ? m : M - 1;
    U128 xv[2][2], old[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const int n = c0 + ch * 64 + j * 32 + 8 * (2 * cc + lh);
        const int ncl = n < p.N ? n : 0;
        xv[j][cc].u = *reinterpret_cast<const uint4*>(EX + (size_t)mc * p.ldex + ncl);
        if (ACC) old[j][cc].u = *reinterpret_cast<const uint4*>(Y + (size_t)mc * p.ldy + ncl);
        else old[j][cc].u = make_uint4(0, 0, 0, 0);
      }
    // ---- dZ tile -> LDS (batches of four rows per thread)
#pragma unroll
    for (int hb = 0; hb < BM / (4 * RS); ++hb) {
      uint4 ru[4], rv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int mm = m0 + r0 + RS * (hb * 4 + i);
        const int mmc = mm < M ? mm : M - 1;
        ru[i] = *reinterpret_cast<const uint4*>(X + (size_t)mmc * p.ldx + q * 8);
        if (PRO == CX_PRO_AFFINE2) rv[i] = *reinterpret_cast<const uint4*>(X2 + (size_t)mmc * p.ldx2 + q * 8);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = r0 + RS * (hb * 4 + i);
        // branch-free: a branch on the row bound lets the compiler sink the loads under it, one vmcnt(0) each
        U128 o;
        if (PRO == CX_PRO_NONE) {
          o.u = ru[i];
        } else {
          o.u = cx_affine2_8(ru[i], rv[i], aco, aco + KD, aco + 2 * KD);
        }
        const unsigned keep = m0 + row < M ? 0xffffffffu : 0u;
        o.u.x &= keep; o.u.y &= keep; o.u.z &= keep; o.u.w &= keep;
        *reinterpret_cast<uint4*>(At + row * PITCH + q * 16) = o.u;
      }
    }
    __syncthreads();                              // dZ tile (first time: weights, coefficients) visible

    // ---- input gradient of this wave's 32 pixels x 64 channels: D[row = channel][col = pixel]
    f32x16 accd[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) accd[j][r] = 0.f;
    {
      const char* Ab = At + (pw * 32 + lrow) * PITCH + lh * 16;
      const char* Wb = Wt + (ch * 64 + lrow) * PITCH + lh * 16;
#pragma unroll
      for (int kk = 0; kk < KD / 16; ++kk) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(Ab + kk * 32);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const bf16x8 wf = *reinterpret_cast<const bf16x8*>(Wb + j * 32 * PITCH + kk * 32);
          accd[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, af, accd[j], 0, 0, 0);
        }
      }
    }
    // ---- mask epilogue on the accumulators; relu(bn(x)) goes to LDS for the second product
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const int cl = ch * 64 + j * 32 + 8 * (2 * cc + lh);
        const int n = c0 + cl;
        float v[8];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(accd[j][8 * cc + r4]), __float_as_uint(accd[j][8 * cc + 4 + r4]),
                                                           false, false);
          v[r4] = __uint_as_float(sw[0]);
          v[4 + r4] = __uint_as_float(sw[1]);
        }
        asm volatile("" ::: "memory");
        const float4 a0 = *reinterpret_cast<const float4*>(ecoef + cl), a1 = *reinterpret_cast<const float4*>(ecoef + cl + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(ecoef + BC + cl), b1 = *reinterpret_cast<const float4*>(ecoef + BC + cl + 4);
        const float4 e0 = *reinterpret_cast<const float4*>(ecoef + 4 * BC + cl), e1 = *reinterpret_cast<const float4*>(ecoef + 4 * BC + cl + 4);
        const float esc[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        const float esh[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        const float esl[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
        U128 o, xh;
        cx_mask_epi8(v, xv[j][cc].u, old[j][cc].u, esc, esh, esl, pok, s1[j][cc], s2[j][cc], o.u, xh.u);      // S2 = r * (sum dz*x - mu * S1), affine part at the end
        if (pok && n < p.N) *reinterpret_cast<uint4*>(Y + (size_t)m * p.ldy + n) = o.u;
        *reinterpret_cast<uint4*>(Xh + (ch * 2 + j) * XH_BYTES + (pw * 32 + lrow) * XH_PITCH + (2 * cc + lh) * 16) = xh.u;
      }
    __syncthreads();                              // relu(bn(x)) tile visible

    // ---- weight gradient: dW[n][c] += sum over the 128 pixels of the tile (k = pixel, transposing LDS reads)
#pragma unroll
    for (int kk = 0; kk < BM / 16; ++kk) {
      if (BC == 64) {                               // one c sub-tile, two n sub-tiles
        const bf16x8 bfr = tr_frag(Xh + wc_[0] * XH_BYTES, XH_PITCH, kk * 16, 0, lane);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const bf16x8 af = tr_frag(At, PITCH, kk * 16, wn_[i] * 32, lane);
          accw[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, accw[i], 0, 0, 0);
        }
      } else {                                      // one n sub-tile, two c sub-tiles
        const bf16x8 af = tr_frag(At, PITCH, kk * 16, wn_[0] * 32, lane);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const bf16x8 bfr = tr_frag(Xh + wc_[i] * XH_BYTES, XH_PITCH, kk * 16, 0, lane);
          accw[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, accw[i], 0, 0, 0);
        }
      }
    }
    __syncthreads();                              // both LDS tiles free for the next pixel tile
  }

  // ---- one burst of atomics per workgroup: dW tile, then S1 / S2
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = c0 + wc_[i] * 32 + lrow;
    if (c < p.N) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = wn_[i] * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        dw_out(dw, slab, (size_t)KD * p.N, split, slab ? (size_t)n * p.N + c : (size_t)n * dw_ld + c, accw[i][r]);
      }
    }
  }
  {
    float* scratch = reinterpret_cast<float*>(At);               // both tiles are free now (ecoef stays)
    wg_stat_begin<NT / 64>(scratch, BC, tid, NT);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float t1v = 0.f, t2v = 0.f;
#pragma unroll
      for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float a = half_sum(s1[j][cc][e]);
          const float b = half_sum(s2[j][cc][e]);
          if (lrow == 8 * cc + e) { t1v = a; t2v = b; }
        }
      if (lrow < 16) {
        const int cl = ch * 64 + j * 32 + 8 * (2 * (lrow >> 3) + lh) + (lrow & 7);
        wg_stat_put(scratch, BC, wave, cl, t1v, ecoef[3 * BC + cl] * (t2v - ecoef[2 * BC + cl] * t1v));
      }
    }
    wg_stat_end<NT / 64>(scratch, BC, tid, NT, p.stat_sum, p.stat_sq, p.stat_det, p.stat_det ? split : (int)blockIdx.x, p.stat_replicas,
                         p.stat_rstride, c0, p.N);
  }
}

// ---- v2 (round 2): the 128-channel form with 64-pixel tiles and the NEXT tile's operands in flight under the current tile's
// MFMA / epilogue phases.  v1 issued a tile's loads, waited, then ran two MFMA phases and a VALU epilogue with nothing in flight
// (3.3 TB/s of real traffic).  Halving the tile halves every per-thread register block (accumulators, channel sums, staging), which
// pays for one prefetched tile in registers: dZ / y1 of tile t+1 are requested right after tile t's staging barrier, x / old dX of
// tile t+2 right after tile t's epilogue.  The dZ and relu(bn(x)) images are double buffered in LDS: two barriers per tile.
//
// Round 4: the read-modify-write operands travel in ROW-COALESCED form.  Rounds 1-3 loaded x / old dX and stored dX in the
// accumulator layout of the input-gradient MFMA (lane = pixel, 8 consecutive channels): a wave-instruction then touches 32 bytes
// (two lanes) of 32 different pixels, and each 128-byte line is completed by four instructions of four waves.  A bare read-modify-
// write stream in that shape runs at 3.9-4.2 TB/s, in the staging shape (16 lanes = one pixel's 256 bytes, a wave-instruction =
// 8 whole lines) at 4.85-5.07 TB/s (scratch/segbench.hip, profiles/r04_segbench.txt) -- and the kernel sat at exactly the first
// figure (4.4 / 3.8 / 3.5 TB/s on the 80x80 / 40x40 / 20x20 maps).  Now x and old dX are requested in the staging shape (two tiles
// ahead, as before), parked in LDS tiles next to the dZ image, read from there in the accumulator layout by the mask epilogue, which
// writes the new dX back into the same LDS slot; after the epilogue's barrier the tile leaves as whole rows.
constexpr int BM2 = 64;
// Tiles are walked from the END of the pixel range: the kernel in front of this one on the stream (the 3x3 input gradient that wrote
// dZ) walked forward, so its most recent ~200 MB -- what the 256 MB memory-side cache still holds -- are this kernel's first reads
// (interleaved A/B on one box: 27.02 / 27.00 / 27.00 -> 26.92 / 26.95 / 26.96 ms per DenseNet121 step; -DCX_PW_FWD_ORDER: forward).
#ifdef CX_PW_FWD_ORDER
#define TILE_OF(mt) (mt)
#else
#define TILE_OF(mt) (m_tiles - 1 - (mt))
#endif
constexpr int A2_BYTES = BM2 * PITCH;
constexpr int XH2_BYTES = BM2 * XH_PITCH;
typedef uint32_t pw_u32x4 __attribute__((ext_vector_type(4)));   // (HIP's uint4 is a struct: register sets that are only copied end up in scratch)
__device__ __forceinline__ uint4 as_u4(const pw_u32x4 v) { return make_uint4(v[0], v[1], v[2], v[3]); }

template <int PRO, bool ACC>
__global__ __launch_bounds__(512, 1) void pw_bwd2_kernel(const CxConv p, float* __restrict__ dw, const int M, const int c_tiles,
                                                         const int tiles_per_split, float* __restrict__ slab, const int dw_ld) {
  constexpr int BC = 128, NT = 512;
  constexpr int W_BYTES = BC * PITCH;
  constexpr int COEF_BYTES = (5 * BC + 3 * KD) * 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ecoef = reinterpret_cast<float*>(smem);
  char* Wt = smem + COEF_BYTES;                                 // [128 c][272 B]
  char* At = Wt + W_BYTES;                                      // [2][64 px][272 B]   normalised dZ
  char* Xh = At + 2 * A2_BYTES;                                 // [2][4][64 px][64 B] relu(bn(x)), 32-channel images
  char* Xt = Xh + 2 * 4 * XH2_BYTES;                            // [64 px][272 B]      x of the current tile
  char* Ot = Xt + A2_BYTES;                                     // [2][64 px][272 B]   old dX -> new dX of the tile (in place)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 31, lh = lane >> 5;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int ct = id % c_tiles, split = id / c_tiles;
  const int c0 = ct * BC;
  const int m_tiles = (M + BM2 - 1) / BM2;
  const int t0 = split * tiles_per_split;
  const int t1 = (t0 + tiles_per_split < m_tiles) ? t0 + tiles_per_split : m_tiles;

  const bf16* __restrict__ X = reinterpret_cast<const bf16*>(p.x);
  const bf16* __restrict__ X2 = reinterpret_cast<const bf16*>(p.x2);
  const bf16* __restrict__ Wp = reinterpret_cast<const bf16*>(p.w);
  const bf16* __restrict__ EX = reinterpret_cast<const bf16*>(p.ex);
  bf16* __restrict__ Y = reinterpret_cast<bf16*>(p.y);

  if (tid < BC) {
    const int n = c0 + tid;
    const bool ok = n < p.N;
    ecoef[tid] = ok ? p.e_sc[n] : 0.f;
    ecoef[BC + tid] = ok ? p.e_sh[n] : 0.f;
    ecoef[2 * BC + tid] = ok ? p.e_mu[n] : 0.f;
    ecoef[3 * BC + tid] = ok ? p.e_r[n] : 0.f;
    ecoef[4 * BC + tid] = ok ? p.e_scale[n] : 0.f;
  }
  const int q = tid & 15, r0 = tid >> 4;                        // staging: 16-byte chunk q of rows r0 and r0 + 32 (dZ, x, old / new dX)
#pragma unroll
  for (int i = 0; i < BC / 32; ++i) {
    const int n = c0 + r0 + 32 * i;
    const uint4 v = *reinterpret_cast<const uint4*>(Wp + (size_t)(n < p.N ? n : 0) * KD + q * 8);
    *reinterpret_cast<uint4*>(Wt + (r0 + 32 * i) * PITCH + q * 16) = n < p.N ? v : make_uint4(0, 0, 0, 0);
  }
  if (PRO == CX_PRO_AFFINE2 && tid < KD) {
    ecoef[5 * BC + tid] = p.pa[tid];
    ecoef[5 * BC + KD + tid] = p.pb[tid];
    ecoef[5 * BC + 2 * KD + tid] = p.pc[tid];
  }
  const float* aco = ecoef + 5 * BC + q * 8;

  // input gradient: wave = (pixel half pw, 32-channel quarter cq); weight gradient: n sub-tile wn, c sub-tiles wc0, wc0 + 1
  const int pw = wave & 1, cq = wave >> 1;
  const int wn = wave & 3, wc0 = (wave >> 2) * 2;
  f32x16 accw[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) accw[i][r] = 0.f;
  float s1[2][8], s2[2][8];
#pragma unroll
  for (int cc = 0; cc < 2; ++cc)
#pragma unroll
    for (int e = 0; e < 8; ++e) s1[cc][e] = s2[cc][e] = 0.f;

  // this thread's channel chunk of the buffer rows it stages / stores (clamped for loads, exact for stores)
  const bool qok = c0 + q * 8 < p.N;
  const int qcl = qok ? c0 + q * 8 : 0;

  pw_u32x4 ru[2], rv[2];
  pw_u32x4 xsA[2], osA[2], xsB[2], osB[2];                      // x / old dX of two tiles, staging shape
  auto request_rmw = [&](int mt, pw_u32x4 (&xs)[2], pw_u32x4 (&os)[2]) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int mm = TILE_OF(mt) * BM2 + r0 + 32 * i;
      const int mmc = mm < M ? mm : M - 1;
      // non-temporal: these rows are not read again before ~1 GB of other traffic has passed (A/B on one box: -3 % per launch)
      if (PW_NT & 1) xs[i] = __builtin_nontemporal_load(reinterpret_cast<const pw_u32x4*>(EX + (size_t)mmc * p.ldex + qcl));
      else xs[i] = *reinterpret_cast<const pw_u32x4*>(EX + (size_t)mmc * p.ldex + qcl);
      if (ACC) {
        if (PW_NT & 2) os[i] = __builtin_nontemporal_load(reinterpret_cast<const pw_u32x4*>(Y + (size_t)mmc * p.ldy + qcl));
        else os[i] = *reinterpret_cast<const pw_u32x4*>(Y + (size_t)mmc * p.ldy + qcl);
      } else os[i] = pw_u32x4{0u, 0u, 0u, 0u};
    }
  };
  auto request_dz = [&](int mt) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int mm = TILE_OF(mt) * BM2 + r0 + 32 * i;
      const int mmc = mm < M ? mm : M - 1;
      ru[i] = *reinterpret_cast<const pw_u32x4*>(X + (size_t)mmc * p.ldx + q * 8);
      if (PRO == CX_PRO_AFFINE2) rv[i] = *reinterpret_cast<const pw_u32x4*>(X2 + (size_t)mmc * p.ldx2 + q * 8);
    }
  };
  // ---- first tile's dZ / y1, first two tiles' x / old dX
  request_dz(t0);
  request_rmw(t0, xsA, osA);
  request_rmw(t0 + 1 < t1 ? t0 + 1 : t0, xsB, osB);
  __syncthreads();                                              // coefficients (read by the staging) and weights visible

  // One 64-pixel tile; SEL = LDS buffer of the double-buffered images and the register set its x / old dX arrived in.
  auto tile = [&](const int mt, auto SelC, pw_u32x4 (&xs)[2], pw_u32x4 (&os)[2]) __attribute__((always_inline)) {
    constexpr int SEL = decltype(SelC)::value;
    const int m0 = TILE_OF(mt) * BM2;
    const bool tvalid = mt < t1;                // odd tile counts: the second half of the last pair is a masked dummy
    char* Ab_ = At + SEL * A2_BYTES;
    char* Xb_ = Xh + SEL * (4 * XH2_BYTES);
    char* Ob_ = Ot + SEL * A2_BYTES;
    // ---- dZ tile (already in registers) -> LDS, normalised; x and old dX as they came
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = r0 + 32 * i;
      U128 o;
      if (PRO == CX_PRO_NONE) o.u = as_u4(ru[i]);
      else o.u = cx_affine2_8(as_u4(ru[i]), as_u4(rv[i]), aco, aco + KD, aco + 2 * KD);
      const unsigned keep = (tvalid && m0 + row < M) ? 0xffffffffu : 0u;
      o.u.x &= keep; o.u.y &= keep; o.u.z &= keep; o.u.w &= keep;
      *reinterpret_cast<uint4*>(Ab_ + row * PITCH + q * 16) = o.u;
      *reinterpret_cast<pw_u32x4*>(Xt + row * PITCH + q * 16) = xs[i];
      *reinterpret_cast<pw_u32x4*>(Ob_ + row * PITCH + q * 16) = os[i];
    }
    __syncthreads();                              // tile visible; the other At / Xh / Ot buffers are free (their readers passed here)
    // ---- next tile's dZ / y1 (clamped tile index: the last iteration re-requests its own tile instead of branching)
    request_dz(mt + 1 < t1 ? mt + 1 : t1 - 1);
    // ---- input gradient of this wave's 32 pixels x 32 channels: D[row = channel][col = pixel]
    f32x16 accd;
#pragma unroll
    for (int r = 0; r < 16; ++r) accd[r] = 0.f;
    {
      const char* Ab = Ab_ + (pw * 32 + lrow) * PITCH + lh * 16;
      const char* Wb = Wt + (cq * 32 + lrow) * PITCH + lh * 16;
#pragma unroll
      for (int kk = 0; kk < KD / 16; ++kk) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(Ab + kk * 32);
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(Wb + kk * 32);
        accd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, af, accd, 0, 0, 0);
      }
    }
    // ---- mask epilogue on the accumulator layout (operands from the LDS tiles); relu(bn(x)) goes to LDS for the second product,
    // the new dX back into the slot the old one came from
    const int m = m0 + pw * 32 + lrow;
    const bool pok = tvalid && m < M;
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const int cl = cq * 32 + 8 * (2 * cc + lh);
      const int toff = (pw * 32 + lrow) * PITCH + cl * 2;
      const uint4 xv = *reinterpret_cast<const uint4*>(Xt + toff);
      const uint4 old = *reinterpret_cast<const uint4*>(Ob_ + toff);
      float v[8];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(accd[8 * cc + r4]), __float_as_uint(accd[8 * cc + 4 + r4]), false, false);
        v[r4] = __uint_as_float(sw[0]);
        v[4 + r4] = __uint_as_float(sw[1]);
      }
      const float4 a0 = *reinterpret_cast<const float4*>(ecoef + cl), a1 = *reinterpret_cast<const float4*>(ecoef + cl + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(ecoef + BC + cl), b1 = *reinterpret_cast<const float4*>(ecoef + BC + cl + 4);
      const float4 e0 = *reinterpret_cast<const float4*>(ecoef + 4 * BC + cl), e1 = *reinterpret_cast<const float4*>(ecoef + 4 * BC + cl + 4);
      const float esc[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      const float esh[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      const float esl[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
      U128 o, xh;
      cx_mask_epi8(v, xv, old, esc, esh, esl, pok, s1[cc], s2[cc], o.u, xh.u);
      *reinterpret_cast<uint4*>(Ob_ + toff) = o.u;
      *reinterpret_cast<uint4*>(Xb_ + cq * XH2_BYTES + (pw * 32 + lrow) * XH_PITCH + (2 * cc + lh) * 16) = xh.u;
    }
    // ---- x / old dX of the tile after next, into the register set this tile has just handed to LDS
    request_rmw(mt + 2 < t1 ? mt + 2 : t1 - 1, xs, os);
    __syncthreads();                              // relu(bn(x)) and the new dX tile visible
    // ---- the new dX leaves as whole rows (16 lanes = 256 contiguous bytes of one pixel)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = r0 + 32 * i;
      const pw_u32x4 o = *reinterpret_cast<const pw_u32x4*>(Ob_ + row * PITCH + q * 16);
      if (tvalid && m0 + row < M && qok) {
        if (PW_NT & 4) __builtin_nontemporal_store(o, reinterpret_cast<pw_u32x4*>(Y + (size_t)(m0 + row) * p.ldy + qcl));
        else *reinterpret_cast<pw_u32x4*>(Y + (size_t)(m0 + row) * p.ldy + qcl) = o;
      }
    }
    // ---- weight gradient: dW[n][c] += sum over the 64 pixels of the tile
#pragma unroll
    for (int kk = 0; kk < BM2 / 16; ++kk) {
      const bf16x8 af = tr_frag(Ab_, PITCH, kk * 16, wn * 32, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bf16x8 bfr = tr_frag(Xb_ + (wc0 + i) * XH2_BYTES, XH_PITCH, kk * 16, 0, lane);
        accw[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, accw[i], 0, 0, 0);
      }
    }
  };

  const int n_pairs = (t1 - t0 + 1) >> 1;
  for (int pr_ = 0; pr_ < n_pairs; ++pr_) {
    tile(t0 + 2 * pr_, std::integral_constant<int, 0>{}, xsA, osA);
    tile(t0 + 2 * pr_ + 1, std::integral_constant<int, 1>{}, xsB, osB);
  }

#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = c0 + (wc0 + i) * 32 + lrow;
    if (c < p.N) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        dw_out(dw, slab, (size_t)KD * p.N, split, slab ? (size_t)n * p.N + c : (size_t)n * dw_ld + c, accw[i][r]);
      }
    }
  }
  {
    __syncthreads();                                             // (the last tile's row stores read Ot)
    float* scratch = reinterpret_cast<float*>(At);               // both tiles are free now (ecoef stays)
    wg_stat_begin<8>(scratch, BC, tid, NT);
    float t1v = 0.f, t2v = 0.f;
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float a = half_sum(s1[cc][e]);
        const float b = half_sum(s2[cc][e]);
        if (lrow == 8 * cc + e) { t1v = a; t2v = b; }
      }
    if (lrow < 16) {
      const int cl = cq * 32 + 8 * (2 * (lrow >> 3) + lh) + (lrow & 7);
      wg_stat_put(scratch, BC, wave, cl, t1v, ecoef[3 * BC + cl] * (t2v - ecoef[2 * BC + cl] * t1v));
    }
    wg_stat_end<8>(scratch, BC, tid, NT, p.stat_sum, p.stat_sq, p.stat_det, p.stat_det ? split : (int)blockIdx.x, p.stat_replicas,
                   p.stat_rstride, c0, p.N);
  }
}

// ---- two dense layers per pass (round 4).  Layer l (A) and layer l - 1 (B) of a dense block both add their input
// gradient to the gradient-buffer channels [0, K') that both read (K' = layer B's input width; layer A's 32 newest channels are the
// slice layer B produced and go through the single-layer kernel first, because layer B's dZ depends on them).  With the x / dX tiles
// in LDS (pw_bwd2 above) the second layer of the pair needs NO x / dX traffic: per tile the workgroup stages dZ_A, x and old dX,
// runs layer A (input-gradient MFMA, mask epilogue that updates the dX tile in place and leaves relu(bn_A(x)) for the weight
// gradient), then layer B on the same tiles, and stores the dX rows once: per pixel 1024 + 6 K' bytes instead of 1024 + 12 K' for the
// two single-layer passes.  The dX tile carries bf16 between the layers, exactly the rounding the separate passes apply, so dX is
// bit-identical to them.  Budget: LDS 2 x W[128][128] + 2 dZ images + one relu(bn(x)) image set + x tile + dX tile + coefficients =
// 161,792 B; registers: both layers' weight-gradient accumulators (64) and channel sums (64) -- one x / old-dX register set (requested
// one pair-tile ahead) instead of two.
struct PairLayer {
  const bf16* x;         // dz2 (B,H,W,128)
  const bf16* x2;        // y1
  const bf16* w;         // packed transposed weights [c][128]
  const float *pa, *pb, *pc;
  const float *e_sc, *e_sh, *e_mu, *e_r, *e_scale;
  float *stat_sum, *stat_sq;
  float* dw;
  float* slab;
  int ldx, ldx2, dw_ld, stat_rstride;
};

template <bool ACC>
__global__ __launch_bounds__(512, 1) void pw_bwd2p_kernel(const CxConv p, const PairLayer LA, const PairLayer LB, const int M, const int c_tiles,
                                                          const int tiles_per_split) {
  constexpr int BC = 128, NT = 512;
  constexpr int W_BYTES = BC * PITCH;
  constexpr int CO = 3 * BC + 3 * KD;                           // coefficients per layer in LDS: e_sc, e_sh, e_scale [128]; pa, pb, pc [128]
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ecoef = reinterpret_cast<float*>(smem);                // [2][CO]
  char* Wt = smem + 2 * CO * 4;                                 // [2][128 c][272 B]
  char* At = Wt + 2 * W_BYTES;                                  // [2][64 px][272 B]   normalised dZ of layer A / layer B
  char* Xh = At + 2 * A2_BYTES;                                 // [4][64 px][64 B]    relu(bn(x)) of the layer being processed
  char* Xt = Xh + 4 * XH2_BYTES;                                // [64 px][272 B]      x of the tile
  char* Ot = Xt + A2_BYTES;                                     // [64 px][272 B]      old dX -> + layer A -> + layer B (in place)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 31, lh = lane >> 5;
  const int id = xcd_remap(blockIdx.x, gridDim.x);
  const int ct = id % c_tiles, split = id / c_tiles;
  const int c0 = ct * BC;
  const int m_tiles = (M + BM2 - 1) / BM2;
  const int t0 = split * tiles_per_split;
  const int t1 = (t0 + tiles_per_split < m_tiles) ? t0 + tiles_per_split : m_tiles;
  const bf16* __restrict__ EX = reinterpret_cast<const bf16*>(p.ex);
  bf16* __restrict__ Y = reinterpret_cast<bf16*>(p.y);

  const int q = tid & 15, r0 = tid >> 4;
#pragma unroll
  for (int L = 0; L < 2; ++L) {
    const PairLayer& P = L ? LB : LA;
    float* ec = ecoef + L * CO;
    if (tid < BC) {
      const int n = c0 + tid;
      const bool ok = n < p.N;
      ec[tid] = ok ? P.e_sc[n] : 0.f;
      ec[BC + tid] = ok ? P.e_sh[n] : 0.f;
      ec[2 * BC + tid] = ok ? P.e_scale[n] : 0.f;
      ec[3 * BC + tid] = P.pa[tid];
      ec[3 * BC + KD + tid] = P.pb[tid];
      ec[3 * BC + 2 * KD + tid] = P.pc[tid];
    }
#pragma unroll
    for (int i = 0; i < BC / 32; ++i) {
      const int n = c0 + r0 + 32 * i;
      const uint4 v = *reinterpret_cast<const uint4*>(P.w + (size_t)(n < p.N ? n : 0) * KD + q * 8);
      *reinterpret_cast<uint4*>(Wt + L * W_BYTES + (r0 + 32 * i) * PITCH + q * 16) = n < p.N ? v : make_uint4(0, 0, 0, 0);
    }
  }

  const int pw = wave & 1, cq = wave >> 1;
  const int wn = wave & 3, wc0 = (wave >> 2) * 2;
  f32x16 accw[2][2];
  // channel sums.  A lane's epilogue yields 2 x 16 values per layer and tile (sum of g and of g * x for its pixel's 16 channels): kept per
  // lane over the tiles as pw_bwd2 keeps them they would be 64 registers for the pair.  Each tile they are folded over the pixel
  // lanes instead -- lanes ^16 (permlane16_swap: rows 0 / 2 then hold the g sums, rows 1 / 3 the g * x sums) and ^8 (row_ror:8:
  // lanes with bit 3 clear keep channels e < 4, the others e >= 4) -- into 4 per (layer, cc); the remaining 8 lanes fold at the end.
  float st[2][2][4];
#pragma unroll
  for (int L = 0; L < 2; ++L) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) accw[L][i][r] = 0.f;
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int e = 0; e < 4; ++e) st[L][cc][e] = 0.f;
  }
  const bool hi8 = (lane & 8) != 0;
  const bool qok = c0 + q * 8 < p.N;
  const int qcl = qok ? c0 + q * 8 : 0;

  pw_u32x4 ru[2], rv[2], xs[2], os[2];
  auto request_rmw = [&](int mt) __attribute__((always_inline)) {
    const int mtc = mt < t1 ? mt : t1 - 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int mm = TILE_OF(mtc) * BM2 + r0 + 32 * i;
      const int mmc = mm < M ? mm : M - 1;
      xs[i] = __builtin_nontemporal_load(reinterpret_cast<const pw_u32x4*>(EX + (size_t)mmc * p.ldex + qcl));
      if (ACC) os[i] = __builtin_nontemporal_load(reinterpret_cast<const pw_u32x4*>(Y + (size_t)mmc * p.ldy + qcl));
      else os[i] = pw_u32x4{0u, 0u, 0u, 0u};
    }
  };
  auto request_dz = [&](const PairLayer& P, int mt) __attribute__((always_inline)) {
    const int mtc = mt < t1 ? mt : t1 - 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int mm = TILE_OF(mtc) * BM2 + r0 + 32 * i;
      const int mmc = mm < M ? mm : M - 1;
      ru[i] = *reinterpret_cast<const pw_u32x4*>(P.x + (size_t)mmc * P.ldx + q * 8);
      rv[i] = *reinterpret_cast<const pw_u32x4*>(P.x2 + (size_t)mmc * P.ldx2 + q * 8);
    }
  };
  // dZ of layer L (in ru / rv) -> its LDS image, normalised
  auto stage_dz = [&](const int L, const int m0) __attribute__((always_inline)) {
    const float* aco = ecoef + L * CO + 3 * BC + q * 8;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = r0 + 32 * i;
      U128 o;
      o.u = cx_affine2_8(as_u4(ru[i]), as_u4(rv[i]), aco, aco + KD, aco + 2 * KD);
      const unsigned keep = m0 + row < M ? 0xffffffffu : 0u;
      o.u.x &= keep; o.u.y &= keep; o.u.z &= keep; o.u.w &= keep;
      *reinterpret_cast<uint4*>(At + L * A2_BYTES + row * PITCH + q * 16) = o.u;
    }
  };
  // input gradient of this wave's 32 pixels x 32 channels for layer L, then the mask epilogue on the LDS tiles
  auto dgrad_epi = [&](auto LC, const int m0) __attribute__((always_inline)) {
    constexpr int L = decltype(LC)::value;
    f32x16 accd;
#pragma unroll
    for (int r = 0; r < 16; ++r) accd[r] = 0.f;
    {
      const char* Ab = At + L * A2_BYTES + (pw * 32 + lrow) * PITCH + lh * 16;
      const char* Wb = Wt + L * W_BYTES + (cq * 32 + lrow) * PITCH + lh * 16;
#pragma unroll
      for (int kk = 0; kk < KD / 16; ++kk) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(Ab + kk * 32);
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(Wb + kk * 32);
        accd = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, af, accd, 0, 0, 0);
      }
    }
    return accd;
  };
  auto epilogue = [&](auto LC, const f32x16& accd, const int m0) __attribute__((always_inline)) {
    constexpr int L = decltype(LC)::value;
    const float* ec = ecoef + L * CO;
    const int m = m0 + pw * 32 + lrow;
    const bool pok = m < M;
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const int cl = cq * 32 + 8 * (2 * cc + lh);
      const int toff = (pw * 32 + lrow) * PITCH + cl * 2;
      const uint4 xv = *reinterpret_cast<const uint4*>(Xt + toff);
      const uint4 old = *reinterpret_cast<const uint4*>(Ot + toff);
      float v[8];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(accd[8 * cc + r4]), __float_as_uint(accd[8 * cc + 4 + r4]), false, false);
        v[r4] = __uint_as_float(sw[0]);
        v[4 + r4] = __uint_as_float(sw[1]);
      }
      const float4 a0 = *reinterpret_cast<const float4*>(ec + cl), a1 = *reinterpret_cast<const float4*>(ec + cl + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(ec + BC + cl), b1 = *reinterpret_cast<const float4*>(ec + BC + cl + 4);
      const float4 e0 = *reinterpret_cast<const float4*>(ec + 2 * BC + cl), e1 = *reinterpret_cast<const float4*>(ec + 2 * BC + cl + 4);
      const float esc[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
      const float esh[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      const float esl[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
      U128 o, xh;
      float t1[8], t2[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) t1[e] = t2[e] = 0.f;
      cx_mask_epi8(v, xv, old, esc, esh, esl, pok, t1, t2, o.u, xh.u);
      float r8[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(t1[e]), __float_as_uint(t2[e]), false, false);
        r8[e] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float keep = hi8 ? r8[e + 4] : r8[e], send = hi8 ? r8[e] : r8[e + 4];
        st[L][cc][e] += keep + __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(send), 0x128, 0xf, 0xf, false));
      }
      *reinterpret_cast<uint4*>(Ot + toff) = o.u;
      *reinterpret_cast<uint4*>(Xh + cq * XH2_BYTES + (pw * 32 + lrow) * XH_PITCH + (2 * cc + lh) * 16) = xh.u;
    }
  };
  auto wgrad = [&](auto LC) __attribute__((always_inline)) {
    constexpr int L = decltype(LC)::value;
#pragma unroll
    for (int kk = 0; kk < BM2 / 16; ++kk) {
      const bf16x8 af = tr_frag(At + L * A2_BYTES, PITCH, kk * 16, wn * 32, lane);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bf16x8 bfr = tr_frag(Xh + (wc0 + i) * XH2_BYTES, XH_PITCH, kk * 16, 0, lane);
        accw[L][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, accw[L][i], 0, 0, 0);
      }
    }
  };
  using L0 = std::integral_constant<int, 0>;
  using L1 = std::integral_constant<int, 1>;

  request_dz(LA, t0);
  request_rmw(t0);
  __syncthreads();                                              // coefficients and weights visible

  for (int mt = t0; mt < t1; ++mt) {
    const int m0 = TILE_OF(mt) * BM2;
    // ---- dZ_A, x, old dX of this tile -> LDS (the same thread restages the slots it stored from: no barrier needed against the
    // previous tile's row stores)
    stage_dz(0, m0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = r0 + 32 * i;
      *reinterpret_cast<pw_u32x4*>(Xt + row * PITCH + q * 16) = xs[i];
      *reinterpret_cast<pw_u32x4*>(Ot + row * PITCH + q * 16) = os[i];
    }
    request_dz(LB, mt);                           // layer B's dZ of this tile, the next tile's x / old dX
    request_rmw(mt + 1);
    __syncthreads();                              // B1
    {
      const f32x16 accd = dgrad_epi(L0{}, m0);
      epilogue(L0{}, accd, m0);
    }
    stage_dz(1, m0);                              // (the B image's last readers -- the previous tile's weight gradient -- are behind B1)
    request_dz(LA, mt + 1);
    __syncthreads();                              // B2: relu(bn_A(x)), the dX tile after layer A and the dZ_B image visible
    wgrad(L0{});
    const f32x16 accd = dgrad_epi(L1{}, m0);
    __syncthreads();                              // B3: every wave is done with relu(bn_A(x))
    epilogue(L1{}, accd, m0);
    __syncthreads();                              // B4: relu(bn_B(x)) and the final dX tile visible
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = r0 + 32 * i;
      const pw_u32x4 o = *reinterpret_cast<const pw_u32x4*>(Ot + row * PITCH + q * 16);
      if (m0 + row < M && qok) __builtin_nontemporal_store(o, reinterpret_cast<pw_u32x4*>(Y + (size_t)(m0 + row) * p.ldy + qcl));
    }
    wgrad(L1{});
  }

  __syncthreads();                                               // the weight gradients' readers of At / Xh are done
#pragma unroll
  for (int L = 0; L < 2; ++L) {
    const PairLayer& P = L ? LB : LA;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = c0 + (wc0 + i) * 32 + lrow;
      if (c < p.N) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          dw_out(P.dw, P.slab, (size_t)KD * p.N, split, P.slab ? (size_t)n * p.N + c : (size_t)n * P.dw_ld + c, accw[L][i][r]);
        }
      }
    }
    float* scratch = reinterpret_cast<float*>(At);
    wg_stat_begin<8>(scratch, BC, tid, NT);
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = st[L][cc][e];
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        v += __shfl_xor(v, 4);
        const float gx = __shfl_xor(v, 16);                      // the g * x sum from the lane 16 up
        if ((lrow & 0x17) == 0) {
          const int cl = cq * 32 + 8 * (2 * cc + lh) + e + 4 * (lrow >> 3);
          const int n = c0 + cl;
          const float mu = n < p.N ? P.e_mu[n] : 0.f, rr = n < p.N ? P.e_r[n] : 0.f;
          wg_stat_put(scratch, BC, wave, cl, v, rr * (gx - mu * v));
        }
      }
    wg_stat_end<8>(scratch, BC, tid, NT, P.stat_sum, P.stat_sq, p.stat_det, p.stat_det ? split : (int)blockIdx.x, p.stat_replicas,
                   P.stat_rstride, c0, p.N);
  }
}

template <bool ACC>
int launch_bwd2p(const CxConv& pa_, const CxConv& pb_, float* dw_a, const int dw_ld_a, float* dw_b, float* scratch, long long scratch_floats,
                 hipStream_t st) {
  constexpr int BC = 128;
  const CxConv& p = pa_;
  const long long M = (long long)p.B * p.Ho * p.Wo;
  const int m_tiles = (int)((M + BM2 - 1) / BM2);
  const int c_tiles = (p.N + BC - 1) / BC;
  int splits = 256 / c_tiles;
  if (splits < 1) splits = 1;
  if (splits > m_tiles) splits = m_tiles;
  const int tps = (m_tiles + splits - 1) / splits;
  splits = (m_tiles + tps - 1) / tps;
  const size_t smem = 2 * (3 * BC + 3 * KD) * 4 + 2 * BC * PITCH + 2 * A2_BYTES + 4 * XH2_BYTES + 2 * A2_BYTES;
  if (const int e = stat_rows_check(pa_, splits)) return e;
  if (const int e = stat_rows_check(pb_, splits)) return e;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_bwd2p_kernel<ACC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr = true;
  }
  const size_t total = (size_t)KD * p.N;
  // both layers' slabs come out of the one scratch region (layer A first), or neither does
  float* slab_a = dw_slab(scratch, scratch_floats, 2ll * splits, (long long)total);
  float* slab_b = slab_a ? slab_a + (size_t)splits * total : nullptr;
  auto mk = [&](const CxConv& c, float* dw, int dw_ld, float* slab) {
    PairLayer L;
    L.x = (const bf16*)c.x; L.x2 = (const bf16*)c.x2; L.w = (const bf16*)c.w;
    L.pa = c.pa; L.pb = c.pb; L.pc = c.pc;
    L.e_sc = c.e_sc; L.e_sh = c.e_sh; L.e_mu = c.e_mu; L.e_r = c.e_r; L.e_scale = c.e_scale;
    L.stat_sum = c.stat_sum; L.stat_sq = c.stat_sq;
    L.dw = dw; L.slab = slab; L.ldx = c.ldx; L.ldx2 = c.ldx2; L.dw_ld = dw_ld; L.stat_rstride = c.stat_rstride;
    return L;
  };
  CX_KTAG("pw_bwd2p_kernel<%s>", ACC ? "true" : "false");
  hipLaunchKernelGGL((pw_bwd2p_kernel<ACC>), dim3(c_tiles * splits), dim3(512), smem, st, p, mk(pa_, dw_a, dw_ld_a, slab_a),
                     mk(pb_, dw_b, pb_.N, slab_b), (int)M, c_tiles, tps);
  if (const int e = launch_status()) return e;
  if (!slab_a) return 0;
  if (const int e = cx_dw_reduce_ld(dw_a, slab_a, total, splits, p.N, dw_ld_a, st)) return e;
  return cx_dw_reduce_ld(dw_b, slab_b, total, splits, p.N, pb_.N, st);
}

template <int PRO, bool ACC>
int launch_bwd2(const CxConv& p, float* dw, const int dw_ld, float* scratch, long long scratch_floats, hipStream_t st) {
  constexpr int BC = 128;
  const long long M = (long long)p.B * p.Ho * p.Wo;
  const int m_tiles = (int)((M + BM2 - 1) / BM2);
  const int c_tiles = (p.N + BC - 1) / BC;
  // grid target: one workgroup per CU.  A workgroup's fixed cost (34 KB weight tile, 64 KB weight-gradient slab) weighs against
  // 6-25 tiles of work on the 40x40 / 20x20 / 10x10 maps: 8-24 % less time than two per CU there (scratch/bench_pw.py); since the
  // weight gradients leave as slabs (one per workgroup, summed by a second launch) it is also ahead on the 80x80 maps (+0.6 % of the
  // DenseNet121 step; three per CU: -5 %)
  static const int wgs_env = cx_diag_int("CX_PW_BWD_WGS", 0);
  const int wgs = wgs_env ? wgs_env : 256;
  int splits = wgs / c_tiles;
  if (splits < 1) splits = 1;
  if (splits > m_tiles) splits = m_tiles;
  const int tps = (m_tiles + splits - 1) / splits;
  splits = (m_tiles + tps - 1) / tps;
  const size_t smem = (5 * BC + 3 * KD) * 4 + BC * PITCH + 2 * A2_BYTES + 2 * 4 * XH2_BYTES + 3 * A2_BYTES;
  if (const int e = stat_rows_check(p, splits)) return e;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_bwd2_kernel<PRO, ACC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr = true;
  }
  const size_t total = (size_t)KD * p.N;
  float* slab = dw_slab(scratch, scratch_floats, splits, (long long)total);
  CX_KTAG("pw_bwd2_kernel<%d, %s>", PRO, ACC ? "true" : "false");
  hipLaunchKernelGGL((pw_bwd2_kernel<PRO, ACC>), dim3(c_tiles * splits), dim3(512), smem, st, p, dw, (int)M, c_tiles, tps, slab, dw_ld);
  if (const int e = launch_status()) return e;
  return slab ? cx_dw_reduce_ld(dw, slab, total, splits, p.N, dw_ld, st) : 0;
}

template <int PRO, bool ACC, int BC>
int launch_bwd_bc(const CxConv& p, float* dw, const int dw_ld, float* scratch, long long scratch_floats, hipStream_t st) {
  const long long M = (long long)p.B * p.Ho * p.Wo;
  const int m_tiles = (int)((M + BM - 1) / BM);
  const int c_tiles = (p.N + BC - 1) / BC;
  int splits = (BC == 64 ? 1024 : 512) / c_tiles;    // ~4 (2) workgroups per CU in the grid
  if (splits < 1) splits = 1;
  if (splits > m_tiles) splits = m_tiles;
  const int tps = (m_tiles + splits - 1) / splits;
  splits = (m_tiles + tps - 1) / tps;
  const size_t smem = (5 * BC + 3 * KD) * 4 + BC * PITCH + A_BYTES + (BC / 32) * XH_BYTES;
  if (const int e = stat_rows_check(p, splits)) return e;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_bwd_kernel<PRO, ACC, BC>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)smem);
    attr = true;
  }
  const size_t total = (size_t)KD * p.N;
  float* slab = dw_slab(scratch, scratch_floats, splits, (long long)total);
  CX_KTAG("pw_bwd_kernel<%d, %s, %d>", PRO, ACC ? "true" : "false", BC);
  hipLaunchKernelGGL((pw_bwd_kernel<PRO, ACC, BC>), dim3(c_tiles * splits), dim3(BC * 4), smem, st, p, dw, (int)M, c_tiles, tps, slab, dw_ld);
  if (const int e = launch_status()) return e;
  return slab ? cx_dw_reduce_ld(dw, slab, total, splits, p.N, dw_ld, st) : 0;
}

template <int PRO, bool ACC>
int launch_bwd(const CxConv& p, float* dw, const int dw_ld, float* scratch, long long scratch_floats, hipStream_t st) {
  static const int force = cx_diag_int("CX_PW_BWD_BC", 0);
  const bool wide = force ? force == 128 : p.N >= 32;   // measured crossover (round 2: the v2 kernel also wins on the 64- and 96-channel layers)
  static const int v1 = cx_diag_int("CX_PW_BWD_V1", 0);   // diagnostic: the round-1 kernel
  if (wide && !v1) return launch_bwd2<PRO, ACC>(p, dw, dw_ld, scratch, scratch_floats, st);
  return wide ? launch_bwd_bc<PRO, ACC, 128>(p, dw, dw_ld, scratch, scratch_floats, st)
              : launch_bwd_bc<PRO, ACC, 64>(p, dw, dw_ld, scratch, scratch_floats, st);
}

}  // namespace

extern "C" int cx_conv1x1_dgrad_wgrad_ws(const CxConv* pp, float* dw, float* scratch, int64_t scratch_floats, void* stream) {
  return cx_conv1x1_dgrad_wgrad_ld_ws(pp, dw, pp ? pp->N : 0, scratch, scratch_floats, stream);
}

extern "C" int cx_conv1x1_dgrad_wgrad_ld_ws(const CxConv* pp, float* dw, int dw_ld, float* scratch, int64_t scratch_floats, void* stream) {
  if (!pp || !dw) return CX_EINVAL;
  const CxConv& p = *pp;
  if (dw_ld < p.N) return CX_EINVAL;
  if (!p.x || !p.w || !p.y || !p.ex || !p.e_sc || !p.e_sh || !p.e_mu || !p.e_r || !p.e_scale || !p.stat_sum || !p.stat_sq) return CX_EINVAL;
  if (p.dtype != CX_DT_BF16) return CX_EUNSUPPORTED;
  if (p.mode != CX_MODE_CONV || p.kh != 1 || p.kw != 1 || p.stride != 1 || p.pad != 0 || p.tstride > 1) return CX_EUNSUPPORTED;
  if (p.epilogue != CX_EPI_MASK || p.K != KD || (p.N % 8) || p.N <= 0) return CX_EUNSUPPORTED;
  if (p.prologue != CX_PRO_AFFINE2 && p.prologue != CX_PRO_NONE) return CX_EUNSUPPORTED;
  if (p.prologue == CX_PRO_AFFINE2 && (!p.x2 || !p.pa || !p.pb || !p.pc || (p.ldx2 % 8) || !aligned16(p.x2))) return CX_EINVAL;
  if (p.B <= 0 || p.Ho != p.H || p.Wo != p.W || (long long)p.B * p.H * p.W >= (1ll << 31)) return CX_ESHAPE;
  if ((p.ldx % 8) || (p.ldy % 8) || (p.ldex % 8) || p.ldx < KD || !aligned16(p.x) || !aligned16(p.y) || !aligned16(p.ex) || !aligned16(p.w))
    return CX_EALIGN;
  if (p.stat_replicas < 0 || (p.stat_replicas > 1 && p.stat_rstride < p.N)) return CX_EINVAL;
  hipStream_t st = as_stream(stream);
  if (p.prologue == CX_PRO_AFFINE2)
    return p.accumulate ? launch_bwd<CX_PRO_AFFINE2, true>(p, dw, dw_ld, scratch, scratch_floats, st)
                        : launch_bwd<CX_PRO_AFFINE2, false>(p, dw, dw_ld, scratch, scratch_floats, st);
  return p.accumulate ? launch_bwd<CX_PRO_NONE, true>(p, dw, dw_ld, scratch, scratch_floats, st)
                      : launch_bwd<CX_PRO_NONE, false>(p, dw, dw_ld, scratch, scratch_floats, st);
}

static int pair_check(const CxConv& p) {
  if (!p.x || !p.x2 || !p.w || !p.y || !p.ex || !p.pa || !p.pb || !p.pc || !p.e_sc || !p.e_sh || !p.e_mu || !p.e_r || !p.e_scale || !p.stat_sum ||
      !p.stat_sq)
    return CX_EINVAL;
  if (p.dtype != CX_DT_BF16 || p.mode != CX_MODE_CONV || p.kh != 1 || p.kw != 1 || p.stride != 1 || p.pad != 0 || p.tstride > 1 ||
      p.epilogue != CX_EPI_MASK || p.prologue != CX_PRO_AFFINE2 || p.K != KD || (p.N % 8) || p.N <= 0)
    return CX_EUNSUPPORTED;
  if (p.B <= 0 || p.Ho != p.H || p.Wo != p.W || (long long)p.B * p.H * p.W >= (1ll << 31)) return CX_ESHAPE;
  if ((p.ldx % 8) || (p.ldx2 % 8) || (p.ldy % 8) || (p.ldex % 8) || p.ldx < KD || !aligned16(p.x) || !aligned16(p.x2) || !aligned16(p.y) ||
      !aligned16(p.ex) || !aligned16(p.w))
    return CX_EALIGN;
  if (p.stat_replicas < 0 || (p.stat_replicas > 1 && p.stat_rstride < p.N)) return CX_EINVAL;
  return 0;
}

// ABI 9: two dense layers per pass (see pw_bwd2p_kernel).  a = the later layer (restricted to the channels both layers read), b = the
// earlier one; they share ex / y / N / geometry / accumulate / stat_det / stat_replicas.
extern "C" int cx_conv1x1_dgrad_wgrad_pair_ws(const CxConv* pa, const CxConv* pb, float* dw_a, int dw_ld_a, float* dw_b, float* scratch,
                                              int64_t scratch_floats, void* stream) {
  if (!pa || !pb || !dw_a || !dw_b) return CX_EINVAL;
  if (const int e = pair_check(*pa)) return e;
  if (const int e = pair_check(*pb)) return e;
  const CxConv &a = *pa, &b = *pb;
  if (a.ex != b.ex || a.y != b.y || a.N != b.N || a.B != b.B || a.H != b.H || a.W != b.W || a.ldex != b.ldex || a.ldy != b.ldy ||
      a.accumulate != b.accumulate || a.stat_det != b.stat_det || a.stat_replicas != b.stat_replicas ||
      dw_ld_a < a.N || a.stat_sum == b.stat_sum)
    return CX_EINVAL;
  hipStream_t st = as_stream(stream);
  return a.accumulate ? launch_bwd2p<true>(a, b, dw_a, dw_ld_a, dw_b, scratch, scratch_floats, st)
                      : launch_bwd2p<false>(a, b, dw_a, dw_ld_a, dw_b, scratch, scratch_floats, st);
}

extern "C" int cx_conv1x1_dgrad_wgrad(const CxConv* pp, float* dw, void* stream) {
  return cx_conv1x1_dgrad_wgrad_ws(pp, dw, nullptr, 0, stream);
}
