// 1x1 convolution with FEW input channels and MANY output channels: the bottleneck expansion of the ResNets (conv3: 256 -> 1024 on the
// 20x20 maps, 36 times per step, attn_aug_conv.py:159-211) and the input gradient of the reduction (conv1: dY 256 -> dX 1024).
//
//   Y[m][n] = sum_k pro(X[m][k]) * W[n][k]          K = 64 | 128 | 256,  N % 128 == 0,  N >= 2 K
//
// conv_mm.hip runs these as (M / 128) x (N / 128) independent tiles of FOUR k-steps each: a tile is pipeline fill, prologue arithmetic
// and epilogue with nothing in between, and every N tile stages (loads, normalises, writes to LDS) the same 128 x K activations again
// -- 82 us for 131 MB and 27 GFLOP on the 20x20 maps, three times the HBM time.  Here the roles of the operands are swapped:
//
//   * a wave's 64 activation rows live in REGISTERS as MFMA operand fragments for the whole kernel (K / 2 VGPRs: 128 at K = 256):
//     loaded straight from global memory in fragment layout (lane = row, 16 contiguous bytes per k group), normalised once;
//   * the workgroup (2 x 2 waves, 128 rows) walks ALL output channels in chunks of 128; the weights stream through a two-stage LDS
//     ring (64 input channels per step, from L2, requested one step ahead) -- the only operand the MFMAs read from LDS: half a
//     ds_read_b128 per MFMA instead of one, no activation staging inside the loop at all;
//   * a chunk's 64 x 64 accumulators of a wave leave through a wave-private LDS tile (two 32-row halves, no workgroup barrier), as
//     whole 128-byte rows with the channel sums of the values as stored; the two row-halves of the workgroup meet per chunk in LDS and
//     the workgroup writes ONE statistic row (deterministic mode) for all N channels.
//
// Two workgroups share a CU (76 KB of LDS, 256 VGPRs each wave); rows per workgroup are chosen so that the launch is whole rounds
// of 512 workgroups (M = 51200: 512 tiles of 100 rows).
#include <cstdlib>
#include "common.h"

namespace {

// (timing ablations behind -DXS_ABL: bit 0 drops the chunk epilogue, bit 1 the weight stream -- profiles/r05_xs_kernel.txt)
#ifndef XS_ABL
#define XS_ABL 0
#endif
constexpr int XS_PITCH = 144;                      // weight stage rows: 64 bf16 + 16 B pad
constexpr int XS_STAGE = 128 * XS_PITCH;           // 18432 B
constexpr int XS_EP = 68;                          // fp32 pitch of the wave-private epilogue tile (64 columns + 4)
typedef uint32_t xs_u32x4 __attribute__((ext_vector_type(4)));

template <int PRO>
struct XsCoef {
  static constexpr int v = (PRO == CX_PRO_NONE) ? 0 : (PRO == CX_PRO_AFFINE_RELU ? 2 : 3);
};

// one dword (two channels) of an activation fragment
template <int PRO>
__device__ __forceinline__ uint32_t xs_pro(const uint32_t g, const uint32_t y, const float* ca, const float* cb, const float* cc, const int j) {
  if (PRO == CX_PRO_NONE) return g;
  if (PRO == CX_PRO_AFFINE_RELU)
    return cx_relu_pk(cx_packbf(fmaf(cx_bf_lo(g), ca[2 * j], cb[2 * j]), fmaf(cx_bf_hi(g), ca[2 * j + 1], cb[2 * j + 1])));
  return cx_packbf(fmaf(cx_bf_lo(g), ca[2 * j], fmaf(cx_bf_lo(y), cb[2 * j], cc[2 * j])),
                   fmaf(cx_bf_hi(g), ca[2 * j + 1], fmaf(cx_bf_hi(y), cb[2 * j + 1], cc[2 * j + 1])));
}

template <int KC, int PRO, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void pw_xs_kernel(const CxConv p, const int M, const int rpt) {
  constexpr int K = KC * 64, NKK = K / 16;
  constexpr int NCO = XsCoef<PRO>::v;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* coef = reinterpret_cast<float*>(smem);                         // [NCO][K]
  char* Bs = smem + NCO * K * 4;                                        // [2][128 n][XS_PITCH]
  float* Es = reinterpret_cast<float*>(Bs + 2 * XS_STAGE);              // [4 waves][32][XS_EP]
  float* St = Es + 4 * 32 * XS_EP;                                      // [2 chunks][2 row halves][2][128]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = lane & 31, lh = lane >> 5;
  const int mt = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = mt * rpt;
  const char* __restrict__ X = reinterpret_cast<const char*>(p.x);
  const char* __restrict__ X2 = reinterpret_cast<const char*>(p.x2);
  const char* __restrict__ Wb = reinterpret_cast<const char*>(p.w);
  bf16* __restrict__ Y = reinterpret_cast<bf16*>(p.y);

  if (PRO != CX_PRO_NONE) {
    for (int i = tid; i < K; i += 256) {
      coef[i] = p.pa[i];
      coef[K + i] = p.pb[i];
      if (PRO == CX_PRO_AFFINE2) coef[2 * K + i] = p.pc[i];
    }
  }

  // ---- weight stream: thread = 16-byte chunk q of rows r0 + 32 u of a stage (128 output channels x 64 input channels)
  const int q = tid & 7, r0 = tid >> 3;
  const int NC = p.N >> 7, T = NC * KC;
  const char* wthr = Wb + ((size_t)r0 * K + q * 8) * 2;
  xs_u32x4 wreg[4];
  auto issue_w = [&](int t) __attribute__((always_inline)) {
    t = t < T ? t : T - 1;                             // (past the end: a harmless re-read of the last step)
    const int nc = t / KC, ks = t - nc * KC;
    const char* src = wthr + ((size_t)nc * 128 * K + ks * 64) * 2;
#pragma unroll
    for (int u = 0; u < 4; ++u) wreg[u] = *reinterpret_cast<const xs_u32x4*>(src + (size_t)u * 32 * K * 2);
  };
  auto store_w = [&](int stage) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < 4; ++u) *reinterpret_cast<xs_u32x4*>(Bs + stage * XS_STAGE + (r0 + 32 * u) * XS_PITCH + q * 16) = wreg[u];
  };
  issue_w(0);

  // ---- the wave's activation rows as operand fragments: afr[i][kk] = rows wm * 64 + i * 32 + lrow, channels kk * 16 + lh * 8 + 0..7
  xs_u32x4 afr[2][NKK];
  bool rok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rl = wm * 64 + i * 32 + lrow;
    const int m = m0 + rl;
    rok[i] = rl < rpt && m < M;
    const size_t mc = rok[i] ? m : M - 1;
    const char* xp = X + (mc * p.ldx + lh * 8) * 2;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) afr[i][kk] = *reinterpret_cast<const xs_u32x4*>(xp + kk * 32);
  }
  __syncthreads();                                     // coefficient table visible
  if (PRO != CX_PRO_NONE) {
    const float* cf = coef;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      xs_u32x4 sec[PRO == CX_PRO_AFFINE2 ? NKK : 1];
      if (PRO == CX_PRO_AFFINE2) {
        const size_t mc = rok[i] ? m0 + wm * 64 + i * 32 + lrow : M - 1;
        const char* xp = X2 + (mc * p.ldx2 + lh * 8) * 2;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) sec[kk] = *reinterpret_cast<const xs_u32x4*>(xp + kk * 32);
      }
#pragma unroll
      for (int kk = 0; kk < NKK; ++kk) {
        // (the table address is made to depend on the fragment it is for -- an opaque zero derived from the loaded register --: the
        // compiler otherwise reads all 16 k groups' coefficients ahead of the global loads' arrival, 256 registers, and spills them)
        uint32_t z;
        asm volatile("v_and_b32 %0, 0, %1" : "=v"(z) : "v"(afr[i][kk][0]));
        const int c0 = kk * 16 + lh * 8 + (int)z;
        float ca[8], cb[8], cc[8];
        *reinterpret_cast<float4*>(ca) = *reinterpret_cast<const float4*>(cf + c0);
        *reinterpret_cast<float4*>(ca + 4) = *reinterpret_cast<const float4*>(cf + c0 + 4);
        *reinterpret_cast<float4*>(cb) = *reinterpret_cast<const float4*>(cf + K + c0);
        *reinterpret_cast<float4*>(cb + 4) = *reinterpret_cast<const float4*>(cf + K + c0 + 4);
        if (PRO == CX_PRO_AFFINE2) {
          *reinterpret_cast<float4*>(cc) = *reinterpret_cast<const float4*>(cf + 2 * K + c0);
          *reinterpret_cast<float4*>(cc + 4) = *reinterpret_cast<const float4*>(cf + 2 * K + c0 + 4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) afr[i][kk][j] = xs_pro<PRO>(afr[i][kk][j], PRO == CX_PRO_AFFINE2 ? sec[kk][j] : 0u, ca, cb, cc, j);
        // (... and the group's results pass through a volatile statement: volatile statements keep their order, so group kk + 1's
        // table reads cannot start before group kk's arithmetic has consumed its coefficients)
        asm volatile("" : "+v"(afr[i][kk]));
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {                        // rows past the tile / the tensor multiply as zeros
    const uint32_t keep = rok[i] ? 0xffffffffu : 0u;
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) afr[i][kk] &= keep;
  }

  store_w(0);
  issue_w(1);

  // ---- epilogue geometry: item (row = it >> 3, 8-channel group cq = it & 7) of a 32 x 64 half tile, it = lane + 64 u
  const int cq = lane & 7;
  float* Ew = Es + wave * 32 * XS_EP;
  const bool want_stats = p.stat_sum != nullptr;
  const int srow = p.stat_det ? mt : (int)blockIdx.x;

  for (int nc = 0; nc < NC; ++nc) {
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KC; ++ks) {
      const int t = nc * KC + ks;
      __syncthreads();                                 // stage t & 1 is complete, nobody reads the other one any more
      if (ks == 0 && nc > 0 && want_stats && tid < 128) {
        // the previous chunk's channel sums: row halves in order, one statistic row per workgroup
        const float* s = St + ((nc - 1) & 1) * 512;
        const int n = (nc - 1) * 128 + tid;
        const float a = s[tid] + s[256 + tid];
        float b = s[128 + tid] + s[384 + tid];
        if (EPI == CX_EPI_JOIN) b = p.e_r[n] * (b - p.e_mu[n] * a);
        if (p.stat_det) {
          p.stat_sum[(size_t)srow * p.stat_rstride + n] = a;
          p.stat_sq[(size_t)srow * p.stat_rstride + n] = b;
        } else {
          const size_t rep = p.stat_replicas > 1 ? (size_t)(srow % p.stat_replicas) * p.stat_rstride : 0;
          atomicAdd(&p.stat_sum[rep + n], a);
          atomicAdd(&p.stat_sq[rep + n], b);
        }
      }
      const char* Bt = Bs + (t & 1) * XS_STAGE + (wn * 64 + lrow) * XS_PITCH + lh * 16;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(Bt + kk * 32);
        const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(Bt + 32 * XS_PITCH + kk * 32);
        const bf16x8 a0 = __builtin_bit_cast(bf16x8, afr[0][ks * 4 + kk]);
        const bf16x8 a1 = __builtin_bit_cast(bf16x8, afr[1][ks * 4 + kk]);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        if (kk == 1 && !(XS_ABL & 2)) {                // the next step's weights (requested a step ago) to the free stage, the one after requested
          store_w((t + 1) & 1);
          issue_w(t + 2);
        }
      }
    }

    // ---- chunk epilogue, wave-private: 32-row halves through Ew, whole 128-byte rows out
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
    const int ncol = nc * 128 + wn * 64 + cq * 8;
#pragma unroll
    for (int i = 0; i < ((XS_ABL & 1) ? 0 : 2); ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) Ew[((r & 3) + 8 * (r >> 2) + 4 * lh) * XS_EP + j * 32 + lrow] = acc[i][j][r];
      // (one wave: its LDS instructions complete in order, the reads below see the writes above)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int row = (lane >> 3) + 8 * u;
        const int rl = wm * 64 + i * 32 + row;
        const int m = m0 + rl;
        const float4 v0 = *reinterpret_cast<const float4*>(Ew + row * XS_EP + cq * 8);
        const float4 v1 = *reinterpret_cast<const float4*>(Ew + row * XS_EP + cq * 8 + 4);
        if (rl < rpt && m < M) {
          float t8[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
          bf16* yp = Y + (size_t)m * p.ldy + ncol;
          if (EPI == CX_EPI_JOIN) {
            // conv_mm.hip's join epilogue: the gradient of the join's output, rounded as a plain store would have stored it, then the
            // join's ReLU mask (the forward's sign bits) and the sums of its BatchNorm's backward
            const uint4 old = *reinterpret_cast<const uint4*>(yp);
            const uint4 xv = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16*>(p.ex) + (size_t)m * p.ldex + ncol);
            const unsigned mk = p.emask[cx_side_chunk((size_t)m, ncol >> 3, (size_t)M, p.N)];
            const uint32_t ow[4] = {old.x, old.y, old.z, old.w}, xw[4] = {xv.x, xv.y, xv.z, xv.w};
            uint32_t w4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const uint32_t bits = mk >> (2 * j);
              const uint32_t keep = ((bits & 1u) ? 0x0000ffffu : 0u) | ((bits & 2u) ? 0xffff0000u : 0u);
              w4[j] = cx_packbf(t8[2 * j] + cx_bf_lo(ow[j]), t8[2 * j + 1] + cx_bf_hi(ow[j])) & keep;
              const float dl = cx_bf_lo(w4[j]), du = cx_bf_hi(w4[j]);
              s1[2 * j] += dl;
              s1[2 * j + 1] += du;
              s2[2 * j] = fmaf(dl, cx_bf_lo(xw[j]), s2[2 * j]);          // (mean and 1 / std are applied to the workgroup's sums: S2 = r (sum dz x - mu sum dz),
              s2[2 * j + 1] = fmaf(du, cx_bf_hi(xw[j]), s2[2 * j + 1]);  //  as the fused 1x1 backward of the DenseNets does)
            }
            *reinterpret_cast<uint4*>(yp) = make_uint4(w4[0], w4[1], w4[2], w4[3]);
          } else {
            if (p.accumulate) {
              const uint4 old = *reinterpret_cast<const uint4*>(yp);
              const uint32_t ow[4] = {old.x, old.y, old.z, old.w};
#pragma unroll
              for (int j = 0; j < 4; ++j) { t8[2 * j] += cx_bf_lo(ow[j]); t8[2 * j + 1] += cx_bf_hi(ow[j]); }
            }
            *reinterpret_cast<uint4*>(yp) = cx_pack8_stats(t8, true, true, s1, s2);
          }
        }
      }
    }
    if (XS_ABL & 1) {                                  // (keeps the accumulators alive when the epilogue is ablated)
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += acc[0][0][r] + acc[0][1][r] + acc[1][0][r] + acc[1][1][r];
      if (t == 1.2345f) Y[0] = f2bf(t);
    }
    if (want_stats) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int d = 8; d < 64; d <<= 1) {
          s1[j] += __shfl_xor(s1[j], d);
          s2[j] += __shfl_xor(s2[j], d);
        }
      }
      if (lane < 8) {
        float* s = St + (nc & 1) * 512 + wm * 256 + wn * 64 + cq * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[j] = s1[j]; s[128 + j] = s2[j]; }
      }
    }
  }
  if (want_stats) {
    __syncthreads();
    if (tid < 128) {
      const float* s = St + ((NC - 1) & 1) * 512;
      const int n = (NC - 1) * 128 + tid;
      const float a = s[tid] + s[256 + tid];
      float b = s[128 + tid] + s[384 + tid];
      if (EPI == CX_EPI_JOIN) b = p.e_r[n] * (b - p.e_mu[n] * a);
      if (p.stat_det) {
        p.stat_sum[(size_t)srow * p.stat_rstride + n] = a;
        p.stat_sq[(size_t)srow * p.stat_rstride + n] = b;
      } else {
        const size_t rep = p.stat_replicas > 1 ? (size_t)(srow % p.stat_replicas) * p.stat_rstride : 0;
        atomicAdd(&p.stat_sum[rep + n], a);
        atomicAdd(&p.stat_sq[rep + n], b);
      }
    }
  }
}

template <int KC, int PRO, int EPI>
int launch_xs(const CxConv& p, hipStream_t st, int M, int rpt) {
  const int m_tiles = (M + rpt - 1) / rpt;
  const size_t smem = (size_t)XsCoef<PRO>::v * KC * 64 * 4 + 2 * XS_STAGE + 4 * 32 * XS_EP * 4 + 2 * 512 * 4;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_xs_kernel<KC, PRO, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    attr = true;
  }
  CX_KTAG("pw_xs_kernel<%d, %d, %d>", KC, PRO, EPI);
  hipLaunchKernelGGL((pw_xs_kernel<KC, PRO, EPI>), dim3(m_tiles), dim3(256), smem, st, p, M, rpt);
  return launch_status();
}

template <int PRO, int EPI>
int launch_xs_k(const CxConv& p, hipStream_t st, int M, int rpt) {
  if (p.K == 256) return launch_xs<4, PRO, EPI>(p, st, M, rpt);
  if (p.K == 128) return launch_xs<2, PRO, EPI>(p, st, M, rpt);
  return launch_xs<1, PRO, EPI>(p, st, M, rpt);
}

}  // namespace

// Called by cx_conv_gemm ahead of cx_try_conv_mm.  kernel_hint (ABI 10): on = 0 keeps the call off this kernel, form = 5 takes it
// whenever the shape is supported (tests, micro-benchmarks).
int cx_try_pw_xs(const CxConv& p, hipStream_t st, bool* handled) {
  *handled = false;
  static const int env_on = cx_diag_int("CX_XS", 1);
  const int hint_on = (p.kernel_hint & 0xff) - 1, hint_form = ((p.kernel_hint >> 8) & 0xff) - 1;
  if (hint_on == 0 || (hint_form >= 0 && hint_form != 5) || (!env_on && hint_form != 5)) return 0;
  if (p.mode != CX_MODE_CONV || p.dtype != CX_DT_BF16 || p.kh != 1 || p.kw != 1 || p.stride != 1 || p.pad != 0 || p.tstride > 1) return 0;
  if (p.K != 64 && p.K != 128 && p.K != 256) return 0;
  if ((p.N % 128) || p.N < 2 * p.K || p.N < 256) return 0;
  // (the join epilogue is correct here -- bit-identical to conv_mm.hip's -- but its operand loads sit behind the accumulators' trip through
  // LDS with no registers left to request them early: 166 against 114 us on the 20x20 maps.  Only on request, form 5.)
  if (p.epilogue != CX_EPI_STORE && !(p.epilogue == CX_EPI_JOIN && p.prologue == CX_PRO_AFFINE2 && hint_form == 5)) return 0;
  if (p.prologue != CX_PRO_NONE && p.prologue != CX_PRO_AFFINE_RELU && p.prologue != CX_PRO_AFFINE2) return 0;
  if (p.Ho != p.H || p.Wo != p.W) return 0;
  const long long M = (long long)p.B * p.H * p.W;
  if (M >= (1ll << 31)) return 0;
  // whole rounds of 512 workgroups (two per CU): rows per workgroup <= 128, a multiple of 4
  const long long full = (M + 127) / 128;
  const long long tiles = (full + 511) / 512 * 512;
  int rpt = (int)((M + tiles - 1) / tiles);
  rpt = (rpt + 3) & ~3;
  rpt = rpt < 64 ? 64 : rpt > 128 ? 128 : rpt;
  if (p.stat_det && p.stat_sum && ((M + rpt - 1) / rpt > p.stat_replicas)) rpt = 128;
  if (const int e = stat_rows_check(p, (int)((M + rpt - 1) / rpt))) {
    if (hint_form == 5) { *handled = true; return e; }
    return 0;                                          // (conv_mm.hip needs fewer rows)
  }
  *handled = true;
  if (p.epilogue == CX_EPI_JOIN) return launch_xs_k<CX_PRO_AFFINE2, CX_EPI_JOIN>(p, st, (int)M, rpt);
  if (p.prologue == CX_PRO_NONE) return launch_xs_k<CX_PRO_NONE, CX_EPI_STORE>(p, st, (int)M, rpt);
  if (p.prologue == CX_PRO_AFFINE_RELU) return launch_xs_k<CX_PRO_AFFINE_RELU, CX_EPI_STORE>(p, st, (int)M, rpt);
  return launch_xs_k<CX_PRO_AFFINE2, CX_EPI_STORE>(p, st, (int)M, rpt);
}
