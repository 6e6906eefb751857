// Input gradient of the dense-layer bottleneck 1x1 convolution (K = 128 gradient channels -> N buffer channels) with the
// ReLU + BatchNorm-backward epilogue (CX_EPI_MASK, read-modify-write of the block's gradient buffer).
//
//   dX[m][n] (+)= e_scale[n] * [ex[m][n]*e_sc[n]+e_sh[n] > 0] * sum_k dZ(m,k) * W[n][k]
//
// The generic implicit-GEMM kernel gives every 64-channel N tile its own workgroup, and each of them re-reads and
// re-normalises the same 128x128 dZ tile (two tensors under CX_PRO_AFFINE2): with N up to 1024 that re-fetch is most of the
// kernel's L2 traffic.  Here a workgroup keeps its dZ tile in LDS and walks up to TPC consecutive N tiles over it.
//   * MFMA operands are swapped (A = weights, B = dZ), so an accumulator lane owns ONE pixel and, after a
//     v_permlane32_swap between the wave halves, 8 consecutive channels: the read-modify-write operands are loaded and
//     stored 16 B per lane straight from / to registers -- no LDS transposition, no barrier in the epilogue.
//   * those operands (old gradient, mask source) are requested at the top of the N-tile iteration, before the barrier and
//     the MFMAs, and the next weight tile is fetched behind the current one (one barrier per N tile).
//   * per-channel sums S1, S2 reduce over the 32 pixel lanes with DPP adds and go to the replicated statistics vectors.
#include "common.h"

namespace {

constexpr int KD = 128;                 // gradient channels of the bottleneck
constexpr int BM = 128;                 // pixels per workgroup
constexpr int BN = 64;                  // buffer channels per N tile
constexpr int PITCH = KD * 2 + 16;      // 272 B: consecutive rows shift by one 16-B slot (conflict-free ds_read_b128)
constexpr int A_BYTES = BM * PITCH;
constexpr int W_BYTES = BN * PITCH;
constexpr int MAX_TPC = 8;              // N tiles per workgroup (coefficient vectors for 256 channels live in LDS)

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// sum over the 32 lanes of a wave half (both halves at once)
__device__ __forceinline__ float half_sum(float v) {
  v = dpp_add<0xB1>(v);       // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);       // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);      // row_half_mirror
  v = dpp_add<0x140>(v);      // row_mirror
  return v + __shfl_xor(v, 16);
}

template <int PRO, bool ACC>
__global__ __launch_bounds__(256, 2) void pw_dgrad_kernel(const CxConv p, const int M, const int n_tiles, const int tpc,
                                                         const int n_chunks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ecoef = reinterpret_cast<float*>(smem);                 // [5][MAX_TPC*BN]: e_sc, e_sh, e_mu, e_r, e_scale
  char* At = smem + 5 * MAX_TPC * BN * 4;
  char* Wt = At + A_BYTES;                                        // two weight tiles

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int wgid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = wgid / n_chunks, nc = wgid - mt * n_chunks;
  const int t0 = nc * tpc;
  const int t1 = (t0 + tpc < n_tiles) ? t0 + tpc : n_tiles;
  const int m0 = mt * BM;

  const bf16* __restrict__ X = reinterpret_cast<const bf16*>(p.x);
  const bf16* __restrict__ X2 = reinterpret_cast<const bf16*>(p.x2);
  const bf16* __restrict__ Wp = reinterpret_cast<const bf16*>(p.w);
  const bf16* __restrict__ EX = reinterpret_cast<const bf16*>(p.ex);
  bf16* __restrict__ Y = reinterpret_cast<bf16*>(p.y);

  // ---- epilogue coefficient vectors of this workgroup's channel range
  {
    const int nbase = t0 * BN, cnt = (t1 - t0) * BN;
    for (int i = tid; i < cnt; i += 256) {
      const int n = nbase + i;
      const bool ok = n < p.N;
      ecoef[0 * MAX_TPC * BN + i] = ok ? p.e_sc[n] : 0.f;
      ecoef[1 * MAX_TPC * BN + i] = ok ? p.e_sh[n] : 0.f;
      ecoef[2 * MAX_TPC * BN + i] = ok ? p.e_mu[n] : 0.f;
      ecoef[3 * MAX_TPC * BN + i] = ok ? p.e_r[n] : 0.f;
      ecoef[4 * MAX_TPC * BN + i] = ok ? p.e_scale[n] : 0.f;
    }
  }

  // ---- the dZ tile: 128 rows x 16 chunks, thread = chunk q of rows (tid>>4) + 16 i
  const int q = tid & 15, r0 = tid >> 4;
  {
    float ca[8], cb[8], cc[8];
    if (PRO == CX_PRO_AFFINE2) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { ca[j] = p.pa[q * 8 + j]; cb[j] = p.pb[q * 8 + j]; cc[j] = p.pc[q * 8 + j]; }
    }
#pragma unroll
    for (int hb = 0; hb < 2; ++hb) {
      uint4 ru[4], rv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int m = m0 + r0 + 16 * (hb * 4 + i);
        const int mc = m < M ? m : M - 1;                        // unconditional, clamped
        ru[i] = *reinterpret_cast<const uint4*>(X + (size_t)mc * p.ldx + q * 8);
        if (PRO == CX_PRO_AFFINE2) rv[i] = *reinterpret_cast<const uint4*>(X2 + (size_t)mc * p.ldx2 + q * 8);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = r0 + 16 * (hb * 4 + i);
        U128 o;
        if (m0 + row >= M) {
          o.u = make_uint4(0, 0, 0, 0);
        } else if (PRO == CX_PRO_NONE) {
          o.u = ru[i];
        } else {
          U128 u, v;
          u.u = ru[i];
          v.u = rv[i];
#pragma unroll
          for (int j = 0; j < 8; ++j) o.e[j] = f2bf(fmaf(bf2f(u.e[j]), ca[j], fmaf(bf2f(v.e[j]), cb[j], cc[j])));
        }
        *reinterpret_cast<uint4*>(At + row * PITCH + q * 16) = o.u;
      }
    }
  }

  // ---- weight tile staging: 64 rows x 16 chunks, thread = chunk q of rows (tid>>4) + 16 i
  uint4 rw[4];
  auto load_w = [&](int nt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = nt * BN + r0 + 16 * i;
      const int ncl = n < p.N ? n : 0;
      rw[i] = *reinterpret_cast<const uint4*>(Wp + (size_t)ncl * KD + q * 8);
    }
  };
  auto store_w = [&](int nt) {
    char* Wb = Wt + (nt & 1) * W_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int n = nt * BN + r0 + 16 * i;
      *reinterpret_cast<uint4*>(Wb + (r0 + 16 * i) * PITCH + q * 16) = n < p.N ? rw[i] : make_uint4(0, 0, 0, 0);
    }
  };
  load_w(t0);
  store_w(t0);

  const size_t rep = p.stat_replicas > 1 ? (size_t)(blockIdx.x % p.stat_replicas) * p.stat_rstride : 0;

  for (int nt = t0; nt < t1; ++nt) {
    const bool more = nt + 1 < t1;
    if (more) load_w(nt + 1);                 // older than the epilogue operands: its wait does not drain them

    // ---- read-modify-write operands of this tile: lane = pixel lrow of sub-tile i, chunks 2*cc + lh of this wave's 32 channels
    const int nw = nt * BN + wn * 32;
    U128 xv[2][2], old[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + (wm * 2 + i) * 32 + lrow;
      const int mc = m < M ? m : M - 1;
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const int n = nw + 8 * (2 * cc + lh);
        const int ncl = n < p.N ? n : 0;
        xv[i][cc].u = *reinterpret_cast<const uint4*>(EX + (size_t)mc * p.ldex + ncl);
        if (ACC) old[i][cc].u = *reinterpret_cast<const uint4*>(Y + (size_t)mc * p.ldy + ncl);
        else old[i][cc].u = make_uint4(0, 0, 0, 0);
      }
    }

    __syncthreads();                          // weight tile nt (and, first time, the dZ tile and coefficients) visible

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    {
      const char* Wb = Wt + (nt & 1) * W_BYTES + (wn * 32 + lrow) * PITCH + lh * 16;
      const char* Ab = At + ((wm * 2) * 32 + lrow) * PITCH + lh * 16;
#pragma unroll
      for (int kk = 0; kk < KD / 16; ++kk) {
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(Wb + kk * 32);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const bf16x8 af = *reinterpret_cast<const bf16x8*>(Ab + i * 32 * PITCH + kk * 32);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, af, acc[i], 0, 0, 0);   // D[row = channel][col = pixel]
        }
      }
    }
    if (more) store_w(nt + 1);                // the other buffer: last read before this iteration's barrier

    // ---- epilogue straight from the accumulators
    float s1[2][8], s2[2][8];
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int j = 0; j < 8; ++j) s1[cc][j] = s2[cc][j] = 0.f;
    const int cbase = (nt - t0) * BN + wn * 32;     // offset into the LDS coefficient vectors
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const int cl = cbase + 8 * (2 * cc + lh);
      const int n = nw + 8 * (2 * cc + lh);
      float esc[8], esh[8], emu[8], er[8], escale[8];
#pragma unroll
      for (int j4 = 0; j4 < 2; ++j4) {
        const float4 a = *reinterpret_cast<const float4*>(ecoef + 0 * MAX_TPC * BN + cl + 4 * j4);
        const float4 b = *reinterpret_cast<const float4*>(ecoef + 1 * MAX_TPC * BN + cl + 4 * j4);
        const float4 c = *reinterpret_cast<const float4*>(ecoef + 2 * MAX_TPC * BN + cl + 4 * j4);
        const float4 d = *reinterpret_cast<const float4*>(ecoef + 3 * MAX_TPC * BN + cl + 4 * j4);
        const float4 e = *reinterpret_cast<const float4*>(ecoef + 4 * MAX_TPC * BN + cl + 4 * j4);
        esc[4 * j4] = a.x; esc[4 * j4 + 1] = a.y; esc[4 * j4 + 2] = a.z; esc[4 * j4 + 3] = a.w;
        esh[4 * j4] = b.x; esh[4 * j4 + 1] = b.y; esh[4 * j4 + 2] = b.z; esh[4 * j4 + 3] = b.w;
        emu[4 * j4] = c.x; emu[4 * j4 + 1] = c.y; emu[4 * j4 + 2] = c.z; emu[4 * j4 + 3] = c.w;
        er[4 * j4] = d.x; er[4 * j4 + 1] = d.y; er[4 * j4 + 2] = d.z; er[4 * j4 + 3] = d.w;
        escale[4 * j4] = e.x; escale[4 * j4 + 1] = e.y; escale[4 * j4 + 2] = e.z; escale[4 * j4 + 3] = e.w;
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        // registers 8cc..8cc+3 hold channels 8*(2cc) + 4*lh + 0..3, registers 8cc+4..8cc+7 channels 8*(2cc+1) + 4*lh + 0..3;
        // swapping the upper half of the first group with the lower half of the second leaves every lane with the eight
        // consecutive channels 8*(2cc+lh) .. +7 of its pixel
        float v[8];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[i][8 * cc + r4]),
                                                           __float_as_uint(acc[i][8 * cc + 4 + r4]), false, false);
          v[r4] = __uint_as_float(sw[0]);
          v[4 + r4] = __uint_as_float(sw[1]);
        }
        const int m = m0 + (wm * 2 + i) * 32 + lrow;
        U128 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xf = bf2f(xv[i][cc].e[j]);
          const float dz = (fmaf(xf, esc[j], esh[j]) > 0.f) ? v[j] : 0.f;
          s1[cc][j] += dz;
          s2[cc][j] += dz * (xf - emu[j]) * er[j];
          o.e[j] = f2bf(fmaf(escale[j], dz, bf2f(old[i][cc].e[j])));
        }
        if (m < M && n < p.N) *reinterpret_cast<uint4*>(Y + (size_t)m * p.ldy + n) = o.u;
      }
    }
    // ---- statistics: 32 pixel lanes -> one total per channel; lane lrow = 8cc + j of each half adds channel 8*(2cc+lh) + j
    {
      float t1v = 0.f, t2v = 0.f;
#pragma unroll
      for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float a = half_sum(s1[cc][j]);
          const float b = half_sum(s2[cc][j]);
          if (lrow == 8 * cc + j) { t1v = a; t2v = b; }
        }
      if (lrow < 16) {
        const int n = nw + 8 * (2 * (lrow >> 3) + lh) + (lrow & 7);
        if (n < p.N) {
          atomicAdd(&p.stat_sum[rep + n], t1v);
          atomicAdd(&p.stat_sq[rep + n], t2v);
        }
      }
    }
  }
}

template <int PRO, bool ACC>
int launch_pw(const CxConv& p, hipStream_t st) {
  const long long M = (long long)p.B * p.Ho * p.Wo;
  const int m_tiles = (int)((M + BM - 1) / BM);
  const int n_tiles = (p.N + BN - 1) / BN;
  // as many N tiles per workgroup as keeps >= ~4 workgroups per CU in the grid
  int tpc = MAX_TPC;
  while (tpc > 1 && (long long)m_tiles * ((n_tiles + tpc - 1) / tpc) < 1024) --tpc;
  const int n_chunks = (n_tiles + tpc - 1) / tpc;
  tpc = (n_tiles + n_chunks - 1) / n_chunks;       // balance the chunks
  const size_t smem = 5 * MAX_TPC * BN * 4 + A_BYTES + 2 * W_BYTES;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_dgrad_kernel<PRO, ACC>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)smem);
    attr_set = true;
  }
  CX_KTAG("pw_dgrad_kernel<%d, %s>", PRO, ACC ? "true" : "false");
  hipLaunchKernelGGL((pw_dgrad_kernel<PRO, ACC>), dim3(m_tiles * n_chunks), dim3(256), smem, st, p, (int)M, n_tiles, tpc, n_chunks);
  return launch_status();
}

}  // namespace

// Called by cx_conv_gemm after its argument validation; *handled = false leaves the call to the generic kernel.
int cx_try_pw_dgrad(const CxConv& p, hipStream_t st, bool* handled) {
  *handled = false;
  if (p.mode != CX_MODE_CONV || p.kh != 1 || p.kw != 1 || p.stride != 1 || p.pad != 0 || p.tstride > 1) return 0;
  if (p.epilogue != CX_EPI_MASK || p.K != KD) return 0;
  if (p.prologue != CX_PRO_AFFINE2 && p.prologue != CX_PRO_NONE) return 0;
  if (p.stat_det) return 0;          // per-wave atomics here: the generic kernel writes deterministic statistic rows
  *handled = true;
  if (p.prologue == CX_PRO_AFFINE2)
    return p.accumulate ? launch_pw<CX_PRO_AFFINE2, true>(p, st) : launch_pw<CX_PRO_AFFINE2, false>(p, st);
  return p.accumulate ? launch_pw<CX_PRO_NONE, true>(p, st) : launch_pw<CX_PRO_NONE, false>(p, st);
}
