#include <cstdlib>
// Weight gradient of the implicit-GEMM convolutions:
//
//   dW[n][c][tap] += sum_m G(m, n) * A(m @ tap, c)          (reduction over output pixels m)
//
// Both MFMA operands are contracted over the SLOW (pixel) axis of the NHWC tensors, so their
// fragments are read from LDS with the gfx950 transposing read ds_read_b64_tr_b16
// (cdna_hip_programming.md T10): tiles are staged [32 pixels][channels] exactly as they lie in
// memory and every 16-lane group pulls a 4-pixel x 16-channel block transposed.  Row pitches are
// chosen == 64 B (mod 256 B) so the four pixel rows of a half-wave fall on disjoint bank quarters.
// G and A are produced by the same fused prologues as the forward kernel (BN backward as a
// two-tensor affine form; BN scale/shift + ReLU (+2x2 average) recomputed from the raw buffer).
// Each workgroup owns an (n-tile, c-tile, tap) and a contiguous pixel range; partial sums are added
// to the fp32 OIHW gradient with global_atomic_add_f32 (consecutive lanes = consecutive c).
#include "common.h"

namespace {

constexpr int PX = 32;   // pixels per step

template <int BNW, int BCW>
struct WGeo {
  static constexpr int WAVES_N = (BNW >= 64) ? 2 : 1;
  static constexpr int WAVES_C = (BCW / 32 < 4 / WAVES_N) ? BCW / 32 : 4 / WAVES_N;
  static constexpr int WAVES_K = 4 / (WAVES_N * WAVES_C);     // leftover waves split the two k16 sub-steps
  static constexpr int TN = BNW / (WAVES_N * 32);
  static constexpr int TC = BCW / (WAVES_C * 32);
  static constexpr int GP = (BNW == 32) ? 64 : BNW * 2 + 64;    // bytes; == 64 (mod 128) keeps rows on disjoint banks
  static constexpr int XP = (BCW == 32) ? 64 : BCW * 2 + 64;
  static constexpr int G_BYTES = PX * GP;
  static constexpr int X_BYTES = PX * XP;
  static constexpr int STAGE = G_BYTES + X_BYTES;
  static constexpr int G_CHUNKS = PX * BNW / 8;
  static constexpr int X_CHUNKS = PX * BCW / 8;
  static constexpr int G_PER = (G_CHUNKS + 255) / 256;
  static constexpr int X_PER = (X_CHUNKS + 255) / 256;
};

__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int pitch, int k0, int ch0, int lane) {
  // fragment of the 32x32x16 MFMA: this lane gets channel ch0 + (lane&31), pixels k0 + 8*(lane>>5) + 0..7
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const char* base = tile + (k0 + 8 * (g >> 1) + q) * pitch + (ch0 + 16 * (g & 1) + 4 * pp) * 2;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  // (joined by a shuffle + bit cast: assembled element by element the compiler emits a v_bfi per dword on the loaded registers and
  // waits for the read right where it is issued, not where the MFMA uses it -- common.h cx_join_tr)
  return cx_join_tr(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base)), __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + 4 * pitch)));
}

template <int BNW, int BCW, int GPRO, int XPRO, int MODE>
__global__ __launch_bounds__(256) void wgrad_kernel(const CxWgrad p, const int M, const int n_tiles, const int c_tiles,
                                                    const int taps, const int splits, const int steps_per_split,
                                                    float* __restrict__ slab) {
  using G = WGeo<BNW, BCW>;
  constexpr int NSRC = (MODE == CX_MODE_POOL2) ? 4 : 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wk = wave / (G::WAVES_N * G::WAVES_C);
  const int wn = (wave / G::WAVES_C) % G::WAVES_N, wc = wave % G::WAVES_C;

  int id = xcd_remap(blockIdx.x, gridDim.x);
  // c-tiles, then taps, then n-tiles vary fastest: the workgroups that re-read one pixel range of G (every c-tile and tap) and of x
  // (every n-tile; the taps' windows overlap) run side by side on one XCD and share it in L2
  const int ct = id % c_tiles;    id /= c_tiles;
  const int tap = id % taps;      id /= taps;
  const int nt = id % n_tiles;    id /= n_tiles;
  const int split = id;
  const int n0 = nt * BNW, c0 = ct * BCW;
  const int wdil = p.dil > 1 ? p.dil : 1;              // taps `dil` pixels apart
  const int dy0 = (MODE == CX_MODE_CONV) ? tap / p.kw : tap;
  const int dy = (MODE == CX_MODE_CONV) ? dy0 * wdil : tap;
  const int dx = (MODE == CX_MODE_CONV) ? (tap - dy0 * p.kw) * wdil : 0;

  const bf16* __restrict__ Gp = reinterpret_cast<const bf16*>(p.g);
  const bf16* __restrict__ G2 = reinterpret_cast<const bf16*>(p.g2);
  const bf16* __restrict__ X = reinterpret_cast<const bf16*>(p.x);

  // per-thread chunk coordinates (fixed channel chunk per thread)
  int grow[G::G_PER], gcq[G::G_PER], xrow[G::X_PER], xcq[G::X_PER];
  float ga[G::G_PER][8], gb[G::G_PER][8], gc[G::G_PER][8], xa[G::X_PER][8], xb[G::X_PER][8];
  bool gact[G::G_PER], xact[G::X_PER];
#pragma unroll
  for (int i = 0; i < G::G_PER; ++i) {
    const int ci = tid + 256 * i;
    gact[i] = ci < G::G_CHUNKS;
    grow[i] = ci / (BNW / 8);
    gcq[i] = ci % (BNW / 8);
    const int n = n0 + gcq[i] * 8;
    gact[i] = gact[i] && n < p.N;
    if (GPRO == CX_PRO_AFFINE2 && gact[i]) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { ga[i][j] = p.ga[n + j]; gb[i][j] = p.gb[n + j]; gc[i][j] = p.gc[n + j]; }
    }
  }
#pragma unroll
  for (int i = 0; i < G::X_PER; ++i) {
    const int ci = tid + 256 * i;
    xact[i] = ci < G::X_CHUNKS;
    xrow[i] = ci / (BCW / 8);
    xcq[i] = ci % (BCW / 8);
    const int c = c0 + xcq[i] * 8;
    xact[i] = xact[i] && c < ((MODE == CX_MODE_STEM) ? 32 : p.K);
    if (XPRO == CX_PRO_AFFINE_RELU && xact[i]) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { xa[i][j] = p.pa[c + j]; xb[i][j] = p.pb[c + j]; }
    }
  }

  uint4 rg[G::G_PER], rg2[G::G_PER], rx[G::X_PER][NSRC];
  bool gv[G::G_PER], xv[G::X_PER];
  const int hw = p.Ho * p.Wo;
  const int step0 = split * steps_per_split;
  int nsteps = (M + PX - 1) / PX - step0;
  if (nsteps > steps_per_split) nsteps = steps_per_split;

  // every load is unconditional on a clamped in-bounds address (a branch around a load serialises the prefetch)
  auto issue_loads = [&](int s) {
    const int mbase = (step0 + s) * PX;
#pragma unroll
    for (int i = 0; i < G::G_PER; ++i) {
      const int m = mbase + grow[i];
      gv[i] = gact[i] && m < M;
      const int mc = m < M ? m : M - 1;
      const int nc = gact[i] ? n0 + gcq[i] * 8 : 0;
      rg[i] = *reinterpret_cast<const uint4*>(Gp + (size_t)mc * p.ldg + nc);
      if (GPRO == CX_PRO_AFFINE2) rg2[i] = *reinterpret_cast<const uint4*>(G2 + (size_t)mc * p.ldg2 + nc);
    }
#pragma unroll
    for (int i = 0; i < G::X_PER; ++i) {
      const int m = mbase + xrow[i];
      const int mc = m < M ? m : M - 1;
      const int b = mc / hw;
      const int rem = mc - b * hw;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      const int cc = xact[i] ? c0 + xcq[i] * 8 : 0;
      if (MODE == CX_MODE_STEM) {
        const int iy = 2 * oy - 3 + dy, ix = 2 * ox - 4 + 2 * xcq[i];
        xv[i] = xact[i] && m < M && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        const int cy = xv[i] ? iy : 0, cx = xv[i] ? ix : 0;
        rx[i][0] = *reinterpret_cast<const uint4*>(X + ((size_t)(b * p.H + cy) * p.W + cx) * 4);
      } else if (MODE == CX_MODE_POOL2) {
        xv[i] = xact[i] && m < M;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const size_t pix = (size_t)(b * p.H + 2 * oy + (a >> 1)) * p.W + 2 * ox + (a & 1);
          rx[i][a] = *reinterpret_cast<const uint4*>(X + pix * p.ldx + cc);
        }
      } else {
        const int iy = oy * p.stride - p.pad + dy, ix = ox * p.stride - p.pad + dx;
        xv[i] = xact[i] && m < M && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        const int cy = xv[i] ? iy : 0, cx = xv[i] ? ix : 0;
        rx[i][0] = *reinterpret_cast<const uint4*>(X + ((size_t)(b * p.H + cy) * p.W + cx) * p.ldx + cc);
      }
    }
  };

  auto write_stage = [&](int buf) {
    char* Gt = smem + buf * G::STAGE;
    char* Xt = Gt + G::G_BYTES;
#pragma unroll
    for (int i = 0; i < G::G_PER; ++i) {
      if (tid + 256 * i < G::G_CHUNKS) {
        U128 o;
        if (!gv[i]) {
          o.u = make_uint4(0, 0, 0, 0);
        } else if (GPRO == CX_PRO_NONE) {
          o.u = rg[i];
        } else {
          o.u = cx_affine2_8(rg[i], rg2[i], ga[i], gb[i], gc[i]);
        }
        *reinterpret_cast<uint4*>(Gt + grow[i] * G::GP + gcq[i] * 16) = o.u;
      }
    }
#pragma unroll
    for (int i = 0; i < G::X_PER; ++i) {
      if (tid + 256 * i < G::X_CHUNKS) {
        U128 o;
        if (!xv[i]) {
          o.u = make_uint4(0, 0, 0, 0);
        } else if (XPRO == CX_PRO_NONE) {
          o.u = rx[i][0];
        } else {
          float acc[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
          for (int a = 0; a < NSRC; ++a) {
            const uint32_t w4[4] = {rx[i][a].x, rx[i][a].y, rx[i][a].z, rx[i][a].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              acc[2 * j] += fmaxf(fmaf(cx_bf_lo(w4[j]), xa[i][2 * j], xb[i][2 * j]), 0.f);
              acc[2 * j + 1] += fmaxf(fmaf(cx_bf_hi(w4[j]), xa[i][2 * j + 1], xb[i][2 * j + 1]), 0.f);
            }
          }
          {
            const float sc_ = NSRC == 4 ? 0.25f : 1.f;
            o.u = make_uint4(cx_packbf(acc[0] * sc_, acc[1] * sc_), cx_packbf(acc[2] * sc_, acc[3] * sc_), cx_packbf(acc[4] * sc_, acc[5] * sc_),
                             cx_packbf(acc[6] * sc_, acc[7] * sc_));
          }
        }
        *reinterpret_cast<uint4*>(Xt + xrow[i] * G::XP + xcq[i] * 16) = o.u;
      }
    }
  };

  f32x16 acc[G::TN][G::TC];
#pragma unroll
  for (int i = 0; i < G::TN; ++i)
#pragma unroll
    for (int j = 0; j < G::TC; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (nsteps > 0) {
    issue_loads(0);
    write_stage(0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
      const int buf = s & 1;
      if (s + 1 < nsteps) issue_loads(s + 1);
      const char* Gt = smem + buf * G::STAGE;
      const char* Xt = Gt + G::G_BYTES;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        if (G::WAVES_K == 2 && kk != wk) continue;
        bf16x8 af[G::TN], bfr[G::TC];
#pragma unroll
        for (int i = 0; i < G::TN; ++i) af[i] = tr_frag(Gt, G::GP, kk * 16, (wn * G::TN + i) * 32, lane);
#pragma unroll
        for (int j = 0; j < G::TC; ++j) bfr[j] = tr_frag(Xt, G::XP, kk * 16, (wc * G::TC + j) * 32, lane);
#pragma unroll
        for (int i = 0; i < G::TN; ++i)
#pragma unroll
          for (int j = 0; j < G::TC; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
      if (s + 1 < nsteps) write_stage(buf ^ 1);
      __syncthreads();
    }
  }

  // ---- atomics into the OIHW gradient
  const int lrow = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int i = 0; i < G::TN; ++i)
#pragma unroll
    for (int j = 0; j < G::TC; ++j) {
      const int c = c0 + (wc * G::TC + j) * 32 + lrow;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + (wn * G::TN + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (n >= p.N) continue;
        if (MODE == CX_MODE_STEM) {
          const int kx = (c >> 2) - 1, ch = c & 3;
          if (c < 32 && kx >= 0 && ch < 3)
            dw_out(p.dw, slab, (size_t)p.N * 147, split * G::WAVES_K + wk, ((size_t)(n * 3 + ch) * 7 + tap) * 7 + kx, acc[i][j][r]);
        } else if (c < p.K) {
          dw_out(p.dw, slab, (size_t)p.N * p.K * taps, split * G::WAVES_K + wk, ((size_t)n * p.K + c) * taps + tap, acc[i][j][r]);
        }
      }
    }
}

template <int BNW, int BCW, int GPRO, int XPRO, int MODE>
int launch(const CxWgrad& p, hipStream_t st) {
  using G = WGeo<BNW, BCW>;
  const int M = p.B * p.Ho * p.Wo;
  const int taps = (MODE == CX_MODE_STEM) ? 7 : p.kh * p.kw;
  const int Kc = (MODE == CX_MODE_STEM) ? 32 : p.K;
  const int n_tiles = (p.N + BNW - 1) / BNW, c_tiles = (Kc + BCW - 1) / BCW;
  const int total_steps = (M + PX - 1) / PX;
  int splits = p.splits;
  if (splits <= 0) {
    splits = 2048 / (n_tiles * c_tiles * taps);      // ~8 workgroups per CU: measured 5-13 % faster than 1024 on the 1x1 layers
    if (splits < 1) splits = 1;
  }
  if (splits > total_steps) splits = total_steps;
  const int sps = (total_steps + splits - 1) / splits;
  splits = (total_steps + sps - 1) / sps;
  const size_t smem = 2 * G::STAGE;
  // (each k-group of waves holds its own partial sums: a slab per (split, group))
  const size_t wtotal = (MODE == CX_MODE_STEM) ? (size_t)p.N * 147 : (size_t)p.N * p.K * taps;
  const int slabs = splits * G::WAVES_K;
  float* slab = dw_slab(p.scratch, p.scratch_floats, slabs, (long long)wtotal);
  CX_KTAG("wgrad_kernel<%d, %d, %d, %d, %d>", BNW, BCW, GPRO, XPRO, MODE);
  hipLaunchKernelGGL((wgrad_kernel<BNW, BCW, GPRO, XPRO, MODE>), dim3(n_tiles * c_tiles * taps * splits), dim3(256), smem,
                     st, p, M, n_tiles, c_tiles, taps, splits, sps, slab);
  if (const int e = launch_status()) return e;
  return slab ? cx_dw_reduce(p.dw, slab, wtotal, slabs, st) : 0;
}


// ------------------------------------------------------------------------------------------------ stem (7x7 s2 p3)
// All seven row-taps in one workgroup: the output-gradient tile G (two 64-channel tensors under AFFINE2, ~1.7 GB at
// bs=256) is staged once per 32-pixel step and multiplied against the seven 8-pixel x 4-channel input windows, instead
// of one workgroup per tap re-reading G seven times (measured 12.2 GB fetched per launch before).
constexpr int ST_GP = 64 * 2 + 64;          // G tile pitch (bytes)
constexpr int ST_XP = 64;                   // X tile pitch: 32 bf16
constexpr int ST_G_BYTES = PX * ST_GP;
constexpr int ST_X_BYTES = PX * ST_XP;
constexpr int ST_STAGE = ST_G_BYTES + 7 * ST_X_BYTES;

// STRIP (Wo a multiple of 32: a step is 32 consecutive outputs of ONE image row): the seven input rows of the step are staged
// once as flat 72-pixel strips (8 B per pixel) and the MFMA operand of output pixel px is the strip read at a pitch of 16 B --
// rows that overlap by six pixels -- instead of 32 separate 8-pixel windows per row (14 KB through L1 per step; 4 KB now, one
// 16-B chunk per thread).
constexpr int ST_SROW = 36 * 16;            // strip row: 72 pixels
template <int GPRO, bool STRIP>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const CxWgrad p, const int M, const int steps_per_split,
                                                         float* __restrict__ slab) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bf16* __restrict__ Gp = reinterpret_cast<const bf16*>(p.g);
  const bf16* __restrict__ G2 = reinterpret_cast<const bf16*>(p.g2);
  const bf16* __restrict__ X = reinterpret_cast<const bf16*>(p.x);
  const int hw = p.Ho * p.Wo;

  // G: one 16-B chunk per thread (32 px x 8 chunks); X: 7 taps x 32 px x 4 chunks = 896 chunks, 4 slots per thread
  const int grow = tid >> 3, gcq = tid & 7;
  float ga[8], gb[8], gc[8];
  if (GPRO == CX_PRO_AFFINE2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) { ga[j] = p.ga[gcq * 8 + j]; gb[j] = p.gb[gcq * 8 + j]; gc[j] = p.gc[gcq * 8 + j]; }
  }
  const int step0 = xcd_remap(blockIdx.x, gridDim.x) * steps_per_split;      // (consecutive pixel ranges share input rows: one XCD)
  int nsteps = (M + PX - 1) / PX - step0;
  if (nsteps > steps_per_split) nsteps = steps_per_split;

  uint4 rg, rg2, rx[4];
  bool gv, xv[4];
  auto issue_loads = [&](int s) {
    const int mbase = (step0 + s) * PX;
    {
      const int m = mbase + grow;
      gv = m < M;
      const int mc = gv ? m : M - 1;
      rg = *reinterpret_cast<const uint4*>(Gp + (size_t)mc * p.ldg + gcq * 8);
      if (GPRO == CX_PRO_AFFINE2) rg2 = *reinterpret_cast<const uint4*>(G2 + (size_t)mc * p.ldg2 + gcq * 8);
    }
    if (STRIP) {
      const int cc = tid < 252 ? tid : 0;
      const int tap = cc / 36, q = cc - tap * 36;
      const int b = mbase / hw;
      const int rem = mbase - b * hw;
      const int oy = rem / p.Wo, ox0 = rem - oy * p.Wo;
      const int iy = 2 * oy - 3 + tap, ix = 2 * ox0 - 4 + 2 * q;
      xv[0] = tid < 252 && mbase < M && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const int cy = xv[0] ? iy : 0, cx = xv[0] ? ix : 0;
      rx[0] = *reinterpret_cast<const uint4*>(X + ((size_t)(b * p.H + cy) * p.W + cx) * 4);
    } else
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ci = tid + 256 * i;                // < 896 for i < 3; i == 3 only for tid < 128
      const int cc = ci < 896 ? ci : 0;
      const int tap = cc >> 7, row = (cc >> 2) & 31, q = cc & 3;
      const int m = mbase + row;
      const int mc = m < M ? m : M - 1;
      const int b = mc / hw;
      const int rem = mc - b * hw;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      const int iy = 2 * oy - 3 + tap, ix = 2 * ox - 4 + 2 * q;
      xv[i] = ci < 896 && m < M && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
      const int cy = xv[i] ? iy : 0, cx = xv[i] ? ix : 0;
      rx[i] = *reinterpret_cast<const uint4*>(X + ((size_t)(b * p.H + cy) * p.W + cx) * 4);
    }
  };
  auto write_stage = [&](int buf) {
    char* Gt = smem + buf * ST_STAGE;
    char* Xt = Gt + ST_G_BYTES;
    U128 o;
    if (!gv) {
      o.u = make_uint4(0, 0, 0, 0);
    } else if (GPRO == CX_PRO_NONE) {
      o.u = rg;
    } else {
      o.u = cx_affine2_8(rg, rg2, ga, gb, gc);
    }
    *reinterpret_cast<uint4*>(Gt + grow * ST_GP + gcq * 16) = o.u;
    if (STRIP) {
      if (tid < 252) *reinterpret_cast<uint4*>(Xt + tid * 16) = xv[0] ? rx[0] : make_uint4(0, 0, 0, 0);     // [tap][36 chunks]
    } else
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ci = tid + 256 * i;
      if (ci < 896) {
        const int tap = ci >> 7, row = (ci >> 2) & 31, q = ci & 3;
        *reinterpret_cast<uint4*>(Xt + tap * ST_X_BYTES + row * ST_XP + q * 16) = xv[i] ? rx[i] : make_uint4(0, 0, 0, 0);
      }
    }
  };

  // wave w owns taps w and w+4 (wave 3: tap 3 only), both 32-channel halves of N = 64
  f32x16 acc[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][i][r] = 0.f;

  if (nsteps > 0) {
    issue_loads(0);
    write_stage(0);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
      const int buf = s & 1;
      if (s + 1 < nsteps) issue_loads(s + 1);
      const char* Gt = smem + buf * ST_STAGE;
      const char* Xt = Gt + ST_G_BYTES;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 af[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) af[i] = tr_frag(Gt, ST_GP, kk * 16, i * 32, lane);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int tap = wave + 4 * t;
          if (tap < 7) {
            const bf16x8 bfr = STRIP ? tr_frag(Xt + tap * ST_SROW, 16, kk * 16, 0, lane)
                                     : tr_frag(Xt + tap * ST_X_BYTES, ST_XP, kk * 16, 0, lane);
#pragma unroll
            for (int i = 0; i < 2; ++i) acc[t][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr, acc[t][i], 0, 0, 0);
          }
        }
      }
      if (s + 1 < nsteps) write_stage(buf ^ 1);
      __syncthreads();
    }
  }

  const int c = lane & 31, lh = lane >> 5;          // c = 4*(input pixel in the 8-wide window) + channel
  const int kx = (c >> 2) - 1, ch = c & 3;
  if (kx >= 0 && ch < 3) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int tap = wave + 4 * t;
      if (tap >= 7) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          dw_out(p.dw, slab, (size_t)64 * 147, (int)blockIdx.x, ((size_t)(n * 3 + ch) * 7 + tap) * 7 + kx, acc[t][i][r]);
        }
    }
  }
}

template <int GPRO>
int launch_stem(const CxWgrad& p, hipStream_t st) {
  const int M = p.B * p.Ho * p.Wo;
  const int total_steps = (M + PX - 1) / PX;
  int splits = p.splits > 0 ? p.splits : 1024;
  if (splits > total_steps) splits = total_steps;
  const int sps = (total_steps + splits - 1) / splits;
  splits = (total_steps + sps - 1) / sps;
  const size_t wtotal = (size_t)64 * 147;
  float* slab = dw_slab(p.scratch, p.scratch_floats, splits, (long long)wtotal);
  static const int strip_on = cx_diag_int("CX_STEM_STRIP", 1);
  if (strip_on && p.Wo % PX == 0 && (p.W & 1) == 0) {
    CX_KTAG("stem_wgrad_kernel<%d, true>", GPRO);
    hipLaunchKernelGGL((stem_wgrad_kernel<GPRO, true>), dim3(splits), dim3(256), 2 * ST_STAGE, st, p, M, sps, slab);
  } else {
    CX_KTAG("stem_wgrad_kernel<%d, false>", GPRO);
    hipLaunchKernelGGL((stem_wgrad_kernel<GPRO, false>), dim3(splits), dim3(256), 2 * ST_STAGE, st, p, M, sps, slab);
  }
  if (const int e = launch_status()) return e;
  return slab ? cx_dw_reduce(p.dw, slab, wtotal, splits, st) : 0;
}


// ------------------------------------------------------------------------------------------------ 1x1, N a multiple of 128
// dW[128 n][K c] += sum_px dZ[px][n] * A[px][c].  The generic kernel above walks 32 pixels per barrier with 4 MFMAs per
// wave and keeps ~20 KB per workgroup in flight; it measured 1.9-2.9 TB/s on these layers.  This variant: 512 threads, a
// 128 x 128 tile (the dZ operand -- two tensors under AFFINE2 -- is re-read by half as many channel tiles), 64 pixels per
// step (48 KB per workgroup in flight, two workgroups per CU), prologue vectors in LDS, 8 MFMAs per wave and step.
constexpr int PW_PX = 64;
constexpr int PW_PITCH = 128 * 2 + 64;                 // == 64 B (mod 256 B) for the transposing reads
constexpr int PW_TILE = PW_PX * PW_PITCH;
constexpr int PW_STAGE = 2 * PW_TILE;
constexpr int PW_COEF = 5 * 128 * 4;                   // ga gb gc (dZ channels) | pa pb (this tile's input channels)

template <int GPRO, int XPRO>
__global__ __launch_bounds__(512, 2) void pw_wgrad_kernel(const CxWgrad p, const int M, const int c_tiles, const int n_tiles,
                                                         const int steps_per_split, float* __restrict__ slab) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* coef = reinterpret_cast<float*>(smem);
  char* tiles = smem + PW_COEF;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave >> 2, wc = wave & 3;
  int id = xcd_remap(blockIdx.x, gridDim.x);
  const int ct = id % c_tiles;                         // channel tiles of one pixel range are neighbours: dZ shared in L2
  id /= c_tiles;
  const int nt = id % n_tiles;                         // N > 128 (ResNet 1x1 convolutions): 128-row tiles of dW
  const int split = id / n_tiles;
  const int c0 = ct * 128, n0 = nt * 128;
  const bf16* __restrict__ Gp = reinterpret_cast<const bf16*>(p.g);
  const bf16* __restrict__ G2 = reinterpret_cast<const bf16*>(p.g2);
  const bf16* __restrict__ X = reinterpret_cast<const bf16*>(p.x);

  if (tid < 128) {
    coef[tid] = GPRO == CX_PRO_AFFINE2 ? p.ga[n0 + tid] : 1.f;
    coef[128 + tid] = GPRO == CX_PRO_AFFINE2 ? p.gb[n0 + tid] : 0.f;
    coef[256 + tid] = GPRO == CX_PRO_AFFINE2 ? p.gc[n0 + tid] : 0.f;
    const int c = c0 + tid;
    coef[384 + tid] = (XPRO == CX_PRO_AFFINE_RELU && c < p.K) ? p.pa[c] : 0.f;
    coef[512 + tid] = (XPRO == CX_PRO_AFFINE_RELU && c < p.K) ? p.pb[c] : 0.f;
  }
  __syncthreads();

  // chunk slot i of this thread: row (tid >> 4) + 32 i of the 64-pixel step, 16-B channel chunk q = tid & 15 (same for all)
  const int q = tid & 15, r0 = tid >> 4;
  const bool xact = c0 + q * 8 < p.K;
  const int xc = xact ? c0 + q * 8 : 0;
  const int step0 = split * steps_per_split;
  int nsteps = (M + PW_PX - 1) / PW_PX - step0;
  if (nsteps > steps_per_split) nsteps = steps_per_split;

  uint4 rg[2], rg2[2], rx[2];
  bool rv[2];
  auto issue_loads = [&](int s) __attribute__((always_inline)) {
    const int mbase = (step0 + s) * PW_PX;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = mbase + r0 + 32 * i;
      rv[i] = m < M;
      const int mc = rv[i] ? m : M - 1;                // unconditional loads on clamped addresses
      rg[i] = *reinterpret_cast<const uint4*>(Gp + (size_t)mc * p.ldg + n0 + q * 8);
      if (GPRO == CX_PRO_AFFINE2) rg2[i] = *reinterpret_cast<const uint4*>(G2 + (size_t)mc * p.ldg2 + n0 + q * 8);
      rx[i] = *reinterpret_cast<const uint4*>(X + (size_t)mc * p.ldx + xc);
    }
  };
  auto write_stage = [&](int buf) __attribute__((always_inline)) {
    char* Gt = tiles + buf * PW_STAGE;
    char* Xt = Gt + PW_TILE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      U128 og, ox;
      if (!rv[i]) {
        og.u = make_uint4(0, 0, 0, 0);
      } else if (GPRO == CX_PRO_NONE) {
        og.u = rg[i];
      } else {
        U128 u, v;
        u.u = rg[i];
        v.u = rg2[i];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          og.e[j] = f2bf(fmaf(bf2f(u.e[j]), coef[q * 8 + j], fmaf(bf2f(v.e[j]), coef[128 + q * 8 + j], coef[256 + q * 8 + j])));
      }
      if (!rv[i] || !xact) {
        ox.u = make_uint4(0, 0, 0, 0);
      } else if (XPRO == CX_PRO_NONE) {
        ox.u = rx[i];
      } else {
        U128 v;
        v.u = rx[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) ox.e[j] = f2bf(fmaxf(fmaf(bf2f(v.e[j]), coef[384 + q * 8 + j], coef[512 + q * 8 + j]), 0.f));
      }
      *reinterpret_cast<uint4*>(Gt + (r0 + 32 * i) * PW_PITCH + q * 16) = og.u;
      *reinterpret_cast<uint4*>(Xt + (r0 + 32 * i) * PW_PITCH + q * 16) = ox.u;
    }
  };

  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // ONE LDS stage (40 KB, so that two workgroups share a CU) and two barriers per 64-pixel step; the next step's 48 KB are in
  // flight in registers while this one is multiplied
  if (nsteps > 0) {
    issue_loads(0);
    for (int s = 0; s < nsteps; ++s) {
      write_stage(0);
      __syncthreads();
      if (s + 1 < nsteps) issue_loads(s + 1);
      const char* Gt = tiles;
      const char* Xt = Gt + PW_TILE;
#pragma unroll
      for (int kk = 0; kk < PW_PX / 16; ++kk) {
        const bf16x8 bfr = tr_frag(Xt, PW_PITCH, kk * 16, wc * 32, lane);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const bf16x8 af = tr_frag(Gt, PW_PITCH, kk * 16, (wn * 2 + i) * 32, lane);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[i], 0, 0, 0);
        }
      }
      __syncthreads();
    }
  }

  const int lrow = lane & 31, lh = lane >> 5;
  const int c = c0 + wc * 32 + lrow;
  if (c < p.K) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + (wn * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        dw_out(p.dw, slab, (size_t)p.N * p.K, split, (size_t)n * p.K + c, acc[i][r]);
      }
  }
}

template <int GPRO, int XPRO>
int launch_pw_wgrad(const CxWgrad& p, hipStream_t st) {
  const int M = p.B * p.Ho * p.Wo;
  const int c_tiles = (p.K + 127) / 128, n_tiles = p.N / 128;
  const int total_steps = (M + PW_PX - 1) / PW_PX;
  int splits = p.splits > 0 ? p.splits : 1024 / (c_tiles * n_tiles);
  if (splits < 1) splits = 1;
  if (splits > total_steps) splits = total_steps;
  const int sps = (total_steps + splits - 1) / splits;
  splits = (total_steps + sps - 1) / sps;
  const size_t smem = PW_COEF + PW_STAGE;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_wgrad_kernel<GPRO, XPRO>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)smem);
    attr = true;
  }
  const size_t wtotal = (size_t)p.N * p.K;
  float* slab = dw_slab(p.scratch, p.scratch_floats, splits, (long long)wtotal);
  CX_KTAG("pw_wgrad_kernel<%d, %d>", GPRO, XPRO);
  hipLaunchKernelGGL((pw_wgrad_kernel<GPRO, XPRO>), dim3(c_tiles * n_tiles * splits), dim3(512), smem, st, p, M, c_tiles, n_tiles, sps, slab);
  if (const int e = launch_status()) return e;
  return slab ? cx_dw_reduce(p.dw, slab, wtotal, splits, st) : 0;
}

template <int GPRO, int XPRO, int MODE>
int launch_tile(const CxWgrad& p, hipStream_t st) {
  if (MODE == CX_MODE_STEM) return launch<64, 32, GPRO, XPRO, MODE>(p, st);
  if (p.N == 32) return launch<32, 128, GPRO, XPRO, MODE>(p, st);
  // (a partial last 128-row tile is masked like a partial 64-row one; from 96 rows up the wider tile halves the re-reads of x)
  if (p.N % 128 == 0 || p.N % 128 >= 96 || p.N >= 224) return launch<128, 64, GPRO, XPRO, MODE>(p, st);
  return launch<64, 64, GPRO, XPRO, MODE>(p, st);
}

}  // namespace

int cx_try_pc_wgrad(const CxWgrad& p, hipStream_t st, bool* handled);      // conv3x3_pc.hip
int cx_try_ring_wgrad(const CxWgrad& p, hipStream_t st, bool* handled);    // conv3x3_ring.hip
int cx_try_strip_wgrad(const CxWgrad& p, hipStream_t st, bool* handled);   // conv3x3_strip.hip

int cx_conv_wgrad_f32(const CxWgrad& p, hipStream_t st);                    // conv_f32.hip

int cx_try_wgrad_mm(const CxWgrad& p, hipStream_t st, bool* handled);          // wgrad_mm.hip

extern "C" int cx_conv_wgrad(const CxWgrad* pp, void* stream) {
  if (!pp) return CX_EINVAL;
  const CxWgrad& p = *pp;
  if (!p.g || !p.x || !p.dw) return CX_EINVAL;
  if (p.B <= 0 || p.H <= 0 || p.W <= 0 || p.Ho <= 0 || p.Wo <= 0) return CX_ESHAPE;
  if (p.dtype == CX_DT_F32) {             // fp32 storage mode (conv_f32.hip)
    if (p.K <= 0 || p.N <= 0 || (long long)p.B * p.Ho * p.Wo >= (1ll << 31)) return CX_ESHAPE;
    if (!aligned16(p.g) || !aligned16(p.x)) return CX_EALIGN;
    if (p.mode == CX_MODE_CONV) {
      if (p.kh <= 0 || p.kw <= 0 || p.stride <= 0 || p.pad < 0 || p.dil < 0) return CX_ESHAPE;
      const int wd = p.dil > 1 ? p.dil : 1;
      if (p.Ho != (p.H + 2 * p.pad - wd * (p.kh - 1) - 1) / p.stride + 1 || p.Wo != (p.W + 2 * p.pad - wd * (p.kw - 1) - 1) / p.stride + 1)
        return CX_ESHAPE;
    } else if (p.mode == CX_MODE_POOL2) {
      if ((p.H & 1) || (p.W & 1) || p.Ho != p.H / 2 || p.Wo != p.W / 2 || p.kh != 1 || p.kw != 1) return CX_ESHAPE;
    }
    return cx_conv_wgrad_f32(p, as_stream(stream));
  }
  if (p.dtype != CX_DT_BF16) return CX_EINVAL;
  if (p.K <= 0 || p.N <= 0 || (p.K % 8) || (p.N % 8)) return CX_ESHAPE;
  if ((long long)p.B * p.Ho * p.Wo >= (1ll << 31)) return CX_ESHAPE;
  if (p.mode == CX_MODE_STEM ? (p.ldx != 4) : (p.ldx % 8 != 0)) return CX_EALIGN;
  if ((p.ldg % 8) || !aligned16(p.g) || !aligned16(p.x)) return CX_EALIGN;
  if (p.g_prologue == CX_PRO_AFFINE2 && (!p.g2 || !p.ga || !p.gb || !p.gc || (p.ldg2 % 8) || !aligned16(p.g2))) return CX_EINVAL;
  if (p.x_prologue == CX_PRO_AFFINE_RELU && (!p.pa || !p.pb)) return CX_EINVAL;
  hipStream_t st = as_stream(stream);
  const bool g2 = p.g_prologue == CX_PRO_AFFINE2;
  if (p.g_prologue != CX_PRO_NONE && !g2) return CX_EUNSUPPORTED;
  if (p.mode == CX_MODE_CONV) {
    if (p.kh <= 0 || p.kw <= 0 || p.stride <= 0 || p.pad < 0 || p.dil < 0) return CX_ESHAPE;
    const int wd = p.dil > 1 ? p.dil : 1;
    if (p.Ho != (p.H + 2 * p.pad - wd * (p.kh - 1) - 1) / p.stride + 1 || p.Wo != (p.W + 2 * p.pad - wd * (p.kw - 1) - 1) / p.stride + 1)
      return CX_ESHAPE;
    if (wd == 1) {             // (a dilated convolution's weight gradient runs on the generic tile kernel below)
      bool handled = false;
      int rc = cx_try_pc_wgrad(p, st, &handled);
      if (handled) return rc;
      rc = cx_try_ring_wgrad(p, st, &handled);
      if (handled) return rc;
      rc = cx_try_wgrad_mm(p, st, &handled);             // wide 1x1 and 3x3 (N % 128 == 0); the dense layers' N = 32 passes through
      if (handled) return rc;
      rc = cx_try_strip_wgrad(p, st, &handled);
      if (handled) return rc;
    }
    const long long pw_min = 1ll << 23;              // ResNet152 1x1 layers: 2^23 measured 1 % faster end to end than 2^25
    // the dense-layer bottleneck with enough work to fill the chip with 512-thread workgroups (measured crossover against
    // the generic kernel: 102 k pixels x 512 channels)
    if (p.kh == 1 && p.kw == 1 && p.stride == 1 && p.pad == 0 && p.N % 128 == 0 && p.K >= 64 &&
        (long long)p.B * p.Ho * p.Wo * p.K >= pw_min) {
      if (p.x_prologue == CX_PRO_AFFINE_RELU)
        return g2 ? launch_pw_wgrad<CX_PRO_AFFINE2, CX_PRO_AFFINE_RELU>(p, st) : launch_pw_wgrad<CX_PRO_NONE, CX_PRO_AFFINE_RELU>(p, st);
      if (p.x_prologue == CX_PRO_NONE)
        return g2 ? launch_pw_wgrad<CX_PRO_AFFINE2, CX_PRO_NONE>(p, st) : launch_pw_wgrad<CX_PRO_NONE, CX_PRO_NONE>(p, st);
    }
    if (p.x_prologue == CX_PRO_AFFINE_RELU)
      return g2 ? launch_tile<CX_PRO_AFFINE2, CX_PRO_AFFINE_RELU, CX_MODE_CONV>(p, st)
                : launch_tile<CX_PRO_NONE, CX_PRO_AFFINE_RELU, CX_MODE_CONV>(p, st);
    if (p.x_prologue == CX_PRO_NONE)
      return g2 ? launch_tile<CX_PRO_AFFINE2, CX_PRO_NONE, CX_MODE_CONV>(p, st)
                : launch_tile<CX_PRO_NONE, CX_PRO_NONE, CX_MODE_CONV>(p, st);
    return CX_EUNSUPPORTED;
  }
  if (p.mode == CX_MODE_POOL2) {
    if (p.x_prologue != CX_PRO_AFFINE_RELU) return CX_EUNSUPPORTED;
    if ((p.H & 1) || (p.W & 1) || p.Ho != p.H / 2 || p.Wo != p.W / 2) return CX_ESHAPE;
    return g2 ? launch_tile<CX_PRO_AFFINE2, CX_PRO_AFFINE_RELU, CX_MODE_POOL2>(p, st)
              : launch_tile<CX_PRO_NONE, CX_PRO_AFFINE_RELU, CX_MODE_POOL2>(p, st);
  }
  if (p.mode == CX_MODE_STEM) {
    if (p.x_prologue != CX_PRO_NONE || p.K != 32 || p.N != 64) return CX_EUNSUPPORTED;
    return g2 ? launch_stem<CX_PRO_AFFINE2>(p, st) : launch_stem<CX_PRO_NONE>(p, st);
  }
  return CX_EUNSUPPORTED;
}
