// Implicit-GEMM convolution (forward and input-gradient) for NHWC bf16 on gfx950.
//
//   Y[m][n] = sum_{tap,c} A(m, tap, c) * W[tap][n][c]        m = (b, oy, ox) flattened
//
// A is never stored: it is produced in the staging pass from the raw dense-block buffer by the fused
// prologue (BN scale/shift + ReLU, 2x2 average, or the two-tensor affine form of BN backward).
// Tile: 128 output pixels x BN channels x 32 input channels per step, 4 waves, MFMA 32x32x16 bf16 with
// fp32 accumulation; LDS double buffered (one barrier per step), global loads for step s+1 are in
// flight while step s is multiplied.  LDS rows are padded to 80 B so the ds_read_b128 fragment reads
// are bank-conflict free (MI355X_MICROARCH.md LDS table).  The epilogue stages the fp32 tile through
// LDS so that global stores / mask loads are 16 B per lane along the channel axis.
#include "common.h"

namespace {

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int PITCH = 80;              // bytes per LDS row: 32 bf16 + 16 B pad
constexpr int A_BYTES = BM * PITCH;

template <int BN>
struct Geo {
  static constexpr int WAVES_N = (BN >= 64) ? 2 : 1;
  static constexpr int WAVES_M = 4 / WAVES_N;
  static constexpr int TM = BM / (WAVES_M * 32);
  static constexpr int TN = BN / (WAVES_N * 32);
  static constexpr int B_BYTES = BN * PITCH;
  static constexpr int STAGE = A_BYTES + B_BYTES;
  static constexpr int EPITCH = BN + 4;                       // floats per epilogue row
  static constexpr int EPI_BYTES = 64 * EPITCH * 4;
  static constexpr int MAIN_BYTES = (2 * STAGE > EPI_BYTES) ? 2 * STAGE : EPI_BYTES;
};

template <int PRO>
struct NCoef {
  static constexpr int v = (PRO == CX_PRO_NONE) ? 0 : (PRO == CX_PRO_AFFINE_RELU ? 2 : 3);
};

template <int BN, int PRO, int MODE, int EPI>
__global__ __launch_bounds__(256) void conv_gemm_kernel(const CxConv p, const int M, const int n_tiles) {
  using G = Geo<BN>;
  constexpr int NSRC = (MODE == CX_MODE_POOL2) ? 4 : 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* coef = reinterpret_cast<float*>(smem);                       // [NCoef][K]
  char* tiles = smem + NCoef<PRO>::v * p.K * 4;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / G::WAVES_N, wn = wave % G::WAVES_N;
  const int wgid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = wgid / n_tiles, nt = wgid % n_tiles;
  const int n0 = nt * BN;

  // ---- coefficient table + stat zeroing
  if (PRO != CX_PRO_NONE) {
    for (int i = tid; i < p.K; i += 256) {
      coef[i] = p.pa[i];
      coef[p.K + i] = p.pb[i];
      if (PRO == CX_PRO_AFFINE2) coef[2 * p.K + i] = p.pc[i];
    }
  }

  // ---- per-thread A rows
  const int qa = tid & 3;                 // 8-channel chunk inside the 32-wide K step
  int rb[2], riy[2], rix[2];
  bool rvalid[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = mt * BM + (tid >> 2) + 64 * i;
    rvalid[i] = m < M;
    const int mm = rvalid[i] ? m : 0;
    const int hw = p.Ho * p.Wo;
    rb[i] = mm / hw;
    const int rem = mm - rb[i] * hw;
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    if (MODE == CX_MODE_CONV) {
      riy[i] = oy * p.stride - p.pad;
      rix[i] = ox * p.stride - p.pad;
    } else if (MODE == CX_MODE_POOL2) {
      riy[i] = 2 * oy;
      rix[i] = 2 * ox;
    } else {                              // STEM: row taps ky, 8 pixels starting at 2*ox-4
      riy[i] = 2 * oy - 3;
      rix[i] = 2 * ox - 4 + 2 * qa;
    }
  }
  const int kpt = (MODE == CX_MODE_STEM) ? 1 : (p.K + BK - 1) / BK;         // K steps per tap (last one may be partial: K % 8 == 0)
  const int taps = (MODE == CX_MODE_STEM) ? 7 : p.kh * p.kw;
  const int nsteps = taps * kpt;
  const bf16* __restrict__ X = reinterpret_cast<const bf16*>(p.x);
  const bf16* __restrict__ X2 = reinterpret_cast<const bf16*>(p.x2);
  const bf16* __restrict__ Wp = reinterpret_cast<const bf16*>(p.w);

  uint4 ra[2][NSRC], ra2[2], rbw[2];
  bool av[2], wv[2] = {false, false};

  // NOTE: every global load below is UNCONDITIONAL (invalid rows read a clamped, in-bounds address and are
  // zeroed when staged).  A branch around a load makes hipcc wait vmcnt(0) per load (cdna_hip_programming.md,
  // "Three .s-level traps" (c)), which serialises the whole prefetch.
  // Steps are requested in order: the tap / channel-step position advances incrementally (no per-step integer divisions), and a
  // transposed stride is 1 or 2 (shift and mask).
  const int ts_sh = p.tstride > 1 ? 1 : 0;
  const int cdil = p.dil > 1 ? p.dil : 1;
  int q_tap = 0, q_kc = 0, q_dy = 0, q_dx = 0, w_kc = 0;
  auto issue_loads = [&](int s) {
    (void)s;
    const int tap = q_tap, kc = q_kc;
    const int dy = (MODE == CX_MODE_CONV) ? q_dy * cdil : tap;       // cdil: taps `dil` pixels apart (1 unless CxConv.dil > 1)
    const int dx = (MODE == CX_MODE_CONV) ? q_dx * cdil : 0;
    w_kc = kc;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (MODE == CX_MODE_STEM) {
        const int iy = riy[i] + dy, ix = rix[i];
        av[i] = rvalid[i] & (iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.W);
        const int cy = av[i] ? iy : 0, cx = av[i] ? ix : 0;
        ra[i][0] = *reinterpret_cast<const uint4*>(X + ((size_t)(rb[i] * p.H + cy) * p.W + cx) * 4);
      } else if (MODE == CX_MODE_POOL2) {
        av[i] = rvalid[i];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const size_t pix = (size_t)(rb[i] * p.H + riy[i] + (a >> 1)) * p.W + rix[i] + (a & 1);
          ra[i][a] = *reinterpret_cast<const uint4*>(X + pix * p.ldx + kc * BK + qa * 8);
        }
      } else {
        int iy = riy[i] + dy, ix = rix[i] + dx;
        // input gradient of a stride-2 conv: only source positions on the stride grid exist
        bool ok = rvalid[i] & (iy >= 0) & (ix >= 0) & (((iy | ix) & ts_sh) == 0);
        iy >>= ts_sh;
        ix >>= ts_sh;
        const bool kok = kc * BK + qa * 8 < p.K;                  // partial last K step
        av[i] = ok & (iy < p.H) & (ix < p.W) & kok;
        const int cy = av[i] ? iy : 0, cx = av[i] ? ix : 0;
        const int ck = kok ? kc * BK + qa * 8 : 0;
        const size_t pix = (size_t)(rb[i] * p.H + cy) * p.W + cx;
        ra[i][0] = *reinterpret_cast<const uint4*>(X + pix * p.ldx + ck);
        if (PRO == CX_PRO_AFFINE2) ra2[i] = *reinterpret_cast<const uint4*>(X2 + pix * p.ldx2 + ck);
      }
    }
    // weights: rows n = tid>>2 (+64), chunk qa
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int nb = (tid >> 2) + 64 * i;
      if (nb < BN) {                       // compile-time per (i, BN) except BN=32 (uniform per wave)
        const int n = n0 + nb;
        const bool kok = (MODE == CX_MODE_STEM) || (kc * BK + qa * 8 < p.K);
        wv[i] = n < p.N && kok;
        const int nc = wv[i] ? n : 0;
        const size_t off = ((size_t)tap * p.N + nc) * (size_t)(MODE == CX_MODE_STEM ? BK : p.K) + (kok ? kc * BK + qa * 8 : 0);
        rbw[i] = *reinterpret_cast<const uint4*>(Wp + off);
      }
    }
    if (++q_kc == kpt) {
      q_kc = 0;
      ++q_tap;
      if (++q_dx == p.kw) {
        q_dx = 0;
        ++q_dy;
      }
    }
  };

  auto write_stage = [&](int s, int buf) {
    (void)s;
    const int tap = 0, kc = w_kc;             // the step requested last
    char* A = tiles + buf * G::STAGE;
    char* Bt = A + A_BYTES;
    const int c0 = kc * BK + qa * 8;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      U128 o;
      if (!av[i]) {
        o.u = make_uint4(0, 0, 0, 0);
      } else if (PRO == CX_PRO_NONE) {
        o.u = ra[i][0];
      } else if (PRO == CX_PRO_AFFINE_RELU) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int a = 0; a < NSRC; ++a) {
          const uint32_t w4[4] = {ra[i][a].x, ra[i][a].y, ra[i][a].z, ra[i][a].w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            acc[2 * j] += fmaxf(fmaf(cx_bf_lo(w4[j]), coef[c0 + 2 * j], coef[p.K + c0 + 2 * j]), 0.f);
            acc[2 * j + 1] += fmaxf(fmaf(cx_bf_hi(w4[j]), coef[c0 + 2 * j + 1], coef[p.K + c0 + 2 * j + 1]), 0.f);
          }
        }
        {
          const float sc_ = NSRC == 4 ? 0.25f : 1.f;
          o.u = make_uint4(cx_packbf(acc[0] * sc_, acc[1] * sc_), cx_packbf(acc[2] * sc_, acc[3] * sc_), cx_packbf(acc[4] * sc_, acc[5] * sc_),
                           cx_packbf(acc[6] * sc_, acc[7] * sc_));
        }
      } else {
        o.u = cx_affine2_8(ra[i][0], ra2[i], coef + c0, coef + p.K + c0, coef + 2 * p.K + c0);
      }
      *reinterpret_cast<uint4*>(A + ((tid >> 2) + 64 * i) * PITCH + qa * 16) = o.u;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int nb = (tid >> 2) + 64 * i;
      if (nb < BN) *reinterpret_cast<uint4*>(Bt + nb * PITCH + qa * 16) = wv[i] ? rbw[i] : make_uint4(0, 0, 0, 0);
    }
    (void)tap;
  };

  f32x16 acc[G::TM][G::TN];
#pragma unroll
  for (int i = 0; i < G::TM; ++i)
#pragma unroll
    for (int j = 0; j < G::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  issue_loads(0);
  __syncthreads();                       // coefficient table visible
  write_stage(0, 0);
  __syncthreads();

  const int lrow = lane & 31, lh = lane >> 5;
  for (int s = 0; s < nsteps; ++s) {
    const int buf = s & 1;
    if (s + 1 < nsteps) issue_loads(s + 1);
    const char* A = tiles + buf * G::STAGE;
    const char* Bt = A + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 af[G::TM], bfr[G::TN];
#pragma unroll
      for (int i = 0; i < G::TM; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(A + ((wm * G::TM + i) * 32 + lrow) * PITCH + kk * 32 + lh * 16);
#pragma unroll
      for (int j = 0; j < G::TN; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8*>(Bt + ((wn * G::TN + j) * 32 + lrow) * PITCH + kk * 32 + lh * 16);
#pragma unroll
      for (int i = 0; i < G::TM; ++i)
#pragma unroll
        for (int j = 0; j < G::TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (s + 1 < nsteps) write_stage(s + 1, buf ^ 1);
    __syncthreads();
  }

  // ---------------------------------------------------------------- epilogue (two 64-row halves)
  constexpr int CPR = BN / 8;              // 16-byte chunks per row
  constexpr int RPP = 256 / CPR;           // rows per pass
  const int cq = tid % CPR, rr = tid / CPR;
  const int nch = n0 + cq * 8;             // first channel of this thread's chunk
  const bool nvalid = nch < p.N;
  float* etile = reinterpret_cast<float*>(tiles);
  bf16* __restrict__ Y = reinterpret_cast<bf16*>(p.y);
  const bf16* __restrict__ EX = reinterpret_cast<const bf16*>(p.ex);
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
  float esc[8], esh[8], emu[8], er[8], escale[8];
  if (EPI == CX_EPI_MASK && nvalid) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      esc[j] = p.e_sc[nch + j];
      esh[j] = p.e_sh[nch + j];
      emu[j] = p.e_mu[nch + j];
      er[j] = p.e_r[nch + j];
      escale[j] = p.e_scale[nch + j];
    }
  }
  const bool want_stats = p.stat_sum != nullptr;

  // Every mask / read-modify-write operand of the tile is requested up front: the passes below would otherwise each
  // pay one full HBM round trip (load -> use -> store, 8 times per tile), which is what bounds the 1x1 input-gradient
  // kernels.  Loads are unconditional on clamped in-bounds addresses.
  constexpr int NPASS = 64 / RPP;
  U128 xv[2][NPASS], old[2][NPASS];
  {
    const int ncl = nvalid ? nch : 0;
    if (EPI == CX_EPI_MASK) {
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
          const int m = mt * BM + half * 64 + pass * RPP + rr;
          const int mc = m < M ? m : M - 1;
          xv[half][pass].u = *reinterpret_cast<const uint4*>(EX + (size_t)mc * p.ldex + ncl);
        }
    }
    if (p.accumulate) {
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) {
          const int m = mt * BM + half * 64 + pass * RPP + rr;
          const int mc = m < M ? m : M - 1;
          old[half][pass].u = *reinterpret_cast<const uint4*>(Y + (size_t)mc * p.ldy + ncl);
        }
    } else {
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int pass = 0; pass < NPASS; ++pass) old[half][pass].u = make_uint4(0, 0, 0, 0);
    }
  }

#pragma unroll
  for (int half = 0; half < 2; ++half) {
    if (wm / (G::WAVES_M / 2) == half) {
      const int wml = wm % (G::WAVES_M / 2);
#pragma unroll
      for (int i = 0; i < G::TM; ++i)
#pragma unroll
        for (int j = 0; j < G::TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (wml * G::TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int col = (wn * G::TN + j) * 32 + lrow;
            etile[row * G::EPITCH + col] = acc[i][j][r];
          }
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      const int row = pass * RPP + rr;
      const int m = mt * BM + half * 64 + row;
      if (m < M && nvalid) {
        const float4 v0 = *reinterpret_cast<const float4*>(etile + row * G::EPITCH + cq * 8);
        const float4 v1 = *reinterpret_cast<const float4*>(etile + row * G::EPITCH + cq * 8 + 4);
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        U128 o;
        const uint32_t ow[4] = {old[half][pass].u.x, old[half][pass].u.y, old[half][pass].u.z, old[half][pass].u.w};
        if (EPI == CX_EPI_STORE) {
          float t[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) { t[2 * j] = v[2 * j] + cx_bf_lo(ow[j]); t[2 * j + 1] = v[2 * j + 1] + cx_bf_hi(ow[j]); }
          o.u = cx_pack8_stats(t, true, true, s1, s2);
        } else {
          const uint32_t xw[4] = {xv[half][pass].u.x, xv[half][pass].u.y, xv[half][pass].u.z, xv[half][pass].u.w};
          uint32_t w4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float xl = cx_bf_lo(xw[j]), xu = cx_bf_hi(xw[j]);
            const float dl = (fmaf(xl, esc[2 * j], esh[2 * j]) > 0.f) ? v[2 * j] : 0.f;
            const float du = (fmaf(xu, esc[2 * j + 1], esh[2 * j + 1]) > 0.f) ? v[2 * j + 1] : 0.f;
            s1[2 * j] += dl;
            s1[2 * j + 1] += du;
            s2[2 * j] += dl * (xl - emu[2 * j]) * er[2 * j];
            s2[2 * j + 1] += du * (xu - emu[2 * j + 1]) * er[2 * j + 1];
            w4[j] = cx_packbf(fmaf(escale[2 * j], dl, cx_bf_lo(ow[j])), fmaf(escale[2 * j + 1], du, cx_bf_hi(ow[j])));
          }
          o.u = make_uint4(w4[0], w4[1], w4[2], w4[3]);
        }
        *reinterpret_cast<uint4*>(Y + (size_t)m * p.ldy + nch) = o.u;
      }
    }
    __syncthreads();
  }

  if (want_stats) {
    // lanes sharing a channel chunk inside a wave are CPR apart
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int d = CPR; d < 64; d <<= 1) {
        s1[j] += __shfl_xor(s1[j], d);
        s2[j] += __shfl_xor(s2[j], d);
      }
    }
    float* scratch = reinterpret_cast<float*>(tiles);            // the tile buffers are free now
    wg_stat_begin<4>(scratch, BN, tid, 256);
    if (lane < CPR) {
#pragma unroll
      for (int j = 0; j < 8; ++j) wg_stat_put(scratch, BN, wave, cq * 8 + j, s1[j], s2[j]);
    }
    // row = pixel tile: the n tiles of one pixel tile write disjoint channels of the same row
    wg_stat_end<4>(scratch, BN, tid, 256, p.stat_sum, p.stat_sq, p.stat_det, p.stat_det ? mt : (int)blockIdx.x, p.stat_replicas,
                   p.stat_rstride, n0, p.N);
  }
}

template <int BN, int PRO, int MODE, int EPI>
int launch(const CxConv& p, hipStream_t st) {
  using G = Geo<BN>;
  const long long M = (long long)p.B * p.Ho * p.Wo;
  const int m_tiles = (int)((M + BM - 1) / BM);
  const int n_tiles = (p.N + BN - 1) / BN;
  const size_t smem = (size_t)NCoef<PRO>::v * p.K * 4 + G::MAIN_BYTES + 2 * BN * 4;
  if (smem > 160 * 1024) return CX_ESHAPE;
  if (const int e = stat_rows_check(p, m_tiles)) return e;
  static bool attr_set = false;
  if (!attr_set && smem > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_gemm_kernel<BN, PRO, MODE, EPI>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  CX_KTAG("conv_gemm_kernel<%d, %d, %d, %d>", BN, PRO, MODE, EPI);
  hipLaunchKernelGGL((conv_gemm_kernel<BN, PRO, MODE, EPI>), dim3(m_tiles * n_tiles), dim3(256), smem, st, p, (int)M,
                     n_tiles);
  return launch_status();
}

template <int PRO, int MODE, int EPI>
int launch_bn(const CxConv& p, hipStream_t st) {
  if (p.N % 128 == 0) return launch<128, PRO, MODE, EPI>(p, st);
  if (p.N == 32) return launch<32, PRO, MODE, EPI>(p, st);
  return launch<64, PRO, MODE, EPI>(p, st);
}

}  // namespace

int cx_try_pc_fwd(const CxConv& p, hipStream_t st, bool* handled);        // conv3x3_pc.hip
int cx_try_ring_fwd(const CxConv& p, hipStream_t st, bool* handled);      // conv3x3_ring.hip
int cx_try_strip_fwd(const CxConv& p, hipStream_t st, bool* handled);     // conv3x3_strip.hip
int cx_try_ring_dgrad(const CxConv& p, hipStream_t st, bool* handled);    // conv3x3_ring.hip
int cx_try_strip_dgrad(const CxConv& p, hipStream_t st, bool* handled);
int cx_try_pw_dgrad(const CxConv& p, hipStream_t st, bool* handled);        // conv1x1_dgrad.hip
int cx_try_pw_fwd(const CxConv& p, hipStream_t st, bool* handled);          // conv1x1_fwd.hip
int cx_try_pw_fwdk(const CxConv& p, hipStream_t st, bool* handled);         // conv1x1_fwdk.hip
int cx_try_pw_xs(const CxConv& p, hipStream_t st, bool* handled);           // conv1x1_xs.hip
int cx_try_stem_fwd(const CxConv& p, hipStream_t st, bool* handled);        // conv_stem.hip
int cx_conv_gemm_f32(const CxConv& p, hipStream_t st);                      // conv_f32.hip
int cx_try_conv_mm(const CxConv& p, hipStream_t st, bool* handled);         // conv_mm.hip

thread_local int cx_tl_stat_rows = 0;
thread_local int cx_tl_pro_out = 0;
thread_local char cx_tl_kernel[112] = "";
extern "C" const char* cx_last_kernel(void) { return cx_tl_kernel; }
extern "C" int cx_last_stat_rows(void) { return cx_tl_stat_rows; }
extern "C" int cx_last_pro_out(void) { return cx_tl_pro_out; }

static inline int dil_extent(int k, int dil) { return (dil > 1 ? dil : 1) * (k - 1) + 1; }

extern "C" int cx_conv_gemm(const CxConv* pp, void* stream) {
  if (!pp) return CX_EINVAL;
  const CxConv& p = *pp;
  cx_tl_pro_out = 0;
  if (!p.x || !p.w || !p.y) return CX_EINVAL;
  if (p.B <= 0 || p.H <= 0 || p.W <= 0 || p.Ho <= 0 || p.Wo <= 0) return CX_ESHAPE;
  if (p.dtype == CX_DT_F32) {             // fp32 storage mode: one generic kernel family (conv_f32.hip)
    if (p.K <= 0 || p.N <= 0) return CX_ESHAPE;
    if (!aligned16(p.x) || !aligned16(p.y) || !aligned16(p.w)) return CX_EALIGN;
    if ((long long)p.B * p.Ho * p.Wo >= (1ll << 31)) return CX_ESHAPE;
    if (p.prologue != CX_PRO_NONE && (!p.pa || !p.pb)) return CX_EINVAL;
    if (p.prologue == CX_PRO_AFFINE2 && (!p.x2 || !p.pc)) return CX_EINVAL;
    if (p.epilogue == CX_EPI_MASK && (!p.ex || !p.e_sc || !p.e_sh || !p.e_mu || !p.e_r || !p.e_scale || !p.stat_sum || !p.stat_sq))
      return CX_EINVAL;
    if ((p.stat_sum == nullptr) != (p.stat_sq == nullptr)) return CX_EINVAL;
    if (p.mode == CX_MODE_CONV) {
      if (p.kh <= 0 || p.kw <= 0 || p.stride <= 0 || p.pad < 0 || p.ldx < p.K || p.dil < 0) return CX_ESHAPE;
      const int keh = dil_extent(p.kh, p.dil), kew = dil_extent(p.kw, p.dil);
      if (p.tstride > 1) {
        const int fpad = keh - 1 - p.pad;
        if (p.stride != 1 || fpad < 0) return CX_ESHAPE;
        if (p.H != (p.Ho + 2 * fpad - keh) / p.tstride + 1 || p.W != (p.Wo + 2 * fpad - kew) / p.tstride + 1) return CX_ESHAPE;
      } else if (p.Ho != (p.H + 2 * p.pad - keh) / p.stride + 1 || p.Wo != (p.W + 2 * p.pad - kew) / p.stride + 1) {
        return CX_ESHAPE;
      }
    } else if (p.mode == CX_MODE_POOL2) {
      if ((p.H & 1) || (p.W & 1) || p.Ho != p.H / 2 || p.Wo != p.W / 2 || p.ldx < p.K) return CX_ESHAPE;
    }
    return cx_conv_gemm_f32(p, as_stream(stream));
  }
  if (p.dtype != CX_DT_BF16) return CX_EINVAL;
  if (p.K <= 0 || p.N <= 0 || (p.K % 8) || (p.N % 8)) return CX_ESHAPE;
  if (p.mode == CX_MODE_POOL2 && (p.K % 32)) return CX_ESHAPE;
  if (p.mode == CX_MODE_STEM ? (p.ldx != 4) : (p.ldx % 8 != 0)) return CX_EALIGN;
  if ((p.ldy % 8) || !aligned16(p.x) || !aligned16(p.y) || !aligned16(p.w)) return CX_EALIGN;
  if ((long long)p.B * p.Ho * p.Wo >= (1ll << 31)) return CX_ESHAPE;
  if (p.prologue != CX_PRO_NONE && (!p.pa || !p.pb)) return CX_EINVAL;
  if (p.prologue == CX_PRO_AFFINE2 && (!p.x2 || !p.pc || (p.ldx2 % 8) || !aligned16(p.x2))) return CX_EINVAL;
  if (p.epilogue == CX_EPI_MASK) {
    if (!p.ex || !p.e_sc || !p.e_sh || !p.e_mu || !p.e_r || !p.e_scale || !p.stat_sum || !p.stat_sq) return CX_EINVAL;
    if ((p.ldex % 8) || !aligned16(p.ex)) return CX_EALIGN;
  }
  if (p.epilogue == CX_EPI_JOIN) {
    if (!p.ex || !p.emask || !p.e_mu || !p.e_r || !p.stat_sum || !p.stat_sq || !p.accumulate) return CX_EINVAL;
    if ((p.ldex % 8) || !aligned16(p.ex)) return CX_EALIGN;
  }
  if ((p.stat_sum == nullptr) != (p.stat_sq == nullptr)) return CX_EINVAL;
  if (p.stat_replicas < 0 || (p.stat_replicas > 1 && p.stat_rstride < p.N)) return CX_EINVAL;
  hipStream_t st = as_stream(stream);
  if (p.prologue == CX_PRO_JOIN) {
    // the residual join of the block below in the prologue of a 1x1 stride-1 convolution: conv_mm.hip has the only implementation
    if (!p.x2 || !p.pc || !p.pro_out) return CX_EINVAL;
    if ((p.ldx2 % 8) || (p.ldpo % 8) || !aligned16(p.x2) || !aligned16(p.pro_out) || (((uintptr_t)p.po_lo) & 7) || (p.x3 && (((uintptr_t)p.x3) & 7)))
      return CX_EALIGN;
    if (p.mode != CX_MODE_CONV || p.epilogue != CX_EPI_STORE || p.kh != 1 || p.kw != 1 || p.stride != 1 || p.pad != 0 || p.tstride > 1 ||
        p.accumulate || (p.K % 64) || p.Ho != p.H || p.Wo != p.W || p.ldx < p.K || p.ldx2 < p.K || p.ldpo < p.K)
      return CX_EUNSUPPORTED;
    bool handled = false;
    const int rc = cx_try_conv_mm(p, st, &handled);
    return handled ? rc : CX_EUNSUPPORTED;
  }
  if (p.mode == CX_MODE_CONV) {
    if (p.kh <= 0 || p.kw <= 0 || p.stride <= 0 || p.pad < 0 || p.dil < 0) return CX_ESHAPE;
    if (p.tstride > 2) return CX_EUNSUPPORTED;        // the reference's strides are 1 and 2
    const int keh = dil_extent(p.kh, p.dil), kew = dil_extent(p.kw, p.dil);
    if (p.tstride > 1) {
      // (B,H,W) is the strided conv's OUTPUT gradient, (Ho,Wo) its input: H = (Ho + 2*fwd_pad - kh)/tstride + 1 with
      // pad = kh-1-fwd_pad; stride of the implicit GEMM itself is 1
      const int fpad = keh - 1 - p.pad;
      if (p.stride != 1 || fpad < 0) return CX_ESHAPE;
      if (p.H != (p.Ho + 2 * fpad - keh) / p.tstride + 1 || p.W != (p.Wo + 2 * fpad - kew) / p.tstride + 1) return CX_ESHAPE;
    } else if (p.Ho != (p.H + 2 * p.pad - keh) / p.stride + 1 || p.Wo != (p.W + 2 * p.pad - kew) / p.stride + 1) {
      return CX_ESHAPE;
    }
    if (p.ldx < p.K) return CX_ESHAPE;
    if (p.dil <= 1) {          // (a dilated convolution runs on the generic implicit GEMM below: the tiled kernels assume adjacent taps)
      bool handled = false;
      int rc = cx_try_pc_fwd(p, st, &handled);
      if (handled) return rc;
      rc = cx_try_ring_fwd(p, st, &handled);
      if (handled) return rc;
      rc = cx_try_strip_fwd(p, st, &handled);
      if (handled) return rc;
      rc = cx_try_ring_dgrad(p, st, &handled);
      if (handled) return rc;
      rc = cx_try_strip_dgrad(p, st, &handled);
      if (handled) return rc;
      rc = cx_try_pw_dgrad(p, st, &handled);
      if (handled) return rc;
      rc = cx_try_pw_fwd(p, st, &handled);
      if (handled) return rc;
      rc = cx_try_pw_fwdk(p, st, &handled);
      if (handled) return rc;
      rc = cx_try_pw_xs(p, st, &handled);
      if (handled) return rc;
      rc = cx_try_conv_mm(p, st, &handled);
      if (handled) return rc;
    }
    if (p.epilogue == CX_EPI_STORE) {
      if (p.prologue == CX_PRO_NONE) return launch_bn<CX_PRO_NONE, CX_MODE_CONV, CX_EPI_STORE>(p, st);
      if (p.prologue == CX_PRO_AFFINE_RELU) return launch_bn<CX_PRO_AFFINE_RELU, CX_MODE_CONV, CX_EPI_STORE>(p, st);
      if (p.prologue == CX_PRO_AFFINE2) return launch_bn<CX_PRO_AFFINE2, CX_MODE_CONV, CX_EPI_STORE>(p, st);
    } else if (p.epilogue == CX_EPI_MASK) {
      if (p.prologue == CX_PRO_AFFINE2) return launch_bn<CX_PRO_AFFINE2, CX_MODE_CONV, CX_EPI_MASK>(p, st);
      if (p.prologue == CX_PRO_NONE) return launch_bn<CX_PRO_NONE, CX_MODE_CONV, CX_EPI_MASK>(p, st);
    }
    return CX_EUNSUPPORTED;
  }
  if (p.mode == CX_MODE_POOL2) {
    if (p.prologue != CX_PRO_AFFINE_RELU || p.epilogue != CX_EPI_STORE) return CX_EUNSUPPORTED;
    if ((p.H & 1) || (p.W & 1) || p.Ho != p.H / 2 || p.Wo != p.W / 2 || p.ldx < p.K) return CX_ESHAPE;
    return launch_bn<CX_PRO_AFFINE_RELU, CX_MODE_POOL2, CX_EPI_STORE>(p, st);
  }
  if (p.mode == CX_MODE_STEM) {
    if (p.prologue != CX_PRO_NONE || p.epilogue != CX_EPI_STORE) return CX_EUNSUPPORTED;
    if (p.K != 32 || (p.W & 1) || p.Ho != (p.H + 6 - 7) / 2 + 1 || p.Wo != (p.W + 6 - 7) / 2 + 1) return CX_ESHAPE;
    {
      bool handled = false;
      const int rc = cx_try_stem_fwd(p, st, &handled);
      if (handled) return rc;
    }
    return launch_bn<CX_PRO_NONE, CX_MODE_STEM, CX_EPI_STORE>(p, st);
  }
  return CX_EUNSUPPORTED;
}
