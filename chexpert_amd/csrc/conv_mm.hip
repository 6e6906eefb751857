// Implicit-GEMM convolution for the MFMA-heavy shapes (ResNet bottlenecks, EfficientNet / AAConv 1x1 with wide channels):
// K >= 64 input channels per tap (a multiple of 8; the last 64-channel step may be partial), N % 128 == 0 output channels, NHWC bf16,
// any kernel size / stride, transposed stride 1 or 2.
//
//   Y[m][n] = sum_{tap,c} A(m, tap, c) * W[tap][n][c]        m = (b, oy, ox) flattened
//
// What differs from conv_gemm.hip (which stays the path for narrow / ragged channel counts): these shapes are bound by the
// vector-memory path of a CU (L2 -> L1 -> registers, ~56-64 B/clk/CU), not by HBM, and a k-step of the 128x128x32 kernel
// is one L2 round trip long (loads requested at the top of a step are staged at its bottom: bytes in flight per CU ~ 48 KB).
// Here: 64 input channels per step (half the barriers), 128 x 128, 256 x 128 or 128 x 256 output tiles, and the operands of step
// s+2 are requested while step s multiplies - two register sets, so two steps are always in flight.  Measured with s_memtime
// stamps, a step is bound by vector-instruction issue (two waves per SIMD: ~8 cycles per instruction per wave), not by MFMA or
// bandwidth, so the instruction streams are kept short: per staged row one bit-field extract (tap validity, precomputed per
// tile), one add of the step's scalar byte offset and one AND; 32-bit offsets against scalar bases; ReLU on packed bf16.
// Prologues (BN + ReLU, two-tensor BN-backward form) are applied once per element on the way to LDS; epilogues (store /
// accumulate / ReLU-mask + BN-backward sums, statistic rows) are those of conv_gemm.hip.
#include <cstdlib>
#include <type_traits>
#include "common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));   // (HIP's uint4 is a struct: copies of it become memcpys that pin register sets to scratch)

#ifndef CX_MM_INTERLEAVE
#define CX_MM_INTERLEAVE 1
#endif
constexpr bool kInterleave = CX_MM_INTERLEAVE;
constexpr int BK = 64;
constexpr int PITCH = 144;             // bytes per LDS row: 64 bf16 + 16 B pad (36 dwords: ds_read_b128 fragment reads conflict-free)

// A workgroup is WMW x WNW waves, each wave owns a 64 x 64 output block: (2,2) = 128 x 128 with 256 threads (two workgroups per
// CU), (4,2) = 256 x 128 and (2,4) = 128 x 256 with 512 threads (one per CU).  The 128 x 256 form transforms each activation
// once per 256 output channels: the prologue arithmetic per MFMA halves, which is what bounds these kernels (vector issue).
template <int WMW, int WNW>
struct MG {
  static constexpr int BM = 64 * WMW;
  static constexpr int BN = 64 * WNW;
  static constexpr int NW = WMW * WNW;
  static constexpr int NT = 64 * NW;
  static constexpr int RSTEP = NT / 8;                  // rows covered by one staging pass (8 chunks of 16 B per row)
  static constexpr int NA = BM / RSTEP;                 // activation rows per thread per step
  static constexpr int NB = BN / RSTEP;                 // weight rows per thread per step
  static constexpr int A_BYTES = BM * PITCH;
  static constexpr int B_BYTES = BN * PITCH;
  static constexpr int STAGE = A_BYTES + B_BYTES;
  static constexpr int EPITCH = BN + 4;
  static constexpr int MAIN_BYTES = 2 * STAGE;          // > 64 * EPITCH * 4 (epilogue tile) and > NW * 2 * BN * 4 (statistics)
};

template <int PRO>
struct NCoef {
  static constexpr int v = (PRO == CX_PRO_NONE) ? 0 : ((PRO == CX_PRO_AFFINE_RELU || PRO == CX_PRO_JOIN) ? 2 : 3);   // JOIN: pa, pc
};

typedef short s16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t packbf(float a, float b) {      // one v_cvt_pk_bf16_f32 (RNE)
  union {
    bf16x2 h;
    uint32_t u;
  } o;
  o.h = __builtin_convertvector(f32x2{a, b}, bf16x2);
  return o.u;
}
// relu on a packed pair of bf16: as 16-bit integers the negative values (sign bit) are below zero
__device__ __forceinline__ uint32_t relu_pk(uint32_t v) {
  union {
    uint32_t u;
    s16x2 s;
  } a, r;
  a.u = v;
  r.s = __builtin_elementwise_max(a.s, s16x2{0, 0});
  return r.u;
}
// 16 B at base + 32-bit byte offset (the scalar-base form of global_load: no 64-bit vector arithmetic per load)
__device__ __forceinline__ u32x4 ld16(const char* base, uint32_t off) { return *reinterpret_cast<const u32x4*>(base + (size_t)off); }

// Geometry of one launch.  A plain convolution is one launch over all output pixels.  The input gradient of a stride-2
// convolution (CxConv.tstride == 2) is up to four launches, one per parity class (oy & 1, ox & 1) of the output pixels: inside a
// class every pixel has the same valid taps (1, 2, 2 or 4 of a 3x3 instead of 9 with three quarters of the products zero), they
// form a dense grid in the gradient image, and the rows of the implicit GEMM enumerate the class's pixels (b, oy', ox').
struct Cls {
  int nty, ntx;                   // taps walked: nty x ntx
  int i_mul, iy_add, ix_add;      // first tap's source pixel of row (oy', ox'): (oy' * i_mul + iy_add, ox' * i_mul + ix_add)
  int wy0, wx0, wstep;            // weight tap of walked tap (ty, tx): (wy0 + wstep * ty) * kw + wx0 + wstep * tx
  int Hq, Wq, Mq;                 // class image size, rows (B * Hq * Wq)
  int o_mul, oy_add, ox_add;      // output pixel of row (oy', ox'): (oy' * o_mul + oy_add, ox' * o_mul + ox_add)
  int row0;                       // first statistic row of the launch
  int rpt;                        // rows of the GEMM per workgroup tile, <= 128 (the tile's further rows are masked): a bandwidth-bound launch
                                  // whose 128-row tiles would fill the chip 1.56 times (400 tiles: the second round on 144 of 256 CUs)
                                  // runs as 512 tiles of 100 rows instead (cx_try_conv_mm)
};

// diagnostic build (DBG): s_memtime phase sums of the main loop per workgroup, waves 0 and NW-1: [issue, multiply, stage, barrier, steps]
__device__ unsigned long long conv_mm_stamps[2048 * 2 * 8];

template <int WMW, int WNW, int PRO, int EPI, bool DBG = false>
__global__ __launch_bounds__(64 * WMW * WNW) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_mm_kernel(const CxConv p, const Cls c,
                                                                                                          const int n_tiles) {
  const int M = c.Mq;
  using G = MG<WMW, WNW>;
  constexpr int BN = G::BN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int Kp = (p.K + BK - 1) / BK * BK;                            // K rounded up to whole steps (K % 8 == 0)
  float* coef = reinterpret_cast<float*>(smem);                       // [NCoef][Kp]
  char* tiles = smem + NCoef<PRO>::v * Kp * 4;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WNW, wn = wave % WNW;
  const int wgid = xcd_remap(blockIdx.x, gridDim.x);
  const int mt = wgid / n_tiles, nt = wgid - mt * n_tiles;
  const int n0 = nt * BN;

  if (PRO != CX_PRO_NONE) {
    for (int i = tid; i < Kp; i += G::NT) {
      const bool in = i < p.K;
      coef[i] = in ? p.pa[i] : 0.f;
      coef[Kp + i] = in ? (PRO == CX_PRO_JOIN ? p.pc[i] : p.pb[i]) : 0.f;
      if (PRO == CX_PRO_AFFINE2) coef[2 * Kp + i] = in ? p.pc[i] : 0.f;
    }
  }

  // ---- staging geometry: 16-byte chunk qa of rows r0 + RSTEP * i.  Per row: byte offset of its (tap 0, channel 0) element and
  // one validity bit per tap; inside the loop a row costs a bit-field extract, an add of the step's scalar offset and an AND.
  const int qa = tid & 7;
  const int r0 = tid >> 3;
  const int ntaps = c.nty * c.ntx;
  uint32_t roff[G::NA], roff2[G::NA], vbits[G::NA];
  uint32_t roff3[PRO == CX_PRO_JOIN ? G::NA : 1], roffp[PRO == CX_PRO_JOIN ? G::NA : 1];
#pragma unroll
  for (int i = 0; i < G::NA; ++i) {
    const int m = mt * c.rpt + r0 + G::RSTEP * i;
    const bool ok = m < M && r0 + G::RSTEP * i < c.rpt;
    const int mm = ok ? m : 0;
    const int hw = c.Hq * c.Wq;
    const int b = mm / hw;
    const int rem = mm - b * hw;
    const int oy = rem / c.Wq, ox = rem - oy * c.Wq;
    const int iy0 = oy * c.i_mul + c.iy_add, ix0 = ox * c.i_mul + c.ix_add;
    const int pix = (b * p.H + iy0) * p.W + ix0;          // may be "negative": only used where the tap is valid
    roff[i] = ((uint32_t)pix * (uint32_t)p.ldx + qa * 8) * 2u;
    roff2[i] = ((uint32_t)pix * (uint32_t)p.ldx2 + qa * 8) * 2u;
    if (PRO == CX_PRO_JOIN) {                // 1x1, stride 1: row m IS pixel m of the side planes (common.h cx_side_chunk, K % 64 == 0:
      roff3[i] = (uint32_t)pix * 8u + qa;    // chunk index of (row, channel step 0); a step further on is M * 8 chunks further on)
      roffp[i] = ((uint32_t)pix * (uint32_t)p.ldpo + qa * 8) * 2u;
    }
    // tap (dy, dx) is valid where row iy0 + dy and column ix0 + dx exist: separable, so kh + kw tests instead of kh * kw
    uint32_t xb = 0, bits = 0;
#pragma nounroll
    for (int dx = 0; dx < c.ntx; ++dx) xb |= ((uint32_t)(ix0 + dx) < (uint32_t)p.W) ? (1u << dx) : 0u;
#pragma nounroll
    for (int dy = 0; dy < c.nty; ++dy) bits |= (ok && (uint32_t)(iy0 + dy) < (uint32_t)p.H) ? (xb << (dy * c.ntx)) : 0u;
    vbits[i] = bits;
  }
  const int kpt = Kp / BK;
  const int nsteps = ntaps * kpt;
  // the last channel step of a tap may be partial: this thread's 16-byte chunk exists there iff kl_mask (it reads offset 0 and is
  // zeroed when staged otherwise)
  const uint32_t kl_mask = ((kpt - 1) * BK + qa * 8 < p.K) ? 0xffffffffu : 0u;
  const char* __restrict__ X = reinterpret_cast<const char*>(p.x);
  const char* __restrict__ X2 = reinterpret_cast<const char*>(p.x2);
  const char* __restrict__ Wb = reinterpret_cast<const char*>(p.w);
  const char* __restrict__ X3 = reinterpret_cast<const char*>(p.x3);           // CX_PRO_JOIN: lo plane of the identity operand (or null)
  char* __restrict__ PO = reinterpret_cast<char*>(p.pro_out);
  char* __restrict__ PL = reinterpret_cast<char*>(p.po_lo);
  const uint32_t m8 = (uint32_t)M * 8u;                                         // chunks per 64-channel block of a side plane
  const bool join_lo = PRO == CX_PRO_JOIN && p.x3 != nullptr;
  const bool want_lo = PRO == CX_PRO_JOIN && p.po_lo != nullptr;               // two-plane output (else the single bf16 plane of cx_affine2_relu_mask)
  const bool join_out = PRO == CX_PRO_JOIN && nt == 0;                         // the first N tile of a row block writes the side outputs
  uint32_t woff[G::NB];
#pragma unroll
  for (int i = 0; i < G::NB; ++i) {          // N % 8 tiles: rows past N re-read row N - 1, their accumulator columns are never stored
    const int nrow = n0 + r0 + G::RSTEP * i;
    woff[i] = ((uint32_t)(nrow < p.N ? nrow : p.N - 1) * (uint32_t)p.K + qa * 8) * 2u;
  }
  const uint32_t wtap = (uint32_t)p.N * (uint32_t)p.K * 2u;      // weight bytes per tap

  // Activations: two register sets (requests run two steps ahead of the multiplication).  Weights: one set, requested one step
  // ahead - each row is requested again right after it has been stored to LDS (they come from L2, and a second set does not fit).
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  struct Regs {
    u32x4 a[G::NA], a2[G::NA];
    u32x2 a3[PRO == CX_PRO_JOIN ? G::NA : 1];
    int tap, kc;
  };
  Regs set0, set1;
  u32x4 wreg[G::NB];
  // next step to request: tap (row, column), channel step; byte offsets of the step inside the activation / weight tensors
  int q_tap = 0, q_dy = 0, q_dx = 0, q_kc = 0;
  uint32_t q_x = 0, q_x2 = 0, q_w = (uint32_t)(c.wy0 * p.kw + c.wx0) * wtap;

  // Every load is unconditional on an in-bounds address (a branch around a load serialises the prefetch): invalid taps read
  // offset 0 and are zeroed when staged.
  auto issue_a = [&](Regs& R, int i) __attribute__((always_inline)) {
    const uint32_t km = q_kc == kpt - 1 ? kl_mask : 0xffffffffu;                      // partial last channel step
    const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)vbits[i], q_tap, 1) & km; // 0 or 0xffffffff
    R.a[i] = ld16(X, (roff[i] + q_x) & m);
    if (PRO == CX_PRO_AFFINE2 || PRO == CX_PRO_JOIN) R.a2[i] = ld16(X2, (roff2[i] + q_x2) & m);
    if (PRO == CX_PRO_JOIN) {
      if (join_lo) R.a3[i] = *reinterpret_cast<const u32x2*>(X3 + ((size_t)((roff3[i] + (uint32_t)q_kc * m8) & m) << 3));
      else R.a3[i] = u32x2{0u, 0u};
    }
  };
  // (a weight chunk past K is never zeroed: the activation chunk it meets is, and the weights read instead - offset 0 - are finite)
  auto issue_w = [&](int i) __attribute__((always_inline)) {
    wreg[i] = ld16(Wb, (woff[i] + q_w) & (q_kc == kpt - 1 ? kl_mask : 0xffffffffu));
  };
  auto advance = [&](Regs& R) __attribute__((always_inline)) {
    R.tap = q_tap;
    R.kc = q_kc;
    q_x += BK * 2;
    q_x2 += BK * 2;
    q_w += BK * 2;
    if (++q_kc == kpt) {
      q_kc = 0;
      ++q_tap;
      if (++q_dx == c.ntx) {
        q_dx = 0;
        ++q_dy;
      }
      q_x = (uint32_t)((q_dy * p.W + q_dx) * p.ldx) * 2u;
      q_x2 = (uint32_t)((q_dy * p.W + q_dx) * p.ldx2) * 2u;
      q_w = (uint32_t)((c.wy0 + c.wstep * q_dy) * p.kw + c.wx0 + c.wstep * q_dx) * wtap;
    }
  };

  float ca[8], cb[8], cc[8];
  u32x2 jlo[PRO == CX_PRO_JOIN ? G::NA : 1];           // CX_PRO_JOIN: lo bytes / sign bits of the row being staged
  uint32_t jmask[PRO == CX_PRO_JOIN ? G::NA : 1];
  auto load_coef = [&](const Regs& R) __attribute__((always_inline)) {
    if (PRO != CX_PRO_NONE) {
      const int c0 = R.kc * BK + qa * 8;
      *reinterpret_cast<float4*>(ca) = *reinterpret_cast<const float4*>(coef + c0);
      *reinterpret_cast<float4*>(ca + 4) = *reinterpret_cast<const float4*>(coef + c0 + 4);
      *reinterpret_cast<float4*>(cb) = *reinterpret_cast<const float4*>(coef + Kp + c0);
      *reinterpret_cast<float4*>(cb + 4) = *reinterpret_cast<const float4*>(coef + Kp + c0 + 4);
      if (PRO == CX_PRO_AFFINE2) {
        *reinterpret_cast<float4*>(cc) = *reinterpret_cast<const float4*>(coef + 2 * Kp + c0);
        *reinterpret_cast<float4*>(cc + 4) = *reinterpret_cast<const float4*>(coef + 2 * Kp + c0 + 4);
      }
    }
  };
  // dword j (two channels) of staged row i -> o[j]; after j == 3 the row is masked and written
  auto stage_unit = [&](const Regs& R, int i, int j, u32x4& o, char* A) __attribute__((always_inline)) {
    const uint32_t g = R.a[i][j];
    if (PRO == CX_PRO_NONE) {
      o[j] = g;
    } else if (PRO == CX_PRO_AFFINE_RELU) {
      o[j] = relu_pk(packbf(fmaf(bf_lo(g), ca[2 * j], cb[2 * j]), fmaf(bf_hi(g), ca[2 * j + 1], cb[2 * j + 1])));
    } else if (PRO == CX_PRO_JOIN) {
      // out = relu(bn3(y3) + identity) of the block below (common.h cx_join2: the standalone pass computes the same bits): the hi word
      // is this convolution's operand; hi / lo / sign bits leave as side outputs once the row's four dwords are done
      if (j == 0) jlo[i] = u32x2{0u, 0u}, jmask[i] = 0u;
      uint32_t lo_acc = 0u;
      o[j] = cx_join2(g, R.a2[i][j], join_lo, R.a3[i][j >> 1], j, ca[2 * j], ca[2 * j + 1], 1.f, 1.f, cb[2 * j], cb[2 * j + 1], want_lo, lo_acc,
                      jmask[i]);
      jlo[i][j >> 1] |= lo_acc;
    } else {
      const uint32_t y = R.a2[i][j];
      o[j] = packbf(fmaf(bf_lo(g), ca[2 * j], fmaf(bf_lo(y), cb[2 * j], cc[2 * j])),
                    fmaf(bf_hi(g), ca[2 * j + 1], fmaf(bf_hi(y), cb[2 * j + 1], cc[2 * j + 1])));
    }
    if (j == 3) {
      const uint32_t vm = (uint32_t)__builtin_amdgcn_sbfe((int)vbits[i], R.tap, 1) & (R.kc == kpt - 1 ? kl_mask : 0xffffffffu);
      o &= vm;
      *reinterpret_cast<u32x4*>(A + (r0 + G::RSTEP * i) * PITCH + qa * 16) = o;
      if (PRO == CX_PRO_JOIN) {
        if (join_out && vm) {                // (K % 64 == 0: vm is the row's validity)
          const uint32_t chunk = roff3[i] + (uint32_t)R.kc * m8;
          *reinterpret_cast<u32x4*>(PO + (size_t)(roffp[i] + (uint32_t)R.kc * (BK * 2u))) = o;
          if (want_lo) *reinterpret_cast<u32x2*>(PL + ((size_t)chunk << 3)) = jlo[i];
          if (p.po_mask) p.po_mask[chunk] = (uint8_t)jmask[i];
        }
      }
    }
  };
  auto stage_w = [&](int i, char* Bt) __attribute__((always_inline)) {
    *reinterpret_cast<u32x4*>(Bt + (r0 + G::RSTEP * i) * PITCH + qa * 16) = wreg[i];
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int lrow = lane & 31, lh = lane >> 5;
  const char* fragA = tiles + (wm * 64 + lrow) * PITCH + lh * 16;
  const char* fragB = tiles + G::A_BYTES + (wn * 64 + lrow) * PITCH + lh * 16;

  // One step = 16 MFMAs of LDS image `bufc`, and between them, one packet each, the step's other work: the requests of step
  // (this + 2) into Rn (ISSUE) and the prologue + LDS stores of step (this + 1) from Rc into image `bufn` (STAGE).  The packets
  // are fenced (sched_barrier) so that vector, memory and LDS instructions issue while the matrix pipe runs: all waves of a
  // workgroup move in lock step between barriers, and phases of their own would leave every pipe idle most of the time
  // (measured with s_memtime: 2300 cycles per step in phases, of which the MFMAs need 512).
  auto step = [&](Regs& Rn, const Regs& Rc, int bufc, int bufn, auto IssueC, auto StageC) __attribute__((always_inline)) {
    constexpr bool ISSUE = decltype(IssueC)::value, STAGE = decltype(StageC)::value;
    // (the 128 x 128 form has four staged rows AND four weight rows per thread: fenced, it runs out of registers)
    constexpr bool FENCE = kInterleave && !(WMW == 2 && WNW == 2);
    const char* A = fragA + bufc * G::STAGE;
    const char* Bt = fragB + bufc * G::STAGE;
    char* As = tiles + bufn * G::STAGE;
    char* Bs = As + G::A_BYTES;
    u32x4 o[G::NA];
    // fragments of the next k group are read ahead of the current group's MFMAs, except in the two-tensor forms with four staged
    // rows per thread, which have no registers left for a second fragment set
    constexpr int FD = ((PRO == CX_PRO_AFFINE2 || PRO == CX_PRO_JOIN) && G::NA == 4) ? 1 : 2;
    bf16x8 fa[FD][2], fb[FD][2];
    auto read_frags = [&](int slot, int kk) __attribute__((always_inline)) {
      fa[slot][0] = *reinterpret_cast<const bf16x8*>(A + kk * 32);
      fa[slot][1] = *reinterpret_cast<const bf16x8*>(A + 32 * PITCH + kk * 32);
      fb[slot][0] = *reinterpret_cast<const bf16x8*>(Bt + kk * 32);
      fb[slot][1] = *reinterpret_cast<const bf16x8*>(Bt + 32 * PITCH + kk * 32);
    };
    read_frags(0, 0);
    if (STAGE) load_coef(Rc);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int cur = FD == 2 ? (kk & 1) : 0;
      if (FD == 2 && kk < 3) read_frags(cur ^ 1, kk + 1);
      if (FD == 1 && kk > 0) read_frags(0, kk);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc[q >> 1][q & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][q >> 1], fb[cur][q & 1], acc[q >> 1][q & 1], 0, 0, 0);
        if (FENCE) __builtin_amdgcn_sched_barrier(0);
        const int slot = kk * 4 + q;
        if (ISSUE && q == 0 && kk < G::NA) issue_a(Rn, kk);
        if (STAGE) {
          // 4 * NA dword units spread over the 16 slots
          if ((slot * G::NA) % 4 == 0) {
            const int u = slot * G::NA / 4;
            stage_unit(Rc, u >> 2, u & 3, o[u >> 2], As);
          }
          if (q == 2 && kk < G::NB) stage_w(kk, Bs);
        }
        if (ISSUE) {
          if (q == 3 && kk < G::NB) issue_w(kk);         // the row stored one packet ago
          if (slot == 15) advance(Rn);
        }
        if (FENCE) __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  using T = std::true_type;
  using F = std::false_type;

  // prologue of the pipeline: step 0 requested and staged, step 1 requested
  {
#pragma unroll
    for (int i = 0; i < G::NA; ++i) issue_a(set0, i);
#pragma unroll
    for (int i = 0; i < G::NB; ++i) issue_w(i);
    advance(set0);
    __syncthreads();                       // coefficient table visible
    load_coef(set0);
    u32x4 o[G::NA];
#pragma unroll
    for (int i = 0; i < G::NA; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) stage_unit(set0, i, j, o[i], tiles);
#pragma unroll
    for (int i = 0; i < G::NB; ++i) stage_w(i, tiles + G::A_BYTES);
    if (nsteps > 1) {
#pragma unroll
      for (int i = 0; i < G::NA; ++i) issue_a(set1, i);
#pragma unroll
      for (int i = 0; i < G::NB; ++i) issue_w(i);
      advance(set1);
    }
    __syncthreads();
  }

  // Steps in pairs while two more remain to be requested: the requests inside this loop are unconditional, so the compiler's
  // s_waitcnt in the staging packets counts the younger set's loads (vmcnt(n) instead of a full drain).
  int s = 0;
  unsigned long long tacc[5] = {0, 0, 0, 0, 0}, tprev = 0;
  auto stamp = [&](int k) __attribute__((always_inline)) {
    if (DBG) {
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      tacc[k] += t - tprev;
      tprev = t;
    }
  };
  if (DBG) tprev = __builtin_amdgcn_s_memtime();
  for (; s + 3 < nsteps; s += 2) {
    step(set0, set1, 0, 1, T{}, T{});      // multiplies step s, requests s+2, stages s+1
    stamp(0);
    __syncthreads();
    stamp(1);
    step(set1, set0, 1, 0, T{}, T{});
    stamp(0);
    __syncthreads();
    stamp(1);
  }
  if (DBG && (wave == 0 || wave == G::NW - 1) && lane == 0 && blockIdx.x < 2048) {
    unsigned long long* o = conv_mm_stamps + (blockIdx.x * 2 + (wave ? 1 : 0)) * 8;
    for (int k = 0; k < 4; ++k) o[k] = tacc[k];
    o[4] = s;
  }
  // one to three steps left: LDS image 0 = step s, set1 = step s+1 (if any), nothing requested for step s+2 yet
  const int left = nsteps - s;
  if (left == 3) {
    step(set0, set1, 0, 1, T{}, T{});
    __syncthreads();
    step(set1, set0, 1, 0, F{}, T{});
    __syncthreads();
    step(set0, set1, 0, 1, F{}, F{});
  } else if (left == 2) {
    step(set0, set1, 0, 1, F{}, T{});
    __syncthreads();
    step(set1, set0, 1, 0, F{}, F{});
  } else {
    step(set0, set1, 0, 1, F{}, F{});
  }
  __syncthreads();

  // ---------------------------------------------------------------- epilogue: 64-row groups through LDS
  constexpr int CPR = BN / 8;              // 16-byte chunks per row
  constexpr int RPP = G::NT / CPR;         // rows per pass
  constexpr int NPASS = 64 / RPP;
  constexpr int NGRP = G::BM / 64;
  const int cq = tid % CPR, rr = tid / CPR;
  const bool nok = n0 + cq * 8 < p.N;     // a partial last tile (N % 8 == 0): whole 8-channel vectors are in or out
  const int nch = nok ? n0 + cq * 8 : 0;
  float* etile = reinterpret_cast<float*>(tiles);
  bf16* __restrict__ Y = reinterpret_cast<bf16*>(p.y);
  const bf16* __restrict__ EX = reinterpret_cast<const bf16*>(p.ex);
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
  float esc[8], esh[8], emu[8], er[8], escale[8];
  if (EPI == CX_EPI_MASK) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      esc[j] = p.e_sc[nch + j];
      esh[j] = p.e_sh[nch + j];
      emu[j] = p.e_mu[nch + j];
      er[j] = p.e_r[nch + j];
      escale[j] = p.e_scale[nch + j];
    }
  }
  if (EPI == CX_EPI_JOIN) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      emu[j] = p.e_mu[nch + j];
      er[j] = p.e_r[nch + j];
    }
  }
  unsigned mkb[NGRP][NPASS];               // CX_EPI_JOIN: the forward join's sign bits of this row's 8 channels
  const bool want_stats = p.stat_sum != nullptr;

  U128 xv[NGRP][NPASS], old[NGRP][NPASS];
  int mo[NGRP][NPASS];                     // output pixel of each epilogue row (a parity class strides the output image)
#pragma unroll
  for (int g = 0; g < NGRP; ++g)
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      const int m = mt * c.rpt + g * 64 + pass * RPP + rr;
      int mc = m < M ? m : M - 1;
      if (c.o_mul != 1) {
        const int hw = c.Hq * c.Wq;
        const int b = mc / hw;
        const int rem = mc - b * hw;
        const int oy = rem / c.Wq, ox = rem - oy * c.Wq;
        mc = (b * p.Ho + oy * c.o_mul + c.oy_add) * p.Wo + ox * c.o_mul + c.ox_add;
      }
      mo[g][pass] = mc;
      if (EPI == CX_EPI_MASK || EPI == CX_EPI_JOIN) xv[g][pass].u = *reinterpret_cast<const uint4*>(EX + (size_t)mc * p.ldex + nch);
      if (EPI == CX_EPI_JOIN) mkb[g][pass] = p.emask[cx_side_chunk((size_t)mc, nch >> 3, (size_t)p.B * p.Ho * p.Wo, p.N)];
      if (p.accumulate)
        old[g][pass].u = *reinterpret_cast<const uint4*>(Y + (size_t)mc * p.ldy + nch);
      else
        old[g][pass].u = make_uint4(0, 0, 0, 0);
    }

#pragma unroll
  for (int g = 0; g < NGRP; ++g) {
    if (wm == g) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int col = (wn * 2 + j) * 32 + lrow;
            etile[row * G::EPITCH + col] = acc[i][j][r];
          }
    }
    __syncthreads();
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
      const int row = pass * RPP + rr;
      const int m = mt * c.rpt + g * 64 + row;
      if (m < M && g * 64 + row < c.rpt && nok) {
        const float4 v0 = *reinterpret_cast<const float4*>(etile + row * G::EPITCH + cq * 8);
        const float4 v1 = *reinterpret_cast<const float4*>(etile + row * G::EPITCH + cq * 8 + 4);
        float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
        U128 o;
        if (EPI == CX_EPI_STORE) {
          const uint32_t ow[4] = {old[g][pass].u.x, old[g][pass].u.y, old[g][pass].u.z, old[g][pass].u.w};
          float t[8];
#pragma unroll
          for (int j = 0; j < 4; ++j) { t[2 * j] = v[2 * j] + cx_bf_lo(ow[j]); t[2 * j + 1] = v[2 * j + 1] + cx_bf_hi(ow[j]); }
          o.u = cx_pack8_stats(t, true, true, s1, s2);
        } else if (EPI == CX_EPI_JOIN) {
          // the gradient of the join's output, rounded as CX_EPI_STORE would have stored it, then the join's ReLU mask and
          // the sums of its BatchNorm's backward
          const uint32_t ow[4] = {old[g][pass].u.x, old[g][pass].u.y, old[g][pass].u.z, old[g][pass].u.w};
          const uint32_t xw[4] = {xv[g][pass].u.x, xv[g][pass].u.y, xv[g][pass].u.z, xv[g][pass].u.w};
          uint32_t w4[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const uint32_t bits = mkb[g][pass] >> (2 * j);
            const uint32_t keep = ((bits & 1u) ? 0x0000ffffu : 0u) | ((bits & 2u) ? 0xffff0000u : 0u);
            w4[j] = cx_packbf(v[2 * j] + cx_bf_lo(ow[j]), v[2 * j + 1] + cx_bf_hi(ow[j])) & keep;
            const float dl = cx_bf_lo(w4[j]), du = cx_bf_hi(w4[j]);
            s1[2 * j] += dl;
            s1[2 * j + 1] += du;
            s2[2 * j] += dl * (cx_bf_lo(xw[j]) - emu[2 * j]) * er[2 * j];
            s2[2 * j + 1] += du * (cx_bf_hi(xw[j]) - emu[2 * j + 1]) * er[2 * j + 1];
          }
          o.u = make_uint4(w4[0], w4[1], w4[2], w4[3]);
        } else {
          const uint32_t ow[4] = {old[g][pass].u.x, old[g][pass].u.y, old[g][pass].u.z, old[g][pass].u.w};
          const uint32_t xw[4] = {xv[g][pass].u.x, xv[g][pass].u.y, xv[g][pass].u.z, xv[g][pass].u.w};
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
        *reinterpret_cast<uint4*>(Y + (size_t)mo[g][pass] * p.ldy + nch) = o.u;
      }
    }
    __syncthreads();
  }

  if (want_stats) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int d = CPR; d < 64; d <<= 1) {
        s1[j] += __shfl_xor(s1[j], d);
        s2[j] += __shfl_xor(s2[j], d);
      }
    }
    float* scratch = reinterpret_cast<float*>(tiles);
    wg_stat_begin<G::NW>(scratch, BN, tid, G::NT);
    if (lane < CPR) {
#pragma unroll
      for (int j = 0; j < 8; ++j) wg_stat_put(scratch, BN, wave, cq * 8 + j, s1[j], s2[j]);
    }
    wg_stat_end<G::NW>(scratch, BN, tid, G::NT, p.stat_sum, p.stat_sq, p.stat_det, p.stat_det ? c.row0 + mt : (int)blockIdx.x, p.stat_replicas,
                       p.stat_rstride, n0, p.N);
  }
}

#ifdef CX_DIAG                                 // s_memtime stamp instantiations: diagnostic builds only (scratch/stamps_mm.py)
static int g_mm_dbg = 0;
extern "C" int dbg_conv_mm_stamps(unsigned long long* host, int n_words) {     // not part of the ABI
  if (!host) {
    g_mm_dbg = n_words;
    return 0;
  }
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(conv_mm_stamps), (size_t)n_words * 8, 0, hipMemcpyDeviceToHost);
}
#endif

template <int WMW, int WNW, int PRO, int EPI>
int launch(const CxConv& p, const Cls& c, hipStream_t st) {
  using G = MG<WMW, WNW>;
  const int m_tiles = (c.Mq + c.rpt - 1) / c.rpt;
  const int n_tiles = (p.N + G::BN - 1) / G::BN;
  const size_t smem = (size_t)NCoef<PRO>::v * ((p.K + BK - 1) / BK * BK) * 4 + G::MAIN_BYTES;
  if (smem > 160 * 1024) return CX_ESHAPE;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mm_kernel<WMW, WNW, PRO, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    attr_set = true;
  }
#ifdef CX_DIAG
  if (g_mm_dbg && (PRO == CX_PRO_AFFINE_RELU || PRO == CX_PRO_NONE) && EPI == CX_EPI_STORE) {
    constexpr int DP = PRO == CX_PRO_NONE ? CX_PRO_NONE : CX_PRO_AFFINE_RELU;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_mm_kernel<WMW, WNW, DP, CX_EPI_STORE, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((conv_mm_kernel<WMW, WNW, DP, CX_EPI_STORE, true>), dim3(m_tiles * n_tiles), dim3(G::NT), smem, st, p, c, n_tiles);
    return launch_status();
  }
#endif
  CX_KTAG("conv_mm_kernel<%d, %d, %d, %d, false>%s", WMW, WNW, PRO, EPI, p.tstride > 1 ? " x parity classes" : "");
  hipLaunchKernelGGL((conv_mm_kernel<WMW, WNW, PRO, EPI>), dim3(m_tiles * n_tiles), dim3(G::NT), smem, st, p, c, n_tiles);
  return launch_status();
}

// tile form: 1 = 128 x 128 (256 threads), 3 = 128 x 256 (512 threads).  (2 = 256 x 128, MG<4, 2>, was measured and never won:
// it stages twice the activations per weight row, the expensive operand; not instantiated.)
template <int PRO, int EPI>
int launch_form(const CxConv& p, const Cls& c, hipStream_t st, int form) {
  if (form == 3) return launch<2, 4, PRO, EPI>(p, c, st);
  return launch<2, 2, PRO, EPI>(p, c, st);
}

int launch_any(const CxConv& p, const Cls& c, hipStream_t st, int form) {
  if (p.prologue == CX_PRO_JOIN) return launch<2, 4, CX_PRO_JOIN, CX_EPI_STORE>(p, c, st);     // (128 x 256 tiles whatever N: two staged rows per thread)
  if (p.epilogue == CX_EPI_STORE) {
    if (p.prologue == CX_PRO_NONE) return launch_form<CX_PRO_NONE, CX_EPI_STORE>(p, c, st, form);
    if (p.prologue == CX_PRO_AFFINE_RELU) return launch_form<CX_PRO_AFFINE_RELU, CX_EPI_STORE>(p, c, st, form);
    return launch_form<CX_PRO_AFFINE2, CX_EPI_STORE>(p, c, st, form);
  }
  if (p.epilogue == CX_EPI_JOIN) return launch_form<CX_PRO_AFFINE2, CX_EPI_JOIN>(p, c, st, form);
  if (p.prologue == CX_PRO_AFFINE2) return launch_form<CX_PRO_AFFINE2, CX_EPI_MASK>(p, c, st, form);
  return launch_form<CX_PRO_NONE, CX_EPI_MASK>(p, c, st, form);
}

}  // namespace

// CxConv.kernel_hint (ABI 10) lets tests and micro-benchmarks pin the kernel choice per call (on = 0: conv_gemm.hip; form = 1 | 3: tile).

// fewest k-steps per tile for which a shape with a partial last N tile is taken (CX_MM_MIN_STEPS)
static int mm_min_steps() {
  static const int v = cx_diag_int("CX_MM_MIN_STEPS", 14);
  return v;
}

// Called by cx_conv_gemm after validation, once the specialised DenseNet kernels have declined.
int cx_try_conv_mm(const CxConv& p, hipStream_t st, bool* handled) {
  *handled = false;
  static const int env_on0 = cx_diag_int("CX_MM", 1);
  static const int env_form0 = cx_diag_int("CX_MM_FORM", 0);
  const int g_mm_on = (p.kernel_hint & 0xff) - 1, g_mm_form = ((p.kernel_hint >> 8) & 0xff) - 1;      // -1: not pinned
  const bool pjoin = p.prologue == CX_PRO_JOIN;      // (validated by cx_conv_gemm: 1x1, stride 1, K % 64 == 0, store epilogue)
  const int env_on = (p.epilogue == CX_EPI_JOIN || pjoin) ? 1 : g_mm_on >= 0 ? g_mm_on : env_on0;
  const int env_form = g_mm_form >= 0 ? g_mm_form : env_form0;
  if (!env_on || p.mode != CX_MODE_CONV || p.tstride > 2 || (p.K % 8) || p.K < BK || (p.N % 8) || p.kh * p.kw > 32 || p.dtype != CX_DT_BF16) return 0;
  const bool ok_combo = (p.epilogue == CX_EPI_STORE && (p.prologue == CX_PRO_NONE || p.prologue == CX_PRO_AFFINE_RELU || p.prologue == CX_PRO_AFFINE2)) ||
                        (p.epilogue == CX_EPI_MASK && (p.prologue == CX_PRO_NONE || p.prologue == CX_PRO_AFFINE2)) ||
                        (p.epilogue == CX_EPI_JOIN && p.prologue == CX_PRO_AFFINE2 && p.tstride <= 1 && (p.N % 128) == 0) ||
                        (pjoin && p.epilogue == CX_EPI_STORE);
  if (!ok_combo) return 0;
  const bool join = p.epilogue == CX_EPI_JOIN || pjoin;      // this file has the only join epilogue / prologue: taken whatever the tile heuristics say
  // 32-bit byte offsets inside every tensor
  if ((unsigned long long)p.B * p.H * p.W * (unsigned long long)(p.ldx > p.ldx2 ? p.ldx : p.ldx2) * 2 >= (1ull << 32)) return 0;
  if ((unsigned long long)p.kh * p.kw * p.N * p.K * 2 >= (1ull << 32)) return 0;
  if (pjoin && (unsigned long long)p.B * p.H * p.W * (unsigned long long)p.ldpo * 2 >= (1ull << 32)) return 0;
  const int ts = p.tstride > 1 ? 2 : 1;
  // Measured on the ResNet152 shapes (scratch/bench_mm.py): 128 x 256 tiles wherever N allows and a tile has more than four
  // k-steps (1.9-2.7x the generic kernel); with at most four steps a tile is prologue + epilogue, the smaller tile wins, and with
  // one or two steps the generic kernel (32-channel steps, three workgroups per CU) is as fast.
  const int nsteps = (ts == 2 ? (p.kh * p.kw + 3) / 4 : p.kh * p.kw) * ((p.K + BK - 1) / BK);
  if (!env_form && !join && ts == 1 && nsteps <= 2) return 0;
  // Partial last tiles (N a multiple of 8, not of 128: EfficientNet widths, AAConv branches).  Measured on the EfficientNet-B4
  // shapes (scratch/bench_mm_eff.py): with at least 14 k-steps per tile this kernel is 1.3-2x the generic one (K = 960 .. 2688
  // projections and their gradients), with fewer the padded part of the tile costs more than the pipeline gains.
  if (!env_form && !pjoin && (p.N % 128) && (nsteps < mm_min_steps() || p.N < 96)) return 0;
  // partial last tiles (N % 8 == 0): the wider tile only where it does not add padding
  const int pad1 = (p.N + 127) / 128 * 128, pad3 = (p.N + 255) / 256 * 256;
  int form = (pad3 == pad1 && nsteps > 4) ? 3 : 1;
  if (env_form) form = env_form == 3 ? 3 : 1;
  if (form == 3 && (size_t)NCoef<CX_PRO_AFFINE2>::v * ((p.K + BK - 1) / BK * BK) * 4 + MG<2, 4>::MAIN_BYTES > 160 * 1024) form = 1;
  const int bm = 128;

  if (ts == 1) {
    Cls c;
    c.nty = p.kh, c.ntx = p.kw;
    c.i_mul = p.stride, c.iy_add = -p.pad, c.ix_add = -p.pad;
    c.wy0 = c.wx0 = 0, c.wstep = 1;
    c.Hq = p.Ho, c.Wq = p.Wo, c.Mq = p.B * p.Ho * p.Wo;
    c.o_mul = 1, c.oy_add = c.ox_add = 0, c.row0 = 0;
    c.rpt = bm;
    if (pjoin && p.N <= 256) {
      // bandwidth-bound (one N tile): equal rounds on every CU (one workgroup per CU, 256 CUs)
      const int rounds = ((c.Mq + bm - 1) / bm + 255) / 256;
      int r = (c.Mq + rounds * 256 - 1) / (rounds * 256);
      r = (r + 3) & ~3;
      c.rpt = r < 32 ? 32 : r > bm ? bm : r;
    }
    if (const int e = stat_rows_check(p, (c.Mq + c.rpt - 1) / c.rpt)) {
      *handled = true;
      return e;
    }
    *handled = true;
    return launch_any(p, c, st, form);
  }

  // input gradient of a stride-2 convolution: one launch per parity class of the output pixels
  Cls cls[4];
  int n = 0, rows = 0;
  for (int py = 0; py < 2; ++py)
    for (int px = 0; px < 2; ++px) {
      const int d0y = (p.pad + py) & 1, d0x = (p.pad + px) & 1;
      Cls c;
      c.nty = d0y < p.kh ? (p.kh - d0y + 1) / 2 : 0;
      c.ntx = d0x < p.kw ? (p.kw - d0x + 1) / 2 : 0;
      c.Hq = (p.Ho - py + 1) / 2, c.Wq = (p.Wo - px + 1) / 2;
      c.Mq = p.B * c.Hq * c.Wq;
      if (c.Mq <= 0) continue;
      if (c.nty * c.ntx == 0) {
        // no tap reaches these pixels: their gradient is zero.  Accumulating, that is nothing to do (the statistic sums gain
        // zeros); storing, the generic kernel writes the zeros.
        if (!p.accumulate) return 0;
        continue;
      }
      c.i_mul = 1;
      c.iy_add = (py + d0y - p.pad) >> 1, c.ix_add = (px + d0x - p.pad) >> 1;      // (arithmetic shift: exact, may be negative)
      c.wy0 = d0y, c.wx0 = d0x, c.wstep = 2;
      c.o_mul = 2, c.oy_add = py, c.ox_add = px;
      c.row0 = rows;
      c.rpt = bm;
      rows += (c.Mq + bm - 1) / bm;
      cls[n++] = c;
    }
  if (n == 0) return 0;
  *handled = true;
  if (const int e = stat_rows_check(p, rows)) return e;
  for (int i = 0; i < n; ++i)
    if (const int e = launch_any(p, cls[i], st, form)) return e;
  return 0;
}
