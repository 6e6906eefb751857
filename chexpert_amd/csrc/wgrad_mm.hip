// Weight gradient of the wide 1x1 convolutions (ResNet bottlenecks): dW[n][c] += sum_px dZ[px][n] * A[px][c], stride 1,
// N a multiple of 8 (partial last 128-channel tile) output channels, pixels a multiple of 64, NHWC bf16 operands, fp32 OIHW result.
//
// Same recipe as conv_mm.hip, which measured what bounds these kernels (vector-instruction issue with two waves per SIMD, not
// MFMA or bandwidth): tiles of 128 x 256 (or 256 x 128, 128 x 128) outputs so that each operand element is transformed for 256
// (128) outputs instead of 128 / 32, 64 pixels per step, the operands of step s+2 requested while step s multiplies (two register
// sets), prologue coefficients of the thread's fixed channel chunk held in registers, and the step's loads / prologue arithmetic /
// LDS stores issued in packets between the MFMAs.  Both operands have the reduction index (pixel) as their slow memory axis:
// fragments come from the [pixel][channel] LDS images through ds_read_b64_tr_b16.
// The pixel range is split over workgroups; partial tiles leave through fp32 atomics, or, with CxWgrad.scratch, through slabs
// that a second launch adds in split order (bit-reproducible).
#include <cstdlib>
#include <type_traits>
#include "common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

constexpr int PX = 64;                                 // pixels per step

template <int WA, int WB>
struct WG {
  static constexpr int TN = 64 * WA;                   // dZ channels (rows of dW) per tile
  static constexpr int TC = 64 * WB;                   // input channels (columns of dW) per tile
  static constexpr int NW = WA * WB;
  static constexpr int NT = 64 * NW;
  static constexpr int GQ = TN / 8, XQ = TC / 8;       // 16-byte chunks per image row
  static constexpr int RG = NT / GQ, RX = NT / XQ;     // rows covered by one staging pass
  static constexpr int NG = PX / RG, NX = PX / RX;     // chunks per thread per step
  static constexpr int GP = TN * 2 + 64;               // LDS pitches: == 64 B (mod 256 B) for the transposing reads
  static constexpr int XP = TC * 2 + 64;
  static constexpr int STAGE = PX * (GP + XP);
  static constexpr int UNITS = (NG + NX) * 4;          // dword units of prologue work per step
};

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t packbf(float a, float b) {
  union {
    bf16x2 h;
    uint32_t u;
  } o;
  o.h = __builtin_convertvector(f32x2{a, b}, bf16x2);
  return o.u;
}
__device__ __forceinline__ uint32_t relu_pk(uint32_t v) {
  union {
    uint32_t u;
    s16x2 s;
  } a, r;
  a.u = v;
  r.s = __builtin_elementwise_max(a.s, s16x2{0, 0});
  return r.u;
}
__device__ __forceinline__ u32x4 ld16(const char* base, uint32_t off) { return *reinterpret_cast<const u32x4*>(base + (size_t)off); }

__device__ __forceinline__ bf16x8 tr_frag(const char* tile, int pitch, int k0, int ch0, int lane) {
  // fragment of the 32x32x16 MFMA: this lane gets channel ch0 + (lane&31), pixels k0 + 8*(lane>>5) + 0..7
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const char* base = tile + (k0 + 8 * (g >> 1) + q) * pitch + (ch0 + 16 * (g & 1) + 4 * pp) * 2;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  // (joined by a shuffle + bit cast: assembled element by element the compiler emits a v_bfi per dword on the loaded registers and
  // waits for the read right where it is issued, not where the MFMA uses it -- common.h cx_join_tr)
  return cx_join_tr(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base)), __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + 4 * pitch)));
}

template <int WA, int WB, int GPRO, int XPRO>
__global__ __launch_bounds__(64 * WA * WB) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad_mm_kernel(const CxWgrad p, const int c_tiles,
                                                                                                        const int n_tiles,
                                                                                                        const int total_steps,
                                                                                                        const int steps_per_split,
                                                                                                        float* __restrict__ slab) {
  using G = WG<WA, WB>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wa = wave / WB, wb = wave % WB;
  int id = xcd_remap(blockIdx.x, gridDim.x);
  const int ct = id % c_tiles;                         // tiles of one pixel range are neighbours: operands shared in L2
  id /= c_tiles;
  const int nt = id % n_tiles;
  const int split = id / n_tiles;
  const int c0 = ct * G::TC, n0 = nt * G::TN;

  // this thread's chunk columns and their prologue coefficients (fixed for the whole kernel)
  const int qg = tid % G::GQ, rg0 = tid / G::GQ;
  const int qx = tid % G::XQ, rx0 = tid / G::XQ;
  const bool xact = c0 + qx * 8 < p.K;
  const uint32_t xmask = xact ? 0xffffffffu : 0u;
  const int xc = xact ? c0 + qx * 8 : 0;
  const bool gact = n0 + qg * 8 < p.N;            // partial last N tile (N % 8 == 0): chunks past N are staged as zeros
  const uint32_t gmask = gact ? 0xffffffffu : 0u;
  const int gn = gact ? n0 + qg * 8 : 0;
  float ga[8], gb[8], gc[8], pa[8], pb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    ga[j] = GPRO == CX_PRO_AFFINE2 ? p.ga[gn + j] : 1.f;
    gb[j] = GPRO == CX_PRO_AFFINE2 ? p.gb[gn + j] : 0.f;
    gc[j] = GPRO == CX_PRO_AFFINE2 ? p.gc[gn + j] : 0.f;
    pa[j] = XPRO == CX_PRO_AFFINE_RELU ? p.pa[xc + j] : 1.f;
    pb[j] = XPRO == CX_PRO_AFFINE_RELU ? p.pb[xc + j] : 0.f;
  }

  const int step0 = split * steps_per_split;
  int nsteps = total_steps - step0;
  if (nsteps > steps_per_split) nsteps = steps_per_split;

  // byte offsets of the thread's chunks inside a 64-pixel step; the step itself is a scalar byte offset
  uint32_t goff[G::NG], goff2[G::NG], xoff[G::NX];
#pragma unroll
  for (int i = 0; i < G::NG; ++i) {
    goff[i] = ((uint32_t)(rg0 + G::RG * i) * (uint32_t)p.ldg + gn) * 2u;
    goff2[i] = ((uint32_t)(rg0 + G::RG * i) * (uint32_t)p.ldg2 + gn) * 2u;
  }
#pragma unroll
  for (int i = 0; i < G::NX; ++i) xoff[i] = ((uint32_t)(rx0 + G::RX * i) * (uint32_t)p.ldx + xc) * 2u;
  const char* __restrict__ Gb = reinterpret_cast<const char*>(p.g);
  const char* __restrict__ G2b = reinterpret_cast<const char*>(p.g2);
  const char* __restrict__ Xb = reinterpret_cast<const char*>(p.x);
  uint32_t q_g = (uint32_t)step0 * PX * (uint32_t)p.ldg * 2u, q_g2 = (uint32_t)step0 * PX * (uint32_t)p.ldg2 * 2u,
           q_x = (uint32_t)step0 * PX * (uint32_t)p.ldx * 2u;
  const uint32_t sg = PX * (uint32_t)p.ldg * 2u, sg2 = PX * (uint32_t)p.ldg2 * 2u, sx = PX * (uint32_t)p.ldx * 2u;

  // dZ (two tensors under AFFINE2): two register sets, requests two steps ahead.  Activations: one set, each chunk requested
  // again right after it has been stored to LDS (one step ahead) - a second set does not fit beside the coefficients.
  struct Regs {
    u32x4 g[G::NG], g2[G::NG];
  };
  Regs set0, set1;
  u32x4 xreg[G::NX];
  auto issue_g = [&](Regs& R, int i) __attribute__((always_inline)) {
    R.g[i] = ld16(Gb, goff[i] + q_g);
    if (GPRO == CX_PRO_AFFINE2) R.g2[i] = ld16(G2b, goff2[i] + q_g2);
  };
  auto issue_x = [&](int i) __attribute__((always_inline)) { xreg[i] = ld16(Xb, xoff[i] + q_x); };
  auto advance = [&]() __attribute__((always_inline)) {
    q_g += sg;
    q_g2 += sg2;
    q_x += sx;
  };
  // dword j (two channels) of chunk i -> o[j]; after j == 3 the chunk is written
  auto unit_g = [&](const Regs& R, int i, int j, u32x4& o, char* Gt) __attribute__((always_inline)) {
    const uint32_t g = R.g[i][j];
    if (GPRO == CX_PRO_NONE) {
      o[j] = g;
    } else {
      const uint32_t y = R.g2[i][j];
      o[j] = packbf(fmaf(bf_lo(g), ga[2 * j], fmaf(bf_lo(y), gb[2 * j], gc[2 * j])),
                    fmaf(bf_hi(g), ga[2 * j + 1], fmaf(bf_hi(y), gb[2 * j + 1], gc[2 * j + 1])));
    }
    if (j == 3) {
      o &= gmask;
      *reinterpret_cast<u32x4*>(Gt + (rg0 + G::RG * i) * G::GP + qg * 16) = o;
    }
  };
  auto unit_x = [&](int i, int j, u32x4& o, char* Xt, bool reissue) __attribute__((always_inline)) {
    const uint32_t x = xreg[i][j];
    if (XPRO == CX_PRO_NONE) {
      o[j] = x;
    } else {
      o[j] = relu_pk(packbf(fmaf(bf_lo(x), pa[2 * j], pb[2 * j]), fmaf(bf_hi(x), pa[2 * j + 1], pb[2 * j + 1])));
    }
    if (j == 3) {
      o &= xmask;
      *reinterpret_cast<u32x4*>(Xt + (rx0 + G::RX * i) * G::XP + qx * 16) = o;
      if (reissue) issue_x(i);
    }
  };
  auto unit = [&](const Regs& R, int u, u32x4& o, char* St, bool reissue) __attribute__((always_inline)) {
    if (u < G::NG * 4)
      unit_g(R, u >> 2, u & 3, o, St);
    else
      unit_x((u - G::NG * 4) >> 2, u & 3, o, St + PX * G::GP, reissue);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // One step = 16 MFMAs of LDS image `bufc`; between them, in fenced packets: the requests of step (this + 2) into Rn and the
  // prologue + LDS stores of step (this + 1) from Rc into image `bufn`.
  auto step = [&](Regs& Rn, const Regs& Rc, int bufc, int bufn, auto IssueC, auto StageC) __attribute__((always_inline)) {
    constexpr bool ISSUE = decltype(IssueC)::value, STAGE = decltype(StageC)::value;
    constexpr bool FENCE = G::NG + G::NX <= 6;               // (the 128 x 128 form keeps 8 chunks per set: no registers to spare)
    const char* Gt = smem + bufc * G::STAGE;
    const char* Xt = Gt + PX * G::GP;
    char* Sn = smem + bufn * G::STAGE;
    u32x4 o;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 af[2], bfr[2];
      af[0] = tr_frag(Gt, G::GP, kk * 16, (wa * 2) * 32, lane);
      af[1] = tr_frag(Gt, G::GP, kk * 16, (wa * 2 + 1) * 32, lane);
      bfr[0] = tr_frag(Xt, G::XP, kk * 16, (wb * 2) * 32, lane);
      bfr[1] = tr_frag(Xt, G::XP, kk * 16, (wb * 2 + 1) * 32, lane);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc[q >> 1][q & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[q >> 1], bfr[q & 1], acc[q >> 1][q & 1], 0, 0, 0);
        if (FENCE) __builtin_amdgcn_sched_barrier(0);
        const int slot = kk * 4 + q;
        if (ISSUE && q == 0 && kk < G::NG) issue_g(Rn, kk);
        if (STAGE) {
#pragma unroll
          for (int u = slot * G::UNITS / 16; u < (slot + 1) * G::UNITS / 16; ++u) unit(Rc, u, o, Sn, ISSUE);
        }
        if (ISSUE && slot == 15) advance();
        if (FENCE) __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  using T = std::true_type;
  using F = std::false_type;

  if (nsteps > 0) {
    // prologue of the pipeline: step 0 requested and staged, step 1 requested
#pragma unroll
    for (int i = 0; i < G::NG; ++i) issue_g(set0, i);
#pragma unroll
    for (int i = 0; i < G::NX; ++i) issue_x(i);
    advance();
    if (nsteps > 1) {
#pragma unroll
      for (int i = 0; i < G::NG; ++i) issue_g(set1, i);
    }
    {
      u32x4 o;
#pragma unroll
      for (int u = 0; u < G::UNITS; ++u) unit(set0, u, o, smem, nsteps > 1);      // (re-requests the activations of step 1)
    }
    if (nsteps > 1) advance();
    __syncthreads();

    int s = 0;
    for (; s + 3 < nsteps; s += 2) {
      step(set0, set1, 0, 1, T{}, T{});      // multiplies step s, requests s+2, stages s+1
      __syncthreads();
      step(set1, set0, 1, 0, T{}, T{});
      __syncthreads();
    }
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
  }

  // ---- partial tile -> fp32 OIHW gradient
  if (p.splits == -7) return;           // (timing-only diagnostic, CX_WGRAD_MM_NOSTORE: results are wrong)
  const int lrow = lane & 31, lh = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = c0 + (wb * 2 + j) * 32 + lrow;
      if (c < p.K) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n0 + (wa * 2 + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          if (n < p.N) dw_out(p.dw, slab, (size_t)p.N * p.K, split, (size_t)n * p.K + c, acc[i][j][r]);
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------ 3x3, stride 1, pad 1
// dW[n][c][dy][dx] += sum_px dZ[px][n] * A[px + (dy-1, dx-1)][c] for N % 128 == 0, K % 128 == 0 (the ResNet bottleneck 3x3 layers).
// A workgroup owns 128 n x 128 c x the THREE taps of one kernel row dy, so one staged dZ tile (the expensive operand: two tensors
// and nine prologue operations per element pair) and one staged activation strip serve three products: the pixels are walked in
// a zero-padded index space (each image row is W + 2 positions, columns 0 and W + 1 are zeros), where the dx = 0, 1, 2 neighbours of
// position p are positions p - 1, p, p + 1 of the strip and never wrap into another image row; the rows above / below the image
// (dy = 0 / 2 at y = 0 / H - 1) are zero rows of the strip.  What a padded position is (real pixel or zero, and which pixel) comes
// from a small LDS table that 136 threads fill three steps ahead (two exact divisions by multiplication per entry), so a staged
// chunk costs one table read, one multiply-add and one AND.  Per step: 24 MFMAs per wave against 16 staged chunks per thread pair.
//
// Stride 2 (S2; the first 3x3 of a ResNet stage, the 3x3 branch of an attention-augmented transition): the same workgroup, the pixels
// walked in the padded OUTPUT index space (each output row is Wo + 1 positions, position 0 a zero), and the activation row
// 2 oy + dy - 1 staged DE-INTERLEAVED as two strips -- even columns 2 ox, odd columns 2 ox + 1 -- so that the three taps of the kernel
// row are again unit shifts: dx = 1 reads the even strip at the position, dx = 2 the odd strip at the position, dx = 0 the odd strip
// one position to the left (column 2 ox - 1; the pad position makes ox = 0 read zeros).  The generic kernel (conv_wgrad.hip) gave
// every (tap, 64-channel tile) its own workgroup: dZ staged and transformed 36 times for a 256-channel layer instead of 6.
struct W3Geo {
  int B, H, W, P, TP;              // P positions per padded row (stride 1: W + 2; stride 2: Wo + 1), TP = B * (rows) * P positions
  uint32_t mP, mH;                 // ceil(2^32 / P), ceil(2^32 / rows per image): q / P == umulhi(q, mP) for q * P < 2^32
  int Ho, Wo;                      // stride 2: the gradient's image (rows per image = Ho); stride 1: == H, W
};

constexpr int W3_GP = 128 * 2 + 64, W3_XP = 128 * 2 + 64;
constexpr int W3_XROWS = 68;       // 66 strip rows used (64 + the two dx neighbours), padded
template <bool S2>
struct W3L {
  static constexpr int NSTRIP = S2 ? 2 : 1;
  static constexpr int STAGE = PX * W3_GP + NSTRIP * W3_XROWS * W3_XP;
  static constexpr int TAB = 64 + NSTRIP * W3_XROWS;
};

template <int GPRO, int XPRO, bool S2>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad3_kernel(const CxWgrad p, const W3Geo g, const int c_tiles,
                                                                                              const int n_tiles, const int total_steps,
                                                                                              const int steps_per_split,
                                                                                              float* __restrict__ slab) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int W3_STAGE = W3L<S2>::STAGE, W3_TAB = W3L<S2>::TAB;
  uint32_t* tab = reinterpret_cast<uint32_t*>(smem + 2 * W3_STAGE);          // [3][W3_TAB]: bit 31 = real pixel, low bits = its index
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wa = wave >> 2, wb = wave & 3;             // 64 n x 32 c x 3 taps per wave
  int id = xcd_remap(blockIdx.x, gridDim.x);
  const int ct = id % c_tiles;
  id /= c_tiles;
  const int nt = id % n_tiles;
  id /= n_tiles;
  const int dy = id % 3;
  const int split = id / 3;
  const int c0 = ct * 128, n0 = nt * 128;
  const int xshift = (dy - 1) * g.P - 1;               // strip row j of a step = position (first position of the step) + xshift + j
  const int ylo = dy == 2 ? 1 : 0, yhi = dy == 0 ? g.H - 2 : g.H - 1;      // image rows of the strip that are this dy's neighbours

  const int q16 = tid & 15, r32 = tid >> 4;             // chunk column, row (+ 32 i) of both images
  // partial last N tile (N a multiple of 8: the 3x3 branches of the attention-augmented layers): this thread's dZ chunk lies past N ->
  // it requests offset 0 and stages zeros, and the epilogue stores no row past N
  const bool nok = n0 + q16 * 8 < p.N;
  const uint32_t nmask = nok ? 0xffffffffu : 0u;
  const int ncoef = nok ? n0 + q16 * 8 : 0;
  float ga[8], gb[8], gc[8], pa[8], pb[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    ga[j] = GPRO == CX_PRO_AFFINE2 ? p.ga[ncoef + j] : 1.f;
    gb[j] = GPRO == CX_PRO_AFFINE2 ? p.gb[ncoef + j] : 0.f;
    gc[j] = GPRO == CX_PRO_AFFINE2 ? p.gc[ncoef + j] : 0.f;
    pa[j] = XPRO == CX_PRO_AFFINE_RELU ? p.pa[c0 + q16 * 8 + j] : 1.f;
    pb[j] = XPRO == CX_PRO_AFFINE_RELU ? p.pb[c0 + q16 * 8 + j] : 0.f;
  }
  const int step0 = split * steps_per_split;
  int nsteps = total_steps - step0;
  if (nsteps > steps_per_split) nsteps = steps_per_split;

  const char* __restrict__ Gb = reinterpret_cast<const char*>(p.g);
  const char* __restrict__ G2b = reinterpret_cast<const char*>(p.g2);
  const char* __restrict__ Xb = reinterpret_cast<const char*>(p.x);
  const uint32_t ldg2b = p.ldg * 2, ldg22b = p.ldg2 * 2, ldx2b = p.ldx * 2;
  const uint32_t gcol = (n0 + q16 * 8) * 2, xcol = (c0 + q16 * 8) * 2;

  // table of step t (absolute): entries 0..63 = the dZ positions, 64.. = the strip positions
  auto fill_table = [&](int t) __attribute__((always_inline)) {
    if (tid < W3_TAB) {
      if (!S2) {
        const int q = tid < 64 ? t * PX + tid : t * PX + (tid - 64) + xshift;
        bool ok = q >= 0 && q < g.TP;
        const uint32_t qc = ok ? (uint32_t)q : 0u;
        const uint32_t R = __umulhi(qc, g.mP);             // padded row = b * H + y
        const uint32_t xp = qc - R * g.P;
        const uint32_t y = R - __umulhi(R, g.mH) * g.H;
        ok = ok && (xp - 1u) < (uint32_t)g.W;
        if (tid >= 64) ok = ok && (int)y >= ylo && (int)y <= yhi;
        tab[(t % 3) * W3_TAB + tid] = ok ? (0x80000000u | (R * g.W + xp - 1u)) : 0u;
      } else {
        // entries 0..63: dZ positions; 64..131: even strip (row j = position j of the step); 132..199: odd strip (row j = position j - 1)
        const int odd = tid >= 64 + W3_XROWS;
        const int j = tid < 64 ? tid : tid - 64 - odd * W3_XROWS;
        const int q = t * PX + j - odd;
        bool ok = q >= 0 && q < g.TP && (tid < 64 || j < 64 + odd);
        const uint32_t qc = ok ? (uint32_t)q : 0u;
        const uint32_t R = __umulhi(qc, g.mP);             // padded output row = b * Ho + oy
        const uint32_t xp = qc - R * g.P;
        ok = ok && xp >= 1u;
        const uint32_t bimg = __umulhi(R, g.mH);
        const uint32_t oy = R - bimg * g.Ho;
        uint32_t pix = R * g.Wo + xp - 1u;
        if (tid >= 64) {
          const int iy = 2 * (int)oy + dy - 1, ix = 2 * ((int)xp - 1) + odd;
          ok = ok && iy >= 0 && iy < g.H && ix < g.W;
          pix = (bimg * g.H + (uint32_t)(iy < 0 ? 0 : iy)) * g.W + (uint32_t)(ix < 0 ? 0 : ix);
        }
        tab[(t % 3) * W3_TAB + tid] = ok ? (0x80000000u | pix) : 0u;
      }
    }
  };

  struct Regs {
    u32x4 g[2], g2[2];
  };
  Regs set0, set1;
  u32x4 xreg[S2 ? 6 : 3];              // [2]: strip rows 64, 65 (threads 0..31); stride 2: [3..5] the odd strip
  auto issue_g = [&](Regs& R, int i, int t) __attribute__((always_inline)) {
    const uint32_t e = tab[(t % 3) * W3_TAB + r32 + 32 * i];
    const uint32_t m = (uint32_t)((int)e >> 31), pix = e & 0x7fffffffu;
    R.g[i] = ld16(Gb, (__umul24(pix, ldg2b) + gcol) & m & nmask);
    if (GPRO == CX_PRO_AFFINE2) R.g2[i] = ld16(G2b, (__umul24(pix, ldg22b) + gcol) & m & nmask);
  };
  // strip chunk i: rows r32 + 32 (i % 3) of strip i / 3 (stride 1: one strip, i = 0..2)
  auto issue_x = [&](int i, int t) __attribute__((always_inline)) {
    const uint32_t e = tab[(t % 3) * W3_TAB + 64 + (i / 3) * W3_XROWS + r32 + 32 * (i % 3)];
    const uint32_t m = (uint32_t)((int)e >> 31), pix = e & 0x7fffffffu;
    xreg[i] = ld16(Xb, (__umul24(pix, ldx2b) + xcol) & m);
  };
  // dword j of dZ chunk i of step t -> o[j]; after j == 3 the chunk is masked and written
  auto unit_g = [&](const Regs& R, int i, int j, u32x4& o, char* Gt, int t) __attribute__((always_inline)) {
    const uint32_t gw = R.g[i][j];
    if (GPRO == CX_PRO_NONE) {
      o[j] = gw;
    } else {
      const uint32_t y = R.g2[i][j];
      o[j] = packbf(fmaf(bf_lo(gw), ga[2 * j], fmaf(bf_lo(y), gb[2 * j], gc[2 * j])),
                    fmaf(bf_hi(gw), ga[2 * j + 1], fmaf(bf_hi(y), gb[2 * j + 1], gc[2 * j + 1])));
    }
    if (j == 3) {
      o &= (uint32_t)((int)tab[(t % 3) * W3_TAB + r32 + 32 * i] >> 31) & nmask;
      *reinterpret_cast<u32x4*>(Gt + (r32 + 32 * i) * W3_GP + q16 * 16) = o;
    }
  };
  auto unit_x = [&](int i, int j, u32x4& o, char* Xt, int t, bool reissue) __attribute__((always_inline)) {
    const uint32_t x = xreg[i][j];
    if (XPRO == CX_PRO_NONE) {
      o[j] = x;
    } else {
      o[j] = relu_pk(packbf(fmaf(bf_lo(x), pa[2 * j], pb[2 * j]), fmaf(bf_hi(x), pa[2 * j + 1], pb[2 * j + 1])));
    }
    if (j == 3) {
      const int row = (i / 3) * W3_XROWS + r32 + 32 * (i % 3);
      o &= (uint32_t)((int)tab[(t % 3) * W3_TAB + 64 + row] >> 31);
      *reinterpret_cast<u32x4*>(Xt + row * W3_XP + q16 * 16) = o;
      if (reissue) issue_x(i, t + 1);
    }
  };
  // the whole third chunk of a strip (rows 64, 65) of threads 0..31 (stride 2: only the odd strip has a row 64)
  auto extra_x = [&](char* Xt, int t, bool reissue) __attribute__((always_inline)) {
    if (tid < 32) {
      u32x4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) unit_x(S2 ? 5 : 2, j, o, Xt, t, reissue);
    }
  };

  f32x16 acc[2][3];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][t][r] = 0.f;

  // One step = 24 MFMAs of LDS image `bufc` (step t); in fenced packets between them: the table of step t + 3, the dZ requests of
  // step t + 2 into Rn, the prologue + LDS stores of step t + 1 from Rc / the strip registers into image `bufn`, each strip chunk
  // requested again for step t + 2 as soon as it is stored.
  auto step = [&](Regs& Rn, const Regs& Rc, int bufc, int bufn, int t, auto IssueC, auto StageC) __attribute__((always_inline)) {
    constexpr bool ISSUE = decltype(IssueC)::value, STAGE = decltype(StageC)::value;
    const char* Gt = smem + bufc * W3_STAGE;
    const char* Xt = Gt + PX * W3_GP;
    char* Gn = smem + bufn * W3_STAGE;
    char* Xn = Gn + PX * W3_GP;
    u32x4 o;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      bf16x8 af[2], bfr[3];
      af[0] = tr_frag(Gt, W3_GP, kk * 16, wa * 64, lane);
      af[1] = tr_frag(Gt, W3_GP, kk * 16, wa * 64 + 32, lane);
      if (!S2) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) bfr[dx] = tr_frag(Xt, W3_XP, kk * 16 + dx, wb * 32, lane);
      } else {
        bfr[0] = tr_frag(Xt + W3_XROWS * W3_XP, W3_XP, kk * 16, wb * 32, lane);           // odd strip, one position to the left
        bfr[1] = tr_frag(Xt, W3_XP, kk * 16, wb * 32, lane);                              // even strip
        bfr[2] = tr_frag(Xt + W3_XROWS * W3_XP, W3_XP, kk * 16 + 1, wb * 32, lane);       // odd strip
      }
#pragma unroll
      for (int q = 0; q < 6; ++q) {
        acc[q & 1][q >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[q & 1], bfr[q >> 1], acc[q & 1][q >> 1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        const int slot = kk * 6 + q;                       // 24 slots
        if (slot == 0) fill_table(t + 3);
        if (ISSUE && (slot == 1 || slot == 2)) issue_g(Rn, slot - 1, t + 2);
        if (STAGE) {
          // 16 dword units (8 dZ, 8 strip) on slots 3..18, the extra strip chunk on slot 20
          if (slot >= 3 && slot < 11) unit_g(Rc, (slot - 3) >> 2, (slot - 3) & 3, o, Gn, t + 1);
          if (slot >= 11 && slot < 19) unit_x((slot - 11) >> 2, (slot - 11) & 3, o, Xn, t + 1, ISSUE);
          if (S2) {                                          // the odd strip's two full chunks on the four free slots, two units each
            const int f = slot == 19 ? 0 : slot == 21 ? 1 : slot == 22 ? 2 : slot == 23 ? 3 : -1;
            if (f >= 0) {
              unit_x(3 + (f >> 1), 2 * (f & 1), o, Xn, t + 1, ISSUE);
              unit_x(3 + (f >> 1), 2 * (f & 1) + 1, o, Xn, t + 1, ISSUE);
            }
          }
          if (slot == 20) extra_x(Xn, t + 1, ISSUE);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  using T = std::true_type;
  using F = std::false_type;

  if (nsteps > 0) {
    fill_table(step0);
    fill_table(step0 + 1);
    fill_table(step0 + 2);
    __syncthreads();
    issue_g(set0, 0, step0);
    issue_g(set0, 1, step0);
    issue_x(0, step0);
    issue_x(1, step0);
    if (S2) {
      issue_x(3, step0);
      issue_x(4, step0);
    }
    if (tid < 32) issue_x(S2 ? 5 : 2, step0);
    if (nsteps > 1) {
      issue_g(set1, 0, step0 + 1);
      issue_g(set1, 1, step0 + 1);
    }
    {
      u32x4 o;
#pragma unroll
      for (int u = 0; u < 8; ++u) unit_g(set0, u >> 2, u & 3, o, smem, step0);
#pragma unroll
      for (int u = 0; u < 8; ++u) unit_x(u >> 2, u & 3, o, smem + PX * W3_GP, step0, nsteps > 1);
      if (S2) {
#pragma unroll
        for (int u = 0; u < 8; ++u) unit_x(3 + (u >> 2), u & 3, o, smem + PX * W3_GP, step0, nsteps > 1);
      }
      extra_x(smem + PX * W3_GP, step0, nsteps > 1);
    }
    __syncthreads();

    int s = 0;
    for (; s + 3 < nsteps; s += 2) {
      step(set0, set1, 0, 1, step0 + s, T{}, T{});      // multiplies step s, requests s+2, stages s+1
      __syncthreads();
      step(set1, set0, 1, 0, step0 + s + 1, T{}, T{});
      __syncthreads();
    }
    const int left = nsteps - s;
    if (left == 3) {
      step(set0, set1, 0, 1, step0 + s, T{}, T{});
      __syncthreads();
      step(set1, set0, 1, 0, step0 + s + 1, F{}, T{});
      __syncthreads();
      step(set0, set1, 0, 1, step0 + s + 2, F{}, F{});
    } else if (left == 2) {
      step(set0, set1, 0, 1, step0 + s, F{}, T{});
      __syncthreads();
      step(set1, set0, 1, 0, step0 + s + 1, F{}, F{});
    } else {
      step(set0, set1, 0, 1, step0 + s, F{}, F{});
    }
  }

  // ---- partial tile -> slab [split][tap][n][c] (c contiguous: the OIHW positions of one tap are 36 bytes apart, which as
  // atomics costs nine memory-side requests per useful one - measured 400 us per launch); dw3_reduce_kernel adds the slabs in split
  // order and transposes to OIHW
  const int lrow = lane & 31, lh = lane >> 5;
  const size_t nk = (size_t)p.N * p.K;
  const int c = c0 + wb * 32 + lrow;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = n0 + wa * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (n < p.N) slab[((size_t)split * 9 + dy * 3 + dx) * nk + (size_t)n * p.K + c] = acc[i][dx][r];
      }
}

// dw[(n*K + c)*9 + tap] += sum over the splits (in split order) of slab[split][tap][n][c]: one thread per (n, c)
__global__ __launch_bounds__(256) void dw3_reduce_kernel(float* __restrict__ dw, const float* __restrict__ slab, size_t nk, int splits) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nk) return;
  float a[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) a[t] = 0.f;
  for (int s = 0; s < splits; ++s) {
#pragma unroll
    for (int t = 0; t < 9; ++t) a[t] += slab[((size_t)s * 9 + t) * nk + i];
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) dw[i * 9 + t] += a[t];
}

static inline int w3_splits(const CxWgrad& p);

template <int GPRO, int XPRO, bool S2>
int launch3(const CxWgrad& p, hipStream_t st) {
  W3Geo g;
  g.B = p.B, g.H = p.H, g.W = p.W;
  g.Ho = S2 ? p.Ho : p.H, g.Wo = S2 ? p.Wo : p.W;
  g.P = S2 ? p.Wo + 1 : p.W + 2;
  g.TP = p.B * g.Ho * g.P;
  g.mP = 0xffffffffu / (uint32_t)g.P + 1u;
  g.mH = 0xffffffffu / (uint32_t)g.Ho + 1u;
  const int c_tiles = p.K / 128, n_tiles = (p.N + 127) / 128;
  const int total_steps = (g.TP + PX - 1) / PX;
  const int splits = w3_splits(p);
  const int sps = (total_steps + splits - 1) / splits;
  const size_t smem = 2 * W3L<S2>::STAGE + 3 * W3L<S2>::TAB * 4;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3_kernel<GPRO, XPRO, S2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr = true;
  }
  const size_t nk = (size_t)p.N * p.K;
  float* slab = p.scratch;                               // (the dispatcher checked its size)
  // the sum below runs at once, behind the kernel on the same stream: in stream order the region is free again when this call
  // returns, so a deferring caller (ops._wgrad_used) must not advance its arena past it -- and must not see the previous launch's figure
  cx_tl_slab_floats_v = 0;
  CX_KTAG("wgrad3_kernel<%d, %d, %s>", GPRO, XPRO, S2 ? "true" : "false");
  hipLaunchKernelGGL((wgrad3_kernel<GPRO, XPRO, S2>), dim3(c_tiles * n_tiles * 3 * splits), dim3(512), smem, st, p, g, c_tiles, n_tiles, total_steps,
                     sps, slab);
  if (const int e = launch_status()) return e;
  hipLaunchKernelGGL(dw3_reduce_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, st, p.dw, slab, nk, splits);
  return launch_status();
}

// pixel-range splits of the 3x3 kernel (shared by the dispatcher's workspace check and the launcher)
static inline int w3_splits(const CxWgrad& p) {
  const int total_steps = ((p.stride == 2 ? p.B * p.Ho * (p.Wo + 1) : p.B * p.H * (p.W + 2)) + PX - 1) / PX;
  int splits = p.splits > 0 ? p.splits : 256 / ((p.K / 128) * ((p.N + 127) / 128) * 3);
  if (splits < 1) splits = 1;
  if (splits > total_steps) splits = total_steps;
  const int sps = (total_steps + splits - 1) / splits;
  return (total_steps + sps - 1) / sps;
}

template <int WA, int WB, int GPRO, int XPRO>
int launch(const CxWgrad& p, hipStream_t st, int wgs_target) {
  using G = WG<WA, WB>;
  const long long M = (long long)p.B * p.Ho * p.Wo;
  const int c_tiles = (p.K + G::TC - 1) / G::TC, n_tiles = (p.N + G::TN - 1) / G::TN;
  const int total_steps = (int)(M / PX);
  int splits = p.splits > 0 ? p.splits : wgs_target / (c_tiles * n_tiles);
  if (splits < 1) splits = 1;
  if (splits > total_steps) splits = total_steps;
  const int sps = (total_steps + splits - 1) / splits;
  splits = (total_steps + sps - 1) / sps;
  const size_t smem = 2 * G::STAGE;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_mm_kernel<WA, WB, GPRO, XPRO>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)smem);
    attr = true;
  }
  CxWgrad q = p;
#ifdef CX_DIAG_TIMING   // timing-only ablation (results wrong): never in the product library
  static const bool nostore = cx_diag_set("CX_WGRAD_MM_NOSTORE");
  if (nostore) q.splits = -7;
#endif
  const size_t total = (size_t)p.N * p.K;
  float* slab = dw_slab(p.scratch, p.scratch_floats, splits, (long long)total);
  CX_KTAG("wgrad_mm_kernel<%d, %d, %d, %d>", WA, WB, GPRO, XPRO);
  hipLaunchKernelGGL((wgrad_mm_kernel<WA, WB, GPRO, XPRO>), dim3(c_tiles * n_tiles * splits), dim3(G::NT), smem, st, q, c_tiles, n_tiles,
                     total_steps, sps, slab);
  if (const int e = launch_status()) return e;
  return slab ? cx_dw_reduce(p.dw, slab, total, splits, st) : 0;
}

template <int GPRO, int XPRO>
int launch_form(const CxWgrad& p, hipStream_t st, int form) {
  if (form == 3) return launch<2, 4, GPRO, XPRO>(p, st, 256);      // 128 x 256, one 512-thread workgroup per CU
  if (form == 2) return launch<4, 2, GPRO, XPRO>(p, st, 256);      // 256 x 128
  return launch<2, 2, GPRO, XPRO>(p, st, 512);                     // 128 x 128, two 256-thread workgroups per CU
}

}  // namespace

// CxWgrad.kernel_hint (ABI 10) pins the kernel choice per call for tests and micro-benchmarks (on = 0: conv_wgrad.hip's kernels; form 1 | 2 | 3).

// Called by cx_conv_wgrad after validation (bf16, MODE_CONV).
int cx_try_wgrad_mm(const CxWgrad& p, hipStream_t st, bool* handled) {
  *handled = false;
  static const int env_on0 = cx_diag_int("CX_WGRAD_MM", 1);
  static const int env_form0 = cx_diag_int("CX_WGRAD_MM_FORM", 0);
  const int g_wm_on = (p.kernel_hint & 0xff) - 1, g_wm_form = ((p.kernel_hint >> 8) & 0xff) - 1;      // -1: not pinned
  const int on = g_wm_on >= 0 ? g_wm_on : env_on0;
  const int env_form = g_wm_form >= 0 ? g_wm_form : env_form0;
  if (!on || p.mode != CX_MODE_CONV) return 0;
  if (p.kh == 3 && p.kw == 3 && (p.stride == 1 || p.stride == 2) && p.pad == 1) {
    // the bottleneck 3x3 layers (wgrad3_kernel); padded positions, pixel indices and byte offsets must fit their fields
    static const int env3 = cx_diag_int("CX_WGRAD3", 1);
    const int on3 = g_wm_form == 0 ? 0 : env3;          // kernel_hint = CX_KERNEL_HINT(1, 0): the strip kernel
    // N: any multiple of 8 from 96 up (partial last tile; CX_WGRAD3_MIN_N, 0 = multiples of 128 only)
    static const int min_n3 = cx_diag_int("CX_WGRAD3_MIN_N", 96);
    if (!on3 || (p.N % 8) || ((p.N % 128) && (min_n3 <= 0 || p.N < min_n3)) || (p.K % 128) || p.W < 2 || p.H < 2) return 0;
    // its partial tiles leave through the slab workspace only (see the kernel's epilogue): without one the strip kernel runs
    if (!p.scratch || (long long)w3_splits(p) * 9 * p.N * p.K > p.scratch_floats) return 0;
    const bool s2 = p.stride == 2;
    if (s2 && (p.Ho != (p.H - 1) / 2 + 1 || p.Wo != (p.W - 1) / 2 + 1)) return 0;
    // measured at 128 images (scratch/bench_w3s2.py): 226 / 178 / 212 us against the generic kernel's 374 / 378 / 338 on the three
    // attention-augmented transitions (K = 256 / 512 / 1024), 121 / 125 against 228 on ResNet152's layer3.0 / layer4.0 (K = 256 / 512),
    // but 255 against 241 on layer2.0 (K = 128 on the 80x80 map: one channel tile, the strips of the large map staged by 3 x 85
    // workgroups each) -- taken from two channel tiles up unless the call pins the form
    if (s2 && p.K < 256 && g_wm_form != 3) return 0;
    const unsigned long long pw = s2 ? p.Wo + 1 : p.W + 2, ph = s2 ? p.Ho : p.H;
    const unsigned long long tp = (unsigned long long)p.B * ph * pw;
    if (tp * pw >= (1ull << 32) || (unsigned long long)p.B * ph * ph >= (1ull << 32) || tp >= (1ull << 24)) return 0;
    int ldm = p.ldg > p.ldx ? p.ldg : p.ldx;
    if (p.g_prologue == CX_PRO_AFFINE2 && p.ldg2 > ldm) ldm = p.ldg2;
    if (ldm >= (1 << 23) || (unsigned long long)p.B * p.H * p.W * ldm * 2 >= (1ull << 32)) return 0;
    const bool g2_ = p.g_prologue == CX_PRO_AFFINE2;
    if (p.g_prologue != CX_PRO_NONE && !g2_) return 0;
    if (p.x_prologue != CX_PRO_NONE && p.x_prologue != CX_PRO_AFFINE_RELU) return 0;
    *handled = true;
    if (s2) {
      if (p.x_prologue == CX_PRO_AFFINE_RELU)
        return g2_ ? launch3<CX_PRO_AFFINE2, CX_PRO_AFFINE_RELU, true>(p, st) : launch3<CX_PRO_NONE, CX_PRO_AFFINE_RELU, true>(p, st);
      return g2_ ? launch3<CX_PRO_AFFINE2, CX_PRO_NONE, true>(p, st) : launch3<CX_PRO_NONE, CX_PRO_NONE, true>(p, st);
    }
    if (p.x_prologue == CX_PRO_AFFINE_RELU)
      return g2_ ? launch3<CX_PRO_AFFINE2, CX_PRO_AFFINE_RELU, false>(p, st) : launch3<CX_PRO_NONE, CX_PRO_AFFINE_RELU, false>(p, st);
    return g2_ ? launch3<CX_PRO_AFFINE2, CX_PRO_NONE, false>(p, st) : launch3<CX_PRO_NONE, CX_PRO_NONE, false>(p, st);
  }
  if (p.kh != 1 || p.kw != 1 || p.stride != 1 || p.pad != 0) return 0;
  if ((p.N % 8) || (p.K % 8) || p.K < 64) return 0;
  // partial last N tile (dZ chunks past N staged as zeros): EfficientNet-B4 +1.4 % with every width taken (CX_WGRAD_MM_MIN_N = 24,
  // 96, 256: +1.4 / +1.2 / +0.9 %; 0 = multiples of 128 only)
  static const int min_n = cx_diag_int("CX_WGRAD_MM_MIN_N", 24);
  if ((p.N % 128) && (min_n <= 0 || p.N < min_n)) return 0;
  const long long M = (long long)p.B * p.Ho * p.Wo;
  if (M % PX) return 0;
  const unsigned long long ldmax = (unsigned long long)(p.ldg > p.ldx ? p.ldg : p.ldx);
  if ((unsigned long long)M * (ldmax > (unsigned long long)p.ldg2 ? ldmax : (unsigned long long)p.ldg2) * 2 >= (1ull << 32)) return 0;
  const bool g2 = p.g_prologue == CX_PRO_AFFINE2;
  if (p.g_prologue != CX_PRO_NONE && !g2) return 0;
  if (p.x_prologue != CX_PRO_NONE && p.x_prologue != CX_PRO_AFFINE_RELU) return 0;
  int form = 3;                   // measured on the ResNet152 shapes (scratch/bench_wmm.py): 128 x 256 wins at every K >= 64
  if (env_form) form = env_form;
  if (form == 2 && (p.N % 256)) form = 1;
  if (form == 2 && (p.N % 128)) form = 3;
  *handled = true;
  if (p.x_prologue == CX_PRO_AFFINE_RELU)
    return g2 ? launch_form<CX_PRO_AFFINE2, CX_PRO_AFFINE_RELU>(p, st, form) : launch_form<CX_PRO_NONE, CX_PRO_AFFINE_RELU>(p, st, form);
  return g2 ? launch_form<CX_PRO_AFFINE2, CX_PRO_NONE>(p, st, form) : launch_form<CX_PRO_NONE, CX_PRO_NONE>(p, st, form);
}
