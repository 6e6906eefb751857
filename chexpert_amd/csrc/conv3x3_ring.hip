// 3x3 stride-1 pad-1 forward of the dense layers (K = 128 -> N = 32), second generation: one 512-thread workgroup per CU,
// the nine 32x128 weight slices AND the input-row ring both live in LDS, every wave computes whole 32-pixel sub-tiles.
//
// The first strip kernel (conv3x3_strip.hip) gave each of three waves one kernel row and met the partial sums in LDS:
// 24 MFMAs, then two barriers and a 128-thread reduction per sub-tile -- the matrix pipe idled two thirds of the time
// (1.46 TB/s on 80x80 maps).  Here a sub-tile is 72 back-to-back MFMAs of one wave, no cross-wave reduction:
//   * operands swapped (A = weight slice [32 out][16 k], B = ring pixels [16 k][32 px]): both fragments are one
//     conflict-free ds_read_b128 (272-B pitches); an accumulator lane owns one pixel and, after v_permlane32_swap,
//     8 consecutive output channels -> two 16-B stores per lane, no LDS transposition, no scratch;
//   * per-lane channel sums stay in registers for the whole workgroup (N = 32: 32 registers), reduced once at the end;
//   * ring layout as before: padded-flat rows of W+2 pixels with zero pad columns, tap (dy,dx) of flat pixel m is ring
//     pixel m + dy*(W+2) + dx, two mirror pixels past the end cover the wrap;
//   * the R new rows of the next step are requested before the MFMAs of the current one (<= 7 x 16 B per thread) and are
//     normalised (BN + ReLU) while they are written to the ring; two barriers per step.
#include <cstdlib>
#include <type_traits>
#include "common.h"

namespace {

constexpr int XP = 272;                  // bytes per ring pixel / weight row: 128 bf16 + 16 pad
constexpr int NT = 512;                  // threads
constexpr int W_ROWS = 9 * 32;
constexpr int W_BYTES = W_ROWS * XP;
constexpr int RING_PX_MAX = (160 * 1024 - W_BYTES - 1024 - 256) / XP;     // 309

struct RingGeo {
  int B, H, W, P, R, Q;         // P = W+2, Q = (R+2)*P ring pixels
  int Wt, ntx;                  // forward: column tiles of Wt pixels (P = Wt+2, ntx per row; B counts tiles = images * ntx)
  int spi;                      // steps per image = ceil(H/R)
  int steps_per_wg;
  unsigned mP;                  // ceil(2^32 / P): m / P = umulhi(m, mP) for the flat pixel indices of a step (m < 2^16)
  unsigned mSpi;                // ceil(2^32 / spi): step index / spi (exact while steps < 2^32 / spi: the launchers check)
  int ntx_shift;                // ntx is 1 or 2
};

__device__ __forceinline__ int wrapq(int v, int q) { return v >= q ? v - q : v; }
// step index / steps per image by multiply-high; spi == 1 (a whole image per step: small maps) has no 32-bit magic number (2^32)
__device__ __forceinline__ int div_spi(const RingGeo& g, int v) { return g.spi == 1 ? v : (int)__umulhi((unsigned)v, g.mSpi); }

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float half_sum(float v) {      // sum over the 32 lanes of each wave half
  v = dpp_add<0xB1>(v);
  v = dpp_add<0x4E>(v);
  v = dpp_add<0x141>(v);
  v = dpp_add<0x140>(v);
  return v + __shfl_xor(v, 16);
}

#ifdef CX_RING_STAMPS
// diagnostic build (scratch/stamps_ring.py): s_memtime sums per phase of the input-gradient kernel, wave 0 of each workgroup
__device__ unsigned long long ring_stamps[1024 * 8];
__device__ unsigned long long ring_fwd_stamps[1024 * 8];
__device__ __forceinline__ unsigned long long rstamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define RSTAMP(i) { const unsigned long long t_ = rstamp(); st_acc[i] += t_ - st_prev; st_prev = t_; }
#else
#define RSTAMP(i)
#endif
template <int NCH, int DEPTH>
__global__ __launch_bounds__(NT, 1) void conv3x3_ring_fwd_kernel(const bf16* __restrict__ x, int ldx, const float* __restrict__ sc,
                                                                const float* __restrict__ sh, const bf16* __restrict__ wpk,
                                                                bf16* __restrict__ y, int ldy, float* stat_sum, float* stat_sq,
                                                                int stat_replicas, int stat_rstride, int stat_det, const RingGeo g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* coef = reinterpret_cast<float*>(smem);                // [2][128]
  char* wl = smem + 1024 + 256;                                // [9*32][272 B]
  char* ring = wl + W_BYTES;                                   // [(Q+2)][272 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 31, lh = lane >> 5;
  const int P = g.P, R = g.R, Q = g.Q, W = g.W, H = g.H, Wt = g.Wt, ntx = g.ntx;       // W: image width, Wt = P - 2: column-tile width

  // ---- one-time setup: weights [tap][n][k] -> LDS rows of 272 B, ring zeroed (pad columns stay zero), coefficients
  for (int i = tid; i < W_ROWS * 16; i += NT) {
    const int row = i >> 4, c = i & 15;
    *reinterpret_cast<uint4*>(wl + row * XP + c * 16) = *reinterpret_cast<const uint4*>(wpk + (size_t)row * 128 + c * 8);
  }
  for (int i = tid; i < (Q + 2) * (XP / 16); i += NT) reinterpret_cast<uint4*>(ring)[i] = make_uint4(0, 0, 0, 0);
  __syncthreads();

  const int total_steps = g.B * g.spi;
  const int u0 = blockIdx.x * g.steps_per_wg;
  const int u1 = min(total_steps, u0 + g.steps_per_wg);
  const int chunks_per_row = P * 16;          // the two halo columns of a column tile are loaded (zeros outside the image)

  // DEPTH register sets of new rows: the rows of steps u+1 .. u+DEPTH are in flight while step u is multiplied (at W = 80 a
  // step is one image row and shorter than an HBM round trip under load)
  uint4 pre[DEPTH][NCH];
  bool pv[DEPTH][NCH];
  int base_row = 0;           // image row held by ring slot 0: slot(y) = (y - base_row) mod (R+2)
  // chunk slot i of this thread: chunk id tid + NT*i -> (row inside the group of new rows, pixel, 16-B channel chunk).  NT and
  // 16*W are multiples of 16, so the channel chunk is tid & 15 for every slot: its BN scale/shift stay in registers
  int crow[NCH], cpx[NCH];
  const int cc8 = tid & 15;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int cid = tid + NT * i;
    crow[i] = cid / chunks_per_row;
    cpx[i] = (cid - crow[i] * chunks_per_row) >> 4;
  }
  float csc[8], csh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { csc[j] = sc[cc8 * 8 + j]; csh[j] = sh[cc8 * 8 + j]; }
  // retire these loads here: a load still pending at the loop header makes every first use inside the loop a vmcnt(0), which
  // drains the DEPTH row sets in flight each step
#pragma unroll
  for (int j = 0; j < 8; ++j) asm volatile("" ::"v"(csc[j]), "v"(csh[j]));
  // rows [y0, y0+n) of column tile bv = image * ntx + tile -> registers (ring position p of a row is image column x0 - 1 + p);
  // unconditional loads on clamped addresses, validity applied when staged
  // A chunk's byte offset inside the row group is a constant of the thread (voff); the group's own offset is wave-uniform.  Chunks
  // outside the image (or past the n rows asked for) read offset 0 and are zeroed when staged.  (Per chunk: one add, two range tests
  // and a select -- clamping row and column and rebuilding the 64-bit address was ~22 vector instructions per chunk, 7 chunks per
  // step, in a kernel that is bound by vector-instruction issue.)  32-bit offsets: the launcher checks the tensor is < 4 GB.
  uint32_t voff[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) voff[i] = ((uint32_t)(crow[i] * W + cpx[i]) * (uint32_t)ldx + (uint32_t)cc8 * 8u) * 2u;
  const char* __restrict__ xb = reinterpret_cast<const char*>(x);
  auto issue_rows = [&](uint4 (&pre)[NCH], bool (&pv)[NCH], int bv, int y0, int n) __attribute__((always_inline)) {
    const int b = bv >> g.ntx_shift, x0 = (bv - (b << g.ntx_shift)) * Wt - 1;
    const uint32_t sbase = (uint32_t)((b * H + y0) * W + x0) * (uint32_t)ldx * 2u;      // (wraps for y0 = -1 / x0 = -1: only valid chunks use it)
    const uint32_t r_lo = (uint32_t)max(-y0, 0), r_n = (uint32_t)max(min(n, H - y0), 0) - r_lo;      // rows  [r_lo, r_lo + r_n)
    const uint32_t c_lo = (uint32_t)max(-x0, 0), c_n = (uint32_t)max(W - x0, 0) - c_lo;              // pixels [c_lo, c_lo + c_n)
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      pv[i] = ((uint32_t)crow[i] - r_lo) < r_n && ((uint32_t)cpx[i] - c_lo) < c_n;
      pre[i] = *reinterpret_cast<const uint4*>(xb + (size_t)(pv[i] ? sbase + voff[i] : 0u));
    }
  };
  auto write_rows = [&](uint4 (&pre)[NCH], bool (&pv)[NCH], int y0, int n) __attribute__((always_inline)) {
    int slot_y0 = (y0 - base_row) % (R + 2);          // wave-uniform
    if (slot_y0 < 0) slot_y0 += R + 2;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      if (crow[i] < n) {
        int slot = slot_y0 + crow[i];                   // crow <= R: one conditional subtract wraps it
        if (slot >= R + 2) slot -= R + 2;
        U128 o;
        o.u = cx_affine_relu8(pre[i], csc, csh);
        { const unsigned keep = pv[i] ? 0xffffffffu : 0u; o.u.x &= keep; o.u.y &= keep; o.u.z &= keep; o.u.w &= keep; }   // no per-element branch
        const int pos = slot * P + cpx[i];
        *reinterpret_cast<uint4*>(ring + pos * XP + cc8 * 16) = o.u;
        if (pos < 2) *reinterpret_cast<uint4*>(ring + (Q + pos) * XP + cc8 * 16) = o.u;   // mirror of pixels 0,1
      }
    }
  };
  // new rows of step v (a continuation step of its image inside this workgroup's range), else a harmless clamped load
  int ulim = u1;              // end of the pass being walked (below)
  auto issue_step = [&](uint4 (&pre)[NCH], bool (&pv)[NCH], int v) __attribute__((always_inline)) {
    const int bv = div_spi(g, v), sv = v - bv * g.spi;
    const bool ok = v < ulim && sv != 0;
    issue_rows(pre, pv, ok ? bv : 0, ok ? sv * R + 1 : 0, ok ? R : 0);
  };

  float s1[2][8], s2[2][8];
#pragma unroll
  for (int cc = 0; cc < 2; ++cc)
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[cc][j] = s2[cc][j] = 0.f;
  const int nsub = (R * P + 31) / 32;
  const char* wbase = wl + lrow * XP + lh * 16;
#ifdef CX_RING_STAMPS
  unsigned long long st_acc[7] = {0, 0, 0, 0, 0, 0, 0}, st_prev = rstamp(), st_steps = 0;
#endif

  auto step = [&](int u, auto KI) __attribute__((always_inline)) {
    constexpr int k = decltype(KI)::value;
    uint4 (&cur)[NCH] = pre[k];
    bool (&cv)[NCH] = pv[k];
    const int bv = div_spi(g, u), yc = (u - bv * g.spi) * R;
    const int b = bv >> g.ntx_shift, x0 = (bv - (b << g.ntx_shift)) * Wt;
    RSTAMP(0)                                          // (between steps: restart of an image, loop overhead)
    write_rows(cur, cv, yc + 1, R);
    RSTAMP(1)
    __syncthreads();                                   // the window of this step is complete
    RSTAMP(2)
    issue_step(cur, cv, u + DEPTH);                    // in flight under the MFMAs of this and the next DEPTH-1 steps
    RSTAMP(3)
    int slot0 = (yc - 1 - base_row) % (R + 2);
    if (slot0 < 0) slot0 += R + 2;
    const int ws = slot0 * P;

    for (int s = wave; s < nsub; s += NT / 64) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const int m = s * 32 + lrow;
      const int pix = min(m, R * P - 1);
      // Software pipeline over the nine taps inside the wave: fragment reads run two groups ahead of the MFMAs (three
      // register sets), so the LDS latency hides under the matrix pipe even when this is the only computing
      // wave of its SIMD (W = 80: 3 sub-tiles per step for 8 waves).  Read-then-multiply per tap measured 130 cycles per
      // MFMA; the sched_barriers keep the scheduler from hoisting every read of the unrolled loop (it spilled).
      const char* ap0 = ring + wrapq(wrapq(ws + pix, Q), Q) * XP + lh * 16;
      const char* ap1 = ring + wrapq(wrapq(ws + pix + P, Q), Q) * XP + lh * 16;
      const char* ap2 = ring + wrapq(wrapq(ws + pix + 2 * P, Q), Q) * XP + lh * 16;
      // the 72 (tap, k-step) products in groups of 3, three register sets: the reads of group g+2 go out before the MFMAs of
      // group g (6 MFMAs = 192 cycles ahead; 72 fragment registers)
      constexpr int G = 3, NG = 72 / G;
      bf16x8 fa[3][G], fb[3][G];
      auto load_grp = [&](int gi, bf16x8 (&A)[G], bf16x8 (&B)[G]) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < G; ++j) {
          const int f = gi * G + j, t = f >> 3, ks = f & 7;
          const int dy = t / 3, dx = t - dy * 3;
          A[j] = *reinterpret_cast<const bf16x8*>(wbase + t * 32 * XP + ks * 32);
          B[j] = *reinterpret_cast<const bf16x8*>((dy == 0 ? ap0 : dy == 1 ? ap1 : ap2) + dx * XP + ks * 32);
        }
      };
      load_grp(0, fa[0], fb[0]);
      load_grp(1, fa[1], fb[1]);
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) {
        if (gi + 2 < NG) load_grp(gi + 2, fa[(gi + 2) % 3], fb[(gi + 2) % 3]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < G; ++j)
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[gi % 3][j], fb[gi % 3][j], acc, 0, 0, 0);   // D[row = out channel][col = pixel]
        __builtin_amdgcn_sched_barrier(0);
      }
#ifdef CX_RING_STAMPS
      asm volatile("" ::"v"(acc[0]));
#endif
      RSTAMP(4)
      const int oy = (int)__umulhi((unsigned)m, g.mP), ox = m - oy * P;      // m / P without the ~20-instruction division
      const int yy = yc + oy;
      const bool valid = m < R * P && ox < Wt && x0 + ox < W && yy < H;
      bf16* yrow = y + ((size_t)(b * H + (valid ? yy : 0)) * W + (valid ? x0 + ox : 0)) * ldy;
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        // registers 8cc..8cc+3 / 8cc+4..8cc+7: channels 16cc + 4*lh + e / 16cc + 8 + 4*lh + e; the swap of the upper half of
        // the first group with the lower half of the second leaves channels 8*(2cc+lh) .. +7 of this lane's pixel
        U128 o;
        float t[8];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[8 * cc + r4]), __float_as_uint(acc[8 * cc + 4 + r4]),
                                                           false, false);
          t[r4] = __uint_as_float(sw[0]);
          t[4 + r4] = __uint_as_float(sw[1]);
        }
        o.u = cx_pack8_stats(t, valid, true, s1[cc], s2[cc]);
        if (valid) *reinterpret_cast<uint4*>(yrow + 8 * (2 * cc + lh)) = o.u;
      }
      RSTAMP(5)
    }
    __syncthreads();                                   // every wave is done with the oldest rows of the ring
    RSTAMP(6)
#ifdef CX_RING_STAMPS
    ++st_steps;
#endif
  };

  // One image (or the part of it in this workgroup's range) at a time: the window rows yc-1, yc of its first step are built
  // synchronously and the pipeline is refilled outside the step loop.
  // The range is walked from a per-workgroup offset and wraps (two passes): with every workgroup starting at row 0 of its
  // own image, all 256 read addresses that differ by multiples of the image size at the same moment.
  const int rot = u1 - u0 > 1 ? (int)((blockIdx.x * 37u) % (unsigned)(u1 - u0)) : 0;
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
  ulim = pass == 0 ? u1 : u0 + rot;
  for (int ui = pass == 0 ? u0 + rot : u0; ui < ulim;) {
    const int b = div_spi(g, ui), yc = (ui - b * g.spi) * R;
    const int ue = min(ulim, (b + 1) * g.spi);
    base_row = yc - 1;
    auto sync_rows = [&](int y0) __attribute__((always_inline)) {
      issue_rows(pre[0], pv[0], b, y0, 1);
      write_rows(pre[0], pv[0], y0, 1);
    };
    sync_rows(yc - 1);
    sync_rows(yc);
    issue_rows(pre[0], pv[0], b, yc + 1, R);
#pragma unroll
    for (int d = 1; d < DEPTH; ++d) issue_step(pre[d], pv[d], ui + d);
    for (int u = ui; u < ue; u += DEPTH) {
      step(u, std::integral_constant<int, 0>());
      if (DEPTH > 1) { if (u + 1 >= ue) break; step(u + 1, std::integral_constant<int, 1 % DEPTH>()); }
      if (DEPTH > 2) { if (u + 2 >= ue) break; step(u + 2, std::integral_constant<int, 2 % DEPTH>()); }
    }
    ui = ue;
  }
  }

#ifdef CX_RING_STAMPS
  if (tid == 0 && blockIdx.x < 1024) {
    for (int i = 0; i < 7; ++i) ring_fwd_stamps[blockIdx.x * 8 + i] = st_acc[i];
    ring_fwd_stamps[blockIdx.x * 8 + 7] = st_steps;
  }
#endif
  if (stat_sum) {
    float* scratch = reinterpret_cast<float*>(wl);               // the weight slices are no longer read
    wg_stat_begin<NT / 64>(scratch, 32, tid, NT);
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float a = half_sum(s1[cc][j]);
        const float c = half_sum(s2[cc][j]);
        if (lrow == 8 * cc + j) { t1 = a; t2 = c; }
      }
    if (lrow < 16) {
      const int n = 8 * (2 * (lrow >> 3) + lh) + (lrow & 7);
      wg_stat_put(scratch, 32, wave, n, t1, t2);
    }
    wg_stat_end<NT / 64>(scratch, 32, tid, NT, stat_sum, stat_sq, stat_det, (int)blockIdx.x, stat_replicas, stat_rstride, 0, 32);
  }
}

template <int NCH, int DEPTH>
int launch_ring(const CxConv& p, hipStream_t st, const RingGeo& g) {
  const size_t smem = 1024 + 256 + W_BYTES + (size_t)(g.Q + 2) * XP;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_ring_fwd_kernel<NCH, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    attr = true;
  }
  const int total = g.B * g.spi;
  const int grid = (total + g.steps_per_wg - 1) / g.steps_per_wg;
  if (const int e = stat_rows_check(p, grid)) return e;
  CX_KTAG("conv3x3_ring_fwd_kernel<%d, %d>", NCH, DEPTH);
  hipLaunchKernelGGL((conv3x3_ring_fwd_kernel<NCH, DEPTH>), dim3(grid), dim3(NT), smem, st, (const bf16*)p.x, p.ldx, p.pa, p.pb,
                     (const bf16*)p.w, (bf16*)p.y, p.ldy, p.stat_sum, p.stat_sq, p.stat_replicas, p.stat_rstride, p.stat_det, g);
  return launch_status();
}


// ------------------------------------------------------------------------------------------------ input gradient
// dz2[m][0:128] = mask(y1[m]) * sum_taps dY2[m @ tap][0:32] . Wt[tap][0:128][0:32]     (K = 9 x 32, N = 128, CX_EPI_MASK)
// Same ring design.  The ring holds the 32-channel gradient slice after the deferred BN correction (AFFINE2 of the gradient
// and activation slices), 80 B per pixel, so a step covers up to 256 flat pixels = 8 sub-tiles; a work item is
// (sub-tile, half of the 128 output channels) = 36 MFMAs, and wave w always takes channel half w & 1, so the 64 per-lane
// S1/S2 accumulators keep one meaning for the whole workgroup.
constexpr int GP = 80;                   // bytes per ring pixel / weight row: 32 bf16 + 16 pad (5 slots: conflict-free)
constexpr int DW_ROWS = 9 * 128;
constexpr int DW_BYTES = DW_ROWS * GP;
constexpr int EC_BYTES = 5 * 128 * 4 + 3 * 32 * 4;   // epilogue vectors [5][128] + AFFINE2 vectors [3][32]
constexpr int MAX_ITEMS = 2;             // work items per wave and step

template <int NCH>
__global__ __launch_bounds__(NT, 1) void conv3x3_ring_dgrad_kernel(
    const bf16* __restrict__ gsl, int ldg, const bf16* __restrict__ g2, int ldg2, const float* __restrict__ ga,
    const float* __restrict__ gb, const float* __restrict__ gc, const bf16* __restrict__ wpk, const bf16* __restrict__ ex, int ldex,
    const float* __restrict__ e_sc, const float* __restrict__ e_sh, const float* __restrict__ e_mu, const float* __restrict__ e_r,
    const float* __restrict__ e_scale, bf16* __restrict__ y, int ldy, float* S1, float* S2, int stat_replicas, int stat_rstride,
    int stat_det, bf16* __restrict__ po, int ldpo, const RingGeo g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ecoef = reinterpret_cast<float*>(smem);               // e_sc, e_sh, e_mu, e_r, e_scale [128] each
  char* wl = smem + EC_BYTES;                                  // [9*128][80 B]
  char* ring = wl + DW_BYTES;                                  // [(Q+2)][80 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 31, lh = lane >> 5;
  const int h2 = wave & 1;                                     // this wave's half of the output channels
  const int P = g.P, R = g.R, Q = g.Q, W = g.W, H = g.H;

  for (int i = tid; i < DW_ROWS * 4; i += NT) {
    const int row = i >> 2, c = i & 3;
    *reinterpret_cast<uint4*>(wl + row * GP + c * 16) = *reinterpret_cast<const uint4*>(wpk + (size_t)row * 32 + c * 8);
  }
  for (int i = tid; i < (Q + 2) * (GP / 16); i += NT) reinterpret_cast<uint4*>(ring)[i] = make_uint4(0, 0, 0, 0);
  if (tid < 128) {
    ecoef[tid] = e_sc[tid]; ecoef[128 + tid] = e_sh[tid]; ecoef[256 + tid] = e_mu[tid]; ecoef[384 + tid] = e_r[tid];
    ecoef[512 + tid] = e_scale[tid];
  }
  if (tid < 32) { ecoef[640 + tid] = ga[tid]; ecoef[672 + tid] = gb[tid]; ecoef[704 + tid] = gc[tid]; }
  __syncthreads();

  const int total_steps = g.B * g.spi;
  const int u0 = blockIdx.x * g.steps_per_wg;
  const int u1 = min(total_steps, u0 + g.steps_per_wg);
  const int cpr = W * 4;

  // new gradient rows: chunk slot i of this thread = chunk id tid + NT*i -> (row, pixel); the channel chunk is tid & 3 for
  // every slot (NT and 4*W are multiples of 4), so its AFFINE2 coefficients stay in registers
  uint4 pg[NCH], pg2[NCH];
  bool gv[NCH];
  int base_row = 0;
  int crow[NCH], cpx[NCH];
  const int cc4 = tid & 3;
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    const int cid = tid + NT * i;
    crow[i] = cid / cpr;
    cpx[i] = (cid - crow[i] * cpr) >> 2;
  }
  const float* kco = ecoef + 640 + cc4 * 8;          // ga | gb | gc of this thread's channel chunk (LDS)

  // (chunk offsets inside the row group are constants of the thread, the group's offset is wave-uniform; rows outside the image
  // read offset 0 and are zeroed when staged -- see the forward kernel)
  uint32_t vog[NCH], vog2[NCH];
#pragma unroll
  for (int i = 0; i < NCH; ++i) {
    vog[i] = ((uint32_t)(crow[i] * W + cpx[i]) * (uint32_t)ldg + (uint32_t)cc4 * 8u) * 2u;
    vog2[i] = ((uint32_t)(crow[i] * W + cpx[i]) * (uint32_t)ldg2 + (uint32_t)cc4 * 8u) * 2u;
  }
  const char* __restrict__ gslb = reinterpret_cast<const char*>(gsl);
  const char* __restrict__ g2b = reinterpret_cast<const char*>(g2);
  auto issue_rows = [&](int b, int y0, int n) __attribute__((always_inline)) {
    const uint32_t row0 = (uint32_t)((b * H + y0) * W);
    const uint32_t sg = row0 * (uint32_t)ldg * 2u, sg2 = row0 * (uint32_t)ldg2 * 2u;
    const uint32_t r_lo = (uint32_t)max(-y0, 0), r_n = (uint32_t)max(min(n, H - y0), 0) - r_lo;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      gv[i] = ((uint32_t)crow[i] - r_lo) < r_n;
      pg[i] = *reinterpret_cast<const uint4*>(gslb + (size_t)(gv[i] ? sg + vog[i] : 0u));
      pg2[i] = *reinterpret_cast<const uint4*>(g2b + (size_t)(gv[i] ? sg2 + vog2[i] : 0u));
    }
  };
  auto write_rows = [&](int b, int y0, int n) __attribute__((always_inline)) {
    int slot_y0 = (y0 - base_row) % (R + 2);
    if (slot_y0 < 0) slot_y0 += R + 2;
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      if (crow[i] < n) {
        int slot = slot_y0 + crow[i];
        if (slot >= R + 2) slot -= R + 2;
        U128 o;
        o.u = cx_affine2_8(pg[i], pg2[i], kco, kco + 32, kco + 64);
        // side output (CxConv.pro_out): the corrected gradient slice as a dense tensor for the weight-gradient kernel.  Halo rows
        // are staged -- and stored -- by two workgroups: the same bits twice.
        if (po && gv[i]) *reinterpret_cast<uint4*>(po + ((size_t)(b * H + y0 + crow[i]) * W + cpx[i]) * ldpo + cc4 * 8) = o.u;
        { const unsigned keep = gv[i] ? 0xffffffffu : 0u; o.u.x &= keep; o.u.y &= keep; o.u.z &= keep; o.u.w &= keep; }   // no per-element branch
        const int pos = slot * P + cpx[i] + 1;
        *reinterpret_cast<uint4*>(ring + pos * GP + cc4 * 16) = o.u;
        if (pos < 2) *reinterpret_cast<uint4*>(ring + (Q + pos) * GP + cc4 * 16) = o.u;
      }
    }
  };

  float s1[2][2][8], s2[2][2][8];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int e = 0; e < 8; ++e) s1[j][cc][e] = s2[j][cc][e] = 0.f;
  const int nsub = (R * P + 31) / 32;                 // <= 4 * MAX_ITEMS
  const char* wbase = wl + (h2 * 64 + lrow) * GP + lh * 16;

#ifdef CX_RING_STAMPS
  unsigned long long st_acc[7] = {0, 0, 0, 0, 0, 0, 0}, st_prev = rstamp();
#endif
  for (int u = u0; u < u1; ++u) {
    const int b = div_spi(g, u), yc = (u - b * g.spi) * R;
    if (u == u0 || yc == 0) {
      base_row = yc - 1;
      issue_rows(b, yc - 1, 1);
      write_rows(b, yc - 1, 1);
      issue_rows(b, yc, 1);
      write_rows(b, yc, 1);
      issue_rows(b, yc + 1, R);
    }
    RSTAMP(0)
    write_rows(b, yc + 1, R);
    RSTAMP(1)
    __syncthreads();                                   // the window of this step is complete
    RSTAMP(2)
    const bool next_cont = (u + 1 < u1) && (div_spi(g, u + 1) == b);
    if (next_cont) issue_rows(b, yc + R + 1, R);
    RSTAMP(3)
    int slot0 = (yc - 1 - base_row) % (R + 2);
    if (slot0 < 0) slot0 += R + 2;
    const int ws = slot0 * P;

#pragma unroll 1
    for (int t = 0; t < MAX_ITEMS; ++t) {
      const int s = (wave >> 1) + 4 * t;
      if (s < nsub) {
        // mask source of this item (256 B per pixel, the bulk of the traffic): requested ahead of the item's 36 MFMAs, 16 B
        // per lane in the accumulator's own lane = pixel layout
        U128 xv[2][2];
        const int m = s * 32 + lrow;
        const int oy = (int)__umulhi((unsigned)m, g.mP), ox = m - oy * P;
        const int yy = yc + oy;
        const bool pok = m < R * P && ox < W && yy < H;
        const int poff = (b * H + min(yy, H - 1)) * W + min(ox, W - 1);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int cc = 0; cc < 2; ++cc)
            xv[j][cc].u = *reinterpret_cast<const uint4*>(ex + (size_t)poff * ldex + (2 * h2 + j) * 32 + 8 * (2 * cc + lh));
        f32x16 acc[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        const int pix = min(s * 32 + lrow, R * P - 1);
        // software pipeline over the nine taps inside the wave (as in the forward kernel): the 6 fragment reads of tap t+1 go
        // out before the 4 MFMAs of tap t (two register sets)
        const char* ap0 = ring + wrapq(wrapq(ws + pix, Q), Q) * GP + lh * 16;
        const char* ap1 = ring + wrapq(wrapq(ws + pix + P, Q), Q) * GP + lh * 16;
        const char* ap2 = ring + wrapq(wrapq(ws + pix + 2 * P, Q), Q) * GP + lh * 16;
        bf16x8 fb[2][2], fa[2][2][2];
        auto load_tap = [&](int t, bf16x8 (&B)[2], bf16x8 (&A)[2][2]) __attribute__((always_inline)) {
          const int dy = t / 3, dx = t - dy * 3;
          const char* ap = (dy == 0 ? ap0 : dy == 1 ? ap1 : ap2) + dx * GP;
          const char* wp = wbase + t * 128 * GP;
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            B[ks] = *reinterpret_cast<const bf16x8*>(ap + ks * 32);
#pragma unroll
            for (int j = 0; j < 2; ++j) A[ks][j] = *reinterpret_cast<const bf16x8*>(wp + (j * 32) * GP + ks * 32);
          }
        };
        load_tap(0, fb[0], fa[0]);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          if (t + 1 < 9) load_tap(t + 1, fb[(t + 1) & 1], fa[(t + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[t & 1][ks][j], fb[t & 1][ks], acc[j], 0, 0, 0);   // D[channel][pixel]
          __builtin_amdgcn_sched_barrier(0);
        }
#ifdef CX_RING_STAMPS
        asm volatile("" ::"v"(acc[0][0]), "v"(acc[1][0]));
#endif
        RSTAMP(4)
        bf16* yrow = y + (size_t)poff * ldy;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int cc = 0; cc < 2; ++cc) {
            const int n = (2 * h2 + j) * 32 + 8 * (2 * cc + lh);
            float v[8];
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
              const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[j][8 * cc + r4]),
                                                               __float_as_uint(acc[j][8 * cc + 4 + r4]), false, false);
              v[r4] = __uint_as_float(sw[0]);
              v[4 + r4] = __uint_as_float(sw[1]);
            }
            // phased so that only two coefficient vectors are live at a time (the compiler otherwise hoists all 160 values)
            float xf[8], dz[8];
            {
              asm volatile("" ::: "memory");
              const float4 a0 = *reinterpret_cast<const float4*>(ecoef + n), a1 = *reinterpret_cast<const float4*>(ecoef + n + 4);
              const float4 b0 = *reinterpret_cast<const float4*>(ecoef + 128 + n), b1 = *reinterpret_cast<const float4*>(ecoef + 128 + n + 4);
              const float esc[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
              const float esh[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
              const uint32_t xw[4] = {xv[j][cc].u.x, xv[j][cc].u.y, xv[j][cc].u.z, xv[j][cc].u.w};
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                xf[e] = (e & 1) ? cx_bf_hi(xw[e >> 1]) : cx_bf_lo(xw[e >> 1]);
                dz[e] = (pok && fmaf(xf[e], esc[e], esh[e]) > 0.f) ? v[e] : 0.f;
                s1[j][cc][e] += dz[e];
              }
            }
            // S2 = sum dz * (x - mu) * r = r * (sum dz*x - mu * S1): only sum dz*x is accumulated per element, the per-channel
            // affine part is applied once, after the final reduction
#pragma unroll
            for (int e = 0; e < 8; ++e) s2[j][cc][e] = fmaf(dz[e], xf[e], s2[j][cc][e]);
            U128 o;
            {
              asm volatile("" ::: "memory");
              const float4 e0 = *reinterpret_cast<const float4*>(ecoef + 512 + n), e1 = *reinterpret_cast<const float4*>(ecoef + 512 + n + 4);
              const float esl[8] = {e0.x, e0.y, e0.z, e0.w, e1.x, e1.y, e1.z, e1.w};
              o.u = make_uint4(cx_packbf(esl[0] * dz[0], esl[1] * dz[1]), cx_packbf(esl[2] * dz[2], esl[3] * dz[3]),
                               cx_packbf(esl[4] * dz[4], esl[5] * dz[5]), cx_packbf(esl[6] * dz[6], esl[7] * dz[7]));
            }
            if (pok) *reinterpret_cast<uint4*>(yrow + n) = o.u;
          }
        RSTAMP(5)
      }
    }
    __syncthreads();                                   // every wave is done with the oldest rows of the ring
    RSTAMP(6)
  }
#ifdef CX_RING_STAMPS
  if (tid == 0 && blockIdx.x < 1024) {
    for (int i = 0; i < 7; ++i) ring_stamps[blockIdx.x * 8 + i] = st_acc[i];
    ring_stamps[blockIdx.x * 8 + 7] = (unsigned long long)(u1 - u0);
  }
#endif

  {
    float* scratch = reinterpret_cast<float*>(wl);               // the weight slices are no longer read (ecoef stays)
    wg_stat_begin<NT / 64>(scratch, 128, tid, NT);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float a = half_sum(s1[j][cc][e]);
          const float c = half_sum(s2[j][cc][e]);
          if (lrow == 8 * cc + e) { t1 = a; t2 = c; }
        }
      if (lrow < 16) {
        const int n = (2 * h2 + j) * 32 + 8 * (2 * (lrow >> 3) + lh) + (lrow & 7);
        wg_stat_put(scratch, 128, wave, n, t1, ecoef[384 + n] * (t2 - ecoef[256 + n] * t1));
      }
    }
    wg_stat_end<NT / 64>(scratch, 128, tid, NT, S1, S2, stat_det, (int)blockIdx.x, stat_replicas, stat_rstride, 0, 128);
  }
}

template <int NCH>
int launch_ring_dgrad(const CxConv& p, hipStream_t st, const RingGeo& g) {
  const size_t smem = EC_BYTES + DW_BYTES + (size_t)(g.Q + 2) * GP;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_ring_dgrad_kernel<NCH>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    attr = true;
  }
  const int total = g.B * g.spi;
  const int grid = (total + g.steps_per_wg - 1) / g.steps_per_wg;
  if (const int e = stat_rows_check(p, grid)) return e;
  CX_KTAG("conv3x3_ring_dgrad_kernel<%d>", NCH);
  hipLaunchKernelGGL((conv3x3_ring_dgrad_kernel<NCH>), dim3(grid), dim3(NT), smem, st, (const bf16*)p.x, p.ldx, (const bf16*)p.x2,
                     p.ldx2, p.pa, p.pb, p.pc, (const bf16*)p.w, (const bf16*)p.ex, p.ldex, p.e_sc, p.e_sh, p.e_mu, p.e_r, p.e_scale,
                     (bf16*)p.y, p.ldy, p.stat_sum, p.stat_sq, p.stat_replicas, p.stat_rstride, p.stat_det, (bf16*)p.pro_out, p.ldpo, g);
  cx_tl_pro_out = p.pro_out ? 1 : 0;
  return launch_status();
}


// ------------------------------------------------------------------------------------------------ weight gradient
// dW[32 n][128 c][3][3] += sum_px dY[px][n] * A[px @ tap][c]: both MFMA operands are contracted over pixels, so both are
// read from pixel-major LDS images with the transposing ds_read_b64_tr_b16.  The first strip kernel gave a workgroup one
// 32-channel slice (its 3 waves issued 3 MFMAs per 8 LDS reads, each input row was fetched in four 64-B pieces by four
// workgroups and dY four times).  Here ONE 768-thread workgroup per CU keeps the whole 128-channel ring (320-B pitch:
// == 64 B mod 256 B for the transposing reads) and the dY strip of the step; wave (cs, dy) owns the three taps (dy, 0..2)
// of channel sub-tile cs -- 36 accumulator tiles spread over 12 waves, three waves per SIMD, nothing reduced across waves.
constexpr int WNT = 768;
constexpr int WXP = 320;                 // ring pitch: 128 bf16 + 64 B
constexpr int WGP = 64;                  // dY strip pitch: 32 bf16
constexpr int WCOEF = 352 * 4;           // prologue vectors

__device__ __forceinline__ bf16x8 tr2(const char* a0, const char* a1) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  // (joined by a shuffle + bit cast: assembled element by element the compiler emits a v_bfi per dword on the loaded registers and
  // waits for the read right where it is issued, not where the MFMA uses it -- common.h cx_join_tr)
  return cx_join_tr(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0)), __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a1)));
}

template <int NX, int NG>
__global__ __launch_bounds__(WNT, 1) void conv3x3_ring_wgrad_kernel(
    const bf16* __restrict__ gsl, int ldg, const bf16* __restrict__ g2, int ldg2, const float* __restrict__ ga,
    const float* __restrict__ gb, const float* __restrict__ gc, int g_affine2, const bf16* __restrict__ x, int ldx,
    const float* __restrict__ pa, const float* __restrict__ pb, float* __restrict__ dw, const RingGeo g,
    float* __restrict__ slab) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int P = g.P, R = g.R, Q = g.Q, W = g.W, H = g.H;
  const int nk = (R * P + 15) / 16;
  float* coef = reinterpret_cast<float*>(smem);                    // pa[128] pb[128] ga[32] gb[32] gc[32]
  char* ring = smem + WCOEF;                                       // [(Q+2)][320 B]
  char* gst = ring + (size_t)(Q + 2) * WXP;                        // [nk*16][64 B]
  float* red = reinterpret_cast<float*>(ring);                     // [32 n][32 c][9], aliased on the ring at the end
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cs = wave & 3, dy = wave >> 2;                         // channel sub-tile, kernel row

  for (int i = tid; i < ((Q + 2) * WXP + nk * 16 * WGP) / 16; i += WNT) reinterpret_cast<uint4*>(ring)[i] = make_uint4(0, 0, 0, 0);
  if (tid < 128) { coef[tid] = pa[tid]; coef[128 + tid] = pb[tid]; }
  if (tid < 32) {
    coef[256 + tid] = g_affine2 ? ga[tid] : 1.f;
    coef[288 + tid] = g_affine2 ? gb[tid] : 0.f;
    coef[320 + tid] = g_affine2 ? gc[tid] : 0.f;
  }

  // chunk slot i of this thread: chunk id tid + 768*i; 768 is a multiple of 16 and of 4, so the channel chunk of the input
  // rows is tid & 15 and that of the gradient rows tid & 3 for every slot (one LDS address per thread for its coefficients)
  const int cx8 = tid & 15, cg4 = tid & 3;
  const float* xco = coef + cx8 * 8;
  const float* gco = coef + 256 + cg4 * 8;
  __syncthreads();

  const int total_steps = g.B * g.spi;
  const int u0 = blockIdx.x * g.steps_per_wg;
  const int u1 = min(total_steps, u0 + g.steps_per_wg);
  const int xcpr = W * 16, gcpr = W * 4;

  uint4 pre[NX], pg[NG], pg2[NG];
  bool pv[NX], gv[NG];
  int base_row = 0;
  int xrow[NX], xpx[NX], grow[NG], gpx[NG];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int cid = tid + WNT * i;
    xrow[i] = cid / xcpr;
    xpx[i] = (cid - xrow[i] * xcpr) >> 4;
  }
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    const int cid = tid + WNT * i;
    grow[i] = cid / gcpr;
    gpx[i] = (cid - grow[i] * gcpr) >> 2;
  }
  // (chunk offsets inside the row group: constants of the thread; see the forward kernel)
  uint32_t vox[NX], vog[NG], vog2[NG];
#pragma unroll
  for (int i = 0; i < NX; ++i) vox[i] = ((uint32_t)(xrow[i] * W + xpx[i]) * (uint32_t)ldx + (uint32_t)cx8 * 8u) * 2u;
#pragma unroll
  for (int i = 0; i < NG; ++i) {
    vog[i] = ((uint32_t)(grow[i] * W + gpx[i]) * (uint32_t)ldg + (uint32_t)cg4 * 8u) * 2u;
    vog2[i] = ((uint32_t)(grow[i] * W + gpx[i]) * (uint32_t)ldg2 + (uint32_t)cg4 * 8u) * 2u;
  }
  const char* __restrict__ xb = reinterpret_cast<const char*>(x);
  const char* __restrict__ gslb = reinterpret_cast<const char*>(gsl);
  const char* __restrict__ g2b = reinterpret_cast<const char*>(g2);
  auto issue_rows = [&](int b, int y0, int n) __attribute__((always_inline)) {
    const uint32_t sx = (uint32_t)((b * H + y0) * W) * (uint32_t)ldx * 2u;
    const uint32_t r_lo = (uint32_t)max(-y0, 0), r_n = (uint32_t)max(min(n, H - y0), 0) - r_lo;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      pv[i] = ((uint32_t)xrow[i] - r_lo) < r_n;
      pre[i] = *reinterpret_cast<const uint4*>(xb + (size_t)(pv[i] ? sx + vox[i] : 0u));
    }
  };
  auto write_rows = [&](int y0, int n) __attribute__((always_inline)) {
    int slot_y0 = (y0 - base_row) % (R + 2);
    if (slot_y0 < 0) slot_y0 += R + 2;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      if (xrow[i] < n) {
        int slot = slot_y0 + xrow[i];
        if (slot >= R + 2) slot -= R + 2;
        U128 o;
        o.u = cx_affine_relu8(pre[i], xco, xco + 128);
        { const unsigned keep = pv[i] ? 0xffffffffu : 0u; o.u.x &= keep; o.u.y &= keep; o.u.z &= keep; o.u.w &= keep; }   // no per-element branch
        const int pos = slot * P + xpx[i] + 1;
        *reinterpret_cast<uint4*>(ring + pos * WXP + cx8 * 16) = o.u;
        if (pos < 2) *reinterpret_cast<uint4*>(ring + (Q + pos) * WXP + cx8 * 16) = o.u;   // mirror of pixels 0,1
      }
    }
  };
  auto issue_g = [&](int b, int yc) __attribute__((always_inline)) {
    const uint32_t row0 = (uint32_t)((b * H + yc) * W);
    const uint32_t sg = row0 * (uint32_t)ldg * 2u, sg2 = row0 * (uint32_t)ldg2 * 2u;
    const uint32_t r_n = (uint32_t)max(min(R, H - yc), 0);
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      gv[i] = (uint32_t)grow[i] < r_n;
      pg[i] = *reinterpret_cast<const uint4*>(gslb + (size_t)(gv[i] ? sg + vog[i] : 0u));
      pg2[i] = *reinterpret_cast<const uint4*>(g2b + (size_t)(gv[i] ? sg2 + vog2[i] : 0u));      // g2 == gsl when there is no second tensor
    }
  };
  auto write_g = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < NG; ++i) {
      if (grow[i] < R) {
        U128 o;
        o.u = cx_affine2_8(pg[i], pg2[i], gco, gco + 32, gco + 64);
        { const unsigned keep = gv[i] ? 0xffffffffu : 0u; o.u.x &= keep; o.u.y &= keep; o.u.z &= keep; o.u.w &= keep; }   // no per-element branch
        *reinterpret_cast<uint4*>(gst + (grow[i] * P + gpx[i]) * WGP + cg4 * 16) = o.u;     // pad columns / tail stay zero
      }
    }
  };

  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // lane constants of the transposing reads: 4 pixel rows x 16 channels per 16-lane group
  const int gq = (lane & 15) >> 2, gp = lane & 3, gg = lane >> 4;
  const int lrow = 8 * (gg >> 1) + gq;                      // pixel row inside a 16-pixel k-step (second read: +4)
  const int gcol = (16 * (gg & 1) + 4 * gp) * 2;            // byte offset of the channel block
  const int xcol = cs * 64 + gcol;
  const int QB = Q * WXP;

  for (int u = u0; u < u1; ++u) {
    const int b = div_spi(g, u), yc = (u - b * g.spi) * R;
    if (u == u0 || yc == 0) {
      base_row = yc - 1;
      issue_rows(b, yc - 1, 1);
      write_rows(yc - 1, 1);
      issue_rows(b, yc, 1);
      write_rows(yc, 1);
      issue_rows(b, yc + 1, R);
      issue_g(b, yc);
    }
    write_rows(yc + 1, R);
    write_g();
    __syncthreads();
    const bool next_cont = (u + 1 < u1) && (div_spi(g, u + 1) == b);
    if (next_cont) {
      issue_rows(b, yc + R + 1, R);
      issue_g(b, yc + R);
    }
    int slot0 = (yc - 1 - base_row) % (R + 2);
    if (slot0 < 0) slot0 += R + 2;
    const int ws = slot0 * P;
    int o0 = wrapq(wrapq(ws + lrow + dy * P, Q), Q) * WXP;
    int o1 = wrapq(wrapq(ws + lrow + 4 + dy * P, Q), Q) * WXP;
#pragma unroll 2
    for (int kk = 0; kk < nk; ++kk) {
      const char* gbase = gst + (kk * 16 + lrow) * WGP + gcol;
      const bf16x8 af = tr2(gbase, gbase + 4 * WGP);
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const bf16x8 bfr = tr2(ring + o0 + xcol + dx * WXP, ring + o1 + xcol + dx * WXP);
        acc[dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[dx], 0, 0, 0);
      }
      o0 += 16 * WXP;
      if (o0 >= QB) o0 -= QB;
      o1 += 16 * WXP;
      if (o1 >= QB) o1 -= QB;
    }
    __syncthreads();
  }

  // ---- per channel sub-tile: transpose through LDS into OIHW order, atomics over contiguous runs (288 floats per n)
  for (int round = 0; round < 4; ++round) {
    if (cs == round) {
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          red[(n * 32 + (lane & 31)) * 9 + dy * 3 + dx] = acc[dx][r];
        }
    }
    __syncthreads();
    for (int idx = tid; idx < 32 * 288; idx += WNT) {
      const int n = idx / 288, i = idx - n * 288;
      dw_out(dw, slab, (size_t)32 * 128 * 9, (int)blockIdx.x, ((size_t)n * 128 + round * 32) * 9 + i, red[idx]);
    }
    __syncthreads();
  }
}

template <int NX, int NG>
int launch_ring_wgrad(const CxWgrad& p, hipStream_t st, const RingGeo& g, size_t smem) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_ring_wgrad_kernel<NX, NG>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    attr = true;
  }
  const int total = g.B * g.spi;
  const int grid = (total + g.steps_per_wg - 1) / g.steps_per_wg;
  const bool a2 = p.g_prologue == CX_PRO_AFFINE2;
  const size_t wtotal = (size_t)32 * 128 * 9;
  float* slab = dw_slab(p.scratch, p.scratch_floats, grid, (long long)wtotal);
  CX_KTAG("conv3x3_ring_wgrad_kernel<%d, %d>", NX, NG);
  hipLaunchKernelGGL((conv3x3_ring_wgrad_kernel<NX, NG>), dim3(grid), dim3(WNT), smem, st, (const bf16*)p.g, p.ldg,
                     (const bf16*)(a2 ? p.g2 : p.g), a2 ? p.ldg2 : p.ldg, p.ga, p.gb, p.gc, (int)a2, (const bf16*)p.x, p.ldx, p.pa, p.pb,
                     p.dw, g, slab);
  if (const int e = launch_status()) return e;
  return slab ? cx_dw_reduce(p.dw, slab, wtotal, grid, st) : 0;
}

}  // namespace

// Eligibility + launch, called from cx_conv_gemm ahead of the first-generation strip kernel.
int cx_try_ring_fwd(const CxConv& p, hipStream_t st, bool* handled) {
  *handled = false;
  if (p.mode != CX_MODE_CONV || p.kh != 3 || p.kw != 3 || p.stride != 1 || p.pad != 1 || p.tstride > 1) return 0;
  if (p.K != 128 || p.N != 32 || p.prologue != CX_PRO_AFFINE_RELU || p.epilogue != CX_EPI_STORE || p.accumulate) return 0;
  if (p.W < 4) return 0;
  if ((unsigned long long)p.B * p.H * p.W * (unsigned long long)p.ldx * 2ull >= (1ull << 32)) return 0;      // 32-bit chunk offsets in the kernel
  RingGeo g;
  // Wide maps as two column tiles: an 80-pixel row is 2.5 sub-tiles for 8 waves and one row is all the ring holds, so a step was
  // 3 busy waves and ~3.6 us of fixed latency (290 us per launch at bs = 256); halves of 40 columns take 5 rows per step (7
  // sub-tiles) for one extra halo column per row.
  static const int tile_on = cx_diag_int("CX_RING_TILE", 1);
  g.ntx = (tile_on && p.W >= 64 && p.W % 2 == 0) ? 2 : 1;
  g.Wt = p.W / g.ntx;
  g.B = p.B * g.ntx; g.H = p.H; g.W = p.W; g.P = g.Wt + 2;
  int rmax = (RING_PX_MAX - 2) / g.P - 2;
  if (rmax < 1) return 0;
  if (rmax > p.H) rmax = p.H;
  g.spi = (p.H + rmax - 1) / rmax;
  g.R = (p.H + g.spi - 1) / g.spi;                   // balanced steps
  // at most 7 chunks of new rows per thread
  while (g.R > 1 && g.R * g.P * 16 > 7 * NT) --g.R;
  if (g.R * g.P * 16 > 7 * NT) return 0;
  g.spi = (p.H + g.R - 1) / g.R;
  g.Q = (g.R + 2) * g.P;
  g.mP = 0xffffffffu / (unsigned)g.P + 1u;
  g.mSpi = 0xffffffffu / (unsigned)g.spi + 1u;
  g.ntx_shift = 0;
  if ((unsigned long long)p.B * 2ull * g.spi * g.spi >= (1ull << 32)) return 0;      // exactness of the multiply-high divisions
  g.ntx_shift = g.ntx - 1;
  const int total = g.B * g.spi;
  // one workgroup per CU; ranges aligned to whole images when there are enough of them (no window rebuild inside an image)
  int spw = (total + 255) / 256;
  if (g.B >= 256) spw = ((g.B + 255) / 256) * g.spi;
  if (spw < 1) spw = 1;
  g.steps_per_wg = spw;
  const int need = (g.R * g.P * 16 + NT - 1) / NT;
  *handled = true;
  if (need <= 3) return launch_ring<3, 3>(p, st, g);      // short steps (one 80-pixel row): three row groups in flight
  if (need <= 5) return launch_ring<5, 2>(p, st, g);
  return launch_ring<7, 1>(p, st, g);
}

int cx_try_ring_dgrad(const CxConv& p, hipStream_t st, bool* handled) {
  *handled = false;
  if (p.mode != CX_MODE_CONV || p.kh != 3 || p.kw != 3 || p.stride != 1 || p.pad != 1 || p.tstride > 1) return 0;
  if (p.K != 32 || p.N != 128 || p.prologue != CX_PRO_AFFINE2 || p.epilogue != CX_EPI_MASK || p.accumulate) return 0;
  if (p.W < 4 || p.W + 2 > 256) return 0;
  if ((unsigned long long)p.B * p.H * p.W * (unsigned long long)(p.ldx > p.ldx2 ? p.ldx : p.ldx2) * 2ull >= (1ull << 32)) return 0;   // 32-bit chunk offsets
  RingGeo g;
  g.B = p.B; g.H = p.H; g.W = p.W; g.P = p.W + 2;
  int rmax = (32 * 4 * MAX_ITEMS) / g.P;             // at most 8 sub-tiles (16 work items over 8 waves) per step
  if (rmax < 1) return 0;
  if (rmax > p.H) rmax = p.H;
  g.spi = (p.H + rmax - 1) / rmax;
  g.R = (p.H + g.spi - 1) / g.spi;
  g.spi = (p.H + g.R - 1) / g.R;
  g.Q = (g.R + 2) * g.P;
  g.mP = 0xffffffffu / (unsigned)g.P + 1u;
  g.mSpi = 0xffffffffu / (unsigned)g.spi + 1u;
  g.ntx_shift = 0;
  if ((unsigned long long)p.B * 2ull * g.spi * g.spi >= (1ull << 32)) return 0;      // exactness of the multiply-high divisions
  if (EC_BYTES + DW_BYTES + (size_t)(g.Q + 2) * GP > 160 * 1024) return 0;
  const int need = (g.R * p.W * 4 + NT - 1) / NT;
  if (need > 2) return 0;
  const int total = g.B * g.spi;
  int spw = (total + 255) / 256;
  if (g.B >= 256) spw = ((g.B + 255) / 256) * g.spi;
  if (spw < 1) spw = 1;
  g.steps_per_wg = spw;
  *handled = true;
  return need <= 1 ? launch_ring_dgrad<1>(p, st, g) : launch_ring_dgrad<2>(p, st, g);
}

int cx_try_ring_wgrad(const CxWgrad& p, hipStream_t st, bool* handled) {
  *handled = false;
  if (p.mode != CX_MODE_CONV || p.kh != 3 || p.kw != 3 || p.stride != 1 || p.pad != 1) return 0;
  if (p.K != 128 || p.N != 32 || p.x_prologue != CX_PRO_AFFINE_RELU) return 0;
  if (p.g_prologue != CX_PRO_NONE && p.g_prologue != CX_PRO_AFFINE2) return 0;
  // every workgroup ends with 147 KB of fp32 atomics (~30 us at the per-CU atomic issue rate): pays off on large maps only
  // (measured 337 vs 556 us at 80x80, 128 vs 118 us at 40x40); smaller maps stay on the first-generation strip kernel
  if (p.W < 56 || (long long)p.H * p.W < 3136) return 0;
  {
    const unsigned long long ldm = (unsigned long long)(p.ldx > p.ldg ? p.ldx : p.ldg) > (unsigned long long)p.ldg2 ? (unsigned long long)(p.ldx > p.ldg ? p.ldx : p.ldg) : (unsigned long long)p.ldg2;
    if ((unsigned long long)p.B * p.H * p.W * ldm * 2ull >= (1ull << 32)) return 0;      // 32-bit chunk offsets in the kernel
  }
  RingGeo g;
  g.B = p.B; g.H = p.H; g.W = p.W; g.P = p.W + 2;
  // largest R with: <= 16 k-steps of 16 pixels, ring + strip in 160 KB, <= 5 / 2 chunks of new rows per thread, and the
  // 36 KB transposition buffer of the epilogue inside the ring
  int rmax = 0;
  for (int r = 1; r <= p.H; ++r) {
    const int nk = (r * g.P + 15) / 16;
    const size_t bytes = WCOEF + (size_t)((r + 2) * g.P + 2) * WXP + (size_t)nk * 16 * WGP;
    if (nk > 16 || bytes > 160 * 1024 || r * p.W * 16 > 5 * WNT || r * p.W * 4 > 2 * WNT) break;
    rmax = r;
  }
  if (rmax < 1) return 0;
  g.spi = (p.H + rmax - 1) / rmax;
  g.R = (p.H + g.spi - 1) / g.spi;
  g.spi = (p.H + g.R - 1) / g.R;
  g.Q = (g.R + 2) * g.P;
  g.mP = 0xffffffffu / (unsigned)g.P + 1u;
  g.mSpi = 0xffffffffu / (unsigned)g.spi + 1u;
  g.ntx_shift = 0;
  if ((unsigned long long)p.B * 2ull * g.spi * g.spi >= (1ull << 32)) return 0;      // exactness of the multiply-high divisions
  const int nk = (g.R * g.P + 15) / 16;
  size_t smem = WCOEF + (size_t)(g.Q + 2) * WXP + (size_t)nk * 16 * WGP;
  if (smem < WCOEF + 32 * 288 * 4) smem = WCOEF + 32 * 288 * 4;
  const int total = g.B * g.spi;
  int spw = (total + 255) / 256;
  if (g.B >= 256) spw = ((g.B + 255) / 256) * g.spi;
  if (spw < 1) spw = 1;
  g.steps_per_wg = spw;
  const int nx = (g.R * p.W * 16 + WNT - 1) / WNT, ng = (g.R * p.W * 4 + WNT - 1) / WNT;
  *handled = true;
  if (nx <= 2 && ng <= 1) return launch_ring_wgrad<2, 1>(p, st, g, smem);
  if (nx <= 3) return launch_ring_wgrad<3, 2>(p, st, g, smem);
  return launch_ring_wgrad<5, 2>(p, st, g, smem);
}

#ifdef CX_RING_STAMPS
extern "C" int dbg_ring_stamps(unsigned long long* host, int n_words) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ring_stamps), (size_t)n_words * 8, 0, hipMemcpyDeviceToHost);
}
extern "C" int dbg_ring_fwd_stamps(unsigned long long* host, int n_words) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(ring_fwd_stamps), (size_t)n_words * 8, 0, hipMemcpyDeviceToHost);
}
#endif
