// 3x3 stride-1 pad-1 convolutions of the dense layers as persistent "strip" kernels (gfx950).
//
// The generic implicit GEMM re-stages its A tile once per tap (9x L2 reads, two MFMAs per barrier at
// N=32).  Here a workgroup walks down an image R rows at a time and keeps a ring of R+2 normalised
// input rows in LDS in a PADDED-FLAT layout: every row slot is W+2 pixels wide with zero pad columns, so
// output pixel m (flat index over R x (W+2)) reads tap (dy,dx) at flat index m + dy*(W+2) + dx -- no
// per-tap staging, no edge branches; the two pad columns yield garbage outputs that are never stored.
// Each step loads only the R NEW rows from HBM (register prefetch under the MFMAs of the current step).
//
//   forward  (K=128 -> N=32): ring rows = relu(bn2(y1)), weights resident in LDS in MFMA-fragment order
//                             (72 fragments of 1 KiB, ds_read_b128 lane-linear => conflict-free);
//                             A fragments by ds_read_b128 from 272-B pixel rows (conflict-free for 32
//                             consecutive pixels).  3 waves, one 32-pixel sub-tile each, 72 MFMA per sub-tile.
//   wgrad    (dW[32][128][3][3]): ring rows (320-B pitch) are read with the transposing ds_read_b64_tr_b16,
//                             the gradient strip G (32 ch) likewise; each wave owns 32 input channels x 9
//                             taps (9 accumulator tiles) for the whole pixel range of the workgroup;
//                             one burst of fp32 atomics per workgroup at the end.
#include "common.h"

namespace {

constexpr int XP_FWD = 272;     // bytes per ring pixel, forward (128 bf16 + 16 pad): 68 banks == 4 (mod 64)
constexpr int XP_WG = 320;      // wgrad ring pitch: == 64 B (mod 256 B) for the transposing reads
constexpr int NCHX = 9;         // max 16-B chunks of new input rows per thread and step (forward, 192 threads)

struct StripGeo {
  int B, H, W, P, R, Q;         // P = W+2, Q = (R+2)*P ring pixels
  int spi;                      // steps per image = ceil(H/R)
  int steps_per_wg;
};

__device__ __forceinline__ int wrapq(int v, int q) { return v >= q ? v - q : v; }

// ------------------------------------------------------------------------------------------------ forward
__global__ __launch_bounds__(192) void conv3x3_strip_fwd_kernel(const bf16* __restrict__ x, int ldx,
                                                                 const float* __restrict__ sc, const float* __restrict__ sh,
                                                                 const bf16* __restrict__ wpk, bf16* __restrict__ y, int ldy,
                                                                 float* stat_sum, float* stat_sq, const StripGeo g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wl = smem;                                             // [9][8][64][16 B] fragment-ordered weights
  char* ring = wl + 9 * 8 * 64 * 16;                           // [(Q+2)][272 B]
  float* coef = reinterpret_cast<float*>(ring + (size_t)(g.Q + 2) * XP_FWD);   // [2][128]
  float* lstat = coef + 256;                                   // [2][32]
  float* scratch = lstat + 64;                                 // [3 waves][32][36]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int P = g.P, R = g.R, Q = g.Q, W = g.W, H = g.H;

  // ---- one-time setup: weights -> fragment order, ring zero, coefficient table
  for (int c = tid; c < 9 * 32 * 16; c += 192) {
    const int c8 = c & 15, n = (c >> 4) & 31, tap = c >> 9;
    const uint4 v = *reinterpret_cast<const uint4*>(wpk + ((size_t)(tap * 32 + n) * 128 + c8 * 8));
    const int ks = c8 >> 1, h = c8 & 1;
    *reinterpret_cast<uint4*>(wl + (((tap * 8 + ks) * 64) + n + 32 * h) * 16) = v;
  }
  for (int i = tid; i < (Q + 2) * (XP_FWD / 16); i += 192) reinterpret_cast<uint4*>(ring)[i] = make_uint4(0, 0, 0, 0);
  for (int i = tid; i < 128; i += 192) { coef[i] = sc[i]; coef[128 + i] = sh[i]; }
  if (tid < 64) lstat[tid] = 0.f;
  __syncthreads();

  const int total_steps = g.B * g.spi;
  const int u0 = blockIdx.x * g.steps_per_wg;
  const int u1 = min(total_steps, u0 + g.steps_per_wg);
  const int chunks_per_row = W * 16;

  uint4 pre[NCHX];
  bool pv[NCHX];
  int base_row = 0;           // image row held by ring slot 0 ... slot(y) = (y - base_row) mod (R+2)
  // loads rows [y0, y0+n) of image b into registers (16-B chunks spread over the threads)
  auto issue_rows = [&](int b, int y0, int n) {
#pragma unroll
    for (int i = 0; i < NCHX; ++i) {
      const int cid = tid + 192 * i;
      pv[i] = false;
      if (cid < n * chunks_per_row) {
        const int r = cid / chunks_per_row, rem = cid - r * chunks_per_row;
        const int px = rem >> 4, c8 = rem & 15, yy = y0 + r;
        if (yy >= 0 && yy < H) {
          pv[i] = true;
          pre[i] = *reinterpret_cast<const uint4*>(x + ((size_t)(b * H + yy) * W + px) * ldx + c8 * 8);
        }
      }
    }
  };
  auto write_rows = [&](int y0, int n) {
#pragma unroll
    for (int i = 0; i < NCHX; ++i) {
      const int cid = tid + 192 * i;
      if (cid < n * chunks_per_row) {
        const int r = cid / chunks_per_row, rem = cid - r * chunks_per_row;
        const int px = rem >> 4, c8 = rem & 15;
        int slot = (y0 + r - base_row) % (R + 2);
        if (slot < 0) slot += R + 2;
        U128 o;
        if (pv[i]) {
          U128 v;
          v.u = pre[i];
#pragma unroll
          for (int j = 0; j < 8; ++j) o.e[j] = f2bf(fmaxf(fmaf(bf2f(v.e[j]), coef[c8 * 8 + j], coef[128 + c8 * 8 + j]), 0.f));
        } else {
          o.u = make_uint4(0, 0, 0, 0);
        }
        const int pos = slot * P + px + 1;
        *reinterpret_cast<uint4*>(ring + (size_t)pos * XP_FWD + c8 * 16) = o.u;
        if (pos < 2) *reinterpret_cast<uint4*>(ring + (size_t)(Q + pos) * XP_FWD + c8 * 16) = o.u;   // mirror of pixels 0,1
      }
    }
  };

  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
  const int nsub = (R * P + 31) / 32;
  float* my_scr = scratch + wave * 32 * 36;
  bool have_window = false;
  int prev_b = -1, prev_yc = 0;

  for (int u = u0; u < u1; ++u) {
    const int b = u / g.spi, yc = (u - b * g.spi) * R;
    const bool cont = have_window && b == prev_b && yc == prev_yc + R;
    if (!cont) {
      // (re)build the window rows yc-1 .. yc: two synchronous row loads, then prefetch the R new rows
      __syncthreads();
      base_row = yc - 1;
      issue_rows(b, yc - 1, 1);
      write_rows(yc - 1, 1);
      issue_rows(b, yc, 1);
      write_rows(yc, 1);
      issue_rows(b, yc + 1, R);
    }
    write_rows(yc + 1, R);
    __syncthreads();
    const bool next_cont = (u + 1 < u1) && ((u + 1) / g.spi == b);
    if (next_cont) issue_rows(b, yc + R + 1, R);
    int slot0 = (yc - 1 - base_row) % (R + 2);
    if (slot0 < 0) slot0 += R + 2;
    const int ws = slot0 * P;

    for (int s = wave; s < nsub; s += 3) {
      const int pix = min(s * 32 + (lane & 31), R * P - 1);
      int aoff[3];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) aoff[dy] = wrapq(wrapq(ws + pix + dy * P, Q), Q) * XP_FWD + (lane >> 5) * 16;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
          for (int ks = 0; ks < 8; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(ring + aoff[dy] + dx * XP_FWD + ks * 32);
            const bf16x8 bw = *reinterpret_cast<const bf16x8*>(wl + ((((dy * 3 + dx) * 8 + ks) * 64) + lane) * 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bw, acc, 0, 0, 0);
          }
      // ---- epilogue of the sub-tile: transpose through per-wave LDS scratch, 16-B stores along channels
#pragma unroll
      for (int r = 0; r < 16; ++r) my_scr[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 36 + (lane & 31)] = acc[r];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int pr = pass * 16 + (lane >> 2), c = lane & 3;
        const int m = s * 32 + pr;
        const int oy = m / P, ox = m - oy * P;
        const int yy = yc + oy;
        if (m < R * P && ox < W && yy < H) {
          const float4 v0 = *reinterpret_cast<const float4*>(my_scr + pr * 36 + c * 8);
          const float4 v1 = *reinterpret_cast<const float4*>(my_scr + pr * 36 + c * 8 + 4);
          const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
          U128 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            o.e[j] = f2bf(v[j]);
            const float rv = bf2f(o.e[j]);
            s1[j] += rv;
            s2[j] += rv * rv;
          }
          *reinterpret_cast<uint4*>(y + ((size_t)(b * H + yy) * W + ox) * ldy + c * 8) = o.u;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();
    have_window = true;
    prev_b = b;
    prev_yc = yc;
  }

  if (stat_sum) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
      for (int d = 4; d < 64; d <<= 1) {
        s1[j] += __shfl_xor(s1[j], d);
        s2[j] += __shfl_xor(s2[j], d);
      }
    }
    if (lane < 4) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        atomicAdd(&lstat[lane * 8 + j], s1[j]);
        atomicAdd(&lstat[32 + lane * 8 + j], s2[j]);
      }
    }
    __syncthreads();
    if (tid < 32) {
      atomicAdd(&stat_sum[tid], lstat[tid]);
      atomicAdd(&stat_sq[tid], lstat[32 + tid]);
    }
  }
}

// ------------------------------------------------------------------------------------------------ weight gradient
__device__ __forceinline__ bf16x8 tr2(const char* a0, const char* a1) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  U64 lo, hi;
  lo.s = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0));
  hi.s = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a1));
  bf16x8 r;
  r[0] = lo.e[0]; r[1] = lo.e[1]; r[2] = lo.e[2]; r[3] = lo.e[3];
  r[4] = hi.e[0]; r[5] = hi.e[1]; r[6] = hi.e[2]; r[7] = hi.e[3];
  return r;
}

constexpr int NCHW = 4;         // 16-B chunks per thread and step, for the input rows and for the gradient rows
constexpr int WP = 64;          // bytes per ring / strip pixel: 32 channels; rows of a half-wave hit disjoint bank quarters

// Workgroup = (32-input-channel tile ct, pixel range).  The four waves split the 16-pixel k-steps of a
// strip; each keeps 9 accumulator tiles (one per tap) of dW[32 n][32 c].  At the end the waves are summed
// through LDS and added to the OIHW gradient with atomics over 1152-B contiguous runs per output channel.
__global__ __launch_bounds__(256) void conv3x3_strip_wgrad_kernel(
    const bf16* __restrict__ gsl, int ldg, const bf16* __restrict__ g2, int ldg2, const float* __restrict__ ga,
    const float* __restrict__ gb, const float* __restrict__ gc, int g_affine2, const bf16* __restrict__ x, int ldx,
    const float* __restrict__ pa, const float* __restrict__ pb, float* __restrict__ dw, const StripGeo g, const int n_splits) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int P = g.P, R = g.R, Q = g.Q, W = g.W, H = g.H;
  const int nk = (R * P + 15) / 16;
  char* ring = smem;                                               // [(Q+2)][64 B]
  char* gst = ring + (size_t)(Q + 2) * WP;                         // [nk*16][64 B]
  float* coef = reinterpret_cast<float*>(gst + (size_t)nk * 16 * WP);   // pa[32] pb[32] ga[32] gb[32] gc[32]
  float* red = coef + 160;                                         // [9][32][32] final cross-wave sum
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wg = xcd_remap(blockIdx.x, gridDim.x);
  const int ct = wg & 3, split = wg >> 2;                          // the 4 channel tiles of a pixel range are neighbours
  const int c0 = ct * 32;

  for (int i = tid; i < (Q + 2) * (WP / 16); i += 256) reinterpret_cast<uint4*>(ring)[i] = make_uint4(0, 0, 0, 0);
  for (int i = tid; i < nk * 16 * (WP / 16); i += 256) reinterpret_cast<uint4*>(gst)[i] = make_uint4(0, 0, 0, 0);
  for (int i = tid; i < 9 * 32 * 32; i += 256) red[i] = 0.f;
  if (tid < 32) {
    coef[tid] = pa[c0 + tid];
    coef[32 + tid] = pb[c0 + tid];
    coef[64 + tid] = g_affine2 ? ga[tid] : 1.f;
    coef[96 + tid] = g_affine2 ? gb[tid] : 0.f;
    coef[128 + tid] = g_affine2 ? gc[tid] : 0.f;
  }
  __syncthreads();

  const int total_steps = g.B * g.spi;
  const int u0 = split * g.steps_per_wg;
  const int u1 = min(total_steps, u0 + g.steps_per_wg);
  const int cpr = W * 4;                 // chunks per row (32 channels)

  uint4 pre[NCHW], pg[NCHW], pg2[NCHW];
  bool pv[NCHW], gv[NCHW];
  int base_row = 0;
  auto issue_rows = [&](int b, int y0, int n) {
#pragma unroll
    for (int i = 0; i < NCHW; ++i) {
      const int cid = tid + 256 * i;
      pv[i] = false;
      if (cid < n * cpr) {
        const int r = cid / cpr, rem = cid - r * cpr;
        const int px = rem >> 2, c8 = rem & 3, yy = y0 + r;
        if (yy >= 0 && yy < H) {
          pv[i] = true;
          pre[i] = *reinterpret_cast<const uint4*>(x + ((size_t)(b * H + yy) * W + px) * ldx + c0 + c8 * 8);
        }
      }
    }
  };
  auto write_rows = [&](int y0, int n) {
#pragma unroll
    for (int i = 0; i < NCHW; ++i) {
      const int cid = tid + 256 * i;
      if (cid < n * cpr) {
        const int r = cid / cpr, rem = cid - r * cpr;
        const int px = rem >> 2, c8 = rem & 3;
        int slot = (y0 + r - base_row) % (R + 2);
        if (slot < 0) slot += R + 2;
        U128 o;
        if (pv[i]) {
          U128 v;
          v.u = pre[i];
#pragma unroll
          for (int j = 0; j < 8; ++j) o.e[j] = f2bf(fmaxf(fmaf(bf2f(v.e[j]), coef[c8 * 8 + j], coef[32 + c8 * 8 + j]), 0.f));
        } else {
          o.u = make_uint4(0, 0, 0, 0);
        }
        const int pos = slot * P + px + 1;
        *reinterpret_cast<uint4*>(ring + (size_t)pos * WP + c8 * 16) = o.u;
        if (pos < 2) *reinterpret_cast<uint4*>(ring + (size_t)(Q + pos) * WP + c8 * 16) = o.u;   // mirror of pixels 0,1
      }
    }
  };
  auto issue_g = [&](int b, int yc) {
#pragma unroll
    for (int i = 0; i < NCHW; ++i) {
      const int cid = tid + 256 * i;
      gv[i] = false;
      if (cid < R * cpr) {
        const int r = cid / cpr, rem = cid - r * cpr;
        const int px = rem >> 2, c8 = rem & 3, yy = yc + r;
        if (yy < H) {
          gv[i] = true;
          const size_t pixel = (size_t)(b * H + yy) * W + px;
          pg[i] = *reinterpret_cast<const uint4*>(gsl + pixel * ldg + c8 * 8);
          if (g_affine2) pg2[i] = *reinterpret_cast<const uint4*>(g2 + pixel * ldg2 + c8 * 8);
        }
      }
    }
  };
  auto write_g = [&]() {
#pragma unroll
    for (int i = 0; i < NCHW; ++i) {
      const int cid = tid + 256 * i;
      if (cid < R * cpr) {
        const int r = cid / cpr, rem = cid - r * cpr;
        const int px = rem >> 2, c8 = rem & 3;
        U128 o;
        if (gv[i]) {
          U128 u, v;
          u.u = pg[i];
          if (g_affine2) {
            v.u = pg2[i];
#pragma unroll
            for (int j = 0; j < 8; ++j)
              o.e[j] = f2bf(fmaf(bf2f(u.e[j]), coef[64 + c8 * 8 + j], fmaf(bf2f(v.e[j]), coef[96 + c8 * 8 + j], coef[128 + c8 * 8 + j])));
          } else {
            o.u = u.u;
          }
        } else {
          o.u = make_uint4(0, 0, 0, 0);
        }
        *reinterpret_cast<uint4*>(gst + (size_t)(r * P + px) * WP + c8 * 16) = o.u;   // pad columns / tail stay zero
      }
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // lane constants of the transposing reads (conv_wgrad.hip tr_frag): 4 pixel rows x 16 channels per 16-lane group
  const int gq = (lane & 15) >> 2, gp = lane & 3, gg = lane >> 4;
  const int lrow = 8 * (gg >> 1) + gq;                      // pixel row inside a 16-pixel k-step (second read: +4)
  const int gcol = (16 * (gg & 1) + 4 * gp) * 2;            // byte offset of the channel block
  const int QB = Q * WP;
  bool have_window = false;
  int prev_b = -1, prev_yc = 0;

  for (int u = u0; u < u1; ++u) {
    const int b = u / g.spi, yc = (u - b * g.spi) * R;
    const bool cont = have_window && b == prev_b && yc == prev_yc + R;
    if (!cont) {
      __syncthreads();
      base_row = yc - 1;
      issue_rows(b, yc - 1, 2);
      write_rows(yc - 1, 2);
      issue_rows(b, yc + 1, R);
      issue_g(b, yc);
    }
    write_rows(yc + 1, R);
    write_g();
    __syncthreads();
    const bool next_cont = (u + 1 < u1) && ((u + 1) / g.spi == b);
    if (next_cont) {
      issue_rows(b, yc + R + 1, R);
      issue_g(b, yc + R);
    }
    int slot0 = (yc - 1 - base_row) % (R + 2);
    if (slot0 < 0) slot0 += R + 2;
    const int ws = slot0 * P;
    // running byte offsets (pixel part) of this lane's two pixel rows for the three kernel rows dy
    int o0[3], o1[3];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      o0[dy] = wrapq(wrapq(ws + 16 * wave + lrow + dy * P, Q), Q) * WP;
      o1[dy] = wrapq(wrapq(ws + 16 * wave + lrow + 4 + dy * P, Q), Q) * WP;
    }
    for (int kk = wave; kk < nk; kk += 4) {
      const char* gbase = gst + (size_t)(kk * 16 + lrow) * WP + gcol;
      const bf16x8 af = tr2(gbase, gbase + 4 * WP);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const bf16x8 bfr = tr2(ring + o0[dy] + gcol + dx * WP, ring + o1[dy] + gcol + dx * WP);
          acc[dy * 3 + dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[dy * 3 + dx], 0, 0, 0);
        }
        o0[dy] += 64 * WP;
        if (o0[dy] >= QB) o0[dy] -= QB;
        o1[dy] += 64 * WP;
        if (o1[dy] >= QB) o1[dy] -= QB;
      }
    }
    __syncthreads();
    have_window = true;
    prev_b = b;
    prev_yc = yc;
  }

  // ---- sum the four waves through LDS, then atomics over contiguous OIHW runs
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      atomicAdd(&red[(t * 32 + n) * 32 + (lane & 31)], acc[t][r]);
    }
  __syncthreads();
  for (int idx = tid; idx < 32 * 288; idx += 256) {
    const int n = idx / 288, i = idx - n * 288;
    const int c = i / 9, t = i - c * 9;
    atomicAdd(dw + ((size_t)n * 128 + c0) * 9 + i, red[(t * 32 + n) * 32 + c]);
  }
  (void)n_splits;
}

inline StripGeo make_geo(int B, int H, int W, int target_wgs, int min_steps, int max_flat) {
  StripGeo g;
  g.B = B; g.H = H; g.W = W; g.P = W + 2;
  g.R = max_flat / g.P;
  if (g.R < 1) g.R = 1;
  if (g.R > H) g.R = H;
  g.Q = (g.R + 2) * g.P;
  g.spi = (H + g.R - 1) / g.R;
  const int total = B * g.spi;
  int spw = (total + target_wgs - 1) / target_wgs;
  if (spw < min_steps) spw = min_steps;
  if (spw > total) spw = total;
  g.steps_per_wg = spw;
  return g;
}

}  // namespace

// Eligibility + launch, called from cx_conv_gemm / cx_conv_wgrad (same ABI, faster path).
int cx_try_strip_fwd(const CxConv& p, hipStream_t st, bool* handled) {
  *handled = false;
  if (p.mode != CX_MODE_CONV || p.kh != 3 || p.kw != 3 || p.stride != 1 || p.pad != 1) return 0;
  if (p.K != 128 || p.N != 32 || p.prologue != CX_PRO_AFFINE_RELU || p.epilogue != CX_EPI_STORE) return 0;
  if (p.W + 2 > 96 || p.W < 4) return 0;
  StripGeo g = make_geo(p.B, p.H, p.W, 256, 4, 96);
  if (g.R * p.W * 16 > NCHX * 192) return 0;
  const size_t smem = 9 * 8 * 64 * 16 + (size_t)(g.Q + 2) * XP_FWD + 256 * 4 + 64 * 4 + 3 * 32 * 36 * 4;
  if (smem > 160 * 1024) return 0;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_strip_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    attr = true;
  }
  const int total = g.B * g.spi;
  const int grid = (total + g.steps_per_wg - 1) / g.steps_per_wg;
  hipLaunchKernelGGL(conv3x3_strip_fwd_kernel, dim3(grid), dim3(192), smem, st, (const bf16*)p.x, p.ldx, p.pa, p.pb,
                     (const bf16*)p.w, (bf16*)p.y, p.ldy, p.stat_sum, p.stat_sq, g);
  *handled = true;
  return launch_status();
}

int cx_try_strip_wgrad(const CxWgrad& p, hipStream_t st, bool* handled) {
  *handled = false;
  if (p.mode != CX_MODE_CONV || p.kh != 3 || p.kw != 3 || p.stride != 1 || p.pad != 1) return 0;
  if (p.K != 128 || p.N != 32 || p.x_prologue != CX_PRO_AFFINE_RELU) return 0;
  if (p.g_prologue != CX_PRO_NONE && p.g_prologue != CX_PRO_AFFINE2) return 0;
  if (p.W + 2 > 128 || p.W < 4) return 0;
  // pixel-range splits: enough workgroups to fill the chip, few enough that the final atomics stay small
  const long long px = (long long)p.B * p.H * p.W;
  int target = (int)(px / 12800);
  if (target < 8) target = 8;
  if (target > 128) target = 128;
  StripGeo g = make_geo(p.B, p.H, p.W, target, 2, 256);
  if (g.R * p.W * 4 > NCHW * 256 || g.Q < 64) return 0;
  const int nk = (g.R * g.P + 15) / 16;
  const size_t smem = (size_t)(g.Q + 2) * WP + (size_t)nk * 16 * WP + 160 * 4 + 9 * 32 * 32 * 4;
  if (smem > 160 * 1024) return 0;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_strip_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
    attr = true;
  }
  const int total = g.B * g.spi;
  const int splits = (total + g.steps_per_wg - 1) / g.steps_per_wg;
  hipLaunchKernelGGL(conv3x3_strip_wgrad_kernel, dim3(splits * 4), dim3(256), smem, st, (const bf16*)p.g, p.ldg, (const bf16*)p.g2,
                     p.ldg2, p.ga, p.gb, p.gc, (int)(p.g_prologue == CX_PRO_AFFINE2), (const bf16*)p.x, p.ldx, p.pa, p.pb, p.dw, g,
                     splits);
  *handled = true;
  return launch_status();
}
