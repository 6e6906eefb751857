#include <cstdlib>
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
constexpr int NCHX = 7;         // max 16-B chunks of new input rows per thread and step (forward, 192 threads)

struct StripGeo {
  int B, H, W, P, R, Q;         // P = W+2, Q = (R+2)*P ring pixels
  int spi;                      // steps per image = ceil(H/R)
  int steps_per_wg;
};

__device__ __forceinline__ int wrapq(int v, int q) { return v >= q ? v - q : v; }

// ------------------------------------------------------------------------------------------------ forward
// 3 waves; wave dy keeps the weights of its three taps (dy, 0..2) in REGISTERS (24 fragments = 96 VGPRs), so
// LDS only holds the input ring (~67 KB at W=80) and two workgroups share a CU: while one stages rows /
// stores its tile the other feeds the matrix pipe.  The three per-row partial sums of a 32-pixel sub-tile
// meet in a 12 KB LDS scratch, where the tile is also transposed for 16-B stores along the channel axis.
__global__ __launch_bounds__(192, 2) void conv3x3_strip_fwd_kernel(const bf16* __restrict__ x, int ldx,
                                                                 const float* __restrict__ sc, const float* __restrict__ sh,
                                                                 const bf16* __restrict__ wpk, bf16* __restrict__ y, int ldy,
                                                                 float* stat_sum, float* stat_sq, int stat_replicas, int stat_rstride,
                                                                 int stat_det, const StripGeo g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* coef = reinterpret_cast<float*>(smem);                // [2][128]
  float* lstat = coef + 256;                                   // [2][32]
  float* scratch = lstat + 64;                                 // [3 waves][32 px][32 ch]
  char* ring = reinterpret_cast<char*>(scratch + 3 * 1024);    // [(Q+2)][272 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int P = g.P, R = g.R, Q = g.Q, W = g.W, H = g.H;

  // ---- one-time setup
  bf16x8 wr[3][8];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      wr[dx][ks] = *reinterpret_cast<const bf16x8*>(wpk + ((size_t)((wave * 3 + dx) * 32 + (lane & 31)) * 128 + ks * 16 + (lane >> 5) * 8));
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
  // step-invariant decomposition of this thread's chunks: row inside the group of new rows, pixel, channel chunk
  int crow[NCHX], cpx[NCHX], cc8[NCHX];
#pragma unroll
  for (int i = 0; i < NCHX; ++i) {
    const int cid = tid + 192 * i;
    crow[i] = cid / chunks_per_row;
    const int rem = cid - crow[i] * chunks_per_row;
    cpx[i] = rem >> 4;
    cc8[i] = rem & 15;
  }
  // loads rows [y0, y0+n) of image b into registers.  Loads are unconditional on clamped addresses
  // (a branch around a load makes hipcc wait per load); validity is applied when the rows are staged.
  auto issue_rows = [&](int b, int y0, int n) {
#pragma unroll
    for (int i = 0; i < NCHX; ++i) {
      const int yy = y0 + crow[i];
      pv[i] = crow[i] < n && yy >= 0 && yy < H;
      const int yc_ = min(max(yy, 0), H - 1);
      pre[i] = *reinterpret_cast<const uint4*>(x + ((size_t)(b * H + yc_) * W + cpx[i]) * ldx + cc8[i] * 8);
    }
  };
  auto write_rows = [&](int y0, int n) {
#pragma unroll
    for (int i = 0; i < NCHX; ++i) {
      if (crow[i] < n) {
        int slot = (y0 + crow[i] - base_row) % (R + 2);
        if (slot < 0) slot += R + 2;
        U128 o, v;
        v.u = pre[i];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          o.e[j] = f2bf(fmaxf(fmaf(bf2f(v.e[j]), coef[cc8[i] * 8 + j], coef[128 + cc8[i] * 8 + j]), 0.f));
        { const unsigned keep = pv[i] ? 0xffffffffu : 0u; o.u.x &= keep; o.u.y &= keep; o.u.z &= keep; o.u.w &= keep; }   // no per-element branch
        const int pos = slot * P + cpx[i] + 1;
        *reinterpret_cast<uint4*>(ring + pos * XP_FWD + cc8[i] * 16) = o.u;
        if (pos < 2) *reinterpret_cast<uint4*>(ring + (Q + pos) * XP_FWD + cc8[i] * 16) = o.u;   // mirror of pixels 0,1
      }
    }
  };

  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
  const int nsub = (R * P + 31) / 32;           // <= 3
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

    for (int s = 0; s < nsub; ++s) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      {
        const int pix = min(s * 32 + (lane & 31), R * P - 1);
        const char* ap = ring + wrapq(wrapq(ws + pix + wave * P, Q), Q) * XP_FWD + (lane >> 5) * 16;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
          for (int ks = 0; ks < 8; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(ap + dx * XP_FWD + ks * 32);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, wr[dx][ks], acc, 0, 0, 0);
          }
      }
      // the three kernel-row partials meet in LDS; 128 threads sum / round / store 16 B each
      float* my = scratch + wave * 1024;
#pragma unroll
      for (int r = 0; r < 16; ++r) my[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[r];
      __syncthreads();
      if (tid < 128) {
        const int pr = tid >> 2, c = tid & 3;
        const int m = s * 32 + pr;
        const int oy = m / P, ox = m - oy * P;
        const int yy = yc + oy;
        if (m < R * P && ox < W && yy < H) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = 0.f;
#pragma unroll
          for (int w3 = 0; w3 < 3; ++w3) {
            const float4 v0 = *reinterpret_cast<const float4*>(scratch + w3 * 1024 + pr * 32 + c * 8);
            const float4 v1 = *reinterpret_cast<const float4*>(scratch + w3 * 1024 + pr * 32 + c * 8 + 4);
            v[0] += v0.x; v[1] += v0.y; v[2] += v0.z; v[3] += v0.w;
            v[4] += v1.x; v[5] += v1.y; v[6] += v1.z; v[7] += v1.w;
          }
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
      __syncthreads();
    }
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
    wg_stat_begin<3>(scratch, 32, tid, 192);
    if (lane < 4 && wave < 2) {
#pragma unroll
      for (int j = 0; j < 8; ++j) wg_stat_put(scratch, 32, wave, lane * 8 + j, s1[j], s2[j]);
    }
    wg_stat_end<3>(scratch, 32, tid, 192, stat_sum, stat_sq, stat_det, (int)blockIdx.x, stat_replicas, stat_rstride, 0, 32);
  }
}

// ------------------------------------------------------------------------------------------------ input gradient
// dz2[m][128] = mask(y1) * sum_taps dY2[m @ tap][32] * Wt[tap][128][32]   (K = 9 x 32, N = 128)
// Same skeleton as the forward kernel: wave dy holds its three taps x four 32-channel output tiles in
// registers (24 fragments), the ring holds the 32-channel gradient slice after the deferred BN correction
// (AFFINE2 of the gradient buffer slice and the activation slice), the three partial 32 px x 128 ch tiles meet
// in LDS where the ReLU/BN mask epilogue (CX_EPI_MASK) runs with 16-B accesses; y1 is prefetched under the MFMAs.
constexpr int GP = 80;          // ring pitch: 32 bf16 + 16 B pad (20 banks: conflict-free ds_read_b128 over 32 pixels)
constexpr int NCHD = 2;         // 16-B chunks of new gradient rows per thread and step (per tensor)

__global__ __launch_bounds__(192, 2) void conv3x3_strip_dgrad_kernel(
    const bf16* __restrict__ gsl, int ldg, const bf16* __restrict__ g2, int ldg2, const float* __restrict__ ga,
    const float* __restrict__ gb, const float* __restrict__ gc, const bf16* __restrict__ wpk, const bf16* __restrict__ ex, int ldex,
    const float* __restrict__ e_sc, const float* __restrict__ e_sh, const float* __restrict__ e_mu, const float* __restrict__ e_r,
    const float* __restrict__ e_scale, bf16* __restrict__ y, int ldy, float* S1, float* S2, int stat_replicas, int stat_rstride,
    int stat_det, const StripGeo g) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* coef = reinterpret_cast<float*>(smem);                // ga gb gc [32] each (96) + pad
  float* lstat = coef + 128;                                   // [2][128]
  float* econst = lstat + 256;                                 // e_sc, e_sh, e_mu, e_r, e_scale [128] each
  float* scratch = econst + 640;                               // [3 waves][32 px][128 ch]
  char* ring = reinterpret_cast<char*>(scratch + 3 * 4096);    // [(Q+2)][80 B]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int P = g.P, R = g.R, Q = g.Q, W = g.W, H = g.H;

  bf16x8 wr[3][2][4];                                          // [dx][ks][n-tile]
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        wr[dx][ks][nt] = *reinterpret_cast<const bf16x8*>(wpk + ((size_t)((wave * 3 + dx) * 128 + nt * 32 + (lane & 31)) * 32 + ks * 16 + (lane >> 5) * 8));
  for (int i = tid; i < (Q + 2) * (GP / 16); i += 192) reinterpret_cast<uint4*>(ring)[i] = make_uint4(0, 0, 0, 0);
  if (tid < 32) { coef[tid] = ga[tid]; coef[32 + tid] = gb[tid]; coef[64 + tid] = gc[tid]; }
  for (int i = tid; i < 256; i += 192) lstat[i] = 0.f;
  // epilogue constants of this thread's fixed 8-channel chunk
  const int ec = (tid & 15) * 8;
  float s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s1[j] = s2[j] = 0.f;
  if (tid < 128) {
    econst[tid] = e_sc[tid]; econst[128 + tid] = e_sh[tid]; econst[256 + tid] = e_mu[tid]; econst[384 + tid] = e_r[tid];
    econst[512 + tid] = e_scale[tid];
  }
  __syncthreads();

  const int total_steps = g.B * g.spi;
  const int u0 = blockIdx.x * g.steps_per_wg;
  const int u1 = min(total_steps, u0 + g.steps_per_wg);
  const int cpr = W * 4;

  uint4 pg[NCHD], pg2[NCHD];
  bool gv[NCHD];
  int base_row = 0;
  int crow[NCHD], cpx[NCHD], cc8[NCHD];
#pragma unroll
  for (int i = 0; i < NCHD; ++i) {
    const int cid = tid + 192 * i;
    crow[i] = cid / cpr;
    const int rem = cid - crow[i] * cpr;
    cpx[i] = rem >> 2;
    cc8[i] = rem & 3;
  }
  auto issue_rows = [&](int b, int y0, int n) {
#pragma unroll
    for (int i = 0; i < NCHD; ++i) {
      const int yy = y0 + crow[i];
      gv[i] = crow[i] < n && yy >= 0 && yy < H;
      const int yc_ = min(max(yy, 0), H - 1);
      const size_t pixel = (size_t)(b * H + yc_) * W + cpx[i];
      pg[i] = *reinterpret_cast<const uint4*>(gsl + pixel * ldg + cc8[i] * 8);
      pg2[i] = *reinterpret_cast<const uint4*>(g2 + pixel * ldg2 + cc8[i] * 8);
    }
  };
  auto write_rows = [&](int y0, int n) {
#pragma unroll
    for (int i = 0; i < NCHD; ++i) {
      if (crow[i] < n) {
        int slot = (y0 + crow[i] - base_row) % (R + 2);
        if (slot < 0) slot += R + 2;
        U128 o, u, v;
        u.u = pg[i];
        v.u = pg2[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float t = fmaf(bf2f(u.e[j]), coef[cc8[i] * 8 + j], fmaf(bf2f(v.e[j]), coef[32 + cc8[i] * 8 + j], coef[64 + cc8[i] * 8 + j]));
          o.e[j] = f2bf(t);
        }
        { const unsigned keep = gv[i] ? 0xffffffffu : 0u; o.u.x &= keep; o.u.y &= keep; o.u.z &= keep; o.u.w &= keep; }   // no per-element branch
        const int pos = slot * P + cpx[i] + 1;
        *reinterpret_cast<uint4*>(ring + pos * GP + cc8[i] * 16) = o.u;
        if (pos < 2) *reinterpret_cast<uint4*>(ring + (Q + pos) * GP + cc8[i] * 16) = o.u;
      }
    }
  };

  const int nsub = (R * P + 31) / 32;
  bool have_window = false;
  int prev_b = -1, prev_yc = 0;

  for (int u = u0; u < u1; ++u) {
    const int b = u / g.spi, yc = (u - b * g.spi) * R;
    const bool cont = have_window && b == prev_b && yc == prev_yc + R;
    if (!cont) {
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

    for (int s = 0; s < nsub; ++s) {
      // prefetch the mask source (y1) of this sub-tile: 32 px x 16 chunks = 512 items over 192 threads
      uint4 exv[3];
      bool exok[3];
      int eoff[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int item = tid + 192 * i;
        const int pr = item >> 4;
        const int m = s * 32 + pr;
        const int oy = m / P, ox = m - oy * P;
        const int yy = yc + oy;
        exok[i] = item < 512 && m < R * P && ox < W && yy < H;
        const int oyc = min(yy, H - 1), oxc = min(ox, W - 1);
        eoff[i] = (b * H + oyc) * W + oxc;
        exv[i] = *reinterpret_cast<const uint4*>(ex + (size_t)eoff[i] * ldex + ec);
      }
      f32x16 acc[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
      {
        const int pix = min(s * 32 + (lane & 31), R * P - 1);
        const char* ap = ring + wrapq(wrapq(ws + pix + wave * P, Q), Q) * GP + (lane >> 5) * 16;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 a = *reinterpret_cast<const bf16x8*>(ap + dx * GP + ks * 32);
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, wr[dx][ks][nt], acc[nt], 0, 0, 0);
          }
      }
      float* my = scratch + wave * 4096;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) my[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 128 + nt * 32 + (lane & 31)] = acc[nt][r];
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const int item = tid + 192 * i;
        const int pr = item >> 4;
        if (exok[i]) {
          float v[8];
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = 0.f;
#pragma unroll
          for (int w3 = 0; w3 < 3; ++w3) {
            const float4 v0 = *reinterpret_cast<const float4*>(scratch + w3 * 4096 + pr * 128 + ec);
            const float4 v1 = *reinterpret_cast<const float4*>(scratch + w3 * 4096 + pr * 128 + ec + 4);
            v[0] += v0.x; v[1] += v0.y; v[2] += v0.z; v[3] += v0.w;
            v[4] += v1.x; v[5] += v1.y; v[6] += v1.z; v[7] += v1.w;
          }
          U128 xv, o;
          xv.u = exv[i];
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float xf = bf2f(xv.e[j]);
            const float dz = (fmaf(xf, econst[ec + j], econst[128 + ec + j]) > 0.f) ? v[j] : 0.f;
            s1[j] += dz;
            s2[j] += dz * (xf - econst[256 + ec + j]) * econst[384 + ec + j];
            o.e[j] = f2bf(econst[512 + ec + j] * dz);
          }
          *reinterpret_cast<uint4*>(y + (size_t)eoff[i] * ldy + ec) = o.u;
        }
      }
      __syncthreads();
    }
    have_window = true;
    prev_b = b;
    prev_yc = yc;
  }

  // statistics: lanes l, l+16, l+32, l+48 share a channel chunk
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    s1[j] += __shfl_xor(s1[j], 16);
    s1[j] += __shfl_xor(s1[j], 32);
    s2[j] += __shfl_xor(s2[j], 16);
    s2[j] += __shfl_xor(s2[j], 32);
  }
  wg_stat_begin<3>(scratch, 128, tid, 192);
  if (lane < 16) {
#pragma unroll
    for (int j = 0; j < 8; ++j) wg_stat_put(scratch, 128, wave, ec + j, s1[j], s2[j]);
  }
  wg_stat_end<3>(scratch, 128, tid, 192, S1, S2, stat_det, (int)blockIdx.x, stat_replicas, stat_rstride, 0, 128);
}

// ------------------------------------------------------------------------------------------------ weight gradient
__device__ __forceinline__ bf16x8 tr2(const char* a0, const char* a1) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  // (joined by a shuffle + bit cast: assembled element by element the compiler emits a v_bfi per dword on the loaded registers and
  // waits for the read right where it is issued, not where the MFMA uses it -- common.h cx_join_tr)
  return cx_join_tr(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a0)), __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(a1)));
}

constexpr int NCHW1 = 5;        // 16-B chunks per thread and step (192 threads), for the input rows and for the gradient rows
constexpr int NCHW4 = 3;        // the same with four wave groups (768 threads, <= 168 VGPRs)
constexpr int WP = 64;          // bytes per ring / strip pixel: 32 channels; rows of a half-wave hit disjoint bank quarters

// Workgroup = (32-input-channel tile ct, pixel range), 3 waves: wave dy owns the three taps (dy, 0..2) of
// dW[32 n][32 c] (48 accumulator registers), so nothing is reduced across waves and ~47 KB of LDS / ~170
// VGPRs let 2-3 workgroups share a CU (their staging and MFMA phases overlap).  At the end the tile is
// transposed through LDS and added to the OIHW gradient with atomics over 1152-B contiguous runs.
//
// NG > 1 (small maps, one workgroup per CU): NG groups of three waves share one staged strip and split its 16-pixel
// k-steps round-robin, so a step can be NG times longer (up to a whole image: more bytes in flight per barrier) while its
// MFMA and staging phases shrink by NG; the groups' tiles meet in LDS before the atomics.
#ifdef CX_STRIP_STAMPS
// diagnostic build (scratch/stamps_strip.py): s_memtime sums per phase of the weight-gradient body, wave 0 of each workgroup
__device__ unsigned long long strip_stamps[1024 * 8];
__device__ __forceinline__ unsigned long long sstamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define SSTAMP(i) { const unsigned long long t_ = sstamp(); st_acc[i] += t_ - st_prev; st_prev = t_; }
#else
#define SSTAMP(i)
#endif
template <int NG, int NCHW>
__device__ __forceinline__ void strip_wgrad_body(
    const bf16* __restrict__ gsl, int ldg, const bf16* __restrict__ g2, int ldg2, const float* __restrict__ ga,
    const float* __restrict__ gb, const float* __restrict__ gc, int g_affine2, const bf16* __restrict__ x, int ldx,
    const float* __restrict__ pa, const float* __restrict__ pb, float* __restrict__ dw, const int K, const int N,
    const int c_tiles, const int n_tiles, const StripGeo& g, float* __restrict__ slab, const int wg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int P = g.P, R = g.R, Q = g.Q, W = g.W, H = g.H;
  const int nk = (R * P + 15) / 16;
  float* coef = reinterpret_cast<float*>(smem);                    // pa[32] pb[32] ga[32] gb[32] gc[32]
  char* ring = smem + 160 * 4;                                     // [(Q+2)][64 B]
  char* gst = ring + (size_t)(Q + 2) * WP;                         // [nk*16][64 B]
  float* red = reinterpret_cast<float*>(ring);                     // [32 n][32 c][9] aliased on ring+strip at the end
  constexpr int NTHR = 192 * NG;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = (tid >> 6) % 3, grp = (tid >> 6) / 3;           // wave = kernel row dy, grp = k-step residue
  // (32 input channels, 32 output channels) tile pairs of one pixel range are neighbours, input tile fastest: the dense
  // layers are 4 x 1 pairs (K = 128, N = 32), a ResNet 256 -> 256 conv is 8 x 8
  const int pairs = c_tiles * n_tiles;
  const int split = wg / pairs, pair = wg - split * pairs;
  const int nt = pair / c_tiles, ct = pair - nt * c_tiles;
  const int c0 = ct * 32, n0 = nt * 32;
  gsl += n0;
  g2 += n0;

  for (int i = tid; i < ((Q + 2) + nk * 16) * (WP / 16); i += NTHR) reinterpret_cast<uint4*>(ring)[i] = make_uint4(0, 0, 0, 0);
  if (tid < 32) {
    coef[tid] = pa[c0 + tid];
    coef[32 + tid] = pb[c0 + tid];
    const int nn = n0 + tid < N ? n0 + tid : N - 1;   // the last output-channel tile may be partial (N % 8 == 0)
    coef[64 + tid] = g_affine2 ? ga[nn] : 1.f;
    coef[96 + tid] = g_affine2 ? gb[nn] : 0.f;
    coef[128 + tid] = g_affine2 ? gc[nn] : 0.f;
  }
  __syncthreads();

  const int total_steps = g.B * g.spi;
  const int u0 = split * g.steps_per_wg;
  const int u1 = min(total_steps, u0 + g.steps_per_wg);
  const int cpr = W * 4;                 // chunks per row (32 channels)

  uint4 pre[NCHW], pg[NCHW], pg2[NCHW];
  bool pv[NCHW], gv[NCHW];
  int base_row = 0;
  int crow[NCHW], cpx[NCHW], cc8[NCHW];
#pragma unroll
  for (int i = 0; i < NCHW; ++i) {
    const int cid = tid + NTHR * i;
    crow[i] = cid / cpr;
    const int rem = cid - crow[i] * cpr;
    cpx[i] = rem >> 2;
    cc8[i] = rem & 3;
  }
  // unconditional loads on clamped addresses; validity applied when staging (see forward kernel)
  // (chunk offsets inside the row group: constants of the thread; the group's offset is wave-uniform; rows outside the image read
  // offset 0 and are zeroed when staged -- conv3x3_ring.hip, forward kernel)
  uint32_t vox[NCHW], vog[NCHW], vog2[NCHW];
#pragma unroll
  for (int i = 0; i < NCHW; ++i) {
    vox[i] = ((uint32_t)(crow[i] * W + cpx[i]) * (uint32_t)ldx + (uint32_t)(c0 + cc8[i] * 8)) * 2u;
    const int co = n0 + cc8[i] * 8 < N ? cc8[i] * 8 : 0;
    vog[i] = ((uint32_t)(crow[i] * W + cpx[i]) * (uint32_t)ldg + (uint32_t)co) * 2u;
    vog2[i] = ((uint32_t)(crow[i] * W + cpx[i]) * (uint32_t)ldg2 + (uint32_t)co) * 2u;
  }
  const char* __restrict__ xb = reinterpret_cast<const char*>(x);
  const char* __restrict__ gslb = reinterpret_cast<const char*>(gsl);
  const char* __restrict__ g2b = reinterpret_cast<const char*>(g2);
  // (one chunk at a time: the step loop requests the next step's chunks between its multiplications)
  auto issue_rows_i = [&](int i, uint32_t sx, uint32_t r_lo, uint32_t r_n) __attribute__((always_inline)) {
    pv[i] = ((uint32_t)crow[i] - r_lo) < r_n;
    pre[i] = *reinterpret_cast<const uint4*>(xb + (size_t)(pv[i] ? sx + vox[i] : 0u));
  };
  auto issue_rows = [&](int b, int y0, int n) {
    const uint32_t sx = (uint32_t)((b * H + y0) * W) * (uint32_t)ldx * 2u;
    const uint32_t r_lo = (uint32_t)max(-y0, 0), r_n = (uint32_t)max(min(n, H - y0), 0) - r_lo;
#pragma unroll
    for (int i = 0; i < NCHW; ++i) issue_rows_i(i, sx, r_lo, r_n);
  };
  auto write_rows = [&](int y0, int n) {
#pragma unroll
    for (int i = 0; i < NCHW; ++i) {
      if (crow[i] < n) {
        int slot = (y0 + crow[i] - base_row) % (R + 2);
        if (slot < 0) slot += R + 2;
        U128 o;
        o.u = cx_affine_relu8(pre[i], coef + cc8[i] * 8, coef + 32 + cc8[i] * 8);
        { const unsigned keep = pv[i] ? 0xffffffffu : 0u; o.u.x &= keep; o.u.y &= keep; o.u.z &= keep; o.u.w &= keep; }   // no per-element branch
        const int pos = slot * P + cpx[i] + 1;
        *reinterpret_cast<uint4*>(ring + pos * WP + cc8[i] * 16) = o.u;
        if (pos < 2) *reinterpret_cast<uint4*>(ring + (Q + pos) * WP + cc8[i] * 16) = o.u;   // mirror of pixels 0,1
      }
    }
  };
  auto issue_g_i = [&](int i, uint32_t sg, uint32_t sg2, uint32_t r_n) __attribute__((always_inline)) {
    const bool nok = n0 + cc8[i] * 8 < N;          // chunk of 8 output channels inside the tensor
    gv[i] = (uint32_t)crow[i] < r_n && nok;
    pg[i] = *reinterpret_cast<const uint4*>(gslb + (size_t)(gv[i] ? sg + vog[i] : 0u));
    if (g_affine2) pg2[i] = *reinterpret_cast<const uint4*>(g2b + (size_t)(gv[i] ? sg2 + vog2[i] : 0u));
  };
  auto issue_g = [&](int b, int yc) {
    const uint32_t row0 = (uint32_t)((b * H + yc) * W);
    const uint32_t sg = row0 * (uint32_t)ldg * 2u, sg2 = row0 * (uint32_t)ldg2 * 2u;
    const uint32_t r_n = (uint32_t)max(min(R, H - yc), 0);
#pragma unroll
    for (int i = 0; i < NCHW; ++i) issue_g_i(i, sg, sg2, r_n);
  };
  auto write_g = [&]() {
#pragma unroll
    for (int i = 0; i < NCHW; ++i) {
      if (crow[i] < R) {
        U128 o;
        // (without the two-tensor prologue the staged values are the loaded ones: a = 1, b = c = 0 reproduces them exactly)
        o.u = g_affine2 ? cx_affine2_8(pg[i], pg2[i], coef + 64 + cc8[i] * 8, coef + 96 + cc8[i] * 8, coef + 128 + cc8[i] * 8) : pg[i];
        { const unsigned keep = gv[i] ? 0xffffffffu : 0u; o.u.x &= keep; o.u.y &= keep; o.u.z &= keep; o.u.w &= keep; }   // no per-element branch
        *reinterpret_cast<uint4*>(gst + (crow[i] * P + cpx[i]) * WP + cc8[i] * 16) = o.u;   // pad columns / tail stay zero
      }
    }
  };

  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // lane constants of the transposing reads (conv_wgrad.hip tr_frag): 4 pixel rows x 16 channels per 16-lane group
  const int gq = (lane & 15) >> 2, gp = lane & 3, gg = lane >> 4;
  const int lrow = 8 * (gg >> 1) + gq;                      // pixel row inside a 16-pixel k-step (second read: +4)
  const int gcol = (16 * (gg & 1) + 4 * gp) * 2;            // byte offset of the channel block
  const int QB = Q * WP;
  bool have_window = false;
  int prev_b = -1, prev_yc = 0;

#ifdef CX_STRIP_STAMPS
  unsigned long long st_acc[6] = {0, 0, 0, 0, 0, 0}, st_prev = sstamp();
#endif
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
    SSTAMP(0)                         // restart: halo round trip
    write_rows(yc + 1, R);
    write_g();
    SSTAMP(1)                         // wait for the requested rows + staging
    __syncthreads();
    SSTAMP(2)                         // barrier 1
    const bool next_cont = (u + 1 < u1) && ((u + 1) / g.spi == b);
    SSTAMP(3)
    int slot0 = (yc - 1 - base_row) % (R + 2);
    if (slot0 < 0) slot0 += R + 2;
    const int ws = slot0 * P;
    // running byte offsets of this lane's two pixel rows in kernel row dy = wave
    int o0 = ((ws + lrow + wave * P + grp * 16) % Q) * WP;
    int o1 = ((ws + lrow + 4 + wave * P + grp * 16) % Q) * WP;
    auto kstep = [&](int kk) __attribute__((always_inline)) {
      const char* gbase = gst + (kk * 16 + lrow) * WP + gcol;
      const bf16x8 af = tr2(gbase, gbase + 4 * WP);
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const bf16x8 bfr = tr2(ring + o0 + gcol + dx * WP, ring + o1 + gcol + dx * WP);
        acc[dx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[dx], 0, 0, 0);
      }
      o0 += 16 * NG * WP;
      while (o0 >= QB) o0 -= QB;
      o1 += 16 * NG * WP;
      while (o1 >= QB) o1 -= QB;
    };
    int kk = grp;
    if (next_cont) {
      // the next step's rows are requested one chunk per k-step: requested in one burst ahead of the loop, the 12 waves spent
      // ~2000 cycles a step queueing 72 KB at the CU's vector-memory port with the matrix pipe idle (scratch/stamps_strip.py)
      const int yn = yc + R + 1;
      const uint32_t sx = (uint32_t)((b * H + yn) * W) * (uint32_t)ldx * 2u;
      const uint32_t r_lo = (uint32_t)max(-yn, 0), r_nx = (uint32_t)max(min(R, H - yn), 0) - r_lo;
      const uint32_t row0 = (uint32_t)((b * H + yc + R) * W);
      const uint32_t sg = row0 * (uint32_t)ldg * 2u, sg2 = row0 * (uint32_t)ldg2 * 2u;
      const uint32_t r_ng = (uint32_t)max(min(R, H - (yc + R)), 0);
#pragma unroll
      for (int i = 0; i < NCHW; ++i) {
        issue_rows_i(i, sx, r_lo, r_nx);
        issue_g_i(i, sg, sg2, r_ng);
        if (kk < nk) { kstep(kk); kk += NG; }
      }
    }
    for (; kk < nk; kk += NG) kstep(kk);
#ifdef CX_STRIP_STAMPS
    asm volatile("" ::"v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[2][0]));
#endif
    SSTAMP(4)                         // multiply
    __syncthreads();
    SSTAMP(5)                         // barrier 2
    have_window = true;
    prev_b = b;
    prev_yc = yc;
  }
#ifdef CX_STRIP_STAMPS
  if (tid == 0 && wg < 1024) {
    for (int i = 0; i < 6; ++i) strip_stamps[wg * 8 + i] = st_acc[i];
    strip_stamps[wg * 8 + 6] = (unsigned long long)(u1 - u0);
  }
#endif

  // ---- transpose through LDS into OIHW order, then atomics over contiguous runs (288 floats per output channel)
  for (int round = 0; round < NG; ++round) {
    if (grp == round) {
#pragma unroll
      for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
          float* slot = red + (n * 32 + (lane & 31)) * 9 + wave * 3 + dx;
          *slot = round == 0 ? acc[dx][r] : *slot + acc[dx][r];
        }
    }
    __syncthreads();
  }
  for (int idx = tid; idx < 32 * 288; idx += NTHR) {
    const int n = idx / 288, i = idx - n * 288;
    if (n0 + n < N) dw_out(dw, slab, (size_t)N * K * 9, split, ((size_t)(n0 + n) * K + c0) * 9 + i, red[idx]);
  }
}

template <int NG, int NCHW>
__global__ __launch_bounds__(192 * NG) void conv3x3_strip_wgrad_kernel(
    const bf16* __restrict__ gsl, int ldg, const bf16* __restrict__ g2, int ldg2, const float* __restrict__ ga,
    const float* __restrict__ gb, const float* __restrict__ gc, int g_affine2, const bf16* __restrict__ x, int ldx,
    const float* __restrict__ pa, const float* __restrict__ pb, float* __restrict__ dw, const int K, const int N,
    const int c_tiles, const int n_tiles, const StripGeo g, float* __restrict__ slab) {
  strip_wgrad_body<NG, NCHW>(gsl, ldg, g2, ldg2, ga, gb, gc, g_affine2, x, ldx, pa, pb, dw, K, N, c_tiles, n_tiles, g, slab,
                             xcd_remap(blockIdx.x, gridDim.x));
}

// The 3x3 weight gradients of SEVERAL dense layers of one block in one launch (cx_conv3x3_wgrad_batch).  A dense layer's weight
// gradient feeds nothing but the flat gradient buffer, so it need not run between its layer's input-gradient kernels: with the
// corrected gradient slice left behind as a dense tensor (CxConv.pro_out) and the saved bottleneck tensor, the launches of a whole
// block are independent of each other.  On the 20x20 / 10x10 maps one such launch is 24-36 us for 2-9 us of memory traffic --
// launch, first-window latency and the partial-tile epilogue of 256 workgroups; batched, workgroup = (layer, split, channel tile),
// the workgroups of the next layer start while the slow ones of this layer finish, each workgroup walks a longer pixel range
// (fewer partial tiles: parallelism comes from the layers), and nothing sits on the input-gradient chain.
struct StripBatch {
  const bf16* g[CX_WGRAD_BATCH_MAX];
  const bf16* x[CX_WGRAD_BATCH_MAX];
  const float* pa[CX_WGRAD_BATCH_MAX];
  const float* pb[CX_WGRAD_BATCH_MAX];
  float* slab[CX_WGRAD_BATCH_MAX];
};

template <int NG, int NCHW>
__global__ __launch_bounds__(192 * NG) void conv3x3_strip_wgrad_batch_kernel(const StripBatch bt, const int per_layer, int ldg, int ldx,
                                                                             const int K, const int N, const int c_tiles,
                                                                             const int n_tiles, const StripGeo g) {
  const int layer = blockIdx.x / per_layer;
  const int r = blockIdx.x - layer * per_layer;
  const int wg = (per_layer & 7) ? r : xcd_remap(r, per_layer);    // per_layer % 8 == 0: the low bits of r are still the XCD
  strip_wgrad_body<NG, NCHW>(bt.g[layer], ldg, bt.g[layer], ldg, nullptr, nullptr, nullptr, 0, bt.x[layer], ldx, bt.pa[layer],
                             bt.pb[layer], nullptr, K, N, c_tiles, n_tiles, g, bt.slab[layer], wg);
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
  StripGeo g = make_geo(p.B, p.H, p.W, 512, 4, 96);      // >= 2 workgroups per CU
  if (g.R * p.W * 16 > NCHX * 192) return 0;
  const size_t smem = (256 + 64 + 3 * 1024) * 4 + (size_t)(g.Q + 2) * XP_FWD;
  if (smem > 80 * 1024) return 0;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_strip_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              80 * 1024);
    attr = true;
  }
  const int total = g.B * g.spi;
  const int grid = (total + g.steps_per_wg - 1) / g.steps_per_wg;
  *handled = true;
  if (const int e = stat_rows_check(p, grid)) return e;
  CX_KTAG("conv3x3_strip_fwd_kernel");
  hipLaunchKernelGGL(conv3x3_strip_fwd_kernel, dim3(grid), dim3(192), smem, st, (const bf16*)p.x, p.ldx, p.pa, p.pb,
                     (const bf16*)p.w, (bf16*)p.y, p.ldy, p.stat_sum, p.stat_sq, p.stat_replicas, p.stat_rstride, p.stat_det, g);
  return launch_status();
}

int cx_try_strip_dgrad(const CxConv& p, hipStream_t st, bool* handled) {
  *handled = false;
  if (p.mode != CX_MODE_CONV || p.kh != 3 || p.kw != 3 || p.stride != 1 || p.pad != 1) return 0;
  if (p.K != 32 || p.N != 128 || p.prologue != CX_PRO_AFFINE2 || p.epilogue != CX_EPI_MASK || p.accumulate) return 0;
  if (p.W + 2 > 96 || p.W < 4) return 0;
  StripGeo g = make_geo(p.B, p.H, p.W, 512, 4, 96);
  if (g.R * p.W * 4 > NCHD * 192) return 0;
  const size_t smem = (128 + 256 + 640 + 3 * 4096) * 4 + (size_t)(g.Q + 2) * GP;
  if (smem > 80 * 1024) return 0;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_strip_dgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              80 * 1024);
    attr = true;
  }
  const int total = g.B * g.spi;
  const int grid = (total + g.steps_per_wg - 1) / g.steps_per_wg;
  *handled = true;
  if (const int e = stat_rows_check(p, grid)) return e;
  CX_KTAG("conv3x3_strip_dgrad_kernel");
  hipLaunchKernelGGL(conv3x3_strip_dgrad_kernel, dim3(grid), dim3(192), smem, st, (const bf16*)p.x, p.ldx, (const bf16*)p.x2, p.ldx2,
                     p.pa, p.pb, p.pc, (const bf16*)p.w, (const bf16*)p.ex, p.ldex, p.e_sc, p.e_sh, p.e_mu, p.e_r, p.e_scale,
                     (bf16*)p.y, p.ldy, p.stat_sum, p.stat_sq, p.stat_replicas, p.stat_rstride, p.stat_det, g);
  return launch_status();
}

int cx_try_strip_wgrad(const CxWgrad& p, hipStream_t st, bool* handled) {
  *handled = false;
  if (p.mode != CX_MODE_CONV || p.kh != 3 || p.kw != 3 || p.stride != 1 || p.pad != 1) return 0;
  if (p.K % 32 || p.N % 8 || p.K > 2048 || p.N > 2048 || p.x_prologue != CX_PRO_AFFINE_RELU) return 0;
  if (p.g_prologue != CX_PRO_NONE && p.g_prologue != CX_PRO_AFFINE2) return 0;
  if (p.W + 2 > 128 || p.W < 4) return 0;
  {
    unsigned long long ldm = (unsigned long long)(p.ldx > p.ldg ? p.ldx : p.ldg);
    if (p.g_prologue == CX_PRO_AFFINE2 && (unsigned long long)p.ldg2 > ldm) ldm = (unsigned long long)p.ldg2;
    if ((unsigned long long)p.B * p.H * p.W * ldm * 2ull >= (1ull << 32)) return 0;      // 32-bit chunk offsets in the kernel
  }
  const long long px = (long long)p.B * p.H * p.W;
  const int c_tiles = p.K / 32, n_tiles = (p.N + 31) / 32, pairs = c_tiles * n_tiles;
  // pixel-range splits: the dense layers (4 tile pairs) take 64; wider convolutions ~768 workgroups in all, at least 4 splits
  int wide_splits = 768 / pairs;                 // 384 / 1536 / 3072 workgroups measured 1-3 % slower on ResNet152
  if (wide_splits < 4) wide_splits = 4;
  if (wide_splits > 64) wide_splits = 64;
  const int split_target = pairs == 4 ? 64 : wide_splits;     // 96 / 128 / 192 splits measured slower on the dense layers
  // small maps: four wave groups per workgroup on strips of up to 880 flat pixels, one workgroup per CU
  {
    int flat = NCHW4 * 768 / (p.W * 4) * (p.W + 2);          // rows the staging registers hold
    if (flat > 880) flat = 880;
    StripGeo g = make_geo(p.B, p.H, p.W, split_target, 1, flat);
    const int nk = (g.R * g.P + 15) / 16;
    size_t smem = 160 * 4 + (size_t)(g.Q + 2) * WP + (size_t)nk * 16 * WP;
    if (smem < 160 * 4 + 32 * 288 * 4) smem = 160 * 4 + 32 * 288 * 4;
    static const bool off = cx_diag_set("CX_SW_NG1");
    if (!off && g.R * p.W * 4 <= NCHW4 * 768 && 2 * p.W * 4 <= NCHW4 * 768 && g.Q >= 64 && smem <= 150 * 1024) {
      static bool attr = false;
      if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_strip_wgrad_kernel<4, NCHW4>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
        attr = true;
      }
      const int total = g.B * g.spi;
      const int splits = (total + g.steps_per_wg - 1) / g.steps_per_wg;
      const size_t wtotal = (size_t)p.N * p.K * 9;
      float* slab = dw_slab(p.scratch, p.scratch_floats, splits, (long long)wtotal);
      CX_KTAG("conv3x3_strip_wgrad_kernel<4, %d>", NCHW4);
      hipLaunchKernelGGL((conv3x3_strip_wgrad_kernel<4, NCHW4>), dim3(splits * pairs), dim3(768), smem, st, (const bf16*)p.g, p.ldg,
                         (const bf16*)p.g2, p.ldg2, p.ga, p.gb, p.gc, (int)(p.g_prologue == CX_PRO_AFFINE2), (const bf16*)p.x, p.ldx,
                         p.pa, p.pb, p.dw, p.K, p.N, c_tiles, n_tiles, g, slab);
      *handled = true;
      if (const int e = launch_status()) return e;
      return slab ? cx_dw_reduce(p.dw, slab, wtotal, splits, st) : 0;
    }
  }
  // pixel-range splits: >= 2 workgroups per CU over the 4 channel tiles, few enough that the final atomics
  // (147 KB per split) stay small next to the activations (384 B per pixel)
  int target = (int)(px / 3200);
  if (target < 64) target = 64;
  if (target > 192) target = 192;
  if (pairs != 4) target = split_target;
  StripGeo g = make_geo(p.B, p.H, p.W, target, 1, 256);
  if (g.R * p.W * 4 > NCHW1 * 192 || 2 * p.W * 4 > NCHW1 * 192 || g.Q < 16) return 0;
  const int nk = (g.R * g.P + 15) / 16;
  size_t smem = 160 * 4 + (size_t)(g.Q + 2) * WP + (size_t)nk * 16 * WP;
  if (smem < 160 * 4 + 32 * 288 * 4) smem = 160 * 4 + 32 * 288 * 4;
  if (smem > 64 * 1024) return 0;
  const int total = g.B * g.spi;
  const int splits = (total + g.steps_per_wg - 1) / g.steps_per_wg;
  const size_t wtotal = (size_t)p.N * p.K * 9;
  float* slab = dw_slab(p.scratch, p.scratch_floats, splits, (long long)wtotal);
  CX_KTAG("conv3x3_strip_wgrad_kernel<1, %d>", NCHW1);
  hipLaunchKernelGGL((conv3x3_strip_wgrad_kernel<1, NCHW1>), dim3(splits * pairs), dim3(192), smem, st, (const bf16*)p.g, p.ldg,
                     (const bf16*)p.g2, p.ldg2, p.ga, p.gb, p.gc, (int)(p.g_prologue == CX_PRO_AFFINE2), (const bf16*)p.x, p.ldx, p.pa, p.pb,
                     p.dw, p.K, p.N, c_tiles, n_tiles, g, slab);
  *handled = true;
  if (const int e = launch_status()) return e;
  return slab ? cx_dw_reduce(p.dw, slab, wtotal, splits, st) : 0;
}

#ifdef CX_STRIP_STAMPS
extern "C" int dbg_strip_stamps(unsigned long long* host, int n_words) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(strip_stamps), (size_t)n_words * 8, 0, hipMemcpyDeviceToHost);
}
#endif

// ABI 8.  See StripBatch above; include/chexpert_hip.h has the contract.
int cx_conv3x3_wgrad_batch(const CxWgrad* geo, const CxWgradBatch* items, void* stream) {
  if (!geo || !items) return CX_EINVAL;
  const CxWgrad& p = *geo;
  const int n = items->n;
  if (n <= 0 || n > CX_WGRAD_BATCH_MAX) return CX_EINVAL;
  if (p.dtype != CX_DT_BF16 || p.mode != CX_MODE_CONV || p.kh != 3 || p.kw != 3 || p.stride != 1 || p.pad != 1) return CX_EUNSUPPORTED;
  if (p.K != 128 || p.N != 32 || p.x_prologue != CX_PRO_AFFINE_RELU || p.g_prologue != CX_PRO_NONE) return CX_EUNSUPPORTED;
  if (p.W + 2 > 128 || p.W < 4 || p.Ho != p.H || p.Wo != p.W || (p.ldg % 8) || (p.ldx % 8)) return CX_EUNSUPPORTED;
  if ((unsigned long long)p.B * p.H * p.W * (unsigned long long)(p.ldx > p.ldg ? p.ldx : p.ldg) * 2ull >= (1ull << 32)) return CX_EUNSUPPORTED;
  for (int i = 0; i < n; ++i)
    if (!items->g[i] || !items->x[i] || !items->pa[i] || !items->pb[i] || !items->dw[i]) return CX_EINVAL;
  if (!p.scratch) return CX_EUNSUPPORTED;             // partial tiles go to slabs (ordered sums); no atomic form
  hipStream_t st = as_stream(stream);
  const int c_tiles = 4, n_tiles = 1, pairs = 4;
  static const int env_splits = cx_diag_int("CX_SW_BATCH_SPLITS", 0);
  // pixel-range splits per layer: the layers supply the parallelism, so a workgroup walks a long range (16 images or more) and the
  // partial tiles (147 KB per split and layer) stay a small fraction of the operands
  // (scratch/bench_w2batch.py, 256 images: 40x40 maps 104 / 81 / 84 / 87 us per layer with 8 / 16 / 32 / 64 splits -- 97 per layer
  // launched one by one --, 20x20 22.8 / 23.8 / 25.8 / 30.1 (33.0), 10x10 12.0 / 13.1 / 14.7 / 20.1 (21.3))
  // (two half-size workgroups per CU -- 384 threads, half the rows per step -- measured 44 % SLOWER on the 40x40 maps, 1194 against
  // 828 us for 12 layers: the kernel is bound by its phases, not by bytes in flight; scratch/stamps_strip.py has the shares)
  constexpr int nthr = 768;
  int target = env_splits > 0 ? env_splits : (p.H * p.W > 400 ? 16 : 8);
  int flat = NCHW4 * nthr / (p.W * 4) * (p.W + 2);
  if (flat > 880) flat = 880;
  StripGeo g = make_geo(p.B, p.H, p.W, target, 1, flat);
  const int nk = (g.R * g.P + 15) / 16;
  size_t smem = 160 * 4 + (size_t)(g.Q + 2) * WP + (size_t)nk * 16 * WP;
  if (smem < 160 * 4 + 32 * 288 * 4) smem = 160 * 4 + 32 * 288 * 4;
  if (!(g.R * p.W * 4 <= NCHW4 * nthr && 2 * p.W * 4 <= NCHW4 * nthr && g.Q >= 64 && smem <= 150 * 1024)) return CX_EUNSUPPORTED;
  const int total = g.B * g.spi;
  const int splits = (total + g.steps_per_wg - 1) / g.steps_per_wg;
  const long long wtotal = (long long)p.N * p.K * 9;
  const long long need = (long long)n * splits * wtotal;
  if (need > p.scratch_floats || need >= (1ll << 31)) return CX_EUNSUPPORTED;
  StripBatch bt;
  for (int i = 0; i < CX_WGRAD_BATCH_MAX; ++i) {
    const int j = i < n ? i : 0;
    bt.g[i] = (const bf16*)items->g[j];
    bt.x[i] = (const bf16*)items->x[j];
    bt.pa[i] = items->pa[j];
    bt.pb[i] = items->pb[j];
    bt.slab[i] = p.scratch + (size_t)j * splits * wtotal;
  }
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_strip_wgrad_batch_kernel<4, NCHW4>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    attr = true;
  }
  const int per_layer = splits * pairs;
  CX_KTAG("conv3x3_strip_wgrad_batch_kernel<4, %d>", NCHW4);
  hipLaunchKernelGGL((conv3x3_strip_wgrad_batch_kernel<4, NCHW4>), dim3(per_layer * n), dim3(768), smem, st, bt, per_layer, p.ldg, p.ldx,
                     p.K, p.N, c_tiles, n_tiles, g);
  if (const int e = launch_status()) return e;
  for (int i = 0; i < n; ++i)
    if (const int e = cx_dw_reduce(items->dw[i], bt.slab[i], (size_t)wtotal, splits, st)) return e;
  cx_tl_slab_floats_v = (int)need;
  return 0;
}

