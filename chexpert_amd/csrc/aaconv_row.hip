// Relative-position attention of AAConv2d (/root/reference/models/attn_aug_conv.py:43-100), second generation of the
// per-query kernels of aaconv.hip for the large maps (W = 40 and 20: transition 1 / 2 of the attention DenseNet at 320x320).
//
// aaconv.hip keeps the two relative-logit rows of every query (H + W floats, and W more for their gradient) in LDS: 60 KB per
// 128 queries in the forward pass and 94 KB in the backward pass -- one or two workgroups, 2-4 waves, per CU, and two per-lane
// LDS reads plus a read-modify-write per (query, key) pair.  Here keys stream one key ROW at a time and the map width is a
// template constant, so the column logits rw_i[kx] and their gradient live in registers (the kx loop is unrolled) and the row
// logit rh_i[ky] is one 20-term dot product per key row: LDS holds only the relative tables and the current key row
// (~20-30 KB per workgroup), the pair loop reads LDS only as broadcasts.  Same arithmetic and the same order of the
// per-query sums as aaconv.hip; other widths keep those kernels.
#include <cstdlib>
#include "common.h"

namespace {

constexpr int AQ = 128;      // queries per workgroup (one per thread)
constexpr int DKH = 20;      // head dim of q/k (aaconv.hip)

struct AAGeo {
  int B, H, W, nh, dk, dv, ldq;     // qkv tensor: (B, H*W, ldq) bf16, channels [q dk | k dk | v dv]
};

// stage key row ky: WW keys x (DKH k values, DVH v values) as fp32
template <int DVH, int WW>
__device__ __forceinline__ void stage_row(const bf16* __restrict__ base, int ldq, int kofs, int vofs, int ky, float* Kt, float* Vt, int tid) {
  const size_t j0 = (size_t)ky * WW;
  for (int t = tid; t < WW * 5; t += AQ) {
    const int j = t / 5, c = t - j * 5;
    U64 v;
    v.u = *reinterpret_cast<const uint2*>(base + (j0 + j) * ldq + kofs + c * 4);
#pragma unroll
    for (int e = 0; e < 4; ++e) Kt[j * DKH + c * 4 + e] = bf2f(v.e[e]);
  }
  for (int t = tid; t < WW * DVH; t += AQ) {
    const int j = t / DVH, d = t - j * DVH;
    Vt[t] = bf2f(base[(j0 + j) * ldq + vofs + d]);
  }
}

template <int DVH, int WW>
__global__ __launch_bounds__(AQ) void aa_attn_fwd_row_kernel(const bf16* __restrict__ qkv, const float* __restrict__ rel_h,
                                                            const float* __restrict__ rel_w, float* __restrict__ o,
                                                            float* __restrict__ lse, const AAGeo g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int H = g.H, HW = H * WW;
  const int LH = 2 * H - 1;
  constexpr int LW = 2 * WW - 1;
  float* RH = lds;                       // [DKH][LH]
  float* RW = RH + DKH * LH;             // [DKH][LW]
  float* Kt = RW + DKH * LW;             // [WW][DKH]
  float* Vt = Kt + WW * DKH;             // [WW][DVH]
  const int tid = threadIdx.x;
  const int bn = blockIdx.y, b = bn / g.nh, n = bn - b * g.nh;
  const int i = blockIdx.x * AQ + tid;
  const bool qvalid = i < HW;
  const int ic = qvalid ? i : HW - 1;
  const int qy = ic / WW, qx = ic - qy * WW;
  const bf16* base = qkv + (size_t)b * HW * g.ldq;
  for (int t = tid; t < DKH * LH; t += AQ) RH[t] = rel_h[t];
  for (int t = tid; t < DKH * LW; t += AQ) RW[t] = rel_w[t];
  float q[DKH];
  const float scale = rsqrtf((float)DKH);
  {
    const bf16* qp = base + (size_t)ic * g.ldq + n * DKH;
#pragma unroll
    for (int d = 0; d < DKH; d += 4) {
      U64 v;
      v.u = *reinterpret_cast<const uint2*>(qp + d);
#pragma unroll
      for (int e = 0; e < 4; ++e) q[d + e] = bf2f(v.e[e]) * scale;
    }
  }
  __syncthreads();
  float rw[WW];
#pragma unroll
  for (int kx = 0; kx < WW; ++kx) {
    float a = 0.f;
#pragma unroll
    for (int d = 0; d < DKH; ++d) a = fmaf(q[d], RW[d * LW + kx - qx + WW - 1], a);
    rw[kx] = a;
  }
  float m = -3.0e38f, l = 0.f, acc[DVH];
#pragma unroll
  for (int d = 0; d < DVH; ++d) acc[d] = 0.f;
  const int kofs = g.dk + n * DKH, vofs = 2 * g.dk + n * DVH;
  for (int ky = 0; ky < H; ++ky) {
    __syncthreads();
    stage_row<DVH, WW>(base, g.ldq, kofs, vofs, ky, Kt, Vt, tid);
    float rhv = 0.f;
#pragma unroll
    for (int d = 0; d < DKH; ++d) rhv = fmaf(q[d], RH[d * LH + ky - qy + H - 1], rhv);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < WW; ++j) {
      float s = rhv + rw[j];
      const float4* kp = reinterpret_cast<const float4*>(Kt + j * DKH);
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        const float4 kv = kp[c];
        s = fmaf(q[4 * c], kv.x, fmaf(q[4 * c + 1], kv.y, fmaf(q[4 * c + 2], kv.z, fmaf(q[4 * c + 3], kv.w, s))));
      }
      if (s > m) {
        const float c = __expf(m - s);
        l = fmaf(l, c, 1.f);
#pragma unroll
        for (int d = 0; d < DVH; ++d) acc[d] = fmaf(acc[d], c, Vt[j * DVH + d]);
        m = s;
      } else {
        const float p = __expf(s - m);
        l += p;
#pragma unroll
        for (int d = 0; d < DVH; ++d) acc[d] = fmaf(p, Vt[j * DVH + d], acc[d]);
      }
    }
  }
  if (qvalid) {
    const float inv = 1.f / l;
    float* op = o + ((size_t)b * HW + i) * g.dv + n * DVH;
#pragma unroll
    for (int d = 0; d < DVH; ++d) op[d] = acc[d] * inv;
    lse[(size_t)bn * HW + i] = m + __logf(l);
  }
}

// dq, d key_rel_h, d key_rel_w
template <int DVH, int WW>
__global__ __launch_bounds__(AQ) void aa_attn_bwd_q_row_kernel(const bf16* __restrict__ qkv, const float* __restrict__ rel_h,
                                                              const float* __restrict__ rel_w, const float* __restrict__ o,
                                                              const float* __restrict__ d_o, const float* __restrict__ lse,
                                                              float* __restrict__ dqkv, float* __restrict__ d_rel_h,
                                                              float* __restrict__ d_rel_w, float* __restrict__ slab_h, float* __restrict__ slab_w, const AAGeo g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int H = g.H, HW = H * WW;
  const int LH = 2 * H - 1;
  constexpr int LW = 2 * WW - 1;
  float* RH = lds;
  float* RW = RH + DKH * LH;
  float* dRH = RW + DKH * LW;            // workgroup partials
  float* dRW = dRH + DKH * LH;
  float* Kt = dRW + DKH * LW;
  float* Vt = Kt + WW * DKH;
  float* Qs = Vt + WW * DVH;             // [AQ][DKH + 1] scaled queries of the workgroup
  float* dr = Qs + AQ * (DKH + 1);       // [AQ] d rh_i[ky] of the key row just finished
  float* dwq = dr + AQ;                  // [AQ][WW + 1] d rw_i[kx], parked once at the end
  const int tid = threadIdx.x;
  const int bn = blockIdx.y, b = bn / g.nh, n = bn - b * g.nh;
  const int i = blockIdx.x * AQ + tid;
  const bool qvalid = i < HW;
  const int ic = qvalid ? i : HW - 1;
  const int qy = ic / WW, qx = ic - qy * WW;
  const bf16* base = qkv + (size_t)b * HW * g.ldq;
  for (int t = tid; t < DKH * LH; t += AQ) { RH[t] = rel_h[t]; dRH[t] = 0.f; }
  for (int t = tid; t < DKH * LW; t += AQ) { RW[t] = rel_w[t]; dRW[t] = 0.f; }
  float q[DKH];
  const float scale = rsqrtf((float)DKH);
  {
    const bf16* qp = base + (size_t)ic * g.ldq + n * DKH;
#pragma unroll
    for (int d = 0; d < DKH; d += 4) {
      U64 v;
      v.u = *reinterpret_cast<const uint2*>(qp + d);
#pragma unroll
      for (int e = 0; e < 4; ++e) q[d + e] = bf2f(v.e[e]) * scale;
    }
  }
  float dO[DVH], delta = 0.f;
  {
    const float* op = o + ((size_t)b * HW + ic) * g.dv + n * DVH;
    const float* dp = d_o + ((size_t)b * HW + ic) * g.dv + n * DVH;
#pragma unroll
    for (int d = 0; d < DVH; ++d) { dO[d] = qvalid ? dp[d] : 0.f; delta = fmaf(dO[d], op[d], delta); }
  }
  const float L = lse[(size_t)bn * HW + ic];
#pragma unroll
  for (int d = 0; d < DKH; ++d) Qs[tid * (DKH + 1) + d] = qvalid ? q[d] : 0.f;
  const int i0 = blockIdx.x * AQ;                      // first query of the workgroup; its image rows qy_a .. qy_b
  const int qy_a = i0 / WW, qy_b = min(i0 + AQ - 1, HW - 1) / WW;
  __syncthreads();
  float rw[WW], drw[WW], dq[DKH];
#pragma unroll
  for (int kx = 0; kx < WW; ++kx) {
    float a = 0.f;
#pragma unroll
    for (int d = 0; d < DKH; ++d) a = fmaf(q[d], RW[d * LW + kx - qx + WW - 1], a);
    rw[kx] = a;
    drw[kx] = 0.f;
  }
#pragma unroll
  for (int d = 0; d < DKH; ++d) dq[d] = 0.f;
  const int kofs = g.dk + n * DKH, vofs = 2 * g.dk + n * DVH;
  for (int ky = 0; ky < H; ++ky) {
    __syncthreads();
    stage_row<DVH, WW>(base, g.ldq, kofs, vofs, ky, Kt, Vt, tid);
    const int r = ky - qy + H - 1;
    float rhv = 0.f;
#pragma unroll
    for (int d = 0; d < DKH; ++d) rhv = fmaf(q[d], RH[d * LH + r], rhv);
    __syncthreads();
    float drh = 0.f;
#pragma unroll
    for (int j = 0; j < WW; ++j) {
      float s = rhv + rw[j];
      const float4* kp = reinterpret_cast<const float4*>(Kt + j * DKH);
      float4 kv[5];
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        kv[c] = kp[c];
        s = fmaf(q[4 * c], kv[c].x, fmaf(q[4 * c + 1], kv[c].y, fmaf(q[4 * c + 2], kv[c].z, fmaf(q[4 * c + 3], kv[c].w, s))));
      }
      const float p = __expf(s - L);
      float dp = 0.f;
#pragma unroll
      for (int d = 0; d < DVH; ++d) dp = fmaf(dO[d], Vt[j * DVH + d], dp);
      const float ds = p * (dp - delta);
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        dq[4 * c] = fmaf(ds, kv[c].x, dq[4 * c]);
        dq[4 * c + 1] = fmaf(ds, kv[c].y, dq[4 * c + 1]);
        dq[4 * c + 2] = fmaf(ds, kv[c].z, dq[4 * c + 2]);
        dq[4 * c + 3] = fmaf(ds, kv[c].w, dq[4 * c + 3]);
      }
      drh += ds;
      drw[j] += ds;
    }
    // key row complete: fold d rh_i[ky] into dq and into the workgroup's d key_rel_h partial.  Queries of one image row share
    // the table column r; LDS float atomics cost ~7 cycles per lane even on distinct words and ~45 on the same word (they were
    // half of this kernel), so no atomics: each query parks its scalar, then thread (image row, d) sums its row's products
    // and owns the word it adds to.
    const float drh_v = qvalid ? drh : 0.f;
    if (qvalid) {                          // its own basic block: merged into the unrolled pair loop's block it spilled 2 KB per lane
#pragma unroll
      for (int d = 0; d < DKH; ++d) dq[d] = fmaf(drh, RH[d * LH + r], dq[d]);
    }
    dr[tid] = drh_v;
    __syncthreads();
    for (int oo = tid; oo < (qy_b - qy_a + 1) * DKH; oo += AQ) {
      const int seg = oo / DKH, d = oo - seg * DKH, yy = qy_a + seg;
      const int l0 = max(yy * WW - i0, 0), l1 = min((yy + 1) * WW - i0, AQ);
      float t = 0.f;
      for (int l = l0; l < l1; ++l) t = fmaf(dr[l], Qs[l * (DKH + 1) + d], t);
      dRH[d * LH + ky - yy + H - 1] += t;
    }
  }
  // d rw_i[kx] -> dq and d key_rel_w.  Park the per-query column sums in LDS (plain stores), then one thread per table word
  // sums every (query, key column) pair that lands on it: no atomics.
#pragma unroll
  for (int kx = 0; kx < WW; ++kx) {
    const float dv_ = qvalid ? drw[kx] : 0.f;
    const int r = kx - qx + WW - 1;
#pragma unroll
    for (int d = 0; d < DKH; ++d) dq[d] = fmaf(dv_, RW[d * LW + r], dq[d]);
    dwq[tid * (WW + 1) + kx] = dv_;
  }
  __syncthreads();
  for (int oo = tid; oo < LW * DKH; oo += AQ) {        // thread owns table word (d, rr): every (query, kx) with kx - qx + W - 1 = rr
    const int rr = oo / DKH, d = oo - rr * DKH;
    int xq = i0 % WW;                                   // qx of query l
    float t = 0.f;
    for (int l = 0; l < AQ; ++l) {
      const int kx = xq + rr - (WW - 1);
      if (kx >= 0 && kx < WW) t = fmaf(dwq[l * (WW + 1) + kx], Qs[l * (DKH + 1) + d], t);
      if (++xq == WW) xq = 0;
    }
    dRW[d * LW + rr] += t;
  }
  if (qvalid) {
    float* dqp = dqkv + ((size_t)b * HW + i) * (2 * g.dk + g.dv) + n * DKH;
#pragma unroll
    for (int d = 0; d < DKH; ++d) dqp[d] = dq[d] * scale;       // q~ = q * scale
  }
  __syncthreads();
  {   // the workgroup's partial tables: one slab per workgroup (summed in workgroup order afterwards) or fp32 atomics
    const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    for (int t = tid; t < DKH * LH; t += AQ) { if (slab_h) slab_h[wg * (DKH * LH) + t] = dRH[t]; else atomicAdd(&d_rel_h[t], dRH[t]); }
    for (int t = tid; t < DKH * LW; t += AQ) { if (slab_w) slab_w[wg * (DKH * LW) + t] = dRW[t]; else atomicAdd(&d_rel_w[t], dRW[t]); }
  }
}

// dk, dv: one lane per KEY, queries stream one query ROW at a time.  aaconv.hip re-reads the two relative tables per
// (key, query, d): 40 per-lane LDS reads per pair.  Per query row the row term folds into the key (kr = k + RH[:, ky-qy+H-1],
// 20 reads per row) and the column term is the query's own logit row rw_i[.], built once per row by the whole workgroup
// (WW x WW x 20 FMAs) and read back with ONE per-lane LDS read per pair.
template <int DVH, int WW>
__global__ __launch_bounds__(AQ) void aa_attn_bwd_k_row_kernel(const bf16* __restrict__ qkv, const float* __restrict__ rel_h,
                                                              const float* __restrict__ rel_w, const float* __restrict__ o,
                                                              const float* __restrict__ d_o, const float* __restrict__ lse,
                                                              float* __restrict__ dqkv, const AAGeo g) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int H = g.H, HW = H * WW;
  const int LH = 2 * H - 1;
  constexpr int LW = 2 * WW - 1;
  float* RH = lds;
  float* RW = RH + DKH * LH;
  float* Qt = RW + DKH * LW;             // [WW][DKH] scaled queries of the row
  float* Dt = Qt + WW * DKH;             // [WW][DVH] dO
  float* Et = Dt + WW * DVH;             // [WW][2]   delta, lse
  float* rwT = Et + 2 * WW;              // [WW][WW+1] rw_i[kx] of the row's queries
  const int tid = threadIdx.x;
  const int bn = blockIdx.y, b = bn / g.nh, n = bn - b * g.nh;
  const int j = blockIdx.x * AQ + tid;
  const bool kvalid = j < HW;
  const int jc = kvalid ? j : HW - 1;
  const int ky = jc / WW, kx = jc - ky * WW;
  const bf16* base = qkv + (size_t)b * HW * g.ldq;
  for (int t = tid; t < DKH * LH; t += AQ) RH[t] = rel_h[t];
  for (int t = tid; t < DKH * LW; t += AQ) RW[t] = rel_w[t];
  float k[DKH], v[DVH], dk[DKH], dv[DVH];
  {
    const bf16* kp = base + (size_t)jc * g.ldq + g.dk + n * DKH;
#pragma unroll
    for (int d = 0; d < DKH; d += 4) {
      U64 u;
      u.u = *reinterpret_cast<const uint2*>(kp + d);
#pragma unroll
      for (int e = 0; e < 4; ++e) k[d + e] = bf2f(u.e[e]);
    }
#pragma unroll
    for (int d = 0; d < DVH; ++d) v[d] = bf2f(base[(size_t)jc * g.ldq + 2 * g.dk + n * DVH + d]);
  }
#pragma unroll
  for (int d = 0; d < DKH; ++d) dk[d] = 0.f;
#pragma unroll
  for (int d = 0; d < DVH; ++d) dv[d] = 0.f;
  const float scale = rsqrtf((float)DKH);
  for (int qy = 0; qy < H; ++qy) {
    __syncthreads();
    const size_t i0 = (size_t)qy * WW;
    for (int t = tid; t < WW * 5; t += AQ) {
      const int ii = t / 5, c = t - ii * 5;
      U64 u;
      u.u = *reinterpret_cast<const uint2*>(base + (i0 + ii) * g.ldq + n * DKH + c * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) Qt[ii * DKH + c * 4 + e] = bf2f(u.e[e]) * scale;
    }
    for (int t = tid; t < WW; t += AQ) {
      float de = 0.f;
#pragma unroll
      for (int d = 0; d < DVH; ++d) {
        const float dd = d_o[((size_t)b * HW + i0 + t) * g.dv + n * DVH + d];
        Dt[t * DVH + d] = dd;
        de = fmaf(dd, o[((size_t)b * HW + i0 + t) * g.dv + n * DVH + d], de);
      }
      Et[2 * t] = de;
      Et[2 * t + 1] = lse[(size_t)bn * HW + i0 + t];
    }
    __syncthreads();
    for (int t = tid; t < WW * WW; t += AQ) {           // rw_i[kx] = q~_i . key_rel_w[:, kx - qx + W - 1], i = (qy, qx)
      const int qx = t / WW, kk = t - qx * WW;
      float a = 0.f;
#pragma unroll
      for (int d = 0; d < DKH; ++d) a = fmaf(Qt[qx * DKH + d], RW[d * LW + kk - qx + WW - 1], a);
      rwT[qx * (WW + 1) + kk] = a;
    }
    float kr[DKH];
    const int rhh = ky - qy + H - 1;
#pragma unroll
    for (int d = 0; d < DKH; ++d) kr[d] = k[d] + RH[d * LH + rhh];
    __syncthreads();
#pragma unroll 4
    for (int ii = 0; ii < WW; ++ii) {
      const float4* qp = reinterpret_cast<const float4*>(Qt + ii * DKH);
      float s = rwT[ii * (WW + 1) + kx];
      float4 qv[5];
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        qv[c] = qp[c];
        s = fmaf(qv[c].x, kr[4 * c], fmaf(qv[c].y, kr[4 * c + 1], fmaf(qv[c].z, kr[4 * c + 2], fmaf(qv[c].w, kr[4 * c + 3], s))));
      }
      const float p = __expf(s - Et[2 * ii + 1]);
      float dp = 0.f;
#pragma unroll
      for (int d = 0; d < DVH; ++d) {
        dp = fmaf(Dt[ii * DVH + d], v[d], dp);
        dv[d] = fmaf(p, Dt[ii * DVH + d], dv[d]);
      }
      const float ds = p * (dp - Et[2 * ii]);
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        dk[4 * c] = fmaf(ds, qv[c].x, dk[4 * c]);
        dk[4 * c + 1] = fmaf(ds, qv[c].y, dk[4 * c + 1]);
        dk[4 * c + 2] = fmaf(ds, qv[c].z, dk[4 * c + 2]);
        dk[4 * c + 3] = fmaf(ds, qv[c].w, dk[4 * c + 3]);
      }
    }
  }
  if (kvalid) {
    float* dp = dqkv + ((size_t)b * HW + j) * (2 * g.dk + g.dv);
#pragma unroll
    for (int d = 0; d < DKH; ++d) dp[g.dk + n * DKH + d] = dk[d];
#pragma unroll
    for (int d = 0; d < DVH; ++d) dp[2 * g.dk + n * DVH + d] = dv[d];
  }
}

// ------------------------------------------------------------------------------------------------ query side on MFMA (W = 40)
// The per-pair dot products of aa_attn_bwd_q_row_kernel (20 FMAs for the logit, 20 for dq += ds * k: 40 of its ~50 vector
// instructions per (query, key) pair) move to the matrix pipe.  A wave owns 32 queries; a key row (40 keys, padded to 2 x 32) is
// one tile:  S^T = K Q^T  (keys in accumulator rows, queries in lanes: lane (q, half) sees 16 + 4 of the row's 40 keys, always the
// same columns kx, so its relative-column logits rw_q[kx] and their gradients are 20 registers indexed at compile time),
// then per element  p = exp(S + rh + rw - lse),  ds = p (dO . v - delta),  and  dQ += dS K  with dS moved from accumulator to
// operand layout by v_permlane32_swap as one fp16 term (see aa_op below; dq is checked to 1e-3).
// The relative-table gradients are skewed matrix products on the same pipe after the key loop.
constexpr int AQM = 128;               // queries per workgroup (4 waves x 32)
constexpr int KB_PITCH = 80;           // bf16 key image: 32 d (20 used) + 16 B pad

typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// Operand type of the two backward kernels' matrix products.  q and k arrive as bf16; dS = p (dP - delta) is produced in fp32 and has
// to become a 16-bit operand of dQ += dS K / dK += dS^T Q.  As bf16 that costs 2^-9 per term (measured 1.3e-3 / 1.7e-3 rms on dq / dk --
// as much as rounding the RESULT to bf16) unless it is split into hi + lo terms (rounds 3-4: 2^-17, at +20 vector instructions and one
// more MFMA per 16 keys).  As fp16 one term carries 2^-12: q and k convert exactly (8-bit mantissas; magnitudes below 6e-5 land on
// the fp16 subnormal grid, 3e-8 absolute; saturated at +-65504), the products and their fp32 sums are the same numbers the bf16 MFMA
// of the forward produced, and v_cvt_pk_f16_f32 costs what v_cvt_pk_bf16_f32 does.  AA_BWD_F16 = 0 restores bf16 hi + lo.
#ifndef AA_BWD_F16
#define AA_BWD_F16 1
#endif
#if AA_BWD_F16
typedef _Float16 aa_op;
typedef f16x8 aa_opx8;
#define AA_MFMA __builtin_amdgcn_mfma_f32_32x32x16_f16
#define AA_DS_TERMS 1
__device__ __forceinline__ aa_op aa_to_op(float v) { return (aa_op)__builtin_amdgcn_fmed3f(v, -65504.f, 65504.f); }
__device__ __forceinline__ uint32_t pk_op(float a, float b) {
  union { f16x2v h; uint32_t u; } o;
  o.h = __builtin_convertvector(f32x2v{a, b}, f16x2v);
  return o.u;
}
// four bf16 (as loaded) -> four operand elements
__device__ __forceinline__ uint2 aa_ops_of_bf4(const uint2 v) {
  return make_uint2(pk_op(__builtin_amdgcn_fmed3f(cx_bf_lo(v.x), -65504.f, 65504.f), __builtin_amdgcn_fmed3f(cx_bf_hi(v.x), -65504.f, 65504.f)),
                    pk_op(__builtin_amdgcn_fmed3f(cx_bf_lo(v.y), -65504.f, 65504.f), __builtin_amdgcn_fmed3f(cx_bf_hi(v.y), -65504.f, 65504.f)));
}
#else
typedef bf16 aa_op;
typedef bf16x8 aa_opx8;
#define AA_MFMA __builtin_amdgcn_mfma_f32_32x32x16_bf16
#ifndef AA_DS_TERMS
#define AA_DS_TERMS 2                  // bf16 terms of dS in the dQ / dK products: 2 = hi + lo (2^-17 relative), 1 = hi only (2^-9)
#endif
__device__ __forceinline__ aa_op aa_to_op(float v) { return f2bf(v); }
__device__ __forceinline__ uint32_t pk_op(float a, float b) {
  union { bf16x2v h; uint32_t u; } o;
  o.h = __builtin_convertvector(f32x2v{a, b}, bf16x2v);
  return o.u;
}
__device__ __forceinline__ uint2 aa_ops_of_bf4(const uint2 v) { return v; }
#endif
__device__ __forceinline__ float aa_lo_of(float v, uint32_t packed, int half) {        // v - (element `half` of `packed` as a float)
  union { uint32_t u; aa_op e[2]; } o;
  o.u = packed;
  return v - (float)o.e[half];
}
// dS is proportional to the gradient dO that arrives at the attention output, and a training step's dO can be anywhere (1e-3 .. 1e-8:
// loss averaging, depth): as an fp16 operand dS = p (dO . v - delta) would sink into the subnormals (6e-8) and flush.  Both backward
// kernels therefore scale dO and delta by a power of two that brings the workgroup's largest |dO| into [1, 2) -- exact -- carry the
// factor through everything that is linear in dS (dQ, dK, dV, the table gradients) and remove it at the outputs.  bf16 operands
// (AA_BWD_F16 = 0) have fp32's exponent range and the factor is 1.
struct AaScale { float up, down; };
__device__ __forceinline__ AaScale aa_scale_of(const float amax) {
  AaScale r;
  const uint32_t eb = (__float_as_uint(amax) >> 23) & 0xffu;
  const bool ok = AA_BWD_F16 && eb >= 1u && eb <= 253u;
  r.up = ok ? __uint_as_float((254u - eb) << 23) : 1.f;       // 2^(127 - eb): amax * up in [1, 2)
  r.down = ok ? __uint_as_float(eb << 23) : 1.f;
  return r;
}
// workgroup maximum of a non-negative value through four LDS words (256 threads); contains one barrier
__device__ __forceinline__ float aa_wg_max(float v, float* slot, const int lane, const int wave) {
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) v = fmaxf(v, __shfl_xor(v, d));
  if (lane == 0) slot[wave] = v;
  __syncthreads();
  return fmaxf(fmaxf(slot[0], slot[1]), fmaxf(slot[2], slot[3]));
}
__device__ __forceinline__ aa_opx8 tr_frag_k(const char* tile, int pitch, int k0, int lane) {
  // B operand of D[q][d] += dS[q][k] K[k][d]: this lane gets column d = lane & 31, keys k0 + 8 * (lane >> 5) + 0..7
  const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const char* base = tile + (k0 + 8 * (g >> 1) + q) * pitch + (16 * (g & 1) + 4 * pp) * 2;
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  // (joined by a shuffle + bit cast: assembled element by element the compiler emits a v_bfi per dword on the loaded registers and
  // waits for the read right where it is issued, not where the MFMA uses it -- common.h cx_join_tr)
  return __builtin_bit_cast(aa_opx8, cx_join_tr(__builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base)), __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(base + 4 * pitch))));
}

template <int DVH, int WW>
__global__ __launch_bounds__(256, 3) void aa_attn_bwd_q_mfma_kernel(const bf16* __restrict__ qkv, const float* __restrict__ rel_h,
                                                                const float* __restrict__ rel_w, const float* __restrict__ o,
                                                                const float* __restrict__ d_o, const float* __restrict__ lse,
                                                                float* __restrict__ dqkv, float* __restrict__ d_rel_h,
                                                                float* __restrict__ d_rel_w, float* __restrict__ slab_h, float* __restrict__ slab_w, const AAGeo g) {
  // WW = 40: 16 + 4 keys per lane (two 32-key tiles, the second one a quarter full); WW = 20: 12 slots per lane in one tile, the
  // last four of the upper lane half (kx 20..23) past the row: their logit offset is -inf, so p = ds = 0
  static_assert(WW == 40 || WW == 20, "key rows of 40 or 20");
  constexpr int LW = 2 * WW - 1, NT = 256;
  constexpr int NE = WW == 40 ? 20 : 12;               // accumulator slots of a key row seen by one lane
  constexpr int NG16 = WW == 40 ? 3 : 2;               // 16-key groups of dQ += dS K
  constexpr float LOG2E = 1.4426950408889634f;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int H = g.H, HW = H * WW;
  const int LH = 2 * H - 1;
  float* RH = lds;
  float* RW = RH + DKH * LH;
  // (round 5: three workgroups per CU instead of two -- 168 VGPRs under the launch bound -- and the table-gradient sums moved from
  // owner-computes vector loops (13 K instructions per wave, more than the whole key loop) to the matrix pipe, fp32 operands
  // (v_mfma_f32_32x32x2_f32: exact products, fixed summation order): their results are staged IN PLACE of the tables they are
  // the gradients of, which are dead by then, so the workgroup needs no separate partial-sum arrays: 62 -> 50 KB of LDS.)
  constexpr int KR = WW == 40 ? 48 : 32; // rows of a key image: the row's keys, then zero rows up to the last one an operand read touches
  constexpr int VR = WW == 40 ? 40 : 24; // rows of a value image
  constexpr int DRP = AQM + 1;           // pitch of dr2 (the prologue's stores walk down a column)
  float* Vt = RW + DKH * LW;             // 2 x [VR][DVH] fp32 (rows past the key row zero)
  float* Qs = Vt + 2 * VR * DVH;         // [AQM][DKH + 1] scaled queries of the workgroup
  float* dwq = Qs + AQM * (DKH + 1);     // before key row ky: dr2[ky][DRP] = rh_q[ky] (the relative row logit, from the prologue); after it:
                                         // d rh_q[ky]; after the loop [AQM][WW + 1] d rw_q[kx]; at the very end [AQM][DKH + 1] relative-term
                                         // part of dq: AQM * max(H, WW + 1) + H (rounded up to 4) floats
  float* dr2 = dwq;
  char* Kb = reinterpret_cast<char*>(dwq + AQM * (H > WW + 1 ? H : WW + 1) + ((H + 3) & ~3));     // 2 x bf16 [KR keys][KB_PITCH], 16-byte aligned
  float* Smax = reinterpret_cast<float*>(Kb + 2 * KR * KB_PITCH);                                  // 4 words: the waves' largest |dO|
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 31, lh = lane >> 5;
  const int ql = wave * 32 + lrow;                              // query of this lane inside the workgroup
  const int bn = blockIdx.y, b = bn / g.nh, n = bn - b * g.nh;
  const int i0 = blockIdx.x * AQM;
  const int i = i0 + ql;
  const bool qvalid = i < HW;
  const int ic = qvalid ? i : HW - 1;
  const int qy = ic / WW, qx = ic - qy * WW;
  const bf16* base = qkv + (size_t)b * HW * g.ldq;
  for (int t = tid; t < DKH * LH; t += NT) RH[t] = rel_h[t];
  for (int t = tid; t < DKH * LW; t += NT) RW[t] = rel_w[t];
  const size_t wg_ = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
  for (int t = tid; t < 2 * KR * KB_PITCH / 4; t += NT) reinterpret_cast<uint32_t*>(Kb)[t] = 0u;
  for (int t = tid; t < 2 * VR * DVH; t += NT) Vt[t] = 0.f;

  // the query: bf16 operand fragments (B operand of S^T = K Q^T: d = kk * 16 + lh * 8 + 0..7) and fp32 scaled copy
  float q[DKH];
  const float scale = rsqrtf((float)DKH);
  aa_opx8 qf[2];
  {
    const bf16* qp = base + (size_t)ic * g.ldq + n * DKH;
    aa_op qb[DKH];
#pragma unroll
    for (int d = 0; d < DKH; d += 4) {
      U64 v;
      v.u = *reinterpret_cast<const uint2*>(qp + d);
#pragma unroll
      for (int e = 0; e < 4; ++e) { qb[d + e] = aa_to_op(bf2f(v.e[e])); q[d + e] = bf2f(v.e[e]) * scale; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      qf[0][e] = lh ? qb[8 + e] : qb[e];
      qf[1][e] = (lh == 0 && e < 4) ? qb[16 + e] : aa_to_op(0.f);
    }
  }
  float dO[DVH], delta = 0.f;
  {
    const float* op = o + ((size_t)b * HW + ic) * g.dv + n * DVH;
    const float* dp = d_o + ((size_t)b * HW + ic) * g.dv + n * DVH;
#pragma unroll
    for (int d = 0; d < DVH; ++d) { dO[d] = qvalid ? dp[d] : 0.f; delta = fmaf(dO[d], op[d], delta); }
  }
  const float Ll = lse[(size_t)bn * HW + ic] * LOG2E;
  if (lh == 0) {
#pragma unroll
    for (int d = 0; d < DKH; ++d) Qs[ql * (DKH + 1) + d] = qvalid ? q[d] : 0.f;
  }
  // the workgroup's gradient scale (see AaScale): dS and everything summed from it carry `up` until the outputs
  float amax = 0.f;
#pragma unroll
  for (int d = 0; d < DVH; ++d) amax = fmaxf(amax, fabsf(dO[d]));
  const AaScale gs = aa_scale_of(aa_wg_max(amax, Smax, lane, wave));       // (its barrier also publishes Qs, the tables and the zero fill)
#pragma unroll
  for (int d = 0; d < DVH; ++d) dO[d] *= gs.up;
  delta *= gs.up;

  // this lane's key columns: kx = (e & 3) + 8 * (e >> 2) + 4 * lh for e < 16 (keys 0..31), 32 + (e - 16) + 4 * lh after
  // (a rolled loop over d with the query read back from LDS: fully unrolled, the 400 table reads are hoisted and spilled)
  float rwl[NE], drwl[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) rwl[e] = drwl[e] = 0.f;
  {
    const float* rwb = RW - qx + WW - 1;
#pragma unroll 1
    for (int d = 0; d < DKH; ++d) {
      const float qd = Qs[ql * (DKH + 1) + d];
#pragma unroll
      for (int e = 0; e < NE; ++e) {
        const int kx = (e < 16 ? (e & 3) + 8 * (e >> 2) : 32 + (e - 16)) + 4 * lh;
        rwl[e] = fmaf(qd, rwb[d * LW + (kx < WW ? kx : WW - 1)], rwl[e]);
      }
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int kx = (e < 16 ? (e & 3) + 8 * (e >> 2) : 32 + (e - 16)) + 4 * lh;
      rwl[e] = kx < WW ? rwl[e] * LOG2E : -1.0e30f;
    }
  }
  // The relative row logits of the wave's 32 queries, rh_q[ky] = q . key_rel_h[:, ky - qy + H - 1] for every key row, on the matrix pipe
  // (fp32 operands): T[q][c] = sum_d Qs[q][d] RH[d][c] over 32-column tiles of the table, entry (q, c) parked at dr2[c - (H-1) + qy(q)][q]
  // where the key loop picks it up before it overwrites the slot with d rh_q[ky].
  {
    const int qw0 = i0 + wave * 32, qy0 = qw0 / WW, x0 = qw0 - qy0 * WW;
    const float* qrow = Qs + (wave * 32 + lrow) * (DKH + 1) + lh;
#pragma unroll 1
    for (int t = 0; t * 32 < LH; ++t) {
      const int c = t * 32 + lrow;
      const float* rcol = RH + lh * LH + min(c, LH - 1);
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int s2 = 0; s2 < DKH / 2; ++s2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qrow[2 * s2], rcol[2 * s2 * LH], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int qo = (r & 3) + 8 * (r >> 2) + 4 * lh;
        int yq = qy0 + (x0 + qo >= WW ? 1 : 0);
        if (WW < 32) yq += x0 + qo >= 2 * WW ? 1 : 0;
        const int ky = c - (H - 1) + yq;
        if (c < LH && (unsigned)ky < (unsigned)H) dr2[ky * DRP + wave * 32 + qo] = acc[r];
      }
    }
  }
  f32x16 dqa;                            // D[q][d]: rows = the wave's queries, columns = d (lane & 31)
#pragma unroll
  for (int r = 0; r < 16; ++r) dqa[r] = 0.f;
  float dqr[10];                         // relative-term part of dq for d = 10 * lh + 0..9 (lane = query layout)
#pragma unroll
  for (int d = 0; d < 10; ++d) dqr[d] = 0.f;
  const float sl = scale * LOG2E;
  const int kofs = g.dk + n * DKH, vofs = 2 * g.dk + n * DVH;

  // key row ky -> image (ky & 1): bf16 keys (5 chunks of 4 channels per key), fp32 values.  Requested at the top of the previous
  // row into registers (unconditional loads on clamped indices), stored to LDS at its bottom: the latency hides under the row
  const int sj = min(tid / 5, WW - 1), sc = tid - (tid / 5) * 5;
  const int vj = min(tid / DVH, WW - 1), vd = tid - (tid / DVH) * DVH;
  uint2 kreg;
  bf16 vreg;
  auto load_keys = [&](int ky) __attribute__((always_inline)) {
    const size_t j0 = (size_t)ky * WW;
    kreg = *reinterpret_cast<const uint2*>(base + (j0 + sj) * g.ldq + kofs + sc * 4);
    vreg = base[(j0 + vj) * g.ldq + vofs + vd];
  };
  auto store_keys = [&](const int img) __attribute__((always_inline)) {
    if (tid < WW * 5) *reinterpret_cast<uint2*>(Kb + (img * KR + sj) * KB_PITCH + sc * 8) = aa_ops_of_bf4(kreg);
    if (tid < WW * DVH) Vt[(img * VR + vj) * DVH + vd] = bf2f(vreg);
  };
  load_keys(0);
  store_keys(0);                          // (the zero fill and the prologue's barrier are behind us)
  load_keys(H > 1 ? 1 : 0);
  const int r1 = 32 + min(lrow, KR - 33); // second tile's key row of this lane (rows past the image: its last zero row)
  for (int ky = 0; ky < H; ++ky) {
    __syncthreads();                      // image ky & 1 is complete, and nobody reads the other one (row ky - 1) any more
    store_keys((ky + 1) & 1);             // row ky + 1, requested a row ago
    load_keys(ky + 2 < H ? ky + 2 : H - 1);  // in flight under this row's arithmetic
    const char* Kc = Kb + (ky & 1) * (KR * KB_PITCH);
    const float* Vc = Vt + (ky & 1) * (VR * DVH);
    const int r = ky - qy + H - 1;
    const float rhv = dr2[ky * DRP + ql];
    const float rhl = qvalid ? fmaf(rhv, LOG2E, -Ll) : -1.0e30f;         // (rows past the map: p = 0)

    // S^T tiles: keys 0..31 and 32..63
    f32x16 st0, st1;
#pragma unroll
    for (int e = 0; e < 16; ++e) st0[e] = st1[e] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const aa_opx8 k0 = *reinterpret_cast<const aa_opx8*>(Kc + lrow * KB_PITCH + kk * 32 + lh * 16);
      st0 = AA_MFMA(k0, qf[kk], st0, 0, 0, 0);
      if (WW == 40) {
        const aa_opx8 k1 = *reinterpret_cast<const aa_opx8*>(Kc + r1 * KB_PITCH + kk * 32 + lh * 16);
        st1 = AA_MFMA(k1, qf[kk], st1, 0, 0, 0);
      }
    }
    float ds[NE], drh = 0.f;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int kx = (e < 16 ? (e & 3) + 8 * (e >> 2) : 32 + (e - 16)) + 4 * lh;      // (slots past the row read zero rows of Vc)
      const float sv = e < 16 ? st0[e] : st1[e - 16];
      const float p = __builtin_amdgcn_exp2f(fmaf(sv, sl, rhl + rwl[e]));
      float dp = 0.f;
#pragma unroll
      for (int d = 0; d < DVH; ++d) dp = fmaf(dO[d], Vc[kx * DVH + d], dp);
      ds[e] = p * (dp - delta);
      drh += ds[e];
      drwl[e] += ds[e];
    }
    // dQ += dS K over the key groups [0,16), [16,32), [32,48): accumulator rows -> 8 consecutive keys per lane, hi + lo bf16
#pragma unroll
    for (int g16 = 0; g16 < NG16; ++g16) {
      float v[8];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        // (accumulator slots that do not exist - keys 40..47, or 24..31 of a 20-wide row - are zeros)
        const float a = 8 * g16 + r4 < NE ? ds[8 * g16 + r4] : 0.f;
        const float c = 8 * g16 + 4 + r4 < (WW == 40 ? 16 : NE) ? ds[8 * g16 + 4 + r4] : 0.f;
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(c), false, false);
        v[r4] = __uint_as_float(sw[0]);
        v[4 + r4] = __uint_as_float(sw[1]);
      }
      union { aa_opx8 h; uint32_t u[4]; } hi, lo;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        hi.u[j] = pk_op(v[2 * j], v[2 * j + 1]);
        if (AA_DS_TERMS > 1) lo.u[j] = pk_op(aa_lo_of(v[2 * j], hi.u[j], 0), aa_lo_of(v[2 * j + 1], hi.u[j], 1));
      }
      const aa_opx8 kt = tr_frag_k(Kc, KB_PITCH, g16 * 16, lane);
      dqa = AA_MFMA(hi.h, kt, dqa, 0, 0, 0);
      if (AA_DS_TERMS > 1) dqa = AA_MFMA(lo.h, kt, dqa, 0, 0, 0);
    }
    // key row complete: d rh_q[ky] of the query (both lane halves) folds into dq now and is parked for the d key_rel_h sums
    drh += __shfl_xor(drh, 32);
#pragma unroll
    for (int d = 0; d < 10; ++d) dqr[d] = fmaf(drh, RH[(10 * lh + d) * LH + r], dqr[d]);
    if (lh == 0) dr2[ky * DRP + ql] = qvalid ? drh : 0.f;       // parked for the d key_rel_h sums after the loop
  }
  __syncthreads();
  // Table gradients as skewed matrix products over the workgroup's AQM queries l, one 32-row tile of table columns rr per wave:
  //   d key_rel_h[d][rr] = sum_l A[rr][l] Qs[l][d],  A[rr][l] = dr2[rr + yy(l) - (H-1)][l]      (yy = image row of query l)
  //   d key_rel_w[d][rr] = sum_l A[rr][l] Qs[l][d],  A[rr][l] = dwq[l][rr + xq(l) - (W-1)]      (xq = image column of query l)
  // v_mfma_f32_32x32x2_f32: lane (i = lane & 31, k = lane >> 5) supplies A[i][k] and B[k][j = lane & 31]; the 64 steps take
  // queries l = 2 s + k in index order.  Output columns j >= DKH and rows rr past the table are never stored.
  auto table_tile = [&](const bool by_row, const int L) __attribute__((always_inline)) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if (wave * 32 < L) {
      const int rr = wave * 32 + lrow;
      int l = lh, yy = (i0 + lh) / WW, xq = (i0 + lh) - yy * WW;
      const float* qcol = Qs + min(lrow, DKH);
#pragma unroll 8
      for (int s2 = 0; s2 < AQM / 2; ++s2) {
        const int kk = by_row ? rr + yy - (H - 1) : rr + xq - (WW - 1);
        const int kc = min(max(kk, 0), (by_row ? H : WW) - 1);
        const float av = by_row ? dr2[kc * DRP + l] : dwq[l * (WW + 1) + kc];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kk == kc ? av : 0.f, qcol[l * (DKH + 1)], acc, 0, 0, 0);
        l += 2;
        xq += 2;
        if (xq >= WW) { xq -= WW; ++yy; }
      }
    }
    return acc;
  };
  auto table_store = [&](const f32x16& acc, float* out, const int L) __attribute__((always_inline)) {
    if (lrow < DKH) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (rr < L) out[lrow * L + rr] = acc[r] * gs.down;
      }
    }
  };
  {
    const f32x16 th = table_tile(true, LH);
    table_store(th, RH, LH);              // key_rel_h is no longer read: its gradient takes its place
  }
  __syncthreads();                        // dr2 consumed: its space becomes dwq
  // d rw_q[kx] -> dq and d key_rel_w
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int kx = (e < 16 ? (e & 3) + 8 * (e >> 2) : 32 + (e - 16)) + 4 * lh;
    if (kx < WW) dwq[ql * (WW + 1) + kx] = drwl[e];
  }
  __syncthreads();
#pragma unroll 4
  for (int kx = 0; kx < WW; ++kx) {
    const float dv_ = dwq[ql * (WW + 1) + kx];
    const int rr = kx - qx + WW - 1;
#pragma unroll
    for (int d = 0; d < 10; ++d) dqr[d] = fmaf(dv_, RW[(10 * lh + d) * LW + rr], dqr[d]);
  }
  const f32x16 tw = table_tile(false, LW);
  __syncthreads();                        // every wave is done with key_rel_w and dwq
  table_store(tw, RW, LW);
  // dq = (matrix part [query rows][d columns] + relative part [query lanes][d]) * scale, through LDS
  float* dqx = dwq;                                     // [AQM][DKH + 1]
#pragma unroll
  for (int d = 0; d < 10; ++d) dqx[ql * (DKH + 1) + 10 * lh + d] = dqr[d];
  __syncthreads();
  if (lrow < DKH) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qq = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (i0 + qq < HW) {
        float* dqp = dqkv + ((size_t)b * HW + i0 + qq) * (2 * g.dk + g.dv) + n * DKH;
        dqp[lrow] = (dqa[r] + dqx[qq * (DKH + 1) + lrow]) * (scale * gs.down);
      }
    }
  }
  // (RH and RW are adjacent: one flat range of DKH * (LH + LW) words, the layout of the slab row pair)
  for (int t = tid; t < DKH * LH; t += NT) { if (slab_h) slab_h[wg_ * (DKH * LH) + t] = RH[t]; else atomicAdd(&d_rel_h[t], RH[t]); }
  for (int t = tid; t < DKH * LW; t += NT) { if (slab_w) slab_w[wg_ * (DKH * LW) + t] = RW[t]; else atomicAdd(&d_rel_w[t], RW[t]); }
}

// ------------------------------------------------------------------------------------------------ key side on MFMA
// The mirror image of the query-side kernel: a wave owns 32 KEYS (B operand, registers for the whole kernel), queries stream as
// flat 32-query tiles:  S = Q K^T  (queries in accumulator rows, keys in lanes), per element  p = exp2(S*sl + G + U),
// ds = p (dO . v - delta),  dv += p dO  on the vector pipe (lane = key),  dK += dS^T Q  by MFMA with dS moved from accumulator
// to operand layout by v_permlane32_swap and split into hi + lo bf16, exactly as the query side does for dQ.
// What the query side keeps in registers per lane -- the relative logits of ITS query -- is here a function of (query, key)
// with the query changing per accumulator row, so the two relative terms of a tile are built as small tables in LDS, one tile
// ahead, also on the matrix pipe:
//   G[q][kx] = scale * q . key_rel_w[:, kx - qx + W - 1]   = a skewed window of Q (32 x 20) x RW (20 x 2W-1): waves 0..2 own
//              one 32-column tile of the product each (RW as hi + lo bf16 fragments in registers) and store the entries that
//              fall into their query's window at [q][kx]
//   U[q][kyl] = scale * q . key_rel_h[:, ky0 + kyl - qy + H - 1] - lse[q]  for the (at most NKR) key rows the workgroup's 128
//              keys touch: wave 3, Q x (32 consecutive columns of RH, hi + lo bf16 images in LDS)
// so the pair loop reads two table words and one [dO | delta] broadcast per pair: ~14 vector instructions per (query, key)
// against ~50 of aa_attn_bwd_k_row_kernel.  Each key's sums run over the query tiles in index order inside one workgroup:
// no atomics, nothing order-dependent.
template <int DVH, int WW>
__global__ __launch_bounds__(256, (WW == 40 && DVH <= 2) ? 4 : 3) void aa_attn_bwd_k_mfma_kernel(const bf16* __restrict__ qkv, const float* __restrict__ rel_h,
                                                                const float* __restrict__ rel_w, const float* __restrict__ o,
                                                                const float* __restrict__ d_o, const float* __restrict__ lse,
                                                                float* __restrict__ dqkv, const AAGeo g) {
  static_assert(WW == 40 || WW == 20, "key rows of 40 or 20");
  constexpr int LW = 2 * WW - 1, NT = 256;
  constexpr int NGT = (LW + 31) / 32;                  // 32-column tiles of Q x RW: 3 / 2
  constexpr int NKR = 127 / WW + 2;                    // key rows 128 consecutive keys can touch: 5 / 8
  constexpr int DP = DVH + 1;                          // [dO | delta] per query
  constexpr int GP = WW + 1, UP = NKR + 6;             // table pitches (G: one dump column; U: the row shifts)
  constexpr float LOG2E = 1.4426950408889634f;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int H = g.H, HW = H * WW, LH = 2 * H - 1;
  char* Qi = reinterpret_cast<char*>(lds);                                   // 3 x bf16 [32][KB_PITCH] raw queries of a tile
  float* Gs = reinterpret_cast<float*>(Qi + 3 * 32 * KB_PITCH);              // 2 x [32][GP]
  float* Us = Gs + 2 * 32 * GP;                                              // 2 x [32][UP]
  float* Dd = Us + 2 * 32 * UP;                                              // 3 x [32][DP]
  float* Ls = Dd + 3 * 32 * DP;                                              // 3 x [32]: -lse * log2(e) of the tile's queries (-1e30 past the map)
  char* RHhi = reinterpret_cast<char*>(Ls + 3 * 32);                         // bf16 [LH][KB_PITCH]: key_rel_h^T, hi and lo parts
  char* RHlo = RHhi + (size_t)LH * KB_PITCH;
  float* Smax = reinterpret_cast<float*>(RHlo + (size_t)LH * KB_PITCH);       // 4 words: the waves' largest |dO|
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 31, lh = lane >> 5;
  const int bn = blockIdx.y, b = bn / g.nh, n = bn - b * g.nh;
  const int j0 = blockIdx.x * 128;
  const int j = j0 + wave * 32 + lrow;                                       // this lane's key (both lane halves)
  const bool kvalid = j < HW;
  const int jc = kvalid ? j : HW - 1;
  const int ky0 = j0 / WW;
  const int kyl = jc / WW - ky0, kxl = jc - (jc / WW) * WW;
  const bf16* base = qkv + (size_t)b * HW * g.ldq;
  const float scale = rsqrtf((float)DKH);
  const float sl = scale * LOG2E;
  const int ntl = (HW + 31) / 32;

  // the gradient scale of this (image, head) (see AaScale): every query's dO meets this workgroup's keys
  AaScale gs;
  {
    float amax = 0.f;
    const float* dob = d_o + (size_t)b * HW * g.dv + n * DVH;
    for (int t = tid; t < HW * DVH; t += NT) {
      const int i = t / DVH, d = t - i * DVH;
      amax = fmaxf(amax, fabsf(dob[(size_t)i * g.dv + d]));
    }
    gs = aa_scale_of(aa_wg_max(amax, Smax, lane, wave));
  }
  for (int t = tid; t < 3 * 32 * KB_PITCH / 4; t += NT) reinterpret_cast<uint32_t*>(Qi)[t] = 0u;
  for (int t = tid; t < LH * 32; t += NT) {
    const int r = t >> 5, d = t & 31;
    const float v = d < DKH ? rel_h[d * LH + r] : 0.f;
    const aa_op hi = aa_to_op(v);
    *reinterpret_cast<aa_op*>(RHhi + r * KB_PITCH + d * 2) = hi;
    *reinterpret_cast<aa_op*>(RHlo + r * KB_PITCH + d * 2) = aa_to_op(v - (float)hi);
  }
  // key_rel_w columns 32 * wave + lrow as B-operand fragments (k = d), hi + lo
  aa_opx8 rwhi[2], rwlo[2];
  {
    const int rr = min(32 * wave + lrow, LW - 1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int d = kk * 16 + lh * 8 + e;
        const float v = (d < DKH && wave < NGT) ? rel_w[d * LW + rr] * sl : 0.f;      // (the logit scale rides on the table operand)
        const aa_op hi = aa_to_op(v);
        rwhi[kk][e] = hi;
        rwlo[kk][e] = aa_to_op(v - (float)hi);
      }
  }
  // the key: operand fragments (B operand of S = Q K^T: d = kk * 16 + lh * 8 + 0..7) and the fp32 values
  aa_opx8 kf[2];
  float v[DVH], dv[DVH];
  {
    const bf16* kp = base + (size_t)jc * g.ldq + g.dk + n * DKH;
    aa_op kb[DKH];
#pragma unroll
    for (int d = 0; d < DKH; d += 4) {
      U64 u;
      u.u = *reinterpret_cast<const uint2*>(kp + d);
#pragma unroll
      for (int e = 0; e < 4; ++e) kb[d + e] = aa_to_op(bf2f(u.e[e]));
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      kf[0][e] = lh ? kb[8 + e] : kb[e];
      kf[1][e] = (lh == 0 && e < 4) ? kb[16 + e] : aa_to_op(0.f);
    }
#pragma unroll
    for (int d = 0; d < DVH; ++d) { v[d] = bf2f(base[(size_t)jc * g.ldq + 2 * g.dk + n * DVH + d]); dv[d] = 0.f; }
  }
  f32x16 dka;                            // D[key][d]: rows = the wave's keys, columns = d (lane & 31)
#pragma unroll
  for (int r = 0; r < 16; ++r) dka[r] = 0.f;

  // a tile's raw inputs: threads 0..159 one 8-byte chunk of a query, threads 160..191 one query's dO / o / lse
  const int sq = tid < 160 ? tid / 5 : (tid - 160) & 31, sc = tid - (tid / 5) * 5;
  uint2 qreg = make_uint2(0u, 0u);
  float dreg[DP], lreg = 0.f;
#pragma unroll
  for (int d = 0; d < DP; ++d) dreg[d] = 0.f;
  auto load_tile = [&](int u) __attribute__((always_inline)) {
    const int i = min(u * 32 + sq, HW - 1);
    if (tid < 160) {
      qreg = *reinterpret_cast<const uint2*>(base + (size_t)i * g.ldq + n * DKH + sc * 4);
    } else if (tid < 192) {
      const float* op = o + ((size_t)b * HW + i) * g.dv + n * DVH;
      const float* dp = d_o + ((size_t)b * HW + i) * g.dv + n * DVH;
      float de = 0.f;
#pragma unroll
      for (int d = 0; d < DVH; ++d) { dreg[d] = dp[d] * gs.up; de = fmaf(dp[d], op[d], de); }
      dreg[DVH] = de * gs.up;
      lreg = u * 32 + sq < HW ? -lse[(size_t)bn * HW + i] * LOG2E : -1.0e30f;
    }
  };
  auto store_tile = [&](int u) __attribute__((always_inline)) {
    const int bq = u % 3;
    if (tid < 160) {
      *reinterpret_cast<uint2*>(Qi + (bq * 32 + sq) * KB_PITCH + sc * 8) = aa_ops_of_bf4(qreg);
    } else if (tid < 192) {
#pragma unroll
      for (int d = 0; d < DP; ++d) Dd[(bq * 32 + sq) * DP + d] = dreg[d];
      Ls[bq * 32 + sq] = lreg;
    }
  };
  // the relative-logit tables of tile u (its queries are in Qi[u % 3]).  Every store is unconditional: an entry outside its query's
  // window goes to the row's dump column, the row shifts of U are an index offset.
  const int role = __builtin_amdgcn_readfirstlane(wave);
  auto build_tables = [&](int u) __attribute__((always_inline)) {
    const int bq = u % 3, bt = u & 1;
    const int y0 = (u * 32) / WW, x0 = u * 32 - y0 * WW;
    const int m = x0 + 4 * lh;                                        // query column before wrapping = m + qo
    if (role < NGT) {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const aa_opx8 a = *reinterpret_cast<const aa_opx8*>(Qi + (bq * 32 + lrow) * KB_PITCH + kk * 32 + lh * 16);
        acc = AA_MFMA(a, rwhi[kk], acc, 0, 0, 0);
        acc = AA_MFMA(a, rwlo[kk], acc, 0, 0, 0);
      }
      const int kb = 32 * role + lrow - (WW - 1) + m;                 // kx = kb + qo - WW * wraps
      float* gdst = Gs + (bt * 32 + 4 * lh) * GP;
      // (W = 40: a tile starts at a column that is a multiple of 8, so whether query qo + 4 lh is past the end of its image row
      // is the same for both lane halves and all four queries of an accumulator group -- a scalar per group, not a compare and a
      // select per entry)
      const int jw = (WW - x0) >> 3;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int qo = (e & 3) + 8 * (e >> 2);
        int kx;
        if (WW == 40) {
          kx = kb + (qo - ((e >> 2) >= jw ? WW : 0));
        } else {
          kx = kb + qo;
          kx -= m + qo >= WW ? WW : 0;
          kx -= m + qo >= 2 * WW ? WW : 0;
        }
        gdst[qo * GP + min((unsigned)kx, (unsigned)WW)] = acc[e];
      }
    } else if (role == 3) {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      const int dmax = min(u * 32 + 31, HW - 1) / WW - y0;            // image rows of the tile's queries: y0 .. y0 + dmax
      const int rrow = min(ky0 - (y0 + dmax) + H - 1 + lrow, LH - 1);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const aa_opx8 a = *reinterpret_cast<const aa_opx8*>(Qi + (bq * 32 + lrow) * KB_PITCH + kk * 32 + lh * 16);
        const aa_opx8 bh = *reinterpret_cast<const aa_opx8*>(RHhi + rrow * KB_PITCH + kk * 32 + lh * 16);
        const aa_opx8 bl = *reinterpret_cast<const aa_opx8*>(RHlo + rrow * KB_PITCH + kk * 32 + lh * 16);
        acc = AA_MFMA(a, bh, acc, 0, 0, 0);
        acc = AA_MFMA(a, bl, acc, 0, 0, 0);
      }
      // column n of the product is key row ky0 + n - (dmax - wraps): stored at n - dmax + wraps + 2, read at kyl + 2
      if (lrow < NKR + 2) {
        float* udst = Us + (bt * 32 + 4 * lh) * UP + lrow - dmax + 2;
        const float* nl = Ls + bq * 32 + 4 * lh;
        const int jw = (WW - x0) >> 3;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int qo = (e & 3) + 8 * (e >> 2);
          int c;
          if (WW == 40) {
            c = (e >> 2) >= jw ? 1 : 0;                                // (scalar, see the G table)
          } else {
            c = m + qo >= WW ? 1 : 0;
            c += m + qo >= 2 * WW ? 1 : 0;
          }
          c = min(c, dmax);                                            // (queries past the map wrap further: same window, nl = -1e30)
          udst[qo * UP + c] = fmaf(acc[e], sl, nl[qo]);
        }
      }
    }
  };

  load_tile(0);
  __syncthreads();                         // zero fill of Qi done
  store_tile(0);
  load_tile(min(1, ntl - 1));
  __syncthreads();
  store_tile(1);
  build_tables(0);
  load_tile(min(2, ntl - 1));
  __syncthreads();
  for (int t = 0; t < ntl; ++t) {
    // tiles t and t + 1 are in Qi, the tables of tile t are built, tile t + 2 is in registers
    store_tile(t + 2);
    load_tile(min(t + 3, ntl - 1));
    if (t + 1 < ntl) build_tables(t + 1);
    const int bq = t % 3, bt = t & 1;
    f32x16 st;
#pragma unroll
    for (int e = 0; e < 16; ++e) st[e] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const aa_opx8 a = *reinterpret_cast<const aa_opx8*>(Qi + (bq * 32 + lrow) * KB_PITCH + kk * 32 + lh * 16);
      st = AA_MFMA(a, kf[kk], st, 0, 0, 0);
    }
    const float* gp = Gs + (bt * 32 + 4 * lh) * GP + kxl;
    const float* up = Us + (bt * 32 + 4 * lh) * UP + kyl + 2;
    const float* dp = Dd + (bq * 32 + 4 * lh) * DP;
    float ds[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int qo = (e & 3) + 8 * (e >> 2);
      const float p = __builtin_amdgcn_exp2f(fmaf(st[e], sl, gp[qo * GP] + up[qo * UP]));
      float a = -dp[qo * DP + DVH];
#pragma unroll
      for (int d = 0; d < DVH; ++d) {
        const float dd = dp[qo * DP + d];
        a = fmaf(dd, v[d], a);
        dv[d] = fmaf(p, dd, dv[d]);
      }
      ds[e] = p * a;
    }
    // dK += dS^T Q over the query groups [0,16), [16,32): accumulator rows -> 8 consecutive queries per lane, hi + lo bf16
#pragma unroll
    for (int g16 = 0; g16 < 2; ++g16) {
      float w[8];
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(ds[8 * g16 + r4]), __float_as_uint(ds[8 * g16 + 4 + r4]), false, false);
        w[r4] = __uint_as_float(sw[0]);
        w[4 + r4] = __uint_as_float(sw[1]);
      }
      union { aa_opx8 h; uint32_t u[4]; } hi, lo;
#pragma unroll
      for (int jj = 0; jj < 4; ++jj) {
        hi.u[jj] = pk_op(w[2 * jj], w[2 * jj + 1]);
        if (AA_DS_TERMS > 1) lo.u[jj] = pk_op(aa_lo_of(w[2 * jj], hi.u[jj], 0), aa_lo_of(w[2 * jj + 1], hi.u[jj], 1));
      }
      const aa_opx8 qt = tr_frag_k(Qi + bq * 32 * KB_PITCH, KB_PITCH, g16 * 16, lane);
      dka = AA_MFMA(hi.h, qt, dka, 0, 0, 0);
      if (AA_DS_TERMS > 1) dka = AA_MFMA(lo.h, qt, dka, 0, 0, 0);
    }
    __syncthreads();
  }
  const int ctot = 2 * g.dk + g.dv;
#pragma unroll
  for (int d = 0; d < DVH; ++d) dv[d] += __shfl_xor(dv[d], 32);
  if (kvalid && lh == 0) {
    float* op = dqkv + ((size_t)b * HW + j) * ctot + 2 * g.dk + n * DVH;
#pragma unroll
    for (int d = 0; d < DVH; ++d) op[d] = dv[d] * gs.down;
  }
  if (lrow < DKH) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kk = j0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (kk < HW) dqkv[((size_t)b * HW + kk) * ctot + g.dk + n * DKH + lrow] = dka[r] * (scale * gs.down);
    }
  }
}

// Forward on the same tiles: S^T = K Q^T per key row by MFMA, online softmax per query over the lane's 20 (12) accumulator slots
// and its partner half (the two halves of a query share the running maximum), p V on the vector pipe (DVH <= 6).
template <int DVH, int WW>
__global__ __launch_bounds__(256, (WW == 20 || DVH <= 2) ? 4 : 3) void aa_attn_fwd_mfma_kernel(const bf16* __restrict__ qkv, const float* __restrict__ rel_h,
                                                              const float* __restrict__ rel_w, float* __restrict__ o,
                                                              float* __restrict__ lse, const AAGeo g) {
  static_assert(WW == 40 || WW == 20, "key rows of 40 or 20");
  constexpr int LW = 2 * WW - 1, NT = 256;
  constexpr int NE = WW == 40 ? 20 : 12;
  constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int H = g.H, HW = H * WW;
  const int LH = 2 * H - 1;
  float* RH = lds;
  float* RW = RH + DKH * LH;
  float* Vt = RW + DKH * LW;             // [64][DVH] fp32 (rows past the key row zero)
  char* Kb = reinterpret_cast<char*>(Vt + 64 * DVH);            // bf16 [64 keys][KB_PITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 31, lh = lane >> 5;
  const int ql = wave * 32 + lrow;
  const int bn = blockIdx.y, b = bn / g.nh, n = bn - b * g.nh;
  const int i = blockIdx.x * AQM + ql;
  const bool qvalid = i < HW;
  const int ic = qvalid ? i : HW - 1;
  const int qy = ic / WW, qx = ic - qy * WW;
  const bf16* base = qkv + (size_t)b * HW * g.ldq;
  for (int t = tid; t < DKH * LH; t += NT) RH[t] = rel_h[t];
  for (int t = tid; t < DKH * LW; t += NT) RW[t] = rel_w[t];
  for (int t = tid; t < 64 * KB_PITCH / 4; t += NT) reinterpret_cast<uint32_t*>(Kb)[t] = 0u;
  for (int t = tid; t < 64 * DVH; t += NT) Vt[t] = 0.f;
  float q[DKH];
  const float scale = rsqrtf((float)DKH);
  bf16x8 qf[2];
  {
    const bf16* qp = base + (size_t)ic * g.ldq + n * DKH;
    bf16 qb[DKH];
#pragma unroll
    for (int d = 0; d < DKH; d += 4) {
      U64 v;
      v.u = *reinterpret_cast<const uint2*>(qp + d);
#pragma unroll
      for (int e = 0; e < 4; ++e) { qb[d + e] = v.e[e]; q[d + e] = bf2f(v.e[e]) * scale; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      qf[0][e] = lh ? qb[8 + e] : qb[e];
      qf[1][e] = (lh == 0 && e < 4) ? qb[16 + e] : f2bf(0.f);
    }
  }
  __syncthreads();
  float rwl[NE];
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int kx = (e < 16 ? (e & 3) + 8 * (e >> 2) : 32 + (e - 16)) + 4 * lh;
    const int kc = kx < WW ? kx : WW - 1;
    float a = 0.f;
#pragma unroll
    for (int d = 0; d < DKH; ++d) a = fmaf(q[d], RW[d * LW + kc - qx + WW - 1], a);
    rwl[e] = kx < WW ? a * LOG2E : -1.0e30f;
  }
  float m = -3.0e38f, l = 0.f, acc[DVH];
#pragma unroll
  for (int d = 0; d < DVH; ++d) acc[d] = 0.f;
  const float sl = scale * LOG2E;
  const int kofs = g.dk + n * DKH, vofs = 2 * g.dk + n * DVH;
  const int sj = min(tid / 5, WW - 1), sc = tid - (tid / 5) * 5;
  const int vj = min(tid / DVH, WW - 1), vd = tid - (tid / DVH) * DVH;
  uint2 kreg;
  bf16 vreg;
  auto load_keys = [&](int ky) __attribute__((always_inline)) {
    const size_t j0 = (size_t)ky * WW;
    kreg = *reinterpret_cast<const uint2*>(base + (j0 + sj) * g.ldq + kofs + sc * 4);
    vreg = base[(j0 + vj) * g.ldq + vofs + vd];
  };
  load_keys(0);
  for (int ky = 0; ky < H; ++ky) {
    __syncthreads();
    if (tid < WW * 5) *reinterpret_cast<uint2*>(Kb + sj * KB_PITCH + sc * 8) = kreg;
    if (tid < WW * DVH) Vt[vj * DVH + vd] = bf2f(vreg);
    __syncthreads();
    load_keys(ky + 1 < H ? ky + 1 : ky);
    float rhv = 0.f;
#pragma unroll
    for (int d = 0; d < DKH; ++d) rhv = fmaf(q[d], RH[d * LH + ky - qy + H - 1], rhv);
    const float rhl = rhv * LOG2E;
    f32x16 st0, st1;
#pragma unroll
    for (int e = 0; e < 16; ++e) st0[e] = st1[e] = 0.f;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const bf16x8 k0 = *reinterpret_cast<const bf16x8*>(Kb + lrow * KB_PITCH + kk * 32 + lh * 16);
      st0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k0, qf[kk], st0, 0, 0, 0);
      if (WW == 40) {
        const bf16x8 k1 = *reinterpret_cast<const bf16x8*>(Kb + (32 + lrow) * KB_PITCH + kk * 32 + lh * 16);
        st1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k1, qf[kk], st1, 0, 0, 0);
      }
    }
    float s2[NE], mx = -3.0e38f;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      s2[e] = fmaf(e < 16 ? st0[e] : st1[e - 16], sl, rhl + rwl[e]);
      mx = fmaxf(mx, s2[e]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mn = fmaxf(m, mx);
    const float alpha = __builtin_amdgcn_exp2f(m - mn);
    m = mn;
    l *= alpha;
#pragma unroll
    for (int d = 0; d < DVH; ++d) acc[d] *= alpha;
#pragma unroll
    for (int e = 0; e < NE; ++e) {
      const int kx = (e < 16 ? (e & 3) + 8 * (e >> 2) : 32 + (e - 16)) + 4 * lh;
      const float p = __builtin_amdgcn_exp2f(s2[e] - mn);
      l += p;
#pragma unroll
      for (int d = 0; d < DVH; ++d) acc[d] = fmaf(p, Vt[kx * DVH + d], acc[d]);
    }
  }
  l += __shfl_xor(l, 32);
#pragma unroll
  for (int d = 0; d < DVH; ++d) acc[d] += __shfl_xor(acc[d], 32);
  if (qvalid && lh == 0) {
    const float inv = 1.f / l;
    float* op = o + ((size_t)b * HW + i) * g.dv + n * DVH;
#pragma unroll
    for (int d = 0; d < DVH; ++d) op[d] = acc[d] * inv;
    lse[(size_t)bn * HW + i] = m * LN2 + __logf(l);
  }
}

template <int DVH, int WW>
int launch_row(int which, const void* qkv, const float* rel_h, const float* rel_w, float* o, const float* d_o, float* lse, float* dqkv,
               float* d_rel_h, float* d_rel_w, float* slab_h, float* slab_w, const AAGeo& g, hipStream_t st) {
  const dim3 grid((g.H * WW + AQ - 1) / AQ, g.B * g.nh);
  const size_t tables = (size_t)DKH * (2 * g.H - 1 + 2 * WW - 1);
  if (which == 0) {
    static const bool f_row = cx_diag_set("CX_AA_F_ROW");          // diagnostic: the per-query VALU kernel
    if ((WW == 40 || WW == 20) && !f_row) {
      const size_t smem_m = (tables + 64 * DVH) * 4 + 64 * KB_PITCH;
      hipLaunchKernelGGL((aa_attn_fwd_mfma_kernel<DVH, WW>), dim3((g.H * WW + AQM - 1) / AQM, g.B * g.nh), dim3(256), smem_m, st,
                         (const bf16*)qkv, rel_h, rel_w, o, lse, g);
    } else {
      const size_t smem = (tables + (size_t)WW * (DKH + DVH)) * 4;
      hipLaunchKernelGGL((aa_attn_fwd_row_kernel<DVH, WW>), grid, dim3(AQ), smem, st, (const bf16*)qkv, rel_h, rel_w, o, lse, g);
    }
  } else {
    static const bool q_row = cx_diag_set("CX_AA_Q_ROW");          // diagnostic: the per-query VALU kernel
    if ((WW == 40 || WW == 20) && !q_row) {
      const size_t smem_m = (tables + 2 * (WW == 40 ? 40 : 24) * DVH + (size_t)AQM * (DKH + 1 + (g.H > WW + 1 ? g.H : WW + 1)) + ((g.H + 3) & ~3)) * 4 +
                            2 * (WW == 40 ? 48 : 32) * KB_PITCH + 16;
      static bool attr_m = false;
      if (!attr_m) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&aa_attn_bwd_q_mfma_kernel<DVH, WW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                  96 * 1024);
        attr_m = true;
      }
      hipLaunchKernelGGL((aa_attn_bwd_q_mfma_kernel<DVH, WW>), dim3((g.H * WW + AQM - 1) / AQM, g.B * g.nh), dim3(256), smem_m, st,
                         (const bf16*)qkv, rel_h, rel_w, o, d_o, lse, dqkv, d_rel_h, d_rel_w, slab_h, slab_w, g);
    } else {
      const size_t smem = (2 * tables + (size_t)WW * (DKH + DVH) + (size_t)AQ * (DKH + 2 + WW + 1)) * 4;
      hipLaunchKernelGGL((aa_attn_bwd_q_row_kernel<DVH, WW>), grid, dim3(AQ), smem, st, (const bf16*)qkv, rel_h, rel_w, o, d_o, lse, dqkv,
                         d_rel_h, d_rel_w, slab_h, slab_w, g);
    }
    static const bool k_row = cx_diag_set("CX_AA_K_ROW");          // diagnostic: the one-lane-per-key VALU kernel
    if ((WW == 40 || WW == 20) && !k_row) {
      constexpr int NKR = 127 / WW + 2;
      const size_t smem_k = (size_t)3 * 32 * KB_PITCH + ((size_t)2 * 32 * (WW + 1) + 2 * 32 * (NKR + 6) + 3 * 32 * (DVH + 1) + 3 * 32) * 4 +
                            (size_t)2 * (2 * g.H - 1) * KB_PITCH + 16;
      hipLaunchKernelGGL((aa_attn_bwd_k_mfma_kernel<DVH, WW>), dim3((g.H * WW + 127) / 128, g.B * g.nh), dim3(256), smem_k, st,
                         (const bf16*)qkv, rel_h, rel_w, o, d_o, lse, dqkv, g);
    } else {
      const size_t smem_k = (tables + (size_t)WW * (DKH + DVH + 2 + WW + 1)) * 4;
      hipLaunchKernelGGL((aa_attn_bwd_k_row_kernel<DVH, WW>), grid, dim3(AQ), smem_k, st, (const bf16*)qkv, rel_h, rel_w, o, d_o, lse, dqkv, g);
    }
  }
  return launch_status();
}

template <int WW>
int launch_row_w(int which, int dvh, const void* qkv, const float* rel_h, const float* rel_w, float* o, const float* d_o, float* lse,
                 float* dqkv, float* d_rel_h, float* d_rel_w, float* slab_h, float* slab_w, const AAGeo& g, hipStream_t st, bool* handled) {
  *handled = true;
  switch (dvh) {
    case 1: return launch_row<1, WW>(which, qkv, rel_h, rel_w, o, d_o, lse, dqkv, d_rel_h, d_rel_w, slab_h, slab_w, g, st);
    case 2: return launch_row<2, WW>(which, qkv, rel_h, rel_w, o, d_o, lse, dqkv, d_rel_h, d_rel_w, slab_h, slab_w, g, st);
    case 3: return launch_row<3, WW>(which, qkv, rel_h, rel_w, o, d_o, lse, dqkv, d_rel_h, d_rel_w, slab_h, slab_w, g, st);
    case 4: return launch_row<4, WW>(which, qkv, rel_h, rel_w, o, d_o, lse, dqkv, d_rel_h, d_rel_w, slab_h, slab_w, g, st);
    case 6: return launch_row<6, WW>(which, qkv, rel_h, rel_w, o, d_o, lse, dqkv, d_rel_h, d_rel_w, slab_h, slab_w, g, st);
    default: *handled = false; return 0;
  }
}

}  // namespace

// which: 0 forward (o, lse), 1 the whole backward (dq, dk, dv, d key_rel_h, d key_rel_w).  *handled = false: width not covered.
int cx_try_aa_row(int which, const void* qkv, const float* rel_h, const float* rel_w, float* o, const float* d_o, float* lse, float* dqkv,
                  float* d_rel_h, float* d_rel_w, float* slab_h, float* slab_w, int B, int H, int W, int nh, int dk, int dv, int ldq,
                  hipStream_t st, bool* handled) {
  *handled = false;
  if (H > 64) return 0;                  // tables: 20 * (2H-1 + 2W-1) floats per copy
  const AAGeo g{B, H, W, nh, dk, dv, ldq};
  const int dvh = dv / nh;
  if (W == 40) return launch_row_w<40>(which, dvh, qkv, rel_h, rel_w, o, d_o, lse, dqkv, d_rel_h, d_rel_w, slab_h, slab_w, g, st, handled);
  if (W == 20) return launch_row_w<20>(which, dvh, qkv, rel_h, rel_w, o, d_o, lse, dqkv, d_rel_h, d_rel_w, slab_h, slab_w, g, st, handled);
  return 0;
}
