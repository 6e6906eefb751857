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
                                                              float* __restrict__ d_rel_w, const AAGeo g) {
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
  for (int t = tid; t < DKH * LH; t += AQ) atomicAdd(&d_rel_h[t], dRH[t]);
  for (int t = tid; t < DKH * LW; t += AQ) atomicAdd(&d_rel_w[t], dRW[t]);
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

template <int DVH, int WW>
int launch_row(int which, const void* qkv, const float* rel_h, const float* rel_w, float* o, const float* d_o, float* lse, float* dqkv,
               float* d_rel_h, float* d_rel_w, const AAGeo& g, hipStream_t st) {
  const dim3 grid((g.H * WW + AQ - 1) / AQ, g.B * g.nh);
  const size_t tables = (size_t)DKH * (2 * g.H - 1 + 2 * WW - 1);
  if (which == 0) {
    const size_t smem = (tables + (size_t)WW * (DKH + DVH)) * 4;
    hipLaunchKernelGGL((aa_attn_fwd_row_kernel<DVH, WW>), grid, dim3(AQ), smem, st, (const bf16*)qkv, rel_h, rel_w, o, lse, g);
  } else {
    const size_t smem = (2 * tables + (size_t)WW * (DKH + DVH) + (size_t)AQ * (DKH + 2 + WW + 1)) * 4;
    hipLaunchKernelGGL((aa_attn_bwd_q_row_kernel<DVH, WW>), grid, dim3(AQ), smem, st, (const bf16*)qkv, rel_h, rel_w, o, d_o, lse, dqkv,
                       d_rel_h, d_rel_w, g);
    const size_t smem_k = (tables + (size_t)WW * (DKH + DVH + 2 + WW + 1)) * 4;
    hipLaunchKernelGGL((aa_attn_bwd_k_row_kernel<DVH, WW>), grid, dim3(AQ), smem_k, st, (const bf16*)qkv, rel_h, rel_w, o, d_o, lse, dqkv, g);
  }
  return launch_status();
}

template <int WW>
int launch_row_w(int which, int dvh, const void* qkv, const float* rel_h, const float* rel_w, float* o, const float* d_o, float* lse,
                 float* dqkv, float* d_rel_h, float* d_rel_w, const AAGeo& g, hipStream_t st, bool* handled) {
  *handled = true;
  switch (dvh) {
    case 1: return launch_row<1, WW>(which, qkv, rel_h, rel_w, o, d_o, lse, dqkv, d_rel_h, d_rel_w, g, st);
    case 2: return launch_row<2, WW>(which, qkv, rel_h, rel_w, o, d_o, lse, dqkv, d_rel_h, d_rel_w, g, st);
    case 3: return launch_row<3, WW>(which, qkv, rel_h, rel_w, o, d_o, lse, dqkv, d_rel_h, d_rel_w, g, st);
    case 4: return launch_row<4, WW>(which, qkv, rel_h, rel_w, o, d_o, lse, dqkv, d_rel_h, d_rel_w, g, st);
    case 6: return launch_row<6, WW>(which, qkv, rel_h, rel_w, o, d_o, lse, dqkv, d_rel_h, d_rel_w, g, st);
    default: *handled = false; return 0;
  }
}

}  // namespace

// which: 0 forward (o, lse), 1 the whole backward (dq, dk, dv, d key_rel_h, d key_rel_w).  *handled = false: width not covered.
int cx_try_aa_row(int which, const void* qkv, const float* rel_h, const float* rel_w, float* o, const float* d_o, float* lse, float* dqkv,
                  float* d_rel_h, float* d_rel_w, int B, int H, int W, int nh, int dk, int dv, int ldq, hipStream_t st, bool* handled) {
  *handled = false;
  if (H > 64) return 0;                  // tables: 20 * (2H-1 + 2W-1) floats per copy
  const AAGeo g{B, H, W, nh, dk, dv, ldq};
  const int dvh = dv / nh;
  if (W == 40) return launch_row_w<40>(which, dvh, qkv, rel_h, rel_w, o, d_o, lse, dqkv, d_rel_h, d_rel_w, g, st, handled);
  if (W == 20) return launch_row_w<20>(which, dvh, qkv, rel_h, rel_w, o, d_o, lse, dqkv, d_rel_h, d_rel_w, g, st, handled);
  return 0;
}
