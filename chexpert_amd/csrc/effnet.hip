// EfficientNet building blocks (/root/reference/models/efficientnet.py:27-131) that are not implicit GEMMs:
// depthwise k x k convolution (forward / input gradient / weight gradient) with the preceding BatchNorm + Swish
// applied on load, squeeze-and-excitation (global pool, two tiny FCs, per-(b,c) scaling), Swish / BatchNorm
// backward glue.  All of it is HBM-bound elementwise / stencil work: 16 B per lane along the channel axis,
// per-block LDS reduction before fp32 atomics.  The 1x1 expand / project / head convolutions go through
// cx_conv_gemm / cx_conv_wgrad on materialised bf16 activations.
#include <type_traits>
#include "common.h"

// dwconv.hip: tiled depthwise kernels for k in {3,5}, stride in {1,2}, pad = k/2 (which: 0 forward, 1 dgrad, 2 wgrad)
int cx_try_dw_tile(int which, const void* x, const float* w, const float* sc, const float* sh, const float* mean, const float* rstd,
                   const void* gq, const void* g2, const float* ga, const float* gb, const float* gc, void* y, void* dz, float* s1,
                   float* s2, float* dw, int accumulate, int B, int H, int W, int C, int k, int stride, int pad, int stat_rows,
                   float* scratch, long long scratch_floats, hipStream_t st, bool* handled);

namespace {

__device__ __forceinline__ float sigmoidf_(float z) { return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * z)); }
__device__ __forceinline__ float swishf_(float z) { return z * sigmoidf_(z); }
__device__ __forceinline__ float dswishf_(float z) {
  const float s = sigmoidf_(z);
  return s * (1.f + z * (1.f - s));
}

// eight consecutive per-channel floats as two 16-B loads (C % 8 == 0 keeps them aligned)
__device__ __forceinline__ void load8(const float* __restrict__ p, float (&o)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p);
  const f32x4 b = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
  for (int e = 0; e < 4; ++e) { o[e] = a[e]; o[4 + e] = b[e]; }
}

inline int grid_for(size_t n, int block, int cap) {
  size_t g = (n + block - 1) / block;
  if (g > (size_t)cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}
inline int threads_for(int CP) { return CP * (256 / CP > 0 ? 256 / CP : 1); }

// shared reduction of per-thread 8-channel partials (thread's chunk column cq is loop invariant).
// det == 0: LDS atomics, then one global atomic per channel (lds = NS * C floats, zeroed by the caller).
// det != 0: every thread parks its partials in its own LDS slot (lds = blockDim.x * NS * 8 floats), the threads sharing a chunk are
// added in thread order and the workgroup plain-stores statistic row `row` (dst[k][row * C + c], CxConv.stat_det convention):
// the same bits every run.
template <int NS>
__device__ __forceinline__ void flush_partials(float (&s)[NS][8], int cq, int C, float* lds, float* const (&dst)[NS], int det = 0,
                                               int row = 0) {
  if (det) {
    const int CP = C / 8;
    __syncthreads();                       // (the atomic mode's zero-fill of lds may still be in flight in other waves)
#pragma unroll
    for (int k = 0; k < NS; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) lds[(threadIdx.x * NS + k) * 8 + j] = s[k][j];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      const int chunk = c >> 3, j = c & 7;
#pragma unroll
      for (int k = 0; k < NS; ++k) {
        float t = 0.f;
        for (int th = chunk; th < (int)blockDim.x; th += CP) t += lds[(th * NS + k) * 8 + j];
        if (dst[k]) dst[k][(size_t)row * C + c] = t;
      }
    }
    return;
  }
#pragma unroll
  for (int k = 0; k < NS; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) atomicAdd(&lds[k * C + cq * 8 + j], s[k][j]);
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x)
#pragma unroll
    for (int k = 0; k < NS; ++k)
      if (dst[k]) atomicAdd(&dst[k][c], lds[k * C + c]);
}

template <typename T>
__global__ void nchw3_to_nhwc8_kernel(const float* __restrict__ x, T* __restrict__ y, size_t hw, size_t total) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const size_t b = idx / hw, p = idx - b * hw;
  const float* src = x + b * 3 * hw + p;
  float o_f[8];
  o_f[0] = V8<T>::rnd(src[0]);
  o_f[1] = V8<T>::rnd(src[hw]);
  o_f[2] = V8<T>::rnd(src[2 * hw]);
#pragma unroll
  for (int j = 3; j < 8; ++j) o_f[j] = V8<T>::rnd(0.f);
  V8<T>::st(y + idx * 8, o_f);
}

// decoded grey bytes -> the three identical whitened channels (chexpert.py:70-72), NHWC8 bf16 (channels 3..7 zero)
template <typename T>
__global__ void u8_to_nhwc8_kernel(const uint8_t* __restrict__ x, T* __restrict__ y, float scale, float shift, size_t total) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const float g = V8<T>::rnd(fmaf((float)x[idx], scale, shift));
  float o_f[8];
  o_f[0] = g; o_f[1] = g; o_f[2] = g;
#pragma unroll
  for (int j = 3; j < 8; ++j) o_f[j] = V8<T>::rnd(0.f);
  V8<T>::st(y + idx * 8, o_f);
}

// ---------------------------------------------------------------------------------------------- depthwise conv
// forward: y[p][c] = sum_t act(x[p@t][c]) * w[c][t],  act = swish(x*sc+sh) or identity (sc == nullptr)
template <typename T>
__global__ void dwconv_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w, const float* __restrict__ sc,
                                  const float* __restrict__ sh, T* __restrict__ y, float* g1, float* g2, int B, int H, int W, int C,
                                  int Ho, int Wo, int k, int stride, int pad, int det) {
  extern __shared__ float lds[];          // [2][C]
  const int CP = C / 8;
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  const int cq = threadIdx.x % CP;
  float s[2][8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[0][j] = s[1][j] = 0.f;
  float fsc[8], fsh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { fsc[j] = sc ? sc[cq * 8 + j] : 1.f; fsh[j] = sc ? sh[cq * 8 + j] : 0.f; }
  const size_t npix = (size_t)B * Ho * Wo, ppb = blockDim.x / CP;
  for (size_t pix = (size_t)blockIdx.x * ppb + threadIdx.x / CP; pix < npix; pix += (size_t)gridDim.x * ppb) {
    const int b = pix / ((size_t)Ho * Wo);
    const int rem = pix - (size_t)b * Ho * Wo;
    const int oy = rem / Wo, ox = rem - oy * Wo;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int dy = 0; dy < k; ++dy) {
      const int iy = oy * stride - pad + dy;
      if (iy < 0 || iy >= H) continue;
      for (int dx = 0; dx < k; ++dx) {
        const int ix = ox * stride - pad + dx;
        if (ix < 0 || ix >= W) continue;
        typename V8<T>::raw v;
        v = V8<T>::ld(x + ((size_t)(b * H + iy) * W + ix) * C + cq * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float a = V8<T>::get(v, j);
          if (sc) a = V8<T>::rnd(swishf_(fmaf(a, fsc[j], fsh[j])));       // same rounding as a materialised activation
          acc[j] = fmaf(a, w[(size_t)(cq * 8 + j) * k * k + dy * k + dx], acc[j]);
        }
      }
    }
    float o_f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      o_f[j] = V8<T>::rnd(acc[j]);
      const float rv = o_f[j];
      s[0][j] += rv;
      s[1][j] += rv * rv;
    }
    V8<T>::st(y + pix * C + cq * 8, o_f);
  }
  float* const dst[2] = {g1, g2};
  if (g1) flush_partials<2>(s, cq, C, lds, dst, det, (int)blockIdx.x);
}

// input gradient: da[p][c] = sum_t dY[(p + pad - t)/stride][c] * w[c][t];  dY = g*ga + g2*gb + gc
//   dz = da * swish'(x*sc+sh) (or da when sc == nullptr);  S1 += dz, S2 += dz * (x-mean)*rstd
template <typename T>
__global__ void dwconv_dgrad_kernel(const T* __restrict__ g, const T* __restrict__ g2, const float* __restrict__ ga,
                                    const float* __restrict__ gb, const float* __restrict__ gc, const float* __restrict__ w,
                                    const T* __restrict__ x, const float* __restrict__ sc, const float* __restrict__ sh,
                                    const float* __restrict__ mean, const float* __restrict__ rstd, T* __restrict__ dz, float* S1,
                                    float* S2, int B, int H, int W, int C, int Ho, int Wo, int k, int stride, int pad, int accumulate,
                                    int det) {
  extern __shared__ float lds[];
  const int CP = C / 8;
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  const int cq = threadIdx.x % CP;
  float s[2][8], fa[8], fb[8], fc[8], fsc[8], fsh[8], fmu[8], fr[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cq * 8 + j;
    s[0][j] = s[1][j] = 0.f;
    fa[j] = ga[c]; fb[j] = gb[c]; fc[j] = gc[c];
    fsc[j] = sc ? sc[c] : 1.f; fsh[j] = sc ? sh[c] : 0.f; fmu[j] = sc ? mean[c] : 0.f; fr[j] = sc ? rstd[c] : 0.f;
  }
  const size_t npix = (size_t)B * H * W, ppb = blockDim.x / CP;
  for (size_t pix = (size_t)blockIdx.x * ppb + threadIdx.x / CP; pix < npix; pix += (size_t)gridDim.x * ppb) {
    const int b = pix / ((size_t)H * W);
    const int rem = pix - (size_t)b * H * W;
    const int iy = rem / W, ix = rem - iy * W;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    for (int dy = 0; dy < k; ++dy) {
      const int ny = iy + pad - dy;
      if (ny < 0 || ny % stride) continue;
      const int oy = ny / stride;
      if (oy >= Ho) continue;
      for (int dx = 0; dx < k; ++dx) {
        const int nx = ix + pad - dx;
        if (nx < 0 || nx % stride) continue;
        const int ox = nx / stride;
        if (ox >= Wo) continue;
        const size_t op = ((size_t)b * Ho + oy) * Wo + ox;
        typename V8<T>::raw u, v;
        u = V8<T>::ld(g + op * C + cq * 8);
        v = V8<T>::ld(g2 + op * C + cq * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float dy_ = V8<T>::rnd(fmaf(V8<T>::get(u, j), fa[j], fmaf(V8<T>::get(v, j), fb[j], fc[j])));
          acc[j] = fmaf(dy_, w[(size_t)(cq * 8 + j) * k * k + dy * k + dx], acc[j]);
        }
      }
    }
    typename V8<T>::raw xv, old;
    float o_f[8];
    xv = V8<T>::ld(x + pix * C + cq * 8);
    if (accumulate) old = V8<T>::ld(dz + pix * C + cq * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xf = V8<T>::get(xv, j);
      float d = acc[j];
      if (sc) d *= dswishf_(fmaf(xf, fsc[j], fsh[j]));
      s[0][j] += d;
      s[1][j] += d * (xf - fmu[j]) * fr[j];
      if (accumulate) d += V8<T>::get(old, j);
      o_f[j] = V8<T>::rnd(d);
    }
    V8<T>::st(dz + pix * C + cq * 8, o_f);
  }
  float* const dst[2] = {S1, S2};
  if (S1) flush_partials<2>(s, cq, C, lds, dst, det, (int)blockIdx.x);
}

// weight gradient: dW[c][t] += sum_p dY[p][c] * act(x[p@t][c]); one tap per blockIdx.y
template <typename T>
__global__ void dwconv_wgrad_kernel(const T* __restrict__ g, const T* __restrict__ g2, const float* __restrict__ ga,
                                    const float* __restrict__ gb, const float* __restrict__ gc, const T* __restrict__ x,
                                    const float* __restrict__ sc, const float* __restrict__ sh, float* __restrict__ dw, int B, int H, int W,
                                    int C, int Ho, int Wo, int k, int stride, int pad, float* __restrict__ slab) {
  extern __shared__ float lds[];          // [C]
  const int CP = C / 8;
  for (int i = threadIdx.x; i < C; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  const int cq = threadIdx.x % CP;
  const int tap = blockIdx.y, dy = tap / k, dx = tap - dy * k;
  float s[1][8], fa[8], fb[8], fc[8], fsc[8], fsh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cq * 8 + j;
    s[0][j] = 0.f;
    fa[j] = ga[c]; fb[j] = gb[c]; fc[j] = gc[c];
    fsc[j] = sc ? sc[c] : 1.f; fsh[j] = sc ? sh[c] : 0.f;
  }
  const size_t npix = (size_t)B * Ho * Wo, ppb = blockDim.x / CP;
  for (size_t pix = (size_t)blockIdx.x * ppb + threadIdx.x / CP; pix < npix; pix += (size_t)gridDim.x * ppb) {
    const int b = pix / ((size_t)Ho * Wo);
    const int rem = pix - (size_t)b * Ho * Wo;
    const int oy = rem / Wo, ox = rem - oy * Wo;
    const int iy = oy * stride - pad + dy, ix = ox * stride - pad + dx;
    if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
    typename V8<T>::raw u, v, xv;
    u = V8<T>::ld(g + pix * C + cq * 8);
    v = V8<T>::ld(g2 + pix * C + cq * 8);
    xv = V8<T>::ld(x + ((size_t)(b * H + iy) * W + ix) * C + cq * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float dy_ = V8<T>::rnd(fmaf(V8<T>::get(u, j), fa[j], fmaf(V8<T>::get(v, j), fb[j], fc[j])));
      float a = V8<T>::get(xv, j);
      if (sc) a = V8<T>::rnd(swishf_(fmaf(a, fsc[j], fsh[j])));
      s[0][j] = fmaf(dy_, a, s[0][j]);
    }
  }
  if (slab) {        // reproducible: thread slots folded in thread order, one partial (row blockIdx.x of the slab) per workgroup and tap
    const int CPs = C / 8;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) lds[threadIdx.x * 8 + j] = s[0][j];
    __syncthreads();
    const size_t total = (size_t)C * k * k;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      float t = 0.f;
      for (int th = c >> 3; th < (int)blockDim.x; th += CPs) t += lds[th * 8 + (c & 7)];
      slab[(size_t)blockIdx.x * total + (size_t)c * k * k + tap] = t;
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) atomicAdd(&lds[cq * 8 + j], s[0][j]);
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) atomicAdd(&dw[(size_t)c * k * k + tap], lds[c]);
}

// ---------------------------------------------------------------------------------------------- SE / activation glue
// pooled[b][c] = mean_hw act(x*sc+sh), act: 0 none, 1 relu, 2 swish
template <typename T>
__global__ void gap_affine_act_kernel(const T* __restrict__ x, const float* __restrict__ sc, const float* __restrict__ sh,
                                      float* __restrict__ pooled, int HW, int C, int act, int splits, float* __restrict__ rows) {
  // four pixel rows of a thread are requested before the first is consumed (one 16-B load in flight per lane left this kernel
  // at 0.8 TB/s); the rows of a workgroup meet in LDS, one global atomic per channel and workgroup
  extern __shared__ float lds[];
  constexpr int U = 4;
  const int CP = C / 8;
  const int b = blockIdx.y, sp = blockIdx.x;
  const int cq = threadIdx.x % CP, rr = threadIdx.x / CP, rpp = blockDim.x / CP;
  const int len = (HW + splits - 1) / splits, p0 = sp * len, p1 = min(HW, p0 + len);
  for (int i = threadIdx.x; i < C; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  float a[8], fsc[8], fsh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = 0.f; fsc[j] = sc[cq * 8 + j]; fsh[j] = sh[cq * 8 + j]; }
  const T* __restrict__ xb = x + (size_t)b * HW * C + cq * 8;
  for (int p = p0 + rr; p < p1; p += U * rpp) {
    typename V8<T>::raw v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int pp = p + u * rpp;
      v[u] = V8<T>::ld(xb + (size_t)(pp < p1 ? pp : p1 - 1) * C);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = p + u * rpp < p1;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float z = fmaf(V8<T>::get(v[u], j), fsc[j], fsh[j]);
        const float t = act == 2 ? swishf_(z) : (act == 1 ? fmaxf(z, 0.f) : z);
        a[j] += ok ? t : 0.f;
      }
    }
  }
  const float inv = 1.f / HW;
  if (rows) {        // reproducible: pixel lanes folded in lane order, the split's partial mean plain-stored into row `sp` (rows summed in
                     // row order by the launch that follows)
    __syncthreads();
    for (int j = 0; j < 8; ++j) lds[threadIdx.x * 8 + j] = a[j];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      float t = 0.f;
      for (int th = c >> 3; th < (int)blockDim.x; th += CP) t += lds[th * 8 + (c & 7)];
      rows[((size_t)sp * gridDim.y + b) * C + c] = t * inv;
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) atomicAdd(&lds[cq * 8 + j], a[j]);
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) atomicAdd(&pooled[(size_t)b * C + c], lds[c] * inv);
}

// SE excitation: h1 = W1 pooled + b1; s = sigmoid(W2 swish(h1) + b2)
// Round 5: grid (sample, channel slice).  One 1024-thread block per sample left 192 of the 256 CUs idle at bs = 64 and walked each W1
// row with ONE load in flight per lane (41 us per launch, 32 launches per B4 step).  Every block now computes the whole h1 of its
// sample (R x C multiply-adds: cheap) with the sample's pooled vector in LDS and four W1 rows per wave in flight, then its slice of the
// second product; block (b, 0) stores h1.  The sums keep their order (lane-strided partials, xor butterfly): bit-identical results.
__global__ __launch_bounds__(1024) void se_fwd_kernel(const float* __restrict__ pooled, const float* __restrict__ w1, const float* __restrict__ b1,
                              const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ h1, float* __restrict__ s,
                              int C, int R, int c_per_slice, const float* __restrict__ rows, int n_rows, float* __restrict__ pooled_out) {
  extern __shared__ float lds[];          // [R] swish(h1) | [C] pooled of this sample
  float* pl = lds + R;
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  if (rows) {
    // cx_gap_se_fwd: the pool's per-split partial means (row sp of sample b at rows[(sp * B + b) * C + c]) are added here, in row
    // order, instead of by a launch of their own between the pool and this kernel; block (b, 0) leaves the sum for the backward pass
    // (the order of cx_rows_reduce -- rows q, q + 64, ... first, then over q -- so that this form gives the same bits as the separate
    // launch did: on the B4 fixture a 1e-7 change of the pooled means re-draws enough bf16 roundings downstream to move the train
    // logits from 8.0e-3 to 1.2e-2 of the reference; both are draws of the same noise, but the recorded one stays)
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      float t = 0.f;
      for (int q = 0; q < 64 && q < n_rows; ++q) {
        float sa = 0.f;
        for (int sp = q; sp < n_rows; sp += 64) sa += rows[((size_t)sp * gridDim.x + b) * C + c];
        t += sa;
      }
      pl[c] = t;
      if (blockIdx.y == 0) pooled_out[(size_t)b * C + c] = t;
    }
  } else {
    for (int c = threadIdx.x; c < C; c += blockDim.x) pl[c] = pooled[(size_t)b * C + c];
  }
  __syncthreads();
  for (int r0 = wave * 4; r0 < R; r0 += nw * 4) {
    float a[4] = {0.f, 0.f, 0.f, 0.f};
    const float* wr[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wr[j] = w1 + (size_t)min(r0 + j, R - 1) * C;
    for (int c = lane; c < C; c += 64) {
      const float pv = pl[c];
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = fmaf(wr[j][c], pv, a[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) a[j] += __shfl_xor(a[j], d);
      if (lane == 0 && r0 + j < R) {
        const float v = a[j] + b1[r0 + j];
        if (blockIdx.y == 0) h1[(size_t)b * R + r0 + j] = v;
        lds[r0 + j] = swishf_(v);
      }
    }
  }
  __syncthreads();
  // eight lanes per channel: a W2 row (R contiguous floats) is read 32 B at a time, the partial sums meet in a fixed butterfly
  const int l8 = threadIdx.x & 7;
  const int c_lo = blockIdx.y * c_per_slice, c_hi = min(C, c_lo + c_per_slice);
  for (int c = c_lo + (threadIdx.x >> 3); c < c_hi; c += blockDim.x >> 3) {
    float a = 0.f;
    for (int r = l8; r < R; r += 8) a = fmaf(w2[(size_t)c * R + r], lds[r], a);
    a += __shfl_xor(a, 1); a += __shfl_xor(a, 2); a += __shfl_xor(a, 4);
    if (l8 == 0) s[(size_t)b * C + c] = sigmoidf_(a + b2[c]);
  }
}

// u = swish(x*sc+sh) * s[b][c]       (the tensor the projection conv consumes)
template <typename T>
__global__ void scale_act_bc_kernel(const T* __restrict__ x, const float* __restrict__ sc, const float* __restrict__ sh,
                                    const float* __restrict__ s, T* __restrict__ u, int HW, int C, size_t total) {
  const int CP = C / 8;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int cq = idx % CP;
    const size_t pix = idx / CP;
    const int b = pix / HW;
    typename V8<T>::raw v;
    float o_f[8];
    v = V8<T>::ld(x + idx * 8);
    float fsc[8], fsh[8], fs[8];
    load8(sc + cq * 8, fsc);
    load8(sh + cq * 8, fsh);
    if (s) load8(s + (size_t)b * C + cq * 8, fs);
#pragma unroll
    for (int j = 0; j < 8; ++j) o_f[j] = V8<T>::rnd(swishf_(fmaf(V8<T>::get(v, j), fsc[j], fsh[j])) * (s ? fs[j] : 1.f));
    V8<T>::st(u + idx * 8, o_f);
  }
}

// linear BatchNorm backward statistics: S1 += sum g, S2 += sum g * (y-mean)*rstd
template <typename T>
__global__ void bn_lin_bwd_stats_kernel(const T* __restrict__ g, const T* __restrict__ y, const float* __restrict__ mean,
                                        const float* __restrict__ rstd, float* S1, float* S2, size_t rows, int C, int det) {
  extern __shared__ float lds[];
  const int CP = C / 8;
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  const int cq = threadIdx.x % CP;
  float s[2][8], fmu[8], fr[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s[0][j] = s[1][j] = 0.f; fmu[j] = mean[cq * 8 + j]; fr[j] = rstd[cq * 8 + j]; }
  const size_t ppb = blockDim.x / CP, stride = (size_t)gridDim.x * ppb;
  constexpr int U = 4;                       // pixel rows in flight per thread
  for (size_t pix0 = (size_t)blockIdx.x * ppb + threadIdx.x / CP; pix0 < rows; pix0 += U * stride) {
    typename V8<T>::raw u[U], v[U];
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const size_t pix = pix0 + i * stride < rows ? pix0 + i * stride : pix0;
      u[i] = V8<T>::ld(g + pix * C + cq * 8);
      v[i] = V8<T>::ld(y + pix * C + cq * 8);
    }
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const bool ok = pix0 + i * stride < rows;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float gf = ok ? V8<T>::get(u[i], j) : 0.f;
        s[0][j] += gf;
        s[1][j] += gf * (V8<T>::get(v[i], j) - fmu[j]) * fr[j];
      }
    }
  }
  float* const dst[2] = {S1, S2};
  flush_partials<2>(s, cq, C, lds, dst, det, (int)blockIdx.x);
}

// ds[b][c] = sum_hw du * swish(x*sc+sh)
template <typename T>
__global__ void se_bwd_reduce_kernel(const T* __restrict__ du, const T* __restrict__ x, const float* __restrict__ sc,
                                     const float* __restrict__ sh, float* __restrict__ ds, int HW, int C, int splits, float* __restrict__ rows) {
  extern __shared__ float lds[];
  constexpr int U = 4;                       // pixel rows in flight per thread (see gap_affine_act_kernel)
  const int CP = C / 8;
  const int b = blockIdx.y, sp = blockIdx.x;
  const int cq = threadIdx.x % CP, rr = threadIdx.x / CP, rpp = blockDim.x / CP;
  const int len = (HW + splits - 1) / splits, p0 = sp * len, p1 = min(HW, p0 + len);
  for (int i = threadIdx.x; i < C; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  float a[8], fsc[8], fsh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = 0.f; fsc[j] = sc[cq * 8 + j]; fsh[j] = sh[cq * 8 + j]; }
  const size_t base = (size_t)b * HW * C + cq * 8;
  for (int p = p0 + rr; p < p1; p += U * rpp) {
    typename V8<T>::raw v[U], d[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int pp = p + u * rpp;
      const size_t off = base + (size_t)(pp < p1 ? pp : p1 - 1) * C;
      v[u] = V8<T>::ld(x + off);
      d[u] = V8<T>::ld(du + off);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = p + u * rpp < p1;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float t = V8<T>::get(d[u], j) * swishf_(fmaf(V8<T>::get(v[u], j), fsc[j], fsh[j]));
        a[j] += ok ? t : 0.f;
      }
    }
  }
  if (rows) {        // reproducible (see gap_affine_act_kernel)
    __syncthreads();
    for (int j = 0; j < 8; ++j) lds[threadIdx.x * 8 + j] = a[j];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      float t = 0.f;
      for (int th = c >> 3; th < (int)blockDim.x; th += CP) t += lds[th * 8 + (c & 7)];
      rows[((size_t)sp * gridDim.y + b) * C + c] = t;
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) atomicAdd(&lds[cq * 8 + j], a[j]);
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) atomicAdd(&ds[(size_t)b * C + c], lds[c]);
}

// SE backward through the two FCs: one block per sample
// Workgroup = (group of G images, slice of CS channels).  The weight-gradient outer products are summed over the group before
// they meet the global atomics (one workgroup per image sent B*C*R contended atomics per matrix: 224 us per call at B = 128);
// every slice recomputes the group's dlogit2 / dh1 (a few hundred thousand FMAs) rather than exchanging them.  With a slab
// workspace (sl_*: one slab per image group) the group sums are plain-stored and added in group order afterwards (reproducible);
// without, they meet in fp32 atomics.
__global__ __launch_bounds__(1024) void se_bwd_kernel(const float* __restrict__ ds, const float* __restrict__ s,
                                                      const float* __restrict__ h1, const float* __restrict__ pooled,
                                                      const float* __restrict__ w1, const float* __restrict__ w2, float* dw1, float* db1,
                                                      float* dw2, float* db2, float* __restrict__ dpooled, int B, int C, int R, int G,
                                                      int CS, float* sl_w1, float* sl_b1, float* sl_w2, float* sl_b2) {
  extern __shared__ float lds[];          // [G][C] dlogit2, [G][R] a1 = swish(h1), [G][R] dh1
  float* dl2 = lds;
  float* a1 = lds + (size_t)G * C;
  float* dh1 = a1 + G * R;
  const int b0 = blockIdx.x * G, ng = min(G, B - b0);
  const int cs0 = blockIdx.y * CS, ncs = min(CS, C - cs0);
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < ng * R; i += nt) a1[i] = swishf_(h1[(size_t)b0 * R + i]);
  for (int i = tid; i < ng * C; i += nt) {
    const float sv = s[(size_t)b0 * C + i];
    dl2[i] = ds[(size_t)b0 * C + i] * sv * (1.f - sv);
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6, nw = nt >> 6;
  for (int r = wave; r < R; r += nw) {                // a wave per hidden unit: each W2 element is read once for the whole group
    float a[16];
#pragma unroll
    for (int gi = 0; gi < 16; ++gi) a[gi] = 0.f;
    for (int c = lane; c < C; c += 64) {
      const float wv = w2[(size_t)c * R + r];
#pragma unroll
      for (int gi = 0; gi < 16; ++gi)
        if (gi < ng) a[gi] = fmaf(wv, dl2[gi * C + c], a[gi]);
    }
#pragma unroll
    for (int gi = 0; gi < 16; ++gi) {
#pragma unroll
      for (int d = 32; d >= 1; d >>= 1) a[gi] += __shfl_xor(a[gi], d);
      if (lane == 0 && gi < ng) dh1[gi * R + r] = a[gi] * dswishf_(h1[(size_t)(b0 + gi) * R + r]);
    }
  }
  for (int i = tid; i < ncs * R; i += nt) {           // dW2[c][r] += sum_g dlogit2[g][c] * a1[g][r]
    const int cl = i / R, r = i - cl * R, c = cs0 + cl;
    float a = 0.f;
    for (int gi = 0; gi < ng; ++gi) a = fmaf(dl2[gi * C + c], a1[gi * R + r], a);
    dw_out(dw2, sl_w2, (size_t)C * R, (int)blockIdx.x, (size_t)c * R + r, a);
  }
  for (int cl = tid; cl < ncs; cl += nt) {
    float a = 0.f;
    for (int gi = 0; gi < ng; ++gi) a += dl2[gi * C + cs0 + cl];
    dw_out(db2, sl_b2, (size_t)C, (int)blockIdx.x, (size_t)(cs0 + cl), a);
  }
  __syncthreads();
  if (blockIdx.y == 0)
    for (int r = tid; r < R; r += nt) {
      float a = 0.f;
      for (int gi = 0; gi < ng; ++gi) a += dh1[gi * R + r];
      dw_out(db1, sl_b1, (size_t)R, (int)blockIdx.x, (size_t)r, a);
    }
  for (int i = tid; i < R * ncs; i += nt) {           // dW1[r][c] += sum_g dh1[g][r] * pooled[g][c]
    const int r = i / ncs, c = cs0 + i - r * ncs;
    float a = 0.f;
    for (int gi = 0; gi < ng; ++gi) a = fmaf(dh1[gi * R + r], pooled[(size_t)(b0 + gi) * C + c], a);
    dw_out(dw1, sl_w1, (size_t)R * C, (int)blockIdx.x, (size_t)r * C + c, a);
  }
  for (int i = tid; i < ng * ncs; i += nt) {
    const int gi = i / ncs, c = cs0 + i - gi * ncs;
    float a = 0.f;
    for (int r = 0; r < R; ++r) a = fmaf(w1[(size_t)r * C + c], dh1[gi * R + r], a);
    dpooled[(size_t)(b0 + gi) * C + c] = a;
  }
}

// The same in two passes over (group of 16 images, slice of CS channels) workgroups, for callers with a workspace.  The single
// kernel above makes every slice recompute the group's dh1 = W2^T dlogit2 over ALL C channels (3 M multiply-adds behind
// broadcast LDS reads: 50-480 us per call on EfficientNet-B4's shapes, measured in scratch/bench_se.py); here a slice only
// multiplies its own CS channels (pass A, partial dh1 rows to the workspace) and pass B adds the slices' rows in slice order.
//   A: part[group][slice][g][r] = sum_{c in slice} W2[c][r] dlogit2[g][c];  dW2 / db2 of the slice
//   B: dh1 = (sum_slices part) * swish'(h1);  dW1 / db1;  dpooled of the slice
__global__ __launch_bounds__(512) void se_bwd_a_kernel(const float* __restrict__ ds, const float* __restrict__ s,
                                                       const float* __restrict__ h1, const float* __restrict__ w2, float* dw2,
                                                       float* db2, float* __restrict__ part, int B, int C, int R, int CS,
                                                       float* sl_w2, float* sl_b2, const float* __restrict__ ds_rows, int n_rows) {
  extern __shared__ float lds[];          // [16][CS] dlogit2 of the slice, [16][R] a1 = swish(h1)
  float* dl2 = lds;
  float* a1 = lds + 16 * CS;
  const int b0 = blockIdx.x * 16, ng = min(16, B - b0);
  const int cs0 = blockIdx.y * CS, ncs = min(CS, C - cs0);
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int i = tid; i < ng * R; i += nt) a1[i] = swishf_(h1[(size_t)b0 * R + i]);
  for (int i = tid; i < ng * ncs; i += nt) {
    const int gi = i / ncs, cl = i - gi * ncs;
    const size_t at = (size_t)(b0 + gi) * C + cs0 + cl;
    const float sv = s[at];
    float dsv;
    if (ds_rows) {       // cx_se_bwd_fused: the reduce kernel's per-split partial sums, added here in row order (no launch between)
      dsv = 0.f;
      for (int q = 0; q < 64 && q < n_rows; ++q) {       // (cx_rows_reduce's order: the same bits as the separate launch)
        float sa = 0.f;
        for (int sp = q; sp < n_rows; sp += 64) sa += ds_rows[(size_t)sp * B * C + at];
        dsv += sa;
      }
    } else {
      dsv = ds[at];
    }
    dl2[gi * CS + cl] = dsv * sv * (1.f - sv);
  }
  __syncthreads();
  float* prow = part + ((size_t)blockIdx.x * gridDim.y + blockIdx.y) * 16 * R;
  for (int i = tid; i < ng * R; i += nt) {             // consecutive threads = consecutive r: W2 rows are read along their length
    const int gi = i / R, r = i - gi * R;
    float a = 0.f;
    for (int cl = 0; cl < ncs; ++cl) a = fmaf(w2[(size_t)(cs0 + cl) * R + r], dl2[gi * CS + cl], a);
    prow[i] = a;
  }
  for (int i = tid; i < ncs * R; i += nt) {            // dW2[c][r] += sum_g dlogit2[g][c] * a1[g][r]
    const int cl = i / R, r = i - cl * R;
    float a = 0.f;
    for (int gi = 0; gi < ng; ++gi) a = fmaf(dl2[gi * CS + cl], a1[gi * R + r], a);
    dw_out(dw2, sl_w2, (size_t)C * R, (int)blockIdx.x, (size_t)(cs0 + cl) * R + r, a);
  }
  for (int cl = tid; cl < ncs; cl += nt) {
    float a = 0.f;
    for (int gi = 0; gi < ng; ++gi) a += dl2[gi * CS + cl];
    dw_out(db2, sl_b2, (size_t)C, (int)blockIdx.x, (size_t)(cs0 + cl), a);
  }
}

__global__ __launch_bounds__(512) void se_bwd_b_kernel(const float* __restrict__ part, const float* __restrict__ h1,
                                                       const float* __restrict__ pooled, const float* __restrict__ w1, float* dw1,
                                                       float* db1, float* __restrict__ dpooled, int B, int C, int R, int CS,
                                                       float* sl_w1, float* sl_b1) {
  extern __shared__ float lds[];          // [16][R] dh1
  float* dh1 = lds;
  const int b0 = blockIdx.x * 16, ng = min(16, B - b0);
  const int cs0 = blockIdx.y * CS, ncs = min(CS, C - cs0);
  const int tid = threadIdx.x, nt = blockDim.x, nsl = gridDim.y;
  const float* prow = part + (size_t)blockIdx.x * nsl * 16 * R;
  for (int i = tid; i < ng * R; i += nt) {
    float a = 0.f;
    for (int sl = 0; sl < nsl; ++sl) a += prow[(size_t)sl * 16 * R + i];
    dh1[i] = a * dswishf_(h1[(size_t)b0 * R + i]);
  }
  __syncthreads();
  if (blockIdx.y == 0)
    for (int r = tid; r < R; r += nt) {
      float a = 0.f;
      for (int gi = 0; gi < ng; ++gi) a += dh1[gi * R + r];
      dw_out(db1, sl_b1, (size_t)R, (int)blockIdx.x, (size_t)r, a);
    }
  for (int i = tid; i < R * ncs; i += nt) {            // dW1[r][c] += sum_g dh1[g][r] * pooled[g][c]
    const int r = i / ncs, c = cs0 + i - r * ncs;
    float a = 0.f;
    for (int gi = 0; gi < ng; ++gi) a = fmaf(dh1[gi * R + r], pooled[(size_t)(b0 + gi) * C + c], a);
    dw_out(dw1, sl_w1, (size_t)R * C, (int)blockIdx.x, (size_t)r * C + c, a);
  }
  for (int i = tid; i < ng * ncs; i += nt) {
    const int gi = i / ncs, c = cs0 + i - gi * ncs;
    float a = 0.f;
    for (int r = 0; r < R; ++r) a = fmaf(w1[(size_t)r * C + c], dh1[gi * R + r], a);
    dpooled[(size_t)(b0 + gi) * C + c] = a;
  }
}

// dz = (du * s[b][c] + dpooled[b][c]/HW) * swish'(x*sc+sh);  S1 += dz, S2 += dz * (x-mean)*rstd.   du / s may be null
template <typename T, int U, int MAXT>
__global__ __launch_bounds__(MAXT) void se_act_bwd_kernel(const T* __restrict__ du, const T* __restrict__ x, const float* __restrict__ sc,
                                  const float* __restrict__ sh, const float* __restrict__ mean, const float* __restrict__ rstd,
                                  const float* __restrict__ s, const float* __restrict__ dpooled, T* __restrict__ dz, float* S1,
                                  float* S2, int B, int HW, int C, int det) {
  extern __shared__ float lds[];
  const int CP = C / 8;
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) lds[i] = 0.f;
  __syncthreads();
  const int cq = threadIdx.x % CP;
  float st[2][8], fsc[8], fsh[8], fmu[8], fr[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = cq * 8 + j;
    st[0][j] = st[1][j] = 0.f;
    fsc[j] = sc[c]; fsh[j] = sh[c]; fmu[j] = mean[c]; fr[j] = rstd[c];
  }
  const float inv = 1.f / HW;
  const size_t npix = (size_t)B * HW, ppb = blockDim.x / CP, stride = (size_t)gridDim.x * ppb;
  // U pixel rows in flight per thread: every load of the rows (activations, gradient, per-sample vectors) is requested before the
  // first is consumed
  const unsigned np = (unsigned)npix, str = (unsigned)stride;
  for (unsigned pix0 = blockIdx.x * (unsigned)ppb + threadIdx.x / CP; pix0 < np; pix0 += U * str) {
    typename V8<T>::raw v[U], d[U];
    float fdp[U][8], fs[U][8];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned pix = pix0 + u * str < np ? pix0 + u * str : pix0;
      const unsigned b = pix / (unsigned)HW;
      v[u] = V8<T>::ld(x + (size_t)pix * C + cq * 8);
      if (du) d[u] = V8<T>::ld(du + (size_t)pix * C + cq * 8);
      if (dpooled) load8(dpooled + (size_t)b * C + cq * 8, fdp[u]);
      if (du && s) load8(s + (size_t)b * C + cq * 8, fs[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned pix = pix0 + u * str;
      const bool ok = pix < np;
      float o_f[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xf = V8<T>::get(v[u], j);
        float da = dpooled ? fdp[u][j] * inv : 0.f;
        if (du) da = fmaf(V8<T>::get(d[u], j), s ? fs[u][j] : 1.f, da);
        const float dzv = ok ? da * dswishf_(fmaf(xf, fsc[j], fsh[j])) : 0.f;
        st[0][j] += dzv;
        st[1][j] += dzv * (xf - fmu[j]) * fr[j];
        o_f[j] = V8<T>::rnd(dzv);
      }
      if (ok) V8<T>::st(dz + (size_t)pix * C + cq * 8, o_f);
    }
  }
  float* const dst[2] = {S1, S2};
  flush_partials<2>(st, cq, C, lds, dst, det, (int)blockIdx.x);
}

// out = a*pa + (b ? b : 0)*pb + pc   (BatchNorm output + optional skip, no activation)
// ps (optional): per-sample scale of the a-branch = the DropConnect mask / keep probability (efficientnet.py:44-51)
template <typename T>
__global__ void affine2_out_kernel(const T* __restrict__ a, const T* __restrict__ b, const float* __restrict__ pa,
                                   const float* __restrict__ pb, const float* __restrict__ pc, const float* __restrict__ ps,
                                   size_t rows_per_sample, T* __restrict__ out, size_t rows, int C) {
  const int CP = C / 8;
  const size_t total = rows * CP;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int cq = idx % CP;
    const float sb = ps ? ps[(idx / CP) / rows_per_sample] : 1.f;
    typename V8<T>::raw u, v;
    float o_f[8];
    u = V8<T>::ld(a + idx * 8);
    if (b) v = V8<T>::ld(b + idx * 8);
    float fpa[8], fpb[8], fpc[8];
    load8(pa + cq * 8, fpa);
    load8(pc + cq * 8, fpc);
    if (b) load8(pb + cq * 8, fpb);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float r = sb * fmaf(V8<T>::get(u, j), fpa[j], fpc[j]);
      if (b) r = fmaf(V8<T>::get(v, j), fpb[j], r);
      o_f[j] = V8<T>::rnd(r);
    }
    V8<T>::st(out + idx * 8, o_f);
  }
}

// out[row][:] = ps[row / rows_per_sample] * g[row][:]   (gradient of the DropConnect-ed branch)
template <typename T>
__global__ void scale_rows_kernel(const T* __restrict__ g, const float* __restrict__ ps, size_t rows_per_sample, T* __restrict__ out,
                                  size_t rows, int C) {
  const int CP = C / 8;
  const size_t total = rows * CP;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const float sb = ps[(idx / CP) / rows_per_sample];
    typename V8<T>::raw u;
    float o_f[8];
    u = V8<T>::ld(g + idx * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) o_f[j] = V8<T>::rnd(sb * V8<T>::get(u, j));
    V8<T>::st(out + idx * 8, o_f);
  }
}

// counter-based Bernoulli mask, already divided by the keep probability: out[i] in {0, 1/keep}
__global__ void dropout_mask_kernel(float* __restrict__ out, size_t n, float keep, unsigned long long seed) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long z = seed + 0x9E3779B97F4A7C15ull * (i + 1);          // splitmix64
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  const float u = (float)(z >> 40) * (1.0f / 16777216.0f);               // 24 bits -> [0,1)
  out[i] = u < keep ? 1.f / keep : 0.f;
}

// the same with the seed assembled on the device: seed = base + step * 1000003, step read from device memory (a captured graph
// draws fresh masks at every replay)
__global__ void dropout_mask_dev_kernel(float* __restrict__ out, size_t n, float keep, unsigned long long base,
                                        const unsigned long long* __restrict__ step) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned long long z = base + step[0] * 1000003ull + 0x9E3779B97F4A7C15ull * (i + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  const float u = (float)(z >> 40) * (1.0f / 16777216.0f);
  out[i] = u < keep ? 1.f / keep : 0.f;
}
__global__ void counter_add_kernel(unsigned long long* ctr, unsigned long long inc) {
  if (threadIdx.x == 0 && blockIdx.x == 0) ctr[0] += inc;
}

__global__ void mul_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = a[i] * b[i];
}

__global__ void linear_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                  float* __restrict__ y, int C, int N) {
  const int b = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int k = wave; k < N; k += blockDim.x / 64) {
    float acc = 0.f;
    for (int c = lane; c < C; c += 64) acc = fmaf(x[(size_t)b * C + c], w[(size_t)k * C + c], acc);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d);
    if (lane == 0) y[(size_t)b * N + k] = acc + (bias ? bias[k] : 0.f);
  }
}

}  // namespace

// ---- launchers, templated on the storage type (bf16 / the fp32 parity mode)
namespace {

template <typename T>
int nchw3_to_nhwc8_t(const float* x, void* y, int B, int H, int W, void* stream) {
  if (!x || !y || B <= 0 || H <= 0 || W <= 0) return CX_EINVAL;
  const size_t hw = (size_t)H * W, total = hw * B;
  hipLaunchKernelGGL(nchw3_to_nhwc8_kernel<T>, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), x, (T*)y, hw, total);
  return launch_status();
}

template <typename T>
int u8_to_nhwc8_t(const uint8_t* x, void* y, size_t npix, float mean, float std, void* stream) {
  if (!x || !y || std <= 0.f) return CX_EINVAL;
  hipLaunchKernelGGL(u8_to_nhwc8_kernel<T>, dim3((npix + 255) / 256), dim3(256), 0, as_stream(stream), x, (T*)y, 1.f / (255.f * std),
                     -mean / std, npix);
  return launch_status();
}

// stat_rows > 0 (forward / input gradient / the two statistics kernels below): DETERMINISTIC statistic rows as CxConv.stat_det --
// row r at stat_sum[r * C + c], at most stat_rows rows, cx_last_stat_rows() tells how many; 0: fp32 atomics into [C] vectors.
inline size_t det_smem(size_t base, int th, int ns, int det) {
  const size_t need = (size_t)th * ns * 8 * sizeof(float);
  return det && need > base ? need : base;
}

template <typename T>
int dwconv_fwd_t(const void* x, const float* w, const float* sc, const float* sh, void* y, float* stat_sum, float* stat_sq, int B, int H,
                 int W, int C, int k, int stride, int pad, int stat_rows, void* stream) {
  if (!x || !w || !y || C % 8 || C > 4096 || k < 1 || stride < 1 || (sc && !sh)) return CX_EINVAL;
  if (stat_rows > 0 && (!stat_sum || !stat_sq)) return CX_EINVAL;
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  const int CP = C / 8, th = threads_for(CP);
  if (CP > 1024) return CX_ESHAPE;
  if constexpr (std::is_same<T, bf16>::value) {      // the tiled fast paths are bf16 kernels
    bool handled = false;
    const int rc = cx_try_dw_tile(0, x, w, sc, sh, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, y, nullptr, stat_sum,
                                  stat_sq, nullptr, 0, B, H, W, C, k, stride, pad, stat_rows, nullptr, 0, as_stream(stream), &handled);
    if (handled) return rc;
  }
  const size_t npix = (size_t)B * Ho * Wo;
  int grid = grid_for(npix, th / CP, 4096);
  const int det = stat_rows > 0 ? 1 : 0;
  if (det) { if (grid > stat_rows) grid = stat_rows; cx_tl_stat_rows = grid; }
  const size_t smem = det_smem(2 * C * sizeof(float), th, 2, det);
  if (smem > 64 * 1024) return CX_ESHAPE;
  hipLaunchKernelGGL(dwconv_fwd_kernel<T>, dim3(grid), dim3(th), smem, as_stream(stream),
                     (const T*)x, w, sc, sh, (T*)y, stat_sum, stat_sq, B, H, W, C, Ho, Wo, k, stride, pad, det);
  return launch_status();
}

template <typename T>
int dwconv_dgrad_t(const void* g, const void* g2, const float* ga, const float* gb, const float* gc, const float* w, const void* x,
                   const float* sc, const float* sh, const float* mean, const float* rstd, void* dz, float* S1, float* S2, int B, int H,
                   int W, int C, int k, int stride, int pad, int accumulate, int stat_rows, void* stream) {
  if (!g || !g2 || !ga || !gb || !gc || !w || !x || !dz || C % 8 || C / 8 > 1024) return CX_EINVAL;
  if (sc && (!sh || !mean || !rstd || !S1 || !S2)) return CX_EINVAL;
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  const int CP = C / 8, th = threads_for(CP);
  const int det = (stat_rows > 0 && S1) ? 1 : 0;
  if constexpr (std::is_same<T, bf16>::value) {      // the tiled fast paths are bf16 kernels
    bool handled = false;
    const int rc = cx_try_dw_tile(1, x, w, sc, sh, mean, rstd, g, g2, ga, gb, gc, nullptr, dz, S1, S2, nullptr, accumulate, B, H, W, C, k,
                                  stride, pad, det ? stat_rows : 0, nullptr, 0, as_stream(stream), &handled);
    if (handled) return rc;
  }
  const size_t npix = (size_t)B * H * W;
  int grid = grid_for(npix, th / CP, 4096);
  if (det) { if (grid > stat_rows) grid = stat_rows; cx_tl_stat_rows = grid; }
  const size_t smem = det_smem(2 * C * sizeof(float), th, 2, det);
  if (smem > 64 * 1024) return CX_ESHAPE;
  hipLaunchKernelGGL(dwconv_dgrad_kernel<T>, dim3(grid), dim3(th), smem, as_stream(stream),
                     (const T*)g, (const T*)g2, ga, gb, gc, w, (const T*)x, sc, sh, mean, rstd, (T*)dz, S1, S2, B, H, W, C, Ho,
                     Wo, k, stride, pad, accumulate, det);
  return launch_status();
}

// scratch (optional): slab workspace of the CxWgrad.scratch protocol -- the partial dW of every workgroup is plain-stored and added
// in workgroup order (immediately, or by the deferred table sum); null / too small: fp32 atomics
template <typename T>
int dwconv_wgrad_t(const void* g, const void* g2, const float* ga, const float* gb, const float* gc, const void* x, const float* sc,
                   const float* sh, float* dw, int B, int H, int W, int C, int k, int stride, int pad, float* scratch,
                   int64_t scratch_floats, void* stream) {
  if (!g || !g2 || !ga || !gb || !gc || !x || !dw || C % 8 || C / 8 > 1024) return CX_EINVAL;
  const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
  const int CP = C / 8, th = threads_for(CP);
  if constexpr (std::is_same<T, bf16>::value) {      // the tiled fast paths are bf16 kernels
    bool handled = false;
    const int rc = cx_try_dw_tile(2, x, nullptr, sc, sh, nullptr, nullptr, g, g2, ga, gb, gc, nullptr, nullptr, nullptr, nullptr, dw, 0, B,
                                  H, W, C, k, stride, pad, 0, scratch, scratch_floats, as_stream(stream), &handled);
    if (handled) return rc;
  }
  const size_t npix = (size_t)B * Ho * Wo;
  const int gx = grid_for(npix, th / CP, 256);
  const long long total = (long long)C * k * k;
  float* slab = dw_slab(scratch, scratch_floats, gx, total);
  const size_t smem = det_smem(C * sizeof(float), th, 1, slab != nullptr);
  hipLaunchKernelGGL(dwconv_wgrad_kernel<T>, dim3(gx, k * k), dim3(th), smem, as_stream(stream),
                     (const T*)g, (const T*)g2, ga, gb, gc, (const T*)x, sc, sh, dw, B, H, W, C, Ho, Wo, k, stride, pad, slab);
  if (const int e = launch_status()) return e;
  return slab ? cx_dw_reduce(dw, slab, (size_t)total, gx, as_stream(stream)) : 0;
}

// scratch (cx_gap_affine_act / cx_se_bwd_reduce): splits * B * C floats make the per-(image, channel) sums reproducible -- every pixel
// split plain-stores its partial into its own row, a second launch adds the rows in order; NULL / too small: fp32 atomics
template <typename T>
int gap_affine_act_t(const void* x, const float* sc, const float* sh, float* pooled, int B, int HW, int C, int act, float* scratch,
                     int64_t scratch_floats, void* stream) {
  if (!x || !sc || !sh || !pooled || C % 8 || C / 8 > 1024) return CX_EINVAL;
  const int CP = C / 8, th = threads_for(CP);
  int splits = 1024 / B;
  if (splits < 1) splits = 1;
  if (splits > HW / 16 + 1) splits = HW / 16 + 1;
  float* rows = (scratch && (int64_t)splits * B * C <= scratch_floats) ? scratch : nullptr;
  if (!rows) {
    hipError_t e = hipMemsetAsync(pooled, 0, (size_t)B * C * sizeof(float), as_stream(stream));
    if (e != hipSuccess) return (int)e;
  }
  const size_t smem = det_smem(C * sizeof(float), th, 1, rows != nullptr);
  hipLaunchKernelGGL(gap_affine_act_kernel<T>, dim3(splits, B), dim3(th), smem, as_stream(stream), (const T*)x, sc, sh, pooled, HW, C, act,
                     splits, rows);
  if (const int e = launch_status()) return e;
  return rows ? cx_rows_reduce(pooled, rows, splits, B * C, B * C, 0, stream) : 0;
}

template <typename T>
int scale_act_bc_t(const void* x, const float* sc, const float* sh, const float* s, void* u, int B, int HW, int C, void* stream) {
  if (!x || !sc || !sh || !u || C % 8) return CX_EINVAL;
  const size_t total = (size_t)B * HW * (C / 8);
  hipLaunchKernelGGL(scale_act_bc_kernel<T>, dim3(grid_for(total, 256, 8192)), dim3(256), 0, as_stream(stream), (const T*)x, sc, sh, s,
                     (T*)u, HW, C, total);
  return launch_status();
}

template <typename T>
int bn_lin_bwd_stats_t(const void* g, const void* y, const float* mean, const float* rstd, float* S1, float* S2, size_t rows, int C,
                       int stat_rows, void* stream) {
  if (!g || !y || !mean || !rstd || !S1 || !S2 || C % 8 || C / 8 > 1024) return CX_EINVAL;
  const int CP = C / 8, th = threads_for(CP);
  int grid = grid_for(rows, 8 * (th / CP), 1024);
  const int det = stat_rows > 0 ? 1 : 0;
  if (det) { if (grid > stat_rows) grid = stat_rows; cx_tl_stat_rows = grid; }
  const size_t smem = det_smem(2 * C * sizeof(float), th, 2, det);
  if (smem > 64 * 1024) return CX_ESHAPE;
  hipLaunchKernelGGL(bn_lin_bwd_stats_kernel<T>, dim3(grid), dim3(th), smem, as_stream(stream),
                     (const T*)g, (const T*)y, mean, rstd, S1, S2, rows, C, det);
  return launch_status();
}

template <typename T>
int se_bwd_reduce_t(const void* du, const void* x, const float* sc, const float* sh, float* ds, int B, int HW, int C, float* scratch,
                    int64_t scratch_floats, void* stream) {
  if (!du || !x || !sc || !sh || !ds || C % 8 || C / 8 > 1024) return CX_EINVAL;
  const int CP = C / 8, th = threads_for(CP);
  int splits = 1024 / B;
  if (splits < 1) splits = 1;
  if (splits > HW / 16 + 1) splits = HW / 16 + 1;
  float* rows = (scratch && (int64_t)splits * B * C <= scratch_floats) ? scratch : nullptr;
  if (!rows) {
    hipError_t e = hipMemsetAsync(ds, 0, (size_t)B * C * sizeof(float), as_stream(stream));
    if (e != hipSuccess) return (int)e;
  }
  const size_t smem = det_smem(C * sizeof(float), th, 1, rows != nullptr);
  hipLaunchKernelGGL(se_bwd_reduce_kernel<T>, dim3(splits, B), dim3(th), smem, as_stream(stream), (const T*)du, (const T*)x, sc, sh, ds,
                     HW, C, splits, rows);
  if (const int e = launch_status()) return e;
  return rows ? cx_rows_reduce(ds, rows, splits, B * C, B * C, 0, stream) : 0;
}

template <typename T>
int se_act_bwd_t(const void* du, const void* x, const float* sc, const float* sh, const float* mean, const float* rstd, const float* s,
                 const float* dpooled, void* dz, float* S1, float* S2, int B, int HW, int C, int stat_rows, void* stream) {
  if (!x || !sc || !sh || !mean || !rstd || !dz || !S1 || !S2 || (!du && !dpooled) || C % 8 || C / 8 > 1024) return CX_EINVAL;
  const int CP = C / 8, th = threads_for(CP);
  if ((size_t)B * HW >= (1u << 31)) return CX_ESHAPE;
  const int det = stat_rows > 0 ? 1 : 0;
  const size_t smem = det_smem(2 * C * sizeof(float), th, 2, det);
  if (smem > 64 * 1024) return CX_ESHAPE;
  if (th <= 512) {
    int grid = grid_for((size_t)B * HW, 8 * (th / CP), 1024);
    if (det) { if (grid > stat_rows) grid = stat_rows; cx_tl_stat_rows = grid; }
    hipLaunchKernelGGL((se_act_bwd_kernel<T, 4, 512>), dim3(grid), dim3(th), smem,
                       as_stream(stream), (const T*)du, (const T*)x, sc, sh, mean, rstd, s, dpooled, (T*)dz, S1, S2, B, HW, C, det);
  } else {
    int grid = grid_for((size_t)B * HW, th / CP, 2048);
    if (det) { if (grid > stat_rows) grid = stat_rows; cx_tl_stat_rows = grid; }
    hipLaunchKernelGGL((se_act_bwd_kernel<T, 1, 1024>), dim3(grid), dim3(th), smem,
                       as_stream(stream), (const T*)du, (const T*)x, sc, sh, mean, rstd, s, dpooled, (T*)dz, S1, S2, B, HW, C, det);
  }
  return launch_status();
}

template <typename T>
int affine2_out_t(const void* a, const void* b, const float* pa, const float* pb, const float* pc, const float* sample_scale,
                   size_t rows_per_sample, void* out, size_t rows, int C, void* stream) {
  if (!a || !pa || !pc || !out || (b && !pb) || C % 8 || (sample_scale && rows_per_sample == 0)) return CX_EINVAL;
  hipLaunchKernelGGL(affine2_out_kernel<T>, dim3(grid_for(rows * (C / 8), 256, 8192)), dim3(256), 0, as_stream(stream), (const T*)a,
                     (const T*)b, pa, pb, pc, sample_scale, sample_scale ? rows_per_sample : (size_t)1, (T*)out, rows, C);
  return launch_status();
}

template <typename T>
int scale_rows_t(const void* g, const float* sample_scale, size_t rows_per_sample, void* out, size_t rows, int C, void* stream) {
  if (!g || !sample_scale || !out || rows_per_sample == 0 || C % 8) return CX_EINVAL;
  hipLaunchKernelGGL(scale_rows_kernel<T>, dim3(grid_for(rows * (C / 8), 256, 8192)), dim3(256), 0, as_stream(stream), (const T*)g,
                     sample_scale, rows_per_sample, (T*)out, rows, C);
  return launch_status();
}

}  // namespace

static int se_fwd_launch(const float* pooled, const float* w1, const float* b1, const float* w2, const float* b2, float* h1, float* s, int B,
                         int C, int R, const float* rows, int n_rows, float* pooled_out, void* stream) {
  CX_KTAG("se_fwd_kernel");
  if ((size_t)(R + C) * sizeof(float) > 64 * 1024) return CX_ESHAPE;
  // channel slices so that B x slices fills the chip (each block repeats the first product: slices <= 8)
  int slices = B >= 256 ? 1 : (256 + B - 1) / B;
  if (slices > 8) slices = 8;
  const int cps = ((C + slices - 1) / slices + 127) / 128 * 128;      // whole passes of the 128 channels a block's threads cover
  slices = (C + cps - 1) / cps;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&se_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    attr = true;
  }
  hipLaunchKernelGGL(se_fwd_kernel, dim3(B, slices), dim3(1024), (size_t)(R + C) * sizeof(float), as_stream(stream), pooled, w1, b1, w2, b2,
                     h1, s, C, R, cps, rows, n_rows, pooled_out);
  return launch_status();
}

// squeeze + excite as two launches: the pool's split rows are summed by the excitation kernel (no reduce launch between them)
template <typename T>
static int gap_se_fwd_t(const void* x, const float* sc, const float* sh, float* pooled, const float* w1, const float* b1, const float* w2,
                        const float* b2, float* h1, float* s, int B, int HW, int C, int R, int act, float* scratch, int64_t scratch_floats,
                        void* stream) {
  if (!x || !sc || !sh || !pooled || !w1 || !b1 || !w2 || !b2 || !h1 || !s || R <= 0 || R > 1024 || C % 8 || C / 8 > 1024) return CX_EINVAL;
  const int CP = C / 8, th = threads_for(CP);
  int splits = 1024 / B;
  if (splits < 1) splits = 1;
  if (splits > HW / 16 + 1) splits = HW / 16 + 1;
  if (!scratch || (int64_t)splits * B * C > scratch_floats) {          // no row workspace: the three-launch form (atomics)
    if (const int e = gap_affine_act_t<T>(x, sc, sh, pooled, B, HW, C, act, scratch, scratch_floats, stream)) return e;
    return cx_se_fwd(pooled, w1, b1, w2, b2, h1, s, B, C, R, stream);
  }
  const size_t smem = det_smem(C * sizeof(float), th, 1, true);
  hipLaunchKernelGGL(gap_affine_act_kernel<T>, dim3(splits, B), dim3(th), smem, as_stream(stream), (const T*)x, sc, sh, pooled, HW, C, act,
                     splits, scratch);
  if (const int e = launch_status()) return e;
  return se_fwd_launch(nullptr, w1, b1, w2, b2, h1, s, B, C, R, scratch, splits, pooled, stream);
}
extern "C" {

int cx_nchw3_to_nhwc8(const float* x, void* y, int B, int H, int W, void* stream) {
  return nchw3_to_nhwc8_t<bf16>(x, y, B, H, W, stream);
}
int cx_nchw3_to_nhwc8_f32(const float* x, void* y, int B, int H, int W, void* stream) {
  return nchw3_to_nhwc8_t<float>(x, y, B, H, W, stream);
}

int cx_u8_to_nhwc8(const uint8_t* x, void* y, size_t npix, float mean, float std, void* stream) {
  return u8_to_nhwc8_t<bf16>(x, y, npix, mean, std, stream);
}
int cx_u8_to_nhwc8_f32(const uint8_t* x, void* y, size_t npix, float mean, float std, void* stream) {
  return u8_to_nhwc8_t<float>(x, y, npix, mean, std, stream);
}

int cx_dwconv_fwd(const void* x, const float* w, const float* sc, const float* sh, void* y, float* stat_sum, float* stat_sq, int B, int H,
                  int W, int C, int k, int stride, int pad, int stat_rows, void* stream) {
  return dwconv_fwd_t<bf16>(x, w, sc, sh, y, stat_sum, stat_sq, B, H, W, C, k, stride, pad, stat_rows, stream);
}
int cx_dwconv_fwd_f32(const void* x, const float* w, const float* sc, const float* sh, void* y, float* stat_sum, float* stat_sq, int B, int H,
                  int W, int C, int k, int stride, int pad, int stat_rows, void* stream) {
  return dwconv_fwd_t<float>(x, w, sc, sh, y, stat_sum, stat_sq, B, H, W, C, k, stride, pad, stat_rows, stream);
}

int cx_dwconv_dgrad(const void* g, const void* g2, const float* ga, const float* gb, const float* gc, const float* w, const void* x,
                    const float* sc, const float* sh, const float* mean, const float* rstd, void* dz, float* S1, float* S2, int B, int H,
                    int W, int C, int k, int stride, int pad, int accumulate, int stat_rows, void* stream) {
  return dwconv_dgrad_t<bf16>(g, g2, ga, gb, gc, w, x, sc, sh, mean, rstd, dz, S1, S2, B, H, W, C, k, stride, pad, accumulate, stat_rows, stream);
}
int cx_dwconv_dgrad_f32(const void* g, const void* g2, const float* ga, const float* gb, const float* gc, const float* w, const void* x,
                    const float* sc, const float* sh, const float* mean, const float* rstd, void* dz, float* S1, float* S2, int B, int H,
                    int W, int C, int k, int stride, int pad, int accumulate, int stat_rows, void* stream) {
  return dwconv_dgrad_t<float>(g, g2, ga, gb, gc, w, x, sc, sh, mean, rstd, dz, S1, S2, B, H, W, C, k, stride, pad, accumulate, stat_rows, stream);
}

int cx_dwconv_wgrad(const void* g, const void* g2, const float* ga, const float* gb, const float* gc, const void* x, const float* sc,
                    const float* sh, float* dw, int B, int H, int W, int C, int k, int stride, int pad, float* scratch,
                    int64_t scratch_floats, void* stream) {
  return dwconv_wgrad_t<bf16>(g, g2, ga, gb, gc, x, sc, sh, dw, B, H, W, C, k, stride, pad, scratch, scratch_floats, stream);
}
int cx_dwconv_wgrad_f32(const void* g, const void* g2, const float* ga, const float* gb, const float* gc, const void* x, const float* sc,
                    const float* sh, float* dw, int B, int H, int W, int C, int k, int stride, int pad, float* scratch,
                    int64_t scratch_floats, void* stream) {
  return dwconv_wgrad_t<float>(g, g2, ga, gb, gc, x, sc, sh, dw, B, H, W, C, k, stride, pad, scratch, scratch_floats, stream);
}

int cx_gap_affine_act(const void* x, const float* sc, const float* sh, float* pooled, int B, int HW, int C, int act, float* scratch,
                      int64_t scratch_floats, void* stream) {
  return gap_affine_act_t<bf16>(x, sc, sh, pooled, B, HW, C, act, scratch, scratch_floats, stream);
}
int cx_gap_affine_act_f32(const void* x, const float* sc, const float* sh, float* pooled, int B, int HW, int C, int act, float* scratch,
                      int64_t scratch_floats, void* stream) {
  return gap_affine_act_t<float>(x, sc, sh, pooled, B, HW, C, act, scratch, scratch_floats, stream);
}

int cx_se_fwd(const float* pooled, const float* w1, const float* b1, const float* w2, const float* b2, float* h1, float* s, int B, int C,
              int R, void* stream) {
  if (!pooled || !w1 || !b1 || !w2 || !b2 || !h1 || !s || R <= 0 || R > 1024) return CX_EINVAL;
  return se_fwd_launch(pooled, w1, b1, w2, b2, h1, s, B, C, R, nullptr, 0, nullptr, stream);
}

int cx_gap_se_fwd(const void* x, const float* sc, const float* sh, float* pooled, const float* w1, const float* b1, const float* w2,
                  const float* b2, float* h1, float* s, int B, int HW, int C, int R, int act, float* scratch, int64_t scratch_floats,
                  void* stream) {
  return gap_se_fwd_t<bf16>(x, sc, sh, pooled, w1, b1, w2, b2, h1, s, B, HW, C, R, act, scratch, scratch_floats, stream);
}
int cx_gap_se_fwd_f32(const void* x, const float* sc, const float* sh, float* pooled, const float* w1, const float* b1, const float* w2,
                      const float* b2, float* h1, float* s, int B, int HW, int C, int R, int act, float* scratch, int64_t scratch_floats,
                      void* stream) {
  return gap_se_fwd_t<float>(x, sc, sh, pooled, w1, b1, w2, b2, h1, s, B, HW, C, R, act, scratch, scratch_floats, stream);
}

int cx_scale_act_bc(const void* x, const float* sc, const float* sh, const float* s, void* u, int B, int HW, int C, void* stream) {
  return scale_act_bc_t<bf16>(x, sc, sh, s, u, B, HW, C, stream);
}
int cx_scale_act_bc_f32(const void* x, const float* sc, const float* sh, const float* s, void* u, int B, int HW, int C, void* stream) {
  return scale_act_bc_t<float>(x, sc, sh, s, u, B, HW, C, stream);
}

int cx_bn_lin_bwd_stats(const void* g, const void* y, const float* mean, const float* rstd, float* S1, float* S2, size_t rows, int C,
                        int stat_rows, void* stream) {
  return bn_lin_bwd_stats_t<bf16>(g, y, mean, rstd, S1, S2, rows, C, stat_rows, stream);
}
int cx_bn_lin_bwd_stats_f32(const void* g, const void* y, const float* mean, const float* rstd, float* S1, float* S2, size_t rows, int C,
                        int stat_rows, void* stream) {
  return bn_lin_bwd_stats_t<float>(g, y, mean, rstd, S1, S2, rows, C, stat_rows, stream);
}

int cx_se_bwd_reduce(const void* du, const void* x, const float* sc, const float* sh, float* ds, int B, int HW, int C, float* scratch,
                     int64_t scratch_floats, void* stream) {
  return se_bwd_reduce_t<bf16>(du, x, sc, sh, ds, B, HW, C, scratch, scratch_floats, stream);
}
int cx_se_bwd_reduce_f32(const void* du, const void* x, const float* sc, const float* sh, float* ds, int B, int HW, int C, float* scratch,
                     int64_t scratch_floats, void* stream) {
  return se_bwd_reduce_t<float>(du, x, sc, sh, ds, B, HW, C, scratch, scratch_floats, stream);
}

// ds_rows != nullptr: the per-(image, channel) sums arrive as n_rows split rows (two-pass form only: returns CX_ESHAPE when it
// cannot run, the caller then reduces the rows into ds and calls again without them)
static int se_bwd_impl(const float* ds, const float* ds_rows, int n_rows, const float* s, const float* h1, const float* pooled, const float* w1,
                       const float* w2, float* dw1, float* db1, float* dw2, float* db2, float* dpooled, int B, int C, int R, float* scratch,
                       int64_t scratch_floats, void* stream) {
  if ((!ds && !ds_rows) || !s || !h1 || !pooled || !w1 || !w2 || !dw1 || !db1 || !dw2 || !db2 || !dpooled || R <= 0) return CX_EINVAL;
  hipStream_t st = as_stream(stream);
  {
    // two-pass form: needs a workspace for the slabs (one per group of 16 images) and the slices' partial dh1 rows
    const int CS = 64, groups = (B + 15) / 16, slices = (C + CS - 1) / CS;
    const long long per = 2ll * C * R + C + R, nslab = (long long)groups * per, npart = (long long)groups * slices * 16 * R;
    float* wsb = dw_slab(scratch, scratch_floats, 1, nslab + npart);
    const size_t smem_a = (size_t)16 * (CS + R) * sizeof(float), smem_b = (size_t)16 * R * sizeof(float);
    if (wsb && smem_a <= 48 * 1024) {
      float *s1 = wsb, *sb1 = s1 + (size_t)groups * R * C, *s2 = sb1 + (size_t)groups * R, *sb2 = s2 + (size_t)groups * C * R;
      float* part = wsb + nslab;
      CX_KTAG("se_bwd_a_kernel");
      hipLaunchKernelGGL(se_bwd_a_kernel, dim3(groups, slices), dim3(512), smem_a, st, ds, s, h1, w2, dw2, db2, part, B, C, R, CS, s2,
                         sb2, ds_rows, n_rows);
      hipLaunchKernelGGL(se_bwd_b_kernel, dim3(groups, slices), dim3(512), smem_b, st, (const float*)part, h1, pooled, w1, dw1, db1,
                         dpooled, B, C, R, CS, s1, sb1);
      if (const int e = launch_status()) return e;
      if (const int e = cx_dw_reduce(dw1, s1, (size_t)R * C, groups, st)) return e;
      if (const int e = cx_dw_reduce(db1, sb1, (size_t)R, groups, st)) return e;
      if (const int e = cx_dw_reduce(dw2, s2, (size_t)C * R, groups, st)) return e;
      const int e = cx_dw_reduce(db2, sb2, (size_t)C, groups, st);
      cx_tl_slab_floats_v = (int)(nslab + npart);
      return e;
    }
  }
  if (ds_rows) return CX_ESHAPE;
  // no workspace: one kernel, fp32 atomics for the weight gradients
  int G = (int)((120 * 1024) / ((size_t)(C + 2 * R) * sizeof(float)));      // images per workgroup: what 120 KB of LDS hold, at most 16
  if (G > 16) G = 16;
  if (G < 1) return CX_ESHAPE;
  const size_t smem = (size_t)G * (C + 2 * R) * sizeof(float);
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&se_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    attr = true;
  }
  const int CS = 64, groups = (B + G - 1) / G;
  cx_tl_slab_floats_v = 0;
  hipLaunchKernelGGL(se_bwd_kernel, dim3(groups, (C + CS - 1) / CS), dim3(1024), smem, st, ds, s, h1, pooled, w1, w2,
                     dw1, db1, dw2, db2, dpooled, B, C, R, G, CS, nullptr, nullptr, nullptr, nullptr);
  return launch_status();
}

int cx_se_bwd(const float* ds, const float* s, const float* h1, const float* pooled, const float* w1, const float* w2, float* dw1,
              float* db1, float* dw2, float* db2, float* dpooled, int B, int C, int R, float* scratch, int64_t scratch_floats,
              void* stream) {
  return se_bwd_impl(ds, nullptr, 0, s, h1, pooled, w1, w2, dw1, db1, dw2, db2, dpooled, B, C, R, scratch, scratch_floats, stream);
}

}  // extern "C" (a template follows)

// SELayer backward up to d pooled in three launches instead of four: the reduce kernel's split rows go straight into the first FC
// pass (cx_se_bwd_reduce + cx_se_bwd with the rows summed by se_bwd_a_kernel); `ds` is only written when the fused form cannot run
template <typename T>
static int se_bwd_fused_t(const void* du, const void* x, const float* sc, const float* sh, float* ds, const float* s, const float* h1,
                          const float* pooled, const float* w1, const float* w2, float* dw1, float* db1, float* dw2, float* db2, float* dpooled,
                          int B, int HW, int C, int R, float* rows_scratch, int64_t rows_floats, float* scratch, int64_t scratch_floats,
                          void* stream) {
  if (!du || !x || !sc || !sh || !ds || C % 8 || C / 8 > 1024) return CX_EINVAL;
  const int CP = C / 8, th = threads_for(CP);
  int splits = 1024 / B;
  if (splits < 1) splits = 1;
  if (splits > HW / 16 + 1) splits = HW / 16 + 1;
  const bool rows_ok = rows_scratch && (int64_t)splits * B * C <= rows_floats;
  if (rows_ok) {
    const size_t smem = det_smem(C * sizeof(float), th, 1, true);
    hipLaunchKernelGGL(se_bwd_reduce_kernel<T>, dim3(splits, B), dim3(th), smem, as_stream(stream), (const T*)du, (const T*)x, sc, sh, ds, HW,
                       C, splits, rows_scratch);
    if (const int e = launch_status()) return e;
    const int rc = se_bwd_impl(nullptr, rows_scratch, splits, s, h1, pooled, w1, w2, dw1, db1, dw2, db2, dpooled, B, C, R, scratch,
                               scratch_floats, stream);
    if (rc != CX_ESHAPE) return rc;
    if (const int e = cx_rows_reduce(ds, rows_scratch, splits, B * C, B * C, 0, stream)) return e;      // (no slab workspace: the old sequence)
  } else if (const int e = se_bwd_reduce_t<T>(du, x, sc, sh, ds, B, HW, C, rows_scratch, rows_floats, stream)) {
    return e;
  }
  return se_bwd_impl(ds, nullptr, 0, s, h1, pooled, w1, w2, dw1, db1, dw2, db2, dpooled, B, C, R, scratch, scratch_floats, stream);
}

extern "C" {

int cx_se_bwd_fused(const void* du, const void* x, const float* sc, const float* sh, float* ds, const float* s, const float* h1,
                    const float* pooled, const float* w1, const float* w2, float* dw1, float* db1, float* dw2, float* db2, float* dpooled,
                    int B, int HW, int C, int R, float* rows_scratch, int64_t rows_floats, float* scratch, int64_t scratch_floats,
                    void* stream) {
  return se_bwd_fused_t<bf16>(du, x, sc, sh, ds, s, h1, pooled, w1, w2, dw1, db1, dw2, db2, dpooled, B, HW, C, R, rows_scratch, rows_floats,
                              scratch, scratch_floats, stream);
}
int cx_se_bwd_fused_f32(const void* du, const void* x, const float* sc, const float* sh, float* ds, const float* s, const float* h1,
                        const float* pooled, const float* w1, const float* w2, float* dw1, float* db1, float* dw2, float* db2, float* dpooled,
                        int B, int HW, int C, int R, float* rows_scratch, int64_t rows_floats, float* scratch, int64_t scratch_floats,
                        void* stream) {
  return se_bwd_fused_t<float>(du, x, sc, sh, ds, s, h1, pooled, w1, w2, dw1, db1, dw2, db2, dpooled, B, HW, C, R, rows_scratch, rows_floats,
                               scratch, scratch_floats, stream);
}

int cx_se_act_bwd(const void* du, const void* x, const float* sc, const float* sh, const float* mean, const float* rstd, const float* s,
                  const float* dpooled, void* dz, float* S1, float* S2, int B, int HW, int C, int stat_rows, void* stream) {
  return se_act_bwd_t<bf16>(du, x, sc, sh, mean, rstd, s, dpooled, dz, S1, S2, B, HW, C, stat_rows, stream);
}
int cx_se_act_bwd_f32(const void* du, const void* x, const float* sc, const float* sh, const float* mean, const float* rstd, const float* s,
                  const float* dpooled, void* dz, float* S1, float* S2, int B, int HW, int C, int stat_rows, void* stream) {
  return se_act_bwd_t<float>(du, x, sc, sh, mean, rstd, s, dpooled, dz, S1, S2, B, HW, C, stat_rows, stream);
}

int cx_affine2_out(const void* a, const void* b, const float* pa, const float* pb, const float* pc, const float* sample_scale,
                   size_t rows_per_sample, void* out, size_t rows, int C, void* stream) {
  return affine2_out_t<bf16>(a, b, pa, pb, pc, sample_scale, rows_per_sample, out, rows, C, stream);
}
int cx_affine2_out_f32(const void* a, const void* b, const float* pa, const float* pb, const float* pc, const float* sample_scale,
                   size_t rows_per_sample, void* out, size_t rows, int C, void* stream) {
  return affine2_out_t<float>(a, b, pa, pb, pc, sample_scale, rows_per_sample, out, rows, C, stream);
}

int cx_scale_rows(const void* g, const float* sample_scale, size_t rows_per_sample, void* out, size_t rows, int C, void* stream) {
  return scale_rows_t<bf16>(g, sample_scale, rows_per_sample, out, rows, C, stream);
}
int cx_scale_rows_f32(const void* g, const float* sample_scale, size_t rows_per_sample, void* out, size_t rows, int C, void* stream) {
  return scale_rows_t<float>(g, sample_scale, rows_per_sample, out, rows, C, stream);
}

int cx_dropout_mask(float* out, size_t n, float keep_prob, unsigned long long seed, void* stream) {
  if (!out || !(keep_prob > 0.f) || keep_prob > 1.f) return CX_EINVAL;
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), out, n, keep_prob, seed);
  return launch_status();
}

int cx_dropout_mask_dev(float* out, size_t n, float keep_prob, unsigned long long base, const unsigned long long* step, void* stream) {
  if (!out || !step || !(keep_prob > 0.f) || keep_prob > 1.f) return CX_EINVAL;
  hipLaunchKernelGGL(dropout_mask_dev_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), out, n, keep_prob, base, step);
  return launch_status();
}

int cx_counter_add(unsigned long long* counter, unsigned long long inc, void* stream) {
  if (!counter) return CX_EINVAL;
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(64), 0, as_stream(stream), counter, inc);
  return launch_status();
}

int cx_mul_f32(const float* a, const float* b, float* out, size_t n, void* stream) {
  if (!a || !b || !out) return CX_EINVAL;
  hipLaunchKernelGGL(mul_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, as_stream(stream), a, b, out, n);
  return launch_status();
}

int cx_linear_fwd(const float* x, const float* w, const float* bias, float* y, int B, int C, int N, void* stream) {
  if (!x || !w || !y) return CX_EINVAL;
  hipLaunchKernelGGL(linear_fwd_kernel, dim3(B), dim3(256), 0, as_stream(stream), x, w, bias, y, C, N);
  return launch_status();
}

}  // extern "C"
