// Shared device helpers for the gfx950 kernels (wave64, MFMA 32x32x16 bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/chexpert_hip.h"

// Diagnostic switches.  The product library never reads the environment: `cx_diag_int` / `cx_diag_set` return their defaults
// unless the library is a diagnostic build (`make diag`: -DCX_DIAG), so a stray variable cannot change what a training run
// computes (tests/test_host_cpu.py checks that libchexpert_hip.so does not import getenv).  Switches that make a kernel compute
// WRONG results for timing ablations need -DCX_DIAG_TIMING on top (`make diag-timing`).
#ifdef CX_DIAG
#include <stdlib.h>
static inline int cx_diag_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
static inline bool cx_diag_set(const char* name) { return getenv(name) != nullptr; }
#else
static inline int cx_diag_int(const char*, int dflt) { return dflt; }
static inline bool cx_diag_set(const char*) { return false; }
#endif

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// the two 4-pixel halves of a transposing LDS read pair (ds_read_b64_tr_b16) as one MFMA operand, with no ALU instruction in between
typedef short cx_s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 cx_join_tr(const s16x4 lo, const s16x4 hi) {
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

union U128 {
  uint4 u;
  bf16x8 h;
  bf16 e[8];
};

union U64 {
  uint2 u;
  bf16x4 h;
  s16x4 s;
  bf16 e[4];
};

__device__ __forceinline__ float bf2f(bf16 v) { return (float)v; }
__device__ __forceinline__ bf16 f2bf(float v) { return (bf16)v; }   // RNE, v_cvt_pk_bf16_f32 on gfx950

// Prologue arithmetic on a staged 16-byte chunk (8 bf16), one dword = two channels at a time: unpack by shift / mask, fp32 fma, ONE
// v_cvt_pk_bf16_f32 per dword, ReLU as a packed 16-bit integer max (a negative bf16 is a negative int16; -0 becomes +0).  Per
// element (`o.e[j] = f2bf(fmaxf(fmaf(bf2f(v.e[j]), ...)))`) the compiler converts pairs that are NOT neighbours in memory and
// re-orders them with eight more and / shift / or: 44 vector instructions per chunk against 28 here -- the 3x3 forward kernel issues
// ~900 vector instructions per wave and step next to 72 MFMAs (it is VALU-issue bound), a third of them in this prologue.
// Rounding is the same instruction on the same values: results are bit-identical.
typedef float cx_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 cx_bf16x2 __attribute__((ext_vector_type(2)));
typedef short cx_s16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float cx_bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float cx_bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t cx_packbf(float a, float b) {
  union { cx_bf16x2 h; uint32_t u; } o;
  o.h = __builtin_convertvector(cx_f32x2{a, b}, cx_bf16x2);
  return o.u;
}
__device__ __forceinline__ uint32_t cx_relu_pk(uint32_t v) {
  union { uint32_t u; cx_s16x2 s; } a, r;
  a.u = v;
  r.s = __builtin_elementwise_max(a.s, cx_s16x2{0, 0});
  return r.u;
}
// relu(x * sc + sh) of 8 channels; sc / sh: the 8 coefficients of this chunk (registers or LDS)
__device__ __forceinline__ uint4 cx_affine_relu8(const uint4 v, const float* __restrict__ sc, const float* __restrict__ sh) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  uint32_t o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    o[j] = cx_relu_pk(cx_packbf(fmaf(cx_bf_lo(w[j]), sc[2 * j], sh[2 * j]), fmaf(cx_bf_hi(w[j]), sc[2 * j + 1], sh[2 * j + 1])));
  return make_uint4(o[0], o[1], o[2], o[3]);
}
// u * a + v * b + c of 8 channels (deferred BatchNorm correction of a gradient slice)
__device__ __forceinline__ uint4 cx_affine2_8(const uint4 u, const uint4 v, const float* __restrict__ a, const float* __restrict__ b,
                                              const float* __restrict__ c) {
  const uint32_t p[4] = {u.x, u.y, u.z, u.w}, q[4] = {v.x, v.y, v.z, v.w};
  uint32_t o[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
    o[j] = cx_packbf(fmaf(cx_bf_lo(p[j]), a[2 * j], fmaf(cx_bf_lo(q[j]), b[2 * j], c[2 * j])),
                     fmaf(cx_bf_hi(p[j]), a[2 * j + 1], fmaf(cx_bf_hi(q[j]), b[2 * j + 1], c[2 * j + 1])));
  return make_uint4(o[0], o[1], o[2], o[3]);
}

// Store epilogue on 8 channels: t -> bf16 (one conversion per dword) and, where `stats`, the channel sums of the values AS STORED
// (masked by `valid`: a row past the end of the tensor contributes nothing).
__device__ __forceinline__ uint4 cx_pack8_stats(const float (&t)[8], const bool valid, const bool stats, float (&s1)[8], float (&s2)[8]) {
  uint32_t w[4];
  const uint32_t keep = valid ? 0xffffffffu : 0u;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    w[j] = cx_packbf(t[2 * j], t[2 * j + 1]);
    if (stats) {
      const uint32_t m = w[j] & keep;
      const float rl = cx_bf_lo(m), rh = cx_bf_hi(m);
      s1[2 * j] += rl;
      s1[2 * j + 1] += rh;
      s2[2 * j] += rl * rl;
      s2[2 * j + 1] += rh * rh;
    }
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// Mask epilogue of the fused 1x1 backward on 8 channels, dword pairs: pre = x*esc + esh (the forward's BN), dz = [pre > 0] * v,
// S1 += dz, S2 += dz * x, o = bf16(esl * dz + old), xh = bf16(relu(pre)) (the weight-gradient operand); `pok` = the pixel exists.
__device__ __forceinline__ void cx_mask_epi8(const float (&v)[8], const uint4 xv, const uint4 old, const float (&esc)[8], const float (&esh)[8],
                                             const float (&esl)[8], const bool pok, float (&s1)[8], float (&s2)[8], uint4& o, uint4& xh) {
  const uint32_t xw[4] = {xv.x, xv.y, xv.z, xv.w}, ow[4] = {old.x, old.y, old.z, old.w};
  const uint32_t keep = pok ? 0xffffffffu : 0u;
  uint32_t oo[4], hh[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float xl = cx_bf_lo(xw[j]), xu = cx_bf_hi(xw[j]);
    const float pl = fmaf(xl, esc[2 * j], esh[2 * j]), pu = fmaf(xu, esc[2 * j + 1], esh[2 * j + 1]);
    const float dl = (pok && pl > 0.f) ? v[2 * j] : 0.f, du = (pok && pu > 0.f) ? v[2 * j + 1] : 0.f;
    s1[2 * j] += dl;
    s1[2 * j + 1] += du;
    s2[2 * j] = fmaf(dl, xl, s2[2 * j]);
    s2[2 * j + 1] = fmaf(du, xu, s2[2 * j + 1]);
    oo[j] = cx_packbf(fmaf(esl[2 * j], dl, cx_bf_lo(ow[j])), fmaf(esl[2 * j + 1], du, cx_bf_hi(ow[j])));
    hh[j] = cx_relu_pk(cx_packbf(pl, pu)) & keep;
  }
  o = make_uint4(oo[0], oo[1], oo[2], oo[3]);
  xh = make_uint4(hh[0], hh[1], hh[2], hh[3]);
}

// ---- residual stream of the Bottleneck ResNets (attn_aug_conv.py:202-209: out = relu(bn3(conv3) + identity)) ----------------
// The stream passes through 50 joins; rounded to bf16 at each of them it alone moves the resnet152 train logits by 1.0e-2 of their
// abs-max (north_star's whole bf16 budget).  It is stored as TWO planes: `hi` = the value rounded to bf16 (RNE) -- the tensor every
// convolution and weight gradient reads, unchanged -- and `lo` = one signed byte per element, the next 8 mantissa bits: with
// O = the fp32 bit pattern of the value (>= 0: it is a ReLU output, so bit patterns order like magnitudes) and Hs = hi << 16,
// lo = clamp(((O + 0x80) - Hs) >> 8, -127, 127); the join that reads the stream as its identity operand rebuilds
// bits = Hs + (lo << 8): 16 significant bits (2^-17 relative) at 3 bytes per element instead of 4.  Only the join reads `lo`.
__device__ __forceinline__ float cx_stream_dec(const uint32_t hs, const uint32_t lo_dword, const int byte) {
  return __uint_as_float(hs + ((uint32_t)__builtin_amdgcn_sbfe((int)lo_dword, byte * 8, 8) << 8));
}
// two channels: values a, b >= 0 -> packed hi word; their lo bytes are or-ed into `lo` at byte positions `byte`, `byte + 1`
__device__ __forceinline__ uint32_t cx_stream_enc2(const float a, const float b, uint32_t& lo, const int byte) {
  const uint32_t w = cx_packbf(a, b);
  int d0 = (int)(__float_as_uint(a) + 0x80u - (w << 16)) >> 8;
  int d1 = (int)(__float_as_uint(b) + 0x80u - (w & 0xffff0000u)) >> 8;
  d0 = min(max(d0, -127), 127);            // (-128 would be the exact half-way point below hi: decoding and rounding again could then
  d1 = min(max(d1, -127), 127);            // pick the other neighbour -- clamped, hi + lo always rounds back to hi)
  lo |= (((uint32_t)d0 & 0xffu) << (byte * 8)) | (((uint32_t)d1 & 0xffu) << (byte * 8 + 8));
  return w;
}
// Layout of the per-element side planes of a (rows, C) tensor -- sign bits (one byte per 8-channel chunk) and the lo plane of the
// residual stream (8 bytes per chunk): blocked by 64 channels where C % 64 == 0, [C / 64][rows][8 chunks], so that a k-step of the
// convolution that produces them in its prologue (64 channels of 128 consecutive rows) writes whole 128-byte lines at once instead of
// 8-byte pieces of 16 different lines per row that leave the L2 one by one; flat [rows][C / 8] otherwise.  Index of chunk cq of row m.
__host__ __device__ __forceinline__ size_t cx_side_chunk(const size_t m, const int cq, const size_t rows, const int C) {
  return (C & 63) ? m * (size_t)(C >> 3) + cq : ((size_t)(cq >> 3) * rows + m) * 8 + (cq & 7);
}

// The join on two channels (dword j of an 8-channel chunk), shared by the standalone pass (cx_join_fwd) and the prologue of the next
// block's conv1 (CX_PRO_JOIN) so that both give the same bits: t = relu(u * a + (v * b + c)) with u = conv3's raw output and
// v = the identity operand (hi [+ lo]; `lo_in` null-equivalent: has_lo = false) ; returns the packed hi word, adds lo bytes and the
// two sign bits (t > 0) at position 2 j of `mask`.
__device__ __forceinline__ uint32_t cx_join2(const uint32_t u, const uint32_t v, const bool has_lo, const uint32_t vlo, const int j,
                                             const float a0, const float a1, const float b0, const float b1, const float c0, const float c1,
                                             const bool want_lo, uint32_t& lo_out, uint32_t& mask) {
  const int byte = (j & 1) * 2;
  const float v0 = has_lo ? cx_stream_dec(v << 16, vlo, byte) : cx_bf_lo(v);
  const float v1 = has_lo ? cx_stream_dec(v & 0xffff0000u, vlo, byte + 1) : cx_bf_hi(v);
  const float t0 = fmaxf(fmaf(cx_bf_lo(u), a0, fmaf(v0, b0, c0)), 0.f);
  const float t1 = fmaxf(fmaf(cx_bf_hi(u), a1, fmaf(v1, b1, c1)), 0.f);
  uint32_t w;
  if (want_lo) {
    w = cx_stream_enc2(t0, t1, lo_out, byte);
    mask |= ((t0 > 0.f ? 1u : 0u) | (t1 > 0.f ? 2u : 0u)) << (2 * j);
  } else {      // single-plane form: the sign bits are those of the value as stored (what the rounded tensor's consumers see)
    w = cx_packbf(t0, t1);
    mask |= (((w & 0xffffu) != 0u ? 1u : 0u) | ((w >> 16) != 0u ? 2u : 0u)) << (2 * j);
  }
  return w;
}

// 16-byte accesses with the non-temporal hint (streams that are not read again before the caches have turned over)
typedef uint32_t cx_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 cx_ld16_nt(const void* p) {
  const cx_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const cx_u32x4*>(p));
  return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void cx_st16_nt(void* p, const uint4 v) {
  __builtin_nontemporal_store(cx_u32x4{v.x, v.y, v.z, v.w}, reinterpret_cast<cx_u32x4*>(p));
}

// eight consecutive channels of one pixel in the storage type T (bf16: one 16-B access; fp32 parity mode: two)
template <typename T> struct V8;
template <> struct V8<bf16> {
  typedef uint4 raw;
  static __device__ __forceinline__ raw ld(const bf16* p) { return *reinterpret_cast<const uint4*>(p); }
  static __device__ __forceinline__ float get(const raw& r, int j) {
    const uint32_t w = j < 2 ? r.x : j < 4 ? r.y : j < 6 ? r.z : r.w;
    return (j & 1) ? cx_bf_hi(w) : cx_bf_lo(w);
  }
  static __device__ __forceinline__ void st(bf16* p, const float (&v)[8]) {
    *reinterpret_cast<uint4*>(p) = make_uint4(cx_packbf(v[0], v[1]), cx_packbf(v[2], v[3]), cx_packbf(v[4], v[5]), cx_packbf(v[6], v[7]));
  }
  static __device__ __forceinline__ float rnd(float v) { return bf2f(f2bf(v)); }      // the value as stored
};
template <> struct V8<float> {
  struct raw { float4 a, b; };
  static __device__ __forceinline__ raw ld(const float* p) {
    raw r;
    r.a = *reinterpret_cast<const float4*>(p);
    r.b = *reinterpret_cast<const float4*>(p + 4);
    return r;
  }
  static __device__ __forceinline__ float get(const raw& r, int j) {
    const float v[8] = {r.a.x, r.a.y, r.a.z, r.a.w, r.b.x, r.b.y, r.b.z, r.b.w};
    return v[j];
  }
  static __device__ __forceinline__ void st(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
  static __device__ __forceinline__ float rnd(float v) { return v; }
};

// four consecutive channels (attention heads: dkh = 20 = 5 x 4) and one scalar, in the storage type T
template <typename T> struct V4;
template <> struct V4<bf16> {
  typedef uint2 raw;
  static __device__ __forceinline__ raw ld(const bf16* p) { return *reinterpret_cast<const uint2*>(p); }
  static __device__ __forceinline__ float get(const raw& r, int j) { U64 u; u.u = r; return bf2f(u.e[j]); }
  static __device__ __forceinline__ float ld1(const bf16* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st1(bf16* p, float v) { *p = f2bf(v); }
};
template <> struct V4<float> {
  typedef float4 raw;
  static __device__ __forceinline__ raw ld(const float* p) { return *reinterpret_cast<const float4*>(p); }
  static __device__ __forceinline__ float get(const raw& r, int j) { const float v[4] = {r.x, r.y, r.z, r.w}; return v[j]; }
  static __device__ __forceinline__ float ld1(const float* p) { return *p; }
  static __device__ __forceinline__ void st1(float* p, float v) { *p = v; }
};

// XCD-aware bijective remap: blocks b and b+8 share an XCD (round-robin dispatch); give each XCD a
// contiguous range of tile ids so neighbouring tiles (same A rows / same weights) hit one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// the kernel instantiation the most recent conv / weight-gradient entry point of this thread dispatched to, as rocprofv3 spells it
// (cx_last_kernel(): bench.py tags its per-kernel event timings with it instead of mirroring the dispatch rules)
extern thread_local char cx_tl_kernel[112];
#define CX_KTAG(...) snprintf(cx_tl_kernel, sizeof(cx_tl_kernel), __VA_ARGS__)

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }
static inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }

// ---- workgroup statistics ---------------------------------------------------------------------------------------------
// Per-channel partial sums leave a workgroup through LDS: each wave deposits the channels it owns in its own slot, the slots
// are summed in wave order (a fixed order: fp32 addition is not associative) and the workgroup row is either plain-stored
// (CxConv.stat_det: one writer per element, consumers sum the rows in row order) or added to a replica with ONE atomic per
// channel (legacy mode).  `scratch` = NW * 2 * CW floats of LDS that nothing else uses any more.
extern thread_local int cx_tl_stat_rows;
extern thread_local int cx_tl_pro_out;       // 1: the launch wrote CxConv.pro_out

template <int NW>
__device__ __forceinline__ void wg_stat_begin(float* scratch, int CW, int tid, int nthreads) {
  __syncthreads();
  for (int i = tid; i < NW * 2 * CW; i += nthreads) scratch[i] = 0.f;
  __syncthreads();
}
__device__ __forceinline__ void wg_stat_put(float* scratch, int CW, int wave, int ch, float a, float b) {
  scratch[(wave * 2) * CW + ch] += a;
  scratch[(wave * 2 + 1) * CW + ch] += b;
}
template <int NW>
__device__ __forceinline__ void wg_stat_end(float* scratch, int CW, int tid, int nthreads, float* sum, float* sq, int det, int row,
                                            int replicas, int rstride, int n_base, int N) {
  __syncthreads();
  for (int c = tid; c < CW; c += nthreads) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      a += scratch[(w * 2) * CW + c];
      b += scratch[(w * 2 + 1) * CW + c];
    }
    const int n = n_base + c;
    if (n < N) {
      if (det) {
        sum[(size_t)row * rstride + n] = a;
        sq[(size_t)row * rstride + n] = b;
      } else {
        const size_t rep = replicas > 1 ? (size_t)(row % replicas) * rstride : 0;
        atomicAdd(&sum[rep + n], a);
        atomicAdd(&sq[rep + n], b);
      }
    }
  }
}
// ---- weight-gradient partial tiles --------------------------------------------------------------------------------------
// A weight-gradient kernel splits the pixel range over workgroups.  Its partial sums either go to dw through fp32 atomics (order,
// and with it the last bits, change from run to run) or, with a slab workspace, each (split, element) is plain-stored exactly once
// and cx_dw_reduce adds the slabs to dw in split order.
__device__ __forceinline__ void dw_out(float* dw, float* slab, size_t total, int split, size_t idx, float v) {
  if (slab)
    slab[(size_t)split * total + idx] = v;
  else
    atomicAdd(dw + idx, v);
}
// the slab to hand to a kernel: p's workspace if it holds splits * total floats, else null (atomics)
extern thread_local int cx_tl_slab_floats_v;      // floats of the caller's slab the most recent launch of this thread used (0: atomics)
static inline float* dw_slab(float* scratch, long long scratch_floats, long long splits, long long total) {
  const bool ok = scratch && splits * total <= scratch_floats && splits * total < (1ll << 31);
  cx_tl_slab_floats_v = ok ? (int)(splits * total) : 0;
  return ok ? scratch : nullptr;
}
int cx_dw_reduce(float* dw, const float* slab, size_t total, int splits, hipStream_t st);      // elementwise.hip
int cx_dw_reduce_ld(float* dw, const float* slab, size_t total, int splits, int cols, int dw_ld, hipStream_t st);   // ... into columns of a wider matrix

static inline int stat_rows_check(const CxConv& p, int rows) {
  if (!p.stat_det || !p.stat_sum) return 0;
  if (rows > p.stat_replicas || p.stat_rstride < p.N) return CX_ESTATROWS;
  cx_tl_stat_rows = rows;
  return 0;
}
