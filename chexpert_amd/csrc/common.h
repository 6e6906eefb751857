// Shared device helpers for the gfx950 kernels (wave64, MFMA 32x32x16 bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/chexpert_hip.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

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

// XCD-aware bijective remap: blocks b and b+8 share an XCD (round-robin dispatch); give each XCD a
// contiguous range of tile ids so neighbouring tiles (same A rows / same weights) hit one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

static inline hipStream_t as_stream(void* s) { return (hipStream_t)s; }
static inline int launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
static inline bool aligned16(const void* p) { return (((uintptr_t)p) & 15) == 0; }
