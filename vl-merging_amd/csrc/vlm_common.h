// Shared device/host helpers for the vl-merging MI355X (gfx950) hot path.  CDNA4 only: 64-lane
// wavefronts, MFMA, 160 KiB LDS.  No CUDA compatibility layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/vlm_hip.h"

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

#define VLM_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return VLM_ERR_LAUNCH;            \
  } while (0)

static inline int vlm_ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

__device__ __forceinline__ float bf2f(bf16_t v) { return (float)v; }
__device__ __forceinline__ bf16_t f2bf(float v) { return (bf16_t)v; }

// wave64 all-lane reductions: four DPP row rotations (row_ror:8/4/2/1 leave every lane of a 16-lane row holding the
// row's total; VALU-only, ~4 cycles each) + two cross-row exchanges through ds_bpermute -- instead of six ds_bpermute
// round trips (~100 cycles of latency each on the LDS pipe), which bounded the one-wave-per-row kernels.
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f32<0x128>(v);  // row_ror:8
  v += dpp_f32<0x124>(v);  // row_ror:4
  v += dpp_f32<0x122>(v);  // row_ror:2
  v += dpp_f32<0x121>(v);  // row_ror:1
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_f32<0x128>(v));
  v = fmaxf(v, dpp_f32<0x124>(v));
  v = fmaxf(v, dpp_f32<0x122>(v));
  v = fmaxf(v, dpp_f32<0x121>(v));
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}
