// Common device helpers for the lang2seg gfx950 kernels (wave64, CDNA4 only).
#pragma once
#include "knobs.h"
#include <hip/hip_runtime.h>
#include <cstdio>
#include <stdint.h>
#include <functional>

#define L2S_OK 0
#define L2S_EINVAL 1
#define L2S_ELAUNCH 2

typedef uint16_t bf16_t;  // raw bf16 bits

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return __builtin_bit_cast(uint16_t, b);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// generic dtype-tagged load/store (dtype: 0 = f32, 1 = bf16)
__device__ __forceinline__ float ldx(const void* p, long i, int dt) {
  return dt ? bf2f(((const bf16_t*)p)[i]) : ((const float*)p)[i];
}
__device__ __forceinline__ void stx(void* p, long i, int dt, float v) {
  if (dt) ((bf16_t*)p)[i] = f2bf(v); else ((float*)p)[i] = v;
}

// Wave-wide reductions on the DPP path (no LDS traffic: the ds_bpermute butterflies these replace cost six LDS round trips per sum, and
// their results must not be consumed behind a younger LDS write, see lang.hip cap_att_bwd_step_kernel): a scan inside each row of 16
// lanes (row_shr 1, 2, 4, 8), then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3; lane 63 holds the result and is
// broadcast.  The result is the same in every lane; the summation order is fixed.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_take(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_take<0x111, 0xf>(0.f, v);     // row_shr:1
  v += dpp_take<0x112, 0xf>(0.f, v);     // row_shr:2
  v += dpp_take<0x114, 0xf>(0.f, v);     // row_shr:4
  v += dpp_take<0x118, 0xf>(0.f, v);     // row_shr:8   -> lane 15 of each row: the row's sum
  v += dpp_take<0x142, 0xa>(0.f, v);     // row_bcast:15 into rows 1, 3
  v += dpp_take<0x143, 0xc>(0.f, v);     // row_bcast:31 into rows 2, 3 -> lane 63: the wave's sum
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_take<0x111, 0xf>(v, v));
  v = fmaxf(v, dpp_take<0x112, 0xf>(v, v));
  v = fmaxf(v, dpp_take<0x114, 0xf>(v, v));
  v = fmaxf(v, dpp_take<0x118, 0xf>(v, v));
  v = fmaxf(v, dpp_take<0x142, 0xa>(v, v));
  v = fmaxf(v, dpp_take<0x143, 0xc>(v, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
// block reductions for blockDim.x <= 1024 (result valid in every thread)
__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float r = 0.f;
  for (int i = 0; i < nw; ++i) r += sh[i];
  return r;
}
__device__ __forceinline__ float block_max(float v, float* sh) {
  v = wave_max(v);
  int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[w] = v;
  __syncthreads();
  float r = sh[0];
  for (int i = 1; i < nw; ++i) r = fmaxf(r, sh[i]);
  return r;
}

// exact a / b for 0 <= a < 2^24 using a precomputed float reciprocal (one fix-up step)
__device__ __forceinline__ int fast_div(int a, int b, float rb) {
  int q = (int)(__int2float_rn(a) * rb);
  int r = a - q * b;
  if (r < 0) --q; else if (r >= b) ++q;
  return q;
}

// Launch tape (tape.hip): while recording, every launch site also appends a replayable closure (argument copies by value).
namespace l2s {
bool recording();
void record(hipStream_t s, std::function<void(hipStream_t)> fn);
}
#define L2S_LAUNCH(kern, grid, block, shmem, stream, ...)                                                              \
  do {                                                                                                                \
    if (l2s::recording()) l2s::record((stream), [=](hipStream_t s_) { hipLaunchKernelGGL(kern, grid, block, shmem, s_, __VA_ARGS__); }); \
    hipLaunchKernelGGL(kern, grid, block, shmem, stream, __VA_ARGS__);                                                 \
  } while (0)

static inline int l2s_check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) fprintf(stderr, "lang2seg_hip: launch failed: %s\n", hipGetErrorString(e));
  return e == hipSuccess ? L2S_OK : L2S_ELAUNCH;
}
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
