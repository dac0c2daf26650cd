// Common device helpers for the lang2seg gfx950 kernels (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
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
  return e == hipSuccess ? L2S_OK : L2S_ELAUNCH;
}
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
