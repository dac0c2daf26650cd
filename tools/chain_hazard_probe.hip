// Does a chain of tiny dependent launches on ONE stream always see the previous launch's stores?  (round 4: the encoder's LSTM read a
// partially stale h(t-1) in launch-tape replays of the VGG step.)  Pattern of lstm_step_fwd_kernel: kernel t reads ALL of row t of a
// [T+1][H] array and every wave's lane 0 stores ONE float of row t+1, so a 128-byte line of the row is written by 32 waves of 8
// workgroups (different XCDs).  The array is reused every "step" with new data; launches are issued back to back from C++.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/chain_hazard_probe tools/chain_hazard_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float wsum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// h_next[j] = seed + sum_k h_prev[k] * w(j, k) with w = 1 / H for every (j, k): h_next[j] = seed + mean(h_prev)   (exact in fp32 for our values)
template <int MODE>
__global__ __launch_bounds__(256) void step_kernel(const float* h_prev, float* h_next, int H, float seed) {
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (j >= H) return;
  float a = 0.f;
  for (int k = lane * 4; k < H; k += 256) {
    float4 v;
    if (MODE == 1) {            // system-coherent loads
      v.x = __builtin_nontemporal_load(h_prev + k); v.y = __builtin_nontemporal_load(h_prev + k + 1);
      v.z = __builtin_nontemporal_load(h_prev + k + 2); v.w = __builtin_nontemporal_load(h_prev + k + 3);
    } else v = *(const float4*)(h_prev + k);
    a += v.x + v.y + v.z + v.w;
  }
  a = wsum(a);
  if (lane == 0) h_next[j] = seed + a / (float)H;
}
__global__ void heavy_kernel(float* p, long n, int iters) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  float v = 0.f;
  for (int it = 0; it < iters; ++it) for (long k = i; k < n; k += (long)gridDim.x * blockDim.x) v += p[k];
  if (v == 12345.f) p[0] = v;
}

int main(int argc, char** argv) {
  const int H = 512, T = 6, STEPS = argc > 1 ? atoi(argv[1]) : 2000;
  const int with_heavy = argc > 2 ? atoi(argv[2]) : 1;
  hipStream_t s, s2; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  float* hs; CK(hipMalloc(&hs, (T + 1) * H * 4)); CK(hipMemset(hs, 0, (T + 1) * H * 4));
  float* big; const long nbig = 64L << 20; CK(hipMalloc(&big, nbig * 4)); CK(hipMemset(big, 0, nbig * 4));
  std::vector<float> host((T + 1) * H);
  for (int mode = 0; mode < 2; ++mode) {
    long bad_steps = 0, bad_elems = 0; int first_bad = -1;
    for (int step = 0; step < STEPS; ++step) {
      if (with_heavy && step % 4 == 0) hipLaunchKernelGGL(heavy_kernel, dim3(2048), dim3(256), 0, s2, big, nbig, 1);
      const float seed = (float)(step % 64 + 1);
      for (int t = 0; t < T; ++t) {
        if (mode == 0) hipLaunchKernelGGL((step_kernel<0>), dim3(H / 4), dim3(256), 0, s, (const float*)(hs + t * H), hs + (t + 1) * H, H, seed);
        else hipLaunchKernelGGL((step_kernel<1>), dim3(H / 4), dim3(256), 0, s, (const float*)(hs + t * H), hs + (t + 1) * H, H, seed);
      }
      CK(hipMemcpyAsync(host.data(), hs, (T + 1) * H * 4, hipMemcpyDeviceToHost, s));
      CK(hipStreamSynchronize(s));
      // expected: row 0 = 0; row t+1 = seed + mean(row t) = (t+1) * seed
      bool bad = false;
      for (int t = 0; t < T; ++t)
        for (int j = 0; j < H; ++j)
          if (host[(t + 1) * H + j] != (float)(t + 1) * seed) { bad = true; ++bad_elems; }
      if (bad) { ++bad_steps; if (first_bad < 0) first_bad = step; }
    }
    printf("mode %d (%s loads), %d steps of %d dependent launches, heavy side stream %d: %ld bad steps, %ld bad elements, first bad step %d\n",
           mode, mode ? "nontemporal" : "plain", STEPS, T, with_heavy, bad_steps, bad_elems, first_bad);
  }
  return 0;
}
