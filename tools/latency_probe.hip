// Launch-boundary vs in-kernel grid-barrier cost on one GPU (decides whether chaining small dependent convolutions inside one
// persistent launch can pay).  Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/latency_probe tools/latency_probe.hip
//   empty      : N dependent launches of an empty kernel (grid G x 256)                     -> launch-to-launch floor
//   touch      : N dependent launches, each thread reads what the previous launch wrote (other workgroup's data) and writes
//   barrier    : ONE launch, G resident workgroups, N x { read neighbour's data, write, device-scope release, counter barrier, acquire }
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 9999) p[0] = 1; }

__global__ __launch_bounds__(256) void touch_kernel(const float* __restrict__ in, float* __restrict__ out, int n) {
  const int g = gridDim.x, b = (blockIdx.x + g / 2 + 1) % g;            // a workgroup far away (most likely on another XCD)
  const int i = b * 256 + threadIdx.x;
  float v = in[i % n];
  out[blockIdx.x * 256 + threadIdx.x] = v + 1.f;
}

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);                  // agent scope by default for global atomics in HIP
    while (__atomic_load_n(counter, __ATOMIC_ACQUIRE) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void barrier_kernel(float* a, float* b, int n, int iters, unsigned* counter, int touch) {
  const int g = gridDim.x, nb = (blockIdx.x + g / 2 + 1) % g;
  float* in = a; float* out = b;
  for (int it = 0; it < iters; ++it) {
    if (touch) {
      float v = in[(nb * 256 + threadIdx.x) % n];
      out[blockIdx.x * 256 + threadIdx.x] = v + 1.f;
      __threadfence();                                                  // release: my stores visible device-wide
    }
    grid_barrier(counter, (unsigned)(it + 1) * g);
    if (touch) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    float* t = in; in = out; out = t;
  }
}

int main(int argc, char** argv) {
  const int N = 2000;
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int G : {152, 256, 608}) {
    const int n = G * 256;
    float *a, *b; unsigned* c;
    CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&c, 4));
    CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
    float ms;
    // empty
    for (int w = 0; w < 2; ++w) {
      CK(hipEventRecord(e0, s));
      for (int i = 0; i < N; ++i) hipLaunchKernelGGL(empty_kernel, dim3(G), dim3(256), 0, s, (int*)nullptr);
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("G=%4d  empty   %.2f us/launch\n", G, ms * 1e3 / N);
    for (int lds : {0, 65536}) {
      for (int w = 0; w < 2; ++w) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < N; ++i) { hipLaunchKernelGGL(touch_kernel, dim3(G), dim3(256), lds, s, (const float*)a, b, n); float* t = a; a = b; b = t; }
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      }
      printf("G=%4d  touch   %.2f us/launch (dynamic LDS %d)\n", G, ms * 1e3 / N, lds);
    }
    if (G <= 256) {
      for (int touch : {0, 1}) {
        for (int w = 0; w < 2; ++w) {
          CK(hipMemsetAsync(c, 0, 4, s));
          CK(hipEventRecord(e0, s));
          hipLaunchKernelGGL(barrier_kernel, dim3(G), dim3(256), 0, s, a, b, n, N, c, touch);
          CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("G=%4d  barrier %.2f us/iteration (touch %d)\n", G, ms * 1e3 / N, touch);
      }
      // verify the data really travelled: after N iterations every element is N (touch = 1 ran last, twice from a state of k)
      std::vector<float> h(n); CK(hipMemcpy(h.data(), (N % 2) ? b : a, n * 4, hipMemcpyDeviceToHost));
      float mn = 1e30f, mx = -1e30f; for (float v : h) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
      printf("G=%4d  barrier data check: min %.0f max %.0f (equal => every read saw the previous iteration's store)\n", G, mn, mx);
    }
    CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(c));
  }
  return 0;
}
