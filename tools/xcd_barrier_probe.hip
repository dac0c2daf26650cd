// Price of a barrier among the workgroups of ONE XCD against a chip-wide one (verdict r3 item 4: "captioner recurrence on one XCD - price
// the barrier first").  A launch of 256 workgroups x 256 threads; workgroup b runs on XCD b % 8 (round-robin dispatch, checked with
// s_getreg XCC_ID).  Mode 0: the 32 workgroups with b % 8 == 0 iterate N barriers (one agent-scope atomic add per workgroup, then every
// workgroup polls the counter), the others exit.  Mode 1: all 256 workgroups take part.  Mode 2: as mode 0, plus every workgroup stores 2 KiB
// before arriving and reads 2 KiB of its neighbour's after the barrier (what a recurrence step exchanges: h, c of the previous token).
//   hipcc --offload-arch=gfx950 -O3 tools/xcd_barrier_probe.hip -o /tmp/xcd_probe && /tmp/xcd_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

__global__ __launch_bounds__(256) void probe(unsigned* counter, float* xchg, unsigned* xcc_seen, long long* cycles, int iters, int mode) {
  const int b = blockIdx.x;
  if (threadIdx.x == 0) xcc_seen[b] = xcc_id();
  const bool one = mode != 1;
  if (one && (b & 7) != 0) return;
  const int members = one ? gridDim.x / 8 : gridDim.x;
  const int me = one ? b >> 3 : b;
  float acc = 0.f;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (mode == 2) {
      // 2 KiB per workgroup, write-through so that the other CUs of the XCD see it in L2
      float* mine = xchg + (size_t)((it & 1) * members + me) * 512;
      for (int i = threadIdx.x; i < 512; i += 256) __builtin_nontemporal_store((float)(it + i), mine + i);
      __threadfence();
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned want = (unsigned)(it + 1) * members;
      while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    if (mode == 2) {
      const float* other = xchg + (size_t)((it & 1) * members + (me + 1) % members) * 512;
      for (int i = threadIdx.x; i < 512; i += 256) acc += __builtin_nontemporal_load(other + i);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) cycles[b] = t1 - t0;
  if (acc == 12345.678f) xchg[0] = acc;
}

int main() {
  unsigned *counter, *xcc; float* xchg; long long* cyc;
  hipMalloc(&counter, 4); hipMalloc(&xcc, 256 * 4); hipMalloc(&xchg, 2 * 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
  const int iters = 2000;
  const char* names[3] = {"one XCD (32 workgroups)", "whole chip (256 workgroups)", "one XCD + 2 KiB exchange per workgroup"};
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(counter, 0, 4); hipMemset(cyc, 0, 256 * 8);
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipEventRecord(a);
      hipLaunchKernelGGL(probe, dim3(256), dim3(256), 0, 0, counter, xchg, xcc, cyc, iters, mode);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      std::vector<unsigned> x(256); hipMemcpy(x.data(), xcc, 256 * 4, hipMemcpyDeviceToHost);
      int bad = 0; for (int i = 0; i < 256; ++i) bad += (x[i] != (unsigned)(i & 7));
      if (rep) printf("%-44s %7.3f us per barrier (%d barriers, launch %.1f us; %d of 256 workgroups NOT on XCD b %% 8)\n", names[mode], ms * 1e3 / iters, iters, ms * 1e3, bad);
    }
  }
  return 0;
}
