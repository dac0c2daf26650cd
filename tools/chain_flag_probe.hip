// What would replacing the launch boundary between two DEPENDENT small launches by a flag cost?
//   In the step the backbone is a chain of 166 launches of ~8 us of which 3.2 us are the boundary (a chain of
// one-thread launches on one stream: tools/overlap_bound.py).  Here: a chain of launches shaped like layer3's convolutions - 152
// workgroups x 256 threads, each reads 8 KB of the previous launch's output (1.2 MB in all, as a 2394 x 256 bf16 map), spins ~4 us of
// dependent FMAs, writes 8 KB - in three forms:
//   A  one stream, stream order is the dependency (today's form)
//   B  two streams in turn; launch k waits IN THE KERNEL for a counter that launch k-1's workgroups bump behind an agent-scope
//      release fence, and passes an agent-scope acquire fence before it reads (launch k+2 follows k on its stream, so two are in flight)
//   C  as B, with the spin of FMAs BEFORE the wait (what a convolution could do with its weights: they do not depend on the previous launch)
//   D  as B without fences: write-through stores (sc0 sc1) + s_waitcnt vmcnt(0) before the counter, sc0 sc1 loads behind it
// Every element of the last output must equal the chain length (a stale read anywhere breaks it).  The counters are cleared by one memset per
// replay.  Every wait is bounded; a timeout is reported, not hung on.
//   hipcc --offload-arch=gfx950 -O3 tools/chain_flag_probe.hip -o /tmp/chain_flag_probe && /tmp/chain_flag_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstring>

constexpr int NWG = 152, NT = 256, ELEMS = NWG * NT * 8;       // 8 x 4-byte per thread = 8 KB per workgroup, 1.2 MB per buffer
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__device__ __forceinline__ float spin_fma(float x, int n) {
  for (int i = 0; i < n; ++i) { x = __builtin_fmaf(x, 1.0000001f, 1e-9f); asm volatile("" : "+v"(x)); }
  return x;
}

// mode 0: no flags.  mode 1: wait, then work.  mode 2: the FMA spin first, then wait, then the memory part.
__global__ __launch_bounds__(NT) void link(const int* __restrict__ in, int* __restrict__ out, int* wait_flag, int* done_flag, int mode, int spins, int* tmo) {
  const int tid = threadIdx.x, b = blockIdx.x;
  float junk = 0.f;
  if (mode == 2) junk = spin_fma((float)tid, spins);
  typedef int v4i __attribute__((ext_vector_type(4)));
  if (mode == 3) {
    // D: no fences.  The output is stored write-through (sc0 sc1: the line does not stay dirty in this XCD's L2), the counter is bumped after
    // s_waitcnt vmcnt(0); the reader requests the data with sc0 sc1 loads (served from memory / MALL, never from a stale L2 line).
    if (wait_flag) {
      if (tid == 0) {
        int n = 0;
        while (__hip_atomic_load(wait_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < NWG) {
          __builtin_amdgcn_s_sleep(1);
          if (++n > (1 << 22)) { atomicOr(tmo, 1); break; }
        }
      }
      __syncthreads();
    }
    const int src = (b * 37 + 11) % NWG;
    const int* ip = in + ((long)src * NT + tid) * 8;
    v4i v0, v1;
    asm volatile("global_load_dwordx4 %0, %2, off sc0 sc1\n\tglobal_load_dwordx4 %1, %2, off offset:16 sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v0), "=&v"(v1) : "v"(ip) : "memory");
    junk = spin_fma((float)(v0.x & 1), spins);
    const int add = 1 + (junk > 1e30f ? 1 : 0);
    v0 += add; v1 += add;
    int* op = out + ((long)b * NT + tid) * 8;
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc0 sc1\n\ts_waitcnt vmcnt(0)" :: "v"(op), "v"(v0), "v"(v1) : "memory");
    __syncthreads();
    if (tid == 0 && done_flag) __hip_atomic_fetch_add(done_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (mode && wait_flag) {
    if (tid == 0) {
      int n = 0;
      while (__hip_atomic_load(wait_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < NWG) {
        __builtin_amdgcn_s_sleep(1);
        if (++n > (1 << 22)) { atomicOr(tmo, 1); break; }
      }
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  // read the previous output from ANOTHER workgroup's region (so that the data crosses compute units / XCDs)
  const int src = (b * 37 + 11) % NWG;
  const int4* ip = (const int4*)(in + ((long)src * NT + tid) * 8);
  int4 v0 = ip[0], v1 = ip[1];
  if (mode != 2) junk = spin_fma((float)(v0.x & 1), spins);
  int4* op = (int4*)(out + ((long)b * NT + tid) * 8);
  const int add = 1 + (junk > 1e30f ? 1 : 0);
  op[0] = make_int4(v0.x + add, v0.y + add, v0.z + add, v0.w + add);
  op[1] = make_int4(v1.x + add, v1.y + add, v1.z + add, v1.w + add);
  if (mode && done_flag) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(done_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

int main() {
  constexpr int CHAIN = 72, REPS = 20;
  int *buf[3], *flags, *tmo;
  for (int i = 0; i < 3; ++i) CHECK(hipMalloc(&buf[i], ELEMS * 4));
  CHECK(hipMalloc(&flags, (CHAIN + 1) * 64 * 4)); CHECK(hipMalloc(&tmo, 4)); CHECK(hipMemset(tmo, 0, 4));
  hipStream_t s[2]; CHECK(hipStreamCreate(&s[0])); CHECK(hipStreamCreate(&s[1]));
  hipEvent_t e0, e1, ez, ej; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventCreateWithFlags(&ez, hipEventDisableTiming)); CHECK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
  // calibrate the FMA spin to ~4 us
  int spins = 200;
  std::vector<int> host(ELEMS);
  for (int mode = 0; mode < 4; ++mode) {
    float best = 1e9f; bool ok = true;
    for (int rep = 0; rep < REPS + 2; ++rep) {
      CHECK(hipMemsetAsync(buf[0], 0, ELEMS * 4, s[0]));
      CHECK(hipMemsetAsync(flags, 0, (CHAIN + 1) * 64 * 4, s[0]));
      CHECK(hipEventRecord(ez, s[0])); CHECK(hipStreamWaitEvent(s[1], ez, 0));
      CHECK(hipEventRecord(e0, s[0]));
      for (int k = 0; k < CHAIN; ++k) {
        hipStream_t st = mode ? s[k & 1] : s[0];
        int* wf = k ? flags + (k - 1) * 64 : nullptr;        // (a counter per launch, 256 bytes apart)
        hipLaunchKernelGGL(link, dim3(NWG), dim3(NT), 0, st, (const int*)buf[k % 3], buf[(k + 1) % 3], wf, flags + k * 64, mode, spins, tmo);
      }
      CHECK(hipEventRecord(ej, s[1])); CHECK(hipStreamWaitEvent(s[0], ej, 0));
      CHECK(hipEventRecord(e1, s[0]));
      CHECK(hipStreamSynchronize(s[0])); CHECK(hipStreamSynchronize(s[1]));
      float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
      if (rep >= 2 && ms < best) best = ms;
      CHECK(hipMemcpy(host.data(), buf[CHAIN % 3], ELEMS * 4, hipMemcpyDeviceToHost));
      for (int i = 0; i < ELEMS; ++i) if (host[i] != CHAIN) { ok = false; break; }
    }
    int t; CHECK(hipMemcpy(&t, tmo, 4, hipMemcpyDeviceToHost));
    printf("mode %c: %.2f us per launch (best of %d chains of %d)  values %s  timeouts %d\n", "ABCD"[mode], best * 1e3f / CHAIN, REPS, CHAIN, ok ? "all correct" : "WRONG", t);
  }
  return 0;
}
