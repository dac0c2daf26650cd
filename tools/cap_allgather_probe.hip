// Price of the exchange primitive a one-XCD captioner recurrence would use (verdict r4 item 1a): 32 workgroups with equal
// blockIdx.x % 8 (one XCD under round-robin placement; speed only) all-gather small vectors as 8-byte {value, tag} granules:
// every producer publishes its share with ONE `global_store_dwordx2 sc1` wave instruction, every workgroup polls all granules
// with `sc1` loads (no counter, no fence, no flag), stages the values in LDS and passes one workgroup barrier.  Per iteration three
// all-gathers, as a token of the recurrence has: 512 values (att_h), 196 (attention dots), 512 (h); the value a workgroup publishes
// depends on the sum of what it gathered before, so the chain is a real dependency chain, and every gathered word is checked.
//   mode 0: granule buffers of their own per (iteration, phase), zeroed by the host before the launch, constant tag
//   mode 1: three buffers reused every iteration, tag = epoch (safe: a producer can only rewrite buffer X after an all-gather
//           that every workgroup entered after consuming X)
// "beside": a streaming kernel (16-byte loads over 1 GiB, every CU, ~32 KiB in flight per CU) runs on a second stream meanwhile.
//   hipcc --offload-arch=gfx950 -O3 tools/cap_allgather_probe.hip -o /tmp/cap_ag_probe && /tmp/cap_ag_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr int NWG = 32, NT = 512;
constexpr int PH_N[3] = {512, 196, 512};
constexpr int PH_OFF[3] = {0, 512, 768};     // granule offsets inside one iteration's block (1280 granules = 10 KiB)
constexpr int IT_GRAN = 1280;

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

// every thread polls granule `tid` (if < n) until its tag matches; returns false on timeout (bounded spin)
__device__ __forceinline__ bool gather(gu64* g, int n, unsigned tag, float* lds, unsigned* tmo) {
  const int tid = threadIdx.x;
  bool ok = tid >= n;
  unsigned spins = 0;
  float v = 0.f;
  while (true) {
    if (!ok) {
      const u64 x = __hip_atomic_load(g + tid, RLX_AGENT);
      if ((unsigned)(x >> 32) == tag) { ok = true; v = __uint_as_float((unsigned)x); }
    }
    if (__all(ok)) break;
    if (++spins > (1u << 22)) { if ((tid & 63) == 0) atomicOr(tmo, 1u); break; }
  }
  if (tid < n) lds[tid] = v;
  __syncthreads();
  return true;
}

// `spread` = 0: grid of 256, the workgroups with b % 8 == 0 take part (one XCD); 1: grid of 32, all take part (4 per XCD).
// The dynamic LDS request (`own` launches ask for 128 KiB) keeps LDS-using workgroups of other kernels off the probe's CUs.
__global__ __launch_bounds__(NT) void probe(u64* gran_, unsigned* xcc_seen, long long* ticks, unsigned* tmo, unsigned* errs, int iters, int mode, int spread) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  float* lds = dyn;
  float* red = dyn + 512;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) xcc_seen[b] = xcc_id();
  if (!spread && (b & 7) != 0) return;
  const int me = spread ? b : b >> 3;
  gu64* gran = (gu64*)gran_;
  float carry = 0.f;
  unsigned bad = 0;
  const long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ph = 0; ph < 3; ++ph) {
      const int n = PH_N[ph];
      gu64* g = gran + (mode == 0 ? (size_t)it * IT_GRAN : 0) + PH_OFF[ph];
      const unsigned tag = mode == 0 ? 1u : (unsigned)(it * 3 + ph + 1);
      // my share: granules [lo, hi) (16 each for n = 512; 7 each for the first 28 workgroups for n = 196), ONE wave instruction
      const int per = n == 512 ? 16 : 7;
      const int lo = me * per, hi = min(n, lo + per);
      if (tid < hi - lo) {
        const float val = (float)((it * 3 + ph) % 7) + (float)(lo + tid) * 0.001f + carry;
        __hip_atomic_store(g + lo + tid, ((u64)tag << 32) | __float_as_uint(val), RLX_AGENT);
      }
      gather(g, n, tag, lds, tmo);
      // check every word and form the next carry from the sum (a dependency on everything gathered)
      float s = 0.f;
      if (tid < n) {
        const float expect = (float)((it * 3 + ph) % 7) + (float)tid * 0.001f + carry;
        const float got = lds[tid];
        if (got != expect) ++bad;
        s = got - expect;
      }
      for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
      if ((tid & 63) == 0) red[tid >> 6] = s;
      __syncthreads();
      float tot = 0.f;
      for (int w = 0; w < NT / 64; ++w) tot += red[w];
      carry = tot == 0.f ? (float)((it + ph) & 3) * 0.25f : 1e9f;      // the same on every workgroup unless something was wrong
      __syncthreads();
    }
  }
  const long long t1 = __builtin_amdgcn_s_memrealtime();
  if (tid == 0) ticks[b] = t1 - t0;
  if (bad) atomicAdd(errs, bad);
}

__global__ __launch_bounds__(256) void stream_load(const float4* src, size_t n16, float* sink, int passes) {
  extern __shared__ __attribute__((aligned(16))) float ldyn[];     // only requested: 0 or 40 KiB (an LDS-using neighbour, as every GEMM tile of the step is)
  if (passes < 0) ldyn[threadIdx.x] = 0.f;
  float acc = 0.f;
  for (int p = 0; p < passes; ++p)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256 * 8) {
      float4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { const size_t j = i + (size_t)k * gridDim.x * 256; v[k] = j < n16 ? src[j] : make_float4(0, 0, 0, 0); }
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += v[k].x + v[k].w;
    }
  if (acc == 1.2345f) sink[0] = acc;
}

int main() {
  const int iters = 400;
  u64* gran; unsigned *xcc, *tmo, *errs; long long* ticks; float4* big; float* sink;
  const size_t gran_bytes = (size_t)iters * IT_GRAN * 8;
  hipMalloc(&gran, gran_bytes); hipMalloc(&xcc, 256 * 4); hipMalloc(&tmo, 4); hipMalloc(&errs, 4); hipMalloc(&ticks, 256 * 8);
  const size_t big_n16 = (size_t)1 << 26;   // 1 GiB
  hipMalloc(&big, big_n16 * 16); hipMalloc(&sink, 4);
  hipMemset(big, 0, big_n16 * 16);
  hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  hipFuncSetAttribute((const void*)stream_load, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  hipStream_t sa, sb; hipStreamCreate(&sa); hipStreamCreate(&sb);
  // load: 0 none; 1 streaming workgroups everywhere, also on the probe's CUs (4 x 256 threads per CU, launched first);
  //       2 the probe owns its CUs' LDS (128 KiB) and is launched first, the streaming workgroups (40 KiB of LDS each) fill the other CUs
  const char* lname[3] = {"idle chip", "streaming, shared CUs", "streaming, probe owns its CUs"};
  for (int spread = 0; spread < 2; ++spread)
    for (int load = 0; load < 3; ++load)
      for (int mode = 0; mode < 2; ++mode) {
        std::vector<double> per;
        unsigned tmo_h = 0, err_h = 0; int badx = 0;
        for (int rep = 0; rep < 5; ++rep) {
          hipMemsetAsync(gran, 0, gran_bytes, sa); hipMemsetAsync(tmo, 0, 4, sa); hipMemsetAsync(errs, 0, 4, sa); hipMemsetAsync(ticks, 0, 256 * 8, sa);
          hipStreamSynchronize(sa);
          const size_t plds = load == 2 ? 128 * 1024 : 4096;
          if (load == 1) hipLaunchKernelGGL(stream_load, dim3(256 * 4), dim3(256), 0, sb, big, big_n16, sink, 40);
          hipLaunchKernelGGL(probe, dim3(spread ? 32 : 256), dim3(NT), plds, sa, gran, xcc, ticks, tmo, errs, iters, mode, spread);
          if (load == 2) hipLaunchKernelGGL(stream_load, dim3(256 * 4), dim3(256), 40 * 1024, sb, big, big_n16, sink, 40);
          hipStreamSynchronize(sa);
          hipDeviceSynchronize();
          std::vector<long long> t(256); hipMemcpy(t.data(), ticks, 256 * 8, hipMemcpyDeviceToHost);
          std::vector<unsigned> x(256); hipMemcpy(x.data(), xcc, 256 * 4, hipMemcpyDeviceToHost);
          unsigned a, e; hipMemcpy(&a, tmo, 4, hipMemcpyDeviceToHost); hipMemcpy(&e, errs, 4, hipMemcpyDeviceToHost);
          tmo_h |= a; err_h += e;
          long long mx = 0; for (int i = 0; i < 256; ++i) mx = std::max(mx, t[i]);
          badx = 0; for (int i = 0; i < (spread ? 32 : 256); ++i) badx += (x[i] != x[i & 7]);
          if (rep) per.push_back(mx * 0.01 / (iters * 3.0));      // 100 MHz ticks -> us per all-gather
        }
        std::sort(per.begin(), per.end());
        printf("%-14s %-30s %-22s %6.2f us per all-gather (min %.2f max %.2f of 4 runs; timeouts %u, wrong words %u, off-class workgroups %d)\n",
               spread ? "4 per XCD" : "one XCD", lname[load], mode ? "reused buffers + epoch" : "own buffer per phase",
               per[per.size() / 2], per.front(), per.back(), tmo_h, err_h, badx);
      }
  printf("(%d x 3 all-gathers of 512 / 196 / 512 granules among 32 workgroups of 512 threads per run)\n", iters);
  return 0;
}
