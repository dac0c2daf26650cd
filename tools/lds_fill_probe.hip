// What a CU can pull into LDS, by LDS-DMA (`buffer_load_dwordx4 ... lds`, 1 KiB per wave instruction) and by register staging (global_load_dwordx4 +
// ds_write_b128), from a working set that fits one XCD's L2 (2 MB shared by every workgroup), the Infinity Cache (64 MB, every workgroup its own
// stream) or neither (1 GB).  Round 4: igemm_dma256_kernel and the LDS-DMA 1x1 weight-gradient tile both settle at 32 KB per slice and CU in
// ~0.95 us = 34 GB/s per CU whatever their pipeline depth - is that the fill path or where the bytes come from?
//   hipcc --offload-arch=gfx950 -O3 tools/lds_fill_probe.hip -o /tmp/lds_fill && /tmp/lds_fill
// One workgroup of 8 waves per CU (grid 256 / 128 / 32), every wave keeps D requests of 1 KiB in flight and waits with a counted vmcnt.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef int i32x4s __attribute__((ext_vector_type(4)));

template <int MODE, int D>   // MODE 0: LDS-DMA, 1: registers + ds_write
__global__ __launch_bounds__(512) void fill(const char* __restrict__ src, long span, long stride_wg, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const char* base = src + (blockIdx.x * stride_wg) % span;
  i32x4s r; r.x = (int)(uintptr_t)src; r.y = (int)((uintptr_t)src >> 32); r.z = 0x7fffffff; r.w = 0x00020000;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (unsigned)(wave * D * 1024);
  unsigned off = (unsigned)((base - src) + wave * 1024 + lane * 16);
  const unsigned wrap = (unsigned)(span - 8 * 1024 * 2);
  float acc = 0.f;
  if (MODE == 0) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        const unsigned lds = lds0 + d * 1024;
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(off), "s"(r), "s"(lds) : "memory");
        off += 8 * 1024; if (off >= wrap) off -= wrap;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  } else {
    for (int it = 0; it < iters; ++it) {
      uint4 v[D];
#pragma unroll
      for (int d = 0; d < D; ++d) { v[d] = *(const uint4*)(src + off); off += 8 * 1024; if (off >= wrap) off -= wrap; }
#pragma unroll
      for (int d = 0; d < D; ++d) *(uint4*)(smem + wave * D * 1024 + d * 1024 + lane * 16) = v[d];
    }
  }
  __syncthreads();
  acc = *(float*)(smem + (threadIdx.x & 255) * 4);
  if (acc == 123.456f) sink[0] = acc;
}

template <int MODE, int D>
static void run(const char* name, const char* buf, long span, long stride, int grid) {
  const int iters = 4000 / D;
  hipFuncSetAttribute((const void*)fill<MODE, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * D * 1024);
  float* sink; hipMalloc(&sink, 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(a);
    hipLaunchKernelGGL((fill<MODE, D>), dim3(grid), dim3(512), 8 * D * 1024, 0, buf, span, stride, iters, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
  }
  const double bytes = (double)grid * 8 * iters * D * 1024;
  printf("%-28s %-10s depth %d  grid %3d: %7.1f GB/s per CU  %6.2f TB/s chip\n", name, MODE ? "registers" : "LDS-DMA", D, grid, bytes / grid / best / 1e6, bytes / best / 1e9);
  hipFree(sink);
}

int main() {
  char* buf; hipMalloc(&buf, 1L << 30); hipMemset(buf, 1, 1L << 30);
  struct { const char* name; long span, stride; } sets[3] = {{"2 MB shared (L2)", 2L << 20, 0}, {"64 MB (Infinity Cache)", 64L << 20, 256L << 10}, {"1 GB (HBM)", 1L << 30, 4L << 20}};
  for (int g : {256, 128}) {
    for (auto& s : sets) {
      run<0, 2>(s.name, buf, s.span, s.stride, g); run<0, 4>(s.name, buf, s.span, s.stride, g); run<0, 8>(s.name, buf, s.span, s.stride, g);
      run<1, 4>(s.name, buf, s.span, s.stride, g); run<1, 8>(s.name, buf, s.span, s.stride, g);
    }
  }
  return 0;
}
