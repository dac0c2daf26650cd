// Do the per-CU pipes of an igemm slice overlap?  Workgroups of 256 threads run N iterations of the 64x64 tile's per-slice traffic in
// isolation and combined (bit mask MODE): 1 = 16 KiB of buffer_load_b128 from an L2-resident window, issued two iterations ahead, 2 = 16 KiB
// of ds_write_b128 + 32 KiB of ds_read_b128 in the tile's swizzled layout, 4 = 8 MFMA 16x16x32 bf16 per wave; one barrier per iteration.
// Further variants: the ring kernels' double-buffered order (PIPE), 32 workgroups sharing one window (hot lines), the loader's row-strided
// slabs (STRIDED), two slices per barrier (probe2), the 128x128 tile's slice (probe3), the LDS alone (probe4), the fill by LDS-DMA (probe5; probe6 for the 128x128 slice).
// Check the instruction counts of a variant in the ISA before believing its number (dead-code elimination removed the fragment reads of an
// early build).  Results and reading: DESIGN.md section 4.1c.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/_pipe_overlap_probe tools/pipe_overlap_probe.hip ; run it on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE, bool STRIDED, bool PIPE>
__global__ __launch_bounds__(256) void probe(const uint4* __restrict__ src, uint4* __restrict__ out, int iters, int win_u4, int share, int stride) {
  __shared__ uint4 lds[2048];                                   // 32 KiB: two 16 KiB slices
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)((blockIdx.x / share) & 255) * win_u4)   /* `share` workgroups read the same window in step */, 0, win_u4 * 16, 0x00020000);
  uint4 acc = make_uint4(0, 0, 0, 0), ld[3][4], fr[8];
  f32x4 c[4];
  for (int i = 0; i < 4; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 8; ++i) fr[i] = make_uint4(tid, i, 0x3f803f80u, 0x3f803f80u);
  for (int s = 0; s < 3; ++s) for (int j = 0; j < 4; ++j) ld[s][j] = make_uint4(s, j, tid, 1);
  unsigned off = tid * 16;
  const unsigned wbytes = (unsigned)win_u4 * 16u;
  // stride == 0: a contiguous 16 KiB per iteration.  stride > 0: the tile loader's pattern — 128 rows `stride` bytes apart, the 8 lanes of
  // a row read its next 128 bytes, the slab moves 128 bytes along the rows per iteration (rows = pixels / filters, bytes = channels)
  unsigned kofs = 0;
  auto issue = [&](uint4 (&d)[4]) {
    if (!STRIDED) {
#pragma unroll
      for (int j = 0; j < 4; ++j) d[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + j * 4096, 0, 0));
      off += 16384; if (off >= wbytes) off -= wbytes;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        d[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(((tid >> 3) + 32 * j) * stride + (tid & 7) * 16), (int)kofs, 0));
      kofs += 128; if (kofs >= (unsigned)stride) kofs = 0;
    }
  };
  // fragment-read pattern of the 64x64 tile: 128-byte rows, 16-byte chunk index xor (row & 7): conflict-free for ds_read_b128
  const int frow = lane & 15, fg = lane >> 4, wm = wave >> 1, wn = wave & 1;
  auto body = [&](uint4 (&cur)[4], uint4 (&nxt)[4], int buf) {
    if (PIPE) {
      // the ring kernels' order: barrier; fragment reads of the slice filled during the PREVIOUS iteration and the fill of the next slice
      // into the other buffer go to the LDS together; loads; MFMAs
      __syncthreads();
      if (MODE & 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int half = j >> 2, t = j & 1, kg = (j >> 1) & 1;
          const int row = (half ? 64 + wn * 32 : wm * 32) + t * 16 + frow, ch = (kg * 4 + fg) ^ (row & 7);
          fr[j] = lds[buf * 1024 + row * 8 + ch];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = (tid >> 3) + 32 * j, ch = (tid & 7) ^ (row & 7);
          lds[(buf ^ 1) * 1024 + row * 8 + ch] = (MODE & 1) ? cur[j] : make_uint4(tid, j, buf, 1);
        }
      } else if (MODE & 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(cur[j].x), "v"(cur[j].y), "v"(cur[j].z), "v"(cur[j].w));
      }
      if (MODE & 1) issue(cur);                                 // refill the set just stored (three sets: two iterations of lead)
      if (MODE & 4) {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          c[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fr[j]), __builtin_bit_cast(bf16x8, fr[(j + 1) & 7]), c[j & 3], 0, 0, 0);
      } else if (MODE & 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" :: "v"(fr[j].x), "v"(fr[j].y), "v"(fr[j].z), "v"(fr[j].w));
      }
      return;
    }
    if (MODE & 1) issue(nxt);                                   // two iterations ahead
    if (MODE & 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int row = (tid >> 3) + 32 * j, ch = (tid & 7) ^ (row & 7);
        lds[buf * 1024 + row * 8 + ch] = (MODE & 1) ? cur[j] : fr[j];
      }
    } else if (MODE & 1) {
#pragma unroll
      for (int j = 0; j < 4; ++j) asm volatile("" :: "v"(cur[j].x), "v"(cur[j].y), "v"(cur[j].z), "v"(cur[j].w));   // the loaded registers are needed here
    }
    __syncthreads();
    if (MODE & 2) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int half = j >> 2, t = j & 1, kg = (j >> 1) & 1;     // A rows (wm) / B rows (wn), two 16-row tiles, two k groups
        const int row = (half ? 64 + wn * 32 : wm * 32) + t * 16 + frow, ch = (kg * 4 + fg) ^ (row & 7);
        fr[j] = lds[buf * 1024 + row * 8 + ch];
      }
    }
    if (MODE & 4) {
      if (!(MODE & 2)) asm volatile("" : "+v"(fr[0].x), "+v"(fr[1].x));   // opaque operands: the MFMAs cannot be hoisted or folded
#pragma unroll
      for (int j = 0; j < 8; ++j)
        c[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fr[j]), __builtin_bit_cast(bf16x8, fr[(j + 1) & 7]), c[j & 3], 0, 0, 0);
    } else if (MODE & 2) {
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" :: "v"(fr[j].x), "v"(fr[j].y), "v"(fr[j].z), "v"(fr[j].w));           // every fragment read has to return
    }
  };
  if (MODE & 1) { issue(ld[0]); issue(ld[1]); if (PIPE) issue(ld[2]); }
  for (int it = 0; it < iters; it += 6) {                       // one barrier per iteration, as the tile kernels
    body(ld[0], ld[2], 0); body(ld[1], ld[0], 1); body(ld[2], ld[1], 0);
    body(ld[0], ld[2], 1); body(ld[1], ld[0], 0); body(ld[2], ld[1], 1);
  }
  for (int i = 0; i < 4; ++i) { asm volatile("" :: "v"(c[i][0]), "v"(c[i][1]), "v"(c[i][2]), "v"(c[i][3])); acc.x ^= __float_as_uint(c[i][0]); acc.y ^= __float_as_uint(c[i][1]); }
  for (int j = 0; j < 8; ++j) acc.z ^= fr[j].z;
  for (int s = 0; s < 3; ++s) for (int j = 0; j < 4; ++j) acc.w ^= ld[s][j].w;
  if (acc.x == 0x12345678u && acc.w == 0x9abcdef0u) out[tid] = acc;   // keep everything alive
}

// Two slices of K per barrier (the 256-byte-row idea): per iteration 32 KiB of loads, 32 KiB of fill, 64 KiB of fragment reads, 16 MFMAs per
// wave, ONE barrier; double-buffered order as above.  Reported per 64 of K (half an iteration).
__global__ __launch_bounds__(256) void probe2(const uint4* __restrict__ src, uint4* __restrict__ out, int iters, int win_u4) {
  __shared__ uint4 lds[4096];                                   // 64 KiB: two buffers of two slices
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)(blockIdx.x & 255) * win_u4), 0, win_u4 * 16, 0x00020000);
  uint4 acc = make_uint4(0, 0, 0, 0), ld[3][8], fr[16];
  f32x4 c[4];
  for (int i = 0; i < 4; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 16; ++i) fr[i] = make_uint4(tid, i, 0x3f803f80u, 0x3f803f80u);
  unsigned off = tid * 16;
  const unsigned wbytes = (unsigned)win_u4 * 16u;
  auto issue = [&](uint4 (&d)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + j * 4096, 0, 0));
    off += 32768; if (off >= wbytes) off -= wbytes;
  };
  const int frow = lane & 15, fg = lane >> 4, wm = wave >> 1, wn = wave & 1;
  auto body = [&](uint4 (&cur)[8], int buf) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int h = j >> 3, half = (j >> 2) & 1, t = j & 1, kg = (j >> 1) & 1;
      const int row = (half ? 64 + wn * 32 : wm * 32) + t * 16 + frow, ch = (kg * 4 + fg) ^ (row & 7);
      fr[j] = lds[buf * 2048 + h * 1024 + row * 8 + ch];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = (tid >> 3) + 32 * (j & 3), ch = (tid & 7) ^ (row & 7);
      lds[(buf ^ 1) * 2048 + (j >> 2) * 1024 + row * 8 + ch] = cur[j];
    }
    issue(cur);
#pragma unroll
    for (int j = 0; j < 16; ++j)
      c[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fr[j]), __builtin_bit_cast(bf16x8, fr[(j + 1) & 15]), c[j & 3], 0, 0, 0);
  };
  issue(ld[0]); issue(ld[1]); issue(ld[2]);
  for (int it = 0; it < iters; it += 6) {
    body(ld[0], 0); body(ld[1], 1); body(ld[2], 0); body(ld[0], 1); body(ld[1], 0); body(ld[2], 1);
  }
  for (int i = 0; i < 4; ++i) { asm volatile("" :: "v"(c[i][0]), "v"(c[i][1]), "v"(c[i][2]), "v"(c[i][3])); acc.x ^= __float_as_uint(c[i][0]); }
  if (acc.x == 0x12345678u) out[tid] = acc;
}

// The 128x128 tile's slice (4 waves, wave tile 64x64): 32 KiB of loads, 32 KiB of fill, 64 KiB of fragment reads, 32 MFMAs per wave, one barrier;
// double-buffered order.  Half the operand bytes per MFMA of the 64x64 tile.
__global__ __launch_bounds__(256) void probe3(const uint4* __restrict__ src, uint4* __restrict__ out, int iters, int win_u4) {
  __shared__ uint4 lds[4096];                                   // 64 KiB: two buffers of two slices
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + (size_t)(blockIdx.x & 255) * win_u4), 0, win_u4 * 16, 0x00020000);
  uint4 acc = make_uint4(0, 0, 0, 0), ld[3][8], fr[16];
  f32x4 c[16];
  for (int i = 0; i < 16; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < 16; ++i) fr[i] = make_uint4(tid, i, 0x3f803f80u, 0x3f803f80u);
  unsigned off = tid * 16;
  const unsigned wbytes = (unsigned)win_u4 * 16u;
  auto issue = [&](uint4 (&d)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) d[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, off + j * 4096, 0, 0));
    off += 32768; if (off >= wbytes) off -= wbytes;
  };
  const int frow = lane & 15, fg = lane >> 4, wm = wave >> 1, wn = wave & 1;
  auto body = [&](uint4 (&cur)[8], int buf) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int h = j >> 3, half = (j >> 2) & 1, t = j & 1, kg = (j >> 1) & 1;
      const int row = (half ? 64 + wn * 32 : wm * 32) + t * 16 + frow, ch = (kg * 4 + fg) ^ (row & 7);
      fr[j] = lds[buf * 2048 + h * 1024 + row * 8 + ch];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = (tid >> 3) + 32 * (j & 3), ch = (tid & 7) ^ (row & 7);
      lds[(buf ^ 1) * 2048 + (j >> 2) * 1024 + row * 8 + ch] = cur[j];
    }
    issue(cur);
#pragma unroll
    for (int kg = 0; kg < 2; ++kg)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          c[i * 4 + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fr[8 + kg * 4 + j]), __builtin_bit_cast(bf16x8, fr[kg * 4 + i]), c[i * 4 + j], 0, 0, 0);
  };
  issue(ld[0]); issue(ld[1]); issue(ld[2]);
  for (int it = 0; it < iters; it += 6) {
    body(ld[0], 0); body(ld[1], 1); body(ld[2], 0); body(ld[0], 1); body(ld[1], 0); body(ld[2], 1);
  }
  for (int i = 0; i < 16; ++i) { asm volatile("" :: "v"(c[i][0]), "v"(c[i][1]), "v"(c[i][2]), "v"(c[i][3])); acc.x ^= __float_as_uint(c[i][0]); }
  if (acc.x == 0x12345678u) out[tid] = acc;
}

// LDS rates alone: W ds_write_b128 and R ds_read_b128 per thread and iteration (the tile's swizzled, conflict-free addresses), one barrier.
template <int W, int R>
__global__ __launch_bounds__(256) void probe4(uint4* __restrict__ out, int iters) {
  __shared__ uint4 lds[2048];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int frow = lane & 15, fg = lane >> 4, wm = wave >> 1, wn = wave & 1;
  uint4 fr[R > 0 ? R : 1], v = make_uint4(tid, 1, 2, 3);
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 1;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int half = (j >> 2) & 1, t = j & 1, kg = (j >> 1) & 1;
      const int row = (half ? 64 + wn * 32 : wm * 32) + t * 16 + frow, ch = (kg * 4 + fg) ^ (row & 7);
      fr[j] = lds[buf * 1024 + (((j >> 3) * 512 + row * 8 + ch) & 1023)];
    }
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const int row = (tid >> 3) + 32 * (j & 3), ch = (tid & 7) ^ (row & 7);
      lds[(buf ^ 1) * 1024 + row * 8 + ch] = v;
    }
#pragma unroll
    for (int j = 0; j < R; ++j) asm volatile("" :: "v"(fr[j].x), "v"(fr[j].y), "v"(fr[j].z), "v"(fr[j].w));
  }
  if (v.x == 0x12345678u) out[tid] = v;
}
template <int W, int R> void run4(int g, uint4* out, int N, hipEvent_t e0, hipEvent_t e1) {
  float ms = 0;
  for (int w = 0; w < 3; ++w) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((probe4<W, R>), dim3(g), dim3(256), 0, 0, out, N);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  const double ns = ms * 1e6 / N, bytes = 256.0 * 16 * (W + R);
  printf("grid %3d  LDS only: %d writes + %2d reads (b128) per thread   %7.1f ns / iteration = %5.1f B/ns per CU\n", g, W, R, ns, bytes / ns); fflush(stdout);
}

// The 64x64 tile's slice with the fill done by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write): per iteration 4 DMA
// wave-instructions per wave (16 KiB per workgroup) into an LDS ring of four slices, two iterations ahead; 8 fragment reads and 8 MFMAs per
// wave; one barrier; vmcnt(4) before the barrier = the fill issued in the previous iteration has landed.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <bool MFMA>
__global__ __launch_bounds__(256) void probe5(const uint4* __restrict__ src, uint4* __restrict__ out, int iters, int win_u4) {
  __shared__ uint4 lds[4096];                                   // 64 KiB: four slices
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const uint4* base = src + (size_t)(blockIdx.x & 255) * win_u4;
  uint4 acc = make_uint4(0, 0, 0, 0), fr[8];
  f32x4 c[4];
  for (int i = 0; i < 4; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fg = lane >> 4, wm = wave >> 1, wn = wave & 1;
  unsigned off = 0;
  auto fill = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) glds16(base + ((off + (wave * 4 + j) * 64 + lane) & (unsigned)(win_u4 - 1)), lds0 + buf * 16384 + (wave * 4 + j) * 1024);
    off += 1024;
  };
  fill(0); fill(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 3;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int half = j >> 2, t = j & 1, kg = (j >> 1) & 1;
      const int row = (half ? 64 + wn * 32 : wm * 32) + t * 16 + frow, ch = (kg * 4 + fg) ^ (row & 7);
      fr[j] = lds[buf * 1024 + row * 8 + ch];
    }
    fill((it + 2) & 3);
    if (MFMA) {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        c[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fr[j]), __builtin_bit_cast(bf16x8, fr[(j + 1) & 7]), c[j & 3], 0, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) asm volatile("" :: "v"(fr[j].x), "v"(fr[j].y), "v"(fr[j].z), "v"(fr[j].w));
    }
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int i = 0; i < 4; ++i) { asm volatile("" :: "v"(c[i][0]), "v"(c[i][1]), "v"(c[i][2]), "v"(c[i][3])); acc.x ^= __float_as_uint(c[i][0]); }
  if (acc.x == 0x12345678u) out[tid] = acc;
}

// the 128x128 tile's slice (probe3) with the LDS-DMA fill: 8 DMA instructions per wave and iteration into four 32 KiB slices (dynamic LDS)
template <bool MFMA>
__global__ __launch_bounds__(256) void probe6(const uint4* __restrict__ src, uint4* __restrict__ out, int iters, int win_u4) {
  extern __shared__ uint4 lds[];                                // 128 KiB: four slices of 32 KiB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds;
  const uint4* base = src + (size_t)(blockIdx.x & 255) * win_u4;
  uint4 acc = make_uint4(0, 0, 0, 0), fr[16];
  f32x4 c[16];
  for (int i = 0; i < 16; ++i) c[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int frow = lane & 15, fg = lane >> 4, wm = wave >> 1, wn = wave & 1;
  unsigned off = 0;
  auto fill = [&](int buf) {
#pragma unroll
    for (int j = 0; j < 8; ++j) glds16(base + ((off + (wave * 8 + j) * 64 + lane) & (unsigned)(win_u4 - 1)), lds0 + buf * 32768 + (wave * 8 + j) * 1024);
    off += 2048;
  };
  fill(0); fill(1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int it = 0; it < iters; ++it) {
    const int buf = it & 3;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int half = j >> 3, t = j & 3, kg = (j >> 2) & 1;        // A rows (wm) / B rows (wn), four 16-row tiles, two k groups
      const int row = (half ? 128 + wn * 64 : wm * 64) + t * 16 + frow, ch = (kg * 4 + fg) ^ (row & 7);
      fr[j] = lds[buf * 2048 + row * 8 + ch];
    }
    fill((it + 2) & 3);
    if (MFMA) {
#pragma unroll
      for (int kg = 0; kg < 2; ++kg)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int jj = 0; jj < 4; ++jj)
            c[i * 4 + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fr[8 + kg * 4 + jj]), __builtin_bit_cast(bf16x8, fr[kg * 4 + i]), c[i * 4 + jj], 0, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) asm volatile("" :: "v"(fr[j].x), "v"(fr[j].y), "v"(fr[j].z), "v"(fr[j].w));
    }
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int i = 0; i < 16; ++i) { asm volatile("" :: "v"(c[i][0]), "v"(c[i][1]), "v"(c[i][2]), "v"(c[i][3])); acc.x ^= __float_as_uint(c[i][0]); }
  if (acc.x == 0x12345678u) out[tid] = acc;
}

template <int MODE, bool STRIDED = false, bool PIPE = false> float run(int g, const uint4* src, uint4* out, int N, int win, hipEvent_t e0, hipEvent_t e1, int share = 1, int stride = 0) {
  float ms = 0;
  for (int w = 0; w < 3; ++w) {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((probe<MODE, STRIDED, PIPE>), dim3(g), dim3(256), 0, 0, src, out, N, win, share, stride);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
  }
  return ms;
}

int main() {
  const int G = 256, N = 12000, WIN = 4096;                     // 64 KiB window per workgroup: L2 resident (2 MiB per XCD)
  uint4 *src, *out;
  CK(hipMalloc(&src, (size_t)G * WIN * 16)); CK(hipMalloc(&out, 4096)); CK(hipMemset(src, 1, (size_t)G * WIN * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[8] = {"barriers only", "loads", "lds", "loads+lds", "mfma", "loads+mfma", "lds+mfma", "loads+lds+mfma"};
  for (int g : {120, 256, 512}) {
    float ms[8] = {run<0>(g, src, out, N, WIN, e0, e1), run<1>(g, src, out, N, WIN, e0, e1), run<2>(g, src, out, N, WIN, e0, e1), run<3>(g, src, out, N, WIN, e0, e1),
                   run<4>(g, src, out, N, WIN, e0, e1), run<5>(g, src, out, N, WIN, e0, e1), run<6>(g, src, out, N, WIN, e0, e1), run<7>(g, src, out, N, WIN, e0, e1)};
    for (int m = 0; m < 8; ++m) { printf("grid %3d  %-16s %7.1f ns / iteration\n", g, names[m], ms[m] * 1e6 / N); fflush(stdout); }
  }
  // the ring kernels' double-buffered order (fragment reads of slice t and the fill of slice t+1 in one LDS phase)
  for (int g : {120, 256, 512}) {
    const float a = run<2, false, true>(g, src, out, N, WIN, e0, e1), b = run<3, false, true>(g, src, out, N, WIN, e0, e1), c = run<6, false, true>(g, src, out, N, WIN, e0, e1), d = run<7, false, true>(g, src, out, N, WIN, e0, e1);
    printf("grid %3d  pipelined order   lds %7.1f   loads+lds %7.1f   lds+mfma %7.1f   loads+lds+mfma %7.1f ns / iteration\n", g, a * 1e6 / N, b * 1e6 / N, c * 1e6 / N, d * 1e6 / N); fflush(stdout);
  }
  for (int g : {120, 256}) {
    float ms = 0;
    for (int w = 0; w < 3; ++w) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(probe2, dim3(g), dim3(256), 0, 0, (const uint4*)src, out, N / 2, WIN);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("grid %3d  two slices per barrier, loads+lds+mfma %7.1f ns per 64 of K\n", g, ms * 1e6 / N); fflush(stdout);
  }
  for (int g : {120, 256, 512}) {
    float ms = 0;
    for (int w = 0; w < 3; ++w) {
      CK(hipEventRecord(e0, 0));
      hipLaunchKernelGGL(probe3, dim3(g), dim3(256), 0, 0, (const uint4*)src, out, N / 2, WIN);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    }
    printf("grid %3d  128x128 tile slice (32 MFMAs per wave), loads+lds+mfma %7.1f ns per slice\n", g, ms * 1e6 / (N / 2)); fflush(stdout);
  }
  run4<4, 0>(256, out, N, e0, e1); run4<0, 8>(256, out, N, e0, e1); run4<0, 16>(256, out, N, e0, e1); run4<4, 8>(256, out, N, e0, e1); run4<8, 16>(256, out, N, e0, e1); run4<0, 16>(512, out, N, e0, e1);
  for (int g : {120, 256, 512}) {
    float ms[2] = {0, 0};
    for (int w = 0; w < 3; ++w) {
      CK(hipEventRecord(e0, 0)); hipLaunchKernelGGL(probe5<false>, dim3(g), dim3(256), 0, 0, (const uint4*)src, out, N, WIN); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[0], e0, e1));
      CK(hipEventRecord(e0, 0)); hipLaunchKernelGGL(probe5<true>, dim3(g), dim3(256), 0, 0, (const uint4*)src, out, N, WIN); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms[1], e0, e1));
    }
    printf("grid %3d  LDS-DMA fill: fill+reads %7.1f   fill+reads+mfma %7.1f ns / iteration\n", g, ms[0] * 1e6 / N, ms[1] * 1e6 / N); fflush(stdout);
  }
  CK(hipFuncSetAttribute((const void*)probe6<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  for (int g : {256}) {
    float ms = 0;
    for (int w = 0; w < 3; ++w) { CK(hipEventRecord(e0, 0)); hipLaunchKernelGGL(probe6<true>, dim3(g), dim3(256), 131072, 0, (const uint4*)src, out, N / 2, WIN); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); }
    printf("grid %3d  128x128 slice, LDS-DMA fill: fill+reads+mfma %7.1f ns / iteration\n", g, ms * 1e6 / (N / 2)); fflush(stdout);
  }
  // hot lines: `share` workgroups (consecutive ids = different XCDs) stream the SAME window at the same time, as the workgroups of one
  // tile column do with a weight slice
  for (int share : {1, 32}) {
    const float a = run<1>(256, src, out, N, WIN, e0, e1, share), b = run<7>(256, src, out, N, WIN, e0, e1, share);
    printf("grid 256  share %3d   loads %7.1f   loads+lds+mfma %7.1f ns / iteration\n", share, a * 1e6 / N, b * 1e6 / N); fflush(stdout);
  }
  // row-strided slabs (128 rows x 128 bytes per iteration), 8 distinct windows of 128 rows shared by 32 workgroups each (L2 resident)
  for (int stride : {2048, 4608}) {
    const int win = 128 * stride / 16;
    for (int g : {120, 256}) {
      const float a = run<1, true>(g, src, out, N, win, e0, e1, 32, stride), b = run<3, true>(g, src, out, N, win, e0, e1, 32, stride), c = run<7, true>(g, src, out, N, win, e0, e1, 32, stride);
      printf("grid %3d  row stride %4d B   loads %7.1f   loads+lds %7.1f   loads+lds+mfma %7.1f ns / iteration\n", g, stride, a * 1e6 / N, b * 1e6 / N, c * 1e6 / N); fflush(stdout);
    }
  }
  return 0;
}
