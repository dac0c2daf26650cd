// Does an SMEM instruction count as a wait state between a VALU write of a VGPR and a DPP read of it on gfx950?
// (round 4: hipcc filled one of the two required wait states of the wave_sum DPP chain in lstm_step_fwd_kernel with an s_load_dwordx4; the
// sums of exactly those chains were sporadically wrong inside launch-tape replays.)  Three forms of the same 64-lane sum:
//   A: the gap between dependent DPP adds = { one independent VALU, one s_load_dword }   (what the compiler emitted)
//   B: the gap = s_nop 1                                                             (two explicit wait states)
//   C: the gap = { one independent VALU } only                                        (one wait state: known to be illegal, as a control)
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/_dpp_hazard_probe tools/dpp_hazard_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

#define STEP(ctrl) "v_add_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
#define BC15 "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n"
#define BC31 "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n"
template <int FORM>
__device__ __forceinline__ float wsum(float v, const int* sp) {
  int junk; float j2 = 1.f;
  if (FORM == 0) {
    asm volatile("s_nop 1\n" STEP("row_shr:1") "v_mov_b32 %2, 0\n s_load_dword %1, %3, 0x0\n" STEP("row_shr:2") "v_mov_b32 %2, 0\n s_load_dword %1, %3, 0x4\n"
                 STEP("row_shr:4") "v_mov_b32 %2, 0\n s_load_dword %1, %3, 0x8\n" STEP("row_shr:8") "v_mov_b32 %2, 0\n s_load_dword %1, %3, 0xc\n"
                 BC15 "v_mov_b32 %2, 0\n s_load_dword %1, %3, 0x10\n" BC31 "s_nop 1\n s_waitcnt lgkmcnt(0)\n"
                 : "+v"(v), "=&s"(junk), "=&v"(j2) : "s"(sp) : "memory");
  } else if (FORM == 1) {
    asm volatile("s_nop 1\n" STEP("row_shr:1") "s_nop 1\n" STEP("row_shr:2") "s_nop 1\n" STEP("row_shr:4") "s_nop 1\n" STEP("row_shr:8") "s_nop 1\n" BC15 "s_nop 1\n" BC31 "s_nop 1\n"
                 : "+v"(v));
  } else {
    asm volatile("s_nop 1\n" STEP("row_shr:1") "v_mov_b32 %1, 0\n" STEP("row_shr:2") "v_mov_b32 %1, 0\n" STEP("row_shr:4") "v_mov_b32 %1, 0\n" STEP("row_shr:8") "v_mov_b32 %1, 0\n"
                 BC15 "v_mov_b32 %1, 0\n" BC31 "s_nop 1\n"
                 : "+v"(v), "=&v"(j2));
  }
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
template <int FORM>
__global__ __launch_bounds__(256) void probe(const float* x, const int* sp, unsigned* bad, int iters) {
  const int lane = threadIdx.x & 63;
  unsigned nb = 0;
  for (int it = 0; it < iters; ++it) {
    const float v = x[(blockIdx.x * 256 + threadIdx.x + it * 64) & 65535];       // small integers: every order of additions is exact
    float ref = 0.f;
    for (int l = 0; l < 64; ++l) ref += __shfl(v, l, 64);
    const float s = wsum<FORM>(v, sp);
    if (s != ref) ++nb;
  }
  if (nb) atomicAdd(bad, nb);
}
__global__ void heavy(float* p, long n) {
  long i = blockIdx.x * (long)blockDim.x + threadIdx.x; float v = 0.f;
  for (long k = i; k < n; k += (long)gridDim.x * blockDim.x) v += p[k];
  if (v == 12345.f) p[0] = v;
}
int main() {
  float* x; int* sp; unsigned* bad; float* big; const long nbig = 64L << 20;
  CK(hipMalloc(&x, 65536 * 4)); CK(hipMalloc(&sp, 256)); CK(hipMalloc(&bad, 4)); CK(hipMalloc(&big, nbig * 4)); CK(hipMemset(big, 0, nbig * 4)); CK(hipMemset(sp, 0, 256));
  std::vector<float> h(65536); for (int i = 0; i < 65536; ++i) h[i] = (float)((i * 2654435761u >> 20) % 17) - 8.f;
  CK(hipMemcpy(x, h.data(), 65536 * 4, hipMemcpyHostToDevice));
  hipStream_t s, s2; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  const char* names[3] = {"A: VALU + s_load in the gap", "B: s_nop 1 in the gap", "C: one VALU in the gap (control)"};
  for (int co = 0; co < 2; ++co)
    for (int form = 0; form < 3; ++form) {
      CK(hipMemset(bad, 0, 4));
      for (int rep = 0; rep < 50; ++rep) {
        if (co) hipLaunchKernelGGL(heavy, dim3(2048), dim3(256), 0, s2, big, nbig);
        for (int g : {128, 1024}) {
          if (form == 0) hipLaunchKernelGGL(probe<0>, dim3(g), dim3(256), 0, s, (const float*)x, (const int*)sp, bad, 200);
          if (form == 1) hipLaunchKernelGGL(probe<1>, dim3(g), dim3(256), 0, s, (const float*)x, (const int*)sp, bad, 200);
          if (form == 2) hipLaunchKernelGGL(probe<2>, dim3(g), dim3(256), 0, s, (const float*)x, (const int*)sp, bad, 200);
        }
      }
      CK(hipDeviceSynchronize());
      unsigned nb; CK(hipMemcpy(&nb, bad, 4, hipMemcpyDeviceToHost));
      printf("co-running heavy kernel %d  %-36s wrong sums: %u of %ld\n", co, names[form], nb, 50L * (128 + 1024) * 256 * 200);
    }
  return 0;
}
