// Direct 3x3 convolution (stride 1, pad 1) with an LDS-staged input patch, on CDNA4 matrix cores (gfx950), NHWC activations.
//
// Replaces cuDNN behind the 3x3 nn.Conv2d of the reference's bottlenecks / RPN, forward and data gradient
// (pyutils/mask-faster-rcnn/lib/nets/resnet_v1_cycle_res5_2.py:86, network_cycle_res5_2.py:236-239).
//
// Why another kernel: a small feature map (38x63 = 2394 pixels) gives a GEMM-shaped 3x3 kernel ~150 workgroups, and what a CU can pull
// out of L2 (20-30 B/clk, measured: the 64x64 implicit-GEMM tile moves 590 KB per workgroup in 14 us whatever its pipeline depth, slice
// size or split) bounds the launch.  The implicit GEMM re-reads every input pixel nine times, once per tap.  Here a workgroup owns
// BM consecutive pixels x BN output channels and stages, per 64-channel chunk, ONE patch of input rows that covers all nine taps
// (BM + 2 (W+1) + 2 rows); the taps read their MFMA fragments from that patch at row offsets.  Pixels live in a virtual layout with one
// zero column appended to every image row and one zero row to every image, so a tap that leaves the image lands on zeros and no
// fragment is ever masked.  Per 128x32 tile and 256 input channels the workgroup loads 132 KB of input + 147 KB of weights instead of
// 590 KB + 147 KB.
//
// Measured (round 2, tools/conv_bench.py): layer3 3x3 14.4 us against 15.1 us for the implicit GEMM, but layer2 (21 vs 13.5 us), the RPN
// convolution (79 vs 65 us) and layer4 on the map (45 vs 34 us) are slower, and so is the step (134.3 vs 136.1 img/s): with ~150 workgroups
// these launches are paced by fixed latencies (a one-slice launch already takes 4 us) and by instruction issue of a single wave per SIMD
// (PMC: 350 of a slice's 970 cycles are issue, 128 of them MFMA), not by bytes.  l2s_conv_igemm therefore uses this kernel only when
// L2S_CONV3X3_PATCH=1; it stays in the library as a checked alternative (tests/test_kernels_gpu.py::test_conv3x3_patch).
//
// Pipeline step = (channel chunk, filter row ky): the three taps of the row share one barrier: 24 MFMAs per wave and step.
// LDS: patch double-buffered per chunk, weights (3 taps x BN rows) double-buffered per step; 128-byte rows, 16-byte chunks XOR-swizzled
// by (row & 7) (conflict-free ds_read_b128 at any row offset).  Operands are fetched one step ahead through wave-uniform buffer
// descriptors (out-of-image rows carry the out-of-range offset and read zeros).
#include "common.h"
#include "../../include/lang2seg_hip.h"
#include <stdlib.h>

namespace {

constexpr unsigned OOR = 0x80000000u;
constexpr int RB = 128;        // bytes of channels per LDS row

template <typename T> struct MmaP;
template <> struct MmaP<bf16_t> {
  static __device__ __forceinline__ f32x4 run(const uint4& a, const uint4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct MmaP<float> {
  static __device__ __forceinline__ f32x4 run(const uint4& a, const uint4& b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    return c;
  }
};

typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));

// BM pixels x BN channels per workgroup; WGM x WGN waves; PRMAX = patch rows the LDS is sized for (>= BM + 2 (W + 1) + 2)
template <typename T, int BM, int BN, int WGM, int WGN, int PRMAX>
__global__ __launch_bounds__(64 * WGM * WGN) void conv3x3_patch_kernel(const l2s_conv_desc p) {
  constexpr int ES = (int)sizeof(T), VE = 16 / ES, BK = RB / ES;
  constexpr int NT = 64 * WGM * WGN;
  constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16;
  constexpr int NPA = (PRMAX * 8 + NT - 1) / NT;            // patch vectors per thread
  constexpr int NPB = (3 * BN * 8 + NT - 1) / NT;           // weight vectors per thread and step
  constexpr int ABUF = PRMAX * RB, BBUF = 3 * BN * RB;
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][ABUF] patches, [2][BBUF] weights

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int H = p.IH, W = p.IW, Wv = W + 1, Hv = H + 1;
  const int Mv = p.n_img * Hv * Wv;                          // virtual pixels
  const int PR = BM + 2 * Wv + 2;                            // patch rows in use
  int mt, nt;
  {
    const int MT = (Mv + BM - 1) / BM, NTl = (p.Cout + BN - 1) / BN, G = MT * NTl;
    const int L = blockIdx.x, x = L & 7, slot = L >> 3, q = G >> 3, r = G & 7;
    const int t = x * q + min(x, r) + slot;                  // per-XCD contiguous chunks of tiles, n fastest (the patch is shared)
    mt = t / NTl; nt = t - mt * NTl;
  }
  const int v0 = mt * BM, n0 = nt * BN;
  const long xpix = (long)p.n_img * H * W;
  const auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)(((xpix - 1) * p.ldx + p.Cin) * (long)ES), 0x00020000);
  const auto rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)((long)p.Cout * 9 * p.Cin * (long)ES), 0x00020000);

  // ---- loader coordinates (fixed for the whole K loop) ----
  unsigned voffA[NPA]; int ldsA[NPA];
#pragma unroll
  for (int j = 0; j < NPA; ++j) {
    const int vid = tid + j * NT, r = vid >> 3, cv = vid & 7;
    const int u = v0 - Wv - 1 + r;                           // virtual pixel of patch row r
    bool ok = r < PR && u >= 0 && u < Mv;
    int real = 0;
    if (ok) {
      const int t = u / Wv, x = u - t * Wv, n = t / Hv, y = t - n * Hv;
      ok = x < W && y < H;
      real = (n * H + y) * W + x;
    }
    voffA[j] = ok ? (unsigned)(((long)real * p.ldx + cv * VE) * ES) : OOR;
    ldsA[j] = r < PRMAX ? r * RB + ((cv ^ (r & 7)) << 4) : -1;
  }
  unsigned voffB[NPB]; int ldsB[NPB];
#pragma unroll
  for (int j = 0; j < NPB; ++j) {
    const int vid = tid + j * NT, r = vid >> 3, cv = vid & 7;    // r = kx * BN + nn
    const int kx = r / BN, nn = r - kx * BN, n = n0 + nn;
    const bool ok = r < 3 * BN && n < p.Cout;
    voffB[j] = ok ? (unsigned)((((long)n * 9 + kx) * p.Cin + cv * VE) * ES) : OOR;
    ldsB[j] = r < 3 * BN ? r * RB + ((cv ^ (r & 7)) << 4) : -1;
  }
  const int NC = p.Cin / BK;                                 // channel chunks
  const int NS = NC * 3;                                     // steps (chunk, ky)
  uint4 ra[NPA], rb[NPB];
  auto issueA = [&](int c) {
    const int so = c * BK * ES;
#pragma unroll
    for (int j = 0; j < NPA; ++j) ra[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, voffA[j], so, 0));
  };
  auto issueB = [&](int s) {
    const int c = s / 3, ky = s - c * 3;
    const int so = (ky * 3 * p.Cin + c * BK) * ES;
#pragma unroll
    for (int j = 0; j < NPB; ++j) rb[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rw, voffB[j], so, 0));
  };
  auto storeA = [&](int buf) {
    char* a = smem + buf * ABUF;
#pragma unroll
    for (int j = 0; j < NPA; ++j) if (ldsA[j] >= 0) *(uint4*)(a + ldsA[j]) = ra[j];
  };
  auto storeB = [&](int buf) {
    char* b = smem + 2 * ABUF + buf * BBUF;
#pragma unroll
    for (int j = 0; j < NPB; ++j) if (ldsB[j] >= 0) *(uint4*)(b + ldsB[j]) = rb[j];
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;
  auto compute = [&](int s) {
    const int c = s / 3, ky = s - c * 3;
    const char* a = smem + (c & 1) * ABUF;
    const char* b = smem + 2 * ABUF + (s & 1) * BBUF;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int rowA0 = wm * WM + fr + ky * Wv + kx;         // output pixel q, tap (ky, kx) -> patch row q + ky Wv + kx
      const int rowB0 = kx * BN + wn * WN + fr;
#pragma unroll
      for (int kg = 0; kg < 2; ++kg) {
        uint4 fa[TM], fb[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) { const int r = rowA0 + i * 16; fa[i] = *(const uint4*)(a + r * RB + (((kg * 4 + fg) ^ (r & 7)) << 4)); }
#pragma unroll
        for (int j = 0; j < TN; ++j) { const int r = rowB0 + j * 16; fb[j] = *(const uint4*)(b + r * RB + (((kg * 4 + fg) ^ (r & 7)) << 4)); }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = MmaP<T>::run(fb[j], fa[i], acc[i][j]);
      }
    }
  };

  // prologue: patch of chunk 0 and weights of step 0 to LDS; step 1's weights in flight
  issueA(0); issueB(0);
  storeA(0); storeB(0);
  if (NS > 1) issueB(1);
  bool a_pending = false;                                    // ra holds the next chunk's patch
  for (int s = 0; s < NS; ++s) {
    __syncthreads();
    const int c = s / 3, ky = s - c * 3;
    if (s + 1 < NS) {
      storeB((s + 1) & 1);                                   // (read last in step s-1, before this barrier)
      if (ky == 2 && a_pending) { storeA((c + 1) & 1); a_pending = false; }
      if (s + 2 < NS) issueB(s + 2);
      if (ky == 0 && c + 1 < NC) { issueA(c + 1); a_pending = true; }   // two steps ahead of its first use
    }
    compute(s);
  }

  // ---- epilogue: lane owns pixel (lane & 15) x 4 consecutive channels ((lane >> 4) * 4 + r) of each 16x16 tile ----
  const bool outf32 = (p.flags & L2S_CONV_OUT_F32) || ES == 4;
  const int OS = outf32 ? 4 : 2;
  const auto ry = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, 0x7FFFFFFF, 0x00020000);
  const auto radd = __builtin_amdgcn_make_buffer_rsrc((void*)(p.add ? p.add : p.y), 0, 0x7FFFFFFF, 0x00020000);
  const auto rref = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ref ? p.ref : p.y), 0, 0x7FFFFFFF, 0x00020000);
  auto unpack = [](const u32x4v& q, float (&f)[4]) {
    if (ES == 4) { f[0] = __uint_as_float(q.x); f[1] = __uint_as_float(q.y); f[2] = __uint_as_float(q.z); f[3] = __uint_as_float(q.w); }
    else { f[0] = __uint_as_float(q.x << 16); f[1] = __uint_as_float(q.x & 0xFFFF0000u); f[2] = __uint_as_float(q.y << 16); f[3] = __uint_as_float(q.y & 0xFFFF0000u); }
  };
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int v = v0 + wm * WM + i * 16 + fr;
    bool okp = v < Mv;
    int real = 0;
    if (okp) {
      const int t = v / Wv, x = v - t * Wv, n = t / Hv, y = t - n * Hv;
      okp = x < W && y < H;
      real = (n * H + y) * W + x;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * WN + j * 16 + fg * 4;
      const bool ok = okp && n < p.Cout;                     // Cout % 4 == 0
      float vv[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (p.bias && n < p.Cout) { const float4 b4 = *(const float4*)(p.bias + n); vv[0] += b4.x; vv[1] += b4.y; vv[2] += b4.z; vv[3] += b4.w; }
      if (p.add) {
        const unsigned o = ok ? (unsigned)(((long)real * p.ldadd + n) * ES) : OOR;
        u32x4v q;
        if (ES == 4) q = __builtin_amdgcn_raw_buffer_load_b128(radd, o, 0, 0);
        else { const u32x2 t2 = __builtin_amdgcn_raw_buffer_load_b64(radd, o, 0, 0); q = (u32x4v){t2.x, t2.y, 0u, 0u}; }
        float a4[4]; unpack(q, a4); vv[0] += a4[0]; vv[1] += a4[1]; vv[2] += a4[2]; vv[3] += a4[3];
      }
      if (p.flags & L2S_CONV_RELU) { vv[0] = fmaxf(vv[0], 0.f); vv[1] = fmaxf(vv[1], 0.f); vv[2] = fmaxf(vv[2], 0.f); vv[3] = fmaxf(vv[3], 0.f); }
      if (p.ref) {
        const unsigned o = ok ? (unsigned)(((long)real * p.ldref + n) * ES) : OOR;
        u32x4v q;
        if (ES == 4) q = __builtin_amdgcn_raw_buffer_load_b128(rref, o, 0, 0);
        else { const u32x2 t2 = __builtin_amdgcn_raw_buffer_load_b64(rref, o, 0, 0); q = (u32x4v){t2.x, t2.y, 0u, 0u}; }
        float r4[4]; unpack(q, r4);
#pragma unroll
        for (int e = 0; e < 4; ++e) if (!(r4[e] > 0.f)) vv[e] = 0.f;
      }
      const unsigned o = ok ? (unsigned)(((long)real * p.ldy + n) * OS) : OOR;
      if (OS == 4) {
        __builtin_amdgcn_raw_buffer_store_b128((u32x4v){__float_as_uint(vv[0]), __float_as_uint(vv[1]), __float_as_uint(vv[2]), __float_as_uint(vv[3])}, ry, o, 0, 0);
      } else {
        u32x2 pk;
        pk.x = (uint32_t)f2bf(vv[0]) | ((uint32_t)f2bf(vv[1]) << 16);
        pk.y = (uint32_t)f2bf(vv[2]) | ((uint32_t)f2bf(vv[3]) << 16);
        __builtin_amdgcn_raw_buffer_store_b64(pk, ry, o, 0, 0);
      }
    }
  }
}

template <typename T, int BM, int BN, int WGM, int WGN, int PRMAX>
int launch_patch(const l2s_conv_desc& d, hipStream_t st) {
  const int Mv = d.n_img * (d.IH + 1) * (d.IW + 1);
  dim3 grid(cdiv(Mv, BM) * cdiv(d.Cout, BN));
  const size_t lds = 2 * (size_t)PRMAX * RB + 2 * (size_t)3 * BN * RB;
  static bool attr_done = false;
  if (!attr_done) { (void)hipFuncSetAttribute((const void*)conv3x3_patch_kernel<T, BM, BN, WGM, WGN, PRMAX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done = true; }
  L2S_LAUNCH((conv3x3_patch_kernel<T, BM, BN, WGM, WGN, PRMAX>), grid, dim3(64 * WGM * WGN), lds, st, d);
  return l2s_check_launch();
}

}  // namespace

// 1: this problem is handled (launched); 0: not eligible (caller falls back to the implicit GEMM); < 0: error
extern "C" int l2s_conv3x3_patch_try(const l2s_conv_desc* d, int dtype, hipStream_t stream) {
  if (!d) return 0;
  const int bk = dtype == L2S_BF16 ? 64 : 32;
  if (d->KH != 3 || d->KW != 3 || d->stride != 1 || d->pad != 1 || d->OH != d->IH || d->OW != d->IW) return 0;
  if (d->Cin % bk || d->Cout % 4 || (d->flags & (L2S_CONV_SCATTER | L2S_CONV_DECONV2X2)) || d->tile) return 0;
  if (d->ldy % 4 || (d->add && d->ldadd % 4) || (d->ref && d->ldref % 4)) return 0;
  const long esz = dtype == L2S_BF16 ? 2 : 4;
  const long M = (long)d->n_img * d->IH * d->IW;
  if (M * d->ldx * esz >= (1L << 31) || (long)d->Cout * 9 * d->Cin * esz >= (1L << 31) || M * d->ldy * 4 >= (1L << 31)) return 0;
  // feature maps (one image, a few thousand pixels): the launches that are bound by what a CU pulls out of L2.  RoI batches (256 7x7
  // maps) stay on the implicit GEMM: a quarter of their virtual pixels would be padding.
  if (d->n_img != 1 || M > 16384) return 0;
  const int pr = 128 + 2 * (d->IW + 1) + 2;
  int rc;
  if (pr <= 288) rc = dtype == L2S_BF16 ? launch_patch<bf16_t, 128, 32, 4, 1, 288>(*d, stream) : launch_patch<float, 128, 32, 4, 1, 288>(*d, stream);
  else if (pr <= 416) rc = dtype == L2S_BF16 ? launch_patch<bf16_t, 128, 32, 4, 1, 416>(*d, stream) : launch_patch<float, 128, 32, 4, 1, 416>(*d, stream);
  else return 0;
  return rc == L2S_OK ? 1 : -rc;
}
