// Implicit-GEMM convolution on CDNA4 matrix cores (gfx950), NHWC activations.
//
//   l2s_conv_igemm : forward conv / 1x1 / Linear / data-gradient, no im2col buffer.
//                    C[m][n] = sum_k A(m,k) * W[n][k],  m = output pixel, n = output channel,
//                    k = (tap, cin); A gathered on the fly from the NHWC input (zero padded).
//   (the weight gradient lives in conv_wgrad.hip)
//
// Replaces the cuDNN calls behind nn.Conv2d / F.conv2d / nn.Linear in the reference
// (pyutils/mask-faster-rcnn/lib/nets/resnet_v1_cycle_res5_2.py:83-88,121,147,324-335 and
// network_cycle_res5_2.py:236-251,279-301).
//
// Kernels: igemm_ring_kernel (128x128, and 64x64 for f32 / fp32-output launches), igemm_ws64_kernel (bf16 64x64: four loader waves +
// four multiplier waves), igemm_sp_kernel (224x128 / 256x128, 8 waves, software pipelined, f32 mode); igemm_kernel is the
// round-1 generic form (K tails, >= 2 GiB operands); igemm_dma_kernel / igemm_ks64_kernel / igemm_p3_kernel (LDS-DMA fill, round 3);
// igemm_pdma_kernel (the LDS-DMA tile as a persistent workgroup, round 4, on request).  conv_plan() picks one per launch.
// Tiling: 256 threads = 4 waves (2x2); block tile BMxBN (128x128 or 64x64); K consumed in 128-byte
// slices per row (64 bf16 / 32 f32) staged global -> registers -> LDS (double buffered, one barrier per
// slice); 16x16 MFMA tiles: v_mfma_f32_16x16x32_bf16 or the exact-f32 v_mfma_f32_16x16x4_f32
// ("verification mode").  LDS rows are padded by 16 B so the ds_read_b128 fragment reads are
// conflict-free.  The MFMA is issued with operands swapped (rows = n, cols = m) so that each lane ends
// up with 4 consecutive output channels of one pixel -> 8/16-byte NHWC stores.
#include "common.h"
#include "roi_sample.h"
#include "../../include/lang2seg_hip.h"
#include <stdlib.h>

namespace {

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
  static __device__ __forceinline__ f32x4 run(const uint4& a, const uint4& b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
  }
};
template <> struct Mma<float> {
  // 16 k per 64-byte group: lane group g holds k = 4g..4g+3; step e multiplies k = 4g+e of every group.
  static __device__ __forceinline__ f32x4 run(const uint4& a, const uint4& b, f32x4 c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    return c;
  }
};

constexpr int ROWB = 128;        // bytes of K per LDS row per slice
constexpr int LROW = ROWB + 16;  // padded LDS row

// ---- branch-free epilogue (all row pitches and Cout multiples of 4, extents < 2 GiB): every bias / residual / mask
// operand of the wave's TM x TN sub-tiles is requested before the first use (buffer loads; out-of-range rows and
// channels carry offset 0x80000000, so loads return 0 and stores are dropped), instead of one dependent HBM round trip
// per sub-tile ----
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
template <typename T, int TM, int TN, int WM, int WN, bool OUTF32>
__device__ __forceinline__ void igemm_epilogue_fast(const l2s_conv_desc& p, f32x4 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int fr, int fg, int M) {
  constexpr unsigned NOPE = 0x80000000u;
  constexpr int ES = (int)sizeof(T), OS = (OUTF32 || sizeof(T) == 4) ? 4 : 2;
  const int ohw = p.OH * p.OW;
  const int Cq = (p.flags & L2S_CONV_DECONV2X2) ? (p.Cout >> 2) : p.Cout;
  const auto ry = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, 0x7FFFFFFF, 0x00020000);
  const auto radd = __builtin_amdgcn_make_buffer_rsrc((void*)(p.add ? p.add : p.y), 0, 0x7FFFFFFF, 0x00020000);
  const auto rref = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ref ? p.ref : p.y), 0, 0x7FFFFFFF, 0x00020000);
  const auto rbias = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias ? (const void*)p.bias : (const void*)p.y), 0, 0x7FFFFFFF, 0x00020000);
  int orow[TM][TN], ocol[TM][TN]; bool ok[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * WM + i * 16 + fr;
    int n_img = 0, oy = 0, ox = 0, r0 = m;
    if (p.flags & (L2S_CONV_SCATTER | L2S_CONV_DECONV2X2)) {
      n_img = m / ohw; const int rem = m - n_img * ohw; oy = rem / p.OW; ox = rem - oy * p.OW;
      if (p.flags & L2S_CONV_SCATTER) r0 = (n_img * p.out_h + oy * p.out_stride) * p.out_w + ox * p.out_stride;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * WN + j * 16 + fg * 4;
      int oc = n, r2 = r0;
      if (p.flags & L2S_CONV_DECONV2X2) {
        const int tap = n / Cq; oc = n - tap * Cq;
        r2 = (n_img * 2 * p.OH + 2 * oy + (tap >> 1)) * (2 * p.OW) + 2 * ox + (tap & 1);
      }
      orow[i][j] = r2; ocol[i][j] = oc;
      ok[i][j] = (m < M) && (n < p.Cout);
    }
  }
  f32x4 bv[TN];
  if (p.bias) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * WN + j * 16 + fg * 4;
      const unsigned o = n < p.Cout ? (unsigned)(ocol[0][j] * 4) : NOPE;
      bv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rbias, o, 0, 0));
    }
  }
  // residual and ReLU-mask operands: 4 channels = 8 bytes (bf16) or 16 bytes (f32) per lane and sub-tile
  u32x4v av[TM][TN], rv[TM][TN];
  if (p.add) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const unsigned o = ok[i][j] ? (unsigned)((orow[i][j] * p.ldadd + ocol[i][j]) * ES) : NOPE;
        if (ES == 4) av[i][j] = __builtin_amdgcn_raw_buffer_load_b128(radd, o, 0, 0);
        else { const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(radd, o, 0, 0); av[i][j] = (u32x4v){t.x, t.y, 0u, 0u}; }
      }
  }
  if (p.ref) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const unsigned o = ok[i][j] ? (unsigned)((orow[i][j] * p.ldref + ocol[i][j]) * ES) : NOPE;
        if (ES == 4) rv[i][j] = __builtin_amdgcn_raw_buffer_load_b128(rref, o, 0, 0);
        else { const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rref, o, 0, 0); rv[i][j] = (u32x4v){t.x, t.y, 0u, 0u}; }
      }
  }
  auto unpack = [](const u32x4v& q, float (&f)[4]) {
    if (ES == 4) { f[0] = __uint_as_float(q.x); f[1] = __uint_as_float(q.y); f[2] = __uint_as_float(q.z); f[3] = __uint_as_float(q.w); }
    else { f[0] = __uint_as_float(q.x << 16); f[1] = __uint_as_float(q.x & 0xFFFF0000u); f[2] = __uint_as_float(q.y << 16); f[3] = __uint_as_float(q.y & 0xFFFF0000u); }
  };
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      if (p.bias) { v[0] += bv[j][0]; v[1] += bv[j][1]; v[2] += bv[j][2]; v[3] += bv[j][3]; }
      if (p.add) { float a[4]; unpack(av[i][j], a); v[0] += a[0]; v[1] += a[1]; v[2] += a[2]; v[3] += a[3]; }
      if (p.flags & L2S_CONV_RELU) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
      if (p.ref) {
        float r[4]; unpack(rv[i][j], r);
#pragma unroll
        for (int e = 0; e < 4; ++e) if (!(r[e] > 0.f)) v[e] = 0.f;
      }
      const unsigned o = ok[i][j] ? (unsigned)((orow[i][j] * p.ldy + ocol[i][j]) * OS) : NOPE;
      if (OS == 4) {
        __builtin_amdgcn_raw_buffer_store_b128((u32x4v){__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])}, ry, o, 0, 0);
      } else {
        u32x2 pk;
        pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
        pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
        __builtin_amdgcn_raw_buffer_store_b64(pk, ry, o, 0, 0);
      }
    }
}

// ---- LDS-staged epilogue of the 128x128 bf16 tile (4 waves as 2 x 2, wave tile 64 x 64): the MFMA layout gives a lane 4 channels of 16
// different pixels, i.e. 8-byte accesses in 32-byte runs to the output, the residual and the ReLU-mask operand.  Staging the fp32 tile
// (+ bias) through LDS, 64 rows at a time, turns them into 16-byte accesses that cover whole 256-byte rows.  Same arithmetic order as
// igemm_epilogue_fast (bias, residual, ReLU, mask in fp32, one rounding). ----
template <int TM, int TN, int WM, int WN, int WGM, int NT, int BN, int PASSES>
__device__ __forceinline__ void igemm_epilogue_lds128(const l2s_conv_desc& p, f32x4 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int fr, int fg, int M, char* smem,
                                                      const u32x4v* pre_add = nullptr, const u32x4v* pre_ref = nullptr, const f32x4* pre_bias = nullptr) {
  // pre_add / pre_ref (single-pass tiles): the residual / ReLU-mask operands of this thread's chunks, requested by the caller BEFORE its K
  // loop with igemm_epilogue_prefetch (same chunk map), so that their latency does not sit between the last MFMA and the stores
  // (tile width BN = WGN x WN; PASSES passes of RP rows: the waves whose rows fall into pass h stage their sub-tiles, then all NT threads
  // store them; one pass when the whole fp32 tile fits the kernel's LDS)
  constexpr int LDW = BN + 4;                                // floats per staged row (+ 4: shifts consecutive rows by 4 banks)
  constexpr int RP = WM * WGM / PASSES, CPR = BN / 8;        // rows per pass, 8-channel chunks per row
  constexpr int CH = RP * CPR, IT = (CH + NT - 1) / NT;      // chunks of one pass, chunks per thread
  static_assert((WM * WGM) % PASSES == 0 && RP % WM == 0, "passes");
  constexpr unsigned NOPE = 0x80000000u;
  float* st = (float*)smem;
  const auto ry = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, 0x7FFFFFFF, 0x00020000);
  const auto radd = __builtin_amdgcn_make_buffer_rsrc((void*)(p.add ? p.add : p.y), 0, 0x7FFFFFFF, 0x00020000);
  const auto rref = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ref ? p.ref : p.y), 0, 0x7FFFFFFF, 0x00020000);
  const int tid = threadIdx.x;
  f32x4 bv[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * WN + j * 16 + fg * 4;
    bv[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (pre_bias) bv[j] = pre_bias[j];                       // (requested by the caller before its K loop)
    else if (p.bias && n < p.Cout) bv[j] = *(const f32x4*)(p.bias + n);
  }
#pragma unroll
  for (int h = 0; h < PASSES; ++h) {
    __syncthreads();                                         // the K loop's (h = 0) / the previous pass's LDS reads are done
    if ((wm * WM) / RP == h) {
      const int r0 = wm * WM - h * RP;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          f32x4 v = acc[i][j];
          v[0] += bv[j][0]; v[1] += bv[j][1]; v[2] += bv[j][2]; v[3] += bv[j][3];
          *(f32x4*)(st + (r0 + i * 16 + fr) * LDW + wn * WN + j * 16 + fg * 4) = v;
        }
    }
    __syncthreads();
    // RP rows x CPR chunks of 8 channels; CPR lanes cover one output row of the tile
    u32x4v av[IT], rv[IT];
    unsigned off[IT];
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      const int c = tid + NT * k, row = c / CPR, col = (c % CPR) * 8;
      const int m = m0 + h * RP + row, n = n0 + col;
      const bool ok = c < CH && m < M && n < p.Cout;
      off[k] = ok ? (unsigned)m : NOPE;
      if (PASSES == 1 && pre_add) { if (p.add) av[k] = pre_add[k]; }
      else if (p.add) av[k] = __builtin_amdgcn_raw_buffer_load_b128(radd, ok ? (unsigned)((m * p.ldadd + n) * 2) : NOPE, 0, 0);
      if (PASSES == 1 && pre_ref) { if (p.ref) rv[k] = pre_ref[k]; }
      else if (p.ref) rv[k] = __builtin_amdgcn_raw_buffer_load_b128(rref, ok ? (unsigned)((m * p.ldref + n) * 2) : NOPE, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      const int c = min(tid + NT * k, CH - 1), row = c / CPR, col = (c % CPR) * 8;
      const f32x4 lo = *(const f32x4*)(st + row * LDW + col), hi = *(const f32x4*)(st + row * LDW + col + 4);
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      if (p.add) {
        const unsigned w[4] = {av[k].x, av[k].y, av[k].z, av[k].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] += __uint_as_float(w[e] << 16); v[2 * e + 1] += __uint_as_float(w[e] & 0xFFFF0000u); }
      }
      if (p.flags & L2S_CONV_RELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if (p.ref) {
        const unsigned w[4] = {rv[k].x, rv[k].y, rv[k].z, rv[k].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (!(__uint_as_float(w[e] << 16) > 0.f)) v[2 * e] = 0.f;
          if (!(__uint_as_float(w[e] & 0xFFFF0000u) > 0.f)) v[2 * e + 1] = 0.f;
        }
      }
      u32x4v pk;
      pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16); pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
      pk.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16); pk.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
      const int n = n0 + col;
      __builtin_amdgcn_raw_buffer_store_b128(pk, ry, off[k] != NOPE ? (unsigned)((off[k] * p.ldy + n) * 2) : NOPE, 0, 0);
    }
  }
}

// the chunk map of igemm_epilogue_lds128 (one pass): thread tid owns chunks tid + NT k of the ROWS x BN tile, 8 channels each
template <int ROWS, int NT, int BN>
__device__ __forceinline__ void igemm_epilogue_prefetch(const l2s_conv_desc& p, int m0, int n0, int M, int tid, u32x4v* av, u32x4v* rv) {
  constexpr int CPR = BN / 8, CH = ROWS * CPR, IT = (CH + NT - 1) / NT;
  constexpr unsigned NOPE = 0x80000000u;
  const auto radd = __builtin_amdgcn_make_buffer_rsrc((void*)(p.add ? p.add : p.y), 0, 0x7FFFFFFF, 0x00020000);
  const auto rref = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ref ? p.ref : p.y), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    const int c = tid + NT * k, row = c / CPR, col = (c % CPR) * 8;
    const int m = m0 + row, n = n0 + col;
    const bool ok = c < CH && m < M && n < p.Cout;
    if (p.add) av[k] = __builtin_amdgcn_raw_buffer_load_b128(radd, ok ? (unsigned)((m * p.ldadd + n) * 2) : NOPE, 0, 0);
    if (p.ref) rv[k] = __builtin_amdgcn_raw_buffer_load_b128(rref, ok ? (unsigned)((m * p.ldref + n) * 2) : NOPE, 0, 0);
  }
}

// ---- shared epilogue: lane owns pixel (lane&15) x 4 consecutive channels ((lane>>4)*4 + r) of each 16x16 accumulator tile ----
template <typename T, int TM, int TN, int WM, int WN, bool OUTF32>
__device__ __forceinline__ void igemm_epilogue(const l2s_conv_desc& p, f32x4 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int fr, int fg, int M) {
  const int ohw = p.OH * p.OW;
  const int Cq = (p.flags & L2S_CONV_DECONV2X2) ? (p.Cout >> 2) : p.Cout;
  const bool vec_ok = ((p.ldy & 3) == 0) && ((p.ldadd & 3) == 0) && ((p.ldref & 3) == 0) && ((Cq & 3) == 0);
  {
    const long orows = (p.flags & L2S_CONV_SCATTER) ? (long)p.n_img * p.out_h * p.out_w : ((p.flags & L2S_CONV_DECONV2X2) ? 4L * M : (long)M);
    const long lim = 1L << 31;
    const bool small = orows * p.ldy * 4 < lim && (!p.add || orows * p.ldadd * 4 < lim) && (!p.ref || orows * p.ldref * 4 < lim);
    if (vec_ok && small) { igemm_epilogue_fast<T, TM, TN, WM, WN, OUTF32>(p, acc, m0, n0, wm, wn, fr, fg, M); return; }
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * WM + i * 16 + fr;
    if (m >= M) continue;
    long orow = m;
    int n_img = 0, oy = 0, ox = 0;
    if (p.flags & (L2S_CONV_SCATTER | L2S_CONV_DECONV2X2)) {
      n_img = m / ohw; int rem = m - n_img * ohw; oy = rem / p.OW; ox = rem - oy * p.OW;
      if (p.flags & L2S_CONV_SCATTER) orow = ((long)n_img * p.out_h + (long)oy * p.out_stride) * p.out_w + (long)ox * p.out_stride;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * WN + j * 16 + fg * 4;
      if (n >= p.Cout) continue;
      int oc = n; long orow2 = orow;
      if (p.flags & L2S_CONV_DECONV2X2) {
        int tap = n / Cq; oc = n - tap * Cq;
        orow2 = ((long)n_img * 2 * p.OH + 2 * oy + (tap >> 1)) * (2 * p.OW) + 2 * ox + (tap & 1);
      }
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[i][j][r];
      const bool full = (n + 3 < p.Cout);
      if (full && vec_ok) {
        if (p.bias) { const float4 b4 = *(const float4*)(p.bias + oc); v[0] += b4.x; v[1] += b4.y; v[2] += b4.z; v[3] += b4.w; }
        if (p.add) {
          const T* ap = (const T*)p.add + orow2 * p.ldadd + oc;
          if (sizeof(T) == 4) { const float4 a4 = *(const float4*)ap; v[0] += a4.x; v[1] += a4.y; v[2] += a4.z; v[3] += a4.w; }
          else { const uint2 a2 = *(const uint2*)ap; v[0] += __uint_as_float(a2.x << 16); v[1] += __uint_as_float(a2.x & 0xFFFF0000u);
                 v[2] += __uint_as_float(a2.y << 16); v[3] += __uint_as_float(a2.y & 0xFFFF0000u); }
        }
        if (p.flags & L2S_CONV_RELU) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
        if (p.ref) {
          const T* rp = (const T*)p.ref + orow2 * p.ldref + oc;
          float rv[4];
          if (sizeof(T) == 4) { const float4 r4 = *(const float4*)rp; rv[0] = r4.x; rv[1] = r4.y; rv[2] = r4.z; rv[3] = r4.w; }
          else { const uint2 r2 = *(const uint2*)rp; rv[0] = __uint_as_float(r2.x << 16); rv[1] = __uint_as_float(r2.x & 0xFFFF0000u);
                 rv[2] = __uint_as_float(r2.y << 16); rv[3] = __uint_as_float(r2.y & 0xFFFF0000u); }
#pragma unroll
          for (int r = 0; r < 4; ++r) if (!(rv[r] > 0.f)) v[r] = 0.f;
        }
        if (OUTF32 || sizeof(T) == 4) {
          *(float4*)((float*)p.y + orow2 * p.ldy + oc) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          uint2 pk;
          pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
          pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
          *(uint2*)((T*)p.y + orow2 * p.ldy + oc) = pk;
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (n + r >= p.Cout) break;
          if (p.bias) v[r] += p.bias[oc + r];
          if (p.add) v[r] += Elem<T>::ld((const T*)p.add + orow2 * p.ldadd + oc + r);
          if (p.flags & L2S_CONV_RELU) v[r] = fmaxf(v[r], 0.f);
          if (p.ref) { if (!(Elem<T>::ld((const T*)p.ref + orow2 * p.ldref + oc + r) > 0.f)) v[r] = 0.f; }
          if (OUTF32) ((float*)p.y)[orow2 * p.ldy + oc + r] = v[r];
          else Elem<T>::st((T*)p.y + orow2 * p.ldy + oc + r, v[r]);
        }
      }
    }
  }
}

// split-K: the workgroup's fp32 partial tile goes to slab blockIdx.z of the workspace ([split][M][Cout] floats); splitk_reduce_kernel adds
// the slabs in slab order and applies the epilogue, so the result does not depend on which workgroup finishes first (no atomics).
// Cout % 4 == 0 (checked by the launcher): a lane's four channels are one 16-byte store.
template <int TM, int TN, int WM, int WN>
__device__ __forceinline__ void igemm_splitk_slab(const l2s_conv_desc& p, f32x4 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int fr, int fg, int M) {
  float* slab = p.ws + (long)blockIdx.z * M * p.Cout;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * WM + i * 16 + fr;
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * WN + j * 16 + fg * 4;
      if (n < p.Cout) *(f32x4*)(slab + (long)m * p.Cout + n) = acc[i][j];
    }
  }
}

template <typename T, int BM, int BN, bool OUTF32>
__global__ __launch_bounds__(256) void igemm_kernel(const l2s_conv_desc p) {
  constexpr int VE = 16 / (int)sizeof(T);    // elements per 16-byte vector
  constexpr int BK = ROWB / (int)sizeof(T);  // K elements per slice
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 16, TN = WN / 16;
  constexpr int NA = BM / 32, NB = BN / 32;  // 16-byte vectors per thread per slice
  constexpr int BUF = (BM + BN) * ROWB;      // unpadded 128-byte rows, XOR-swizzled 16-byte chunks (chunk ^ (row & 7))
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int M = p.n_img * p.OH * p.OW;
  const int K = p.KH * p.KW * p.Cin;
  // XCD-aware tile order (1-D grid): consecutive workgroup ids are dealt round-robin over the 8 XCDs, so id % 8 labels
  // the XCD.  The tile list is cut into 8 contiguous chunks, one per XCD, ordered so that a chunk's operands fit the
  // XCD's 4 MiB L2 (xcd_mode 0: M-chunks, n fastest -> A tile reused across its n-tiles, all of W resident;
  // xcd_mode 1: N-chunks, m fastest -> one W column block resident, A streamed).  Speed only, never correctness.
  int mt, nt;
  {
    const int MT = (M + BM - 1) / BM, NT = (p.Cout + BN - 1) / BN, G = MT * NT;
    const int L = blockIdx.x, x = L & 7, slot = L >> 3, q = G >> 3, r = G & 7;
    const int t = x * q + min(x, r) + slot;
    if (p.xcd_mode == 0) { mt = t / NT; nt = t - mt * NT; } else { nt = t / MT; mt = t - nt * MT; }
  }
  const int m0 = mt * BM, n0 = nt * BN;
  const T* __restrict__ X = (const T*)p.x;
  const T* __restrict__ Wt = (const T*)p.w;

  // ---- per-thread loader coordinates: rows lrow + 32 j, 16-byte chunk cv; fixed for the whole K loop ----
  const int lrow = tid >> 3, cv = tid & 7;
  const int wchunk = ((cv ^ (lrow & 7)) << 4);          // swizzled LDS chunk offset (row & 7 == lrow & 7 for every j)
  int a_iy0[NA], a_ix0[NA]; long a_base[NA]; bool a_ok[NA];
  const int ohw = p.OH * p.OW;
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    int m = m0 + lrow + 32 * j;
    a_ok[j] = m < M;
    int mm = a_ok[j] ? m : 0;
    int n_img = mm / ohw, rem = mm - n_img * ohw;
    int oy = rem / p.OW, ox = rem - oy * p.OW;
    a_iy0[j] = oy * p.stride - p.pad;
    a_ix0[j] = ox * p.stride - p.pad;
    a_base[j] = (long)n_img * p.IH * p.IW;
  }
  const T* pb[NB]; bool b_ok[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    int n = n0 + lrow + 32 * j;
    b_ok[j] = n < p.Cout;
    pb[j] = Wt + (long)(b_ok[j] ? n : 0) * K + cv * VE;
  }

  const int KT_all = (K + BK - 1) / BK;
  int kt0 = 0, KT = KT_all;
  if (gridDim.z > 1) {                      // split-K: this workgroup owns slices [kt0, KT)
    const int per = (KT_all + gridDim.z - 1) / gridDim.z;
    kt0 = blockIdx.z * per; KT = min(KT_all, kt0 + per);          // (the launcher picks a split that leaves no range empty)
  }
  // loader state, advanced incrementally: (tap, c0) and one pointer per row; pointers are rebuilt only when the tap changes
  int c0 = kt0 * BK, tap = 0;
  if (p.KH * p.KW > 1) { tap = c0 / p.Cin; c0 -= tap * p.Cin; }
  const T* pa[NA]; bool va[NA];
  auto set_tap = [&](int t) {
    const int ky = t / p.KW, kx = t - ky * p.KW;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int iy = a_iy0[j] + ky, ix = a_ix0[j] + kx;
      va[j] = a_ok[j] && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
      pa[j] = X + ((a_base[j] + (long)(va[j] ? iy : 0) * p.IW + (va[j] ? ix : 0)) * p.ldx + c0 + cv * VE);
    }
  };
  set_tap(tap);
#pragma unroll
  for (int j = 0; j < NB; ++j) pb[j] += (long)kt0 * BK;

  uint4 ra[NA], rb[NB];
  auto load_slice = [&]() {
    const bool kin = (c0 + cv * VE) < p.Cin;             // K tail (only 1x1 / Linear with Cin % BK != 0)
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (va[j] && kin) v = *(const uint4*)pa[j];
      ra[j] = v;
      pa[j] += BK;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (b_ok[j] && kin) v = *(const uint4*)pb[j];
      rb[j] = v;
      pb[j] += BK;
    }
    c0 += BK;
    if (c0 >= p.Cin && p.KH * p.KW > 1) { c0 = 0; ++tap; if (tap < p.KH * p.KW) set_tap(tap); }
  };
  auto store_slice = [&](int buf) {
    char* a = smem + buf * BUF + lrow * ROWB + wchunk;
    char* b = a + BM * ROWB;
#pragma unroll
    for (int j = 0; j < NA; ++j) *(uint4*)(a + 32 * j * ROWB) = ra[j];
#pragma unroll
    for (int j = 0; j < NB; ++j) *(uint4*)(b + 32 * j * ROWB) = rb[j];
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  load_slice();
  store_slice(0);
  __syncthreads();
  const int fr = lane & 15, fg = lane >> 4;
  const int swz = fr & 7;
  const int offa = (wm * WM + fr) * ROWB, offb = BM * ROWB + (wn * WN + fr) * ROWB;
  for (int kt = kt0; kt < KT; ++kt) {
    const int cur = (kt - kt0) & 1;
    if (kt + 1 < KT) load_slice();
    const char* base = smem + cur * BUF;
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
      const int ch = ((kg * 4 + fg) ^ swz) << 4;
      uint4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *(const uint4*)(base + offa + i * 16 * ROWB + ch);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *(const uint4*)(base + offb + j * 16 * ROWB + ch);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::run(fb[j], fa[i], acc[i][j]);
    }
    if (kt + 1 < KT) store_slice(cur ^ 1);
    __syncthreads();
  }

  if (gridDim.z > 1) { igemm_splitk_slab<TM, TN, WM, WN>(p, acc, m0, n0, wm, wn, fr, fg, M); return; }
  igemm_epilogue<T, TM, TN, WM, WN, OUTF32>(p, acc, m0, n0, wm, wn, fr, fg, M);
}

// ------------------------------------------------------------------------------------------------
// Ring variant (the default): same tiling / LDS image / epilogue as igemm_kernel, but the operands are fetched with
// buffer_load_dwordx4 through two wave-uniform buffer descriptors into a ring of D register sets, D slices ahead of
// the MFMAs.  Per load the address is descriptor + 32-bit per-row voffset (rebuilt only when the filter tap changes)
// + scalar slice offset, so the K loop carries no per-load address arithmetic; out-of-image taps, rows >= M and
// channels >= Cout get voffset = 0x80000000, which the descriptor's range check turns into zeros without a branch
// (branch-free loads are what lets hipcc keep D slices in flight with counted vmcnt instead of draining at every use).
// Iteration t:  barrier -> ds_write slice t+1 (issued D iterations ago) -> issue slice t+1+D -> MFMAs on slice t.
// Requires Cin % BK == 0 for every tap (no K tail) and operand extents < 2 GiB; the launcher falls back otherwise.
// ------------------------------------------------------------------------------------------------
// minimum workgroups per CU the 64x64 / depth-2 ring tile is compiled for: 3 (168 VGPRs, three waves per SIMD) measured 174.5 vs 170.5 img/s
// for 1 (144 VGPRs but scheduled for two); 4 spills
#ifndef L2S_RING64_MINWG
#define L2S_RING64_MINWG 3
#endif
constexpr unsigned OOR = 0x80000000u;

template <typename T, int BM, int BN, int WGM, int WGN, int D, bool OUTF32>
__global__ __launch_bounds__(64 * WGM * WGN, (BM * BN >= 128 * 128 && WGM * WGN == 4) ? 2 : ((BM * BN <= 64 * 64 && D == 2 && L2S_RING64_MINWG > 1) ? L2S_RING64_MINWG : 1)) void igemm_ring_kernel(const l2s_conv_desc p) {
  if (p.prio) __builtin_amdgcn_s_setprio(3);
  constexpr int VE = 16 / (int)sizeof(T);
  constexpr int RB = ROWB;                      // bytes of K per LDS row per slice
  constexpr int BK = RB / (int)sizeof(T);
  constexpr int CPR = RB / 16;                  // 16-byte chunks (= loader threads) per row
  constexpr int NTG = 64 * WGM * WGN;           // WGM x WGN waves, wave tile WM x WN
  constexpr int LR = NTG / CPR;                 // rows covered by one loader pass
  constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16;
  constexpr int NA = BM / LR, NB = BN / LR;
  constexpr int BUF = (BM + BN) * RB;
  extern __shared__ __attribute__((aligned(16))) char smem_all[];

  const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int M = p.n_img * p.OH * p.OW;
  const int K = p.KH * p.KW * p.Cin;
  int mt, nt;
  {
    const int MT = (M + BM - 1) / BM, NT = (p.Cout + BN - 1) / BN, G = MT * NT;
    const int L = blockIdx.x, x = L & 7, slot = L >> 3, q = G >> 3, r = G & 7;
    const int t = x * q + min(x, r) + slot;
    if (p.xcd_mode == 0) { mt = t / NT; nt = t - mt * NT; } else { nt = t / MT; mt = t - nt * MT; }
  }
  const int m0 = mt * BM, n0 = nt * BN;
  char* smem = smem_all;
  const long xpix = (long)p.n_img * p.IH * p.IW;
  const auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)(((xpix - 1) * p.ldx + p.Cin) * (long)sizeof(T)), 0x00020000);
  const auto rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)((long)p.Cout * K * (long)sizeof(T)), 0x00020000);

  const int lrow = tid / CPR, cv = tid % CPR;
  const int wchunk = ((cv ^ (lrow & (CPR - 1))) << 4);      // (LR is a multiple of CPR: every row of this thread has the same low bits)
  int a_iy0[NA], a_ix0[NA], a_base[NA]; bool a_ok[NA];
  const int ohw = p.OH * p.OW;
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    int m = m0 + lrow + LR * j;
    a_ok[j] = m < M;
    int mm = a_ok[j] ? m : 0;
    int n_img = mm / ohw, rem = mm - n_img * ohw;
    int oy = rem / p.OW, ox = rem - oy * p.OW;
    a_iy0[j] = oy * p.stride - p.pad;
    a_ix0[j] = ox * p.stride - p.pad;
    a_base[j] = n_img * p.IH * p.IW;
  }
  unsigned voffB[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int n = n0 + lrow + LR * j;
    voffB[j] = n < p.Cout ? (unsigned)(((long)n * K + cv * VE) * (long)sizeof(T)) : OOR;
  }
  // issue-side state (uniform): slice index, tap, channel offset inside the tap
  const int KT = K / BK;
  const int taps = p.KH * p.KW;
  int it = 0, c0 = 0, tap = 0;
  unsigned voffA[NA];
  auto set_tap = [&](int t) {
    const int ky = t / p.KW, kx = t - ky * p.KW;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int iy = a_iy0[j] + ky, ix = a_ix0[j] + kx;
      const bool v = a_ok[j] && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
      voffA[j] = v ? (unsigned)(((long)(a_base[j] + iy * p.IW + ix) * p.ldx + cv * VE) * (long)sizeof(T)) : OOR;
    }
  };
  set_tap(tap);
  auto issue = [&](uint4 (&a)[NA], uint4 (&b)[NB]) {
    const int sa = c0 * (int)sizeof(T), sb = it * (BK * (int)sizeof(T));
#pragma unroll
    for (int j = 0; j < NA; ++j) a[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, voffA[j], sa, 0));
#pragma unroll
    for (int j = 0; j < NB; ++j) b[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rw, voffB[j], sb, 0));
    ++it; c0 += BK;
    if (c0 >= p.Cin && taps > 1) { c0 = 0; ++tap; if (tap < taps) set_tap(tap); }
  };
  auto store_slice = [&](int buf, const uint4 (&ra)[NA], const uint4 (&rb)[NB]) {
    char* a = smem + buf * BUF + lrow * RB + wchunk;
    char* b = a + BM * RB;
#pragma unroll
    for (int j = 0; j < NA; ++j) *(uint4*)(a + LR * j * RB) = ra[j];
#pragma unroll
    for (int j = 0; j < NB; ++j) *(uint4*)(b + LR * j * RB) = rb[j];
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // ring set s holds slice k with k % D == s.  Prologue: slices 0..D-1 issued; slice 0 to LDS; slice D re-issued into set 0.
  uint4 ra[D][NA], rb[D][NB];
#pragma unroll
  for (int s = 0; s < D; ++s)
    if (s < KT) issue(ra[s], rb[s]);
  store_slice(0, ra[0], rb[0]);
  if (D < KT) issue(ra[0], rb[0]);

  const int fr = lane & 15, fg = lane >> 4;
  const int swz = fr & (CPR - 1);
  const int offa = (wm * WM + fr) * RB, offb = BM * RB + (wn * WN + fr) * RB;
  auto compute = [&](int t) {
    const char* base = smem + (t & 1) * BUF;
#pragma unroll
    for (int kg = 0; kg < RB / 64; ++kg) {
      const int ch = ((kg * 4 + fg) ^ swz) << 4;
      uint4 fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *(const uint4*)(base + offa + i * 16 * RB + ch);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[j] = *(const uint4*)(base + offb + j * 16 * RB + ch);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::run(fb[j], fa[i], acc[i][j]);
    }
  };
  // fragment reads of a whole slice first (slice t's buffer is complete since the barrier), THEN the LDS fill of slice t+1 and the next
  // global loads, then the MFMAs: the reads' latency passes under the fill / load issue instead of in front of every MFMA group
  constexpr int KG = RB / 64;
  uint4 fra[KG][TM], frb[KG][TN];
  auto read_all = [&](int t) {
    const char* base = smem + (t & 1) * BUF;
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
      const int ch = ((kg * 4 + fg) ^ swz) << 4;
#pragma unroll
      for (int i = 0; i < TM; ++i) fra[kg][i] = *(const uint4*)(base + offa + i * 16 * RB + ch);
#pragma unroll
      for (int j = 0; j < TN; ++j) frb[kg][j] = *(const uint4*)(base + offb + j * 16 * RB + ch);
    }
  };
  auto mma_all = [&]() {
#pragma unroll
    for (int kg = 0; kg < KG; ++kg)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::run(frb[kg][j], fra[kg][i], acc[i][j]);
  };
  constexpr bool early = BM * BN <= 64 * 64;    // (the 128x128 tile has no registers to spare)
  // steady state: every iteration stores slice t+1 and issues slice t+1+D, no conditions (so the compiler's vmcnt is the
  // exact count of the D-1 younger sets); unrolled by D so that the ring set index is a compile-time constant.
  int t0 = 0;
  if (early) {
    for (; t0 + 2 * D <= KT; t0 += D) {
#pragma unroll
      for (int s = 0; s < D; ++s) {
        const int t = t0 + s;
        const int nxt = (s + 1) % D;
        __syncthreads();
        read_all(t);
        store_slice((t + 1) & 1, ra[nxt], rb[nxt]);
        issue(ra[nxt], rb[nxt]);
        mma_all();
      }
    }
  } else {
    for (; t0 + 2 * D <= KT; t0 += D) {
#pragma unroll
      for (int s = 0; s < D; ++s) {
        const int t = t0 + s;
        const int nxt = (s + 1) % D;
        __syncthreads();
        store_slice((t + 1) & 1, ra[nxt], rb[nxt]);
        issue(ra[nxt], rb[nxt]);
        compute(t);
      }
    }
  }
  // tail: fewer than 2 D slices left
#pragma unroll
  for (int s = 0; s < 2 * D; ++s) {
    const int t = t0 + s;
    if (t < KT) {
      const int nxt = (s + 1) % D;
      __syncthreads();
      if (early) read_all(t);
      if (t + 1 < KT) {
        store_slice((t + 1) & 1, ra[nxt], rb[nxt]);
        if (t + 1 + D < KT) issue(ra[nxt], rb[nxt]);
      }
      if (early) mma_all(); else compute(t);
    }
  }
  if constexpr (sizeof(T) == 2 && !OUTF32 && ((BM == 128 && BN == 128) || (BM == 128 && BN == 64) || (BM == 64 && BN == 64)) && WGM == 2 && WGN == 2) {
    // whole-row 16-byte accesses through an LDS-staged fp32 tile when every row pitch allows it
    const bool plain = !(p.flags & (L2S_CONV_DECONV2X2 | L2S_CONV_SCATTER)) && !(p.ldy & 7) && !(p.ldadd & 7) && !(p.ldref & 7) && !(p.Cout & 7) &&
                       !((uintptr_t)p.y & 15) && !((uintptr_t)p.add & 15) && !((uintptr_t)p.ref & 15) && !((uintptr_t)p.bias & 15) &&
                       (long)M * p.ldy * 2 < (1L << 31) && (!p.add || (long)M * p.ldadd * 2 < (1L << 31)) && (!p.ref || (long)M * p.ldref * 2 < (1L << 31));
    if (plain) { igemm_epilogue_lds128<TM, TN, WM, WN, WGM, NTG, BN, (BM == 128 ? 2 : 1)>(p, acc, m0, n0, wm, wn, fr, fg, M, smem_all); return; }
  }
  igemm_epilogue<T, TM, TN, WM, WN, OUTF32>(p, acc, m0, n0, wm, wn, fr, fg, M);
}

// ------------------------------------------------------------------------------------------------
// Wave-specialised 64x64 bf16 tile (8 waves): waves 4-7 LOAD (global -> register ring -> LDS), waves 0-3 MULTIPLY (LDS -> fragments ->
// MFMA).  The plain ring kernel gives one wave per SIMD the whole instruction stream of a slice, in order: address updates, buffer
// loads, the vmcnt wait, LDS fill, barrier, fragment reads, the lgkmcnt wait, MFMAs (970 cycles per slice with 128 of MFMA, 4.1a).
// Here a SIMD holds one wave of each kind, so the loader's waits pass under the multiplier's MFMAs and the multiplier's stream per
// slice is a barrier, 8 ds_read_b128 and 8 MFMAs, the MFMAs of slice t-1 issued behind the reads of slice t.
//   barrier #k: slice k complete in LDS buffer k & 1 (loaders arrive after their ds_writes, multipliers after the reads of slice
//   k-1 returned: __syncthreads' lgkmcnt(0)), so the loaders may overwrite buffer (k+1) & 1 with slice k+1 right behind it.
// The loaders leave after the last slice (a finished wave no longer counts at s_barrier); the epilogue is the ring kernel's.
// ------------------------------------------------------------------------------------------------
template <int D, int BN, bool STAMP = false>
__global__ __launch_bounds__(512, 2) void igemm_ws64_kernel(const l2s_conv_desc p) {
  if (p.prio) __builtin_amdgcn_s_setprio(3);
  // STAMP (tools/ws64_stamps.py, algo 8): lane 0 of the first multiplier wave stores the 100 MHz clock at four points of every workgroup
  // into p.ws: [workgroup][entry, first slice landed, K loop done, stores drained]
  unsigned long long t_entry = 0;
  if constexpr (STAMP) t_entry = __builtin_amdgcn_s_memrealtime();
  typedef bf16_t T;
  constexpr int BM = 64, RB = 128, BK = 64, CPR = 8, LR = 32, NA = BM / LR, NB = BN / LR;
  constexpr int WM = 32, WN = BN / 2, TM = 2, TN = WN / 16, KG = 2, BUF = (BM + BN) * RB;
  extern __shared__ __attribute__((aligned(16))) char smem_all[];
  char* smem = smem_all;
  const int wave_all = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int M = p.n_img * p.OH * p.OW;
  const int K = p.KH * p.KW * p.Cin;
  int mt, nt;
  {
    const int MT = (M + BM - 1) / BM, NT = (p.Cout + BN - 1) / BN, G = MT * NT;
    const int L = blockIdx.x, x = L & 7, slot = L >> 3, q = G >> 3, r = G & 7;
    const int t = x * q + min(x, r) + slot;
    if (p.xcd_mode == 0) { mt = t / NT; nt = t - mt * NT; } else { nt = t / MT; mt = t - nt * MT; }
  }
  const int m0 = mt * BM, n0 = nt * BN;
  // split-K (gridDim.z > 1): this workgroup multiplies slices [kt0, kt0 + KT) and leaves its partial tile in slab blockIdx.z
  int kt0 = 0, KT = K / BK;
  if (gridDim.z > 1) {
    const int per = (KT + (int)gridDim.z - 1) / (int)gridDim.z;
    kt0 = (int)blockIdx.z * per; KT = min(KT, kt0 + per) - kt0;
  }

  if (wave_all >= 4) {
    // ---------------- loaders ----------------
    const int tid = (int)threadIdx.x - 256;
    const long xpix = (long)p.n_img * p.IH * p.IW;
    const auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)(((xpix - 1) * p.ldx + p.Cin) * 2L), 0x00020000);
    const auto rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)((long)p.Cout * K * 2L), 0x00020000);
    const int lrow = tid / CPR, cv = tid % CPR;
    const int wchunk = ((cv ^ (lrow & (CPR - 1))) << 4);
    int a_iy0[NA], a_ix0[NA], a_base[NA]; bool a_ok[NA];
    const int ohw = p.OH * p.OW;
    const float r_ohw = 1.0f / (float)ohw, r_ow = 1.0f / (float)p.OW;   // (M < 2^24: exact quotients with one fix-up step, no integer division)
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int m = m0 + lrow + LR * j;
      a_ok[j] = m < M;
      const int mm = a_ok[j] ? m : 0;
      const int n_img = p.n_img > 1 ? fast_div(mm, ohw, r_ohw) : 0, rem = mm - n_img * ohw;
      const int oy = fast_div(rem, p.OW, r_ow), ox = rem - oy * p.OW;
      a_iy0[j] = oy * p.stride - p.pad;
      a_ix0[j] = ox * p.stride - p.pad;
      a_base[j] = n_img * p.IH * p.IW;
    }
    unsigned voffB[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int n = n0 + lrow + LR * j;
      voffB[j] = n < p.Cout ? (unsigned)(((long)n * K + cv * 8) * 2L) : OOR;
    }
    const int taps = p.KH * p.KW;
    int it = kt0, c0 = kt0 * BK, tap = 0;
    if (taps > 1) { tap = c0 / p.Cin; c0 -= tap * p.Cin; }
    unsigned voffA[NA];
    auto set_tap = [&](int t) {
      const int ky = t / p.KW, kx = t - ky * p.KW;
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const int iy = a_iy0[j] + ky, ix = a_ix0[j] + kx;
        const bool v = a_ok[j] && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
        voffA[j] = v ? (unsigned)(((long)(a_base[j] + iy * p.IW + ix) * p.ldx + cv * 8) * 2L) : OOR;
      }
    };
    set_tap(tap);
    uint4 ra[D][NA], rb[D][NB];
    auto issue = [&](uint4 (&a)[NA], uint4 (&b)[NB]) {
      const int sa = c0 * 2, sb = it * RB;
#pragma unroll
      for (int j = 0; j < NA; ++j) a[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, voffA[j], sa, 0));
#pragma unroll
      for (int j = 0; j < NB; ++j) b[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rw, voffB[j], sb, 0));
      ++it; c0 += BK;
      if (c0 >= p.Cin && taps > 1) { c0 = 0; ++tap; if (tap < taps) set_tap(tap); }
    };
    auto store_slice = [&](int buf, const uint4 (&a4)[NA], const uint4 (&b4)[NB]) {
      char* a = smem + buf * BUF + lrow * RB + wchunk;
      char* b = a + BM * RB;
#pragma unroll
      for (int j = 0; j < NA; ++j) *(uint4*)(a + LR * j * RB) = a4[j];
#pragma unroll
      for (int j = 0; j < NB; ++j) *(uint4*)(b + LR * j * RB) = b4[j];
    };
#pragma unroll
    for (int s = 0; s < D; ++s)
      if (s < KT) issue(ra[s], rb[s]);
    store_slice(0, ra[0], rb[0]);
    if (D < KT) issue(ra[0], rb[0]);
    __syncthreads();                                          // barrier #0
    int t0 = 0;
    for (; t0 + 2 * D <= KT; t0 += D) {
#pragma unroll
      for (int s = 0; s < D; ++s) {
        const int t = t0 + s;
        const int nxt = (s + 1) % D;
        store_slice((t + 1) & 1, ra[nxt], rb[nxt]);
        issue(ra[nxt], rb[nxt]);
        __syncthreads();                                      // barrier #(t+1)
      }
    }
#pragma unroll
    for (int s = 0; s < 2 * D; ++s) {
      const int t = t0 + s;
      if (t + 1 < KT) {
        const int nxt = (s + 1) % D;
        store_slice((t + 1) & 1, ra[nxt], rb[nxt]);
        if (t + 1 + D < KT) issue(ra[nxt], rb[nxt]);
        __syncthreads();
      }
    }
    return;
  }

  // ---------------- multipliers ----------------
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = wave_all;
  const int wm = wave >> 1, wn = wave & 1;
  const int fr = lane & 15, fg = lane >> 4;
  const bool plain = !(p.flags & (L2S_CONV_DECONV2X2 | L2S_CONV_SCATTER)) && !(p.ldy & 7) && !(p.ldadd & 7) && !(p.ldref & 7) && !(p.Cout & 7) &&
                     !((uintptr_t)p.y & 15) && !((uintptr_t)p.add & 15) && !((uintptr_t)p.ref & 15) && !((uintptr_t)p.bias & 15) &&
                     (long)M * p.ldy * 2 < (1L << 31) && (!p.add || (long)M * p.ldadd * 2 < (1L << 31)) && (!p.ref || (long)M * p.ldref * 2 < (1L << 31));
  const bool staged = plain && gridDim.z == 1;
  // residual / ReLU-mask operands of the LDS-staged epilogue, requested before the K loop (these waves issue no other global load)
  constexpr int EIT = (BM * (BN / 8) + 255) / 256;
  u32x4v eav[EIT], erv[EIT];
  // (behind a K loop of >= 8 slices they go out at once; with fewer slices they would compete with the loaders' first, critical loads -
  // layer3 conv3 7.5 -> 8.5 us - so they follow barrier #0, when the first slice has landed)
  const bool pre = staged;
  if (pre && KT >= 8) igemm_epilogue_prefetch<BM, 256, BN>(p, m0, n0, M, tid, eav, erv);
  f32x4 ebv[BN / 32];                                          // the wave's bias columns (an L2 round trip that used to open the epilogue)
#pragma unroll
  for (int j = 0; j < BN / 32; ++j) {
    const int n = n0 + (wave_all & 1) * (BN / 2) + j * 16 + (lane >> 4) * 4;
    ebv[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (staged && p.bias && n < p.Cout) ebv[j] = *(const f32x4*)(p.bias + n);
  }
  const int swz = fr & (CPR - 1);
  const int offa = (wm * WM + fr) * RB, offb = BM * RB + (wn * WN + fr) * RB;
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  uint4 fa0[KG][TM], fb0[KG][TN], fa1[KG][TM], fb1[KG][TN];
  auto read_all = [&](int t, uint4 (&fa)[KG][TM], uint4 (&fb)[KG][TN]) {
    const char* base = smem + (t & 1) * BUF;
#pragma unroll
    for (int kg = 0; kg < KG; ++kg) {
      const int ch = ((kg * 4 + fg) ^ swz) << 4;
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[kg][i] = *(const uint4*)(base + offa + i * 16 * RB + ch);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[kg][j] = *(const uint4*)(base + offb + j * 16 * RB + ch);
    }
  };
  auto mma_all = [&](const uint4 (&fa)[KG][TM], const uint4 (&fb)[KG][TN]) {
#pragma unroll
    for (int kg = 0; kg < KG; ++kg)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::run(fb[kg][j], fa[kg][i], acc[i][j]);
  };
  __syncthreads();                                            // barrier #0
  unsigned long long t_first = 0, t_loop = 0;
  if constexpr (STAMP) t_first = __builtin_amdgcn_s_memrealtime();
  read_all(0, fa0, fb0);
  if (pre && KT < 8) igemm_epilogue_prefetch<BM, 256, BN>(p, m0, n0, M, tid, eav, erv);
  int t = 1;
  for (; t + 2 <= KT; t += 2) {
    __syncthreads(); read_all(t, fa1, fb1); mma_all(fa0, fb0);
    __syncthreads(); read_all(t + 1, fa0, fb0); mma_all(fa1, fb1);
  }
  if (t < KT) { __syncthreads(); read_all(t, fa1, fb1); mma_all(fa0, fb0); mma_all(fa1, fb1); }
  else mma_all(fa0, fb0);
  if (gridDim.z > 1) { igemm_splitk_slab<TM, TN, WM, WN>(p, acc, m0, n0, wm, wn, fr, fg, M); return; }
  if constexpr (STAMP) {
    asm volatile("s_nop 0" : "+v"(acc[0][0]), "+v"(acc[TM - 1][TN - 1]));         // (the last MFMAs have retired)
    t_loop = __builtin_amdgcn_s_memrealtime();
  }
  if (staged) {
    igemm_epilogue_lds128<TM, TN, WM, WN, 2, 256, BN, 1>(p, acc, m0, n0, wm, wn, fr, fg, M, smem_all, eav, erv, ebv);
    if constexpr (STAMP) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (tid == 0 && p.ws) {
        unsigned long long* o = (unsigned long long*)p.ws + 4L * (blockIdx.x + (long)gridDim.x * blockIdx.z);
        o[0] = t_entry; o[1] = t_first; o[2] = t_loop; o[3] = __builtin_amdgcn_s_memrealtime();
      }
    }
    return;
  }
  igemm_epilogue<T, TM, TN, WM, WN, false>(p, acc, m0, n0, wm, wn, fr, fg, M);
}

// ------------------------------------------------------------------------------------------------
// Software-pipelined large-tile variant (256x128 tile, 8 waves = 2 per SIMD, one workgroup per CU): the ring loader of
// igemm_ring_kernel plus THREE LDS slice buffers, so that slice t+1 is already complete in LDS while slice t is being
// multiplied.  Iteration t (one barrier):
//   phase A: ds_read the second-half fragments (k 32..63) of slice t | ds_write slice t+2 | issue loads of slice t+2+D |
//            16 MFMAs on the first-half fragments (read during the previous iteration)
//   phase B: ds_read the first-half fragments of slice t+1 (visible since this iteration's barrier) | 16 MFMAs on the
//            second-half fragments
// so no MFMA ever waits for an LDS read issued after the barrier, and the LDS fill of slice t+2 hides under the MFMAs.
// ------------------------------------------------------------------------------------------------
template <typename T, int BM, int BN, int WGM, int WGN, int D, bool OUTF32, bool TAPIN>
__global__ __launch_bounds__(64 * WGM * WGN) void igemm_sp_kernel(const l2s_conv_desc p) {
  constexpr int VE = 16 / (int)sizeof(T);
  constexpr int BK = ROWB / (int)sizeof(T);
  constexpr int NTG = 64 * WGM * WGN, LR = NTG / 8;
  constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16;
  constexpr int NA = (BM + LR - 1) / LR, NB = BN / LR;      // BM need not be a multiple of the loader pass (224 = 3.5 passes)
  constexpr int BUF = (BM + BN) * ROWB;
  static_assert(D >= 2, "ring depth");
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int M = p.n_img * p.OH * p.OW;
  const int K = p.KH * p.KW * p.Cin;
  int mt, nt;
  {
    const int MT = (M + BM - 1) / BM, NT = (p.Cout + BN - 1) / BN, G = MT * NT;
    const int L = blockIdx.x, x = L & 7, slot = L >> 3, q = G >> 3, r = G & 7;
    const int t = x * q + min(x, r) + slot;
    if (p.xcd_mode == 0) { mt = t / NT; nt = t - mt * NT; } else { nt = t / MT; mt = t - nt * MT; }
  }
  const int m0 = mt * BM, n0 = nt * BN;
  const long xpix = (long)p.n_img * p.IH * p.IW;
  const auto rx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)(((xpix - 1) * p.ldx + p.Cin) * (long)sizeof(T)), 0x00020000);
  const auto rw = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)((long)p.Cout * K * (long)sizeof(T)), 0x00020000);

  const int lrow = tid >> 3, cv = tid & 7;
  const int wchunk = ((cv ^ (lrow & 7)) << 4);
  int a_iy0[NA], a_ix0[NA], a_base[NA]; bool a_ok[NA];
  const int ohw = p.OH * p.OW;
#pragma unroll
  for (int j = 0; j < NA; ++j) {
    int m = m0 + lrow + LR * j;
    a_ok[j] = m < M && (lrow + LR * j) < BM;
    int mm = a_ok[j] ? m : 0;
    int n_img = mm / ohw, rem = mm - n_img * ohw;
    int oy = rem / p.OW, ox = rem - oy * p.OW;
    a_iy0[j] = oy * p.stride - p.pad;
    a_ix0[j] = ox * p.stride - p.pad;
    a_base[j] = n_img * p.IH * p.IW;
  }
  unsigned voffB[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const int n = n0 + lrow + LR * j;
    voffB[j] = n < p.Cout ? (unsigned)(((long)n * K + cv * VE) * (long)sizeof(T)) : OOR;
  }
  const int KT = K / BK;
  const int taps = p.KH * p.KW;
  int it = 0, c0 = 0, tap = 0;
  unsigned voffA[NA];
  // TAPIN (filters with 2..32 taps): K is walked channel-chunk-major, taps innermost, so the 9 taps of a 3x3 filter re-read the
  // same 128-byte lines of the input within 9 consecutive slices (they hit in L2 / L1) instead of once per full pass over the
  // channels.  Measured on the dominant launch: the fabric-side read traffic was 21x the algorithmic bytes with the tap-major
  // walk (profiles/r01_pmc_traffic_before_tap_inner.json).  Per row: byte offset of tap (0,0) + a bit mask of the in-image taps.
  int vbase[NA]; unsigned tmask[NA];
  auto set_tap = [&](int t) {
    const int ky = t / p.KW, kx = t - ky * p.KW;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int iy = a_iy0[j] + ky, ix = a_ix0[j] + kx;
      const bool v = a_ok[j] && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
      voffA[j] = v ? (unsigned)(((long)(a_base[j] + iy * p.IW + ix) * p.ldx + cv * VE) * (long)sizeof(T)) : OOR;
    }
  };
  if (TAPIN) {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      vbase[j] = ((a_base[j] + a_iy0[j] * p.IW + a_ix0[j]) * p.ldx + cv * VE) * (int)sizeof(T);
      unsigned m = 0;
      for (int t = 0; t < taps; ++t) {
        const int ky = t / p.KW, kx = t - ky * p.KW;
        const int iy = a_iy0[j] + ky, ix = a_ix0[j] + kx;
        if (a_ok[j] && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW) m |= 1u << t;
      }
      tmask[j] = m;
    }
  } else {
    set_tap(0);
  }
  auto issue = [&](uint4 (&a)[NA], uint4 (&b)[NB]) {
    int sa, sb;
    if (TAPIN) {
      const int ky = tap / p.KW, kx = tap - ky * p.KW;
      const int toff = (ky * p.IW + kx) * p.ldx * (int)sizeof(T);
      sa = c0 * (int)sizeof(T); sb = (tap * p.Cin + c0) * (int)sizeof(T);
#pragma unroll
      for (int j = 0; j < NA; ++j) voffA[j] = ((tmask[j] >> tap) & 1u) ? (unsigned)(vbase[j] + toff) : OOR;
    } else {
      sa = c0 * (int)sizeof(T); sb = it * (BK * (int)sizeof(T));
    }
#pragma unroll
    for (int j = 0; j < NA; ++j) a[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rx, voffA[j], sa, 0));
#pragma unroll
    for (int j = 0; j < NB; ++j) b[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rw, voffB[j], sb, 0));
    ++it;
    if (TAPIN) {
      if (++tap == taps) { tap = 0; c0 += BK; }
    } else {
      c0 += BK;
      if (c0 >= p.Cin && taps > 1) { c0 = 0; ++tap; if (tap < taps) set_tap(tap); }
    }
  };
  auto store_slice = [&](int buf, const uint4 (&ra)[NA], const uint4 (&rb)[NB]) {
    char* a = smem + buf * BUF + lrow * ROWB + wchunk;
    char* b = a + BM * ROWB;
#pragma unroll
    for (int j = 0; j < NA; ++j) if ((BM % LR) == 0 || j < NA - 1 || lrow < (BM % LR)) *(uint4*)(a + LR * j * ROWB) = ra[j];
#pragma unroll
    for (int j = 0; j < NB; ++j) *(uint4*)(b + LR * j * ROWB) = rb[j];
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;
  const int swz = fr & 7;
  const int offa = (wm * WM + fr) * ROWB, offb = BM * ROWB + (wn * WN + fr) * ROWB;
  const int ch0 = ((0 * 4 + fg) ^ swz) << 4, ch1 = ((1 * 4 + fg) ^ swz) << 4;
  auto read_frags = [&](int buf, int ch, uint4 (&fa)[TM], uint4 (&fb)[TN]) {
    const char* base = smem + buf * BUF;
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = *(const uint4*)(base + offa + i * 16 * ROWB + ch);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = *(const uint4*)(base + offb + j * 16 * ROWB + ch);
  };
  auto mma = [&](const uint4 (&fa)[TM], const uint4 (&fb)[TN]) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::run(fb[j], fa[i], acc[i][j]);
  };

  // prologue: slices 0..D-1 in flight; slices 0 and 1 to LDS (their sets re-issued with slices D, D+1); first-half fragments of slice 0
  uint4 ra[D][NA], rb[D][NB];
#pragma unroll
  for (int s = 0; s < D; ++s)
    if (s < KT) issue(ra[s], rb[s]);
  store_slice(0, ra[0], rb[0]);
  if (D < KT) issue(ra[0], rb[0]);
  if (1 < KT) {
    store_slice(1, ra[1], rb[1]);
    if (D + 1 < KT) issue(ra[1], rb[1]);
  }
  __syncthreads();
  uint4 fa0[TM], fb0[TN], fa1[TM], fb1[TN];
  read_frags(0, ch0, fa0, fb0);

  int t0 = 0, b0 = 0;                        // b0 = t0 % 3
  for (; t0 + 2 * D + 2 <= KT; t0 += D) {    // steady state: t + 2 + D < KT for every t in the body
#pragma unroll
    for (int s = 0; s < D; ++s) {
      const int set = (s + 2) % D;           // t0 % D == 0
      const int bt = b0, bt1 = b0 == 2 ? 0 : b0 + 1, bt2 = bt1 == 2 ? 0 : bt1 + 1;
      if (s > 0 || t0 > 0) __syncthreads();
      read_frags(bt, ch1, fa1, fb1);
      store_slice(bt2, ra[set], rb[set]);
      issue(ra[set], rb[set]);
      mma(fa0, fb0);
      read_frags(bt1, ch0, fa0, fb0);
      mma(fa1, fb1);
      b0 = bt1;
    }
  }
  // tail
#pragma unroll
  for (int s = 0; s < 2 * D + 2; ++s) {
    const int t = t0 + s;
    if (t < KT) {
      const int set = (s + 2) % D;
      const int bt = b0, bt1 = b0 == 2 ? 0 : b0 + 1, bt2 = bt1 == 2 ? 0 : bt1 + 1;
      if (t > 0) __syncthreads();
      read_frags(bt, ch1, fa1, fb1);
      if (t + 2 < KT) {
        store_slice(bt2, ra[set], rb[set]);
        if (t + 2 + D < KT) issue(ra[set], rb[set]);
      }
      mma(fa0, fb0);
      if (t + 1 < KT) read_frags(bt1, ch0, fa0, fb0);
      mma(fa1, fb1);
      b0 = bt1;
    }
  }
  if constexpr (sizeof(T) == 2 && !OUTF32 && BN == 128) {
    const bool plain = !(p.flags & (L2S_CONV_DECONV2X2 | L2S_CONV_SCATTER)) && !(p.ldy & 7) && !(p.ldadd & 7) && !(p.ldref & 7) && !(p.Cout & 7) &&
                       !((uintptr_t)p.y & 15) && !((uintptr_t)p.add & 15) && !((uintptr_t)p.ref & 15) && !((uintptr_t)p.bias & 15) &&
                       (long)M * p.ldy * 2 < (1L << 31) && (!p.add || (long)M * p.ldadd * 2 < (1L << 31)) && (!p.ref || (long)M * p.ldref * 2 < (1L << 31));
    if (plain) { igemm_epilogue_lds128<TM, TN, WM, WN, WGM, NTG, BN, 1>(p, acc, m0, n0, wm, wn, fr, fg, M, smem); return; }
  }
  igemm_epilogue<T, TM, TN, WM, WN, OUTF32>(p, acc, m0, n0, wm, wn, fr, fg, M);
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA tile (bf16): BM x BN = 256 x 128, eight waves as 4 (rows) x 2 (columns), K in 128-byte slices through a ring of three LDS
// stages that the operands enter by `buffer_load_dwordx4 ... lds` - no register staging, no ds_write, no VGPRs for data in flight.
// A wave instruction of the DMA writes 1 KiB = 8 rows linearly, so the XOR swizzle of the 16-byte chunks (chunk ^ (row & 7)) sits on
// the per-lane SOURCE offset and on the fragment reads; rows outside the image / tile / matrix carry an offset >= 0x80000000, which
// the buffer descriptor's range check turns into zeros written to LDS.
// The two halves of the workgroup (waves 0-3 = group 0, waves 4-7 = group 1: the two waves of every SIMD) run the same K loop ONE
// SLOT APART.  A slot is the time between two workgroup barriers; in every slot one group is in its LOAD segment (the 16 fragment
// reads of slice t, then the six DMA requests of slice t+2) while the other is in its MULTIPLY segment (the slice's 32 MFMAs, with the
// address arithmetic of the next requests between them), so a SIMD's matrix pipe always has one wave feeding it:
//     slot        2t          2t+1         2t+2         2t+3
//     group 0     LOAD(t)     MUL(t)       LOAD(t+1)    MUL(t+1)      waits for ITS share of slice t+1 at the end of MUL(t)
//     group 1     MUL(t-1)    LOAD(t)      MUL(t)       LOAD(t+1)     waits for its share of slice t+1 at the end of LOAD(t)
// Slice s is first read in slot 2s; both groups retire their share of it before the barrier that ends slot 2s-1 (requested 3 resp. 2
// slots earlier, with only the requests of slice s+1 younger: a counted vmcnt).  A stage is refilled only after every wave has passed
// a barrier behind an `s_waitcnt lgkmcnt(0)` that retired its reads of that stage.  The DMA and every wait are inline asm, so hipcc's
// waitcnt pass neither sees nor drains them; the fragment reads are ordinary LDS loads.
// What the in-kernel clock stamps say (tools/dma_stamps.py, algo 4): a wave needs ~22 cycles per ds_read_b128 (the SIMD's 64 B/clk
// return path: 350 cycles for its 16 KiB whatever the other waves do), and a DMA request costs what its address arithmetic costs
// (73 cycles each with the offsets computed in place, which made LOAD 770 cycles against MULTIPLY's 540) - hence the offsets of the
// next slice are prepared between the MFMAs.
// ------------------------------------------------------------------------------------------------
typedef int i32x4s __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma_b128(const i32x4s& rsrc, unsigned voff, unsigned soff, unsigned lds) {
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), "s"(soff), "s"(lds) : "memory");
}
__device__ __forceinline__ void wg_barrier() { asm volatile("s_barrier" ::: "memory"); }
__device__ __forceinline__ void wait_lgkm0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int BM, int BN, bool STAMP = false>
__global__ __launch_bounds__(512) void igemm_dma_kernel(const l2s_conv_desc p) {
  typedef bf16_t T;
  constexpr int WGM = 4, WGN = 2, WM = BM / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16;
  constexpr int PA = BM / 64, PB = BN / 64, NP = PA + PB;    // 1-KiB DMA pieces (8 rows x 128 B) per wave and slice
  constexpr int STG = (BM + BN) * ROWB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int wm = wave >> 1, wn = wave & 1;
  const int M = p.n_img * p.OH * p.OW;
  const int K = p.KH * p.KW * p.Cin;
  int mt, nt;
  {
    const int MT = (M + BM - 1) / BM, NT = (p.Cout + BN - 1) / BN, G = MT * NT;
    const int L = blockIdx.x, x = L & 7, slot = L >> 3, q = G >> 3, r = G & 7;
    const int t = x * q + min(x, r) + slot;
    if (p.xcd_mode == 0) { mt = t / NT; nt = t - mt * NT; } else { nt = t / MT; mt = t - nt * MT; }
  }
  const int m0 = mt * BM, n0 = nt * BN;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const long xpix = (long)p.n_img * p.IH * p.IW;
  i32x4s rx, rw;
  rx.x = (int)(uintptr_t)p.x; rx.y = (int)((uintptr_t)p.x >> 32); rx.z = (int)(((xpix - 1) * p.ldx + p.Cin) * 2L); rx.w = 0x00020000;
  rw.x = (int)(uintptr_t)p.w; rw.y = (int)((uintptr_t)p.w >> 32); rw.z = (int)((long)p.Cout * K * 2L); rw.w = 0x00020000;

  // ---- DMA coordinates: lane -> (row of an 8-row piece, physical chunk); source chunk = physical ^ row ----
  const int prow = lane >> 3, sch = (lane & 7) ^ prow;
  const int ohw = p.OH * p.OW;
  int vbase[PA]; unsigned ntmask[PA];                  // byte offset of tap (0,0); bit (ky KW + kx) SET = that tap is outside the image
#pragma unroll
  for (int j = 0; j < PA; ++j) {
    const int m = m0 + 8 * (wave + 8 * j) + prow;
    const bool ok = m < M;
    const int mm = ok ? m : 0;
    const int n_img = mm / ohw, rem = mm - n_img * ohw;
    const int oy = rem / p.OW, ox = rem - oy * p.OW;
    const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
    vbase[j] = (((n_img * p.IH + iy0) * p.IW + ix0) * p.ldx + sch * 8) * 2;
    unsigned mk = 0;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = iy0 + ky, ix = ix0 + kx;
        if (ky < p.KH && kx < p.KW && ok && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW) mk |= 1u << (ky * p.KW + kx);
      }
    ntmask[j] = ~mk;
  }
  unsigned voffB[PB];
#pragma unroll
  for (int j = 0; j < PB; ++j) {
    const int n = n0 + 8 * (wave + 8 * j) + prow;
    voffB[j] = n < p.Cout ? (unsigned)(((long)n * K + sch * 8) * 2L) : OOR;
  }
  // issue cursor (uniform): K is walked channel-chunk-major with the taps innermost (the nine taps of a 3x3 filter re-read the same
  // input lines within nine consecutive slices).  prep() turns the cursor into the offsets of the next slice to request.
  const int KT = K / 64;
  int c0 = 0, ky = 0, kx = 0, tapi = 0;
  unsigned vo[PA], soA = 0, soB = 0;
  int toff = 0;
  auto prep_s = [&]() {                                // scalar half: offsets of the slice at the cursor, then the cursor moves on (no branches)
    toff = (ky * p.IW + kx) * p.ldx * 2;
    soA = (unsigned)(c0 * 2); soB = (unsigned)((tapi * p.Cin + c0) * 2);
  };
#ifdef L2S_TOOLS
  // tools build: l2s_conv_desc.prio bit 8 = the A operand is requested for tap 0 only (the other taps' requests carry the out-of-range offset: zeros
  // land in LDS, nothing crosses L2) - the operand traffic of a patch tile that stages the input once for the nine taps; results are garbage
  const bool ko_a = (p.prio >> 8) & 1;
  auto prep_v = [&](int j) { vo[j] = ((((ntmask[j] >> tapi) & 1u) | (unsigned)(ko_a && tapi != 0)) << 31) | (unsigned)(vbase[j] + toff); asm volatile("" : "+v"(vo[j])); };
#else
  auto prep_v = [&](int j) { vo[j] = (((ntmask[j] >> tapi) & 1u) << 31) | (unsigned)(vbase[j] + toff); asm volatile("" : "+v"(vo[j])); };   // (the empty asm pins the arithmetic where it is written: hipcc otherwise sinks it behind the MFMAs)
#endif
  auto prep_adv = [&]() {
    const int nkx = kx + 1; const bool wx = nkx == p.KW; kx = wx ? 0 : nkx;
    const int nky = ky + (wx ? 1 : 0); const bool wy = nky == p.KH; ky = wy ? 0 : nky;
    tapi = wy ? 0 : tapi + 1; c0 += wy ? 64 : 0;
  };
  auto prep = [&]() {
    prep_s();
#pragma unroll
    for (int j = 0; j < PA; ++j) prep_v(j);
    prep_adv();
  };
  const unsigned ldsA = lds0 + (unsigned)(wave * 1024), ldsB = ldsA + (unsigned)(BM * ROWB);
  auto request = [&](int stage) {                      // the prepared slice -> LDS stage `stage`
    const unsigned sb = (unsigned)(stage * STG);
#pragma unroll
    for (int j = 0; j < PA; ++j) dma_b128(rx, vo[j], soA, ldsA + sb + (unsigned)(j * 8192));
#pragma unroll
    for (int j = 0; j < PB; ++j) dma_b128(rw, voffB[j], soB, ldsB + sb + (unsigned)(j * 8192));
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fg = lane >> 4;
  const int swz = fr & 7;
  const int offa = (wm * WM + fr) * ROWB, offb = BM * ROWB + (wn * WN + fr) * ROWB;
  uint4 fa[2][TM], fb[2][TN];
  auto read_all = [&](int stage) {
    const char* base = smem + stage * STG;
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
      const int ch = ((kg * 4 + fg) ^ swz) << 4;
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[kg][i] = *(const uint4*)(base + offa + i * 16 * ROWB + ch);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[kg][j] = *(const uint4*)(base + offb + j * 16 * ROWB + ch);
    }
  };
  // the slice's 32 MFMAs with the address arithmetic of the slice after next spread between them (a handful of SALU / VALU
  // instructions per gap: they issue while the matrix pipe works)
  auto mma_all = [&]() {
    int q = 0;
#pragma unroll
    for (int kg = 0; kg < 2; ++kg)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = Mma<T>::run(fb[kg][j], fa[kg][i], acc[i][j]);
          if (q == 2) { __builtin_amdgcn_sched_barrier(0); prep_s(); __builtin_amdgcn_sched_barrier(0); }
          if (q >= 6 && (q - 6) % 4 == 0 && (q - 6) / 4 < PA) { __builtin_amdgcn_sched_barrier(0); prep_v((q - 6) / 4); __builtin_amdgcn_sched_barrier(0); }
          if (q == 6 + 4 * PA) { __builtin_amdgcn_sched_barrier(0); prep_adv(); __builtin_amdgcn_sched_barrier(0); }
          ++q;
        }
  };

  // ---- prologue: slices 0 and 1 requested, slice 2 prepared; everybody waits for its share of slice 0 ----
  prep(); request(0);
  if (1 < KT) { prep(); request(1); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory"); }
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (2 < KT) prep();
  wg_barrier();
  if (grp == 1) wg_barrier();                          // one slot behind group 0
  int st = 0;                                          // t % 3
  // STAMP (tools/dma_stamps.py, algo 4): waves 0 and 4 of workgroup 0 store the shader clock at seven points of their first 24 slices
  // into 8 KiB of LDS behind the ring (requested by the launcher for this build only); copied to p.ws at the end
  unsigned long long* stamps = (unsigned long long*)(smem + 3 * STG) + (grp * 24 * 8);
  const bool stamping = STAMP && blockIdx.x == 0 && (wave & 3) == 0 && lane == 0;
  // (l2s_conv_desc.prio bit 9 in this build: NO stamps inside the loop, only the two clocks before and behind it - the loop as the product
  // runs it, for the clock it holds: MI355X_MICROARCH.md DVFS item 6)
  const bool fine = STAMP && !(p.prio & 0x200);
  unsigned long long loop_c0 = 0, loop_r0 = 0;
  if constexpr (STAMP) { loop_c0 = __builtin_amdgcn_s_memtime(); loop_r0 = __builtin_amdgcn_s_memrealtime(); }
  auto stamp = [&](int t, int i) {
    if constexpr (STAMP) {
      if (fine && t < 24) {
        const unsigned long long c = __builtin_amdgcn_s_memtime(); if (stamping) stamps[t * 8 + i] = c;
        // (the constant 100 MHz clock next to the first stamp of a slice: delta s_memtime / delta s_memrealtime = the core clock the loop runs at)
        if (i == 0) { const unsigned long long r = __builtin_amdgcn_s_memrealtime(); if (stamping) ((unsigned long long*)(smem + 3 * STG))[2 * 24 * 8 + grp * 24 + t] = r; }
      }
    }
  };
  for (int t = 0; t < KT; ++t) {
    const int st2 = st == 0 ? 2 : st - 1;              // (t + 2) % 3: the stage of slice t-1
    stamp(t, 0);
    read_all(st);
    stamp(t, 1);
    if (t + 2 < KT) request(st2);
    stamp(t, 2);
    wait_lgkm0();
    stamp(t, 3);
    if (grp == 1) {
      if (t + 2 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");   // slice t+1 landed (slice t+2 may be in flight)
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    wg_barrier();
    stamp(t, 4);
    __builtin_amdgcn_sched_barrier(0);
    mma_all();                                         // (+ offsets of slice t+3, requested in the next LOAD segment)
    __builtin_amdgcn_sched_barrier(0);
    stamp(t, 5);
    if (grp == 0) {
      if (t + 2 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    stamp(t, 6);
    wg_barrier();
    stamp(t, 7);
    st = st == 2 ? 0 : st + 1;
  }
  if (grp == 0) wg_barrier();                          // group 1's last MULTIPLY slot
  if constexpr (STAMP) {
    const unsigned long long loop_c1 = __builtin_amdgcn_s_memtime(), loop_r1 = __builtin_amdgcn_s_memrealtime();
    if (stamping) {
      unsigned long long* o = (unsigned long long*)(smem + 3 * STG) + 2 * 24 * 8 + 48 + grp * 2;
      o[0] = loop_c1 - loop_c0; o[1] = loop_r1 - loop_r0;
    }
    __syncthreads();
    if (blockIdx.x == 0 && tid < 2 * 24 * 8 + 2 * 24 + 4 && p.ws) ((unsigned long long*)p.ws)[tid] = ((unsigned long long*)(smem + 3 * STG))[tid];
    __syncthreads();
  }
  {
    const bool plain = !(p.flags & (L2S_CONV_DECONV2X2 | L2S_CONV_SCATTER)) && !(p.ldy & 7) && !(p.ldadd & 7) && !(p.ldref & 7) && !(p.Cout & 7) &&
                       !((uintptr_t)p.y & 15) && !((uintptr_t)p.add & 15) && !((uintptr_t)p.ref & 15) && !((uintptr_t)p.bias & 15) &&
                       (long)M * p.ldy * 2 < (1L << 31) && (!p.add || (long)M * p.ldadd * 2 < (1L << 31)) && (!p.ref || (long)M * p.ldref * 2 < (1L << 31));
    if (plain) { igemm_epilogue_lds128<TM, TN, WM, WN, WGM, 512, BN, 1>(p, acc, m0, n0, wm, wn, fr, fg, M, smem); return; }
  }
  igemm_epilogue<T, TM, TN, WM, WN, false>(p, acc, m0, n0, wm, wn, fr, fg, M);
}

// ------------------------------------------------------------------------------------------------
// The same LDS-DMA pipeline on tiles of 196 ROWS (round 5).  The dominant launch - layer4's 3x3 on the 256 RoIs, M = 12544 = 49 x 256,
// N = 512 - is 196 tiles of 256 x 128 on 256 CUs: inside its busy CUs the 256-row kernel runs within ~15 % of the matrix pipe's rate at
// the clock the chip holds, and a quarter of the chip idles.  M = 64 x 196, so a 196-row tile gives 64 x 4 = EXACTLY 256
// workgroups whose tiles are 13 row fragments (208 rows, 6 % padding) instead of 16.  13 does not divide by the four wave rows: wave row 0
// keeps four fragments (rows 0-63) and wave rows 1-3 take three (rows 64-111, 112-159, 160-207), i.e. the two groups' MULTIPLY slots are
// 32 and 24 MFMAs long - 56 per slice instead of 64 - while the LDS geometry, the requests (the pieces of rows >= 208 carry the
// out-of-range offset), the K walk and every wait stay those of igemm_dma_kernel<256,128>.  The two fragment counts are two
// instantiations of the body, chosen per wave (the barriers of a workgroup count arrivals, not program locations).
// Rows 196-207 of a tile belong to the next tile: they are computed (their operands are valid rows) and not stored.
// ------------------------------------------------------------------------------------------------
constexpr int T196 = 196, R208 = 208;
template <int TMW>
__device__ __forceinline__ void igemm_dma196_body(const l2s_conv_desc& p, char* smem, int wave, int lane) {
  typedef bf16_t T;
  constexpr int BM = 256, BN = 128, TN = 4;
  constexpr int PA = BM / 64, PB = BN / 64, NP = PA + PB;
  constexpr int STG = (BM + BN) * ROWB;
  const int tid = threadIdx.x;
  const int grp = wave >> 2;
  const int wm = wave >> 1, wn = wave & 1;
  const int rbase = wm == 0 ? 0 : 16 + 48 * wm;             // 0, 64, 112, 160
  const int M = p.n_img * p.OH * p.OW;
  const int K = p.KH * p.KW * p.Cin;
  int mt, nt;
  {
    const int MT = (M + T196 - 1) / T196, NT = (p.Cout + BN - 1) / BN, G = MT * NT;
    const int L = blockIdx.x, x = L & 7, slot = L >> 3, q = G >> 3, r = G & 7;
    const int t = x * q + min(x, r) + slot;
    if (p.xcd_mode == 0) { mt = t / NT; nt = t - mt * NT; } else { nt = t / MT; mt = t - nt * MT; }
  }
  const int m0 = mt * T196, n0 = nt * BN;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  const long xpix = (long)p.n_img * p.IH * p.IW;
  i32x4s rx, rw;
  rx.x = (int)(uintptr_t)p.x; rx.y = (int)((uintptr_t)p.x >> 32); rx.z = (int)(((xpix - 1) * p.ldx + p.Cin) * 2L); rx.w = 0x00020000;
  rw.x = (int)(uintptr_t)p.w; rw.y = (int)((uintptr_t)p.w >> 32); rw.z = (int)((long)p.Cout * K * 2L); rw.w = 0x00020000;
  const int prow = lane >> 3, sch = (lane & 7) ^ prow;
  const int ohw = p.OH * p.OW;
  int vbase[PA]; unsigned ntmask[PA];
#pragma unroll
  for (int j = 0; j < PA; ++j) {
    const int lr = 8 * (wave + 8 * j) + prow;                // row of the tile
    const int m = m0 + lr;
    const bool ok = m < M && lr < R208;
    const int mm = ok ? m : 0;
    const int n_img = mm / ohw, rem = mm - n_img * ohw;
    const int oy = rem / p.OW, ox = rem - oy * p.OW;
    const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
    vbase[j] = (((n_img * p.IH + iy0) * p.IW + ix0) * p.ldx + sch * 8) * 2;
    unsigned mk = 0;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int iy = iy0 + ky, ix = ix0 + kx;
        if (ky < p.KH && kx < p.KW && ok && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW) mk |= 1u << (ky * p.KW + kx);
      }
    ntmask[j] = ~mk;
  }
  unsigned voffB[PB];
#pragma unroll
  for (int j = 0; j < PB; ++j) {
    const int n = n0 + 8 * (wave + 8 * j) + prow;
    voffB[j] = n < p.Cout ? (unsigned)(((long)n * K + sch * 8) * 2L) : OOR;
  }
  const int KT = K / 64;
  int c0 = 0, ky = 0, kx = 0, tapi = 0;
  unsigned vo[PA], soA = 0, soB = 0;
  int toff = 0;
  auto prep_s = [&]() {
    toff = (ky * p.IW + kx) * p.ldx * 2;
    soA = (unsigned)(c0 * 2); soB = (unsigned)((tapi * p.Cin + c0) * 2);
  };
  auto prep_v = [&](int j) { vo[j] = (((ntmask[j] >> tapi) & 1u) << 31) | (unsigned)(vbase[j] + toff); asm volatile("" : "+v"(vo[j])); };
  auto prep_adv = [&]() {
    const int nkx = kx + 1; const bool wx = nkx == p.KW; kx = wx ? 0 : nkx;
    const int nky = ky + (wx ? 1 : 0); const bool wy = nky == p.KH; ky = wy ? 0 : nky;
    tapi = wy ? 0 : tapi + 1; c0 += wy ? 64 : 0;
  };
  auto prep = [&]() {
    prep_s();
#pragma unroll
    for (int j = 0; j < PA; ++j) prep_v(j);
    prep_adv();
  };
  const unsigned ldsA = lds0 + (unsigned)(wave * 1024), ldsB = ldsA + (unsigned)(BM * ROWB);
  auto request = [&](int stage) {
    const unsigned sb = (unsigned)(stage * STG);
#pragma unroll
    for (int j = 0; j < PA; ++j) dma_b128(rx, vo[j], soA, ldsA + sb + (unsigned)(j * 8192));
#pragma unroll
    for (int j = 0; j < PB; ++j) dma_b128(rw, voffB[j], soB, ldsB + sb + (unsigned)(j * 8192));
  };
  f32x4 acc[TMW][TN];
#pragma unroll
  for (int i = 0; i < TMW; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fg = lane >> 4;
  const int swz = fr & 7;
  const int offa = (rbase + fr) * ROWB, offb = BM * ROWB + (wn * 64 + fr) * ROWB;
  uint4 fa[2][TMW], fb[2][TN];
  auto read_all = [&](int stage) {
    const char* base = smem + stage * STG;
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
      const int ch = ((kg * 4 + fg) ^ swz) << 4;
#pragma unroll
      for (int i = 0; i < TMW; ++i) fa[kg][i] = *(const uint4*)(base + offa + i * 16 * ROWB + ch);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[kg][j] = *(const uint4*)(base + offb + j * 16 * ROWB + ch);
    }
  };
  auto mma_all = [&]() {
    int q = 0;
#pragma unroll
    for (int kg = 0; kg < 2; ++kg)
#pragma unroll
      for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = Mma<T>::run(fb[kg][j], fa[kg][i], acc[i][j]);
          if (q == 2) { __builtin_amdgcn_sched_barrier(0); prep_s(); __builtin_amdgcn_sched_barrier(0); }
          if (q >= 5 && (q - 5) % 4 == 0 && (q - 5) / 4 < PA) { __builtin_amdgcn_sched_barrier(0); prep_v((q - 5) / 4); __builtin_amdgcn_sched_barrier(0); }
          if (q == 5 + 4 * PA) { __builtin_amdgcn_sched_barrier(0); prep_adv(); __builtin_amdgcn_sched_barrier(0); }
          ++q;
        }
  };
  prep(); request(0);
  if (1 < KT) { prep(); request(1); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory"); }
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (2 < KT) prep();
  wg_barrier();
  if (grp == 1) wg_barrier();
  int st = 0;
  for (int t = 0; t < KT; ++t) {
    const int st2 = st == 0 ? 2 : st - 1;
    read_all(st);
    if (t + 2 < KT) request(st2);
    wait_lgkm0();
    if (grp == 1) {
      if (t + 2 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    wg_barrier();
    __builtin_amdgcn_sched_barrier(0);
    mma_all();
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 0) {
      if (t + 2 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    wg_barrier();
    st = st == 2 ? 0 : st + 1;
  }
  if (grp == 0) wg_barrier();
  // ---- epilogue: the fp32 tile (+ bias) through LDS (row pitch BN + 4 floats), then bias / residual / ReLU / mask in fp32, one rounding,
  // 16-byte stores of whole rows (the arithmetic of igemm_epilogue_lds128); rows >= 196 of the tile are the next tile's ----
  {
    constexpr int LDW = BN + 4, CPR = BN / 8, CH = R208 * CPR, NTH = 512, IT = (CH + NTH - 1) / NTH;
    constexpr unsigned NOPE = 0x80000000u;
    float* stg = (float*)smem;
    const auto ry = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, 0x7FFFFFFF, 0x00020000);
    const auto radd = __builtin_amdgcn_make_buffer_rsrc((void*)(p.add ? p.add : p.y), 0, 0x7FFFFFFF, 0x00020000);
    const auto rref = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ref ? p.ref : p.y), 0, 0x7FFFFFFF, 0x00020000);
    f32x4 bv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * 64 + j * 16 + fg * 4;
      bv[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (p.bias && n < p.Cout) bv[j] = *(const f32x4*)(p.bias + n);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        f32x4 v = acc[i][j];
        v[0] += bv[j][0]; v[1] += bv[j][1]; v[2] += bv[j][2]; v[3] += bv[j][3];
        *(f32x4*)(stg + (rbase + i * 16 + fr) * LDW + wn * 64 + j * 16 + fg * 4) = v;
      }
    __syncthreads();
    u32x4v av[IT], rv[IT];
    unsigned off[IT];
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      const int c = tid + NTH * k, row = c / CPR, col = (c % CPR) * 8;
      const int m = m0 + row, n = n0 + col;
      const bool ok = c < CH && row < T196 && m < M && n < p.Cout;
      off[k] = ok ? (unsigned)m : NOPE;
      if (p.add) av[k] = __builtin_amdgcn_raw_buffer_load_b128(radd, ok ? (unsigned)((m * p.ldadd + n) * 2) : NOPE, 0, 0);
      if (p.ref) rv[k] = __builtin_amdgcn_raw_buffer_load_b128(rref, ok ? (unsigned)((m * p.ldref + n) * 2) : NOPE, 0, 0);
    }
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      const int c = min(tid + NTH * k, CH - 1), row = c / CPR, col = (c % CPR) * 8;
      const f32x4 lo = *(const f32x4*)(stg + row * LDW + col), hi = *(const f32x4*)(stg + row * LDW + col + 4);
      float v[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      if (p.add) {
        const unsigned w[4] = {av[k].x, av[k].y, av[k].z, av[k].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] += __uint_as_float(w[e] << 16); v[2 * e + 1] += __uint_as_float(w[e] & 0xFFFF0000u); }
      }
      if (p.flags & L2S_CONV_RELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if (p.ref) {
        const unsigned w[4] = {rv[k].x, rv[k].y, rv[k].z, rv[k].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (!(__uint_as_float(w[e] << 16) > 0.f)) v[2 * e] = 0.f;
          if (!(__uint_as_float(w[e] & 0xFFFF0000u) > 0.f)) v[2 * e + 1] = 0.f;
        }
      }
      u32x4v pk;
      pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16); pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
      pk.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16); pk.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
      const int n = n0 + col;
      __builtin_amdgcn_raw_buffer_store_b128(pk, ry, off[k] != NOPE ? (unsigned)((off[k] * p.ldy + n) * 2) : NOPE, 0, 0);
    }
  }
}
__global__ __launch_bounds__(512) void igemm_dma196_kernel(const l2s_conv_desc p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if ((wave >> 1) == 0) igemm_dma196_body<4>(p, smem, wave, lane);
  else igemm_dma196_body<3>(p, smem, wave, lane);
}

// ---- per-WAVE epilogue of a 64 x 64 sub-tile (bf16 out, plain row-major output): the arithmetic of igemm_epilogue_fast (bias, residual,
// ReLU, mask in fp32, one rounding; operands requested up front in the MFMA layout), but the results leave through 2 KiB of LDS that
// belong to this wave alone: 16 rows x 64 channels at a time are written in the MFMA layout (8 bytes per lane) and read back row-major
// (16 bytes per lane), so the stores cover whole 128-byte runs instead of 32-byte ones (the persistent tile with direct stores:
// 80 us for the 100 MB of output + residual of the N = 2048 launches).  No barrier: LDS operations of one wave are ordered.
// 16-byte chunks are XOR-swizzled with (row >> 1) & 7: writes and reads of a half-wave cover all 64 banks. ----
// One operand array serves the residual OR the mask (a launch with both takes the direct epilogue): 32 registers live across the
// tile's last MUL slot instead of 64 (with both arrays the kernel spilled)
template <int TM, int TN, int WM, int WN>
__device__ __forceinline__ void igemm_epilogue_wave_prefetch(const l2s_conv_desc& p, int m0, int n0, int wm, int wn, int lane, int M, f32x4 (&bv)[TN], u32x2 (&ov)[TM][TN]) {
  constexpr unsigned NOPE = 0x80000000u;
  const int fr = lane & 15, fg = lane >> 4;
  const void* op = p.add ? p.add : p.ref;
  const int ldo = p.add ? p.ldadd : p.ldref;
  const auto rop = __builtin_amdgcn_make_buffer_rsrc((void*)(op ? op : p.y), 0, 0x7FFFFFFF, 0x00020000);
  const auto rbias = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias ? (const void*)p.bias : (const void*)p.y), 0, 0x7FFFFFFF, 0x00020000);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * WN + j * 16 + fg * 4;
    bv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rbias, (p.bias && n < p.Cout) ? (unsigned)(n * 4) : NOPE, 0, 0));
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + wm * WM + i * 16 + fr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * WN + j * 16 + fg * 4;
      ov[i][j] = __builtin_amdgcn_raw_buffer_load_b64(rop, (op && m < M && n < p.Cout) ? (unsigned)(((long)m * ldo + n) * 2) : NOPE, 0, 0);
    }
  }
}
template <int TM, int TN, int WM, int WN>
__device__ __forceinline__ void igemm_epilogue_wave(const l2s_conv_desc& p, f32x4 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int lane, int M, char* wlds,
                                                    const f32x4 (&bv)[TN], const u32x2 (&ov)[TM][TN]) {
  static_assert(TN == 4 && WN == 64, "64-channel sub-tiles");
  constexpr unsigned NOPE = 0x80000000u;
  const int fr = lane & 15, fg = lane >> 4;
  const auto ry = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, 0x7FFFFFFF, 0x00020000);
  // staging addresses: write (row fr, 8 bytes of chunk 2 j + (fg >> 1)); read (row lane >> 3 (+ 8), chunk lane & 7)
  const int wsw = (fr >> 1) & 7;
  const int rrow = lane >> 3, rch = lane & 7;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      float v[4] = {acc[i][j][0] + bv[j][0], acc[i][j][1] + bv[j][1], acc[i][j][2] + bv[j][2], acc[i][j][3] + bv[j][3]};   // (no bias: zeros were loaded)
      const float o[4] = {__uint_as_float(ov[i][j].x << 16), __uint_as_float(ov[i][j].x & 0xFFFF0000u), __uint_as_float(ov[i][j].y << 16), __uint_as_float(ov[i][j].y & 0xFFFF0000u)};
      if (p.add) { v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3]; }
      if (p.flags & L2S_CONV_RELU) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
      if (p.ref) {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (!(o[e] > 0.f)) v[e] = 0.f;
      }
      u32x2 pk;
      pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
      pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
      *(u32x2*)(wlds + fr * 128 + (((2 * j + (fg >> 1)) ^ wsw) << 4) + ((fg & 1) << 3)) = pk;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int row = rrow + 8 * h;
      const u32x4v q = *(const u32x4v*)(wlds + row * 128 + ((rch ^ ((row >> 1) & 7)) << 4));
      const int m = m0 + wm * WM + i * 16 + row, n = n0 + wn * WN + rch * 8;
      const unsigned o = (m < M && n < p.Cout) ? (unsigned)(((long)m * p.ldy + n) * 2) : NOPE;
      __builtin_amdgcn_raw_buffer_store_b128(q, ry, o, 0, 0);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// 256 x 256 LDS-DMA tile for the wide plain GEMMs (round 4: layer4 on the RoIs, conv3 / downsample forward and the conv1 / downsample
// data gradients - M = 12544, N = 1024 / 2048, K = 512 ... 2048).  Same pipeline as igemm_dma_kernel (two wave groups one slot apart,
// three LDS stages, counted vmcnt), different arithmetic per slot: the 256 x 128 tile multiplies 32 MFMAs per wave and slot against
// 16 fragment reads + 6 DMA requests, and its LOAD slot (1100-1300 cycles by the stamps of tools/dma_stamps.py) is twice its MULTIPLY
// slot - 37 % of a CU's matrix pipe.  Here a wave owns 64 x 128 of the tile and a slice is 32 channels (64-byte rows, so that three
// stages of 512 rows still fit): the same 32 MFMAs per slot against 12 fragment reads + 4 requests, and no offset arithmetic at all
// (1x1 / stride 1: the slice's byte offset is a scalar).  1 KiB requests = 16 rows x 64 B; a row's four 16-byte chunks are
// XOR-swizzled with (row >> 2) & 3: the 16 rows a quarter-wave reads then cover all 64 banks.
// Epilogue per wave through 4 KiB of LDS of its own (16 rows x 128 channels at a time, 16-byte chunks XOR row): bias, residual, ReLU /
// mask in fp32, one rounding, 16-byte stores of whole 256-byte runs.
// ------------------------------------------------------------------------------------------------
template <int TM, int TN, bool BOTH>
__device__ __forceinline__ void igemm_epilogue_wave128(const l2s_conv_desc& p, f32x4 (&acc)[TM][TN], int m0, int n0, int wm, int wn, int lane, int M, char* wlds) {
  static_assert(TN == 8, "128-channel sub-tiles");
  constexpr unsigned NOPE = 0x80000000u;
  constexpr int WM = TM * 16, WN = TN * 16;
  const int fr = lane & 15, fg = lane >> 4;
  const void* op = p.add ? p.add : p.ref;
  const int ldo = p.add ? p.ldadd : p.ldref;
  const auto rop = __builtin_amdgcn_make_buffer_rsrc((void*)(op ? op : p.y), 0, 0x7FFFFFFF, 0x00020000);
  const auto rbias = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias ? (const void*)p.bias : (const void*)p.y), 0, 0x7FFFFFFF, 0x00020000);
  const auto ry = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, 0x7FFFFFFF, 0x00020000);
  f32x4 bv[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * WN + j * 16 + fg * 4;
    bv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rbias, (p.bias && n < p.Cout) ? (unsigned)(n * 4) : NOPE, 0, 0));
  }
  // BOTH: residual AND mask (the data gradient of a block's first 1x1: dx = mask(conv^T(dz) + g)): `op` is the residual, `rv` the mask operand
  const auto rref = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ref ? p.ref : p.y), 0, 0x7FFFFFFF, 0x00020000);
  u32x2 ov[2][TN], rv[BOTH ? 2 : 1][BOTH ? TN : 1];       // operands of row block i, requested one block ahead
  auto fetch = [&](int i, u32x2 (&o)[TN]) {
    const int m = m0 + wm * WM + i * 16 + fr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * WN + j * 16 + fg * 4;
      o[j] = __builtin_amdgcn_raw_buffer_load_b64(rop, (op && m < M && n < p.Cout) ? (unsigned)(((long)m * ldo + n) * 2) : NOPE, 0, 0);
      if constexpr (BOTH) rv[i & 1][j] = __builtin_amdgcn_raw_buffer_load_b64(rref, (m < M && n < p.Cout) ? (unsigned)(((long)m * p.ldref + n) * 2) : NOPE, 0, 0);
    }
  };
  fetch(0, ov[0]);
  const int rrow = lane >> 4, rch = lane & 15;             // read side: 4 rows x 16 chunks per instruction
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    if (i + 1 < TM) fetch(i + 1, ov[(i + 1) & 1]);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const u32x2 oo = ov[i & 1][j];
      float v[4] = {acc[i][j][0] + bv[j][0], acc[i][j][1] + bv[j][1], acc[i][j][2] + bv[j][2], acc[i][j][3] + bv[j][3]};   // (no bias: zeros were loaded)
      const float o[4] = {__uint_as_float(oo.x << 16), __uint_as_float(oo.x & 0xFFFF0000u), __uint_as_float(oo.y << 16), __uint_as_float(oo.y & 0xFFFF0000u)};
      if (p.add) { v[0] += o[0]; v[1] += o[1]; v[2] += o[2]; v[3] += o[3]; }
      if (p.flags & L2S_CONV_RELU) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
      if (p.ref) {
        float q[4] = {o[0], o[1], o[2], o[3]};
        if constexpr (BOTH) {
          const u32x2 rr = rv[i & 1][j];
          q[0] = __uint_as_float(rr.x << 16); q[1] = __uint_as_float(rr.x & 0xFFFF0000u); q[2] = __uint_as_float(rr.y << 16); q[3] = __uint_as_float(rr.y & 0xFFFF0000u);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) if (!(q[e] > 0.f)) v[e] = 0.f;
      }
      u32x2 pk;
      pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
      pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
      *(u32x2*)(wlds + fr * 256 + (((2 * j + (fg >> 1)) ^ fr) << 4) + ((fg & 1) << 3)) = pk;
    }
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int row = rrow + 4 * h;
      const u32x4v q = *(const u32x4v*)(wlds + row * 256 + ((rch ^ row) << 4));
      const int m = m0 + wm * WM + i * 16 + row, n = n0 + wn * WN + rch * 8;
      const unsigned o = (m < M && n < p.Cout) ? (unsigned)(((long)m * p.ldy + n) * 2) : NOPE;
      __builtin_amdgcn_raw_buffer_store_b128(q, ry, o, 0, 0);
    }
  }
}

__global__ __launch_bounds__(512) void igemm_dma256_kernel(const l2s_conv_desc p) {
  typedef bf16_t T;
  constexpr int BM = 256, BN = 256, WM = 64, WN = 128, TM = 4, TN = 8;
  constexpr int RB = 64;                                   // bytes of K per LDS row and slice (32 channels)
  constexpr int STG = (BM + BN) * RB;                      // 32 KiB per stage
  constexpr int NP = 4;                                    // 1-KiB requests per wave and slice: two of A, two of B
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int wm = wave >> 1, wn = wave & 1;
  const int M = p.n_img * p.OH * p.OW;
  const int K = p.Cin;                                     // 1x1
  int mt, nt;
  {
    const int MT = (M + BM - 1) / BM, NT = (p.Cout + BN - 1) / BN, G = MT * NT;
    const int L = blockIdx.x, x = L & 7, slot = L >> 3, q = G >> 3, r = G & 7;
    const int t = x * q + min(x, r) + slot;
    if (p.xcd_mode == 0) { mt = t / NT; nt = t - mt * NT; } else { nt = t / MT; mt = t - nt * MT; }
  }
  const int m0 = mt * BM, n0 = nt * BN;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  i32x4s rx, rw;
  rx.x = (int)(uintptr_t)p.x; rx.y = (int)((uintptr_t)p.x >> 32); rx.z = (int)(((long)(M - 1) * p.ldx + p.Cin) * 2L); rx.w = 0x00020000;
  rw.x = (int)(uintptr_t)p.w; rw.y = (int)((uintptr_t)p.w >> 32); rw.z = (int)((long)p.Cout * K * 2L); rw.w = 0x00020000;
  // DMA coordinates: lane -> (row of a 16-row piece, physical chunk); source chunk = physical ^ ((row >> 2) & 3)
  const int prow = lane >> 2, sch = (lane & 3) ^ ((prow >> 2) & 3);
  unsigned voA[2], voB[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = m0 + 16 * (wave + 8 * j) + prow, n = n0 + 16 * (wave + 8 * j) + prow;
    voA[j] = m < M ? (unsigned)(((long)m * p.ldx + sch * 8) * 2L) : OOR;
    voB[j] = n < p.Cout ? (unsigned)(((long)n * K + sch * 8) * 2L) : OOR;
  }
  const int KT = K / 32;
  const unsigned ldsA = lds0 + (unsigned)(wave * 1024), ldsB = ldsA + (unsigned)(BM * RB);
  auto request = [&](int stage, int t) {                   // slice t -> LDS stage `stage`
    const unsigned sb = (unsigned)(stage * STG), so = (unsigned)(t * RB);
#pragma unroll
    for (int j = 0; j < 2; ++j) dma_b128(rx, voA[j], so, ldsA + sb + (unsigned)(j * 8192));
#pragma unroll
    for (int j = 0; j < 2; ++j) dma_b128(rw, voB[j], so, ldsB + sb + (unsigned)(j * 8192));
  };
  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fg = lane >> 4;
  const int ch = (fg ^ ((fr >> 2) & 3)) << 4;
  const int offa = (wm * WM + fr) * RB + ch, offb = BM * RB + (wn * WN + fr) * RB + ch;
  uint4 fa[TM], fb[TN];
  auto read_all = [&](int stage) {
    const char* base = smem + stage * STG;
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[i] = *(const uint4*)(base + offa + i * 16 * RB);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[j] = *(const uint4*)(base + offb + j * 16 * RB);
  };
  auto mma_all = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::run(fb[j], fa[i], acc[i][j]);
  };
  // prologue: slices 0 and 1 requested; everybody waits for its share of slice 0
  request(0, 0);
  if (1 < KT) { request(1, 1); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory"); }
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  wg_barrier();
  if (grp == 1) wg_barrier();                              // one slot behind group 0
  int st = 0;
  for (int t = 0; t < KT; ++t) {
    const int st2 = st == 0 ? 2 : st - 1;                  // (t + 2) % 3: the stage of slice t - 1
    read_all(st);
    if (t + 2 < KT) request(st2, t + 2);
    wait_lgkm0();
    if (grp == 1) {
      if (t + 2 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");   // slice t + 1 landed (slice t + 2 may be in flight)
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    wg_barrier();
    __builtin_amdgcn_sched_barrier(0);
    mma_all();
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 0) {
      if (t + 2 < KT) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    wg_barrier();
    st = st == 2 ? 0 : st + 1;
  }
  if (grp == 0) wg_barrier();                              // group 1's last MULTIPLY slot
  if (p.add && p.ref) igemm_epilogue_wave128<TM, TN, true>(p, acc, m0, n0, wm, wn, lane, M, smem + 3 * STG + wave * 4096);
  else igemm_epilogue_wave128<TM, TN, false>(p, acc, m0, n0, wm, wn, lane, M, smem + 3 * STG + wave * 4096);
}

// ------------------------------------------------------------------------------------------------
// Persistent form of the LDS-DMA tile for the multi-round 1x1 launches (bf16, plain GEMM: layer4 on the RoIs, conv3 / downsample forward
// and the conv1 / downsample data gradients: M = 12544, N = 1024..2048, K = 512..1024 - 784 tiles of 8 or 16 slices).  As separate
// workgroups every tile pays its own prologue (~2 us until the first slice has crossed L2 -> LDS) and its own epilogue with nothing
// else of the workgroup in flight: 65 us for 26 GFLOP.  Here ONE workgroup per CU walks its tiles and the slice stream never stops: the
// requests run NST - 1 slices ahead ACROSS tile boundaries, so the first slices of tile i + 1 are on their way while tile i multiplies
// and while each wave writes its 64 x 64 sub-tile out (per-wave epilogue: bias, residual, ReLU / mask, no barrier, no LDS).
// A wave drains its own vector-memory counter once after its epilogue: stores and loads complete out of order with respect to each
// other, so the counted waits of the pipeline are only valid while nothing but DMA requests is outstanding.
// ------------------------------------------------------------------------------------------------
template <int BM, int BN>
__global__ __launch_bounds__(512) void igemm_pdma_kernel(const l2s_conv_desc p) {
  typedef bf16_t T;
  constexpr int WGM = 4, WGN = 2, WM = BM / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16;
  constexpr int PA = BM / 64, PB = BN / 64, NP = PA + PB;
  constexpr int STG = (BM + BN) * ROWB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int wm = wave >> 1, wn = wave & 1;
  const int M = p.n_img * p.OH * p.OW;
  const int K = p.Cin;
  const int KT = K / 64;
  const int MT = (M + BM - 1) / BM, NT = (p.Cout + BN - 1) / BN, G = MT * NT;
  // this workgroup's tiles: L = blockIdx.x, + gridDim.x, ... in the XCD-aware order of the one-shot kernel (gridDim.x is a multiple of 8)
  const int ntl = (G - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
  auto tile_of = [&](int i, int& m0_, int& n0_) {
    const int Lx = blockIdx.x + i * gridDim.x, x = Lx & 7, slot = Lx >> 3, q = G >> 3, r = G & 7;
    const int t = x * q + min(x, r) + slot;
    int mt, nt;
    if (p.xcd_mode == 0) { mt = t / NT; nt = t - mt * NT; } else { nt = t / MT; mt = t - nt * MT; }
    m0_ = mt * BM; n0_ = nt * BN;
  };
  if (ntl <= 0) return;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  i32x4s rx, rw;
  rx.x = (int)(uintptr_t)p.x; rx.y = (int)((uintptr_t)p.x >> 32); rx.z = (int)((((long)M - 1) * p.ldx + p.Cin) * 2L); rx.w = 0x00020000;
  rw.x = (int)(uintptr_t)p.w; rw.y = (int)((uintptr_t)p.w >> 32); rw.z = (int)((long)p.Cout * K * 2L); rw.w = 0x00020000;
  const int prow = lane >> 3, sch = (lane & 7) ^ prow;

  // ---- request cursor: (tile index rq_i, slice rq_k); bases of the cursor's tile ----
  int rq_i = 0, rq_k = 0;
  unsigned vbA[PA], vbB[PB];
  auto set_req_tile = [&](int i) {
    int m0_, n0_;
    tile_of(i < ntl ? i : ntl - 1, m0_, n0_);
#pragma unroll
    for (int j = 0; j < PA; ++j) {
      const int m = m0_ + 8 * (wave + 8 * j) + prow;
      vbA[j] = m < M ? (unsigned)(((long)m * p.ldx + sch * 8) * 2L) : OOR;
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      const int n = n0_ + 8 * (wave + 8 * j) + prow;
      vbB[j] = n < p.Cout ? (unsigned)(((long)n * K + sch * 8) * 2L) : OOR;
    }
  };
  set_req_tile(0);
  unsigned voA[PA], voB[PB], so = 0;
  auto prep = [&]() {                                  // offsets of the slice at the cursor; the cursor moves on (into the next tile behind the last slice)
#pragma unroll
    for (int j = 0; j < PA; ++j) { voA[j] = vbA[j]; asm volatile("" : "+v"(voA[j])); }
#pragma unroll
    for (int j = 0; j < PB; ++j) { voB[j] = vbB[j]; asm volatile("" : "+v"(voB[j])); }
    so = (unsigned)(rq_k * 128);
    if (++rq_k == KT) { rq_k = 0; ++rq_i; set_req_tile(rq_i); }
  };
  const unsigned ldsA = lds0 + (unsigned)(wave * 1024), ldsB = ldsA + (unsigned)(BM * ROWB);
  auto request = [&](int stage) {
    const unsigned sb = (unsigned)(stage * STG);
#pragma unroll
    for (int j = 0; j < PA; ++j) dma_b128(rx, voA[j], so, ldsA + sb + (unsigned)(j * 8192));
#pragma unroll
    for (int j = 0; j < PB; ++j) dma_b128(rw, voB[j], so, ldsB + sb + (unsigned)(j * 8192));
  };

  f32x4 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int fr = lane & 15, fg = lane >> 4;
  const int swz = fr & 7;
  const int offa = (wm * WM + fr) * ROWB, offb = BM * ROWB + (wn * WN + fr) * ROWB;
  uint4 fa[2][TM], fb[2][TN];
  auto read_all = [&](int stage) {
    const char* base = smem + stage * STG;
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
      const int ch = ((kg * 4 + fg) ^ swz) << 4;
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[kg][i] = *(const uint4*)(base + offa + i * 16 * ROWB + ch);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[kg][j] = *(const uint4*)(base + offb + j * 16 * ROWB + ch);
    }
  };
  auto mma_all = [&](bool do_prep) {
    int q = 0;
#pragma unroll
    for (int kg = 0; kg < 2; ++kg)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = Mma<T>::run(fb[kg][j], fa[kg][i], acc[i][j]);
          if (q == 4) { __builtin_amdgcn_sched_barrier(0); if (do_prep) prep(); __builtin_amdgcn_sched_barrier(0); }
          ++q;
        }
  };

  const bool wide = !(p.ldy & 7) && !(p.Cout & 7) && !((uintptr_t)p.y & 15) && !(p.add && p.ref);     // whole 16-byte pieces of output rows; one epilogue operand
  const int SL = ntl * KT;                              // this workgroup's slices
  // ---- prologue ----
  prep(); request(0);
  if (1 < SL) { prep(); request(1); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory"); }
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (2 < SL) prep();
  wg_barrier();
  if (grp == 1) wg_barrier();
  int st = 0, ti = 0, tk = 0;                          // stage of the slice, tile index and slice within the tile
  int m0, n0;
  tile_of(0, m0, n0);
  for (int g = 0; g < SL; ++g) {
    const int st2 = st == 0 ? 2 : st - 1;
    read_all(st);
    if (g + 2 < SL) request(st2);
    // the epilogue's bias / residual / mask operands of this tile: requested behind the last slice's reads, a whole MUL slot before their use.
    // (They are younger than the wave's requests of slices g + 1 and g + 2: the counted wait below becomes "everything but the operands".)
    wait_lgkm0();
    if (grp == 1) {
      if (g + 2 < SL) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    wg_barrier();
    __builtin_amdgcn_sched_barrier(0);
    mma_all(g + 3 < SL);
    __builtin_amdgcn_sched_barrier(0);
    if (tk == KT - 1) {
      // the tile is complete in this wave's registers: write its 64 x 64 part out, start the next tile from zero
      asm volatile("s_nop 0" : "+v"(acc[0][0]), "+v"(acc[TM - 1][TN - 1]));
      if (wide) {
        // operands requested only now: the fragments are dead, so the 48 operand registers fit (requested a slot earlier they made the
        // kernel spill, and every scratch access drains the DMA ring through the compiler's own vmcnt waits: 124 us instead of 80)
        f32x4 ebv[TN]; u32x2 eov[TM][TN];
        igemm_epilogue_wave_prefetch<TM, TN, WM, WN>(p, m0, n0, wm, wn, lane, M, ebv, eov);
        igemm_epilogue_wave<TM, TN, WM, WN>(p, acc, m0, n0, wm, wn, lane, M, smem + 3 * STG + wave * 2048, ebv, eov);
      }
      else igemm_epilogue_fast<T, TM, TN, WM, WN, false>(p, acc, m0, n0, wm, wn, fr, fg, M);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // stores and loads retire out of order with respect to each other
      tk = 0; ++ti;
      if (ti < ntl) tile_of(ti, m0, n0);
    } else {
      ++tk;
      if (grp == 0) {
        if (g + 2 < SL) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    wg_barrier();
    st = st == 2 ? 0 : st + 1;
  }
  if (grp == 0) wg_barrier();
}

// ------------------------------------------------------------------------------------------------
// Small-M tile (bf16), the layer2 / layer3 launches: 64 x 64 outputs per workgroup, K in 256-byte slices (128 bf16) through a ring of D
// LDS stages filled by LDS-DMA.  Waves 4-7 only REQUEST (eight 1-KiB pieces = 4 rows x 256 B per wave and slice, offsets as in the large
// tile above, D-1 slices ahead, a counted vmcnt before the slice's barrier).  Waves 0-3 MULTIPLY, and they split the slice's K, not the
// tile: wave w owns the whole 64 x 64 accumulator over k = 32 w .. 32 w + 31 of every slice, so a slice costs it 8 fragment reads for 16
// MFMAs (the 2 x 2 spatial split of igemm_ws64_kernel: 8 reads for 8 MFMAs - a SIMD returns LDS data at 64 B/clk, ~20 cycles per
// ds_read_b128, which is what paced that tile).  One barrier per slice: "slice t has landed" and, for the requesters, "the stage of slice
// t-1 is free" (its reads were consumed by MFMAs issued before the barrier).  The four partial accumulators are added through LDS in a
// fixed order (wave 0 + 1 + 2 + 3) by the epilogue, which also stages the rows for 16-byte stores; the residual / ReLU-mask operands of
// the epilogue are requested before the K loop.
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(512) void igemm_ks64_kernel(const l2s_conv_desc p) {
  if (p.prio) __builtin_amdgcn_s_setprio(3);
  typedef bf16_t T;
  constexpr int BM = 64, BN = 64, RB2 = 256, BK = 128, STG = (BM + BN) * RB2, PW = 4;   // PW: pieces per requester wave and operand
  constexpr unsigned NOPE = 0x80000000u;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int M = p.n_img * p.OH * p.OW;
  const int K = p.KH * p.KW * p.Cin;
  int mt, nt;
  {
    const int MT = (M + BM - 1) / BM, NT = (p.Cout + BN - 1) / BN, G = MT * NT;
    const int L = blockIdx.x, x = L & 7, slot = L >> 3, q = G >> 3, r = G & 7;
    const int t = x * q + min(x, r) + slot;
    if (p.xcd_mode == 0) { mt = t / NT; nt = t - mt * NT; } else { nt = t / MT; mt = t - nt * MT; }
  }
  const int m0 = mt * BM, n0 = nt * BN;
  const int KT = K / BK;

  if (wave >= 4) {
    // ---------------- requesters ----------------
    const int w = wave - 4;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    const long xpix = (long)p.n_img * p.IH * p.IW;
    i32x4s rx, rw;
    rx.x = (int)(uintptr_t)p.x; rx.y = (int)((uintptr_t)p.x >> 32); rx.z = (int)(((xpix - 1) * p.ldx + p.Cin) * 2L); rx.w = 0x00020000;
    rw.x = (int)(uintptr_t)p.w; rw.y = (int)((uintptr_t)p.w >> 32); rw.z = (int)((long)p.Cout * K * 2L); rw.w = 0x00020000;
    const int prow = lane >> 4, sch = (lane & 15) ^ ((4 * w + prow) & 15);   // piece row, source chunk (16-byte chunks XOR-swizzled by row & 15)
    const int ohw = p.OH * p.OW;
    const float r_ohw = 1.0f / (float)ohw, r_ow = 1.0f / (float)p.OW;
    int vbase[PW]; unsigned ntmask[PW], voffB[PW];
#pragma unroll
    for (int j = 0; j < PW; ++j) {
      const int m = m0 + 4 * (w + 4 * j) + prow;
      const bool ok = m < M;
      const int mm = ok ? m : 0;
      const int n_img = p.n_img > 1 ? fast_div(mm, ohw, r_ohw) : 0, rem = mm - n_img * ohw;
      const int oy = fast_div(rem, p.OW, r_ow), ox = rem - oy * p.OW;
      const int iy0 = oy * p.stride - p.pad, ix0 = ox * p.stride - p.pad;
      vbase[j] = (((n_img * p.IH + iy0) * p.IW + ix0) * p.ldx + sch * 8) * 2;
      unsigned mk = 0;
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int iy = iy0 + ky, ix = ix0 + kx;
          if (ky < p.KH && kx < p.KW && ok && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW) mk |= 1u << (ky * p.KW + kx);
        }
      ntmask[j] = ~mk;
      const int n = n0 + 4 * (w + 4 * j) + prow;
      voffB[j] = n < p.Cout ? (unsigned)(((long)n * K + sch * 8) * 2L) : OOR;
    }
    int c0 = 0, ky = 0, kx = 0, tapi = 0;
    const unsigned ldsA = lds0 + (unsigned)(w * 1024), ldsB = ldsA + (unsigned)(BM * RB2);
    auto request = [&](int stage) {                    // the slice at the cursor -> LDS stage `stage`; the cursor moves on
      const int toff = (ky * p.IW + kx) * p.ldx * 2;
      const unsigned soA = (unsigned)(c0 * 2), soB = (unsigned)((tapi * p.Cin + c0) * 2), sb = (unsigned)(stage * STG);
      unsigned vo[PW];
#pragma unroll
      for (int j = 0; j < PW; ++j) vo[j] = (((ntmask[j] >> tapi) & 1u) << 31) | (unsigned)(vbase[j] + toff);
#pragma unroll
      for (int j = 0; j < PW; ++j) dma_b128(rx, vo[j], soA, ldsA + sb + (unsigned)(j * 4096));
#pragma unroll
      for (int j = 0; j < PW; ++j) dma_b128(rw, voffB[j], soB, ldsB + sb + (unsigned)(j * 4096));
      const int nkx = kx + 1; const bool wx = nkx == p.KW; kx = wx ? 0 : nkx;
      const int nky = ky + (wx ? 1 : 0); const bool wy = nky == p.KH; ky = wy ? 0 : nky;
      tapi = wy ? 0 : tapi + 1; c0 += wy ? BK : 0;
    };
    auto wait_younger = [&](int younger) {             // all requests landed except those of the `younger` most recent slices
      if (D >= 4 && younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * PW) : "memory");
      else if (younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    static_assert(D == 3 || D == 4, "ring depth");
    int issued = 0;
#pragma unroll
    for (int s = 0; s < D - 1; ++s)
      if (s < KT) { request(s); ++issued; }
    wait_younger(issued - 1);
    wg_barrier();                                      // barrier #0: slice 0 landed
    int sf = D - 1;                                    // stage of the next request: (t + D - 1) % D
    for (int t = 0; t + 1 < KT; ++t) {
      if (issued < KT) { request(sf); ++issued; }
      sf = sf == D - 1 ? 0 : sf + 1;
      wait_younger(issued - (t + 2));                  // slice t+1 landed
      wg_barrier();                                    // barrier #(t+1)
    }
    return;
  }

  // ---------------- multipliers: wave w = K quarter w of every slice ----------------
  const int fr = lane & 15, fg = lane >> 4;
  const bool plain = !(p.flags & (L2S_CONV_DECONV2X2 | L2S_CONV_SCATTER)) && !(p.ldy & 7) && !(p.ldadd & 7) && !(p.ldref & 7) && !(p.Cout & 7) &&
                     !((uintptr_t)p.y & 15) && !((uintptr_t)p.add & 15) && !((uintptr_t)p.ref & 15) && !((uintptr_t)p.bias & 15) &&
                     (long)M * p.ldy * 2 < (1L << 31) && (!p.add || (long)M * p.ldadd * 2 < (1L << 31)) && (!p.ref || (long)M * p.ldref * 2 < (1L << 31));
  // epilogue operands of this thread's two output chunks (row = chunk / 8, 8 channels each), requested now: their latency passes under the K loop
  const auto ry = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, 0x7FFFFFFF, 0x00020000);
  const auto radd = __builtin_amdgcn_make_buffer_rsrc((void*)(p.add ? p.add : p.y), 0, 0x7FFFFFFF, 0x00020000);
  const auto rref = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ref ? p.ref : p.y), 0, 0x7FFFFFFF, 0x00020000);
  u32x4v av[2], rv[2]; unsigned orow[2];
  f32x4 bq[2] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};   // bias of the thread's 8 columns (both chunks: same columns)
  if (plain && p.bias && n0 + (tid & 7) * 8 < p.Cout) { bq[0] = *(const f32x4*)(p.bias + n0 + (tid & 7) * 8); bq[1] = *(const f32x4*)(p.bias + n0 + (tid & 7) * 8 + 4); }
  if (plain) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c = tid + 256 * k, row = c >> 3, col = (c & 7) * 8;
      const int m = m0 + row, n = n0 + col;
      const bool ok = m < M && n < p.Cout;
      orow[k] = ok ? (unsigned)m : NOPE;
      if (p.add) av[k] = __builtin_amdgcn_raw_buffer_load_b128(radd, ok ? (unsigned)((m * p.ldadd + n) * 2) : NOPE, 0, 0);
      if (p.ref) rv[k] = __builtin_amdgcn_raw_buffer_load_b128(rref, ok ? (unsigned)((m * p.ldref + n) * 2) : NOPE, 0, 0);
    }
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int ch = ((4 * wave + fg) ^ fr) << 4;          // this lane's 16-byte chunk of a 256-byte row (row & 15 == fr)
  const int offa = fr * RB2 + ch, offb = BM * RB2 + fr * RB2 + ch;
  int st = 0;
  for (int t = 0; t < KT; ++t) {
    wg_barrier();                                      // barrier #t: slice t landed
    const char* base = smem + st * STG;
    uint4 fa[4], fb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = *(const uint4*)(base + offa + i * 16 * RB2);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = *(const uint4*)(base + offb + j * 16 * RB2);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = Mma<T>::run(fb[j], fa[i], acc[i][j]);
    st = st == D - 1 ? 0 : st + 1;
  }
  // ---- the four K quarters meet in LDS: part[w][row][col] fp32, row pitch 68 floats ----
  constexpr int LDW = BN + 4;
  float* part = (float*)smem;
  __syncthreads();                                     // every wave is done reading the ring (the requesters have left)
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) *(f32x4*)(part + (wave * BM + i * 16 + fr) * LDW + j * 16 + fg * 4) = acc[i][j];
  __syncthreads();
  if (plain) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c = tid + 256 * k, row = c >> 3, col = (c & 7) * 8;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 lo = *(const f32x4*)(part + (q * BM + row) * LDW + col), hi = *(const f32x4*)(part + (q * BM + row) * LDW + col + 4);
        v[0] += lo[0]; v[1] += lo[1]; v[2] += lo[2]; v[3] += lo[3]; v[4] += hi[0]; v[5] += hi[1]; v[6] += hi[2]; v[7] += hi[3];
      }
      const int n = n0 + col;
      if (p.bias) { v[0] += bq[0][0]; v[1] += bq[0][1]; v[2] += bq[0][2]; v[3] += bq[0][3]; v[4] += bq[1][0]; v[5] += bq[1][1]; v[6] += bq[1][2]; v[7] += bq[1][3]; }
      if (p.add) {
        const unsigned wv[4] = {av[k].x, av[k].y, av[k].z, av[k].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[2 * e] += __uint_as_float(wv[e] << 16); v[2 * e + 1] += __uint_as_float(wv[e] & 0xFFFF0000u); }
      }
      if (p.flags & L2S_CONV_RELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if (p.ref) {
        const unsigned wv[4] = {rv[k].x, rv[k].y, rv[k].z, rv[k].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          if (!(__uint_as_float(wv[e] << 16) > 0.f)) v[2 * e] = 0.f;
          if (!(__uint_as_float(wv[e] & 0xFFFF0000u) > 0.f)) v[2 * e + 1] = 0.f;
        }
      }
      u32x4v pk;
      pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16); pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
      pk.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16); pk.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
      __builtin_amdgcn_raw_buffer_store_b128(pk, ry, orow[k] != NOPE ? (unsigned)((orow[k] * p.ldy + n) * 2) : NOPE, 0, 0);
    }
    return;
  }
  // general output forms (strided scatter, pixel shuffle, odd pitches): wave w finishes rows 16 w .. 16 w + 15 through the shared epilogue
  f32x4 fin[1][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    f32x4 sum = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 4; ++q) { const f32x4 v = *(const f32x4*)(part + (q * BM + wave * 16 + fr) * LDW + j * 16 + fg * 4); sum[0] += v[0]; sum[1] += v[1]; sum[2] += v[2]; sum[3] += v[3]; }
    fin[0][j] = sum;
  }
  igemm_epilogue<T, 1, 4, 16, 64, false>(p, fin, m0, n0, wave, 0, fr, fg, M);
}

// ------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 on ONE map (bf16): the "patch" tile.  What paces the small-M launches is the rate at which one CU pulls operand
// bytes out of L2 (~70 GB/s, tools/ws64_stamps.py: 265 ns per 16 KiB slice whatever the prefetch depth), and an im2col-style tile pulls
// every input pixel nine times: 64 x 64 outputs of layer3's conv2 cost (64 + 64) rows x 2304 x 2 B = 590 KB per workgroup.  Here a workgroup
// owns 128 CONSECUTIVE pixels (row-major over the map) x BN output channels and stages, per 32-channel step, the input pixels
// m0 - (W + 1) .. m0 + 127 + (W + 1) ONCE ("patch", PR rows of 64 B) next to the nine taps' weight rows (9 BN rows of 64 B).  Tap (ky, kx)
// of output pixel m reads input pixel m + (ky - 1) W + (kx - 1): in the patch that is row (m - m0) + ky W + kx, the same shift for all
// 128 pixels, so the MFMA A fragment of a tap is a plain ds_read_b128 at a shifted row - no gather, no per-tap copy.  What the shift gets
// wrong are the map's left / right edges (kx = 0 at x = 0 and kx = 2 at x = W - 1 would read the neighbouring row's pixel): those lanes'
// fragments are zeroed in registers; above / below the map the patch rows are out-of-range buffer addresses, i.e. zeros.
// layer3 conv2 (BN = 32: 19 x 8 = 152 workgroups): 16 KB (patch) + 18 KB (weights) per step x 8 steps = 278 KB per workgroup instead of 590.
// Waves 4-7 request (LDS-DMA, NP one-KiB pieces per wave and step, NS - 1 steps ahead), waves 0-3 multiply: wave w owns pixels 32 w .. 32 w + 31
// x all BN channels, and walks the nine taps of a step software-pipelined (the fragments of tap q + 1 are requested before the MFMAs of
// tap q: one wave per SIMD, nothing else hides the LDS latency); 2 + BN / 16 fragment reads for 2 BN / 16 MFMAs per tap.  One barrier per
// step, as in igemm_ks64_kernel.
// Measured (isolated launch, forward form): layer3 conv2 9.5 us (wave-specialised tile 13.0, K-split tile 10.8), layer2 conv2 9.2 (11.7),
// layer4-on-the-map conv2 21.2 (31.9), RPN 3x3 39.0 (62.0).  Knock-outs on layer3's shape: requests alone 3.2 us over 8 steps (the ~70 GB/s
// per CU again), multipliers alone 5.6 us - the LDS read path (144 KB of fragment reads per step and CU, ~80 % of 128 B/clk) now paces
// it; a split by filter row (three waves over the whole 128 x BN tile, 0.6 reads per MFMA instead of 1) was built and is slower (12.2 us:
// three taps per barrier leave nothing to pipeline).  Inside the step the LDS footprint matters as much as the launch time: with a ring
// of 4 stages (139 KB) the tile keeps the weight-gradient workgroups off its CUs and the step gains 0.8 %; with 2 stages (72 KB) 3.4 %.
// ------------------------------------------------------------------------------------------------
template <int BN, int PR, int NS>
__global__ __launch_bounds__(512) void igemm_p3_kernel(const l2s_conv_desc p) {
  if (p.prio) __builtin_amdgcn_s_setprio(3);
  typedef bf16_t T;
  constexpr int BM = 128, RB = 64, TN = BN / 16;
  constexpr int PA = PR / 16, PB = 9 * BN / 16;                  // one-KiB pieces (16 rows x 64 B) of the patch / of the weights per step
  constexpr int NP = (PA + PB + 3) / 4;                          // pieces per requester wave (the last ones may be padding)
  constexpr int STG = 4 * NP * 1024;
  static_assert(PR % 16 == 0 && BN % 16 == 0 && NS >= 2 && NS <= 4, "patch tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int W = p.OW, M = p.OH * p.OW;
  const int K = 9 * p.Cin;
  int mt, nt;
  {
    const int MT = (M + BM - 1) / BM, NT = (p.Cout + BN - 1) / BN, G = MT * NT;
    const int L = blockIdx.x, x = L & 7, slot = L >> 3, q = G >> 3, r = G & 7;
    const int t = x * q + min(x, r) + slot;                      // an XCD's chunk of tiles: a few pixel tiles x every channel tile (shared patch in L2)
    mt = t / NT; nt = t - mt * NT;
  }
  const int m0 = mt * BM, n0 = nt * BN;
  const int KT = p.Cin / 32;

  if (wave >= 4) {
    // ---------------- requesters ----------------
    const int w = wave - 4;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    i32x4s rx, rw;
    rx.x = (int)(uintptr_t)p.x; rx.y = (int)((uintptr_t)p.x >> 32); rx.z = (int)((((long)M - 1) * p.ldx + p.Cin) * 2L); rx.w = 0x00020000;
    rw.x = (int)(uintptr_t)p.w; rw.y = (int)((uintptr_t)p.w >> 32); rw.z = (int)((long)p.Cout * K * 2L); rw.w = 0x00020000;
    const int prow = lane >> 2, pch = lane & 3;                   // row of the piece, 16-byte chunk of the row
    unsigned voff[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      const int k = w + 4 * j;                                    // piece index: patch pieces first, then weight pieces, then padding
      unsigned v = OOR;
      if (k < PA) {
        const int idx = m0 - (W + 1) + 16 * k + prow;
        if (idx >= 0 && idx < M) v = (unsigned)((idx * p.ldx + pch * 8) * 2);
      } else if (k < PA + PB) {
        const int brow = 16 * (k - PA) + prow, tap = brow / BN, n = n0 + brow - tap * BN;
        if (n < p.Cout) v = (unsigned)(((long)n * K + tap * p.Cin + pch * 8) * 2L);
      }
      voff[j] = v;
    }
    int c0 = 0;
    auto request = [&](int stage) {
      const unsigned so = (unsigned)(c0 * 2), base = lds0 + (unsigned)(stage * STG + w * 1024);
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        const bool patch = w + 4 * j < PA;                        // (wave-uniform choice of the descriptor: scalar selects)
        i32x4s r;
        r.x = patch ? rx.x : rw.x; r.y = patch ? rx.y : rw.y; r.z = patch ? rx.z : rw.z; r.w = rx.w;
        dma_b128(r, voff[j], so, base + (unsigned)(j * 4096));
      }
      c0 += 32;
    };
    auto wait_younger = [&](int younger) {                        // everything landed except the `younger` most recent steps
      if (NS >= 4 && younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory");
      else if (younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    int issued = 0;
#pragma unroll
    for (int s_ = 0; s_ < NS - 1; ++s_)
      if (s_ < KT) { request(s_); ++issued; }
    wait_younger(issued - 1);
    wg_barrier();                                                 // barrier #0: step 0 landed
    int sf = NS - 1;
    for (int t = 0; t + 1 < KT; ++t) {
      if (issued < KT) { request(sf); ++issued; }
      sf = sf == NS - 1 ? 0 : sf + 1;
      wait_younger(issued - (t + 2));                             // step t + 1 landed
      wg_barrier();                                               // barrier #(t + 1)
    }
    return;
  }

  // ---------------- multipliers: wave w owns pixels 32 w .. 32 w + 31 x all BN channels, all nine taps ----------------
  const int fr = lane & 15, fg = lane >> 4;
  unsigned keepL[2], keepR[2];                                  // all-ones, or zero for an edge lane (ANDed into the fragment registers)
  {
    const float rw_ = 1.0f / (float)W;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = min(m0 + 32 * wave + 16 * i + fr, M - 1);
      const int y = fast_div(m, W, rw_), x = m - y * W;
      keepL[i] = x == 0 ? 0u : 0xFFFFFFFFu; keepR[i] = x == W - 1 ? 0u : 0xFFFFFFFFu;
    }
  }
  f32x4 acc[2][TN];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int arow = (32 * wave + fr) * RB + fg * 16;             // + (16 i + ky W + kx) * 64
  const int brow = PA * 1024 + fr * RB + fg * 16;               // + (tap BN + 16 j) * 64
  int st = 0;
  wg_barrier();                                                 // barrier #0
  for (int t = 0; t < KT; ++t) {
    const char* base = smem + st * STG;
    // the nine taps of the step, software-pipelined inside the wave (one wave per SIMD: nothing else hides the LDS latency): the
    // fragments of tap q + 1 are requested before the MFMAs of tap q are issued
    uint4 fa[2][2], fb[2][TN];
    auto load = [&](int q, int b_) {
      const int ky = q / 3, kx = q - 3 * ky;
#pragma unroll
      for (int i = 0; i < 2; ++i) fa[b_][i] = *(const uint4*)(base + arow + (ky * W + 16 * i + kx) * RB);
#pragma unroll
      for (int j = 0; j < TN; ++j) fb[b_][j] = *(const uint4*)(base + brow + (q * BN + 16 * j) * RB);
    };
    load(0, 0);
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const int b_ = q & 1, kx = q % 3;
      if (q + 1 < 9) load(q + 1, b_ ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      if (kx != 1) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const unsigned k_ = kx == 0 ? keepL[i] : keepR[i];
          fa[b_][i].x &= k_; fa[b_][i].y &= k_; fa[b_][i].z &= k_; fa[b_][i].w &= k_;
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::run(fb[b_][j], fa[b_][i], acc[i][j]);
      __builtin_amdgcn_sched_barrier(0);
    }
    st = st == NS - 1 ? 0 : st + 1;
    if (t + 1 < KT) { wait_lgkm0(); wg_barrier(); }             // this step's reads have returned: its stage may be refilled
  }
  igemm_epilogue<T, 2, TN, 32, BN, false>(p, acc, m0, n0, wave, 0, fr, fg, M);
}

// ------------------------------------------------------------------------------------------------
// RoIAlign fused into the first bottleneck of the RoI head (bf16): crop-and-resize (NET:107-149) -> layer4.0.conv1 (1x1, + bias + ReLU) and
// layer4.0.downsample (1x1, + bias), RES:271-273 - three launches of the unfused path (the crop kernel and two convolutions that each
// re-read the 25.7 MB crop) in one.  ONE WORKGROUP PER RoI (256 RoIs = 256 compute units):
//   * the P x P x C crop of the RoI (49 x 1024 bf16 = 98 KiB) is gathered ONCE, by all 512 threads, into LDS - bilinear blend in fp32,
//     one rounding, the arithmetic of the stand-alone crop kernel (roi_sample.h) - and written once to `pooled`: the weight gradients
//     of both convolutions and the backward pass read it, the forward pass never does again;
//   * the two weight matrices are then walked as ONE matrix of N1 + N2 rows: wave w owns columns 16 w .. 16 w + 15 of every
//     128-column tile and reads its weights STRAIGHT INTO MFMA B-FRAGMENT LAYOUT (a [Cout][K] row is K-contiguous: 16 bytes per lane),
//     eight 128-byte K slices ahead in registers - no LDS, no barrier and no cross-wave traffic for half of the operand bytes; the crop
//     fragments come from LDS (8 reads for 8 MFMAs per slice).  Rows 49..63 of the fourth row tile alias row 48 and are never stored.
// What paces it: every workgroup streams all (N1 + N2) x C weights (5.2 MB) from L2 - the price of gathering the crop once.  The first
// form of this kernel staged the weights through a three-stage LDS-DMA ring (16 KiB per slice, all the LDS the crop leaves): 173-183 us,
// because 32 KiB in flight per CU is a third of what the L2 round trip needs; the register ring keeps 128 KiB in flight.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void roialign_block0_kernel(const l2s_roi_block0_desc p) {
  typedef bf16_t T;
  constexpr int BN = 128, PF = 8;                                       // columns per tile; K slices of weights in flight per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int roi = blockIdx.x, PP = p.P * p.P, C = p.C;
  const int rowb = C * 2;                                               // bytes of one crop row in LDS
  char* As = smem;                                                      // [PP][C] bf16, 16-byte chunks XOR-swizzled by (row & 15)
  const T* __restrict__ feat = (const T*)p.feat;
  const int fr = lane & 15, fg = lane >> 4;

  // ---- weight stream: slice t = (n-tile t / KS, K slice t % KS) of the N1 + N2 rows ----
  const int KS = C / 64, NT1 = p.N1 / BN, NT = NT1 + p.N2 / BN, TOT = NT * KS;
  const auto rw1 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w1, 0, (int)((long)p.N1 * C * 2L), 0x00020000);
  const auto rw2 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w2, 0, (int)((long)p.N2 * C * 2L), 0x00020000);
  const unsigned voffB = (unsigned)(((16 * wave + fr) * C + fg * 8) * 2);
  int rq_nt = 0, rq_ks = 0;                                             // request cursor
  uint4 pf[PF][2];
  auto request = [&](uint4 (&dst)[2]) {
    const bool first = rq_nt < NT1;
    const int ntr = first ? rq_nt : rq_nt - NT1;
    const unsigned so = (unsigned)((ntr * BN * C + rq_ks * 64) * 2);
    const auto r = first ? rw1 : rw2;
    dst[0] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, voffB, so, 0));
    dst[1] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(r, voffB, so + 64u, 0));
    if (++rq_ks == KS) { rq_ks = 0; ++rq_nt; }
  };
#pragma unroll
  for (int j = 0; j < PF; ++j)
    if (j < TOT) request(pf[j]);

  // ---- the crop: sample table, then 8 channels per thread and step ----
  {
    RoiTaps* tab = (RoiTaps*)(smem + PP * rowb);
    if (tid < PP) {
      const int py = tid / p.P, px = tid - py * p.P;
      tab[tid] = roi_taps(roi_sample(p.rois + roi * 5, p.H, p.W, p.P, py, px, p.spatial_scale, (float)p.H, (float)p.W), p.H, p.W, C);
    }
    __syncthreads();
    const int cpr = C / 8;                                              // 16-byte chunks per row
    T* pooled = (T*)p.pooled + (long)roi * PP * C;
    // four chunks per trip: their sixteen loads are requested before the first blend (the taps are branch-free, roi_sample.h); one chunk per
    // trip waited for its own loads thirteen times over (54 us for this phase alone)
    constexpr int UN = 4;
    const int total = PP * cpr;
    for (int i0 = tid; i0 < total; i0 += 512 * UN) {
      RoiQuad q[UN]; int row[UN], ch[UN];
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        const int idx = min(i0 + 512 * u, total - 1);
        row[u] = idx / cpr; ch[u] = idx - row[u] * cpr;
        q[u] = roi_load8(feat, tab[row[u]], ch[u] * 8);
      }
#pragma unroll
      for (int u = 0; u < UN; ++u) {
        if (i0 + 512 * u < total) {
          const uint4 v = roi_mix8(q[u], tab[row[u]]);
          *(uint4*)(As + row[u] * rowb + ((ch[u] ^ (row[u] & 15)) << 4)) = v;
          *(uint4*)(pooled + (long)row[u] * C + ch[u] * 8) = v;
        }
      }
    }
    __syncthreads();
  }

  // ---- the product ----
  int arow[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) arow[i] = min(16 * i + fr, PP - 1);
  f32x4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  int nt = 0, ks = 0;
  auto step = [&](uint4 (&fb)[2], bool more) {
    if (p.debug & 1) { acc[0][0] += __uint_as_float(fb[0].x ^ fb[1].y); }
    else {
#pragma unroll
    for (int kg = 0; kg < 2; ++kg) {
      const int ch = ks * 8 + kg * 4 + fg;
      uint4 fa[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = *(const uint4*)(As + arow[i] * rowb + ((ch ^ (arow[i] & 15)) << 4));
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = Mma<T>::run(fb[kg], fa[i], acc[i]);
    }
    }
    if (more && !(p.debug & 2)) request(fb);                            // the register pair just consumed takes the slice PF ahead
    if (++ks == KS) {
      // this wave's 16 columns of the 128-column tile are complete: + bias (+ ReLU for conv1), 4 channels = 8 bytes per lane and row tile
      const bool first = nt < NT1;
      const int n = (first ? nt : nt - NT1) * BN + 16 * wave + fg * 4;
      const float* bias = first ? p.b1 : p.b2;
      const int ldy = first ? p.N1 : p.N2;
      T* y = (T*)(first ? p.y1 : p.y2) + (long)roi * PP * ldy + n;
      const f32x4 bv = bias ? *(const f32x4*)(bias + n) : (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = 16 * i + fr;
        float v0 = acc[i][0] + bv[0], v1 = acc[i][1] + bv[1], v2 = acc[i][2] + bv[2], v3 = acc[i][3] + bv[3];
        if (first) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
        if (row < PP && !((p.debug & 4) && nt > 0)) {
          uint2 pk;
          pk.x = (uint32_t)f2bf(v0) | ((uint32_t)f2bf(v1) << 16); pk.y = (uint32_t)f2bf(v2) | ((uint32_t)f2bf(v3) << 16);
          *(uint2*)(y + (long)row * ldy) = pk;
        }
        acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
      ks = 0; ++nt;
    }
  };
  int t = 0;
  for (; t + 2 * PF <= TOT; t += PF) {                                  // steady state: every slice re-arms its register pair
#pragma unroll
    for (int j = 0; j < PF; ++j) step(pf[j], true);
  }
#pragma unroll
  for (int j = 0; j < 2 * PF; ++j)                                      // tail: fewer than 2 PF slices left
    if (t + j < TOT) step(pf[j % PF], t + j + PF < TOT);
}

// y = epilogue(slab 0 + slab 1 + ... in this order); 4 channels per thread
template <typename T, bool OUTF32>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const l2s_conv_desc p, long total, int split) {
  const long e = (blockIdx.x * (long)blockDim.x + threadIdx.x) * 4;
  if (e >= total) return;
  const long m = e / p.Cout; const int n = (int)(e - m * p.Cout);
  float4 v4 = *(const float4*)(p.ws + e);
  for (int s = 1; s < split; ++s) {
    const float4 u = *(const float4*)(p.ws + s * total + e);
    v4.x += u.x; v4.y += u.y; v4.z += u.z; v4.w += u.w;
  }
  float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (p.bias) v[r] += p.bias[n + r];
    if (p.add) v[r] += Elem<T>::ld((const T*)p.add + m * p.ldadd + n + r);
    if (p.flags & L2S_CONV_RELU) v[r] = fmaxf(v[r], 0.f);
    if (p.ref) { if (!(Elem<T>::ld((const T*)p.ref + m * p.ldref + n + r) > 0.f)) v[r] = 0.f; }
  }
  if (OUTF32) { float* o = (float*)p.y + m * p.ldy + n; for (int r = 0; r < 4; ++r) o[r] = v[r]; }
  else { T* o = (T*)p.y + m * p.ldy + n; for (int r = 0; r < 4; ++r) Elem<T>::st(o + r, v[r]); }
}

template <typename T, bool OUTF32>
int launch_splitk_reduce(const l2s_conv_desc& d, int split, hipStream_t st) {
  const long total = (long)d.n_img * d.OH * d.OW * d.Cout;
  L2S_LAUNCH((splitk_reduce_kernel<T, OUTF32>), dim3((int)((total / 4 + 255) / 256)), dim3(256), 0, st, d, total, split);
  return l2s_check_launch();
}

// the split the launch will use: only when the caller forces one (d.split_k > 1, with a workspace), limited by the K slices (>= 2 per
// range, no empty range) and by the workspace; 1 = no split.  An automatic rule (few tiles x many slices) was built and measured: the
// second launch costs more than the split saves on every shape tried - 80 tiles x 32 slices 10.9 -> 13.1 us, 80 x 72 (3x3) 14.8 -> 16.7,
// 120 x 16 7.4 -> 11.0 - and inside the step it made 600x800 images 7 % slower (91 extra launches on layer3's chain) before the profile
// showed it.  split_k = 0 therefore means "no split".
static int splitk_factor(const l2s_conv_desc& d, int KT, long tiles64, bool auto_ok) {
  (void)tiles64; (void)auto_ok;
  if (!d.ws || d.split_k <= 1 || (d.flags & (L2S_CONV_SCATTER | L2S_CONV_DECONV2X2)) || (d.Cout % 4)) return 1;
  const long out = (long)d.n_img * d.OH * d.OW * d.Cout;
  int s = d.split_k;
  if (s > KT / 2) s = KT / 2;
  while (s > 1 && (long)s * out > (long)d.ws_floats) --s;
  while (s > 1 && (s - 1) * cdiv(KT, s) >= KT) --s;                 // no empty range
  return s < 1 ? 1 : s;
}

template <typename T, int BM, int BN, bool OUTF32>
int launch_igemm(const l2s_conv_desc& d, hipStream_t st) {
  const int M = d.n_img * d.OH * d.OW;
  int split = 1;
  if (BM == 64 && d.split_k > 1) {                                  // forced only: the generic kernel serves K tails and the f32 mode
    const int K = d.KH * d.KW * d.Cin, KT = cdiv(K, ROWB / (int)sizeof(T));
    split = splitk_factor(d, KT, (long)cdiv(M, BM) * cdiv(d.Cout, BN), false);
  }
  dim3 grid(cdiv(M, BM) * cdiv(d.Cout, BN), 1, split);
  size_t lds = 2 * (BM + BN) * ROWB;
  static bool attr_done = false;
  if (!attr_done) { (void)hipFuncSetAttribute((const void*)igemm_kernel<T, BM, BN, OUTF32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done = true; }
  L2S_LAUNCH((igemm_kernel<T, BM, BN, OUTF32>), grid, dim3(256), lds, st, d);
  if (split > 1) return launch_splitk_reduce<T, OUTF32>(d, split, st);
  return l2s_check_launch();
}

template <typename T, int BM, int BN, int WGM, int WGN, int D, bool OUTF32>
int launch_igemm_ring(const l2s_conv_desc& d, hipStream_t st) {
  const int M = d.n_img * d.OH * d.OW;
  dim3 grid(cdiv(M, BM) * cdiv(d.Cout, BN));
  size_t lds = (size_t)2 * (BM + BN) * ROWB;
  static bool attr_done = false;
  if (!attr_done) { (void)hipFuncSetAttribute((const void*)igemm_ring_kernel<T, BM, BN, WGM, WGN, D, OUTF32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done = true; }
  L2S_LAUNCH((igemm_ring_kernel<T, BM, BN, WGM, WGN, D, OUTF32>), grid, dim3(64 * WGM * WGN), lds, st, d);
  return l2s_check_launch();
}

template <int D, int BN, bool STAMP = false>
int launch_igemm_ws64(const l2s_conv_desc& d, hipStream_t st, int split = 1) {
  const int M = d.n_img * d.OH * d.OW;
  dim3 grid(cdiv(M, 64) * cdiv(d.Cout, BN), 1, split);
  const size_t lds = 2 * 128 * 128;                                           // (also holds the staged 64 x (BN + 4) fp32 tile of the epilogue)
  L2S_LAUNCH((igemm_ws64_kernel<D, BN, STAMP>), grid, dim3(512), lds, st, d);
  if (split > 1) return launch_splitk_reduce<bf16_t, false>(d, split, st);
  return l2s_check_launch();
}

template <typename T, int BM, int BN, int WGM, int WGN, int D, bool OUTF32, bool TAPIN>
int launch_igemm_sp(const l2s_conv_desc& d, hipStream_t st) {
  const int M = d.n_img * d.OH * d.OW;
  dim3 grid(cdiv(M, BM) * cdiv(d.Cout, BN));
  size_t lds = (size_t)3 * (BM + BN) * ROWB;
  static bool attr_done = false;
  if (!attr_done) { (void)hipFuncSetAttribute((const void*)igemm_sp_kernel<T, BM, BN, WGM, WGN, D, OUTF32, TAPIN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done = true; }
  L2S_LAUNCH((igemm_sp_kernel<T, BM, BN, WGM, WGN, D, OUTF32, TAPIN>), grid, dim3(64 * WGM * WGN), lds, st, d);
  return l2s_check_launch();
}

int launch_roialign_block0(const l2s_roi_block0_desc& d, hipStream_t st) {
  const size_t lds = (size_t)d.P * d.P * d.C * 2 + 64 * sizeof(RoiTaps);   // the crop + the sample table
  static bool attr_done = false;
  if (!attr_done) { (void)hipFuncSetAttribute((const void*)roialign_block0_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_done = true; }
  L2S_LAUNCH(roialign_block0_kernel, dim3(d.R), dim3(512), lds, st, d);
  return l2s_check_launch();
}

template <int D>
int launch_igemm_ks64(const l2s_conv_desc& d, hipStream_t st) {
  const int M = d.n_img * d.OH * d.OW;
  dim3 grid(cdiv(M, 64) * cdiv(d.Cout, 64));
  const size_t lds = (size_t)D * 128 * 256;
  static bool attr_done = false;
  if (!attr_done) { (void)hipFuncSetAttribute((const void*)igemm_ks64_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done = true; }
  L2S_LAUNCH((igemm_ks64_kernel<D>), grid, dim3(512), lds, st, d);
  return l2s_check_launch();
}

template <int BN, int PR, int NS>
int launch_igemm_p3(const l2s_conv_desc& d, hipStream_t st) {
  const int M = d.n_img * d.OH * d.OW;
  dim3 grid(cdiv(M, 128) * cdiv(d.Cout, BN));
  const size_t lds = (size_t)NS * 4 * ((PR / 16 + 9 * BN / 16 + 3) / 4) * 1024;
  static bool attr_done = false;
  if (!attr_done) { (void)hipFuncSetAttribute((const void*)igemm_p3_kernel<BN, PR, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done = true; }
  L2S_LAUNCH((igemm_p3_kernel<BN, PR, NS>), grid, dim3(512), lds, st, d);
  return l2s_check_launch();
}

template <int BM, int BN, bool STAMP = false>
int launch_igemm_dma(const l2s_conv_desc& d, hipStream_t st) {
  const int M = d.n_img * d.OH * d.OW;
  dim3 grid(cdiv(M, BM) * cdiv(d.Cout, BN));
  const size_t lds = (size_t)3 * (BM + BN) * ROWB + (STAMP ? 8192 : 0);
  static bool attr_done = false;
  if (!attr_done) { (void)hipFuncSetAttribute((const void*)igemm_dma_kernel<BM, BN, STAMP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done = true; }
  L2S_LAUNCH((igemm_dma_kernel<BM, BN, STAMP>), grid, dim3(512), lds, st, d);
  return l2s_check_launch();
}

int launch_igemm_dma196(const l2s_conv_desc& d, hipStream_t st) {
  const int M = d.n_img * d.OH * d.OW;
  dim3 grid(cdiv(M, T196) * cdiv(d.Cout, 128));
  const size_t lds = (size_t)3 * (256 + 128) * ROWB;
  static bool attr_done = false;
  if (!attr_done) { (void)hipFuncSetAttribute((const void*)igemm_dma196_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done = true; }
  L2S_LAUNCH(igemm_dma196_kernel, grid, dim3(512), lds, st, d);
  return l2s_check_launch();
}

int launch_igemm_dma256(const l2s_conv_desc& d, hipStream_t st) {
  const int M = d.n_img * d.OH * d.OW;
  dim3 grid(cdiv(M, 256) * cdiv(d.Cout, 256));
  const size_t lds = (size_t)3 * 512 * 64 + 8 * 4096;     // the ring + 4 KiB of epilogue staging per wave = 128 KiB
  static bool attr_done = false;
  if (!attr_done) { (void)hipFuncSetAttribute((const void*)igemm_dma256_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done = true; }
  L2S_LAUNCH(igemm_dma256_kernel, grid, dim3(512), lds, st, d);
  return l2s_check_launch();
}

template <int BM, int BN>
int launch_igemm_pdma(const l2s_conv_desc& d, hipStream_t st) {
  const int M = d.n_img * d.OH * d.OW;
  const int G = cdiv(M, BM) * cdiv(d.Cout, BN);
  const int cap = l2s_knobs::pdma_wgs > 0 ? (l2s_knobs::pdma_wgs & ~7) : 256;  // (tools: fewer resident workgroups than CUs leaves whole CUs to the other queues)
  const int grid = G < cap ? ((G + 7) & ~7) : cap;     // one resident workgroup per CU, a multiple of 8 (the XCD-aware tile order)
  const size_t lds = (size_t)3 * (BM + BN) * ROWB + 8 * 2048;      // the ring + 2 KiB of epilogue staging per wave = 160 KiB
  static bool attr_done = false;
  if (!attr_done) { (void)hipFuncSetAttribute((const void*)igemm_pdma_kernel<BM, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done = true; }
  L2S_LAUNCH((igemm_pdma_kernel<BM, BN>), dim3(grid), dim3(512), lds, st, d);
  return l2s_check_launch();
}

}  // namespace


// ---- kernel choice (one place; l2s_conv_plan_name reports it) ----
enum ConvPlan { PLAN_EINVAL = 0, PLAN_GENERIC64, PLAN_GENERIC128, PLAN_RING64, PLAN_RING128, PLAN_WS64, PLAN_KS64, PLAN_KS64_D3, PLAN_SP224, PLAN_SP256,
                PLAN_DMA256, PLAN_DMA256_STAMPED, PLAN_WS64_SPLITK, PLAN_P3_32_256, PLAN_P3_64_256, PLAN_P3_32_384, PLAN_P3_64_384, PLAN_RING128X64, PLAN_PDMA256, PLAN_DMA256X256, PLAN_DMA196 };
static const char* const PLAN_NAMES[] = {"invalid", "igemm_kernel<64,64>", "igemm_kernel<128,128>", "igemm_ring_kernel<64,64>", "igemm_ring_kernel<128,128>",
                                         "igemm_ws64_kernel", "igemm_ks64_kernel<4>", "igemm_ks64_kernel<3>", "igemm_sp_kernel<224,128>", "igemm_sp_kernel<256,128>",
                                         "igemm_dma_kernel<256,128>", "igemm_dma_kernel<256,128,stamped>", "igemm_ws64_kernel + splitk_reduce_kernel", "igemm_p3_kernel<32,256>", "igemm_p3_kernel<64,256>", "igemm_p3_kernel<32,384>", "igemm_p3_kernel<64,384>", "igemm_ring_kernel<128,64>", "igemm_pdma_kernel<256,128>", "igemm_dma256_kernel", "igemm_dma196_kernel"};
static ConvPlan conv_plan(const l2s_conv_desc* d, int dtype, bool* tapin_out) {
  if (!d || !d->x || !d->w || !d->y || (dtype != L2S_BF16 && dtype != L2S_F32)) return PLAN_EINVAL;
  const bool bf = dtype == L2S_BF16;
  const int ve = bf ? 8 : 4, bk = bf ? 64 : 32;
  const long esz = bf ? 2 : 4;
  const int K = d->KH * d->KW * d->Cin, ntaps = d->KH * d->KW;
  if (d->ldx % ve || K % ve || (ntaps > 1 && d->Cin % bk)) return PLAN_EINVAL;
  if ((d->flags & L2S_CONV_DECONV2X2) && (d->Cout % 16)) return PLAN_EINVAL;
  const long M = (long)d->n_img * d->OH * d->OW;
  if (M >= (1 << 24)) return PLAN_EINVAL;
  const bool f32o = d->flags & L2S_CONV_OUT_F32;
  // tile: 128x128 when it fills the chip (>= ~1 workgroup per CU), else 64x64; the large tiles (one workgroup per CU) when their grid is one
  // round over most of the 256 CUs and the K loop is long enough to pay for the prologue (the layer4@RoIs shapes)
  const long t128 = (long)cdiv(M, 128) * cdiv(d->Cout, 128);
  int tile = d->tile ? d->tile : ((t128 >= 200 && d->Cout >= 96) ? 128 : 64);
  if (!d->tile) {
    const long t256 = (long)cdiv(M, 256) * cdiv(d->Cout, 128);
    if (t256 >= 160 && t256 <= 256 && K >= 1024) tile = ((long)cdiv(M, 224) * cdiv(d->Cout, 128) <= 256) ? 224 : 256;   // 224 rows: 224 instead of 196 workgroups
  }
  const long xb = (long)d->n_img * d->IH * d->IW * d->ldx * esz, wb = (long)d->Cout * K * esz;
  // buffer-descriptor kernels: whole 128-byte K slices per tap and 31-bit operand extents
  const bool desc_ok = (d->Cin % bk == 0) && xb < (1L << 31) && wb < (1L << 31);
  if (tapin_out) *tapin_out = ntaps > 1 && ntaps <= 32 && xb < (1L << 30);     // K walked channel-chunk-major, taps innermost
  const int algo = d->algo;
  if (desc_ok) {
    const int KT = K / bk;
    const long tiles64 = (long)cdiv(M, 64) * cdiv(d->Cout, 64);
    // LDS-DMA 256x128 tile: wherever the large register-staged tiles were chosen, measured 59 vs 74 us on the dominant 3x3 and equal or
    // better on the 1x1 shapes (tools/dma_bench.py)
    const bool dma_ok = bf && !f32o && KT >= 3 && d->KH <= 3 && d->KW <= 3;
    // split-K over workgroups (slabs + ordered reduce): forced by the caller, or chosen for few-tile / long-K launches (splitk_factor)
    if (algo == L2S_ALGO_WS64_STAMPED && bf && !f32o && tile == 64) return PLAN_WS64;   // instrumented build (tools/ws64_stamps.py)
    if (tile == 64 && (algo == L2S_ALGO_AUTO || algo == L2S_ALGO_STAGED) && splitk_factor(*d, KT, tiles64, bf && !f32o) > 1)
      return (bf && !f32o) ? PLAN_WS64_SPLITK : PLAN_GENERIC64;
    // persistent LDS-DMA tile: plain GEMMs (1x1 / stride 1) whose 256x128 tiles need several rounds of workgroups (layer4 on the RoIs, N >= 1024)
    const bool pdma_ok = bf && !f32o && ntaps == 1 && d->stride == 1 && d->pad == 0 && KT >= 3 && !(d->flags & (L2S_CONV_SCATTER | L2S_CONV_DECONV2X2)) &&
                         d->IH == d->OH && d->IW == d->OW && !(d->ldy & 3) && !(d->ldadd & 3) && !(d->ldref & 3) && !(d->Cout & 3) &&
                         M * d->ldy * 2 < (1L << 31) && (!d->add || M * d->ldadd * 2 < (1L << 31)) && (!d->ref || M * d->ldref * 2 < (1L << 31));
    // (on request only: measured 69 / 99 / 78 us against 65 / 92 / 84 us of the one-shot tiles on layer4's 1x1-out / downsample / downsample
    // data-gradient launches - the two wave groups write their sub-tiles out in different slots, and each waits for the other's stores)
    if (pdma_ok && algo == L2S_ALGO_PDMA) return PLAN_PDMA256;
    // 256x256 LDS-DMA tile: the same plain GEMMs when they are wide (N a multiple of 256, >= 1024) and tall enough for more than one round of
    // 256x128 tiles (layer4 on the RoIs: conv3, downsample and the data gradients of conv1 / downsample)
    const bool d256_ok = pdma_ok && d->Cout % 256 == 0 && d->Cout >= 1024 && M >= 4096 && !(d->ldy & 7) && !(d->Cout & 7) &&
                         !((uintptr_t)d->y & 15) && !(d->ldadd & 3) && !(d->ldref & 3);
    if (d256_ok && (algo == L2S_ALGO_DMA256 || (algo == L2S_ALGO_AUTO && l2s_knobs::dma256_auto))) return PLAN_DMA256X256;
    // 196-row tiles of the same pipeline (layer4 on 256 RoIs with N = 512: 256 workgroups of 13 row fragments instead of 196 of 16); plain row-major
    // bf16 outputs only (its epilogue is the LDS-staged one).  ON REQUEST ONLY: measured (round 5, tools/conv_bench.py, same box) 62.8 / 60.5 us
    // against 64.1 / 61.6 us for the 3x3 and 32.5 against 33.8 us for the 1x1 - 2-4 % for 19 % less work per CU, because with all 256 CUs multiplying
    // the chip holds a lower clock (the loop is power-bound) - and the step is 1 % SLOWER with it (213.8 against 216.1 img/s, x2):
    // the 60 CUs the 256-row grid leaves idle are where the caption branch and the other queues run.
    const bool d196_ok = dma_ok && !(d->flags & (L2S_CONV_DECONV2X2 | L2S_CONV_SCATTER)) && !(d->ldy & 7) && !(d->ldadd & 7) && !(d->ldref & 7) && !(d->Cout & 7) &&
                         !((uintptr_t)d->y & 15) && !((uintptr_t)d->add & 15) && !((uintptr_t)d->ref & 15) && !((uintptr_t)d->bias & 15) &&
                         M * d->ldy * 2 < (1L << 31) && (!d->add || M * d->ldadd * 2 < (1L << 31)) && (!d->ref || M * d->ldref * 2 < (1L << 31));
    if (d196_ok && algo == L2S_ALGO_DMA196) return PLAN_DMA196;
    if (dma_ok && (algo == L2S_ALGO_DMA || (algo == L2S_ALGO_AUTO && (tile == 224 || tile == 256)))) return PLAN_DMA256;
    if (dma_ok && algo == L2S_ALGO_DMA_STAMPED) return PLAN_DMA256_STAMPED;
    // patch tile: 3x3 / stride 1 / pad 1 on one map whose row fits the patch (W + 1 <= 128 halo pixels on either side of 128 outputs)
    const bool p3_ok = bf && !f32o && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1 && d->n_img == 1 && d->IH == d->OH && d->IW == d->OW &&
                       d->Cin % 32 == 0 && d->OW + 1 <= 128 && !(d->flags & (L2S_CONV_SCATTER | L2S_CONV_DECONV2X2)) && xb < (1L << 30) && !d->tile;
    if (p3_ok && (algo == L2S_ALGO_AUTO || algo == L2S_ALGO_PATCH)) {
      const long mt128 = cdiv(M, 128);
      const bool narrow = mt128 * cdiv(d->Cout, 32) <= 256;        // 32-channel tiles while they fit one round of workgroups (one per CU)
      const bool small_halo = d->OW + 1 <= 64;
      if (narrow || mt128 * cdiv(d->Cout, 64) <= 256 || algo == L2S_ALGO_PATCH)
        return narrow ? (small_halo ? PLAN_P3_32_256 : PLAN_P3_32_384) : (small_halo ? PLAN_P3_64_256 : PLAN_P3_64_384);
    }
    // K-split 64x64 tile with LDS-DMA fill: whole 128-channel pieces per tap.  Chosen for the 3x3 launches whose 64x64 tiles fit one round
    // of workgroups and that carry no ReLU-mask operand, i.e. forward launches: 13.8 -> 11.2 us (layer3), 61.6 -> 52.9 us (RPN); the 1x1
    // launches gain nothing (their K loop is a few slices), multi-round grids lose (one workgroup per CU), and in the backward pass its
    // 128 KiB of LDS per workgroup cannot start beside the weight-gradient workgroups (step 171.9 vs 174.8 img/s)
    const bool ks_ok = bf && !f32o && d->Cin % 128 == 0 && d->KH <= 3 && d->KW <= 3 && M * d->Cout * 4 < (1L << 31);
    if (ks_ok && algo == L2S_ALGO_KSPLIT) return PLAN_KS64;
    if (ks_ok && algo == L2S_ALGO_KSPLIT_D3) return PLAN_KS64_D3;
    if (ks_ok && algo == L2S_ALGO_AUTO && !d->ref && tile == 64 && ntaps == 9 && tiles64 <= 256 && K >= 1024) return PLAN_KS64;
    if (tile == 224) return PLAN_SP224;
    if (tile == 256) return PLAN_SP256;
    // a 128x128 grid of less than one round of workgroups (two per CU): half-width tiles fill the chip better and halve the epilogue
    // burst per workgroup (layer4 on the map conv3: 19 x 16 tiles, 18.5 -> 15.5 us; layer2 conv3 10.2 -> 8.8 us)
    if (tile == 128 && bf && !f32o && !d->tile && t128 < 512 && d->Cout % 64 == 0 && algo == L2S_ALGO_AUTO) return PLAN_RING128X64;
    if (tile == 128) return PLAN_RING128;
    if (bf && !f32o) {
      // wave-specialised 8-wave form (loaders + multipliers), bf16 output.  Two of its workgroups fit a CU (512 slots): a grid just above
      // a multiple of that pays a whole extra round of workgroups for its last tiles (layer3 conv3 / conv1 data gradient: 608 tiles of 4
      // slices take 9.3 us, ~3 of them for the 96 tiles of the second round: tools/ws64_stamps.py).  The 4-wave ring tile fits three per
      // CU (768 slots); with a short K loop fewer rounds beat the faster K loop: 27.8 -> 26.2 us per layer3 block forward, 32.6 -> 30.1
      // backward (tools/chain_bench.py), the step +1.5 % (same-box A/B)
      if (KT <= 8 && (tiles64 + 767) / 768 < (tiles64 + 511) / 512) return PLAN_RING64;
      return PLAN_WS64;
    }
    return PLAN_RING64;
  }
  if (tile >= 128) return PLAN_GENERIC128;       // K tails, >= 2 GiB operands, split-K with a workspace: the register-staged generic kernel
  return PLAN_GENERIC64;
}
extern "C" const char* l2s_conv_plan_name(const l2s_conv_desc* d, int dtype) { return PLAN_NAMES[conv_plan(d, dtype, nullptr)]; }

extern "C" int l2s_conv_igemm(const l2s_conv_desc* d, int dtype, hipStream_t stream) {
  bool tapin = false;
  const ConvPlan plan = conv_plan(d, dtype, &tapin);
  if (plan == PLAN_EINVAL) return L2S_EINVAL;
  const bool f32o = d->flags & L2S_CONV_OUT_F32;
  // working-set heuristic for the XCD tile order: per-XCD chunk along M keeps all of W + 1/8 of A in L2; if that does not
  // fit (~3 MiB), chunk along N instead (one W column block resident, A streamed)
  l2s_conv_desc dd = *d;
  if (dd.xcd_mode < 0 || dd.xcd_mode > 1) {
    const double esz = dtype == L2S_BF16 ? 2.0 : 4.0;
    const int K = d->KH * d->KW * d->Cin;
    const double wbytes = (double)d->Cout * K * esz, abytes = (double)d->n_img * d->IH * d->IW * d->Cin * esz;
    dd.xcd_mode = (wbytes + abytes / 8.0 <= 3.0 * 1024 * 1024) ? 0 : 1;
    // 1x1: chunks along M even where that sum is larger (round 4, tools/conv_bench.py 0 <mode>: layer4 on the RoIs conv1 39.0 -> 34.2 us, conv3's data
    // gradient 39.4 -> 36.1, downsample 74.5 -> 70.7 / 63.0 -> 56.3, layer4 on the map conv3's data gradient 19.9 -> 16.7; the 3x3 launches, whose
    // W is nine times larger per channel pair, keep the rule: 62.7 against 64.2 us with N-chunks)
    if (d->KH * d->KW == 1) dd.xcd_mode = 0;
  }
#define BYT(CALL_BF, CALL_F32) (dtype == L2S_BF16 ? (CALL_BF) : (CALL_F32))
#define OUT(NAME, T, ...) (f32o ? NAME<T, __VA_ARGS__, true>(dd, stream) : NAME<T, __VA_ARGS__, false>(dd, stream))
  switch (plan) {
    case PLAN_PDMA256: return launch_igemm_pdma<256, 128>(dd, stream);
    case PLAN_DMA256X256: return launch_igemm_dma256(dd, stream);
    case PLAN_DMA256: return launch_igemm_dma<256, 128, false>(dd, stream);
    case PLAN_DMA196: return launch_igemm_dma196(dd, stream);
    case PLAN_DMA256_STAMPED: return launch_igemm_dma<256, 128, true>(dd, stream);
    case PLAN_P3_32_256: return launch_igemm_p3<32, 256, 2>(dd, stream);
    case PLAN_P3_64_256: return launch_igemm_p3<64, 256, 3>(dd, stream);
    case PLAN_P3_32_384: return launch_igemm_p3<32, 384, 3>(dd, stream);
    case PLAN_P3_64_384: return launch_igemm_p3<64, 384, 2>(dd, stream);
    case PLAN_KS64: return launch_igemm_ks64<4>(dd, stream);
    case PLAN_KS64_D3: return launch_igemm_ks64<3>(dd, stream);
    case PLAN_WS64:
      if (d->algo == L2S_ALGO_WS64_STAMPED) return launch_igemm_ws64<3, 64, true>(dd, stream);
      return launch_igemm_ws64<3, 64>(dd, stream);
    case PLAN_WS64_SPLITK: {
      const int M = d->n_img * d->OH * d->OW, KT = d->KH * d->KW * d->Cin / 64;
      return launch_igemm_ws64<3, 64>(dd, stream, splitk_factor(dd, KT, (long)cdiv(M, 64) * cdiv(d->Cout, 64), true));
    }
    case PLAN_SP224:
      if (tapin) return BYT((f32o ? launch_igemm_sp<bf16_t, 224, 128, 2, 4, 2, true, true>(dd, stream) : launch_igemm_sp<bf16_t, 224, 128, 2, 4, 2, false, true>(dd, stream)),
                            (f32o ? launch_igemm_sp<float, 224, 128, 2, 4, 2, true, true>(dd, stream) : launch_igemm_sp<float, 224, 128, 2, 4, 2, false, true>(dd, stream)));
      return BYT((f32o ? launch_igemm_sp<bf16_t, 224, 128, 2, 4, 2, true, false>(dd, stream) : launch_igemm_sp<bf16_t, 224, 128, 2, 4, 2, false, false>(dd, stream)),
                 (f32o ? launch_igemm_sp<float, 224, 128, 2, 4, 2, true, false>(dd, stream) : launch_igemm_sp<float, 224, 128, 2, 4, 2, false, false>(dd, stream)));
    case PLAN_SP256:
      if (tapin) return BYT((f32o ? launch_igemm_sp<bf16_t, 256, 128, 4, 2, 2, true, true>(dd, stream) : launch_igemm_sp<bf16_t, 256, 128, 4, 2, 2, false, true>(dd, stream)),
                            (f32o ? launch_igemm_sp<float, 256, 128, 4, 2, 2, true, true>(dd, stream) : launch_igemm_sp<float, 256, 128, 4, 2, 2, false, true>(dd, stream)));
      return BYT((f32o ? launch_igemm_sp<bf16_t, 256, 128, 4, 2, 2, true, false>(dd, stream) : launch_igemm_sp<bf16_t, 256, 128, 4, 2, 2, false, false>(dd, stream)),
                 (f32o ? launch_igemm_sp<float, 256, 128, 4, 2, 2, true, false>(dd, stream) : launch_igemm_sp<float, 256, 128, 4, 2, 2, false, false>(dd, stream)));
    case PLAN_RING128X64: return launch_igemm_ring<bf16_t, 128, 64, 2, 2, 2, false>(dd, stream);
    case PLAN_RING128: return BYT(OUT(launch_igemm_ring, bf16_t, 128, 128, 2, 2, 2), OUT(launch_igemm_ring, float, 128, 128, 2, 2, 2));
    // 64x64 ring tile: depth 2 with the fragment reads ahead of the LDS fill for bf16 (fp32-output launches), depth 4 for f32
    case PLAN_RING64: return BYT(OUT(launch_igemm_ring, bf16_t, 64, 64, 2, 2, 2), OUT(launch_igemm_ring, float, 64, 64, 2, 2, 4));
    case PLAN_GENERIC128: return BYT(OUT(launch_igemm, bf16_t, 128, 128), OUT(launch_igemm, float, 128, 128));
    case PLAN_GENERIC64: return BYT(OUT(launch_igemm, bf16_t, 64, 64), OUT(launch_igemm, float, 64, 64));
    default: return L2S_EINVAL;
  }
#undef BYT
#undef OUT
}

extern "C" int l2s_roialign_block0_fwd(const l2s_roi_block0_desc* d, hipStream_t stream) {
  if (!d || !d->feat || !d->rois || !d->w1 || !d->w2 || !d->pooled || !d->y1 || !d->y2) return L2S_EINVAL;
  const int PP = d->P * d->P;
  // one RoI per workgroup: the crop must fit LDS next to the weight ring, its rows are whole 128-byte K slices, the outputs whole 128-column tiles
  if (d->R <= 0 || PP < 1 || PP > 64 || d->C % 128 || d->N1 % 128 || d->N2 % 128 || d->N1 <= 0 || d->N2 <= 0) return L2S_EINVAL;
  if ((size_t)PP * d->C * 2 + 64 * sizeof(RoiTaps) > 160 * 1024) return L2S_EINVAL;
  if ((long)d->H * d->W * d->C * 2 >= (1L << 31) || (long)d->N1 * d->C * 2 >= (1L << 31) || (long)d->N2 * d->C * 2 >= (1L << 31)) return L2S_EINVAL;
  return launch_roialign_block0(*d, stream);
}
