// Weight gradient of a convolution on CDNA4 matrix cores (gfx950), NHWC activations:
//
//   dW[co][tap][ci] += sum_m dY[m][co] * X(m, tap)[ci]          (m = output pixel)
//
// Replaces cuDNN's backward-filter behind autograd of nn.Conv2d / nn.Linear / nn.ConvTranspose2d in the reference
// (pyutils/mask-faster-rcnn/lib/nets/resnet_v1_cycle_res5_2.py:83-88,324-335, network_cycle_res5_2.py:236-251,279-301).
//
// The contraction runs over pixels and both operands are pixel-major in HBM, so a K slice is 32 pixels (bf16; 16 in f32
// verification mode) staged global -> register ring -> LDS with rows = pixels; the bf16 MFMA fragments come out of LDS
// through ds_read_b64_tr_b16 (a 4-pixel x 16-channel block per 16 lanes, transposed by the hardware).
//
// Two things differ from a per-tap GEMM:
//   * TX = 3 (3x3, stride 1, pad 1): a workgroup owns (co tile, ci tile, filter ROW ky) and accumulates the three taps
//     kx = 0..2 together.  Pixels are walked in a virtual layout with one zero column appended to every image row
//     (width OW + 1); the X image in LDS holds the slice's pixels shifted by -1 .. +32, and tap kx reads it at row offset kx:
//     a shift that runs off a row end lands on the zero column, so no per-tap masking exists.  dY is staged once per slice
//     instead of once per tap and X 34/32 times instead of three times: a third of the L2 -> LDS traffic per MAC.
//   * no floating-point atomics: with split-K (pixels cut into `split` ranges so that a small-M layer still fills the chip)
//     every workgroup stores its partial tile into its own slab of a workspace, and a second launch adds the slabs to dW in
//     a fixed order.  The result is bit-identical from run to run.  split == 1 adds into dW directly.
#include "common.h"
#include "../../include/lang2seg_hip.h"
#include <stdlib.h>

namespace {

constexpr unsigned OOR = 0x80000000u;

template <typename T> struct WGT;
template <> struct WGT<bf16_t> { static constexpr int BKP = 32; static constexpr int PADB = 32; };   // row skew 8 dwords: tr reads conflict-free
template <> struct WGT<float> { static constexpr int BKP = 16; static constexpr int PADB = 64; };

struct wg_geom {
  int Wv;            // virtual row width (OW, or OW + 1 with the zero column)
  int Mv;            // virtual pixels
  int sa, sb, sc;    // BKP = sa * OH * Wv + sb * Wv + sc
  int split;         // pixel ranges
  int cblocks;       // ci tiles
  long slab;         // floats per slab (= Cout * KH * KW * Cin); 0: add into dW directly
  float* ws;
};

template <typename T, int BM, int BN, int TX, int D>
__global__ __launch_bounds__(256) void wgrad_kernel(const l2s_wgrad_desc p, const wg_geom g) {
  constexpr int ES = (int)sizeof(T);
  constexpr int VE = 16 / ES;
  constexpr int BKP = WGT<T>::BKP;
  constexpr int RB = BKP + TX - 1;                       // rows of the X image
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 16, TN = WN / 16;
  constexpr int LRA = BM * ES + WGT<T>::PADB, LRB = BN * ES + WGT<T>::PADB;
  constexpr int VPA = BM / VE, VPB = BN / VE;            // 16-byte vectors per row
  constexpr int NVA = (BKP * VPA + 255) / 256, NVB = (RB * VPB + 255) / 256;
  constexpr int BUF = BKP * LRA + RB * LRB;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int co0 = blockIdx.x * BM;
  const int tg = blockIdx.y / g.cblocks, ci0 = (blockIdx.y - tg * g.cblocks) * BN;
  const int ky = TX == 3 ? tg : tg / p.KW, kx = TX == 3 ? 0 : tg - ky * p.KW;
  const auto rdy = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, 0x7FFFFFFF, 0x00020000);
  const auto rxx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0x7FFFFFFF, 0x00020000);

  const int nslices = (g.Mv + BKP - 1) / BKP;
  const int per = (nslices + g.split - 1) / g.split;
  const int s_begin = blockIdx.z * per;
  const int s_end = min(nslices, s_begin + per);
  const int NS = max(0, s_end - s_begin);                // (an empty range still has to write its zero slab)

  // ---- loader state: every thread owns NVA vectors of dY and NVB vectors of X per slice; their (image, row, column) in the
  // virtual layout advance by BKP pixels per slice with add / compare steps ----
  const int ohwv = p.OH * g.Wv;
  int an[NVA], ay[NVA], ax[NVA], ac[NVA]; bool aok[NVA];
  int bn[NVB], by[NVB], bx[NVB], bc[NVB]; bool bok[NVB];
  const int pb0 = s_begin * BKP;
#pragma unroll
  for (int j = 0; j < NVA; ++j) {
    const int v = tid + j * 256, r = v / VPA, c = v - r * VPA;
    aok[j] = r < BKP && (co0 + c * VE) < p.Cout;
    ac[j] = co0 + c * VE;
    const int pix = pb0 + min(r, BKP - 1);
    an[j] = pix / ohwv; const int rem = pix - an[j] * ohwv; ay[j] = rem / g.Wv; ax[j] = rem - ay[j] * g.Wv;
  }
#pragma unroll
  for (int j = 0; j < NVB; ++j) {
    const int v = tid + j * 256, r = v / VPB, c = v - r * VPB;
    bok[j] = r < RB && (ci0 + c * VE) < p.Cin;
    bc[j] = ci0 + c * VE;
    const int pix = pb0 + min(r, RB - 1) - (TX == 3 ? 1 : 0);       // TX == 3: image row r holds virtual pixel base - 1 + r
    if (pix < 0) { bn[j] = 0; by[j] = 0; bx[j] = -1; }
    else { bn[j] = pix / ohwv; const int rem = pix - bn[j] * ohwv; by[j] = rem / g.Wv; bx[j] = rem - by[j] * g.Wv; }
  }
  auto issue = [&](uint4 (&ra)[NVA], uint4 (&rb)[NVB]) {
#pragma unroll
    for (int j = 0; j < NVA; ++j) {
      const bool ok = aok[j] && ax[j] < p.OW && an[j] < p.n_img;
      const unsigned o = ok ? (unsigned)((((an[j] * p.OH + ay[j]) * p.OW + ax[j]) * p.lddy + ac[j]) * ES) : OOR;
      ra[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rdy, o, 0, 0));
      ax[j] += g.sc; if (ax[j] >= g.Wv) { ax[j] -= g.Wv; ++ay[j]; }
      ay[j] += g.sb; if (ay[j] >= p.OH) { ay[j] -= p.OH; ++an[j]; }
      an[j] += g.sa;
    }
#pragma unroll
    for (int j = 0; j < NVB; ++j) {
      const int iy = by[j] * p.stride - p.pad + ky;
      const int ix = TX == 3 ? bx[j] : bx[j] * p.stride - p.pad + kx;
      const bool ok = bok[j] && bn[j] < p.n_img && bx[j] >= 0 && bx[j] < p.OW && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
      const unsigned o = ok ? (unsigned)((((bn[j] * p.IH + iy) * p.IW + ix) * p.ldx + bc[j]) * ES) : OOR;
      rb[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rxx, o, 0, 0));
      bx[j] += g.sc; if (bx[j] >= g.Wv) { bx[j] -= g.Wv; ++by[j]; }
      by[j] += g.sb; if (by[j] >= p.OH) { by[j] -= p.OH; ++bn[j]; }
      bn[j] += g.sa;
    }
  };
  auto store_slice = [&](int buf, const uint4 (&ra)[NVA], const uint4 (&rb)[NVB]) {
    char* a = smem + buf * BUF;
    char* b = a + BKP * LRA;
#pragma unroll
    for (int j = 0; j < NVA; ++j) {
      const int v = tid + j * 256, r = v / VPA, c = v - r * VPA;
      if ((BKP * VPA) % 256 == 0 || r < BKP) *(uint4*)(a + r * LRA + c * 16) = ra[j];
    }
#pragma unroll
    for (int j = 0; j < NVB; ++j) {
      const int v = tid + j * 256, r = v / VPB, c = v - r * VPB;
      if ((RB * VPB) % 256 == 0 || r < RB) *(uint4*)(b + r * LRB + c * 16) = rb[j];
    }
  };

  f32x4 acc[TX][TM][TN];
#pragma unroll
  for (int t = 0; t < TX; ++t)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[t][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fg = lane >> 4;
  // The MFMA is issued with X as the row operand and dY as the column operand: D[row = ci][col = co], so a lane ends up
  // with 4 consecutive input channels of one output channel -> 16-byte accesses to dW[co][tap][ci .. ci+3].
  auto compute = [&](int cur) {
    const char* a = smem + cur * BUF;
    const char* b = a + BKP * LRA;
    if constexpr (sizeof(T) == 2) {
      // k mapping inside the 32-pixel slice: lane group g, half h, element e  <->  pixel 16 h + 4 g + e (both operands)
      const int trow = 4 * fg + ((lane >> 2) & 3), tcol = 8 * (lane & 3);
      uint4 fa[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const char* q = a + trow * LRA + (wm * WM + i * 16) * 2 + tcol;
        s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(q));
        s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(q + 16 * LRA));
        fa[i] = __builtin_bit_cast(uint4, (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
      }
#pragma unroll
      for (int t = 0; t < TX; ++t) {
        uint4 fb[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const char* q = b + (trow + t) * LRB + (wn * WN + j * 16) * 2 + tcol;
          s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(q));
          s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(q + 16 * LRB));
          fb[j] = __builtin_bit_cast(uint4, (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[j]), __builtin_bit_cast(bf16x8, fa[i]), acc[t][i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int ks = 0; ks < BKP / 4; ++ks) {
        float fa[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *(const float*)(a + (ks * 4 + fg) * LRA + (wm * WM + i * 16 + fr) * 4);
#pragma unroll
        for (int t = 0; t < TX; ++t) {
          float fb[TN];
#pragma unroll
          for (int j = 0; j < TN; ++j) fb[j] = *(const float*)(b + (ks * 4 + fg + t) * LRB + (wn * WN + j * 16 + fr) * 4);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j], fa[i], acc[t][i][j], 0, 0, 0);
        }
      }
    }
  };

  // register ring: set s holds slice k with k % D == s; iteration t: barrier -> ds_write slice t+1 -> issue slice t+1+D -> MFMAs on t
  uint4 qa[D][NVA], qb[D][NVB];
  if (NS > 0) {
#pragma unroll
    for (int s = 0; s < D; ++s)
      if (s < NS) issue(qa[s], qb[s]);
    store_slice(0, qa[0], qb[0]);
    if (D < NS) issue(qa[0], qb[0]);
    int t0 = 0;
    for (; t0 + 2 * D <= NS; t0 += D) {
#pragma unroll
      for (int s = 0; s < D; ++s) {
        const int t = t0 + s;
        const int nxt = (s + 1) % D;
        __syncthreads();
        store_slice((t + 1) & 1, qa[nxt], qb[nxt]);
        issue(qa[nxt], qb[nxt]);
        compute(t & 1);
      }
    }
#pragma unroll
    for (int s = 0; s < 2 * D; ++s) {
      const int t = t0 + s;
      if (t < NS) {
        const int nxt = (s + 1) % D;
        __syncthreads();
        if (t + 1 < NS) {
          store_slice((t + 1) & 1, qa[nxt], qb[nxt]);
          if (t + 1 + D < NS) issue(qa[nxt], qb[nxt]);
        }
        compute(t & 1);
      }
    }
  }
  // ---- epilogue: partial tile -> own slab (plain stores), or dW += tile when the pixels are not split ----
  const long Kw = (long)p.KH * p.KW * p.Cin;
  float* out = g.slab ? g.ws + (long)blockIdx.z * g.slab : p.dw;
#pragma unroll
  for (int t = 0; t < TX; ++t) {
    const int tap = TX == 3 ? ky * p.KW + t : tg;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int co = co0 + wm * WM + i * 16 + fr;
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int ci = ci0 + wn * WN + j * 16 + fg * 4;
        if (co < p.Cout && ci < p.Cin) {                 // Cin % 4 == 0
          float4* q = (float4*)(out + (long)co * Kw + (long)tap * p.Cin + ci);
          float4 v = make_float4(acc[t][i][j][0], acc[t][i][j][1], acc[t][i][j][2], acc[t][i][j][3]);
          if (!g.slab) { const float4 o = *q; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
          *q = v;
        }
      }
    }
  }
}

// dW[e] += slab_0[e] + slab_1[e] + ... in that order (a fixed summation tree: bit-reproducible)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(float* __restrict__ dw, const float* __restrict__ ws, long n4, long slab, int split) {
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n4; e += (long)gridDim.x * blockDim.x) {
    float4 s = ((const float4*)ws)[e];
    for (int k = 1; k < split; ++k) {
      const float4 v = ((const float4*)(ws + (long)k * slab))[e];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    float4 o = ((float4*)dw)[e];
    o.x += s.x; o.y += s.y; o.z += s.z; o.w += s.w;
    ((float4*)dw)[e] = o;
  }
}

struct wg_plan { int tile_m, tile_n, tx, split; };

// tile / split choice.  Tiles: 3x3 stride-1 layers share the three taps of a filter row (TX = 3); the co tile is 128 wide when the
// layer is large enough to fill the chip that way.  split: enough pixel ranges for ~2 workgroups per CU, at least 8 slices each.
wg_plan plan_wgrad(const l2s_wgrad_desc& d, int dtype, size_t ws_bytes) {
  const int taps = d.KH * d.KW;
  const bool row3 = d.KH == 3 && d.KW == 3 && d.stride == 1 && d.pad == 1 && d.OH == d.IH && d.OW == d.IW;
  static const int tx_on = [] { const char* e = getenv("L2S_WGRAD_TX3"); return e ? atoi(e) : 1; }();
  wg_plan pl;
  pl.tx = (row3 && tx_on) ? 3 : 1;
  const long M = (long)d.n_img * d.OH * d.OW;
  const bool big = M >= 8192 && d.Cout >= 512 && d.Cin >= 512;
  if (pl.tx == 3) { pl.tile_m = (d.tile == 128 || (!d.tile && big)) ? 128 : 64; pl.tile_n = 64; }
  else { pl.tile_m = pl.tile_n = d.tile ? (d.tile == 128 ? 128 : 64) : ((big && taps == 1) ? 128 : 64); }
  const long tiles = (long)cdiv(d.Cout, pl.tile_m) * cdiv(d.Cin, pl.tile_n) * (pl.tx == 3 ? d.KH : taps);
  const int bkp = dtype == L2S_BF16 ? 32 : 16;
  const long Mv = (long)d.n_img * d.OH * (d.OW + (pl.tx == 3 ? 1 : 0));
  const int slices = cdiv(Mv, bkp);
  int split = d.split_k;
  if (split <= 0) {
    static const int min_wg = [] { const char* e = getenv("L2S_WGRAD_MINWG"); return e ? atoi(e) : 448; }();
    split = (int)((min_wg + tiles - 1) / tiles);
    const int maxs = slices / 8 > 0 ? slices / 8 : 1;
    if (split > maxs) split = maxs;
  }
  if (split > slices) split = slices;
  if (split < 1) split = 1;
  if (split > 64) split = 64;
  const long slab = (long)d.Cout * taps * d.Cin * 4;
  if (split > 1 && (long)split * slab > (long)ws_bytes) split = ws_bytes >= (size_t)(2 * slab) ? (int)(ws_bytes / slab) : 1;
  pl.split = split;
  return pl;
}

template <typename T, int BM, int BN, int TX, int D>
int launch_wgrad(const l2s_wgrad_desc& d, const wg_plan& pl, hipStream_t st) {
  constexpr int ES = (int)sizeof(T), BKP = WGT<T>::BKP, RB = BKP + TX - 1;
  constexpr int LRA = BM * ES + WGT<T>::PADB, LRB = BN * ES + WGT<T>::PADB;
  wg_geom g;
  g.Wv = d.OW + (TX == 3 ? 1 : 0);
  g.Mv = d.n_img * d.OH * g.Wv;
  const int ohwv = d.OH * g.Wv;
  g.sa = BKP / ohwv; g.sb = (BKP - g.sa * ohwv) / g.Wv; g.sc = BKP - g.sa * ohwv - g.sb * g.Wv;
  g.split = pl.split;
  g.cblocks = cdiv(d.Cin, BN);
  g.slab = pl.split > 1 ? (long)d.Cout * d.KH * d.KW * d.Cin : 0;
  g.ws = d.ws;
  dim3 grid(cdiv(d.Cout, BM), (TX == 3 ? d.KH : d.KH * d.KW) * g.cblocks, pl.split);
  const size_t lds = 2 * (size_t)(BKP * LRA + RB * LRB);
  L2S_LAUNCH((wgrad_kernel<T, BM, BN, TX, D>), grid, dim3(256), lds, st, d, g);
  if (pl.split > 1) {
    const long n4 = g.slab / 4;
    long gb = (n4 + 255) / 256; if (gb > 4096) gb = 4096;
    float* dw = d.dw; const float* ws = d.ws; const long slab = g.slab; const int split = pl.split;
    L2S_LAUNCH(wgrad_reduce_kernel, dim3((int)gb), dim3(256), 0, st, dw, ws, n4, slab, split);
  }
  return l2s_check_launch();
}

template <typename T>
int dispatch_wgrad(const l2s_wgrad_desc& d, const wg_plan& pl, hipStream_t st) {
  if (pl.tx == 3) return pl.tile_m == 128 ? launch_wgrad<T, 128, 64, 3, 3>(d, pl, st) : launch_wgrad<T, 64, 64, 3, 4>(d, pl, st);
  return pl.tile_m == 128 ? launch_wgrad<T, 128, 128, 1, 3>(d, pl, st) : launch_wgrad<T, 64, 64, 1, 4>(d, pl, st);
}

}  // namespace

extern "C" size_t l2s_wgrad_ws_bytes(const l2s_wgrad_desc* d, int dtype) {
  if (!d) return 0;
  const wg_plan pl = plan_wgrad(*d, dtype, (size_t)1 << 40);
  return pl.split > 1 ? (size_t)pl.split * d->Cout * d->KH * d->KW * d->Cin * 4 : 0;
}

extern "C" int l2s_conv_wgrad(const l2s_wgrad_desc* d, int dtype, hipStream_t stream) {
  if (!d || !d->dy || !d->x || !d->dw) return L2S_EINVAL;
  const int ve = dtype == L2S_BF16 ? 8 : 4;
  if (d->lddy % ve || d->ldx % ve || d->Cin % ve || d->Cout % ve) return L2S_EINVAL;
  const long M = (long)d->n_img * d->OH * d->OW;
  if (M >= (1 << 24)) return L2S_EINVAL;
  const long esz = dtype == L2S_BF16 ? 2 : 4;
  const long xb = (long)d->n_img * d->IH * d->IW * d->ldx * esz, yb = M * d->lddy * esz;
  if (xb >= (1L << 31) || yb >= (1L << 31)) return L2S_EINVAL;        // 32-bit buffer offsets
  if ((long)d->Cout * d->KH * d->KW * d->Cin % 4) return L2S_EINVAL;
  const wg_plan pl = plan_wgrad(*d, dtype, d->ws ? d->ws_bytes : 0);
  if (dtype == L2S_BF16) return dispatch_wgrad<bf16_t>(*d, pl, stream);
  if (dtype == L2S_F32) return dispatch_wgrad<float>(*d, pl, stream);
  return L2S_EINVAL;
}
