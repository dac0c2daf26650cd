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
// What shapes the design:
//   * The layers of this network are small (a 38x63 map is 2394 pixels): one weight gradient alone has 50-150 output tiles and
//     a pixel loop of ~75 slices, i.e. it can neither fill 256 CUs nor hide the ~2 us memory latency of its own loads, and
//     cutting the pixels into split-K ranges pays for the parallelism with partial-sum traffic (fp32 atomics in round 1).
//     Nothing needs a weight gradient before the optimiser, though, so the step DEFERS them: l2s_conv_wgrad_grouped runs the
//     weight gradients of a whole backward stage (e.g. 8 bottlenecks = 24 convolutions) as ONE launch of a few thousand
//     workgroups, each owning a full output tile over all pixels: the chip is full, several workgroups per CU hide each
//     other's latency, there is no split-K, no atomic and no partial-sum traffic, and the result is bit-reproducible.
//   * A tensor that is used twice in the step (layer4 on the RoIs and on the whole map, NET:415-435) is ONE problem with two
//     pixel segments: the workgroup walks both and stores its tile once.
//   * TX = 3 (3x3, stride 1, pad 1): a workgroup owns (co tile, ci tile, filter ROW ky) and accumulates the three taps
//     kx = 0..2 together.  Pixels are walked in a virtual layout with one zero column appended to every image row
//     (width OW + 1); the X image in LDS holds the slice's pixels shifted by -1 .. +32, and tap kx reads it at row offset kx:
//     a shift that runs off a row end lands on the zero column, so no per-tap masking exists.  dY is staged once per slice
//     instead of once per tap and X 34/32 times instead of three times: a third of the L2 -> LDS traffic per MAC.
//   * l2s_conv_wgrad (one problem per launch) remains for callers outside a step; with a workspace it may split the pixels:
//     every workgroup then stores its partial tile into its own slab and a second launch adds the slabs to dW in a fixed
//     order (still no floating-point atomics).
#include "common.h"
#include "../../include/lang2seg_hip.h"
#include "wgrad_internal.h"
#include <stdlib.h>

namespace {

constexpr unsigned OOR = 0x80000000u;

template <typename T> struct WGT;
template <> struct WGT<bf16_t> { static constexpr int BKP = 32; static constexpr int PADB = 32; };   // row skew 8 dwords: tr reads conflict-free
template <> struct WGT<float> { static constexpr int BKP = 16; static constexpr int PADB = 64; };

typedef l2s_wgrad_prob wgp;

// One output tile (co tile, ci tile, tap or filter row) of problem p over the slices [s_begin, s_end) of segment `seg`'s pixels
// (s_end < 0: all slices of every segment).  out: where the tile goes (dW, or a split-K slab); accumulate: out += tile.
template <typename T, int BM, int BN, int TX, int D, int KSTEP = 1, int WGM = 2, int WGN = 2>
__device__ __forceinline__ void wgrad_tile(const wgp& p, int co0, int ci0, int tg, int split_idx, int nsplit, float* out, bool accumulate, char* smem) {
  constexpr int ES = (int)sizeof(T);
  constexpr int VE = 16 / ES;
  constexpr int BKP0 = WGT<T>::BKP;                      // pixels of one MFMA k step
  constexpr int BKP = BKP0 * KSTEP;                      // pixels per slice (one barrier)
  constexpr int RB = BKP + TX - 1;                       // rows of the X image
  constexpr int NT = 64 * WGM * WGN;                     // WGM x WGN waves, wave tile WM x WN
  constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 16, TN = WN / 16;
  constexpr int LRA = BM * ES + WGT<T>::PADB, LRB = BN * ES + WGT<T>::PADB;
  constexpr int VPA = BM / VE, VPB = BN / VE;            // 16-byte vectors per row
  // loader: TPR threads share one pixel row of the slice and own NVA / NVB CONSECUTIVE vectors of it, so a thread carries one
  // (image, row, column) state per operand instead of one per vector (the per-vector form spent ~190 VALU + ~125 SALU instructions
  // per slice and wave on index arithmetic against 32 MFMAs).  TX == 3: the two extra X rows belong to the first 2 TPR threads.
  constexpr int TPR = NT / BKP;
  static_assert(NT % BKP == 0 && VPA % TPR == 0 && VPB % TPR == 0, "loader mapping");
  constexpr int NVA = VPA / TPR, NVB = VPB / TPR;
  constexpr int BUF = BKP * LRA + RB * LRB;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int ky = TX == 3 ? tg : tg / p.KW, kx = TX == 3 ? 0 : tg - ky * p.KW;
  const int fr = lane & 15, fg = lane >> 4;

  f32x4 acc[TX][TM][TN];
#pragma unroll
  for (int t = 0; t < TX; ++t)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) acc[t][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // The MFMA is issued with X as the row operand and dY as the column operand: D[row = ci][col = co], so a lane ends up
  // with 4 consecutive input channels of one output channel -> 16-byte accesses to dW[co][tap][ci .. ci+3].
  auto compute = [&](int cur) {
    const char* a = smem + cur * BUF;
    const char* b = a + BKP * LRA;
    if constexpr (sizeof(T) == 2) {
      // k mapping inside the 32-pixel slice: lane group g, half h, element e  <->  pixel 16 h + 4 g + e (both operands)
      const int trow0 = 4 * fg + ((lane >> 2) & 3), tcol = 8 * (lane & 3);
#pragma unroll
      for (int h = 0; h < KSTEP; ++h) {
        const int trow = trow0 + h * 32;
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

#pragma unroll
  for (int seg = 0; seg < L2S_WGRAD_MAX_SEG; ++seg) {    // (unrolled: the per-segment fields are read with constant indices, p stays in SGPRs)
    if (seg >= p.nseg) break;
    // ---- geometry of this pixel segment (uniform) ----
    const int n_img = p.n_img[seg], IH = p.IH[seg], IW = p.IW[seg], OH = p.OH[seg], OW = p.OW[seg], lddy = p.lddy[seg], ldx = p.ldx[seg];
    const int Wv = OW + (TX == 3 ? 1 : 0);               // virtual row width (with the zero column)
    const int ohwv = OH * Wv;
    const int Mv = n_img * ohwv;
    const int sa = BKP / ohwv, sb = (BKP - sa * ohwv) / Wv, sc = BKP - sa * ohwv - sb * Wv;   // BKP = sa*OH*Wv + sb*Wv + sc
    const int nslices = (Mv + BKP - 1) / BKP;
    const int per = (nslices + nsplit - 1) / nsplit;
    const int s_begin = split_idx * per;
    const int NS = max(0, min(nslices, s_begin + per) - s_begin);
    if (NS == 0) continue;
    const auto rdy = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy[seg], 0, 0x7FFFFFFF, 0x00020000);
    const auto rxx = __builtin_amdgcn_make_buffer_rsrc((void*)p.x[seg], 0, 0x7FFFFFFF, 0x00020000);

    // ---- loader state: the thread's pixel row of the slice (dY, X) and, for the first 2 TPR threads of a filter-row tile, one of the
    // two extra X rows; (image, row, column) in the virtual layout advance by BKP pixels per slice with add / compare steps ----
    const int lrow = tid / TPR, lq = tid - lrow * TPR;
    const int pb0 = s_begin * BKP;
    int an, ay, ax, bn, by, bx, tn = 0, ty = 0, tx = -1;
    const bool tail = TX == 3 && tid < 2 * TPR;
    bool acok[NVA], bcok[NVB];
#pragma unroll
    for (int j = 0; j < NVA; ++j) acok[j] = (co0 + (lq * NVA + j) * VE) < p.Cout;
#pragma unroll
    for (int j = 0; j < NVB; ++j) bcok[j] = (ci0 + (lq * NVB + j) * VE) < p.Cin;
    const unsigned acol = (unsigned)((co0 + lq * NVA * VE) * ES), bcol = (unsigned)((ci0 + lq * NVB * VE) * ES);
    {
      const int pix = pb0 + lrow;
      an = pix / ohwv; const int rem = pix - an * ohwv; ay = rem / Wv; ax = rem - ay * Wv;
    }
    auto place = [&](int pix, int& n, int& y, int& x) {            // TX == 3: image row r holds virtual pixel base - 1 + r
      if (pix < 0) { n = 0; y = 0; x = -1; }
      else { n = pix / ohwv; const int rem = pix - n * ohwv; y = rem / Wv; x = rem - y * Wv; }
    };
    place(pb0 + lrow - (TX == 3 ? 1 : 0), bn, by, bx);
    if (TX == 3) place(pb0 + BKP + lrow - 1, tn, ty, tx);           // (only used by the `tail` threads: lrow is 0 or 1 there)
    auto advance = [&](int& n, int& y, int& x) {
      x += sc; if (x >= Wv) { x -= Wv; ++y; }
      y += sb; if (y >= OH) { y -= OH; ++n; }
      n += sa;
    };
    auto xoff = [&](int n, int y, int x) -> unsigned {
      const int iy = y * p.stride - p.pad + ky;
      const int ix = TX == 3 ? x : x * p.stride - p.pad + kx;
      const bool ok = n < n_img && x >= 0 && x < OW && iy >= 0 && iy < IH && ix >= 0 && ix < IW;
      return ok ? (unsigned)(((n * IH + iy) * IW + ix) * ldx * ES) + bcol : OOR;
    };
    auto issue = [&](uint4 (&ra)[NVA], uint4 (&rb)[NVB], uint4 (&rt)[NVB]) {
      {
        const bool ok = ax < OW && an < n_img;
        const unsigned o = ok ? (unsigned)((((an * OH + ay) * OW + ax) * lddy) * ES) + acol : OOR;
#pragma unroll
        for (int j = 0; j < NVA; ++j) ra[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rdy, acok[j] ? o + j * 16 : OOR, 0, 0));
        advance(an, ay, ax);
      }
      {
        const unsigned o = xoff(bn, by, bx);
#pragma unroll
        for (int j = 0; j < NVB; ++j) rb[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rxx, bcok[j] ? o + j * 16 : OOR, 0, 0));
        advance(bn, by, bx);
      }
      if (TX == 3) {
        const unsigned o = tail ? xoff(tn, ty, tx) : OOR;
#pragma unroll
        for (int j = 0; j < NVB; ++j) rt[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rxx, bcok[j] ? o + j * 16 : OOR, 0, 0));
        advance(tn, ty, tx);
      }
    };
    auto store_slice = [&](int buf, const uint4 (&ra)[NVA], const uint4 (&rb)[NVB], const uint4 (&rt)[NVB]) {
      char* a = smem + buf * BUF + lrow * LRA + lq * NVA * 16;
      char* b = smem + buf * BUF + BKP * LRA + lrow * LRB + lq * NVB * 16;
#pragma unroll
      for (int j = 0; j < NVA; ++j) *(uint4*)(a + j * 16) = ra[j];
#pragma unroll
      for (int j = 0; j < NVB; ++j) *(uint4*)(b + j * 16) = rb[j];
      if (TX == 3 && tail) {
#pragma unroll
        for (int j = 0; j < NVB; ++j) *(uint4*)(b + BKP * LRB + j * 16) = rt[j];
      }
    };

    // register ring: set s holds slice k with k % D == s; iteration t: barrier -> ds_write slice t+1 -> issue slice t+1+D -> MFMAs on t
    uint4 qa[D][NVA], qb[D][NVB], qt[D][TX == 3 ? NVB : 1];
    auto iss = [&](int s_) { if constexpr (TX == 3) issue(qa[s_], qb[s_], qt[s_]); else issue(qa[s_], qb[s_], qb[s_]); };
    auto sto = [&](int buf, int s_) { if constexpr (TX == 3) store_slice(buf, qa[s_], qb[s_], qt[s_]); else store_slice(buf, qa[s_], qb[s_], qb[s_]); };
#pragma unroll
    for (int s = 0; s < D; ++s)
      if (s < NS) iss(s);
    __syncthreads();                                     // (the previous segment's last slice may still be read)
    sto(0, 0);
    if (D < NS) iss(0);
    int t0 = 0;
    for (; t0 + 2 * D <= NS; t0 += D) {
#pragma unroll
      for (int s = 0; s < D; ++s) {
        const int t = t0 + s;
        const int nxt = (s + 1) % D;
        __syncthreads();
        sto((t + 1) & 1, nxt);
        iss(nxt);
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
          sto((t + 1) & 1, nxt);
          if (t + 1 + D < NS) iss(nxt);
        }
        compute(t & 1);
      }
    }
  }
  // ---- epilogue ----
  const long Kw = (long)p.KH * p.KW * p.Cin;
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
          if (accumulate) { const float4 o = *q; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
          *q = v;
        }
      }
    }
  }
}

template <typename T, int BM, int BN, int TX, int KSTEP = 1> constexpr size_t wgrad_lds() {
  return 2 * (size_t)(WGT<T>::BKP * KSTEP * (BM * sizeof(T) + WGT<T>::PADB) + (WGT<T>::BKP * KSTEP + TX - 1) * (BN * sizeof(T) + WGT<T>::PADB));
}

// ---- one problem per launch: grid (co tiles, taps x ci tiles, split) ----
template <typename T, int BM, int BN, int TX, int D>
__global__ __launch_bounds__(256) void wgrad_kernel(const wgp p, int cblocks, float* ws, long slab) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tg = blockIdx.y / cblocks, ci0 = (blockIdx.y - tg * cblocks) * BN;
  float* out = slab ? ws + (long)blockIdx.z * slab : p.dw;
  wgrad_tile<T, BM, BN, TX, D>(p, blockIdx.x * BM, ci0, tg, blockIdx.z, gridDim.z, out, slab == 0, smem);
}


// ---- a whole backward stage per launch: problems in a device table, workgroup -> (problem, tile) through the tile prefix ----
struct wg_prefix { int n; int tile0[L2S_WGRAD_MAX_GROUP + 1]; };

template <typename T, int BM, int BN, int TX, int D, int KSTEP, int WGM = 2, int WGN = 2>
__global__ __launch_bounds__(64 * WGM * WGN) void wgrad_grouped_kernel(const wgp* __restrict__ tab, const wg_prefix pre, float* ws) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8), each with its own L2; XCD x takes the x-th
  // CONTIGUOUS eighth of the tile list, whose neighbours share the X tile (same ci tile and tap, consecutive co tiles / pixel ranges),
  // instead of every XCD fetching every operand tile.  The grid may be smaller than the tile list (l2s_wgrad_grid_cap): a workgroup
  // then walks tiles L, L + gridDim.x, ... (gridDim.x a multiple of 8 keeps a workgroup on its XCD's eighth).
  const int G = pre.tile0[pre.n];
  for (int L = blockIdx.x; L < G; L += gridDim.x) {
    int bid;
    {
      const int x = L & 7, slot = L >> 3, q = G >> 3, r = G & 7;
      bid = x * q + min(x, r) + slot;
    }
    int lo = 0, hi = pre.n;                                // tile0[lo] <= bid < tile0[hi]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (pre.tile0[mid] <= bid) lo = mid; else hi = mid; }
    const wgp p = tab[lo];                                 // uniform: scalar loads
    int t = bid - pre.tile0[lo];
    const int co_tiles = (p.Cout + BM - 1) / BM, ci_tiles = (p.Cin + BN - 1) / BN;
    const int split = p.split > 1 ? p.split : 1;
    // split index fastest, then co: consecutive workgroups (dealt round-robin over the XCDs) share the X tile and the tap
    const int sp = t % split; t /= split;
    const int cot = t % co_tiles, rest = t / co_tiles, cit = rest % ci_tiles, tg = rest / ci_tiles;
    float* out = split > 1 ? ws + p.ws_off + (long)sp * ((long)p.Cout * p.KH * p.KW * p.Cin) : p.dw;
    wgrad_tile<T, BM, BN, TX, D, KSTEP, WGM, WGN>(p, cot * BM, cit * BN, tg, sp, split, out, split == 1 && !(p.flags & 1), smem);
    __syncthreads();                                       // the next tile's first fill may not overtake this tile's last fragment reads
  }
}

// problems of a grouped launch whose pixels were split: dW[e] += slab_0[e] + slab_1[e] + ... (fixed order); grid (blocks, problems)
__global__ __launch_bounds__(256) void wgrad_reduce_grouped_kernel(const wgp* __restrict__ tab, const float* __restrict__ ws) {
  const wgp p = tab[blockIdx.y];
  if (p.split <= 1) return;
  const long slab = (long)p.Cout * p.KH * p.KW * p.Cin, n4 = slab / 4;
  const float* w0 = ws + p.ws_off;
  for (long e = blockIdx.x * (long)blockDim.x + threadIdx.x; e < n4; e += (long)gridDim.x * blockDim.x) {
    float4 s = ((const float4*)w0)[e];
    for (int k = 1; k < p.split; ++k) {
      const float4 v = ((const float4*)(w0 + (long)k * slab))[e];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    if (!(p.flags & 1)) { const float4 o = ((float4*)p.dw)[e]; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
    ((float4*)p.dw)[e] = s;
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

// ---- variants (tile, taps per workgroup): 0 = 64x64 per tap, 1 = 128x128 per tap, 2 = 64x64 filter row, 3 = 128(co)x64 filter row,
// 4 = 256x256 per tap with 8 waves, 5 = the LDS-DMA 128x128 filter-row tile with stream-K balancing (conv_wgrad_dma.hip)
// (4 and 5: bf16 grouped launches only; `tile` = 256 allows them) ----
// The choices below are constants of the product build (csrc/knobs.h):
//  * wgrad_row3_dma_wgs = 128: the stream-K launch of the LDS-DMA filter-row tile takes half the CUs (a workgroup owns its CU - 132 KB of LDS, 240 VGPRs
//    x 8 waves - and the other queues' launches need somewhere to run: 96 / 128 / 160 / 256 -> 190.1 / 189.1 / 189.0 / 185.5 img/s, same box x 3)
//  * wgrad_1x1_dma = 0: the LDS-DMA 256x256 tile of conv_wgrad_dma1.hip (variant 6) for the large 1x1 problems is built, tested, and no faster:
//    447-455 against 425-431 us alone on 120 CUs, 206.9 against 208.4 img/s in the step; both tiles move 32 KB per slice and CU in ~0.95 us
//  * wgrad_row3_wide = 1: 512+ channels on both sides (RPN's 3x3 on the map) take the filter-row tile from any pixel count (its stream-K launch needs no pixel split)
using namespace l2s_knobs;
int variant_of(int Cin, int Cout, int KH, int KW, int stride, int pad, int same_hw, long M, int tile) {
  const bool row3 = KH == 3 && KW == 3 && stride == 1 && pad == 1 && same_hw;
  const bool big = M >= 8192 && Cout >= 512 && Cin >= 512;
  if (row3 && tile == 256 && wgrad_row3_dma && (M >= wgrad_row3_min_m || (wgrad_row3_wide && Cin >= 512 && Cout >= 512)) && Cout % 128 == 0 && Cin % 128 == 0) return 5;
  if (row3) return (tile == 128 || ((!tile || tile == 256) && Cout >= 512)) ? 3 : 2;
  if (wgrad_1x1_dma && tile == 256 && big && KH * KW == 1 && stride == 1 && pad == 0 && same_hw && Cout % 256 == 0 && Cin % 256 == 0) return 6;
  if (tile == 256 && big && KH * KW == 1 && Cout % 256 == 0 && Cin % 256 == 0) return 4;
  return (tile == 128 || ((!tile || tile == 256) && big && KH * KW == 1)) ? 1 : 0;
}
void variant_tile(int v, int& bm, int& bn, int& tx) {
  if (v == 4 || v == 6) { bm = 256; bn = 256; tx = 1; return; }
  if (v == 5) { bm = 128; bn = 128; tx = 3; return; }
  bm = (v == 1 || v == 3) ? 128 : 64; bn = v == 1 ? 128 : 64; tx = v >= 2 ? 3 : 1;
}
long variant_tiles(int v, int Cin, int Cout, int KH, int KW) {
  if (v == 5) return l2s::wgrad_row3_dma_tiles(Cin, Cout);
  int bm, bn, tx; variant_tile(v, bm, bn, tx);
  return (long)cdiv(Cout, bm) * cdiv(Cin, bn) * (tx == 3 ? KH : KH * KW);
}

wgp prob_of(const l2s_wgrad_desc& d) {
  wgp p = {};
  p.dy[0] = d.dy; p.x[0] = d.x; p.dw = d.dw;
  p.n_img[0] = d.n_img; p.IH[0] = d.IH; p.IW[0] = d.IW; p.OH[0] = d.OH; p.OW[0] = d.OW; p.lddy[0] = d.lddy; p.ldx[0] = d.ldx;
  p.nseg = 1; p.Cin = d.Cin; p.Cout = d.Cout; p.KH = d.KH; p.KW = d.KW; p.stride = d.stride; p.pad = d.pad;
  return p;
}

template <typename T, int BM, int BN, int TX, int D>
int launch_single(const l2s_wgrad_desc& d, int split, hipStream_t st) {
  const wgp p = prob_of(d);
  const int cblocks = cdiv(d.Cin, BN);
  const long slab = split > 1 ? (long)d.Cout * d.KH * d.KW * d.Cin : 0;
  dim3 grid(cdiv(d.Cout, BM), (TX == 3 ? d.KH : d.KH * d.KW) * cblocks, split);
  float* ws = d.ws;
  L2S_LAUNCH((wgrad_kernel<T, BM, BN, TX, D>), grid, dim3(256), (wgrad_lds<T, BM, BN, TX>()), st, p, cblocks, ws, slab);
  if (split > 1) {
    const long n4 = slab / 4;
    long gb = (n4 + 255) / 256; if (gb > 4096) gb = 4096;
    float* dw = d.dw; const float* wsc = d.ws;
    L2S_LAUNCH(wgrad_reduce_kernel, dim3((int)gb), dim3(256), 0, st, dw, wsc, n4, slab, split);
  }
  return l2s_check_launch();
}

template <typename T, int BM, int BN, int TX, int D, int KSTEP, int WGM = 2, int WGN = 2>
int launch_grouped(const wgp* tab, const wg_prefix& pre, float* ws, bool any_split, hipStream_t st) {
  size_t lds = wgrad_lds<T, BM, BN, TX, KSTEP>();
  // A grouped launch is resident for hundreds of microseconds next to the main queue's small dependent launches, whose workgroups
  // (32 KiB of LDS) can only start on a CU with that much LDS free: two 128x128 tiles (2 x 74 KiB) leave none, and every main-queue launch
  // then waits for weight-gradient workgroups to retire.  The LDS REQUEST therefore bounds the residency: one 128x128 workgroup per CU
  // (>= 81 KiB requested), two of the smaller tiles (>= 54 KB), which always leaves >= 50 KiB.  158.3 -> 161.9 img/s (A/B in one box).
  constexpr size_t want = 54000, want_big = 83000;
  if (BM * BN >= 128 * 128) { if (want_big > lds) lds = want_big; }
  else if (want > lds) lds = want;
  static bool attr_done = false;
  if (!attr_done) { (void)hipFuncSetAttribute((const void*)wgrad_grouped_kernel<T, BM, BN, TX, D, KSTEP, WGM, WGN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done = true; }
  int grid = pre.tile0[pre.n];
  {
    const int cap = l2s_knobs::wgrad_grid_cap & 0xffff, vmask = l2s_knobs::wgrad_grid_cap >> 16;      // (tools build: cap | variant mask << 16; mask 0 = every variant)
    const int var = BM == 256 ? 4 : (TX == 3 ? (BM == 128 ? 3 : 2) : (BM == 128 ? 1 : 0));
    if (cap > 0 && grid > cap && (vmask == 0 || ((vmask >> var) & 1))) grid = cap;
  }
  L2S_LAUNCH((wgrad_grouped_kernel<T, BM, BN, TX, D, KSTEP, WGM, WGN>), dim3(grid), dim3(64 * WGM * WGN), lds, st, tab, pre, ws);
  if (any_split) {
    const float* wsc = ws;
    L2S_LAUNCH(wgrad_reduce_grouped_kernel, dim3(64, pre.n), dim3(256), 0, st, tab, wsc);
  }
  return l2s_check_launch();
}

bool prob_ok(const wgp& p, int dtype) {
  const int ve = dtype == L2S_BF16 ? 8 : 4;
  const long esz = dtype == L2S_BF16 ? 2 : 4;
  if (p.nseg < 1 || p.nseg > L2S_WGRAD_MAX_SEG || p.Cin % ve || p.Cout % ve || ((long)p.Cout * p.KH * p.KW * p.Cin) % 4) return false;
  for (int s = 0; s < p.nseg; ++s) {
    if (p.lddy[s] % ve || p.ldx[s] % ve) return false;
    const long M = (long)p.n_img[s] * p.OH[s] * p.OW[s];
    if (M >= (1 << 24)) return false;
    if ((long)p.n_img[s] * p.IH[s] * p.IW[s] * p.ldx[s] * esz >= (1L << 31) || M * p.lddy[s] * esz >= (1L << 31)) return false;   // 32-bit buffer offsets
  }
  return true;
}

}  // namespace

extern "C" size_t l2s_wgrad_grouped_ws_bytes(int variant) { return variant == 5 ? l2s::wgrad_row3_dma_ws_bytes(l2s_knobs::wgrad_row3_dma_wgs) : 0; }
extern "C" int l2s_wgrad_variant(int Cin, int Cout, int KH, int KW, int stride, int pad, int same_hw, long M, int tile) {
  return variant_of(Cin, Cout, KH, KW, stride, pad, same_hw, M, tile);
}
extern "C" long l2s_wgrad_tiles(int variant, int Cin, int Cout, int KH, int KW) { return variant_tiles(variant, Cin, Cout, KH, KW); }

extern "C" int l2s_conv_wgrad_grouped(const l2s_wgrad_prob* table_dev, const l2s_wgrad_prob* table_host, int nprob, int variant, int dtype,
                                      float* ws, size_t ws_bytes, hipStream_t stream) {
  if (!table_dev || !table_host || nprob < 1 || nprob > L2S_WGRAD_MAX_GROUP || variant < 0 || variant > 6) return L2S_EINVAL;
  if (variant == 6) {
    if (dtype != L2S_BF16) return L2S_EINVAL;
    for (int i = 0; i < nprob; ++i) if (!prob_ok(table_host[i], dtype) || !table_host[i].dw) return L2S_EINVAL;
    return l2s::wgrad_1x1_dma_launch(table_dev, table_host, nprob, stream);
  }
  if (variant == 5) {
    if (dtype != L2S_BF16) return L2S_EINVAL;
    for (int i = 0; i < nprob; ++i) if (!prob_ok(table_host[i], dtype) || !table_host[i].dw) return L2S_EINVAL;
    return l2s::wgrad_row3_dma_launch(table_dev, table_host, nprob, ws, ws_bytes, l2s_knobs::wgrad_row3_dma_wgs, stream);
  }
  wg_prefix pre;
  pre.n = nprob;
  long t = 0;
  bool any_split = false;
  for (int i = 0; i < nprob; ++i) {
    const wgp& q = table_host[i];
    if (!prob_ok(q, dtype) || !q.dw) return L2S_EINVAL;
    if ((variant == 2 || variant == 3) && !(q.KH == 3 && q.KW == 3 && q.stride == 1 && q.pad == 1)) return L2S_EINVAL;
    const int split = q.split > 1 ? q.split : 1;
    if (split > 1) {
      const long slab = (long)q.Cout * q.KH * q.KW * q.Cin;
      if (!ws || q.ws_off < 0 || (q.ws_off % 4) || (size_t)(q.ws_off + split * slab) * 4 > ws_bytes) return L2S_EINVAL;
      any_split = true;
    }
    pre.tile0[i] = (int)t;
    t += variant_tiles(variant, q.Cin, q.Cout, q.KH, q.KW) * split;
  }
  if (t >= (1L << 30)) return L2S_EINVAL;
  pre.tile0[nprob] = (int)t;
  for (int i = nprob + 1; i <= L2S_WGRAD_MAX_GROUP; ++i) pre.tile0[i] = (int)t;
  // (ring depth 2: with several workgroups per CU the other workgroups cover a load's latency; fewer registers = more of them)
  // KSTEP = MFMA k steps (32 pixels in bf16) per barrier: 2 (64-pixel slices) measured 137.4-139.7 vs 134.9-136.5 img/s for 1 (round 2)
#define GO(T, KS)                                                                                \
  switch (variant) {                                                                             \
    case 0: return launch_grouped<T, 64, 64, 1, 2, KS>(table_dev, pre, ws, any_split, stream);    \
    case 1: return launch_grouped<T, 128, 128, 1, 2, KS>(table_dev, pre, ws, any_split, stream);  \
    case 2: return launch_grouped<T, 64, 64, 3, 2, KS>(table_dev, pre, ws, any_split, stream);    \
    case 3: return launch_grouped<T, 128, 64, 3, 2, KS>(table_dev, pre, ws, any_split, stream);   \
    default: break;                                                                              \
  }
  if (dtype == L2S_BF16) {
    if (variant == 4) return launch_grouped<bf16_t, 256, 256, 1, 2, 1, 4, 2>(table_dev, pre, ws, any_split, stream);
    GO(bf16_t, 2)
  }
  if (dtype == L2S_F32) { GO(float, 1) }
#undef GO
  return L2S_EINVAL;
}

static int plan_split(const l2s_wgrad_desc& d, int variant, int dtype, size_t ws_bytes) {
  int bm, bn, tx; variant_tile(variant, bm, bn, tx);
  const long tiles = variant_tiles(variant, d.Cin, d.Cout, d.KH, d.KW);
  const int bkp = dtype == L2S_BF16 ? 32 : 16;
  const long Mv = (long)d.n_img * d.OH * (d.OW + (tx == 3 ? 1 : 0));
  const int slices = cdiv(Mv, bkp);
  int split = d.split_k;
  if (split <= 0) {
    constexpr int min_wg = 256;
    split = (int)((min_wg + tiles - 1) / tiles);
    const int maxs = slices / 8 > 0 ? slices / 8 : 1;
    if (split > maxs) split = maxs;
  }
  if (split > slices) split = slices;
  if (split < 1) split = 1;
  if (split > 64) split = 64;
  const long slab = (long)d.Cout * d.KH * d.KW * d.Cin * 4;
  if (split > 1 && (long)split * slab > (long)ws_bytes) split = ws_bytes >= (size_t)(2 * slab) ? (int)(ws_bytes / slab) : 1;
  return split;
}

extern "C" size_t l2s_wgrad_ws_bytes(const l2s_wgrad_desc* d, int dtype) {
  if (!d) return 0;
  const int v = variant_of(d->Cin, d->Cout, d->KH, d->KW, d->stride, d->pad, d->OH == d->IH && d->OW == d->IW, (long)d->n_img * d->OH * d->OW, d->tile);
  const int split = plan_split(*d, v, dtype, (size_t)1 << 40);
  return split > 1 ? (size_t)split * d->Cout * d->KH * d->KW * d->Cin * 4 : 0;
}

extern "C" int l2s_conv_wgrad(const l2s_wgrad_desc* d, int dtype, hipStream_t stream) {
  if (!d || !d->dy || !d->x || !d->dw) return L2S_EINVAL;
  if (!prob_ok(prob_of(*d), dtype)) return L2S_EINVAL;
  const int v = variant_of(d->Cin, d->Cout, d->KH, d->KW, d->stride, d->pad, d->OH == d->IH && d->OW == d->IW, (long)d->n_img * d->OH * d->OW, d->tile);
  const int split = plan_split(*d, v, dtype, d->ws ? d->ws_bytes : 0);
#define GO(T)                                                                                    \
  switch (v) {                                                                                   \
    case 0: return launch_single<T, 64, 64, 1, 4>(*d, split, stream);                             \
    case 1: return launch_single<T, 128, 128, 1, 3>(*d, split, stream);                           \
    case 2: return launch_single<T, 64, 64, 3, 4>(*d, split, stream);                             \
    default: return launch_single<T, 128, 64, 3, 3>(*d, split, stream);                           \
  }
  if (dtype == L2S_BF16) { GO(bf16_t) }
  if (dtype == L2S_F32) { GO(float) }
#undef GO
  return L2S_EINVAL;
}
