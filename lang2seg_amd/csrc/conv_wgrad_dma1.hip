// Weight gradient of the large 1x1 / stride 1 problems (resnet.layer4 on the 256 RoIs + on the map: seven problems, 235 GFLOP per step),
// bf16, on the LDS-DMA pattern (round 4, last session) - an alternative to the register-staged 256x256 tile of conv_wgrad.hip (51 % of the
// occupied CUs' matrix pipes, 28 % of its LDS cycles bank conflicts, 2.7 VALU instructions per MFMA: profiles/r04_pmc_wgrad.txt).
// MEASURED NO FASTER and therefore on request only (grouped variant 6; l2s_wgrad_row3_dma(67, 1) makes l2s_wgrad_variant choose it): 447-455 us
// against 425-431 us alone on 120 CUs, whether with three or four LDS stages and whether or not the tiles of an XCD share slabs - like
// igemm_dma256_kernel it moves 32 KB per slice and CU in ~0.95 us = 34 GB/s per CU, the rate at which a CU is served from beyond its XCD's L2.
// Behind autograd of nn.Conv2d (pyutils/mask-faster-rcnn/lib/nets/resnet_v1_cycle_res5_2.py:83-88).
//
//   dW[co][ci] (+)= sum_m dY[m][co] * X[m][ci]            m = the pixels of the problem's one or two segments (RoIs; the map)
//
// * One workgroup per 256 (co) x 256 (ci) tile, over ALL pixels: no pixel split, no slabs, no second launch - layer4's seven problems are 120
//   tiles = 120 workgroups on 120 CUs, which is what the step wants beside the data-gradient chain.
// * Eight waves as 4 (co) x 2 (ci): wave tile 64 x 128, 128 accumulator registers.  A slice = 32 pixels = one MFMA k step: the dY slab is
//   32 rows x 512 B, the X slab 32 rows x 512 B, both by `buffer_load_dwordx4 ... lds` (one request = 1 KiB = two pixel rows, four per wave
//   and slice), 16-byte chunks XOR-swizzled with 2 (row & 15) through the lane's source offset: a ds_read_b64_tr_b16 of 16 rows x 32 B then
//   covers all 64 banks twice (its minimum).  Rows past a segment's last pixel carry offset 0x80000000 (zeros by the range check).
// * Four LDS stages (128 KB, three slices in flight: with two the launch was bound by the requests' latency - 64 KB in flight per CU at ~2 us);
//   the two halves of the workgroup run one slot apart as in igemm_dma_kernel / conv_wgrad_dma.hip: LOAD = the
//   slice's 24 fragment reads + the four requests of the slice two ahead, MUL = 32 bare MFMAs; counted vmcnt.
// * Every accumulator sums its pixels in order, segment 0 first: bit-reproducible; equal to the register-staged tile's sums wherever that one
//   was not split.
#include "common.h"
#include "../../include/lang2seg_hip.h"
#include "wgrad_internal.h"

namespace {

typedef l2s_wgrad_prob wgp;
typedef int i32x4s __attribute__((ext_vector_type(4)));
constexpr unsigned OOR = 0x80000000u;
constexpr int BM = 256, BN = 256, KP = 32, ROWB = 512;
constexpr int A_BYTES = KP * ROWB;                  // 16 KiB: 32 dY rows
constexpr int STG = 2 * A_BYTES;                    // + 32 X rows
constexpr int NP = 4;                               // requests per wave and slice

__device__ __forceinline__ void dma_rows2(const i32x4s& rsrc, unsigned voff, unsigned lds_) {
  const unsigned lds = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds) : "memory");
}
__device__ __forceinline__ void wgb() { asm volatile("s_barrier" ::: "memory"); }

template <int NST>   // LDS stages: NST - 1 slices are requested ahead
__global__ __launch_bounds__(512) void wgrad_1x1_dma_kernel(const wgp* __restrict__ tab, const l2s::wgrad_tile_prefix pre) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int wm = wave >> 1, wn = wave & 1;            // 4 x 2 waves, wave tile 64 (co) x 128 (ci)
  const int fr = lane & 15, fg = lane >> 4;
  // XCD-aware order: workgroup b runs on XCD b % 8; XCD x takes the x-th contiguous eighth of the tile list, whose neighbours share a dY or an X
  // slab (operands far larger than one XCD's 4 MiB L2 are otherwise served at the fabric's ~33 GB/s per CU)
  int bid;
  {
    const int G = pre.tile0[pre.n], b = (int)blockIdx.x, x = b & 7, slot = b >> 3, q = G >> 3, r = G & 7;
    bid = x * q + min(x, r) + slot;
  }
  int lo = 0, hi = pre.n;                             // tile0[lo] <= bid < tile0[hi]
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (pre.tile0[mid] <= bid) lo = mid; else hi = mid; }
  const wgp p = tab[lo];
  const int t = bid - pre.tile0[lo];
  const int co_tiles = p.Cout / BM;
  const int co0 = (t % co_tiles) * BM, ci0 = (t / co_tiles) * BN;
  const int M0 = p.n_img[0] * p.OH[0] * p.OW[0], M1 = p.nseg > 1 ? p.n_img[1] * p.OH[1] * p.OW[1] : 0;
  const int ns0 = (M0 + KP - 1) / KP, KT = ns0 + (M1 + KP - 1) / KP;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  auto desc = [](const void* base, long bytes) {
    i32x4s r; r.x = (int)(uintptr_t)base; r.y = (int)((uintptr_t)base >> 32); r.z = (int)bytes; r.w = 0x00020000; return r;
  };
  const i32x4s rdy0 = desc(p.dy[0], (long)M0 * p.lddy[0] * 2L), rx0 = desc(p.x[0], (long)M0 * p.ldx[0] * 2L);
  const i32x4s rdy1 = desc(p.nseg > 1 ? p.dy[1] : p.dy[0], (long)M1 * p.lddy[p.nseg > 1] * 2L), rx1 = desc(p.nseg > 1 ? p.x[1] : p.x[0], (long)M1 * p.ldx[p.nseg > 1] * 2L);
  // ---- DMA lane constants: lane l of a request writes 16 bytes at LDS offset 16 l of the request's two rows: row g = l >> 5, physical chunk
  // l & 31; it fetches source chunk physical ^ 2 (r & 15), r = the row's index in its slab.  Wave w requests rows 2 w + g and 16 + 2 w + g of
  // both slabs: r & 15 = 2 w + g for all four requests.
  const int g = lane >> 5;
  const int sc = (lane & 31) ^ (2 * ((2 * wave + g) & 15));
  const int rowq[2] = {2 * wave + g, 16 + 2 * wave + g};
  // ---- fragment reads (ds_read_b64_tr_b16): lane -> pixel row trow (+16 for the second half of the k step) and 8 bytes at channel offset
  // 16 i + 4 (lane & 3) of the wave's columns; physical chunk = (byte offset >> 4) ^ 2 (row & 15)
  const int trow = 4 * fg + ((lane >> 2) & 3), cb = (lane & 3) >> 1, inb = (lane & 1) << 3;
  unsigned aoff[4], boff[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) aoff[i] = (unsigned)(trow * ROWB + (((wm * 8 + 2 * i + cb) ^ (2 * trow)) << 4) + inb);
#pragma unroll
  for (int j = 0; j < 8; ++j) boff[j] = (unsigned)(A_BYTES + trow * ROWB + (((wn * 16 + 2 * j + cb) ^ (2 * trow)) << 4) + inb);

  auto request = [&](int stage, int ts) {               // slice ts -> LDS stage `stage`
    const bool s1 = ts >= ns0;
    const int m0 = (s1 ? ts - ns0 : ts) * KP;
    const int nv = (s1 ? M1 : M0) - m0;                 // valid rows of the slice (>= 1)
    const int ldd = p.lddy[s1], ldxx = p.ldx[s1];
    const i32x4s rd = s1 ? rdy1 : rdy0, rxx = s1 ? rx1 : rx0;
    const unsigned sb = lds0 + (unsigned)(stage * STG);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = rowq[q];
      const unsigned va = r < nv ? (unsigned)(((m0 + r) * ldd + co0 + sc * 8) * 2) : OOR;   // (< 2^31: prob_ok)
      dma_rows2(rd, va, sb + (unsigned)((wave + 8 * q) * 1024));
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int r = rowq[q];
      const unsigned vb = r < nv ? (unsigned)(((m0 + r) * ldxx + ci0 + sc * 8) * 2) : OOR;
      dma_rows2(rxx, vb, sb + (unsigned)(A_BYTES + (wave + 8 * q) * 1024));
    }
  };
  f32x4 acc[4][8];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  uint4 fa[4], fb[8];
  auto frag = [&](const char* q) {
    const s16x4 l4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(q));
    const s16x4 h4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(q + 16 * ROWB));
    return __builtin_bit_cast(uint4, (s16x8){l4[0], l4[1], l4[2], l4[3], h4[0], h4[1], h4[2], h4[3]});
  };
  auto read_all = [&](int stage) {
    const char* base = smem + stage * STG;
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = frag(base + aoff[i]);
#pragma unroll
    for (int j = 0; j < 8; ++j) fb[j] = frag(base + boff[j]);
  };
  auto mma_all = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[j]), __builtin_bit_cast(bf16x8, fa[i]), acc[i][j], 0, 0, 0);
  };
  // ---- pipeline: slices 0 .. NST-2 requested; everybody waits for its share of slice 0 ----
  // wait until at most `younger` slices' requests of this wave are outstanding (they complete in order)
  auto wait_keep = [&](int younger) {
    if (younger >= 2 && NST >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < KT) request(s, s);
  wait_keep(min(NST - 2, KT - 1));                      // slice 0 landed
  wgb();
  if (grp == 1) wgb();                                  // one slot behind group 0
  int st = 0;
  for (int ts = 0; ts < KT; ++ts) {
    const int stn = st == 0 ? NST - 1 : st - 1;         // (ts + NST - 1) % NST: the stage of slice ts - 1
    read_all(st);
    if (ts + NST - 1 < KT) request(stn, ts + NST - 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int younger = max(0, min(NST - 2, KT - 2 - ts));   // requests that may stay in flight behind slice ts + 1
    if (grp == 1) wait_keep(younger);                   // slice ts + 1 landed
    wgb();
    __builtin_amdgcn_sched_barrier(0);
    mma_all();
    __builtin_amdgcn_sched_barrier(0);
    if (grp == 0) wait_keep(younger);
    wgb();
    st = st == NST - 1 ? 0 : st + 1;
  }
  if (grp == 0) wgb();                                  // group 1's last MUL slot
  // ---- the tile: lane = output channel co (fr), four consecutive input channels (fg) per accumulator ----
  const bool overwrite = p.flags & 1;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long co = co0 + wm * 64 + i * 16 + fr;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int ci = ci0 + wn * 128 + j * 16 + fg * 4;
      float4* q = (float4*)(p.dw + co * p.Cin + ci);
      float4 v = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
      if (!overwrite) { const float4 o = *q; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
      *q = v;
    }
  }
}

}  // namespace

namespace l2s {

bool wgrad_1x1_dma_ok(const l2s_wgrad_prob& q) {
  if (!(q.KH == 1 && q.KW == 1 && q.stride == 1 && q.pad == 0) || q.Cin % BN || q.Cout % BM || q.split > 1 || q.nseg < 1 || q.nseg > 2) return false;
  for (int s = 0; s < q.nseg; ++s) {
    if (q.OH[s] != q.IH[s] || q.OW[s] != q.IW[s] || (long)q.n_img[s] * q.OH[s] * q.OW[s] < 1) return false;
    if (q.lddy[s] % 8 || q.ldx[s] % 8 || ((uintptr_t)q.dy[s] & 15) || ((uintptr_t)q.x[s] & 15)) return false;
  }
  return !((uintptr_t)q.dw & 15);
}
long wgrad_1x1_dma_tiles(int Cin, int Cout) { return (long)(Cout / BM) * (Cin / BN); }

int wgrad_1x1_dma_launch(const l2s_wgrad_prob* tab_dev, const l2s_wgrad_prob* tab_host, int nprob, hipStream_t st) {
  wgrad_tile_prefix pre;
  pre.n = nprob;
  long t = 0;
  for (int i = 0; i < nprob; ++i) {
    if (!wgrad_1x1_dma_ok(tab_host[i])) return L2S_EINVAL;
    pre.tile0[i] = (int)t;
    t += wgrad_1x1_dma_tiles(tab_host[i].Cin, tab_host[i].Cout);
  }
  for (int i = nprob; i <= L2S_WGRAD_MAX_GROUP; ++i) pre.tile0[i] = (int)t;
  static bool attr = false;
  const size_t lds = (size_t)4 * STG;                  // four stages: 128 KiB, three slices (96 KB) in flight per CU
  if (!attr) { (void)hipFuncSetAttribute((const void*)wgrad_1x1_dma_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr = true; }
  L2S_LAUNCH(wgrad_1x1_dma_kernel<4>, dim3((unsigned)t), dim3(512), lds, st, tab_dev, pre);
  return l2s_check_launch();
}

}  // namespace l2s
