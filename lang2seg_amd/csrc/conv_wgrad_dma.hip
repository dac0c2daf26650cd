// Weight gradient of the large 3x3 / stride 1 / pad 1 problems (resnet.layer4 on the 256 RoIs + on the map: 3 x 70 GFLOP per step),
// bf16, rebuilt on the LDS-DMA pattern of igemm_dma_kernel (round 4; the register-staged filter-row tile of conv_wgrad.hip ran this
// launch at 0.17 of peak: 2.6 VALU instructions per MFMA for the loader's pixel walk, 27 % of the LDS cycles bank conflicts, 36 % of a
// wave's life waiting for operands - profiles/r03_pmc_wgrad.txt).  Replaces cuDNN's backward-filter behind autograd of nn.Conv2d
// (pyutils/mask-faster-rcnn/lib/nets/resnet_v1_cycle_res5_2.py:83-88).
//
//   dW[co][ky][kx][ci] += sum_m dY[m][co] * X(m, ky, kx)[ci]
//
// * Tile = 128 (co) x 128 (ci) x the three taps of ONE filter row, eight waves as 2 x 4 (wave tile 64 x 32 x 3 taps, 96 accumulator
//   registers).  Pixels are walked in the virtual layout of conv_wgrad.hip (one zero column appended to every image row), 64 per slice:
//   the dY slab is 64 rows of 256 B, the X slab 66 rows (pixels -1 .. +64) of 256 B, and tap kx reads the X slab one row further down.
// * Both slabs enter LDS by `buffer_load_dwordx4 ... lds`: one request = 1 KiB = FOUR 256-byte pixel rows, the 16-byte chunks
//   XOR-swizzled through the lane's source offset (chunk ^ 2 (row & 7): the ds_read_b64_tr_b16 fragment reads of 8 rows x 32 B then
//   cover all 64 banks; SQ_LDS_BANK_CONFLICT = 0).  No register staging, no ds_write.  Lanes 0..19 of a wave keep the
//   (image row, row, column) state of the 20 rows the wave requests, advance it by adds / compares / selects (no multiply), and five
//   crossbar permutes hand every lane the offset of its row.  Rows outside the image (the zero column, the rows above / below for
//   ky = 0 / 2, pixels past the end) carry offset 0x80000000: the buffer's range check writes zeros to LDS.
// * Four LDS stages (132 KB), one workgroup per CU.  The two halves of the workgroup (the two waves of every SIMD) run one slot
//   apart: LOAD = offsets, the slice's 40 fragment reads into 80 registers, the requests of the slice three ahead; MUL = 48 bare
//   MFMAs.  Counted vmcnt waits (the DMA is inline asm: hipcc's waitcnt pass neither sees nor drains it).
//   In-kernel stamps (tools/wgrad_stamps.py): MUL 720 cycles (15 per MFMA), LOAD ~1150 (the reads come back at ~100 B/clk per CU
//   while the other group's MFMAs and the landing DMA share the CU), slice 3300 cycles; PMC: MFMA pipes 50 % busy, 1.0 VALU per MFMA.
// * 144 tiles of 262 slices do not fill 256 CUs and their remainder would idle half the chip, so the launch is balanced stream-K
//   style: the (tile, slice) units are cut into G equal contiguous ranges, one per workgroup.  A workgroup finishes whole tiles into dW
//   directly; the at most two tiles it shares with its neighbours go to its two slab slots in the workspace, and a second launch adds
//   a shared tile's slabs in workgroup order (fixed order: bit-reproducible, no atomics).
#include "common.h"
#include "../../include/lang2seg_hip.h"
#include "wgrad_internal.h"

namespace {

typedef l2s_wgrad_prob wgp;
typedef int i32x4s __attribute__((ext_vector_type(4)));
constexpr unsigned OOR = 0x80000000u;
constexpr int BM = 128, BN = 128, BKP = 64, ROWB = 256;
constexpr int A_BYTES = BKP * ROWB;                 // 64 dY rows
constexpr int B_ROWS = 68;                          // 66 X rows used (17 requests of 4 rows)
constexpr int STG = A_BYTES + B_ROWS * ROWB;        // 33 792 B per stage
constexpr int NPW = 4;                              // DMA requests per wave and slice: 2 x 4 dY rows, 2 x 4 X rows (wave 0: + the X rows 64..67)
constexpr int TILE_FLOATS = 3 * BM * BN;

// one request = 1 KiB = FOUR 256-byte pixel rows (16 lanes x 16 B each): the vector memory pipe of a CU takes ~16 cycles per wave
// instruction whatever its width - with one row per `buffer_load_dword ... lds` the 136 requests of a slice cost 0.87 us, more than
// its MFMAs (knock-outs: requests alone 128 us of the launch, MFMAs alone 101 us, and the two did not overlap)
__device__ __forceinline__ void dma_rows4(const i32x4s& rsrc, unsigned voff, unsigned lds_) {
  const unsigned lds = (unsigned)__builtin_amdgcn_readfirstlane((int)lds_);   // (uniform; makes the "s" operand a scalar register even where hipcc kept the value in a VGPR)
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(voff), "s"(rsrc), "s"(lds) : "memory");
}
__device__ __forceinline__ long wg_lo(long g, long U, long G) { return g * U / G; }

template <int KO, int NST>   // NST LDS stages (NST - 1 slices requested ahead); KO bit 3 (8): waves 0 / 4 of workgroup 0 stamp the shader clock (tools/wgrad_stamps.py); knock-outs (tools only): 1 = no MFMAs, 2 = no requests after the prologue, 4 = no fragment reads
__global__ __launch_bounds__(512) void wgrad_row3_dma_kernel(const wgp* __restrict__ tab, const l2s::wgrad_sk_plan plan, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;            // 2 x 4 waves, wave tile 64 (co) x 32 (ci)
  const int fr = lane & 15, fg = lane >> 4;
  // XCD-aware: workgroups are dealt round-robin over the 8 XCDs; XCD x takes the x-th contiguous eighth of the unit ranges (neighbouring
  // tiles share their X slab through that XCD's L2)
  const int G = gridDim.x;
  const int L = (G & 7) == 0 ? (int)((blockIdx.x & 7) * (G >> 3) + (blockIdx.x >> 3)) : (int)blockIdx.x;
  const long u_lo = wg_lo(L, plan.U, G), u_hi = wg_lo(L + 1, plan.U, G);
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;

  // ---- lane constants ----
  // DMA: lane l of a request writes 16 bytes at LDS offset 16 l of the request's four rows: row g = l >> 4 of the four, physical chunk
  // l & 15; it fetches source chunk physical ^ 2 (r & 7) of that row, r = the row's index in its slab.  Every request starts at a
  // multiple of four rows, so r & 7 = 4 (request index & 1) + g: two constants per lane.
  const int g4 = lane >> 4;
  unsigned vc[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) vc[par] = (unsigned)(((lane & 15) ^ ((4 * par + g4) << 1)) << 4);
  // fragment reads (ds_read_b64_tr_b16): lane -> pixel row trow (+16 for the second half of the k step) and 8 bytes at channel
  // offset 16 i + 4 (lane & 3) of the wave's columns; chunk = (byte offset >> 4) ^ 2 (row & 7)
  const int trow = 4 * fg + ((lane >> 2) & 3), cb = (lane & 3) >> 1, inb = (lane & 1) << 3;
  unsigned aoff[4], boff[3][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) aoff[i] = (unsigned)(trow * ROWB + (((wm * 8 + 2 * i + cb) ^ ((trow & 7) << 1)) << 4) + inb);
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int j = 0; j < 2; ++j)
      boff[t][j] = (unsigned)(A_BYTES + (trow + t) * ROWB + (((wn * 4 + 2 * j + cb) ^ (((trow + t) & 7) << 1)) << 4) + inb);
  // loader roles: lanes 0..7 hold the state of the wave's dY rows 8 wave + lane; lanes 8..11 of its X rows 4 wave + (lane - 8), lanes
  // 12..15 of X rows 32 + 4 wave + (lane - 12), lanes 16..19 of X rows 64 + (lane - 16) (requested by wave 0)
  const bool roleA = lane < 8;
  const int myrow = roleA ? 8 * wave + lane : (lane < 12 ? 4 * wave + (lane - 8) : (lane < 16 ? 32 + 4 * wave + (lane - 12) : 64 + ((lane - 16) & 3)));
  long u = u_lo;
  int part = 0;
  // XCD-lockstep plan (mode 1): the hardware deals workgroups round-robin over the 8 XCDs, so x = blockIdx & 7 is this workgroup's XCD
  // and ix = blockIdx >> 3 its index there.  Pass q < Q: XCD x works on group 8 q + x (the T = co tiles x ci tiles tiles of one
  // problem and filter row), k1 workgroups per tile, each a k1-th of the slices; last pass: the r remaining groups, m = 8 / r XCDs each.
  const int xcd = blockIdx.x & 7, ix = blockIdx.x >> 3;
  const int npass = plan.mode == 1 ? plan.Q + (plan.r ? 1 : 0) : 0;
  for (int pass = 0; plan.mode == 1 ? pass < npass : u < u_hi; ++pass) {
    int plo, S, tile, s0, s1;
    long slab_id;
    if (plan.mode == 1) {
      const int tl = ix % plan.T, c = ix / plan.T;
      int gt, pieces, piece;                              // global tile, pieces the tile is cut into, this workgroup's piece
      if (pass < plan.Q) { gt = (8 * pass + xcd) * plan.T + tl; pieces = plan.k1; piece = c; slab_id = (long)gt * plan.k1 + c; }
      else {
        const int j = xcd / plan.m, sub = xcd - j * plan.m;
        const int gl = j * plan.T + tl;                   // tile among those of the last pass
        gt = 8 * plan.Q * plan.T + gl; pieces = plan.m * plan.k1; piece = sub * plan.k1 + c;
        slab_id = (long)8 * plan.Q * plan.T * plan.k1 + (long)gl * pieces + piece;
      }
      S = plan.S[0];
      const int tpp = 3 * plan.T;                         // tiles per problem
      plo = gt / tpp; tile = gt - plo * tpp;
      s0 = (int)((long)piece * S / pieces); s1 = (int)((long)(piece + 1) * S / pieces);
      if (pieces == 1) slab_id = -1;                      // whole tile: straight into dW
      if (s0 >= s1) continue;
    } else {
      plo = 0; int phi = plan.n;
      while (phi - plo > 1) { const int mid = (plo + phi) >> 1; if (plan.unit0[mid] <= u) plo = mid; else phi = mid; }
      S = plan.S[plo];
      const long rr = u - plan.unit0[plo];
      tile = (int)(rr / S); s0 = (int)(rr - (long)tile * S);
      s1 = (int)(((long)S - s0) < (u_hi - u) ? (long)S : s0 + (u_hi - u));
      slab_id = (s0 == 0 && s1 == S) ? -1 : (long)L * 2 + part;
    }
    const wgp p = tab[plo];
    const int co_tiles = p.Cout / BM, ci_tiles = p.Cin / BN;
    const int cot = tile % co_tiles, rest = tile / co_tiles, cit = rest % ci_tiles, ky = rest / ci_tiles;
    const int co0 = cot * BM, ci0 = cit * BN;

    f32x4 acc[3][4][2];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[t][i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int sbase = 0;
#pragma unroll
    for (int seg = 0; seg < L2S_WGRAD_MAX_SEG; ++seg) {
      if (seg >= p.nseg) break;
      const int n_img = p.n_img[seg], OH = p.OH[seg], OW = p.OW[seg], lddy = p.lddy[seg], ldx = p.ldx[seg];
      const int Wv = OW + 1, ohwv = OH * Wv, Mv = n_img * ohwv;
      const int nsl = (Mv + BKP - 1) / BKP;
      const int sa_ = (s0 > sbase ? s0 : sbase) - sbase, sb_ = (s1 < sbase + nsl ? s1 : sbase + nsl) - sbase;
      sbase += nsl;
      if (sa_ >= sb_) continue;
      const int n = sb_ - sa_;                          // slices of this segment in this pass
      const int ja = BKP / ohwv, jb = (BKP - ja * ohwv) / Wv, jc = BKP - ja * ohwv - jb * Wv;   // 64 = ja OH Wv + jb Wv + jc
      i32x4s rdy, rxx;
      rdy.x = (int)(uintptr_t)p.dy[seg]; rdy.y = (int)((uintptr_t)p.dy[seg] >> 32); rdy.z = 0x7FFFFFFF; rdy.w = 0x00020000;
      rxx.x = (int)(uintptr_t)p.x[seg]; rxx.y = (int)((uintptr_t)p.x[seg] >> 32); rxx.z = 0x7FFFFFFF; rxx.w = 0x00020000;
      // this lane's row: virtual pixel -> (image row R = image * OH + row, row, column); the X slab starts one pixel early.  Everything the
      // offset needs is kept incrementally (adds, compares, selects: no multiply in the loop): offR = byte offset of image row R (shifted by
      // the filter row for X), offx = byte offset of column x
      const int ld2 = (roleA ? lddy : ldx) * 2, rowbytes = OW * ld2;
      const int Rmax = n_img * OH;
      int pR, py, px, offR, offx;
      {
        const int pix = sa_ * BKP + myrow - (roleA ? 0 : 1);
        int pn;
        if (pix < 0) { pn = 0; py = 0; px = -1; }
        else { pn = pix / ohwv; const int rem = pix - pn * ohwv; py = rem / Wv; px = rem - py * Wv; }
        pR = pn * OH + py;
        offR = (pR + (roleA ? 0 : ky - 1)) * rowbytes + (roleA ? co0 : ci0) * 2;
        offx = px * ld2;
      }
      const bool rowok = roleA || myrow < BKP + 2;       // (X rows 66, 67 of the last request: zeros)
      const int dR = ja * OH + jb, dRb = dR * rowbytes, dxb = jc * ld2, wxb = Wv * ld2;
      const int ylo = roleA ? 0 : 1 - ky, yhi = roleA ? OH : OH + 1 - ky;       // rows whose X row y + ky - 1 is inside the image
      unsigned vo[5];                                   // per-lane source offsets of the wave's (up to) five requests of the prepared slice
      unsigned poff = OOR, pv[5];                       // in flight between the three steps of prep
      auto prep_a = [&]() {                             // this lane's row: offset in the next slice to prepare; the state moves on by 64 pixels
        const bool ok = rowok && pR < Rmax && px >= 0 && px < OW && py >= ylo && py < yhi;
        poff = ok ? (unsigned)(offR + offx) : OOR;
        px += jc; offx += dxb;
        const bool cx = px >= Wv;
        px = cx ? px - Wv : px; offx = cx ? offx - wxb : offx;
        pR += dR + (cx ? 1 : 0); offR += dRb + (cx ? rowbytes : 0);
        py += jb + (cx ? 1 : 0);
        py = py >= OH ? py - OH : py;
      };
      // lane l of request k needs the offset held by role lane 4 k + (l >> 4): one crossbar permute per request, issued BEHIND the wave's own
      // fragment reads in the LOAD slot (issued in the MUL slot they queued behind the OTHER group's 160 reads in the CU's LDS pipe and held
      // this wave's MFMA issue for ~500 cycles per slice; four v_readlane + three selects per request cost ~800 cycles)
      auto prep_b = [&]() {
#pragma unroll
        for (int k = 0; k < 5; ++k) pv[k] = (unsigned)__builtin_amdgcn_ds_bpermute(4 * g4 + 16 * k, (int)poff);
      };
      auto prep_c = [&]() {
        vo[0] = pv[0] + vc[0]; vo[1] = pv[1] + vc[1];   // dY rows 8 wave + 4 k + g: r & 7 = 4 k + g
        vo[2] = pv[2] + vc[wave & 1]; vo[3] = pv[3] + vc[wave & 1];   // X rows 4 wave + g, 32 + 4 wave + g
        vo[4] = pv[4] + vc[0];                          // X rows 64 + g
      };
      auto prep = [&]() { prep_a(); prep_b(); prep_c(); };
      auto request = [&](int stage) {                   // the prepared slice -> LDS stage `stage`
        const unsigned sb = lds0 + (unsigned)(stage * STG);
        dma_rows4(rdy, vo[0], sb + (unsigned)((8 * wave) * ROWB));
        dma_rows4(rdy, vo[1], sb + (unsigned)((8 * wave + 4) * ROWB));
        dma_rows4(rxx, vo[2], sb + (unsigned)(A_BYTES + (4 * wave) * ROWB));
        dma_rows4(rxx, vo[3], sb + (unsigned)(A_BYTES + (32 + 4 * wave) * ROWB));
        if (wave == 0) dma_rows4(rxx, vo[4], sb + (unsigned)(A_BYTES + 64 * ROWB));
      };
      // Fragments of a whole slice (2 k steps x (4 dY + 3 x 2 X) fragments = 80 registers): read in the LOAD slot, multiplied in the MUL slot
      uint4 fa[2][4], fb[2][3][2];
      auto read_all = [&](int stage) {
        const char* base = smem + stage * STG;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const char* q = base + aoff[i] + h * 32 * ROWB;
            s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(q));
            s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(q + 16 * ROWB));
            fa[h][i] = __builtin_bit_cast(uint4, (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
          }
#pragma unroll
          for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              const char* q = base + boff[t][j] + h * 32 * ROWB;
              s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(q));
              s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(q + 16 * ROWB));
              fb[h][t][j] = __builtin_bit_cast(uint4, (s16x8){lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
            }
        }
      };
      // the slice's 48 MFMAs, bare (anything else in this stream costs matrix-pipe time: a wave issues one instruction per four cycles)
      // the slice's 48 MFMAs, bare (anything else in this stream costs matrix-pipe time: a wave issues one instruction per four cycles)
      auto mma_all = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
              for (int j = 0; j < 2; ++j) {
                if (!(KO & 1)) acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fb[h][t][j]), __builtin_bit_cast(bf16x8, fa[h][i]), acc[t][i][j], 0, 0, 0);
              }
      };
      // ---- pipeline (the scheme of igemm_dma_kernel): slice t lives in stage t % 3; the two halves of the workgroup - waves 0-3 and
      // 4-7, the two waves of every SIMD - run the same loop ONE SLOT APART, a slot being the time between two workgroup barriers:
      //     slot        2t          2t+1         2t+2
      //     group 0     LOAD(t)     MUL(t)       LOAD(t+1)       LOAD(t) = the 40 fragment reads of slice t + the 17 requests of slice t+2
      //     group 1     MUL(t-1)    LOAD(t)      MUL(t)          MUL(t)  = its 48 MFMAs (+ the offsets of slice t+3)
      // so the matrix pipe of a SIMD always has one wave feeding it while the other wave's LDS reads and requests are in flight.  A wave
      // retires ITS share of slice t+1 before the barrier that ends slot 2t+1 (counted vmcnt: only its requests of slice t+2 are younger);
      // a stage is refilled only after every wave has passed a barrier behind the lgkmcnt(0) that retired its reads of that stage.
      const int grp = wave >> 2;
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (the previous pass's fragments are in registers everywhere)
      auto wait_keep = [&](int younger) {               // everything but this wave's requests of the `younger` youngest slices has landed
        if (wave == 0) {
          if (younger >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * (NPW + 1)) : "memory");
          else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NPW + 1)) : "memory");
          else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW + 1) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
          if (younger >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NPW) : "memory");
          else if (younger == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NPW) : "memory");
          else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      };
      // prologue: slices 0 .. NST-2 requested, slice NST-1 prepared, this wave's share of slice 0 landed
#pragma unroll
      for (int q = 0; q < NST - 1; ++q)
        if (q < n) { prep(); request(q); }
      wait_keep(n - 1 < NST - 2 ? n - 1 : NST - 2);
      if (NST - 1 < n) prep();
      asm volatile("s_barrier" ::: "memory");
      if (grp == 1) asm volatile("s_barrier" ::: "memory");              // one slot behind group 0
      int st = 0;
      unsigned long long* stamps = (unsigned long long*)(ws + (long)gridDim.x * 3 * TILE_FLOATS) + grp * 32 * 8;
      const bool stamping = (KO & 8) && blockIdx.x == 0 && (wave & 3) == 0 && lane == 0 && pass == 0 && seg == 0;
      auto stamp = [&](int t, int i) {
        if constexpr ((KO & 8) != 0) {
          if (t < 32) { const unsigned long long c = __builtin_amdgcn_s_memtime(); if (stamping) stamps[t * 8 + i] = c; }
        }
      };
      for (int t = 0; t < n; ++t) {
        const int stq = st == 0 ? NST - 1 : st - 1;      // (t + NST - 1) % NST: the stage of slice t - 1
        const int younger = n - 2 - t < NST - 2 ? (n - 2 - t < 0 ? 0 : n - 2 - t) : NST - 2;   // requested slices behind slice t + 1 after this slot's request
        // LOAD slot: the next offsets' arithmetic, the 40 fragment reads of slice t, the five permutes that spread the offsets, the requests
        // of slice t + NST - 1 (offsets prepared one slot earlier) - and ONE wait for the LDS pipe
        stamp(t, 0);
        const bool more = t + NST < n && !(KO & 16);
        if (more) prep_a();
        if (!(KO & 4)) read_all(st);
        if (more) prep_b();
        stamp(t, 1);
        if (t + NST - 1 < n && !(KO & 2)) request(stq);   // (behind the reads; between the MFMAs of the MUL slot they cost as much matrix-pipe time as they save here)
        stamp(t, 2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (more) prep_c();
        stamp(t, 3);
        if (grp == 1) wait_keep(younger);
        asm volatile("s_barrier" ::: "memory");
        stamp(t, 4);
        __builtin_amdgcn_sched_barrier(0);
        mma_all();
        __builtin_amdgcn_sched_barrier(0);
        stamp(t, 5);
        if (grp == 0) wait_keep(younger);
        stamp(t, 6);
        asm volatile("s_barrier" ::: "memory");
        stamp(t, 7);
        st = st == NST - 1 ? 0 : st + 1;
      }
      if (grp == 0) asm volatile("s_barrier" ::: "memory");              // group 1's last MUL slot
    }

    // ---- the tile: whole -> dW (+=), shared with a neighbour -> this workgroup's slab ----
    const bool full = slab_id < 0;
    const long Kw = 9L * p.Cin;
    float* slab = ws + (full ? 0 : slab_id) * TILE_FLOATS;
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int col = wm * 64 + i * 16 + fr;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int cil = wn * 32 + j * 16 + fg * 4;
          float4 v = make_float4(acc[t][i][j][0], acc[t][i][j][1], acc[t][i][j][2], acc[t][i][j][3]);
          if (full) {
            float4* q = (float4*)(p.dw + (long)(co0 + col) * Kw + (long)(ky * 3 + t) * p.Cin + ci0 + cil);
            if (!(p.flags & 1)) { const float4 o = *q; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
            *q = v;
          } else {
            *(float4*)(slab + ((long)(t * BM + col) * BN + cil)) = v;
          }
        }
      }
    part = 1;
    u += s1 - s0;
  }
}

// tiles that more than one workgroup worked on: dW += slab(g_first) + ... + slab(g_last), in that order.  grid (blocks, tiles)
__global__ __launch_bounds__(256) void wgrad_row3_reduce_kernel(const wgp* __restrict__ tab, const l2s::wgrad_sk_plan plan, const float* __restrict__ ws, int G) {
  int T = blockIdx.y, pi = 0;
  for (; pi < plan.n; ++pi) {
    const int tiles = (int)((plan.unit0[pi + 1] - plan.unit0[pi]) / plan.S[pi]);
    if (T < tiles) break;
    T -= tiles;
  }
  if (pi >= plan.n) return;
  const wgp p = tab[pi];
  const int S = plan.S[pi];
  long g0, g1, slab0 = 0;
  const long a = plan.unit0[pi] + (long)T * S, b = a + S;
  if (plan.mode == 1) {
    const long gt = (long)pi * 3 * plan.T + T, first_last = (long)8 * plan.Q * plan.T;
    const int pieces = gt < first_last ? plan.k1 : plan.m * plan.k1;
    if (pieces == 1) return;
    slab0 = gt < first_last ? gt * plan.k1 : first_last * plan.k1 + (gt - first_last) * pieces;
    g0 = 0; g1 = pieces - 1;
  } else {
    g0 = a * G / plan.U; g1 = (b - 1) * G / plan.U;
    while (g0 + 1 < G && wg_lo(g0 + 1, plan.U, G) <= a) ++g0;
    while (g0 > 0 && wg_lo(g0, plan.U, G) > a) --g0;
    while (g1 + 1 < G && wg_lo(g1 + 1, plan.U, G) <= b - 1) ++g1;
    while (g1 > 0 && wg_lo(g1, plan.U, G) > b - 1) --g1;
    if (g0 == g1) return;                               // one workgroup had the whole tile and added it itself
  }
  const int co_tiles = p.Cout / BM, ci_tiles = p.Cin / BN;
  const int cot = T % co_tiles, rest = T / co_tiles, cit = rest % ci_tiles, ky = rest / ci_tiles;
  const long Kw = 9L * p.Cin;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < TILE_FLOATS / 4; e += gridDim.x * blockDim.x) {
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (long g = g0; g <= g1; ++g) {
      const long sl = plan.mode == 1 ? slab0 + g : g * 2 + (wg_lo(g, plan.U, G) >= a ? 0 : 1);
      const float4 v = ((const float4*)(ws + sl * TILE_FLOATS))[e];
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
    const int f = e * 4, t = f / (BM * BN), col = (f - t * BM * BN) / BN, cil = f % BN;
    float4* q = (float4*)(p.dw + (long)(cot * BM + col) * Kw + (long)(ky * 3 + t) * p.Cin + cit * BN + cil);
    if (!(p.flags & 1)) { const float4 o = *q; s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w; }
    *q = s;
  }
}

}  // namespace

namespace l2s {


bool wgrad_row3_dma_ok(const l2s_wgrad_prob& q) {
  if (!(q.KH == 3 && q.KW == 3 && q.stride == 1 && q.pad == 1) || q.Cin % BN || q.Cout % BM || q.split > 1) return false;
  for (int s = 0; s < q.nseg; ++s)
    if (q.OH[s] != q.IH[s] || q.OW[s] != q.IW[s] || (q.OW[s] + 1) * q.OH[s] < 1) return false;
  return true;
}
long wgrad_row3_dma_tiles(int Cin, int Cout) { return (long)(Cout / BM) * (Cin / BN) * 3; }
size_t wgrad_row3_dma_ws_bytes(int G) { return (size_t)G * 3 * TILE_FLOATS * sizeof(float); }   // (upper bound for up to 23 filter-row groups)

int wgrad_row3_dma_launch(const l2s_wgrad_prob* tab_dev, const l2s_wgrad_prob* tab_host, int nprob, float* ws, size_t ws_bytes, int G, hipStream_t st) {
  if (G < 1 || nprob < 1 || nprob > L2S_WGRAD_MAX_GROUP) return L2S_EINVAL;
  wgrad_sk_plan plan;
  plan.n = nprob;
  long U = 0;
  int total_tiles = 0;
  for (int i = 0; i < nprob; ++i) {
    const l2s_wgrad_prob& q = tab_host[i];
    if (!wgrad_row3_dma_ok(q)) return L2S_EINVAL;
    int S = 0;
    for (int s = 0; s < q.nseg; ++s) S += cdiv((long)q.n_img[s] * q.OH[s] * (q.OW[s] + 1), BKP);
    const int tiles = (int)wgrad_row3_dma_tiles(q.Cin, q.Cout);
    plan.unit0[i] = U; plan.S[i] = S;
    U += (long)tiles * S;
    total_tiles += tiles;
  }
  for (int i = nprob; i <= L2S_WGRAD_MAX_GROUP; ++i) plan.unit0[i] = U;
  for (int i = nprob; i < L2S_WGRAD_MAX_GROUP; ++i) plan.S[i] = 1;
  plan.U = U;
  if (U < 1) return L2S_OK;
  if ((long)G > U) G = (int)U;
  // few tiles (layer2's 128-channel 3x3 problems: 12 tiles of 147 slices): every workgroup leaves up to two 192-KiB slabs for the second
  // launch to add, so at least 24 slices of work per workgroup (a multiple of 8 workgroups: the XCD-aware order)
  if (U / 24 < G) { G = (int)(U / 24) & ~7; if (G < 8) G = U < 8 ? (int)U : 8; }
  // XCD-lockstep plan: with the contiguous ranges above, the workgroups of an XCD work on DIFFERENT slices of their tiles at any moment, so
  // every operand slab comes through the fabric once per workgroup (1.25 GB per launch at ~10 TB/s: 118 us, as long as the MFMAs).  If
  // the tiles form groups (problem, filter row) of T tiles that share their dY / X slabs 4 ways, an XCD's G / 8 workgroups take ONE
  // group and walk its slices together - each slab then crosses the fabric once per XCD - and the groups left over after 8 q are
  // spread over all XCDs, slices cut finer.  Needs equal S, T | G / 8 and r | 8.
  plan.mode = 0; plan.T = plan.Ng = plan.k1 = plan.Q = plan.r = plan.m = 0;
  if (l2s_knobs::row3_plan_mode != 0 && (G & 7) == 0) {   // (tools build only: measured 272-281 us against 265-268 for the contiguous ranges)
    const int T = (int)wgrad_row3_dma_tiles(tab_host[0].Cin, tab_host[0].Cout) / 3, P = G / 8;
    bool same = true;
    for (int i = 1; i < nprob; ++i) same = same && plan.S[i] == plan.S[0] && tab_host[i].Cin == tab_host[0].Cin && tab_host[i].Cout == tab_host[0].Cout;
    const int Ng = 3 * nprob, Q = Ng / 8, r = Ng - 8 * Q;
    if (same && T >= 1 && P % T == 0 && (r == 0 || 8 % r == 0) && (long)(P / T) * (Q ? 1 : 8 / (r ? r : 1)) <= plan.S[0]) {
      plan.mode = 1; plan.T = T; plan.Ng = Ng; plan.k1 = P / T; plan.Q = Q; plan.r = r; plan.m = r ? 8 / r : 1;
    }
  }
  long nslabs = plan.mode == 1 ? (long)G * (plan.Q + (plan.r ? 1 : 0)) : (long)G * 2;
  if (plan.mode == 1 && ws_bytes < (size_t)nslabs * TILE_FLOATS * sizeof(float)) { plan.mode = 0; nslabs = (long)G * 2; }   // (many groups: the contiguous plan needs 2 slabs per workgroup)
  if (!ws || ws_bytes < (size_t)nslabs * TILE_FLOATS * sizeof(float)) return L2S_EINVAL;
#define GO(KO_, NST_)                                                                                                                          \
  {                                                                                                                                            \
    constexpr size_t lds = (size_t)NST_ * STG;                                                                                                 \
    static bool attr_done = false;                                                                                                             \
    if (!attr_done) { (void)hipFuncSetAttribute((const void*)wgrad_row3_dma_kernel<KO_, NST_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_done = true; } \
    L2S_LAUNCH((wgrad_row3_dma_kernel<KO_, NST_>), dim3(G), dim3(512), lds, st, tab_dev, plan, ws);                                            \
  }
#ifdef L2S_TOOLS
  switch (l2s_knobs::row3_form) {   // knock-out builds of the kernel for tools/wgrad_stamps.py (their results are garbage); not in the product library
    case 1: GO(1, 4) break; case 2: GO(2, 4) break; case 5: GO(5, 4) break; case 6: GO(6, 4) break; case 7: GO(7, 4) break;
    case 8: GO(8, 4) break; case 12: GO(12, 4) break; case 24: GO(24, 4) break; case 10: GO(10, 4) break;
    default: GO(0, 4) break;
  }
#else
  GO(0, 4)
#endif
#undef GO
  const float* wsc = ws;
  L2S_LAUNCH(wgrad_row3_reduce_kernel, dim3(12, total_tiles), dim3(256), 0, st, tab_dev, plan, wsc, G);
  return l2s_check_launch();
}

}  // namespace l2s
