// RoI path of the lang2seg train step on gfx950, fully on device (no host round trips):
// RPN softmax + box decode, stable top-k sort, greedy NMS (bitmask + single-workgroup scan), anchor / proposal
// target assignment with key-based sampling, mask targets and crop-and-resize RoIAlign.
// Reference: pyutils/mask-faster-rcnn/lib/{layer_utils/proposal_layer.py, anchor_target_layer.py,
// proposal_target_layer.py, model/bbox_transform.py, utils/bbox.py, nms/src/nms.c, nms/src/cuda/nms_kernel.cu,
// nms/src/nms_cuda.c} and nets/network_cycle_res5_2.py:107-149.  Integer / byte / compare work: HBM- and
// latency-bound, wave64-native (one u64 NMS mask word per lane).
#include "common.h"
#include "roi_sample.h"
#include "../../include/lang2seg_hip.h"

namespace {

// ------------------------------------------------------------------ RPN decode
__global__ void rpn_decode_kernel(const float* __restrict__ heads, int ldh, const float* __restrict__ base, int H, int W, int A,
                                  int fs, float im_h, float im_w, float* prob, float* boxes, float* scores) {
  const int n = H * W * A;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int a = i % A, pix = i / A, w = pix % W, h = pix / W;
    const float* hr = heads + (long)pix * ldh;
    const float bg = hr[a], fg = hr[A + a];
    const float m = fmaxf(bg, fg);
    const float e0 = expf(bg - m), e1 = expf(fg - m);
    const float s = e0 + e1;
    prob[(long)pix * 2 * A + a] = e0 / s;
    prob[(long)pix * 2 * A + A + a] = e1 / s;
    scores[i] = e1 / s;
    // anchors (snippets.py:13-29) and bbox_transform_inv (bbox_transform.py:36-62), fp32
    const float ax1 = base[a * 4 + 0] + (float)(w * fs), ay1 = base[a * 4 + 1] + (float)(h * fs);
    const float ax2 = base[a * 4 + 2] + (float)(w * fs), ay2 = base[a * 4 + 3] + (float)(h * fs);
    const float aw = ax2 - ax1 + 1.0f, ah = ay2 - ay1 + 1.0f;
    const float cx = ax1 + 0.5f * aw, cy = ay1 + 0.5f * ah;
    const float* d = hr + 2 * A + a * 4;
    const float pcx = d[0] * aw + cx, pcy = d[1] * ah + cy;
    const float pw = expf(d[2]) * aw, ph = expf(d[3]) * ah;
    float x1 = pcx - 0.5f * pw, y1 = pcy - 0.5f * ph, x2 = pcx + 0.5f * pw, y2 = pcy + 0.5f * ph;
    x1 = fminf(fmaxf(x1, 0.f), im_w - 1.f); y1 = fminf(fmaxf(y1, 0.f), im_h - 1.f);
    x2 = fminf(fmaxf(x2, 0.f), im_w - 1.f); y2 = fminf(fmaxf(y2, 0.f), im_h - 1.f);
    *(float4*)(boxes + (long)i * 4) = make_float4(x1, y1, x2, y2);
  }
}

// ------------------------------------------------------------------ stable descending rank sort
__device__ __forceinline__ uint64_t sort_key(float s, int idx) {
  uint32_t u = __float_as_uint(s);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);       // order-preserving
  return ((uint64_t)u << 32) | (uint32_t)(0xFFFFFFFFu - (uint32_t)idx);  // ties: lower index ranks higher
}
// Stable LSD radix sort (4 passes x 8-bit digits) of (inverted order-preserving score key, index): ascending on the
// inverted key = descending score, stability = lower index first on ties.  Cost is independent of the score distribution
// (freshly initialised RPNs emit 28 728 scores within 1e-3 of 0.5).  Per pass: per-block LDS histogram -> single-workgroup
// exclusive scan over (digit, block) -> scatter with an in-LDS stable local rank.  Blocks of 256 consecutive elements.
__device__ __forceinline__ unsigned int okey(float s) { unsigned int u = __float_as_uint(s); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
// workspace: two (key, index) buffers + one table of per-block digit counts per pass, hist[pass][block][digit]
struct RsWs { unsigned int* kA; int* iA; unsigned int* kB; int* iB; int* hist; };
__device__ __host__ __forceinline__ RsWs rs_ws(int* ws, int n) {
  RsWs w; w.kA = (unsigned int*)ws; w.iA = ws + n; w.kB = (unsigned int*)(ws + 2 * n); w.iB = ws + 3 * n; w.hist = ws + 4 * n;
  return w;
}
// pass 0 counts (keys come straight from the scores) and the clear of the other three tables
__global__ __launch_bounds__(256) void rs_hist0_kernel(const float* scores, int n, int nblk, int* hist) {
  __shared__ int h[256];
  const int t = threadIdx.x, b = blockIdx.x;
  h[t] = 0;
  __syncthreads();
  const int i = b * 256 + t;
  if (i < n) atomicAdd(&h[(~okey(scores[i])) & 255], 1);
  __syncthreads();
  hist[b * 256 + t] = h[t];
#pragma unroll
  for (int p = 1; p < 4; ++p) hist[(p * nblk + b) * 256 + t] = 0;
}
// One pass = ONE launch (round 2: counts, a single-workgroup scan of 28 928 counters, scatter = three dependent launches per pass, the
// scan alone 15 us): every block derives its own bases from the table of the pass (thread d sums digit d over all blocks - coalesced,
// 113 KB from L2 - and over the blocks before this one; a 256-wide exclusive scan of the totals in LDS), ranks its elements stably
// (wave ballots: lanes with the same digit, + the counts of the waves before), scatters, and counts the NEXT pass's digit of every
// element at its destination block with a global atomic.  The last pass writes the sorted (index, score, box) rows of the top k itself.
template <int PASS>
__global__ __launch_bounds__(256) void rs_pass_kernel(const float* __restrict__ scores, const float* __restrict__ boxes, const unsigned int* __restrict__ kin,
                                                      const int* __restrict__ iin, unsigned int* __restrict__ kout, int* __restrict__ iout, int n, int nblk,
                                                      int* __restrict__ hist, int k, float* __restrict__ sboxes, float* __restrict__ sscores, int* __restrict__ sidx) {
  __shared__ int tot[256], base[256], wcnt[4][256];
  const int t = threadIdx.x, b = blockIdx.x, lane = t & 63, wave = t >> 6;
  const int i = b * 256 + t;
  unsigned int key = 0; int idx = 0;
  const bool valid = i < n;
  if (valid) {
    if (PASS == 0) { key = ~okey(scores[i]); idx = i; } else { key = kin[i]; idx = iin[i]; }
  }
  const int d = (int)((key >> (8 * PASS)) & 255u);
  // ---- bases: digit t over all blocks / over the blocks before this one ----
  // (round 5: as 16-byte loads, sixteen in flight per thread - thread (q, g) sums digits 4q .. 4q + 3 over blocks g, g + 4, ... - and one LDS
  // exchange; one 4-byte column per thread, eight loads at a time, was fourteen dependent L2 round trips, ~8 of the pass's 11 us)
  __shared__ int ptot[4][256], pbef[4][256];
  {
    const int q = t & 63, g = t >> 6;
    const int4* hp4 = (const int4*)(hist + (long)PASS * nblk * 256) + q;
    int4 tv = make_int4(0, 0, 0, 0), bv = make_int4(0, 0, 0, 0);
    for (int b0 = g; b0 < nblk; b0 += 64) {
      int4 v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = (b0 + 4 * u < nblk) ? hp4[(long)(b0 + 4 * u) * 64] : make_int4(0, 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        tv.x += v[u].x; tv.y += v[u].y; tv.z += v[u].z; tv.w += v[u].w;
        if (b0 + 4 * u < b) { bv.x += v[u].x; bv.y += v[u].y; bv.z += v[u].z; bv.w += v[u].w; }
      }
    }
    *(int4*)&ptot[g][4 * q] = tv; *(int4*)&pbef[g][4 * q] = bv;
  }
  __syncthreads();
  const int total = ptot[0][t] + ptot[1][t] + ptot[2][t] + ptot[3][t];
  const int before = pbef[0][t] + pbef[1][t] + pbef[2][t] + pbef[3][t];
#pragma unroll
  for (int w = 0; w < 4; ++w) wcnt[w][t] = 0;
  // inclusive scan of the 256 totals: shuffles inside a wave + the three wave sums through LDS (sixteen barriers before)
  int incl = total;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { const int x = __shfl_up(incl, o); if (lane >= o) incl += x; }
  if (lane == 63) tot[wave] = incl;
  __syncthreads();
  for (int w = 0; w < wave; ++w) incl += tot[w];
  base[t] = incl - total + before;                     // elements with a smaller digit anywhere + this digit in earlier blocks
  // ---- stable rank inside the block ----
  unsigned long long same = valid ? ~0ull : 0ull;      // lanes of this wave with the same digit
#pragma unroll
  for (int bit = 0; bit < 8; ++bit) {
    const unsigned long long bal = __ballot(valid && ((d >> bit) & 1));
    same &= ((d >> bit) & 1) ? bal : ~bal;
  }
  same &= __ballot(valid);
  const int below = __popcll(same & ((1ull << lane) - 1ull));
  if (valid && below == 0) wcnt[wave][d] = __popcll(same);
  __syncthreads();
  if (valid) {
    int rank = below;
    for (int w = 0; w < wave; ++w) rank += wcnt[w][d];
    const int pos = base[d] + rank;
    if (PASS < 3) {
      kout[pos] = key; iout[pos] = idx;
      atomicAdd(&hist[((long)(PASS + 1) * nblk + (pos >> 8)) * 256 + ((key >> (8 * (PASS + 1))) & 255u)], 1);
    } else if (pos < k) {
      sidx[pos] = idx; sscores[pos] = scores[idx];
      *(float4*)(sboxes + (long)pos * 4) = *(const float4*)(boxes + (long)idx * 4);
    }
  }
}

// ------------------------------------------------------------------ NMS
// ovr = inter / (area_a + area_b - inter) against thr (nms.c:55-59 `>=`, nms_kernel.cu:56-66 `>`), bit for bit: the IEEE division is only
// evaluated when inter lies within 2^-18 (relative) of thr * union - everywhere else the comparison cannot come out differently (the
// rounding of thr * union and of the quotient are 2^-24 each), and almost every pair of a 12000-box problem is far outside the band.
__device__ __forceinline__ bool nms_hit(const float4& a, float aarea, const float4& b, float barea, float thr, int cmp) {
  const float xx1 = fmaxf(a.x, b.x), yy1 = fmaxf(a.y, b.y), xx2 = fminf(a.z, b.z), yy2 = fminf(a.w, b.w);
  const float w = fmaxf(0.f, xx2 - xx1 + 1.f), h = fmaxf(0.f, yy2 - yy1 + 1.f);
  const float inter = w * h;
  const float uni = aarea + barea - inter;
  const float q = thr * uni;
  if (inter < q * (1.f - 0x1p-18f)) return false;
  if (inter > q * (1.f + 0x1p-18f)) return true;
  const float ovr = inter / uni;                          // nms.c:55-58
  return cmp ? (ovr > thr) : (ovr >= thr);
}
// Round 5: greedy NMS in STAGES of NMS_SB blocks of 64 boxes (4096 boxes), column form.
//   In the training step the scan runs to the LAST of the 12000 sorted boxes (721 - 2000 of them survive with untrained and early-trained
//   RPN heads: tools/proposal_depth.py), so the cost is the 188 dependent block resolutions, not the bytes.  Round 4's scan kept the mask in
//   ROW form (row i = the boxes i suppresses) and fetched the rows of kept boxes - loads that DEPEND on the previous resolutions, two
//   phases of slack for a memory round trip: 0.9 - 1.2 us per block, 225 - 255 us per call.
//   Now a stage needs (1) what the boxes kept by EARLIER stages suppress among its columns - one OR-reduction per column block (the
//   "carry-in", never stored as a mask; all compute units) - and (2) its own diagonal part in COLUMN form: word T[rb][c][lane] = the boxes of
//   block c that suppress box rb * 64 + lane.  Which words the scan loads no longer depends on what it decides: fifteen waves stream them
//   ahead of the chain wave and AND them with the keep words as those appear; the chain wave finds, per block, one finished OR word, the
//   words of the LAG newest blocks and the block's own diagonal word in LDS.
#ifdef L2S_TOOLS
__device__ long long nms_dbg[16];     // tools build: cycles of the last scan launch - [0] chain total, [1] chain waiting for ready, [2] blocks, [3] wave 1 waiting for loads, [4] wave 1 waiting for keep words, [5] wave 1 total
#define NMS_T() ((long long)__builtin_readcyclecounter())
#else
#define NMS_T() 0ll
#endif
constexpr int NMS_SB = 64;     // blocks of 64 boxes per stage
constexpr int NMS_LAG = 6;     // the newest LAG keep words of a block's predecessors are applied by the chain wave itself
constexpr int NMS_RING = 16;   // blocks whose LDS slots exist at a time (> LAG)
constexpr int NMS_SLOT = NMS_LAG + 2;    // LDS words per box: [0] OR of (T & keep) over the older blocks, [1 .. LAG] raw T words, [LAG + 1] the diagonal word (the earlier boxes of the own block that suppress the box)
// blockIdx.y < sbw: stage-local row block rb = blockIdx.y against column block c = 4 blockIdx.x + wave <= rb -> mask[(rb * sbw + c) * 64 + lane]; for
// c == rb the word holds the earlier boxes of the same block that suppress this one (bits below the lane).  blockIdx.y >= sbw (stages after
// the first): 64 entries of the keep list against column block blockIdx.x -> OR into carry[blockIdx.x] (bit = that box is suppressed).
// nms_hit always gets the earlier (higher-scored) box first, as the row-form kernel of rounds 1-4 did.
// (Four column blocks per workgroup, one per wave: 64-thread workgroups - 2080 to 6000 per launch - took 16 - 18 us beside the caption stream's
// convolutions where they take 7 - 10 alone.)
__global__ __launch_bounds__(256) void nms_mask_kernel(const float* __restrict__ boxes, int n, float thr, int cmp, int row0, int sbw, uint64_t* mask,
                                                       unsigned long long* carry_base, const int* __restrict__ keep, const int* nk_p, int max_keep,
                                                       unsigned long long* carry_all, int carry_words) {
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int cbk = blockIdx.x * 4 + wv;
  int nk = 0;
  if (row0 > 0) { nk = *nk_p; if (nk < 0 || nk >= max_keep) return; }
  else if (blockIdx.x == 0 && blockIdx.y == 0) for (int i = threadIdx.x; i < carry_words; i += 256) carry_all[i] = 0ull;
  const bool diag = (int)blockIdx.y < sbw;
  const int rb = blockIdx.y;
  const bool active = cbk < sbw && (diag ? cbk <= rb : ((int)blockIdx.y - sbw) * 64 < nk);
  __shared__ float4 cbox_s[4][64];
  __shared__ float carea_s[4][64];
  float4* cbox = cbox_s[wv]; float* carea = carea_s[wv];
  const int cj = row0 + cbk * 64 + lane;
  const float4 cbv = (active && cj < n) ? *(const float4*)(boxes + (long)cj * 4) : make_float4(0, 0, -1, -1);
  cbox[lane] = cbv;
  carea[lane] = (cbv.z - cbv.x + 1.f) * (cbv.w - cbv.y + 1.f);
  __syncthreads();
  if (!active) return;
  unsigned long long* carry = carry_base;
  const int csize = min(64, n - row0 - cbk * 64);
  if (diag) {
    const int ri = row0 + rb * 64 + lane;
    if (ri >= n) return;
    const float4 a = *(const float4*)(boxes + (long)ri * 4);
    const float aarea = (a.z - a.x + 1.f) * (a.w - a.y + 1.f);
    uint64_t t = 0;
    if (rb == cbk) {
      for (int j = 0; j < lane; ++j)                     // (the diagonal block too: the earlier boxes of the block that suppress this one)
        if (nms_hit(cbox[j], carea[j], a, aarea, thr, cmp)) t |= 1ull << j;
    } else {
      for (int j = 0; j < 64; ++j)
        if (nms_hit(cbox[j], carea[j], a, aarea, thr, cmp)) t |= 1ull << j;
    }
    mask[(long)(rb * sbw + cbk) * 64 + lane] = t;
    return;
  }
  const int ki = ((int)blockIdx.y - sbw) * 64 + lane;
  uint64_t t = 0;
  if (ki < nk) {
    const int ri = keep[ki];
    const float4 a = *(const float4*)(boxes + (long)ri * 4);
    const float aarea = (a.z - a.x + 1.f) * (a.w - a.y + 1.f);
    for (int j = 0; j < csize; ++j)
      if (nms_hit(a, aarea, cbox[j], carea[j], thr, cmp)) t |= 1ull << j;
  }
  unsigned int lo = (unsigned int)t, hi = (unsigned int)(t >> 32);
#pragma unroll
  for (int off = 32; off; off >>= 1) { lo |= __shfl_xor(lo, off); hi |= __shfl_xor(hi, off); }
  if (lane == 0 && (lo | hi)) atomicOr(&carry[cbk], ((unsigned long long)hi << 32) | lo);
}
// The scan of one stage (nms.c:35-63 / nms_cuda.c:47-58): n boxes / cb blocks, numbered from row0 in the keep list; the keep count starts
// from *num_out in the stages after the first, which return at once when the list is already full (RPN_POST_NMS_TOP_N).  One workgroup of
// sixteen waves, no barrier inside the scan: the waves meet through LDS words.
//   wave 0 (the chain), block b: waits for ready[b]; removed = carry | slot[0] | (slot[1 .. LAG] & the keep words of the LAG previous blocks,
//           which it holds itself); then the block itself: kept = alive & ~(suppressed by a kept earlier box of the block), swept to its fixed point;
//           publishes the block's keep word and kdone = b + 1.
//   waves 1 .. 15, wave w owns blocks w - 1, w + 14, ...: loads the block's column-form words (whatever the chain decides), ANDs word c with
//           keep word c as soon as kdone > c, stores the OR, the raw words of the newest LAG blocks and the diagonal word into the block's LDS
//           slot, and sets ready[b].
// A wave that has waited 2^22 polls gives up (it cannot happen while the other waves of the workgroup run): it writes the error sentinel
// *num_out = -1 and ends; the waves that wait for it time out the same way, the later stages return at once on the sentinel, and the
// consumers of the count (gather_rois, the samplers) treat it as an empty list - the context survives, the caller sees num_out < 0.
__global__ __launch_bounds__(1024) void nms_reduce_kernel(const uint64_t* __restrict__ mask, int n, int cb, int max_keep, int* keep, int* num_out,
                                                          int row0, const unsigned long long* __restrict__ carry) {
  constexpr int LAG = NMS_LAG, SLOT = NMS_SLOT, NBULK = 15;
  extern __shared__ unsigned long long slots[];     // [RING][SLOT][64], then keep words [cb], then carry words [cb]
  __shared__ int kdone_sh, stop_sh;
  __shared__ int ready_sh[NMS_SB];
  unsigned long long* ksh = slots + (size_t)NMS_RING * SLOT * 64;
  unsigned long long* csh = ksh + cb;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (scalar: everything a wave branches on is uniform)
  int nk0 = 0;
  if (row0 > 0) { nk0 = *num_out; if (nk0 < 0 || nk0 >= max_keep) return; }
  for (int c = tid; c < cb; c += 1024) { ksh[c] = 0ull; csh[c] = carry ? carry[c] : 0ull; ready_sh[c] = 0; }
  if (tid == 0) { kdone_sh = 0; stop_sh = 0; }
  __syncthreads();
  // The LDS words the waves meet through are plain volatile accesses between compiler barriers: one wave's LDS instructions execute in
  // order, so "data, then flag" needs no wait - and a RELEASE store would wait for vmcnt(0), i.e. for the chain wave's own store into the
  // keep list (2 - 3 us per block; measured: 535 us per call).
  // (through address-space-3 pointers: a volatile access through a generic pointer stays a FLAT load - the vector memory path, sc0 sc1, and
  // an s_waitcnt vmcnt(0) that also waits for every global load in flight; measured: 4300 cycles from the last keep word to a block's ready)
  typedef __attribute__((address_space(3))) volatile int lds_vint;
  typedef __attribute__((address_space(3))) volatile unsigned long long lds_vu64;
  auto lds_ld = [](const int* p) { const int v = *(const lds_vint*)p; asm volatile("" ::: "memory"); return v; };
  auto lds_st = [](int* p, int v) { asm volatile("" ::: "memory"); *(lds_vint*)p = v; };
  auto lds_ld64 = [](const unsigned long long* p) { return *(const lds_vu64*)p; };
  auto lds_st64 = [](unsigned long long* p, unsigned long long v) { *(lds_vu64*)p = v; };
  auto poll = [&](int* p, int want) {                     // until *p >= want
    int spins = 0;
    while (lds_ld(p) < want) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1 << 22)) { *(volatile int*)num_out = -1; __builtin_amdgcn_endpgm(); }   // give up: the count becomes the error sentinel, this wave ends
    }
  };
  if (wave == 0) {
    // ---------------- the chain ----------------
    __builtin_amdgcn_s_setprio(3);
    int nk = nk0;
    unsigned long long kprev[LAG];                          // keep words of blocks b-1, b-2, ...
#pragma unroll
    for (int k = 0; k < LAG; ++k) kprev[k] = 0ull;
    long long t_wait = 0, t_res = 0; const long long t_begin = NMS_T(); int nb_done = 0;
    for (int b = 0; b < cb; ++b) {
      const long long tp = NMS_T();
      poll(&ready_sh[b], 1);
      const long long tr = NMS_T();
      t_wait += tr - tp; ++nb_done;
      const unsigned long long* sl = slots + (size_t)(b % NMS_RING) * SLOT * 64 + lane;
      unsigned long long hit = sl[0];
#pragma unroll
      for (int k = 1; k <= LAG; ++k) hit |= sl[k * 64] & kprev[LAG - k];        // slot k holds column block b - LAG + k - 1
      const unsigned long long d = sl[(LAG + 1) * 64];
      unsigned long long rb = __ballot(hit != 0ull) | csh[b];
      const int lim = min(64, n - b * 64);
      if (lim < 64) rb |= ~0ull << lim;
      // in-block resolution: kept = alive & ~{ i : a KEPT earlier box of the block suppresses i } - the greedy recursion has exactly one solution, and
      // sweeping from "every alive box is kept" reaches it in (longest suppression chain among the alive boxes) + 1 sweeps, 2 - 4 in practice; a sweep is
      // two ANDs, a compare and a ballot (the walk over the row-form word of rounds 3 - 5 cost a readlane round trip per kept box, ~100 cycles each)
      const unsigned long long alive = ~rb;
      unsigned long long kept = alive;
      for (int sweep = 0; sweep < 65; ++sweep) {
        const unsigned long long nk2 = alive & ~__ballot((d & kept) != 0ull);
        if (nk2 == kept) break;
        kept = nk2;
      }
      rb = ~kept;
      const unsigned long long K = ~rb;
      if (lane == 0) {
        lds_st64(&ksh[b], K);
        lds_st(&kdone_sh, b + 1);
      }
      t_res += NMS_T() - tr;
      const int row = b * 64 + lane;
      if ((K >> lane) & 1ull) { const int pos = nk + __popcll(K & ((1ull << lane) - 1ull)); if (pos < max_keep) keep[pos] = row0 + row; }
      nk += __popcll(K);
#pragma unroll
      for (int k = LAG - 1; k > 0; --k) kprev[k] = kprev[k - 1];
      kprev[0] = K;
      if (nk >= max_keep) break;
    }
    if (lane == 0) {
#ifdef L2S_TOOLS
      nms_dbg[0] = NMS_T() - t_begin; nms_dbg[1] = t_wait; nms_dbg[2] = nb_done; nms_dbg[7] = t_res;
#endif
      lds_st(&stop_sh, 1);
      lds_st(&kdone_sh, 1 << 20);
      *num_out = min(nk, max_keep);
    }
    return;
  }
  // ---------------- the column words, ahead of the chain ----------------
  // A block's words are loaded in batches of 32 (all in flight at once; which words is known without the chain), ANDed with the keep words that
  // are already published when the batch arrives, and once more after ONE wait for keep word b - LAG - 1 for the rest: the path from that keep
  // word to ready[b] is one LDS read and a few readlane / AND / OR per late word, and the chain has LAG blocks of its own to cover it.
  long long t_ld = 0, t_k = 0, t_tail = 0; const long long t_begin = NMS_T();
  for (int b = wave - 1; b < cb; b += NBULK) {
    if (lds_ld(&stop_sh)) break;
    const uint64_t* tb = mask + (long)b * cb * 64 + lane;
    unsigned long long* sl = slots + (size_t)(b % NMS_RING) * SLOT * 64 + lane;
    const int nold = b - LAG;                                // column blocks [0, nold) are ANDed here
    unsigned long long acc = 0ull;
    // the words that go to the slot as they are: column blocks nold .. b - 1 and the diagonal word (index b).  Straight-line code throughout:
    // a taken scalar branch costs ~20 cycles, and 64 of them per block (a guard per word) were 4600 cycles between the last keep word and ready.
    unsigned long long raw[LAG + 1];
#pragma unroll
    for (int k = 0; k <= LAG; ++k) raw[k] = tb[(long)max(nold + k, 0) * 64];
    for (int base = 0; base < nold; base += 32) {
      unsigned long long tw[32];
      const long long tl = NMS_T();
#pragma unroll
      for (int i = 0; i < 32; ++i) tw[i] = tb[(long)min(base + i, b) * 64];      // (past nold the word meets a zero keep word)
      // (left to itself the compiler sinks every load next to its use, below the polls - one memory round trip per word)
#pragma unroll
      for (int i = 0; i < 32; ++i) asm volatile("" : "+v"(tw[i]));
      t_ld += NMS_T() - tl;
      const int hi = min(nold, base + 32);                   // this batch ANDs column blocks [base, hi)
      int from = base;
      for (int pass = 0; pass < 2 && from < hi; ++pass) {
        int have = __builtin_amdgcn_readfirstlane(lds_ld(&kdone_sh));
        if (pass == 1) {                                     // (the chain sets kdone past every block when it stops early: no wave waits forever)
          const long long tk = NMS_T();
          int spins = 0;
          while (have < hi) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) { *(volatile int*)num_out = -1; __builtin_amdgcn_endpgm(); }   // give up: the count becomes the error sentinel, this wave ends
            have = __builtin_amdgcn_readfirstlane(lds_ld(&kdone_sh));
          }
          t_k += NMS_T() - tk;
        }
        const int to = min(have, hi);
        // one LDS read for the batch's keep words (lane i reads word base + i), handed out by readlane
        const unsigned long long kw = (lane < 32 && base + lane >= from && base + lane < to) ? lds_ld64(&ksh[base + lane]) : 0ull;
        const unsigned int klo = (unsigned int)kw, khi = (unsigned int)(kw >> 32);
#pragma unroll
        for (int i = 0; i < 32; ++i) {
          const unsigned int k0 = __builtin_amdgcn_readlane(klo, i), k1 = __builtin_amdgcn_readlane(khi, i);
          acc |= tw[i] & (((unsigned long long)k1 << 32) | k0);          // (words outside [from, to) meet a zero keep word)
        }
        from = to;
      }
    }
    // slot b % RING was block b - RING's, which the chain has left (kdone >= nold > b - RING) once the last wait above is over; the first
    // RING blocks of a stage find theirs unused
    const long long tt = NMS_T();
#pragma unroll
    for (int k = 0; k < LAG; ++k) sl[(k + 1) * 64] = nold + k >= 0 ? raw[k] : 0ull;      // (a stage's first blocks have fewer than LAG predecessors)
    sl[(LAG + 1) * 64] = raw[LAG];
    t_tail += NMS_T() - tt;
    lds_st64(&sl[0], acc);
    lds_st(&ready_sh[b], 1);                                // (every lane, behind its own slot words)
  }
#ifdef L2S_TOOLS
  if (wave == 1 && lane == 0) { nms_dbg[3] = t_ld; nms_dbg[4] = t_k; nms_dbg[5] = NMS_T() - t_begin; nms_dbg[6] = t_tail; }
#endif
}
__global__ void gather_rois_kernel(const float* sboxes, const float* sscores, const int* keep, const int* num, int max_keep, float* rois, float* rs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= max_keep) return;
  float4 b = make_float4(0, 0, 0, 0); float sc = 0.f;
  if (i < *num) { int k = keep[i]; b = *(const float4*)(sboxes + (long)k * 4); sc = sscores[k]; }
  rois[i * 5 + 0] = 0.f; rois[i * 5 + 1] = b.x; rois[i * 5 + 2] = b.y; rois[i * 5 + 3] = b.z; rois[i * 5 + 4] = b.w;
  rs[i] = sc;
}

// ------------------------------------------------------------------ anchor targets
struct AtlWs { int* cnt; unsigned long long* gtmax; double* maxov; int* argmax; int* lab; };
__device__ __forceinline__ AtlWs atl_ws(int* ws, int n) {
  AtlWs w;
  w.cnt = ws; w.gtmax = (unsigned long long*)(ws + 16); w.maxov = (double*)(ws + 16 + 64);
  w.argmax = ws + 16 + 64 + 2 * n; w.lab = w.argmax + n;
  return w;
}
__device__ __forceinline__ void anchor_at(const float* base, int i, int A, int W, int fs, float& x1, float& y1, float& x2, float& y2) {
  const int a = i % A, pix = i / A, w = pix % W, h = pix / W;
  x1 = base[a * 4 + 0] + (float)(w * fs); y1 = base[a * 4 + 1] + (float)(h * fs);
  x2 = base[a * 4 + 2] + (float)(w * fs); y2 = base[a * 4 + 3] + (float)(h * fs);
}
__device__ __forceinline__ double iou64(float ax1, float ay1, float ax2, float ay2, const float* g) {
  // utils/bbox.py:21-29 evaluated in float64 (anchor_target_layer.py:62-64)
  const double bx1 = ax1, by1 = ay1, bx2 = ax2, by2 = ay2, qx1 = g[0], qy1 = g[1], qx2 = g[2], qy2 = g[3];
  const double ba = (bx2 - bx1 + 1) * (by2 - by1 + 1), qa = (qx2 - qx1 + 1) * (qy2 - qy1 + 1);
  double iw = fmin(bx2, qx2) - fmax(bx1, qx1) + 1; if (iw < 0) iw = 0;
  double ih = fmin(by2, qy2) - fmax(by1, qy1) + 1; if (ih < 0) ih = 0;
  return iw * ih / (ba + qa - iw * ih);
}
__global__ void atl_init_kernel(int* ws) { if (threadIdx.x < 16 + 64) ws[threadIdx.x] = 0; }
__global__ void atl_iou_kernel(const float* gt, int n_gt, const float* base, int n, int W, int A, int fs, float im_h, float im_w, int* ws) {
  AtlWs w = atl_ws(ws, n);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x1, y1, x2, y2; anchor_at(base, i, A, W, fs, x1, y1, x2, y2);
  const bool inside = x1 >= 0.f && y1 >= 0.f && x2 < im_w && y2 < im_h;
  double best = -1.0; int arg = 0;
  if (inside) {
    for (int g = 0; g < n_gt; ++g) {
      double o = iou64(x1, y1, x2, y2, gt + g * 5);
      if (o > best) { best = o; arg = g; }
      atomicMax(&w.gtmax[g], (unsigned long long)__double_as_longlong(o));
    }
  }
  w.maxov[i] = best; w.argmax[i] = arg;
  w.lab[i] = inside ? -1 : -2;    // -2: outside the image, never a candidate
}
__global__ void atl_label_kernel(const float* gt, int n_gt, const float* base, int n, int W, int A, int fs, float neg_ov, float pos_ov, int* ws) {
  AtlWs w = atl_ws(ws, n);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i < n && w.lab[min(i, n - 1)] != -2;
  int l = -1;
  if (live) {
    float x1, y1, x2, y2; anchor_at(base, i, A, W, fs, x1, y1, x2, y2);
    const double mo = w.maxov[i];
    if (mo < (double)neg_ov) l = 0;                               // ATL:75
    for (int g = 0; g < n_gt; ++g)
      if (iou64(x1, y1, x2, y2, gt + g * 5) == __longlong_as_double((long long)w.gtmax[g])) l = 1;   // ATL:70,78 (all ties)
    if (mo >= (double)pos_ov) l = 1;                              // ATL:81
    w.lab[i] = l;
  }
  // one counter update per wave, not per anchor: 28 728 same-address atomics serialise at the memory side (this kernel took 157 us)
  const unsigned long long m1 = __ballot(l == 1), m0 = __ballot(l == 0);
  if ((threadIdx.x & 63) == (unsigned)__builtin_ctzll(__ballot(1))) {
    if (m1) atomicAdd(&w.cnt[0], __popcll(m1));
    if (m0) atomicAdd(&w.cnt[1], __popcll(m0));
  }
}
// disable the D smallest-key candidates with label == which (single workgroup, radix select on 32-bit keys)
__device__ void select_disable(int* lab, const uint32_t* keys, int n, int which, int D, int* hist /*256*/, int* sh /*4*/) {
  const int tid = threadIdx.x;
  if (D <= 0) return;
  uint32_t prefix = 0, pmask = 0; int need = D;   // find key value t: #(key < t) < D <= #(key <= t)
  for (int pass = 3; pass >= 0; --pass) {
    for (int b = tid; b < 256; b += blockDim.x) hist[b] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += blockDim.x)
      if (lab[i] == which && (keys[i] & pmask) == prefix) atomicAdd(&hist[(keys[i] >> (8 * pass)) & 255], 1);
    __syncthreads();
    if (tid == 0) {
      int acc = 0, b = 0;
      for (; b < 256; ++b) { if (acc + hist[b] >= need) break; acc += hist[b]; }
      sh[0] = b; sh[1] = need - acc;
    }
    __syncthreads();
    prefix |= ((uint32_t)sh[0]) << (8 * pass); pmask |= 255u << (8 * pass); need = sh[1];
    __syncthreads();
  }
  // disable key < prefix, and the first `need` (by index) of key == prefix
  if (tid == 0) sh[2] = 0;
  __syncthreads();
  for (int i0 = 0; i0 < n; i0 += blockDim.x) {
    const int i = i0 + tid;
    bool cand = i < n && lab[i] == which;
    if (cand && keys[i] < prefix) { lab[i] = -1; cand = false; }
    const bool tie = cand && keys[i] == prefix;
    // ties are rare: serialise through an atomic ticket in index order per chunk
    if (tie) { int tk = atomicAdd(&sh[2], 1); if (tk < need) lab[i] = -1; }
    __syncthreads();
  }
}
// On-chip variant of select_disable for n <= 31 * 1024 candidates (a 38x63 map with 12 anchors has 28 728): keys (4 B) and labels
// (1 B) sit in LDS (<= 156 KiB), so the four radix passes and the disabling sweep never leave the CU; the looping version above
// re-reads lab / keys from L2 in a dependent loop for every pass (28 round trips per pass on one CU: 134 us for both calls).
__device__ __forceinline__ void select_disable_lds(signed char* lab, const uint32_t* key, int n, int which, int D, int* hist /*256*/, int* sh /*24*/) {
  const int tid = threadIdx.x;
  if (D <= 0) return;                                    // (uniform: D comes from the shared counters)
  uint32_t prefix = 0, pmask = 0; int need = D;
  for (int pass = 3; pass >= 0; --pass) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
#pragma unroll 4
    for (int i = tid; i < n; i += 1024) {
      const uint32_t k = key[i];
      if (lab[i] == which && (k & pmask) == prefix) atomicAdd(&hist[(k >> (8 * pass)) & 255], 1);
    }
    __syncthreads();
    if (tid < 64) {
      // wave 0: inclusive scan of the 256 bins, 4 per lane; the bin where the running count reaches `need`
      const int c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
      const int tot = c0 + c1 + c2 + c3;
      int inc = tot;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(inc, o, 64); if (tid >= o) inc += v; }
      const int before = inc - tot;
      if (before < need && inc >= need) {
        int acc = before, b = 4 * tid;
        if (acc + c0 < need) { acc += c0; ++b; if (acc + c1 < need) { acc += c1; ++b; if (acc + c2 < need) { acc += c2; ++b; } } }
        sh[0] = b; sh[1] = need - acc;
      }
    }
    __syncthreads();
    prefix |= ((uint32_t)sh[0]) << (8 * pass); pmask |= 255u << (8 * pass); need = sh[1];
    __syncthreads();
  }
  // disable key < prefix and the first `need` (by index) of key == prefix
  if (tid == 0) sh[2] = 0;
  __syncthreads();
  int nt = 0;
#pragma unroll 4
  for (int i = tid; i < n; i += 1024) {
    const uint32_t k = key[i];
    const bool c = lab[i] == which;
    if (c && k < prefix) lab[i] = -1;
    nt += (c && k == prefix);
  }
  if (nt) atomicAdd(&sh[2], nt);
  __syncthreads();
  const int nties = sh[2];
  if (nties == need) {                                   // the usual case: the threshold key occurs exactly as often as it must go
#pragma unroll 4
    for (int i = tid; i < n; i += 1024) if (lab[i] == which && key[i] == prefix) lab[i] = -1;
  } else {
    // equal keys straddle the threshold: the lowest indices go first, one chunk of 1024 at a time
    int left = need;
    for (int i0 = 0; i0 < n && left > 0; i0 += 1024) {   // (`left` is uniform)
      const int i = i0 + tid;
      const bool tie = i < n && lab[i] == which && key[i] == prefix;
      const unsigned long long bm = __ballot(tie);
      const int wv = tid >> 6, ln = tid & 63;
      if (ln == 0) sh[8 + wv] = __popcll(bm);
      __syncthreads();
      int before = 0, total = 0;
      for (int q = 0; q < 16; ++q) { const int c = sh[8 + q]; if (q < wv) before += c; total += c; }
      const int rank = before + __popcll(bm & ((1ull << ln) - 1ull));
      if (tie && rank < left) lab[i] = -1;
      left -= total;
      __syncthreads();
    }
  }
  __syncthreads();
}

constexpr int ATL_LDS_MAX = 31 * 1024;
__global__ __launch_bounds__(1024) void atl_sample_lds_kernel(const uint32_t* fg_keys, const uint32_t* bg_keys, int n, int npad, int batch, float fg_frac, int* ws) {
  extern __shared__ __attribute__((aligned(16))) int lds_all[];       // [256 hist][24 misc][npad keys][npad label bytes]
  int* hist = lds_all; int* sh = lds_all + 256; uint32_t* key = (uint32_t*)(lds_all + 256 + 24);
  signed char* lab = (signed char*)(key + npad);
  AtlWs w = atl_ws(ws, n);
  const int tid = threadIdx.x;
  const int num_fg = (int)(fg_frac * (float)batch);
  const int nfg = w.cnt[0], nbg = w.cnt[1];
#pragma unroll 8
  for (int i = tid; i < n; i += 1024) { lab[i] = (signed char)w.lab[i]; key[i] = fg_keys[i]; }
  __syncthreads();
  select_disable_lds(lab, key, n, 1, nfg - num_fg, hist, sh);          // ATL:88-93
  const int fg_after = min(nfg, num_fg);
  const int num_bg = batch - fg_after;                                // ATL:96
#pragma unroll 8
  for (int i = tid; i < n; i += 1024) key[i] = bg_keys[i];
  __syncthreads();
  select_disable_lds(lab, key, n, 0, nbg - num_bg, hist, sh);         // ATL:97-101
#pragma unroll 8
  for (int i = tid; i < n; i += 1024) w.lab[i] = lab[i];
  if (tid == 0) w.cnt[2] = fg_after + min(nbg, num_bg);               // num_examples = sum(labels >= 0)
}

__global__ __launch_bounds__(1024) void atl_sample_kernel(const uint32_t* fg_keys, const uint32_t* bg_keys, int n, int batch, float fg_frac, int* ws) {
  __shared__ int hist[256];
  __shared__ int sh[4];
  AtlWs w = atl_ws(ws, n);
  const int num_fg = (int)(fg_frac * (float)batch);
  const int nfg = w.cnt[0], nbg = w.cnt[1];
  __syncthreads();
  select_disable(w.lab, fg_keys, n, 1, nfg - num_fg, hist, sh);      // ATL:88-93
  __syncthreads();
  const int fg_after = min(nfg, num_fg);
  const int num_bg = batch - fg_after;                                // ATL:96
  select_disable(w.lab, bg_keys, n, 0, nbg - num_bg, hist, sh);       // ATL:97-101
  __syncthreads();
  if (threadIdx.x == 0) w.cnt[2] = fg_after + min(nbg, num_bg);       // num_examples = sum(labels >= 0)
}
__global__ void atl_out_kernel(const float* gt, const float* base, int n, int H, int W, int A, int fs, const int* ws_c,
                               int* labels, float* targets, float* inw, float* outw) {
  AtlWs w = atl_ws((int*)ws_c, n);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int a = i % A, pix = i / A, ww = pix % W, h = pix / W;
  int l = w.lab[i];
  const bool inside = l != -2;
  if (!inside) l = -1;
  labels[(a * H + h) * W + ww] = l;                                    // ATL:133-134 layout (1,1,A*H,W)
  float t[4] = {0, 0, 0, 0};
  if (inside) {
    float x1, y1, x2, y2; anchor_at(base, i, A, W, fs, x1, y1, x2, y2);
    const float* g = gt + w.argmax[i] * 5;
    const float ew = x2 - x1 + 1.0f, eh = y2 - y1 + 1.0f, ecx = x1 + 0.5f * ew, ecy = y1 + 0.5f * eh;
    const float gw = g[2] - g[0] + 1.0f, gh = g[3] - g[1] + 1.0f, gcx = g[0] + 0.5f * gw, gcy = g[1] + 0.5f * gh;
    t[0] = (gcx - ecx) / ew; t[1] = (gcy - ecy) / eh; t[2] = logf(gw / ew); t[3] = logf(gh / eh);
  }
  const float iw = (l == 1) ? 1.f : 0.f;
  const float ow = (l >= 0) ? (float)(1.0 / (double)w.cnt[2]) : 0.f;
#pragma unroll
  for (int k = 0; k < 4; ++k) { targets[(long)i * 4 + k] = t[k]; inw[(long)i * 4 + k] = iw; outw[(long)i * 4 + k] = ow; }
}


// ------------------------------------------------------------------ RoI max pooling (POOLING_MODE == 'pool')
// layer_utils/roi_pooling/src/cuda/roi_pooling_kernel.cu:15-70 (forward + argmax) and :104-180 (backward), on NHWC maps.
// One thread per (roi, bin, channel); the bin geometry is uniform across the channel lanes of a wave.
__global__ __launch_bounds__(256) void roipool_fwd_kernel(const void* feat, int H, int W, int C, const float* rois, int P, float scale,
                                                         void* out, int* argmax, int dt) {
  const int bin = blockIdx.x;                       // roi * P * P + ph * P + pw
  const int n = bin / (P * P), r = bin - n * P * P, ph = r / P, pw = r - ph * P;
  const float* roi = rois + (long)n * 5;
  const int sw = (int)roundf(roi[1] * scale), sh = (int)roundf(roi[2] * scale);
  const int ew = (int)roundf(roi[3] * scale), eh = (int)roundf(roi[4] * scale);
  const int rw = max(ew - sw + 1, 1), rh = max(eh - sh + 1, 1);
  const float bh = (float)rh / (float)P, bw = (float)rw / (float)P;
  int hs = (int)floorf((float)ph * bh), ws = (int)floorf((float)pw * bw);
  int he = (int)ceilf((float)(ph + 1) * bh), we = (int)ceilf((float)(pw + 1) * bw);
  hs = min(max(hs + sh, 0), H); he = min(max(he + sh, 0), H);
  ws = min(max(ws + sw, 0), W); we = min(max(we + sw, 0), W);
  const bool empty = (he <= hs) || (we <= ws);
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float mx = empty ? 0.f : -3.402823466e+38f; int mi = -1;
    for (int h = hs; h < he; ++h)
      for (int w = ws; w < we; ++w) {
        const float v = ldx(feat, ((long)h * W + w) * C + c, dt);
        if (v > mx) { mx = v; mi = h * W + w; }
      }
    stx(out, (long)bin * C + c, dt, mx);
    argmax[(long)bin * C + c] = mi;
  }
}
__global__ __launch_bounds__(256) void roipool_bwd_kernel(const void* dout, const int* argmax, long n_elem, int C, float* dfeat, int dt) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n_elem; i += (long)gridDim.x * blockDim.x) {
    const int a = argmax[i];
    if (a >= 0) atomicAdd(dfeat + (long)a * C + (int)(i % C), ldx(dout, i, dt));
  }
}

// ------------------------------------------------------------------ proposal targets (single workgroup)
__device__ __forceinline__ float iou32(const float* b, const float* q) {   // utils/bbox.py:21-29 in fp32
  const float ba = (b[2] - b[0] + 1.f) * (b[3] - b[1] + 1.f), qa = (q[2] - q[0] + 1.f) * (q[3] - q[1] + 1.f);
  const float iw = fmaxf(fminf(b[2], q[2]) - fmaxf(b[0], q[0]) + 1.f, 0.f);
  const float ih = fmaxf(fminf(b[3], q[3]) - fmaxf(b[1], q[1]) + 1.f, 0.f);
  const float ua = ba + qa - iw * ih;
  return iw * ih / ua;
}
__global__ __launch_bounds__(1024) void ptl_kernel(const float* rois_in, const float* scores_in, const int* n_rois, int n_max,
                                                   const float* gt, int n_gt, const uint8_t* gt_masks, int im_h, int im_w,
                                                   const uint32_t* fg_keys, const uint32_t* bg_keys, const uint32_t* bg_rand,
                                                   int R, int fg_max, int mask_slots, float fg_thresh, float bg_hi, float bg_lo,
                                                   const float* means4, const float* stds4, const float* inw4, int ncls, int ms,
                                                   float* out_rois, int* labels, float* bt, float* bi, float* bo, float* mt,
                                                   int* counts, int* ws) {
  __shared__ int sh_cnt[4];
  __shared__ int sh_wc[16][2];
  __shared__ unsigned long long s_ki[4096];         // (key << 32) | index: the rank order is one 64-bit compare
  const int tid = threadIdx.x, nt = blockDim.x;
  const int cap = n_max + n_gt;
  int* fg_list = ws; int* bg_list = ws + cap; int* argm = ws + 2 * cap; int* cls = ws + 3 * cap; int* slot = ws + 4 * cap;  // slot[R]
  int n = min(*n_rois, n_max);
  int appended = 0;
  auto roi_ptr = [&](int i, float* b) {
    if (i < n - appended * n_gt) { for (int k = 0; k < 4; ++k) b[k] = rois_in[(long)i * 5 + 1 + k]; }
    else { const float* g = gt + (i - (n - appended * n_gt)) * 5; for (int k = 0; k < 4; ++k) b[k] = g[k]; }
  };
  auto fkey = [&](int i) -> uint32_t { return (i < n - appended * n_gt) ? fg_keys[i] : 0u; };
  auto bkey = [&](int i) -> uint32_t { return (i < n - appended * n_gt) ? bg_keys[i] : 0xFFFFFFFFu; };
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (tid < 4) sh_cnt[tid] = 0;
    __syncthreads();
    // classify candidates; lists are built in index order by a chunked ballot scan
    for (int i0 = 0; i0 < n; i0 += nt) {
      const int i = i0 + tid;
      int kind = 0;  // 1 fg, 2 bg
      if (i < n) {
        float b[4]; roi_ptr(i, b);
        float best = -1.f; int arg = 0;
        for (int g = 0; g < n_gt; ++g) { float o = iou32(b, gt + g * 5); if (o > best) { best = o; arg = g; } }
        argm[i] = arg;
        if (best >= fg_thresh) kind = 1;                                  // PTL:143
        else if (best < bg_hi && best >= bg_lo) kind = 2;                 // PTL:146 (byte add == 2)
        cls[i] = kind;
      }
      // ordered compaction: per-wave ballots, wave bases from the 16 wave counts (index order is kept)
      {
        const unsigned long long mf = __ballot(kind == 1), mb = __ballot(kind == 2);
        const int lane = tid & 63, wv = tid >> 6;
        if (lane == 0) { sh_wc[wv][0] = __popcll(mf); sh_wc[wv][1] = __popcll(mb); }
        __syncthreads();
        int bf = sh_cnt[0], bb = sh_cnt[1], tf = 0, tb = 0;
        for (int w2 = 0; w2 < (nt >> 6); ++w2) {
          const int cf = sh_wc[w2][0], cb = sh_wc[w2][1];
          if (w2 < wv) { bf += cf; bb += cb; }
          tf += cf; tb += cb;
        }
        if (kind == 1) fg_list[bf + __popcll(mf & ((1ull << lane) - 1ull))] = i;
        if (kind == 2) bg_list[bb + __popcll(mb & ((1ull << lane) - 1ull))] = i;
        __syncthreads();
        if (tid == 0) { sh_cnt[0] += tf; sh_cnt[1] += tb; }
        __syncthreads();
      }
    }
    if (sh_cnt[0] > 0 || attempt == 1) break;
    // PTL:159-167: no foreground -> append the gt boxes as candidates and retry
    appended = 1; n += n_gt;
    __syncthreads();
  }
  const int n_fg = sh_cnt[0], n_bg = sh_cnt[1];
  int nfg_sel, nbg_sel; bool fg_repl = false, bg_repl = false;
  if (n_bg > 0) { nfg_sel = min(fg_max, n_fg); nbg_sel = R - nfg_sel; bg_repl = n_bg < nbg_sel; }
  else { nfg_sel = R; nbg_sel = 0; fg_repl = n_fg < R; }
  for (int s = tid; s < R; s += nt) slot[s] = -1;
  __syncthreads();
  // rank by key inside each list; the k smallest keys are emitted in key order (= fg_inds[npr.choice(n,k,False)])
  // LDS path (up to 4096 candidates): bitonic sort of the packed (key, index) words — the O(n^2) rank count of ~1900 background
  // candidates is ~3.6 M compares on ONE compute unit (80+ us); the sort is 66 compare-exchange steps
  auto rank_list = [&](const int* list, int cnt, bool is_fg, int nsel, int slot0) {
    if (cnt <= 4096) {
      int N2 = 2; while (N2 < cnt) N2 <<= 1;
      for (int a = tid; a < N2; a += nt) {
        unsigned long long v = ~0ull;
        if (a < cnt) { const int i = list[a]; v = ((unsigned long long)(is_fg ? fkey(i) : bkey(i)) << 32) | (unsigned int)i; }
        s_ki[a] = v;
      }
      __syncthreads();
      for (int k = 2; k <= N2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
          for (int t = tid; t < (N2 >> 1); t += nt) {
            const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), p = i | j;
            const unsigned long long x = s_ki[i], y = s_ki[p];
            if ((x > y) == ((i & k) == 0)) { s_ki[i] = y; s_ki[p] = x; }
          }
          __syncthreads();
        }
      for (int r = tid; r < min(nsel, cnt); r += nt) slot[slot0 + r] = (int)(unsigned int)s_ki[r];
    } else {
      for (int a = tid; a < cnt; a += nt) {
        const int i = list[a]; const uint32_t ki = is_fg ? fkey(i) : bkey(i); int rank = 0;
        for (int b = 0; b < cnt; ++b) { const int j = list[b]; const uint32_t kj = is_fg ? fkey(j) : bkey(j); rank += (kj < ki) || (kj == ki && j < i); }
        if (rank < nsel) slot[slot0 + rank] = i;
      }
    }
    __syncthreads();
  };
  if (!fg_repl) rank_list(fg_list, n_fg, true, nfg_sel, 0);
  else for (int s = tid; s < nfg_sel; s += nt) slot[s] = fg_list[bg_rand[s] % (uint32_t)n_fg];
  if (nbg_sel > 0) {
    if (!bg_repl) rank_list(bg_list, n_bg, false, nbg_sel, nfg_sel);
    else for (int s = tid; s < nbg_sel; s += nt) slot[nfg_sel + s] = bg_list[bg_rand[s] % (uint32_t)n_bg];
  }
  __syncthreads();
  if (tid == 0) { counts[0] = min(nfg_sel, mask_slots); counts[1] = n_fg; counts[2] = n_bg; counts[3] = appended; }
  // outputs
  const int W4 = 4 * ncls;          // bt / bi / bo arrive zeroed (l2s_proposal_target)
  for (int s = tid; s < R; s += nt) {
    const int i = slot[s];
    float b[4] = {0, 0, 0, 0}; int lab = 0;
    if (i >= 0) {
      roi_ptr(i, b);
      const float* g = gt + argm[i] * 5;
      lab = (s < nfg_sel) ? (int)g[4] : 0;                               // PTL:174
      if (lab > 0) {
        const float ew = b[2] - b[0] + 1.0f, eh = b[3] - b[1] + 1.0f, ecx = b[0] + 0.5f * ew, ecy = b[1] + 0.5f * eh;
        const float gw = g[2] - g[0] + 1.0f, gh = g[3] - g[1] + 1.0f, gcx = g[0] + 0.5f * gw, gcy = g[1] + 0.5f * gh;
        float t[4] = {(gcx - ecx) / ew, (gcy - ecy) / eh, logf(gw / ew), logf(gh / eh)};
        for (int k = 0; k < 4; ++k) {
          bt[(long)s * W4 + 4 * lab + k] = (t[k] - means4[k]) / stds4[k];
          bi[(long)s * W4 + 4 * lab + k] = inw4[k];
          bo[(long)s * W4 + 4 * lab + k] = inw4[k] > 0.f ? 1.f : 0.f;
        }
      }
    }
    out_rois[s * 5 + 0] = 0.f;
    for (int k = 0; k < 4; ++k) out_rois[s * 5 + 1 + k] = b[k];
    labels[s] = lab;
  }
  // mask targets (PTL:193-201): crop gt mask to the roi, PIL-NEAREST resize to ms x ms.  The per-RoI crop geometry goes through
  // LDS first, so that an element costs one global load (issued four at a time) instead of a chain of four dependent ones.
  const int nm = min(nfg_sel, mask_slots);
  int* s_geo = (int*)s_ki;                               // [mask_slots][6]: x1, y1, cw, ch, gt index, valid   (the sort buffer is free now)
  __syncthreads();
  for (int s = tid; s < mask_slots; s += nt) {
    int x1 = 0, y1 = 0, cw = 0, ch = 0, ga = 0, ok = 0;
    if (s < nm && slot[s] >= 0 && mask_slots * 6 <= 8192) {
      const int i = slot[s];
      float b[4]; roi_ptr(i, b);
      x1 = (int)b[0]; y1 = (int)b[1];
      int x2 = (int)b[2] + 1, y2 = (int)b[3] + 1;                       // python slice end, clipped by numpy
      x2 = min(x2, im_w); y2 = min(y2, im_h);
      cw = x2 - x1; ch = y2 - y1; ga = argm[i]; ok = cw > 0 && ch > 0;
    }
    int* g6 = s_geo + s * 6;
    g6[0] = x1; g6[1] = y1; g6[2] = cw; g6[3] = ch; g6[4] = ga; g6[5] = ok;
  }
  __syncthreads();
  const int ms2 = ms * ms, total = mask_slots * ms2;
  for (int e0 = tid; e0 < total; e0 += 4 * nt) {
    long addr[4]; bool val[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int e = e0 + q * nt;
      val[q] = false; addr[q] = 0;
      if (e < total) {
        const int s = e / ms2, r = e - s * ms2, py = r / ms, px = r - py * ms;
        const int* g6 = s_geo + s * 6;
        if (g6[5]) {
          // PIL nearest: xo = 0.5*s; idx_k = (int)xo; xo += s  (float64, sequential adds)
          const int cw = g6[2], ch = g6[3];
          const double sx = (double)cw / (double)ms, sy = (double)ch / (double)ms;
          double xo = 0.5 * sx, yo = 0.5 * sy;
          for (int k = 0; k < px; ++k) xo += sx;
          for (int k = 0; k < py; ++k) yo += sy;
          const int ix = min((int)xo, cw - 1), iy = min((int)yo, ch - 1);
          addr[q] = ((long)g6[4] * im_h + (g6[1] + iy)) * im_w + (g6[0] + ix);
          val[q] = true;
        }
      }
    }
    uint8_t m[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) m[q] = val[q] ? gt_masks[addr[q]] : (uint8_t)0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int e = e0 + q * nt; if (e < total) mt[e] = (float)m[q]; }
  }
}

// ------------------------------------------------------------------ RoIAlign (crop-and-resize)
__global__ void roialign_fwd_kernel(const void* feat, int H, int W, int C, const float* rois, int P, float sscale, float TH, float TW, void* out, int dt) {
  const int cell = blockIdx.x, r = cell / (P * P), rem = cell - r * P * P, py = rem / P, px = rem - py * P;
  const Samp s = roi_sample(rois + r * 5, H, W, P, py, px, sscale, TH, TW);
  const float w00 = (1.f - s.wx1) * (1.f - s.wy1), w01 = s.wx1 * (1.f - s.wy1), w10 = (1.f - s.wx1) * s.wy1, w11 = s.wx1 * s.wy1;
  const bool vx0 = s.x0 >= 0 && s.x0 < W, vx1 = s.x0 + 1 >= 0 && s.x0 + 1 < W, vy0 = s.y0 >= 0 && s.y0 < H, vy1 = s.y0 + 1 >= 0 && s.y0 + 1 < H;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    float v = 0.f;
    if (vy0 && vx0) v += w00 * ldx(feat, ((long)s.y0 * W + s.x0) * C + c, dt);
    if (vy0 && vx1) v += w01 * ldx(feat, ((long)s.y0 * W + s.x0 + 1) * C + c, dt);
    if (vy1 && vx0) v += w10 * ldx(feat, ((long)(s.y0 + 1) * W + s.x0) * C + c, dt);
    if (vy1 && vx1) v += w11 * ldx(feat, ((long)(s.y0 + 1) * W + s.x0 + 1) * C + c, dt);
    stx(out, (long)cell * C + c, dt, v);
  }
}
// bf16, C % 8 == 0: 8 channels (16 bytes) per thread; same arithmetic order as the scalar kernel (roi_sample.h: roi_blend8)
__global__ __launch_bounds__(128) void roialign_fwd_bf16x8_kernel(const bf16_t* __restrict__ feat, int H, int W, int C, const float* __restrict__ rois, int P,
                                                                 float sscale, float TH, float TW, bf16_t* out) {
  const int cell = blockIdx.x, r = cell / (P * P), rem = cell - r * P * P, py = rem / P, px = rem - py * P;
  const RoiTaps t = roi_taps(roi_sample(rois + r * 5, H, W, P, py, px, sscale, TH, TW), H, W, C);
  for (int c = threadIdx.x * 8; c < C; c += blockDim.x * 8) *(uint4*)(out + (long)cell * C + c) = roi_blend8(feat, t, c);
}
__global__ void roialign_bwd_kernel(const void* dout, int H, int W, int C, const float* rois, int P, float sscale, float TH, float TW, float* dfeat, int dt) {
  const int cell = blockIdx.x, r = cell / (P * P), rem = cell - r * P * P, py = rem / P, px = rem - py * P;
  const Samp s = roi_sample(rois + r * 5, H, W, P, py, px, sscale, TH, TW);
  const float w00 = (1.f - s.wx1) * (1.f - s.wy1), w01 = s.wx1 * (1.f - s.wy1), w10 = (1.f - s.wx1) * s.wy1, w11 = s.wx1 * s.wy1;
  const bool vx0 = s.x0 >= 0 && s.x0 < W, vx1 = s.x0 + 1 >= 0 && s.x0 + 1 < W, vy0 = s.y0 >= 0 && s.y0 < H, vy1 = s.y0 + 1 >= 0 && s.y0 + 1 < H;
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float g = ldx(dout, (long)cell * C + c, dt);
    if (vy0 && vx0) atomicAdd(dfeat + ((long)s.y0 * W + s.x0) * C + c, w00 * g);
    if (vy0 && vx1) atomicAdd(dfeat + ((long)s.y0 * W + s.x0 + 1) * C + c, w01 * g);
    if (vy1 && vx0) atomicAdd(dfeat + ((long)(s.y0 + 1) * W + s.x0) * C + c, w10 * g);
    if (vy1 && vx1) atomicAdd(dfeat + ((long)(s.y0 + 1) * W + s.x0 + 1) * C + c, w11 * g);
  }
}


// Gather form of the backward: one workgroup owns one feature pixel, lists the (RoI, bin) samples whose bilinear footprint
// touches it (the sample rows / columns of a RoI are separable, so a thread tests P rows + P columns of its RoI), and sums
// weight * dout over the list with vector loads — no atomics, one read-modify-write of the pixel's C floats, and a fixed
// summation order.  4 channels per thread-iteration (C % 4 == 0).
__device__ __forceinline__ void ra_load4(const float* p, float v[4]) {
  const float4 q = *(const float4*)p; v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
}
__device__ __forceinline__ void ra_load4(const bf16_t* p, float v[4]) {
  const uint2 q = *(const uint2*)p;
  v[0] = __uint_as_float(q.x << 16); v[1] = __uint_as_float(q.x & 0xffff0000u);
  v[2] = __uint_as_float(q.y << 16); v[3] = __uint_as_float(q.y & 0xffff0000u);
}
template <typename T>
__global__ __launch_bounds__(256) void roialign_bwd_gather_kernel(const T* dout, int H, int W, int C, const float* rois, int R, int P,
                                                                  float sscale, float TH, float TW, float* dfeat) {
  constexpr int CAP = 2048;
  __shared__ int e_cell[CAP];
  __shared__ float e_w[CAP];
  __shared__ int sc[256];
  const int pix = blockIdx.x, y = pix / W, x = pix - y * W, t = threadIdx.x, nv = C >> 2;
  bool wrote = false;                                // (uniform) the first chunk WRITES the pixel's channels: dfeat need not arrive zeroed
  for (int r0 = 0; r0 < R; r0 += 256) {
    const int r = r0 + t;
    unsigned my = 0, mx = 0;
    if (r < R) {
      for (int i = 0; i < P; ++i) {
        const Samp s = roi_sample(rois + r * 5, H, W, P, i, i, sscale, TH, TW);
        const float wyv = (s.y0 == y ? 1.f - s.wy1 : 0.f) + (s.y0 + 1 == y ? s.wy1 : 0.f);
        const float wxv = (s.x0 == x ? 1.f - s.wx1 : 0.f) + (s.x0 + 1 == x ? s.wx1 : 0.f);
        if (wyv != 0.f) my |= 1u << i;
        if (wxv != 0.f) mx |= 1u << i;
      }
    }
    const int n = __popc(my) * __popc(mx);
    // inclusive scan of the 256 counts
    sc[t] = n;
    __syncthreads();
    for (int o = 1; o < 256; o <<= 1) {
      const int v = t >= o ? sc[t - o] : 0;
      __syncthreads();
      sc[t] += v;
      __syncthreads();
    }
    const int total = sc[255], off = sc[t] - n;
    __syncthreads();
    for (int wb = 0; wb < total; wb += CAP) {
      if (n && off < wb + CAP && off + n > wb) {
        int k = off;
        for (unsigned a = my; a; a &= a - 1) {
          const int py = __ffs(a) - 1;
          for (unsigned b = mx; b; b &= b - 1, ++k) {
            if (k < wb || k >= wb + CAP) continue;
            const int px = __ffs(b) - 1;
            const Samp s = roi_sample(rois + r * 5, H, W, P, py, px, sscale, TH, TW);
            const float wyv = s.y0 == y ? 1.f - s.wy1 : s.wy1, wxv = s.x0 == x ? 1.f - s.wx1 : s.wx1;
            e_cell[k - wb] = (r * P + py) * P + px;
            e_w[k - wb] = wxv * wyv;
          }
        }
      }
      __syncthreads();
      const int ne = min(CAP, total - wb);
      for (int v = t; v < nv; v += 256) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        const T* base = dout + v * 4;
#pragma unroll 4
        for (int e = 0; e < ne; ++e) {
          float g[4];
          ra_load4(base + (long)e_cell[e] * C, g);
          const float w = e_w[e];
          a0 += w * g[0]; a1 += w * g[1]; a2 += w * g[2]; a3 += w * g[3];
        }
        float4* d = (float4*)(dfeat + (long)pix * C + v * 4);
        float4 q = wrote ? *d : make_float4(0.f, 0.f, 0.f, 0.f); q.x += a0; q.y += a1; q.z += a2; q.w += a3; *d = q;
      }
      wrote = true;
      __syncthreads();
    }
  }
  if (!wrote)                                        // no sample of any RoI touches this pixel
    for (int v = t; v < nv; v += 256) *(float4*)(dfeat + (long)pix * C + v * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
}

}  // namespace

extern "C" int l2s_rpn_decode(const float* heads, int ldh, const float* base_anchors, int H, int W, int A, int feat_stride,
                              float im_h, float im_w, float* prob, float* boxes, float* scores, hipStream_t s) {
  const int n = H * W * A;
  L2S_LAUNCH(rpn_decode_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, heads, ldh, base_anchors, H, W, A, feat_stride, im_h, im_w, prob, boxes, scores);
  return l2s_check_launch();
}
extern "C" long l2s_sort_ws_ints(int n) { return 4L * n + 4L * 256L * cdiv(n, 256) + 64; }
extern "C" int l2s_sort_topk(const float* scores, const float* boxes, int n, int k, int* ws, float* sorted_boxes,
                             float* sorted_scores, int* sorted_idx, hipStream_t s) {
  if (!ws || n <= 0 || k > n) return L2S_EINVAL;
  const int nblk = cdiv(n, 256);
  RsWs w = rs_ws(ws, n);
  L2S_LAUNCH(rs_hist0_kernel, dim3(nblk), dim3(256), 0, s, scores, n, nblk, w.hist);
  L2S_LAUNCH(rs_pass_kernel<0>, dim3(nblk), dim3(256), 0, s, scores, boxes, (const unsigned int*)nullptr, (const int*)nullptr, w.kA, w.iA, n, nblk, w.hist, k, sorted_boxes, sorted_scores, sorted_idx);
  L2S_LAUNCH(rs_pass_kernel<1>, dim3(nblk), dim3(256), 0, s, scores, boxes, (const unsigned int*)w.kA, (const int*)w.iA, w.kB, w.iB, n, nblk, w.hist, k, sorted_boxes, sorted_scores, sorted_idx);
  L2S_LAUNCH(rs_pass_kernel<2>, dim3(nblk), dim3(256), 0, s, scores, boxes, (const unsigned int*)w.kB, (const int*)w.iB, w.kA, w.iA, n, nblk, w.hist, k, sorted_boxes, sorted_scores, sorted_idx);
  L2S_LAUNCH(rs_pass_kernel<3>, dim3(nblk), dim3(256), 0, s, scores, boxes, (const unsigned int*)w.kA, (const int*)w.iA, w.kB, w.iB, n, nblk, w.hist, k, sorted_boxes, sorted_scores, sorted_idx);
  return l2s_check_launch();
}
static inline size_t nms_mask_words(int n) { const int sbw = min(cdiv(n, 64), NMS_SB); return (size_t)sbw * sbw * 64; }
static inline size_t nms_scan_lds(int sbw) { return ((size_t)NMS_RING * NMS_SLOT * 64 + 2 * (size_t)sbw) * 8; }
extern "C" size_t l2s_nms_workspace_bytes(int n) { return (nms_mask_words(n) + (size_t)cdiv(cdiv(n, 64), NMS_SB) * NMS_SB) * 8; }
extern "C" int l2s_nms(const float* sorted_boxes, int n, float thresh, int cmp_mode, int max_keep, uint64_t* mask_ws,
                       int* keep_out, int* num_out, hipStream_t s) {
  if (n <= 0 || max_keep < 1) return L2S_EINVAL;
  const int cb = cdiv(n, 64), nst = cdiv(cb, NMS_SB);
  static bool attr_done = false;
  if (!attr_done) { (void)hipFuncSetAttribute((const void*)nms_reduce_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)nms_scan_lds(NMS_SB)); attr_done = true; }
  unsigned long long* carry_all = (unsigned long long*)mask_ws + nms_mask_words(n);
  for (int st = 0; st < nst; ++st) {
    const int row0 = st * NMS_SB * 64, rows = min(n - row0, NMS_SB * 64), sbw = cdiv(rows, 64);
    unsigned long long* carry = carry_all + (size_t)st * NMS_SB;
    L2S_LAUNCH(nms_mask_kernel, dim3(cdiv(sbw, 4), sbw + (st ? cdiv(max_keep, 64) : 0)), dim3(256), 0, s, sorted_boxes, n, thresh, cmp_mode, row0, sbw, mask_ws,
               carry, (const int*)keep_out, (const int*)num_out, max_keep, carry_all, nst * NMS_SB);
    L2S_LAUNCH(nms_reduce_kernel, dim3(1), dim3(1024), nms_scan_lds(sbw), s, (const uint64_t*)mask_ws, rows, sbw, max_keep, keep_out, num_out, row0,
               st ? (const unsigned long long*)carry : (const unsigned long long*)nullptr);
  }
  return l2s_check_launch();
}
#ifdef L2S_TOOLS
extern "C" int l2s_tools_nms_dbg(long long* host16) { return hipMemcpyFromSymbol(host16, HIP_SYMBOL(nms_dbg), sizeof(long long) * 16) == hipSuccess ? 0 : L2S_ELAUNCH; }
#endif
extern "C" int l2s_gather_rois(const float* sorted_boxes, const float* sorted_scores, const int* keep, const int* num, int max_keep,
                               float* rois, float* roi_scores, hipStream_t s) {
  L2S_LAUNCH(gather_rois_kernel, dim3(cdiv(max_keep, 256)), dim3(256), 0, s, sorted_boxes, sorted_scores, keep, num, max_keep, rois, roi_scores);
  return l2s_check_launch();
}
extern "C" const int* l2s_anchor_target_count(const int* ws) { return ws + 2; }
extern "C" long l2s_anchor_target_ws_ints(int hwa) { return 16 + 64 + 4L * hwa + 16; }
extern "C" int l2s_anchor_target(const float* gt, int n_gt, const float* base_anchors, int H, int W, int A, int feat_stride,
                                 float im_h, float im_w, const uint32_t* fg_keys, const uint32_t* bg_keys,
                                 float neg_ov, float pos_ov, int batch, float fg_frac,
                                 int* labels, float* targets, float* inside_w, float* outside_w, int* ws, hipStream_t s) {
  if (n_gt < 1 || n_gt > 32) return L2S_EINVAL;
  const int n = H * W * A;
  L2S_LAUNCH(atl_init_kernel, dim3(1), dim3(128), 0, s, ws);
  L2S_LAUNCH(atl_iou_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, gt, n_gt, base_anchors, n, W, A, feat_stride, im_h, im_w, ws);
  L2S_LAUNCH(atl_label_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, gt, n_gt, base_anchors, n, W, A, feat_stride, neg_ov, pos_ov, ws);
  if (n <= ATL_LDS_MAX) {
    static bool attr_done = false;
    if (!attr_done) { (void)hipFuncSetAttribute((const void*)atl_sample_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (256 + 24) * 4 + ATL_LDS_MAX * 5); attr_done = true; }
    const int npad = (n + 3) / 4 * 4;
    L2S_LAUNCH(atl_sample_lds_kernel, dim3(1), dim3(1024), (size_t)(256 + 24) * 4 + (size_t)npad * 5, s, fg_keys, bg_keys, n, npad, batch, fg_frac, ws);
  } else {
    L2S_LAUNCH(atl_sample_kernel, dim3(1), dim3(1024), 0, s, fg_keys, bg_keys, n, batch, fg_frac, ws);
  }
  L2S_LAUNCH(atl_out_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, gt, base_anchors, n, H, W, A, feat_stride, (const int*)ws, labels, targets, inside_w, outside_w);
  return l2s_check_launch();
}
extern "C" int l2s_proposal_target(const float* rois, const float* roi_scores, const int* n_rois, int n_max, const float* gt, int n_gt,
                                   const uint8_t* gt_masks, int im_h, int im_w, const uint32_t* fg_keys, const uint32_t* bg_keys,
                                   const uint32_t* bg_rand, int R, int fg_max, int mask_slots, float fg_thresh, float bg_hi, float bg_lo,
                                   const float* means4, const float* stds4, const float* inw4, int ncls, int ms,
                                   float* out_rois, int* labels, float* bbox_targets, float* bbox_inside, float* bbox_outside,
                                   float* mask_targets, int* counts, int* ws, hipStream_t s) {
  if (n_gt < 1 || R < 1 || mask_slots < 1 || mask_slots > R || mask_slots * 6 > 8192) return L2S_EINVAL;
  // the three [R][4 ncls] target arrays are sparse (4 floats per foreground row): cleared by the copy engine / fill kernel
  // instead of 730 stores per thread of the single-workgroup kernel
  const size_t tb = (size_t)R * 4 * ncls * sizeof(float);
  if ((char*)bbox_inside == (char*)bbox_targets + tb && (char*)bbox_outside == (char*)bbox_inside + tb) {      // one allocation: one clear
    if (l2s_memset_async(bbox_targets, 0, 3 * tb, s)) return L2S_ELAUNCH;
  } else if (l2s_memset_async(bbox_targets, 0, tb, s) || l2s_memset_async(bbox_inside, 0, tb, s) || l2s_memset_async(bbox_outside, 0, tb, s)) return L2S_ELAUNCH;
  L2S_LAUNCH(ptl_kernel, dim3(1), dim3(1024), 0, s, rois, roi_scores, n_rois, n_max, gt, n_gt, gt_masks, im_h, im_w,
                     fg_keys, bg_keys, bg_rand, R, fg_max, mask_slots, fg_thresh, bg_hi, bg_lo, means4, stds4, inw4, ncls, ms,
                     out_rois, labels, bbox_targets, bbox_inside, bbox_outside, mask_targets, counts, ws);
  return l2s_check_launch();
}
static int roialign_fwd_launch(const void* feat, int H, int W, int C, const float* rois, int R, int P, float sscale, float TH, float TW,
                               void* out, int dtype, hipStream_t s) {
  if (dtype == L2S_BF16 && !(C & 7) && !((uintptr_t)feat & 15) && !((uintptr_t)out & 15)) {
    L2S_LAUNCH(roialign_fwd_bf16x8_kernel, dim3(R * P * P), dim3(C >= 1024 ? 128 : 64), 0, s, (const bf16_t*)feat, H, W, C, rois, P, sscale, TH, TW, (bf16_t*)out);
    return l2s_check_launch();
  }
  L2S_LAUNCH(roialign_fwd_kernel, dim3(R * P * P), dim3(256), 0, s, feat, H, W, C, rois, P, sscale, TH, TW, out, dtype);
  return l2s_check_launch();
}
static int roialign_bwd_launch(const void* dout, int H, int W, int C, const float* rois, int R, int P, float sscale, float TH, float TW,
                               float* dfeat, int dtype, hipStream_t s) {
  if (C % 4 == 0 && P <= 32) {      // gather form: no atomics, fixed summation order (the scatter kernel below only serves odd channel counts)
    if (dtype) L2S_LAUNCH(roialign_bwd_gather_kernel<bf16_t>, dim3(H * W), dim3(256), 0, s, (const bf16_t*)dout, H, W, C, rois, R, P, sscale, TH, TW, dfeat);
    else L2S_LAUNCH(roialign_bwd_gather_kernel<float>, dim3(H * W), dim3(256), 0, s, (const float*)dout, H, W, C, rois, R, P, sscale, TH, TW, dfeat);
    return l2s_check_launch();
  }
  if (l2s_memset_async(dfeat, 0, (size_t)H * W * C * sizeof(float), s)) return L2S_ELAUNCH;     // the scatter form adds
  L2S_LAUNCH(roialign_bwd_kernel, dim3(R * P * P), dim3(256), 0, s, dout, H, W, C, rois, P, sscale, TH, TW, dfeat, dtype);
  return l2s_check_launch();
}
extern "C" int l2s_roialign_fwd(const void* feat, int H, int W, int C, const float* rois, int R, int P, float spatial_scale,
                                void* out, int dtype, hipStream_t s) {
  return roialign_fwd_launch(feat, H, W, C, rois, R, P, spatial_scale, (float)H, (float)W, out, dtype, s);
}
extern "C" int l2s_roialign_bwd(const void* dout, int H, int W, int C, const float* rois, int R, int P, float spatial_scale,
                                float* dfeat, int dtype, hipStream_t s) {
  return roialign_bwd_launch(dout, H, W, C, rois, R, P, spatial_scale, (float)H, (float)W, dfeat, dtype, s);
}
extern "C" int l2s_cropalign_fwd(const void* feat, int H, int W, int C, const float* rois, int R, int P, float im_h, float im_w,
                                 void* out, int dtype, hipStream_t s) {
  return roialign_fwd_launch(feat, H, W, C, rois, R, P, 1.f, im_h, im_w, out, dtype, s);
}
extern "C" int l2s_cropalign_bwd(const void* dout, int H, int W, int C, const float* rois, int R, int P, float im_h, float im_w,
                                 float* dfeat, int dtype, hipStream_t s) {
  return roialign_bwd_launch(dout, H, W, C, rois, R, P, 1.f, im_h, im_w, dfeat, dtype, s);
}
extern "C" int l2s_roipool_fwd(const void* feat, int H, int W, int C, const float* rois, int R, int P, float spatial_scale, void* out,
                               int* argmax, int dtype, hipStream_t s) {
  L2S_LAUNCH(roipool_fwd_kernel, dim3(R * P * P), dim3(256), 0, s, feat, H, W, C, rois, P, spatial_scale, out, argmax, dtype);
  return l2s_check_launch();
}
extern "C" int l2s_roipool_bwd(const void* dout, const int* argmax, int R, int P, int C, float* dfeat, int dtype, hipStream_t s) {
  const long n = (long)R * P * P * C;
  long g = (n + 255) / 256; if (g > 8192) g = 8192;
  L2S_LAUNCH(roipool_bwd_kernel, dim3((int)g), dim3(256), 0, s, dout, argmax, n, C, dfeat, dtype);
  return l2s_check_launch();
}
