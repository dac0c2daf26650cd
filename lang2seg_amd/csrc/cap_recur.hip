// The att2in2 captioner recurrence (lib/caption_models/AttModel.py:406-423 attention, 446-466 core) as ONE resident launch per direction.
//
// Until round 5 a token cost three dependent launches each way (h GEMVs -> attention dots -> softmax . P + gates; lang.hip), 16.6 us of
// kernel time and 32 us of wall per token inside the step: launch boundaries and queue waits on the critical chain of the caption
// branch.  Here 32 workgroups of 512 threads stay resident for the whole sentence, partition the work BY HIDDEN UNIT and exchange three
// small vectors per token with the all-gather primitive tools/cap_allgather_probe.hip priced (1.4 us on an idle chip, 2.1-2.4 us beside
// a chip-filling streaming kernel as long as the workgroups own their CUs, which the 152 KiB LDS request ensures; profiles/r05_cap_allgather_probe.txt):
//   * a value crosses workgroups as ONE naturally aligned 8-byte {tag, value} granule written by one `global_store_dwordx2 sc1` and
//     polled with `global_load_dwordx2 sc1` (L2-served, never L1): the data IS the flag - no counter, no fence, no release / acquire;
//   * tag = (launch count + 1) << 8 | (phase + 1): the granule buffers are reused by every token and every launch without clearing.
//     Reuse is safe because the three exchanges of a token form a cycle: a producer can only rewrite buffer X after it has passed an
//     exchange Y that every workgroup entered after consuming X.  The launch count lives in device memory (word 0 of the state
//     buffer; workgroup 0 increments it when it leaves), so a replayed launch tape needs no per-launch argument;
//   * every spin is bounded: a workgroup that gives up sets word 1 of the state buffer and leaves; the others follow.
// Weights never leave the register file: a workgroup owns 16 hidden units = 80 rows of W_h2h + 16 rows of W_h2att (96 x 512 fp32 =
// 192 KiB = 96 VGPRs per thread), so the arithmetic stays fp32 in both compute modes (the captioner is fp32 in the reference, and
// its linears were fp32 here too); the projected-attention columns P[:, units] (25 KiB) and the workgroup's 7 rows of patt live in LDS.
//
// forward, per token t (workgroup w, units j in [16w, 16w + 16), locations l in [7w, 7w + 7)):
//   A  att_h[16w..] = W_h2att[rows] h + b  -> exchange 1 (512 values);   s[g][j] = W_h2h[g R + j] h + b   (kept in LDS)
//   B  dots[l] = alpha . tanh(patt[l] + att_h) + b  for the own l (tanh kept for backward)  -> exchange 2 (L values)
//   C  softmax over L (every workgroup), a2c[j] = sum_l w[l] P[l][j] + b, gates, c, h[16w..]  -> exchange 3 (512 values)
// backward, per token t descending (same ownership; W as COLUMNS per thread: thread k holds W[own rows][k]):
//   1  gate backward of the own units -> d(sums) rows (kept), d(a2c)  -> exchange 1 (1024 values)
//   2  d(weight)[l] = P[l] . d(a2c) for the own l  -> exchange 2 (L values)
//   3  softmax backward (every workgroup) -> ddot[L];  d(att_h)[16w..] = alpha sum_l ddot[l] (1 - tanh^2)
//   4  partial d(h)(t-1)[k] = sum over the own 96 rows of {d(sums), d(att_h)}[row] W[row][k], all 512 k  -> exchange 3: a reduce-scatter,
//      32 x 512 granules, each workgroup sums the 32 partial vectors of its 16 units in producer order (bit-reproducible).
#include "common.h"
#include "../../include/lang2seg_hip.h"

namespace {

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
#define RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

constexpr int NWG = L2S_CAP_RECUR_WGS, NT = 512, R = 512, AH = 512, UPW = 16, LPW = 7, LMAX = NWG * LPW;
constexpr unsigned SPIN_LIMIT = 1u << 20;
constexpr size_t LDS_REQ = 152 * 1024;     // owns the CU's LDS: no LDS-using workgroup of another queue co-resides (the price of an exchange doubles to triples otherwise)

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ void put(gu64* g, unsigned tag, float v) { __hip_atomic_store(g, ((u64)tag << 32) | __float_as_uint(v), RLX_AGENT); }

// every thread of the workgroup polls granules tid, tid + NT, ... (NR per thread) of g[0, n) until their tags match, then dst[i] = value.
// Returns false when the workgroup has to give up (bounded spin; *dead is an LDS word shared by the workgroup).
// A give-up also ends the launch's epoch (workgroup 0 bumps state[0] on that path too): the granules the failed launch left behind carry
// the old epoch's tags and cannot satisfy the next launch; state[1] stays set until the host has seen it (Network.train_step reads and
// clears it when a loss comes back non-finite and raises L2SError).
template <int NR>
__device__ __forceinline__ bool gather(gu64* g, int n, unsigned tag, float* dst, int* dead, unsigned* err) {
  const int tid = threadIdx.x;
  bool ok[NR]; float v[NR];
#pragma unroll
  for (int r = 0; r < NR; ++r) { ok[r] = tid + r * NT >= n; v[r] = 0.f; }
  for (unsigned spins = 0;; ++spins) {
    bool all = true;
#pragma unroll
    for (int r = 0; r < NR; ++r) {
      if (!ok[r]) {
        const u64 x = __hip_atomic_load(g + tid + r * NT, RLX_AGENT);
        if ((unsigned)(x >> 32) == tag) { ok[r] = true; v[r] = __uint_as_float((unsigned)x); }
      }
      all = all && ok[r];
    }
    if (__all(all)) break;
    if (spins > SPIN_LIMIT) { if ((tid & 63) == 0) { *dead = 1; atomicOr(err, 1u); } break; }
  }
#pragma unroll
  for (int r = 0; r < NR; ++r) if (tid + r * NT < n) dst[tid + r * NT] = v[r];
  __syncthreads();
  return *dead == 0;
}

__device__ __forceinline__ float dot8(const float4& wa, const float4& wb, const float4& xa, const float4& xb) {
  float a = wa.x * xa.x;
  a = fmaf(wa.y, xa.y, a); a = fmaf(wa.z, xa.z, a); a = fmaf(wa.w, xa.w, a);
  a = fmaf(wb.x, xb.x, a); a = fmaf(wb.y, xb.y, a); a = fmaf(wb.z, xb.z, a); a = fmaf(wb.w, xb.w, a);
  return a;
}

constexpr int F_TOK = 512 + 256 + 512;               // granules of the forward exchanges: att_h | dots (padded) | h
constexpr int B_TOK = 1024 + 256 + NWG * 512;        // backward: d(a2c) | d(weight) | partial d(h), [consumer][producer][16]

__global__ __launch_bounds__(NT) void cap_recur_fwd_kernel(l2s_cap_recur_fwd_args a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* h_l = sm;                 // [512]  h(t-1)
  float* ah_l = sm + 512;          // [512]  att_h
  float* e_l = sm + 1024;          // [256]  dots
  float* w_l = sm + 1280;          // [256]  softmax weights
  float* s_l = sm + 1536;          // [96]   W_h2h rows of the own units times h, + bias
  float* part = sm + 1632;         // [16][32]
  float* aw_l = sm + 2144;         // [512]
  float* patt_l = sm + 2656;       // [7][512]
  float* pc_l = sm + 6240;         // [LMAX][32]: P[l][j] (c < 16) and P[l][R + j] (c >= 16) of the own units
  int* dead = (int*)(sm + 6240 + LMAX * 32);
  __builtin_amdgcn_s_setprio(3);
  const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int S = a.S, L = a.L;
  unsigned* state = a.state;
  gu64* gran = (gu64*)(a.state + 4);
  const unsigned epoch = (state[0] + 1u) << 8;
  if (tid == 0) *dead = 0;

  // ---- resident operands ----
  // rows of a wave: r = 0, 1 -> W_h2att rows 16w + 2wv + r (published first: they head the token's first exchange); r = 2..11 -> W_h2h row
  // g R + 16w + jj with q = 10 wv + r - 2, g = q / 16, jj = q % 16
  float4 wa[12], wb[12];
  float brow = 0.f;
#pragma unroll
  for (int r = 0; r < 12; ++r) {
    const float* row;
    float b;
    if (r < 2) { const int m = UPW * w + 2 * wv + r; row = a.w_h2att + (long)m * R; b = a.b_h2att[m]; }
    else { const int q = 10 * wv + r - 2, n = (q >> 4) * R + UPW * w + (q & 15); row = a.w_h2h + (long)n * R; b = a.b_h2h[n]; }
    wa[r] = *(const float4*)(row + lane * 4);
    wb[r] = *(const float4*)(row + 256 + lane * 4);
    if (lane == r) brow = b;
  }
  for (int i = tid; i < AH; i += NT) aw_l[i] = a.aw[i];
  for (int i = tid; i < LPW * AH; i += NT) {
    const int l = LPW * w + i / AH;
    patt_l[i] = l < L ? a.patt[(long)l * AH + (i % AH)] : 0.f;
  }
  for (int i = tid; i < L * 32; i += NT) {
    const int l = i >> 5, c = i & 31;
    pc_l[i] = a.P[(long)l * 2 * R + (c < 16 ? UPW * w + c : R + UPW * w + c - 16)];
  }
  h_l[tid] = a.hs[tid];                                   // h(-1): row 0 of the state array
  const int j = UPW * w + (tid & 15);                     // the unit of a gate thread (tid < 16)
  float c_prev = 0.f, ba0 = 0.f, ba1 = 0.f, sg[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  if (tid < UPW) {
    c_prev = a.cs[j]; ba0 = a.b_a2c[j]; ba1 = a.b_a2c[R + j];
#pragma unroll
    for (int g = 0; g < 5; ++g) sg[g] = a.sums[(long)g * R + j];
  }
  const float ab = a.ab[0];
  __syncthreads();

  for (int t = 0; t < S; ++t) {
    gu64* G1 = gran; gu64* G2 = gran + 512; gu64* G3 = gran + 768;
    const unsigned tag = epoch + 1u + (unsigned)(t % 80) * 3u;        // < 256: (t % 80) * 3 + 3 <= 240
    // ---- A: the two GEMVs from the register-resident rows ----
    {
      const float4 ha = *(const float4*)(h_l + lane * 4), hb = *(const float4*)(h_l + 256 + lane * 4);
      float mine = 0.f;
#pragma unroll
      for (int r = 0; r < 2; ++r) { const float s = wave_sum(dot8(wa[r], wb[r], ha, hb)); if (lane == r) mine = s; }
      if (lane < 2) put(G1 + UPW * w + 2 * wv + lane, tag, mine + brow);
#pragma unroll
      for (int r = 2; r < 12; ++r) { const float s = wave_sum(dot8(wa[r], wb[r], ha, hb)); if (lane == r) mine = s; }
      if (lane >= 2 && lane < 12) s_l[10 * wv + lane - 2] = mine + brow;
    }
    if (!gather<1>(G1, AH, tag, ah_l, dead, state + 1)) { a.hs[R + tid] = NAN; if (w == 0 && tid == 0) state[0] = state[0] + 1u; return; }      // a give-up poisons h(0): the caption loss turns NaN, nothing passes silently
    // ---- B: attention dots of the own locations (one wave each) ----
    {
      const int l = LPW * w + wv;
      if (wv < LPW && l < L) {
        const float4 pa = *(const float4*)(patt_l + wv * AH + lane * 4), pb = *(const float4*)(patt_l + wv * AH + 256 + lane * 4);
        const float4 xa = *(const float4*)(ah_l + lane * 4), xb = *(const float4*)(ah_l + 256 + lane * 4);
        const float4 ya = *(const float4*)(aw_l + lane * 4), yb = *(const float4*)(aw_l + 256 + lane * 4);
        float4 ta, tb;
        ta.x = tanhf(pa.x + xa.x); ta.y = tanhf(pa.y + xa.y); ta.z = tanhf(pa.z + xa.z); ta.w = tanhf(pa.w + xa.w);
        tb.x = tanhf(pb.x + xb.x); tb.y = tanhf(pb.y + xb.y); tb.z = tanhf(pb.z + xb.z); tb.w = tanhf(pb.w + xb.w);
        float* tw = a.tanh_ws + ((long)t * L + l) * AH;
        *(float4*)(tw + lane * 4) = ta; *(float4*)(tw + 256 + lane * 4) = tb;
        const float s = wave_sum(dot8(ta, tb, ya, yb));
        if (lane == 0) put(G2 + l, tag + 1, s + ab);
      }
    }
    if (!gather<1>(G2, L, tag + 1, e_l, dead, state + 1)) { a.hs[R + tid] = NAN; if (w == 0 && tid == 0) state[0] = state[0] + 1u; return; }      // a give-up poisons h(0): the caption loss turns NaN, nothing passes silently
    // ---- C: softmax over the L locations (every wave computes the statistics; waves 0-3 store a quarter of the weights each) ----
    {
      float v[4], mx = -INFINITY;
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[k] = lane + 64 * k < L ? e_l[lane + 64 * k] : -INFINITY; mx = fmaxf(mx, v[k]); }
      mx = wave_max(mx);
      float e[4], sum = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) { e[k] = lane + 64 * k < L ? expf(v[k] - mx) : 0.f; sum += e[k]; }
      sum = wave_sum(sum);
      if (wv < 4) {
        const float ev = wv == 0 ? e[0] : wv == 1 ? e[1] : wv == 2 ? e[2] : e[3];
        const float wgt = ev / sum;
        w_l[tid] = wgt;                                   // tid = lane + 64 wv; 0 beyond L
        if (w == 0 && tid < L) a.wgt[(long)t * L + tid] = wgt;
      }
    }
    __syncthreads();
    {
      const int c = tid & 31, lg = tid >> 5;
      float acc = 0.f;
      for (int l = lg; l < L; l += 16) acc = fmaf(w_l[l], pc_l[l * 32 + c], acc);
      part[lg * 32 + c] = acc;
    }
    __syncthreads();
    if (tid < UPW) {
      float a0 = ba0, a1 = ba1;
#pragma unroll
      for (int g = 0; g < 16; ++g) { a0 += part[g * 32 + tid]; a1 += part[g * 32 + 16 + tid]; }
      const float ig = sigm(sg[0] + s_l[tid]), fg = sigm(sg[1] + s_l[16 + tid]), og = sigm(sg[2] + s_l[32 + tid]);
      const float t0 = sg[3] + s_l[48 + tid] + a0, t1 = sg[4] + s_l[64 + tid] + a1;
      const float it = fmaxf(t0, t1);
      const float cn = fg * c_prev + ig * it;
      const float tc = tanhf(cn);
      const float hn = og * tc;
      if (t + 1 < S) put(G3 + j, tag + 2, hn);
      a.cs[(long)(t + 1) * R + j] = cn; a.hs[(long)(t + 1) * R + j] = hn;
      float* sv = a.save + (long)t * 6 * R;
      sv[j] = ig; sv[R + j] = fg; sv[2 * R + j] = og; sv[3 * R + j] = (t0 >= t1) ? 0.f : 1.f;     // torch.max(a, b): first wins ties
      sv[4 * R + j] = it; sv[5 * R + j] = tc;
      c_prev = cn;
      if (t + 1 < S) {
#pragma unroll
        for (int g = 0; g < 5; ++g) sg[g] = a.sums[(long)(t + 1) * 5 * R + (long)g * R + j];
      }
    }
    if (t + 1 < S && !gather<1>(G3, R, tag + 2, h_l, dead, state + 1)) { a.hs[R + tid] = NAN; if (w == 0 && tid == 0) state[0] = state[0] + 1u; return; }      // a give-up poisons h(0): the caption loss turns NaN, nothing passes silently
  }
  if (w == 0 && tid == 0) state[0] = state[0] + 1u;
}

__global__ __launch_bounds__(NT) void cap_recur_bwd_kernel(l2s_cap_recur_bwd_args a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* g_l = sm;                 // [1024] d(a2c) of every unit
  float* dw_l = sm + 1024;         // [256]  d(weight)
  float* dd_l = sm + 1280;         // [256]  ddot
  float* v_l = sm + 1536;          // [96]   d(sums) rows of the own units (g * 16 + jj), then d(att_h) of the own 16 channels
  float* part = sm + 1632;         // [32][16]
  float* rs_l = sm + 2144;         // [512]  the 32 partial d(h) vectors of the own units
  float* prow_l = sm + 2656;       // [7][1024] rows of P of the own locations
  int* dead = (int*)(sm + 2656 + LPW * 1024);
  __builtin_amdgcn_s_setprio(3);
  const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int S = a.S, L = a.L;
  unsigned* state = a.state;
  gu64* gran = (gu64*)(a.state + 4);
  const unsigned epoch = (state[0] + 1u) << 8;
  if (tid == 0) *dead = 0;

  // ---- resident operands: thread k holds column k of the own 96 rows (n < 80: W_h2h row (n / 16) R + 16w + n % 16; else W_h2att row 16w + n - 80) ----
  float wr[96];
#pragma unroll
  for (int n = 0; n < 96; ++n) {
    const float* row = n < 80 ? a.w_h2h + (long)((n >> 4) * R + UPW * w + (n & 15)) * R : a.w_h2att + (long)(UPW * w + n - 80) * R;
    wr[n] = row[tid];
  }
  for (int i = tid; i < LPW * 1024; i += NT) {
    const int l = LPW * w + (i >> 10);
    prow_l[i] = l < L ? a.P[(long)l * 2 * R + (i & 1023)] : 0.f;
  }
  const int jj = tid & 15, j = UPW * w + jj;                // gate thread's unit (tid < 16); channel of the d(att_h) sums
  const int lg = tid >> 4;                                  // location group of the d(att_h) sums: l = lg + 32 q
  const float aw_own = a.aw[UPW * w + jj];
  float dh = 0.f, dc = 0.f;                                 // recurrent gradients of the own unit (gate threads)
  __syncthreads();

  for (int t = S - 1; t >= 0; --t) {
    gu64* G1 = gran; gu64* G2 = gran + 1024; gu64* G3 = gran + 1280;
    const unsigned tag = epoch + 1u + (unsigned)(t % 80) * 3u;
    // operands of this token that do not depend on the chain: requested first
    float wq[4], tq[7];
#pragma unroll
    for (int k = 0; k < 4; ++k) wq[k] = lane + 64 * k < L ? a.wgt[(long)t * L + lane + 64 * k] : 0.f;
#pragma unroll
    for (int q = 0; q < 7; ++q) { const int l = lg + 32 * q; tq[q] = l < L ? a.tanh_ws[((long)t * L + l) * AH + UPW * w + jj] : 1.f; }
    // ---- 1: gate backward of the own units (cap_gates_bwd_kernel's arithmetic) ----
    if (tid < UPW) {
      const float* sv = a.save + (long)t * 6 * R;
      const float ig = sv[j], fg = sv[R + j], og = sv[2 * R + j], sel = sv[3 * R + j], it = sv[4 * R + j], tc = sv[5 * R + j];
      const float dhj = dh + a.dho[(long)t * R + j];       // recurrent part + this step's output gradient
      const float dcn = dc + dhj * og * (1.f - tc * tc);
      const float d0s = dcn * it * ig * (1.f - ig), d1s = dcn * a.cs[(long)t * R + j] * fg * (1.f - fg), d2s = dhj * tc * og * (1.f - og);
      const float dit = dcn * ig;
      const float d0 = sel == 0.f ? dit : 0.f, d1 = sel == 0.f ? 0.f : dit;
      put(G1 + j, tag, d0); put(G1 + R + j, tag, d1);
      float* ds = a.dsums + (long)t * 5 * R;
      ds[j] = d0s; ds[R + j] = d1s; ds[2 * R + j] = d2s; ds[3 * R + j] = d0; ds[4 * R + j] = d1;
      float* da = a.da2c + (long)t * 2 * R;
      da[j] = d0; da[R + j] = d1;
      v_l[jj] = d0s; v_l[16 + jj] = d1s; v_l[32 + jj] = d2s; v_l[48 + jj] = d0; v_l[64 + jj] = d1;
      dc = dcn * fg;
    }
    if (!gather<2>(G1, 2 * R, tag, g_l, dead, state + 1)) { a.dsums[tid] = NAN; if (w == 0 && tid == 0) state[0] = state[0] + 1u; return; }      // a give-up poisons d(sums): the captioner's gradients turn NaN
    // ---- 2: d(weight) of the own locations: P[l] . d(a2c), one wave each ----
    {
      const int l = LPW * w + wv;
      if (wv < LPW && l < L) {
        float acc = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 pv = *(const float4*)(prow_l + wv * 1024 + q * 256 + lane * 4), gv = *(const float4*)(g_l + q * 256 + lane * 4);
          acc = fmaf(pv.x, gv.x, fmaf(pv.y, gv.y, fmaf(pv.z, gv.z, fmaf(pv.w, gv.w, acc))));
        }
        acc = wave_sum(acc);
        if (lane == 0) put(G2 + l, tag + 1, acc);
      }
    }
    if (!gather<1>(G2, L, tag + 1, dw_l, dead, state + 1)) { a.dsums[tid] = NAN; if (w == 0 && tid == 0) state[0] = state[0] + 1u; return; }      // a give-up poisons d(sums): the captioner's gradients turn NaN
    // ---- 3: softmax backward (every wave the statistics; waves 0-3 store a quarter of ddot each) ----
    {
      float dwv[4], dot = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) { dwv[k] = lane + 64 * k < L ? dw_l[lane + 64 * k] : 0.f; dot = fmaf(wq[k], dwv[k], dot); }
      dot = wave_sum(dot);
      if (wv < 4) {
        const float wk = wv == 0 ? wq[0] : wv == 1 ? wq[1] : wv == 2 ? wq[2] : wq[3];
        const float dk = wv == 0 ? dwv[0] : wv == 1 ? dwv[1] : wv == 2 ? dwv[2] : dwv[3];
        const float dd = wk * (dk - dot);
        dd_l[tid] = tid < L ? dd : 0.f;
        if (w == 0 && tid < L) a.ddot[(long)t * a.ld_ddot + tid] = dd;
      }
    }
    __syncthreads();
    {
      float p = 0.f;
#pragma unroll
      for (int q = 0; q < 7; ++q) { const int l = lg + 32 * q; if (l < L) p = fmaf(dd_l[l], 1.f - tq[q] * tq[q], p); }
      part[lg * 16 + jj] = p;
    }
    __syncthreads();
    if (tid < UPW) {
      float s = 0.f;
#pragma unroll
      for (int g = 0; g < 32; ++g) s += part[g * 16 + jj];
      const float dah = aw_own * s;
      a.datt_h[(long)t * a.ld_datt_h + j] = dah;
      v_l[80 + jj] = dah;
    }
    if (t == 0) break;                                       // d h(-1) is nobody's input
    __syncthreads();
    // ---- 4: partial d h(t-1) from the own rows, reduce-scattered by unit ----
    {
      float acc = 0.f;
#pragma unroll
      for (int n = 0; n < 96; ++n) acc = fmaf(v_l[n], wr[n], acc);
      put(G3 + (long)(tid >> 4) * 512 + w * 16 + (tid & 15), tag + 2, acc);
    }
    if (!gather<1>(G3 + (long)w * 512, 512, tag + 2, rs_l, dead, state + 1)) { a.dsums[tid] = NAN; if (w == 0 && tid == 0) state[0] = state[0] + 1u; return; }      // a give-up poisons d(sums): the captioner's gradients turn NaN
    if (tid < UPW) {
      float s = 0.f;
#pragma unroll
      for (int p = 0; p < NWG; ++p) s += rs_l[p * 16 + jj];
      dh = s;
    }
  }
  if (w == 0 && tid == 0) state[0] = state[0] + 1u;
}

}  // namespace

extern "C" size_t l2s_cap_recur_state_bytes(int backward) { return 16 + (size_t)(backward ? B_TOK : F_TOK) * 8; }

static int cap_recur_ok(int S, int R_, int AH_, int L) { return S >= 1 && S <= 4096 && R_ == R && AH_ == AH && L >= 1 && L <= LMAX; }
extern "C" int l2s_cap_recur_supported(int S, int R_, int AH_, int L) { return cap_recur_ok(S, R_, AH_, L); }

extern "C" int l2s_cap_recur_fwd(const l2s_cap_recur_fwd_args* a, hipStream_t s) {
  if (!a || !cap_recur_ok(a->S, a->R, a->AH, a->L) || !a->state || ((size_t)a->state & 15)) return L2S_EINVAL;
  static bool attr_done = false;
  if (!attr_done) { (void)hipFuncSetAttribute((const void*)cap_recur_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_REQ); attr_done = true; }
  const l2s_cap_recur_fwd_args v = *a;
  L2S_LAUNCH(cap_recur_fwd_kernel, dim3(NWG), dim3(NT), LDS_REQ, s, v);
  return l2s_check_launch();
}
extern "C" int l2s_cap_recur_bwd(const l2s_cap_recur_bwd_args* a, hipStream_t s) {
  if (!a || !cap_recur_ok(a->S, a->R, a->AH, a->L) || !a->state || ((size_t)a->state & 15)) return L2S_EINVAL;
  static bool attr_done = false;
  if (!attr_done) { (void)hipFuncSetAttribute((const void*)cap_recur_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_REQ); attr_done = true; }
  const l2s_cap_recur_bwd_args v = *a;
  L2S_LAUNCH(cap_recur_bwd_kernel, dim3(NWG), dim3(NT), LDS_REQ, s, v);
  return l2s_check_launch();
}
