// Language side of the lang2seg train step on gfx950: skinny (batch-1) linear layers, the bi-LSTM cell,
// the spatial dynamic-filter correlation and the att2in2 captioner core, forward + backward, fp32.
// Reference: lib/layers/lang_encoder.py:27-82, lib/caption_models/AttModel.py:60-101,406-466,
// lib/misc/utils.py:43-53, pyutils/mask-faster-rcnn/lib/nets/network_cycle_res5_2.py:504-562.
// The recurrences are GEMV-shaped (M = 1): weights are streamed once per call, one wave per output row, wave-level shuffle
// reductions; HBM/L2-bandwidth- and latency-bound.  The row-batch linears outside the recurrences (M = 21 tokens, 196 attention
// locations) run on the exact-fp32 MFMA (linear_nt_mfma_kernel / linear_nn_mfma_kernel).
#include "common.h"
#include "../../include/lang2seg_hip.h"

namespace {

constexpr int MAXM = 24;   // rows handled per wave pass in the skinny kernels
typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sigm(float x) { return 1.f / (1.f + expf(-x)); }

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == 1) return fmaxf(v, 0.f);
  if (act == 2) return tanhf(v);
  return v;
}

// y[m][n] = act(sum_k x[m][k] w[n][k] + b[n] (+ y[m][n]));  a wave owns NW consecutive outputs n and M <= MT rows per pass: every x
// vector it loads is used for NW weight rows (the one-output-per-wave form re-read the whole x block per output: 144 MB of L1 traffic
// for the 21 x 3350 x 512 logit layer, 40 us).  VW = floats per load: 4 (rows 16-byte aligned), 2 (8-byte aligned: K = 3350) or 1.
template <int MT, int VW, int NW>
__global__ __launch_bounds__(256) void linear_fwd_kernel(const float* __restrict__ x, int ldx_, const float* __restrict__ w, int ldw,
                                                        const float* __restrict__ b, float* y, int ldy, int m0, int M, int N, int K,
                                                        int act, int accumulate) {
  const int n0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * NW, lane = threadIdx.x & 63;
  if (n0 >= N) return;
  m0 += blockIdx.y * MT;                       // row chunks of one launch
  float acc[MT][NW];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < NW; ++j) acc[m][j] = 0.f;
  for (int k = lane * VW; k < K; k += 64 * VW) {
    float wv[NW][VW];
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const float* wr = w + (long)min(n0 + j, N - 1) * ldw + k;
      if (VW == 4) { const float4 t = *(const float4*)wr; wv[j][0] = t.x; wv[j][1] = t.y; wv[j][VW > 2 ? 2 : 0] = t.z; wv[j][VW > 3 ? 3 : 0] = t.w; }
      else if (VW == 2) { const float2 t = *(const float2*)wr; wv[j][0] = t.x; wv[j][VW > 1 ? 1 : 0] = t.y; }
      else wv[j][0] = *wr;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if (m0 + m < M) {
        const float* xr = x + (long)(m0 + m) * ldx_ + k;
        float xv[VW];
        if (VW == 4) { const float4 t = *(const float4*)xr; xv[0] = t.x; xv[1] = t.y; xv[VW > 2 ? 2 : 0] = t.z; xv[VW > 3 ? 3 : 0] = t.w; }
        else if (VW == 2) { const float2 t = *(const float2*)xr; xv[0] = t.x; xv[VW > 1 ? 1 : 0] = t.y; }
        else xv[0] = *xr;
#pragma unroll
        for (int j = 0; j < NW; ++j)
#pragma unroll
          for (int e = 0; e < VW; ++e) acc[m][j] = fmaf(wv[j][e], xv[e], acc[m][j]);
      }
    }
  }
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < NW; ++j) acc[m][j] = wave_sum(acc[m][j]);
  if (lane == 0) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
      if (m0 + m < M) {
#pragma unroll
        for (int j = 0; j < NW; ++j)
          if (n0 + j < N) {
            float v = acc[m][j] + (b ? b[n0 + j] : 0.f);
            float* o = y + (long)(m0 + m) * ldy + n0 + j;
            if (accumulate) v += *o;
            *o = act_apply(v, act);
          }
      }
  }
}

// dx[m][k] (+)= sum_n dy[m][n] w[n][k]  (transposed GEMV): block = 64 k-columns (one per lane) x 16 n-groups
// (one per wave); every wave streams whole 256-byte rows of w; cross-wave reduction through LDS; no atomics.
template <int MT>
__global__ __launch_bounds__(1024) void gemvT_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ w, float* dx,
                                                    int lddx, int m0, int M, int N, int K, int accumulate) {
  __shared__ float sh[MT][16][64];
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane;
  float acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) acc[m] = 0.f;
  if (k < K) {
    for (int n = g; n < N; n += 16) {
      const float wv = w[(long)n * K + k];
#pragma unroll
      for (int m = 0; m < MT; ++m)
        if (m0 + m < M) acc[m] = fmaf(dy[(long)(m0 + m) * lddy + n], wv, acc[m]);
    }
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) sh[m][g][lane] = acc[m];
  __syncthreads();
  // 1024 threads reduce MT*64 outputs: thread -> (m = tid / 64, lane)
  for (int o = threadIdx.x; o < MT * 64; o += 1024) {
    const int m = o >> 6, l = o & 63, kk = blockIdx.x * 64 + l;
    if (m0 + m < M && kk < K) {
      float s = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) s += sh[m][j][l];
      float* out = dx + (long)(m0 + m) * lddx + kk;
      *out = accumulate ? (*out + s) : s;
    }
  }
}

// dw[n][k] += sum_m dy[m][n] x[m][k]; db[n] += sum_m dy[m][n].  A block owns 256 columns k and NB rows n (dy[m][n] is uniform: scalar
// loads); single owner per output, no atomics.
__global__ __launch_bounds__(256) void linear_bwd_w_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx_,
                                                          float* dw, float* db, int M, int N, int K) {
  constexpr int NB = 1;                    // (8 rows per block was measured slower inside the step: 378 vs 319 us for the 12 launches)
  const int n0 = blockIdx.y * NB;
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k < K) {
    float acc[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) acc[j] = 0.f;
    for (int m = 0; m < M; ++m) {
      const float xv = x[(long)m * ldx_ + k];
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[j] = fmaf(dy[(long)m * lddy + min(n0 + j, N - 1)], xv, acc[j]);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) if (n0 + j < N) dw[(long)(n0 + j) * K + k] += acc[j];
  }
  if (db && blockIdx.x == 0 && threadIdx.x < NB && n0 + threadIdx.x < N) {
    float s = 0.f;
    for (int m = 0; m < M; ++m) s += dy[(long)m * lddy + n0 + threadIdx.x];
    db[n0 + threadIdx.x] += s;
  }
}

__global__ void act_bwd_kernel(float* dy, const float* y, long n, int act) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float v = y[i];
    if (act == 1) { if (!(v > 0.f)) dy[i] = 0.f; }
    else if (act == 2) dy[i] *= (1.f - v * v);
  }
}

// x *= mul (optional); x = 0 where !(relu_ref > 0); out = x in the activation dtype: dropout mask, ReLU backward and the cast in front of a
// convolution's data gradient in ONE launch (the tail of the captioner's backward pass, on the caption branch's critical chain)
template <typename T>
__global__ void mask_relu_cast_kernel(float* x, const float* __restrict__ mul, const float* __restrict__ relu_ref, T* out, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = x[i];
    if (mul) v *= mul[i];
    if (!(relu_ref[i] > 0.f)) v = 0.f;
    x[i] = v;
    Elem<T>::st(out + i, v);
  }
}

__global__ void embed_fwd_kernel(const float* table, const int64_t* ids, const float* mask, float* out, int T, int D, int relu) {
  const int t = blockIdx.x;
  const float* row = table + ids[t] * (long)D;
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    float v = row[d];
    if (relu) v = fmaxf(v, 0.f);
    if (mask) v *= mask[(long)t * D + d];
    out[(long)t * D + d] = v;
  }
}
// The same token may occur more than once in an expression: the block of its FIRST occurrence adds all of them into the table row, in
// position order (single owner per row: no atomics, bit-reproducible).
__global__ void embed_bwd_kernel(const float* dout, const float* out, const int64_t* ids, const float* mask, float* dtable, int T, int D, int relu) {
  const int t = blockIdx.x;
  const int64_t id = ids[t];
  for (int u = 0; u < t; ++u) if (ids[u] == id) return;          // (uniform) an earlier position owns this row
  float* row = dtable + id * (long)D;
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    float acc = row[d];
    for (int u = t; u < T; ++u) {
      if (ids[u] != id) continue;
      float g = dout[(long)u * D + d];
      if (mask) g *= mask[(long)u * D + d];
      if (relu && out[(long)u * D + d] == 0.f) g = 0.f;   // relu killed it (or the mask did, then g is 0 already)
      acc += g;
    }
    row[d] = acc;
  }
}


__global__ void lstm_cell_fwd_kernel(const float* g, const float* c_prev, float* c, float* h, float* act, int Hh) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= Hh) return;
  const float i = sigm(g[j]), f = sigm(g[Hh + j]), gg = tanhf(g[2 * Hh + j]), o = sigm(g[3 * Hh + j]);
  const float cn = f * c_prev[j] + i * gg;
  c[j] = cn; h[j] = o * tanhf(cn);
  act[j] = i; act[Hh + j] = f; act[2 * Hh + j] = gg; act[3 * Hh + j] = o;
}
__global__ void lstm_cell_bwd_kernel(const float* dh, const float* dc_in, const float* act, const float* c_prev, const float* c,
                                     float* dg, float* dc_prev, int Hh) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= Hh) return;
  const float i = act[j], f = act[Hh + j], gg = act[2 * Hh + j], o = act[3 * Hh + j];
  const float tc = tanhf(c[j]);
  const float dcn = (dc_in ? dc_in[j] : 0.f) + dh[j] * o * (1.f - tc * tc);
  dg[j] = dcn * gg * i * (1.f - i);
  dg[Hh + j] = dcn * c_prev[j] * f * (1.f - f);
  dg[2 * Hh + j] = dcn * i * (1.f - gg * gg);
  dg[3 * Hh + j] = dh[j] * tc * o * (1.f - o);
  dc_prev[j] = dcn * f;
}


// ---- fused bi-LSTM steps (ENC:62, nn.LSTM gate order i,f,g,o).  One launch per time step for BOTH directions (grid.y = direction):
// forward : gates = g_in (x W_ih + b_ih, precomputed for all steps) + W_hh h_prev + b_hh, then the cell update; one wave per unit.
// backward: dh = W_hh^T dg_next (through the [Hh][4Hh] transposed copy; skipped at the first step, where dh is the gradient of the
//           final hidden state) and then the cell backward of this step for the same unit -- both are per-unit once dg_next is known.
struct LstmDir { const float* w; const float* b; const float* g_in; const float* h_prev; const float* c_prev; float* c; float* h; float* act; float* g_out; };
__global__ __launch_bounds__(256) void lstm_step_fwd_kernel(LstmDir d0, LstmDir d1, int Hh) {
  const LstmDir d = blockIdx.y ? d1 : d0;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (j >= Hh) return;
  float a[4] = {0.f, 0.f, 0.f, 0.f};
#if defined(L2S_LSTM_PROBE) && L2S_LSTM_PROBE == 10
  float a2[4] = {0.f, 0.f, 0.f, 0.f};
#endif
#if defined(L2S_LSTM_PROBE) && (L2S_LSTM_PROBE == 13 || L2S_LSTM_PROBE == 14 || (L2S_LSTM_PROBE >= 16 && L2S_LSTM_PROBE <= 19))
  float a3[4] = {0.f, 0.f, 0.f, 0.f};
#endif
#if defined(L2S_LSTM_PROBE) && L2S_LSTM_PROBE == 5
  // probe 5: the same loop with a UNIFORM trip count (Hh % 256 == 0 in every caller of the probe): scalar loop control, no write of EXEC
  // anywhere near the packed FMAs
  const int trips = __builtin_amdgcn_readfirstlane(Hh >> 8);
  for (int it = 0; it < trips; ++it) {
    const int k = lane * 4 + (it << 8);
#else
  for (int k = lane * 4; k < Hh; k += 256) {
#endif
    const float4 hv = *(const float4*)(d.h_prev + k);
#if defined(L2S_LSTM_PROBE) && L2S_LSTM_PROBE == 13
    // probe 13 (packed build): v_pk_fma_f32 on NATURAL register pairs - (w.x, w.y) * (h.x, h.y) and (w.z, w.w) * (h.z, h.w) into a two-lane
    // accumulator per gate, the lanes added after the loop: every packed source is an aligned pair exactly as the load delivered it (no
    // v_mov shuffles, no op_sel); a different summation order, so compared only for run-to-run identity
    {
      typedef float f2v __attribute__((ext_vector_type(2)));
      const f2v h01 = {hv.x, hv.y}, h23 = {hv.z, hv.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 wv = *(const float4*)(d.w + (long)(q * Hh + j) * Hh + k);
        const f2v w01 = {wv.x, wv.y}, w23 = {wv.z, wv.w};
        f2v acc = {a[q], a3[q]};
        acc = __builtin_elementwise_fma(w23, h23, acc);
        acc = __builtin_elementwise_fma(w01, h01, acc);
        a[q] = acc.x; a3[q] = acc.y;
      }
    }
#elif defined(L2S_LSTM_PROBE) && L2S_LSTM_PROBE == 14
    // probe 14 (packed build; NOT the LSTM's arithmetic - compared for run-to-run identity only): natural source pairs as probe 13, but the second
    // factor is one register broadcast to both lanes, i.e. v_pk_fma_f32 ... op_sel_hi:[1,0,1] / op_sel:[0,1,0] without any v_mov shuffle
    {
      typedef float f2v __attribute__((ext_vector_type(2)));
      const f2v hx = {hv.x, hv.x}, hy = {hv.y, hv.y}, hz = {hv.z, hv.z}, hw = {hv.w, hv.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 wv = *(const float4*)(d.w + (long)(q * Hh + j) * Hh + k);
        const f2v w01 = {wv.x, wv.y}, w23 = {wv.z, wv.w};
        f2v acc = {a[q], a3[q]};
        acc = __builtin_elementwise_fma(w23, hw, acc);
        acc = __builtin_elementwise_fma(w01, hz, acc);
        acc = __builtin_elementwise_fma(w23, hy, acc);
        acc = __builtin_elementwise_fma(w01, hx, acc);
        a[q] = acc.x; a3[q] = acc.y;
      }
    }
#elif defined(L2S_LSTM_PROBE) && (L2S_LSTM_PROBE == 18 || L2S_LSTM_PROBE == 19)
    // probes 18 / 19 (NOT the LSTM's arithmetic): the OTHER broadcast form by hand - v_pk_fma_f32 acc, w, hh, acc op_sel:[0,1,0], i.e. BOTH lanes take
    // hh's HIGH register (the low lane through op_sel, the high lane by default) - with hh's unused LOW register holding 18: 1000 x the value,
    // 19: the same value
    {
      typedef float f2v __attribute__((ext_vector_type(2)));
      const float hcs[4] = {hv.w, hv.z, hv.y, hv.x};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 wv = *(const float4*)(d.w + (long)(q * Hh + j) * Hh + k);
        const f2v w01 = {wv.x, wv.y}, w23 = {wv.z, wv.w};
        f2v acc = {a[q], a3[q]};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f2v hh = {L2S_LSTM_PROBE == 19 ? hcs[c] : 1000.f * hcs[c], hcs[c]};
          if (c & 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc) : "v"(w01), "v"(hh));
          else       asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc) : "v"(w23), "v"(hh));
        }
        a[q] = acc.x; a3[q] = acc.y;
      }
    }
#elif defined(L2S_LSTM_PROBE) && (L2S_LSTM_PROBE == 16 || L2S_LSTM_PROBE == 17)
    // probes 16 / 17 (NOT the LSTM's arithmetic): the broadcast form written by hand - v_pk_fma_f32 acc, w, hh, acc op_sel_hi:[1,0,1], i.e. BOTH
    // lanes take hh's LOW register - with hh's unused HIGH register holding 16: the same value (a wrong lane select cannot show), 17: 1000 x the
    // value (a wrong lane select shows as a 1000 x too large term)
    {
      typedef float f2v __attribute__((ext_vector_type(2)));
      const float hcs[4] = {hv.w, hv.z, hv.y, hv.x};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 wv = *(const float4*)(d.w + (long)(q * Hh + j) * Hh + k);
        const f2v w01 = {wv.x, wv.y}, w23 = {wv.z, wv.w};
        f2v acc = {a[q], a3[q]};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const f2v hh = {hcs[c], L2S_LSTM_PROBE == 16 ? hcs[c] : 1000.f * hcs[c]};
          if (c & 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(w01), "v"(hh));
          else       asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(w23), "v"(hh));
        }
        a[q] = acc.x; a3[q] = acc.y;
      }
    }
#elif defined(L2S_LSTM_PROBE) && L2S_LSTM_PROBE == 15
    // probe 15 (packed build; NOT the LSTM's arithmetic): first factors assembled ACROSS two loads - (w_q.c, w_{q+1}.c), the pairs the compiler
    // builds with v_mov_b32 in the product source - times NATURAL pairs of h (no broadcast, no op_sel)
    {
      typedef float f2v __attribute__((ext_vector_type(2)));
      const f2v h01 = {hv.x, hv.y}, h23 = {hv.z, hv.w};
      float4 wq[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) wq[q] = *(const float4*)(d.w + (long)(q * Hh + j) * Hh + k);
#pragma unroll
      for (int q = 0; q < 4; q += 2) {
        f2v acc = {a[q], a[q + 1]};
        acc = __builtin_elementwise_fma((f2v){wq[q].w, wq[q + 1].w}, h23, acc);
        acc = __builtin_elementwise_fma((f2v){wq[q].z, wq[q + 1].z}, h01, acc);
        acc = __builtin_elementwise_fma((f2v){wq[q].y, wq[q + 1].y}, h23, acc);
        acc = __builtin_elementwise_fma((f2v){wq[q].x, wq[q + 1].x}, h01, acc);
        a[q] = acc.x; a[q + 1] = acc.y;
      }
    }
#elif defined(L2S_LSTM_PROBE) && (L2S_LSTM_PROBE == 10 || L2S_LSTM_PROBE == 12)
    // probes 10 / 12 (packed build): the same products in the same order per accumulator, but component by component across the four gates -
    // 10: with a second set of accumulators for .w/.y so that a packed FMA never reads the result of the packed FMA two instructions before it;
    // 12: with four wait states between the component steps
    float4 wq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) wq[q] = *(const float4*)(d.w + (long)(q * Hh + j) * Hh + k);
#if L2S_LSTM_PROBE == 10
#pragma unroll
    for (int q = 0; q < 4; ++q) a2[q] = fmaf(wq[q].w, hv.w, a2[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = fmaf(wq[q].z, hv.z, a[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) a2[q] = fmaf(wq[q].y, hv.y, a2[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = fmaf(wq[q].x, hv.x, a[q]);
#else
#define L2S_NOP4 asm volatile("s_nop 3" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = fmaf(wq[q].w, hv.w, a[q]);
    L2S_NOP4
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = fmaf(wq[q].z, hv.z, a[q]);
    L2S_NOP4
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = fmaf(wq[q].y, hv.y, a[q]);
    L2S_NOP4
#pragma unroll
    for (int q = 0; q < 4; ++q) a[q] = fmaf(wq[q].x, hv.x, a[q]);
    L2S_NOP4
#undef L2S_NOP4
#endif
#else
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 wv = *(const float4*)(d.w + (long)(q * Hh + j) * Hh + k);
      a[q] = fmaf(wv.x, hv.x, fmaf(wv.y, hv.y, fmaf(wv.z, hv.z, fmaf(wv.w, hv.w, a[q]))));
    }
#endif
#if defined(L2S_LSTM_PROBE) && (L2S_LSTM_PROBE == 4 || L2S_LSTM_PROBE == 7)
    // probes 4 / 7: wait states between the last packed FMAs of an iteration and the loop control's write of EXEC (s_andn2_b64 exec)
    asm volatile("s_nop 7" :: "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]));
#endif
  }
#if defined(L2S_LSTM_PROBE) && L2S_LSTM_PROBE == 10
#pragma unroll
  for (int q = 0; q < 4; ++q) a[q] += a2[q];
#endif
#if defined(L2S_LSTM_PROBE) && (L2S_LSTM_PROBE == 13 || L2S_LSTM_PROBE == 14 || (L2S_LSTM_PROBE >= 16 && L2S_LSTM_PROBE <= 19))
#pragma unroll
  for (int q = 0; q < 4; ++q) a[q] += a3[q];
#endif
#if defined(L2S_LSTM_PROBE)
  // tools/lstm_pk_probe.sh only: variants of the hand-off between the accumulation loop - which the compiler pairs into
  // v_pk_fma_f32 when packed fp32 ops are enabled - and the DPP reduction that reads the LOW halves of those pairs first.
#if L2S_LSTM_PROBE == 1
  asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));                       // time only
#elif L2S_LSTM_PROBE == 2 || L2S_LSTM_PROBE == 7
  asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1\n\tv_mov_b32 %2, %2\n\tv_mov_b32 %3, %3" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));   // a plain VALU read + rewrite of each half
#elif L2S_LSTM_PROBE == 3
  asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]));                                           // the constraint alone (register allocation changes, no instruction)
#endif
#endif
#if defined(L2S_LSTM_PROBE) && L2S_LSTM_PROBE == 8
  // probe 8 (packed build): the four reductions strictly one after the other (the compiler then separates dependent DPP steps by s_nop 1,
  // as in the product build) instead of the two interleaved low-half chains it schedules behind v_pk_fma_f32
#pragma unroll
  for (int q = 0; q < 4; ++q) { a[q] = wave_sum(a[q]); __builtin_amdgcn_sched_barrier(0); }
#elif defined(L2S_LSTM_PROBE) && L2S_LSTM_PROBE == 9
  // probe 9 (UNPACKED build): two reductions interleaved step by step by hand, the shape the packed build's scheduler produces
  {
    float x = a[0], y = a[2];
#define L2S_STEP(C, M) x += dpp_take<C, M>(0.f, x); __builtin_amdgcn_sched_barrier(0); y += dpp_take<C, M>(0.f, y); __builtin_amdgcn_sched_barrier(0);
    L2S_STEP(0x111, 0xf) L2S_STEP(0x112, 0xf) L2S_STEP(0x114, 0xf) L2S_STEP(0x118, 0xf) L2S_STEP(0x142, 0xa) L2S_STEP(0x143, 0xc)
#undef L2S_STEP
    a[0] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63));
    a[2] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(y), 63));
    a[1] = wave_sum(a[1]); a[3] = wave_sum(a[3]);
  }
#else
#pragma unroll
  for (int q = 0; q < 4; ++q) a[q] = wave_sum(a[q]);
#endif
  if (lane == 0) {
    float g[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { g[q] = d.g_in[q * Hh + j] + a[q] + d.b[q * Hh + j]; d.g_out[q * Hh + j] = g[q]; }
    const float i = sigm(g[0]), f = sigm(g[1]), gg = tanhf(g[2]), o = sigm(g[3]);
    const float cn = f * d.c_prev[j] + i * gg;
    d.c[j] = cn; d.h[j] = o * tanhf(cn);
    d.act[j] = i; d.act[Hh + j] = f; d.act[2 * Hh + j] = gg; d.act[3 * Hh + j] = o;
  }
}
struct LstmBDir { const float* wT; const float* dg_next; const float* dh_ext; const float* dc_in; const float* act; const float* c_prev; const float* c; float* dg; float* dc_prev; };
__global__ __launch_bounds__(256) void lstm_step_bwd_kernel(LstmBDir d0, LstmBDir d1, int Hh) {
  const LstmBDir d = blockIdx.y ? d1 : d0;
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (j >= Hh) return;
  float dh = 0.f;
  if (d.dg_next) {
    const float* wr = d.wT + (long)j * 4 * Hh;
    for (int k = lane * 4; k < 4 * Hh; k += 256) {
      const float4 wv = *(const float4*)(wr + k), gv = *(const float4*)(d.dg_next + k);
      dh = fmaf(wv.x, gv.x, fmaf(wv.y, gv.y, fmaf(wv.z, gv.z, fmaf(wv.w, gv.w, dh))));
    }
    dh = wave_sum(dh);
  }
  if (lane == 0) {
    if (d.dh_ext) dh += d.dh_ext[j];
    const float i = d.act[j], f = d.act[Hh + j], gg = d.act[2 * Hh + j], o = d.act[3 * Hh + j];
    const float tc = tanhf(d.c[j]);
    const float dcn = (d.dc_in ? d.dc_in[j] : 0.f) + dh * o * (1.f - tc * tc);
    d.dg[j] = dcn * gg * i * (1.f - i);
    d.dg[Hh + j] = dcn * d.c_prev[j] * f * (1.f - f);
    d.dg[2 * Hh + j] = dcn * i * (1.f - gg * gg);
    d.dg[3 * Hh + j] = dh * tc * o * (1.f - o);
    d.dc_prev[j] = dcn * f;
  }
}

// ---------------------------------------------------------------- dynamic filter correlation
__device__ __forceinline__ void spatial_mask7(int y, int x, int H, int W, float m[7]) {
  // NET:530-557 (python-2 true division then int())
  m[0] = 1.f;
  m[1] = (y < (int)(H / 2.0)) ? 1.f : 0.f;
  m[2] = (y >= (int)(H / 2.0)) ? 1.f : 0.f;
  m[3] = (x < (int)(W / 2.0)) ? 1.f : 0.f;
  m[4] = (x >= (int)(W / 2.0)) ? 1.f : 0.f;
  m[5] = (y >= (int)(H / 4.0) && y < (int)(H * 3 / 4.0)) ? 1.f : 0.f;
  m[6] = (x >= (int)(W / 4.0) && x < (int)(W * 3 / 4.0)) ? 1.f : 0.f;
}
// one wave per pixel: 7 masked channel dot products, 7->1 mix, modulate
// gate 0: y = x * response (NET:562); gate 1: y = x * sigmoid(response) (network_cycle_response.py:568-570).  resp keeps the raw response.
__global__ __launch_bounds__(256) void dynfilter_fwd_kernel(const void* x, const float* __restrict__ filt, const float* __restrict__ r, void* y,
                                                           float* resp, float* respk, int H, int W, int C, int dt, int gate) {
  const int pix = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (pix >= H * W) return;
  float d[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int c = lane; c < C; c += 64) {
    const float v = ldx(x, (long)pix * C + c, dt);
#pragma unroll
    for (int k = 0; k < 7; ++k) d[k] = fmaf(v, filt[k * C + c], d[k]);
  }
  float m[7]; spatial_mask7(pix / W, pix % W, H, W, m);
  float rs = 0.f;
#pragma unroll
  for (int k = 0; k < 7; ++k) { d[k] = wave_sum(d[k]) * m[k]; rs = fmaf(r[k], d[k], rs); }
  if (lane < 7) respk[(long)pix * 7 + lane] = d[lane];
  if (lane == 0) resp[pix] = rs;
  const float mult = gate ? sigm(rs) : rs;
  for (int c = lane; c < C; c += 64) stx(y, (long)pix * C + c, dt, ldx(x, (long)pix * C + c, dt) * mult);
}
// bf16, C % 8 == 0, C <= 2048: 8 channels (16 bytes) per lane and trip, the pixel's values stay in registers for the modulation
__global__ __launch_bounds__(256) void dynfilter_fwd_bf16_kernel(const bf16_t* __restrict__ x, const float* __restrict__ filt, const float* __restrict__ r, bf16_t* y,
                                                                float* resp, float* respk, int H, int W, int C, int gate) {
  const int pix = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (pix >= H * W) return;
  float d[7] = {0, 0, 0, 0, 0, 0, 0};
  float xv[4][8];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int c = (lane + 64 * t) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) xv[t][e] = 0.f;
    if (c < C) {
      const uint4 q = *(const uint4*)(x + (long)pix * C + c);
      const uint32_t qw[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { xv[t][2 * e] = __uint_as_float(qw[e] << 16); xv[t][2 * e + 1] = __uint_as_float(qw[e] & 0xFFFF0000u); }
#pragma unroll
      for (int k = 0; k < 7; ++k) {
        const float4 f0 = *(const float4*)(filt + k * C + c), f1 = *(const float4*)(filt + k * C + c + 4);
        d[k] = fmaf(xv[t][0], f0.x, d[k]); d[k] = fmaf(xv[t][1], f0.y, d[k]); d[k] = fmaf(xv[t][2], f0.z, d[k]); d[k] = fmaf(xv[t][3], f0.w, d[k]);
        d[k] = fmaf(xv[t][4], f1.x, d[k]); d[k] = fmaf(xv[t][5], f1.y, d[k]); d[k] = fmaf(xv[t][6], f1.z, d[k]); d[k] = fmaf(xv[t][7], f1.w, d[k]);
      }
    }
  }
  float m[7]; spatial_mask7(pix / W, pix % W, H, W, m);
  float rs = 0.f;
#pragma unroll
  for (int k = 0; k < 7; ++k) { d[k] = wave_sum(d[k]) * m[k]; rs = fmaf(r[k], d[k], rs); }
  if (lane < 7) {
    float dv = d[0];
#pragma unroll
    for (int k = 1; k < 7; ++k) if (lane == k) dv = d[k];
    respk[(long)pix * 7 + lane] = dv;
  }
  if (lane == 0) resp[pix] = rs;
  const float mult = gate ? sigm(rs) : rs;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int c = (lane + 64 * t) * 8;
    if (c < C) {
      uint4 o;
      o.x = (uint32_t)f2bf(xv[t][0] * mult) | ((uint32_t)f2bf(xv[t][1] * mult) << 16); o.y = (uint32_t)f2bf(xv[t][2] * mult) | ((uint32_t)f2bf(xv[t][3] * mult) << 16);
      o.z = (uint32_t)f2bf(xv[t][4] * mult) | ((uint32_t)f2bf(xv[t][5] * mult) << 16); o.w = (uint32_t)f2bf(xv[t][6] * mult) | ((uint32_t)f2bf(xv[t][7] * mult) << 16);
      *(uint4*)(y + (long)pix * C + c) = o;
    }
  }
}
// pass 1: dresp[p] = sum_c dy[p][c] x[p][c]   (dr[k] += sum_p dresp[p]*respk[p][k] is taken by pass 3, in a fixed order)
// (sigmoid gate: times sigma'(response); dresp_extra = gradient of the response BCE loss w.r.t. the raw response)
__global__ __launch_bounds__(256) void dynfilter_bwd1_kernel(const void* dy, const void* x, const float* respk, float* dresp, float* dr, int HW, int C, int dt,
                                                            int gate, const float* resp, const float* dresp_extra) {
  const int pix = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (pix >= HW) return;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) s = fmaf(ldx(dy, (long)pix * C + c, dt), ldx(x, (long)pix * C + c, dt), s);
  s = wave_sum(s);
  if (gate) { const float sg = sigm(resp[pix]); s *= sg * (1.f - sg); }
  if (dresp_extra) s += dresp_extra[pix];
  if (lane == 0) dresp[pix] = s;
}
// pass 2: dx[p][c] = dy*resp + dresp[p]*sum_k r_k m_k[p] f_k[c] (ReLU-masked by relu_ref); part[chunk][k][c] = sum_{p in chunk} dresp r_k m_k x[p][c]
// block = 64 channels x 4 pixel lanes, grid.y over pixel chunks; pass 3 adds the chunks' partial sums to dfilt in chunk order (no atomics)
__global__ __launch_bounds__(256) void dynfilter_bwd2_kernel(const void* dy, const void* x, const float* __restrict__ filt, const float* __restrict__ r,
                                                            const float* __restrict__ resp, const float* __restrict__ dresp, void* dx, const void* ref,
                                                            float* dfilt, int H, int W, int C, int dt, int pchunk, int gate) {
  __shared__ float red[4][7][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), pl = threadIdx.x >> 6;
  const int p0 = blockIdx.y * pchunk, p1 = min(H * W, p0 + pchunk);
  float fk[7], acc[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) { fk[k] = (c < C) ? filt[k * C + c] * r[k] : 0.f; acc[k] = 0.f; }
  for (int p = p0 + pl; p < p1; p += 4) {
    if (c >= C) break;
    float m[7]; spatial_mask7(p / W, p % W, H, W, m);
    const float dr_ = dresp[p];
    const float xv = ldx(x, (long)p * C + c, dt);
    float g = ldx(dy, (long)p * C + c, dt) * (gate ? sigm(resp[p]) : resp[p]);
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 7; ++k) { t = fmaf(m[k], fk[k], t); acc[k] = fmaf(dr_ * m[k], xv, acc[k]); }
    g = fmaf(dr_, t, g);
    if (ref && !(ldx(ref, (long)p * C + c, dt) > 0.f)) g = 0.f;
    stx(dx, (long)p * C + c, dt, g);
  }
#pragma unroll
  for (int k = 0; k < 7; ++k) red[pl][k][threadIdx.x & 63] = acc[k];
  __syncthreads();
  if (pl == 0 && c < C) {
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const int l = threadIdx.x;
      dfilt[((long)blockIdx.y * 7 + k) * C + c] = (red[0][k][l] + red[1][k][l] + red[2][k][l] + red[3][k][l]) * r[k];
    }
  }
}

// bf16 fast paths of passes 1 and 2 (C % 8 == 0 / C % 4 == 0, 16- and 8-byte accesses instead of one bf16 per lane).
__global__ __launch_bounds__(256) void dynfilter_bwd1_bf16_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, float* dresp, int HW, int C,
                                                                 int gate, const float* __restrict__ resp, const float* __restrict__ dresp_extra) {
  const int pix = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (pix >= HW) return;
  float s = 0.f;
  for (int c = lane * 8; c < C; c += 512) {
    const uint4 a = *(const uint4*)(dy + (long)pix * C + c), b = *(const uint4*)(x + (long)pix * C + c);
    const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s = fmaf(__uint_as_float(aw[k] << 16), __uint_as_float(bw[k] << 16), s);
      s = fmaf(__uint_as_float(aw[k] & 0xFFFF0000u), __uint_as_float(bw[k] & 0xFFFF0000u), s);
    }
  }
  s = wave_sum(s);
  if (gate) { const float sg = sigm(resp[pix]); s *= sg * (1.f - sg); }
  if (dresp_extra) s += dresp_extra[pix];
  if (lane == 0) dresp[pix] = s;
}
// block = 256 channels (64 lanes x 4) x 4 pixel lanes
__global__ __launch_bounds__(256) void dynfilter_bwd2_bf16_kernel(const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const float* __restrict__ filt,
                                                                 const float* __restrict__ r, const float* __restrict__ resp, const float* __restrict__ dresp,
                                                                 bf16_t* dx, const bf16_t* __restrict__ ref, float* dfilt, int H, int W, int C, int pchunk, int gate) {
  __shared__ float red[4][7][256];
  const int lane = threadIdx.x & 63, pl = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + lane * 4;
  const int p0 = blockIdx.y * pchunk, p1 = min(H * W, p0 + pchunk);
  float fk[7][4], acc[7][4];
#pragma unroll
  for (int k = 0; k < 7; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) { fk[k][e] = (c < C) ? filt[k * C + c + e] * r[k] : 0.f; acc[k][e] = 0.f; }
  if (c < C) {
    for (int p = p0 + pl; p < p1; p += 4) {
      float m[7]; spatial_mask7(p / W, p % W, H, W, m);
      const float dr_ = dresp[p], mult = gate ? sigm(resp[p]) : resp[p];
      const uint2 xr = *(const uint2*)(x + (long)p * C + c), gr = *(const uint2*)(dy + (long)p * C + c);
      uint2 rr = make_uint2(0x3f803f80u, 0x3f803f80u);
      if (ref) rr = *(const uint2*)(ref + (long)p * C + c);
      const float xv[4] = {__uint_as_float(xr.x << 16), __uint_as_float(xr.x & 0xFFFF0000u), __uint_as_float(xr.y << 16), __uint_as_float(xr.y & 0xFFFF0000u)};
      const float gv[4] = {__uint_as_float(gr.x << 16), __uint_as_float(gr.x & 0xFFFF0000u), __uint_as_float(gr.y << 16), __uint_as_float(gr.y & 0xFFFF0000u)};
      const float rv[4] = {__uint_as_float(rr.x << 16), __uint_as_float(rr.x & 0xFFFF0000u), __uint_as_float(rr.y << 16), __uint_as_float(rr.y & 0xFFFF0000u)};
      float o[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float g = gv[e] * mult;
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 7; ++k) { t = fmaf(m[k], fk[k][e], t); acc[k][e] = fmaf(dr_ * m[k], xv[e], acc[k][e]); }
        g = fmaf(dr_, t, g);
        if (!(rv[e] > 0.f)) g = 0.f;
        o[e] = g;
      }
      *(uint2*)(dx + (long)p * C + c) = make_uint2((uint32_t)f2bf(o[0]) | ((uint32_t)f2bf(o[1]) << 16), (uint32_t)f2bf(o[2]) | ((uint32_t)f2bf(o[3]) << 16));
    }
  }
#pragma unroll
  for (int k = 0; k < 7; ++k)
#pragma unroll
    for (int e = 0; e < 4; ++e) red[pl][k][lane * 4 + e] = acc[k][e];
  __syncthreads();
  // 256 threads: thread t finishes channel blockIdx.x * 256 + t for the seven taps
  const int cc = blockIdx.x * 256 + threadIdx.x;
  if (cc < C) {
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const int l = threadIdx.x;
      dfilt[((long)blockIdx.y * 7 + k) * C + cc] = (red[0][k][l] + red[1][k][l] + red[2][k][l] + red[3][k][l]) * r[k];
    }
  }
}
// pass 3: dfilt[k][c] += sum_chunk part[chunk][k][c] (chunk order); block 0 also dr[k] += sum_p dresp[p] respk[p][k] (fixed tree)
__global__ __launch_bounds__(256) void dynfilter_bwd3_kernel(const float* __restrict__ part, int nchunk, const float* __restrict__ dresp,
                                                            const float* __restrict__ respk, float* dfilt, float* dr, int HW, int C) {
  __shared__ float red[4];
  const int e = blockIdx.x * 256 + threadIdx.x;              // (k, c) flattened
  if (e < 7 * C) {
    float s = 0.f;
    for (int q = 0; q < nchunk; ++q) s += part[(long)q * 7 * C + e];
    dfilt[e] += s;
  }
  if (blockIdx.x == 0) {
    for (int k = 0; k < 7; ++k) {
      float s = 0.f;
      for (int p = threadIdx.x; p < HW; p += 256) s = fmaf(dresp[p], respk[(long)p * 7 + k], s);
      s = block_sum(s, red);
      if (threadIdx.x == 0) dr[k] += s;
    }
  }
}

// ---------------------------------------------------------------- att2in2 attention (L <= 256)
// forward: (A) one wave per location: dots[l] = alpha . tanh(patt[l] + att_h) ; (B) D/64 workgroups: softmax over L (recomputed per
// workgroup, 196 values) and att_res[d] = sum_l w[l] att[l][d] with 4 l-groups per 64 channels.
__global__ __launch_bounds__(256) void cap_att_dots_kernel(const float* __restrict__ patt, const float* __restrict__ att_h, const float* __restrict__ aw,
                                                          const float* __restrict__ ab, int L, int D, float* tanh_ws, float* dots) {
  __builtin_amdgcn_s_setprio(3);   // a link of the caption branch's dependent chain (the step's critical path): issue ahead of co-resident GEMM waves
  const int l = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (l >= L) return;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) {
    const float t = tanhf(patt[(long)l * D + d] + att_h[d]);
    tanh_ws[(long)l * D + d] = t;
    s = fmaf(t, aw[d], s);
  }
  s = wave_sum(s);
  if (lane == 0) dots[l] = s + ab[0];
}
__global__ __launch_bounds__(256) void cap_att_apply_kernel(const float* __restrict__ att, const float* __restrict__ dots, int L, int D, float* weight, float* att_res) {
  __shared__ float w[256];
  __shared__ float red[4];
  __shared__ float part[4][64];
  const int tid = threadIdx.x, lane = tid & 63, g = tid >> 6;
  const float v = tid < L ? dots[tid] : -INFINITY;
  const float mx = block_max(v, red);
  const float e = tid < L ? expf(v - mx) : 0.f;
  const float sum = block_sum(e, red);
  if (tid < L) { w[tid] = e / sum; if (blockIdx.x == 0) weight[tid] = e / sum; }
  __syncthreads();
  const int d = blockIdx.x * 64 + lane;
  float s = 0.f;
  if (d < D) for (int l = g; l < L; l += 4) s = fmaf(w[l], att[(long)l * D + d], s);
  part[g][lane] = s;
  __syncthreads();
  if (g == 0 && d < D) att_res[d] = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
}
// backward: (A) dweight[l] = dres . att[l] (one wave per l); (B) D/64 workgroups: softmax backward over L (recomputed), then the
// per-channel accumulations with 4 l-groups per 64 channels.
__global__ __launch_bounds__(256) void cap_att_bwd_dw_kernel(const float* __restrict__ dres, const float* __restrict__ att, int L, int D, float* dweight) {
  const int l = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (l >= L) return;
  float s = 0.f;
  for (int d = lane; d < D; d += 64) s = fmaf(dres[d], att[(long)l * D + d], s);
  s = wave_sum(s);
  if (lane == 0) dweight[l] = s;
}
__global__ __launch_bounds__(256) void cap_att_bwd_kernel(const float* __restrict__ dres, const float* __restrict__ dweight, const float* __restrict__ tanh_ws,
                                                         const float* __restrict__ weight, const float* __restrict__ aw, int L, int D,
                                                         float* dpatt, float* datt, float* datt_h, float* daw, float* dab) {
  __shared__ float ddot[256];
  __shared__ float wsh[256];
  __shared__ float red[4];
  __shared__ float p1[4][64], p2[4][64];
  const int tid = threadIdx.x, lane = tid & 63, g = tid >> 6;
  const float wl = tid < L ? weight[tid] : 0.f;
  const float dwl = tid < L ? dweight[tid] : 0.f;
  const float dot = block_sum(wl * dwl, red);
  const float dd = wl * (dwl - dot);                      // softmax backward
  if (tid < 256) { ddot[tid] = tid < L ? dd : 0.f; wsh[tid] = wl; }
  const float sdd = block_sum(tid < L ? dd : 0.f, red);
  if (blockIdx.x == 0 && tid == 0) dab[0] += sdd;
  __syncthreads();
  const int d = blockIdx.x * 64 + lane;
  float sah = 0.f, saw = 0.f;
  if (d < D) {
    const float a = aw[d], dr = dres[d];
    for (int l = g; l < L; l += 4) {
      const float t = tanh_ws[(long)l * D + d];
      const float dt_ = ddot[l] * a * (1.f - t * t);
      dpatt[(long)l * D + d] += dt_;
      datt[(long)l * D + d] += wsh[l] * dr;
      sah += dt_;
      saw = fmaf(ddot[l], t, saw);
    }
  }
  p1[g][lane] = sah; p2[g][lane] = saw;
  __syncthreads();
  if (g == 0 && d < D) {
    datt_h[d] = p1[0][lane] + p1[1][lane] + p1[2][lane] + p1[3][lane];
    daw[d] += p2[0][lane] + p2[1][lane] + p2[2][lane] + p2[3][lane];
  }
}

__global__ void cap_gates_fwd_kernel(const float* s, const float* a2c, const float* c_prev, float* c, float* h, float* save, int R) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= R) return;
  const float ig = sigm(s[j]), fg = sigm(s[R + j]), og = sigm(s[2 * R + j]);
  const float t0 = s[3 * R + j] + a2c[j], t1 = s[4 * R + j] + a2c[R + j];
  const float it = fmaxf(t0, t1);
  const float cn = fg * c_prev[j] + ig * it;
  const float tc = tanhf(cn);
  c[j] = cn; h[j] = og * tc;
  save[j] = ig; save[R + j] = fg; save[2 * R + j] = og; save[3 * R + j] = (t0 >= t1) ? 0.f : 1.f;  // torch.max(a,b): first wins ties
  save[4 * R + j] = it; save[5 * R + j] = tc;
}
__global__ void cap_gates_bwd_kernel(const float* dh_a, const float* dh_b, const float* dc_in, const float* save, const float* c_prev, float* ds, float* da2c,
                                     float* dc_prev, int R) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= R) return;
  const float ig = save[j], fg = save[R + j], og = save[2 * R + j], sel = save[3 * R + j], it = save[4 * R + j];
  const float tc = save[5 * R + j];
  const float dhj = dh_a[j] + (dh_b ? dh_b[j] : 0.f);      // recurrent part + this step's output gradient
  const float dcn = (dc_in ? dc_in[j] : 0.f) + dhj * og * (1.f - tc * tc);
  ds[j] = dcn * it * ig * (1.f - ig);
  ds[R + j] = dcn * c_prev[j] * fg * (1.f - fg);
  ds[2 * R + j] = dhj * tc * og * (1.f - og);
  const float dit = dcn * ig;
  const float d0 = sel == 0.f ? dit : 0.f, d1 = sel == 0.f ? 0.f : dit;
  ds[3 * R + j] = d0; ds[4 * R + j] = d1;
  da2c[j] = d0; da2c[R + j] = d1;
  dc_prev[j] = dcn * fg;
}

// log_softmax + masked NLL over S rows (one workgroup per row)
__global__ __launch_bounds__(1024) void lsm_nll_kernel(const float* logits, const int64_t* target, const float* mask, int S, int V1, float gscale,
                                                      float* loss_slot, float* dlogits, float* logprobs) {
  __shared__ float red[16];
  const int srow = blockIdx.x, tid = threadIdx.x;
  const float* lr = logits + (long)srow * V1;
  float mx = -INFINITY;
  for (int v = tid; v < V1; v += blockDim.x) mx = fmaxf(mx, lr[v]);
  mx = block_max(mx, red);
  float se = 0.f;
  for (int v = tid; v < V1; v += blockDim.x) se += expf(lr[v] - mx);
  se = block_sum(se, red);
  const float lse = mx + logf(se);
  float msum = 0.f;
  for (int i = 0; i < S; ++i) msum += mask[i];
  const float mk = mask[srow];
  const int tg = (int)target[srow];
  for (int v = tid; v < V1; v += blockDim.x) {
    const float lp = lr[v] - lse;
    if (logprobs) logprobs[(long)srow * V1 + v] = lp;
    if (dlogits) dlogits[(long)srow * V1 + v] = gscale * mk / msum * (expf(lp) - (v == tg ? 1.f : 0.f));
  }
  if (tid == 0) atomicAdd(loss_slot, -(lr[tg] - lse) * mk / msum);
}


// ---- fused per-step kernels of the att2in2 recurrence (one launch each; the recurrence is a chain of dependent launches,
// so every launch removed is a few microseconds off the caption branch's critical path) ----

// two GEMVs over the same input: y1 = w1 x + b1 (blocks [0, nb1)),  y2 (+)= w2 x + b2 (the rest).  One wave per output.
__global__ __launch_bounds__(256) void linear2_fwd_kernel(const float* __restrict__ x, int K, const float* __restrict__ w1, const float* __restrict__ b1,
                                                         float* y1, int N1, int acc1, const float* __restrict__ w2, const float* __restrict__ b2,
                                                         float* y2, int N2, int acc2, int nb1) {
  __builtin_amdgcn_s_setprio(3);   // a link of the caption branch's dependent chain (the step's critical path): issue ahead of co-resident GEMM waves
  const bool second = (int)blockIdx.x >= nb1;
  const int n = ((int)blockIdx.x - (second ? nb1 : 0)) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int N = second ? N2 : N1;
  if (n >= N) return;
  const float* wr = (second ? w2 : w1) + (long)n * K;
  float acc = 0.f;
  for (int k = lane * 4; k < K; k += 256) {
    const float4 wv = *(const float4*)(wr + k), xv = *(const float4*)(x + k);
    acc = fmaf(wv.x, xv.x, fmaf(wv.y, xv.y, fmaf(wv.z, xv.z, fmaf(wv.w, xv.w, acc))));
  }
  acc = wave_sum(acc);
  if (lane == 0) {
    const float* b = second ? b2 : b1;
    float* o = (second ? y2 : y1) + n;
    float v = acc + (b ? b[n] : 0.f);
    if (second ? acc2 : acc1) v += *o;
    *o = v;
  }
}
// y[n] (+)= w1[n] . x1 + w2[n] . x2   (two inputs, one output vector; one wave per n)
__global__ __launch_bounds__(256) void linear_sum2_kernel(const float* __restrict__ x1, const float* __restrict__ w1, int K1, const float* __restrict__ x2,
                                                         const float* __restrict__ w2, int K2, float* y, int N, int accumulate) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (n >= N) return;
  float a1 = 0.f, a2 = 0.f;
  const float* r1 = w1 + (long)n * K1; const float* r2 = w2 + (long)n * K2;
  for (int k = lane * 4; k < K1; k += 256) {
    const float4 wv = *(const float4*)(r1 + k), xv = *(const float4*)(x1 + k);
    a1 = fmaf(wv.x, xv.x, fmaf(wv.y, xv.y, fmaf(wv.z, xv.z, fmaf(wv.w, xv.w, a1))));
  }
  for (int k = lane * 4; k < K2; k += 256) {
    const float4 wv = *(const float4*)(r2 + k), xv = *(const float4*)(x2 + k);
    a2 = fmaf(wv.x, xv.x, fmaf(wv.y, xv.y, fmaf(wv.z, xv.z, fmaf(wv.w, xv.w, a2))));
  }
  a1 = wave_sum(a1); a2 = wave_sum(a2);
  if (lane == 0) { float v = a1 + a2; if (accumulate) v += y[n]; y[n] = v; }
}
// the same with the four waves of a workgroup splitting the K range of ONE output (four times the workgroups, a quarter of the
// dependent loads per wave): the launch sits on the caption branch's backward chain once per token
__global__ __launch_bounds__(256) void linear_sum2_split_kernel(const float* __restrict__ x1, const float* __restrict__ w1, int K1, const float* __restrict__ x2,
                                                               const float* __restrict__ w2, int K2, float* y, int N, int accumulate) {
  __builtin_amdgcn_s_setprio(3);   // a link of the caption branch's dependent chain (the step's critical path): issue ahead of co-resident GEMM waves
  __shared__ float part[4];
  const int n = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float* r1 = w1 + (long)n * K1; const float* r2 = w2 + (long)n * K2;
  const int q1 = ((K1 / 4 + 3) / 4) * 4, q2 = ((K2 / 4 + 3) / 4) * 4;       // per-wave K ranges, multiples of 4
  float a = 0.f;
  for (int k = wv * q1 + lane * 4; k < min(K1, (wv + 1) * q1); k += 256) {
    const float4 wv4 = *(const float4*)(r1 + k), xv = *(const float4*)(x1 + k);
    a = fmaf(wv4.x, xv.x, fmaf(wv4.y, xv.y, fmaf(wv4.z, xv.z, fmaf(wv4.w, xv.w, a))));
  }
  for (int k = wv * q2 + lane * 4; k < min(K2, (wv + 1) * q2); k += 256) {
    const float4 wv4 = *(const float4*)(r2 + k), xv = *(const float4*)(x2 + k);
    a = fmaf(wv4.x, xv.x, fmaf(wv4.y, xv.y, fmaf(wv4.z, xv.z, fmaf(wv4.w, xv.w, a))));
  }
  a = wave_sum(a);
  if (lane == 0) part[wv] = a;
  __syncthreads();
  if (threadIdx.x == 0) { float v = ((part[0] + part[1]) + part[2]) + part[3]; if (accumulate) v += y[n]; y[n] = v; }
}
// a2c Linear (rows j and R + j) fused with the gate nonlinearity of unit j (ATT:449-462): one wave per unit
__global__ __launch_bounds__(256) void cap_a2c_gates_kernel(const float* __restrict__ ares, const float* __restrict__ w, const float* __restrict__ b, int K,
                                                           const float* __restrict__ s, const float* __restrict__ c_prev, float* c, float* h,
                                                           float* save, int R) {
  const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (j >= R) return;
  const float* r0 = w + (long)j * K; const float* r1 = w + (long)(R + j) * K;
  float a0 = 0.f, a1 = 0.f;
  for (int k = lane * 4; k < K; k += 256) {
    const float4 xv = *(const float4*)(ares + k), w0 = *(const float4*)(r0 + k), w1 = *(const float4*)(r1 + k);
    a0 = fmaf(w0.x, xv.x, fmaf(w0.y, xv.y, fmaf(w0.z, xv.z, fmaf(w0.w, xv.w, a0))));
    a1 = fmaf(w1.x, xv.x, fmaf(w1.y, xv.y, fmaf(w1.z, xv.z, fmaf(w1.w, xv.w, a1))));
  }
  a0 = wave_sum(a0); a1 = wave_sum(a1);
  if (lane == 0) {
    a0 += b[j]; a1 += b[R + j];
    const float ig = sigm(s[j]), fg = sigm(s[R + j]), og = sigm(s[2 * R + j]);
    const float t0 = s[3 * R + j] + a0, t1 = s[4 * R + j] + a1;
    const float it = fmaxf(t0, t1);
    const float cn = fg * c_prev[j] + ig * it;
    const float tc = tanhf(cn);
    c[j] = cn; h[j] = og * tc;
    save[j] = ig; save[R + j] = fg; save[2 * R + j] = og; save[3 * R + j] = (t0 >= t1) ? 0.f : 1.f;
    save[4 * R + j] = it; save[5 * R + j] = tc;
  }
}
// attention backward, the part the recurrence needs at step t: ddot[l] (softmax backward of dweight[l] = dres . att[l],
// recomputed by every workgroup) and datt_h[d] = sum_l ddot[l] aw[d] (1 - tanh^2).  Workgroup = 16 channels x 16 l-groups.
__global__ __launch_bounds__(1024) void cap_att_bwd_step_kernel(const float* __restrict__ dres, const float* __restrict__ att, const float* __restrict__ tanh_ws,
                                                               const float* __restrict__ weight, const float* __restrict__ aw, int L, int D,
                                                               float* ddot_out, float* datt_h) {
  __shared__ float dwl[256];
  __shared__ float ddot[256];
  __shared__ float red[16];
  __shared__ float part[64][17];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // dweight[l] = dres . att[l]: 16 waves x (L / 16) rows, ONE row per trip.  (Round 2: the two-rows-per-trip form -- two interleaved
  // shuffle reductions whose results lane 0 stored with two LDS writes -- sporadically produced a wrong SECOND sum when other kernels
  // shared the CU: the compiler consumes the last ds_bpermute of the second chain after `s_waitcnt lgkmcnt(1)` with the first
  // ds_write already issued, i.e. it relies on a DS permute and a younger DS write retiring in order.  Found by the bit-reproducibility
  // test; tools/scan_lgkm_order.py looks for the pattern in the generated ISA.)
  for (int l = wv; l < L; l += 16) {
    float s0 = 0.f;
    for (int d = lane * 4; d < D; d += 256) {
      const float4 r = *(const float4*)(dres + d);
      const float4 a = *(const float4*)(att + (long)l * D + d);
      s0 = fmaf(a.x, r.x, fmaf(a.y, r.y, fmaf(a.z, r.z, fmaf(a.w, r.w, s0))));
    }
    s0 = wave_sum(s0);
    if (lane == 0) dwl[l] = s0;
  }
  __syncthreads();
  const float wl = tid < L ? weight[tid] : 0.f;
  const float dw = tid < L ? dwl[tid] : 0.f;
  const float dot = block_sum(wl * dw, red);
  const float dd = wl * (dw - dot);
  if (tid < 256) ddot[tid] = tid < L ? dd : 0.f;
  if (blockIdx.x == 0 && tid < L) ddot_out[tid] = dd;
  __syncthreads();
  const int dl = tid & 15, lg = tid >> 4;            // 16 channels x 64 l-groups
  const int d = blockIdx.x * 16 + dl;
  float sah = 0.f;
  if (d < D) {
    const float a = aw[d];
    for (int l = lg; l < L; l += 64) {
      const float t = tanh_ws[(long)l * D + d];
      sah = fmaf(ddot[l] * a, 1.f - t * t, sah);
    }
  }
  part[lg][dl] = sah;
  __syncthreads();
  if (lg == 0 && d < D) {
    float v = 0.f;
    for (int g = 0; g < 64; ++g) v += part[g][dl];
    datt_h[d] = v;
  }
}
// ---------------------------------------------------------------------------------------------------------------------------------
// Projected-attention form of the att2in2 step (round 2).  a2c(att_res) = W (sum_l w_l att_l) + b = sum_l w_l (W att_l) + b because the
// softmax weights sum to one: with P = att . W_a2c^T computed once per sentence ([L][2R], one MFMA GEMM), the per-token chain
//     h2h/h2att GEMVs -> attention dots -> softmax + weighted sum of att -> a2c GEMV + gates          (4 launches, 2 MB of W_a2c per token)
// becomes
//     h2h/h2att GEMVs -> attention dots -> softmax + weighted sum of P's columns + gates               (3 launches),
// and backward  gates -> W_a2c^T GEMV -> attention step -> W_h2h^T / W_h2att^T GEMVs  becomes
//     gates + d(weight)_l = P_l . d(a2c)  ->  attention step  ->  GEMVs                                 (3 launches).
// The step-summed d(P) = sum_t w_t (x) d(a2c)_t is one small GEMM after the loop; d(att) and d(W_a2c) follow from it as two more.
// The caption branch is the train step's critical path, and each dependent launch on it costs ~10 us.
//
// forward: workgroup = 16 units x 16 location groups; softmax over the L dots recomputed per workgroup (L <= 256)
__global__ __launch_bounds__(256) void cap_apply_gates_kernel(const float* __restrict__ P, const float* __restrict__ dots, const float* __restrict__ b,
                                                             const float* __restrict__ s, const float* __restrict__ c_prev, float* c, float* h,
                                                             float* save, float* weight, int L, int R) {
  __builtin_amdgcn_s_setprio(3);   // a link of the caption branch's dependent chain (the step's critical path): issue ahead of co-resident GEMM waves
  __shared__ float w[256];
  __shared__ float red[4];
  __shared__ float p0[16][17], p1[16][17];
  const int tid = threadIdx.x, dl = tid & 15, lg = tid >> 4;
  const float v = tid < L ? dots[tid] : -INFINITY;
  const float mx = block_max(v, red);
  const float e = tid < L ? expf(v - mx) : 0.f;
  const float sum = block_sum(e, red);
  w[tid] = tid < L ? e / sum : 0.f;
  if (blockIdx.x == 0 && tid < L) weight[tid] = e / sum;
  __syncthreads();
  const int j = blockIdx.x * 16 + dl;
  float a0 = 0.f, a1 = 0.f;
  if (j < R) {
    const float* q = P + j;
    for (int l = lg; l < L; l += 16) {
      const float wl = w[l];
      a0 = fmaf(wl, q[(long)l * 2 * R], a0);
      a1 = fmaf(wl, q[(long)l * 2 * R + R], a1);
    }
  }
  p0[lg][dl] = a0; p1[lg][dl] = a1;
  __syncthreads();
  if (lg == 0 && j < R) {
    a0 = b[j]; a1 = b[R + j];
#pragma unroll
    for (int g = 0; g < 16; ++g) { a0 += p0[g][dl]; a1 += p1[g][dl]; }
    const float ig = sigm(s[j]), fg = sigm(s[R + j]), og = sigm(s[2 * R + j]);
    const float t0 = s[3 * R + j] + a0, t1 = s[4 * R + j] + a1;
    const float it = fmaxf(t0, t1);
    const float cn = fg * c_prev[j] + ig * it;
    const float tc = tanhf(cn);
    c[j] = cn; h[j] = og * tc;
    save[j] = ig; save[R + j] = fg; save[2 * R + j] = og; save[3 * R + j] = (t0 >= t1) ? 0.f : 1.f;
    save[4 * R + j] = it; save[5 * R + j] = tc;
  }
}
// backward (1): the gate backward of cap_gates_bwd_kernel recomputed by every workgroup into LDS (2R values; workgroup 0 also stores
// dsums / da2c / dc_prev), then one wave per location: dweight[l] = P[l] . da2c.   R <= 1024.
__global__ __launch_bounds__(256) void cap_gates_bwd_dw_kernel(const float* __restrict__ dh_a, const float* __restrict__ dh_b, const float* __restrict__ dc_in,
                                                              const float* __restrict__ save, const float* __restrict__ c_prev, const float* __restrict__ P,
                                                              float* ds, float* da2c, float* dc_prev, float* dweight, int L, int R) {
  __builtin_amdgcn_s_setprio(3);   // a link of the caption branch's dependent chain (the step's critical path): issue ahead of co-resident GEMM waves
  __shared__ __attribute__((aligned(16))) float g[2048];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const bool out = blockIdx.x == 0;
  for (int j = tid; j < R; j += 256) {
    const float ig = save[j], fg = save[R + j], og = save[2 * R + j], sel = save[3 * R + j], it = save[4 * R + j];
    const float tc = save[5 * R + j];
    const float dhj = dh_a[j] + (dh_b ? dh_b[j] : 0.f);
    const float dcn = (dc_in ? dc_in[j] : 0.f) + dhj * og * (1.f - tc * tc);
    const float dit = dcn * ig;
    const float d0 = sel == 0.f ? dit : 0.f, d1 = sel == 0.f ? 0.f : dit;
    g[j] = d0; g[R + j] = d1;
    if (out) {
      ds[j] = dcn * it * ig * (1.f - ig);
      ds[R + j] = dcn * c_prev[j] * fg * (1.f - fg);
      ds[2 * R + j] = dhj * tc * og * (1.f - og);
      ds[3 * R + j] = d0; ds[4 * R + j] = d1;
      da2c[j] = d0; da2c[R + j] = d1;
      dc_prev[j] = dcn * fg;
    }
  }
  __syncthreads();
  const int l = blockIdx.x * 4 + wv;
  if (l >= L) return;
  const float* q = P + (long)l * 2 * R;
  float a = 0.f;
  for (int n = lane * 4; n < 2 * R; n += 256) {
    const float4 pv = *(const float4*)(q + n), gv = *(const float4*)(g + n);
    a = fmaf(pv.x, gv.x, fmaf(pv.y, gv.y, fmaf(pv.z, gv.z, fmaf(pv.w, gv.w, a))));
  }
  a = wave_sum(a);
  if (lane == 0) dweight[l] = a;
}
// backward (2): softmax backward of the L location weights (recomputed by every workgroup from dweight) and
// datt_h[d] = sum_l ddot[l] aw[d] (1 - tanh^2).  Workgroup = 16 channels x 64 location groups.
__global__ __launch_bounds__(1024) void cap_att_bwd_step2_kernel(const float* __restrict__ dweight, const float* __restrict__ tanh_ws, const float* __restrict__ weight,
                                                                const float* __restrict__ aw, int L, int D, float* ddot_out, float* datt_h) {
  __builtin_amdgcn_s_setprio(3);   // a link of the caption branch's dependent chain (the step's critical path): issue ahead of co-resident GEMM waves
  __shared__ float ddot[256];
  __shared__ float red[16];
  __shared__ float part[64][17];
  const int tid = threadIdx.x;
  const float wl = tid < L ? weight[tid] : 0.f;
  const float dw = tid < L ? dweight[tid] : 0.f;
  const float dot = block_sum(wl * dw, red);
  const float dd = wl * (dw - dot);
  if (tid < 256) ddot[tid] = tid < L ? dd : 0.f;
  if (blockIdx.x == 0 && tid < L) ddot_out[tid] = dd;
  __syncthreads();
  const int dl = tid & 15, lg = tid >> 4;
  const int d = blockIdx.x * 16 + dl;
  float sah = 0.f;
  if (d < D) {
    const float a = aw[d];
    for (int l = lg; l < L; l += 64) {
      const float t = tanh_ws[(long)l * D + d];
      sah = fmaf(ddot[l] * a, 1.f - t * t, sah);
    }
  }
  part[lg][dl] = sah;
  __syncthreads();
  if (lg == 0 && d < D) {
    float v = 0.f;
    for (int gi = 0; gi < 64; ++gi) v += part[gi][dl];
    datt_h[d] = v;
  }
}
// after the loop: everything the per-step kernel left out, summed over the S steps in one launch.  A workgroup owns 16 channels d
// (64 location lanes x 16 channels; round 3: 1024 threads instead of 256 - the launch sits on the caption branch's critical chain and each
// thread's (location, step) loop is a chain of dependent loads: 65-89 -> ~25 us): dpatt / datt rows are written per (l, d), the alpha_net
// weight gradient daw[d] is summed over all locations inside the workgroup in a fixed order (no atomics); workgroup 0 also takes the bias
// gradient.
__global__ __launch_bounds__(1024) void cap_att_bwd_batched_kernel(const float* __restrict__ ddot, const float* __restrict__ weight, const float* __restrict__ dres,
                                                                  int ldr, const float* __restrict__ tanh_ws, const float* __restrict__ aw, int S, int L, int D,
                                                                  float* dpatt, float* datt, float* daw, float* dab) {
  __shared__ float part[64][16];
  __shared__ float red[16];
  const int tid = threadIdx.x, dl = tid & 15, ll = tid >> 4;
  const int d = blockIdx.x * 16 + dl;
  float dwsum = 0.f;
  if (d < D) {
    const float a = aw[d];
    for (int l = ll; l < L; l += 64) {
      float dp = 0.f, da = 0.f, dw = 0.f;
      for (int t = 0; t < S; ++t) {
        const float dd = ddot[(long)t * L + l], w = weight[(long)t * L + l];
        const float th = tanh_ws[((long)t * L + l) * D + d];
        dp = fmaf(dd * a, 1.f - th * th, dp);
        if (dres) da = fmaf(w, dres[(long)t * ldr + d], da);
        dw = fmaf(dd, th, dw);
      }
      dpatt[(long)l * D + d] += dp;
      if (dres) datt[(long)l * D + d] += da;
      dwsum += dw;
    }
  }
  part[ll][dl] = dwsum;
  __syncthreads();
  if (ll == 0 && d < D) {
    float v = 0.f;
    for (int g = 0; g < 64; ++g) v += part[g][dl];
    daw[d] += v;
  }
  if (blockIdx.x == 0) {
    float sb = 0.f;
    for (int e = tid; e < S * L; e += 1024) sb += ddot[e];
    sb = block_sum(sb, red);
    if (tid == 0) dab[0] += sb;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
__global__ void mul_inplace_kernel(float* dx, int lddx, const float* __restrict__ mul, int M, int K) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i < (long)M * K) { const int m = (int)(i / K), k = (int)(i - (long)m * K); dx[(long)m * lddx + k] *= mul[(long)m * lddx + k]; }
}
// Row-batch linears on the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32: a lane holds A[l%16][l/16] and B[l/16][l%16], D rows 4(l/16)+r).
// The wave-per-output kernels above re-read the x block per output and leave the 21 x 3350 x 512 logit layer at 40-200 us; these
// stream every weight exactly once with 16-byte loads, 4 k (or n) per lane per load: the MFMA's k index is a permutation of memory
// order, the same permutation on both operands.
//
// NT: y[m][n] = act(sum_k x[m][k] w[n][k] + b[n] (+ y)).  Workgroup = one 16-column tile x MT 16-row tiles; its 4 waves split K.
template <int MT>
__global__ __launch_bounds__(256) void linear_nt_mfma_kernel(const float* __restrict__ x, int ldx_, const float* __restrict__ w, int ldw,
                                                            const float* __restrict__ b, float* y, int ldy, int M, int N, int K,
                                                            int act, int accumulate) {
  __shared__ float red[4][MT][4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 15, kq = lane >> 4;
  const int n0 = blockIdx.x * 16, m0 = blockIdx.y * (16 * MT);
  const float* wr = w + (long)min(n0 + j, N - 1) * ldw;
  const float* xr[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) xr[t] = x + (long)min(m0 + 16 * t + j, M - 1) * ldx_;
  floatx4 acc[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) acc[t] = floatx4{0.f, 0.f, 0.f, 0.f};
  const int kper = ((K + 63) / 64) * 16;                  // K range of one wave, a multiple of 16
  const int kb = wv * kper, ke = min(K, kb + kper);
#pragma unroll 4
  for (int k = kb; k < ke; k += 16) {
    const int kk = k + 4 * kq;
    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), av[MT];
    const bool in = kk < ke;                              // K % 4 == 0: a lane's four k are all inside or all outside
    if (in) bv = *(const float4*)(wr + kk);
#pragma unroll
    for (int t = 0; t < MT; ++t) av[t] = in ? *(const float4*)(xr[t] + kk) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t].x, bv.x, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t].y, bv.y, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t].z, bv.z, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t].w, bv.w, acc[t], 0, 0, 0);
    }
  }
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) red[wv][t][r][lane] = acc[t][r];
  __syncthreads();
  // wave wv finishes register r = wv of every tile: row 4 (lane/16) + r, column lane % 16
  const int n = n0 + j;
  if (n < N) {
    const float bias = b ? b[n] : 0.f;
#pragma unroll
    for (int t = 0; t < MT; ++t) {
      const int m = m0 + 16 * t + 4 * kq + wv;
      if (m < M) {
        float v = ((red[0][t][wv][lane] + red[1][t][wv][lane]) + red[2][t][wv][lane]) + red[3][t][wv][lane] + bias;
        float* o = y + (long)m * ldy + n;
        if (accumulate) v += *o;
        *o = act_apply(v, act);
      }
    }
  }
}

// NN: dx[m][k] (+)= (sum_n dy[m][n] w[n][k]) (* mul[m][k]) from the weight as stored (no transposed copy).  A wave owns 64 output columns
// as four interleaved 16-column tiles (column k0 + 4 (lane%16) + q belongs to tile q: one 16-byte load of a weight row feeds four
// MFMAs), MT 16-row tiles, and a share of the contraction range; workgroup = 4 waves (LDS sum), grid.y = further split of n whose
// partial sums go to `ws` and are added in order by linear_nn_reduce_kernel (no atomics).
template <int MT>
__global__ __launch_bounds__(256) void linear_nn_mfma_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ w, float* dx,
                                                            int lddx, int M, int N, int K, int accumulate, const float* __restrict__ mul,
                                                            float* ws, int nper) {
  __shared__ float red[4][MT * 4][4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 15, kq = lane >> 4;
  const int k0 = blockIdx.x * 64, m0 = blockIdx.z * (16 * MT);
  const int nb0 = blockIdx.y * nper, ne0 = min(N, nb0 + nper);
  const int wper = ((ne0 - nb0 + 63) / 64) * 16;          // n range of one wave, a multiple of 16
  const int nb = nb0 + wv * wper, ne = min(ne0, nb + wper);
  const int kc = min(k0 + 4 * j, K - 4);                  // K % 4 == 0; columns past K are computed on a clamped address and not stored
  const float* dyr[MT]; bool rv[MT];
#pragma unroll
  for (int t = 0; t < MT; ++t) { const int m = m0 + 16 * t + j; rv[t] = m < M; dyr[t] = dy + (long)min(m, M - 1) * lddy; }
  floatx4 acc[MT][4];
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[t][q] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
  for (int n = nb; n < ne; n += 16) {
    float4 bv[4]; float av[MT][4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int nn = n + 4 * kq + e;
      bv[e] = *(const float4*)(w + (long)min(nn, N - 1) * K + kc);
#pragma unroll
      for (int t = 0; t < MT; ++t) av[t][e] = (nn < ne && rv[t]) ? dyr[t][nn] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int t = 0; t < MT; ++t) {
        acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t][e], bv[e].x, acc[t][0], 0, 0, 0);
        acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t][e], bv[e].y, acc[t][1], 0, 0, 0);
        acc[t][2] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t][e], bv[e].z, acc[t][2], 0, 0, 0);
        acc[t][3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[t][e], bv[e].w, acc[t][3], 0, 0, 0);
      }
  }
#pragma unroll
  for (int t = 0; t < MT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) red[wv][t * 4 + q][r][lane] = acc[t][q][r];
  __syncthreads();
  // wave wv finishes the (tile, column set) pairs p = wv, wv + 4, ...: row 4 (lane/16) + r, column k0 + 4 (lane%16) + q
  const int Mpad = gridDim.z * 16 * MT;
  for (int p = wv; p < MT * 4; p += 4) {
    const int t = p >> 2, q = p & 3;
    const int kcol = k0 + 4 * j + q;
    if (kcol >= K) continue;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + 16 * t + 4 * kq + r;
      if (m >= M) continue;
      float v = ((red[0][p][r][lane] + red[1][p][r][lane]) + red[2][p][r][lane]) + red[3][p][r][lane];
      if (gridDim.y > 1) { ws[((long)blockIdx.y * Mpad + m) * K + kcol] = v; continue; }
      if (mul) v *= mul[(long)m * lddx + kcol];
      float* o = dx + (long)m * lddx + kcol;
      *o = accumulate ? *o + v : v;
    }
  }
}
__global__ void linear_nn_reduce_kernel(const float* __restrict__ ws, int nsplit, int Mpad, float* dx, int lddx, int M, int K, int accumulate,
                                        const float* __restrict__ mul) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= (long)M * K) return;
  const int m = (int)(i / K), k = (int)(i - (long)m * K);
  float v = 0.f;
  for (int sp = 0; sp < nsplit; ++sp) v += ws[((long)sp * Mpad + m) * K + k];
  if (mul) v *= mul[(long)m * lddx + k];
  float* o = dx + (long)m * lddx + k;
  *o = accumulate ? *o + v : v;
}
// TN: dw[n][k] += sum_m dy[m][n] x[m][k] (weight gradient of a row batch), db[n] += sum_m dy[m][n].  The contraction index m is the slow
// index of both operands, so a 16-byte load along n (dy) / along k (x) feeds four interleaved 16-wide tiles: a wave owns a 64 (n) x 64 (k)
// block of dw as 4 x 4 tiles and walks m in steps of 4 (two loads, 16 MFMAs).  Single owner per output: no atomics.  The
// one-thread-per-column kernel above spent 55 us per call on average (13 calls, 0.7 ms of a 6.4 ms step) on these shapes.
template <bool VEC_DY>
__global__ __launch_bounds__(256) void linear_tn_mfma_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx_, float* dw, float* db,
                                                            int M, int N, int K) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, j = lane & 15, kq = lane >> 4;
  const int n0 = (blockIdx.y * 4 + wv) * 64, k0 = blockIdx.x * 64;
  if (n0 >= N) return;
  const int nc = n0 + 4 * j, kc = min(k0 + 4 * j, K - 4);         // K % 4 == 0; clamped columns are computed and not stored
  floatx4 acc[4][4];
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
  for (int m0 = 0; m0 < M; m0 += 4) {
    const int m = m0 + kq;
    const bool mv = m < M;
    const float* dr = dy + (long)min(m, M - 1) * lddy;
    float av[4];
    if (VEC_DY) {
      const float4 t = (mv && nc + 3 < N) ? *(const float4*)(dr + nc) : make_float4(0.f, 0.f, 0.f, 0.f);
      av[0] = t.x; av[1] = t.y; av[2] = t.z; av[3] = t.w;
      if (mv && nc < N && nc + 3 >= N) {
#pragma unroll
        for (int q = 0; q < 4; ++q) av[q] = nc + q < N ? dr[nc + q] : 0.f;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) av[q] = (mv && nc + q < N) ? dr[nc + q] : 0.f;
    }
    const float4 bv = *(const float4*)(x + (long)min(m, M - 1) * ldx_ + kc);   // rows past M meet av == 0
    const float bb[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
    for (int a = 0; a < 4; ++a) cs[a] += av[a];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bb[b], acc[a][b], 0, 0, 0);
  }
  // tile (a, b): D row i <-> n = n0 + 4 i + a (i = 4 (lane / 16) + r), D column lane % 16 <-> k = k0 + 4 (lane % 16) + b: the four b make one float4
  const int kcol = k0 + 4 * j;
  if (kcol < K) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + 4 * (4 * kq + r) + a;
        if (n < N) {
          float4* o = (float4*)(dw + (long)n * K + kcol);
          float4 v = *o;
          v.x += acc[a][0][r]; v.y += acc[a][1][r]; v.z += acc[a][2][r]; v.w += acc[a][3][r];
          *o = v;
        }
      }
  }
  if (db && blockIdx.x == 0) {
    // column sums of dy from the values the lanes already hold: lane (j, kq) summed rows kq, kq + 4, ... of columns nc .. nc + 3
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float v = cs[q];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (kq == 0 && nc + q < N) db[nc + q] += v;
    }
  }
}
}  // namespace

extern "C" int l2s_linear_fwd(const float* x, int ldx_, const float* w, int ldw, const float* b, float* y, int ldy, int M, int N, int K, int act,
                              int accumulate, hipStream_t s) {
  if (M <= 0 || N <= 0) return L2S_OK;
  const bool v4 = !((K & 3) || (ldx_ & 3) || (ldw & 3) || ((uintptr_t)x & 15) || ((uintptr_t)w & 15));
  const bool v2 = !((K & 1) || (ldx_ & 1) || (ldw & 1) || ((uintptr_t)x & 7) || ((uintptr_t)w & 7));
  if (v4 && M >= 2) {
    // row batches: exact-fp32 MFMA tiles, every weight read once
    if (M <= 16) L2S_LAUNCH((linear_nt_mfma_kernel<1>), dim3(cdiv(N, 16), 1), dim3(256), 0, s, x, ldx_, w, ldw, b, y, ldy, M, N, K, act, accumulate);
    else L2S_LAUNCH((linear_nt_mfma_kernel<2>), dim3(cdiv(N, 16), cdiv(M, 32)), dim3(256), 0, s, x, ldx_, w, ldw, b, y, ldy, M, N, K, act, accumulate);
    return l2s_check_launch();
  }
  {
    const int m0 = 0;
    // one output per wave for a single row (GEMV: nothing to reuse), four outputs per wave for row batches
#define LF(MT, NW) do { dim3 grid(cdiv(N, 4 * NW), cdiv(M, MT)); \
                    if (v4) L2S_LAUNCH((linear_fwd_kernel<MT, 4, NW>), grid, dim3(256), 0, s, x, ldx_, w, ldw, b, y, ldy, m0, M, N, K, act, accumulate); \
                    else if (v2) L2S_LAUNCH((linear_fwd_kernel<MT, 2, NW>), grid, dim3(256), 0, s, x, ldx_, w, ldw, b, y, ldy, m0, M, N, K, act, accumulate); \
                    else L2S_LAUNCH((linear_fwd_kernel<MT, 1, NW>), grid, dim3(256), 0, s, x, ldx_, w, ldw, b, y, ldy, m0, M, N, K, act, accumulate); } while (0)
    // NW = 4 (x reused for four outputs) was measured slower inside the step (635 vs 441 us for the 11 row-batch launches: four times
    // fewer waves to hide the load latency on 512..3350 outputs), so every wave keeps one output
    if (M <= 1) LF(1, 1); else if (M <= 8) LF(8, 1); else if (M <= 16) LF(16, 1); else LF(MAXM, 1);
#undef LF
  }
  return l2s_check_launch();
}
// n-splits of the NN kernel for (M, N, K): enough workgroups to reach ~128, at least 64 n per workgroup
static int nn_split(int M, int N, int K) {
  const int tiles = cdiv(K, 64) * cdiv(M, 32);
  int sp = 128 / (tiles > 0 ? tiles : 1);
  const int maxsp = N / 64;
  if (sp > maxsp) sp = maxsp;
  if (sp > 32) sp = 32;
  return sp < 1 ? 1 : sp;
}
extern "C" long l2s_linear_bwd_x_ws_floats(int M, int N, int K) {
  const int sp = nn_split(M, N, K);
  return sp > 1 ? (long)sp * cdiv(M, 32) * 32 * K : 0;
}
extern "C" int l2s_linear_bwd_x(const float* dy, int lddy, const float* w, float* dx, int lddx, int M, int N, int K, int accumulate,
                                const float* mul, float* ws, long ws_floats, hipStream_t s) {
  if (M <= 0) return L2S_OK;
  // (a single row takes this path too when the matrix is large: it streams the weight once over ~128 workgroups instead of K / 64)
  if ((M >= 2 || (long)N * K >= (1L << 20)) && !(K & 3) && !((uintptr_t)w & 15) && K >= 4) {
    int sp = nn_split(M, N, K);
    const int mch = cdiv(M, 32), Mpad = mch * 32;
    if (!ws || ws_floats < (long)sp * Mpad * K) sp = 1;
    const int nper = cdiv(cdiv(N, sp), 16) * 16;
    sp = cdiv(N, nper);
    if (M <= 16 && sp == 1) L2S_LAUNCH((linear_nn_mfma_kernel<1>), dim3(cdiv(K, 64), 1, cdiv(M, 16)), dim3(256), 0, s, dy, lddy, w, dx, lddx, M, N, K, accumulate, mul, ws, nper);
    else L2S_LAUNCH((linear_nn_mfma_kernel<2>), dim3(cdiv(K, 64), sp, mch), dim3(256), 0, s, dy, lddy, w, dx, lddx, M, N, K, accumulate, mul, ws, nper);
    if (sp > 1) L2S_LAUNCH(linear_nn_reduce_kernel, dim3(cdiv((long)M * K, 256)), dim3(256), 0, s, (const float*)ws, sp, Mpad, dx, lddx, M, K, accumulate, mul);
    return l2s_check_launch();
  }
  dim3 grid(cdiv(K, 64));
  if (M == 1) L2S_LAUNCH(gemvT_kernel<1>, grid, dim3(1024), 0, s, dy, lddy, w, dx, lddx, 0, M, N, K, accumulate);
  else for (int m0 = 0; m0 < M; m0 += 8) L2S_LAUNCH(gemvT_kernel<8>, grid, dim3(1024), 0, s, dy, lddy, w, dx, lddx, m0, M, N, K, accumulate);
  if (mul) L2S_LAUNCH(mul_inplace_kernel, dim3(cdiv((long)M * K, 256)), dim3(256), 0, s, dx, lddx, mul, M, K);
  return l2s_check_launch();
}
extern "C" int l2s_linear_bwd_w(const float* dy, int lddy, const float* x, int ldx_, float* dw, float* db, int M, int N, int K, hipStream_t s) {
  if (M <= 0) return L2S_OK;
  if (M >= 2 && K >= 4 && !(K & 3) && !(ldx_ & 3) && !((uintptr_t)x & 15) && !((uintptr_t)dw & 15)) {
    const dim3 grid(cdiv(K, 64), cdiv(N, 256));
    if (!(lddy & 3) && !((uintptr_t)dy & 15)) L2S_LAUNCH((linear_tn_mfma_kernel<true>), grid, dim3(256), 0, s, dy, lddy, x, ldx_, dw, db, M, N, K);
    else L2S_LAUNCH((linear_tn_mfma_kernel<false>), grid, dim3(256), 0, s, dy, lddy, x, ldx_, dw, db, M, N, K);
    return l2s_check_launch();
  }
  L2S_LAUNCH(linear_bwd_w_kernel, dim3(cdiv(K, 256), N), dim3(256), 0, s, dy, lddy, x, ldx_, dw, db, M, N, K);
  return l2s_check_launch();
}
extern "C" int l2s_act_bwd(float* dy, const float* y, long n, int act, hipStream_t s) {
  L2S_LAUNCH(act_bwd_kernel, dim3(cdiv(n, 256) > 1024 ? 1024 : cdiv(n, 256)), dim3(256), 0, s, dy, y, n, act);
  return l2s_check_launch();
}
extern "C" int l2s_mask_relu_cast(float* x, const float* mul, const float* relu_ref, void* out, int out_dtype, long n, hipStream_t s) {
  if (!x || !relu_ref || !out || (out_dtype != L2S_BF16 && out_dtype != L2S_F32)) return L2S_EINVAL;
  const int g = cdiv(n, 256) > 1024 ? 1024 : (int)cdiv(n, 256);
  if (out_dtype == L2S_BF16) { L2S_LAUNCH(mask_relu_cast_kernel<bf16_t>, dim3(g), dim3(256), 0, s, x, mul, relu_ref, (bf16_t*)out, n); }
  else { L2S_LAUNCH(mask_relu_cast_kernel<float>, dim3(g), dim3(256), 0, s, x, mul, relu_ref, (float*)out, n); }
  return l2s_check_launch();
}
extern "C" int l2s_embed_fwd(const float* table, const int64_t* ids, const float* mask, float* out, int T, int D, int relu, hipStream_t s) {
  L2S_LAUNCH(embed_fwd_kernel, dim3(T), dim3(256), 0, s, table, ids, mask, out, T, D, relu);
  return l2s_check_launch();
}
extern "C" int l2s_embed_bwd(const float* dout, const float* out, const int64_t* ids, const float* mask, float* dtable, int T, int D, int relu, hipStream_t s) {
  L2S_LAUNCH(embed_bwd_kernel, dim3(T), dim3(256), 0, s, dout, out, ids, mask, dtable, T, D, relu);
  return l2s_check_launch();
}
extern "C" int l2s_lstm_cell_fwd(const float* gates, const float* c_prev, float* c, float* h, float* act, int Hh, hipStream_t s) {
  L2S_LAUNCH(lstm_cell_fwd_kernel, dim3(cdiv(Hh, 256)), dim3(256), 0, s, gates, c_prev, c, h, act, Hh);
  return l2s_check_launch();
}
extern "C" int l2s_lstm_cell_bwd(const float* dh, const float* dc_in, const float* act, const float* c_prev, const float* c,
                                 float* dgates, float* dc_prev, int Hh, hipStream_t s) {
  L2S_LAUNCH(lstm_cell_bwd_kernel, dim3(cdiv(Hh, 256)), dim3(256), 0, s, dh, dc_in, act, c_prev, c, dgates, dc_prev, Hh);
  return l2s_check_launch();
}
extern "C" int l2s_lstm_step_fwd(const l2s_lstm_fwd_dir* dirs, int ndir, int Hh, hipStream_t s) {
  if (!dirs || ndir < 1 || ndir > 2 || (Hh & 3)) return L2S_EINVAL;
  LstmDir a[2];
  for (int i = 0; i < 2; ++i) { const l2s_lstm_fwd_dir& q = dirs[i < ndir ? i : 0]; a[i] = LstmDir{q.w_hh, q.b_hh, q.gates_in, q.h_prev, q.c_prev, q.c, q.h, q.act, q.gates_out}; }
  L2S_LAUNCH(lstm_step_fwd_kernel, dim3(cdiv(Hh, 4), ndir), dim3(256), 0, s, a[0], a[1], Hh);
  return l2s_check_launch();
}
extern "C" int l2s_lstm_step_bwd(const l2s_lstm_bwd_dir* dirs, int ndir, int Hh, hipStream_t s) {
  if (!dirs || ndir < 1 || ndir > 2 || (Hh & 3)) return L2S_EINVAL;
  LstmBDir a[2];
  for (int i = 0; i < 2; ++i) { const l2s_lstm_bwd_dir& q = dirs[i < ndir ? i : 0]; a[i] = LstmBDir{q.w_hh_T, q.dgates_next, q.dh_ext, q.dc_in, q.act, q.c_prev, q.c, q.dgates, q.dc_prev}; }
  L2S_LAUNCH(lstm_step_bwd_kernel, dim3(cdiv(Hh, 4), ndir), dim3(256), 0, s, a[0], a[1], Hh);
  return l2s_check_launch();
}
extern "C" int l2s_dynfilter_fwd(const void* x, const float* filt, const float* r, void* y, float* resp, float* respk, int H, int W, int C,
                                 int dtype, int gate, hipStream_t s) {
  if (dtype == L2S_BF16 && !(C & 7) && C <= 2048 && !((uintptr_t)x & 15) && !((uintptr_t)y & 15) && !((uintptr_t)filt & 15)) {
    L2S_LAUNCH(dynfilter_fwd_bf16_kernel, dim3(cdiv(H * W, 4)), dim3(256), 0, s, (const bf16_t*)x, filt, r, (bf16_t*)y, resp, respk, H, W, C, gate);
    return l2s_check_launch();
  }
  L2S_LAUNCH(dynfilter_fwd_kernel, dim3(cdiv(H * W, 4)), dim3(256), 0, s, x, filt, r, y, resp, respk, H, W, C, dtype, gate);
  return l2s_check_launch();
}
constexpr int DYN_PCHUNK = 32;   // pixels per partial-sum chunk of pass 2
extern "C" long l2s_dynfilter_ws_floats(int H, int W, int C) { return (long)H * W + (long)cdiv(H * W, DYN_PCHUNK) * 7 * C; }
extern "C" int l2s_dynfilter_bwd_finish(const float* ws, const float* respk, float* dfilt, float* dr, int H, int W, int C, hipStream_t s) {
  const float* dresp_ws = ws; const float* part = ws + (long)H * W;
  L2S_LAUNCH(dynfilter_bwd3_kernel, dim3(cdiv(7 * C, 256)), dim3(256), 0, s, part, cdiv(H * W, DYN_PCHUNK), dresp_ws, respk, dfilt, dr, H * W, C);
  return l2s_check_launch();
}
extern "C" int l2s_dynfilter_bwd(const void* dy, const void* x, const float* filt, const float* r, const float* resp, const float* respk,
                                 void* dx, const void* relu_ref, float* dfilt, float* dr, float* ws, int H, int W, int C, int dtype,
                                 int gate, const float* dresp_extra, hipStream_t s) {
  // ws: [H*W] d(response) followed by [chunks][7][C] partial filter gradients (l2s_dynfilter_ws_floats)
  // dfilt == NULL: only passes 1 and 2 (dx); the caller runs l2s_dynfilter_bwd_finish on whichever stream needs dfilt / dr
  float* dresp_ws = ws; float* part = ws + (long)H * W;
  const int pchunk = DYN_PCHUNK, nchunk = cdiv(H * W, pchunk);
  const bool fast = dtype == L2S_BF16 && !(C & 7) && !((uintptr_t)dy & 15) && !((uintptr_t)x & 15) && !((uintptr_t)dx & 15) && !((uintptr_t)relu_ref & 15);
  if (fast) {
    L2S_LAUNCH(dynfilter_bwd1_bf16_kernel, dim3(cdiv(H * W, 4)), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, dresp_ws, H * W, C, gate, resp, dresp_extra);
    L2S_LAUNCH(dynfilter_bwd2_bf16_kernel, dim3(cdiv(C, 256), nchunk), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, filt, r, resp, (const float*)dresp_ws,
               (bf16_t*)dx, (const bf16_t*)relu_ref, part, H, W, C, pchunk, gate);
  } else {
    L2S_LAUNCH(dynfilter_bwd1_kernel, dim3(cdiv(H * W, 4)), dim3(256), 0, s, dy, x, respk, dresp_ws, dr, H * W, C, dtype, gate, resp, dresp_extra);
    L2S_LAUNCH(dynfilter_bwd2_kernel, dim3(cdiv(C, 64), nchunk), dim3(256), 0, s, dy, x, filt, r, resp, (const float*)dresp_ws, dx, relu_ref,
                       part, H, W, C, dtype, pchunk, gate);
  }
  if (!dfilt) return l2s_check_launch();
  L2S_LAUNCH(dynfilter_bwd3_kernel, dim3(cdiv(7 * C, 256)), dim3(256), 0, s, (const float*)part, nchunk, (const float*)dresp_ws, respk, dfilt, dr, H * W, C);
  return l2s_check_launch();
}
extern "C" int l2s_cap_attention_fwd(const float* patt, const float* att, const float* att_h, const float* aw, const float* ab, int L, int D,
                                     float* tanh_ws, float* weight, float* att_res, hipStream_t s) {
  if (L > 256) return L2S_EINVAL;
  // the softmax weights buffer doubles as the raw-dot scratch between the two launches
  L2S_LAUNCH(cap_att_dots_kernel, dim3(cdiv(L, 4)), dim3(256), 0, s, patt, att_h, aw, ab, L, D, tanh_ws, att_res + D);
  L2S_LAUNCH(cap_att_apply_kernel, dim3(cdiv(D, 64)), dim3(256), 0, s, att, att_res + D, L, D, weight, att_res);
  return l2s_check_launch();
}
extern "C" int l2s_cap_att_dots_fwd(const float* patt, const float* att_h, const float* aw, const float* ab, int L, int D, float* tanh_ws, float* dots,
                                    hipStream_t s) {
  if (L > 256) return L2S_EINVAL;
  L2S_LAUNCH(cap_att_dots_kernel, dim3(cdiv(L, 4)), dim3(256), 0, s, patt, att_h, aw, ab, L, D, tanh_ws, dots);
  return l2s_check_launch();
}
extern "C" int l2s_cap_apply_gates_fwd(const float* P, const float* dots, const float* b_a2c, const float* sums, const float* c_prev, float* c, float* h,
                                       float* save, float* weight, int L, int R, hipStream_t s) {
  if (L > 256) return L2S_EINVAL;
  L2S_LAUNCH(cap_apply_gates_kernel, dim3(cdiv(R, 16)), dim3(256), 0, s, P, dots, b_a2c, sums, c_prev, c, h, save, weight, L, R);
  return l2s_check_launch();
}
extern "C" int l2s_cap_gates_bwd_dw(const float* dh, const float* dh2, const float* dc_in, const float* save, const float* c_prev, const float* P,
                                    float* dsums, float* da2c, float* dc_prev, float* dweight, int L, int R, hipStream_t s) {
  if (R > 1024 || (R & 3) || ((uintptr_t)P & 15)) return L2S_EINVAL;
  L2S_LAUNCH(cap_gates_bwd_dw_kernel, dim3(cdiv(L, 4)), dim3(256), 0, s, dh, dh2, dc_in, save, c_prev, P, dsums, da2c, dc_prev, dweight, L, R);
  return l2s_check_launch();
}
extern "C" int l2s_cap_attention_bwd_step2(const float* dweight, const float* tanh_ws, const float* weight, const float* aw, int L, int D, float* ddot,
                                           float* datt_h, hipStream_t s) {
  if (L > 256) return L2S_EINVAL;
  L2S_LAUNCH(cap_att_bwd_step2_kernel, dim3(cdiv(D, 16)), dim3(1024), 0, s, dweight, tanh_ws, weight, aw, L, D, ddot, datt_h);
  return l2s_check_launch();
}
extern "C" int l2s_cap_attention_bwd(const float* datt_res, const float* att, const float* tanh_ws, const float* weight, const float* aw, int L, int D,
                                     float* dpatt, float* datt, float* datt_h, float* daw, float* dab, hipStream_t s) {
  if (L > 256) return L2S_EINVAL;
  L2S_LAUNCH(cap_att_bwd_dw_kernel, dim3(cdiv(L, 4)), dim3(256), 0, s, datt_res, att, L, D, datt_h + D);
  L2S_LAUNCH(cap_att_bwd_kernel, dim3(cdiv(D, 64)), dim3(256), 0, s, datt_res, datt_h + D, tanh_ws, weight, aw, L, D, dpatt, datt, datt_h, daw, dab);
  return l2s_check_launch();
}
extern "C" int l2s_linear2_fwd(const float* x, int K, const float* w1, const float* b1, float* y1, int N1, int acc1, const float* w2,
                               const float* b2, float* y2, int N2, int acc2, hipStream_t s) {
  if ((K & 3) || ((uintptr_t)x & 15) || ((uintptr_t)w1 & 15) || ((uintptr_t)w2 & 15)) return L2S_EINVAL;
  const int nb1 = cdiv(N1, 4);
  L2S_LAUNCH(linear2_fwd_kernel, dim3(nb1 + cdiv(N2, 4)), dim3(256), 0, s, x, K, w1, b1, y1, N1, acc1, w2, b2, y2, N2, acc2, nb1);
  return l2s_check_launch();
}
extern "C" int l2s_linear_sum2_fwd(const float* x1, const float* w1, int K1, const float* x2, const float* w2, int K2, float* y, int N,
                                   int accumulate, hipStream_t s) {
  if ((K1 & 3) || (K2 & 3) || ((uintptr_t)x1 & 15) || ((uintptr_t)x2 & 15) || ((uintptr_t)w1 & 15) || ((uintptr_t)w2 & 15)) return L2S_EINVAL;
  if (N <= 4096) L2S_LAUNCH(linear_sum2_split_kernel, dim3(N), dim3(256), 0, s, x1, w1, K1, x2, w2, K2, y, N, accumulate);
  else L2S_LAUNCH(linear_sum2_kernel, dim3(cdiv(N, 4)), dim3(256), 0, s, x1, w1, K1, x2, w2, K2, y, N, accumulate);
  return l2s_check_launch();
}
extern "C" int l2s_cap_a2c_gates_fwd(const float* att_res, const float* w_a2c, const float* b_a2c, int K, const float* sums, const float* c_prev,
                                     float* c, float* h, float* save, int R, hipStream_t s) {
  if ((K & 3) || ((uintptr_t)att_res & 15) || ((uintptr_t)w_a2c & 15)) return L2S_EINVAL;
  L2S_LAUNCH(cap_a2c_gates_kernel, dim3(cdiv(R, 4)), dim3(256), 0, s, att_res, w_a2c, b_a2c, K, sums, c_prev, c, h, save, R);
  return l2s_check_launch();
}
extern "C" int l2s_cap_attention_bwd_step(const float* datt_res, const float* att, const float* tanh_ws, const float* weight, const float* aw, int L,
                                          int D, float* ddot, float* datt_h, hipStream_t s) {
  if (L > 256 || (D & 3)) return L2S_EINVAL;
  L2S_LAUNCH(cap_att_bwd_step_kernel, dim3(cdiv(D, 16)), dim3(1024), 0, s, datt_res, att, tanh_ws, weight, aw, L, D, ddot, datt_h);
  return l2s_check_launch();
}
extern "C" int l2s_cap_attention_bwd_batched(const float* ddot, const float* weight, const float* datt_res, int ldr, const float* tanh_ws,
                                             const float* aw, int S, int L, int D, float* dpatt, float* datt, float* daw, float* dab, hipStream_t s) {
  L2S_LAUNCH(cap_att_bwd_batched_kernel, dim3(cdiv(D, 16)), dim3(1024), 0, s, ddot, weight, datt_res, ldr, tanh_ws, aw, S, L, D, dpatt, datt, daw, dab);
  return l2s_check_launch();
}
extern "C" int l2s_cap_gates_fwd(const float* sums, const float* a2c, const float* c_prev, float* c, float* h, float* save, int R, hipStream_t s) {
  L2S_LAUNCH(cap_gates_fwd_kernel, dim3(cdiv(R, 256)), dim3(256), 0, s, sums, a2c, c_prev, c, h, save, R);
  return l2s_check_launch();
}
extern "C" int l2s_cap_gates_bwd(const float* dh, const float* dh2, const float* dc_in, const float* save, const float* c_prev, float* dsums, float* da2c,
                                 float* dc_prev, int R, hipStream_t s) {
  L2S_LAUNCH(cap_gates_bwd_kernel, dim3(cdiv(R, 256)), dim3(256), 0, s, dh, dh2, dc_in, save, c_prev, dsums, da2c, dc_prev, R);
  return l2s_check_launch();
}
extern "C" int l2s_logsoftmax_nll(const float* logits, const int64_t* target, const float* mask, int S, int V1, float gscale, float* loss_slot,
                                  float* dlogits, float* logprobs_opt, hipStream_t s) {
  L2S_LAUNCH(lsm_nll_kernel, dim3(S), dim3(1024), 0, s, logits, target, mask, S, V1, gscale, loss_slot, dlogits, logprobs_opt);
  return l2s_check_launch();
}

