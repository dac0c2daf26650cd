/* lang2seg_hip.h — C ABI of liblang2seg_hip.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the hot path of wenz116/lang2seg: every arithmetic op of
 * Network.train_step (pyutils/mask-faster-rcnn/lib/nets/network_cycle_res5_2.py:702-719) that the
 * reference delegates to cuDNN / THC / its own torch.utils.ffi extensions is one entry point here.
 * Conventions (mirroring the reference FFI in lib/nms/src/nms_cuda.c:17 and
 * lib/layer_utils/roi_pooling/src/roi_pooling_cuda.c:7,49, but re-entrant and stream-explicit):
 *   - plain pointers (DEVICE memory unless stated) + explicit sizes, no torch types;
 *   - caller-owned outputs and workspaces, no hidden allocation, no host sync, graph-capturable;
 *   - every call enqueues on `stream` and returns 0 (L2S_OK) or a nonzero error code, never aborts;
 *   - `dtype` selects the activation storage type: L2S_F32 (verification mode, exact-f32 MFMA) or
 *     L2S_BF16 (bf16 storage, fp32 accumulate).  Parameters, gradients, losses, box math: fp32.
 *   - activations are NHWC ([pixel][channel]); conv weights are [Cout][KH][KW][Cin].
 */
#ifndef LANG2SEG_HIP_H
#define LANG2SEG_HIP_H
#include <stdint.h>
#ifdef __HIPCC__
#include <hip/hip_runtime.h>
#else
typedef struct ihipStream_t* hipStream_t;
#endif
#ifdef __cplusplus
extern "C" {
#endif

#define L2S_F32 0
#define L2S_BF16 1

int l2s_version(void);

/* ---------------------------------------------------------------- convolution / GEMM ------- */
/* replaces nn.Conv2d / F.conv2d / nn.Linear / nn.ConvTranspose2d(k2,s2) forward and data-gradient:
 * resnet_v1_cycle_res5_2.py:83-88,121,147,324-335; network_cycle_res5_2.py:236-251,279-301. */
#define L2S_CONV_RELU 1        /* y = max(y, 0) after bias/add */
#define L2S_CONV_OUT_F32 4     /* y is float regardless of dtype */
#define L2S_CONV_DECONV2X2 8   /* Cout = 4*Cq, n = (dy*2+dx)*Cq + co; pixel-shuffled 2x upsampled output */
#define L2S_CONV_SCATTER 16    /* output pixel (n, oy*out_stride, ox*out_stride) in an out_h x out_w grid */
#define L2S_ALGO_AUTO 0
#define L2S_ALGO_STAGED 1      /* register-staged tiles (ring / wave-specialised / software-pipelined kernels) */
#define L2S_ALGO_DMA 2         /* 256x128 tile, LDS-DMA fill (buffer_load ... lds), two wave groups alternating load / multiply */
#define L2S_ALGO_DMA_STAMPED 4 /* the same kernel with in-kernel clock stamps written to `ws` (tools/dma_stamps.py) */
#define L2S_ALGO_KSPLIT 5      /* 64x64 tile, LDS-DMA fill by four requester waves, four multiplier waves splitting each slice's K */
#define L2S_ALGO_KSPLIT_D3 6   /* the same with a ring of three LDS stages instead of four */
#define L2S_ALGO_PATCH 3       /* 3x3 / stride 1 / pad 1 on one map: 128 pixels x 32 or 64 channels per workgroup, input patch staged once for the nine taps */
#define L2S_ALGO_PDMA 7           /* the LDS-DMA tile as ONE persistent workgroup per CU walking its tiles, requests running ahead across tile boundaries (plain GEMMs) */
#define L2S_ALGO_DMA196 10         /* igemm_dma196_kernel: the 256 x 128 LDS-DMA pipeline on tiles of 196 rows (on request: 2-4 % faster alone, 1 % slower inside the step) */
#define L2S_ALGO_DMA256 9          /* igemm_dma256_kernel: 256x256 LDS-DMA tile for wide plain GEMMs (chosen automatically for them) */
#define L2S_ALGO_WS64_STAMPED 8  /* the 64x64 wave-specialised tile with clock stamps per workgroup in ws (tools/ws64_stamps.py) */
typedef struct {
  const void* x;      /* [n_img, IH, IW, ldx] activations (dtype) */
  const void* w;      /* [Cout][KH*KW*Cin] (dtype) */
  void* y;            /* [rows][ldy] (dtype, or float with OUT_F32) */
  const float* bias;  /* [Cout] or NULL (indexed by co for DECONV2X2) */
  const void* add;    /* [rows][ldadd] (dtype) added before relu/mask, or NULL */
  const void* ref;    /* [rows][ldref] (dtype): y = ref > 0 ? y : 0 (ReLU backward), or NULL */
  int n_img, IH, IW, Cin, OH, OW, Cout, KH, KW, stride, pad;
  int ldx, ldy, ldadd, ldref;
  int flags;
  int out_h, out_w, out_stride; /* SCATTER */
  int tile;                     /* 0 = auto, 64, 128, 224 or 256 (rows of the workgroup tile) */
  int split_k;                  /* K split over workgroups, partial tiles added in a fixed order by a second launch: 0 / 1 = off, n = force (needs ws) */
  int xcd_mode;                 /* tile order over the 8 XCDs: -1 = auto, 0 = M-chunks, 1 = N-chunks (speed only) */
  int algo;                     /* kernel family: 0 = auto; L2S_ALGO_* forces one (benchmarks / tests; same arithmetic, speed only) */
  float* ws;                    /* optional float workspace for split-K partial sums: split x rows x Cout floats (any contents; one per stream) */
  size_t ws_floats;             /* its size; the split is reduced until the slabs fit */
  int prio;                     /* 1: the kernel's waves raise their issue priority (s_setprio 3): launches of a latency-bound dependent chain that shares
                                   CUs with throughput-bound background work (the backbone's convolutions beside the weight-gradient stream) */
} l2s_conv_desc;
int l2s_conv_igemm(const l2s_conv_desc* d, int dtype, hipStream_t stream);
/* name of the kernel l2s_conv_igemm launches for this problem (reporting: bench.py's roofline object); static storage */
const char* l2s_conv_plan_name(const l2s_conv_desc* d, int dtype);

/* weight gradient: dw[Cout][KH*KW*Cin] (float) += sum_pixels dy[p][co] * x(p,tap)[ci]
 * (cuDNN backward-filter behind autograd in the reference: resnet_v1_cycle_res5_2.py:83-88,324-335, network_cycle_res5_2.py:236-251,279-301).
 * No floating-point atomics: with split-K every workgroup stores its partial tile into its own slab of `ws` and a second launch on the
 * same stream adds the slabs to dw in a fixed order, so the result is bit-identical from run to run.  Without a workspace (or when it is
 * too small for two slabs) the pixels are not split.  The caller must not run two launches that accumulate into the SAME dw, or that
 * share a workspace, concurrently on different streams.  l2s_wgrad_ws_bytes: workspace the automatic split of this problem would use. */
typedef struct {
  const void* dy;  /* [n_img*OH*OW][lddy] (dtype) */
  const void* x;   /* [n_img, IH, IW, ldx] (dtype) */
  float* dw;
  int n_img, IH, IW, Cin, OH, OW, Cout, KH, KW, stride, pad;
  int lddy, ldx;
  int split_k;     /* 0 = auto */
  int tile;        /* 0 = auto, 64 or 128 (output-channel tile) */
  float* ws;       /* split-K slabs (device), may be NULL */
  size_t ws_bytes;
} l2s_wgrad_desc;
size_t l2s_wgrad_ws_bytes(const l2s_wgrad_desc* d, int dtype);

/* Deferred, grouped weight gradients: every weight gradient of a backward stage in ONE launch (nothing needs them before the optimiser
 * / the gradient all-reduce of that stage).  Each workgroup owns a whole output tile over all pixels: no split-K, no atomics, no partial
 * sums, bit-reproducible, and a few thousand workgroups fill the chip where a single small layer has ~100 tiles.  A tensor that is used
 * twice in the step (resnet.layer4 on the RoIs and on the whole map, network_cycle_res5_2.py:415-435) is one problem with two pixel
 * segments.  All problems of a launch use the same tile variant (l2s_wgrad_variant: 0 = 64x64 tile per tap, 1 = 128x128 per tap,
 * 2 = 64x64 x three taps of a 3x3 filter row, 3 = 128x64 x filter row); `table_dev` is the device copy of `table_host` and must stay
 * valid until the launch has run.  dw += gradient.  Problems with few output tiles and many pixels (layer2: 9375 pixels, 128 channels)
 * may still split their pixels (`split`): slabs in `ws`, no atomics either. */
#define L2S_WGRAD_MAX_GROUP 64
#define L2S_WGRAD_MAX_SEG 2
typedef struct {
  const void* dy[L2S_WGRAD_MAX_SEG];   /* [n_img*OH*OW][lddy] (dtype) per segment */
  const void* x[L2S_WGRAD_MAX_SEG];    /* [n_img, IH, IW, ldx] (dtype) per segment */
  int n_img[L2S_WGRAD_MAX_SEG], IH[L2S_WGRAD_MAX_SEG], IW[L2S_WGRAD_MAX_SEG], OH[L2S_WGRAD_MAX_SEG], OW[L2S_WGRAD_MAX_SEG];
  int lddy[L2S_WGRAD_MAX_SEG], ldx[L2S_WGRAD_MAX_SEG];
  float* dw;                           /* [Cout][KH*KW*Cin] */
  int nseg, Cin, Cout, KH, KW, stride, pad;
  int split;                           /* <= 1: whole tiles; n: the pixels of every segment are cut into n ranges whose partial tiles go to */
  long ws_off;                         /*       n slabs at float offset ws_off of the launch's workspace, summed in order by a second launch */
  int flags;                           /* 1: dw = gradient (the tensor's first contribution of the step: nothing is read, nothing had to be cleared); */
  int reserved;                        /*    0: dw += gradient */
} l2s_wgrad_prob;
int l2s_wgrad_variant(int Cin, int Cout, int KH, int KW, int stride, int pad, int same_hw, long M, int tile);
long l2s_wgrad_tiles(int variant, int Cin, int Cout, int KH, int KW);
/* variant 5 (bf16, 3x3 / stride 1 / pad 1, >= 8192 pixels, channels multiples of 128): the LDS-DMA filter-row tile, balanced stream-K style
 * over 128 workgroups; its slabs need l2s_wgrad_grouped_ws_bytes(5) of workspace. */
size_t l2s_wgrad_grouped_ws_bytes(int variant);
int l2s_conv_wgrad_grouped(const l2s_wgrad_prob* table_dev, const l2s_wgrad_prob* table_host, int nprob, int variant, int dtype,
                           float* ws, size_t ws_bytes, hipStream_t stream);
int l2s_conv_wgrad(const l2s_wgrad_desc* d, int dtype, hipStream_t stream);

/* shadow weights: dst(dtype)[Cout][taps][Cin] = scale[co] * src[Cout][taps][Cin]  (scale may be NULL) */
int l2s_weight_cast(const float* src, const float* scale, void* dst, int Cout, int taps, int Cin, int dtype, hipStream_t s);
/* data-gradient layout: dst(dtype)[Cin][taps][Cout], tap order reversed (180-degree flip), * scale[co] */
int l2s_weight_transpose(const float* src, const float* scale, void* dst, int Cout, int taps, int Cin, int dtype, hipStream_t s);
/* all data-gradient copies of one step in one launch: table (DEVICE memory) of n descriptors; total_tiles = sum over the descriptors of
 * ceil(Cin / 64) * ceil(Cout / 64) * taps (the launch walks one flat tile index) */
typedef struct { const float* src; const float* scale; void* dst; int Cout, taps, Cin;
                 int force_f32; /* bit 0: dst is f32 whatever the launch's dtype; bit 1: src points at bf16 values (the shadow: scale = NULL) */ } l2s_transpose_desc;
int l2s_weight_transpose_batched(const l2s_transpose_desc* table_dev, int n, int total_tiles, int dtype, hipStream_t s);
/* column sums: out[c] += sum_r a[r][c] (bias gradients), no atomics; ws (nullable, 32*cols floats): partial sums of the row ranges a tall
 * matrix is cut into, added in order by a second launch */
int l2s_colsum(const void* a, int rows, int cols, int lda, float* out, float* ws, long ws_floats, int dtype, hipStream_t s);

/* stem: conv 7x7 s2 p3 (3->64) + frozen-BN affine + ReLU (RES:121-124), input float NHWC; then maxpool k3 s2 p1 (RES:126) */
int l2s_stem_conv(const float* img, const float* w /*[64][7][7][3]*/, const float* scale, const float* bias,
                  void* y, int H, int W, int OH, int OW, int dtype, hipStream_t s);
/* bf16 mode: the same stem AND the pooling behind it in one launch on the matrix cores (stem_mfma.hip): image as two bf16 terms (16
 * significant bits), weights rounded to bf16 in the fragment order l2s_stem_pack() writes (l2s_stem_pack_bytes() bytes; repack whenever
 * conv1's weights change), f32 accumulation / affine / ReLU, one rounding to bf16, pooled output y [PH*PW][64] bf16. */
size_t l2s_stem_pack_bytes(void);
int l2s_stem_pack(const float* w /*[64][7][7][3]*/, void* pack, hipStream_t s);
int l2s_stem_pool_bf16(const float* img, const void* pack, const float* scale, const float* bias, void* y, int H, int W,
                       int OH, int OW, int PH, int PW, hipStream_t s);
/* VGG16 variant (nets/vgg16.py:43-54, network_vgg.py:139-143): conv1_1 (3 -> 64, 3x3, pad 1, bias, ReLU) on the fp32 NHWC image with
 * weights [64][3][3][3]; 2x2 / stride-2 max pooling (floor mode) forward / backward over n_img NHWC maps [IH][IW][C] (also the
 * 14x14 -> 7x7 pool behind the crop-pool).  relu_out != 0: x is a ReLU output, dx is the gradient of its pre-activation. */
/* out = x * mask (fp32 scaled dropout mask), zero where relu_ref <= 0 (nullable): dropout forward / backward of fc6 and fc7 */
int l2s_scale_mask(const void* x, const float* mask, const void* relu_ref, void* out, long n, int dtype, hipStream_t s);
int l2s_conv3x3_c3(const float* img, const float* w, const float* bias, void* y, int H, int W, int dtype, hipStream_t s);
int l2s_maxpool2x2_fwd(const void* x, void* y, int n_img, int IH, int IW, int C, int dtype, hipStream_t s);
int l2s_maxpool2x2_bwd(const void* dy, const void* x, void* dx, int n_img, int IH, int IW, int C, int relu_out, int dtype, hipStream_t s);
int l2s_maxpool3x3s2(const void* x, void* y, int IH, int IW, int C, int OH, int OW, int dtype, hipStream_t s);

/* ---------------------------------------------------------------- pooling / elementwise ---- */
int l2s_fill_f32(float* p, float v, long n, hipStream_t s);
int l2s_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long n, hipStream_t s);
int l2s_mul_f32(const float* a, const float* b, float* out, long n, hipStream_t s);
int l2s_add_f32(const float* a, const float* b, float* out, long n, hipStream_t s);   /* out = a + b (out may alias a or b) */
/* dst(dtype) = a(dtype) + b(dtype) + c(float)   (any of b, c may be NULL) */
int l2s_add3(const void* a, const void* b, const float* c, void* dst, long n, int dtype, hipStream_t s);
/* spatial mean over hw pixels per image: y[n][c] = mean_p x[n][p][c]  (NET:278) and its backward */
int l2s_avgpool_fwd(const void* x, void* y, int n_img, int hw, int C, int dtype, hipStream_t s);
int l2s_avgpool_bwd(const void* dy, void* dx, const void* addend, const void* relu_ref, int n_img, int hw, int C, int dtype, hipStream_t s);
/* adaptive_avg_pool2d (H,W)->(OH,OW), bins [floor(i*H/OH), ceil((i+1)*H/OH)) (NET:419,432); optional {0,1} pixel mask */
int l2s_adaptive_pool_fwd(const void* x, const float* pixmask, void* y, int H, int W, int C, int OH, int OW, int ldy, int dtype, hipStream_t s);
/* dx[p][c] (+)= sum over bins containing p of dy_all[bin][c]/cnt + pixmask[p]*dy_mask[bin][c]/cnt ; then ReLU-masked by relu_ref */
int l2s_adaptive_pool_bwd(const void* dy, int lddy, int off_all, int off_mask, const float* pixmask, void* dx,
                          const void* relu_ref, int H, int W, int C, int OH, int OW, int dtype, hipStream_t s);
/* gt mask (uint8 HxW) -> adaptive avg pool to (h,w) -> >= 0.5 (NET:424-428) */
int l2s_mask_downsample(const uint8_t* mask, float* out, int H, int W, int h, int w, hipStream_t s);
/* counter-based hash RNG: the step counter lives in DEVICE memory (so a captured hipGraph replays with fresh randomness);
 * l2s_counter_inc bumps it once per step; `salt` separates the call sites */
int l2s_counter_inc(uint64_t* counter_dev, hipStream_t s);
/* diagnostics (tools/step_timeline.py): one-thread launch that stores the device's constant 100 MHz clock when the stream reaches it */
int l2s_stamp(uint64_t* slot_dev, hipStream_t s);
/* dropout mask, entries 0 or 1/(1-p) */
int l2s_dropout_mask(float* mask, long n, float p, const uint64_t* seed_dev, uint64_t salt, hipStream_t s);

/* ---------------------------------------------------------------- RoI path ------------------ */
/* RPN pair-softmax + box decode + clip (NET:242-246, proposal_layer.py:42-46, bbox_transform.py:36-80).
 * heads: float [HW][ldh] with columns [0,2A) = cls scores (bg A, fg A), [2A,6A) = bbox deltas.
 * outputs: prob [HW][2A] float, boxes [HW*A][4] float, scores [HW*A] float (fg prob) */
int l2s_rpn_decode(const float* heads, int ldh, const float* base_anchors /*[A][4]*/, int H, int W, int A, int feat_stride,
                   float im_h, float im_w, float* prob, float* boxes, float* scores, hipStream_t s);
/* descending stable sort (ties: lower index first), emits the top-k boxes/scores in order.
 * ws: int32 workspace of l2s_sort_ws_ints(n); sorted_boxes [k][4], sorted_scores [k], sorted_idx [k] (int32) */
long l2s_sort_ws_ints(int n);
int l2s_sort_topk(const float* scores, const float* boxes, int n, int k, int* ws, float* sorted_boxes,
                  float* sorted_scores, int* sorted_idx, hipStream_t s);
/* greedy NMS over score-sorted boxes (replaces gpu_nms / cpu_nms, nms_cuda.c:17, nms.c:4).
 * cmp_mode 0: suppress when IoU >= thresh (cpu_nms, nms.c:59); 1: IoU > thresh (nms_kernel.cu:63).
 * mask_ws: l2s_nms_workspace_bytes(n) bytes (8-byte aligned; the bit mask of ONE 4096-box stage - the boxes are scanned in stages, a
 * stage's mask is that of its own boxes and what earlier stages kept enters as one OR word per 64 boxes - plus those words: 2 MB at
 * n = 12000, where the reference's mask is 18 MB).  max_keep >= 1.  keep_out[max_keep] int32 (indices into the sorted list),
 * num_out[1] int32 — both DEVICE memory (the 18 MB D2H + host loop of nms_cuda.c:47-58 is gone); the stages after the first return
 * at once when keep_out is already full.  num_out[0] = -1: a wave of the scan gave up a bounded wait (not reachable while the scan's
 * workgroup runs as a whole); the remaining stages return at once and l2s_gather_rois emits an empty list. */
size_t l2s_nms_workspace_bytes(int n);
int l2s_nms(const float* sorted_boxes, int n, float thresh, int cmp_mode, int max_keep, uint64_t* mask_ws,
            int* keep_out, int* num_out, hipStream_t s);
/* gather kept proposals: rois[max_keep][5] = [0, box], roi_scores[max_keep]; rows >= *num_out are zero */
int l2s_gather_rois(const float* sorted_boxes, const float* sorted_scores, const int* keep, const int* num, int max_keep,
                    float* rois, float* roi_scores, hipStream_t s);
/* uint32 sampling priority keys from the same counter hash */
int l2s_random_keys(uint32_t* keys, long n, const uint64_t* seed_dev, uint64_t salt, hipStream_t s);

/* anchor_target_layer.py:19-153 on device.  gt float [n_gt][5]; keys uint32 [HWA] (smallest keys are disabled first).
 * outputs: labels int32 [A*H*W] in the (a,h,w) order of ATL:133-134 (-1,0,1), targets/inside/outside float [HW][4A].
 * ws: int32/float scratch of l2s_anchor_target_ws_ints(HWA) ints. */
long l2s_anchor_target_ws_ints(int hwa);
int l2s_anchor_target(const float* gt, int n_gt, const float* base_anchors, int H, int W, int A, int feat_stride,
                      float im_h, float im_w, const uint32_t* fg_keys, const uint32_t* bg_keys,
                      float neg_ov, float pos_ov, int batch, float fg_frac,
                      int* labels, float* targets, float* inside_w, float* outside_w, int* ws, hipStream_t s);

/* proposal_target_layer.py:22-204 on device (torch-0.3 ByteTensor semantics at :146).
 * rois [n_max][5], n_rois device int; gt [n_gt][5]; gt_masks uint8 [n_gt][im_h][im_w].
 * outputs: out_rois [R][5], labels int32 [R], bbox_targets/inside/outside [R][4*ncls], mask_targets float
 * [mask_slots][ms*ms], counts int32[4] = {num_fg (capped at mask_slots), n_fg_cand, n_bg_cand, appended_gt}.  ws: int32 [4*(n_max+n_gt)+16].
 * fg_max = round(FG_FRACTION * R) caps the sampled foreground when background candidates exist; without any (PTL:155-158) all R sampled
 * RoIs are foreground, and the mask targets cover the first mask_slots (fg_max <= mask_slots <= R) of them. */
int l2s_proposal_target(const float* rois, const float* roi_scores, const int* n_rois, int n_max, const float* gt, int n_gt,
                        const uint8_t* gt_masks, int im_h, int im_w, const uint32_t* fg_keys, const uint32_t* bg_keys,
                        const uint32_t* bg_rand, int R, int fg_max, int mask_slots, float fg_thresh, float bg_hi, float bg_lo,
                        const float* means4, const float* stds4, const float* inw4, int ncls, int ms,
                        float* out_rois, int* labels, float* bbox_targets, float* bbox_inside, float* bbox_outside,
                        float* mask_targets, int* counts, int* ws, hipStream_t s);

/* RoIAlign fused into the first bottleneck of the RoI head (bf16 activations): crop-and-resize (network_cycle_res5_2.py:107-149) feeding
 * resnet.layer4[0].conv1 (+ frozen-BN shift + ReLU) and resnet.layer4[0].downsample (+ shift), resnet_v1_cycle_res5_2.py:271-273, in ONE
 * launch, one workgroup per RoI.  `pooled` receives the P x P x C crop exactly as l2s_roialign_fwd writes it (the weight gradients and the
 * backward pass read it; the forward pass does not).  Requirements: P*P <= 64, C % 128 == 0, P*P*C*2 + 4 KiB <= 160 KiB of LDS,
 * N1 % 128 == 0, N2 % 128 == 0; L2S_EINVAL otherwise (callers fall back to l2s_roialign_fwd + two l2s_conv_igemm). */
typedef struct {
  const void* feat;        /* [H*W][C] bf16 */
  const float* rois;       /* [R][5] (batch index, x1, y1, x2, y2) in image pixels */
  const void* w1;          /* [N1][C] bf16 (BN scale folded) */
  const float* b1;         /* [N1] or NULL */
  const void* w2;          /* [N2][C] bf16 */
  const float* b2;         /* [N2] or NULL */
  void* pooled;            /* [R*P*P][C] bf16 */
  void* y1;                /* [R*P*P][N1] bf16 = relu(pooled . w1^T + b1) */
  void* y2;                /* [R*P*P][N2] bf16 = pooled . w2^T + b2 */
  int H, W, C, R, P, N1, N2;
  float spatial_scale;
  int debug;               /* 0; tools/roi_block0_bench.py: 1 = no products, 2 = no weight loads behind the prologue, 4 = no stores */
} l2s_roi_block0_desc;
int l2s_roialign_block0_fwd(const l2s_roi_block0_desc* d, hipStream_t stream);
/* crop-and-resize RoIAlign = affine_grid + grid_sample, align_corners=True, zero padding (NET:107-149) */
int l2s_roialign_fwd(const void* feat, int H, int W, int C, const float* rois, int R, int P, float spatial_scale,
                     void* out, int dtype, hipStream_t s);
/* d(feat) float [H*W][C] of dout [R*P*P][C].  dfeat is WRITTEN (round 5: it need not arrive cleared): a gather per map pixel in a fixed
 * summation order, no atomics, when C % 4 == 0 and P <= 32; otherwise cleared here and scattered into with atomics */
int l2s_roialign_bwd(const void* dout, int H, int W, int C, const float* rois, int R, int P, float spatial_scale,
                     float* dfeat, int dtype, hipStream_t s);

/* _crop_pool_layer_align (NET:151-182, selected by cfg.POOLING_ALIGN at NET:569-570,610-611): the same sampler with the affine
 * grid computed from the RoI in image pixels over the image size (im_info) instead of roi/16 over the map size */
int l2s_cropalign_fwd(const void* feat, int H, int W, int C, const float* rois, int R, int P, float im_h, float im_w,
                      void* out, int dtype, hipStream_t s);
/* d(feat) is WRITTEN, like l2s_roialign_bwd's: a caller that accumulates into an existing gradient adds the result itself */
int l2s_cropalign_bwd(const void* dout, int H, int W, int C, const float* rois, int R, int P, float im_h, float im_w,
                      float* dfeat, int dtype, hipStream_t s);

/* ---------------------------------------------------------------- losses -------------------- */
/* loss slots in the float loss[8] buffer */
#define L2S_LOSS_RPN_CLS 0
#define L2S_LOSS_RPN_BOX 1
#define L2S_LOSS_CLS 2
#define L2S_LOSS_BOX 3
#define L2S_LOSS_MASK 4
#define L2S_LOSS_CAP 5
#define L2S_LOSS_TOTAL 6
#define L2S_LOSS_RESPONSE 7   /* network_7f_response.py:411-419 / network_cycle_response.py:415-423; 0 in the other variants */
/* RPN CE over anchors with label != -1 (NET:377-382) + smooth-L1 sigma=3 (NET:385-390).
 * heads as in l2s_rpn_decode; labels in (a,h,w) order.  dheads(dtype) [HW][ldd] receives d(loss)/d(heads)*gscale. */
/* RoI max pooling, POOLING_MODE == 'pool' (layer_utils/roi_pooling/roi_pool.py:19-50 -> src/cuda/roi_pooling_kernel.cu:15-70,104-180;
 * the reference's second native FFI besides NMS).  feat NHWC [H*W][C]; out [R*P*P][C]; argmax [R*P*P][C] int32 = h*W + w of the
 * maximum (-1: empty bin); backward adds dout into dfeat [H*W][C] f32 through the recorded argmax (dfeat is not cleared). */
int l2s_roipool_fwd(const void* feat, int H, int W, int C, const float* rois, int R, int P, float spatial_scale, void* out, int* argmax,
                    int dtype, hipStream_t s);
int l2s_roipool_bwd(const void* dout, const int* argmax, int R, int P, int C, float* dfeat, int dtype, hipStream_t s);
int l2s_rpn_loss(const float* heads, int ldh, const int* labels, const float* targets, const float* inside_w,
                 const float* outside_w, int H, int W, int A, float sigma, float gscale, float* loss /* += */, void* dheads, int ldd,
                 int dtype, const int* count_dev /* #labels != -1 (l2s_anchor_target_count) or NULL */, int* count_ws /* 1 int, used when count_dev is NULL */,
                 hipStream_t s);
/* device pointer to the number of sampled anchors inside an l2s_anchor_target workspace */
const int* l2s_anchor_target_count(const int* ws);
/* RCNN CE mean over R (NET:393-395) + smooth-L1 sigma=1 (NET:398-402). heads float [R][ldh]: [0,ncls) cls, [ncls,5ncls) bbox */
int l2s_rcnn_loss(const float* heads, int ldh, const int* labels, const float* bbox_targets, const float* inside_w,
                  const float* outside_w, int R, int ncls, float gscale, float* loss, void* dheads, int ldd, int dtype, hipStream_t s);
/* mask BCE-with-logits on the class channel of the first num_fg rois (NET:405-413). score float [fg_max*ms2][ldsc] */
int l2s_mask_loss(const float* score, int ldsc, const int* labels, const float* mask_targets, const int* num_fg, int fg_max,
                  int ms2, float gscale, float* loss, float* dscore /*[fg_max*ms2]*/, hipStream_t s);
/* total = cls + box + rpn_cls + rpn_box + mask + w*cap (NET:448) */
/* response BCE loss of the *_response variants: target = GT mask [mask_h][mask_w] u8 {0,1} resized PIL-NEAREST to [H][W];
 * loss[L2S_LOSS_RESPONSE] += mean BCE; dresp[HW] = gscale * d loss / d response */
int l2s_response_loss(const float* resp, const uint8_t* gt_mask, int mask_h, int mask_w, int H, int W, float gscale, float* loss,
                      float* dresp, hipStream_t s);
/* TEST mode (NET:277-307, 650-658; test_image NET:684-699): heads [R][ldh] = (cls scores | box deltas) ->
 * cls_prob [R][ncls] (softmax), bbox_pred [R][4 ncls] de-normalised with TRAIN.BBOX_NORMALIZE_STDS / MEANS (4 floats each) */
int l2s_rcnn_predict(const float* heads, int ldh, int R, int ncls, const float* stds4, const float* means4, float* cls_prob,
                     float* bbox_pred, hipStream_t s);
/* mask probabilities: score [n_elem][ldsc] (n_elem = n_roi * ms2 mask pixels); labels == NULL -> out [n_elem][ncls] = sigmoid(score)
 * (NET:292-307); labels [n_roi] -> out [n_elem] = sigmoid of the labelled class only (NET:620-624) */
int l2s_mask_prob(const float* score, int ldsc, int ncls, const int* labels, int ms2, long n_elem, float* out, hipStream_t s);
int l2s_total_loss(float* loss, float cap_w, hipStream_t s);
/* mask_pred_net backward (only the label channel carries gradient): dx(dtype)[fg_max*ms2][C] = dscore[p]*W[label][:],
 * dW[label][:] += sum dscore[p]*x[p][:], db[label] += sum dscore[p]; ws: l2s_maskpred_ws_floats(fg_max, C) floats of per-(RoI, pixel chunk) partial sums
 * (added in order by a second launch: no atomics) */
long l2s_maskpred_ws_floats(int fg_max, int C);
int l2s_maskpred_bwd(const float* dscore, const int* labels, const int* num_fg, int fg_max, int ms2, int C,
                     const float* w /*[ncls][C]*/, const void* x, const void* relu_ref, void* dx, float* dw, float* db, float* ws, int dtype,
                     hipStream_t s);
/* its two launches separately: l2s_maskpred_bwd_dx = dx and the partial sums into ws; l2s_maskpred_bwd_reduce = dW / db from ws (may run on another
 * stream behind it: nothing of the data-gradient chain reads dW) */
int l2s_maskpred_bwd_dx(const float* dscore, const int* labels, const int* num_fg, int fg_max, int ms2, int C, const float* w,
                        const void* x, const void* relu_ref, void* dx, float* ws, int dtype, hipStream_t s);
int l2s_maskpred_bwd_reduce(const float* ws, const int* labels, const int* num_fg, int fg_max, int C, float* dw, float* db, hipStream_t s);

/* ---------------------------------------------------------------- language side ------------- */
/* small-M linear layers, fp32: y[m][n] = act(sum_k x[m][k] w[n][k] + b[n] (+ y_in)) ; act 0 none, 1 relu, 2 tanh */
int l2s_linear_fwd(const float* x, int ldx, const float* w, int ldw, const float* b, float* y, int ldy, int M, int N, int K, int act,
                   int accumulate, hipStream_t s);
/* dx[m][k] (+)= (sum_n dy[m][n] w[n][k]) (* mul[m][k], same leading dimension as dx; may be NULL) from the weight as stored.
 * Row batches split the contraction over workgroups when `ws` (>= l2s_linear_bwd_x_ws_floats floats) is given: partial sums in ws,
 * added in a fixed order by a second launch (no atomics); without ws one workgroup per 64 output columns walks all of n. */
long l2s_linear_bwd_x_ws_floats(int M, int N, int K);
int l2s_linear_bwd_x(const float* dy, int lddy, const float* w, float* dx, int lddx, int M, int N, int K, int accumulate,
                     const float* mul, float* ws, long ws_floats, hipStream_t s);
/* dw[n][k] += sum_m dy[m][n] x[m][k]; db[n] += sum_m dy[m][n] */
int l2s_linear_bwd_w(const float* dy, int lddy, const float* x, int ldx, float* dw, float* db, int M, int N, int K, hipStream_t s);
/* activation backward in place: dy *= act'(y)  (act 1 relu, 2 tanh) */
int l2s_act_bwd(float* dy, const float* y, long n, int act, hipStream_t s);
/* x *= mul (mul may be NULL); x = 0 where !(relu_ref > 0); out = x cast to out_dtype: dropout mask + ReLU backward + cast in one launch */
int l2s_mask_relu_cast(float* x, const float* mul, const float* relu_ref, void* out, int out_dtype, long n, hipStream_t s);
/* embedding gather out[t][:] = table[ids[t]][:] * (mask ? mask[t][:] : 1), optional relu; and scatter-add backward */
int l2s_embed_fwd(const float* table, const int64_t* ids, const float* mask, float* out, int T, int D, int relu, hipStream_t s);
int l2s_embed_bwd(const float* dout, const float* out, const int64_t* ids, const float* mask, float* dtable, int T, int D, int relu, hipStream_t s);
/* nn.LSTM cell (gate order i,f,g,o; lang_encoder.py:21-24): gates[4H] pre-activations (already x-proj + h-proj + biases) */
/* Fused bi-LSTM time step (lang_encoder.py:62; nn.LSTM gate order i,f,g,o), one launch for up to two directions.
 * forward : gates_out[4H] = gates_in[4H] (x W_ih^T + b_ih, precomputed) + W_hh h_prev + b_hh; cell update -> c, h, act[4H] (gate activations)
 * backward: dh = w_hh_T[H][4H] . dgates_next (NULL at the first step) + dh_ext (NULL or the gradient of the final hidden state);
 *           then the cell backward of this step: dgates[4H], dc_prev[H] */
typedef struct { const float* w_hh; const float* b_hh; const float* gates_in; const float* h_prev; const float* c_prev;
                 float* c; float* h; float* act; float* gates_out; } l2s_lstm_fwd_dir;
typedef struct { const float* w_hh_T; const float* dgates_next; const float* dh_ext; const float* dc_in; const float* act;
                 const float* c_prev; const float* c; float* dgates; float* dc_prev; } l2s_lstm_bwd_dir;
int l2s_lstm_step_fwd(const l2s_lstm_fwd_dir* dirs, int ndir, int Hh, hipStream_t s);
int l2s_lstm_step_bwd(const l2s_lstm_bwd_dir* dirs, int ndir, int Hh, hipStream_t s);
int l2s_lstm_cell_fwd(const float* gates, const float* c_prev, float* c, float* h, float* act /*[4H] saved*/, int Hh, hipStream_t s);
int l2s_lstm_cell_bwd(const float* dh, const float* dc_in, const float* act, const float* c_prev, const float* c,
                      float* dgates, float* dc_prev, int Hh, hipStream_t s);
/* dynamic-filter correlation (NET:504-562): filt float [7][C] (tanh'ed), r float [7].
 * y(dtype)[HW][C] = x * resp, resp float [HW], respk float [HW][7] (masked per-filter responses) */
int l2s_dynfilter_fwd(const void* x, const float* filt, const float* r, void* y, float* resp, float* respk, int H, int W, int C,
                      int dtype, int gate /*0: y = x*response, 1: y = x*sigmoid(response) (the *_response variants)*/, hipStream_t s);
/* dy(dtype) -> dx(dtype), dfilt float [7][C] (+=), dr float [7] (+=); ws: l2s_dynfilter_ws_floats(H, W, C) floats (d(response) per pixel
 * + per-pixel-chunk partial filter gradients, summed in chunk order by a third launch: no atomics) */
long l2s_dynfilter_ws_floats(int H, int W, int C);
int l2s_dynfilter_bwd(const void* dy, const void* x, const float* filt, const float* r, const float* resp, const float* respk,
                      void* dx, const void* relu_ref, float* dfilt, float* dr, float* ws, int H, int W, int C, int dtype,
                      int gate, const float* dresp_extra /*nullable [HW]: d(response loss)/d(response)*/, hipStream_t s);
/* dfilt == NULL in l2s_dynfilter_bwd: only dx is produced (the main queue needs nothing else); l2s_dynfilter_bwd_finish then turns the
 * workspace into dfilt += / dr += on whichever stream the language-side backward runs on. */
int l2s_dynfilter_bwd_finish(const float* ws, const float* respk, float* dfilt, float* dr, int H, int W, int C, hipStream_t s);
/* att2in2 attention (AttModel.py:406-423): patt [L][D], att [L][D] float; att_h [D]; alpha w[D], b.
 * out: weight [L] (softmax), att_res [D + 256] (the tail is scratch for the raw dots) */
int l2s_cap_attention_fwd(const float* patt, const float* att, const float* att_h, const float* aw, const float* ab, int L, int D,
                          float* tanh_ws /*[L][D]*/, float* weight, float* att_res, hipStream_t s);
int l2s_cap_attention_bwd(const float* datt_res, const float* att, const float* tanh_ws, const float* weight, const float* aw, int L, int D,
                          float* dpatt /*[L][D] +=*/, float* datt /*[L][D] +=*/, float* datt_h /*[D + 256] = (tail: scratch)*/, float* daw /*[D] +=*/, float* dab /*+=*/, hipStream_t s);
/* att2in2 core gates (AttModel.py:446-466): s[5R] = i2h+h2h, a2c[2R]; maxout candidate, no tanh */
/* Fused per-step launches of the att2in2 recurrence (AttModel.py:406-466): the recurrence is a chain of dependent launches on
 * the caption stream, so the per-step work is packed into as few launches as the data dependencies allow.
 *   l2s_linear2_fwd        : h2att(h) and h2h(h) (+= into the i2h sums) from one read of h            (ATT:425,453)
 *   l2s_cap_a2c_gates_fwd  : a2c Linear + sigmoid / maxout / cell update of l2s_cap_gates_fwd          (ATT:449-462)
 *   l2s_linear_sum2_fwd    : dh = W_h2h^T dsums + W_h2att^T datt_h (transposed copies, one launch)
 *   l2s_cap_attention_bwd_step / _batched : the attention backward split into what the recurrence needs at step t
 *       (ddot[L], datt_h[D]) and the step-summed parameter / feature gradients computed once after the loop. */
int l2s_linear2_fwd(const float* x, int K, const float* w1, const float* b1, float* y1, int N1, int acc1, const float* w2, const float* b2,
                    float* y2, int N2, int acc2, hipStream_t s);
int l2s_linear_sum2_fwd(const float* x1, const float* w1 /*[N][K1]*/, int K1, const float* x2, const float* w2 /*[N][K2]*/, int K2, float* y, int N,
                        int accumulate, hipStream_t s);
int l2s_cap_a2c_gates_fwd(const float* att_res, const float* w_a2c /*[2R][K]*/, const float* b_a2c, int K, const float* sums /*[5R]*/,
                          const float* c_prev, float* c, float* h, float* save /*[6R]*/, int R, hipStream_t s);
/* Projected-attention form of the same step (csrc/lang.hip): P = att . W_a2c^T [L][2R] once per sentence, then per token
 *   l2s_cap_att_dots_fwd      : dots[l] = alpha . tanh(patt[l] + att_h) + b                                  (ATT:411-418)
 *   l2s_cap_apply_gates_fwd   : softmax(dots) -> weight[L]; a2c = sum_l weight[l] P[l] + b_a2c; gates / cell update  (ATT:419-423,449-462)
 *   l2s_cap_gates_bwd_dw      : gate backward (as l2s_cap_gates_bwd) + dweight[l] = P[l] . da2c
 *   l2s_cap_attention_bwd_step2 : softmax backward from dweight -> ddot[L], datt_h[D]
 * l2s_cap_attention_bwd_batched accepts datt_res == NULL (no d(att) term: it then comes from d(P)). */
int l2s_cap_att_dots_fwd(const float* patt, const float* att_h, const float* aw, const float* ab, int L, int D, float* tanh_ws, float* dots, hipStream_t s);
int l2s_cap_apply_gates_fwd(const float* P, const float* dots, const float* b_a2c, const float* sums, const float* c_prev, float* c, float* h,
                            float* save, float* weight, int L, int R, hipStream_t s);
int l2s_cap_gates_bwd_dw(const float* dh, const float* dh2, const float* dc_in, const float* save, const float* c_prev, const float* P,
                         float* dsums, float* da2c, float* dc_prev, float* dweight, int L, int R, hipStream_t s);
int l2s_cap_attention_bwd_step2(const float* dweight, const float* tanh_ws, const float* weight, const float* aw, int L, int D, float* ddot,
                                float* datt_h, hipStream_t s);
int l2s_cap_attention_bwd_step(const float* datt_res, const float* att, const float* tanh_ws, const float* weight, const float* aw, int L, int D,
                               float* ddot /*[L]*/, float* datt_h /*[D]*/, hipStream_t s);
int l2s_cap_attention_bwd_batched(const float* ddot /*[S][L]*/, const float* weight /*[S][L]*/, const float* datt_res /*[S][ldr]*/, int ldr,
                                  const float* tanh_ws /*[S][L][D]*/, const float* aw, int S, int L, int D, float* dpatt /*+=*/, float* datt /*+=*/,
                                  float* daw /*+=*/, float* dab /*+=*/, hipStream_t s);
/* A frozen 64-plane bottleneck behind its first 1x1 convolution as ONE launch (csrc/bottleneck_fused.hip; RES:78-114, the frozen layer1 of
 * RES:291-299), bf16, forward only: b = relu(conv3x3(a) + b2); y = relu(conv1x1(b) + b3 + shortcut), shortcut = x (Cx = 256) or conv1x1(x; wd) + bd
 * (Cx = 64); optionally a_next = relu(conv1x1(y; w1n) + b1n) = the next block's first convolution.  Weights BN-folded bf16 [Cout][KH][KW][Cin],
 * biases f32; every pointer 16-byte aligned; any H, W. */
typedef struct {
  const void* a;                    /* [H*W][64]  */
  const void* x;                    /* [H*W][Cx]  */
  const void *w2, *w3, *wd, *w1n;   /* [64][3][3][64], [256][64], [256][64] (Cx = 64) or NULL, [64][256] or NULL */
  const float *b2, *b3, *bd, *b1n;
  void* y;                          /* [H*W][256] */
  void* a_next;                     /* [H*W][64] or NULL */
  int H, W, Cx;
} l2s_bottleneck64_desc;
int l2s_bottleneck64_fwd(const l2s_bottleneck64_desc* d, hipStream_t s);

/* The whole recurrence as ONE resident launch per direction (csrc/cap_recur.hip; ATT:406-423,446-466): L2S_CAP_RECUR_WGS workgroups own 16 hidden units each
 * (weights in registers, fp32), three granule exchanges per token.  Supported for rnn_size = att_hid_size = 512 and L <= 224 locations
 * (l2s_cap_recur_supported); callers fall back to the three launches per token above otherwise.  `state`: a caller-owned buffer of
 * l2s_cap_recur_state_bytes(backward) bytes, 16-byte aligned, ZEROED ONCE when it is allocated and then left to the kernels (word 0: launch
 * count = the exchange epoch; word 1: set to 1 when a bounded spin gave up - results are then invalid); one per direction. */
#define L2S_CAP_RECUR_WGS 32
typedef struct {
  const float *w_h2h, *b_h2h;      /* [5R][R], [5R] */
  const float *w_h2att, *b_h2att;  /* [AH][R], [AH] */
  const float *patt;               /* [L][AH]  ctx2att(att) */
  const float *aw, *ab;            /* alpha_net weight [AH], bias [1] */
  const float *P, *b_a2c;          /* [L][2R] = att . W_a2c^T, [2R] */
  const float *sums;               /* [S][5R]  i2h(x_t) + b_i2h (read only) */
  float *hs, *cs;                  /* [(S+1)][R], row 0 = the initial state */
  float *save;                     /* [S][6R] as l2s_cap_gates_fwd */
  float *tanh_ws;                  /* [S][L][AH] */
  float *wgt;                      /* [S][L] attention weights */
  unsigned* state;
  int S, R, AH, L;
} l2s_cap_recur_fwd_args;
typedef struct {
  const float *w_h2h, *w_h2att, *P, *aw;
  const float *save, *cs, *wgt, *tanh_ws;   /* as written by the forward launch */
  const float *dho;                /* [S][R] gradient of the loss w.r.t. h_t through the logit layer */
  float *dsums, *da2c;             /* [S][5R], [S][2R] */
  float *ddot, *datt_h;            /* [S][ld_ddot] (L used), [S][ld_datt_h] (AH used) */
  unsigned* state;
  int S, R, AH, L, ld_ddot, ld_datt_h;
} l2s_cap_recur_bwd_args;
int l2s_cap_recur_supported(int S, int R, int AH, int L);
size_t l2s_cap_recur_state_bytes(int backward);
int l2s_cap_recur_fwd(const l2s_cap_recur_fwd_args* a, hipStream_t s);
int l2s_cap_recur_bwd(const l2s_cap_recur_bwd_args* a, hipStream_t s);
int l2s_cap_gates_fwd(const float* sums, const float* a2c, const float* c_prev, float* c, float* h, float* save /*[6R]: sig(3R), sel(R), cand(R), tanh(c)(R)*/, int R, hipStream_t s);
int l2s_cap_gates_bwd(const float* dh, const float* dh2 /*nullable: dh = dh + dh2*/, const float* dc_in, const float* save, const float* c_prev, float* dsums /*[5R]*/, float* da2c /*[2R]*/, float* dc_prev, int R, hipStream_t s);
/* log_softmax + masked NLL (AttModel.py:98, misc/utils.py:43-53): logits [S][V1]; dlogits = gscale*(softmax - onehot)*mask/sum(mask) */
int l2s_logsoftmax_nll(const float* logits, const int64_t* target, const float* mask, int S, int V1, float gscale, float* loss_slot,
                       float* dlogits, float* logprobs_opt, hipStream_t s);

/* ---------------------------------------------------------------- input side ----- */
/* Host. COCO compressed run-length string -> run lengths; replaces rleFrString, pyutils/refer/external/maskApi.c:217-231
 * (reached through external/mask.py decode <- lib/loaders/cycle_loader.py:199).  Returns the number of runs written to
 * cnts[0..max_counts), or -1 for a truncated string / too small a buffer. */
int l2s_rle_from_string(const char* s, uint32_t* cnts, int max_counts);
/* Host. prep_im_for_blob's scale (pyutils/mask-faster-rcnn/lib/utils/blob.py:35-43: target_size / short side, capped so that
 * round(scale * long side) <= max_size) and cv2.resize's output size (round-half-even of h * scale, w * scale). */
int l2s_prep_geometry(int h, int w, int target_size, int max_size, double* scale, int* oh, int* ow);
/* Device. blob.py:32-47 on one image: uint8 BGR [h][w][3] -> float32 [oh][ow][3] = cv2.resize(float32(img) - means, fx = fy =
 * scale, INTER_LINEAR); the result is the `data` blob of cycle_loader.py:119-138 for a batch of one image. */
int l2s_prep_image(const uint8_t* img_bgr, int h, int w, double mean_b, double mean_g, double mean_r, double scale,
                   int oh, int ow, float* out, hipStream_t s);
/* Device. The gt mask of one referred object (cycle_loader.py:198-210): rleDecode (maskApi.c:43-47) of its n run-length objects
 * (column-major runs, concatenated in cnts, object r = cnts[offs[r]..offs[r+1])), union over the objects (sum > 0), PIL-nearest
 * resize from [h][w] to [oh][ow] (scipy.misc.imresize 'nearest'); out uint8 {0,1} row-major.  ws: l2s_rle_ws_words() uint32. */
long l2s_rle_ws_words(int total_counts, int oh, int ow);
int l2s_rle_to_mask(const uint32_t* cnts, const int* offs, int n, int total_counts, int h, int w, int oh, int ow,
                    uint32_t* ws, uint8_t* out, hipStream_t s);

/* ---------------------------------------------------------------- launch tape / streams ----- */
/* `to` waits (device side) for everything enqueued so far on `from`; fork or join of the step's branches */
int l2s_stream_fork(hipStream_t from, hipStream_t to);
/* the two halves of a fork under a process-wide name (slot in [0,16)): l2s_event_record marks "everything enqueued so far on s",
 * l2s_event_wait makes s wait for the most recent mark of that slot (no-op if never marked).  Both are tape ops.  Used where the
 * mark and the wait belong to different steps: the first half of the optimiser update (train_val_cycle.py:194-220) is awaited
 * before the next step's first trainable layer while the deferred weight gradients run on behind it. */
int l2s_event_record(int slot, hipStream_t s);
int l2s_event_wait(int slot, hipStream_t s);
int l2s_memset_async(void* p, int value, size_t bytes, hipStream_t s);
int l2s_memcpy_d2d_async(void* dst, const void* src, size_t bytes, hipStream_t s);
/* record every launch / fork / memset issued through this ABI on the registered streams (they still execute), then replay
 * the whole multi-stream step from one call.  Valid while the recorded pointers, shapes and scalars stay the same. */
void* l2s_tape_begin(const hipStream_t* streams, int n);
int l2s_tape_end(void* tape);
long l2s_tape_size(void* tape);
int l2s_tape_run(void* tape, const hipStream_t* streams, int n);
/* segments: l2s_tape_mark() (while recording) cuts the tape where the host must act between launches (RCCL all-reduce of a finished
 * gradient bucket); l2s_tape_run_segment replays segment `seg` in [0, l2s_tape_segments) */
int l2s_tape_pause(int on);   /* 1: launches execute but are not recorded until l2s_tape_pause(0) (host work between two segments) */
int l2s_tape_mark(void);
/* measurement: "record a timing event here" as a tape op (id >= 0 while recording, -1 otherwise); elapsed HIP-event time between two such
   points of the last replayed step (bench.py brackets the dominant launch inside the pipelined replay with these) */
int l2s_tape_time_event(hipStream_t s);
int l2s_time_event_elapsed(int a, int b, float* ms);
int l2s_tape_segments(void* tape);
int l2s_tape_run_segment(void* tape, const hipStream_t* streams, int n, int seg);
int l2s_tape_destroy(void* tape);

/* ---------------------------------------------------------------- optimizer ---------------- */
/* torch.optim.SGD with momentum as configured at train_val_cycle.py:194-220, fused over a flat parameter buffer.
 * seg table (device): per segment {offset, count, rows, wd_flag}; rowscale (optional, per segment offset into a float array,
 * -1 = none) multiplies the gradient per output row (folded frozen-BN scale). */
typedef struct { long offset; long count; int row_len; int weight_decay; long rowscale_off; float lr_mult;
                 int chunk0; /* running count of ceil(count / l2s_sgd_chunk()) over the table's earlier segments (any common origin) */
                 int flags;  /* 1: the gradient is overwritten whole by its producer every step (l2s_wgrad_prob.flags): the update leaves it alone */
                 int reserved; } l2s_sgd_seg;
/* the same update restricted to the elements [lo, hi) of the flat buffer: a rank's shard of a gradient bucket (data parallel, reduce-scatter
 * -> sharded update -> all-gather of the weights).  [chunk_lo, chunk_hi): the work chunks of the table that can hold such elements, numbered from the
 * table's first chunk (chunk_hi < 0: to the end).  flags: 1 = zero the gradients consumed, 2 = only rewrite the shadow from the parameters. */
int l2s_sgd_momentum_range(float* param, float* grad, float* mom, const l2s_sgd_seg* segs, int nseg, const float* rowscale,
                           float lr, float momentum, float wd, float grad_scale, void* shadow, int shadow_dtype, int flags,
                           long lo, long hi, int chunk_lo, int chunk_hi, hipStream_t s);
/* the ranged update with the gradients read from a bf16 buffer instead: element o of the flat buffer takes grad_bf16[o - grad_lo] (grad_lo a
 * multiple of 4, grad_bf16 8-byte aligned, every updated element at or behind grad_lo).  Data parallel: the reduce-scattered bf16 shard of a
 * bucket feeds the update as it is - no cast pass back into the f32 gradient buffer, which this call neither reads nor clears. */
int l2s_sgd_momentum_range_g16(float* param, const void* grad_bf16, long grad_lo, float* mom, const l2s_sgd_seg* segs, int nseg, const float* rowscale,
                               float lr, float momentum, float wd, float grad_scale, void* shadow, int shadow_dtype,
                               long lo, long hi, int chunk_lo, int chunk_hi, hipStream_t s);
int l2s_sgd_chunk(void);          /* elements of one work chunk of the update kernel */
int l2s_sgd_momentum(float* param, float* grad, float* mom, const l2s_sgd_seg* segs, int nseg, const float* rowscale,
                     float lr, float momentum, float wd, float grad_scale, void* shadow /*optional: dtype copy of rowscale*param at the same offsets*/,
                     int shadow_dtype, int clear_grad /*1: the gradient is zeroed as it is read (optimizer.zero_grad() of train_val_cycle.py:383 folded in)*/,
                     hipStream_t s);

#ifdef __cplusplus
}
#endif
#endif
