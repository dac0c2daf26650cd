"""Thin tensor-level wrappers over the C ABI (include/lang2seg_hip.h).

PyTorch is used for device memory and streams only: every wrapper passes raw
`data_ptr()`s + sizes + the current HIP stream to liblang2seg_hip.so.  Outputs are
caller- or wrapper-allocated torch tensors; no arithmetic happens in torch here."""
import ctypes as C
import torch
from . import _lib
from ._lib import ptr, call, F32, BF16, ConvDesc, WgradDesc

TORCH_DT = {F32: torch.float32, BF16: torch.bfloat16}


def stream():
    return torch.cuda.current_stream().cuda_stream


def dt_of(t):
    return BF16 if t.dtype == torch.bfloat16 else F32


def empty(shape, dt, device='cuda'):
    return torch.empty(shape, dtype=TORCH_DT[dt], device=device)


# ------------------------------------------------------------------ streams / tape
def stream_fork(from_stream, to_stream):
    """`to_stream` waits for everything enqueued so far on `from_stream` (torch.cuda.Stream objects)."""
    call('l2s_stream_fork', from_stream.cuda_stream, to_stream.cuda_stream)


def event_record(slot, s):
    """mark 'everything enqueued so far on stream s' under the process-wide slot number (csrc/tape.hip)"""
    call('l2s_event_record', int(slot), s.cuda_stream)


def event_wait(slot, s):
    """stream s waits for the most recent mark of the slot"""
    call('l2s_event_wait', int(slot), s.cuda_stream)


def memset_zero(t):
    call('l2s_memset_async', ptr(t), 0, t.numel() * t.element_size(), stream())


def memcpy(dst, src):
    call('l2s_memcpy_d2d_async', ptr(dst), ptr(src), src.numel() * src.element_size(), stream())


def tape_begin(streams):
    arr = (C.c_void_p * len(streams))(*[s.cuda_stream for s in streams])
    h = _lib.load().l2s_tape_begin(arr, len(streams))
    if not h:
        raise _lib.L2SError('l2s_tape_begin failed')
    return h


def tape_end(h):
    call('l2s_tape_end', h)


def tape_run(h, streams):
    arr = (C.c_void_p * len(streams))(*[s.cuda_stream for s in streams])
    call('l2s_tape_run', h, arr, len(streams))


def tape_mark():
    call('l2s_tape_mark')


def tape_pause(on):
    call('l2s_tape_pause', 1 if on else 0)


def tape_time_event():
    """(measurement) 'record a timing event here' on the current stream's tape order; id >= 0 while a tape is recording, else -1"""
    return int(_lib.load().l2s_tape_time_event(stream()))


def time_event_elapsed(a, b):
    ms = C.c_float(0.0)
    call('l2s_time_event_elapsed', a, b, C.byref(ms))
    return float(ms.value)


def tape_run_segment(h, streams, seg):
    arr = (C.c_void_p * len(streams))(*[s.cuda_stream for s in streams])
    call('l2s_tape_run_segment', h, arr, len(streams), seg)


def tape_destroy(h):
    torch.cuda.synchronize()                     # nothing of the tape may still be in flight
    call('l2s_tape_destroy', h)


def tape_size(h):
    return int(_lib.load().l2s_tape_size(h))


# ------------------------------------------------------------------ conv / gemm
# kernel family of every convolution that does not name one (l2s_conv_desc.algo: 0 = auto); set by the benchmark tools only
# (bench.py --conv-algo, tools/*bench*.py): same arithmetic, speed only
CONV_ALGO = 0
LAST_PLAN = None        # name of the kernel the dispatcher picked for the last conv_igemm call (bench.py's launch timer reads it)
TRACK_PLAN = False      # set by bench.py's LaunchTimer: the extra ctypes call per launch is measurement only


def conv_igemm(x, w, y, n_img, IH, IW, Cin, OH, OW, Cout, KH=1, KW=1, stride=1, pad=0, bias=None, add=None,
               ref=None, relu=False, out_f32=False, deconv=False, scatter=None, ldx=None, ldy=None, ldadd=None,
               ldref=None, tile=0, dt=None, ws=None, split_k=0, xcd_mode=-1, algo=None, prio=0):
    d = ConvDesc()
    d.x, d.w, d.y = ptr(x), ptr(w), ptr(y)
    d.bias, d.add, d.ref = ptr(bias), ptr(add), ptr(ref)
    d.n_img, d.IH, d.IW, d.Cin, d.OH, d.OW, d.Cout = n_img, IH, IW, Cin, OH, OW, Cout
    d.KH, d.KW, d.stride, d.pad = KH, KW, stride, pad
    d.ldx = Cin if ldx is None else ldx
    ocols = (Cout // 4) if deconv else Cout
    d.ldy = ocols if ldy is None else ldy
    d.ldadd = (ocols if ldadd is None else ldadd)
    d.ldref = (ocols if ldref is None else ldref)
    fl = 0
    if relu:
        fl |= _lib.CONV_RELU
    if out_f32:
        fl |= _lib.CONV_OUT_F32
    if deconv:
        fl |= _lib.CONV_DECONV2X2
    if scatter is not None:
        fl |= _lib.CONV_SCATTER
        d.out_h, d.out_w, d.out_stride = scatter
    d.flags = fl
    d.tile = tile
    d.split_k = split_k
    d.xcd_mode = xcd_mode
    d.algo = CONV_ALGO if algo is None else algo
    d.ws = ptr(ws)
    d.ws_floats = 0 if ws is None else ws.numel()
    d.prio = prio
    global LAST_PLAN
    dtv = dt_of(x) if dt is None else dt
    if TRACK_PLAN:
        LAST_PLAN = _lib.load().l2s_conv_plan_name(C.byref(d), dtv).decode()
    call('l2s_conv_igemm', C.byref(d), dtv, stream())
    return y


def roialign_block0_fwd(feat, H, W, C_, rois, R, P, spatial_scale, w1, b1, N1, w2, b2, N2, pooled, y1, y2, debug=0):
    """RoIAlign fused into layer4[0].conv1 (+ shift + ReLU) and layer4[0].downsample (+ shift): one launch, one workgroup per RoI (bf16)"""
    d = _lib.RoiBlock0Desc()
    d.feat, d.rois, d.w1, d.b1, d.w2, d.b2 = ptr(feat), ptr(rois), ptr(w1), ptr(b1), ptr(w2), ptr(b2)
    d.pooled, d.y1, d.y2 = ptr(pooled), ptr(y1), ptr(y2)
    d.H, d.W, d.C, d.R, d.P, d.N1, d.N2, d.spatial_scale = H, W, C_, R, P, N1, N2, float(spatial_scale)
    d.debug = debug
    call('l2s_roialign_block0_fwd', C.byref(d), stream())


def roialign_block0_ok(C_, P, N1, N2):
    """the fused kernel's shape requirements (include/lang2seg_hip.h)"""
    return P * P <= 64 and C_ % 128 == 0 and N1 % 128 == 0 and N2 % 128 == 0 and P * P * C_ * 2 + 4096 <= 160 * 1024


def _wgrad_desc(dy, x, dw, n_img, IH, IW, Cin, OH, OW, Cout, KH, KW, stride, pad, lddy, ldx, split_k, tile, ws):
    d = WgradDesc()
    d.dy, d.x, d.dw = ptr(dy), ptr(x), ptr(dw)
    d.n_img, d.IH, d.IW, d.Cin, d.OH, d.OW, d.Cout = n_img, IH, IW, Cin, OH, OW, Cout
    d.KH, d.KW, d.stride, d.pad = KH, KW, stride, pad
    d.lddy = Cout if lddy is None else lddy
    d.ldx = Cin if ldx is None else ldx
    d.split_k, d.tile = split_k, tile
    d.ws = ptr(ws)
    d.ws_bytes = 0 if ws is None else ws.numel() * ws.element_size()
    return d


def conv_wgrad(dy, x, dw, n_img, IH, IW, Cin, OH, OW, Cout, KH=1, KW=1, stride=1, pad=0, lddy=None, ldx=None,
               split_k=0, tile=0, ws=None):
    """dw += weight gradient.  `ws`: float workspace for the split-K slabs (summed in a fixed order by a second launch on the same
    stream: no float atomics); None = pixels are not split.  One workspace per stream that runs these launches."""
    d = _wgrad_desc(dy, x, dw, n_img, IH, IW, Cin, OH, OW, Cout, KH, KW, stride, pad, lddy, ldx, split_k, tile, ws)
    call('l2s_conv_wgrad', C.byref(d), dt_of(x), stream())
    return dw


def wgrad_ws_bytes(n_img, IH, IW, Cin, OH, OW, Cout, KH=1, KW=1, stride=1, pad=0, dt=BF16, split_k=0, tile=0):
    d = _wgrad_desc(None, None, None, n_img, IH, IW, Cin, OH, OW, Cout, KH, KW, stride, pad, None, None, split_k, tile, None)
    return int(_lib.load().l2s_wgrad_ws_bytes(C.byref(d), dt))


def weight_cast(src, scale, dst, Cout, taps, Cin):
    call('l2s_weight_cast', ptr(src), ptr(scale), ptr(dst), Cout, taps, Cin, dt_of(dst), stream())


def weight_transpose(src, scale, dst, Cout, taps, Cin):
    call('l2s_weight_transpose', ptr(src), ptr(scale), ptr(dst), Cout, taps, Cin, dt_of(dst), stream())


def weight_transpose_batched(table_dev, n, total_tiles, dt):
    call('l2s_weight_transpose_batched', ptr(table_dev), n, total_tiles, dt, stream())


def colsum(a, rows, cols, lda, out, ws=None):
    """out[c] += sum_r a[r][c].  ws (32 * cols floats, one per concurrent call site): tall matrices are summed in row ranges."""
    call('l2s_colsum', ptr(a), rows, cols, lda, ptr(out), ptr(ws), 0 if ws is None else ws.numel(), dt_of(a), stream())


def stem_pack(w, pack=None):
    """conv1's weights [64][7][7][3] f32 -> the bf16 fragment order of l2s_stem_pool_bf16 (into `pack` when given: recorded tapes hold its address)"""
    if pack is None:
        pack = torch.empty(int(_lib.load().l2s_stem_pack_bytes()), dtype=torch.uint8, device=w.device)
    call('l2s_stem_pack', ptr(w), ptr(pack), stream())
    return pack


def stem_pool_bf16(img, pack, scale, bias, y, H, W, OH, OW, PH, PW):
    call('l2s_stem_pool_bf16', ptr(img), ptr(pack), ptr(scale), ptr(bias), ptr(y), H, W, OH, OW, PH, PW, stream())


def stem_conv(img, w, scale, bias, y, H, W, OH, OW):
    call('l2s_stem_conv', ptr(img), ptr(w), ptr(scale), ptr(bias), ptr(y), H, W, OH, OW, dt_of(y), stream())


def scale_mask(x, mask, relu_ref, out):
    call('l2s_scale_mask', ptr(x), ptr(mask), ptr(relu_ref), ptr(out), x.numel(), dt_of(x), stream())


def conv3x3_c3(img, w, bias, y, H, W):
    call('l2s_conv3x3_c3', ptr(img), ptr(w), ptr(bias), ptr(y), H, W, dt_of(y), stream())


def maxpool2x2_fwd(x, y, n_img, IH, IW, Cc):
    call('l2s_maxpool2x2_fwd', ptr(x), ptr(y), n_img, IH, IW, Cc, dt_of(x), stream())


def maxpool2x2_bwd(dy, x, dx, n_img, IH, IW, Cc, relu_out):
    call('l2s_maxpool2x2_bwd', ptr(dy), ptr(x), ptr(dx), n_img, IH, IW, Cc, int(relu_out), dt_of(x), stream())


def maxpool(x, y, IH, IW, Cc, OH, OW):
    call('l2s_maxpool3x3s2', ptr(x), ptr(y), IH, IW, Cc, OH, OW, dt_of(x), stream())


# ------------------------------------------------------------------ elementwise / pooling
def fill(t, v):
    call('l2s_fill_f32', ptr(t), float(v), t.numel(), stream())


def cast(src, dst):
    call('l2s_cast', ptr(src), dt_of(src), ptr(dst), dt_of(dst), src.numel(), stream())


def add3(a, b, c, dst):
    call('l2s_add3', ptr(a), ptr(b), ptr(c), ptr(dst), a.numel(), dt_of(a), stream())


def avgpool_fwd(x, y, n_img, hw, Cc):
    call('l2s_avgpool_fwd', ptr(x), ptr(y), n_img, hw, Cc, dt_of(x), stream())


def avgpool_bwd(dy, dx, addend, ref, n_img, hw, Cc):
    call('l2s_avgpool_bwd', ptr(dy), ptr(dx), ptr(addend), ptr(ref), n_img, hw, Cc, dt_of(dy), stream())


def adaptive_pool_fwd(x, pixmask, y, H, W, Cc, OH, OW, ldy):
    call('l2s_adaptive_pool_fwd', ptr(x), ptr(pixmask), ptr(y), H, W, Cc, OH, OW, ldy, dt_of(x), stream())


def adaptive_pool_bwd(dy, lddy, off_all, off_mask, pixmask, dx, ref, H, W, Cc, OH, OW):
    call('l2s_adaptive_pool_bwd', ptr(dy), lddy, off_all, off_mask, ptr(pixmask), ptr(dx), ptr(ref), H, W, Cc, OH, OW,
         dt_of(dx), stream())


def mask_downsample(mask_u8, out, H, W, h, w):
    call('l2s_mask_downsample', ptr(mask_u8), ptr(out), H, W, h, w, stream())


def counter_inc(counter):
    call('l2s_counter_inc', ptr(counter), stream())


def stamp(buf, slot):
    """diagnostics: element `slot` of the int64 tensor `buf` receives the device clock (100 MHz) when the current stream gets here"""
    call('l2s_stamp', buf.data_ptr() + 8 * int(slot), stream())


def dropout_mask(mask, p, seed_dev, salt):
    call('l2s_dropout_mask', ptr(mask), mask.numel(), float(p), ptr(seed_dev), int(salt), stream())


def random_keys(keys, seed_dev, salt):
    call('l2s_random_keys', ptr(keys), keys.numel(), ptr(seed_dev), int(salt), stream())


# ------------------------------------------------------------------ RoI path
def rpn_decode(heads, ldh, base_anchors, H, W, A, fs, im_h, im_w, prob, boxes, scores):
    call('l2s_rpn_decode', ptr(heads), ldh, ptr(base_anchors), H, W, A, fs, float(im_h), float(im_w), ptr(prob), ptr(boxes),
         ptr(scores), stream())


def sort_ws_ints(n):
    return int(_lib.load().l2s_sort_ws_ints(n))


def sort_topk(scores, boxes, n, k, ws, sboxes, sscores, sidx):
    call('l2s_sort_topk', ptr(scores), ptr(boxes), n, k, ptr(ws), ptr(sboxes), ptr(sscores), ptr(sidx), stream())


def nms_workspace_bytes(n):
    return int(_lib.load().l2s_nms_workspace_bytes(n))


def nms(sboxes, n, thresh, cmp_mode, max_keep, mask_ws, keep, num):
    call('l2s_nms', ptr(sboxes), n, float(thresh), cmp_mode, max_keep, ptr(mask_ws), ptr(keep), ptr(num), stream())


def gather_rois(sboxes, sscores, keep, num, max_keep, rois, roi_scores):
    call('l2s_gather_rois', ptr(sboxes), ptr(sscores), ptr(keep), ptr(num), max_keep, ptr(rois), ptr(roi_scores), stream())


def anchor_target_ws_ints(hwa):
    return int(_lib.load().l2s_anchor_target_ws_ints(hwa))


def anchor_target(gt, n_gt, base_anchors, H, W, A, fs, im_h, im_w, fg_keys, bg_keys, neg_ov, pos_ov, batch, fg_frac,
                  labels, targets, inw, outw, ws):
    call('l2s_anchor_target', ptr(gt), n_gt, ptr(base_anchors), H, W, A, fs, float(im_h), float(im_w), ptr(fg_keys),
         ptr(bg_keys), float(neg_ov), float(pos_ov), batch, float(fg_frac), ptr(labels), ptr(targets), ptr(inw),
         ptr(outw), ptr(ws), stream())


def proposal_target(rois, roi_scores, n_rois, n_max, gt, n_gt, gt_masks, im_h, im_w, fg_keys, bg_keys, bg_rand, R, fg_max, mask_slots,
                    fg_thresh, bg_hi, bg_lo, means4, stds4, inw4, ncls, ms, out_rois, labels, bt, bi, bo, mt, counts, ws):
    call('l2s_proposal_target', ptr(rois), ptr(roi_scores), ptr(n_rois), n_max, ptr(gt), n_gt, ptr(gt_masks), im_h, im_w,
         ptr(fg_keys), ptr(bg_keys), ptr(bg_rand), R, fg_max, mask_slots, float(fg_thresh), float(bg_hi), float(bg_lo), ptr(means4),
         ptr(stds4), ptr(inw4), ncls, ms, ptr(out_rois), ptr(labels), ptr(bt), ptr(bi), ptr(bo), ptr(mt), ptr(counts),
         ptr(ws), stream())


def roialign_fwd(feat, H, W, Cc, rois, R, P, sscale, out):
    call('l2s_roialign_fwd', ptr(feat), H, W, Cc, ptr(rois), R, P, float(sscale), ptr(out), dt_of(feat), stream())


def roialign_bwd(dout, H, W, Cc, rois, R, P, sscale, dfeat):
    call('l2s_roialign_bwd', ptr(dout), H, W, Cc, ptr(rois), R, P, float(sscale), ptr(dfeat), dt_of(dout), stream())


# ------------------------------------------------------------------ losses
def rpn_loss(heads, ldh, labels, targets, inw, outw, H, W, A, sigma, gscale, loss, dheads, ldd, atl_ws=None):
    """atl_ws: the anchor-target workspace (its sampled-anchor count is reused); otherwise the labels are counted."""
    if atl_ws is not None:
        cnt, cws = ptr(atl_ws) + 8, None
    else:
        cws = torch.zeros(1, dtype=torch.int32, device=heads.device)
        cnt, cws = None, ptr(cws)
    call('l2s_rpn_loss', ptr(heads), ldh, ptr(labels), ptr(targets), ptr(inw), ptr(outw), H, W, A, float(sigma), float(gscale),
         ptr(loss), ptr(dheads), ldd, dt_of(dheads), cnt, cws, stream())


def rcnn_loss(heads, ldh, labels, bt, bi, bo, R, ncls, gscale, loss, dheads, ldd):
    call('l2s_rcnn_loss', ptr(heads), ldh, ptr(labels), ptr(bt), ptr(bi), ptr(bo), R, ncls, float(gscale), ptr(loss),
         ptr(dheads), ldd, dt_of(dheads), stream())


def mask_loss(score, ldsc, labels, mt, num_fg, fg_max, ms2, gscale, loss, dscore):
    call('l2s_mask_loss', ptr(score), ldsc, ptr(labels), ptr(mt), ptr(num_fg), fg_max, ms2, float(gscale), ptr(loss),
         ptr(dscore), stream())


def total_loss(loss, cap_w):
    call('l2s_total_loss', ptr(loss), float(cap_w), stream())


def maskpred_bwd(dscore, labels, num_fg, fg_max, ms2, Cc, w, x, ref, dx, dw, db, ws=None):
    if ws is None:                                   # per-(RoI, pixel chunk) partial sums
        ws = torch.empty(int(_lib.load().l2s_maskpred_ws_floats(fg_max, Cc)), dtype=torch.float32, device=dx.device)
    call('l2s_maskpred_bwd', ptr(dscore), ptr(labels), ptr(num_fg), fg_max, ms2, Cc, ptr(w), ptr(x), ptr(ref), ptr(dx),
         ptr(dw), ptr(db), ptr(ws), dt_of(x), stream())


def maskpred_bwd_dx(dscore, labels, num_fg, fg_max, ms2, Cc, w, x, ref, dx, ws):
    """the first launch of maskpred_bwd: dx and the per-(RoI, pixel chunk) partial sums into ws"""
    call('l2s_maskpred_bwd_dx', ptr(dscore), ptr(labels), ptr(num_fg), fg_max, ms2, Cc, ptr(w), ptr(x), ptr(ref), ptr(dx), ptr(ws), dt_of(x), stream())


def maskpred_bwd_reduce(ws, labels, num_fg, fg_max, Cc, dw, db):
    """the second: dW / db from ws, per label in RoI order (on the current stream: the caller puts it on a weight-gradient stream)"""
    call('l2s_maskpred_bwd_reduce', ptr(ws), ptr(labels), ptr(num_fg), fg_max, Cc, ptr(dw), ptr(db), stream())


# ------------------------------------------------------------------ language side (fp32)
def linear_fwd(x, w, b, y, M, N, K, act=0, accumulate=False, ldx=None, ldy=None, ldw=None):
    call('l2s_linear_fwd', ptr(x), K if ldx is None else ldx, ptr(w), K if ldw is None else ldw, ptr(b), ptr(y),
         N if ldy is None else ldy, M, N, K, act, 1 if accumulate else 0, stream())


def linear_bwd_x(dy, w, dx, M, N, K, accumulate=False, lddy=None, lddx=None, mul=None, ws=None):
    """dx[M][K] (+)= (dy[M][N] . w[N][K]) (* mul); ws: float tensor of linear_bwd_x_ws_floats(M, N, K) for the split contraction"""
    call('l2s_linear_bwd_x', ptr(dy), N if lddy is None else lddy, ptr(w), ptr(dx), K if lddx is None else lddx, M, N, K,
         1 if accumulate else 0, ptr(mul), ptr(ws), 0 if ws is None else ws.numel(), stream())


def linear_bwd_x_ws_floats(M, N, K):
    return int(_lib.load().l2s_linear_bwd_x_ws_floats(M, N, K))


def linear_bwd_w(dy, x, dw, db, M, N, K, lddy=None, ldx=None):
    call('l2s_linear_bwd_w', ptr(dy), N if lddy is None else lddy, ptr(x), K if ldx is None else ldx, ptr(dw), ptr(db), M, N,
         K, stream())


def mask_relu_cast(x, mul, relu_ref, out):
    """x *= mul (optional); x = 0 where relu_ref <= 0; out = x in out's dtype (one launch)"""
    call('l2s_mask_relu_cast', ptr(x), ptr(mul), ptr(relu_ref), ptr(out), dt_of(out), x.numel(), stream())


def act_bwd(dy, y, act):
    call('l2s_act_bwd', ptr(dy), ptr(y), dy.numel(), act, stream())


def embed_fwd(table, ids, mask, out, T, D, relu):
    call('l2s_embed_fwd', ptr(table), ptr(ids), ptr(mask), ptr(out), T, D, 1 if relu else 0, stream())


def embed_bwd(dout, out, ids, mask, dtable, T, D, relu):
    call('l2s_embed_bwd', ptr(dout), ptr(out), ptr(ids), ptr(mask), ptr(dtable), T, D, 1 if relu else 0, stream())


def lstm_cell_fwd(gates, c_prev, c, h, act, Hh):
    call('l2s_lstm_cell_fwd', ptr(gates), ptr(c_prev), ptr(c), ptr(h), ptr(act), Hh, stream())


def lstm_cell_bwd(dh, dc_in, act, c_prev, c, dgates, dc_prev, Hh):
    call('l2s_lstm_cell_bwd', ptr(dh), ptr(dc_in), ptr(act), ptr(c_prev), ptr(c), ptr(dgates), ptr(dc_prev), Hh, stream())


def lstm_step_fwd(dirs, Hh):
    """dirs: list (1 or 2) of dicts with the l2s_lstm_fwd_dir fields (tensors)."""
    arr = (_lib.LstmFwdDir * len(dirs))()
    for i, d in enumerate(dirs):
        for k, _ in _lib.LstmFwdDir._fields_:
            setattr(arr[i], k, ptr(d[k]))
    call('l2s_lstm_step_fwd', arr, len(dirs), Hh, stream())


def lstm_step_bwd(dirs, Hh):
    arr = (_lib.LstmBwdDir * len(dirs))()
    for i, d in enumerate(dirs):
        for k, _ in _lib.LstmBwdDir._fields_:
            setattr(arr[i], k, ptr(d.get(k)))
    call('l2s_lstm_step_bwd', arr, len(dirs), Hh, stream())


def dynfilter_fwd(x, filt, r, y, resp, respk, H, W, Cc, gate=0):
    call('l2s_dynfilter_fwd', ptr(x), ptr(filt), ptr(r), ptr(y), ptr(resp), ptr(respk), H, W, Cc, dt_of(x), int(gate), stream())


def dynfilter_ws_floats(H, W, Cc):
    return int(_lib.load().l2s_dynfilter_ws_floats(H, W, Cc))


def dynfilter_bwd(dy, x, filt, r, resp, respk, dx, ref, dfilt, dr, dresp_ws, H, W, Cc, gate=0, dresp_extra=None):
    assert dresp_ws.numel() >= dynfilter_ws_floats(H, W, Cc)
    call('l2s_dynfilter_bwd', ptr(dy), ptr(x), ptr(filt), ptr(r), ptr(resp), ptr(respk), ptr(dx), ptr(ref), ptr(dfilt),
         ptr(dr), ptr(dresp_ws), H, W, Cc, dt_of(x), int(gate), ptr(dresp_extra), stream())


def dynfilter_bwd_finish(dresp_ws, respk, dfilt, dr, H, W, Cc):
    """the part of dynfilter_bwd that only the language-side backward needs (dfilt +=, dr +=); dynfilter_bwd was called with dfilt=None"""
    call('l2s_dynfilter_bwd_finish', ptr(dresp_ws), ptr(respk), ptr(dfilt), ptr(dr), H, W, Cc, stream())


def roipool_fwd(feat, H, W, Cc, rois, R, P, scale, out, argmax):
    call('l2s_roipool_fwd', ptr(feat), H, W, Cc, ptr(rois), R, P, float(scale), ptr(out), ptr(argmax), dt_of(feat), stream())


def cropalign_fwd(feat, H, W, Cc, rois, R, P, im_h, im_w, out):
    call('l2s_cropalign_fwd', ptr(feat), H, W, Cc, ptr(rois), R, P, float(im_h), float(im_w), ptr(out), dt_of(feat), stream())


def cropalign_bwd(dout, H, W, Cc, rois, R, P, im_h, im_w, dfeat):
    call('l2s_cropalign_bwd', ptr(dout), H, W, Cc, ptr(rois), R, P, float(im_h), float(im_w), ptr(dfeat), dt_of(dout), stream())


def roipool_bwd(dout, argmax, R, P, Cc, dfeat):
    call('l2s_roipool_bwd', ptr(dout), ptr(argmax), R, P, Cc, ptr(dfeat), dt_of(dout), stream())


# ------------------------------------------------------------------ input side (loaders)
def rle_from_string(s):
    """COCO compressed RLE string -> uint32 run lengths (host; maskApi.c:217-231)."""
    import numpy as np
    b = s.encode('ascii') if isinstance(s, str) else bytes(s)
    buf = np.empty(max(len(b), 1), np.uint32)
    m = _lib.load().l2s_rle_from_string(b, buf.ctypes.data, buf.size)
    if m < 0:
        raise ValueError('malformed run-length string')
    return buf[:m].copy()


def prep_geometry(h, w, target_size, max_size):
    """(im_scale, out_h, out_w) of prep_im_for_blob (host; blob.py:35-45)."""
    sc, oh, ow = C.c_double(), C.c_int(), C.c_int()
    call('l2s_prep_geometry', int(h), int(w), int(target_size), int(max_size), C.byref(sc), C.byref(oh), C.byref(ow))
    return sc.value, oh.value, ow.value


def prep_image(img_u8, means, scale, out):
    """img_u8 uint8 [h][w][3] BGR (device) -> out float32 [oh][ow][3]."""
    h, w = img_u8.shape[0], img_u8.shape[1]
    call('l2s_prep_image', ptr(img_u8), h, w, float(means[0]), float(means[1]), float(means[2]), float(scale),
         out.shape[0], out.shape[1], ptr(out), stream())


def rle_ws_words(total_counts, oh, ow):
    return int(_lib.load().l2s_rle_ws_words(total_counts, oh, ow))


def rle_to_mask(cnts, offs, n, total, h, w, ws, out):
    """cnts uint32 / offs int32 (device) -> out uint8 [oh][ow]."""
    call('l2s_rle_to_mask', ptr(cnts), ptr(offs), n, total, h, w, out.shape[0], out.shape[1], ptr(ws), ptr(out), stream())


def rcnn_predict(heads, ldh, R, ncls, stds4, means4, cls_prob, bbox_pred):
    call('l2s_rcnn_predict', ptr(heads), ldh, R, ncls, ptr(stds4), ptr(means4), ptr(cls_prob), ptr(bbox_pred), stream())


def mask_prob(score, ldsc, ncls, labels, ms2, n_elem, out):
    call('l2s_mask_prob', ptr(score), ldsc, ncls, ptr(labels), ms2, n_elem, ptr(out), stream())


def response_loss(resp, gt_mask_u8, mask_h, mask_w, H, W, gscale, loss, dresp):
    call('l2s_response_loss', ptr(resp), ptr(gt_mask_u8), mask_h, mask_w, H, W, float(gscale), ptr(loss), ptr(dresp), stream())


def cap_attention_fwd(patt, att, att_h, aw, ab, L, D, tanh_ws, weight, att_res):
    call('l2s_cap_attention_fwd', ptr(patt), ptr(att), ptr(att_h), ptr(aw), ptr(ab), L, D, ptr(tanh_ws), ptr(weight),
         ptr(att_res), stream())


def cap_attention_bwd(datt_res, att, tanh_ws, weight, aw, L, D, dpatt, datt, datt_h, daw, dab):
    call('l2s_cap_attention_bwd', ptr(datt_res), ptr(att), ptr(tanh_ws), ptr(weight), ptr(aw), L, D, ptr(dpatt), ptr(datt),
         ptr(datt_h), ptr(daw), ptr(dab), stream())


def cap_gates_fwd(sums, a2c, c_prev, c, h, save, R):
    call('l2s_cap_gates_fwd', ptr(sums), ptr(a2c), ptr(c_prev), ptr(c), ptr(h), ptr(save), R, stream())


def cap_gates_bwd(dh, dc_in, save, c_prev, dsums, da2c, dc_prev, R, dh2=None):
    call('l2s_cap_gates_bwd', ptr(dh), ptr(dh2), ptr(dc_in), ptr(save), ptr(c_prev), ptr(dsums), ptr(da2c), ptr(dc_prev), R, stream())


def linear2_fwd(x, K, w1, b1, y1, N1, acc1, w2, b2, y2, N2, acc2):
    call('l2s_linear2_fwd', ptr(x), K, ptr(w1), ptr(b1), ptr(y1), N1, int(acc1), ptr(w2), ptr(b2), ptr(y2), N2, int(acc2), stream())


def linear_sum2_fwd(x1, w1, K1, x2, w2, K2, y, N, accumulate=False):
    call('l2s_linear_sum2_fwd', ptr(x1), ptr(w1), K1, ptr(x2), ptr(w2), K2, ptr(y), N, int(accumulate), stream())


def cap_a2c_gates_fwd(att_res, w_a2c, b_a2c, K, sums, c_prev, c, h, save, R):
    call('l2s_cap_a2c_gates_fwd', ptr(att_res), ptr(w_a2c), ptr(b_a2c), K, ptr(sums), ptr(c_prev), ptr(c), ptr(h), ptr(save), R, stream())


def cap_att_dots_fwd(patt, att_h, aw, ab, L, D, tanh_ws, dots):
    call('l2s_cap_att_dots_fwd', ptr(patt), ptr(att_h), ptr(aw), ptr(ab), L, D, ptr(tanh_ws), ptr(dots), stream())


def cap_apply_gates_fwd(P, dots, b_a2c, sums, c_prev, c, h, save, weight, L, R):
    call('l2s_cap_apply_gates_fwd', ptr(P), ptr(dots), ptr(b_a2c), ptr(sums), ptr(c_prev), ptr(c), ptr(h), ptr(save), ptr(weight), L, R, stream())


def bottleneck64_fwd(a, x, w2, b2, w3, b3, y, H, W, wd=None, bd=None, w1n=None, b1n=None, a_next=None):
    """csrc/bottleneck_fused.hip: a frozen 64-plane bottleneck behind its conv1 (+ the next block's conv1) in one launch, bf16"""
    d = _lib.Bottleneck64Desc(ptr(a), ptr(x), ptr(w2), ptr(w3), ptr(wd), ptr(w1n), ptr(b2), ptr(b3), ptr(bd), ptr(b1n), ptr(y), ptr(a_next), H, W, x.shape[-1])
    call('l2s_bottleneck64_fwd', C.byref(d), stream())


def cap_recur_supported(S, R, AH, L):
    return bool(_lib.load().l2s_cap_recur_supported(int(S), int(R), int(AH), int(L)))


def cap_recur_state(backward, device='cuda'):
    """the exchange state of one direction of the resident recurrence: zeroed ONCE here, then owned by the kernels (launch count, give-up flag, granules)"""
    n = int(_lib.load().l2s_cap_recur_state_bytes(int(backward)))
    return torch.zeros((n + 3) // 4, dtype=torch.int32, device=device)


def cap_recur_fwd(w_h2h, b_h2h, w_h2att, b_h2att, patt, aw, ab, P, b_a2c, sums, hs, cs, save, tanh_ws, wgt, state, S, R, AH, L):
    a = _lib.CapRecurFwdArgs(ptr(w_h2h), ptr(b_h2h), ptr(w_h2att), ptr(b_h2att), ptr(patt), ptr(aw), ptr(ab), ptr(P), ptr(b_a2c), ptr(sums), ptr(hs), ptr(cs),
                             ptr(save), ptr(tanh_ws), ptr(wgt), ptr(state), S, R, AH, L)
    call('l2s_cap_recur_fwd', C.byref(a), stream())


def cap_recur_bwd(w_h2h, w_h2att, P, aw, save, cs, wgt, tanh_ws, dho, dsums, da2c, ddot, datt_h, state, S, R, AH, L):
    a = _lib.CapRecurBwdArgs(ptr(w_h2h), ptr(w_h2att), ptr(P), ptr(aw), ptr(save), ptr(cs), ptr(wgt), ptr(tanh_ws), ptr(dho), ptr(dsums), ptr(da2c), ptr(ddot),
                             ptr(datt_h), ptr(state), S, R, AH, L, ddot.stride(0), datt_h.stride(0))
    call('l2s_cap_recur_bwd', C.byref(a), stream())


def cap_gates_bwd_dw(dh, dc_in, save, c_prev, P, dsums, da2c, dc_prev, dweight, L, R, dh2=None):
    call('l2s_cap_gates_bwd_dw', ptr(dh), ptr(dh2), ptr(dc_in), ptr(save), ptr(c_prev), ptr(P), ptr(dsums), ptr(da2c), ptr(dc_prev), ptr(dweight),
         L, R, stream())


def cap_attention_bwd_step2(dweight, tanh_ws, weight, aw, L, D, ddot, datt_h):
    call('l2s_cap_attention_bwd_step2', ptr(dweight), ptr(tanh_ws), ptr(weight), ptr(aw), L, D, ptr(ddot), ptr(datt_h), stream())


def cap_attention_bwd_step(datt_res, att, tanh_ws, weight, aw, L, D, ddot, datt_h):
    call('l2s_cap_attention_bwd_step', ptr(datt_res), ptr(att), ptr(tanh_ws), ptr(weight), ptr(aw), L, D, ptr(ddot), ptr(datt_h), stream())


def cap_attention_bwd_batched(ddot, weight, datt_res, ldr, tanh_ws, aw, S, L, D, dpatt, datt, daw, dab):
    call('l2s_cap_attention_bwd_batched', ptr(ddot), ptr(weight), ptr(datt_res), ldr, ptr(tanh_ws), ptr(aw), S, L, D, ptr(dpatt), ptr(datt),
         ptr(daw), ptr(dab), stream())


def logsoftmax_nll(logits, target, mask, S, V1, gscale, loss_slot, dlogits, logprobs=None):
    call('l2s_logsoftmax_nll', ptr(logits), ptr(target), ptr(mask), S, V1, float(gscale), ptr(loss_slot), ptr(dlogits),
         ptr(logprobs), stream())


def sgd_momentum(param, grad, mom, segs_dev, nseg, rowscale, lr, momentum, wd, gscale=1.0, shadow=None, clear_grad=False):
    call('l2s_sgd_momentum', ptr(param), ptr(grad), ptr(mom), ptr(segs_dev), nseg, ptr(rowscale), float(lr), float(momentum),
         float(wd), float(gscale), ptr(shadow), dt_of(shadow) if shadow is not None else 0, 1 if clear_grad else 0, stream())


def sgd_momentum_range(param, grad, mom, segs_dev, nseg, rowscale, lr, momentum, wd, gscale, shadow, flags, lo, hi, chunk_lo, chunk_hi):
    """the update on the elements [lo, hi) only (flags: 1 = clear the gradients consumed, 2 = shadow rewrite only)"""
    call('l2s_sgd_momentum_range', ptr(param), ptr(grad), ptr(mom), ptr(segs_dev), nseg, ptr(rowscale), float(lr), float(momentum),
         float(wd), float(gscale), ptr(shadow), dt_of(shadow) if shadow is not None else 0, int(flags), int(lo), int(hi), int(chunk_lo), int(chunk_hi), stream())


def sgd_momentum_range_g16(param, grad_bf16, grad_lo, mom, segs_dev, nseg, rowscale, lr, momentum, wd, gscale, shadow, lo, hi, chunk_lo, chunk_hi):
    """the ranged update with the gradients of [lo, hi) read from `grad_bf16[o - grad_lo]` (a reduce-scattered bf16 shard)"""
    call('l2s_sgd_momentum_range_g16', ptr(param), ptr(grad_bf16), int(grad_lo), ptr(mom), ptr(segs_dev), nseg, ptr(rowscale), float(lr), float(momentum),
         float(wd), float(gscale), ptr(shadow), dt_of(shadow) if shadow is not None else 0, int(lo), int(hi), int(chunk_lo), int(chunk_hi), stream())


def sgd_chunk():
    return int(_lib.load().l2s_sgd_chunk())


def add_f32(a, b, out):
    call('l2s_add_f32', ptr(a), ptr(b), ptr(out), a.numel(), stream())


def mul(a, b, out):
    call('l2s_mul_f32', ptr(a), ptr(b), ptr(out), a.numel(), stream())
