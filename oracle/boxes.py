"""Integer / box arithmetic of the RoI path, numpy restatement (test oracle).

Shorthand for citations (relative to /root/reference/pyutils/mask-faster-rcnn/lib):
  GA  = layer_utils/generate_anchors.py     SN = layer_utils/snippets.py
  BT  = model/bbox_transform.py             BB = utils/bbox.py
  PL  = layer_utils/proposal_layer.py       ATL = layer_utils/anchor_target_layer.py
  PTL = layer_utils/proposal_target_layer.py
  NMSC = nms/src/nms.c   NMSK = nms/src/cuda/nms_kernel.cu   NMSH = nms/src/nms_cuda.c
"""
import numpy as np

# ----------------------------------------------------------------------------
# anchors  (GA:41-103, SN:13-29)
# ----------------------------------------------------------------------------

def _whctrs(a):
    w = a[2] - a[0] + 1
    h = a[3] - a[1] + 1
    return w, h, a[0] + 0.5 * (w - 1), a[1] + 0.5 * (h - 1)


def _mk(ws, hs, xc, yc):
    ws = np.asarray(ws, dtype=np.float64)[:, None]
    hs = np.asarray(hs, dtype=np.float64)[:, None]
    return np.hstack((xc - 0.5 * (ws - 1), yc - 0.5 * (hs - 1),
                      xc + 0.5 * (ws - 1), yc + 0.5 * (hs - 1)))


def generate_anchors(base_size=16, ratios=(0.5, 1, 2), scales=(8, 16, 32)):
    """GA:41-52: ratio-major, scale-minor base anchors around (0,0,15,15)."""
    ratios = np.asarray(ratios, dtype=np.float64)
    scales = np.asarray(scales, dtype=np.float64)
    base = np.array([1, 1, base_size, base_size], dtype=np.float64) - 1
    w, h, xc, yc = _whctrs(base)
    size_ratios = (w * h) / ratios
    ws = np.round(np.sqrt(size_ratios))           # GA:84 uses np.round (half-to-even)
    hs = np.round(ws * ratios)
    ratio_anchors = _mk(ws, hs, xc, yc)
    out = []
    for i in range(ratio_anchors.shape[0]):
        w, h, xc, yc = _whctrs(ratio_anchors[i])
        out.append(_mk(w * scales, h * scales, xc, yc))
    return np.vstack(out)


def generate_anchors_pre(height, width, feat_stride=16, anchor_scales=(8, 16, 32),
                         anchor_ratios=(0.5, 1, 2)):
    """SN:13-29: shift the A base anchors over the (H,W) grid, row order (y, x, a)."""
    anchors = generate_anchors(ratios=anchor_ratios, scales=anchor_scales)
    A = anchors.shape[0]
    sx, sy = np.meshgrid(np.arange(width) * feat_stride, np.arange(height) * feat_stride)
    shifts = np.vstack((sx.ravel(), sy.ravel(), sx.ravel(), sy.ravel())).T
    K = shifts.shape[0]
    allanc = anchors.reshape(1, A, 4) + shifts.reshape(K, 1, 4)
    return allanc.reshape(K * A, 4).astype(np.float32), K * A

# ----------------------------------------------------------------------------
# box codec (BT:14-80) and IoU (BB:4-29); fp32 like the reference's torch tensors
# ----------------------------------------------------------------------------

def bbox_transform(ex, gt):
    ex = ex.astype(np.float32); gt = gt.astype(np.float32)
    one = np.float32(1.0); half = np.float32(0.5)
    ew = ex[:, 2] - ex[:, 0] + one
    eh = ex[:, 3] - ex[:, 1] + one
    ecx = ex[:, 0] + half * ew
    ecy = ex[:, 1] + half * eh
    gw = gt[:, 2] - gt[:, 0] + one
    gh = gt[:, 3] - gt[:, 1] + one
    gcx = gt[:, 0] + half * gw
    gcy = gt[:, 1] + half * gh
    return np.stack(((gcx - ecx) / ew, (gcy - ecy) / eh,
                     np.log(gw / ew), np.log(gh / eh)), 1).astype(np.float32)


def bbox_transform_inv(boxes, deltas):
    """BT:36-62. corners = ctr -/+ 0.5*pred_w (no -1)."""
    boxes = boxes.astype(np.float32); deltas = deltas.astype(np.float32)
    one = np.float32(1.0); half = np.float32(0.5)
    w = boxes[:, 2] - boxes[:, 0] + one
    h = boxes[:, 3] - boxes[:, 1] + one
    cx = boxes[:, 0] + half * w
    cy = boxes[:, 1] + half * h
    pcx = deltas[:, 0] * w + cx
    pcy = deltas[:, 1] * h + cy
    pw = np.exp(deltas[:, 2]) * w
    ph = np.exp(deltas[:, 3]) * h
    return np.stack((pcx - half * pw, pcy - half * ph,
                     pcx + half * pw, pcy + half * ph), 1).astype(np.float32)


def clip_boxes(b, im_shape):
    """BT:65-80. im_shape = (H, W)."""
    b = b.copy()
    b[:, 0] = np.clip(b[:, 0], 0, im_shape[1] - 1)
    b[:, 1] = np.clip(b[:, 1], 0, im_shape[0] - 1)
    b[:, 2] = np.clip(b[:, 2], 0, im_shape[1] - 1)
    b[:, 3] = np.clip(b[:, 3], 0, im_shape[0] - 1)
    return b


def bbox_overlaps(boxes, query, dtype=np.float32):
    """BB:4-29, (N,K) IoU with the +1 pixel convention."""
    b = boxes.astype(dtype); q = query.astype(dtype)
    one = dtype(1)
    ba = (b[:, 2] - b[:, 0] + one) * (b[:, 3] - b[:, 1] + one)
    qa = (q[:, 2] - q[:, 0] + one) * (q[:, 3] - q[:, 1] + one)
    iw = np.clip(np.minimum(b[:, 2:3], q[:, 2:3].T) - np.maximum(b[:, 0:1], q[:, 0:1].T) + one, 0, None)
    ih = np.clip(np.minimum(b[:, 3:4], q[:, 3:4].T) - np.maximum(b[:, 1:2], q[:, 1:2].T) + one, 0, None)
    ua = ba[:, None] + qa[None, :] - iw * ih
    return (iw * ih / ua).astype(dtype)

# ----------------------------------------------------------------------------
# NMS (NMSC:35-63 `>=` on CPU ; NMSK:56-66 + NMSH:47-58 `>` on GPU)
# ----------------------------------------------------------------------------

def stable_desc_order(scores):
    """Descending sort, ties broken by lower index (deterministic restatement of
    `scores.sort(0, descending=True)`, nms/pth_nms.py:17,34 / PL:49)."""
    return np.argsort(-scores.astype(np.float32), kind='stable')


def nms(dets, thresh, cmp_mode='ge'):
    """Greedy NMS over dets (n,5)=[x1,y1,x2,y2,score]; returns keep indices into
    `dets`, in descending-score order.  cmp_mode 'ge' = cpu_nms, 'gt' = gpu_nms."""
    dets = dets.astype(np.float32)
    n = dets.shape[0]
    one = np.float32(1.0)
    x1, y1, x2, y2 = dets[:, 0], dets[:, 1], dets[:, 2], dets[:, 3]
    areas = (x2 - x1 + one) * (y2 - y1 + one)
    order = stable_desc_order(dets[:, 4])
    sup = np.zeros(n, dtype=bool)
    keep = []
    thr = np.float32(thresh)
    for _i in range(n):
        i = order[_i]
        if sup[i]:
            continue
        keep.append(i)
        rest = order[_i + 1:]
        xx1 = np.maximum(x1[i], x1[rest]); yy1 = np.maximum(y1[i], y1[rest])
        xx2 = np.minimum(x2[i], x2[rest]); yy2 = np.minimum(y2[i], y2[rest])
        w = np.maximum(np.float32(0), xx2 - xx1 + one)
        h = np.maximum(np.float32(0), yy2 - yy1 + one)
        inter = w * h
        ovr = inter / (areas[i] + areas[rest] - inter)
        hit = (ovr >= thr) if cmp_mode == 'ge' else (ovr > thr)
        sup[rest[hit]] = True
    return np.asarray(keep, dtype=np.int64)

# ----------------------------------------------------------------------------
# proposal_layer (PL:19-68)
# ----------------------------------------------------------------------------

def proposal_layer(rpn_cls_prob, rpn_bbox_pred, im_info, anchors, num_anchors,
                   pre_nms_topN, post_nms_topN, nms_thresh, cmp_mode='ge'):
    """rpn_cls_prob (1,H,W,2A), rpn_bbox_pred (1,H,W,4A) -> rois (k,5), scores (k,), order, keep."""
    A = num_anchors
    scores = rpn_cls_prob[:, :, :, A:].reshape(-1).astype(np.float32)      # PL:42
    deltas = rpn_bbox_pred.reshape(-1, 4)
    props = clip_boxes(bbox_transform_inv(anchors, deltas), im_info[:2])   # PL:45-46
    order = stable_desc_order(scores)
    if pre_nms_topN > 0:
        order = order[:pre_nms_topN]
    props = props[order]; sc = scores[order]
    keep = nms(np.hstack((props, sc[:, None])), nms_thresh, cmp_mode)
    if post_nms_topN > 0:
        keep = keep[:post_nms_topN]
    props = props[keep]; sc = sc[keep]
    rois = np.hstack((np.zeros((props.shape[0], 1), np.float32), props)).astype(np.float32)
    return rois, sc, order, keep

def proposal_top_layer(rpn_cls_prob, rpn_bbox_pred, im_info, anchors, num_anchors, rpn_top_n, rng=None):
    """layer_utils/proposal_top_layer.py:18-67 (TEST.MODE == 'top'): the rpn_top_n best-scoring anchors, decoded and clipped, no NMS.
    Fewer anchors than rpn_top_n: sampled WITH replacement (npr.choice, :46-49) — `rng` supplies the draw."""
    A = num_anchors
    scores = rpn_cls_prob[:, :, :, A:].reshape(-1).astype(np.float32)
    deltas = rpn_bbox_pred.reshape(-1, 4)
    if scores.shape[0] < rpn_top_n:
        top = (rng or np.random).choice(scores.shape[0], size=rpn_top_n, replace=True)
    else:
        top = stable_desc_order(scores)[:rpn_top_n]
    props = clip_boxes(bbox_transform_inv(anchors[top], deltas[top]), im_info[:2])
    rois = np.hstack((np.zeros((props.shape[0], 1), np.float32), props)).astype(np.float32)
    return rois, scores[top]

# ----------------------------------------------------------------------------
# sampling helper: "keep the k smallest keys" == npr.choice(..., replace=False)
# when key[cand[perm[j]]] = j (numpy legacy choice = permutation(n)[:k]).
# ----------------------------------------------------------------------------

def keys_from_rng(cand, rng, n_total, fill=None):
    """Draw rng.permutation(len(cand)) the way npr.choice(replace=False) does and
    turn it into per-element priority keys (uint32).  Elements that are not
    candidates get key 0xFFFFFFFF."""
    keys = np.full(n_total, 0xFFFFFFFF, dtype=np.uint32)
    if len(cand):
        perm = rng.permutation(len(cand))
        keys[np.asarray(cand)[perm]] = np.arange(len(cand), dtype=np.uint32)
    return keys

# ----------------------------------------------------------------------------
# anchor_target_layer (ATL:19-153)
# ----------------------------------------------------------------------------

def anchor_target_layer(H, W, gt_boxes, im_info, all_anchors, A, cfgt, fg_keys=None, bg_keys=None, rng=None):
    """Returns rpn_labels (1,1,A*H,W) float32, bbox_targets/inside/outside (1,H,W,4A).
    Sampling: either explicit uint32 priority keys per anchor (fg_keys / bg_keys; the
    D = count - quota candidates with the SMALLEST keys are disabled, = ATL:91-101 with
    key[cand[perm[j]]] = j), or `rng` (np.random.RandomState) to draw them like npr.choice."""
    total = all_anchors.shape[0]
    inside = np.where((all_anchors[:, 0] >= 0) & (all_anchors[:, 1] >= 0) &
                      (all_anchors[:, 2] < im_info[1]) & (all_anchors[:, 3] < im_info[0]))[0]
    anchors = all_anchors[inside]
    labels = np.full(len(inside), -1, dtype=np.float32)
    ov = bbox_overlaps(anchors, gt_boxes[:, :4], dtype=np.float64)           # ATL:62-64 float64
    argmax = ov.argmax(1)
    maxov = ov[np.arange(len(inside)), argmax]
    gt_max = ov.max(0)
    gt_argmax = np.where(ov == gt_max)[0]                                    # all ties, ATL:70
    labels[maxov < cfgt['RPN_NEGATIVE_OVERLAP']] = 0
    labels[gt_argmax] = 1
    labels[maxov >= cfgt['RPN_POSITIVE_OVERLAP']] = 1
    num_fg = int(cfgt['RPN_FG_FRACTION'] * cfgt['RPN_BATCHSIZE'])
    fg = np.where(labels == 1)[0]
    if len(fg) > num_fg:
        D = len(fg) - num_fg
        if fg_keys is None:
            dis = rng.choice(fg, size=D, replace=False)
        else:
            k = fg_keys[inside[fg]]
            dis = fg[np.argsort(k, kind='stable')[:D]]
        labels[dis] = -1
    num_bg = cfgt['RPN_BATCHSIZE'] - int(np.sum(labels == 1))
    bg = np.where(labels == 0)[0]
    if len(bg) > num_bg:
        D = len(bg) - num_bg
        if bg_keys is None:
            dis = rng.choice(bg, size=D, replace=False)
        else:
            k = bg_keys[inside[bg]]
            dis = bg[np.argsort(k, kind='stable')[:D]]
        labels[dis] = -1
    tgt = bbox_transform(anchors, gt_boxes[argmax, :4])
    inw = np.zeros((len(inside), 4), np.float32)
    inw[labels == 1] = 1.0
    outw = np.zeros((len(inside), 4), np.float32)
    nex = np.sum(labels >= 0)
    outw[labels == 1] = 1.0 / nex
    outw[labels == 0] = 1.0 / nex

    def unmap(d, fill):
        shp = (total,) + d.shape[1:]
        r = np.full(shp, fill, np.float32)
        r[inside] = d
        return r
    labels = unmap(labels, -1); tgt = unmap(tgt, 0); inw = unmap(inw, 0); outw = unmap(outw, 0)
    labels = labels.reshape(1, H, W, A).transpose(0, 3, 1, 2).reshape(1, 1, A * H, W)
    return (labels, tgt.reshape(1, H, W, A * 4), inw.reshape(1, H, W, A * 4),
            outw.reshape(1, H, W, A * 4))

# ----------------------------------------------------------------------------
# scipy<=1.2 misc.imresize(uint8, (h,w), 'nearest') == PIL NEAREST on uint8.
# Closed form verified against PIL 12.2 for every width 1..1000 -> 14:
# xo = 0.5*s, idx_k = int(xo), xo += s with s = in/out in float64.
# ----------------------------------------------------------------------------

def nearest_index(n_in, n_out):
    s = n_in / float(n_out)
    xo = 0.5 * s
    idx = []
    for _ in range(n_out):
        idx.append(min(int(xo), n_in - 1))
        xo += s
    return np.asarray(idx, dtype=np.int64)


def imresize_nearest_u8(a, size):
    return a[nearest_index(a.shape[0], size[0])][:, nearest_index(a.shape[1], size[1])]

# ----------------------------------------------------------------------------
# proposal_target_layer (PTL:22-204) with torch-0.3 ByteTensor semantics at :146
# ----------------------------------------------------------------------------

def proposal_target_layer(rois, scores, gt_boxes, gt_masks, num_classes, cfgt, mask_size=14,
                          fg_keys=None, bg_keys=None, rng=None, bg_rand=None):
    """rois (N,5), gt_boxes (M,5), gt_masks uint8 (M,H,W).
    Sampling without replacement = the k smallest keys, emitted in key order
    (= fg_inds[npr.choice(n, k, replace=False)], PTL:151,154).  With replacement
    (too few bg, PTL:153): index = bg_rand[j] % n (explicit) or rng.randint."""
    rois_per_image = int(cfgt['BATCH_SIZE'])
    fg_per_image = int(round(cfgt['FG_FRACTION'] * rois_per_image))
    all_rois = rois.astype(np.float32); all_scores = scores.astype(np.float32)
    while True:
        ov = bbox_overlaps(all_rois[:, 1:5], gt_boxes[:, :4])
        gt_assign = ov.argmax(1)
        maxov = ov[np.arange(ov.shape[0]), gt_assign]
        labels_all = gt_boxes[gt_assign, 4]
        fg_inds = np.where(maxov >= cfgt['FG_THRESH'])[0]
        bg_inds = np.where((maxov < cfgt['BG_THRESH_HI']) & (maxov >= cfgt['BG_THRESH_LO']))[0]
        if len(fg_inds) == 0:                      # PTL:159-167: append GT boxes and retry
            add = np.hstack((np.zeros((gt_boxes.shape[0], 1), np.float32), gt_boxes[:, :4]))
            all_rois = np.vstack((all_rois, add)).astype(np.float32)
            all_scores = np.concatenate((all_scores, np.zeros(gt_boxes.shape[0], np.float32)))
            fg_keys = None if fg_keys is None else np.concatenate((fg_keys, np.zeros(gt_boxes.shape[0], np.uint32)))
            bg_keys = None if bg_keys is None else np.concatenate((bg_keys, np.full(gt_boxes.shape[0], 0xFFFFFFFF, np.uint32)))
            continue
        break

    def pick(cand, k, keys):
        if keys is None:
            return cand[rng.choice(np.arange(len(cand)), size=int(k), replace=False)]
        o = np.argsort(keys[cand], kind='stable')[:int(k)]
        return cand[o]

    if len(bg_inds) > 0:
        nfg = min(fg_per_image, len(fg_inds))
        fg_sel = pick(fg_inds, nfg, fg_keys)
        nbg = rois_per_image - nfg
        if len(bg_inds) < nbg:
            if bg_rand is None:
                bg_sel = bg_inds[rng.choice(np.arange(len(bg_inds)), size=int(nbg), replace=True)]
            else:
                bg_sel = bg_inds[np.asarray(bg_rand[:nbg], dtype=np.int64) % len(bg_inds)]
        else:
            bg_sel = pick(bg_inds, nbg, bg_keys)
    else:                                           # PTL:155-158: only fg
        nfg = rois_per_image
        if len(fg_inds) < rois_per_image:
            if bg_rand is None:
                fg_sel = fg_inds[rng.choice(np.arange(len(fg_inds)), size=rois_per_image, replace=True)]
            else:
                fg_sel = fg_inds[np.asarray(bg_rand[:rois_per_image], dtype=np.int64) % len(fg_inds)]
        else:
            fg_sel = pick(fg_inds, rois_per_image, fg_keys)
        bg_sel = np.zeros(0, np.int64)
    keep = np.concatenate((fg_sel, bg_sel)).astype(np.int64)
    labels = labels_all[keep].astype(np.float32).copy()
    labels[int(nfg):] = 0
    out_rois = all_rois[keep]
    out_scores = all_scores[keep]
    t = bbox_transform(out_rois[:, 1:5], gt_boxes[gt_assign[keep], :4])
    t = (t - np.asarray(cfgt['BBOX_NORMALIZE_MEANS'], np.float32)) / np.asarray(cfgt['BBOX_NORMALIZE_STDS'], np.float32)
    bt = np.zeros((len(keep), 4 * num_classes), np.float32)
    bi = np.zeros_like(bt)
    for r in np.where(labels > 0)[0]:
        c = int(labels[r])
        bt[r, 4 * c:4 * c + 4] = t[r]
        bi[r, 4 * c:4 * c + 4] = np.asarray(cfgt['BBOX_INSIDE_WEIGHTS'], np.float32)
    bo = (bi > 0).astype(np.float32)
    # mask targets: PTL:193-201 iterates over the *selected* fg_inds
    mt = np.zeros((len(fg_sel), mask_size, mask_size), np.float32)
    for mix, i in enumerate(fg_sel):
        roi = all_rois[i]
        crop = gt_masks[gt_assign[i], int(roi[2]):int(roi[4]) + 1, int(roi[1]):int(roi[3]) + 1]
        mt[mix] = imresize_nearest_u8(crop, (mask_size, mask_size)).astype(np.float32)
    return (out_rois, out_scores, labels.reshape(-1, 1), bt, bi, bo, mt, keep)


# ----------------------------------------------------------------------------
# RoI max pooling (POOLING_MODE == 'pool'): restates the reference's CUDA kernels
# layer_utils/roi_pooling/src/cuda/roi_pooling_kernel.cu:15-70 (forward, argmax) and :104-180 (backward)
# ----------------------------------------------------------------------------

def _c_round(x):
    """C round(): half away from zero, on the float32 product as the kernel computes it."""
    x = np.float32(x)
    return int(np.floor(x + np.float32(0.5))) if x >= 0 else -int(np.floor(-x + np.float32(0.5)))


def roi_pool_fwd(feat_chw, rois, P, spatial_scale):
    """feat (C,H,W) f32, rois (R,5) -> out (R,C,P,P) f32, argmax (R,C,P,P) int32 (index h*W+w, -1 if empty)."""
    C, H, W = feat_chw.shape
    R = rois.shape[0]
    out = np.zeros((R, C, P, P), np.float32)
    arg = np.full((R, C, P, P), -1, np.int32)
    for n in range(R):
        sw, sh = _c_round(rois[n, 1] * np.float32(spatial_scale)), _c_round(rois[n, 2] * np.float32(spatial_scale))
        ew, eh = _c_round(rois[n, 3] * np.float32(spatial_scale)), _c_round(rois[n, 4] * np.float32(spatial_scale))
        rw, rh = max(ew - sw + 1, 1), max(eh - sh + 1, 1)
        bh, bw = np.float32(rh) / np.float32(P), np.float32(rw) / np.float32(P)
        for ph in range(P):
            for pw in range(P):
                hs = int(np.floor(np.float32(ph) * bh)); ws = int(np.floor(np.float32(pw) * bw))
                he = int(np.ceil(np.float32(ph + 1) * bh)); we = int(np.ceil(np.float32(pw + 1) * bw))
                hs = min(max(hs + sh, 0), H); he = min(max(he + sh, 0), H)
                ws = min(max(ws + sw, 0), W); we = min(max(we + sw, 0), W)
                if he <= hs or we <= ws:
                    continue                                  # empty bin: 0, argmax -1
                win = feat_chw[:, hs:he, ws:we].reshape(C, -1)
                k = win.argmax(1)                             # first maximum in (h, w) scan order, like the strict '>' loop
                out[n, :, ph, pw] = win[np.arange(C), k]
                arg[n, :, ph, pw] = (hs + k // (we - ws)) * W + (ws + k % (we - ws))
    return out, arg


def roi_pool_bwd(dout, arg, H, W):
    """dout (R,C,P,P), argmax -> dfeat (C,H,W): every pooled element sends its gradient to its argmax."""
    R, C = dout.shape[:2]
    df = np.zeros((C, H * W), np.float32)
    for c in range(C):
        a = arg[:, c].reshape(-1); g = dout[:, c].reshape(-1)
        m = a >= 0
        np.add.at(df[c], a[m], g[m])
    return df.reshape(C, H, W)
