"""CPU oracle of the input side of the train step — TEST INFRASTRUCTURE ONLY: imported by tests/,
never by the product (lang2seg_amd/loaders calls the HIP kernels of csrc/data.hip through the C ABI).

Restates, in numpy:
  rle_from_string / rle_to_string / rle_encode / rle_decode   pyutils/refer/external/maskApi.c:32-47,203-231
  ref_mask          lib/loaders/cycle_loader.py:198-210 (mask.decode -> sum over segments > 0 -> imresize 'nearest')
  prep_im_for_blob  pyutils/mask-faster-rcnn/lib/utils/blob.py:32-47 with cv2.resize(INTER_LINEAR) written out
  get_batch         lib/loaders/cycle_loader.py:143-357 (the blobs dict), gt_mrcn_loader.py:633-741 (getTestBatch)

Pinning: the run-length functions are checked against the REFERENCE's own maskApi.c compiled from /root/reference
(oracle/_ref/libmaskapi.so, oracle/Makefile) and against tests/golden/ref_rle.npz generated from it
(tests/golden/make_golden_rle.py).  The nearest resize is the PIL closed form of oracle/boxes.py (pinned there against PIL).
prep_im_for_blob is PARITY UNPINNED: cv2 is not installed in this image and the reference holds no image fixtures, so the
bilinear resize follows OpenCV's documented INTER_LINEAR arithmetic for float32 input (coordinate (dx+0.5)/fx-0.5 rounded to
float32, border handling of resize.cpp, horizontal then vertical pass in float32) without a cv2 run to confirm the last bit."""
import ctypes as C
import os

import numpy as np

from . import boxes as B

_HERE = os.path.dirname(os.path.abspath(__file__))
REF_MASKAPI = os.path.join(_HERE, '_ref', 'libmaskapi.so')


# ---------------------------------------------------------------------------- run-length masks
def rle_from_string(s):
    """maskApi.c:217-231"""
    if isinstance(s, bytes):
        s = s.decode('ascii')
    cnts = []
    p = 0
    while p < len(s):
        x = 0; k = 0; more = True
        while more:
            c = ord(s[p]) - 48
            x |= (c & 0x1f) << (5 * k)
            more = bool(c & 0x20)
            p += 1; k += 1
            if not more and (c & 0x10):
                x |= -1 << (5 * k)
        if len(cnts) > 2:
            x += cnts[-2]
        cnts.append(x & 0xffffffff)
    return np.asarray(cnts, np.uint32)


def rle_to_string(cnts):
    """maskApi.c:203-215"""
    out = []
    cnts = [int(c) for c in cnts]
    for i, c in enumerate(cnts):
        x = c
        if i > 2:
            x -= cnts[i - 2]
        more = True
        while more:
            ch = x & 0x1f
            x >>= 5
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(chr(ch + 48))
    return ''.join(out)


def rle_encode(mask_hw):
    """maskApi.c:32-41 on one [h][w] mask (runs over the column-major order, starting with a run of zeros)"""
    t = np.asarray(mask_hw, np.uint8).T.reshape(-1)
    cnts = []
    p = 0; c = 0
    for v in t:
        if v != p:
            cnts.append(c); c = 0; p = v
        c += 1
    cnts.append(c)
    return np.asarray(cnts, np.uint32)


def rle_decode(cnts, h, w):
    """maskApi.c:43-47 -> [h][w] uint8"""
    flat = np.zeros(h * w, np.uint8)
    pos = 0; v = 0
    for c in cnts:
        c = int(c)
        if v:
            flat[pos:pos + c] = 1
        pos += c; v ^= 1
    return flat.reshape(w, h).T.copy()


def ref_mask(rles, out_h, out_w):
    """cycle_loader.py:198-210: rles = [{'size': [h, w], 'counts': str}, ...] of one referred object -> uint8 [out_h][out_w]"""
    if isinstance(rles, dict):
        rles = [rles]
    h, w = int(rles[0]['size'][0]), int(rles[0]['size'][1])
    m = np.zeros((h, w), np.int64)
    for r in rles:
        m += rle_decode(rle_from_string(r['counts']), h, w)
    m = (m > 0).astype(np.uint8)
    return B.imresize_nearest_u8(m, (out_h, out_w))


# ---------------------------------------------------------------------------- image
def prep_scale(h, w, target_size, max_size):
    """blob.py:35-43 and cv2.resize's dsize = round-half-even(size * scale)"""
    smin, smax = min(h, w), max(h, w)
    sc = float(target_size) / float(smin)
    if np.round(sc * smax) > max_size:
        sc = float(max_size) / float(smax)
    return sc, int(np.rint(h * sc)), int(np.rint(w * sc))


def _lin_table(n_dst, inv_scale, n_src, edit_weights):
    d = np.arange(n_dst, dtype=np.float64)
    f = ((d + 0.5) * inv_scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    if edit_weights:
        lo = s < 0
        f[lo] = 0; s[lo] = 0
        hi = s >= n_src - 1
        f[hi] = 0; s[hi] = n_src - 1
        s0, s1 = s, np.minimum(s + 1, n_src - 1)
    else:
        s0, s1 = np.clip(s, 0, n_src - 1), np.clip(s + 1, 0, n_src - 1)
    return s0, s1, (np.float32(1) - f).astype(np.float32), f


def prep_im_for_blob(im_u8_bgr, pixel_means, target_size, max_size):
    """blob.py:32-47 -> (float32 [oh][ow][3], im_scale)"""
    im = im_u8_bgr.astype(np.float32)
    im -= np.asarray(pixel_means, np.float64).reshape(1, 1, 3)           # float64 subtraction rounded back to float32 (numpy in-place rule)
    h, w = im.shape[:2]
    sc, oh, ow = prep_scale(h, w, target_size, max_size)
    inv = 1.0 / sc
    x0, x1, a0, a1 = _lin_table(ow, inv, w, True)
    y0, y1, b0, b1 = _lin_table(oh, inv, h, False)
    a0 = a0[None, :, None]; a1 = a1[None, :, None]
    hz = (im[:, x0] * a0).astype(np.float32) + (im[:, x1] * a1).astype(np.float32)       # horizontal pass, every source row
    hz = hz.astype(np.float32)
    out = (hz[y0] * b0[:, None, None]).astype(np.float32) + (hz[y1] * b1[:, None, None]).astype(np.float32)
    return out.astype(np.float32), sc


# ---------------------------------------------------------------------------- the blobs dict
def xywh_to_xyxy(boxes):
    boxes = np.asarray(boxes, np.float64)
    return np.hstack((boxes[:, 0:2], boxes[:, 0:2] + boxes[:, 2:4] - 1))


def get_batch(info, labels, image_id, image_u8_bgr, pixel_means, target_size, max_size, test=False, cycle=True):
    """cycle_loader.py:180-357 for one image (getTestBatch, gt_mrcn_loader.py:660-739, with test=True): `info` is the parsed
    data.json, `labels` the /labels array of data.h5, `image_u8_bgr` what cv2.imread returns."""
    Refs = {r['ref_id']: r for r in info['refs']}
    Images = {im['image_id']: im for im in info['images']}
    Sents = {s['sent_id']: s for s in info['sentences']}
    blob, sc = prep_im_for_blob(image_u8_bgr, pixel_means, target_size, max_size)
    blob = blob[None]
    ref_ids_out, sent_ids, cats, masks = [], [], [], []
    for ref_id in Images[image_id]['ref_ids']:
        ref = Refs[ref_id]
        m = ref_mask(ref['rle'], blob.shape[1], blob.shape[2])
        for sid in ref['sent_ids']:
            ref_ids_out.append(ref_id); sent_ids.append(sid); cats.append(ref['category_id']); masks.append(m)
    boxes = xywh_to_xyxy(np.vstack([Refs[r]['box'] for r in ref_ids_out]))
    pos_labels = np.vstack([labels[Sents[s]['h5_id']] for s in sent_ids])
    max_len = int((pos_labels != 0).sum(1).max())
    data = {'data': blob, 'im_info': np.array([[blob.shape[1], blob.shape[2], sc]], np.float32),
            'gt_boxes': np.concatenate((boxes * sc, np.array([cats]).T), axis=1).astype(np.float32),
            'gt_masks': np.stack(masks, 0), 'labels': pos_labels[:, :max_len].astype(np.int64),
            'file_name': Images[image_id]['file_name']}
    if cycle and not test:
        cap = pos_labels[:, :max_len]
        label_batch = np.zeros((cap.shape[0], cap.shape[1] + 2), np.int64)
        mask_batch = np.zeros((cap.shape[0], cap.shape[1] + 2), np.float32)
        label_batch[:, 1:-1] = cap
        for ix in range(cap.shape[0]):
            mask_batch[ix, :int((label_batch[ix] != 0).sum()) + 2] = 1
        data.update({'ref_ids': ref_ids_out, 'cap_labels': label_batch, 'cap_masks': mask_batch})
    return data


# ---------------------------------------------------------------------------- the reference's own maskApi.c (when built)
class _RLE(C.Structure):
    _fields_ = [('h', C.c_ulong), ('w', C.c_ulong), ('m', C.c_ulong), ('cnts', C.POINTER(C.c_uint))]


class RefMaskApi(object):
    """ctypes view of oracle/_ref/libmaskapi.so = pyutils/refer/external/maskApi.c compiled as it is."""

    def __init__(self):
        self.lib = C.CDLL(REF_MASKAPI)
        L = self.lib
        L.rleToString.restype = C.c_void_p
        L.rleToString.argtypes = [C.POINTER(_RLE)]
        L.rleFrString.argtypes = [C.POINTER(_RLE), C.c_char_p, C.c_ulong, C.c_ulong]
        L.rleEncode.argtypes = [C.POINTER(_RLE), C.c_void_p, C.c_ulong, C.c_ulong, C.c_ulong]
        L.rleDecode.argtypes = [C.POINTER(_RLE), C.c_void_p, C.c_ulong]
        L.rleFrPoly.argtypes = [C.POINTER(_RLE), C.c_void_p, C.c_ulong, C.c_ulong, C.c_ulong]
        L.rleFree.argtypes = [C.POINTER(_RLE)]
        self.libc = C.CDLL(None)
        self.libc.free.argtypes = [C.c_void_p]

    @staticmethod
    def available():
        return os.path.exists(REF_MASKAPI)

    def _string(self, R):
        p = self.lib.rleToString(C.byref(R))
        s = C.string_at(p).decode('ascii')
        self.libc.free(p)
        return s

    def encode(self, mask_hw):
        """[h][w] uint8 -> compressed string (external/_mask.pyx encode: Fortran-order mask)"""
        m = np.asfortranarray(np.asarray(mask_hw, np.uint8))
        R = _RLE()
        self.lib.rleEncode(C.byref(R), m.ctypes.data, m.shape[0], m.shape[1], 1)
        s = self._string(R)
        self.lib.rleFree(C.byref(R))
        return s

    def from_poly(self, xy, h, w):
        xy = np.ascontiguousarray(xy, np.float64)
        R = _RLE()
        self.lib.rleFrPoly(C.byref(R), xy.ctypes.data, xy.size // 2, h, w)
        s = self._string(R)
        self.lib.rleFree(C.byref(R))
        return s

    def counts(self, s, h, w):
        R = _RLE()
        self.lib.rleFrString(C.byref(R), s.encode('ascii'), h, w)
        c = np.ctypeslib.as_array(R.cnts, shape=(R.m,)).astype(np.uint32).copy() if R.m else np.zeros(0, np.uint32)
        self.lib.rleFree(C.byref(R))
        return c

    def decode(self, s, h, w):
        """-> [h][w] uint8 (external/_mask.pyx decode)"""
        R = _RLE()
        self.lib.rleFrString(C.byref(R), s.encode('ascii'), h, w)
        m = np.zeros((h, w), np.uint8, order='F')
        self.lib.rleDecode(C.byref(R), m.ctypes.data, 1)
        self.lib.rleFree(C.byref(R))
        return np.ascontiguousarray(m)
