#!/usr/bin/env python
"""Generates tests/golden/ref_rle.npz from the REFERENCE's COCO mask API compiled as it is
(oracle/_ref/libmaskapi.so <- /root/reference/pyutils/refer/external/maskApi.c, recipe: oracle/Makefile).

Every case: a binary mask (or a polygon) -> rleEncode / rleFrPoly -> rleToString  (the `counts` strings data.json holds) ->
rleFrString -> rleDecode.  Stored: size, string, the run lengths the reference parsed, the decoded mask (bit-packed).
Run here (needs /root/reference for the build):  make -C oracle && python tests/golden/make_golden_rle.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.data import RefMaskApi


def main():
    api = RefMaskApi()
    rs = np.random.RandomState(20260)
    cases = []          # (h, w, string)

    def add_mask(m):
        m = np.asarray(m, np.uint8)
        cases.append((m.shape[0], m.shape[1], api.encode(m)))

    # degenerate and border cases
    add_mask(np.zeros((1, 1))); add_mask(np.ones((1, 1)))
    add_mask(np.zeros((7, 5))); add_mask(np.ones((7, 5)))
    m = np.zeros((9, 4)); m[0, 0] = 1; add_mask(m)                 # first run of zeros is empty
    m = np.zeros((9, 4)); m[-1, -1] = 1; add_mask(m)
    add_mask(rs.rand(1, 97) > 0.5); add_mask(rs.rand(83, 1) > 0.5)
    add_mask(rs.rand(33, 47) > 0.5)                                # salt and pepper: many short runs
    add_mask(rs.rand(64, 64) > 0.97)                               # long runs -> multi-character counts, negative deltas
    # object-like masks at image sizes of the dataset (COCO: up to 640 on a side)
    for (h, w) in ((375, 500), (480, 640), (640, 427), (333, 500), (120, 160)):
        yy, xx = np.mgrid[0:h, 0:w]
        cx, cy = rs.uniform(0.2, 0.8) * w, rs.uniform(0.2, 0.8) * h
        a, b = rs.uniform(0.05, 0.4) * w, rs.uniform(0.05, 0.4) * h
        add_mask(((xx - cx) / a) ** 2 + ((yy - cy) / b) ** 2 <= 1.0)
        m = np.zeros((h, w)); x0, y0 = rs.randint(0, w // 2), rs.randint(0, h // 2)
        m[y0:y0 + rs.randint(1, h // 2), x0:x0 + rs.randint(1, w // 2)] = 1
        add_mask(m)
    # polygons through the reference's rasteriser (how data.json's refs were made: frPyObjects on COCO polygons)
    n_poly0 = len(cases)
    for (h, w) in ((375, 500), (480, 640), (200, 300), (50, 40)):
        for k in (3, 5, 9, 14):
            ang = np.sort(rs.uniform(0, 2 * np.pi, k))
            rad = rs.uniform(0.15, 0.45, k)
            xy = np.stack([w * (0.5 + rad * np.cos(ang)), h * (0.5 + rad * np.sin(ang))], 1).reshape(-1)
            cases.append((h, w, api.from_poly(xy, h, w)))
    # objects made of several segments (cycle_loader.py:205 sums them): groups of case indices with equal size
    groups = [[10, 11], [12, 13], [n_poly0, n_poly0 + 1, n_poly0 + 2]]

    H = np.array([c[0] for c in cases], np.int32); W = np.array([c[1] for c in cases], np.int32)
    strings = np.array([c[2].encode('ascii') for c in cases])
    cnt_list = [api.counts(c[2], c[0], c[1]) for c in cases]
    offs = np.concatenate([[0], np.cumsum([len(c) for c in cnt_list])]).astype(np.int64)
    masks = [api.decode(c[2], c[0], c[1]) for c in cases]
    moffs = np.concatenate([[0], np.cumsum([(m.size + 7) // 8 for m in masks])]).astype(np.int64)
    packed = np.concatenate([np.packbits(m.reshape(-1)) for m in masks])
    out = os.path.join(ROOT, 'tests', 'golden', 'ref_rle.npz')
    np.savez_compressed(out, h=H, w=W, strings=strings, counts=np.concatenate(cnt_list).astype(np.uint32), counts_off=offs,
                        masks_packed=packed, masks_off=moffs, groups=np.array([g + [-1] * (3 - len(g)) for g in groups], np.int32))
    print('wrote', out, len(cases), 'cases', os.path.getsize(out), 'bytes')


if __name__ == '__main__':
    main()
