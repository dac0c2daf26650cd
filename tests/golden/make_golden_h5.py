#!/usr/bin/env python
"""HDF5 fixtures for the dependency-free reader lang2seg_amd/loaders/h5lite.py, written by h5py itself - the library (and the call) the
reference uses for data.h5: tools/prepro.py:287-289  `f = h5py.File(path, 'w'); f.create_dataset('labels', dtype='int32', data=L)`.
This image's default interpreter has no h5py; /opt/conda/bin/python3.9 does (h5py 3.3.0 / HDF5 1.10.6):
    /opt/conda/bin/python3.9 tests/golden/make_golden_h5.py
Writes tests/golden/h5/*.h5 (a few KB each) and tests/golden/h5/expected.npz (the arrays as h5py reads them back)."""
import os
import numpy as np
import h5py

out = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'h5')
os.makedirs(out, exist_ok=True)
rs = np.random.RandomState(5)
exp = {}

# (1) exactly the reference's call: one int32 dataset, default (earliest) file format -> superblock 0, symbol-table root group, contiguous layout
L = rs.randint(0, 1999, size=(37, 10)).astype('int32'); L[:, 6:] = 0
with h5py.File(os.path.join(out, 'data_prepro.h5'), 'w') as f:
    f.create_dataset('labels', dtype='int32', data=L)
exp['data_prepro'] = L

# (2) refcocog-sized rows (label_length 20), several datasets in the root group (the B-tree / heap walk has to pick the right name),
#     int64 and a big-endian dtype
L2 = rs.randint(0, 3349, size=(11, 20)).astype('int64')
with h5py.File(os.path.join(out, 'multi.h5'), 'w') as f:
    f.create_dataset('aaa', data=np.arange(7, dtype='float32'))
    f.create_dataset('labels', data=L2)
    f.create_dataset('zzz_be', data=np.arange(12, dtype='>i2').reshape(3, 4))
    f.create_group('grp').create_dataset('inner', data=np.arange(5, dtype='uint8'))
exp['multi_labels'] = L2; exp['multi_zzz_be'] = np.arange(12, dtype='i2').reshape(3, 4); exp['multi_inner'] = np.arange(5, dtype='uint8')

# (3) the newest file format (superblock 3, version-2 object headers, link messages instead of a symbol table)
L3 = rs.randint(0, 1999, size=(5, 10)).astype('int32')
with h5py.File(os.path.join(out, 'latest.h5'), 'w', libver='latest') as f:
    f.create_dataset('labels', dtype='int32', data=L3)
exp['latest'] = L3

# (4) a tiny dataset stored compactly inside the object header, and an empty one
with h5py.File(os.path.join(out, 'compact.h5'), 'w') as f:
    dcpl = h5py.h5p.create(h5py.h5p.DATASET_CREATE); dcpl.set_layout(h5py.h5d.COMPACT)
    sid = h5py.h5s.create_simple((2, 3)); tid = h5py.h5t.NATIVE_INT32
    d = h5py.h5d.create(f.id, b'labels', tid, sid, dcpl=dcpl)
    d.write(h5py.h5s.ALL, h5py.h5s.ALL, np.arange(6, dtype='int32').reshape(2, 3))
exp['compact'] = np.arange(6, dtype='int32').reshape(2, 3)

# (5) layouts the reader refuses by name (chunked / compressed): the error must say so
with h5py.File(os.path.join(out, 'chunked_gzip.h5'), 'w') as f:
    f.create_dataset('labels', data=L, chunks=(8, 10), compression='gzip')

np.savez(os.path.join(out, 'expected.npz'), **exp)
for fn in sorted(os.listdir(out)):
    print(fn, os.path.getsize(os.path.join(out, fn)))
