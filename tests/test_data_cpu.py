"""Input side, CPU part: the oracle's run-length restatement against the fixture generated from the
reference's own maskApi.c (and against that library live, where oracle/_ref was built), the host entry points of the C ABI
(no GPU work), and the Loader index."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from oracle import data as OD
from data_util import load_rle_fixture, write_tiny_dataset


def test_oracle_rle_matches_reference_fixture():
    cases, groups = load_rle_fixture()
    assert len(cases) >= 30
    for c in cases:
        cnt = OD.rle_from_string(c['s'])
        assert np.array_equal(cnt, c['counts']), c['s'][:40]
        assert int(cnt.sum()) == c['h'] * c['w']
        assert np.array_equal(OD.rle_decode(cnt, c['h'], c['w']), c['mask'])
        assert OD.rle_to_string(cnt) == c['s']
        assert OD.rle_to_string(OD.rle_encode(c['mask'])) == OD.rle_to_string(OD.rle_encode(OD.rle_decode(cnt, c['h'], c['w'])))
    for g in groups:                 # objects of several segments: union, then nearest resize
        rles = [{'size': [cases[i]['h'], cases[i]['w']], 'counts': cases[i]['s']} for i in g]
        u = np.zeros_like(cases[g[0]]['mask'])
        for i in g:
            u |= cases[i]['mask']
        assert np.array_equal(OD.ref_mask(rles, u.shape[0], u.shape[1]), u)


@pytest.mark.skipif(not OD.RefMaskApi.available(), reason='oracle/_ref/libmaskapi.so not built (needs /root/reference)')
def test_oracle_rle_matches_reference_library_live():
    api = OD.RefMaskApi()
    rs = np.random.RandomState(1)
    for _ in range(40):
        h, w = int(rs.randint(1, 90)), int(rs.randint(1, 90))
        m = (rs.rand(h, w) < rs.uniform(0.02, 0.98)).astype(np.uint8)
        s = api.encode(m)
        assert OD.rle_to_string(OD.rle_encode(m)) == s
        assert np.array_equal(OD.rle_from_string(s), api.counts(s, h, w))
        assert np.array_equal(OD.rle_decode(OD.rle_from_string(s), h, w), api.decode(s, h, w))
        assert np.array_equal(api.decode(s, h, w), m)


def test_host_rle_from_string_matches_fixture():
    from lang2seg_amd import ops as O
    cases, _ = load_rle_fixture()
    for c in cases:
        assert np.array_equal(O.rle_from_string(c['s']), c['counts'])
    with pytest.raises(ValueError):
        O.rle_from_string('0P')            # 'P' - 48 = 0x20: continuation bit set, then the string ends


def test_host_prep_geometry_matches_blob_py():
    from lang2seg_amd import ops as O
    rs = np.random.RandomState(2)
    sizes = [(375, 500), (480, 640), (640, 427), (300, 900), (333, 1000), (600, 1000), (100, 100), (427, 640), (1, 7)]
    sizes += [(int(rs.randint(50, 1300)), int(rs.randint(50, 1300))) for _ in range(200)]
    for h, w in sizes:
        sc, oh, ow = O.prep_geometry(h, w, 600, 1000)
        rsc, roh, row = OD.prep_scale(h, w, 600, 1000)
        assert sc == rsc and (oh, ow) == (roh, row), (h, w)
        assert max(oh, ow) <= 1000


def test_loader_index(tmp_path):
    from lang2seg_amd.loaders.loader import Loader
    info, labels, _ = write_tiny_dataset(str(tmp_path))
    ld = Loader(os.path.join(str(tmp_path), 'data.json'), os.path.join(str(tmp_path), 'data.h5'), verbose=False)
    assert ld.vocab_size == len(info['word_to_ix']) and ld.label_length == 8
    sid = info['refs'][1]['sent_ids'][0]
    assert np.array_equal(ld.fetch_seq(sid), labels[ld.Sentences[sid]['h5_id']])
    seq, sids = ld.fetch_label(info['refs'][0]['ref_id'], 3)
    assert seq.shape == (3, 8) and len(sids) == 3
    enc = ld.encode_labels(['w3 w5 nope', 'w1'])
    assert enc[0, :3].tolist() == [3, 5, ld.word_to_ix['<UNK>']] and enc[1, 0] == 1 and enc[1, 1] == 0
    assert ld.decode_labels(enc)[1] == 'w1'
    assert ld.sentToRef[sid]['ref_id'] == info['refs'][1]['ref_id']


def test_h5lite_reads_what_h5py_wrote(tmp_path):
    """loaders/h5lite.py (no h5py in this image) against files written by h5py itself with the reference's own call
    (tools/prepro.py:287-289: create_dataset('labels', dtype='int32', data=L)) - tests/golden/make_golden_h5.py, run with the one
    interpreter of this container that has h5py: default (earliest) and latest file formats, several datasets and a sub-group in one file,
    int32 / int64 / big-endian int16 / uint8, contiguous and compact layouts; chunked + gzip is refused by name."""
    import os
    from lang2seg_amd.loaders import h5lite
    from lang2seg_amd.loaders.loader import load_labels
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'h5')
    exp = np.load(os.path.join(d, 'expected.npz'))
    for fn, name, key in (('data_prepro.h5', 'labels', 'data_prepro'), ('multi.h5', 'labels', 'multi_labels'), ('multi.h5', 'zzz_be', 'multi_zzz_be'),
                          ('multi.h5', 'grp/inner', 'multi_inner'), ('latest.h5', 'labels', 'latest'), ('compact.h5', 'labels', 'compact')):
        a = h5lite.File(os.path.join(d, fn))[name]
        assert a.dtype == exp[key].dtype and a.shape == exp[key].shape and np.array_equal(a, exp[key]), (fn, name)
    assert h5lite.File(os.path.join(d, 'multi.h5')).keys() == ['aaa', 'grp', 'labels', 'zzz_be']
    assert np.array_equal(load_labels(os.path.join(d, 'data_prepro.h5')), exp['data_prepro'])        # the Loader's entry point
    with pytest.raises(h5lite.H5LiteError, match='compress'):
        h5lite.File(os.path.join(d, 'chunked_gzip.h5'))['labels']
    with pytest.raises(KeyError):
        h5lite.File(os.path.join(d, 'multi.h5'))['missing']
    bad = tmp_path / 'x.h5'
    bad.write_bytes(b'not an hdf5 file' * 40)
    with pytest.raises(h5lite.H5LiteError):
        h5lite.File(str(bad))
