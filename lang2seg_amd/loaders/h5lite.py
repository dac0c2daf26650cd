"""A dependency-free reader for the HDF5 files of the REFER preprocessing: `data.h5` holds ONE dataset, `/labels`, written by
`h5py.File(path, 'w').create_dataset('labels', dtype='int32', data=L)` (tools/prepro.py:287-289 of the reference) and read by
`h5py.File(data_h5, 'r')` in lib/loaders/loader.py:103-104.  h5py is not part of this image, so the loader reads the file format
itself (HDF5 File Format Specification 3.0): superblock versions 0-3, version-1 object headers with symbol-table groups (what
h5py's default `libver='earliest'` writes) and version-2 object headers with link messages (`libver='latest'`), fixed-point and
floating-point datatypes of either byte order, CONTIGUOUS and COMPACT dataset layouts.  Chunked / filtered (compressed) datasets,
dense link storage (fractal heaps), variable-length and compound types raise H5LiteError naming what is unsupported - the
preprocessing never produces them.  Pinned by files written by h5py itself: tests/golden/h5 (tests/golden/make_golden_h5.py)."""
import struct

import numpy as np

SIG = b'\x89HDF\r\n\x1a\n'
UNDEF = None


class H5LiteError(Exception):
    pass


class _Reader(object):
    def __init__(self, buf):
        self.b = buf
        self.O = self.L = 8
        self.base = 0

    def u(self, off, n):
        if off < 0 or off + n > len(self.b):
            raise H5LiteError('truncated file: %d bytes at offset %d' % (n, off))
        return int.from_bytes(self.b[off:off + n], 'little')

    def addr(self, off):
        v = self.u(off, self.O)
        return UNDEF if v == (1 << (8 * self.O)) - 1 else v + self.base


class File(object):
    """`File(path)['labels']` -> numpy array (the whole dataset); `keys()` lists the root group"""

    def __init__(self, path):
        with open(path, 'rb') as f:
            self._r = r = _Reader(f.read())
        self.path = path
        off = 0
        while True:                                           # the superblock may sit at 0, 512, 1024, 2048, ...
            if r.b[off:off + 8] == SIG:
                break
            off = 512 if off == 0 else off * 2
            if off + 8 > len(r.b):
                raise H5LiteError('%s: not an HDF5 file (no superblock signature)' % path)
        ver = r.u(off + 8, 1)
        if ver in (0, 1):
            r.O, r.L = r.u(off + 13, 1), r.u(off + 14, 1)
            p = off + 24 + (4 if ver == 1 else 0)
            r.base = r.u(p, r.O)
            p += 4 * r.O                                       # base, free-space info, end of file, driver info
            # root group symbol table entry: link name offset, object header address, cache type, reserved, scratch pad
            self._root = r.addr(p + r.O)
        elif ver in (2, 3):
            r.O, r.L = r.u(off + 9, 1), r.u(off + 10, 1)
            p = off + 12
            r.base = r.u(p, r.O)
            self._root = r.addr(p + 3 * r.O)
        else:
            raise H5LiteError('%s: superblock version %d is not supported' % (path, ver))
        if r.O not in (4, 8) or r.L not in (4, 8):
            raise H5LiteError('%s: offsets of %d bytes / lengths of %d bytes are not supported' % (path, r.O, r.L))

    # ---------------------------------------------------------------- object headers
    def _messages(self, addr):
        """[(type, flags, offset of the message data, size)] of the object header at addr, continuation blocks included"""
        r = self._r
        out = []
        if r.b[addr:addr + 4] == b'OHDR':
            if r.u(addr + 4, 1) != 2:
                raise H5LiteError('object header version %d' % r.u(addr + 4, 1))
            flags = r.u(addr + 5, 1)
            p = addr + 6
            if flags & 0x20:
                p += 16
            if flags & 0x10:
                p += 4
            n = 1 << (flags & 3)
            size0 = r.u(p, n); p += n
            blocks = [(p, p + size0)]
            track = bool(flags & 0x04)
            while blocks:
                p, end = blocks.pop(0)
                while p + 4 <= end:
                    t, sz, fl = r.u(p, 1), r.u(p + 1, 2), r.u(p + 3, 1)
                    p += 4 + (2 if track else 0)
                    if p + sz > end:
                        break
                    if t == 0x10:
                        ca, cl = r.addr(p), r.u(p + r.O, r.L)
                        if r.b[ca:ca + 4] != b'OCHK':
                            raise H5LiteError('object header continuation without signature')
                        blocks.append((ca + 4, ca + cl - 4))  # (signature in front, checksum behind)
                    elif t != 0:
                        out.append((t, fl, p, sz))
                    p += sz
            return out
        if r.u(addr, 1) != 1:
            raise H5LiteError('object header version %d at %d' % (r.u(addr, 1), addr))
        nmsg, size = r.u(addr + 2, 2), r.u(addr + 8, 4)
        blocks = [(addr + 16, addr + 16 + size)]
        while blocks and len(out) < nmsg + 64:
            p, end = blocks.pop(0)
            while p + 8 <= end:
                t, sz, fl = r.u(p, 2), r.u(p + 2, 2), r.u(p + 4, 1)
                p += 8
                if t == 0x10:
                    blocks.append((r.addr(p), r.addr(p) + r.u(p + r.O, r.L)))
                elif t != 0:
                    out.append((t, fl, p, sz))
                p += (sz + 7) // 8 * 8
        return out

    # ---------------------------------------------------------------- groups
    def _links(self, addr):
        """{name: object header address} of the group at addr"""
        r = self._r
        links = {}
        for t, fl, p, sz in self._messages(addr):
            if t == 0x11:                                     # symbol table: B-tree of symbol nodes + local heap of names
                btree, heap = r.addr(p), r.addr(p + r.O)
                if r.b[heap:heap + 4] != b'HEAP':
                    raise H5LiteError('local heap signature')
                data = r.addr(heap + 8 + 2 * r.L)
                self._walk_btree(btree, data, links)
            elif t == 0x06:                                   # link message (version-2 groups, compact storage)
                if r.u(p, 1) != 1:
                    raise H5LiteError('link message version %d' % r.u(p, 1))
                lf = r.u(p + 1, 1)
                q = p + 2
                ltype = 0
                if lf & 0x08:
                    ltype = r.u(q, 1); q += 1
                if lf & 0x04:
                    q += 8
                if lf & 0x10:
                    q += 1
                n = 1 << (lf & 3)
                ln = r.u(q, n); q += n
                name = bytes(r.b[q:q + ln]).decode('utf-8'); q += ln
                if ltype == 0:
                    links[name] = r.addr(q)
            elif t == 0x02:                                   # link info: a fractal heap address means dense storage
                lf = r.u(p + 1, 1)
                q = p + 2 + (8 if lf & 1 else 0)
                if r.addr(q) is not UNDEF:
                    raise H5LiteError('%s: a group with dense link storage (fractal heap) is not supported' % self.path)
        return links

    def _walk_btree(self, addr, heap_data, links):
        r = self._r
        if r.b[addr:addr + 4] != b'TREE' or r.u(addr + 4, 1) != 0:
            raise H5LiteError('group B-tree node at %d' % addr)
        level, used = r.u(addr + 5, 1), r.u(addr + 6, 2)
        p = addr + 8 + 2 * r.O
        for i in range(used):
            child = r.addr(p + r.L + i * (r.L + r.O))
            if level > 0:
                self._walk_btree(child, heap_data, links)
                continue
            if r.b[child:child + 4] != b'SNOD':
                raise H5LiteError('symbol table node at %d' % child)
            nsym = r.u(child + 6, 2)
            q = child + 8
            for _ in range(nsym):
                noff, oh = r.u(q, r.O), r.addr(q + r.O)
                s = heap_data + noff
                e = r.b.index(b'\0', s)
                links[bytes(r.b[s:e]).decode('utf-8')] = oh
                q += 2 * r.O + 24

    def keys(self):
        return sorted(self._links(self._root))

    def __contains__(self, name):
        try:
            self._resolve(name)
            return True
        except KeyError:
            return False

    def _resolve(self, name):
        addr = self._root
        for part in [p for p in name.split('/') if p]:
            links = self._links(addr)
            if part not in links:
                raise KeyError('%s: no object %r (have %s)' % (self.path, name, sorted(links)))
            addr = links[part]
        return addr

    # ---------------------------------------------------------------- datasets
    def __getitem__(self, name):
        r = self._r
        addr = self._resolve(name)
        shape = dtype = None
        layout = None
        for t, fl, p, sz in self._messages(addr):
            if t == 0x01:                                     # dataspace
                ver, rank = r.u(p, 1), r.u(p + 1, 1)
                q = p + (8 if ver == 1 else 4)
                if ver not in (1, 2):
                    raise H5LiteError('dataspace version %d' % ver)
                shape = tuple(r.u(q + i * r.L, r.L) for i in range(rank))
            elif t == 0x03:                                   # datatype
                cv = r.u(p, 1)
                cls, bits0, size = cv & 15, r.u(p + 1, 1), r.u(p + 4, 4)
                order = '>' if bits0 & 1 else '<'
                if cls == 0:
                    dtype = np.dtype('%s%s%d' % (order, 'i' if bits0 & 8 else 'u', size))
                elif cls == 1:
                    dtype = np.dtype('%sf%d' % (order, size))
                else:
                    raise H5LiteError('%s: datatype class %d of %r is not supported (only integers and floats)' % (self.path, cls, name))
            elif t == 0x08:                                   # data layout
                ver = r.u(p, 1)
                if ver in (3, 4):
                    cls = r.u(p + 1, 1)
                    if cls == 0:
                        n = r.u(p + 2, 2)
                        layout = ('compact', p + 4, n)
                    elif cls == 1:
                        layout = ('contiguous', r.addr(p + 2), r.u(p + 2 + r.O, r.L))
                    else:
                        layout = ('chunked', None, None)
                elif ver in (1, 2):
                    rank, cls = r.u(p + 1, 1), r.u(p + 2, 1)
                    q = p + 8
                    if cls == 1:
                        layout = ('contiguous', r.addr(q), None)
                    elif cls == 0:
                        q += 4 * rank
                        layout = ('compact', q + 4, r.u(q, 4))
                    else:
                        layout = ('chunked', None, None)
                else:
                    raise H5LiteError('data layout version %d' % ver)
            elif t == 0x0B and sz > 0:                        # filter pipeline
                raise H5LiteError('%s: dataset %r is filtered (compressed); rewrite it without compression' % (self.path, name))
        if shape is None or dtype is None or layout is None:
            raise H5LiteError('%s: %r is not a dataset' % (self.path, name))
        if layout[0] == 'chunked':
            raise H5LiteError('%s: dataset %r has a chunked layout; only contiguous and compact datasets are supported '
                              '(h5py writes those unless chunks= / compression= / maxshape= is given)' % (self.path, name))
        count = int(np.prod(shape)) if shape else 1
        nbytes = count * dtype.itemsize
        if count == 0:
            return np.zeros(shape, dtype.newbyteorder('='))
        off = layout[1]
        if off is UNDEF:
            return np.zeros(shape, dtype.newbyteorder('='))   # never written: the fill value (zero)
        if off + nbytes > len(r.b):
            raise H5LiteError('%s: dataset %r runs past the end of the file' % (self.path, name))
        a = np.frombuffer(r.b, dtype=dtype, count=count, offset=off).reshape(shape)
        return a.astype(dtype.newbyteorder('='))              # native byte order, owning copy

    def close(self):
        self._r = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def read_dataset(path, name):
    with File(path) as f:
        return f[name]
