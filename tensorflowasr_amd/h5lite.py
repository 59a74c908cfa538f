"""Minimal read-only HDF5 reader (pure Python + NumPy) for Keras `.weights.h5` files (h5py is not installable here).

Covers what `h5py.File(path, "w")` with default settings writes, which is what keras' H5IOStore uses
(tensorflow_asr/models/base_model.py:55-61 -> keras.Model.save_weights -> saving_lib.H5IOStore): superblock version 0/1,
version-1 object headers (+ continuation blocks), old-style groups (symbol table message -> v1 B-tree -> symbol nodes + local
heap), simple dataspaces, fixed-point / IEEE floating-point datatypes, contiguous and compact dataset layouts, and chunked
layouts without filters (version-3 layout message, v1 chunk B-tree).  Anything else raises H5Error (no silent guesses).
Validated against files written by the real HDF5 library (tests/golden/h5lite_fixture.h5, written by h5py 3.3 / HDF5 1.10.6
with oracle/gen_h5_fixture.py) in tests/test_h5lite.py.

    with H5File(path) as f:
        f.datasets()            -> {"layers/dense/vars/0": ndarray, ...}   (every dataset of the file, by full path)
        f["layers/dense/vars/0"] -> ndarray
"""
import struct

import numpy as np

_UNDEF = 0xFFFFFFFFFFFFFFFF


class H5Error(ValueError):
    pass


class H5File:
    def __init__(self, path):
        with open(path, "rb") as fh:
            self.buf = fh.read()
        b = self.buf
        # the superblock may sit at 0, 512, 1024, ... (user block); keras files have none
        self.base = None
        off = 0
        while off + 8 <= len(b):
            if b[off:off + 8] == b"\x89HDF\r\n\x1a\n":
                self.base = off
                break
            off = 512 if off == 0 else off * 2
        if self.base is None:
            raise H5Error("not an HDF5 file (signature not found)")
        ver = b[self.base + 8]
        if ver not in (0, 1):
            raise H5Error(f"superblock version {ver} is not supported (h5py's default 'earliest' format writes version 0)")
        self.so, self.sl = b[self.base + 13], b[self.base + 14]
        if (self.so, self.sl) != (8, 8):
            raise H5Error("only 8-byte offsets / lengths are supported")
        p = self.base + 24 + (4 if ver == 1 else 0)
        base_addr, _free, _eof, _drv = struct.unpack_from("<QQQQ", b, p)
        self.base += 0 if base_addr in (0, _UNDEF) else base_addr
        p += 32
        # root group symbol table entry
        _name_off, root_hdr, cache, _res = struct.unpack_from("<QQII", b, p)
        self.root = root_hdr
        self._cache = {}

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.buf = None

    # ------------------------------------------------------------------------------------------- object headers
    def _messages(self, addr):
        b = self.buf
        a = self.base + addr
        if b[a:a + 4] == b"OHDR":
            raise H5Error("version-2 object headers (libver='latest') are not supported")
        ver, _r, nmsg, _ref, hsize = struct.unpack_from("<BBHII", b, a)
        if ver != 1:
            raise H5Error(f"object header version {ver} is not supported")
        blocks = [(a + 16, hsize)]
        out = []
        while blocks and len(out) < nmsg:
            p, size = blocks.pop(0)
            end = p + size
            while p + 8 <= end and len(out) < nmsg:
                mtype, msize, _flags = struct.unpack_from("<HHB", b, p)
                data = p + 8
                if mtype == 0x10:  # continuation
                    coff, clen = struct.unpack_from("<QQ", b, data)
                    blocks.append((self.base + coff, clen))
                out.append((mtype, data, msize))
                p = data + msize
        return out

    # ------------------------------------------------------------------------------------------- groups
    def _heap_data(self, heap_addr):
        b = self.buf
        a = self.base + heap_addr
        if b[a:a + 4] != b"HEAP":
            raise H5Error("bad local heap signature")
        _size, _free, data_addr = struct.unpack_from("<QQQ", b, a + 8)
        return self.base + data_addr

    def _group_entries(self, btree_addr, heap_addr):
        b = self.buf
        heap = self._heap_data(heap_addr)
        out = []

        def name_at(off):
            s = heap + off
            e = b.index(b"\x00", s)
            return b[s:e].decode("utf-8")

        def walk(addr):
            a = self.base + addr
            if b[a:a + 4] != b"TREE":
                raise H5Error("bad B-tree signature")
            ntype, level, used = struct.unpack_from("<BBH", b, a + 4)
            if ntype != 0:
                raise H5Error("expected a group B-tree node")
            p = a + 8 + 16  # skip left / right siblings
            for i in range(used):
                child = struct.unpack_from("<Q", b, p + 8 + i * 16)[0]  # key_i (8) child_i (8) ...
                if level > 0:
                    walk(child)
                else:
                    s = self.base + child
                    if b[s:s + 4] != b"SNOD":
                        raise H5Error("bad symbol node signature")
                    nsym = struct.unpack_from("<H", b, s + 6)[0]
                    for k in range(nsym):
                        e = s + 8 + k * 40
                        noff, ohdr = struct.unpack_from("<QQ", b, e)
                        out.append((name_at(noff), ohdr))

        walk(btree_addr)
        return out

    def _children(self, addr):
        """[(name, object header address)] of a group, or None if the object is not a group."""
        for mtype, data, _size in self._messages(addr):
            if mtype == 0x11:  # symbol table message
                bt, hp = struct.unpack_from("<QQ", self.buf, data)
                return self._group_entries(bt, hp)
            if mtype in (0x02, 0x06):
                raise H5Error("new-style (link message) groups are not supported: write the file with h5py's default libver")
        return None

    # ------------------------------------------------------------------------------------------- datasets
    def _dataset(self, addr):
        b = self.buf
        shape = dtype = None
        layout = None
        for mtype, data, _size in self._messages(addr):
            if mtype == 0x01:  # dataspace
                ver, rank, flags = struct.unpack_from("<BBB", b, data)
                p = data + (8 if ver == 1 else 4)
                if ver not in (1, 2):
                    raise H5Error(f"dataspace version {ver}")
                if ver == 2 and b[data + 3] == 2:
                    raise H5Error("null dataspace")
                shape = struct.unpack_from("<" + "Q" * rank, b, p) if rank else ()
            elif mtype == 0x03:  # datatype
                cv = b[data]
                cls, bits0 = cv & 0x0F, b[data + 1]
                size = struct.unpack_from("<I", b, data + 4)[0]
                endian = ">" if (bits0 & 1) else "<"
                if cls == 1:
                    dtype = np.dtype(f"{endian}f{size}")
                elif cls == 0:
                    dtype = np.dtype(f"{endian}{'i' if (bits0 & 0x08) else 'u'}{size}")
                else:
                    raise H5Error(f"datatype class {cls} is not supported (numeric weights only)")
            elif mtype == 0x08:  # layout
                ver = b[data]
                if ver != 3:
                    raise H5Error(f"data layout message version {ver} is not supported")
                lclass = b[data + 1]
                if lclass == 1:
                    la, ls = struct.unpack_from("<QQ", b, data + 2)
                    layout = ("contiguous", la, ls)
                elif lclass == 0:
                    n = struct.unpack_from("<H", b, data + 2)[0]
                    layout = ("compact", data + 4, n)
                elif lclass == 2:
                    nd = b[data + 2]
                    bt = struct.unpack_from("<Q", b, data + 3)[0]
                    dims = struct.unpack_from("<" + "I" * nd, b, data + 11)
                    layout = ("chunked", bt, dims)
                else:
                    raise H5Error(f"layout class {lclass}")
            elif mtype == 0x0B:
                raise H5Error("filtered (compressed) datasets are not supported")
        if shape is None or dtype is None or layout is None:
            return None
        n = int(np.prod(shape)) if shape else 1
        if layout[0] == "contiguous":
            if layout[1] == _UNDEF or n == 0:
                return np.zeros(shape, dtype.newbyteorder("="))
            a = self.base + layout[1]
            arr = np.frombuffer(b, dtype, n, a)
        elif layout[0] == "compact":
            arr = np.frombuffer(b, dtype, n, layout[1])
        else:
            arr = self._read_chunked(layout[1], layout[2], shape, dtype)
        return arr.reshape(shape).astype(dtype.newbyteorder("="), copy=True)

    def _read_chunked(self, btree, cdims, shape, dtype):
        b = self.buf
        rank = len(shape)
        chunk = cdims[:rank]
        out = np.zeros(shape, dtype)

        def walk(addr):
            a = self.base + addr
            if b[a:a + 4] != b"TREE":
                raise H5Error("bad chunk B-tree signature")
            ntype, level, used = struct.unpack_from("<BBH", b, a + 4)
            if ntype != 1:
                raise H5Error("expected a chunk B-tree node")
            ksz = 8 + 8 * (rank + 1)
            p = a + 24
            for i in range(used):
                k = p + i * (ksz + 8)
                csize, fmask = struct.unpack_from("<II", b, k)
                offs = struct.unpack_from("<" + "Q" * rank, b, k + 8)
                child = struct.unpack_from("<Q", b, k + ksz)[0]
                if level > 0:
                    walk(child)
                    continue
                if fmask:
                    raise H5Error("filtered chunks are not supported")
                data = np.frombuffer(b, dtype, int(np.prod(chunk)), self.base + child).reshape(chunk)
                sl = tuple(slice(o, min(o + c, s)) for o, c, s in zip(offs, chunk, shape))
                out[sl] = data[tuple(slice(0, s.stop - s.start) for s in sl)]

        if btree != _UNDEF:
            walk(btree)
        return out

    # ------------------------------------------------------------------------------------------- public
    def datasets(self):
        """{full path: ndarray} of every dataset in the file."""
        out = {}

        def rec(addr, prefix, depth):
            if depth > 64:
                raise H5Error("group nesting too deep (cycle?)")
            kids = self._children(addr)
            if kids is None:
                arr = self._dataset(addr)
                if arr is not None:
                    out[prefix] = arr
                return
            for name, child in kids:
                rec(child, f"{prefix}/{name}" if prefix else name, depth + 1)

        rec(self.root, "", 0)
        return out

    def __getitem__(self, path):
        addr = self.root
        for part in [t for t in path.split("/") if t]:
            kids = self._children(addr)
            if kids is None:
                raise KeyError(path)
            nxt = dict(kids).get(part)
            if nxt is None:
                raise KeyError(path)
            addr = nxt
        arr = self._dataset(addr)
        if arr is None:
            raise KeyError(f"{path} is a group")
        return arr


# ------------------------------------------------------------------------------------------------------------------------
# Writer: the same subset of the format, laid out like h5py's default ("earliest") files - superblock version 0, version-1
# object headers, old-style groups (symbol table message -> v1 B-tree -> symbol nodes + local heap), contiguous little-endian
# datasets.  What keras' H5IOStore needs to READ a `.weights.h5` (keras.Model.load_weights, base_model.py:59-61) is exactly this.
# Validated in tests/test_h5lite.py by reading the written file back with the reader above AND with the real HDF5 library
# (h5py under /opt/conda in the build container, oracle/check_h5_roundtrip.py).
# ------------------------------------------------------------------------------------------------------------------------
_LEAF_K, _INTERNAL_K = 4, 16


class _Out:
    def __init__(self):
        self.b = bytearray()

    def alloc(self, n):
        """reserve n bytes at the next 8-byte boundary -> address"""
        pad = (-len(self.b)) % 8
        self.b += b"\0" * pad
        a = len(self.b)
        self.b += b"\0" * n
        return a

    def put(self, addr, data):
        self.b[addr:addr + len(data)] = data


def _msg(mtype, payload, flags=0):
    payload = payload + b"\0" * ((-len(payload)) % 8)
    return struct.pack("<HHB3x", mtype, len(payload), flags) + payload


def _object_header(out, msgs):
    body = b"".join(msgs)
    a = out.alloc(16 + len(body))
    out.put(a, struct.pack("<BBHII4x", 1, 0, len(msgs), 1, len(body)) + body)
    return a


def _datatype_msg(dt):
    dt = np.dtype(dt)
    if dt.kind == "f" and dt.itemsize in (2, 4, 8):
        exp_bits, man_bits = {2: (5, 10), 4: (8, 23), 8: (11, 52)}[dt.itemsize]
        bits = 8 * dt.itemsize
        head = struct.pack("<BBBBI", 0x11, 0x20, bits - 1, 0, dt.itemsize)  # class 1 (float) v1; LE, implied-msb mantissa; sign bit location
        props = struct.pack("<HHBBBBI", 0, bits, man_bits, exp_bits, 0, man_bits, (1 << (exp_bits - 1)) - 1)
    elif dt.kind in "iu":
        head = struct.pack("<BBBBI", 0x10, 0x08 if dt.kind == "i" else 0x00, 0, 0, dt.itemsize)  # class 0 (fixed point) v1; LE; signed?
        props = struct.pack("<HH", 0, 8 * dt.itemsize)
    else:
        raise H5Error(f"cannot write dtype {dt} (numeric weights only)")
    return _msg(0x03, head + props, flags=1)


def _write_dataset(out, arr):
    arr = np.asarray(arr, order="C")  # (np.ascontiguousarray would turn a scalar into a one-element vector)
    if arr.dtype.byteorder == ">":
        arr = arr.astype(arr.dtype.newbyteorder("<"))
    raw = arr.tobytes()
    daddr = out.alloc(len(raw)) if raw else _UNDEF
    if raw:
        out.put(daddr, raw)
    space = struct.pack("<BBB5x", 1, arr.ndim, 0) + b"".join(struct.pack("<Q", int(s)) for s in arr.shape)
    msgs = [_msg(0x01, space), _datatype_msg(arr.dtype),
            _msg(0x05, struct.pack("<BBBBI", 2, 2, 2, 1, 0), flags=1),  # fill value v2: late allocation, write if set, default value
            _msg(0x08, struct.pack("<BBQQ", 3, 1, daddr, len(raw)))]   # layout v3, contiguous
    return _object_header(out, msgs)


def _write_group(out, children):
    """children: {name: nested dict (group) | ndarray (dataset)} -> (object header address, btree address, heap address)"""
    names = sorted(children, key=lambda s: s.encode("utf-8"))
    # children first (depth-first), so every address is known when this group's tables are written
    entries = []
    for nm in names:
        c = children[nm]
        if isinstance(c, dict):
            hdr, bt, hp = _write_group(out, c)
            entries.append((nm, hdr, 1, struct.pack("<QQ", bt, hp)))
        else:
            entries.append((nm, _write_dataset(out, c), 0, b"\0" * 16))
    # local heap: offset 0 = the empty string (the B-tree's leftmost key), then the link names, each padded to 8 bytes
    heap = bytearray(b"\0" * 8)
    noff = {}
    for nm in names:
        noff[nm] = len(heap)
        e = nm.encode("utf-8") + b"\0"
        heap += e + b"\0" * ((-len(e)) % 8)
    # one free block at the tail (>= 16 bytes: next = 1 (end of list), size) so that the library may add links later
    free_off = len(heap)
    heap += struct.pack("<QQ", 1, 32) + b"\0" * 16
    hdata = out.alloc(len(heap))
    out.put(hdata, bytes(heap))
    hp = out.alloc(32)
    out.put(hp, b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), free_off, hdata))
    # symbol nodes of <= 2 * leaf K entries, in name order
    cap = 2 * _LEAF_K
    level = []  # (address, heap offset of the largest name below it)
    for i in range(0, max(len(entries), 1), cap):
        chunk = entries[i:i + cap]
        a = out.alloc(8 + cap * 40)
        body = b"SNOD" + struct.pack("<BBH", 1, 0, len(chunk))
        for nm, hdr, ctype, scratch in chunk:
            body += struct.pack("<QQII", noff[nm], hdr, ctype, 0) + scratch
        out.put(a, body)
        level.append((a, noff[chunk[-1][0]] if chunk else 0))
    # v1 B-tree (node type 0) over the symbol nodes; key[0] = "" (offset 0), key[i + 1] = largest name in child i
    fan = 2 * _INTERNAL_K
    depth = 0
    while True:
        nodes = []
        for i in range(0, len(level), fan):
            kids = level[i:i + fan]
            a = out.alloc(24 + fan * 8 + (fan + 1) * 8)
            body = b"TREE" + struct.pack("<BBHQQ", 0, depth, len(kids), _UNDEF, _UNDEF) + struct.pack("<Q", 0)
            for child, key in kids:
                body += struct.pack("<QQ", child, key)
            nodes.append((a, body, kids[-1][1]))
        # sibling links and the first key of every node but the leftmost (= the last key of its left neighbour)
        for j, (a, body, _k) in enumerate(nodes):
            left = nodes[j - 1][0] if j else _UNDEF
            right = nodes[j + 1][0] if j + 1 < len(nodes) else _UNDEF
            first = nodes[j - 1][2] if j else 0
            body = body[:8] + struct.pack("<QQ", left, right) + struct.pack("<Q", first) + body[32:]
            out.put(a, body)
        level = [(a, k) for a, _b, k in nodes]
        if len(level) == 1:
            break
        depth += 1
    bt = level[0][0]
    hdr = _object_header(out, [_msg(0x11, struct.pack("<QQ", bt, hp))])
    return hdr, bt, hp


def write_h5(path, datasets, groups=()):
    """Write {"a/b/vars/0": ndarray, ...} (+ optional empty groups, e.g. keras' top-level "vars") as an HDF5 file."""
    tree = {}

    def node(parts):
        cur = tree
        for p in parts:
            nxt = cur.setdefault(p, {})
            if not isinstance(nxt, dict):
                raise H5Error(f"{'/'.join(parts)}: a dataset is in the way")
            cur = nxt
        return cur

    for g in groups:
        node([t for t in g.split("/") if t])
    for p, arr in datasets.items():
        parts = [t for t in p.split("/") if t]
        if not parts:
            raise H5Error("empty dataset path")
        parent = node(parts[:-1])
        if parts[-1] in parent:
            raise H5Error(f"{p}: name already used")
        parent[parts[-1]] = np.asarray(arr)
    out = _Out()
    out.alloc(96)  # superblock, written last
    root, bt, hp = _write_group(out, tree)
    out.alloc(0)
    eof = len(out.b)
    sb = b"\x89HDF\r\n\x1a\n" + struct.pack("<BBBBBBBB", 0, 0, 0, 0, 0, 8, 8, 0) + struct.pack("<HHI", _LEAF_K, _INTERNAL_K, 0)
    sb += struct.pack("<QQQQ", 0, _UNDEF, eof, _UNDEF)
    sb += struct.pack("<QQII", 0, root, 1, 0) + struct.pack("<QQ", bt, hp)
    out.put(0, sb)
    with open(path, "wb") as f:
        f.write(bytes(out.b))
    return eof
