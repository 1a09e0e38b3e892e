"""h5lite — the slice of the HDF5 wire format the ViMoCLIP pipeline uses, read and written natively (no h5py here).

What the reference stores (extract_embeddings.py:50-119, inference.py:94-112, inference_frame_diff.py:240-310,
extract_embeddings_mammalNet.py:85-153) and reads back (TFAM/data/dataset.py:25-66, dataset.py:35-75):
``/{video_id}/embeddings [T,E] f32`` (gzip, chunks ``(1,E)`` or ``(batch,E)``, extendable), ``/{video_id}/labels [C] f32``,
group attributes ``total_frames`` / ``original_frames`` (+ ``error`` / ``skipped_low_ram``), root attributes
(``num_classes`` int, ``dataset_name`` / ``type`` / ``clip_model`` strings) and ``/video_ids`` (variable-length UTF-8 strings).

The file layout written is the "earliest" HDF5 format that h5py produces by default: superblock v0, symbol-table groups
(v1 B-tree + local heap), v1 object headers, layout v3 (contiguous / chunked with a v1 chunk B-tree), filter pipeline v1
(deflate), attribute messages v1, global heap for variable-length strings.  The reader accepts the same plus
continuation blocks, superblock v1, compact layout, fixed-length strings, shuffle / fletcher32 / deflate / h5py-LZF
chunks and per-chunk filter masks.  ``compression="lzf"`` on write declares h5py's filter and stores the chunks raw with
the filter-skipped mask (see ``Dataset._create``).  Files in the "latest" format (superblock >= 2, fractal heaps) raise
NotImplementedError.

The API mirrors the h5py subset the reference calls: ``File(path, mode)`` as a context manager, ``keys / in / [] /
create_group / require_group / create_dataset(shape=, maxshape=, chunks=, compression=, dtype=, data=) / attrs / flush``,
``Dataset.shape / dtype / [...] / [...] = / resize`` and ``string_dtype()``.

Durability: raw chunks are appended as they are written; ``flush()`` writes every *dirty* object (and the groups on its
path to the root) to free space and then commits by rewriting the 96-byte superblock, so a reader (or a resumed run)
always sees the last committed tree.  Regions freed by a commit are reused by later allocations of the same session.

Pinned in tests/test_h5lite.py against libhdf5 1.10.6 (the C library under h5py) through ctypes, both directions.
"""
from __future__ import annotations

import os
import struct
import zlib
from collections import OrderedDict

import numpy as np

UNDEF = 0xFFFFFFFFFFFFFFFF
SIG = b"\x89HDF\r\n\x1a\n"
LEAF_K, INTERNAL_K, CHUNK_K = 4, 16, 32
_VLEN_STR = "vlen_str"


_STR = np.dtype("O", metadata={"vlen": str})    # what h5py.string_dtype() returns: object dtype tagged variable-length str


def string_dtype(encoding="utf-8", length=None):
    if length is not None:
        raise NotImplementedError("fixed-length string dtype is read-only in h5lite")
    return _STR


def _is_str_dtype(dt) -> bool:
    try:
        return dt is not None and not isinstance(dt, tuple) and (dt == _VLEN_STR if isinstance(dt, str) else np.dtype(dt).kind == "O")
    except TypeError:
        return False


def _pad8(b: bytes) -> bytes:
    return b + b"\0" * (-len(b) % 8)


# ------------------------------------------------------------------------------------------------ datatypes
def _dt_encode(dt) -> bytes:
    """Datatype message body (version 1)."""
    if _is_str_dtype(dt):
        base = struct.pack("<B3BIHH4x", 0x10, 0x00, 0, 0, 1, 0, 8)              # base type: 1-byte unsigned char (as libhdf5)
        return struct.pack("<B3BI", 0x19, 0x01, 0x01, 0, 16) + base             # VL, type string, null-term, UTF-8
    if isinstance(dt, tuple):                                                   # ('fixed_str', n, utf8)
        return struct.pack("<B3BI", 0x13, 0x10 if dt[2] else 0x00, 0, 0, dt[1])
    if dt == "bool":
        base = struct.pack("<B3BIHH", 0x10, 0x08, 0, 0, 1, 0, 8)                # int8 signed LE
        names = _pad8(b"FALSE\0") + _pad8(b"TRUE\0")
        return struct.pack("<B3BI", 0x18, 2, 0, 0, 1) + base + names + bytes([0, 1])
    dt = np.dtype(dt)
    if dt.kind == "f":
        if dt.itemsize == 4:
            prop = struct.pack("<HHBBBBI", 0, 32, 23, 8, 0, 23, 127)
            return struct.pack("<B3BI", 0x11, 0x20, 31, 0, 4) + prop
        if dt.itemsize == 8:
            prop = struct.pack("<HHBBBBI", 0, 64, 52, 11, 0, 52, 1023)
            return struct.pack("<B3BI", 0x11, 0x20, 63, 0, 8) + prop
        if dt.itemsize == 2:
            prop = struct.pack("<HHBBBBI", 0, 16, 10, 5, 0, 10, 15)
            return struct.pack("<B3BI", 0x11, 0x20, 15, 0, 2) + prop
    if dt.kind in "iu":
        return struct.pack("<B3BIHH", 0x10, 0x08 if dt.kind == "i" else 0x00, 0, 0, dt.itemsize, 0, 8 * dt.itemsize)
    raise TypeError(f"h5lite cannot store dtype {dt}")


def _dt_decode(buf: bytes, off: int = 0):
    """-> (descr, size, end offset).  descr: numpy dtype | 'vlen_str' | 'bool' | ('fixed_str', n, utf8)."""
    cv, b0, b1, _b2, size = struct.unpack_from("<B3BI", buf, off)
    cls, ver = cv & 15, cv >> 4
    p = off + 8
    if cls == 0:
        if b0 & 1:
            raise NotImplementedError("big-endian integers")
        return np.dtype(("i" if b0 & 8 else "u") + str(size)), size, p + 4
    if cls == 1:
        if b0 & 1:
            raise NotImplementedError("big-endian floats")
        return np.dtype("f" + str(size)), size, p + 12
    if cls == 3:
        return ("fixed_str", size, (b0 >> 4) == 1), size, p
    if cls == 9:
        is_str = (b0 & 15) == 1
        _base, _bsize, end = _dt_decode(buf, p)
        if not is_str:
            raise NotImplementedError("variable-length sequences")
        return _VLEN_STR, size, end
    if cls == 8:
        nmemb = b0 | (b1 << 8)
        base, bsize, q = _dt_decode(buf, p)
        names = []
        for _ in range(nmemb):
            e = buf.index(b"\0", q)
            names.append(buf[q:e])
            q = e + 1 if ver >= 3 else q + ((e - q + 8) // 8) * 8
        q += nmemb * bsize
        if sorted(names) == [b"FALSE", b"TRUE"]:
            return "bool", bsize, q
        return base, bsize, q                                                   # other enums: expose the integer codes
    raise NotImplementedError(f"HDF5 datatype class {cls}")


def _space_encode(shape, maxshape=None) -> bytes:
    rank = len(shape)
    if rank and maxshape is None:
        maxshape = shape                                                        # libhdf5 always stores the maximum dims
    flags = 1 if maxshape is not None and rank else 0
    b = struct.pack("<BBBB4x", 1, rank, flags, 0) + b"".join(struct.pack("<Q", d) for d in shape)
    if flags:
        b += b"".join(struct.pack("<Q", UNDEF if m is None else m) for m in maxshape)
    return b


def _space_decode(buf, off=0):
    ver, rank, flags = struct.unpack_from("<BBB", buf, off)
    if ver == 1:
        p = off + 8
    elif ver == 2:
        p = off + 4
        if buf[off + 3] == 2:                                                   # null dataspace
            return None, None
    else:
        raise NotImplementedError(f"dataspace message version {ver}")
    shape = struct.unpack_from(f"<{rank}Q", buf, p)
    maxshape = None
    if flags & 1:
        maxshape = tuple(None if m == UNDEF else m for m in struct.unpack_from(f"<{rank}Q", buf, p + 8 * rank))
    return tuple(shape), maxshape


def _lzf_decompress(src: bytes, out_len: int) -> bytes:
    out = bytearray()
    i, n = 0, len(src)
    while i < n:
        ctrl = src[i]
        i += 1
        if ctrl < 32:
            out += src[i:i + ctrl + 1]
            i += ctrl + 1
        else:
            ln = ctrl >> 5
            if ln == 7:
                ln += src[i]
                i += 1
            ref = len(out) - ((ctrl & 31) << 8) - src[i] - 1
            i += 1
            for _ in range(ln + 2):
                out.append(out[ref])
                ref += 1
    if len(out) != out_len:
        raise OSError("lzf: corrupt chunk")
    return bytes(out)


# ------------------------------------------------------------------------------------------------ attributes
class AttributeManager:
    def __init__(self, owner):
        self._o = owner

    def _d(self):
        return self._o._attrs

    def __getitem__(self, k):
        return self._d()[k]

    def __setitem__(self, k, v):
        self._o._file._check_writable()
        if isinstance(v, (bool, np.bool_)):
            v = np.bool_(v)
        elif isinstance(v, (int, np.integer)):
            v = np.int64(v)
        elif isinstance(v, (float, np.floating)):
            v = np.float64(v)
        elif isinstance(v, bytes):
            v = v.decode("utf-8")
        elif not isinstance(v, str):
            v = np.asarray(v)
            if v.dtype.kind not in "fiu":
                raise TypeError(f"attribute of dtype {v.dtype} not supported")
        self._d()[k] = v
        self._o._touch()

    def __contains__(self, k):
        return k in self._d()

    def __iter__(self):
        return iter(self._d())

    def __len__(self):
        return len(self._d())

    def keys(self):
        return self._d().keys()

    def items(self):
        return self._d().items()

    def get(self, k, default=None):
        return self._d().get(k, default)


# ------------------------------------------------------------------------------------------------ objects
class _Node:
    def __init__(self, file, parent, name):
        self._file, self._parent, self._name = file, parent, name
        self._attrs = OrderedDict()
        self._addr = None          # committed object header address (None: never written)
        self._extent = []          # [(addr, size)] regions owned by the committed copy of this object's metadata
        self._dirty = True

    @property
    def attrs(self):
        return AttributeManager(self)

    @property
    def name(self):
        if self._parent is None:
            return "/"
        p = self._parent.name
        return (p if p.endswith("/") else p + "/") + self._name

    @property
    def file(self):
        return self._file

    @property
    def parent(self):
        return self._parent or self

    def _touch(self):
        n = self
        while n is not None:
            n._dirty = True
            n = n._parent


class Group(_Node):
    def __init__(self, file, parent, name):
        super().__init__(file, parent, name)
        self._children = None      # name -> _Node | int address (lazy); None = not loaded yet
        self._btree = self._heap = None

    # -- reading ------------------------------------------------------------------------------------------
    def _load(self):
        if self._children is None:
            self._children = OrderedDict()
            if self._btree is not None:
                for nm, addr in self._file._iter_symbols(self._btree, self._heap):
                    self._children[nm] = addr
        return self._children

    def _child(self, nm):
        ch = self._load()
        v = ch[nm]
        if isinstance(v, int):
            v = ch[nm] = self._file._read_object(v, self, nm)
        return v

    def keys(self):
        return list(self._load().keys())

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self._load())

    def __contains__(self, path):
        try:
            self[path]
            return True
        except KeyError:
            return False

    def __getitem__(self, path):
        node = self._file if path.startswith("/") else self
        for part in path.split("/"):
            if not part:
                continue
            if not isinstance(node, Group) or part not in node._load():
                raise KeyError(f"unable to open object '{path}' (component '{part}' not found)")
            node = node._child(part)
        return node

    def get(self, path, default=None):
        try:
            return self[path]
        except KeyError:
            return default

    def values(self):
        return [self._child(k) for k in self.keys()]

    def items(self):
        return [(k, self._child(k)) for k in self.keys()]

    def visititems(self, fn, _prefix=""):
        for k, v in self.items():
            rel = _prefix + k
            r = fn(rel, v)
            if r is None and isinstance(v, Group):
                r = v.visititems(fn, rel + "/")
            if r is not None:
                return r
        return None

    # -- writing ------------------------------------------------------------------------------------------
    def _new_child(self, path, make):
        self._file._check_writable()
        parts = [p for p in path.split("/") if p]
        node = self
        for p in parts[:-1]:
            node = node.require_group(p)
        nm = parts[-1]
        if nm in node._load():
            raise ValueError(f"unable to create '{path}' (name already exists)")
        obj = make(node, nm)
        node._children[nm] = obj
        obj._touch()
        node._touch()
        return obj

    def create_group(self, path):
        return self._new_child(path, lambda parent, nm: Group(self._file, parent, nm)._as_new())

    def _as_new(self):
        self._children = OrderedDict()
        return self

    def require_group(self, path):
        if path in self:
            g = self[path]
            if not isinstance(g, Group):
                raise TypeError(f"'{path}' is not a group")
            return g
        return self.create_group(path)

    def create_dataset(self, path, shape=None, dtype=None, data=None, maxshape=None, chunks=None, compression=None,
                       compression_opts=None, **unused):
        if unused:
            raise TypeError(f"create_dataset: unsupported options {sorted(unused)}")
        return self._new_child(path, lambda parent, nm: Dataset._create(self._file, parent, nm, shape, dtype, data, maxshape,
                                                                        chunks, compression, compression_opts))

    def __delitem__(self, nm):
        self._file._check_writable()
        del self._load()[nm]       # the object's space stays allocated (as with libhdf5 without repacking)
        self._dirty = False
        self._touch()


class Dataset(_Node):
    def __init__(self, file, parent, name):
        super().__init__(file, parent, name)
        self.shape = self.maxshape = self.chunks = None
        self._dt = None            # numpy dtype | 'vlen_str' | ('fixed_str', n, utf8)
        self._filters = []         # [(id, [client data])]
        self._layout = None        # ('contiguous', addr, size) | ('chunked', btree addr) | ('compact', bytes)
        self._index = None         # chunked: {chunk offset tuple: (addr, nbytes, filter mask)}
        self._strings = None       # vlen_str datasets written in this session: list of str

    # -- properties ---------------------------------------------------------------------------------------
    @property
    def dtype(self):
        if self._dt == _VLEN_STR:
            return np.dtype("O")
        if isinstance(self._dt, tuple):
            return np.dtype(f"S{self._dt[1]}")
        return self._dt

    @property
    def compression(self):
        ids = [f[0] for f in self._filters]
        return "gzip" if 1 in ids else ("lzf" if 32000 in ids else None)

    @property
    def compression_opts(self):
        for fid, cd in self._filters:
            if fid == 1:
                return cd[0] if cd else None
        return None

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return int(np.prod(self.shape, dtype=np.int64))

    def __len__(self):
        return self.shape[0]

    # -- creation -----------------------------------------------------------------------------------------
    @classmethod
    def _create(cls, file, parent, nm, shape, dtype, data, maxshape, chunks, compression, compression_opts):
        d = cls(file, parent, nm)
        if data is not None:
            if _is_str_dtype(dtype) or (isinstance(data, np.ndarray) and data.dtype.kind in "OUS") or \
                    (isinstance(data, (list, tuple)) and data and isinstance(data[0], (str, bytes))):
                strs = [s.decode("utf-8") if isinstance(s, bytes) else str(s) for s in np.asarray(data, dtype=object).ravel()]
                d._dt, d._strings = _VLEN_STR, strs
                d.shape = tuple(np.asarray(data, dtype=object).shape)
                d._layout = None
                return d
            data = np.asarray(data, dtype=dtype)
            if data.dtype == np.bool_:
                raise TypeError("bool datasets are not supported")
            shape = data.shape if shape is None else ((shape,) if isinstance(shape, int) else tuple(shape))
            data = data.reshape(shape)
            dtype = data.dtype
        if shape is None:
            raise TypeError("create_dataset needs shape= or data=")
        if _is_str_dtype(dtype):
            raise NotImplementedError("empty variable-length string datasets")
        d.shape = (shape,) if isinstance(shape, int) else tuple(int(s) for s in shape)
        dt = np.dtype("f4" if dtype is None else dtype)
        if dt.kind not in "fiu":
            raise TypeError(f"h5lite cannot store dtype {dt}")
        d._dt = dt.newbyteorder("<") if dt.byteorder == ">" else dt
        if maxshape is not None:
            maxshape = (maxshape,) if isinstance(maxshape, int) else tuple(maxshape)
            if len(maxshape) != len(d.shape):
                raise ValueError("maxshape rank mismatch")
            d.maxshape = tuple(None if m is None else int(m) for m in maxshape)
        else:
            d.maxshape = d.shape
        if compression is True:
            compression = "gzip"
        if compression not in (None, "gzip", "lzf"):
            raise ValueError(f"h5lite writes gzip, lzf-declared or uncompressed data; got compression={compression!r}")
        if compression == "gzip":
            d._filters = [(1, [4 if compression_opts is None else int(compression_opts)])]
        lzf = compression == "lzf"
        need_chunks = compression is not None or d.maxshape != d.shape
        if chunks is True or (chunks is None and need_chunks):
            chunks = cls._guess_chunks(d.shape, d.maxshape, d._dt.itemsize)
        if chunks is not None:
            chunks = (chunks,) if isinstance(chunks, int) else tuple(int(c) for c in chunks)
            if len(chunks) != len(d.shape) or any(c <= 0 for c in chunks):
                raise ValueError("chunk shape must be positive and match the dataset rank")
            d.chunks, d._index, d._layout = chunks, {}, ("chunked", None)
            if lzf:
                # h5py's LZF filter (id 32000; client data: filter revision 4, liblzf 0x0105, chunk bytes).  h5lite does not
                # compress through it: every chunk is stored raw with the filter's bit set in the chunk's filter mask, which
                # is exactly what h5py writes whenever LZF cannot shrink a chunk (float embeddings) -- readable by h5py and,
                # because skipped filters are never invoked, by a stock libhdf5 without the plugin.
                d._filters = [(32000, [4, 0x0105, int(np.prod(chunks)) * d._dt.itemsize])]
        else:
            d._layout = ("contiguous", UNDEF, 0)
        if data is not None and data.size:
            d._write_region(tuple(slice(0, s) for s in d.shape), data)
        return d

    @staticmethod
    def _guess_chunks(shape, maxshape, itemsize):
        ch = [max(1, s if m is not None else max(s, 1024)) for s, m in zip(shape, maxshape)]
        target = 64 * 1024
        i = 0
        while int(np.prod(ch)) * itemsize > target and any(c > 1 for c in ch):
            if ch[i % len(ch)] > 1:
                ch[i % len(ch)] = (ch[i % len(ch)] + 1) // 2
            i += 1
        return tuple(ch)

    # -- raw element codec --------------------------------------------------------------------------------
    def _decode_chunk(self, raw: bytes, mask: int, nelem: int) -> bytes:
        for k in range(len(self._filters) - 1, -1, -1):
            if mask & (1 << k):
                continue
            fid, cd = self._filters[k]
            if fid == 1:
                raw = zlib.decompress(raw)
            elif fid == 2:                                                       # shuffle
                es = cd[0] if cd else self._elem_size()
                a = np.frombuffer(raw, dtype=np.uint8)
                n = len(a) // es
                raw = a[:n * es].reshape(es, n).T.tobytes() + a[n * es:].tobytes()
            elif fid == 3:                                                       # fletcher32: strip the checksum
                raw = raw[:-4]
            elif fid == 32000:
                raw = _lzf_decompress(raw, nelem * self._elem_size())
            else:
                raise NotImplementedError(f"HDF5 filter {fid}")
        return raw

    def _encode_chunk(self, raw: bytes):
        """-> (stored bytes, filter mask): bit k of the mask set = filter k of the pipeline was skipped for this chunk."""
        mask = 0
        for k, (fid, cd) in enumerate(self._filters):
            if fid == 1:
                raw = zlib.compress(raw, cd[0] if cd else 4)
            elif fid == 32000:
                mask |= 1 << k                  # stored uncompressed (see _create)
            else:
                raise NotImplementedError(f"writing through HDF5 filter {fid}")
        return raw, mask

    def _elem_size(self):
        if self._dt == _VLEN_STR:
            return 16
        if isinstance(self._dt, tuple):
            return self._dt[1]
        return self._dt.itemsize

    def _np_storage_dtype(self):
        if self._dt == _VLEN_STR:
            return np.dtype([("len", "<u4"), ("addr", "<u8"), ("idx", "<u4")])
        if isinstance(self._dt, tuple):
            return np.dtype(f"S{self._dt[1]}")
        return self._dt

    # -- reading ------------------------------------------------------------------------------------------
    def _read_all_storage(self, rows=None):
        """Storage-dtype ndarray of the whole dataset, or of rows [r0, r1) of axis 0."""
        f = self._file
        sdt = self._np_storage_dtype()
        shape = self.shape
        r0, r1 = (0, shape[0] if shape else 1) if rows is None else rows
        if not shape:
            kind = self._layout[0]
            raw = self._layout[1] if kind == "compact" else (f._pread(self._layout[1], sdt.itemsize) if self._layout[1] != UNDEF
                                                              else b"\0" * sdt.itemsize)
            return np.frombuffer(raw, dtype=sdt, count=1).reshape(())
        out_shape = (max(0, r1 - r0),) + tuple(shape[1:])
        n_out = int(np.prod(out_shape, dtype=np.int64))
        kind = self._layout[0]
        if kind in ("contiguous", "compact"):
            row_bytes = int(np.prod(shape[1:], dtype=np.int64)) * sdt.itemsize
            if kind == "compact":
                raw = self._layout[1][r0 * row_bytes:r1 * row_bytes]
            elif self._layout[1] == UNDEF or n_out == 0:
                raw = b"\0" * (n_out * sdt.itemsize)
            else:
                raw = f._pread(self._layout[1] + r0 * row_bytes, n_out * sdt.itemsize)
            return np.frombuffer(raw, dtype=sdt, count=n_out).reshape(out_shape).copy()
        out = np.zeros(out_shape, dtype=sdt)
        if n_out == 0:
            return out
        ch = self.chunks
        nelem = int(np.prod(ch))
        for off, (addr, nbytes, mask) in self._chunk_index().items():
            if off[0] + ch[0] <= r0 or off[0] >= r1:
                continue
            raw = self._decode_chunk(f._pread(addr, nbytes), mask, nelem)
            block = np.frombuffer(raw, dtype=sdt, count=nelem).reshape(ch)
            src, dst = [], []
            for ax, (o, c, s) in enumerate(zip(off, ch, shape)):
                lo = max(o, r0) if ax == 0 else o
                hi = min(o + c, r1 if ax == 0 else s)
                if hi <= lo:
                    break
                src.append(slice(lo - o, hi - o))
                dst.append(slice(lo - (r0 if ax == 0 else 0), hi - (r0 if ax == 0 else 0)))
            else:
                out[tuple(dst)] = block[tuple(src)]
        return out

    def _chunk_index(self):
        if self._index is None:
            self._index = {}
            if self._layout[1] not in (None, UNDEF):
                self._file._walk_chunk_btree(self._layout[1], len(self.shape), self._index)
        return self._index

    def _to_user(self, arr):
        if self._dt == _VLEN_STR:
            if self._strings is not None:
                flat = [s.encode("utf-8") for s in self._strings]
                return np.array(flat, dtype=object).reshape(arr.shape if arr is not None else self.shape)
            out = np.empty(arr.shape, dtype=object)
            for idx in np.ndindex(arr.shape):
                e = arr[idx]
                out[idx] = self._file._gheap_get(int(e["addr"]), int(e["idx"])) if e["len"] else b""
            return out
        return arr

    def __getitem__(self, key):
        if self._dt == _VLEN_STR and self._strings is not None:
            full = np.array([s.encode("utf-8") for s in self._strings], dtype=object).reshape(self.shape)
            return full[key if key is not Ellipsis else ()]
        if not isinstance(key, tuple):
            key = (key,)
        rows = None
        rest = key
        if self.shape and key and key[0] is not Ellipsis:
            k0 = key[0]
            n0 = self.shape[0]
            if isinstance(k0, (int, np.integer)):
                k0 = int(k0) + (n0 if k0 < 0 else 0)
                if not 0 <= k0 < n0:
                    raise IndexError(f"index {key[0]} out of range for axis 0 with size {n0}")
                rows, rest = (k0, k0 + 1), (0,) + tuple(key[1:])
            elif isinstance(k0, slice):
                a, b, st = k0.indices(n0)
                if st > 0:
                    rows, rest = (a, max(a, b)), (slice(0, max(0, b - a), st),) + tuple(key[1:])
        arr = self._to_user(self._read_all_storage(rows))
        res = arr[rest] if rest != (Ellipsis,) else arr
        if isinstance(res, np.ndarray) and res.shape == () and self._dt != _VLEN_STR:
            return res[()]
        return res

    def asstr(self):
        ds = self

        class _V:
            def __getitem__(self, key):
                a = ds[key]
                if isinstance(a, bytes):
                    return a.decode("utf-8")
                return np.array([x.decode("utf-8") for x in a.ravel()], dtype=object).reshape(a.shape)
        return _V()

    def __array__(self, dtype=None, copy=None):
        a = self[...]
        return a.astype(dtype) if dtype is not None else a

    # -- writing ------------------------------------------------------------------------------------------
    def resize(self, size, axis=None):
        self._file._check_writable()
        if self.chunks is None:
            raise TypeError("only chunked datasets can be resized")
        if axis is not None:
            new = list(self.shape)
            new[axis] = int(size)
        else:
            new = [int(s) for s in ((size,) if isinstance(size, int) else size)]
        if len(new) != len(self.shape):
            raise ValueError("resize: rank mismatch")
        for n, m in zip(new, self.maxshape):
            if m is not None and n > m:
                raise ValueError(f"unable to set extent {tuple(new)}: exceeds maxshape {self.maxshape}")
        old = self.shape
        self.shape = tuple(new)
        if any(n < o for n, o in zip(new, old)):       # shrinking: drop chunks that fall outside entirely
            idx = self._chunk_index()
            for off in [o for o in idx if any(oo >= n for oo, n in zip(o, new))]:
                a, nb, _ = idx.pop(off)
                self._file._free_later(a, nb)
        self._touch()

    def __setitem__(self, key, value):
        self._file._check_writable()
        if self._dt == _VLEN_STR:
            raise NotImplementedError("variable-length string datasets are written once, through create_dataset(data=...)")
        if not isinstance(key, tuple):
            key = (key,)
        if Ellipsis in key:
            i = key.index(Ellipsis)
            key = key[:i] + (slice(None),) * (len(self.shape) - len(key) + 1) + key[i + 1:]
        key = key + (slice(None),) * (len(self.shape) - len(key))
        region = []
        for k, n in zip(key, self.shape):
            if isinstance(k, (int, np.integer)):
                k = int(k) + (n if k < 0 else 0)
                region.append(slice(k, k + 1))
            else:
                a, b, st = k.indices(n)
                if st != 1:
                    raise NotImplementedError("strided writes")
                region.append(slice(a, max(a, b)))
        tgt_shape = tuple(r.stop - r.start for r in region)
        kept = tuple(n for n, k in zip(tgt_shape, key) if not isinstance(k, (int, np.integer)))
        value = np.broadcast_to(np.asarray(value, dtype=self._dt), kept).reshape(tgt_shape)
        self._write_region(tuple(region), value)

    def _write_region(self, region, value):
        f = self._file
        value = np.ascontiguousarray(value, dtype=self._dt)
        if value.size == 0:
            return
        if self.chunks is None:                                                   # contiguous: allocate once, write in place
            total = self.size * self._dt.itemsize
            if self._layout[1] == UNDEF:
                addr = f._alloc(total)
                self._layout = ("contiguous", addr, total)
                if value.size != self.size:
                    f._pwrite(addr, b"\0" * total)
            if value.size == self.size:
                f._pwrite(self._layout[1], value.tobytes())
            else:
                full = self._read_all_storage()
                full[region] = value
                f._pwrite(self._layout[1], full.tobytes())
            self._touch()
            return
        ch = self.chunks
        idx = self._chunk_index()
        nelem = int(np.prod(ch))
        grid = [range(r.start // c, (r.stop - 1) // c + 1) for r, c in zip(region, ch)]
        for cidx in np.ndindex(*[len(g) for g in grid]):
            off = tuple(g[i] * c for g, i, c in zip(grid, cidx, ch))
            src, dst, full_cover = [], [], True
            for o, c, r in zip(off, ch, region):
                lo, hi = max(o, r.start), min(o + c, r.stop)
                src.append(slice(lo - r.start, hi - r.start))
                dst.append(slice(lo - o, hi - o))
                full_cover &= (lo == o and hi == o + c)
            if full_cover:
                block = value[tuple(src)]
            else:
                if off in idx:
                    a, nb, mask = idx[off]
                    block = np.frombuffer(self._decode_chunk(f._pread(a, nb), mask, nelem), dtype=self._dt, count=nelem).reshape(ch).copy()
                else:
                    block = np.zeros(ch, dtype=self._dt)
                block[tuple(dst)] = value[tuple(src)]
            raw, fmask = self._encode_chunk(np.ascontiguousarray(block).tobytes())
            if off in idx:
                f._free_later(idx[off][0], idx[off][1])
            addr = f._alloc(len(raw))
            f._pwrite(addr, raw)
            idx[off] = (addr, len(raw), fmask)
        self._touch()


# ------------------------------------------------------------------------------------------------ file
class File(Group):
    def __init__(self, path, mode="r", libver=None, **unused):
        if mode not in ("r", "w", "a", "r+", "x", "w-"):
            raise ValueError(f"invalid mode {mode!r}")
        _Node.__init__(self, self, None, "")
        self._children = None
        self._btree = self._heap = None
        self.filename, self.mode = str(path), mode
        self._free, self._pending_free = [], []
        self._gheaps = {}
        exists = os.path.exists(path)
        if mode in ("x", "w-") and exists:
            raise FileExistsError(path)
        if mode in ("r", "r+") and not exists:
            raise FileNotFoundError(f"unable to open file '{path}'")
        self._writable = mode != "r"
        if mode == "r":
            self._fd = open(path, "rb")
        elif mode in ("w", "x", "w-") or not exists:
            self._fd = open(path, "w+b")
            self._children = OrderedDict()
            self._eof = 96
            self._fd.write(b"\0" * 96)
            self.flush()
            return
        else:
            self._fd = open(path, "r+b")
        self._open_existing()

    # -- low level ----------------------------------------------------------------------------------------
    def _check_writable(self):
        if self._fd is None:
            raise ValueError("file is closed")
        if not self._writable:
            raise OSError("file was opened read-only")

    def _pread(self, addr, n):
        self._fd.seek(addr)
        b = self._fd.read(n)
        if len(b) != n:
            raise OSError(f"truncated HDF5 file: wanted {n} bytes at {addr}")
        return b

    def _pwrite(self, addr, b):
        self._fd.seek(addr)
        self._fd.write(b)

    def _alloc(self, n):
        n = max(8, (n + 7) & ~7)
        for i, (a, sz) in enumerate(self._free):
            if sz >= n:
                if sz == n:
                    self._free.pop(i)
                else:
                    self._free[i] = (a + n, sz - n)
                return a
        a = self._eof
        self._eof += n
        return a

    def _free_later(self, addr, n):
        """Region becomes reusable after the next commit (until then the committed tree may still reference it)."""
        if addr not in (None, UNDEF) and n:
            self._pending_free.append((addr, (n + 7) & ~7))

    # -- open ---------------------------------------------------------------------------------------------
    def _open_existing(self):
        hdr = self._pread(0, 8)
        base = 0
        if hdr != SIG:
            for base in (512, 1024, 2048, 4096):
                try:
                    if self._pread(base, 8) == SIG:
                        raise NotImplementedError("HDF5 user block")
                except OSError:
                    break
            raise OSError(f"'{self.filename}' is not an HDF5 file (bad signature)")
        ver = self._pread(8, 1)[0]
        if ver >= 2:
            raise NotImplementedError(f"HDF5 superblock version {ver} ('latest' file format) is not supported by h5lite; "
                                      "re-save with libver='earliest'")
        sb = self._pread(0, 24 + (4 if ver == 1 else 0) + 32 + 40)
        so, sl = sb[13], sb[14]
        if (so, sl) != (8, 8):
            raise NotImplementedError("HDF5 files with offset/length sizes other than 8 bytes")
        p = 24 + (4 if ver == 1 else 0)
        _base, _fs, eof, _drv = struct.unpack_from("<4Q", sb, p)
        _nameoff, root_hdr, ctype = struct.unpack_from("<QQI", sb, p + 32)
        self._eof = max(eof, os.fstat(self._fd.fileno()).st_size if self._writable else eof)
        self._eof = (self._eof + 7) & ~7
        self._read_header_into(self, root_hdr)
        self._addr, self._dirty = root_hdr, False
        if self._writable:
            self._load_recursive(self)

    def _load_recursive(self, g):
        for k in g.keys():
            c = g._child(k)
            if isinstance(c, Group):
                self._load_recursive(c)
            else:
                c._chunk_index() if c.chunks is not None else None

    # -- object header parsing ----------------------------------------------------------------------------
    def _messages(self, addr):
        ver = self._pread(addr, 1)[0]
        if ver != 1:
            if self._pread(addr, 4) == b"OHDR":
                raise NotImplementedError("version 2 object headers ('latest' file format)")
            raise OSError(f"bad object header at {addr}")
        _v, _r, nmsg, _ref, hsize = struct.unpack("<BBHII", self._pread(addr, 12))
        blocks = [(addr + 16, hsize)]
        extent = [(addr, 16 + hsize)]
        msgs = []
        while blocks and len(msgs) < nmsg:
            a, n = blocks.pop(0)
            buf = self._pread(a, n)
            p = 0
            while p + 8 <= n and len(msgs) < nmsg:
                mtype, msize, mflags = struct.unpack_from("<HHB", buf, p)
                body = buf[p + 8:p + 8 + msize]
                p += 8 + msize
                if mflags & 2:
                    raise NotImplementedError("shared object header messages")
                if mtype == 0x10:
                    ca, cl = struct.unpack_from("<QQ", body)
                    blocks.append((ca, cl))
                    extent.append((ca, cl))
                msgs.append((mtype, body))
        return msgs, extent

    def _read_object(self, addr, parent, nm):
        msgs, _ = self._messages(addr)
        types = {m[0] for m in msgs}
        node = Group(self, parent, nm) if 0x11 in types else Dataset(self, parent, nm)
        if 0x11 not in types and 0x08 not in types:
            if 0x02 in types or 0x06 in types:
                raise NotImplementedError("new-style (link message) groups")
            raise OSError(f"object at {addr} is neither a group nor a dataset")
        self._read_header_into(node, addr)
        node._addr, node._dirty = addr, False
        return node

    def _read_header_into(self, node, addr):
        msgs, extent = self._messages(addr)
        node._extent = list(extent)
        for mtype, body in msgs:
            if mtype == 0x11:
                node._btree, node._heap = struct.unpack_from("<QQ", body)
            elif mtype == 0x0C:
                k, v = self._attr_decode(body)
                node._attrs[k] = v
            elif mtype == 0x01 and isinstance(node, Dataset):
                node.shape, mx = _space_decode(body)
                node.maxshape = mx if mx is not None else node.shape
            elif mtype == 0x03 and isinstance(node, Dataset):
                node._dt = _dt_decode(body)[0]
            elif mtype == 0x0B and isinstance(node, Dataset):
                node._filters = self._filters_decode(body)
            elif mtype == 0x08 and isinstance(node, Dataset):
                ver, cls = body[0], body[1]
                if ver != 3:
                    raise NotImplementedError(f"data layout message version {ver}")
                if cls == 0:
                    n = struct.unpack_from("<H", body, 2)[0]
                    node._layout = ("compact", bytes(body[4:4 + n]))
                elif cls == 1:
                    a, n = struct.unpack_from("<QQ", body, 2)
                    node._layout = ("contiguous", a, n)
                elif cls == 2:
                    nd = body[2]
                    bt = struct.unpack_from("<Q", body, 3)[0]
                    dims = struct.unpack_from(f"<{nd}I", body, 11)
                    node.chunks = tuple(dims[:-1])
                    node._layout = ("chunked", bt)
                else:
                    raise NotImplementedError(f"layout class {cls}")
        if isinstance(node, Dataset) and (node.shape is None or node._dt is None or node._layout is None):
            raise OSError(f"dataset header at {addr} is incomplete")

    @staticmethod
    def _filters_decode(body):
        ver, nf = body[0], body[1]
        p = 8 if ver == 1 else 2
        out = []
        for _ in range(nf):
            fid = struct.unpack_from("<H", body, p)[0]
            if ver == 1 or fid >= 256:
                nlen = struct.unpack_from("<H", body, p + 2)[0]
                p += 4
            else:
                nlen = 0
                p += 2
            _flags, ncd = struct.unpack_from("<HH", body, p)
            p += 4
            p += ((nlen + 7) // 8) * 8 if ver == 1 else nlen
            cd = list(struct.unpack_from(f"<{ncd}I", body, p))
            p += 4 * ncd
            if ver == 1 and ncd % 2:
                p += 4
            out.append((fid, cd))
        return out

    def _attr_decode(self, body):
        ver = body[0]
        if ver == 1:
            nsz, dsz, ssz = struct.unpack_from("<HHH", body, 2)
            p = 8
            rnd = lambda n: (n + 7) & ~7
        elif ver in (2, 3):
            nsz, dsz, ssz = struct.unpack_from("<HHH", body, 2)
            p = 8 + (1 if ver == 3 else 0)
            rnd = lambda n: n
        else:
            raise NotImplementedError(f"attribute message version {ver}")
        name = body[p:p + nsz].split(b"\0")[0].decode("utf-8")
        p += rnd(nsz)
        dt, esz, _ = _dt_decode(body, p)
        p += rnd(dsz)
        shape, _ = _space_decode(body, p)
        p += rnd(ssz)
        if shape is None:
            return name, None
        n = int(np.prod(shape, dtype=np.int64))
        raw = body[p:p + n * esz]
        if dt == _VLEN_STR:
            vals = []
            for i in range(n):
                ln, a, ix = struct.unpack_from("<IQI", raw, 16 * i)
                vals.append(self._gheap_get(a, ix).decode("utf-8") if ln else "")
            return name, (vals[0] if shape == () else np.array(vals, dtype=object).reshape(shape))
        if isinstance(dt, tuple):
            vals = [raw[i * esz:(i + 1) * esz].split(b"\0")[0] for i in range(n)]
            vals = [v.decode("utf-8") for v in vals]
            return name, (vals[0] if shape == () else np.array(vals, dtype=object).reshape(shape))
        if dt == "bool":
            a = np.frombuffer(raw, dtype=np.int8, count=n).astype(np.bool_).reshape(shape)
        else:
            a = np.frombuffer(raw, dtype=dt, count=n).reshape(shape).copy()
        return name, (a[()] if shape == () else a)

    # -- group / chunk B-trees, heaps ---------------------------------------------------------------------
    def _heap_data(self, heap_addr):
        h = self._pread(heap_addr, 32)
        if h[:4] != b"HEAP":
            raise OSError("bad local heap signature")
        size, _free, data_addr = struct.unpack_from("<QQQ", h, 8)
        return self._pread(data_addr, size)

    def _iter_symbols(self, btree, heap):
        data = self._heap_data(heap)
        stack = [btree]
        out = []
        while stack:
            a = stack.pop()
            h = self._pread(a, 24)
            if h[:4] != b"TREE":
                raise OSError("bad B-tree signature")
            _t, level, n = struct.unpack_from("<BBH", h, 4)
            body = self._pread(a + 24, (2 * n + 1) * 8)
            kids = [struct.unpack_from("<Q", body, 8 + 16 * i)[0] for i in range(n)]
            if level > 0:
                stack.extend(reversed(kids))
                continue
            for k in kids:
                s = self._pread(k, 8)
                if s[:4] != b"SNOD":
                    raise OSError("bad symbol node signature")
                ns = struct.unpack_from("<H", s, 6)[0]
                ents = self._pread(k + 8, 40 * ns)
                for i in range(ns):
                    noff, oh = struct.unpack_from("<QQ", ents, 40 * i)
                    e = data.index(b"\0", noff)
                    out.append((data[noff:e].decode("utf-8"), oh))
        return out

    def _walk_chunk_btree(self, addr, rank, index):
        if addr == UNDEF:
            return
        h = self._pread(addr, 24)
        if h[:4] != b"TREE":
            raise OSError("bad chunk B-tree signature")
        _t, level, n = struct.unpack_from("<BBH", h, 4)
        ksz = 8 + 8 * (rank + 1)
        body = self._pread(addr + 24, n * (ksz + 8) + ksz)
        for i in range(n):
            p = i * (ksz + 8)
            csize, mask = struct.unpack_from("<II", body, p)
            off = struct.unpack_from(f"<{rank}Q", body, p + 8)
            child = struct.unpack_from("<Q", body, p + ksz)[0]
            if level > 0:
                self._walk_chunk_btree(child, rank, index)
            else:
                index[tuple(off)] = (child, csize, mask)

    def _gheap_get(self, addr, idx):
        objs = self._gheaps.get(addr)
        if objs is None:
            h = self._pread(addr, 16)
            if h[:4] != b"GCOL":
                raise OSError("bad global heap signature")
            size = struct.unpack_from("<Q", h, 8)[0]
            buf = self._pread(addr, size)
            objs, p = {}, 16
            while p + 16 <= size:
                i, _ref, osz = struct.unpack_from("<HH4xQ", buf, p)
                if i == 0:
                    break
                objs[i] = buf[p + 16:p + 16 + osz]
                p += 16 + ((osz + 7) & ~7)
            self._gheaps[addr] = objs
        return objs[idx]

    # -- commit -------------------------------------------------------------------------------------------
    def flush(self):
        self._check_writable()
        if self._dirty:
            self._write_node(self)
        sb = SIG + struct.pack("<BBBBBBBBHHI", 0, 0, 0, 0, 0, 8, 8, 0, LEAF_K, INTERNAL_K, 0)
        sb += struct.pack("<QQQQ", 0, UNDEF, self._eof, UNDEF)
        sb += struct.pack("<QQII", 0, self._addr, 1, 0) + struct.pack("<QQ", self._btree, self._heap)
        assert len(sb) == 96
        self._fd.seek(0, os.SEEK_END)
        if self._fd.tell() < self._eof:
            self._fd.truncate(self._eof)
        self._pwrite(0, sb)
        self._fd.flush()
        self._free.extend(self._pending_free)
        self._pending_free = []
        self._free.sort()
        merged = []
        for a, n in self._free:
            if merged and merged[-1][0] + merged[-1][1] == a:
                merged[-1] = (merged[-1][0], merged[-1][1] + n)
            else:
                merged.append((a, n))
        self._free = merged

    def close(self):
        if self._fd is None:
            return
        if self._writable:
            self.flush()
            os.fsync(self._fd.fileno())
        self._fd.close()
        self._fd = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __bool__(self):
        return self._fd is not None

    # -- object serialisation -----------------------------------------------------------------------------
    def _put(self, node, blob):
        a = self._alloc(len(blob))
        self._pwrite(a, blob)
        node._new_extent.append((a, (len(blob) + 7) & ~7))
        return a

    def _gcol(self, node, strings):
        """One global heap collection holding ``strings`` -> [(addr, index)] per string."""
        body = b""
        refs = []
        for i, s in enumerate(strings, 1):
            body += struct.pack("<HH4xQ", i, 1, len(s)) + _pad8(s)
        size = max(4096, 16 + len(body) + 16)
        size = (size + 7) & ~7
        free = size - 16 - len(body)
        blob = b"GCOL" + struct.pack("<B3xQ", 1, size) + body + struct.pack("<HH4xQ", 0, 0, free) + b"\0" * (free - 16)
        a = self._put(node, blob)
        for i in range(1, len(strings) + 1):
            refs.append((a, i))
        return refs

    def _attr_messages(self, node):
        strs = [v.encode("utf-8") for v in node._attrs.values() if isinstance(v, str)]
        refs = iter(self._gcol(node, strs)) if strs else iter(())
        msgs = []
        for k, v in node._attrs.items():
            name = k.encode("utf-8") + b"\0"
            if isinstance(v, str):
                a, i = next(refs)
                dt, sp, data = _dt_encode(_STR), _space_encode(()), struct.pack("<IQI", len(v.encode("utf-8")), a, i)
            elif isinstance(v, np.bool_):
                dt, sp, data = _dt_encode("bool"), _space_encode(()), bytes([int(v)])
            elif v is None:
                continue
            else:
                v = np.asarray(v)
                v = v.astype(v.dtype.newbyteorder("<")) if v.dtype.byteorder == ">" else v
                dt, sp, data = _dt_encode(v.dtype), _space_encode(v.shape), np.ascontiguousarray(v).tobytes()
            body = struct.pack("<BxHHH", 1, len(name), len(dt), len(sp)) + _pad8(name) + _pad8(dt) + _pad8(sp) + data
            msgs.append((0x0C, body))
        return msgs

    def _header(self, node, msgs):
        body = b""
        for mtype, mb in msgs:
            mb = _pad8(mb)
            if len(mb) > 65528:
                raise ValueError("object header message larger than 64 KiB (attribute too large)")
            body += struct.pack("<HHB3x", mtype, len(mb), 0) + mb
        blob = struct.pack("<BxHII4x", 1, len(msgs), 1, len(body)) + body
        return self._put(node, blob)

    def _write_node(self, node):
        node._new_extent = []
        if isinstance(node, Group):
            ch = node._load()
            entries = []
            for nm in ch:
                c = node._child(nm)
                if c._dirty or c._addr is None:
                    self._write_node(c)
                entries.append((nm.encode("utf-8"), c))
            entries.sort(key=lambda e: e[0])
            node._btree, node._heap = self._write_symbol_table(node, entries)
            msgs = [(0x11, struct.pack("<QQ", node._btree, node._heap))] + self._attr_messages(node)
        else:
            msgs = self._dataset_messages(node) + self._attr_messages(node)
        addr = self._header(node, msgs)
        for a, n in node._extent:
            self._free_later(a, n)
        node._extent, node._addr, node._dirty = node._new_extent, addr, False
        del node._new_extent

    def _write_symbol_table(self, node, entries):
        # local heap: offset 0 = "" (8 zero bytes), then the names
        heap = bytearray(8)
        offs = []
        for nm, _ in entries:
            offs.append(len(heap))
            heap += _pad8(nm + b"\0")
        if len(heap) < 24:                       # room for one free block, as libhdf5 lays out an empty heap
            heap += b"\0" * (24 - len(heap))
        data_addr = self._put(node, bytes(heap))
        heap_addr = self._put(node, b"HEAP" + struct.pack("<B3xQQQ", 0, len(heap), 1, data_addr))   # free list head 1 = none
        # symbol nodes of up to 2*LEAF_K entries
        level = []                               # (address, last key = heap offset of the largest name below)
        per = 2 * LEAF_K
        if not entries:                          # empty group: a B-tree node without children
            size = 24 + 2 * INTERNAL_K * 8 + (2 * INTERNAL_K + 1) * 8
            blob = b"TREE" + struct.pack("<BBHQQ", 0, 0, 0, UNDEF, UNDEF)
            return self._put(node, blob + b"\0" * (size - len(blob))), heap_addr
        for i in range(0, len(entries), per):
            part = entries[i:i + per]
            blob = b"SNOD" + struct.pack("<BxH", 1, len(part))
            for (nm, c), o in zip(part, offs[i:i + per]):
                if isinstance(c, Group):
                    blob += struct.pack("<QQII", o, c._addr, 1, 0) + struct.pack("<QQ", c._btree, c._heap)
                else:
                    blob += struct.pack("<QQII", o, c._addr, 0, 0) + b"\0" * 16
            blob += b"\0" * (8 + 40 * per - len(blob))
            level.append((self._put(node, blob), offs[i + len(part) - 1] if part else 0))
        depth = 0
        per = 2 * INTERNAL_K
        while True:
            nxt = []
            addrs = [self._alloc(24 + per * 8 + (per + 1) * 8) for _ in range(0, len(level), per)]
            for j, i in enumerate(range(0, len(level), per)):
                part = level[i:i + per]
                left = addrs[j - 1] if j > 0 else UNDEF
                right = addrs[j + 1] if j + 1 < len(addrs) else UNDEF
                blob = b"TREE" + struct.pack("<BBHQQ", 0, depth, len(part), left, right)
                first_key = 0 if i == 0 else level[i - 1][1]
                blob += struct.pack("<Q", first_key)
                for a, k in part:
                    blob += struct.pack("<QQ", a, k)
                blob += b"\0" * (24 + per * 8 + (per + 1) * 8 - len(blob))
                self._pwrite(addrs[j], blob)
                node._new_extent.append((addrs[j], len(blob)))
                nxt.append((addrs[j], part[-1][1]))
            if len(nxt) == 1:
                return nxt[0][0], heap_addr
            level, depth = nxt, depth + 1

    def _dataset_messages(self, d):
        msgs = []
        msgs.append((0x01, _space_encode(d.shape, d.maxshape)))
        msgs.append((0x03, _dt_encode(d._dt)))
        if d._dt == _VLEN_STR and d._strings is not None:                       # write the strings + the element table
            keep = d._new_extent                                                 # raw data: written once, not part of the
            d._new_extent = []                                                   # header's (rewritable) extent
            refs = self._gcol(d, [s.encode("utf-8") for s in d._strings]) if d._strings else []
            table = b"".join(struct.pack("<IQI", len(s.encode("utf-8")), a, i) for s, (a, i) in zip(d._strings, refs))
            addr = self._put(d, table) if table else UNDEF
            d._layout = ("contiguous", addr, len(table))
            d._strings = None
            d._new_extent = keep
        if d.chunks is not None:
            msgs.append((0x05, struct.pack("<BBBBI", 2, 3, 2, 1, 0)))           # fill value v2: incremental alloc, write if set, default fill
            if d._filters:
                fb = struct.pack("<BB6x", 1, len(d._filters))
                for fid, cd in d._filters:
                    fname = _pad8(b"deflate\0") if fid == 1 else (_pad8(b"lzf\0") if fid == 32000 else b"")
                    fb += struct.pack("<HHHH", fid, len(fname), 1, len(cd)) + fname + b"".join(struct.pack("<I", c) for c in cd)
                    if len(cd) % 2:
                        fb += b"\0" * 4
                msgs.append((0x0B, fb))
            bt = self._write_chunk_btree(d)
            d._layout = ("chunked", bt)
            dims = tuple(d.chunks) + (d._elem_size(),)
            msgs.append((0x08, struct.pack("<BBB", 3, 2, len(dims)) + struct.pack("<Q", bt) + b"".join(struct.pack("<I", x) for x in dims)))
        else:
            msgs.append((0x05, struct.pack("<BBBBI", 2, 2, 0, 1, 0)))           # late allocation, default fill
            kind = d._layout[0]
            if kind == "compact":
                raw = d._layout[1]
                msgs.append((0x08, struct.pack("<BBH", 3, 0, len(raw)) + raw))
            else:
                msgs.append((0x08, struct.pack("<BBQQ", 3, 1, d._layout[1], d.size * d._elem_size())))
        return msgs

    def _write_chunk_btree(self, d):
        idx = d._chunk_index()
        rank = len(d.shape)
        if not idx:
            return UNDEF
        ksz = 8 + 8 * (rank + 1)
        per = 2 * CHUNK_K
        node_size = 24 + per * 8 + (per + 1) * ksz

        def key(csize, mask, off):
            return struct.pack("<II", csize, mask) + b"".join(struct.pack("<Q", o) for o in off) + struct.pack("<Q", 0)

        items = sorted(idx.items())
        last_off = tuple(o + c for o, c in zip(items[-1][0], d.chunks))          # one-past key of the right-most node
        level = [(off, key(nb, mask, off), addr) for off, (addr, nb, mask) in items]
        depth = 0
        while True:
            nxt = []
            addrs = [self._alloc(node_size) for _ in range(0, len(level), per)]
            for j, i in enumerate(range(0, len(level), per)):
                part = level[i:i + per]
                left = addrs[j - 1] if j > 0 else UNDEF
                right = addrs[j + 1] if j + 1 < len(addrs) else UNDEF
                blob = b"TREE" + struct.pack("<BBHQQ", 1, depth, len(part), left, right)
                for _off, k, a in part:
                    blob += k + struct.pack("<Q", a)
                end = level[i + per][1] if i + per < len(level) else key(0, 0, last_off)
                blob += end
                blob += b"\0" * (node_size - len(blob))
                self._pwrite(addrs[j], blob)
                d._new_extent.append((addrs[j], node_size))
                nxt.append((part[0][0], part[0][1], addrs[j]))
            if len(nxt) == 1:
                return nxt[0][2]
            level, depth = nxt, depth + 1


def is_hdf5(path) -> bool:
    try:
        with open(path, "rb") as f:
            return f.read(8) == SIG
    except OSError:
        return False
