"""ctypes binding of the stock libhdf5 C library (1.10.x, the library h5py wraps) — TEST INFRASTRUCTURE ONLY.

Used by tests/test_h5lite.py and tests/make_h5_golden.py as the independent checker of vimo_clip_amd/h5lite.py: files
written by h5lite must read back through libhdf5, and files written through libhdf5 with the calls h5py makes for
extract_embeddings.py:50-119 (default "earliest" file format, chunked + deflate, int64 / variable-length UTF-8 string
attributes, ``video_ids``) must read back through h5lite.  The product never imports this module.
"""
from __future__ import annotations

import ctypes as C
import glob
import os

import numpy as np

hid_t, herr_t, hsize_t = C.c_int64, C.c_int, C.c_uint64
H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC = 0, 1, 2
H5S_UNLIMITED = 0xFFFFFFFFFFFFFFFF
H5T_VARIABLE = C.c_size_t(-1).value


def _find():
    cands = [os.environ.get("VMC_LIBHDF5", "")] + sorted(glob.glob("/opt/conda/lib/libhdf5.so*")) + \
        sorted(glob.glob("/usr/lib/x86_64-linux-gnu/libhdf5*.so*"))
    for c in cands:
        if c and os.path.exists(c):
            try:
                return C.CDLL(c)
            except OSError:
                continue
    return None


class H5Ref:
    def __init__(self):
        lib = _find()
        if lib is None:
            raise ImportError("libhdf5 not found")
        self.lib = lib
        sig = {
            "H5open": (herr_t, []), "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]), "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]),
            "H5Fclose": (herr_t, [hid_t]), "H5Fflush": (herr_t, [hid_t, C.c_int]),
            "H5Gcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]), "H5Gclose": (herr_t, [hid_t]),
            "H5Oopen": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Oclose": (herr_t, [hid_t]),
            "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]), "H5Screate": (hid_t, [C.c_int]),
            "H5Sclose": (herr_t, [hid_t]), "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
            "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
            "H5Sselect_hyperslab": (herr_t, [hid_t, C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t), C.POINTER(hsize_t), C.POINTER(hsize_t)]),
            "H5Pcreate": (hid_t, [hid_t]), "H5Pclose": (herr_t, [hid_t]), "H5Pset_chunk": (herr_t, [hid_t, C.c_int, C.POINTER(hsize_t)]),
            "H5Pset_deflate": (herr_t, [hid_t, C.c_uint]), "H5Pset_shuffle": (herr_t, [hid_t]),
            "H5Pset_libver_bounds": (herr_t, [hid_t, C.c_int, C.c_int]),
            "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]), "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
            "H5Dclose": (herr_t, [hid_t]), "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]), "H5Dget_space": (hid_t, [hid_t]),
            "H5Dget_type": (hid_t, [hid_t]), "H5Dset_extent": (herr_t, [hid_t, C.POINTER(hsize_t)]),
            "H5Dvlen_reclaim": (herr_t, [hid_t, hid_t, hid_t, C.c_void_p]),
            "H5Acreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]), "H5Aopen": (hid_t, [hid_t, C.c_char_p, hid_t]),
            "H5Aclose": (herr_t, [hid_t]), "H5Awrite": (herr_t, [hid_t, hid_t, C.c_void_p]), "H5Aread": (herr_t, [hid_t, hid_t, C.c_void_p]),
            "H5Aget_type": (hid_t, [hid_t]), "H5Aexists": (C.c_int, [hid_t, C.c_char_p]),
            "H5Tcopy": (hid_t, [hid_t]), "H5Tclose": (herr_t, [hid_t]), "H5Tset_size": (herr_t, [hid_t, C.c_size_t]),
            "H5Tset_cset": (herr_t, [hid_t, C.c_int]), "H5Tget_size": (C.c_size_t, [hid_t]), "H5Tget_class": (C.c_int, [hid_t]),
            "H5Tis_variable_str": (C.c_int, [hid_t]), "H5Tenum_create": (hid_t, [hid_t]),
            "H5Tenum_insert": (herr_t, [hid_t, C.c_char_p, C.c_void_p]), "H5Lexists": (C.c_int, [hid_t, C.c_char_p, hid_t]),
            "H5Eset_auto2": (herr_t, [hid_t, C.c_void_p, C.c_void_p]),
        }
        for n, (res, args) in sig.items():
            f = getattr(lib, n)
            f.restype, f.argtypes = res, args
        lib.H5open()
        lib.H5Eset_auto2(0, None, None)       # no stderr spam on expected failures
        g = lambda n: hid_t.in_dll(lib, n).value
        self.T = {np.dtype("f4"): g("H5T_NATIVE_FLOAT_g"), np.dtype("f8"): g("H5T_NATIVE_DOUBLE_g"), np.dtype("i8"): g("H5T_NATIVE_INT64_g"),
                  np.dtype("i4"): g("H5T_NATIVE_INT32_g"), np.dtype("u1"): g("H5T_NATIVE_UINT8_g"), np.dtype("i1"): g("H5T_NATIVE_INT8_g")}
        self.C_S1 = g("H5T_C_S1_g")
        self.P_DCPL, self.P_FAPL = g("H5P_CLS_DATASET_CREATE_ID_g"), g("H5P_CLS_FILE_ACCESS_ID_g")

    # ------------------------------------------------------------------------------------------ helpers
    def _ok(self, v, what):
        if v < 0:
            raise OSError(f"libhdf5: {what} failed")
        return v

    def _dims(self, t):
        return (hsize_t * len(t))(*[H5S_UNLIMITED if x is None else x for x in t])

    def vstr(self):
        t = self._ok(self.lib.H5Tcopy(self.C_S1), "H5Tcopy")
        self.lib.H5Tset_size(t, H5T_VARIABLE)
        self.lib.H5Tset_cset(t, 1)            # UTF-8, as h5py.string_dtype()
        return t

    def create(self, path, latest=False):
        fapl = 0
        if latest:
            fapl = self.lib.H5Pcreate(self.P_FAPL)
            self.lib.H5Pset_libver_bounds(fapl, 2, 2)      # H5F_LIBVER_V110 / LATEST in 1.10
        f = self._ok(self.lib.H5Fcreate(path.encode(), H5F_ACC_TRUNC, 0, fapl), "H5Fcreate")
        if fapl:
            self.lib.H5Pclose(fapl)
        return f

    def open(self, path, rw=False):
        return self._ok(self.lib.H5Fopen(path.encode(), H5F_ACC_RDWR if rw else H5F_ACC_RDONLY, 0), f"H5Fopen({path})")

    def close(self, f):
        self._ok(self.lib.H5Fclose(f), "H5Fclose")

    def group(self, loc, name):
        g = self._ok(self.lib.H5Gcreate2(loc, name.encode(), 0, 0, 0), "H5Gcreate2")
        return g

    def set_attr(self, loc, name, v):
        L = self.lib
        sp = L.H5Screate(0)
        if isinstance(v, str):
            t = self.vstr()
            a = self._ok(L.H5Acreate2(loc, name.encode(), t, sp, 0, 0), "H5Acreate2")
            buf = (C.c_char_p * 1)(v.encode("utf-8"))
            self._ok(L.H5Awrite(a, t, buf), "H5Awrite")
            L.H5Tclose(t)
        elif isinstance(v, (bool, np.bool_)):                # h5py: enum {FALSE=0, TRUE=1} over int8
            t = L.H5Tenum_create(self.T[np.dtype("i1")])
            for nm, val in (("FALSE", 0), ("TRUE", 1)):
                L.H5Tenum_insert(t, nm.encode(), C.byref(C.c_int8(val)))
            a = self._ok(L.H5Acreate2(loc, name.encode(), t, sp, 0, 0), "H5Acreate2")
            self._ok(L.H5Awrite(a, t, C.byref(C.c_int8(int(v)))), "H5Awrite")
            L.H5Tclose(t)
        else:
            arr = np.asarray(v, dtype=np.int64 if isinstance(v, (int, np.integer)) else np.float64)
            a = self._ok(L.H5Acreate2(loc, name.encode(), self.T[arr.dtype], sp, 0, 0), "H5Acreate2")
            self._ok(L.H5Awrite(a, self.T[arr.dtype], arr.ctypes.data), "H5Awrite")
        L.H5Aclose(a)
        L.H5Sclose(sp)

    def dataset(self, loc, name, data, chunks=None, gzip=None, shuffle=False, maxshape=None):
        L = self.lib
        data = np.ascontiguousarray(data)
        sp = L.H5Screate_simple(data.ndim, self._dims(data.shape), self._dims(maxshape) if maxshape else None)
        dcpl = 0
        if chunks:
            dcpl = L.H5Pcreate(self.P_DCPL)
            L.H5Pset_chunk(dcpl, len(chunks), self._dims(chunks))
            if shuffle:
                L.H5Pset_shuffle(dcpl)
            if gzip is not None:
                L.H5Pset_deflate(dcpl, gzip)
        d = self._ok(L.H5Dcreate2(loc, name.encode(), self.T[data.dtype], sp, 0, dcpl, 0), "H5Dcreate2")
        if data.size:
            self._ok(L.H5Dwrite(d, self.T[data.dtype], 0, 0, 0, data.ctypes.data), "H5Dwrite")
        if dcpl:
            L.H5Pclose(dcpl)
        L.H5Sclose(sp)
        return d

    def append_rows(self, d, block):
        """h5py's ``dset.resize(new_n, axis=0); dset[old:new] = block``."""
        L = self.lib
        block = np.ascontiguousarray(block)
        sp = L.H5Dget_space(d)
        nd = L.H5Sget_simple_extent_ndims(sp)
        dims = (hsize_t * nd)()
        L.H5Sget_simple_extent_dims(sp, dims, None)
        L.H5Sclose(sp)
        old = dims[0]
        dims[0] = old + block.shape[0]
        self._ok(L.H5Dset_extent(d, dims), "H5Dset_extent")
        fs = L.H5Dget_space(d)
        start = (hsize_t * nd)(old, *([0] * (nd - 1)))
        count = self._dims(block.shape)
        self._ok(L.H5Sselect_hyperslab(fs, 0, start, None, count, None), "H5Sselect_hyperslab")
        ms = L.H5Screate_simple(nd, count, None)
        self._ok(L.H5Dwrite(d, self.T[block.dtype], ms, fs, 0, block.ctypes.data), "H5Dwrite")
        L.H5Sclose(ms)
        L.H5Sclose(fs)

    def string_dataset(self, loc, name, strings):
        L = self.lib
        t = self.vstr()
        sp = L.H5Screate_simple(1, self._dims((len(strings),)), None)
        d = self._ok(L.H5Dcreate2(loc, name.encode(), t, sp, 0, 0, 0), "H5Dcreate2")
        buf = (C.c_char_p * len(strings))(*[s.encode("utf-8") for s in strings])
        self._ok(L.H5Dwrite(d, t, 0, 0, 0, buf), "H5Dwrite")
        L.H5Dclose(d)
        L.H5Sclose(sp)
        L.H5Tclose(t)

    # ------------------------------------------------------------------------------------------ reading
    def exists(self, f, path):
        cur = ""
        for part in [p for p in path.split("/") if p]:
            cur += "/" + part
            if self.lib.H5Lexists(f, cur.encode(), 0) <= 0:
                return False
        return True

    def read(self, f, path):
        L = self.lib
        d = self._ok(L.H5Dopen2(f, path.encode(), 0), f"H5Dopen2({path})")
        sp = L.H5Dget_space(d)
        nd = L.H5Sget_simple_extent_ndims(sp)
        dims, mx = (hsize_t * max(nd, 1))(), (hsize_t * max(nd, 1))()
        L.H5Sget_simple_extent_dims(sp, dims, mx)
        shape = tuple(dims[i] for i in range(nd))
        t = L.H5Dget_type(d)
        if L.H5Tis_variable_str(t) > 0:
            n = int(np.prod(shape))
            buf = (C.c_char_p * n)()
            self._ok(L.H5Dread(d, t, 0, 0, 0, buf), "H5Dread")
            out = [buf[i].decode("utf-8") if buf[i] is not None else "" for i in range(n)]
            L.H5Dvlen_reclaim(t, sp, 0, buf)
        else:
            cls, size = L.H5Tget_class(t), L.H5Tget_size(t)
            dt = np.dtype({(1, 4): "f4", (1, 8): "f8", (0, 8): "i8", (0, 4): "i4", (0, 1): "u1"}[(cls, size)])
            out = np.empty(shape, dtype=dt)
            if out.size:
                self._ok(L.H5Dread(d, self.T[dt], 0, 0, 0, out.ctypes.data), "H5Dread")
        L.H5Tclose(t)
        L.H5Sclose(sp)
        L.H5Dclose(d)
        return out

    def maxshape(self, f, path):
        L = self.lib
        d = self._ok(L.H5Dopen2(f, path.encode(), 0), "H5Dopen2")
        sp = L.H5Dget_space(d)
        nd = L.H5Sget_simple_extent_ndims(sp)
        dims, mx = (hsize_t * nd)(), (hsize_t * nd)()
        L.H5Sget_simple_extent_dims(sp, dims, mx)
        L.H5Sclose(sp)
        L.H5Dclose(d)
        return tuple(None if mx[i] == H5S_UNLIMITED else mx[i] for i in range(nd))

    def attr(self, f, obj, name):
        L = self.lib
        o = self._ok(L.H5Oopen(f, obj.encode(), 0), f"H5Oopen({obj})")
        a = self._ok(L.H5Aopen(o, name.encode(), 0), f"H5Aopen({name})")
        t = L.H5Aget_type(a)
        if L.H5Tis_variable_str(t) > 0:
            buf = (C.c_char_p * 1)()
            self._ok(L.H5Aread(a, t, buf), "H5Aread")
            out = buf[0].decode("utf-8")
        else:
            cls, size = L.H5Tget_class(t), L.H5Tget_size(t)
            if cls == 8:                                        # enum (bool)
                v = C.c_int8()
                self._ok(L.H5Aread(a, t, C.byref(v)), "H5Aread")
                out = bool(v.value)
            else:
                dt = np.dtype({(1, 4): "f4", (1, 8): "f8", (0, 8): "i8", (0, 4): "i4"}[(cls, size)])
                v = np.empty((), dtype=dt)
                self._ok(L.H5Aread(a, self.T[dt], v.ctypes.data), "H5Aread")
                out = v[()]
        L.H5Tclose(t)
        L.H5Aclose(a)
        L.H5Oclose(o)
        return out


def load():
    try:
        return H5Ref()
    except (ImportError, OSError, AttributeError, ValueError):
        return None
