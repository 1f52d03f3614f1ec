"""Minimal HDF5 access through ctypes + the libhdf5 C library (h5py is not available).

Used for the *small* parts of a Kover dataset: the header Kover writes before it calls the
k-mer tools (attrs, phenotype, genome_identifiers, phenotype_tags --
bin/kover/core/kover/dataset/create.py:311-354) and for reading files back
(the subset of bin/kover/core/kover/dataset/ds.py:26-148 our tests and tools need).
The big datasets (kmer_sequences, kmer_matrix, kmer_by_matrix_column) are written by
libgrmkmer.so's grm_write_kover_h5.
"""
import ctypes as C
import os

import numpy as np

hid_t = C.c_int64
hsize_t = C.c_ulonglong
H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC = 0, 1, 2
H5S_SCALAR = 0
H5T_INTEGER, H5T_FLOAT, H5T_STRING = 0, 1, 3
H5T_VARIABLE = C.c_size_t(-1).value
_lib = None


class H5Error(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is not None:
        return _lib
    cands = [os.environ.get("GRM_HDF5_LIB"), "libhdf5.so", "libhdf5_serial.so", "libhdf5.so.103", "libhdf5.so.200",
             "/opt/conda/lib/libhdf5.so", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so"]
    L = None
    for c in cands:
        if not c:
            continue
        try:
            L = C.CDLL(c)
            break
        except OSError:
            continue
    if L is None:
        raise H5Error("libhdf5 not found (set GRM_HDF5_LIB=/path/to/libhdf5.so)")
    sig = {
        "H5open": (C.c_int, []),
        "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]),
        "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]),
        "H5Fclose": (C.c_int, [hid_t]),
        "H5Screate": (hid_t, [C.c_int]),
        "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Sclose": (C.c_int, [hid_t]),
        "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
        "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
        "H5Tcopy": (hid_t, [hid_t]),
        "H5Tset_size": (C.c_int, [hid_t, C.c_size_t]),
        "H5Tget_size": (C.c_size_t, [hid_t]),
        "H5Tset_strpad": (C.c_int, [hid_t, C.c_int]),
        "H5Tget_class": (C.c_int, [hid_t]),
        "H5Tget_sign": (C.c_int, [hid_t]),
        "H5Tis_variable_str": (C.c_int, [hid_t]),
        "H5Tclose": (C.c_int, [hid_t]),
        "H5Pcreate": (hid_t, [hid_t]),
        "H5Pset_chunk": (C.c_int, [hid_t, C.c_int, C.POINTER(hsize_t)]),
        "H5Pset_deflate": (C.c_int, [hid_t, C.c_uint]),
        "H5Pget_chunk": (C.c_int, [hid_t, C.c_int, C.POINTER(hsize_t)]),
        "H5Pget_nfilters": (C.c_int, [hid_t]),
        "H5Pget_layout": (C.c_int, [hid_t]),
        "H5Pclose": (C.c_int, [hid_t]),
        "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
        "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
        "H5Dwrite": (C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Dread": (C.c_int, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
        "H5Dget_space": (hid_t, [hid_t]),
        "H5Dget_type": (hid_t, [hid_t]),
        "H5Dget_create_plist": (hid_t, [hid_t]),
        "H5Dclose": (C.c_int, [hid_t]),
        "H5Acreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]),
        "H5Aopen": (hid_t, [hid_t, C.c_char_p, hid_t]),
        "H5Aexists": (C.c_int, [hid_t, C.c_char_p]),
        "H5Awrite": (C.c_int, [hid_t, hid_t, C.c_void_p]),
        "H5Aread": (C.c_int, [hid_t, hid_t, C.c_void_p]),
        "H5Aget_type": (hid_t, [hid_t]),
        "H5Aclose": (C.c_int, [hid_t]),
        "H5Lexists": (C.c_int, [hid_t, C.c_char_p, hid_t]),
        "H5Gcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]),
        "H5Gopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
        "H5Gclose": (C.c_int, [hid_t]),
        "H5Gget_num_objs": (C.c_int, [hid_t, C.POINTER(hsize_t)]),
        "H5Gget_objname_by_idx": (C.c_ssize_t, [hid_t, hsize_t, C.c_char_p, C.c_size_t]),
        "H5Eset_auto2": (C.c_int, [hid_t, C.c_void_p, C.c_void_p]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype, fn.argtypes = res, args
    if L.H5open() < 0:
        raise H5Error("H5open failed")
    L.H5Eset_auto2(0, None, None)
    for g in ("H5T_C_S1_g", "H5T_NATIVE_UINT8_g", "H5T_NATIVE_UINT16_g", "H5T_NATIVE_UINT32_g", "H5T_NATIVE_UINT64_g",
              "H5T_NATIVE_DOUBLE_g", "H5T_NATIVE_INT64_g", "H5P_CLS_DATASET_CREATE_ID_g"):
        setattr(L, g[:-2], hid_t.in_dll(L, g).value)
    _lib = L
    return L


def _chk(v, what):
    if v < 0:
        raise H5Error("HDF5 call failed: " + what)
    return v


class File:
    def __init__(self, path, mode="r"):
        L = lib()
        self.L = L
        p = path.encode()
        if mode == "w":
            self.id = _chk(L.H5Fcreate(p, H5F_ACC_TRUNC, 0, 0), "H5Fcreate " + path)
        elif mode == "r+":
            self.id = _chk(L.H5Fopen(p, H5F_ACC_RDWR, 0), "H5Fopen " + path)
        else:
            self.id = _chk(L.H5Fopen(p, H5F_ACC_RDONLY, 0), "H5Fopen " + path)

    def close(self):
        if self.id is not None:
            _chk(self.L.H5Fclose(self.id), "H5Fclose")
            self.id = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- attributes (variable-length strings and float64 scalars, as h5py writes them) ----
    def _set_attr_on(self, loc, name, value):
        L = self.L
        space = _chk(L.H5Screate(H5S_SCALAR), "H5Screate")
        if isinstance(value, (int, np.integer)) and not isinstance(value, bool):
            a = _chk(L.H5Acreate2(loc, name.encode(), L.H5T_NATIVE_INT64, space, 0, 0), "H5Acreate2 " + name)
            v = C.c_int64(int(value))
            _chk(L.H5Awrite(a, L.H5T_NATIVE_INT64, C.byref(v)), "H5Awrite " + name)
        elif isinstance(value, float):
            a = _chk(L.H5Acreate2(loc, name.encode(), L.H5T_NATIVE_DOUBLE, space, 0, 0), "H5Acreate2 " + name)
            v = C.c_double(value)
            _chk(L.H5Awrite(a, L.H5T_NATIVE_DOUBLE, C.byref(v)), "H5Awrite " + name)
        else:
            t = _chk(L.H5Tcopy(L.H5T_C_S1), "H5Tcopy")
            _chk(L.H5Tset_size(t, H5T_VARIABLE), "H5Tset_size")
            a = _chk(L.H5Acreate2(loc, name.encode(), t, space, 0, 0), "H5Acreate2 " + name)
            s = C.c_char_p(str(value).encode())
            _chk(L.H5Awrite(a, t, C.byref(s)), "H5Awrite " + name)
            L.H5Tclose(t)
        L.H5Aclose(a)
        L.H5Sclose(space)

    def set_attr(self, name, value):
        self._set_attr_on(self.id, name, value)

    def _get_attr_on(self, loc, name):
        L = self.L
        if L.H5Aexists(loc, name.encode()) <= 0:
            raise KeyError(name)
        a = _chk(L.H5Aopen(loc, name.encode(), 0), "H5Aopen " + name)
        t = L.H5Aget_type(a)
        try:
            cls = L.H5Tget_class(t)
            if cls == H5T_FLOAT:
                v = C.c_double()
                _chk(L.H5Aread(a, L.H5T_NATIVE_DOUBLE, C.byref(v)), "H5Aread")
                return v.value
            if cls == H5T_INTEGER:
                v = C.c_int64()
                _chk(L.H5Aread(a, L.H5T_NATIVE_INT64, C.byref(v)), "H5Aread")
                return int(v.value)
            if cls == H5T_STRING:
                if L.H5Tis_variable_str(t) > 0:
                    p = C.c_char_p()
                    _chk(L.H5Aread(a, t, C.byref(p)), "H5Aread")
                    return (p.value or b"").decode()
                n = L.H5Tget_size(t)
                buf = C.create_string_buffer(n + 1)
                _chk(L.H5Aread(a, t, buf), "H5Aread")
                return buf.raw[:n].rstrip(b"\0").decode()
            raise H5Error("unsupported attribute class %d" % cls)
        finally:
            L.H5Tclose(t)
            L.H5Aclose(a)

    def get_attr(self, name):
        return self._get_attr_on(self.id, name)

    def has_attr(self, name):
        return self.L.H5Aexists(self.id, name.encode()) > 0

    def dataset_attr(self, dataset, name):
        d = _chk(self.L.H5Dopen2(self.id, dataset.encode(), 0), "H5Dopen2 " + dataset)
        try:
            return self._get_attr_on(d, name)
        finally:
            self.L.H5Dclose(d)

    # ---- datasets ----
    def _h5type(self, arr):
        L = self.L
        if arr.dtype.kind == "S":
            t = _chk(L.H5Tcopy(L.H5T_C_S1), "H5Tcopy")
            _chk(L.H5Tset_size(t, max(1, arr.dtype.itemsize)), "H5Tset_size")
            _chk(L.H5Tset_strpad(t, 1), "H5Tset_strpad")      # H5T_STR_NULLPAD, as h5py maps numpy 'S'
            return t, True
        m = {np.dtype(np.uint8): L.H5T_NATIVE_UINT8, np.dtype(np.uint16): L.H5T_NATIVE_UINT16,
             np.dtype(np.uint32): L.H5T_NATIVE_UINT32, np.dtype(np.uint64): L.H5T_NATIVE_UINT64,
             np.dtype(np.float64): L.H5T_NATIVE_DOUBLE, np.dtype(np.int64): L.H5T_NATIVE_INT64}
        return m[arr.dtype], False

    def create_dataset(self, name, arr, gzip=0, chunks=None, attrs=None):
        """arr: numpy array of dtype uint8/16/32/64 or 'S<n>' (fixed strings)"""
        L = self.L
        arr = np.ascontiguousarray(arr)
        t, own = self._h5type(arr)
        dims = (hsize_t * max(1, arr.ndim))(*arr.shape)
        space = _chk(L.H5Screate_simple(arr.ndim, dims, None), "H5Screate_simple")
        dcpl = 0
        if arr.size and (gzip or chunks):
            dcpl = _chk(L.H5Pcreate(L.H5P_CLS_DATASET_CREATE_ID), "H5Pcreate")
            ch = chunks or tuple(min(s, 1 << 20) for s in arr.shape)
            _chk(L.H5Pset_chunk(dcpl, arr.ndim, (hsize_t * arr.ndim)(*ch)), "H5Pset_chunk")
            if gzip:
                _chk(L.H5Pset_deflate(dcpl, gzip), "H5Pset_deflate")
        d = _chk(L.H5Dcreate2(self.id, name.encode(), t, space, 0, dcpl, 0), "H5Dcreate2 " + name)
        if arr.size:
            _chk(L.H5Dwrite(d, t, 0, 0, 0, arr.ctypes.data), "H5Dwrite " + name)
        for k, v in (attrs or {}).items():
            self._set_attr_on(d, k, v)
        L.H5Dclose(d)
        if dcpl:
            L.H5Pclose(dcpl)
        L.H5Sclose(space)
        if own:
            L.H5Tclose(t)

    def exists(self, name):
        """link exists; every intermediate group of a path is checked first (H5Lexists requires it)"""
        parts = [p for p in name.split("/") if p]
        for i in range(1, len(parts) + 1):
            if self.L.H5Lexists(self.id, "/".join(parts[:i]).encode(), 0) <= 0:
                return False
        return True

    def create_group(self, name):
        g = _chk(self.L.H5Gcreate2(self.id, name.encode(), 0, 0, 0), "H5Gcreate2 " + name)
        self.L.H5Gclose(g)

    def set_group_attr(self, group, name, value):
        g = _chk(self.L.H5Gopen2(self.id, group.encode(), 0), "H5Gopen2 " + group)
        try:
            self._set_attr_on(g, name, value)
        finally:
            self.L.H5Gclose(g)

    def get_group_attr(self, group, name):
        g = _chk(self.L.H5Gopen2(self.id, group.encode(), 0), "H5Gopen2 " + group)
        try:
            return self._get_attr_on(g, name)
        finally:
            self.L.H5Gclose(g)

    def list_group(self, group):
        g = _chk(self.L.H5Gopen2(self.id, group.encode(), 0), "H5Gopen2 " + group)
        try:
            n = hsize_t()
            _chk(self.L.H5Gget_num_objs(g, C.byref(n)), "H5Gget_num_objs")
            out = []
            for i in range(n.value):
                buf = C.create_string_buffer(512)
                self.L.H5Gget_objname_by_idx(g, i, buf, 512)
                out.append(buf.value.decode())
            return out
        finally:
            self.L.H5Gclose(g)

    def read(self, name):
        L = self.L
        d = _chk(L.H5Dopen2(self.id, name.encode(), 0), "H5Dopen2 " + name)
        space = L.H5Dget_space(d)
        t = L.H5Dget_type(d)
        try:
            nd = L.H5Sget_simple_extent_ndims(space)
            dims = (hsize_t * max(1, nd))()
            L.H5Sget_simple_extent_dims(space, dims, None)
            shape = tuple(int(x) for x in dims[:nd])
            cls, size = L.H5Tget_class(t), L.H5Tget_size(t)
            if cls == H5T_STRING:
                if L.H5Tis_variable_str(t) > 0:
                    raise H5Error("variable-length string datasets are not supported")
                dt = np.dtype("S%d" % size)
            elif cls == H5T_INTEGER:
                dt = np.dtype("%s%d" % ("u" if L.H5Tget_sign(t) == 0 else "i", size))
            elif cls == H5T_FLOAT:
                dt = np.dtype("f%d" % size)
            else:
                raise H5Error("unsupported dataset class %d" % cls)
            out = np.zeros(shape, dtype=dt)
            if out.size:
                _chk(L.H5Dread(d, t, 0, 0, 0, out.ctypes.data), "H5Dread " + name)
            return out
        finally:
            L.H5Tclose(t)
            L.H5Sclose(space)
            L.H5Dclose(d)

    def layout(self, name):
        """-> dict(chunks=tuple|None, n_filters=int)  (rules.py:104-131 relies on .chunks)"""
        L = self.L
        d = _chk(L.H5Dopen2(self.id, name.encode(), 0), "H5Dopen2 " + name)
        space = L.H5Dget_space(d)
        p = L.H5Dget_create_plist(d)
        try:
            nd = L.H5Sget_simple_extent_ndims(space)
            chunks = None
            if L.H5Pget_layout(p) == 2:      # H5D_CHUNKED
                ch = (hsize_t * max(1, nd))()
                L.H5Pget_chunk(p, nd, ch)
                chunks = tuple(int(x) for x in ch[:nd])
            return {"chunks": chunks, "n_filters": L.H5Pget_nfilters(p)}
        finally:
            L.H5Pclose(p)
            L.H5Sclose(space)
            L.H5Dclose(d)
