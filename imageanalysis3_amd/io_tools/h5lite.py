"""Minimal HDF5 access for the FOV save files (``classes/batch_functions.py:305-493``, ``classes/field_of_view.py:1314-1398``).

The reference goes through h5py; this image has no h5py for the system interpreter but ships the HDF5 C library
(libhdf5 1.10), so this module binds the few calls the save-file schema needs with ctypes and exposes them under
h5py's names: ``File`` / ``Group`` / ``Dataset`` with ``require_group``, ``create_dataset(shape, dtype, data, maxshape,
chunks)``, ``ds[i]``, ``ds[i, :n, :]``, ``ds[...] = value``, ``ds.resize(size, axis)``, ``del grp[name]``, ``keys()``,
``in`` and scalar / string ``attrs``.  Files written here are ordinary HDF5 files (h5py reads them and vice versa;
``tests/test_h5_cpu.py`` checks both directions with the h5py of /opt/conda when that interpreter exists).

Library lookup: ``$IA3_HDF5_LIB``, then the loader's search path, then the known install prefixes.  No fallback
container format: without the library every entry point raises ``OSError``.
"""
import ctypes as C
import ctypes.util
import glob
import os
import threading
import numpy as np

hid_t = C.c_int64
hsize_t = C.c_uint64
herr_t = C.c_int

_lock = threading.RLock()   # the packaged library is not built thread-safe
_h5 = None

H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC, H5F_ACC_EXCL = 0, 1, 2, 4
H5P_DEFAULT, H5S_ALL = 0, 0
H5S_SELECT_SET = 0
H5F_LIBVER_EARLIEST, H5F_LIBVER_LATEST = 0, 2          # H5F_LIBVER_V110 in 1.10.x
H5T_INTEGER, H5T_FLOAT, H5T_STRING = 0, 1, 3
H5T_SGN_NONE = 0
H5I_GROUP, H5I_DATASET = 2, 5
H5_INDEX_NAME, H5_ITER_INC = 0, 0
H5S_UNLIMITED = 0xFFFFFFFFFFFFFFFF
H5D_CHUNKED = 2


def _candidates():
    env = os.environ.get("IA3_HDF5_LIB")
    if env:
        yield env
    found = ctypes.util.find_library("hdf5")
    if found:
        yield found
    for pat in ("/opt/conda/lib/libhdf5.so*", "/usr/lib/x86_64-linux-gnu/libhdf5*.so*",
                "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so*", "/usr/local/lib/libhdf5.so*"):
        for p in sorted(glob.glob(pat)):
            yield p


def available():
    try:
        lib()
        return True
    except OSError:
        return False


def lib():
    """The loaded libhdf5 (cached); raises OSError when there is none."""
    global _h5
    if _h5 is not None:
        return _h5
    with _lock:
        if _h5 is not None:
            return _h5
        err = None
        L = None
        for cand in _candidates():
            try:
                L = C.CDLL(cand)
                break
            except OSError as e:
                err = e
        if L is None:
            raise OSError("libhdf5 not found (set IA3_HDF5_LIB); HDF5 save files need the HDF5 C library: %s" % err)
        sig = {
            "H5open": (herr_t, []), "H5get_libversion": (herr_t, [C.POINTER(C.c_uint)] * 3),
            "H5Eset_auto2": (herr_t, [hid_t, C.c_void_p, C.c_void_p]),
            "H5Fcreate": (hid_t, [C.c_char_p, C.c_uint, hid_t, hid_t]), "H5Fopen": (hid_t, [C.c_char_p, C.c_uint, hid_t]),
            "H5Fclose": (herr_t, [hid_t]), "H5Fflush": (herr_t, [hid_t, C.c_int]),
            "H5Pcreate": (hid_t, [hid_t]), "H5Pclose": (herr_t, [hid_t]),
            "H5Pset_libver_bounds": (herr_t, [hid_t, C.c_int, C.c_int]),
            "H5Pset_chunk": (herr_t, [hid_t, C.c_int, C.POINTER(hsize_t)]),
            "H5Pget_chunk": (C.c_int, [hid_t, C.c_int, C.POINTER(hsize_t)]), "H5Pget_layout": (C.c_int, [hid_t]),
            "H5Pset_fclose_degree": (herr_t, [hid_t, C.c_int]),
            "H5Gcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t]), "H5Gopen2": (hid_t, [hid_t, C.c_char_p, hid_t]),
            "H5Gclose": (herr_t, [hid_t]),
            "H5Lexists": (C.c_int, [hid_t, C.c_char_p, hid_t]), "H5Ldelete": (herr_t, [hid_t, C.c_char_p, hid_t]),
            "H5Literate": (herr_t, [hid_t, C.c_int, C.c_int, C.POINTER(hsize_t), C.c_void_p, C.c_void_p]),
            "H5Oopen": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Oclose": (herr_t, [hid_t]), "H5Iget_type": (C.c_int, [hid_t]),
            "H5Dcreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t, hid_t]),
            "H5Dopen2": (hid_t, [hid_t, C.c_char_p, hid_t]), "H5Dclose": (herr_t, [hid_t]),
            "H5Dget_space": (hid_t, [hid_t]), "H5Dget_type": (hid_t, [hid_t]), "H5Dget_create_plist": (hid_t, [hid_t]),
            "H5Dset_extent": (herr_t, [hid_t, C.POINTER(hsize_t)]),
            "H5Dread": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            "H5Dwrite": (herr_t, [hid_t, hid_t, hid_t, hid_t, hid_t, C.c_void_p]),
            "H5Screate_simple": (hid_t, [C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t)]), "H5Screate": (hid_t, [C.c_int]),
            "H5Sclose": (herr_t, [hid_t]), "H5Sget_simple_extent_ndims": (C.c_int, [hid_t]),
            "H5Sget_simple_extent_dims": (C.c_int, [hid_t, C.POINTER(hsize_t), C.POINTER(hsize_t)]),
            "H5Sselect_hyperslab": (herr_t, [hid_t, C.c_int, C.POINTER(hsize_t), C.POINTER(hsize_t), C.POINTER(hsize_t),
                                             C.POINTER(hsize_t)]),
            "H5Tcopy": (hid_t, [hid_t]), "H5Tclose": (herr_t, [hid_t]), "H5Tset_size": (herr_t, [hid_t, C.c_size_t]),
            "H5Tget_class": (C.c_int, [hid_t]), "H5Tget_size": (C.c_size_t, [hid_t]), "H5Tget_sign": (C.c_int, [hid_t]),
            "H5Tis_variable_str": (C.c_int, [hid_t]), "H5Tset_strpad": (herr_t, [hid_t, C.c_int]),
            "H5Aexists": (C.c_int, [hid_t, C.c_char_p]), "H5Adelete": (herr_t, [hid_t, C.c_char_p]),
            "H5Acreate2": (hid_t, [hid_t, C.c_char_p, hid_t, hid_t, hid_t, hid_t]), "H5Aopen": (hid_t, [hid_t, C.c_char_p, hid_t]),
            "H5Awrite": (herr_t, [hid_t, hid_t, C.c_void_p]), "H5Aread": (herr_t, [hid_t, hid_t, C.c_void_p]),
            "H5Aget_type": (hid_t, [hid_t]), "H5Aget_space": (hid_t, [hid_t]), "H5Aclose": (herr_t, [hid_t]),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        if L.H5open() < 0:
            raise OSError("H5open failed")
        L.H5Eset_auto2(0, None, None)   # errors become Python exceptions, not stderr dumps
        _h5 = L
        return L


def _glob(name):
    return hid_t.in_dll(lib(), name).value


def _native(dt):
    """numpy dtype -> (hid_t of the matching native type, owned?)"""
    dt = np.dtype(dt)
    table = {"u1": "H5T_NATIVE_UINT8_g", "u2": "H5T_NATIVE_UINT16_g", "u4": "H5T_NATIVE_UINT32_g",
             "u8": "H5T_NATIVE_UINT64_g", "i1": "H5T_NATIVE_INT8_g", "i2": "H5T_NATIVE_INT16_g",
             "i4": "H5T_NATIVE_INT32_g", "i8": "H5T_NATIVE_INT64_g", "f4": "H5T_NATIVE_FLOAT_g",
             "f8": "H5T_NATIVE_DOUBLE_g"}
    key = dt.kind + str(dt.itemsize)
    if dt.kind == "b":
        key = "i1"
    if key in table:
        return _glob(table[key]), False
    if dt.kind == "S":
        t = lib().H5Tcopy(_glob("H5T_C_S1_g"))
        lib().H5Tset_size(t, max(dt.itemsize, 1))   # fixed length, null padded as h5py does ('S3')
        lib().H5Tset_strpad(t, 1)                   # H5T_STR_NULLPAD: all bytes are payload, as numpy's S dtype
        return t, True
    raise TypeError("h5lite: unsupported dtype %r" % (dt,))


def _numpy_dtype(tid):
    L = lib()
    cls, size = L.H5Tget_class(tid), L.H5Tget_size(tid)
    if cls == H5T_INTEGER:
        return np.dtype(("u" if L.H5Tget_sign(tid) == H5T_SGN_NONE else "i") + str(size))
    if cls == H5T_FLOAT:
        return np.dtype("f" + str(size))
    if cls == H5T_STRING:
        if L.H5Tis_variable_str(tid) > 0:
            raise TypeError("h5lite: variable-length strings are not supported")
        return np.dtype("S" + str(size))
    raise TypeError("h5lite: unsupported HDF5 type class %d" % cls)


def _dims(seq):
    return (hsize_t * len(seq))(*[int(v) for v in seq])


def _guess_chunk(shape, maxshape, itemsize):
    """Chunk shape for ``chunks=True`` (the role of h5py's guess_chunk: about 16 KiB .. 1 MiB, halving the axes in
    turn; unlimited axes start from 1024)."""
    chunk = [1024 if (m is None and s == 0) else max(int(s), 1) for s, m in zip(shape, maxshape)]
    target = min(max(16 * 1024, 24 * 1024 * 2 ** np.log10(max(np.prod(chunk) * itemsize, 1) / (1024. * 1024))), 1024 * 1024)
    i = 0
    while np.prod(chunk) * itemsize > target and np.prod(chunk) > 1:
        ax = i % len(chunk)
        chunk[ax] = int(np.ceil(chunk[ax] / 2.0))
        i += 1
    return tuple(int(c) for c in chunk)


class _Attrs(object):
    def __init__(self, owner):
        self._owner = owner      # keeps the object (and its id) alive while the proxy is in use
        self._oid = owner._id

    def __contains__(self, name):
        with _lock:
            return lib().H5Aexists(self._oid, name.encode()) > 0

    def __setitem__(self, name, value):
        L = lib()
        with _lock:
            if name in self:
                L.H5Adelete(self._oid, name.encode())
            if isinstance(value, str):
                value = np.bytes_(value.encode("utf8"))
            arr = np.array(value, order='C')   # 0-d stays 0-d (a scalar attribute)
            tid, own = _native(arr.dtype)
            sid = L.H5Screate_simple(arr.ndim, _dims(arr.shape), None) if arr.ndim else L.H5Screate(0)
            aid = L.H5Acreate2(self._oid, name.encode(), tid, sid, H5P_DEFAULT, H5P_DEFAULT)
            if aid < 0:
                raise OSError("h5lite: cannot create attribute %r" % name)
            rc = L.H5Awrite(aid, tid, arr.ctypes.data_as(C.c_void_p))
            L.H5Aclose(aid); L.H5Sclose(sid)
            if own:
                L.H5Tclose(tid)
            if rc < 0:
                raise OSError("h5lite: cannot write attribute %r" % name)

    def __getitem__(self, name):
        L = lib()
        with _lock:
            if name not in self:
                raise KeyError(name)
            aid = L.H5Aopen(self._oid, name.encode(), H5P_DEFAULT)
            tid, sid = L.H5Aget_type(aid), L.H5Aget_space(aid)
            try:
                dt = _numpy_dtype(tid)
                nd = L.H5Sget_simple_extent_ndims(sid)
                dims = (hsize_t * max(nd, 1))()
                if nd > 0:
                    L.H5Sget_simple_extent_dims(sid, dims, None)
                out = np.empty(tuple(dims[:nd]), dt)
                mt, own = _native(dt)
                rc = L.H5Aread(aid, mt, out.ctypes.data_as(C.c_void_p))
                if own:
                    L.H5Tclose(mt)
                if rc < 0:
                    raise OSError("h5lite: cannot read attribute %r" % name)
            finally:
                L.H5Tclose(tid); L.H5Sclose(sid); L.H5Aclose(aid)
        if out.ndim == 0:
            v = out[()]
            return v.decode("utf8") if isinstance(v, bytes) else v
        return out


class Dataset(object):
    def __init__(self, did, name, file=None):
        self._id, self.name, self._file = did, name, file

    # -- geometry ---------------------------------------------------------------------------
    def _extent(self):
        L = lib()
        with _lock:
            sid = L.H5Dget_space(self._id)
            nd = L.H5Sget_simple_extent_ndims(sid)
            cur, mx = (hsize_t * max(nd, 1))(), (hsize_t * max(nd, 1))()
            if nd > 0:
                L.H5Sget_simple_extent_dims(sid, cur, mx)
            L.H5Sclose(sid)
        return tuple(int(v) for v in cur[:nd]), tuple(None if v == H5S_UNLIMITED else int(v) for v in mx[:nd])

    @property
    def shape(self):
        return self._extent()[0]

    @property
    def maxshape(self):
        return self._extent()[1]

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def dtype(self):
        L = lib()
        with _lock:
            tid = L.H5Dget_type(self._id)
            try:
                return _numpy_dtype(tid)
            finally:
                L.H5Tclose(tid)

    @property
    def chunks(self):
        L = lib()
        with _lock:
            pl = L.H5Dget_create_plist(self._id)
            try:
                if L.H5Pget_layout(pl) != H5D_CHUNKED:
                    return None
                nd = len(self.shape)
                c = (hsize_t * nd)()
                L.H5Pget_chunk(pl, nd, c)
                return tuple(int(v) for v in c)
            finally:
                L.H5Pclose(pl)

    def __len__(self):
        shp = self.shape
        if not shp:
            raise TypeError("len() of a scalar dataset")
        return shp[0]

    @property
    def attrs(self):
        return _Attrs(self)

    def resize(self, size, axis=None):
        """h5py semantics: ``resize(new_shape)`` or ``resize(new_length, axis)``."""
        shp = list(self.shape)
        if axis is not None:
            shp[int(axis)] = int(size)
        else:
            shp = [int(s) for s in size]
        if self.chunks is None:
            raise TypeError("Only chunked datasets can be resized")
        with _lock:
            if lib().H5Dset_extent(self._id, _dims(shp)) < 0:
                raise ValueError("h5lite: unable to set extent %r (maxshape %r)" % (shp, self.maxshape))

    # -- selections -----------------------------------------------------------------------------
    def _select(self, key):
        """key -> (start, count, result shape) for ints / unit-step slices / Ellipsis."""
        shp = self.shape
        if not isinstance(key, tuple):
            key = (key,)
        if any(k is Ellipsis for k in key):
            i = [k is Ellipsis for k in key].index(True)
            fill = len(shp) - (len(key) - 1)
            key = key[:i] + (slice(None),) * fill + key[i + 1:]
        if len(key) > len(shp):
            raise IndexError("too many indices for a dataset of rank %d" % len(shp))
        key = key + (slice(None),) * (len(shp) - len(key))
        start, count, out = [], [], []
        for k, n in zip(key, shp):
            if isinstance(k, (int, np.integer)):
                k = int(k)
                if k < 0:
                    k += n
                if not 0 <= k < n:
                    raise IndexError("index %d out of range for an axis of length %d" % (k, n))
                start.append(k); count.append(1)
            elif isinstance(k, slice):
                a, b, st = k.indices(n)
                if st != 1:
                    raise NotImplementedError("h5lite: only unit-step slices")
                c = max(b - a, 0)
                start.append(a); count.append(c); out.append(c)
            else:
                raise TypeError("h5lite: unsupported index %r" % (k,))
        return start, count, tuple(out)

    def _spaces(self, start, count):
        L = lib()
        fs = L.H5Dget_space(self._id)
        if len(start):
            if L.H5Sselect_hyperslab(fs, H5S_SELECT_SET, _dims(start), None, _dims(count), None) < 0:
                L.H5Sclose(fs)
                raise OSError("h5lite: bad selection")
            ms = L.H5Screate_simple(len(count), _dims(count), None)
        else:
            ms = L.H5Screate(0)
        return fs, ms

    def __getitem__(self, key):
        if key == ():
            key = Ellipsis
        start, count, oshape = self._select(key)
        dt = self.dtype
        out = np.empty(count, dt)
        if out.size:
            L = lib()
            with _lock:
                fs, ms = self._spaces(start, count)
                mt, own = _native(dt)
                rc = L.H5Dread(self._id, mt, ms, fs, H5P_DEFAULT, out.ctypes.data_as(C.c_void_p))
                L.H5Sclose(fs); L.H5Sclose(ms)
                if own:
                    L.H5Tclose(mt)
            if rc < 0:
                raise OSError("h5lite: read of %s failed" % self.name)
        out = out.reshape(oshape)
        return out[()] if out.ndim == 0 else out

    def __setitem__(self, key, value):
        if key == ():
            key = Ellipsis
        start, count, oshape = self._select(key)
        dt = self.dtype
        val = np.asarray(value)
        if dt.kind == "S" and val.dtype.kind == "U":
            val = np.char.encode(val, "utf8")
        buf = np.ascontiguousarray(np.broadcast_to(val.astype(dt, copy=False), oshape)).reshape(count)
        if not buf.size:
            return
        L = lib()
        with _lock:
            fs, ms = self._spaces(start, count)
            mt, own = _native(dt)
            rc = L.H5Dwrite(self._id, mt, ms, fs, H5P_DEFAULT, buf.ctypes.data_as(C.c_void_p))
            L.H5Sclose(fs); L.H5Sclose(ms)
            if own:
                L.H5Tclose(mt)
        if rc < 0:
            raise OSError("h5lite: write to %s failed" % self.name)

    def __array__(self, dtype=None, copy=None):
        a = self[...]
        return a if dtype is None else a.astype(dtype)

    def _close(self):
        if self._id and self._file is not None and self._file._fid:   # closing the file already closed this id
            with _lock:
                lib().H5Dclose(self._id)
        self._id = 0

    def __del__(self):
        try:
            self._close()
        except Exception:
            pass


_ITER_CB = C.CFUNCTYPE(herr_t, hid_t, C.c_char_p, C.c_void_p, C.c_void_p)


class Group(object):
    def __init__(self, gid, name, owns=True, file=None):
        self._id, self.name, self._owns, self._file = gid, name, owns, file

    @property
    def attrs(self):
        return _Attrs(self)

    def __contains__(self, name):
        L = lib()
        with _lock:
            cur = ""
            for part in [p for p in name.split("/") if p]:   # H5Lexists needs every intermediate link to exist
                cur = cur + "/" + part if cur else part
                if L.H5Lexists(self._id, cur.encode(), H5P_DEFAULT) <= 0:
                    return False
        return True

    def keys(self):
        names = []

        def cb(g, nm, info, data):
            names.append(nm.decode("utf8"))
            return 0
        fn = _ITER_CB(cb)
        with _lock:
            idx = hsize_t(0)
            if lib().H5Literate(self._id, H5_INDEX_NAME, H5_ITER_INC, C.byref(idx), fn, None) < 0:
                raise OSError("h5lite: cannot list %s" % self.name)
        return names

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    def _child(self, name):
        return (self.name.rstrip("/") + "/" + name) if self.name else name

    def __getitem__(self, name):
        L = lib()
        if name not in self:
            raise KeyError("Unable to open object (object %r doesn't exist)" % name)
        with _lock:
            oid = L.H5Oopen(self._id, name.encode(), H5P_DEFAULT)
            if oid < 0:
                raise KeyError(name)
            kind = L.H5Iget_type(oid)
            L.H5Oclose(oid)
            if kind == H5I_GROUP:
                return Group(L.H5Gopen2(self._id, name.encode(), H5P_DEFAULT), self._child(name), file=self._file)
            if kind == H5I_DATASET:
                return Dataset(L.H5Dopen2(self._id, name.encode(), H5P_DEFAULT), self._child(name), self._file)
        raise TypeError("h5lite: %r is neither a group nor a dataset" % name)

    def __setitem__(self, name, value):
        """``grp[name] = array`` creates a dataset holding the array (h5py semantics)."""
        arr = np.asarray(value)
        if arr.dtype.kind == "U":
            arr = np.char.encode(arr, "utf8")
        self.create_dataset(name, shape=arr.shape, dtype=arr.dtype, data=arr)

    def __delitem__(self, name):
        if name not in self:
            raise KeyError("Couldn't delete link (name %r doesn't exist)" % name)
        with _lock:
            if lib().H5Ldelete(self._id, name.encode(), H5P_DEFAULT) < 0:
                raise OSError("h5lite: cannot delete %r" % name)

    def create_group(self, name):
        if name in self:
            raise ValueError("Unable to create group (name already exists)")
        with _lock:
            gid = lib().H5Gcreate2(self._id, name.encode(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
        if gid < 0:
            raise OSError("h5lite: cannot create group %r" % name)
        return Group(gid, self._child(name), file=self._file)

    def require_group(self, name):
        if name in self:
            g = self[name]
            if not isinstance(g, Group):
                raise TypeError("Incompatible object (%s) already exists" % type(g).__name__)
            return g
        return self.create_group(name)

    def create_dataset(self, name, shape=None, dtype=None, data=None, maxshape=None, chunks=None):
        L = lib()
        if data is not None:
            data = np.asarray(data)
            if data.dtype.kind == "U":
                data = np.char.encode(data, "utf8")
            if shape is None:
                shape = data.shape
            if dtype is None:
                dtype = data.dtype
        if shape is None:
            raise TypeError("One of data, shape or dtype must be specified")
        if isinstance(shape, (int, np.integer)):
            shape = (int(shape),)
        shape = tuple(int(s) for s in shape)
        dt = np.dtype("f4") if dtype is None else np.dtype(dtype)   # h5py's default dtype is 'f'
        if name in self:
            raise ValueError("Unable to create dataset (name already exists)")
        if maxshape is not None:
            maxshape = tuple(maxshape)
            if len(maxshape) != len(shape):
                raise ValueError("maxshape must have the rank of shape")
        if chunks is True or (chunks is None and maxshape is not None):
            chunks = _guess_chunk(shape, maxshape if maxshape is not None else shape, dt.itemsize)
        elif chunks is not None:
            chunks = tuple(int(c) for c in chunks)
            if len(chunks) != len(shape):
                raise ValueError("chunks must have the rank of shape")
        with _lock:
            tid, own = _native(dt)
            mx = None if maxshape is None else (hsize_t * len(shape))(*[H5S_UNLIMITED if m is None else int(m) for m in maxshape])
            sid = L.H5Screate_simple(len(shape), _dims(shape), mx) if len(shape) else L.H5Screate(0)
            dcpl = L.H5Pcreate(_glob("H5P_CLS_DATASET_CREATE_ID_g"))
            if chunks is not None and len(shape):
                L.H5Pset_chunk(dcpl, len(chunks), _dims(chunks))
            did = L.H5Dcreate2(self._id, name.encode(), tid, sid, H5P_DEFAULT, dcpl, H5P_DEFAULT)
            L.H5Pclose(dcpl); L.H5Sclose(sid)
            if own:
                L.H5Tclose(tid)
        if did < 0:
            raise OSError("h5lite: cannot create dataset %r" % name)
        ds = Dataset(did, self._child(name), self._file)
        if data is not None:
            ds[...] = data.reshape(shape)
        return ds

    def _close(self):
        if self._id and self._owns and self._file is not None and self._file._fid:
            with _lock:
                lib().H5Gclose(self._id)
        self._id = 0

    def __del__(self):
        try:
            self._close()
        except Exception:
            pass


class File(Group):
    """``h5py.File(name, mode, libver=...)`` for modes r, r+, w, w-/x, a."""

    def __init__(self, filename, mode="r", libver=None):
        L = lib()
        self.filename = filename
        fn = os.fsencode(filename)
        with _lock:
            fapl = L.H5Pcreate(_glob("H5P_CLS_FILE_ACCESS_ID_g"))
            L.H5Pset_fclose_degree(fapl, 3)   # H5F_CLOSE_STRONG: closing the file closes what is left open in it
            if libver == "latest":
                L.H5Pset_libver_bounds(fapl, H5F_LIBVER_LATEST, H5F_LIBVER_LATEST)
            elif libver not in (None, "earliest"):
                L.H5Pclose(fapl)
                raise ValueError("h5lite: libver must be None, 'earliest' or 'latest'")
            if mode == "r":
                fid = L.H5Fopen(fn, H5F_ACC_RDONLY, fapl)
            elif mode == "r+":
                fid = L.H5Fopen(fn, H5F_ACC_RDWR, fapl)
            elif mode == "w":
                fid = L.H5Fcreate(fn, H5F_ACC_TRUNC, H5P_DEFAULT, fapl)
            elif mode in ("w-", "x"):
                fid = L.H5Fcreate(fn, H5F_ACC_EXCL, H5P_DEFAULT, fapl)
            elif mode == "a":
                fid = L.H5Fopen(fn, H5F_ACC_RDWR, fapl) if os.path.isfile(filename) else L.H5Fcreate(fn, H5F_ACC_EXCL, H5P_DEFAULT, fapl)
            else:
                L.H5Pclose(fapl)
                raise ValueError("Invalid mode; must be one of r, r+, w, w-, x, a")
            L.H5Pclose(fapl)
        if fid < 0:
            raise OSError("Unable to open file %r (mode %r)" % (filename, mode))
        Group.__init__(self, fid, "/", owns=False, file=self)
        self._fid = fid

    def flush(self):
        with _lock:
            lib().H5Fflush(self._fid, 1)

    def close(self):
        if getattr(self, "_fid", 0):
            with _lock:
                lib().H5Fclose(self._fid)
            self._fid = 0
            self._id = 0

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
