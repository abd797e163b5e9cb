"""HDF5 files without h5py: the subset of h5py.File the feature / match stores use, over the HDF5 C library through ctypes.

The reference writes and reads its stores with h5py (extract_localization.py:266-272, hloc/match_features.py:108-119,
hloc/triangulation.py:57-62, it_loc/localize_cv2.py:677-680).  h5py is a binding of libhdf5; this image carries the C library
(/opt/conda/lib/libhdf5.so, 1.10.6) but not the binding, so the stores below call the same library directly -- the files are ordinary
HDF5 (h5dump lists them, h5py opens them): nested groups along the '/' of an image name, one dataset per key, little-endian IEEE
types, fp16 as the 16-bit float type h5py itself builds (sign 15, exponent 10..14 bias 15, mantissa 0..9).

    f = h5lite.File(path, "w" | "a" | "r")          g = f.create_group("db/1.jpg")       g.create_dataset("keypoints", data=a)
    f["db/1.jpg"]["keypoints"].__array__()          "db/1.jpg" in f                      f.keys() (top level, like h5py)
    f.leaf_groups()  (every group that holds datasets, full names, file order)            f.write_group(name, {key: array})

Not a general binding: contiguous datasets of fixed numeric types, whole-dataset reads and writes.  libhdf5 is not thread-safe in
this build; every call goes through one module-wide lock (h5py does the same).
"""
import ctypes
import ctypes.util
import os
import threading
import weakref

import numpy as np

_lock = threading.RLock()
_lib = None
_ids = {}
_open_files = {}        # realpath -> WeakSet of open File objects
_spaces = {}            # shape -> dataspace id (dataspaces belong to no file: kept for the process)

H5F_ACC_RDONLY, H5F_ACC_RDWR, H5F_ACC_TRUNC, H5F_ACC_EXCL = 0, 1, 2, 4
H5P_DEFAULT = 0
H5S_ALL = 0
H5T_INTEGER, H5T_FLOAT = 0, 1
H5_INDEX_NAME, H5_INDEX_CRT_ORDER = 0, 1
H5_ITER_INC, H5_ITER_NATIVE = 0, 2
H5O_TYPE_GROUP, H5O_TYPE_DATASET = 0, 1

hid_t = ctypes.c_int64
hsize_t = ctypes.c_uint64


def _candidates():
    env = os.environ.get("SFD2_LIBHDF5")
    if env:
        yield env
    found = ctypes.util.find_library("hdf5")
    if found:
        yield found
    for d in ("/opt/conda/lib", "/usr/lib/x86_64-linux-gnu", "/usr/lib/x86_64-linux-gnu/hdf5/serial", "/usr/lib64", "/usr/local/lib"):
        for n in ("libhdf5.so", "libhdf5_serial.so"):
            yield os.path.join(d, n)
        if os.path.isdir(d):
            for n in sorted(os.listdir(d)):
                if n.startswith(("libhdf5.so.", "libhdf5_serial.so.")):
                    yield os.path.join(d, n)


def load():
    """The HDF5 C library, or None when this host has none (callers then fall back to the stand-in stores)."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib or None
        for cand in _candidates():
            try:
                lib = ctypes.CDLL(cand)
                lib.H5open.restype = ctypes.c_int
                if lib.H5open() < 0:
                    continue
                maj, mnr, rel = ctypes.c_uint(), ctypes.c_uint(), ctypes.c_uint()
                lib.H5get_libversion(ctypes.byref(maj), ctypes.byref(mnr), ctypes.byref(rel))
                if (maj.value, mnr.value) < (1, 10):
                    continue           # hid_t is 32 bits before 1.10: not this binding's ABI
                _bind(lib)
                lib.H5Eset_auto2(hid_t(0), None, None)       # errors are return codes here, not a stack dump on stderr
                lib.version = (maj.value, mnr.value, rel.value)
                lib.path = cand
                _lib = lib
                return lib
            except (OSError, AttributeError):
                continue
        _lib = False
        return None


def available():
    return load() is not None


def _bind(lib):
    P, I, U, S = ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_char_p
    sig = {
        "H5Fcreate": (hid_t, [S, U, hid_t, hid_t]), "H5Fopen": (hid_t, [S, U, hid_t]), "H5Fclose": (I, [hid_t]), "H5Fflush": (I, [hid_t, I]),
        "H5Gcreate2": (hid_t, [hid_t, S, hid_t, hid_t, hid_t]), "H5Gopen2": (hid_t, [hid_t, S, hid_t]), "H5Gclose": (I, [hid_t]),
        "H5Pcreate": (hid_t, [hid_t]), "H5Pset_create_intermediate_group": (I, [hid_t, U]), "H5Pclose": (I, [hid_t]),
        "H5Pset_fclose_degree": (I, [hid_t, I]), "H5Pset_libver_bounds": (I, [hid_t, I, I]),
        "H5Screate_simple": (hid_t, [I, P, P]), "H5Screate": (hid_t, [I]), "H5Sclose": (I, [hid_t]), "H5Sget_simple_extent_ndims": (I, [hid_t]),
        "H5Sget_simple_extent_dims": (I, [hid_t, P, P]),
        "H5Dcreate2": (hid_t, [hid_t, S, hid_t, hid_t, hid_t, hid_t, hid_t]), "H5Dopen2": (hid_t, [hid_t, S, hid_t]), "H5Dclose": (I, [hid_t]),
        "H5Dwrite": (I, [hid_t, hid_t, hid_t, hid_t, hid_t, P]), "H5Dread": (I, [hid_t, hid_t, hid_t, hid_t, hid_t, P]),
        "H5Dget_space": (hid_t, [hid_t]), "H5Dget_type": (hid_t, [hid_t]),
        "H5Tcopy": (hid_t, [hid_t]), "H5Tclose": (I, [hid_t]), "H5Tget_class": (I, [hid_t]), "H5Tget_size": (ctypes.c_size_t, [hid_t]),
        "H5Tget_sign": (I, [hid_t]), "H5Tset_fields": (I, [hid_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_size_t]),
        "H5Tset_size": (I, [hid_t, ctypes.c_size_t]), "H5Tset_ebias": (I, [hid_t, ctypes.c_size_t]), "H5Tset_precision": (I, [hid_t, ctypes.c_size_t]),
        "H5Tset_offset": (I, [hid_t, ctypes.c_size_t]), "H5Tget_order": (I, [hid_t]),
        "H5Lexists": (I, [hid_t, S, hid_t]), "H5Oopen": (hid_t, [hid_t, S, hid_t]), "H5Oclose": (I, [hid_t]), "H5Iget_type": (I, [hid_t]),
        "H5Lget_name_by_idx": (ctypes.c_ssize_t, [hid_t, S, I, I, hsize_t, P, ctypes.c_size_t, hid_t]),
        "H5Gget_info": (I, [hid_t, P]), "H5Oget_info_by_name": (I, [hid_t, S, P, hid_t]),
        "H5Eset_auto2": (I, [hid_t, P, P]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    for sym in ("H5T_NATIVE_DOUBLE_g", "H5T_NATIVE_FLOAT_g", "H5T_IEEE_F64LE_g", "H5T_IEEE_F32LE_g", "H5T_STD_I8LE_g", "H5T_STD_I16LE_g", "H5T_STD_I32LE_g",
                "H5T_STD_I64LE_g", "H5T_STD_U8LE_g", "H5T_STD_U16LE_g", "H5T_STD_U32LE_g", "H5T_STD_U64LE_g", "H5P_CLS_LINK_CREATE_ID_g",
                "H5P_CLS_FILE_ACCESS_ID_g"):
        _ids[sym] = hid_t.in_dll(lib, sym).value
    # IEEE binary16 as h5py defines it (h5py/h5t.pyx: copy of IEEE_F32LE with 16-bit fields)
    t = lib.H5Tcopy(_ids["H5T_IEEE_F32LE_g"])
    if lib.H5Tset_fields(t, 15, 10, 5, 0, 10) < 0 or lib.H5Tset_size(t, 2) < 0 or lib.H5Tset_ebias(t, 15) < 0:
        raise OSError("libhdf5: cannot build the 16-bit float type")
    _ids["F16LE"] = t
    lcpl = lib.H5Pcreate(_ids["H5P_CLS_LINK_CREATE_ID_g"])
    lib.H5Pset_create_intermediate_group(lcpl, 1)
    _ids["LCPL_MKPARENTS"] = lcpl
    fapl = lib.H5Pcreate(_ids["H5P_CLS_FILE_ACCESS_ID_g"])
    lib.H5Pset_fclose_degree(fapl, 3)        # H5F_CLOSE_STRONG: closing the file closes the groups still open on it (h5py's default as well)
    # object formats from HDF5 1.8 on (compact link storage in groups instead of the 1.6 symbol tables): a pair group with its two datasets is created
    # 1.3x faster, every HDF5 >= 1.8 reads the file (h5py has been built on >= 1.8 since 2.0).  SFD2_H5_LIBVER=earliest keeps the oldest formats.
    if os.environ.get("SFD2_H5_LIBVER", "v18").lower() != "earliest":
        lib.H5Pset_libver_bounds(fapl, 1, 2)     # H5F_LIBVER_V18 .. H5F_LIBVER_LATEST (= V110 in 1.10: nothing written here needs it)
    _ids["FAPL_STRONG"] = fapl


_FILE_TYPES = {"f8": "H5T_IEEE_F64LE_g", "f4": "H5T_IEEE_F32LE_g", "f2": "F16LE", "i1": "H5T_STD_I8LE_g", "i2": "H5T_STD_I16LE_g", "i4": "H5T_STD_I32LE_g",
               "i8": "H5T_STD_I64LE_g", "u1": "H5T_STD_U8LE_g", "u2": "H5T_STD_U16LE_g", "u4": "H5T_STD_U32LE_g", "u8": "H5T_STD_U64LE_g"}


_type_cache = {}


def _h5type(dtype):
    t = _type_cache.get(dtype)
    if t is None:
        dt = np.dtype(dtype)
        key = dt.kind + str(dt.itemsize)
        if key not in _FILE_TYPES or dt.byteorder == ">":
            raise TypeError(f"h5lite: dtype {dt} is not one the stores hold")
        t = _type_cache[dtype] = _ids[_FILE_TYPES[key]]
    return t


_tls = threading.local()


def _quiet():
    """A thread-safe libhdf5 keeps an error stack PER THREAD, each with automatic printing on: switch it off once per thread (errors are return codes here)."""
    if not getattr(_tls, "quiet", False):
        load().H5Eset_auto2(hid_t(0), None, None)
        _tls.quiet = True


def _chk(rc, what):
    if rc < 0:
        raise OSError(f"libhdf5: {what} failed")
    return rc


class Dataset:
    """f[group][key]: shape, dtype, __array__(), [()] / [...] (the whole dataset is read on first use and kept)."""

    def __init__(self, parent_id, name, full, open_id=None):
        lib = load()
        self.name = full
        with _lock:
            d = open_id if open_id is not None else _chk(lib.H5Dopen2(parent_id, name.encode(), H5P_DEFAULT), f"open dataset {full}")
            try:
                sp = lib.H5Dget_space(d)
                nd = lib.H5Sget_simple_extent_ndims(sp)
                dims = (hsize_t * max(nd, 1))()
                if nd > 0:
                    lib.H5Sget_simple_extent_dims(sp, dims, None)
                lib.H5Sclose(sp)
                t = lib.H5Dget_type(d)
                cls, size, sign = lib.H5Tget_class(t), lib.H5Tget_size(t), lib.H5Tget_sign(t)
                lib.H5Tclose(t)
                if cls == H5T_FLOAT:
                    dt = np.dtype(f"<f{size}")
                elif cls == H5T_INTEGER:
                    dt = np.dtype(f"<{'i' if sign else 'u'}{size}")
                else:
                    raise TypeError(f"h5lite: dataset {full} has a type class ({cls}) the stores do not use")
                self.shape, self.dtype = tuple(int(dims[i]) for i in range(nd)), dt
                out = np.empty(self.shape, dtype=dt)
                if out.size:
                    _chk(lib.H5Dread(d, _h5type(dt), H5S_ALL, H5S_ALL, H5P_DEFAULT, out.ctypes.data), f"read {full}")
                self._a = out
            finally:
                lib.H5Dclose(d)

    def __array__(self, dtype=None, copy=None):
        return self._a if dtype is None else self._a.astype(dtype)

    def __getitem__(self, idx):
        return self._a[idx]

    def __len__(self):
        return len(self._a)


class Group:
    def __init__(self, file, gid, name):
        self._file, self._id, self.name = file, gid, name

    def _close(self):
        if self._id:
            with _lock:
                load().H5Oclose(self._id)
            self._id = 0

    def __del__(self):
        try:
            if self._file._id:          # (ids die with their file)
                self._close()
        except Exception:
            pass

    def _full(self, key):
        return (self.name.rstrip("/") + "/" + key) if self.name != "/" else "/" + key

    def create_dataset(self, key, data=None, dtype=None):
        lib = load()
        a = np.ascontiguousarray(data if dtype is None else np.asarray(data, dtype=dtype))
        if a.dtype.byteorder == ">":
            a = a.astype(a.dtype.newbyteorder("<"))
        if a.dtype == np.bool_:
            a = a.astype(np.uint8)
        t = _h5type(a.dtype)
        with _lock:
            self._file._writable()
            if lib.H5Lexists(self._id, key.encode(), H5P_DEFAULT) > 0:
                raise ValueError(f"Unable to create dataset (name already exists): {self._full(key)}")      # h5py's message
            if a.ndim:
                dims = (hsize_t * a.ndim)(*a.shape)
                sp = _chk(lib.H5Screate_simple(a.ndim, dims, None), "dataspace")
            else:
                sp = _chk(lib.H5Screate(0), "scalar dataspace")        # H5S_SCALAR
            try:
                d = _chk(lib.H5Dcreate2(self._id, key.encode(), t, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT), f"create dataset {self._full(key)}")
                try:
                    if a.size:
                        _chk(lib.H5Dwrite(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data), f"write {self._full(key)}")
                finally:
                    lib.H5Dclose(d)
            finally:
                lib.H5Sclose(sp)
        return None

    def _names(self, gid=None):
        lib = load()
        gid = self._id if gid is None else gid
        out = []
        with _lock:
            i = 0
            buf = ctypes.create_string_buffer(4096)
            while True:
                n = lib.H5Lget_name_by_idx(gid, b".", H5_INDEX_NAME, H5_ITER_INC, i, buf, 4096, H5P_DEFAULT)
                if n < 0:
                    break
                out.append(buf.value.decode())
                i += 1
        return out

    def keys(self):
        return self._names()

    def _open_any(self, gid, key):
        """(id, is_group) of a child that exists -- H5Oopen + H5Iget_type: no call that fails on purpose (a thread-safe libhdf5 keeps an error stack per thread and
        prints it unless every thread has switched that off)."""
        lib = load()
        oid = _chk(lib.H5Oopen(gid, key.encode(), H5P_DEFAULT), f"open {key}")
        return oid, lib.H5Iget_type(oid) == 2        # H5I_GROUP

    def _kind(self, key):
        """'group' / 'dataset' / None of a direct child."""
        lib = load()
        with _lock:
            if lib.H5Lexists(self._id, key.encode(), H5P_DEFAULT) <= 0:
                return None
            oid, is_group = self._open_any(self._id, key)
            lib.H5Oclose(oid)
            return "group" if is_group else "dataset"

    def __contains__(self, name):
        lib = load()
        _quiet()
        with _lock:
            cur = ""
            for part in [p for p in name.split("/") if p]:          # H5Lexists wants every intermediate link to exist
                cur = part if not cur else cur + "/" + part
                if lib.H5Lexists(self._id, cur.encode(), H5P_DEFAULT) <= 0:
                    return False
            return True

    def __getitem__(self, name):
        lib = load()
        if name not in self:
            raise KeyError(name)
        with _lock:
            oid, is_group = self._open_any(self._id, name)
            if is_group:
                return Group(self._file, oid, self._full(name))
            return Dataset(self._id, name, self._full(name), open_id=oid)      # (closes the id)

    def create_group(self, name):
        lib = load()
        with _lock:
            self._file._writable()
            if name in self:
                raise ValueError(f"Unable to create group (name already exists): {self._full(name)}")
            g = _chk(lib.H5Gcreate2(self._id, name.encode(), _ids["LCPL_MKPARENTS"], H5P_DEFAULT, H5P_DEFAULT), f"create group {self._full(name)}")
        return Group(self._file, g, self._full(name))


class File(Group):
    """h5py.File's role: File(path, mode) with mode 'r' (must exist), 'a' (read / write, created if missing), 'w' (truncate)."""
    threadsafe_reads = True           # (serialised by the module lock)

    def __init__(self, path, mode="a"):
        lib = load()
        if lib is None:
            raise OSError("h5lite: no HDF5 C library on this host (set SFD2_LIBHDF5=<path to libhdf5.so>)")
        if mode not in ("r", "a", "w", "r+"):
            raise ValueError(mode)
        self.path, self.mode = str(path), mode
        self._id = 0
        _quiet()
        key = os.path.realpath(self.path)

        def attempt():
            if mode == "w" or (mode == "a" and not os.path.exists(self.path)):
                return lib.H5Fcreate(self.path.encode(), H5F_ACC_TRUNC, H5P_DEFAULT, _ids["FAPL_STRONG"])
            if not os.path.exists(self.path):
                raise FileNotFoundError(self.path)
            return lib.H5Fopen(self.path.encode(), H5F_ACC_RDONLY if mode == "r" else H5F_ACC_RDWR, _ids["FAPL_STRONG"])
        with _lock:
            fid = attempt()
            if fid < 0 and mode != "r":
                # libhdf5 refuses a second open of one file with other access flags inside a process.  A READ handle somebody forgot (datasets handed
                # out are arrays in memory already) must not block the writer: read-only handles of this path are closed, then once more
                for other in list(_open_files.get(key, ())):
                    if other.mode == "r":
                        other.close()
                fid = attempt()
            if fid < 0:
                raise OSError(f"libhdf5: cannot open {self.path} (mode {mode})" + (": open for writing elsewhere in this process" if _open_files.get(key) else ""))
            _open_files.setdefault(key, weakref.WeakSet()).add(self)
        Group.__init__(self, self, fid, "/")

    def _writable(self):
        if not self._id:
            raise ValueError("file is closed")
        if self.mode == "r":
            raise IOError("store opened read-only")

    def write_group(self, name, datasets):
        """create_group + one create_dataset per item under one lock round (what the pipelined drivers' writer threads call).  The group is new, so no dataset
        name can be taken: no existence checks; dataspaces of the shapes seen so far are kept (a match store writes two shapes 200 000 times)."""
        lib = load()
        _quiet()
        arrs = {}
        for k, v in datasets.items():
            a = v if (type(v) is np.ndarray and v.flags.c_contiguous) else np.ascontiguousarray(v)
            if a.dtype.byteorder == ">":
                a = a.astype(a.dtype.newbyteorder("<"))
            if a.dtype == np.bool_:
                a = a.astype(np.uint8)
            arrs[k] = (a, _h5type(a.dtype))
        with _lock:
            self._writable()
            g = lib.H5Gcreate2(self._id, name.encode(), _ids["LCPL_MKPARENTS"], H5P_DEFAULT, H5P_DEFAULT)
            if g < 0:
                if name in self:
                    raise ValueError(f"Unable to create group (name already exists): /{name}")
                raise OSError(f"libhdf5: create group /{name} failed")
            try:
                for k, (a, t) in arrs.items():
                    sp = _spaces.get(a.shape)
                    if sp is None:
                        dims = (hsize_t * max(a.ndim, 1))(*a.shape)
                        sp = _chk(lib.H5Screate_simple(a.ndim, dims, None), "dataspace")
                        if len(_spaces) < 256:
                            _spaces[a.shape] = sp
                    d = lib.H5Dcreate2(g, k.encode(), t, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT)
                    if sp is not _spaces.get(a.shape):
                        lib.H5Sclose(sp)
                    _chk(d, f"create dataset /{name}/{k}")
                    try:
                        if a.size:
                            _chk(lib.H5Dwrite(d, t, H5S_ALL, H5S_ALL, H5P_DEFAULT, a.ctypes.data), f"write /{name}/{k}")
                    finally:
                        lib.H5Dclose(d)
            finally:
                lib.H5Gclose(g)

    def leaf_groups(self):
        """Full names (without the leading '/') of every group that holds at least one dataset, depth first in name order: the image / pair
        names the stores were written with ('db/1.jpg' is group '1.jpg' inside group 'db')."""
        lib = load()
        out = []

        def walk(gid, prefix):
            names = self._names(gid)
            groups, has_dataset = [], False
            for n in names:
                oid, is_group = self._open_any(gid, n)
                if is_group:
                    groups.append((n, oid))
                else:
                    has_dataset = True
                    lib.H5Oclose(oid)
            if prefix and (has_dataset or not names):
                out.append(prefix)               # (an empty group is still a name somebody wrote)
            for n, g in groups:
                try:
                    walk(g, n if not prefix else prefix + "/" + n)
                finally:
                    lib.H5Oclose(g)
        with _lock:
            walk(self._id, "")
        return out

    def flush(self):
        with _lock:
            if self._id:
                load().H5Fflush(self._id, 1)

    def close(self):
        with _lock:
            if self._id:
                load().H5Fclose(self._id)
                self._id = 0
                ws = _open_files.get(os.path.realpath(self.path))
                if ws is not None:
                    ws.discard(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
