"""Feature / match stores with the reference's group and dataset names and dtypes.

The reference writes one HDF5 group per image (extract_localization.py:266-270: keypoints
(N,2) f64, descriptors (128,N) f64, scores (N,) f64, image_size (2,)) and one group per pair
(hloc/match_features.py:99-119: matches0 int16 (N,), matching_scores0 fp16 (N,)), and reads
them back as ``f[name]['keypoints'].__array__()`` (it_loc/localize_cv2.py:571-574,
hloc/triangulation.py:57-111).  The stores below ARE HDF5 files with exactly that layout wherever an HDF5 library
exists: through h5py when it imports, otherwise through the HDF5 C library itself (h5lite.py: libhdf5 over ctypes -- this image
carries the C library but not the binding).  ``open_store`` wraps either in H5Store, whose one addition to the h5py.File subset the
pipelines use is that keys() lists the image / pair names (every group holding datasets, full path) instead of the top-level links.
Where no HDF5 library exists, or on request (``standin=`` / SFD2_STORE=pack), the same names and arrays go behind the same mapping
interface into a stand-in, so callers are written once.  Two stand-ins:

  PackStore (``<name>.pack/``)  one append-only data file + a text index, read back through
      one memory map.  Written for the pipelined drivers: a 4 MB feature group is one write(), a
      pair's two small datasets cost microseconds instead of a zip archive each, and readers on
      several threads share the map (no per-group open / CRC pass).
      HDF5 costs ~250 us per pair group in the library itself (one group + two datasets: ~4 k pairs/s, h5py or not), the pipelined
      match driver produces 50-80 k pairs/s: a run that must not wait for its store writes a PackStore and converts afterwards
      (tools/pack_to_h5.py, ``pack_to_h5`` below).
  NpzStore (``<name>.npzdir/``)  one ``.npz`` shard per group; what rounds 2-4 wrote, still read.
"""
import json
import mmap
import os
import threading
import zipfile

import numpy as np

try:  # pragma: no cover - not installed in the build image
    import h5py
except Exception:  # pragma: no cover
    h5py = None


def _shard_name(group):
    return group.replace("%", "%25").replace("/", "%2F") + ".npz"


def _group_name(shard):
    return shard[:-4].replace("%2F", "/").replace("%25", "%")


class _Dataset:
    """What h5py hands out for f[group][key]: supports __array__(), [()] and [...]."""

    def __init__(self, arr):
        self._a = arr
        self.shape, self.dtype = arr.shape, arr.dtype

    def __array__(self, dtype=None, copy=None):
        return self._a if dtype is None else self._a.astype(dtype)

    def __getitem__(self, idx):
        return self._a[idx]

    def __len__(self):
        return len(self._a)


class _NpzGroup:
    def __init__(self, store, name):
        self._store, self._name, self._data = store, name, {}

    def create_dataset(self, key, data=None):
        self._data[key] = np.asarray(data)
        self._store._flush(self._name, self._data)
        return _Dataset(self._data[key])

    def __getitem__(self, key):
        return _Dataset(self._data[key])

    def __contains__(self, key):
        return key in self._data

    def keys(self):
        return self._data.keys()


class NpzStore:
    """Directory of .npz shards with the h5py.File subset the pipelines use:
    create_group / __getitem__ / __contains__ / keys / close / context manager."""
    threadsafe_reads = True

    def __init__(self, path, mode="a"):
        if mode not in ("r", "a", "w"):
            raise ValueError(mode)
        self.path, self.mode = str(path), mode
        if mode == "r" and not os.path.isdir(self.path):
            raise FileNotFoundError(self.path)
        os.makedirs(self.path, exist_ok=True)
        if mode == "w":
            for f in os.listdir(self.path):
                if f.endswith(".npz"):
                    os.remove(os.path.join(self.path, f))

    def _flush(self, name, data):
        if self.mode == "r":
            raise IOError("store opened read-only")
        tmp = os.path.join(self.path, _shard_name(name) + ".tmp")
        with open(tmp, "wb") as fh:
            np.savez(fh, **data)
        os.replace(tmp, os.path.join(self.path, _shard_name(name)))

    def create_group(self, name):
        if name in self:
            raise ValueError(f"Unable to create group (name already exists): {name}")   # h5py's behaviour
        return _NpzGroup(self, name)

    def __contains__(self, name):
        return os.path.exists(os.path.join(self.path, _shard_name(name)))

    def __getitem__(self, name):
        p = os.path.join(self.path, _shard_name(name))
        if not os.path.exists(p):
            raise KeyError(name)
        g = _NpzGroup(self, name)
        try:
            with np.load(p, allow_pickle=False) as z:
                g._data = {k: z[k] for k in z.files}
        except zipfile.BadZipFile as e:
            raise IOError(f"corrupt shard {p}") from e
        return g

    def keys(self):
        return sorted(_group_name(f) for f in os.listdir(self.path) if f.endswith(".npz"))

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class _PackGroup:
    def __init__(self, store, name, entries):
        self._store, self._name, self._entries = store, name, entries    # key -> (dtype str, shape, offset)

    def create_dataset(self, key, data=None):
        if key in self._entries:
            raise ValueError(f"Unable to create dataset (name already exists): {self._name}/{key}")
        self._store._append(self._name, {key: np.asarray(data)}, new_group=False)
        return self[key]

    def __getitem__(self, key):
        dt, shape, off = self._entries[key]
        return _Dataset(self._store._view(dt, shape, off))

    def __contains__(self, key):
        return key in self._entries

    def keys(self):
        return self._entries.keys()


class PackStore:
    """``<path>/data.bin`` (every dataset's bytes, 64-byte aligned, append-only) + ``<path>/index.jsonl`` (one line per
    write: group name and its datasets' dtype / shape / offset).  A group exists once its index line is on disk, and the data
    file is flushed before a line is written, so a killed writer leaves a readable store: trailing data without a line is ignored,
    and on open the index ends at the first line that is torn or names bytes beyond the end of data.bin (later lines are dropped with it).  The h5py.File subset the pipelines
    use -- create_group / __getitem__ / __contains__ / keys / close / context manager -- plus write_group(name, {key:
    array}) (one lock round, one write() per dataset, one index line): what the writer threads of the pipelined drivers
    call.  Thread-safe; reads go through ONE shared read-only memory map (remapped when the file has grown)."""
    threadsafe_reads = True

    def __init__(self, path, mode="a"):
        if mode not in ("r", "a", "w"):
            raise ValueError(mode)
        self.path, self.mode = str(path), mode
        self._data_path = os.path.join(self.path, "data.bin")
        self._index_path = os.path.join(self.path, "index.jsonl")
        if mode == "r" and not os.path.exists(self._index_path):
            raise FileNotFoundError(self.path)
        os.makedirs(self.path, exist_ok=True)
        if mode == "w":
            for f in (self._data_path, self._index_path):
                if os.path.exists(f):
                    os.remove(f)
        self._lock = threading.Lock()
        self._groups = {}          # name -> {key: (dtype str, shape tuple, offset)}
        self._map = None
        self._map_len = 0
        good = 0
        if os.path.exists(self._index_path):
            data_size = os.path.getsize(self._data_path) if os.path.exists(self._data_path) else 0
            with open(self._index_path, "r") as fh:
                for line in fh:
                    if not line.endswith("\n"):
                        break      # a torn last line: the write never completed
                    try:
                        rec = json.loads(line)
                        ends = [int(off) + int(np.prod(shape, dtype=np.int64)) * np.dtype(dt).itemsize for dt, shape, off in rec["d"].values()]
                    except (ValueError, KeyError, TypeError):
                        break      # not a record: everything from here on is dropped
                    if ends and max(ends) > data_size:
                        break      # the line names bytes data.bin does not hold (a writer killed between the two files): appends are
                                   # sequential, so every later line is beyond the end as well -- the store ends at the last whole group
                    good += len(line.encode())
                    g = self._groups.setdefault(rec["g"], {})
                    for key, (dt, shape, off) in rec["d"].items():
                        g[key] = (dt, tuple(shape), int(off))
        self._fd = self._fi = None
        self._end = 0
        if mode != "r":
            if os.path.exists(self._index_path) and os.path.getsize(self._index_path) != good:
                with open(self._index_path, "r+b") as fh:
                    fh.truncate(good)       # drop the torn tail so the next line starts on its own
            self._fd = open(self._data_path, "ab")
            self._fi = open(self._index_path, "a")
            self._end = self._fd.tell()
            # a killed writer may have left data behind the last indexed dataset: new data goes behind it, aligned
            if self._end % 64:
                self._fd.write(b"\0" * (64 - self._end % 64))
                self._end = self._fd.tell()

    # -- writing
    def _append(self, name, datasets, new_group):
        if self.mode == "r":
            raise IOError("store opened read-only")
        arrs = {k: np.ascontiguousarray(v) for k, v in datasets.items()}
        with self._lock:
            if new_group and name in self._groups:
                raise ValueError(f"Unable to create group (name already exists): {name}")   # h5py's behaviour
            g = self._groups.setdefault(name, {})
            rec = {}
            for k, a in arrs.items():
                if k in g:
                    raise ValueError(f"Unable to create dataset (name already exists): {name}/{k}")
                off = self._end
                self._fd.write(a.data if a.size else b"")
                pad = (-a.nbytes) % 64
                if pad:
                    self._fd.write(b"\0" * pad)
                self._end += a.nbytes + pad
                rec[k] = (a.dtype.str, list(a.shape), off)
                g[k] = (a.dtype.str, tuple(a.shape), off)
            self._fd.flush()    # data first: an index line never reaches the OS ahead of the bytes it names
            self._fi.write(json.dumps({"g": name, "d": rec}, separators=(",", ":")) + "\n")

    def write_group(self, name, datasets):
        """create_group + one create_dataset per item, as one append."""
        self._append(name, datasets, new_group=True)

    def write_rows(self, names, blocks):
        """len(names) new groups at once: row i of every 2-D array in `blocks` ({key: [k, n] array}) becomes dataset `key` of group names[i].
        One lock round, ONE write() per block and one for the k index lines (the match driver's writer: a query's 50 pair groups cost 50 x 25 us as
        single appends, its [50, 4096] int16 / fp16 blocks are two contiguous pieces of memory).  Rows keep the 64-byte alignment of single appends
        (a block whose row size is not a multiple of 64 bytes goes row by row)."""
        if self.mode == "r":
            raise IOError("store opened read-only")
        arrs = {k: np.ascontiguousarray(v) for k, v in blocks.items()}
        k = len(names)
        if any(a.ndim != 2 or a.shape[0] != k for a in arrs.values()):
            raise ValueError("write_rows: every block must be [len(names), n]")
        if any((a.shape[1] * a.dtype.itemsize) % 64 for a in arrs.values()) or k == 0:
            for i, nm in enumerate(names):
                self._append(nm, {key: a[i] for key, a in arrs.items()}, new_group=True)
            return
        with self._lock:
            for nm in names:
                if nm in self._groups:
                    raise ValueError(f"Unable to create group (name already exists): {nm}")
            if len(set(names)) != k:
                raise ValueError("write_rows: duplicate group names")
            base = {}
            for key, a in arrs.items():
                base[key] = self._end
                self._fd.write(a.data if a.size else b"")
                self._end += a.nbytes
            lines = []
            for i, nm in enumerate(names):
                rec, g = {}, {}
                for key, a in arrs.items():
                    off = base[key] + i * a.shape[1] * a.dtype.itemsize
                    rec[key] = (a.dtype.str, [int(a.shape[1])], off)
                    g[key] = (a.dtype.str, (int(a.shape[1]),), off)
                self._groups[nm] = g
                lines.append(json.dumps({"g": nm, "d": rec}, separators=(",", ":")))
            self._fd.flush()    # data first, as in _append
            self._fi.write("\n".join(lines) + "\n")

    def create_group(self, name):
        self._append(name, {}, new_group=True)
        return _PackGroup(self, name, self._groups[name])

    def flush(self):
        with self._lock:
            if self._fd is not None:
                self._fd.flush()
                self._fi.flush()

    # -- reading
    def _view(self, dt, shape, off):
        dtype = np.dtype(dt)
        n = int(np.prod(shape, dtype=np.int64)) if len(shape) else 1
        if n == 0:
            return np.zeros(shape, dtype=dtype)
        need = off + n * dtype.itemsize
        with self._lock:
            if self._map is None or self._map_len < need:
                if self._fd is not None:
                    self._fd.flush()
                with open(self._data_path, "rb") as fh:
                    size = os.fstat(fh.fileno()).st_size
                    if size < need:
                        raise IOError(f"corrupt store {self.path}: dataset beyond the end of data.bin")
                    self._map = mmap.mmap(fh.fileno(), size, access=mmap.ACCESS_READ)   # older views keep the old map alive
                    self._map_len = size
            m = self._map
        return np.frombuffer(m, dtype=dtype, count=n, offset=off).reshape(shape)

    def __contains__(self, name):
        return name in self._groups

    def __getitem__(self, name):
        if name not in self._groups:
            raise KeyError(name)
        return _PackGroup(self, name, self._groups[name])

    def keys(self):
        return sorted(self._groups.keys())

    def close(self):
        with self._lock:
            if self._fd is not None:
                self._fd.close()
                self._fi.close()
                self._fd = self._fi = None
                self.mode = "r"

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class H5Store:
    """An HDF5 file behind the stores' interface.  backend: the h5py module, or sfd2_amd.h5lite (same File / group / dataset surface).
    create_group / write_group / write_rows / __getitem__ / __contains__ / close / context manager as the stand-ins; keys() = the names the
    groups were written with (full paths of every group that holds datasets -- 'db/1.jpg' is HDF5 group '1.jpg' inside group 'db'), in
    name order.  Thread-safe (one lock: neither libhdf5 nor an h5py.File is safe for concurrent use)."""
    threadsafe_reads = True

    def __init__(self, path, mode, backend):
        self.path, self.mode, self._backend = str(path), mode, backend
        self._lock = threading.RLock()
        self._f = backend.File(self.path, mode)
        self._names = None          # filled by the first keys()

    def _walk(self):
        if hasattr(self._f, "leaf_groups"):
            return list(self._f.leaf_groups())
        out = []

        def visit(name, obj):
            if hasattr(obj, "keys"):                      # a group
                kids = list(obj.keys())
                if not kids or any(not hasattr(obj[k], "keys") for k in kids):
                    out.append(name)
        self._f.visititems(visit)
        return sorted(out)

    def keys(self):
        with self._lock:
            if self._names is None:
                self._names = set(self._walk())
            return sorted(self._names)

    def __contains__(self, name):
        with self._lock:
            return name in self._f

    def __getitem__(self, name):
        with self._lock:
            if name not in self._f:
                raise KeyError(name)
            return self._f[name]

    def create_group(self, name):
        with self._lock:
            if self.mode == "r":
                raise IOError("store opened read-only")
            g = self._f.create_group(name)
            if self._names is not None:
                self._names.add(name)
            return g

    def write_group(self, name, datasets):
        with self._lock:
            if hasattr(self._f, "write_group"):
                if self.mode == "r":
                    raise IOError("store opened read-only")
                self._f.write_group(name, datasets)
                if self._names is not None:
                    self._names.add(name)
                return
            g = self.create_group(name)
            for k, v in datasets.items():
                g.create_dataset(k, data=v)

    def write_rows(self, names, blocks):
        """Row i of every [k, n] block becomes dataset `key` of group names[i] (PackStore.write_rows' contract; here one group at a time -- the format's cost)."""
        arrs = {k: np.asarray(v) for k, v in blocks.items()}
        if any(a.ndim != 2 or a.shape[0] != len(names) for a in arrs.values()):
            raise ValueError("write_rows: every block must be [len(names), n]")
        with self._lock:
            for nm in names:
                if nm in self._f:
                    raise ValueError(f"Unable to create group (name already exists): {nm}")
            if len(set(names)) != len(names):
                raise ValueError("write_rows: duplicate group names")
            for i, nm in enumerate(names):
                self.write_group(nm, {k: a[i] for k, a in arrs.items()})

    def flush(self):
        with self._lock:
            self._f.flush()

    def close(self):
        with self._lock:
            self._f.close()
            self.mode = "r"

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


STANDIN = "pack"     # the stand-in a '.h5' path becomes when no HDF5 library exists or one is asked for: "pack" (PackStore) or "npz" (NpzStore)
STORE = os.environ.get("SFD2_STORE", "auto")     # "auto": HDF5 when a library exists, else STANDIN; "h5" (required); "pack" / "npz": that stand-in


def hdf5_backend():
    """h5py when it imports, else h5lite when the HDF5 C library loads, else None."""
    if h5py is not None:
        return h5py
    from . import h5lite
    return h5lite if h5lite.available() else None


def open_store(path, mode="a", standin=None):
    """``path`` ending in .h5 -> an HDF5 file (the reference's format) through h5py or, without it, the HDF5 C library (H5Store).  With no HDF5
    library on the host, or with ``standin="pack"`` / ``"npz"`` (module default STORE, environment SFD2_STORE), a stand-in at ``path`` with the trailing
    .h5 replaced: a PackStore (<name>.pack) or an NpzStore (<name>.npzdir).  Reading (mode "r") without an explicit choice picks whichever of the three
    exists, the HDF5 file first.  A path ending in .pack / .npzdir names a stand-in directly."""
    path = str(path)
    base = path[:-3] if path.endswith(".h5") else None
    if base is None:
        if path.endswith(".npzdir"):
            return NpzStore(path, mode)
        return PackStore(path, mode)
    kind = standin or (STORE if STORE != "auto" else None)
    if kind not in (None, "h5", "pack", "npz"):
        raise ValueError(f"unknown stand-in store {kind!r}")
    backend = hdf5_backend() if kind in (None, "h5") or mode == "r" else None
    if kind == "h5" and backend is None:
        raise RuntimeError("SFD2_STORE=h5 but neither h5py nor the HDF5 C library (sfd2_amd/h5lite.py) is available")
    if mode == "r":
        # whatever exists, the asked-for kind first, then HDF5, PackStore, NpzStore
        for k in ([kind] if kind else []) + ["h5", "pack", "npz"]:
            if k == "h5" and backend is not None and os.path.isfile(path):
                return H5Store(path, mode, backend)
            if k == "pack" and os.path.exists(os.path.join(base + ".pack", "index.jsonl")):
                return PackStore(base + ".pack", mode)
            if k == "npz" and os.path.isdir(base + ".npzdir"):
                return NpzStore(base + ".npzdir", mode)
        raise FileNotFoundError(path)
    if backend is not None:
        return H5Store(path, mode, backend)
    kind = kind or STANDIN
    if kind == "npz":
        return NpzStore(base + ".npzdir", mode)
    return PackStore(base + ".pack", mode)


def pack_to_h5(src, dst, backend=None, progress=None):
    """Every group of a stand-in store (PackStore / NpzStore directory) into an HDF5 file with the reference's layout: the hand-over of a fast run's
    output to the reference's consumers (hloc/triangulation.py:57-111, it_loc/localize_cv2.py:677-680).  Returns the number of groups written."""
    backend = backend or hdf5_backend()
    if backend is None:
        raise RuntimeError("no HDF5 library (h5py or libhdf5) on this host")
    rd = NpzStore(src, "r") if str(src).endswith(".npzdir") else PackStore(src, "r")
    n = 0
    with H5Store(dst, "w", backend) as out:
        for name in rd.keys():
            g = rd[name]
            out.write_group(name, {k: g[k].__array__() for k in g.keys()})
            n += 1
            if progress is not None and n % 1000 == 0:
                progress(n)
    rd.close()
    return n


def write_features(store, name, pred):
    """extract_localization.py:266-270: one group per image, one dataset per key, dtypes as produced
    (keypoints / descriptors / scores float64, image_size integer)."""
    if hasattr(store, "write_group"):
        store.write_group(name, pred)
        return store[name]
    grp = store.create_group(name)
    for k, v in pred.items():
        grp.create_dataset(k, data=v)
    return grp


def write_matches(store, pair, matches0, scores0):
    """hloc/match_features.py:108-116: matches0 -> int16, matching_scores0 -> fp16."""
    from .match_features import cast_for_storage
    m, s = cast_for_storage(matches0, scores0)
    if hasattr(store, "write_group"):
        store.write_group(pair, {"matches0": m, "matching_scores0": s})
        return store[pair]
    grp = store.create_group(pair)
    grp.create_dataset("matches0", data=m)
    grp.create_dataset("matching_scores0", data=s)
    return grp
