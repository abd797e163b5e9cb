"""Feature / match stores with the reference's group and dataset names and dtypes.

The reference writes one HDF5 group per image (extract_localization.py:266-270: keypoints
(N,2) f64, descriptors (128,N) f64, scores (N,) f64, image_size (2,)) and one group per pair
(hloc/match_features.py:99-119: matches0 int16 (N,), matching_scores0 fp16 (N,)), and reads
them back as ``f[name]['keypoints'].__array__()`` (it_loc/localize_cv2.py:571-574,
hloc/triangulation.py:57-111).  h5py is a third-party dependency that this image does not
carry: when it imports, the stores below ARE h5py files with exactly that layout; when it does
not, the same names and arrays go into a directory of ``.npz`` shards (one per group) behind
the same mapping interface, so callers are written once.
"""
import os
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


def open_store(path, mode="a"):
    """``path`` ending in .h5 with h5py importable -> h5py.File (the reference's format);
    otherwise an NpzStore at ``path`` (a trailing .h5 becomes .npzdir)."""
    path = str(path)
    if h5py is not None and path.endswith(".h5"):
        return h5py.File(path, mode)
    if path.endswith(".h5"):
        path = path[:-3] + ".npzdir"
    return NpzStore(path, mode)


def write_features(store, name, pred):
    """extract_localization.py:266-270: one group per image, one dataset per key, dtypes as produced
    (keypoints / descriptors / scores float64, image_size integer)."""
    grp = store.create_group(name)
    for k, v in pred.items():
        grp.create_dataset(k, data=v)
    return grp


def write_matches(store, pair, matches0, scores0):
    """hloc/match_features.py:108-116: matches0 -> int16, matching_scores0 -> fp16."""
    from .match_features import cast_for_storage
    m, s = cast_for_storage(matches0, scores0)
    grp = store.create_group(pair)
    grp.create_dataset("matches0", data=m)
    grp.create_dataset("matching_scores0", data=s)
    return grp
