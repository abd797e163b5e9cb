#!/usr/bin/env python3
"""PackStore / NpzStore directory -> HDF5 file with the reference's layout (one group per image or pair name, nested along '/', one dataset per key,
dtypes as stored): the hand-over of a run that wrote the fast stand-in store (SFD2_STORE=pack, or a host without any HDF5 library) to the reference's
consumers, which open the files with h5py (hloc/triangulation.py:57-111, it_loc/localize_cv2.py:677-680; writers extract_localization.py:266-272,
hloc/match_features.py:108-119).

    python tools/pack_to_h5.py outputs/feats-ressegnetv2-n4096-r1600.pack [out.h5]        (default: the same name with .h5)

Runs wherever h5py imports or libhdf5 can be loaded (sfd2_amd/h5lite.py; SFD2_LIBHDF5=<path> to point at one)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main(argv):
    if len(argv) < 2 or argv[1] in ("-h", "--help"):
        print(__doc__)
        return 2
    from sfd2_amd import feature_io as fio
    src = argv[1].rstrip("/")
    if not os.path.isdir(src):
        print(f"{src}: not a store directory", file=sys.stderr)
        return 1
    dst = argv[2] if len(argv) > 2 else os.path.splitext(src)[0] + ".h5"
    backend = fio.hdf5_backend()
    if backend is None:
        print("no HDF5 library on this host: install h5py or set SFD2_LIBHDF5=<path to libhdf5.so>", file=sys.stderr)
        return 1
    t0 = time.perf_counter()
    n = fio.pack_to_h5(src, dst, backend, progress=lambda k: print(f"  {k} groups ...", file=sys.stderr))
    dt = time.perf_counter() - t0
    print(f"{dst}: {n} groups in {dt:.2f} s ({n / max(dt, 1e-9):.0f} groups/s) through {backend.__name__}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
