"""cProfile of match_features.main (grouped driver) on a synthetic feature store: where the driver's thread spends a query group."""
import cProfile, os, pstats, sys, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from sfd2_amd import feature_io as fio, match_features as mf
td = tempfile.mkdtemp()
rs = np.random.RandomState(0)
NQ, NDB, K = 256, 128, 50
feats_name = "feats-synth"
st = fio.open_store(os.path.join(td, feats_name + ".h5"), "w")
names = [f"db/{i:04d}.jpg" for i in range(NDB)] + [f"query/{i:04d}.jpg" for i in range(NQ)]
for nm in names:
    d = rs.standard_normal((128, 4096)); d /= np.linalg.norm(d, axis=0, keepdims=True)
    st.write_group(nm, {"descriptors": d.astype(np.float64), "keypoints": np.zeros((4096, 2)), "scores": np.zeros(4096), "image_size": np.array([1600, 1200])})
st.close()
pairs = [f"query/{q:04d}.jpg db/{(q * 7 + j) % NDB:04d}.jpg" for q in range(NQ) for j in range(K)]
mf.main(mf.confs["NNM"], pairs[:K * 4], feats_name, td, pairs_name="warm")
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
mf.main(mf.confs["NNM"], pairs, feats_name, td, pairs_name="run")
pr.disable()
dt = time.perf_counter() - t0
print(f"{len(pairs) / dt:.0f} pairs/s under the profiler ({dt * 1e3 / NQ:.3f} ms per query group)")
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
