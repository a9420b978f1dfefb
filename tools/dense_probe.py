"""Association time on the dense-room map of tests/test_dense_map_gpu.py with the fine index on / off under the round-4 build options (debug aid)."""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import lili_om_amd as L
from lili_om_amd import synth
from test_dense_map_gpu import _dense_room

mp = _dense_room()
rng = np.random.default_rng(7)
pick = rng.choice(mp.shape[0], 6000)
qw = mp[pick].astype(np.float64) + rng.normal(0, 0.01, (6000, 3)) + rng.uniform(-0.1, 0.1, (6000, 3))
t_true = np.array([0.4, -0.3, 1.5]); ang = np.radians(15.0)
q_true = np.array([np.cos(ang / 2), 0, 0, np.sin(ang / 2)])
q_local = synth.quat_rot(q_true * np.array([1, -1, -1, -1]), qw - t_true).astype(np.float32)
P = L.make_params("rot")
for guess, narrow in ((0, 0), (0, 1), (1, 1)):
    ctx = L.Context(0)
    ctx.set_option("map_guess_box", guess); ctx.set_option("map_narrow_counts", narrow)
    for fine in (1, 0, 1, 0):
        ctx.set_option("fine_grid", fine)
        m = L.ScanToMapMatcher(ctx, P)
        m.set_input_cloud(L.KIND_SURF, mp)
        m.set_queries(1, L.KIND_SURF, q_local[600:])
        m.find_corresponding_surf_features(1, q_true, t_true)
        ctx.sync(); tic = time.perf_counter()
        for _ in range(10):
            m.find_corresponding_surf_features(1, q_true, t_true, want_count=False)
        ctx.sync()
        print(f"guess {guess} narrow {narrow} fine {fine}: {(time.perf_counter() - tic) / 10 * 1e6:.1f} us  density {m.map_density(L.KIND_SURF)}  cells {m.map_info(L.KIND_SURF)}  stats {m.map_build_stats()}", flush=True)
    ctx.close()
