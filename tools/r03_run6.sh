#!/bin/bash
# round 3, GPU run 6: incremental local map + 8-bit sort, plane-fit differential test, reference tests, functor inputs dump, full suite, local-map timing
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r03f; mkdir -p $OUT
( time timeout 900 python -m pytest tests/test_voxel_gpu.py tests/test_plane_fit_gpu.py tests/test_reference_gpu.py -m gpu -q ) > $OUT/pytest_new.log 2>&1
tail -40 $OUT/pytest_new.log
timeout 300 python tools/dump_functor_inputs.py gpurun_out/functor_inputs.npz 2>&1 | tail -3
( time timeout 1200 python -m pytest tests -m gpu -q --deselect tests/test_voxel_gpu.py --deselect tests/test_plane_fit_gpu.py --deselect tests/test_reference_gpu.py ) > $OUT/pytest_all.log 2>&1
tail -8 $OUT/pytest_all.log
for inc in 1 0; do for dig in 8 4; do
python - <<PY
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
import lili_om_amd as L
from lili_om_amd import synth
w = synth.make_workload(n_map=200_000, half_extent=(150.0, 150.0), verbose=False)
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = L.Context(0, stream=s.cuda_stream)
ctx.set_option("localmap_incremental", $inc); ctx.set_option("sort_digit_bits", $dig)
feats = np.ascontiguousarray(np.concatenate([w["scan_xyz"][::10], np.zeros((w["scan_xyz"][::10].shape[0], 1), np.float32)], 1))
lm = L.api.LocalMap(ctx, L.KIND_SURF, width=50, leaf=0.4, max_sq_radius=1.0)
for k in range(50): lm.push(feats, [0.8 * k, 0.1 * k, 0.0], [1.0, 0.0, 0.0, 0.0])
n_raw, n_map = lm.commit()
kf = [0]
def one():
    kf[0] += 1
    lm.push(feats, [0.8 * (50 + kf[0]), 0.1 * (50 + kf[0]), 0.0], [1.0, 0.0, 0.0, 0.0]); lm.commit()
for _ in range(3): one()
torch.cuda.synchronize(); tic = time.perf_counter()
for _ in range(20): one()
torch.cuda.synchronize()
print("LOCALMAP incremental=$inc digit_bits=$dig: %.4f ms per keyframe (%d -> %d points) stats %s" % ((time.perf_counter() - tic) / 20 * 1e3, n_raw, n_map, lm.stats()))
kfc = np.concatenate([w["scan_xyz"], np.zeros((w["scan_xyz"].shape[0], 1), np.float32)], 1)
L.api.voxel_filter(ctx, kfc, 0.4); torch.cuda.synchronize(); tic = time.perf_counter()
for _ in range(10): L.api.voxel_filter(ctx, kfc, 0.4)
torch.cuda.synchronize(); print("VOXEL200k digit_bits=$dig: %.4f ms" % ((time.perf_counter() - tic) / 10 * 1e3))
ctx.close()
PY
done; done
