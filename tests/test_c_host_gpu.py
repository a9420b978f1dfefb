"""The C ABI driven by a plain C++ host program (examples/s2m_demo.cpp: no Python, no PyTorch in that process):
its result must be bit-identical to the same calls made through the ctypes binding."""
import os
import struct
import subprocess

import numpy as np
import pytest

import lili_om_amd as L
from lili_om_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEMO = os.path.join(ROOT, "examples", "s2m_demo")


@pytest.mark.parametrize("variant", ["rot", "frontend"])
def test_cpp_host_matches_python_binding(gpu_ctx, tmp_path, variant):
    if not os.path.exists(DEMO):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-s"])
    room = synth.make_room(seed=31, n_query=7000, n_edge_query=100)
    P = L.make_params(variant)
    t, q = room["t_true"], room["q_true"]
    if variant == "rot":
        t, q = L.api.body_pose_from_lidar(t, q, P)
    t0, q0 = synth.perturbed_pose(t, q, np.random.default_rng(17), 0.1, 0.8)
    path = tmp_path / "in.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("<qqii", room["map_xyz"].shape[0], room["q_xyz"].shape[0], int(P.variant), 0))
        f.write(np.concatenate([t0, q0]).astype("<f8").tobytes())
        f.write(np.ascontiguousarray(room["map_xyz"], "<f4").tobytes())
        f.write(np.ascontiguousarray(room["q_xyz"], "<f4").tobytes())
    out = subprocess.run([DEMO, str(path), "7"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    tok = out.stdout.split()
    pose_c = np.array([float(v) for v in tok[:7]])
    status_c, n_c = int(tok[8]), int(tok[10])
    m = L.ScanToMapMatcher(gpu_ctx, P)
    m.set_input_cloud(L.KIND_SURF, room["map_xyz"])
    m.set_queries(0, L.KIND_SURF, room["q_xyz"])
    m.pose_set(0, t0, q0)
    m.iterate(0, 7, L.MASK_SURF)
    tp, qp, st = m.pose_get(0)
    Q2, T2 = L.api.assoc_transform(tp, qp, P) if variant == "rot" else (qp, tp)
    n_p = m.find_corresponding_surf_features(0, Q2, T2)
    assert status_c == st == 0
    assert np.array_equal(pose_c, np.concatenate([tp, qp]))          # %.17g round-trips doubles exactly
    assert n_c == n_p > 1000


def test_cpp_host_window_seam_one_call_equals_single_calls(gpu_ctx, tmp_path):
    """examples/s2m_demo --window 3: the blocking seam of a 3-keyframe window from C++ — ONE lili_s2m_linearize_window per evaluation returns the bits of
    three lili_s2m_linearize calls (the program exits 3 otherwise) and prints both timings."""
    import json
    if not os.path.exists(DEMO):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-s"])
    room = synth.make_room(seed=41, n_query=2750, n_edge_query=10)
    P = L.make_params("rot")
    t, q = L.api.body_pose_from_lidar(room["t_true"], room["q_true"], P)
    path = tmp_path / "win.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("<qqii", room["map_xyz"].shape[0], room["q_xyz"].shape[0], int(P.variant), 0))
        f.write(np.concatenate([t, q]).astype("<f8").tobytes())
        f.write(np.ascontiguousarray(room["map_xyz"], "<f4").tobytes())
        f.write(np.ascontiguousarray(room["q_xyz"], "<f4").tobytes())
    out = subprocess.run([DEMO, str(path), "--window", "3", "100"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.stdout, out.stderr)
    r = json.loads(out.stdout.strip().splitlines()[-1])
    print(r)
    assert r["window_equals_single_calls_bit_for_bit"] is True and r["correspondences_slot0"] > 1500
    assert r["us_per_window_evaluation"] < r["us_per_evaluation_as_K_single_calls"]


def test_cpp_host_backend_keyframe_one_call_equals_separate_calls():
    """examples/backend_demo: the back end's per-keyframe work before ceres::Solve as ONE lili_backend_keyframe_prepare (the joining keyframe taken from its slot on the
    device) against the calls one by one through host buffers, from plain C++: counts and the window's Gram records bit for bit (the program exits 3 otherwise),
    ring popping included (width 5 < 14 keyframes)."""
    import json
    demo = os.path.join(ROOT, "examples", "backend_demo")
    if not os.path.exists(demo):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-s"])
    out = subprocess.run([demo, "14", "1500", "200", "5", "2"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout, out.stderr)
    r = json.loads(out.stdout.strip().splitlines()[-1])
    print(r)
    assert r["one_call_equals_separate_calls_bit_for_bit"] is True and r["correspondences_total"] > 20000
    assert r["ms_per_keyframe_one_call"] < r["ms_per_keyframe_separate_calls"]


def _write_rot_scan_file(path, n_az=400, n_map=200_000):
    """scan.bin of examples/rot_scan_demo: a synthetic 64-ring scan, surf + edge maps, the perturbed body pose as the prediction"""
    import struct
    import numpy as np
    import lili_om_amd as L
    from lili_om_amd import synth
    w = synth.make_workload(n_map=n_map, n_az=n_az, half_extent=(150.0, 150.0), verbose=False)
    raw = np.concatenate([w["scan_xyz"], np.full((w["scan_xyz"].shape[0], 1), 50.0, np.float32)], 1).astype("<f4")
    P = L.make_params("rot")
    tb, qb = L.api.body_pose_from_lidar(w["lidar_t"], w["lidar_q"], P)
    t0, q0 = synth.perturbed_pose(tb, qb, np.random.default_rng(synth.SEED_POSE), 0.2, 1.0)
    sm, em = np.ascontiguousarray(w["map_xyz"], "<f4"), np.ascontiguousarray(w["edge_map_xyz"], "<f4")
    with open(path, "wb") as f:
        f.write(struct.pack("<iii", raw.shape[0], sm.shape[0], em.shape[0]))
        f.write(np.asarray(t0, "<f8").tobytes()); f.write(np.asarray(q0, "<f8").tobytes())
        f.write(raw.tobytes()); f.write(sm.tobytes()); f.write(em.tobytes())
    return raw.shape[0]


def test_cpp_host_rot_scan_one_call_equals_separate_calls(tmp_path):
    """examples/rot_scan_demo (BASELINE configs[0] from plain C++): lili_frontend_frame_rot on the caller's maps — the matcher enqueued behind the extractor for guessed
    feature counts from the second repetition on — against lili_extract_rot -> lili_s2m_set_queries x 2 -> lili_s2m_pose_set -> lili_s2m_iterate -> lili_s2m_pose_get:
    the same pose to the last bit (the program exits 3 otherwise)."""
    demo = os.path.join(ROOT, "examples", "rot_scan_demo")
    if not os.path.exists(demo):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "-s"])
    path = str(tmp_path / "scan.bin")
    _write_rot_scan_file(path)
    out = subprocess.run([demo, path, "6", "2"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, (out.stdout, out.stderr)
    print(out.stdout)
    lines = out.stdout.strip().splitlines()
    assert "poses_equal_bit_for_bit 1" in lines[2]
    one, sep = lines[0].split(), lines[1].split()
    assert one[2:9] == sep[2:9] and int(one[10]) == 0 and int(one[14]) > 300

