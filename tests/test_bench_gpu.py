"""bench.py as the driver runs it: the JSON contract, the parity self-checks that make a run FAIL (exit status 3) when a GPU-vs-oracle
pose delta exceeds the north star's 1e-4 m / 1e-4 rad, and the multi-rank path (ranks sharing GPU 0: the pool has one GPU per box) with the replica mode
and the lili_p2p mailbox round trip."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _last_json(text):
    for line in reversed(text.strip().splitlines()):
        if line.startswith("{"):
            return json.loads(line)
    raise AssertionError("no JSON line in: " + text[-2000:])


def test_bench_default_run_passes_its_own_parity_checks():
    """python bench.py --gpus 1 --steps 20 --warmup 5 — the driver's command: exit status 0, every parity field inside the tolerance."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5"], cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["metric"].startswith("scan-to-map iterations/s") and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["value"] > 1000
    assert d["parity_failures"] == []
    pd = d["pose_delta_vs_cpu"]
    assert pd["pass"] and pd["dt_m"] <= 1e-4 and pd["dang_rad"] <= 1e-4 and pd["gn_status"] == 0
    assert d["final_pose"]["gn_status"] == 0
    for key in ("0", "1", "4"):
        c = d["extras"]["configs"][key]
        assert "error" not in c, (key, c)
        assert c["parity"]["pass"] and c["parity"]["dt_m"] <= 1e-4 and c["parity"]["dang_rad"] <= 1e-4, (key, c["parity"])
    assert d["extras"]["configs"]["1"]["frames"] == 100                       # BASELINE configs[1]: the first 100 frames
    rl = d["roofline"]
    assert rl["bound"] == "hbm" and 0 < rl["frac"] < 1 and abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-5
    cb = d["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] in ("port", "reference")
    seam = d["extras"]["configs"]["4"]["cpp_seam"]
    assert seam.get("window_equals_single_calls_bit_for_bit") is True, seam


@pytest.mark.parametrize("world", [2, 4])
def test_bench_ranks_sharing_one_gpu(world):
    """torch.distributed.run --nproc-per-node N bench.py --gpus N (gloo control plane, all ranks on GPU 0): the strong split through lili_p2p with its
    mailbox round trip, the replica mode without any collective, both bit-identical-pose checks, exit status 0."""
    env = dict(os.environ, LILI_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "20", "--warmup", "10", "--n-map", "500000", "--n-az", "500", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == world and d["scaling"] == "strong" and d["parity_failures"] == []
    mg = d["multi_gpu_check"]
    assert mg["ranks"] == world and mg["final_pose_bit_identical_on_all_ranks"] and mg["max_gn_status"] == 0
    assert mg["p2p_mailbox_round_trip"] is True and d["config"]["collectives"].startswith("lili_p2p")
    assert mg["replicas_final_pose_bit_identical_on_all_ranks"] and mg["replicas_max_gn_status"] == 0
    ex = d["extras"]
    assert ex["replicas_iterations_per_s"] > 0 and ex["weak_scaling_iterations_per_s"] > 0
    # the slot-per-rank window (one keyframe per rank at full size, one gather per iteration): all ranks end with the same bits for every slot
    assert ex.get("window_gather_error") is None, ex.get("window_gather_error")
    assert ex["window_gather_slot_iterations_per_s"] > 0 and ex["window_gather_poses_bit_identical_on_all_ranks"] is True and ex["window_gather_max_gn_status"] == 0
    assert ex["window_gather_dt_truth_m"] < 0.05
