#!/usr/bin/env python
"""Closes the third-party pin where the real libraries exist (SURVEY §8c mitigation 3).
    make -C oracle/refshim REAL_DEPS=1                      # the reference's sources against REAL Eigen / PCL / Ceres -> oracle/_ref_real/
    LILI_REF_DIR=oracle/_ref_real python tools/diff_ref_golden.py
re-runs tests/golden/make_ref_golden.py's generators on that build (into a temporary directory) and compares every array with the
committed fixture (produced with the stand-in headers): bit-equal arrays, and for the others the number of differing elements and the
largest absolute / relative difference.  Differences are what Eigen's / PCL's / FLANN's / Ceres' internals contribute beyond the oracle's
restatement of them (kNN tie order, in-voxel summation order, QR / eigen-solver rounding).  Cannot run in the graft image (no such libraries)."""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
G = os.path.join(ROOT, "tests", "golden")
spec = importlib.util.spec_from_file_location("make_ref_golden", os.path.join(G, "make_ref_golden.py"))
M = importlib.util.module_from_spec(spec)
spec.loader.exec_module(M)

RUNS = {"ref_rot": "run_rot", "ref_livox": "run_livox", "ref_factors": "run_factors", "ref_backend": "run_backend", "ref_format": "run_format",
        "ref_marg": "run_marg", "ref_localmap": "run_localmap"}       # ref_frontend needs libref_lo (stand-in ceres::Solve hook): not in the real build


def main():
    print("reference build under test:", os.environ.get("LILI_REF_DIR", "oracle/_ref (stand-in headers: expect all-equal)"))
    worst = 0.0
    for name, fn in RUNS.items():
        try:
            new = getattr(M, fn)()
        except Exception as e:      # noqa: BLE001
            print(f"{name}: generator failed: {e!r}")
            continue
        old = np.load(os.path.join(G, name + ".npz"))
        for k in sorted(old.files):
            a, b = np.asarray(old[k]), np.asarray(new[k])
            if a.shape != b.shape:
                print(f"{name}:{k}: SHAPE {a.shape} -> {b.shape}")
                worst = float("inf")
                continue
            if a.tobytes() == b.tobytes():
                continue
            d = np.abs(a.astype(np.float64) - b.astype(np.float64))
            rel = d / np.maximum(np.abs(a.astype(np.float64)), 1e-300)
            print(f"{name}:{k}: {int((a != b).sum())} of {a.size} elements differ, max abs {d.max():.3e}, max rel {rel.max():.3e}")
            worst = max(worst, float(d.max()))
    print("largest absolute difference over all fixtures:", worst)


if __name__ == "__main__":
    main()
