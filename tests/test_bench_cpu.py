"""bench.py / bench_configs.py without a GPU: what turns a parity field into exit status 3, and that no secondary measurement shares the headline's
lili context (pose slots live in the context; three helpers on the shared context overwrote slot 1, the parity check started from it and the
driver's record read 1.47 m / 1.99 rad with exit status 0)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_exit_status_follows_the_parity_fields():
    """The helper that decides: a config whose pose delta is over the tolerance, or that did not run, or whose solver status is not 0, is a failure."""
    sys.path.insert(0, ROOT)
    import bench_configs as BC
    ok = {"gn_status": 0, "parity": {"dt_m": 1e-9, "dang_rad": 1e-9, "pass": True}}
    assert BC.parity_failures(ok) == []
    assert BC.parity_failures({"gn_status": 0, "parity": {"dt_m": 1.467, "dang_rad": 1.989, "pass": False}})          # r03's driver record would have failed
    assert BC.parity_failures({"gn_status": 1, "parity": ok["parity"]})
    assert BC.parity_failures({"error": "boom"})
    assert BC._parity(2e-4, 0.0)["pass"] is False and BC._parity(0.0, 2e-4)["pass"] is False and BC._parity(5e-5, 5e-5)["pass"] is True


def test_secondary_measurements_never_get_the_headline_context():
    src = open(os.path.join(ROOT, "bench.py")).read()
    extras = src[src.index("def own_ctx():"):src.index('if extras:\n            out["extras"] = extras')]
    # inside the extras block every helper call and every matcher is built on a context of its own (cx), never on `ctx` / `m`
    assert not re.search(r"\bBC\.\w+\(L, ctx\b", extras) and not re.search(r"secondary_stages\(L, ctx\b", extras)
    assert not re.search(r"ScanToMapMatcher\(ctx\b", extras) and "m.pose_set(" not in extras and "m.pose_copy(" not in extras
    assert extras.count("cx = own_ctx()") >= 6 and extras.count("cx.close()") >= 6
    # the parity registration starts from the explicit pose, right after the timed region, and a failure is fatal
    assert "m.pose_set(0, t0, q0)\n        m.iterate(0, ips, L.MASK_SURF)" in src and "sys.exit(exit_code)" in src and "exit_code = 3 if failures else 0" in src
