"""Time per evaluation of the device LM loop (lili_s2m_solve_lm) next to the one-launch-per-iteration Gauss-Newton loop (lili_s2m_iterate_inner).
    python tools/lm_time.py [out.json]"""
import json
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import lili_om_amd as L          # noqa: E402
from lili_om_amd import synth   # noqa: E402

w = synth.make_workload(n_map=5_000_000, half_extent=(460.0, 380.0))
order = np.argsort(w["scan_ring"], kind="stable")
scan = np.ascontiguousarray(w["scan_xyz"][order])
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
ctx = L.Context(0, stream=s.cuda_stream)
P = L.make_params("rot")
m = L.ScanToMapMatcher(ctx, P)
m.map_focus(w["lidar_t"], float(np.linalg.norm(w["scan_xyz"], axis=1).max()) + 3.0)
m.set_input_cloud(L.KIND_SURF, w["map_xyz"])
tb, qb = L.api.body_pose_from_lidar(w["lidar_t"], w["lidar_q"], P)
t0, q0 = synth.perturbed_pose(tb, qb, np.random.default_rng(synth.SEED_POSE), 0.05, 0.5)
out = []
for n, slots in ((3000, 1), (3000, 3), (20000, 1), (200000, 1)):
    q = scan[:n] if n >= 25000 else np.ascontiguousarray(scan[:: max(1, scan.shape[0] // n)][:n])
    for k in range(slots):
        m.set_queries(k, L.KIND_SURF, q)
    m.pose_set(7, t0, q0)
    def reset():
        for k in range(slots):
            m.pose_copy(k, 7)
    reset()
    for k in range(slots):
        m.associate_dev(k, L.MASK_SURF)
    summ = m.solve_lm(0, L.MASK_SURF) if slots == 1 else m.solve_lm_window(list(range(slots)), L.MASK_SURF)[0]
    evals = len(summ["log"]) + 1
    reps = 30
    def timed(fn):
        reset(); fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            reset(); fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / reps
    t_reset = timed(lambda: None)
    if slots == 1:
        t_lm = timed(lambda: m.solve_lm(0, L.MASK_SURF, want_summary=False))
        t_gn = timed(lambda: m.iterate_inner(0, evals, L.MASK_SURF, want_cost=True))
    else:
        t_lm = timed(lambda: m.solve_lm_window(list(range(slots)), L.MASK_SURF, want_summary=False))
        t_gn = timed(lambda: [m.iterate_inner(k, evals, L.MASK_SURF, want_cost=True) for k in range(slots)])
    row = dict(n=n, slots=slots, evaluations=evals, iterations=summ["iterations"], successful=summ["successful_steps"], termination=summ["termination"],
               lm_us_per_solve=round(t_lm - t_reset, 2), lm_us_per_evaluation=round((t_lm - t_reset) / evals, 3),
               gn_launch_per_iteration_us=round((t_gn - t_reset) / evals / slots, 3), final_cost=summ["final_cost"])
    out.append(row)
    print(json.dumps(row), flush=True)
# The MARGINAL cost of an evaluation: the figures above divide a whole solve — launch, set-up of the first point, summary — by its 4-5 evaluations.
# With all three tolerances at zero the loop runs to max_iterations; the slope of the solve time over the number of evaluations is what one more evaluation costs.
for n, slots in ((3000, 1), (3000, 3)):
    q = np.ascontiguousarray(scan[:: max(1, scan.shape[0] // n)][:n])
    for k in range(slots):
        m.set_queries(k, L.KIND_SURF, q)
    m.pose_set(7, t0, q0)
    for k in range(slots):
        m.pose_copy(k, 7); m.associate_dev(k, L.MASK_SURF)
    pts = []
    for it in (4, 8, 16, 32):
        opt = m.lm_options(max_iterations=it, function_tolerance=0.0, gradient_tolerance=0.0, parameter_tolerance=0.0)
        def reset():
            for k in range(slots):
                m.pose_copy(k, 7)
        reset()
        summ = m.solve_lm(0, L.MASK_SURF, options=opt) if slots == 1 else m.solve_lm_window(list(range(slots)), L.MASK_SURF, options=opt)[0]
        evals = len(summ["log"]) + 1
        reps = 20
        reset(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            reset()
            if slots == 1:
                m.solve_lm(0, L.MASK_SURF, options=opt, want_summary=False)
            else:
                m.solve_lm_window(list(range(slots)), L.MASK_SURF, options=opt, want_summary=False)
        e1.record(); torch.cuda.synchronize()
        pts.append((evals, e0.elapsed_time(e1) * 1e3 / reps, summ["termination"]))
    (ea, ta, _), (eb, tb_, _) = pts[0], pts[-1]
    row = dict(n=n, slots=slots, marginal_us_per_evaluation=round((tb_ - ta) / max(eb - ea, 1), 3), fixed_us_per_solve=round(ta - ea * (tb_ - ta) / max(eb - ea, 1), 2),
               points=[dict(evaluations=e, us_per_solve=round(t, 2), termination=term) for e, t, term in pts],
               note="all three tolerances 0: the solve runs to max_iterations; slope and intercept of the solve time over the evaluations (incl. the pose resets of the loop)")
    out.append(row)
    print(json.dumps(row), flush=True)
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
ctx.close()
