"""ON THE GPU: the correspondence records lili_s2m_associate produces for the scene of tests/test_reference_gpu.py::test_gpu_linearize_vs_reference_functors
(both variants) -> gpurun_out/functor_inputs.npz.  Back in the build container, tests/golden/make_ref_golden.py::make_functor_fixture evaluates the REFERENCE's
functors (oracle/_ref/libref_factors.so) on them and writes tests/golden/ref_functor_rows.npz.
    python tools/dump_functor_inputs.py gpurun_out/functor_inputs.npz"""
import sys
import numpy as np
sys.path.insert(0, ".")
import lili_om_amd as L          # noqa: E402
from lili_om_amd import synth   # noqa: E402

out = {}
ctx = L.Context(0)
for variant in ("livox", "rot"):
    room = synth.make_room(seed=12, n_query=1500, n_edge_query=200)
    P = L.make_params(variant)
    m = L.ScanToMapMatcher(ctx, P)
    m.set_input_cloud(L.KIND_SURF, np.c_[room["map_xyz"], room["map_refl"]] if variant == "livox" else room["map_xyz"])
    m.set_input_cloud(L.KIND_EDGE, room["edge_map_xyz"])
    m.set_queries(0, L.KIND_SURF, np.c_[room["q_xyz"], room["q_refl"]] if variant == "livox" else room["q_xyz"])
    m.set_queries(0, L.KIND_EDGE, room["eq_xyz"])
    tb, qb = L.api.body_pose_from_lidar(room["t_true"], room["q_true"], P)
    t0, q0 = synth.perturbed_pose(tb, qb, np.random.default_rng(3), 0.05, 0.5)
    Q2, T2 = L.api.assoc_transform(t0, q0, P)
    ns = m.find_corresponding_surf_features(0, Q2, T2)
    ne = m.find_corresponding_corner_features(0, Q2, T2)
    rs, re_ = m.surf_records(0, ns), m.edge_records(0, ne)
    for k in ("cp", "n", "d", "score"):
        out[f"{variant}_s_{k}"] = rs[k]
    for k in ("cp", "a", "b", "s"):
        out[f"{variant}_e_{k}"] = re_[k]
    out[f"{variant}_qlb"], out[f"{variant}_tlb"] = np.array(list(P.q_lb)), np.array(list(P.t_lb))
    out[f"{variant}_t0"], out[f"{variant}_q0"] = np.asarray(t0, np.float64), np.asarray(q0, np.float64)
    out[f"{variant}_ss"] = P.scale_surf_num / ns if P.scale_surf_num else 1.0
    out[f"{variant}_se"] = P.scale_edge_num / ne if P.scale_edge_num else 1.0
    print(variant, ns, ne)
np.savez_compressed(sys.argv[1], **out)
ctx.close()
