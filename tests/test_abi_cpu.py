"""CPU tests of the boundary: the C-ABI library loads, exports every symbol include/lili_hip.h declares,
refuses to run without a GPU (no fallback), and its host-side helpers agree with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import lili_om_amd as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    txt = open(os.path.join(ROOT, "include", "lili_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lili_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = L.load_library()
    names = _header_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/lili_hip.h but not exported by liblili_hip.so"
    # and the binding binds exactly the declared set
    assert sorted(L.api.exported_symbols()) == names
    assert lib.lili_abi_version() == 1


def test_struct_layouts_match_header(tmp_path):
    """sizeof/offsetof from the C header (compiled with gcc as plain C) vs the ctypes mirrors."""
    import subprocess
    src = tmp_path / "lay.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "lili_hip.h"\nint main(void){'
                   'printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(lili_cloud), sizeof(lili_s2m_params),'
                   'offsetof(lili_s2m_params, q_lb), offsetof(lili_s2m_params, scale_surf_num),'
                   'offsetof(lili_cloud, aux_offset), offsetof(lili_s2m_params, kd_max_radius));return 0;}')
    exe = tmp_path / "lay"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    vals = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    P, Cl = L.api.S2MParams, L.api.Cloud
    assert vals == [C.sizeof(Cl), C.sizeof(P), P.q_lb.offset, P.scale_surf_num.offset, Cl.aux_offset.offset,
                    P.kd_max_radius.offset]


def test_every_struct_mirror_matches_the_header(tmp_path):
    """size and the offset of EVERY field of every structure the ctypes binding mirrors, against the C header compiled as plain C: a mirror that drifts from
    its struct corrupts memory silently (lili_frontend_result, lili_lm_summary and the others joined the ABI after the first layout test was written)."""
    import subprocess
    pairs = [("lili_cloud", L.api.Cloud), ("lili_feature_out", L.api.FeatureOut), ("lili_rot_params", L.api.RotParams), ("lili_livox_params", L.api.LivoxParams),
             ("lili_frontend_options", L.api.FrontendOptions), ("lili_frontend_result", L.api.FrontendResult), ("lili_lm_options", L.api.LmOptions),
             ("lili_lm_iteration", L.api.LmIteration), ("lili_lm_summary", L.api.LmSummary), ("lili_s2m_params", L.api.S2MParams), ("lili_imu_state", L.api.ImuState)]
    lines, expect = [], []
    for cname, T in pairs:
        lines.append(f'printf("%zu\\n", sizeof({cname}));')
        expect.append(C.sizeof(T))
        for fname, _ in T._fields_:
            lines.append(f'printf("%zu\\n", offsetof({cname}, {fname}));')
            expect.append(getattr(T, fname).offset)
    src = tmp_path / "lay_all.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "lili_hip.h"\nint main(void){' + "".join(lines) + "return 0;}")
    exe = tmp_path / "lay_all"
    subprocess.check_call(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert got == expect, [(i, a, b) for i, (a, b) in enumerate(zip(got, expect)) if a != b]


def test_every_bound_function_has_a_declared_signature():
    """every lili_* entry point the Python package calls has argtypes / restype in api._SIGS (ctypes' default would pass a 64-bit pointer as a C int), and every
    signature names a function the header declares"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    used = set()
    for f in os.listdir(os.path.join(root, "lili_om_amd")):
        if f.endswith(".py"):
            used |= set(re.findall(r"lib\.(lili_[a-z0-9_]+)", open(os.path.join(root, "lili_om_amd", f)).read()))
    sigs = set(L.api._SIGS.keys())
    assert used <= sigs, sorted(used - sigs)
    hdr = open(os.path.join(root, "include", "lili_hip.h")).read()
    declared = set(re.findall(r"\b(lili_[a-z0-9_]+)\s*\(", hdr))
    assert sigs <= declared, sorted(sigs - declared)


def test_documented_options_are_the_implemented_ones():
    """include/lili_hip.h lists every name lili_set_option accepts and no name it does not (round 4's header still advertised knobs whose experiments had been
    closed and removed): names compared between the comment block above the declaration and the strcmp chain in lili_api.hip (names behind an #ifdef are build
    variants, not options of the shipped library)."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "lili_om_amd", "csrc", "lili_api.hip")).read()
    body = src[src.index("int lili_set_option("):]
    body = body[:body.index("\n}\n")]
    body = re.sub(r"#ifdef.*?#endif", "", body, flags=re.S)
    implemented = set(re.findall(r'strcmp\(name, "([a-z0-9_]+)"\)', body))
    hdr = open(os.path.join(root, "include", "lili_hip.h")).read()
    doc = hdr[:hdr.index("int lili_set_option(")]
    doc = doc[doc.rindex("/* Tuning knobs"):]
    removed = doc[doc.index("(the closed experiments"):doc.index("an unknown name")]
    documented = set(re.findall(r'"([a-z0-9_]+)"', doc)) - set(re.findall(r'"([a-z0-9_]+)"', removed))
    assert implemented == documented, (sorted(implemented - documented), sorted(documented - implemented))
    assert len(implemented) >= 20


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(L.LiliError):
        L.Context(0)


def test_product_does_not_import_oracle():
    import sys
    for name, mod in list(sys.modules.items()):
        if name.startswith("lili_om_amd"):
            src = getattr(mod, "__file__", None)
            if src and src.endswith(".py"):
                assert "oracle" not in re.sub(r'""".*?"""', "", open(src).read(), flags=re.S).replace("# ", ""), name
    for f in os.listdir(os.path.join(ROOT, "lili_om_amd", "csrc")):
        if f.endswith((".hip", ".h")):
            assert "oracle/" not in open(os.path.join(ROOT, "lili_om_amd", "csrc", f)).read()


def test_host_gn_step_matches_oracle(oracle):
    rng = np.random.default_rng(0)
    for _ in range(20):
        J = rng.normal(size=(50, 8))
        G = J.T @ J
        t = rng.normal(size=3)
        q = rng.normal(size=4); q /= np.linalg.norm(q)
        s1, t1, q1, d1 = L.api.gn_step_host(G, t, q)
        s2, t2, q2, d2 = oracle.gn_step(G, t, q)
        assert s1 == s2 == 0
        assert np.allclose(t1, t2, rtol=1e-12, atol=1e-12) and np.allclose(q1, q2, rtol=1e-12, atol=1e-12)
        assert np.allclose(d1, d2, rtol=1e-10, atol=1e-12)
    s1, *_ = L.api.gn_step_host(np.zeros((8, 8)), [0, 0, 0], [1, 0, 0, 0])
    assert s1 == 1


def test_assoc_transform_matches_reference_algebra(oracle):
    P = L.make_params("rot")
    t = np.array([1.0, 2.0, 3.0]); q = np.array([0.9, 0.1, -0.3, 0.2]); q /= np.linalg.norm(q)
    Q2, T2 = L.api.assoc_transform(t, q, P)
    qlb = np.array(list(P.q_lb)); tlb = np.array(list(P.t_lb))
    # Q2 * q_lb == Q (up to |q_lb|^2 handled by Eigen's inverse), T2 + Q2 * t_lb == T
    assert np.allclose(oracle.qrot(Q2, tlb) + T2, t)
    back = np.array([Q2[0] * qlb[0] - Q2[1:] @ qlb[1:], *(Q2[0] * qlb[1:] + qlb[0] * Q2[1:] + np.cross(Q2[1:], qlb[1:]))])
    assert np.allclose(back, q)


def test_gram_to_factor_reproduces_normal_equations():
    """The ceres adapter's square-root block has the same J^T J, J^T r and cost as the residuals it replaces."""
    rng = np.random.default_rng(1)
    for n in (3, 7, 400):                      # n < 8 gives a rank-deficient Gram
        Jr = rng.normal(size=(n, 8)) * rng.uniform(0.1, 10, size=8)
        s = Jr[:, 7] ** 2
        rho1 = 1.0 / (1.0 + s)                 # Cauchy: corrected rows = sqrt(rho') [J r]
        Jc = Jr * np.sqrt(rho1)[:, None]
        G = Jc.T @ Jc
        cost = 0.5 * np.log1p(s).sum()
        res, jac = L.api.gram_to_factor(G, cost)
        scale = np.abs(G).max()
        assert np.abs(jac.T @ jac - G[:7, :7]).max() <= 1e-12 * scale
        assert np.abs(jac.T @ res - G[:7, 7]).max() <= 1e-12 * scale
        assert abs(0.5 * res @ res - cost) <= 1e-12 * max(cost, 1.0)
        assert not jac[8].any()


def test_keyframe_map_pose_is_the_reference_composition():
    """lili_om_amd.api.keyframe_map_pose (host-side pose algebra of the local-map binding) vs the oracle's restatement of
    L/src/BackendFusion.cpp:1425-1426, itself pinned to the reference text by tests/test_reference_cpu.py: bit for bit."""
    import numpy as np
    from lili_om_amd import api
    from oracle import oracle
    rng = np.random.default_rng(8)
    for _ in range(50):
        q_po, q_bl = rng.normal(size=4), rng.normal(size=4)
        q_po /= np.linalg.norm(q_po)
        t_po, t_bl = rng.uniform(-50, 50, 3), rng.uniform(-0.5, 0.5, 3)
        asm = oracle.LocalMapAssembly(3, 0.4, 0.4, q_bl, t_bl)
        q_ref, t_ref = asm.lidar_pose(np.concatenate([q_po, t_po]))
        t, q = api.keyframe_map_pose(t_po, q_po, t_bl, q_bl)
        assert q.tobytes() == q_ref.tobytes() and t.tobytes() == t_ref.tobytes()
