"""CPU-only checks of the product boundary: the C-ABI library loads and exports every symbol
include/cimpc.h declares, argument validation fails loudly, and there is no CPU fallback."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "cimpc.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cimpc_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from contactimplicitmpc.jl_amd import _lib
    lib = _lib.load()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/cimpc.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in _lib.SIGNATURES"
    assert lib.cimpc_version() >= 100


def test_default_options_match_reference_defaults():
    from contactimplicitmpc.jl_amd import _lib
    import ctypes as C
    lib = _lib.load()
    ip = _lib.IpOpts(); nt = _lib.NewtonOpts()
    lib.cimpc_default_ip_opts(C.byref(ip)); lib.cimpc_default_newton_opts(C.byref(nt))
    # policy.jl:48-61
    assert (ip.r_tol, ip.kappa_tol, ip.undercut, ip.gamma_reg) == (1e-8, 2e-4, 5.0, 0.1)
    assert (nt.r_tol, nt.max_iter, nt.beta_init) == (3e-4, 5, 1e-5)


def test_no_cpu_fallback_and_loud_errors():
    import torch
    from contactimplicitmpc.jl_amd import CIMPCSolver, CimpcError
    if torch.cuda.is_available():
        pytest.skip("GPU present: the no-device path cannot be exercised")
    with pytest.raises(CimpcError) as e:
        CIMPCSolver(11, 8, 2, 4, 8, H_ref=4, H=2, B=1)
    assert "no HIP device" in str(e.value) or "NO_DEVICE" in str(e.value) or "-2" in str(e.value)


def test_product_does_not_import_oracle():
    """The shipped path may not import / link anything under oracle/."""
    pkg = os.path.join(ROOT, "contactimplicitmpc")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")) or f == "Makefile":
                src = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle" not in src.replace("test oracle", ""), f"{f} mentions oracle/"
    src = open(os.path.join(ROOT, "include", "cimpc.h")).read()
    assert "oracle" not in src


def test_struct_layouts_match_the_header(tmp_path):
    """The ctypes mirrors (_lib.py) must have the size and field offsets a C compiler gives the structs of
    include/cimpc.h - an ABI drift between the header and the Python host would otherwise corrupt options
    silently."""
    import ctypes as C
    import subprocess
    from contactimplicitmpc.jl_amd import _lib
    structs = {"cimpc_dims": _lib.Dims, "cimpc_ip_opts": _lib.IpOpts, "cimpc_newton_opts": _lib.NewtonOpts,
               "cimpc_stats": _lib.Stats, "cimpc_profile": _lib.Profile}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "cimpc.h"', 'int main(void) {']
    for cname, py in structs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in py._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    for cname, py in structs.items():
        assert int(got[cname]) == C.sizeof(py), cname
        for fname, _ in py._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(py, fname).offset, f"{cname}.{fname}"


def test_reference_objective_test():
    """test/controller/objective.jl:1-12: a TrackingObjective over H steps holds H weight matrices of every kind."""
    from contactimplicitmpc.jl_amd.trajectory import Dims, Objective, QUADRUPED
    H = 10
    obj = Objective.tracking(Dims(**QUADRUPED), H)
    assert len(obj.q) == H and len(obj.u) == H and len(obj.gamma) == H and len(obj.b) == H
    assert obj.q.shape == (H, 11, 11) and obj.u.shape == (H, 8, 8) and obj.gamma.shape == (H, 4, 4) and obj.b.shape == (H, 8, 8)


def test_oracle_containers_are_its_own():
    """the checker must not share its small containers with the product (a bug in a shared update_theta / copy_traj /
    layout would be common-mode and invisible to the parity tests).  oracle/ imports nothing from the product except the seeded
    INPUT generator; the two copies of the index layouts agree with each other AND with numbers derived from the reference
    (src/simulation/index.jl:289-327, 371-384; quadruped: nz = 43, ntheta = 34, test/controller/newton.jl sizes)."""
    import numpy as np
    import oracle.dims as od
    import oracle.newton as on
    from contactimplicitmpc.jl_amd import trajectory as pt
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in os.listdir(os.path.join(root, "oracle")):
        if f.endswith(".py") and f != "synth.py":
            src = open(os.path.join(root, "oracle", f)).read()
            assert not re.search(r"^\s*(from|import)\s+contactimplicitmpc", src, flags=re.M), f
    assert od.Dims is not pt.Dims and on.Traj is not pt.Traj and on.copy_traj is not pt.copy_traj and on.Objective is not pt.Objective
    names = ("PUSHBOT", "HOPPER_2D", "QUADRUPED", "CENTROIDAL", "FLAMINGO", "HOPPER_3D", "WALLEDCARTPOLE", "PARTICLE", "PARTICLE_2D", "CENTROIDAL_WALL")
    for nm in names:
        assert getattr(od, nm) == getattr(pt, nm), nm
        for mode in (0, 1):
            a, b = od.Dims(**getattr(od, nm), mode=mode), pt.Dims(**getattr(pt, nm), mode=mode)
            for k in ("nx", "ny", "nz", "nth", "nd", "nr"):
                assert getattr(a, k) == getattr(b, k), (nm, k)
            for k in ("ix", "iy1", "iy2", "ig1", "ib1", "iq0", "iq1", "iu1", "iw1"):
                np.testing.assert_array_equal(getattr(a, k), getattr(b, k))
    # reference-derived numbers (quadruped, flat_2D_lc): nz = nq + 4 nc + 2 nb = 43, ntheta = 2 nq + nu + nw + 2 = 34;
    # z = [q2 (11); gamma1 (4); b1 (8); psi1 (4); s1 (4); eta1 (8); s2 (4)], theta = [q0; q1; u1 (8); w1 (2); mu; h]
    q = od.Dims(**od.QUADRUPED)
    assert (q.nz, q.nth, q.ny, q.nd, q.nr) == (43, 34, 16, 11, 19)
    np.testing.assert_array_equal(q.iy1, np.arange(11, 27)); np.testing.assert_array_equal(q.iy2, np.arange(27, 43))
    np.testing.assert_array_equal(q.ig1, np.arange(11, 15)); np.testing.assert_array_equal(q.ib1, np.arange(15, 23))
    np.testing.assert_array_equal(q.iu1, np.arange(22, 30)); np.testing.assert_array_equal(q.iw1, np.arange(30, 32))
    qf = od.Dims(**od.QUADRUPED, mode=1)
    assert (qf.nd, qf.nr) == (23, 31)                                  # :configurationforce, newton_residual.jl:18-54
    # update_theta! (trajectory.jl:67-82) and copy_traj! (newton.jl:105-128): same result from both copies, on known values
    H = 3
    rng = np.random.default_rng(0)
    mk = lambda T: T(q=np.arange((H + 2) * 11, dtype=float).reshape(H + 2, 11), u=100 + np.arange(H * 8, dtype=float).reshape(H, 8),
                     w=200 + np.arange(H * 2, dtype=float).reshape(H, 2), gamma=np.zeros((H, 4)), b=np.zeros((H, 8)), theta=-np.ones((H, 34)))
    A, B = mk(on.Traj), mk(pt.Traj)
    A.update_theta(q); B.update_theta(pt.Dims(**pt.QUADRUPED))
    np.testing.assert_array_equal(A.theta, B.theta)
    np.testing.assert_array_equal(A.theta[1, :11], A.q[1]); np.testing.assert_array_equal(A.theta[1, 11:22], A.q[2])
    np.testing.assert_array_equal(A.theta[1, 22:30], A.u[1]); np.testing.assert_array_equal(A.theta[1, 30:32], A.w[1])
    assert (A.theta[:, 32:] == -1).all()                                # mu, h untouched
    C = mk(on.Traj); C.q[:] = 0; C.theta[:] = 0
    on.copy_traj(C, A, H)
    np.testing.assert_array_equal(C.q, A.q); np.testing.assert_array_equal(C.theta, A.theta)


def test_every_environment_override_is_documented():
    """The library reads a dozen environment overrides, once, in cimpc_create (cimpc_host.cpp: Knobs::read_environment); each one is an
    A/B handle for the scripts under scripts/ and must be named in README.md - and nothing else in the kernels' sources may read the
    environment (untested configurations)."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    names = set()
    for f in glob.glob(os.path.join(root, "contactimplicitmpc", "jl_amd", "csrc", "*")):
        if not f.endswith((".cpp", ".h", ".hip")):
            continue
        src = open(f).read()
        found = set(re.findall(r'(?:env_int|getenv)\("([A-Z0-9_]+)"', src))
        if found:
            assert f.endswith("cimpc_host.cpp"), (f, found)          # only the host's create path reads the environment
        names |= found
    readme = open(os.path.join(root, "README.md")).read()
    # (round 5: + CIMPC_KKT_TWISTED, CIMPC_KKT_TW_NB - both driven by tests/test_gpu_round5.py; round 6: + CIMPC_KKT_DUO, CIMPC_LAZY_DZ -
    #  the A/B switches of the duo KKT kernel and the lazy sensitivity commit, both driven by tests/test_gpu_round6.py - and CIMPC_KKT_DUO_HINT)
    assert 8 <= len(names) <= 22, sorted(names)      # (+ CIMPC_ASYNC_KKT_TW, driven by tests/test_gpu_round6.py)
    for n in sorted(names):
        tail = n[len("CIMPC_"):]
        assert n in readme or ("_" + tail.split("_", 1)[-1]) in readme, n + " is not documented in README.md"
