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
