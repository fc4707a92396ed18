"""`contactimplicitmpc/jl_amd/csrc/plant_model.h` is host- and device-compilable: built here with g++ (tests/native/
plant_model_check.cpp) and compared with the numpy plant of oracle/plant.py - residual and dual-number Jacobian of both
planar-chain models and of hopper_2D."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from oracle import plant as pl

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_plant_model_header_matches_the_numpy_plant(tmp_path):
    exe = str(tmp_path / "plant_check")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(HERE, "native", "plant_model_check.cpp")])
    for mid, P in ((0, pl.QuadrupedPlant()), (1, pl.FlamingoPlant()), (2, pl.HopperPlant()), (3, pl.CentroidalPlant(True)),
                   (4, pl.CentroidalPlant(False)), (5, pl.ParticlePlant())):
        d = P.dims
        rng = np.random.default_rng(mid)
        z, th, kappa = rng.uniform(0.1, 1.0, d.nz), rng.uniform(0.1, 1.0, d.nth), 1e-3
        inp = f"{mid} {kappa} " + " ".join(repr(float(v)) for v in z) + " " + " ".join(repr(float(v)) for v in th)
        out = subprocess.run([exe], input=inp, capture_output=True, text=True, check=True).stdout.split("\n")
        assert [int(v) for v in out[0].split()] == [d.nz, d.nth]
        r = np.array(out[1].split(), dtype=float)
        J = np.array(out[2].split(), dtype=float).reshape(d.nz, d.nz)
        np.testing.assert_allclose(r, P.residual(z, th, kappa), rtol=0, atol=1e-12)
        np.testing.assert_allclose(J, P.jacobian_z(z, th), rtol=0, atol=1e-12 * np.abs(J).max())
