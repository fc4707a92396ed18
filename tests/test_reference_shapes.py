"""The reference's stored Jacobian buffers (src/simulation/<model>/flat/jacobians.jld2, written by
generate_simulation.jl:20-23) pin the SHAPES of rz (nz x nz) and rθ (nz x nθ) of every model; the fixture
tests/golden/jacobian_shapes.json was produced from them by tests/golden/make_jacobian_shapes.py.  (Their payload is
uninitialised `similar()` memory - no sparsity pattern to pin, see the generator's docstring.)"""
import json
import os

import pytest

from contactimplicitmpc.jl_amd.trajectory import Dims, CENTROIDAL, HOPPER_2D, PUSHBOT, QUADRUPED

HERE = os.path.dirname(os.path.abspath(__file__))
SHAPES = json.load(open(os.path.join(HERE, "golden", "jacobian_shapes.json")))
FLAMINGO = dict(nq=9, nu=6, nw=2, nc=4, nb=8)          # src/dynamics/flamingo/model.jl


@pytest.mark.parametrize("name,dims", [("quadruped", QUADRUPED), ("hopper_2D", HOPPER_2D), ("pushbot", PUSHBOT),
                                        ("flamingo", FLAMINGO), ("centroidal_quadruped", CENTROIDAL)])
def test_index_layout_sizes_match_the_reference_files(name, dims):
    d = Dims(**dims)
    s = SHAPES[name]
    assert s["rz_shape"] == [d.nz, d.nz]
    assert s["rth_shape_as_stored"] == [d.nth, d.nz]          # Julia (nz, nθ) column-major = numpy (nθ, nz)


@pytest.mark.parametrize("name", ["quadruped", "hopper_2D", "flamingo", "centroidal_quadruped"])
def test_model_restatements_have_the_reference_sizes(name):
    from contactimplicitmpc.jl_amd import lcp_models
    m = lcp_models.MODELS[name]()
    assert SHAPES[name]["rz_shape"] == [m.nz, m.nz] and SHAPES[name]["rth_shape_as_stored"] == [m.nth, m.nz]


def test_every_reference_model_is_covered_by_the_generic_dimension_rule():
    """nz = nq + 4 nc + 2 nb and nθ = 2 nq + nu + nw + 2 (index.jl:371-384) reproduce the stored shapes of the models
    this library has no dedicated kernel for, with their (nq, nu, nw, nc, nb) from src/dynamics/*/model.jl."""
    others = {"hopper_3D": dict(nq=7, nu=3, nw=3, nc=1, nb=4), "point_foot_quadruped": dict(nq=18, nu=12, nw=3, nc=4, nb=16),
              "centroidal_quadruped_box": dict(nq=18, nu=12, nw=3, nc=4, nb=16), "walledcartpole": dict(nq=4, nu=1, nw=4, nc=2, nb=4),
              "particle": dict(nq=3, nu=3, nw=3, nc=1, nb=4),
              "particle_2D": dict(nq=2, nu=2, nw=2, nc=1, nb=2),
              "centroidal_quadruped_wall": dict(nq=18, nu=12, nw=3, nc=8, nb=32)}
    for name, dm in others.items():
        d = Dims(**dm)
        assert SHAPES[name]["rz_shape"] == [d.nz, d.nz], name
        assert SHAPES[name]["rth_shape_as_stored"] == [d.nth, d.nz], name
