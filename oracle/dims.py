"""Problem dimensions and index layouts of the CPU checker (test infrastructure): the checker's OWN restatement of
src/simulation/index.jl:13-107, 117-178, 187-327, 371-384 - z order [q2; g1; b1; psi1; s1; eta1; s2], th order
[q0; q1; u1; w1; mu; h], x = q2, y1 = [g1; b1; psi1], y2 = [s1; eta1; s2]; Newton block sizes newton_residual.jl:18-54, 69-98.
Deliberately NOT imported from the product package (contactimplicitmpc/jl_amd/trajectory.py holds the host side's copy):
tests/test_abi_and_host.py::test_oracle_containers_are_its_own checks that the two agree and that both match the
reference-derived numbers, so a mistake in either one shows up instead of cancelling."""
from dataclasses import dataclass

import numpy as np

MODE_CONFIGURATION = 0        # :configuration
MODE_CONFIGURATIONFORCE = 1   # :configurationforce


@dataclass(frozen=True)
class Dims:
    nq: int
    nu: int
    nw: int
    nc: int
    nb: int          # nc * friction_dim(env)
    mode: int = MODE_CONFIGURATION

    # ---- interior-point (per-knot LCP) sizes -------------------------------
    @property
    def nx(self):
        return self.nq

    @property
    def ny(self):
        return 2 * self.nc + self.nb

    @property
    def nz(self):   # index.jl:371-377
        return self.nq + 4 * self.nc + 2 * self.nb

    @property
    def nth(self):  # index.jl:379-384
        return 2 * self.nq + self.nu + self.nw + 2

    # ---- Newton (horizon) sizes ---------------------------------------------
    @property
    def nd(self):
        return self.nq if self.mode == MODE_CONFIGURATION else self.nq + self.nc + self.nb

    @property
    def nr(self):
        return self.nq + self.nu if self.mode == MODE_CONFIGURATION \
            else self.nq + self.nu + self.nc + self.nb

    # ---- index vectors (0-based) ---------------------------------------------
    @property
    def ix(self):
        return np.arange(0, self.nq)

    @property
    def iy1(self):
        return np.arange(self.nq, self.nq + self.ny)

    @property
    def iy2(self):
        return np.arange(self.nq + self.ny, self.nq + 2 * self.ny)

    idyn = ix
    irst = iy1   # the residual uses the same 3-group partition sizes
    ibil = iy2

    # z sub-blocks
    @property
    def ig1(self):
        return np.arange(self.nq, self.nq + self.nc)

    @property
    def ib1(self):
        return np.arange(self.nq + self.nc, self.nq + self.nc + self.nb)

    # theta sub-blocks
    @property
    def iq0(self):
        return np.arange(0, self.nq)

    @property
    def iq1(self):
        return np.arange(self.nq, 2 * self.nq)

    @property
    def iu1(self):
        return np.arange(2 * self.nq, 2 * self.nq + self.nu)

    @property
    def iw1(self):
        return np.arange(2 * self.nq + self.nu, 2 * self.nq + self.nu + self.nw)


# dimension constants of the models named by BASELINE.json configs
# (reference model files cited there)
PUSHBOT = dict(nq=2, nu=2, nw=2, nc=2, nb=4)        # pushbot/model.jl:126-130
HOPPER_2D = dict(nq=4, nu=2, nw=2, nc=1, nb=2)      # hopper_2D/model.jl:100-104
QUADRUPED = dict(nq=11, nu=8, nw=2, nc=4, nb=8)     # quadruped/model.jl:500-504
CENTROIDAL = dict(nq=18, nu=12, nw=3, nc=4, nb=16)  # centroidal_quadruped/model.jl:187-190 (= point_foot_quadruped, centroidal_quadruped_box)
FLAMINGO = dict(nq=9, nu=6, nw=2, nc=4, nb=8)       # flamingo/model.jl
HOPPER_3D = dict(nq=7, nu=3, nw=3, nc=1, nb=4)      # hopper_3D/model.jl:96-99
WALLEDCARTPOLE = dict(nq=4, nu=1, nw=4, nc=2, nb=4)  # walledcartpole/model.jl:143-147
PARTICLE = dict(nq=3, nu=3, nw=3, nc=1, nb=4)       # particle/model.jl:114-117
PARTICLE_2D = dict(nq=2, nu=2, nw=2, nc=1, nb=2)    # particle_2D/model.jl
CENTROIDAL_WALL = dict(nq=18, nu=12, nw=3, nc=8, nb=32)   # centroidal_quadruped_wall (nz = 114, ny = 48: runtime-dimension kernel)
