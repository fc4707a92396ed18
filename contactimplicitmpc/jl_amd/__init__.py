"""contactimplicitmpc.jl_amd - MI355X (gfx950) solver core of ContactImplicitMPC.jl's
per-MPC-step path (batched linearized-LCP interior-point solves + horizon Newton/KKT).

The product is the C-ABI library `libcimpc_hip.so` (include/cimpc.h); this package is
the thin Python host that mirrors the reference's interface for the path.
"""
from ._lib import CimpcError, LIB_PATH, load  # noqa: F401
from .solver import (CIMPCSolver, InteriorPointOptions, NewtonOptions,  # noqa: F401
                     MODE_CONFIGURATION, MODE_CONFIGURATIONFORCE)
from .trajectory import Dims, Traj, Objective  # noqa: F401  (host-side mirrors of the reference's data types)
