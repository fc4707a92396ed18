"""Problem dimensions and index layouts: re-export of the host-side definitions
(contactimplicitmpc/jl_amd/trajectory.py, which cites src/simulation/index.jl) so that the checker
and the device path are fed from the same layouts."""
from contactimplicitmpc.jl_amd.trajectory import (Dims, MODE_CONFIGURATION, MODE_CONFIGURATIONFORCE,  # noqa: F401
                                                  PUSHBOT, HOPPER_2D, QUADRUPED, CENTROIDAL, FLAMINGO, HOPPER_3D,
                                                  WALLEDCARTPOLE, PARTICLE, PARTICLE_2D, CENTROIDAL_WALL)
