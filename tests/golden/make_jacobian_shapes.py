"""Generates tests/golden/jacobian_shapes.json from the reference's own files (run in the build container, where
/root/reference exists; the JSON is the committed fixture).

`src/simulation/<model>/flat/jacobians.jld2` is written by `generate_residual_expressions` + `JLD2.save(path_jac, "rz_sp", rz_sp,
"rθ_sp", rθ_sp)` (src/simulation/generate_simulation.jl:20-23 and the following blocks).  Read with the from-the-format
JLD2 reader of this repository the two arrays turn out to be DENSE `similar(...)` buffers whose payload is uninitialised
memory (small integers, pointers, -4096 patterns when viewed as Int64): they carry no sparsity pattern, so they cannot pin
the structure of the Jacobians.  What they do pin is the SHAPE the reference's code generation works with: rz is nz x nz and
rθ is nz x nθ for every model - i.e. the index layout restated in contactimplicitmpc/jl_amd/trajectory.py (Dims.nz, Dims.nth)
and the model dimensions of lcp_models.py."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from contactimplicitmpc.jl_amd import gait_io  # noqa: E402

REF = "/root/reference/src/simulation"
out = {}
for model in sorted(os.listdir(REF)):
    p = os.path.join(REF, model, "flat", "jacobians.jld2")
    if not os.path.exists(p):
        continue
    try:
        d = gait_io.read_jld2(p)
    except Exception as e:      # outside the reader's subset: recorded, not guessed
        out[model] = {"error": type(e).__name__}
        continue
    rz, rth = d["rz_sp"], d["rθ_sp"]
    as_int = rz.view(np.int64)
    out[model] = {"rz_shape": list(rz.shape), "rth_shape_as_stored": list(rth.shape),      # numpy view of Julia's column-major (nz, nθ)
                  "payload_looks_uninitialised": bool((np.abs(as_int) > (1 << 40)).mean() > 0.01 or (as_int == -4096).any())}
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "jacobian_shapes.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1))
