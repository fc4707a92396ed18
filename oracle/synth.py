"""Seeded synthetic inputs: re-export of contactimplicitmpc/jl_amd/synthetic.py (the generator is an
input producer shared by bench.py, the tests and this checker; it holds no solver arithmetic)."""
from contactimplicitmpc.jl_amd.synthetic import make_problem, make_objective, make_rollout  # noqa: F401
