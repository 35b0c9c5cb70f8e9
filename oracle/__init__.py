"""CPU oracle for the SampleNeRFRO hot path — TEST INFRASTRUCTURE ONLY.

This package is a numpy restatement of the reference algorithm
(`/root/reference/rnerf/{eikonal_utils,ior_utils,model_utils,models,math_utils}.py`,
`train.py:75-183`).  It is the checker for the HIP path in `samplenerfro_amd/`.

PARITY UNPINNED: the reference ships no tests / golden vectors for this path and
JAX/flax cannot be imported in the build container, so this
restatement is pinned only by the analytic known-answer tests in
`tests/test_oracle_kat.py` and by the Random123 threefry2x32 vector.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this package.  The product (`samplenerfro_amd`) never does.
"""
