"""CPU oracle for the lang2seg train-step hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product package (`lang2seg_amd/`)
imports this directory; only `tests/`, `__graft_entry__.smoke()` and the
`cpu_baseline` leg of `bench.py` may.  It is a from-scratch restatement
(numpy for integer/box arithmetic, torch-CPU fp32 for the float graph) of the
reference algorithm; every function cites the reference file:line it follows.

Pinning: the reference holds no tests or golden vectors for this path except
the anchor docstring (generate_anchors.py:14-39).  The oracle is pinned against
fixtures produced by running the *reference itself* in the dev container through
an import harness (tests/golden/make_golden.py, committed with its outputs in
tests/golden/*.npz).  tests/test_oracle_vs_golden.py checks that pin on CPU.
"""
