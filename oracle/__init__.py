"""oracle/ -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (plain C, `ppo_oracle.c`) of the reference's PPO hot path plus the harness that drives the
compiled reference itself (`ref_harness.cpp` -> `oracle/_ref/`).  Only `tests/`, `__graft_entry__.smoke()` and
`bench.py`'s `cpu_baseline` leg may import this package; nothing under `ppo-libtorch_amd/` does.
"""
from .oracle import *  # noqa: F401,F403
