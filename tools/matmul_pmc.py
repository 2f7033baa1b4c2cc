"""One shape of ppo_matmul a few times (for rocprofv3 --pmc / --kernel-trace):  python tools/matmul_pmc.py [K] [precision 0|1] [M] [N]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402
K = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
prec = int(sys.argv[2]) if len(sys.argv) > 2 else 0
M = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
N = int(sys.argv[4]) if len(sys.argv) > 4 else 256
P = load_package(); B = P.binding
ctx = P.Context(P.make_config(num_envs=8, num_steps=8))
rng = np.random.default_rng(0)
a = rng.standard_normal((M, K)).astype(np.float32); b = (rng.standard_normal((N, K)) * 0.1).astype(np.float32)
d_a, d_b, d_c = ctx.dev(a), ctx.dev(b), ctx.empty((M, N), np.float32)
for _ in range(5):
    B.matmul_launch(ctx, False, False, M, N, K, d_a, K, d_b, K, d_c, N, B.MM_EPI_NONE, None, 0, prec)
ctx.sync()
ctx.close()
