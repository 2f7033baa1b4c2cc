"""K sweep of ppo_matmul at M = 65536, N = 256: separates the per-chunk cost from prologue + epilogue."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402
P = load_package(); B = P.binding
ctx = P.Context(P.make_config(num_envs=8, num_steps=8))
rng = np.random.default_rng(0)
M, N = 65536, 256
for K in (32, 64, 256, 1024):
    a = rng.standard_normal((M, K)).astype(np.float32); b = (rng.standard_normal((N, K)) * 0.1).astype(np.float32)
    d_a, d_b, d_c = ctx.dev(a), ctx.dev(b), ctx.empty((M, N), np.float32)
    bias = ctx.dev(rng.standard_normal(N).astype(np.float32))
    for epi in (B.MM_EPI_NONE, B.MM_EPI_BIAS_TANH):
        for prec, pname in ((B.MM_F32X3, "f32x3"), (B.MM_BF16, "bf16")):
            def run(n):
                for _ in range(n):
                    B.matmul_launch(ctx, False, False, M, N, K, d_a, K, d_b, K, d_c, N, epi, bias, 0, prec)
                ctx.sync()
            run(3); n = 30; t0 = time.perf_counter(); run(n); us = (time.perf_counter() - t0) / n * 1e6
            print(json.dumps({"K": K, "epi": epi, "precision": pname, "us": round(us, 1)}), flush=True)
    for x in (d_a, d_b, d_c, bias): x.free()
ctx.close()
