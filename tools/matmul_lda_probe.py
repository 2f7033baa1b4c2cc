"""Does a power-of-two row pitch of the A operand cost memory-channel conflicts?  M = 65536, N = 256, K in (256, 1024), lda = K and K + 32."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402
P = load_package(); B = P.binding
ctx = P.Context(P.make_config(num_envs=8, num_steps=8))
rng = np.random.default_rng(0)
M, N = 65536, 256
for K in (256, 1024):
    for pad in (0, 32):
        lda = K + pad
        a = rng.standard_normal((M, lda)).astype(np.float32); b = (rng.standard_normal((N, K)) * 0.1).astype(np.float32)
        d_a, d_b, d_c = ctx.dev(a), ctx.dev(b), ctx.empty((M, N), np.float32)
        for prec, pname in ((B.MM_F32X3, "f32x3"), (B.MM_BF16, "bf16")):
            def run(n):
                for _ in range(n):
                    B.matmul_launch(ctx, False, False, M, N, K, d_a, lda, d_b, K, d_c, N, B.MM_EPI_NONE, None, 0, prec)
                ctx.sync()
            run(3); n = 30; t0 = time.perf_counter(); run(n); us = (time.perf_counter() - t0) / n * 1e6
            print(json.dumps({"K": K, "lda": lda, "precision": pname, "us": round(us, 1)}), flush=True)
        for x in (d_a, d_b, d_c): x.free()
ctx.close()
