"""Rates of ppo_matmul (kernels_gemm.hip) at the layer shapes of BASELINE configs[4] (minibatch 65 536 rows, 376 -> 256 -> 256 ...):
   python tools/matmul_bench.py [rows]
Prints one JSON line per (shape, orientation, precision): microseconds per launch and TFLOP/s (2 M N K, the f32 product count)."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package  # noqa: E402


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    P = load_package()
    B = P.binding
    ctx = P.Context(P.make_config(num_envs=8, num_steps=8))
    rng = np.random.default_rng(0)
    cases = [("forward 256->256", False, False, rows, 256, 256, B.MM_EPI_BIAS_TANH), ("forward 376->256", False, False, rows, 256, 376, B.MM_EPI_BIAS_TANH),
             ("forward head 256->11", False, False, rows, 11, 256, B.MM_EPI_BIAS), ("d(input) 256<-256", False, True, rows, 256, 256, B.MM_EPI_DTANH),
             ("d(weight) 256x256", True, True, 256, 256, rows, B.MM_EPI_NONE), ("d(weight) 11x256", True, True, 11, 256, rows, B.MM_EPI_NONE)]
    for name, ta, tb, M, N, K, epi in cases:
        a = rng.standard_normal((K, M) if ta else (M, K)).astype(np.float32)
        b = (rng.standard_normal((K, N) if tb else (N, K)) * 0.1).astype(np.float32)
        d_a, d_b, d_c = ctx.dev(a), ctx.dev(b), ctx.empty((M, N), np.float32)
        aux = None
        if epi in (B.MM_EPI_BIAS, B.MM_EPI_BIAS_TANH):
            aux = ctx.dev(rng.standard_normal(N).astype(np.float32))
        elif epi == B.MM_EPI_DTANH:
            aux = ctx.dev(np.tanh(rng.standard_normal((M, N))).astype(np.float32))
        for prec, pname in ((B.MM_F32X3, "f32x3"), (B.MM_BF16, "bf16")):
            def run(n):
                for _ in range(n):
                    B.matmul_launch(ctx, ta, tb, M, N, K, d_a, a.shape[1], d_b, b.shape[1], d_c, N, epi, aux, N if epi == B.MM_EPI_DTANH else 0, prec)
                ctx.sync()
            run(3)
            n = 30
            t0 = time.perf_counter()
            run(n)
            us = (time.perf_counter() - t0) / n * 1e6
            print(json.dumps({"case": name, "M": M, "N": N, "K": K, "precision": pname, "us": round(us, 1), "tflops": round(2.0 * M * N * K / us / 1e6, 1)}), flush=True)
        for x in (d_a, d_b, d_c) + ((aux,) if aux is not None else ()):
            x.free()
    ctx.close()


if __name__ == "__main__":
    main()
