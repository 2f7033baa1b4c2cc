"""ppo_matmul (include/ppo_hip.h): the Linear-layer product of networks wider than 2 x 64 on the matrix cores, against numpy float64.

PPO_MM_F32X3 carries every f32 operand as three exact bf16 terms and drops the three smallest of the nine cross products:
|c - exact| <= ~4 x 2^-24 sum_k |a||b| + f32 accumulation (the test allows 2e-6 of sum_k |a||b|, a bound an f32 fmaf chain
meets as well).  PPO_MM_BF16 rounds the operands to bf16 once: checked against float64 on the SAME rounded operands.
Shapes cover full tiles, ragged edges in every dimension, the narrow-tile variants (N <= 32, M <= 32), unaligned leading dimensions
and every operand orientation the forward / backward of a layer uses."""
import numpy as np
import pytest

from __graft_entry__ import load_package

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def P():
    return load_package()


@pytest.fixture(scope="module")
def ctx(P):
    c = P.Context(P.make_config(num_envs=8, num_steps=8))
    yield c
    c.close()


def bf16_rne(x):
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


SHAPES = [(128, 128, 32), (256, 256, 256), (300, 256, 376), (1000, 11, 256), (517, 1, 256), (11, 256, 1000), (1, 376, 777),
          (130, 130, 40), (64, 40, 7), (2048, 256, 376)]


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, True), (True, False)])
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_matmul_f32x3_matches_float64(P, ctx, M, N, K, ta, tb):
    rng = np.random.default_rng(M * 131 + N * 17 + K + 2 * ta + tb)
    A = (rng.standard_normal((M, K)) * np.exp(rng.uniform(-3, 3, (M, K)))).astype(np.float32)
    B = (rng.standard_normal((N, K)) * 0.3).astype(np.float32)
    a = np.ascontiguousarray(A.T) if ta else A
    b = np.ascontiguousarray(B.T) if tb else B
    c = P.binding.matmul(ctx, a, b, ta, tb)
    ref = A.astype(np.float64) @ B.astype(np.float64).T
    bound = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64).T
    assert c.shape == (M, N)
    assert np.all(np.abs(c - ref) <= 2e-6 * bound + 1e-30), np.max(np.abs(c - ref) / (bound + 1e-30))


@pytest.mark.parametrize("M,N,K", [(256, 256, 256), (300, 11, 376), (130, 130, 40)])
def test_matmul_bf16_matches_float64_on_rounded_operands(P, ctx, M, N, K):
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K)).astype(np.float32)
    B = (rng.standard_normal((N, K)) * 0.3).astype(np.float32)
    c = P.binding.matmul(ctx, A, B, precision=P.binding.MM_BF16)
    ref = bf16_rne(A).astype(np.float64) @ bf16_rne(B).astype(np.float64).T
    bound = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64).T
    assert np.all(np.abs(c - ref) <= 2e-6 * bound)
    # and it is what it says: bf16-level agreement with the unrounded product
    full = A.astype(np.float64) @ B.astype(np.float64).T
    assert np.max(np.abs(c - full) / bound) < 2 ** -7


def test_matmul_epilogues(P, ctx):
    B_ = P.binding
    rng = np.random.default_rng(5)
    M, N, K = 200, 256, 96
    x = rng.standard_normal((M, K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) * 0.2).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    z = x.astype(np.float64) @ W.astype(np.float64).T + bias
    np.testing.assert_allclose(B_.matmul(ctx, x, W, epilogue=B_.MM_EPI_BIAS, aux=bias), z, rtol=0, atol=2e-5)
    h = B_.matmul(ctx, x, W, epilogue=B_.MM_EPI_BIAS_TANH, aux=bias)
    np.testing.assert_allclose(h, np.tanh(z), rtol=0, atol=2e-6)
    # d(input) of the layer above: dz' = (dz W')(1 - h^2) with W' [out, N] read as [k = out][N]
    out = 40
    Wn = (rng.standard_normal((out, N)) * 0.2).astype(np.float32)
    dz = rng.standard_normal((M, out)).astype(np.float32)
    got = B_.matmul(ctx, dz, Wn, False, True, epilogue=B_.MM_EPI_DTANH, aux=h)
    ref = (dz.astype(np.float64) @ Wn.astype(np.float64)) * (1.0 - h.astype(np.float64) ** 2)
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5)


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, True)])
def test_matmul_both_main_loops(P, ctx, ta, tb):
    """Both main loops of the layer product (kernels_gemm.hip picks one by batch size and precision): the eight-wave double-buffered
    one (up to 8192 rows, and every plain-bf16 product) and the two-workgroups-per-CU one (larger fp32-accurate products) give results
    within the same bound -- shapes with an odd number of chunk pairs, ragged edges, a contraction shorter than the prefetch distance,
    and one shape on the far side of the row threshold."""
    for M, N, K in [(300, 256, 376), (256, 256, 256), (130, 130, 40), (512, 384, 1000), (8200, 256, 136)]:
        rng = np.random.default_rng(M + N + K)
        A = rng.standard_normal((M, K)).astype(np.float32)
        B = (rng.standard_normal((N, K)) * 0.3).astype(np.float32)
        a = np.ascontiguousarray(A.T) if ta else A
        b = np.ascontiguousarray(B.T) if tb else B
        c = P.binding.matmul(ctx, a, b, ta, tb)
        ref = A.astype(np.float64) @ B.astype(np.float64).T
        bound = np.abs(A).astype(np.float64) @ np.abs(B).astype(np.float64).T
        assert np.all(np.abs(c - ref) <= 2e-6 * bound + 1e-30), (M, N, K)


def test_matmul_degenerate_sizes_and_bad_arguments(P, ctx):
    B_ = P.binding
    # K = 0: the product is empty, the epilogue still applies
    bias = np.arange(5, dtype=np.float32)
    d_a, d_b, d_c, d_x = ctx.dev(np.zeros((3, 1), np.float32)), ctx.dev(np.zeros((5, 1), np.float32)), ctx.dev(np.full((3, 5), 7.0, np.float32)), ctx.dev(bias)
    B_.matmul_launch(ctx, False, False, 3, 5, 0, d_a, 1, d_b, 1, d_c, 5, B_.MM_EPI_BIAS, d_x, 0)
    ctx.sync()
    assert np.array_equal(d_c.download(), np.tile(bias, (3, 1)))
    # M = 0 / N = 0: nothing is written, status OK
    d_c.upload(np.full((3, 5), 7.0, np.float32))
    B_.matmul_launch(ctx, False, False, 0, 5, 1, d_a, 1, d_b, 1, d_c, 5)
    B_.matmul_launch(ctx, False, False, 3, 0, 1, d_a, 1, d_b, 1, d_c, 5)
    ctx.sync()
    assert np.all(d_c.download() == 7.0)
    # an epilogue without its operand, an unknown epilogue / precision: PPO_ERR_INVALID, not a launch
    for kw in (dict(epilogue=B_.MM_EPI_BIAS), dict(epilogue=9, d_aux=d_x), dict(precision=5)):
        with pytest.raises(B_.PPOError):
            B_.matmul_launch(ctx, False, False, 3, 5, 1, d_a, 1, d_b, 1, d_c, 5, **kw)
