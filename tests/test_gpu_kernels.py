"""Per-kernel parity of the HIP path against numpy / the oracle, through the C-ABI.  GPU only."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import pgpfa_oracle as orc

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, np.max(np.abs(b)))


@pytest.fixture(scope='module')
def hip():
    from funs import _hip
    return _hip


@pytest.fixture(scope='module')
def small_ctx(hip):
    ctx = hip.Context(7, 2, 20, 2, 10.0)
    yield ctx
    ctx.close()


@pytest.mark.parametrize('M,N,K', [(128, 128, 16), (200, 150, 64), (300, 70, 160), (64, 260, 32), (513, 129, 48), (130, 260, 784)])
@pytest.mark.parametrize('mfma,tile', [(1, 64), (1, 128), (0, 64), (0, 128)])
def test_gemm_nt(small_ctx, M, N, K, mfma, tile):
    """C = alpha A B^T + beta C, asymmetric operands (catches transposed fragment layouts); both workgroup tile sizes."""
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    A = rng.standard_normal((M, K))
    B = rng.standard_normal((N, K)) + np.arange(N)[:, None] * 0.01
    C0 = rng.standard_normal((M, N))
    small_ctx.set_option('use_mfma', mfma)
    small_ctx.set_option('small_tile_below', 1 << 30 if tile == 64 else 0)
    try:
        C = small_ctx.test_gemm_nt(A, B, C0, alpha=-0.7, beta=1.3)
    finally:
        small_ctx.set_option('use_mfma', 1)
        small_ctx.set_option('small_tile_below', 1 << 30)
    ref = -0.7 * A @ B.T + 1.3 * C0
    assert rel(C, ref) <= 1e-13 * K


@pytest.mark.parametrize('M,N,K', [(128, 128, 16), (260, 70, 128), (64, 300, 48), (200, 150, 512), (640, 130, 1040)])
@pytest.mark.parametrize('mfma,tile', [(1, 64), (1, 128), (0, 64), (0, 128)])
def test_gemm_nn(small_ctx, M, N, K, mfma, tile):
    """The K x N (multi-RHS) operand form used by the shared-factor triangular sweeps (the long-K cases run split-K)."""
    rng = np.random.default_rng(M + N + K)
    A = rng.standard_normal((M, K))
    B = rng.standard_normal((K, N)) + np.arange(N)[None, :] * 0.01
    C0 = rng.standard_normal((M, N))
    small_ctx.set_option('use_mfma', mfma)
    small_ctx.set_option('small_tile_below', 1 << 30 if tile == 64 else 0)
    try:
        C = small_ctx.test_gemm_nn(A, B, C0, alpha=-1.0, beta=1.0)
    finally:
        small_ctx.set_option('use_mfma', 1)
        small_ctx.set_option('small_tile_below', 1 << 30)
    assert rel(C, C0 - A @ B) <= 1e-13 * K


@pytest.mark.parametrize('n,batch', [(100, 3), (128, 2), (300, 2), (700, 2), (1100, 1)])
@pytest.mark.parametrize('mfma', [1, 0])
def test_potrf_and_inverse(small_ctx, n, batch, mfma):
    """Blocked Cholesky (diag-block kernel + TRSM/SYRK GEMMs, super-panels) and L^-T L^-1 inverse."""
    rng = np.random.default_rng(n)
    A = []
    for b in range(batch):
        Q = rng.standard_normal((n, n))
        A.append(Q @ Q.T / n + np.diag(0.5 + rng.random(n)))
    A = np.stack(A)
    small_ctx.set_option('use_mfma', mfma)
    try:
        L, inv = small_ctx.test_potrf(A, want_inverse=True)
    finally:
        small_ctx.set_option('use_mfma', 1)
    for b in range(batch):
        assert rel(L[b], np.linalg.cholesky(A[b])) <= 1e-11
        assert rel(inv[b], np.linalg.inv(A[b])) <= 1e-10


def test_potrf_rejects_indefinite(small_ctx, hip):
    A = np.eye(140)
    A[77, 77] = -1.0
    with pytest.raises(hip.HipBackendError):
        small_ctx.test_potrf(A)


def test_profiling_entry_points(small_ctx):
    """The measurement helpers of the C-ABI: phase timings of the diagonal-block kernel and the per-shape GEMM report."""
    t0, t1, t3 = (small_ctx.bench_potrf_diag(4, 3, ph) for ph in (0, 1, 3))
    assert 0.0 < t0 < t1 < t3 < 1e4                                   # microseconds: load/store < + Cholesky steps < + inverse
    rng = np.random.default_rng(5)
    A, B = rng.standard_normal((130, 80)), rng.standard_normal((80, 90))
    small_ctx.set_option('profile', 2)
    try:
        C = small_ctx.test_gemm_nn(A, B)
        rep = small_ctx.gemm_shape_report()
    finally:
        small_ctx.set_option('profile', 0)
    assert rel(C, A @ B) <= 1e-13 * 80
    assert 'f64 NN M=130 N=90' in rep and 'TFLOP/s' in rep


def test_gram_and_inverse(hip, c1):
    g = load_golden('c1_callbacks.npz')
    ctx = hip.Context(30, 3, 100, 20, c1['binSize'])
    try:
        ctx.upload_counts(c1['Y'])
        ctx.set_params(c1['init_C'], c1['init_d'], c1['init_tau'])
        K = ctx.gram()
        assert np.max(np.abs(K - g['K'])) <= 1e-15                   # closed form, SURVEY 8c
        Kinv = ctx.gram_inverse()
        # cond(K) ~ 7e4: explicit inverses agree to ~cond*eps relative
        assert rel(Kinv, np.linalg.inv(g['K'])) <= 1e-9
        assert rel(np.einsum('kij,kjl->kil', K, Kinv), np.broadcast_to(np.eye(100), (3, 100, 100))) <= 1e-9
    finally:
        ctx.close()


def test_laplace_callbacks_vs_golden(hip, c1):
    """a4-a6: objective, gradient, Hessian at the reference's probe point (1e-9 rel)."""
    g = load_golden('c1_callbacks.npz')
    ctx = hip.Context(30, 3, 100, 20, c1['binSize'])
    try:
        ctx.upload_counts(c1['Y'])
        ctx.set_params(c1['init_C'], c1['init_d'], c1['init_tau'])
        f, grad = ctx.laplace_eval(np.array([0]), g['xprobe'][None])
        assert abs(f[0] - g['f']) <= 1e-9 * abs(g['f'])
        assert rel(grad[0].reshape(-1), g['g']) <= 1e-9
        H = ctx.laplace_hessian(0, g['xprobe'])
        assert rel(H, g['H']) <= 1e-9
        # several trials at once, against the oracle's structured form
        rng = np.random.default_rng(3)
        X = 0.2 * rng.standard_normal((5, 3, 100))
        idx = np.array([3, 0, 19, 7, 7])
        f, grad = ctx.laplace_eval(idx, X)
        Kinv = np.linalg.inv(g['K'])
        for i, r in enumerate(idx):
            fr = orc.nlp(X[i], c1['Ys'][r], c1['init_C'], c1['init_d'], Kinv)
            gr = orc.nlp_grad(X[i], c1['Ys'][r], c1['init_C'], c1['init_d'], Kinv)
            assert abs(f[i] - fr) <= 1e-9 * abs(fr)
            assert rel(grad[i], gr) <= 1e-9
    finally:
        ctx.close()


def test_counts_validation(hip):
    ctx = hip.Context(4, 2, 10, 2, 10.0)
    try:
        Y = np.zeros((2, 4, 10))
        Y[1, 2, 3] = 300.0
        ctx.upload_counts(Y)                # counts above one byte are accepted (second byte plane) ...
        assert ctx.info('counts_two_bytes') == 1.0 and ctx.counts()[1, 2, 3] == 300
        Y[1, 2, 3] = 65536.0                # ... up to 65535
        with pytest.raises(hip.HipBackendError):
            ctx.upload_counts(Y)
        Y[1, 2, 3] = -1.0
        with pytest.raises(hip.HipBackendError):
            ctx.upload_counts(Y)
        Y[1, 2, 3] = 1.5
        with pytest.raises(hip.HipBackendError):
            ctx.upload_counts(Y)
        with pytest.raises(hip.HipBackendError):
            ctx.estep_laplace()             # no parameters set yet -> loud failure, no fallback
    finally:
        ctx.close()


def test_mstep_cd_costgrad_vs_golden(hip, c1):
    """a9: cost/grad on the reference's own E-step output, uploaded through set_posterior (1e-10 rel)."""
    g = load_golden('c1_mstep.npz')
    lap = load_golden('c1_laplace.npz')
    ctx = hip.Context(30, 3, 100, 20, c1['binSize'])
    try:
        ctx.upload_counts(c1['Y'])
        ctx.set_params(c1['init_C'], c1['init_d'], c1['init_tau'])
        vsmgp = np.zeros((20, 100, 100, 3))
        ctx.set_posterior(None, lap['post_mean'], lap['post_vsm'], vsmgp)
        for v, cref, gref in ((g['v0'], g['cost0'], g['grad0']), (g['v1'], g['cost1'], g['grad1'])):
            cost, grad = ctx.mstep_cd_costgrad(v)
            assert abs(cost - cref) <= 1e-10 * abs(cref)
            assert rel(grad, gref) <= 1e-10
        cost, grad = ctx.mstep_cd_costgrad(g['v1'], g['v0'], 1.0 / float(g['prior_step']) ** 2)
        assert abs(cost - g['costp']) <= 1e-10 * abs(g['costp'])
        assert rel(grad, g['gradp']) <= 1e-10
    finally:
        ctx.close()


def test_mstep_tau_costgrad_vs_golden(hip, c1):
    """a11: PautoSum from uploaded posterior blocks, then cost/grad at the reference's probes."""
    g = load_golden('c1_mstep.npz')
    lap = load_golden('c1_laplace.npz')
    res, _, _ = orc.laplace(c1['Ys'], c1['init'], c1['binSize'], mode='exact', return_cov=False)
    ctx = hip.Context(30, 3, 100, 20, c1['binSize'])
    try:
        ctx.upload_counts(c1['Y'])
        ctx.set_params(c1['init_C'], c1['init_d'], c1['init_tau'])
        ctx.set_posterior(None, np.stack(res['post_mean']), np.stack(res['post_vsm']), np.stack(res['post_vsmGP']))
        n = ctx.mstep_precomp()
        assert n == 20
        P_ref, _ = orc.make_precomp(res)
        P = ctx.pautosum()
        assert rel(P, P_ref) <= 1e-12
        for k in range(3):
            for pv in g['pprobe']:
                cost, grad = ctx.mstep_tau_costgrad(k, pv)
                cref = orc.tau_cost(pv, P_ref[k], 20)
                gref = orc.tau_grad(pv, P_ref[k], 20)[0]
                assert abs(cost - cref) <= 1e-9 * abs(cref)
                assert abs(grad - gref) <= 1e-8 * max(1.0, abs(gref))
    finally:
        ctx.close()


def test_library_reports_info(small_ctx):
    assert small_ctx.info('n_pad') == 128
    assert small_ctx.info('hbm_bytes_allocated') > 0


def test_tau_batch_equals_single_and_lockstep_matches_bfgs(hip, c1):
    """Batched tau cost/grad == per-latent evaluation; the lockstep solver lands on scipy BFGS's optimum."""
    import funs
    from conftest import Experiment
    res, _, _ = orc.laplace(c1['Ys'], c1['init'], c1['binSize'], mode='exact', return_cov=False)
    ctx = hip.Context(30, 3, 100, 20, c1['binSize'])
    try:
        ctx.upload_counts(c1['Y'])
        ctx.set_params(c1['init_C'], c1['init_d'], c1['init_tau'])
        ctx.set_posterior(None, np.stack(res['post_mean']), np.stack(res['post_vsm']), np.stack(res['post_vsmGP']))
        ctx.mstep_precomp()
        logp = np.array([-6.0, -7.5, -5.2])
        cb, gb = ctx.mstep_tau_costgrad_batch(logp)
        for k in range(3):
            c1_, g1_ = ctx.mstep_tau_costgrad(k, logp[k])
            assert abs(cb[k] - c1_) <= 1e-12 * abs(c1_) and abs(gb[k] - g1_) <= 1e-10 * max(1.0, abs(g1_))
        # several candidate points per latent in one pass (candidate-major) == one pass per candidate set
        cand = np.stack([logp, logp + 0.3, logp - 0.2, logp + 0.01])
        cm, gm = ctx.mstep_tau_costgrad_multi(cand)
        for j in range(4):
            cj, gj = ctx.mstep_tau_costgrad_batch(cand[j])
            assert np.max(np.abs(cm[j] - cj) / np.abs(cj)) <= 1e-12 and np.max(np.abs(gm[j] - gj)) <= 1e-9 * np.max(np.abs(gj))
        with pytest.raises(hip.HipBackendError):
            ctx.mstep_tau_costgrad_multi(np.zeros((5, 3)))
    finally:
        ctx.close()
    exp = Experiment(c1['Ys'], c1['binSize'])
    tau_o, _ = orc.learn_tau(c1['init'], res, c1['binSize'])
    for solver in ('lockstep', 'secant', 'scipy'):
        funs.learning.TAU_SOLVER = solver
        try:
            tau, det = funs.learning.learnGPparams(c1['init'], res, exp)
        finally:
            funs.learning.TAU_SOLVER = 'lockstep'
        assert np.max(np.abs(np.log(tau) - np.log(tau_o))) <= 1e-7, solver


@pytest.mark.parametrize('q,p,T', [(13, 7, 41), (30, 3, 100), (50, 10, 70), (20, 16, 33)])
def test_poisson_pass_mfma_matches_vector_kernel_and_oracle(hip, q, p, T):
    """The matrix-core Poisson pass (objective, gradient, per-bin curvature blocks) against the vector kernel
    (use_mfma = 0) and the oracle, at sizes that exercise every padding rule (neurons, latents, bins)."""
    rng = np.random.default_rng(q * 100 + p)
    R = 3
    Y = rng.poisson(0.7, size=(R, q, T)).astype(np.uint8)
    C = 0.4 * rng.standard_normal((q, p))
    d = -0.5 + 0.3 * rng.standard_normal(q)
    tau = 0.05 + 0.2 * rng.random(p)
    X = 0.3 * rng.standard_normal((R, p, T))
    out = {}
    for mfma in (1, 0):
        ctx = hip.Context(q, p, T, R, 10.0)
        try:
            ctx.upload_counts(Y)
            ctx.set_option('use_mfma', mfma)
            ctx.set_params(C, d, tau)
            f, g = ctx.laplace_eval(np.arange(R), X)
            H = ctx.laplace_hessian(1, X[1])
            out[mfma] = (f, g, H)
        finally:
            ctx.close()
    assert rel(out[1][0], out[0][0]) <= 1e-13
    assert rel(out[1][1], out[0][1]) <= 1e-12
    assert rel(out[1][2], out[0][2]) <= 1e-12
    Kinv = np.linalg.inv(orc.make_K(tau, T, 10.0))
    for r in range(R):
        assert abs(out[1][0][r] - orc.nlp(X[r], Y[r].astype(float), C, d, Kinv)) <= 1e-10 * abs(out[1][0][r])
        assert rel(out[1][1][r], orc.nlp_grad(X[r], Y[r].astype(float), C, d, Kinv)) <= 1e-9


@pytest.mark.parametrize('q,T,R', [(30, 100, 20), (13, 41, 3), (70, 600, 2), (200, 500, 4)])
def test_count_moments_exact(hip, q, T, R):
    """Integer first/second moments of the resident counts (dot4 kernel + 64-bit atomics): bit-exact against numpy's
    integer arithmetic, including counts up to 255, neuron tiles with a ragged edge and bins beyond one LDS chunk."""
    rng = np.random.default_rng(q + T)
    Y = rng.poisson(0.8, size=(R, q, T))
    Y[0, 0, :7] = 255
    Y[R - 1, q - 1, T - 3:] = 254
    Y = np.minimum(Y, 255).astype(np.uint8)
    ctx = hip.Context(q, 2, T, R, 10.0)
    try:
        ctx.upload_counts(Y)
        for idx in (None, np.array([R - 1, 0], dtype=np.int32)):
            s, S, ns = ctx.count_moments(idx)
            Ysel = Y if idx is None else Y[idx]
            flat = np.transpose(Ysel, (1, 0, 2)).reshape(q, -1).astype(np.int64)
            assert ns == flat.shape[1]
            assert np.array_equal(s, flat.sum(axis=1))
            assert np.array_equal(S, flat @ flat.T)
    finally:
        ctx.close()
