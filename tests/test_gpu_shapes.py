"""Seeded random problem shapes through both covariance engines and the dual evaluation, against the oracle / plain numpy.

The fixed-size parity tests sit on the shapes the kernels were tuned at; these walk the instantiation boundaries instead (latent widths
1..32 across the 4/8/10/12/16/20/24/32 buckets, bin counts that are not multiples of the 16/32/64-bin tiles, neuron counts that are not
multiples of 16, one to a few trials).  Same tolerances as tests/test_gpu_parity.py: modes 1e-8, objective 1e-9 rel, covariance blocks 1e-8 rel."""
import numpy as np
import pytest

from oracle import pgpfa_oracle as orc

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, np.max(np.abs(b)))


def _shape(seed):
    rng = np.random.default_rng(1000 + seed)
    p = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 10, 11, 12, 13, 15, 16, 17, 20, 21, 24, 25, 29, 32]))
    T = int(rng.integers(6, 80))
    q = int(rng.integers(max(3, p // 2), 48))
    R = int(rng.integers(1, 5))
    return q, p, T, R, rng


@pytest.mark.parametrize('seed', range(14))
@pytest.mark.parametrize('cov_mode', [1, 2])
def test_laplace_estep_random_shapes(seed, cov_mode):
    from funs import _hip
    q, p, T, R, rng = _shape(seed)
    _, Ys, _ = orc.synth_dataset(q, p, T, R, seed=seed, dOffset=0.0)
    Y = np.stack(Ys).astype(np.uint8)
    par = {'C': 0.3 * rng.standard_normal((q, p)) / np.sqrt(max(1, p / 4)), 'd': np.log(Y.mean(axis=(0, 2)) + 0.1),
           'tau': 0.03 + 0.25 * rng.random(p)}
    res, nll_o, _ = orc.laplace([y.astype(float) for y in Ys], par, 10.0, mode='exact', return_cov=False)
    ctx = _hip.Context(q, p, T, R, 10.0)
    try:
        ctx.upload_counts(Y)
        ctx.set_option('cov_mode', cov_mode)
        ctx.set_option('keep_trial_vsmgp', 1)
        ctx.set_params(par['C'], par['d'], par['tau'])
        obj, _, status = ctx.estep_laplace()
        tag = 'shape q=%d p=%d T=%d R=%d cov_mode=%d (low-rank plan %d)' % (q, p, T, R, cov_mode, int(ctx.info('plan_lowrank')))
        assert np.all(status == 0), tag
        assert abs(-obj / R - nll_o) <= 1e-9 * abs(nll_o), tag
        assert np.max(np.abs(ctx.post_mean() - np.stack(res['post_mean']))) <= 1e-8, tag
        assert rel(ctx.post_vsm(), np.stack(res['post_vsm'])) <= 1e-8, tag
        assert rel(ctx.post_vsmgp(), np.stack(res['post_vsmGP'])) <= 1e-8, tag
    finally:
        ctx.close()


@pytest.mark.parametrize('seed', range(10))
@pytest.mark.parametrize('lowrank', [0, 1])
def test_dual_evaluation_random_shapes(seed, lowrank):
    """Dual cost + gradient (inference.py:196-219, with the reference's 1e-6 jitter, inference.py:190) at random shapes: dense engine
    and low-rank engine (the jitter carried by the per-bin blocks) against the oracle."""
    from funs import _hip
    q, p, T, R, rng = _shape(100 + seed)
    _, Ys, _ = orc.synth_dataset(q, p, T, R, seed=seed, dOffset=0.0)
    Y = np.stack(Ys).astype(np.uint8)
    C, d, tau = 0.25 * rng.standard_normal((q, p)), np.log(Y.mean(axis=(0, 2)) + 0.1), 0.2 + 0.4 * rng.random(p)
    lam = 0.05 + 0.5 * rng.random((R, q * T))
    idx = np.arange(R, dtype=np.int32)
    ctx = _hip.Context(q, p, T, R, 10.0)
    try:
        ctx.upload_counts(Y)
        if lowrank:
            ctx.set_option('cov_mode', 2)
            ctx.set_option('dual_lowrank', 1)
        else:
            ctx.set_option('cov_mode', 1)
        ctx.set_params(C, d, tau)
        cost, grad = ctx.dual_costgrad_batch(idx, lam)
        used_lowrank = bool(ctx.info('plan_lowrank'))
    finally:
        ctx.close()
    tag = 'shape q=%d p=%d T=%d R=%d lowrank=%d (plan %d)' % (q, p, T, R, lowrank, used_lowrank)
    K_big = orc.make_K_big(orc.make_K(tau, T, 10.0))
    C_big, d_big = orc.make_Cd_big(C, d, T)
    Kinv_big = np.linalg.inv(K_big)
    for i in range(R):
        y = Ys[i].reshape(-1).astype(float)
        # both engines evaluate the reference's function, 1e-6 diagonal jitter included (inference.py:190)
        ref_cost = orc.dual_cost(lam[i], y, C_big, K_big, Kinv_big, d_big)
        ref_grad = orc.dual_grad(lam[i], y, C_big, K_big, Kinv_big, d_big)
        assert abs(cost[i] - ref_cost) <= 1e-8 * abs(ref_cost), tag
        assert rel(grad[i], ref_grad) <= 1e-7, tag


@pytest.mark.parametrize('q,p,T,R', [(300, 4, 40, 3), (517, 10, 24, 2), (260, 12, 36, 2)])
def test_mstep_many_neurons(q, p, T, R):
    """(C,d) M-step passes beyond 256 neurons (several neuron tiles per workgroup pass, the count-term kernel's second sweep): cost,
    gradient and the device Newton iteration against the oracle."""
    from funs import _hip
    _, Ys, _ = orc.synth_dataset(q, p, T, R, seed=q, dOffset=0.0)
    Y = np.stack(Ys).astype(np.uint8)
    rng = np.random.default_rng(q)
    par = {'C': 0.3 * rng.standard_normal((q, p)) / np.sqrt(p), 'd': np.log(Y.mean(axis=(0, 2)) + 0.1), 'tau': 0.05 + 0.2 * rng.random(p)}
    ctx = _hip.Context(q, p, T, R, 10.0)
    try:
        ctx.upload_counts(Y)
        ctx.set_params(par['C'], par['d'], par['tau'])
        obj, _, status = ctx.estep_laplace()
        assert np.all(status == 0)
        pm, vs = [m for m in ctx.post_mean()], [m for m in ctx.post_vsm()]
        Yf = [y.astype(float) for y in Ys]
        v = orc.cd_to_vec(par['C'], par['d']) + 0.01 * rng.standard_normal(q * (p + 1))
        cost, grad = ctx.mstep_cd_costgrad(v)
        assert abs(cost - orc.mstep_cd_cost(v, Yf, pm, vs, p, q)) <= 1e-10 * abs(cost)
        assert rel(grad, orc.mstep_cd_grad(v, Yf, pm, vs, p, q)) <= 1e-9
        cost_n, delta, dec = ctx.mstep_cd_newton_pass(v)
        assert abs(cost_n.sum() - cost) <= 1e-10 * abs(cost) and np.all(dec >= 0)
        g0 = np.max(np.abs(grad))
        for _ in range(25):                                  # full Newton steps (a nearly silent neuron takes a dozen from 0.01 off)
            v = v + delta.reshape(-1)
            cost_n, delta, dec = ctx.mstep_cd_newton_pass(v)
            if np.max(np.abs(delta)) < 1e-12:
                break
        g_end = np.max(np.abs(orc.mstep_cd_grad(v, Yf, pm, vs, p, q)))
        assert g_end <= 1e-10 * max(1.0, g0), (g_end, g0)
        cost_c, _, _ = ctx.mstep_cd_chord_pass(v)
        assert abs(cost_c.sum() - cost_n.sum()) <= 1e-12 * abs(cost_n.sum())
    finally:
        ctx.close()
