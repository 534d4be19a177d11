"""Round-3 GPU parity tests: spike counts above 255 pinned to the reference (tests/golden/c1_highcount.npz), BASELINE config 5
as a workload (the dual-variational E-step run to convergence at 500 neurons x 20 latents x 1000 bins, checked against plain
numpy on the dense matrices), the per-trial parameter snapshots behind lazily rebuilt covariance blocks (minibatch EM), stale
count terms, and the growing workspace arena.  Everything goes through the C-ABI wrapper or the drop-in `funs` surface; the
oracle, numpy and the golden vectors are the checkers."""
import numpy as np
import pytest

from conftest import Experiment, load_golden
from oracle import pgpfa_oracle as orc

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, np.max(np.abs(b)))


@pytest.fixture(scope='module')
def funs_mod():
    import funs
    return funs


# ---------------------------------------------------------------------------------------------------------------
# counts above 255 (reference util.py:741,750 keeps counts as float64 of any size)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dtype', ['uint16', 'float64'])
@pytest.mark.parametrize('mfma', [1, 0])
def test_counts_above_255_vs_reference(c1, dtype, mfma):
    """3 trials of config 1 with bins of 256 ... 1200 spikes (one of exactly 300), every use of the counts against the values
    captured from the reference: objective / gradient at a probe (inference.py:12-48) 1e-9, E-step modes vs the polished modes
    1e-8 and covariance blocks 1e-8 rel, the (C,d) cost / gradient on the reference's own posterior (learning.py:20-91) 1e-10,
    dual cost / gradient (inference.py:196-219) 1e-9, the exact integer count moments; matrix-core and vector kernels."""
    from funs import _hip
    g = load_golden('c1_highcount.npz')
    Y = g['Y']
    assert Y.max() == 1200 and Y[0, 4, 17] == 300
    R, q, T = Y.shape
    p = 3
    ctx = _hip.Context(q, p, T, R, c1['binSize'])
    try:
        ctx.set_option('use_mfma', mfma)
        ctx.upload_counts(Y.astype(dtype))
        assert ctx.info('counts_two_bytes') == 1.0
        assert np.array_equal(ctx.counts(), Y)
        ctx.set_params(c1['init_C'], c1['init_d'], c1['init_tau'])
        f, grad = ctx.laplace_eval(np.array([0], dtype=np.int32), g['xprobe'][None, :])
        assert abs(f[0] - float(g['f'])) <= 1e-9 * abs(float(g['f'])) and rel(grad.reshape(-1), g['g']) <= 1e-9
        obj, iters, status = ctx.estep_laplace()
        assert np.all(status == 0)
        assert np.max(np.abs(ctx.post_mean().reshape(R, -1) - g['polished'])) <= 1e-8
        assert rel(ctx.post_vsm(), g['post_vsm_polished']) <= 1e-8
        assert abs(obj - float(g['nlp_polished_sum'])) <= 1e-9 * abs(float(g['nlp_polished_sum']))
        # vs the reference's own early-stopped output (its slack)
        assert np.max(np.abs(ctx.post_mean() - g['post_mean'])) <= 5e-3 and abs(-obj / R - float(g['nll'])) <= 1e-4 * abs(float(g['nll']))
        # PautoSum from the device posterior vs the reference's (early-stopped modes: its slack again)
        ctx.mstep_precomp()
        assert rel(ctx.pautosum(), g['PautoSum']) <= 5e-3
        # (C,d) cost / gradient on the reference's posterior, both kernels
        ctx.set_posterior(None, g['post_mean'], g['post_vsm'])
        for cd_mfma in (1, 0):
            ctx.set_option('cd_mfma', cd_mfma)
            cost, gr = ctx.mstep_cd_costgrad(g['v1'])
            assert abs(cost - float(g['cost1'])) <= 1e-10 * abs(float(g['cost1'])) and rel(gr, g['grad1']) <= 1e-10
        # the Newton pass sees the same cost (its per-neuron costs sum to it)
        cost_n, _, _ = ctx.mstep_cd_newton_pass(g['v1'])
        assert abs(np.sum(cost_n) - float(g['cost1'])) <= 1e-10 * abs(float(g['cost1']))
        # dual cost / gradient of trial 0
        dc, dg = ctx.dual_costgrad(0, g['lam'])
        assert abs(dc - float(g['dual_cost'])) <= 1e-9 * abs(float(g['dual_cost'])) and rel(dg, g['dual_grad']) <= 1e-9
        s, cross, n = ctx.count_moments()
        assert n == R * T and np.array_equal(s, g['raster_sum'].astype(np.int64)) and np.array_equal(cross, g['raster_cross'].astype(np.int64))
        # a one-byte tensor afterwards drops the second plane again
        ctx.upload_counts(c1['Y'][:R])
        assert ctx.info('counts_two_bytes') == 0.0 and np.array_equal(ctx.counts(), c1['Y'][:R])
        with pytest.raises(_hip.HipBackendError):
            ctx.upload_counts(np.full((R, q, T), 70000.0))
    finally:
        ctx.close()


def test_counts_above_255_through_the_engine(funs_mod, c1):
    """The drop-in surface with such counts: experiment.data[r]['Y'] as float arrays (the reference's format), one batch-EM iteration;
    E-step objective and the M-step's (C,d) optimality against the oracle."""
    g = load_golden('c1_highcount.npz')
    Ys = [g['Y'][r].astype(float) for r in range(3)]
    exp = Experiment(Ys, c1['binSize'])
    par = {'C': c1['init_C'].copy(), 'd': c1['init_d'].copy(), 'tau': c1['init_tau'].copy()}
    infRes, nll, _ = funs_mod.inference.laplace(exp, par)
    res, nll_o, _ = orc.laplace(Ys, par, c1['binSize'], mode='exact', return_cov=False)
    assert abs(nll - nll_o) <= 1e-9 * abs(nll_o)
    new, _ = funs_mod.learning.updateParams(par, infRes, exp, CdOptimMethod='newton')
    gr0 = orc.mstep_cd_grad(orc.cd_to_vec(par['C'], par['d']), Ys, res['post_mean'], res['post_vsm'], 3, 30)
    gr = orc.mstep_cd_grad(orc.cd_to_vec(new['C'], new['d']), Ys, res['post_mean'], res['post_vsm'], 3, 30)
    assert np.max(np.abs(gr)) <= 1e-8 * np.max(np.abs(gr0))            # (the loud bins put the gradient at the start at ~1e2)
    funs_mod._session.drop_sessions()


def test_generator_writes_two_byte_counts(c1):
    """pgpfa_generate at rates of several hundred spikes per bin: the plane of high bytes appears by itself, the counts come back as
    uint16 and follow the Poisson law of the latents the same call returns (mean and variance per neuron over bins and trials)."""
    from funs import _hip
    q, p, T, R = 6, 2, 64, 40
    rng = np.random.default_rng(5)
    C, tau = 0.2 * rng.standard_normal((q, p)), np.array([0.2, 0.4])
    d = np.log(np.array([3.0, 40.0, 200.0, 300.0, 600.0, 900.0]))
    ctx = _hip.Context(q, p, T, R, 10.0)
    try:
        ctx.set_params(C, d, tau)
        X, Y = ctx.generate(1234)
        assert ctx.info('counts_two_bytes') == 1.0 and Y.dtype == np.uint16 and Y.max() > 255
        rate = np.exp(np.einsum('qk,rkt->rqt', C, X) + d[None, :, None])
        z = (Y - rate) / np.sqrt(rate)
        assert abs(z.mean()) <= 5.0 / np.sqrt(z.size) and abs(z.var() - 1.0) <= 0.05
        # same seed, same draw - now with the plane present from the start
        X2, Y2 = ctx.generate(1234)
        assert np.array_equal(Y, Y2) and np.array_equal(X, X2)
        obj, _, status = ctx.estep_laplace()
        assert np.all(status == 0) and np.isfinite(obj)
    finally:
        ctx.close()


# ---------------------------------------------------------------------------------------------------------------
# lazily rebuilt blocks belong to the trial's own E-step
# ---------------------------------------------------------------------------------------------------------------
def test_rebuilt_blocks_use_the_parameters_of_the_trials_own_estep(c1):
    """Minibatch EM: E-step A on trials 0..9 under theta_1, parameters move, E-step B on trials 10..19 under theta_2.  post_vsmGP
    (rebuilt on demand under the sum-only low-rank plan) and post_cov of a trial of A must be the blocks at theta_1."""
    from funs import _hip
    q, p, T = 30, 3, 100
    th1 = {'C': c1['init_C'], 'd': c1['init_d'], 'tau': c1['init_tau']}
    rng = np.random.default_rng(8)
    th2 = {'C': th1['C'] + 0.1 * rng.standard_normal((q, p)), 'd': th1['d'] - 0.2, 'tau': th1['tau'] * np.array([0.7, 1.3, 0.9])}
    A, B = np.arange(10, dtype=np.int32), np.arange(10, 20, dtype=np.int32)
    ctx = _hip.Context(q, p, T, 20, c1['binSize'])
    try:
        ctx.upload_counts(c1['Y'])
        ctx.set_option('cov_mode', 2)
        ctx.set_params(th1['C'], th1['d'], th1['tau'])
        _, _, st = ctx.estep_laplace(A)
        assert np.all(st == 0) and ctx.info('last_cov_lowrank') == 1.0
        ctx.set_params(th2['C'], th2['d'], th2['tau'])
        _, _, st = ctx.estep_laplace(B)
        assert np.all(st == 0)
        gp3 = ctx.post_vsmgp(np.array([3, 12], dtype=np.int32))
        cov3, cov12 = ctx.post_cov(3), ctx.post_cov(12)
        X = ctx.post_mean()
        # the context is back on theta_2 afterwards
        f2, _ = ctx.laplace_eval(np.array([12], dtype=np.int32), X[12][None])
    finally:
        ctx.close()
    for trial, th, gp, cov in ((3, th1, gp3[0], cov3), (12, th2, gp3[1], cov12)):
        Kinv = np.linalg.inv(orc.make_K(th['tau'], T, c1['binSize']))
        H = orc.nlp_hess(X[trial], c1['Ys'][trial], th['C'], th['d'], Kinv)
        S = np.linalg.inv(H)
        vsmGP, _ = orc.marginal_blocks(S, p, T)
        assert rel(cov, S) <= 1e-8, trial
        assert rel(gp, vsmGP) <= 1e-8, trial
    Kinv2 = np.linalg.inv(orc.make_K(th2['tau'], T, c1['binSize']))
    assert abs(f2[0] - orc.nlp(X[12], c1['Ys'][12], th2['C'], th2['d'], Kinv2)) <= 1e-9 * abs(f2[0])


def test_count_terms_follow_replaced_modes(c1):
    """The hoisted count terms of the (C,d) cost (sum_t y m_t per neuron) are recomputed after pgpfa_set_modes rewrote the modes
    (they used to survive it)."""
    from funs import _hip
    q, p, T, R = 30, 3, 100, 6
    ctx = _hip.Context(q, p, T, R, c1['binSize'])
    try:
        ctx.upload_counts(c1['Y'][:R])
        ctx.set_params(c1['init_C'], c1['init_d'], c1['init_tau'])
        ctx.estep_laplace()
        v = orc.cd_to_vec(c1['init_C'], c1['init_d'])
        ctx.mstep_cd_costgrad(v)                                   # caches the count terms of the E-step's modes
        X = ctx.post_mean()
        vsm = ctx.post_vsm()
        X2 = X + 0.05 * np.random.default_rng(0).standard_normal(X.shape)
        ctx.set_modes(None, X2.reshape(R, -1))
        cost, grad = ctx.mstep_cd_costgrad(v)
        Ys = c1['Ys'][:R]
        assert abs(cost - orc.mstep_cd_cost(v, Ys, list(X2), list(vsm), p, q)) <= 1e-10 * abs(cost)
        assert rel(grad, orc.mstep_cd_grad(v, Ys, list(X2), list(vsm), p, q)) <= 1e-9
    finally:
        ctx.close()


# ---------------------------------------------------------------------------------------------------------------
# workspace arena: grows with the ranks, results unchanged
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('vmm', [1, 0])
def test_workspace_grows_with_the_ranks(vmm):
    """Low-rank plan at config-2 dimensions: an E-step at long timescales, then at short ones (ranks up by more than the plan's
    head-room: the workspace is re-planned and the arena grows - by mapping more memory behind the same address range, or by
    re-allocation when the virtual-memory calls are switched off); modes and blocks against the dense engine each time."""
    from funs import _hip
    q, p, T, R = 100, 5, 200, 24
    _, Ys, _ = orc.synth_dataset(q, p, T, R, seed=3)
    Y = np.stack(Ys).astype(np.uint8)
    rng = np.random.default_rng(3)
    C, d = rng.random((q, p)) - 0.5, -2.0 * rng.random(q) - 1.0
    out = {}
    for cov_mode in (2, 1):
        ctx = _hip.Context(q, p, T, R, 10.0)
        try:
            ctx.set_option('workspace_vmm', vmm)
            ctx.set_option('workspace_granule_mb', 64)          # (1 GiB by default: this problem is smaller than one granule)
            ctx.upload_counts(Y)
            ctx.set_option('cov_mode', cov_mode)
            res = []
            for tau in (np.full(p, 0.6), np.full(p, 0.08), np.full(p, 0.03)):
                ctx.set_params(C, d, tau)
                obj, _, st = ctx.estep_laplace()
                assert np.all(st == 0) and ctx.info('last_cov_lowrank') == float(cov_mode == 2)
                ctx.mstep_precomp()
                res.append((obj, ctx.post_mean(), ctx.post_vsm(), ctx.pautosum(), ctx.info('lowrank_rtot'), ctx.info('arena_bytes')))
            out[cov_mode] = res
        finally:
            ctx.close()
    ranks = [r[4] for r in out[2]]
    arena = [r[5] for r in out[2]]
    assert ranks[2] > 2.0 * ranks[0] and arena[2] >= arena[1] >= arena[0] and arena[2] > arena[0]
    for a, b in zip(out[2], out[1]):
        assert abs(a[0] - b[0]) <= 1e-10 * abs(b[0]) and np.max(np.abs(a[1] - b[1])) <= 1e-8
        assert rel(a[2], b[2]) <= 1e-8 and rel(a[3], b[3]) <= 1e-8


# ---------------------------------------------------------------------------------------------------------------
# BASELINE config 5: the variational E-step at 500 neurons x 20 latents x 1000 bins
# ---------------------------------------------------------------------------------------------------------------
def _dense_dual_reference(C, d, tau, y, lam, T, bin_ms):
    """Plain numpy on the dense matrices of one trial (n = p T): dual cost and gradient of inference.py:196-219 with the
    jittered covariance of inference.py:188-191, posterior mean of inference.py:193-194 and the per-bin blocks of that covariance."""
    import scipy.linalg as sl
    q, p = C.shape
    n = p * T
    K = orc.make_K(tau, T, bin_ms)                           # [p][T][T]
    lam2, y2 = lam.reshape(q, T), y.reshape(q, T)
    H = np.zeros((n, n))
    for k in range(p):
        H[k * T:(k + 1) * T, k * T:(k + 1) * T] = np.linalg.inv(K[k])
    W = np.einsum('nk,nt,nl->tkl', C, lam2, C)               # [T][p][p]
    ar = np.arange(T)
    for k in range(p):
        for l in range(p):
            H[k * T + ar, l * T + ar] += W[:, k, l]
    H[np.arange(n), np.arange(n)] *= (1.0 + 1e-6)            # inference.py:190
    L, info = sl.lapack.dpotrf(H, lower=1, overwrite_a=1)
    assert info == 0
    logdet = 2.0 * np.sum(np.log(np.diag(L)))
    S, info = sl.lapack.dpotri(L, lower=1, overwrite_c=1)
    assert info == 0
    blocks = np.empty((T, p, p))
    for k in range(p):
        for l in range(k + 1):                               # lower triangle of S is valid
            blocks[:, k, l] = S[k * T + ar, l * T + ar]
            blocks[:, l, k] = blocks[:, k, l]
    del S, L, H
    lmy = lam2 - y2
    v = C.T @ lmy                                            # [p][T]
    Kv = np.einsum('kts,ks->kt', K, v)
    quad = np.einsum('nk,tkl,nl->nt', C, blocks, C)
    cost = 0.5 * np.sum(v * Kv) - d @ lmy.sum(axis=1) - 0.5 * logdet + np.sum(lam2 * (np.log(lam2) - 1.0))
    grad = C @ Kv - d[:, None] + np.log(lam2) - 0.5 * quad
    return cost, grad.reshape(-1), -Kv, blocks


@pytest.mark.timeout(3000)
def test_config5_variational_estep_to_convergence(funs_mod, monkeypatch):
    """BASELINE config 5 as a workload: inference.dualVariational (inference.py:259-432) at 500 neurons, 20 latents, 1000 bins on
    4 trials, run to the optimum on the device by the variational fixed point (the default solver since round 4; round 3 ran the device
    L-BFGS here: 3000-6000 iterations per trial to scipy's decrease test) - in FP64 and in mixed precision (DUAL_F32) - through the
    low-rank engine (default at this size).  Checked against plain numpy on the dense 20 000 x 20 000 matrices of trial 0: dual cost
    1e-8, the reference's dual gradient (inference.py:215-219) BELOW 1e-6 in the max-norm in numpy's own arithmetic - the optimum is the
    zero of the reference's function, not of ours -, device gradient vs numpy 1e-8 absolute, covariance blocks 1e-7, posterior mean 1e-9;
    the structured posterior-mean identity post_mean = -K C_big (lambda - y) on every trial; a few L-BFGS iterations from the optimum do
    not lower the dual (1e-10 rel); mixed vs FP64: cost 1e-5 at the same lambda, the two optima within 1e-5 of each other in the bound
    (round 3: 2e-3, what two decrease-test stops left); a warm restart (prevOptimRes) settles in one pass."""
    import bench
    inf = funs_mod.inference
    q, p, T, R = 500, 20, 1000, 4
    true, Ys = bench.synth_shard(q, p, T, R, 12, 0)
    par = {'C': true['C'], 'd': true['d'], 'tau': np.linspace(0.1, 0.5, p)}
    exp = Experiment([y.astype(float) for y in Ys], 10.0)
    m = q * T
    assert inf.DUAL_SOLVER == 'fixedpoint'
    out = {}
    for f32 in (False, True):
        monkeypatch.setattr(inf, 'DUAL_F32', f32)
        funs_mod._session.drop_sessions()
        infRes, nll, vlb, opt = inf.dualVariational(exp, dict(par), optimizeLogLambda=True)
        sess = infRes.session
        assert sess.ctx.info('plan_lowrank') == 1.0
        lam = np.exp(np.stack(opt))
        idx = np.arange(R, dtype=np.int32)
        sess.ctx.set_option('dual_f32', 0)
        cost, grad = sess.ctx.dual_costgrad_batch(idx, lam)           # FP64 evaluation at either run's optimum
        out[f32] = dict(nll=nll, vlb=vlb, lam=lam, iters=infRes.dual_iterations.copy(), cost=cost, grad=grad,
                        pm=np.stack([infRes['post_mean'][r] for r in range(R)]), vsm0=infRes['post_vsm'][0].copy())
        if f32:
            # the mixed-precision evaluation at the FP64 run's optimum (same lambda: what the precision itself changes)
            sess.ctx.set_option('dual_f32', 1)
            out['mixed_at_f64_opt'] = sess.ctx.dual_costgrad_batch(idx, out[False]['lam'])
            sess.ctx.set_option('dual_f32', 0)
        else:
            # a few quasi-Newton iterations from the optimum find nothing lower
            rho_l, fopt_l, it_l = sess.ctx.dual_lbfgs(idx, np.log(lam), max_iter=5)
            out['lbfgs_from_opt'] = fopt_l
            # warm restart: one pass
            ir_w, nll_w, vlb_w, _ = inf.dualVariational(exp, dict(par), optimizeLogLambda=True, prevOptimRes=opt)
            out['warm'] = (ir_w.dual_iterations.copy(), vlb_w, nll_w)
        print('config 5, %s: fixed-point passes %s, bound %.6f, nll %.6f, max |dual gradient| %.2e' % (
            'mixed' if f32 else 'f64', out[f32]['iters'], vlb, nll, np.max(np.abs(grad))))
    funs_mod._session.drop_sessions()
    a, b = out[False], out[True]
    assert np.all(a['iters'] >= 2) and np.all(a['iters'] <= 12) and np.all(b['iters'] <= 12)
    assert np.max(np.abs(a['grad'])) <= 1e-6 and np.max(np.abs(b['grad'])) <= 1e-5
    assert np.all(out['lbfgs_from_opt'] >= a['cost'] - 1e-10 * np.abs(a['cost']))
    assert np.all(out['warm'][0] == 1) and abs(out['warm'][1] - a['vlb']) <= 1e-9 * abs(a['vlb'])
    # mixed precision: the same function (cost 1e-5 rel, gradient 1e-3 of the FP64 gradient's scale at a generic point is covered by
    # test_dual_mixed_precision_at_config5_dimensions; here at the optimum, absolute) and the same optimum
    cm, gm = out['mixed_at_f64_opt']
    assert np.max(np.abs(cm - a['cost']) / np.abs(a['cost'])) <= 1e-5
    assert np.max(np.abs(gm - a['grad'])) <= 1e-4
    assert abs(b['vlb'] - a['vlb']) <= 1e-5 * abs(a['vlb']) and abs(b['nll'] - a['nll']) <= 1e-5 * abs(a['nll'])
    # structured identity on every trial: post_mean = -K C_big (lambda - y)  (inference.py:194)
    K = orc.make_K(par['tau'], T, 10.0)
    for r in range(R):
        v = par['C'].T @ (a['lam'][r].reshape(q, T) - Ys[r])
        assert rel(a['pm'][r], -np.einsum('kts,ks->kt', K, v)) <= 1e-9
    # dense numpy, trial 0, at the FP64 optimum
    cost, grad, mean, blocks = _dense_dual_reference(par['C'], par['d'], par['tau'], Ys[0].astype(float).reshape(-1), a['lam'][0], T, 10.0)
    print('config 5 trial 0 vs dense numpy: cost %.3e, numpy max |grad| %.3e, device - numpy grad %.3e, blocks %.3e (f64) / %.3e (mixed, at its own optimum vs f64)'
          % (abs(a['cost'][0] - cost) / abs(cost), np.max(np.abs(grad)), np.max(np.abs(a['grad'][0] - grad)), rel(a['vsm0'], blocks), rel(b['vsm0'], a['vsm0'])))
    assert abs(a['cost'][0] - cost) <= 1e-8 * abs(cost)
    assert np.max(np.abs(grad)) <= 1e-6
    assert np.max(np.abs(a['grad'][0] - grad)) <= 1e-8
    assert rel(a['vsm0'], blocks) <= 1e-7
    assert rel(a['pm'][0], mean) <= 1e-9


# ---------------------------------------------------------------------------------------------------------------
# sum over trials of the covariance blocks by the exact split form (csrc/split.h)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dims', ['c1', 'c2', 'c3'])
def test_split_accumulation_matches_the_fp64_product(c1, dims):
    """PautoSum through the split form (first term collapsed to r_k x r_k, cross term FP64 with a single-precision operand, the
    (eps ||Wt||)^2 term on the FP16 matrix cores in two halves) against the full-width FP64 product of the same engine, 1e-9 of the
    largest entry, and post_vsm identical; a chunk whose correction is too large for the precision argument (forced here with the
    threshold) takes the FP64 product; a population firing 12 x faster (eps ||Wt|| several times larger) still agrees to 1e-9 under the
    default threshold."""
    from funs import _hip
    import bench
    if dims == 'c1':
        q, p, T, R = 30, 3, 100, 20
        Y = c1['Y']
        par = {'C': c1['init_C'], 'd': c1['init_d'], 'tau': c1['init_tau']}
    else:
        q, p, T, R = (100, 5, 200, 40) if dims == 'c2' else (200, 10, 500, 48)
        true, Ys = bench.synth_shard(q, p, T, R, 12, 0)
        Y = np.stack(Ys)
        par = {'C': true['C'], 'd': true['d'], 'tau': np.linspace(0.1, 0.5, p)}
    out = {}
    for mode in ('fp64', 'split', 'forced_fallback', 'loud', 'very_loud'):
        ctx = _hip.Context(q, p, T, R, 10.0)
        try:
            ctx.upload_counts(Y)
            ctx.set_option('cov_mode', 2)
            ctx.set_option('split_cov', 0 if mode == 'fp64' else 1)
            if mode == 'forced_fallback':
                ctx.set_option('split_max_norm', 1e-9)
            elif 'PGPFA_SPLIT_MAX' in __import__('os').environ:
                ctx.set_option('split_max_norm', float(__import__('os').environ['PGPFA_SPLIT_MAX']))
            d = par['d']
            if mode in ('loud', 'very_loud'):
                # a population firing 12 x (50 x) faster, counts redrawn at those rates: eps ||Wt|| several times the usual (beyond the guard)
                d = par['d'] + (2.5 if mode == 'loud' else 3.9)
                rng = np.random.default_rng(5)
                ctx.upload_counts(np.minimum(rng.poisson(np.exp(d)[None, :, None] * np.ones((R, 1, T))), 60000).astype(np.uint16))
            ctx.set_option('measure_mix', 1)
            ctx.set_params(par['C'], d, par['tau'])
            obj, _, st = ctx.estep_laplace()
            assert np.all(st == 0) and ctx.info('last_cov_lowrank') == 1.0
            ctx.mstep_precomp()
            out[mode] = (ctx.pautosum(), ctx.post_vsm(), ctx.info('last_split_cov'), ctx.info('last_eps_wt_rms'), obj)
            if mode == 'loud':
                ctx.set_option('split_cov', 0)
                ctx.estep_laplace()
                ctx.mstep_precomp()
                out['loud_fp64'] = (ctx.pautosum(), ctx.post_vsm())
        finally:
            ctx.close()
    ref = out['fp64']
    print('%s: rms eps||Wt|| = %.2e (loud %.2e, split used %d); split vs FP64 product: PautoSum %.2e, post_vsm %.2e; loud %.2e'
          % (dims, out['split'][3], out['loud'][3], out['loud'][2], rel(out['split'][0], ref[0]), rel(out['split'][1], ref[1]),
             rel(out['loud'][0], out['loud_fp64'][0])))
    assert out['split'][2] == 1.0 and out['fp64'][2] == 0.0 and out['forced_fallback'][2] == 0.0
    assert rel(out['split'][0], ref[0]) <= 1e-9 and rel(out['split'][1], ref[1]) <= 1e-12
    assert rel(out['forced_fallback'][0], ref[0]) <= 1e-13
    assert rel(out['loud'][0], out['loud_fp64'][0]) <= 1e-9 and rel(out['loud'][1], out['loud_fp64'][1]) <= 1e-12
    if dims != 'c1':
        # rates high enough for the guard (rms of eps ||Wt|| above 0.07) to keep the FP64 product by itself
        assert out['very_loud'][3] > 0.07 and out['very_loud'][2] == 0.0 and np.all(np.isfinite(out['very_loud'][0]))


@pytest.mark.parametrize('q,p,T,R', [(30, 3, 100, 4), (200, 10, 77, 3), (37, 7, 130, 2), (17, 1, 5, 2), (129, 10, 64, 2)])
def test_cd_newton_pass_matrix_core_form(q, p, T, R):
    """The two-stage matrix-core Newton pass of the (C,d) M-step (mstep_cd_hess_mfma_kernel) against the vector kernel on the same
    posterior and against plain numpy: per-neuron costs 1e-11, Newton step H_n^-1 g_n and decrement g_n^T H_n^-1 g_n 1e-9 (they involve
    every entry of the gradient and of the packed Hessian of learning.py:20-91's cost, restated below per neuron)."""
    from funs import _hip
    _, Ys, _ = orc.synth_dataset(q, p, T, R, seed=7 * q + T, dOffset=0.0)
    Y = np.stack(Ys).astype(np.uint8)
    rng = np.random.default_rng(q + T)
    par = {'C': 0.3 * rng.standard_normal((q, p)) / np.sqrt(p), 'd': np.log(Y.mean(axis=(0, 2)) + 0.1), 'tau': 0.05 + 0.2 * rng.random(p)}
    ctx = _hip.Context(q, p, T, R, 10.0)
    try:
        ctx.upload_counts(Y)
        ctx.set_params(par['C'], par['d'], par['tau'])
        _, _, status = ctx.estep_laplace()
        assert np.all(status == 0)
        pm, vs = ctx.post_mean(), ctx.post_vsm()
        v = orc.cd_to_vec(par['C'], par['d']) + 0.01 * rng.standard_normal(q * (p + 1))
        out = {}
        for form in (1, 0):
            ctx.set_option('cd_hess_mfma', form)
            out[form] = ctx.mstep_cd_newton_pass(v)
        for a, b in zip(out[1], out[0]):
            assert rel(a, b) <= 1e-10
        # plain numpy: theta_n = (c_n, d_n), w = m_t + V_t c_n, yhat = exp(d_n + c_n.m_t + c_n^T V_t c_n / 2)
        vv = v.reshape(p + 1, q)
        Cn, dn = vv[:p].T, vv[p]
        cost_ref, delta_ref, dec_ref = np.zeros(q), np.zeros((p + 1, q)), np.zeros(q)
        for n in range(q):
            g, H, cst = np.zeros(p + 1), np.zeros((p + 1, p + 1)), 0.0
            for r in range(R):
                m = pm[r].reshape(p, T)
                Vc = vs[r] @ Cn[n]                                   # [T][p]
                w1 = np.concatenate([m.T + Vc, np.ones((T, 1))], axis=1)
                hh = dn[n] + Cn[n] @ m
                yh = np.exp(hh + 0.5 * Vc @ Cn[n])
                y = Ys[r][n].astype(float)
                cst += np.sum(y * hh - yh)
                g += yh @ w1 - np.concatenate([m @ y, [y.sum()]])
                H += (w1 * yh[:, None]).T @ w1
                H[:p, :p] += np.einsum('t,tij->ij', yh, vs[r])
            cost_ref[n] = -cst / R
            delta_ref[:, n] = -np.linalg.solve(H / R, g / R)
            dec_ref[n] = (g / R) @ np.linalg.solve(H / R, g / R)
        cost_n, delta, dec = out[1]
        assert rel(cost_n, cost_ref) <= 1e-11
        assert rel(delta.reshape(p + 1, q), delta_ref) <= 1e-9 and rel(dec, dec_ref) <= 1e-9
    finally:
        ctx.close()
