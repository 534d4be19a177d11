"""Round-2 GPU parity tests: config-3-size fixture from the reference, stochastic EM at config-3 dimensions (config 4's
minibatch loop), the reference's callbacks by name, regressions for the workspace re-plan / stale-view / duplicate-index
findings, and the 2-rank RCCL path (skipped on a 1-GPU box).  Everything goes through the drop-in `funs` surface or the
C-ABI wrapper; the oracle and the golden vectors are the checkers."""
import os
import subprocess
import sys
import warnings

import numpy as np
import pytest

from conftest import Experiment, ROOT, load_golden
from oracle import pgpfa_oracle as orc

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, np.max(np.abs(b)))


@pytest.fixture(scope='module')
def funs_mod():
    import funs
    return funs


# ---------------------------------------------------------------------------------------------------------------
# config 3 pinned to the reference
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('cov_mode', [0, 1])
def test_c3_size_spot_check_vs_reference(funs_mod, cov_mode):
    """One trial at config 3's dimensions against c3_spot.npz (the real reference's inference.laplace, then polished on its own
    callbacks): post_mean 1e-7 vs the polished mode, post_vsm / post_vsmGP 1e-8 rel vs the inverse of the reference Hessian there;
    vs the reference's raw early-stopped answer 5e-3 / 1e-4 (its slack here: 2.7e-5 in the mode).  cov_mode 0 = the plan the
    product picks (low-rank engine at these timescales), 1 = dense engine."""
    from funs import _hip
    g = load_golden('c3_spot.npz')
    q, p, T = 200, 10, 500
    ctx = _hip.Context(q, p, T, 1, float(g['binSize']))
    try:
        ctx.upload_counts(g['Y'])
        ctx.set_option('cov_mode', cov_mode)
        ctx.set_option('keep_trial_vsmgp', 1)
        ctx.set_params(g['init_C'], g['init_d'], g['init_tau'])
        obj, iters, status = ctx.estep_laplace()
        assert np.all(status == 0)
        assert ctx.info('last_cov_lowrank') == (1.0 if cov_mode == 0 else 0.0)
        X = ctx.post_mean()[0]
        assert np.max(np.abs(X.reshape(-1) - g['polished'])) <= 1e-7
        assert abs(obj - float(g['polished_f'])) <= 1e-10 * abs(float(g['polished_f']))
        assert rel(ctx.post_vsm()[0], g['polished_vsm']) <= 1e-8
        G = ctx.post_vsmgp()[0]                                           # (T, T, p)
        assert rel(np.stack([np.diag(G[:, :, k]) for k in range(p)]), g['polished_vsmGP_diag']) <= 1e-8
        assert rel(G[::50, :, :], g['polished_vsmGP_rows']) <= 1e-8
        # the reference's own numbers
        assert np.max(np.abs(X - g['post_mean'])) <= 5e-3
        assert abs(-obj - float(g['nll'])) <= 1e-4
        assert rel(ctx.post_vsm()[0], g['post_vsm']) <= 1e-5
    finally:
        ctx.close()


# ---------------------------------------------------------------------------------------------------------------
# config 4's loop at config 3's dimensions
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.timeout(3000)
@pytest.mark.parametrize('tau_method', ['TNC', 'lockstep'])
def test_online_em_at_config3_dimensions(funs_mod, tau_method):
    """Stochastic EM as BASELINE config 4 runs it on one GPU at its REAL size: 8192 trials resident (the 0.8-GB count tensor, trial indices
    far above 2047 in every minibatch), minibatches of 1024, three iterations of the
    engine's 'diag' loop (engine.py:288-448).  Size-independent checks, each against the reference's arithmetic restated by the
    oracle / plain numpy: (1) the minibatch index stream is the reference's RNG stream (util.py:459-473); (2) every returned
    mode of every minibatch is a stationary point of the reference's log-posterior (inference.py:34-48) under that iteration's
    parameters, and the reported nPLL is the mean of negLogPosteriorUnNorm there; (3) the new (C,d) make the oracle's gradient
    of MStepObservationCostWithPrior (learning.py:488-534) vanish on the minibatch's posterior; (4) each new timescale is a
    the stopping point of the reference's TNC call on
    its prior-regularised cost and (inconsistent, learning.py:733-734) gradient, restated by the oracle on the device's PautoSum;
    (5) the configuration's 8-way split of a minibatch: the eight contiguous slices the ranks of an 8-GPU job would take (shard_slice) run
    one after the other on this GPU give the same modes (1e-8) and their PautoSum / nPLL contributions ADD UP to the whole minibatch's
    (1e-9) - what the RCCL all-reduce of the M-step statistics sums.
    tau_method: the engine's default 'TNC' (the reference's scipy call per latent on device evaluations) and, round 6, 'lockstep' (the zero of
    the reference's regularised gradient for all latents together) - check (4) holds both to the oracle's TNC stopping point."""
    import bench
    q, p, T, Rres, batch, iters = 200, 10, 500, 8192, 1024, 3
    true, Ys = bench.synth_shard(q, p, T, Rres, 12, 0)
    exp = bench.Shard(Ys, 10.0)
    exp._pgpfa_local_shard = False
    np.random.seed(0)
    init = funs_mod.util.initializeParams(p, q, exp)
    init = {k: np.real(np.asarray(v)).astype(np.float64) for k, v in init.items()}
    seen = {}

    # spy on the M-step to capture each iteration's posterior while it is resident
    real_update = funs_mod.learning.updateParamsWithPrior

    def spy(oldParams, infRes, experiment, *a, **kw):
        out = real_update(oldParams, infRes, experiment, *a, **kw)
        n = len(seen)
        pick = np.array([0, 511, 1023])
        seen[n] = {'idx': np.asarray(experiment.batchTrIdx).copy(), 'old': {k: np.array(v) for k, v in oldParams.items()},
                   'new': {k: np.array(v) for k, v in out[0].items()}, 'pick': pick,
                   'pm_all': infRes.session.ctx.post_mean(infRes.trial_idx), 'pv_all': infRes.session.ctx.post_vsm(infRes.trial_idx),
                   'P': infRes.session.ctx.pautosum(), 'status': infRes.newton_status.copy()}
        return out
    funs_mod.learning.updateParamsWithPrior = spy
    try:
        np.random.seed(1)
        fit = funs_mod.engine.PPGPFAfit(exp, initParams=init, inferenceMethod='laplace', EMmode='Online', maxEMiter=iters, batchSize=batch,
                                        onlineParamUpdateMethod='diag', CdOptimMethod='newton', tauOptimMethod=tau_method, quiet=True)
    finally:
        funs_mod.learning.updateParamsWithPrior = real_update
    # (1) index stream
    np.random.seed(1)
    for n in range(iters):
        assert np.array_equal(seen[n]['idx'], np.random.choice(Rres, batch, replace=False))
    Yf = [y.astype(np.float64) for y in Ys]
    for n in range(iters):
        s = seen[n]
        par, idx = s['old'], s['idx']
        assert np.all(s['status'] == 0)
        Kinv = np.linalg.inv(orc.make_K(par['tau'], T, 10.0))
        # (2) stationarity of all 1024 modes + objective
        f = 0.0
        worst = 0.0
        for j, r in enumerate(idx):
            X = s['pm_all'][j]
            h = par['C'] @ X + par['d'][:, None]
            e = np.exp(h)
            KX = np.einsum('kts,ks->kt', Kinv, X)
            worst = max(worst, float(np.max(np.abs(par['C'].T @ (e - Ys[r]) + KX))))
            f += np.sum(e) - np.sum(Ys[r] * h) + 0.5 * np.sum(X * KX)
        assert worst <= 1e-6
        assert abs(fit.posteriorLikelihood[n] + f / batch) <= 1e-9 * abs(f / batch)
        # (3) (C,d): gradient of the reference's prior-regularised cost at the new point, on the whole minibatch
        step = 1.0 / (n + 1) ** 0.75
        old_vec = orc.cd_to_vec(par['C'], par['d'])
        inv_prior = -np.eye(old_vec.size) / step ** 2
        new_vec = orc.cd_to_vec(s['new']['C'], s['new']['d'])
        gcd = orc.mstep_cd_grad_prior(new_vec, old_vec, inv_prior, [Yf[r] for r in idx], list(s['pm_all']), list(s['pv_all']), p, q)
        assert np.max(np.abs(gcd)) <= 2e-6          # (the device Newton stops at a predicted parameter error of 1e-10; Hessian scale ~1e3)
        # (4) timescales: the reference's own optimiser call (learning.py:819-825: TNC, gtol 1e-10, on its cost and its inconsistent
        # gradient) restated by the oracle on the device's PautoSum must stop where the product stopped.  TNC ends on rounding
        # noise of a cost of order 1e6 here, so two correct evaluations agree to ~1e-4 in tau (config 1: 3.5e-5 measured)
        import scipy.optimize as op
        for k in (0, 5, 9):
            tb = par['tau'][k] * 100.0
            out = op.minimize(orc.tau_cost_prior, np.log(1.0 / tb ** 2), args=(s['P'][k], batch, 10.0, par['tau'][k], step),
                              jac=orc.tau_grad_prior, options={'disp': False, 'gtol': 1e-10}, method='TNC')
            tau_o = (1.0 / np.exp(out.x[0])) ** 0.5 / 100.0
            assert abs(s['new']['tau'][k] - tau_o) <= 1e-3 * tau_o
    assert len(fit.paramSeq) == iters + 1 and np.all(np.isfinite(fit.optimParams['tau']))
    assert max(int(np.max(seen[n]['idx'])) for n in range(iters)) > 8000 and min(int(np.min(seen[n]['idx'])) for n in range(iters)) < 100
    # (5) the last minibatch again under its own parameters: whole, then as the eight slices of an 8-rank job
    from funs import _session
    sess, _ = _session.session_for(exp, p)
    ctx = sess.ctx
    s = seen[iters - 1]
    ctx.set_params(s['old']['C'], s['old']['d'], s['old']['tau'])
    idx = s['idx'].astype(np.int32)
    obj_all, _, st = ctx.estep_laplace(idx)
    assert np.all(st == 0)
    ctx.mstep_precomp()
    P_all, pm_all = ctx.pautosum(), ctx.post_mean(idx)
    assert np.max(np.abs(pm_all - s['pm_all'])) <= 1e-8
    P_sum, obj_sum = np.zeros_like(P_all), 0.0
    for r in range(8):
        lo, hi = _session.shard_slice(batch, r, 8)
        obj_r, _, st = ctx.estep_laplace(idx[lo:hi])
        assert np.all(st == 0) and hi - lo == 128
        ctx.mstep_precomp()
        P_sum += ctx.pautosum()
        obj_sum += obj_r
        assert np.max(np.abs(ctx.post_mean(idx[lo:hi]) - pm_all[lo:hi])) <= 1e-8
    assert np.max(np.abs(P_sum - P_all)) <= 1e-9 * np.max(np.abs(P_all)) and abs(obj_sum - obj_all) <= 1e-10 * abs(obj_all)
    _session.drop_sessions()


# ---------------------------------------------------------------------------------------------------------------
# the reference's callbacks by name and positional signature
# ---------------------------------------------------------------------------------------------------------------
def test_named_laplace_callbacks_vs_golden(funs_mod, c1):
    """negLogPosteriorUnNorm / _grad / _hess (inference.py:12-65) called exactly as the reference's callers do - with the
    Kronecker big matrices - against the values captured from the reference: 1e-9 rel."""
    g = load_golden('c1_callbacks.npz')
    inf = funs_mod.inference
    p, q, T = 3, 30, 100
    K = g['K']
    K_big = np.zeros((p * T, p * T))
    for k in range(p):
        K_big[k * T:(k + 1) * T, k * T:(k + 1) * T] = K[k]
    K_bigInv = np.linalg.inv(K_big)
    C_big = np.kron(c1['init_C'], np.eye(T)).T                     # util.py:594-597
    d_big = np.kron(c1['init_d'], np.ones(T)).T
    ybar = c1['Ys'][0].reshape(-1)
    x = g['xprobe']
    f = inf.negLogPosteriorUnNorm(x, ybar, C_big, d_big, K_bigInv, p, q)
    assert abs(f - float(g['f'])) <= 1e-9 * abs(float(g['f']))
    assert rel(inf.negLogPosteriorUnNorm_grad(x, ybar, C_big, d_big, K_bigInv, p, q), g['g']) <= 1e-9
    assert rel(inf.negLogPosteriorUnNorm_hess(x, ybar, C_big, d_big, K_bigInv, p, q), g['H']) <= 1e-9
    # another trial's counts through the same matrices (mcmc.py:25 calls it in a loop), and the product's own builders
    y1 = c1['Ys'][1].reshape(-1)
    f1 = inf.negLogPosteriorUnNorm(x, y1, C_big, d_big, K_bigInv, p, q)
    assert abs(f1 - orc.nlp_big(x, y1, C_big, d_big, K_bigInv)) <= 1e-9 * abs(f1)
    Cb2, db2 = funs_mod.util.makeCd_big(c1['init'], T)
    Kb2, _ = funs_mod.util.makeK_big(dict(c1['init']), T * c1['binSize'], c1['binSize'])
    f2 = inf.negLogPosteriorUnNorm(x, ybar, Cb2, db2, np.linalg.inv(Kb2), p, q)
    assert abs(f2 - float(g['f'])) <= 1e-9 * abs(float(g['f']))
    # matrices that are not such structures are refused, not silently mis-evaluated
    bad = K_bigInv.copy()
    bad[0, p * T - 1] = bad[p * T - 1, 0] = 0.5
    with pytest.raises(ValueError):
        inf.negLogPosteriorUnNorm(x, ybar, C_big, d_big, bad, p, q)
    with pytest.raises(ValueError):
        inf.negLogPosteriorUnNorm(x, ybar, C_big + 0.1, d_big, K_bigInv, p, q)


@pytest.mark.parametrize('engine', ['dense', 'lowrank'])
def test_named_dual_callbacks_vs_golden(funs_mod, engine, monkeypatch):
    """(both covariance engines: the low-rank one carries the reference's jitter in the per-bin blocks)
    dualProblem / dualProblem_grad / dualProblemRho(_grad) / VIPostCov / VIPostMean (inference.py:188-256) by the
    reference's signatures on the variational toy problem (20 neurons, 2 latents, T = 50): dual cost / gradient against the values
    captured from the reference 1e-9, posterior mean / covariance / precision against the oracle's big-matrix restatement 1e-9."""
    g = load_golden('var_toy.npz')
    inf = funs_mod.inference
    monkeypatch.setattr(inf, 'COV_MODE', 2 if engine == 'lowrank' else 1)
    while inf._BIG_CACHE:
        inf._BIG_CACHE.pop()[5].close()
    q, p, T = 20, 2, 50
    par = {'C': g['init_C'], 'd': g['init_d'], 'tau': g['init_tau']}
    K = orc.make_K(par['tau'], T, float(g['binSize']))
    K_big = orc.make_K_big(K)
    K_bigInv = np.linalg.inv(K_big)
    C_big, d_big = orc.make_Cd_big(par['C'], par['d'], T)
    lam = g['lam_probe']
    ybar = g['Y'][0].reshape(-1).astype(float)
    cost = inf.dualProblem(lam, ybar, C_big, K_big, K_bigInv, d_big)
    assert abs(cost - float(g['dual_cost'])) <= 1e-9 * abs(float(g['dual_cost']))
    assert rel(inf.dualProblem_grad(lam, ybar, C_big, K_big, K_bigInv, d_big), g['dual_grad']) <= 1e-9
    rho = np.log(lam)
    assert abs(inf.dualProblemRho(rho, ybar, C_big, K_big, K_bigInv, d_big) - float(g['dual_cost'])) <= 1e-9 * abs(float(g['dual_cost']))
    assert rel(inf.dualProblemRho_grad(rho, ybar, C_big, K_big, K_bigInv, d_big), g['dual_grad'] * lam) <= 1e-9
    cov, prec = inf.VIPostCov(K_bigInv, C_big, lam)
    cov_o, prec_o = orc.vi_post_cov(K_bigInv, C_big, lam)
    assert rel(prec, prec_o) <= 1e-9 and rel(cov, cov_o) <= 1e-8
    assert rel(inf.VIPostMean(K_big, C_big, ybar, lam), orc.vi_post_mean(K_big, C_big, ybar, lam)) <= 1e-9
    while inf._BIG_CACHE:
        inf._BIG_CACHE.pop()[5].close()


# ---------------------------------------------------------------------------------------------------------------
# regressions for the round-1 findings
# ---------------------------------------------------------------------------------------------------------------
def test_per_trial_vsmgp_survives_workspace_replan(c1):
    """The per-trial post_vsmGP buffer is allocated lazily in the middle of an E-step; a later E-step over a longer trial list
    re-plans (frees and re-carves) the chunk workspace.  The buffer must survive that: 5-trial minibatch, then all 20 trials,
    then post_vsmGP of both against the oracle."""
    from funs import _hip
    ctx = _hip.Context(30, 3, 100, 20, c1['binSize'])
    try:
        ctx.upload_counts(c1['Y'])
        ctx.set_option('keep_trial_vsmgp', 1)
        ctx.set_params(c1['init_C'], c1['init_d'], c1['init_tau'])
        mini = np.array([3, 7, 11, 0, 19], dtype=np.int32)
        _, _, st = ctx.estep_laplace(mini)
        assert np.all(st == 0)
        chunk_small = ctx.info('chunk_trials')
        gp_mini = ctx.post_vsmgp(mini)
        _, _, st = ctx.estep_laplace()
        assert np.all(st == 0) and ctx.info('chunk_trials') >= chunk_small
        gp_all = ctx.post_vsmgp()
        res, _, _ = orc.laplace(c1['Ys'], c1['init'], c1['binSize'], mode='exact', return_cov=False)
        for r in range(20):
            assert rel(gp_all[r], res['post_vsmGP'][r]) <= 1e-8
        for j, r in enumerate(mini):
            assert rel(gp_mini[j], res['post_vsmGP'][r]) <= 1e-8
        # and the other way round: dense plan forced, then back
        ctx.set_option('cov_mode', 1)
        _, _, st = ctx.estep_laplace(mini, warm_start=True)
        ctx.set_option('cov_mode', 0)
        _, _, st2 = ctx.estep_laplace()
        assert np.all(st == 0) and np.all(st2 == 0)
        ctx.mstep_precomp()
        P, n = orc.make_precomp(res)
        assert rel(ctx.pautosum(), P) <= 1e-8
        assert rel(ctx.post_vsmgp(np.array([5], dtype=np.int32))[0], res['post_vsmGP'][5]) <= 1e-8
    finally:
        ctx.close()


def test_duplicate_and_oversized_trial_lists_are_rejected(c1):
    from funs import _hip
    ctx = _hip.Context(30, 3, 100, 4, c1['binSize'])
    try:
        ctx.upload_counts(c1['Y'][:4])
        ctx.set_params(c1['init_C'], c1['init_d'], c1['init_tau'])
        with pytest.raises(_hip.HipBackendError, match='twice'):
            ctx.estep_laplace(np.array([0, 1, 1], dtype=np.int32))
        with pytest.raises(_hip.HipBackendError):
            ctx.estep_laplace(np.array([0, 1, 2, 3, 0], dtype=np.int32))
        with pytest.raises(_hip.HipBackendError, match='out of range'):
            ctx.estep_laplace(np.array([0, 4], dtype=np.int32))
        with pytest.raises(_hip.HipBackendError, match='twice'):
            ctx.set_modes(np.array([2, 2], dtype=np.int32), np.zeros((2, 300)))
        _, _, st = ctx.estep_laplace(np.array([3, 1], dtype=np.int32))
        assert np.all(st == 0)
    finally:
        ctx.close()


def test_superseded_infres_views_raise_and_snapshots_survive(funs_mod, c1):
    exp = Experiment(c1['Ys'][:6], c1['binSize'])
    par = {k: v.copy() for k, v in c1['init'].items()}
    res1, _, opt1 = funs_mod.inference.laplace(exp, dict(par))
    m0 = res1['post_mean'][0].copy()
    res1.materialize(('post_vsm',))
    v3 = res1['post_vsm'][3].copy()
    par2 = {'C': par['C'] * 1.1, 'd': par['d'] - 0.1, 'tau': par['tau'] * 0.9}
    res2, _, _ = funs_mod.inference.laplace(exp, dict(par2))
    assert np.array_equal(res1['post_mean'][0], m0) and np.array_equal(res1['post_vsm'][3], v3)        # fetched / snapshotted before
    with pytest.raises(funs_mod._hip.HipBackendError, match='superseded'):
        res1['post_mean'][1]
    with pytest.raises(funs_mod._hip.HipBackendError, match='superseded'):
        res1['post_vsmGP'][0]
    with pytest.raises(funs_mod._hip.HipBackendError, match='superseded'):
        opt1[2]
    assert np.max(np.abs(res2['post_mean'][1] - m0)) > 0            # the new E-step's views are live
    # engine: the true-parameter extraction must not silently replace fit.infRes
    class Ds(Experiment):
        pass
    ds = Ds(c1['Ys'][:6], c1['binSize'])
    ds.params = {'C': c1['true_C'], 'd': c1['true_d'], 'tau': c1['true_tau']}
    fit = funs_mod.engine.PPGPFAfit(ds, initParams={k: v.copy() for k, v in c1['init'].items()}, EMmode='Batch', maxEMiter=1,
                                    extractAllTraj=True, extractAllTraj_trueParams=True, quiet=True)
    ref, _, _ = orc.laplace(c1['Ys'][:6], fit.optimParams, c1['binSize'], mode='exact', return_cov=False)
    tru, _, _ = orc.laplace(c1['Ys'][:6], ds.params, c1['binSize'], mode='exact', return_cov=False)
    for r in range(6):
        assert np.max(np.abs(fit.infRes['post_mean'][r] - ref['post_mean'][r])) <= 1e-7
        assert np.max(np.abs(fit.infRes_trueParams['post_mean'][r] - tru['post_mean'][r])) <= 1e-7


def test_failed_mode_search_raises(funs_mod, c1):
    """Parameters that overflow the rates: the reference would carry NaNs into the M-step; here the E-step raises."""
    exp = Experiment(c1['Ys'][:4], c1['binSize'])
    bad = {'C': c1['init_C'] * 0.0 + 50.0, 'd': c1['init_d'] + 800.0, 'tau': c1['init_tau'].copy()}
    with pytest.raises(funs_mod._hip.HipBackendError):
        funs_mod.inference.laplace(exp, bad)
    with warnings.catch_warnings():
        warnings.simplefilter('error')
        funs_mod.inference.laplace(exp, {k: v.copy() for k, v in c1['init'].items()})      # a healthy E-step neither warns nor raises


# ---------------------------------------------------------------------------------------------------------------
# two ranks over RCCL (needs two GPUs)
# ---------------------------------------------------------------------------------------------------------------
_RANK_CODE = r'''
import os, sys, json, numpy as np
ROOT = %r
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'poisson-gpfa_amd')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import funs
from conftest import Experiment
d = np.load(os.path.join(ROOT, 'tests', 'golden', 'c1_dataset.npz'))
exp = Experiment([d['Y'][r].astype(float) for r in range(20)], 10.0)
init = {'C': d['init_C'].copy(), 'd': d['init_d'].copy(), 'tau': d['init_tau'].copy()}
fit = funs.engine.PPGPFAfit(exp, initParams=init, EMmode='Batch', maxEMiter=2, CdOptimMethod='newton', quiet=True)
np.random.seed(1)
fit2 = funs.engine.PPGPFAfit(exp, initParams={k: v.copy() for k, v in init.items()}, EMmode='Online', maxEMiter=3, batchSize=6,
                             CdOptimMethod='newton', quiet=True)
from funs._session import session_for
sess, _ = session_for(exp, 3)
out = {'comm': [bool(sess.comm_ready), sess.rank, sess.size], 'nll': [float(v) for v in fit.posteriorLikelihood],
       'C': np.asarray(fit.optimParams['C']).tolist(), 'tau': np.asarray(fit.optimParams['tau']).tolist(),
       'nll_online': [float(v) for v in fit2.posteriorLikelihood], 'tau_online': np.asarray(fit2.optimParams['tau']).tolist(),
       'n_local': len(fit.infRes['post_mean'])}
with open(os.environ['OUT'] + '.' + os.environ.get('RANK', '0'), 'w') as fh:
    json.dump(out, fh)
'''


@pytest.mark.timeout(1800)
def test_two_rank_rccl_matches_single_rank(tmp_path):
    """Batch and stochastic EM with the trials sharded over two GPUs (RCCL all-reduce of the M-step statistics inside the
    C-ABI, unique-id rendezvous through the file keyed on the launcher) against the same run on one GPU."""
    from funs import _hip
    if _hip.device_count() < 2:
        pytest.skip('needs two GPUs (device_count = %d)' % _hip.device_count())
    import json
    script = tmp_path / 'rank.py'
    script.write_text(_RANK_CODE % ROOT)
    base = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'PGPFA_FORCE_COMM'):
        base.pop(k, None)
    one = tmp_path / 'one'
    subprocess.run([sys.executable, str(script)], env=dict(base, OUT=str(one)), check=True, timeout=800)
    ref = json.load(open(str(one) + '.0'))
    two = tmp_path / 'two'
    procs = [subprocess.Popen([sys.executable, str(script)],
                              env=dict(base, OUT=str(two), RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT='29663'))
             for r in range(2)]
    assert [pr.wait(timeout=800) for pr in procs] == [0, 0]
    res = [json.load(open(str(two) + '.%d' % r)) for r in range(2)]
    assert res[0]['comm'] == [True, 0, 2] and res[1]['comm'] == [True, 1, 2]
    assert res[0]['n_local'] == 10 and res[1]['n_local'] == 10
    for r in range(2):
        assert np.allclose(res[r]['nll'], ref['nll'], rtol=1e-9, atol=0)
        assert np.allclose(res[r]['C'], ref['C'], rtol=0, atol=1e-8)
        assert np.allclose(res[r]['tau'], ref['tau'], rtol=1e-8, atol=0)
        assert np.allclose(res[r]['nll_online'], ref['nll_online'], rtol=1e-6, atol=0)
        assert np.allclose(res[r]['tau_online'], ref['tau_online'], rtol=1e-4, atol=0)
    assert res[0]['C'] == res[1]['C']


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` without a launcher starts the ranks itself (the driver's scaling run): dry run of the
    launcher on any box, a real 2-rank step when two GPUs are visible."""
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run'], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr[-1000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line['world'] == 2 and line['rank'] == 0 and line['master'].startswith('127.0.0.1:')
    from funs import _hip
    if _hip.device_count() < 2:
        return
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--config', 'c1', '--steps', '2', '--warmup', '1',
                          '--no-cpu-baseline'], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 2 and line['value'] > 0


# ---------------------------------------------------------------------------------------------------------------
# synthetic population drawn on the device (reference recipe util.py:705-750)
# ---------------------------------------------------------------------------------------------------------------
def test_device_generator_distributions(funs_mod):
    """The device sampler is not NumPy's stream, so it is checked against the distributions of the reference recipe: latents
    x_k ~ N(0, K(tau_k)) (sample covariance over 4000 trials against the oracle's Gram matrix, entries within 6 standard errors;
    zero mean), counts y ~ Poisson(exp(Cx+d)) given the returned latents (standardised residuals: mean 0, variance 1, per-count
    frequencies at small rates against the Poisson pmf by chi-square; large rates exercise the rejection sampler), and it is a
    pure function of the seed."""
    from funs import _hip
    q, p, T, R = 12, 3, 40, 4000
    rng = np.random.default_rng(5)
    C = rng.uniform(-0.5, 0.5, (q, p))
    d = rng.uniform(-2.0, 0.0, q) - 1.0
    d[-2:] = np.array([2.6, 3.3])                       # two busy neurons: rates 10-60 per bin -> transformed rejection
    tau = np.array([0.03, 0.12, 0.4])
    ctx = _hip.Context(q, p, T, R, 10.0)
    try:
        ctx.set_params(C, d, tau)
        X, Y = ctx.generate(1234)
        X2, Y2 = ctx.generate(1234)
        assert np.array_equal(X, X2) and np.array_equal(Y, Y2)
        X3, Y3 = ctx.generate(1235, np.arange(8, dtype=np.int32))
        assert not np.array_equal(X3, X[:8])
        # counts generated into the resident tensor are what the E-step then sees
        Xs, Ys = ctx.generate(1234, np.arange(8, dtype=np.int32))
        assert np.array_equal(Ys, Y[:8])
        obj, _, st = ctx.estep_laplace(np.arange(8, dtype=np.int32))
        assert np.all(st == 0) and np.isfinite(obj)
    finally:
        ctx.close()
    K = orc.make_K(tau, T, 10.0)
    for k in range(p):
        xs = X[:, k, :]                                 # (R, T)
        assert np.max(np.abs(xs.mean(axis=0))) <= 6.0 * np.sqrt(1.0 / R)
        S = xs.T @ xs / R
        se = np.sqrt((K[k] ** 2 + np.outer(np.diag(K[k]), np.diag(K[k]))) / R)       # Wishart standard errors
        assert np.max(np.abs(S - K[k]) / se) <= 6.0
    lam = np.exp(np.einsum('nk,rkt->rnt', C, X) + d[None, :, None])
    Yf = Y.astype(np.float64)
    z = (Yf - lam) / np.sqrt(lam)
    for n in range(q):
        m = z[:, n, :].size
        assert abs(z[:, n, :].mean()) <= 6.0 / np.sqrt(m)
        # Var[(y-lam)/sqrt(lam)] = 1 ; its own variance is (2 + 1/lam)/m
        tol = 6.0 * np.sqrt((2.0 + np.mean(1.0 / lam[:, n, :])) / m)
        assert abs(np.mean(z[:, n, :] ** 2) - 1.0) <= tol
    assert lam[:, -1, :].mean() > 20 and Y[:, -1, :].max() > 40
    # pmf check on a band of nearly constant small rate
    sel = (lam > 0.19) & (lam < 0.21)
    from scipy.stats import poisson
    obs = np.bincount(Y[sel].astype(int), minlength=6)[:6]
    expc = np.array([poisson.pmf(c, lam[sel]).sum() for c in range(6)])
    keep = expc > 5
    chi2 = np.sum((obs[keep] - expc[keep]) ** 2 / expc[keep])
    assert chi2 <= 30.0, (chi2, obs, expc)
    # the drop-in constructor
    ds = funs_mod.util.dataset(trialDur=400, binSize=10, numTrials=6, xdim=2, ydim=9, seed=3, sampler='device')
    assert len(ds.data) == 6 and ds.data[0]['Y'].shape == (9, 40) and ds.data[0]['X'].shape == (2, 40) and ds.data[0]['Y'].dtype == np.uint8
    ds2 = funs_mod.util.dataset(trialDur=400, binSize=10, numTrials=6, xdim=2, ydim=9, seed=3, sampler='device')
    assert all(np.array_equal(a['Y'], b['Y']) for a, b in zip(ds.data, ds2.data))


def test_batched_mcmc_chains_match_single_chains(funs_mod, c1, c1_experiment):
    """Lockstep elliptical-slice chains over several trials (one batched device evaluation per proposal round): chain i must be
    the single-trial sampler's chain under the same seed (1e-12: same arithmetic, same draws), and the chain of the golden
    fixture's trial under its seed must be the reference's (1e-9)."""
    from funs import mcmc
    g = load_golden('c1_mcmc.npz')
    params = {k: v.copy() for k, v in c1['init'].items()}
    nsamp = int(g['n_samples'])
    trials = [int(g['trial']), 3, 11]
    seeds = [int(g['seed']), 7, 8]
    chains = mcmc.PosteriorMCMC_batch(c1_experiment, dict(params), nsamp, trials, seeds)
    assert chains.shape == (3,) + g['chain'].shape
    assert np.max(np.abs(chains[0] - g['chain'])) <= 1e-9
    for i in (1, 2):
        np.random.seed(seeds[i])
        single = mcmc.PosteriorMCMC(c1_experiment, dict(params), nsamp, trials[i])
        assert np.max(np.abs(chains[i] - single)) <= 1e-12


# ---------------------------------------------------------------------------------------------------------------
# single precision: the FP32 instantiation of the MFMA GEMM and the mixed-precision dual evaluation (BASELINE config 5)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (200, 77, 300), (513, 1030, 48)])
def test_f32_gemm_vs_numpy(M, N, K):
    """v_mfma_f32_16x16x4_f32 instantiation of the tile kernel (both operand forms) against float64 numpy products of the
    float-rounded operands: 2e-6 relative to |A||B| (single-precision accumulation over K <= 300)."""
    from funs import _hip
    rng = np.random.default_rng(M + N + K)
    A, Bt, Bn = rng.standard_normal((M, K)), rng.standard_normal((N, K)), rng.standard_normal((K, N))
    C0 = rng.standard_normal((M, N))
    ctx = _hip.Context(4, 2, 16, 1, 10.0)
    try:
        A32, Bt32, Bn32, C32 = (x.astype(np.float32).astype(np.float64) for x in (A, Bt, Bn, C0))
        got = ctx.test_gemm_nt(A, Bt, C0, alpha=0.5, beta=2.0, f32=True)
        ref = 0.5 * A32 @ Bt32.T + 2.0 * C32
        assert np.max(np.abs(got - ref)) <= 2e-6 * (np.abs(A32) @ np.abs(Bt32).T).max()
        got = ctx.test_gemm_nn(A, Bn, f32=True)
        assert np.max(np.abs(got - A32 @ Bn32)) <= 2e-6 * (np.abs(A32) @ np.abs(Bn32)).max()
    finally:
        ctx.close()


def _dual_eval(q, p, T, Y, C, d, tau, lam, f32):
    from funs import _hip
    R = Y.shape[0]
    ctx = _hip.Context(q, p, T, R, 10.0)
    try:
        ctx.upload_counts(Y)
        ctx.set_option('cov_mode', 2)
        ctx.set_option('dual_lowrank', 1)
        ctx.set_option('dual_f32', 1 if f32 else 0)
        ctx.set_params(C, d, tau)
        cost, grad = ctx.dual_costgrad_batch(np.arange(R, dtype=np.int32), lam)
        assert ctx.info('plan_lowrank') == 1.0
        return cost, grad
    finally:
        ctx.close()


def test_mixed_precision_dual_evaluation_vs_numpy():
    """dual_f32: r x r Cholesky, its inverse and Yt in single precision on the FP32 matrix cores, everything else in FP64.  At 40
    neurons x 20 latents x 200 bins (n = 4000, dense numpy is seconds) against the oracle's dual (reference jitter included): cost 2e-6 rel,
    gradient 2e-4 of its largest entry (the covariance blocks c_n^T Sigma_t c_n inherit cond(B) * 6e-8); the FP64 engine on the
    same input keeps 1e-8 / 1e-7."""
    q, p, T, R = 40, 20, 200, 2
    _, Ys, _ = orc.synth_dataset(q, p, T, R, seed=33, dOffset=0.0)
    Y = np.stack(Ys).astype(np.uint8)
    rng = np.random.default_rng(33)
    C, d, tau = 0.25 * rng.standard_normal((q, p)), np.log(Y.mean(axis=(0, 2)) + 0.1), 0.15 + 0.3 * rng.random(p)
    lam = 0.05 + 0.5 * rng.random((R, q * T))
    c64, g64 = _dual_eval(q, p, T, Y, C, d, tau, lam, False)
    c32, g32 = _dual_eval(q, p, T, Y, C, d, tau, lam, True)
    K_big = orc.make_K_big(orc.make_K(tau, T, 10.0))
    C_big, d_big = orc.make_Cd_big(C, d, T)
    Kinv_big = np.linalg.inv(K_big)
    for i in range(R):
        y = Ys[i].reshape(-1).astype(float)
        ref_cost = orc.dual_cost(lam[i], y, C_big, K_big, Kinv_big, d_big)
        ref_grad = orc.dual_grad(lam[i], y, C_big, K_big, Kinv_big, d_big)
        assert abs(c64[i] - ref_cost) <= 1e-8 * abs(ref_cost) and rel(g64[i], ref_grad) <= 1e-7
        print('mixed precision: cost rel %.2e, grad rel %.2e' % (abs(c32[i] - ref_cost) / abs(ref_cost), rel(g32[i], ref_grad)))
        assert abs(c32[i] - ref_cost) <= 2e-6 * abs(ref_cost)
        assert rel(g32[i], ref_grad) <= 2e-4


@pytest.mark.timeout(1800)
def test_mixed_precision_dual_evaluation_at_config5_dimensions():
    """BASELINE config 5's dimensions (500 neurons, 20 latents, 1000 bins: n = 20 000, 500 000 dual variables per trial), two
    trials: the mixed-precision evaluation against the FP64 low-rank engine on the same input (that one is pinned to plain numpy
    at sizes numpy can invert, above and in test_dual_evaluation_lowrank_engine_wide_latent_state): cost 1e-5 rel, gradient 1e-3 of
    its largest entry."""
    import bench
    q, p, T, R = 500, 20, 1000, 2
    true, Ys = bench.synth_shard(q, p, T, R, 12, 0)
    Y = np.stack(Ys)
    rng = np.random.default_rng(12)
    tau = np.linspace(0.1, 0.5, p)
    lam = np.exp(true['C'] @ np.zeros((p, T)) + true['d'][:, None]).reshape(1, -1) * (0.5 + rng.random((R, q * T)))
    c64, g64 = _dual_eval(q, p, T, Y, true['C'], true['d'], tau, lam, False)
    c32, g32 = _dual_eval(q, p, T, Y, true['C'], true['d'], tau, lam, True)
    print('config-5 dims: cost rel %s, grad rel %s' % (np.abs(c32 - c64) / np.abs(c64), [rel(g32[i], g64[i]) for i in range(R)]))
    assert np.all(np.isfinite(c64)) and np.all(np.isfinite(g32))
    assert np.max(np.abs(c32 - c64) / np.abs(c64)) <= 1e-5
    assert max(rel(g32[i], g64[i]) for i in range(R)) <= 1e-3


@pytest.mark.parametrize('w32', [1, 0])
def test_host_free_pcg_path_reaches_the_same_modes(c1, w32):
    """The inner PCG without host round trips (device stop flag, fused per-bin passes; with and without the packed single-precision
    curvature in the matvec) is only chosen by itself for large chunks: forced on at config-1 size it must land on the polished
    modes of the reference (1e-8) with the covariance blocks of the oracle, like the round-trip form."""
    from funs import _hip
    g = load_golden('c1_laplace.npz')
    ctx = _hip.Context(30, 3, 100, 20, c1['binSize'])
    try:
        ctx.upload_counts(c1['Y'])
        ctx.set_option('cov_mode', 2)
        ctx.set_option('pcg_fused', 2)
        ctx.set_option('pcg_w32', w32)
        ctx.set_params(c1['init_C'], c1['init_d'], c1['init_tau'])
        obj, iters, status = ctx.estep_laplace()
        assert np.all(status == 0) and ctx.info('plan_lowrank') == 1.0 and ctx.info('last_pcg_iterations') > 0
        assert np.max(np.abs(ctx.post_mean().reshape(20, -1) - g['polished'])) <= 1e-8
        res, nll, _ = orc.laplace(c1['Ys'], c1['init'], c1['binSize'], mode='exact', return_cov=False)
        assert abs(-obj / 20 - nll) <= 1e-9 * abs(nll)
        assert rel(ctx.post_vsm()[:4], np.stack(res['post_vsm'][:4])) <= 1e-8
        # warm restart: a handful of iterations, same modes
        obj2, _, st2 = ctx.estep_laplace(warm_start=True)
        assert np.all(st2 == 0) and abs(obj2 - obj) <= 1e-10 * abs(obj)
    finally:
        ctx.close()


def test_em_iteration_log_json_lines(c1, c1_experiment, tmp_path, monkeypatch):
    """PGPFA_LOG_JSONL=<path>: one JSON line per EM iteration with the values the reference prints (iteration, nPLL) plus timings and the
    device's work counters (metrics / logging)."""
    import json
    import funs
    path = tmp_path / 'em.jsonl'
    monkeypatch.setenv('PGPFA_LOG_JSONL', str(path))
    init = {k: v.copy() for k, v in c1['init'].items()}
    fit = funs.engine.PPGPFAfit(c1_experiment, initParams=init, inferenceMethod='laplace', EMmode='Batch', maxEMiter=3, quiet=True,
                                CdOptimMethod='newton')
    lines = [json.loads(l) for l in open(path)]
    assert [l['iteration'] for l in lines] == [1, 2, 3]
    assert np.allclose([l['nPLL'] for l in lines], fit.posteriorLikelihood, rtol=0, atol=0)
    assert all(l['estep_s'] > 0 and l['mstep_s'] > 0 and l['chunk_trials'] == 20 for l in lines)
    assert lines[0]['em_mode'] == 'Batch' and lines[0]['VLB'] is None
