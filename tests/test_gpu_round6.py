"""Round-6 GPU parity tests.

* The workload of the driver's bench line ITSELF: batch EM at 200 neurons x 10 latents x 500 bins x 1024 trials from the Poisson-PCA start with the
  options `bench.py` runs (extrapolated warm start of the modes, `CdOptimMethod='newton'` with its extrapolated start, lockstep timescale finder),
  every iteration checked against the reference's arithmetic (engine.py:180-238; inference.py:12-48; learning.py:20-91, 175-255).
* An E-step behind a JUMP of the parameters (cross-validation folds util.py:180-335, engine.py:523-540, a second fit in one process): the resident
  modes belong to other parameters; the start guard and the guarded extrapolation keep such an E-step at the cost of a cold one.

Everything goes through the drop-in `funs` surface; numpy and the oracle are the checkers."""
import time

import numpy as np
import pytest

from oracle import pgpfa_oracle as orc

pytestmark = pytest.mark.gpu

# nPLL of the first four EM iterations as `python bench.py` prints them (key `nll`; profiles/r05_bench_c3_driver_protocol.json and every run since):
# the driver line leans on this fit being the same fit on every box
BENCH_NLL = [-49192.873468, -48026.698574, -47803.722180, -47616.751845]


@pytest.fixture(scope='module')
def funs_mod():
    import funs
    return funs


def _stationarity(par, Ys, pm, T, bin_ms=10.0):
    """max |gradient of the reference's log-posterior| (inference.py:34-48) over the trials, and the mean of negLogPosteriorUnNorm there"""
    Kinv = np.linalg.inv(orc.make_K(par['tau'], T, bin_ms))
    worst, f = 0.0, 0.0
    for r, Y in enumerate(Ys):
        X = pm[r]
        h = par['C'] @ X + par['d'][:, None]
        e = np.exp(h)
        KX = np.einsum('kts,ks->kt', Kinv, X)
        worst = max(worst, float(np.max(np.abs(par['C'].T @ (e - Y) + KX))))
        f += np.sum(e) - np.sum(Y * h) + 0.5 * np.sum(X * KX)
    return worst, f / len(Ys)


@pytest.mark.timeout(1800)
def test_bench_workload_batch_em_1024_trials(funs_mod):
    """Four batch-EM iterations of the bench's own workload and loop (bench.py: em_step).  Every iteration: (1) all 1024 modes are stationary
    points of the reference's log-posterior under that iteration's parameters (<= 1e-6) and no trial reports a non-zero status; (2) the reported
    nPLL is the mean of negLogPosteriorUnNorm at the modes (1e-9) and equals the value bench.py prints (1e-8); (3) the new (C,d) zero the oracle's
    gradient of MStepObservationCost on the whole posterior (2e-6: the device Newton stops at a predicted parameter error of 1e-10, Hessian
    scale ~1e3); (4) every new timescale zeroes the oracle's MStepGPtimescaleCost_grad on the device's PautoSum (1e-6 R, as the 48-trial test)."""
    import bench
    from funs import _session
    q, p, T, R = 200, 10, 500, 1024
    _session.drop_sessions()
    true, Ys = bench.synth_shard(q, p, T, R, 12, 0)
    exp = bench.Shard(Ys, 10.0)
    np.random.seed(0)
    params = {k: np.real(np.asarray(v)).astype(np.float64) for k, v in funs_mod.util.initializeParams(p, q, exp).items()}
    Yf = [y.astype(np.float64) for y in Ys]
    optim = None
    all_idx = np.arange(R, dtype=np.int32)
    for it in range(4):
        infRes, nll, optim = funs_mod.inference.laplace(exp, params, prevOptimRes=optim)
        ctx = infRes.session.ctx
        assert np.all(infRes.newton_status == 0)
        new, _ = funs_mod.learning.updateParams(params, infRes, exp, CdOptimMethod='newton')
        pm, pv, P = ctx.post_mean(all_idx), ctx.post_vsm(all_idx), ctx.pautosum()
        worst, f_mean = _stationarity(params, Yf, pm, T)
        print('bench workload, iteration %d: worst |grad| %.2e, nPLL %.6f (bench %.6f), rank %d, extrapolated starts %s'
              % (it, worst, nll, BENCH_NLL[it], int(ctx.info('lowrank_rtot')), it >= 2))
        assert worst <= 1e-6
        assert abs(nll + f_mean) <= 1e-9 * abs(f_mean)
        assert abs(nll - BENCH_NLL[it]) <= 1e-8 * abs(BENCH_NLL[it])
        g_cd = orc.mstep_cd_grad(orc.cd_to_vec(new['C'], new['d']), Yf, list(pm), list(pv), p, q)
        assert np.max(np.abs(g_cd)) <= 2e-6
        logp = np.log(1.0 / (new['tau'] * 100.0) ** 2)
        for k in range(p):
            assert abs(orc.tau_grad(logp[k], P[k], R)[0]) <= 1e-6 * R
        params = new
    _session.drop_sessions()


@pytest.mark.timeout(1800)
def test_estep_behind_a_parameter_jump_costs_what_a_cold_one_costs(funs_mod):
    """Three EM iterations of a fit at 200 x 10 x 500 x 512 trials, then the parameters JUMP to the generating ones (other loadings, ranks x 2), two
    E-steps there, then every timescale is halved.  The resident modes of the fit mean nothing under the new loadings (log rates of +-100): the
    start guard restarts those slots at zero, the extrapolation is skipped behind the jump (the difference of the last two modes is the jump's
    effect), no trial goes through the dense per-trial retry - and every one of these E-steps returns the modes of the reference's log-posterior
    (1e-6), at no more than 3 x the library time of the cold E-step at the same parameters (the test of VERDICT round 5, item 4)."""
    import bench
    from funs import _session
    q, p, T, R = 200, 10, 500, 512
    _session.drop_sessions()
    true, Ys = bench.synth_shard(q, p, T, R, 12, 0)
    exp = bench.Shard(Ys, 10.0)
    Yf = [y.astype(np.float64) for y in Ys]
    np.random.seed(0)
    params = {k: np.real(np.asarray(v)).astype(np.float64) for k, v in funs_mod.util.initializeParams(p, q, exp).items()}
    optim = None
    all_idx = np.arange(R, dtype=np.int32)
    for it in range(3):
        infRes, nll, optim = funs_mod.inference.laplace(exp, params, prevOptimRes=optim)
        params, _ = funs_mod.learning.updateParams(params, infRes, exp, CdOptimMethod='newton')
    ctx = infRes.session.ctx
    truth = {k: np.asarray(v, dtype=np.float64).copy() for k, v in true.items()}
    half = dict(truth, tau=truth['tau'] * 0.5)

    def estep(par, tag):
        nonlocal optim
        infRes, nll, optim = funs_mod.inference.laplace(exp, par, prevOptimRes=optim)
        ms = ctx.info('last_estep_ms')
        worst, f_mean = _stationarity(par, Yf, ctx.post_mean(all_idx), T)
        print('%-28s library %7.1f ms, cold restarts %4d, dense retries %d, param step %.3g (prev %.3g), worst |grad| %.2e'
              % (tag, ms, ctx.info('last_cold_restarts'), ctx.info('last_dense_retries'), ctx.info('last_param_step'), ctx.info('last_param_step_prev'), worst))
        assert np.all(infRes.newton_status == 0) and ctx.info('last_dense_retries') == 0.0
        assert worst <= 1e-6 and abs(nll + f_mean) <= 1e-9 * abs(f_mean)
        return ms, nll

    ms_jump, nll_jump = estep(truth, 'jump to the truth')
    assert ctx.info('last_cold_restarts') >= 0.9 * R                  # (the fit's latent space is a rotation of the generating one)
    ms_after, _ = estep(truth, 'same parameters again')
    assert ctx.info('last_cold_restarts') == 0.0
    ms_half, _ = estep(half, 'timescales halved')
    ms_half2, _ = estep(half, 'same parameters again')
    # the yardstick: cold E-steps at the same parameters in a fresh session (same workspace plan sizes)
    _session.drop_sessions()
    sess, _ = _session.session_for(exp, p)
    cold = {}
    for tag, par in (('truth', truth), ('half', half)):
        sess.set_params(par)
        sess.ctx.estep_laplace(all_idx, warm_start=0)               # (plans the workspace)
        t0 = time.time()
        obj, _, st = sess.ctx.estep_laplace(all_idx, warm_start=0)
        cold[tag] = sess.ctx.info('last_estep_ms')
        assert np.all(st == 0)
        if tag == 'truth':
            assert abs(-obj / R - nll_jump) <= 1e-9 * abs(nll_jump)
    print('cold E-steps: truth %.1f ms, halved timescales %.1f ms' % (cold['truth'], cold['half']))
    assert ms_jump <= 3.0 * cold['truth'] and ms_after <= 3.0 * cold['truth']
    assert ms_half <= 3.0 * cold['half'] and ms_half2 <= 3.0 * cold['half']
    _session.drop_sessions()


@pytest.mark.timeout(1800)
def test_lockstep_timescale_update_with_prior_at_config3_dimensions(funs_mod):
    """learning.updateParamsWithPrior on the posterior of 256 trials at 200 x 10 x 500 with tauOptimMethod='lockstep' (round 6, opt-in) against the
    reference's 'TNC' on the same resident posterior, for the step sizes of stochastic-EM iterations 1, 4 and 11: every new timescale within 1e-3
    (relative) of where the reference's scipy call stops - the tolerance config 4's test states - and a zero of the reference's regularised
    gradient expression (learning.py:726-769, restated by the oracle on the device's PautoSum; 1e-6 R).  (C,d) are the same numbers either way."""
    import bench
    from funs import _session
    import warnings
    q, p, T, R = 200, 10, 500, 256
    _session.drop_sessions()
    true, Ys = bench.synth_shard(q, p, T, R, 12, 0)
    exp = bench.Shard(Ys, 10.0)
    np.random.seed(0)
    params = {k: np.real(np.asarray(v)).astype(np.float64) for k, v in funs_mod.util.initializeParams(p, q, exp).items()}
    infRes, nll, _ = funs_mod.inference.laplace(exp, params)
    prior = np.diag(np.ones(q * (p + 1)))
    P = None
    for n in (0, 3, 10):
        sz = 1.0 / (n + 1) ** 0.75
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            t0 = time.time()
            ref, _, _ = funs_mod.learning.updateParamsWithPrior(params, infRes, exp, 'newton', 'TNC', sz, sz, prior, covOpts='useDiag')
            t1 = time.time()
        new, det, _ = funs_mod.learning.updateParamsWithPrior(params, infRes, exp, 'newton', 'lockstep', sz, sz, prior, covOpts='useDiag')
        t2 = time.time()
        if P is None:
            P = infRes.session.ctx.pautosum()
        print('prior step %.3f: M-step with TNC %.1f ms, with the lockstep finder %.1f ms (%d device passes); max relative difference of tau %.2e'
              % (sz, (t1 - t0) * 1e3, (t2 - t1) * 1e3, det['tau'][0].nfev, np.max(np.abs(new['tau'] - ref['tau']) / ref['tau'])))
        assert np.max(np.abs(new['tau'] - ref['tau']) / ref['tau']) <= 1e-3
        assert np.max(np.abs(new['C'] - ref['C'])) <= 1e-9 and np.max(np.abs(new['d'] - ref['d'])) <= 1e-9
        assert all(d.success for d in det['tau'])
        for k in range(p):
            pv = np.log(1.0 / (new['tau'][k] * 100.0) ** 2)
            assert abs(orc.tau_grad_prior(pv, P[k], R, 10.0, params['tau'][k], sz)[0]) <= 1e-6 * R
    _session.drop_sessions()


def test_timescale_rounds_beside_the_cd_passes_change_nothing(funs_mod, monkeypatch):
    """learning.updateParams with the rounds of the lockstep timescale finder started on the side stream beside the (C,d) Newton passes
    (pgpfa_mstep_tau_costgrad_multi_begin / _end, round 6) against the same update with one problem after the other (learning.py:295-309: the
    reference's order): the SAME parameters, bit for bit - same sample points, same device arithmetic - over three EM iterations at 100 x 5 x 200 x 64
    and at 200 x 10 x 500 x 128; and the asynchronous pass returns the bits of the synchronous one."""
    import bench
    from funs import _session
    learning = funs_mod.learning
    for q, p, T, R in ((100, 5, 200, 64), (200, 10, 500, 128)):
        true, Ys = bench.synth_shard(q, p, T, R, 12, 0)
        out = {}
        for overlap in (True, False):
            _session.drop_sessions()
            monkeypatch.setattr(learning, 'M_STEP_OVERLAP', overlap)
            exp = bench.Shard(Ys, 10.0)
            np.random.seed(0)
            params = {k: np.real(np.asarray(v)).astype(np.float64) for k, v in funs_mod.util.initializeParams(p, q, exp).items()}
            optim, seq = None, []
            for it in range(3):
                infRes, nll, optim = funs_mod.inference.laplace(exp, params, prevOptimRes=optim)
                params, _ = learning.updateParams(params, infRes, exp, CdOptimMethod='newton')
                seq.append((nll, params['C'].copy(), params['d'].copy(), params['tau'].copy()))
            out[overlap] = seq
            if overlap:
                ctx = infRes.session.ctx
                Q = np.log(1.0 / (params['tau'] * 100.0) ** 2)[None, :] + np.array([-0.1, 0.0, 0.05, 0.2])[:, None]
                ctx.mstep_precomp()
                f0, g0 = ctx.mstep_tau_costgrad_multi(Q)
                ctx.mstep_tau_costgrad_multi_begin(Q)
                f1, g1 = ctx.mstep_tau_costgrad_multi_end()
                assert np.array_equal(f0, f1) and np.array_equal(g0, g1)
        for a, b in zip(out[True], out[False]):
            assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    _session.drop_sessions()


def test_replan_under_a_spent_growth_budget_runs_in_chunks_and_changes_nothing(funs_mod):
    """A re-plan that finds the arena too small and its growth budget spent (workspace_grow_budget_ms; pages another process has just released
    map at up to 40 ms per GB) settles for the slots that fit and walks the trial list in balanced chunks.  Forced here: 384 trials at
    200 x 10 x 500, a first E-step at the Poisson-PCA start (plan for ranks <= 832), then the generating parameters (rank 1120: the slabs
    no longer fit) with a budget of 1 us and a floor of 64 slots.  The chunked E-step must return what a fresh context with an unbounded plan
    returns: same modes (1e-8), same objective (1e-11 rel), same PautoSum (1e-9 of its largest entry), all statuses 0."""
    import bench
    from funs import _hip
    q, p, T, R = 200, 10, 500, 384
    true, Ys = bench.synth_shard(q, p, T, R, 12, 0)
    Y = np.stack(Ys)
    exp = bench.Shard(Ys, 10.0)
    np.random.seed(0)
    init = {k: np.real(np.asarray(v)).astype(np.float64) for k, v in funs_mod.util.initializeParams(p, q, exp).items()}
    out = {}
    for bounded in (True, False):
        ctx = _hip.Context(q, p, T, R, 10.0)
        try:
            ctx.upload_counts(Y)
            if bounded:
                ctx.set_option('workspace_pool', 0)               # (an arena of its own: a pooled one of an earlier test may already hold what the re-plan asks for)
                ctx.set_option('workspace_grow_budget_ms', 1e-3)
                ctx.set_option('workspace_grow_floor_slots', 64)
                ctx.set_params(init['C'], init['d'], init['tau'])
                _, _, st = ctx.estep_laplace()
                assert np.all(st == 0) and ctx.info('chunk_trials') == float(R)
            ctx.set_params(true['C'], true['d'], true['tau'])
            obj, _, st = ctx.estep_laplace()
            assert np.all(st == 0) and ctx.info('plan_lowrank') == 1.0
            chunk = ctx.info('chunk_trials')
            ctx.mstep_precomp()
            out[bounded] = (obj, ctx.post_mean().copy(), ctx.pautosum().copy(), chunk)
            print('bounded growth %s: chunk_trials %d, arena %.1f GB, plans %d' % (bounded, chunk, ctx.info('arena_bytes') / 1e9, ctx.info('plans')))
        finally:
            ctx.close()
    a, b = out[True], out[False]
    assert 64 <= a[3] < R and a[3] % 8 == 0 and b[3] == float(R)
    assert abs(a[0] - b[0]) <= 1e-11 * abs(b[0])
    assert np.max(np.abs(a[1] - b[1])) <= 1e-8
    assert np.max(np.abs(a[2] - b[2])) <= 1e-9 * np.max(np.abs(b[2]))


@pytest.mark.parametrize('shape', [(40, 7, 70, 6), (33, 9, 130, 5), (25, 10, 64, 3), (30, 3, 100, 8), (50, 1, 90, 4), (40, 12, 70, 5), (45, 17, 100, 9),
                                   (50, 20, 64, 3), (60, 10, 300, 24), (200, 10, 500, 32)])
@pytest.mark.parametrize('keep', [0, 1])
def test_compact_rank_offsets_change_nothing(shape, keep):
    """rank_gran = 4 (round 6: the latents' ranks rounded to 4 in the r x r system, in L^-T and in the columns of Yt / D, products with F_k still issued
    at the rank rounded to 16 against zero columns of F_k; B assembled on padded 16-blocks and stored through a map; the panel yt_mix stages gathered
    through the same map) against rank_gran = 16 (every latent owns whole 16-blocks, rounds 1-5): the same low-rank factors, so the same numbers up to
    the order of the sums - objective 1e-12 rel, modes 1e-9 (both stop on a predicted error of 1e-9), post_vsm / PautoSum 1e-10 rel, per-trial
    post_vsmGP (keep = 1: the products that skip the zero columns left of a latent's first, and the partial clearing of L^-T) 1e-10 rel - with fewer
    rows: lowrank_rtot must not grow and the widths 7 / 9 / 12 / 17 / 20, bins that are no multiple of 4 and the fused / stand-alone mixing
    passes all run it.  And the dual-variational evaluation through the same engine (cost 1e-12 rel, gradient 1e-9 of its largest entry)."""
    from funs import _hip
    q, p, T, R = shape
    rng = np.random.default_rng(q * 1000 + p)
    _, Ys, _ = orc.synth_dataset(q, p, T, R, seed=p, dOffset=0.0)
    Y = np.stack(Ys).astype(np.uint8)
    par = {'C': 0.3 * rng.standard_normal((q, p)) / np.sqrt(max(1, p / 4)), 'd': np.log(Y.mean(axis=(0, 2)) + 0.1), 'tau': 0.05 + 0.3 * rng.random(p)}
    lam = np.exp(0.2 * rng.standard_normal((min(R, 4), q * T)) - 1.0)
    out = {}
    for g in (16, 4):
        ctx = _hip.Context(q, p, T, R, 10.0)
        try:
            ctx.upload_counts(Y)
            ctx.set_option('cov_mode', 2)
            ctx.set_option('rank_gran', g)
            ctx.set_option('keep_trial_vsmgp', keep)
            ctx.set_params(par['C'], par['d'], par['tau'])
            obj, _, st = ctx.estep_laplace()
            assert np.all(st == 0) and ctx.info('plan_lowrank') == 1.0 and ctx.info('last_dense_retries') == 0.0
            rank = ctx.info('lowrank_rtot')
            ctx.mstep_precomp()
            res = [obj, ctx.post_mean().copy(), ctx.post_vsm().copy(), ctx.pautosum().copy(), rank]
            res.append(ctx.post_vsmgp(np.arange(min(R, 3), dtype=np.int32)).copy() if keep else None)
            ctx.set_option('dual_lowrank', 1)
            res.append(ctx.dual_costgrad_batch(np.arange(lam.shape[0], dtype=np.int32), lam))
            out[g] = res
        finally:
            ctx.close()
    a, b = out[4], out[16]
    assert a[4] <= b[4] and a[4] % 4 == 0
    assert abs(a[0] - b[0]) <= 1e-12 * abs(b[0])
    assert np.max(np.abs(a[1] - b[1])) <= 1e-9
    assert np.max(np.abs(a[2] - b[2])) <= 1e-10 * np.max(np.abs(b[2]))
    assert np.max(np.abs(a[3] - b[3])) <= 1e-10 * np.max(np.abs(b[3]))
    if keep:
        assert np.max(np.abs(a[5] - b[5])) <= 1e-10 * np.max(np.abs(b[5]))
    (ca, ga), (cb, gb) = a[6], b[6]
    assert np.max(np.abs(ca - cb) / np.abs(cb)) <= 1e-12 and np.max(np.abs(ga - gb)) <= 1e-9 * np.max(np.abs(gb))
