"""Round-5 GPU parity tests: the BASELINE configurations at their REAL per-GPU sizes.

* config 5's share of one GPU: inference.dualVariational (reference inference.py:188-219, 259-432) on 256 trials at 500 neurons x 20
  latents x 1000 bins (2048 trials over 8 GPUs), FP64 and mixed precision - about 200 GB of chunk workspace, the wide (17..20 latents)
  kernels at full occupancy, a workspace re-plan between a 128-trial and a 256-trial call;
* the variational fixed point handing a trial back instead of failing the call, dual variables that stay readable after
  an interleaved Laplace E-step.

One dense oracle evaluation of one trial is a 20 000 x 20 000 factorisation + inverse on the host, so size-independent properties stand in
on all trials and the dense numpy restatement of the reference's callbacks checks two sampled ones.  Everything goes through the drop-in
`funs` surface or the C-ABI wrapper; numpy and the oracle are the checkers."""
import numpy as np
import pytest

from conftest import Experiment
from oracle import pgpfa_oracle as orc

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, np.max(np.abs(b)))


@pytest.fixture(scope='module')
def funs_mod():
    import funs
    return funs


def _free_hbm_gb():
    from funs import _hip
    ctx = _hip.Context(2, 1, 4, 1, 10.0)
    try:
        return ctx.info('hbm_bytes_free') / 1e9
    finally:
        ctx.close()


@pytest.mark.timeout(3000)
def test_config5_per_gpu_share_of_256_trials(funs_mod, monkeypatch):
    """256 trials at 500 x 20 x 1000 through inference.dualVariational (fixed-point solver, low-rank engine), FP64 and mixed.
    (1) every trial settles (status 0, <= 12 passes); (2) on TWO sampled trials (first and last of the list: both ends of the chunk) the
    reference's dualProblem_grad (inference.py:215-219), restated in dense numpy on the 20 000 x 20 000 matrices, is below 1e-6 in the
    max-norm at the returned lambda, the dual cost agrees 1e-8, the covariance blocks 1e-7; (3) post_mean = -K C_big (lambda - y)
    (inference.py:194) on ALL 256 trials, 1e-9; (4) the M-step statistic is additive: PautoSum of the 256-trial call = PautoSum of the
    call on trials 0..127 + that on 128..255 (1e-8 of its largest entry) - the first of those calls plans the workspace for 128 slots, the
    256-trial call re-plans it (chunk_trials 128 -> 256) and must change nothing: same dual optimum (1e-10 rel) and posterior means (1e-7:
    the fixed point stops at 1e-8 in the offsets) for the first 128 trials under either plan; (5) mixed precision: bound and nPLL within
    1e-5 rel of the FP64 run, the FP64 dual gradient at the mixed run's lambda below 1e-5; (6) a warm restart from the resident optimum
    settles every trial in one pass."""
    import bench
    from test_gpu_round3 import _dense_dual_reference
    inf = funs_mod.inference
    q, p, T, R = 500, 20, 1000, 256
    true, Ys = bench.synth_shard(q, p, T, R, 12, 0)
    par = {'C': true['C'], 'd': true['d'], 'tau': np.linspace(0.1, 0.5, p)}
    exp = Experiment([y for y in Ys], 10.0)
    halves = [Experiment(Ys[:128], 10.0), Experiment(Ys[128:], 10.0)]
    assert inf.DUAL_SOLVER == 'fixedpoint'
    monkeypatch.setattr(inf, 'DUAL_F32', False)
    funs_mod._session.drop_sessions()

    def run(e, **kw):
        infRes, nll, vlb, opt = inf.dualVariational(e, dict(par), optimizeLogLambda=False, **kw)
        ctx = infRes.session.ctx
        ctx.mstep_precomp()
        return infRes, nll, vlb, opt, ctx.pautosum()

    # (4) first half: plans the workspace for 128 slots.  One session per Experiment object, so the halves run in sessions of their own -
    # the re-plan is exercised inside the 256-trial session below by a 128-trial call on ITS context first.
    P_half, vlb_half, n_half = [], [], []
    import gc
    gc.collect()
    free_gb = _free_hbm_gb()
    print('config 5, 256 trials: %.1f GB of HBM free at the start' % free_gb)
    for h in halves:
        ir, nll_h, vlb_h, _, P = run(h)
        cx = ir.session.ctx
        print('config 5 half: chunk_trials %g, plan_lowrank %g, rank %g, HBM free %.1f / %.1f GB, this context %.1f GB (arena %.1f GB)'
              % (cx.info('chunk_trials'), cx.info('plan_lowrank'), cx.info('lowrank_rtot'), cx.info('hbm_bytes_free') / 1e9, cx.info('hbm_bytes_total') / 1e9,
                 cx.info('hbm_bytes_allocated') / 1e9, cx.info('arena_bytes') / 1e9))
        assert ir.session.ctx.info('chunk_trials') == 128.0 and ir.session.ctx.info('plan_lowrank') == 1.0
        P_half.append(P); vlb_half.append(vlb_h); n_half.append(nll_h)
        funs_mod._session.drop_sessions()
    # the 256-trial session: a 128-trial fixed point on its context first (plan for 128), then the whole list (re-plan to 256)
    sess, _ = funs_mod._session.session_for(exp, p)
    sess.set_params(par)
    ctx = sess.ctx
    ctx.set_option('dual_lowrank', 1); ctx.set_option('dual_f32', 0)
    first = np.arange(128, dtype=np.int32)
    _, fopt_a, passes_a, st_a = ctx.dual_fixed_point(first, None, want_rho=False)
    assert ctx.info('chunk_trials') == 128.0 and np.all(st_a == 0)
    ctx.dual_finalize(first, None)
    pm_a = ctx.post_mean(first)
    infRes, nll, vlb, opt, P_all = run(exp)
    assert infRes.session is sess and ctx.info('chunk_trials') == 256.0 and ctx.info('plan_lowrank') == 1.0
    iters = infRes.dual_iterations.copy()
    assert np.all(iters >= 2) and np.all(iters <= 12)
    pm = ctx.post_mean(np.arange(R, dtype=np.int32))
    assert np.max(np.abs(pm[:128] - pm_a)) <= 1e-7
    cost_all, _ = ctx.dual_costgrad_batch(first, ctx.dual_lambda(first))
    assert np.max(np.abs(cost_all - fopt_a) / np.abs(fopt_a)) <= 1e-10
    scale = np.max(np.abs(P_all))
    # (every trial's fixed point stops at 1e-8 in its offsets, and the mode searches of a chunk share one preconditioner: lambda of a trial agrees to
    #  ~1e-8 between two chunkings, not to rounding - measured 1.2e-9 of the largest entry)
    assert np.max(np.abs(P_all - (P_half[0] + P_half[1]))) <= 1e-8 * scale
    assert abs(vlb - 0.5 * (vlb_half[0] + vlb_half[1])) <= 1e-10 * abs(vlb) and abs(nll - 0.5 * (n_half[0] + n_half[1])) <= 1e-10 * abs(nll)
    # (3) structured identity on all trials, lambda read back in blocks of 32 trials (4 MB each)
    K = orc.make_K(par['tau'], T, 10.0)
    worst = 0.0
    for c0 in range(0, R, 32):
        idx = np.arange(c0, c0 + 32, dtype=np.int32)
        lam = ctx.dual_lambda(idx)
        for j, r in enumerate(idx):
            v = par['C'].T @ (lam[j].reshape(q, T) - Ys[r])
            worst = max(worst, rel(pm[r], -np.einsum('kts,ks->kt', K, v)))
    assert worst <= 1e-9
    # (2) dense numpy on two sampled trials
    for r in (0, R - 1):
        lam_r = np.asarray(opt[r], dtype=np.float64)
        cost, grad, mean, blocks = _dense_dual_reference(par['C'], par['d'], par['tau'], Ys[r].astype(float).reshape(-1), lam_r, T, 10.0)
        c_dev, _ = ctx.dual_costgrad_batch(np.array([r], dtype=np.int32), lam_r[None])
        print('config 5, 256 trials, trial %d vs dense numpy: cost %.2e, numpy max |dual gradient| %.2e, blocks %.2e, mean %.2e'
              % (r, abs(c_dev[0] - cost) / abs(cost), np.max(np.abs(grad)), rel(infRes['post_vsm'][r], blocks), rel(pm[r], mean)))
        assert np.max(np.abs(grad)) <= 1e-6
        assert abs(c_dev[0] - cost) <= 1e-8 * abs(cost)
        assert rel(infRes['post_vsm'][r], blocks) <= 1e-7 and rel(pm[r], mean) <= 1e-9
    # (6) warm restart from the resident optimum: one pass each
    ir_w, nll_w, vlb_w, _ = inf.dualVariational(exp, dict(par), optimizeLogLambda=False, prevOptimRes=opt)
    assert np.all(ir_w.dual_iterations == 1) and abs(vlb_w - vlb) <= 1e-9 * abs(vlb)
    # (5) mixed precision on the same trials (same session: the plan stays)
    monkeypatch.setattr(inf, 'DUAL_F32', True)
    ir_m, nll_m, vlb_m, opt_m = inf.dualVariational(exp, dict(par), optimizeLogLambda=False)
    assert np.all(ir_m.dual_iterations <= 12)
    assert abs(vlb_m - vlb) <= 1e-5 * abs(vlb) and abs(nll_m - nll) <= 1e-5 * abs(nll)
    ctx.set_option('dual_f32', 0)
    pick = np.array([0, 100, R - 1], dtype=np.int32)
    _, g64 = ctx.dual_costgrad_batch(pick, ctx.dual_lambda(pick))
    print('config 5, 256 trials: passes f64 %s..%s, mixed %s..%s; FP64 dual gradient at the mixed optimum %.2e'
          % (iters.min(), iters.max(), ir_m.dual_iterations.min(), ir_m.dual_iterations.max(), np.max(np.abs(g64))))
    assert np.max(np.abs(g64)) <= 1e-5
    funs_mod._session.drop_sessions()


def test_fixed_point_hands_back_a_trial_whose_mode_search_fails(funs_mod):
    """a mode search that does not settle inside the variational fixed point (here: the outer Newton cap set to one
    iteration) must hand ITS trials back (status 2) from a finite lambda instead of failing the whole call; inference.dualVariational then
    finishes them with the device L-BFGS, and the result is the optimum the unrestricted run finds (bound 1e-6 rel)."""
    g = np.load(__import__('os').path.join(__import__('os').path.dirname(__file__), 'golden', 'var_toy.npz'))
    Y = g['Y']
    par = {'C': g['init_C'], 'd': g['init_d'], 'tau': g['init_tau']}
    R, q, T = Y.shape
    p = par['C'].shape[1]
    from funs import _hip
    ctx = _hip.Context(q, p, T, R, float(g['binSize']))
    try:
        ctx.upload_counts(Y)
        ctx.set_params(par['C'], par['d'], par['tau'])
        _, f_ref, passes, st = ctx.dual_fixed_point(None, None, want_rho=False)
        assert np.all(st == 0)
        lam_ref = ctx.dual_lambda()
        ctx.set_option('pcg_outer_max', 1)                 # the shared Newton-PCG gives up after one outer iteration ...
        ctx.set_option('newton_max_iter', 1)               # ... and so does the per-trial Newton: the cold mode search cannot settle
        _, f_b, passes_b, st_b = ctx.dual_fixed_point(None, None, want_rho=False)
        assert np.all(st_b == 2), st_b                     # handed back, not an exception
        lam_b = ctx.dual_lambda()
        assert np.all(np.isfinite(lam_b)) and np.all(lam_b > 0)
        ctx.set_option('pcg_outer_max', 12); ctx.set_option('newton_max_iter', 60)
        rho, f_l, it_l = ctx.dual_lbfgs(np.arange(R, dtype=np.int32), np.log(lam_b))
        assert np.max(np.abs(f_l - f_ref) / np.abs(f_ref)) <= 1e-5
    finally:
        ctx.close()


def test_dual_variables_survive_an_interleaved_laplace_estep(funs_mod):
    """varOptimRes of a variational E-step that went through pgpfa_dual_finalize(lam) (the hand-back path and the
    'device' / 'scipy' solvers) stays readable - and usable as prevOptimRes - after a Laplace E-step on the same trials, as the reference's
    plain arrays do (inference.py:326, 398)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'var_toy.npz'))
    Y = g['Y']
    R, q, T = Y.shape
    from funs import _hip
    ctx = _hip.Context(q, g['init_C'].shape[1], T, R, float(g['binSize']))
    try:
        ctx.upload_counts(Y)
        ctx.set_params(g['init_C'], g['init_d'], g['init_tau'])
        idx = np.arange(R, dtype=np.int32)
        lam = np.exp(0.1 * np.random.default_rng(0).standard_normal((R, q * T))) * 0.5
        ctx.dual_finalize(idx, lam)                       # lam given by the caller: kept in lam_keep
        assert np.array_equal(ctx.dual_lambda(idx), lam)
        ctx.estep_laplace()                               # supersedes the posterior, not the dual variables
        assert np.array_equal(ctx.dual_lambda(idx), lam)
        _, fopt, passes, st = ctx.dual_fixed_point(idx, None, resident=True, want_rho=False)      # start = 3 from those variables
        assert np.all(st == 0)
        # new counts drop them
        ctx.upload_counts(Y)
        with pytest.raises(_hip.HipBackendError):
            ctx.dual_lambda(idx)
    finally:
        ctx.close()


@pytest.mark.parametrize('tau_ms,expect', [(10.0, 'dense'), (30.0, 'lowrank'), (300.0, 'lowrank')])
def test_auto_plan_picks_the_faster_covariance_engine(tau_ms, expect):
    """Both ends of the design at 100 neurons x 5 latents x 400 bins, 64 trials: with every timescale at ONE bin the pivoted Cholesky of the Gram
    matrices has full rank and the low-rank engine costs more than the dense factorisation it replaces; at 3 bins (rank 1520 of 2000) it is still
    ahead - measured, which is why lowrank_pays() takes it up to a flop ratio of 1.7 since round 5 - and at 30 bins it costs a fraction.
    The auto plan (cov_mode 0; util.py:599-619 sets the rank through the timescales) must pick the engine that is faster when each is forced
    (margin 1.25 on the best of three warm E-steps), and all three must agree on the result (modes 1e-8, covariance blocks 1e-8 rel)."""
    import time
    from funs import _hip
    q, p, T, R = 100, 5, 400, 64
    rng = np.random.default_rng(7)
    C, d = rng.random((q, p)) - 0.5, -1.0 - 2.0 * rng.random(q)
    tau = np.full(p, tau_ms * 1e-3)
    Y = rng.poisson(np.exp(d)[None, :, None] * np.ones((R, q, T))).astype(np.uint8)
    out = {}
    for name, mode in (('auto', 0), ('dense', 1), ('lowrank', 2)):
        ctx = _hip.Context(q, p, T, R, 10.0)
        try:
            ctx.upload_counts(Y)
            ctx.set_option('cov_mode', mode)
            ctx.set_params(C, d, tau)
            obj, _, st = ctx.estep_laplace()
            assert np.all(st == 0)
            best = 1e9
            for _ in range(3):
                t0 = time.time()
                obj, _, st = ctx.estep_laplace(warm_start=True)
                best = min(best, time.time() - t0)
            out[name] = (best, 'lowrank' if ctx.info('last_cov_lowrank') else 'dense', ctx.post_mean().copy(), ctx.post_vsm().copy(), ctx.info('lowrank_rtot'))
        finally:
            ctx.close()
    print('tau %.0f ms (rank %d of %d): auto -> %s %.1f ms; dense %.1f ms, low-rank %.1f ms'
          % (tau_ms, out['auto'][4], p * T, out['auto'][1], out['auto'][0] * 1e3, out['dense'][0] * 1e3, out['lowrank'][0] * 1e3))
    assert out['dense'][1] == 'dense' and out['lowrank'][1] == 'lowrank' and out['auto'][1] == expect
    assert out['auto'][0] <= 1.25 * min(out['dense'][0], out['lowrank'][0])
    for name in ('dense', 'lowrank'):
        assert np.max(np.abs(out[name][2] - out['auto'][2])) <= 1e-8 and rel(out[name][3], out['auto'][3]) <= 1e-8


# ---------------------------------------------------------------------------------------------------------------
# Yt = F L^-T and the mixing pass as one kernel
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dims', [(20, 1, 40, 3, 0.1, 0.1), (30, 2, 70, 4, 0.05, 0.4), (25, 3, 16, 5, 0.03, 0.06), (40, 4, 130, 12, 0.05, 0.3), (200, 10, 500, 16, 0.02, 0.12), (200, 10, 500, 16, 0.1, 0.5), (60, 8, 333, 8, 0.01, 0.05),
                                  (60, 6, 333, 8, 0.1, 0.6), (50, 7, 200, 8, 0.05, 0.2), (80, 9, 150, 8, 0.05, 0.3), (60, 12, 120, 8, 0.05, 0.3)])
def test_product_and_mixing_in_one_kernel_match_the_two_passes(dims):
    """`yt_mix = 1` (default): under the split covariance form the tiles of Yt = F L^-T of all latents are formed on the FP64 matrix cores, mixed against G_t in
    registers and only the correction D and post_vsm are written (csrc/ytmix.h) - against `yt_mix = 0`, the batched product followed by the mixing
    pass.  Same products, sums over K and over the columns in another order: E-step objective identical (the covariance phase does not enter it),
    post_vsm 1e-12, PautoSum 1e-10 of the largest entry (its FP16 term sees D rounded from values that differ in the last FP64 bits).  Shapes: bins not a
    multiple of the 128-bin workgroup (down to 16 bins and one latent: a workgroup mostly empty, a single column block), rank totals above the 256-row panel chunk (two chunks per column block), unequal ranks per latent (timescales
    spread 6 x), 7 and 9 latents - run by the 8- and 10-wide instantiations with an empty last latent - and 12, beyond the kernel's widths: the
    two-pass route, and `last_yt_mix_fused` says so."""
    from funs import _hip
    q, p, T, R, tau_lo, tau_hi = dims
    import bench
    true, Ys = bench.synth_shard(q, p, T, R, 12, 0)
    Y = np.stack(Ys)
    C, d = true['C'], true['d']
    tau = np.linspace(tau_lo, tau_hi, p)
    out = {}
    for fused in (1, 0):
        ctx = _hip.Context(q, p, T, R, 10.0)
        try:
            ctx.upload_counts(Y)
            ctx.set_option('cov_mode', 2)
            ctx.set_option('yt_mix', fused)
            ctx.set_params(C, d, tau)
            obj, _, status = ctx.estep_laplace()
            assert np.all(status == 0) and ctx.info('last_split_cov') == 1.0 and ctx.info('plan_lowrank') == 1.0
            ctx.mstep_precomp()
            out[fused] = (obj, ctx.post_vsm().copy(), ctx.pautosum().copy(), ctx.info('lowrank_rtot'), ctx.info('last_yt_mix_fused'))
        finally:
            ctx.close()
    b = out[0]
    assert b[4] == 0.0
    for fused in (1,):
        a = out[fused]
        print(f'\nfused product + mixing (yt_mix = {fused}) at {dims}: rank total {a[3]:.0f}, kernel {a[4]:.0f}, post_vsm '
              f'{np.max(np.abs(a[1] - b[1])) / np.max(np.abs(b[1])):.2e}, PautoSum {np.max(np.abs(a[2] - b[2])) / np.max(np.abs(b[2])):.2e}')
        # which route ran: 1 = one kernel, 0 = product and mixing pass apart
        want = float(fused) if p <= 10 else 0.0
        assert a[4] == want
        assert a[0] == b[0]
        assert np.max(np.abs(a[1] - b[1])) <= 1e-12 * np.max(np.abs(b[1]))
        assert np.max(np.abs(a[2] - b[2])) <= 1e-10 * np.max(np.abs(b[2]))
    if (q, p) == (200, 10):
        assert b[3] > 256


# ---------------------------------------------------------------------------------------------------------------
# post_vsm for 11..20 latents on the 4 x 4 x 4 block shape of the matrix cores
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('dims', [(60, 12, 150, 6), (80, 14, 200, 6), (70, 16, 132, 5), (90, 17, 210, 6), (100, 20, 333, 8), (100, 20, 328, 8)])
@pytest.mark.parametrize('f32', [0, 1])
def test_block_form_of_the_covariance_blocks_matches_the_padded_tiles(dims, f32):
    """`vsm_b4 = 1` (default): post_vsm[t] for 11..20 latents as 4 x 4 x 4 block products - four bins per instruction, instructions over the lower pairs of
    four-latent blocks (model.h, post_vsm_b4_kernel) - against `vsm_b4 = 0`, the 16 x 16 x 4 tiles padded to 16 or 32 rows, through one variational
    fixed point (the blocks feed the variance offsets of every pass, so a wrong entry would move the optimum): dual objective 1e-12 rel, posterior
    means 1e-10, covariance blocks 1e-12 of the largest entry.  FP64 and mixed precision (single-precision panel); latent counts that fill the last
    block (12, 16, 20) and that do not (14, 17); `vsm_b4 = 2` (64 bins per workgroup, 16-byte loads) where the bin count is a multiple of the vector
    width (150 and 210 are not in single precision: those shapes run the scalar form under either value); bins not a multiple of the workgroup."""
    from funs import _hip
    import bench
    q, p, T, R = dims
    true, Ys = bench.synth_shard(q, p, T, R, 12, 0)
    Y = np.stack(Ys)
    tau = np.linspace(0.08, 0.4, p)
    idx = np.arange(R, dtype=np.int32)
    out = {}
    for b4 in (2, 1, 0):
        ctx = _hip.Context(q, p, T, R, 10.0)
        try:
            ctx.upload_counts(Y)
            ctx.set_option('vsm_b4', b4)
            ctx.set_option('dual_lowrank', 1); ctx.set_option('dual_f32', f32)
            ctx.set_params(true['C'], true['d'], tau)
            _, fopt, passes, status = ctx.dual_fixed_point(idx, None, want_rho=False)
            # (in single precision a trial may stall above the 1e-8 bar and be handed back - status 2 - whichever kernel forms the blocks: the
            #  two runs must agree on that too)
            assert np.all(status == 0) or f32
            ctx.dual_finalize(idx, None)
            out[b4] = (fopt.copy(), ctx.post_mean(idx).copy(), ctx.post_vsm(idx).copy(), passes.copy(), status.copy())
        finally:
            ctx.close()
    b = out[0]
    for b4 in (2, 1):
        a = out[b4]
        print(f'\nblock form (vsm_b4 = {b4}) at {dims}, f32 {f32}: objective {np.max(np.abs(a[0] - b[0]) / np.abs(b[0])):.2e}, means {np.max(np.abs(a[1] - b[1])):.2e}, '
              f'blocks {np.max(np.abs(a[2] - b[2])) / np.max(np.abs(b[2])):.2e}, passes {a[3].max()} / {b[3].max()}')
        assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
        assert np.max(np.abs(a[0] - b[0]) / np.abs(b[0])) <= 1e-12
        assert np.max(np.abs(a[1] - b[1])) <= 1e-10
        assert np.max(np.abs(a[2] - b[2])) <= 1e-12 * np.max(np.abs(b[2]))


@pytest.mark.parametrize('T', [500, 301, 258, 700])
def test_pivoted_cholesky_with_two_bins_per_thread_finds_the_same_factors(T):
    """rbf_pivchol2_kernel (beyond 256 bins: two bins per row thread, four column groups, pivot search on the diagonal in registers) against
    rbf_pivchol_kernel: same ranks (same pivots: the largest remaining diagonal entry, lowest index on ties), and - the sums being grouped
    differently - E-step results that agree to rounding: objective 1e-12 relative, modes 1e-9, PautoSum 1e-10 of its largest entry.  301 bins:
    the last row pair is half empty; 700 bins: two trips of 512 rows per step; timescales from 2 to 60 bins: ranks from a handful to most of the bins."""
    from funs import _hip
    import bench
    q, p, R = 40, 4, 8
    true, Ys = bench.synth_shard(q, p, T, R, 9, 0)
    Y = np.stack(Ys)
    tau = np.array([0.02, 0.07, 0.2, 0.6])
    out = {}
    for pairs in (1, 0):
        ctx = _hip.Context(q, p, T, R, 10.0)
        try:
            ctx.upload_counts(Y)
            ctx.set_option('pivchol_pairs', pairs)
            ctx.set_option('cov_mode', 2)
            ctx.set_params(true['C'], true['d'], tau)
            obj, _, status = ctx.estep_laplace()
            assert np.all(status == 0) and ctx.info('plan_lowrank') == 1.0
            ctx.mstep_precomp()
            out[pairs] = (obj, ctx.post_mean().copy(), ctx.pautosum().copy(), ctx.info('lowrank_rtot'))
        finally:
            ctx.close()
    a, b = out[1], out[0]
    assert a[3] == b[3] and a[3] > 0
    assert abs(a[0] - b[0]) <= 1e-12 * abs(b[0])
    assert np.max(np.abs(a[1] - b[1])) <= 1e-9
    assert np.max(np.abs(a[2] - b[2])) <= 1e-10 * np.max(np.abs(b[2]))


def test_cd_newton_from_an_extrapolated_start_finds_the_same_minimiser(funs_mod, monkeypatch):
    """Batch EM with CdOptimMethod='newton': from the second M-step on the (C,d) iteration starts one previous displacement ahead of the
    parameters it is handed (learning._newton_cd).  The minimiser of the convex per-neuron cost does not depend on the start: five EM
    iterations with and without give the same parameters (1e-8; each M-step stops on a predicted error of 1e-10) and objectives (1e-10
    relative), in no more device passes."""
    funs = funs_mod
    import bench
    q, p, T, R = 60, 4, 120, 48
    true, Ys = bench.synth_shard(q, p, T, R, 5, 0)
    runs = {}
    for ext in (2, 1, 0):
        monkeypatch.setattr(funs.learning, 'CD_EXTRAPOLATE', ext)
        exp = bench.Shard(Ys, 10.0)
        sess, _ = funs._session.session_for(exp, p)
        params = {'C': true['C'] * 0.8, 'd': true['d'] + 0.1, 'tau': np.full(p, 0.2)}
        optim, hist, passes = None, [], 0
        for it in range(5):
            infRes, nll, optim = funs.inference.laplace(exp, params, prevOptimRes=optim)
            params, _ = funs.learning.updateParams(params, infRes, exp, CdOptimMethod='newton')
            passes += sum(sess._cd_passes)
            hist.append((float(nll), params['C'].copy(), params['d'].copy(), params['tau'].copy()))
        runs[ext] = (hist, passes)
        funs._session.drop_sessions(exp)
    assert runs[2][1] <= runs[0][1] and runs[1][1] <= runs[0][1], [runs[k][1] for k in (2, 1, 0)]
    for a, b in zip(runs[2][0] + runs[1][0], runs[0][0] + runs[0][0]):
        assert abs(a[0] - b[0]) <= 1e-10 * abs(b[0])
        for x, y in zip(a[1:], b[1:]):
            assert np.max(np.abs(x - y)) <= 1e-8
