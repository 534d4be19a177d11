"""Round 4: the inner PCG without the prior mat-vec (pcg.h: Chronopoulos-Gear step, Kt^-1 z = r - Wb z) and the variational fixed point
(pgpfa_dual_fixed_point) against the oracle's restatement of the reference (inference.py:12-65, 188-219), against the forms they replace,
and at the instantiation edges (latent widths that are not a template width, bins that are not a multiple of the 64-bin tile)."""
import numpy as np
import pytest

from conftest import load_golden
from oracle import pgpfa_oracle as orc

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(1e-300, np.max(np.abs(b)))


@pytest.fixture(scope='module')
def funs_mod():
    import funs
    return funs


# ---------------------------------------------------------------------------------------------------------------
# inner PCG: two-kernel step without K^-1 in the loop
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('shape', [(40, 7, 70, 6), (33, 9, 130, 5), (25, 10, 64, 3), (30, 3, 100, 8), (50, 1, 90, 4),
                                   (40, 12, 70, 5), (45, 17, 100, 9), (50, 20, 64, 3), (36, 14, 37, 10)])
def test_pcg_step_without_prior_matvec_vs_oracle(shape):
    """Round 5: pcg_form = 2 (the default: private vectors on line-aligned rows - 70 / 130 / 37 bins pad to 80 / 144 / 48 -, start kernel, closing inside
    kernel A, one upload per solve) next to forms 1 and 0; and 11 .. 20 latents, where the step runs in pcgw_*_kernel (32-bin tiles, two slots per
    wave, packed triangles in row chunks: widths 12, 14 -> 16, 17 -> 20, 20; 9 and 10 slots leave a partly filled group of 8) against the oracle and
    against the split kernels of round 3.
    Forced on at small shapes (it is chosen by itself only for large chunks): latent widths 7 and 9 run the 8- and 10-wide
    instantiations with clamped component indices, 70 / 130 bins leave a partly filled 64-bin tile.  Modes against the oracle's exact
    Newton (1e-8), objective 1e-9 rel, covariance blocks 1e-8 rel; the split kernels of round 3 (pcg_form = 0) on the same problem land on
    the same modes, and so do the products of the preconditioner through the general GEMM kernel (thin_products = 0, or 1: only `Sb u`) instead
    of thin.h's kernels (bins that are not a multiple of 4 take their scalar-load instantiation; ranks below 64 leave row tiles empty)."""
    from funs import _hip
    q, p, T, R = shape
    rng = np.random.default_rng(q * 1000 + p)
    _, Ys, _ = orc.synth_dataset(q, p, T, R, seed=p, dOffset=0.0)
    Y = np.stack(Ys).astype(np.uint8)
    par = {'C': 0.3 * rng.standard_normal((q, p)) / np.sqrt(max(1, p / 4)), 'd': np.log(Y.mean(axis=(0, 2)) + 0.1), 'tau': 0.05 + 0.3 * rng.random(p)}
    res, nll_o, _ = orc.laplace([y.astype(float) for y in Ys], par, 10.0, mode='exact', return_cov=False)
    modes = {}
    # (form 2 twice: with the solve's z, s, p, q, t / y stored in single precision - the default, key (2, 2) - and all FP64, key (2, -2))
    for form, thin in (((2, 2), (2, -2), (1, 2), (0, 2), (1, 0), (1, 1)) if p <= 10 else ((2, 2), (2, -2), (0, 2))):
        ctx = _hip.Context(q, p, T, R, 10.0)
        try:
            ctx.upload_counts(Y)
            ctx.set_option('cov_mode', 2)
            ctx.set_option('pcg_fused', 2)
            ctx.set_option('pcg_form', form)
            ctx.set_option('thin_products', abs(thin))
            ctx.set_option('pcg_vec32', 1 if thin > 0 else 0)
            ctx.set_params(par['C'], par['d'], par['tau'])
            obj, _, status = ctx.estep_laplace()
            assert np.all(status == 0) and ctx.info('plan_lowrank') == 1.0 and ctx.info('last_pcg_iterations') > 0
            assert abs(-obj / R - nll_o) <= 1e-9 * abs(nll_o)
            modes[form, thin] = ctx.post_mean().copy()
            assert np.max(np.abs(modes[form, thin] - np.stack(res['post_mean']))) <= 1e-8
            assert rel(ctx.post_vsm(), np.stack(res['post_vsm'])) <= 1e-8
            obj2, _, st2 = ctx.estep_laplace(warm_start=True)
            assert np.all(st2 == 0) and abs(obj2 - obj) <= 1e-10 * abs(obj)
        finally:
            ctx.close()
    assert np.max(np.abs(modes[2, 2] - modes[0, 2])) <= 2e-9
    assert np.max(np.abs(modes[2, -2] - modes[0, 2])) <= 2e-9
    if p <= 10:
        assert np.max(np.abs(modes[1, 2] - modes[0, 2])) <= 2e-9
        assert np.max(np.abs(modes[1, 2] - modes[1, 0])) <= 2e-9
        assert np.max(np.abs(modes[1, 2] - modes[1, 1])) <= 2e-9


def test_pcg_forms_agree_at_config3_dimensions():
    """200 neurons x 10 latents x 500 bins, 96 trials (the chunk is large enough for the host-free iteration by itself): the step without the
    prior mat-vec against the split kernels of round 3 - same modes (1e-8: both stop on a predicted error of 1e-9), same objective (1e-11
    rel), same PautoSum (1e-9 of its largest entry); and every mode is stationary for the reference's log-posterior restated in structured
    numpy (gradient <= 1e-6)."""
    from funs import _hip
    import bench
    q, p, T, R = 200, 10, 500, 96
    true, Ys = bench.synth_shard(q, p, T, R, 12, 0)
    Y = np.stack(Ys)
    par = {'C': true['C'], 'd': true['d'], 'tau': np.linspace(0.1, 0.5, p)}
    out = {}
    for form in (2, 1, 0):
        ctx = _hip.Context(q, p, T, R, 10.0)
        try:
            ctx.upload_counts(Y)
            ctx.set_option('pcg_form', form)
            ctx.set_params(par['C'], par['d'], par['tau'])
            obj, _, status = ctx.estep_laplace()
            assert np.all(status == 0) and ctx.info('plan_lowrank') == 1.0
            ctx.mstep_precomp()
            out[form] = (obj, ctx.post_mean().copy(), ctx.pautosum().copy(), ctx.info('last_pcg_iterations'))
        finally:
            ctx.close()
    for form in (2, 1):
        assert abs(out[form][0] - out[0][0]) <= 1e-11 * abs(out[0][0]) and np.max(np.abs(out[form][1] - out[0][1])) <= 1e-8
    a, b = out[2], out[0]
    assert abs(a[0] - b[0]) <= 1e-11 * abs(b[0])
    assert np.max(np.abs(a[1] - b[1])) <= 1e-8
    assert np.max(np.abs(a[2] - b[2])) <= 1e-9 * np.max(np.abs(b[2]))
    assert a[3] <= 1.3 * b[3]                       # (the model matrix of the inner solve does not cost iterations)
    # stationarity in the reference's log-posterior (inference.py:30-47), structured: C^T(exp(Cx+d) - y) + K^-1 x
    K = orc.make_K(par['tau'], T, 10.0)
    Kinv = np.linalg.inv(K)
    for r in (0, 17, 95):
        X = a[1][r].reshape(p, T)
        g = par['C'].T @ (np.exp(par['C'] @ X + par['d'][:, None]) - Y[r]) + np.einsum('kts,ks->kt', Kinv, X)
        assert np.max(np.abs(g)) <= 1e-6


# ---------------------------------------------------------------------------------------------------------------
# variational fixed point
# ---------------------------------------------------------------------------------------------------------------
def _dense(par, T, bin_ms):
    K_big = orc.make_K_big(orc.make_K(par['tau'], T, bin_ms))
    C_big, d_big = orc.make_Cd_big(par['C'], par['d'], T)
    return K_big, np.linalg.inv(K_big), C_big, d_big


@pytest.mark.parametrize('engine', ['dense', 'lowrank'])
def test_fixed_point_is_the_zero_of_the_reference_dual_gradient(c1, engine):
    """pgpfa_dual_fixed_point on config-1 data, both covariance engines, cold start (lambda = 0.5, the reference's, inference.py:302):
    at the returned lambda the oracle's restatement of dualProblem_grad (inference.py:215-219; dense matrices, jitter included) vanishes
    to 1e-7 in the max-norm, fopt is the oracle's dualProblem there (1e-9 rel), the device L-BFGS started anywhere does not get below it,
    and a restart from the optimum is settled after one pass."""
    from funs import _hip
    q, p, T = 30, 3, 100
    idx = np.array([0, 5, 11, 19], dtype=np.int32)
    par = {'C': c1['init_C'], 'd': c1['init_d'], 'tau': c1['init_tau']}
    K_big, Kinv_big, C_big, d_big = _dense(par, T, c1['binSize'])
    ctx = _hip.Context(q, p, T, 20, c1['binSize'])
    try:
        ctx.upload_counts(c1['Y'])
        ctx.set_option('cov_mode', 2 if engine == 'lowrank' else 1)
        ctx.set_option('dual_lowrank', 1 if engine == 'lowrank' else 0)
        ctx.set_params(par['C'], par['d'], par['tau'])
        rho, fopt, passes, status = ctx.dual_fixed_point(idx, np.full((4, q * T), np.log(0.5)))
        assert ctx.info('plan_lowrank') == float(engine == 'lowrank')
        assert np.all(status == 0) and np.all(passes >= 2) and np.all(passes <= 12)
        for i, tr in enumerate(idx):
            ybar = c1['Ys'][tr].reshape(-1)
            lam = np.exp(rho[i])
            g = orc.dual_grad(lam, ybar, C_big, K_big, Kinv_big, d_big)
            assert np.max(np.abs(g)) <= 1e-7
            f = orc.dual_cost(lam, ybar, C_big, K_big, Kinv_big, d_big)
            assert abs(fopt[i] - f) <= 1e-9 * abs(f)
        rho_l, fopt_l, iters = ctx.dual_lbfgs(idx, np.full((4, q * T), np.log(0.5)))
        assert np.all(fopt <= fopt_l + 1e-9 * np.abs(fopt_l))
        assert np.max(np.abs(fopt - fopt_l)) <= 1e-2             # (where scipy's decrease test stops: BASELINE.md section 2)
        rho2, fopt2, passes2, status2 = ctx.dual_fixed_point(idx, rho, warm=True)
        assert np.all(status2 == 0) and np.all(passes2 == 1)
        assert np.max(np.abs(rho2 - rho)) <= 1e-7 and np.max(np.abs(fopt2 - fopt)) <= 1e-9 * np.max(np.abs(fopt))
        # the cold start set on the device (rho0 = None: lambda = 0.5, nothing uploaded) is the same start; lambda comes back beside rho, and the
        # finalize call takes the optimum from the device (lam = None) - same posterior as from the host copy
        rho4, fopt4, passes4, status4, lam4 = ctx.dual_fixed_point(idx, None, want_lam=True)
        assert np.array_equal(rho4, rho) and np.array_equal(fopt4, fopt) and np.array_equal(passes4, passes)
        assert np.max(np.abs(lam4 - np.exp(rho4))) <= 1e-15 * np.max(lam4)
        nlp_dev = ctx.dual_finalize(idx, None)
        pm_dev, vs_dev = ctx.post_mean(idx).copy(), ctx.post_vsm(idx).copy()
        nlp_host = ctx.dual_finalize(idx, lam4)
        assert nlp_dev == nlp_host and np.array_equal(ctx.post_mean(idx), pm_dev) and np.array_equal(ctx.post_vsm(idx), vs_dev)
        with pytest.raises(_hip.HipBackendError):
            ctx.dual_finalize(idx, None)                     # (the host copy replaced the resident optimum of these trials)
        # the pass cap is reported, not hidden
        _, _, passes3, status3 = ctx.dual_fixed_point(idx, np.full((4, q * T), np.log(0.5)), max_outer=1)
        assert np.all(status3 == 1) and np.all(passes3 == 1)
    finally:
        ctx.close()


def test_dual_variational_through_the_fixed_point_vs_reference(funs_mod):
    """inference.dualVariational with DUAL_SOLVER = 'fixedpoint' (the default) on the toy the reference's own E-step was captured on
    (tests/golden/var_toy.npz): bound / nPLL / posterior within the reference's own stopping slack (1e-3, 2e-3), the bound not above the
    one the device L-BFGS stops at, every trial converged in a handful of passes, the returned lambda a zero of the oracle's dual gradient;
    a trial the fixed point hands back (forced with one pass) is finished by L-BFGS and lands on the same bound."""
    from conftest import Experiment
    inf = funs_mod.inference
    g = load_golden('var_toy.npz')
    Ys = [g['Y'][r].astype(float) for r in range(g['Y'].shape[0])]
    exp = Experiment(Ys, float(g['binSize']))
    params = {'C': g['init_C'].copy(), 'd': g['init_d'].copy(), 'tau': g['init_tau'].copy()}
    assert inf.DUAL_SOLVER == 'fixedpoint'
    infRes, nll, vlb, opt = inf.dualVariational(exp, params)
    infRes.materialize()
    lam3_view = opt[3].copy()                              # (read now: this entry stays valid)
    assert np.all(infRes.dual_iterations >= 2) and np.all(infRes.dual_iterations <= 12)
    assert abs(vlb - float(g['estep_vlb'])) <= 1e-3 and abs(nll - float(g['estep_nll'])) <= 1e-3
    K_big, Kinv_big, C_big, d_big = _dense(params, Ys[0].shape[1], float(g['binSize']))
    for r in (0, 4, 9):
        assert np.max(np.abs(infRes['post_mean'][r] - g['estep_post_mean'][r])) <= 2e-3
        assert rel(infRes['post_vsm'][r], g['estep_post_vsm'][r]) <= 2e-3
        assert np.max(np.abs(orc.dual_grad(opt[r], Ys[r].reshape(-1), C_big, K_big, Kinv_big, d_big))) <= 1e-7
    try:
        inf.DUAL_SOLVER = 'device'
        _, nll_l, vlb_l, _ = inf.dualVariational(exp, params)
        inf.DUAL_SOLVER = 'fixedpoint'
        old = inf.DUAL_FP_MAX_PASSES
        inf.DUAL_FP_MAX_PASSES = 1
        ir_b, nll_b, vlb_b, _ = inf.dualVariational(exp, params)
        inf.DUAL_FP_MAX_PASSES = old
    finally:
        inf.DUAL_SOLVER = 'fixedpoint'
    assert vlb <= vlb_l + 1e-9 and abs(vlb - vlb_l) <= 1e-3
    assert np.all(ir_b.dual_iterations > 1) and abs(vlb_b - vlb) <= 1e-3
    # the returned varOptimRes is a view of device state: entries read before the later runs above stay what they were, the others now
    # belong to a superseded E-step and say so
    assert np.array_equal(opt[3], lam3_view)
    with pytest.raises(funs_mod._hip.HipBackendError):
        opt[7]
    infRes, nll, vlb, opt = inf.dualVariational(exp, params)
    opt.materialize()
    # warm start from the optimum (prevOptimRes, engine.py:200): one pass, same numbers - from a host list of arrays (what the reference passes) ...
    host_opt = [np.array(o) for o in opt]
    ir_w, nll_w, vlb_w, opt_w = inf.dualVariational(exp, params, prevOptimRes=host_opt)
    assert np.all(ir_w.dual_iterations == 1) and abs(vlb_w - vlb) <= 1e-9 * abs(vlb) and abs(nll_w - nll) <= 1e-8 * abs(nll)
    # ... and from the lazy device-resident list the call itself returned (no bytes move)
    assert isinstance(opt_w, funs_mod._session.DeviceDualOptimRes)
    ir_r, nll_r, vlb_r, opt_r = inf.dualVariational(exp, params, prevOptimRes=opt_w)
    assert np.all(ir_r.dual_iterations == 1) and abs(vlb_r - vlb) <= 1e-9 * abs(vlb)
    assert np.max(np.abs(opt_r[2] - host_opt[2])) <= 1e-7 * np.max(host_opt[2])
    funs_mod._session.drop_sessions()


# ---------------------------------------------------------------------------------------------------------------
# copies as kernels / the runtime's copies; the two mixing passes
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize('option', [('copy_kernels', 0), ('mix_slot', 0), ('mix_slot', 1), ('mix_slot', 2), ('mix_slot', 3), ('poisson_tiles', 1)])
def test_alternative_paths_give_the_same_numbers(c1, option):
    """The small copies through the runtime (`copy_kernels = 0`: hipMemcpyAsync + hipStreamSynchronize instead of kernels through mapped staging and
    a sequence number) and the stand-alone mixing passes of the split accumulation (`yt_mix = 0`: Yt formed by the batched product, then `mix_slot` = 0: 64-bin
    pass, 1 / 2: a thread per bin, 3: two column halves per bin) against the defaults (product and mixing in one kernel) on the same E-step + M-step
    statistics: copies move bytes - bit-identical; the mixing passes add the same products in another order - 1e-13; one 16-bin tile per wave in
    the matrix-core Poisson pass (`poisson_tiles = 1`) instead of two: the same numbers per bin, the objective's partial sums in another order."""
    from funs import _hip
    out = []
    for variant in (False, True):
        ctx = _hip.Context(30, 3, 100, 20, c1['binSize'])
        try:
            ctx.upload_counts(c1['Y'])
            ctx.set_option('cov_mode', 2)
            if variant:
                if option[0] == 'mix_slot':
                    ctx.set_option('yt_mix', 0)
                ctx.set_option(option[0], option[1])
            ctx.set_params(c1['init_C'], c1['init_d'], c1['init_tau'])
            obj, _, status = ctx.estep_laplace()
            assert np.all(status == 0) and ctx.info('last_split_cov') == 1.0
            ctx.mstep_precomp()
            cost, grad = ctx.mstep_cd_costgrad(np.concatenate([c1['init_C'].T.reshape(-1), c1['init_d']]))
            out.append((obj, ctx.post_mean().copy(), ctx.post_vsm().copy(), ctx.pautosum().copy(), cost, grad.copy()))
        finally:
            ctx.close()
    a, b = out
    tol = 0.0 if option[0] == 'copy_kernels' else 1e-13
    assert abs(a[0] - b[0]) <= tol * abs(a[0]) and abs(a[4] - b[4]) <= tol * abs(a[4])
    for i in (1, 2, 3, 5):
        assert np.max(np.abs(a[i] - b[i])) <= tol * np.max(np.abs(a[i]))


# ---------------------------------------------------------------------------------------------------------------
# clearing only what is read of the L^-T slabs
# ---------------------------------------------------------------------------------------------------------------
def test_partial_clear_of_the_inverse_slabs_is_bitwise_the_full_clear():
    """`mt_fill = 1` clears, before a slot's L^-T is formed, only the entries the consumers of the low-rank engine read and the inverse does not write.  If
    that set is right the results do not change by a bit against clearing the whole slab - over E-steps whose timescales (and with them the rank layout of
    the slabs) change from one to the next, so that a stale entry of the previous layout would be a wrong number, not a zero: E-step objective, posterior
    means, covariance blocks and PautoSum compared with `==`."""
    from funs import _hip
    q, p, T, R = 60, 6, 200, 24
    rng = np.random.default_rng(5)
    _, Ys, _ = orc.synth_dataset(q, p, T, R, seed=3, dOffset=0.0)
    Y = np.stack(Ys).astype(np.uint8)
    C = 0.3 * rng.standard_normal((q, p))
    d = np.log(Y.mean(axis=(0, 2)) + 0.1)
    taus = [np.linspace(0.3, 0.08, p), np.linspace(0.05, 0.25, p), np.full(p, 0.12), np.linspace(0.4, 0.03, p)]
    out = {}
    for fill in (1, 0):
        ctx = _hip.Context(q, p, T, R, 10.0)
        try:
            ctx.upload_counts(Y)
            ctx.set_option('cov_mode', 2)
            ctx.set_option('mt_fill', fill)
            res = []
            for tau in taus:
                ctx.set_params(C, d, tau)
                obj, _, status = ctx.estep_laplace()
                assert np.all(status == 0) and ctx.info('plan_lowrank') == 1.0
                ctx.mstep_precomp()
                res.append((obj, ctx.post_mean().copy(), ctx.post_vsm().copy(), ctx.pautosum().copy(), ctx.info('lowrank_rtot')))
            out[fill] = res
        finally:
            ctx.close()
    assert len({r[4] for r in out[1]}) > 1                      # the rank did change between the E-steps
    for a, b in zip(out[1], out[0]):
        assert a[0] == b[0]
        for i in (1, 2, 3):
            assert np.array_equal(a[i], b[i])


@pytest.mark.parametrize('shape', [(40, 18, 130, 3), (50, 20, 200, 4), (30, 17, 64, 2)])
def test_wide_mixing_pass_with_lanes_along_the_bins(shape):
    """17..20 latents: the in-place mixing pass of the full-width covariance product with the lanes along the bins (`mix_wide = 1`, mix_vsm_wide2_kernel:
    18 and 17 latents run the 20-wide instantiation with masked rows, 130 bins leave a partly filled 64-bin tile) against the kernel it replaces
    (lanes along the latents) - the option switches the per-bin application of the shared preconditioner too (apply_bin_wide2_kernel): modes 2e-9, per-bin
    covariance blocks and PautoSum 1e-8 of their largest entry between the two, modes and blocks against the oracle's exact Newton (1e-8)."""
    from funs import _hip
    q, p, T, R = shape
    rng = np.random.default_rng(p * 7 + T)
    _, Ys, _ = orc.synth_dataset(q, p, T, R, seed=p, dOffset=0.0)
    Y = np.stack(Ys).astype(np.uint8)
    par = {'C': 0.3 * rng.standard_normal((q, p)) / np.sqrt(p / 4), 'd': np.log(Y.mean(axis=(0, 2)) + 0.1), 'tau': 0.05 + 0.3 * rng.random(p)}
    res, _, _ = orc.laplace([y.astype(float) for y in Ys], par, 10.0, mode='exact', return_cov=False)
    out = {}
    for wide in (1, 0):
        ctx = _hip.Context(q, p, T, R, 10.0)
        try:
            ctx.upload_counts(Y)
            ctx.set_option('cov_mode', 2)
            ctx.set_option('mix_wide', wide)
            ctx.set_option('split_cov', 0)                     # (the in-place pass belongs to the full-width product)
            ctx.set_params(par['C'], par['d'], par['tau'])
            obj, _, status = ctx.estep_laplace()
            assert np.all(status == 0) and ctx.info('plan_lowrank') == 1.0 and ctx.info('last_split_cov') == 0.0
            ctx.mstep_precomp()
            out[wide] = (ctx.post_vsm().copy(), ctx.pautosum().copy(), ctx.post_mean().copy())
        finally:
            ctx.close()
    # (the option also switches the per-bin application of the shared preconditioner in the inner solves: the modes agree to the stopping tolerance)
    assert np.max(np.abs(out[1][2] - out[0][2])) <= 2e-9
    for i in (0, 1):
        assert np.max(np.abs(out[1][i] - out[0][i])) <= 1e-8 * np.max(np.abs(out[0][i]))
    for wide in (1, 0):
        assert np.max(np.abs(out[wide][2] - np.stack(res['post_mean']))) <= 1e-8
        assert rel(out[wide][0], np.stack(res['post_vsm'])) <= 1e-8


def test_split_accumulation_at_20_latents():
    """The split form of the sum over trials of the covariance blocks beyond 16 latents (17..20: the mixing pass of mix_vsm_wide2_kernel<20, true>, every column of
    Yt present): PautoSum against the full-width FP64 product of the same engine 1e-9 of its largest entry, post_vsm 1e-12; at 18 latents the 20-wide
    instantiation runs with masked rows."""
    from funs import _hip
    for q, p, T, R in ((60, 20, 130, 6), (50, 18, 200, 5)):
        rng = np.random.default_rng(p)
        _, Ys, _ = orc.synth_dataset(q, p, T, R, seed=p + 1, dOffset=0.0)
        Y = np.stack(Ys).astype(np.uint8)
        par = {'C': 0.3 * rng.standard_normal((q, p)) / np.sqrt(p / 4), 'd': np.log(Y.mean(axis=(0, 2)) + 0.1), 'tau': 0.05 + 0.3 * rng.random(p)}
        out = {}
        for split in (1, 0):
            ctx = _hip.Context(q, p, T, R, 10.0)
            try:
                ctx.upload_counts(Y)
                ctx.set_option('cov_mode', 2)
                ctx.set_option('split_cov', split)
                ctx.set_params(par['C'], par['d'], par['tau'])
                obj, _, status = ctx.estep_laplace()
                assert np.all(status == 0) and ctx.info('plan_lowrank') == 1.0 and ctx.info('last_split_cov') == float(split)
                ctx.mstep_precomp()
                out[split] = (ctx.pautosum().copy(), ctx.post_vsm().copy())
            finally:
                ctx.close()
        print('%d latents: split vs FP64 product: PautoSum %.2e, post_vsm %.2e' % (p, rel(out[1][0], out[0][0]), rel(out[1][1], out[0][1])))
        assert rel(out[1][0], out[0][0]) <= 1e-9 and rel(out[1][1], out[0][1]) <= 1e-12
