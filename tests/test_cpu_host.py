"""CPU-side checks: the C-ABI library loads and exports every symbol include/pgpfa.h declares, the
host utilities reproduce the reference's layouts and RNG streams, the product fails loudly without a
GPU, and the trial-sharding arithmetic (2 ranks, gloo) reproduces the unsharded statistics."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, load_golden
from oracle import pgpfa_oracle as orc


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from funs import _hip
    lib = _hip.load_library()
    header = open(os.path.join(ROOT, 'include', 'pgpfa.h')).read()
    declared = sorted(set(re.findall(r'\b(pgpfa_[a-z0-9_]+)\s*\(', header)))
    assert declared, 'no declarations parsed'
    for name in declared:
        assert hasattr(lib, name), name
    assert sorted(_hip.EXPORTED_SYMBOLS) == declared
    assert lib.pgpfa_version() >= 100


def test_no_gpu_means_loud_failure():
    """No CPU fallback: without a device the context constructor raises."""
    from funs import _hip
    if _hip.device_count() > 0:
        pytest.skip('a GPU is visible')
    with pytest.raises(_hip.HipBackendError):
        _hip.Context(4, 2, 10, 2, 10.0)
    from funs import inference
    from conftest import Experiment
    exp = Experiment([np.zeros((4, 10)), np.zeros((4, 10))], 10.0)
    with pytest.raises(_hip.HipBackendError):
        inference.laplace(exp, {'C': np.zeros((4, 2)), 'd': np.zeros(4), 'tau': np.ones(2) * 0.1})


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'poisson-gpfa_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert 'pgpfa_oracle' not in src and 'import oracle' not in src and 'from oracle' not in src, f


def test_vec_layout_matches_reference(c1):
    from funs import util
    g = load_golden('c1_callbacks.npz')
    v = util.CdtoVecCd(c1['init_C'], c1['init_d'])
    assert np.array_equal(v, g['vecCd0'])
    C, d = util.vecCdtoCd(v, 3, 30)
    assert np.array_equal(C, c1['init_C']) and np.array_equal(d, c1['init_d'])


def test_dataset_generator_reproduces_reference_stream(c1):
    """util.dataset() defaults == config 1; the reference sampler must give the golden counts bit for bit."""
    from funs import util
    ds = util.dataset()
    Y = np.stack([tr['Y'] for tr in ds.data])
    assert np.array_equal(Y, c1['Y'])
    assert np.array_equal(ds.params['C'], c1['true_C']) and np.array_equal(ds.params['d'], c1['true_d'])
    big = util.dataset(trialDur=400, numTrials=3, xdim=2, ydim=6, seed=3, sampler='cholesky')
    assert big.data[0]['Y'].shape == (6, 40) and big.T == 40
    # the one-SVD sampler against numpy's own legacy multivariate_normal / poisson calls in the reference's order
    # (util.py:705-750) at a size and seed the golden file does not cover, latents included
    ds = util.dataset(trialDur=300, numTrials=4, xdim=3, ydim=7, seed=41)
    np.random.seed(41)
    C = np.random.rand(7, 3) - 0.5
    d = np.random.rand(7) * (-2) - 1
    tau = np.abs(np.random.rand(3)) + 0.01
    from oracle import pgpfa_oracle as orc
    K_big = orc.make_K_big(orc.make_K(tau, 30, 10.0))
    for tr in ds.data:
        X = np.reshape(np.random.multivariate_normal(np.zeros(90), K_big, 1), [3, 30])
        Y = np.random.poisson(lam=np.exp(C @ X + d[:, None]))
        assert np.allclose(tr['X'], X, rtol=0, atol=1e-12) and np.array_equal(tr['Y'], Y)


def test_subsample_stream_matches_reference(c1_experiment):
    from funs import util
    g = load_golden('c1_em_online.npz')
    np.random.seed(1)
    for it in range(4):
        sub = util.subsampleTrials(c1_experiment, 5)
        assert np.array_equal(sub.batchTrIdx, g['batchTrIdx'][it])
        assert sub._pgpfa_parent is c1_experiment and sub.numTrials == 5
        assert all(sub.data[i] is c1_experiment.data[j] for i, j in enumerate(sub.batchTrIdx))


def test_shard_slices_partition():
    from funs._session import shard_slice
    for n in (0, 1, 7, 20, 1024, 1025):
        for size in (1, 2, 3, 8):
            cuts = [shard_slice(n, r, size) for r in range(size)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(size - 1))
            lens = [b - a for a, b in cuts]
            assert max(lens) - min(lens) <= 1


_WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, 'poisson-gpfa_amd'))
from oracle import pgpfa_oracle as orc
from funs._session import shard_slice
rank, size = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=size)
g = np.load(os.path.join({root!r}, 'tests', 'golden', 'c1_dataset.npz'))
Ys = [g['Y'][r].astype(float) for r in range(6)]
params = {{'C': g['init_C'], 'd': g['init_d'], 'tau': g['init_tau']}}
lo, hi = shard_slice(len(Ys), rank, size)
mine = Ys[lo:hi]
res, nll, _ = orc.laplace(mine, params, 10.0, mode='exact', return_cov=False)
v = orc.cd_to_vec(params['C'], params['d']) * 1.01
f, dC, dd = orc.mstep_cd_terms(v, mine, res['post_mean'], res['post_vsm'], 3, 30)
P, n = orc.make_precomp(res)
buf = torch.from_numpy(np.concatenate([[-nll * len(mine), len(mine), f], dC.ravel(), dd, P.ravel()]))
dist.all_reduce(buf)
if rank == 0:
    np.save(os.environ['OUT'], buf.numpy())
dist.destroy_process_group()
'''


@pytest.mark.timeout(600)
def test_two_rank_gloo_sharded_statistics_equal_unsharded(tmp_path):
    """Trial-sharded sufficient statistics (objective sum, (C,d) cost/grad sums, PautoSum) summed over
    2 gloo ranks equal the unsharded ones up to summation order - the contract the RCCL path relies on."""
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER.format(root=ROOT))
    out = tmp_path / 'reduced.npy'
    env = dict(os.environ, OUT=str(out), OMP_NUM_THREADS='2')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', '29617', str(script)]
    subprocess.run(cmd, check=True, env=env, timeout=550)
    red = np.load(out)
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'c1_dataset.npz'))
    Ys = [g['Y'][r].astype(float) for r in range(6)]
    params = {'C': g['init_C'], 'd': g['init_d'], 'tau': g['init_tau']}
    res, nll, _ = orc.laplace(Ys, params, 10.0, mode='exact', return_cov=False)
    v = orc.cd_to_vec(params['C'], params['d']) * 1.01
    f, dC, dd = orc.mstep_cd_terms(v, Ys, res['post_mean'], res['post_vsm'], 3, 30)
    P, n = orc.make_precomp(res)
    ref = np.concatenate([[-nll * 6, 6, f], dC.ravel(), dd, P.ravel()])
    assert np.max(np.abs(red - ref) / (1.0 + np.abs(ref))) <= 1e-10


def test_timescale_root_finder_host_logic():
    """The host side of the lockstep timescale M-step (learning._lockstep_multi / _newton_poly_root) on analytic
    stand-ins for the device cost/gradient: roots of ten independent convex problems found together, with and without
    the displacement hint of the previous EM iteration, and the safeguarded Newton on the cubic interpolant against
    plain bisection."""
    from funs import learning
    rng = np.random.default_rng(7)
    k = 10
    roots = rng.normal(-6.0, 1.0, k)
    a = 1.0 + rng.random(k)

    def evaluate(Q):                                   # f_k = a/2 d^2 + d^4/10, g_k = f_k'
        d = Q - roots[None, :]
        return 0.5 * a * d * d + 0.1 * d ** 4, a * d + 0.4 * d ** 3
    for hint in (None, 0.3 * np.ones(k), -0.5 * np.ones(k)):
        p0 = roots + 0.3 * rng.standard_normal(k)
        pv, fv, gv, rounds, done = learning._lockstep_multi(evaluate, p0, d_hint=hint)
        assert np.all(done) and rounds <= 8
        assert np.max(np.abs(pv - roots)) <= 1e-9
    # a start far from the optimum (no sign change among the first candidates): geometric stepping out, then bracketing
    pv, _, _, rounds, done = learning._lockstep_multi(evaluate, roots + 3.0, d_hint=None)
    assert np.all(done) and np.max(np.abs(pv - roots)) <= 1e-9 and rounds <= 12

    def bisect(X, Y, lo, hi):
        n = X.shape[0]
        coef = Y.copy()
        for lvl in range(1, n):
            coef[lvl:] = (coef[lvl:] - coef[lvl - 1:-1]) / (X[lvl:] - X[:n - lvl])
        lo, hi = lo.copy(), hi.copy()
        for _ in range(80):
            mid = 0.5 * (lo + hi)
            v = coef[n - 1].copy()
            for i in range(n - 2, -1, -1):
                v = v * (mid - X[i]) + coef[i]
            hi = np.where(v > 0, mid, hi)
            lo = np.where(v > 0, lo, mid)
        return 0.5 * (lo + hi)
    for _ in range(50):
        c3 = rng.normal(0.0, 0.5, k)
        X = np.sort(roots[None, :] + rng.normal(0.0, 0.3, (4, k)), axis=0)
        Y = a * (X - roots) + c3 * (X - roots) ** 3 + 0.3 * (X - roots) ** 2
        lo = np.max(np.where(Y < 0, X, -np.inf), axis=0)
        hi = np.min(np.where(Y > 0, X, np.inf), axis=0)
        ok = np.isfinite(lo) & np.isfinite(hi) & (hi > lo)
        if not ok.any():
            continue
        lo, hi = np.where(ok, lo, 0.0), np.where(ok, hi, 1.0)
        r = learning._newton_poly_root(X, Y, lo, hi, ok)
        assert np.max(np.abs(r - bisect(X, Y, lo, hi))[ok]) <= 1e-13


def test_lockstep_timescale_update_with_prior_stops_where_the_reference_tnc_stops(c1, monkeypatch):
    """learning.learnGPparamsWithPrior(tauOptimMethod='lockstep') on a host stand-in for the device pass (the oracle's MStepGPtimescaleCost
    and its gradient on the PautoSum of config 1's posterior): the root of the reference's regularised gradient (learning.py:726-769: the prior
    term without the chain factor) found for all latents together lands where the reference's own TNC call on its cost / gradient pair stops
    (learning.py:819-825, restated by oracle.learn_tau_prior) - 1e-3 relative in tau, the tolerance of the GPU tests - for the step sizes of
    the first, fourth and eleventh stochastic-EM iteration; and the same point again from the displacement hint of the call before."""
    from funs import learning
    from oracle import pgpfa_oracle as orc
    import warnings
    par = {'C': c1['init_C'], 'd': c1['init_d'], 'tau': c1['init_tau']}
    Ys = c1['Ys'][:8]
    bs = c1['binSize']
    infRes = orc.laplace(Ys, par, bs, mode='exact')[0]
    P, R = orc.make_precomp(infRes)

    class Ctx:
        passes = 0

        def mstep_precomp(self):
            return R

        def mstep_tau_costgrad_multi(self, logp):
            Ctx.passes += 1
            logp = np.asarray(logp).reshape(-1, P.shape[0])
            cost = np.array([[orc.tau_cost(v, P[k], R) for k, v in enumerate(row)] for row in logp])
            grad = np.array([[orc.tau_grad(v, P[k], R)[0] for k, v in enumerate(row)] for row in logp])
            return cost, grad

    class Sess:
        ctx, p, T, post_stamp = Ctx(), P.shape[0], P.shape[1], 0
    sess = Sess()
    monkeypatch.setattr(learning, '_resident_session', lambda infRes, experiment, xdim: sess)

    class Exp:
        binSize = bs
    for n in (0, 3, 10):
        step = 1.0 / (n + 1) ** 0.75
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            tau_ref, _ = orc.learn_tau_prior(par, infRes, bs, 'TNC', step)
        for again in range(2):
            Ctx.passes = 0
            tau, details = learning.learnGPparamsWithPrior(par, infRes, Exp(), 'lockstep', step)
            assert np.max(np.abs(tau - tau_ref) / tau_ref) <= 1e-3
            assert all(d.success for d in details) and Ctx.passes <= 6
            # the point is the zero of the reference's gradient expression
            for k in range(P.shape[0]):
                pv = np.log(1.0 / (tau[k] * 1000.0 / bs) ** 2)
                g = orc.tau_grad_prior(pv, P[k], R, bs, par['tau'][k], step)[0]
                assert abs(g) <= 1e-6 * R


class _SeparableCost:
    """Stand-in for the device side of the Newton (C,d) driver: q independent convex costs f_n(th) = sum_i w_i exp(x_i . th) - y_n . th over
    th in R^D, with the methods and return conventions of funs._hip.Context that learning._newton_cd calls (vectors laid out [D][q])."""

    def __init__(self, X, w, Y):
        self.X, self.w, self.Y = X, w, Y                     # [I][D], [I], [q][D]
        self.H = None
        self.calls = {'full': 0, 'chord': 0, 'cost': 0}

    def _parts(self, vec):
        th = np.asarray(vec, dtype=np.float64).reshape(self.X.shape[1], -1).T            # [q][D]
        e = self.w[None, :] * np.exp(th @ self.X.T)                                         # [q][I]
        cost = e.sum(axis=1) - np.einsum('nd,nd->n', self.Y, th)
        grad = e @ self.X - self.Y
        return th, e, cost, grad

    def _step(self, grad):
        delta = -np.stack([np.linalg.solve(self.H[n], grad[n]) for n in range(grad.shape[0])])
        return delta.T.reshape(-1), -np.einsum('nd,nd->n', grad, delta)

    def mstep_cd_newton_pass(self, vec, prior_center=None, inv_s2=0.0):
        _, e, cost, grad = self._parts(vec)
        self.H = np.einsum('ni,id,ie->nde', e, self.X, self.X)
        self.calls['full'] += 1
        return (cost,) + self._step(grad)

    def mstep_cd_chord_pass(self, vec, prior_center=None, inv_s2=0.0):
        _, _, cost, grad = self._parts(vec)
        self.calls['chord'] += 1
        return (cost,) + self._step(grad)

    def mstep_cd_cost_per_neuron(self, vec, prior_center=None, inv_s2=0.0):
        self.calls['cost'] += 1
        return self._parts(vec)[2]


def test_newton_cd_driver_from_an_extrapolated_start(monkeypatch):
    """learning._newton_cd on a host stand-in for the device passes: a slowly drifting family of separable convex costs, minimised one after the
    other from the previous minimiser (what batch EM does to the (C,d) update).  Started one displacement ahead (CD_EXTRAPOLATE = 1) or along
    the trend of the last two (2) the driver lands on the same minimisers as from the parameters it is handed (0) - 1e-9, the gradient there
    below 1e-8 - and takes fewer device passes; handed anything else than its previous result it does not extrapolate."""
    import types
    from funs import learning
    rng = np.random.default_rng(3)
    q, D, I = 7, 4, 60
    X = rng.normal(size=(I, D)) * 0.7
    Y = np.abs(rng.normal(size=(q, I))) @ X / I * 8.0
    base = rng.random(I) + 0.5
    drift = rng.normal(size=I) * 0.6
    out = {}
    for mode in (0, 1, 2):
        monkeypatch.setattr(learning, 'CD_EXTRAPOLATE', mode)
        sess = types.SimpleNamespace(q=q, p=D - 1, ctx=None)
        x = np.zeros(D * q)
        mins, passes = [], 0
        for it in range(10):
            sess.ctx = _SeparableCost(X, base * np.exp(drift * (1.0 - 0.85 ** it)), Y)      # displacements that shrink by 15 % per iteration
            if it:
                sess.ctx.H = H_prev                                    # the Hessians of the previous M-step stay resident
            x, fun, n = learning._newton_cd(sess, x, hess_key=b'all', extrapolate=True)
            H_prev = sess.ctx.H
            passes += n if it >= 2 else 0
            grad = sess.ctx._parts(x)[3]
            assert np.max(np.abs(grad)) <= 1e-8
            mins.append(x.copy())
        out[mode] = (mins, passes)
        # a caller that hands in something else than the previous result gets no extrapolation: the track does not match
        sess.ctx.calls = {'full': 0, 'chord': 0, 'cost': 0}
        x2, _, _ = learning._newton_cd(sess, mins[-1] + 1e-3, hess_key=b'all', extrapolate=True)
        assert np.max(np.abs(x2 - mins[-1])) <= 1e-9
    for mode in (1, 2):
        for a, b in zip(out[mode][0], out[0][0]):
            assert np.max(np.abs(a - b)) <= 1e-9
    assert out[2][1] <= out[1][1] <= out[0][1] and out[2][1] < out[0][1], [out[m][1] for m in (0, 1, 2)]


def test_scalar_form_of_the_timescale_root_finder_is_the_array_form_bit_for_bit(monkeypatch):
    """learning._lockstep_multi (plain floats, latent by latent: what the M-step runs) against learning._lockstep_multi_np (the array
    statement of the same algorithm) on 600 random families of convex problems - quartic, cosh and softplus costs; 1 to 11 latents; starts
    from 0.01 to 5 away; no hint, a good hint, a random one, a hint with gaps, a zero hint: same roots, costs, gradients, rounds and verdicts,
    bit for bit.  (With the early acceptance of round 6 switched off: the array form does not have it; it is tested below.)"""
    from funs import learning
    monkeypatch.setattr(learning, 'TAU_EARLY_ACCEPT', False)
    rng = np.random.default_rng(11)
    for trial in range(600):
        k = int(rng.integers(1, 12))
        roots = 2.0 * rng.normal(size=k)
        a = 0.5 + 3.0 * rng.random(k)
        c = 0.5 * rng.random(k)
        kind = trial % 3

        def evaluate(Q):
            x = np.asarray(Q) - roots[None, :]
            if kind == 0:
                return a * (x ** 2 / 2 + c * x ** 4 / 4), a * (x + c * x ** 3)
            if kind == 1:
                return a * (np.cosh(x) - 1.0), a * np.sinh(x)
            return 2.0 * a * (np.log1p(np.exp(x)) + np.log1p(np.exp(-x))), 2.0 * a * np.tanh(x / 2)
        p0 = roots + (0.01, 0.1, 1.0, 5.0)[trial % 4] * rng.normal(size=k)
        hint = (None, (roots - p0) * (1.0 + 0.1 * rng.normal(size=k)), rng.normal(size=k), np.where(rng.random(k) < 0.3, np.nan, roots - p0),
                np.zeros(k))[trial % 5]
        A = learning._lockstep_multi_np(evaluate, p0, d_hint=hint)
        B = learning._lockstep_multi(evaluate, p0, d_hint=hint)
        assert A[3] == B[3] and np.array_equal(A[4], B[4])
        for x, y in zip(A[:3], B[:3]):
            assert np.array_equal(x, y)


def test_timescale_root_finder_accepts_a_root_on_its_interpolation_error(monkeypatch):
    """Round 6: a root of the cubic through four samples is accepted when the interpolation-error term - the next divided difference with the
    nearest fifth sample, times prod (r - x_i), over the cubic's slope - is below xtol, instead of waiting for a second prediction to agree with
    it.  On 300 random families (quartic, cosh, softplus; the hint of a previous M-step off by 10 %, as in a running fit): the roots of either rule are within 2e-8 of the true ones (both may stop on |gradient| <= 1e-8), never more rounds, and at least a third of the families
    one round fewer."""
    from funs import learning
    rng = np.random.default_rng(23)
    fewer = 0
    for trial in range(300):
        k = int(rng.integers(1, 12))
        roots = 2.0 * rng.normal(size=k)
        a = 0.5 + 3.0 * rng.random(k)
        c = 0.5 * rng.random(k)
        kind = trial % 3

        def evaluate(Q):
            x = np.asarray(Q) - roots[None, :]
            if kind == 0:
                return a * (x ** 2 / 2 + c * x ** 4 / 4), a * (x + c * x ** 3)
            if kind == 1:
                return a * (np.cosh(x) - 1.0), a * np.sinh(x)
            return 2.0 * a * (np.log1p(np.exp(x)) + np.log1p(np.exp(-x))), 2.0 * a * np.tanh(x / 2)
        p0 = roots + 0.05 * rng.normal(size=k)
        hint = (roots - p0) * (1.0 + 0.1 * rng.normal(size=k))
        monkeypatch.setattr(learning, 'TAU_EARLY_ACCEPT', False)
        base = learning._lockstep_multi(evaluate, p0, d_hint=hint)
        monkeypatch.setattr(learning, 'TAU_EARLY_ACCEPT', True)
        fast = learning._lockstep_multi(evaluate, p0, d_hint=hint)
        assert np.all(fast[4]) and np.all(base[4])
        # (either rule may also stop on |gradient| <= 1e-8 at a sample: up to 1e-8 / curvature, curvature >= 0.5 here)
        assert np.max(np.abs(fast[0] - roots)) <= 2e-8 and np.max(np.abs(base[0] - roots)) <= 2e-8
        assert fast[3] <= base[3]
        fewer += int(fast[3] < base[3])
    assert fewer >= 100, fewer


# ---- multi-rank start-up: rendezvous handshake and launcher supervision (no GPU: the unique id is a stub) -----------------
_RDZV = r"""
import os, sys, time
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, 'poisson-gpfa_amd'))
from funs import _hip, _session
_hip.comm_unique_id = lambda: bytes(range(128))
w = _session.WORLD
assert w.enabled and w.size == int(os.environ['WORLD_SIZE'])
try:
    uid, path = w.exchange_unique_id()
except _hip.HipBackendError as exc:
    sys.stderr.write('RDZV-FAILED: %s\n' % exc)
    sys.exit(7)
assert uid == bytes(range(128))
print('ok', w.rank, os.path.basename(path))
"""


def _rdzv_rank(tmp_path, rank, size, port, timeout_s, ppid_file=None):
    script = tmp_path / 'rdzv.py'
    script.write_text(_RDZV.format(root=ROOT))
    env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(size), LOCAL_RANK=str(rank), MASTER_PORT=str(port),
               PGPFA_RDZV_DIR=str(tmp_path), PGPFA_RDZV_TIMEOUT=str(timeout_s))
    return subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def test_rendezvous_handshake_and_timeouts(tmp_path):
    """All ranks present: same id everywhere.  A rank that never shows up (died at start-up): every other rank leaves with
    a non-zero status after the timeout instead of entering ncclCommInitRank and hanging there.  A stale id file of an
    earlier job with the same key is not taken for rank 0's."""
    procs = [_rdzv_rank(tmp_path, r, 3, 41001, 60) for r in range(3)]
    outs = [pr.communicate(timeout=120) for pr in procs]
    assert [pr.returncode for pr in procs] == [0, 0, 0], outs
    assert all(o[0].startswith('ok %d ' % r) for r, o in enumerate(outs))
    # rank 2 never started: ranks 0 and 1 time out (rank 1 has read the id and acknowledged; the go-ahead never comes)
    procs = [_rdzv_rank(tmp_path, r, 3, 41002, 3) for r in range(2)]
    outs = [pr.communicate(timeout=120) for pr in procs]
    assert [pr.returncode for pr in procs] == [7, 7], outs
    assert 'acknowledgement' in outs[0][1] and 'go-ahead' in outs[1][1]
    # rank 0 never publishes; a leftover id file with the same key but an old timestamp lies around
    key = 'pgpfa_uid_41003_none_0_%d_1' % os.getpid()
    stale = tmp_path / key
    stale.write_bytes(bytes(128))
    old = os.path.getmtime(str(stale)) - 3600
    os.utime(str(stale), (old, old))
    pr = _rdzv_rank(tmp_path, 1, 2, 41003, 3)
    out = pr.communicate(timeout=120)
    assert pr.returncode == 7 and "rank 0's unique id" in out[1], out


def test_rendezvous_survives_leftovers_of_a_restarted_attempt(tmp_path):
    """a restarted worker group under the same key finds the previous attempt's id file, acknowledgements and
    go-ahead, all recent.  Rank 1 starts FIRST and reads the leftovers (old id, matching old go-ahead is removed by rank 0 only when
    it starts); it must end up with rank 0's NEW id, never with the old one, and rank 0 must not take the leftover acknowledgement
    of the old id for rank 1's.  Also: a rank that starts long after rank 0 published (more than the old fixed 120-s window would
    allow relative to the file's age) still joins while rank 0 waits."""
    import time
    key = 'pgpfa_uid_41004_none_0_%d_1' % os.getpid()
    old_uid = bytes(128)
    (tmp_path / key).write_bytes(old_uid)
    (tmp_path / (key + '.ack.1')).write_bytes(old_uid)
    pr1 = _rdzv_rank(tmp_path, 1, 2, 41004, 60)
    time.sleep(1.5)                                    # rank 1 has read and acknowledged the leftover id by now; no go-ahead yet
    assert pr1.poll() is None
    pr0 = _rdzv_rank(tmp_path, 0, 2, 41004, 60)
    outs = [pr.communicate(timeout=120) for pr in (pr0, pr1)]
    assert [pr0.returncode, pr1.returncode] == [0, 0], outs      # the script asserts uid == bytes(range(128)), the NEW id
    # leftover go-ahead with the OLD id next to a leftover id file: rank 1 alone must time out waiting, not leave with the old id
    key = 'pgpfa_uid_41005_none_0_%d_1' % os.getpid()
    (tmp_path / key).write_bytes(old_uid)
    (tmp_path / (key + '.go')).write_bytes(bytes([1]) * 128 + bytes(16))
    pr = _rdzv_rank(tmp_path, 1, 2, 41005, 3)
    out = pr.communicate(timeout=120)
    assert pr.returncode == 7 and 'go-ahead' in out[1], out
    # the COMPLETE leftovers of a crashed run under the same key - id file and a go-ahead that matches it, both recent.  Rank 1
    # starts first: it must not leave with the dead id (the go-ahead does not carry its nonce) but join rank 0's new attempt
    key = 'pgpfa_uid_41007_none_0_%d_1' % os.getpid()
    dead = bytes([7]) * 128
    (tmp_path / key).write_bytes(dead)
    (tmp_path / (key + '.go')).write_bytes(dead + bytes(16))
    pr1 = _rdzv_rank(tmp_path, 1, 2, 41007, 60)
    time.sleep(1.5)
    assert pr1.poll() is None                          # still waiting: the leftover go-ahead was not taken
    pr0 = _rdzv_rank(tmp_path, 0, 2, 41007, 60)
    outs = [pr.communicate(timeout=120) for pr in (pr0, pr1)]
    assert [pr0.returncode, pr1.returncode] == [0, 0], outs
    # a restart count in the environment is part of the key: the files of attempt 0 are not even looked at by attempt 1
    env_key = 'pgpfa_uid_41006_none_1_%d_1' % os.getpid()
    script = tmp_path / 'rdzv.py'
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', LOCAL_RANK=str(r), MASTER_PORT='41006', PGPFA_RDZV_DIR=str(tmp_path),
                   PGPFA_RDZV_TIMEOUT='60', TORCHELASTIC_RESTART_COUNT='1')
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [pr.communicate(timeout=120) for pr in procs]
    assert [pr.returncode for pr in procs] == [0, 0], outs
    assert all(env_key in o[0] for o in outs), outs


def test_bench_launcher_stops_all_ranks_when_one_dies():
    """bench.py --gpus N as its own launcher: rank 1 exits non-zero at start-up while rank 0 would wait for ever; the parent
    must end rank 0 and return non-zero promptly."""
    import time
    env = dict(os.environ, PGPFA_DRYRUN_DIE='1', PGPFA_DRYRUN_HANG='0')
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run'], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and 'rank 1 exited with status 3' in out.stderr, out
    assert time.time() - t0 < 60
    # and the overall timeout
    env = dict(os.environ, PGPFA_DRYRUN_HANG='0', PGPFA_BENCH_TIMEOUT='3')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run'], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and 'timeout after 3 s' in out.stderr, out


def test_bench_launcher_reports_a_rank_stuck_in_comm_init():
    """ncclCommInitRank has no timeout - a rank that hangs in it must end as a non-zero exit WITH a message inside the
    comm-phase deadline (a helper thread ends the process; nothing is re-executed), and the launcher then stops the other ranks."""
    import time
    env = dict(os.environ, PGPFA_DRYRUN_COMM_HANG='1', PGPFA_DRYRUN_HANG='0', PGPFA_COMM_TIMEOUT='2', PGPFA_BENCH_TIMEOUT='100')
    t0 = time.time()
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run'], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode != 0, out
    assert 'rank 1: pgpfa_comm_init (ncclCommInitRank) did not finish within 2 s' in out.stderr, out
    assert 'rank 1 exited with status 13' in out.stderr, out
    assert time.time() - t0 < 60
    # the launcher's own default limit stays below the 1800 s of whoever runs it
    src = open(os.path.join(ROOT, 'bench.py')).read()
    import re
    assert float(re.search(r"PGPFA_BENCH_TIMEOUT', '(\d+)'", src).group(1)) <= 900


def test_header_documents_every_option_and_info_key():
    """include/pgpfa.h is the C-ABI's documentation: every key pgpfa_set_option accepts and every info key the library sets is named there."""
    import re
    csrc = os.path.join(ROOT, 'poisson-gpfa_amd', 'csrc')
    src = ''.join(open(os.path.join(csrc, f)).read() for f in sorted(os.listdir(csrc)) if f.endswith('.hip'))
    hdr = open(os.path.join(ROOT, 'include', 'pgpfa.h')).read()
    options = sorted(set(re.findall(r'k == "([a-z0-9_]+)"', src)))
    infos = sorted(set(re.findall(r'c->info\["([a-z0-9_]+)"\]', src)))
    assert len(options) > 30 and len(infos) > 15
    missing = [k for k in options + infos if '"%s"' % k not in hdr]
    assert not missing, missing


def test_dual_variational_hands_unsettled_trials_to_lbfgs(monkeypatch):
    """Host logic of inference.dualVariational under DUAL_SOLVER = 'fixedpoint' (the default), against a test double of the context: the cold start
    is set on the device (no rho goes up), nothing comes down (want_rho / want_lam off), the finalize call takes the resident optimum (lam = None)
    and the returned varOptimRes is a lazy list that reads a trial's lambda (or its log) from the device when asked; handed back as prevOptimRes
    of the next call on the same trials it is a resident warm start.  Trials the fixed point reports as not settled (status 1: pass cap, 2: not
    contracting) go to the device L-BFGS from the lambda the fixed point left, exactly those and no others; their optimum and bound replace
    the fixed point's, iteration counts add, and the finalize call then gets the merged lambda from the host."""
    from funs import _hip, _session, inference
    from conftest import Experiment
    calls = {}
    q, p, T, R = 4, 2, 6, 5
    state = {'status': np.zeros(R, np.int32), 'lam': 1.0 + np.arange(R)[:, None] + np.zeros((R, q * T))}

    class Ctx:
        def __init__(self, q_, p_, T_, R_, bin_ms, device=0):
            self.q, self.p, self.T, self.R = q_, p_, T_, R_
        def close(self): pass
        def set_option(self, k, v): calls.setdefault('options', {})[k] = v
        def info(self, k): return 0.0
        def upload_counts(self, Y): pass
        def set_params(self, C, d, tau): pass
        def dual_fixed_point(self, idx, rho0=None, max_outer=40, tol=1e-8, warm=False, want_lam=False, want_rho=True, resident=False):
            calls['fp'] = dict(idx=np.array(idx), rho0=None if rho0 is None else np.array(rho0), max_outer=max_outer, tol=tol, warm=warm,
                               want_lam=want_lam, want_rho=want_rho, resident=resident)
            return (None, -np.arange(len(idx), dtype=float), np.full(len(idx), 5, np.int32), state['status'][:len(idx)].copy())
        def dual_lambda(self, idx=None):
            calls.setdefault('lambda_reads', []).append(np.array(idx))
            return state['lam'][np.asarray(idx)].copy()
        def dual_lbfgs(self, idx, rho0, max_iter=15000, factr=1e7, pgtol=1e-5):
            calls['lbfgs'] = (np.array(idx), np.array(rho0))
            return np.array(rho0) + np.log(2.0), np.full(len(idx), -100.0), np.full(len(idx), 70, np.int32)
        def dual_finalize(self, idx, lam):
            calls['finalize'] = (np.array(idx), None if lam is None else np.array(lam))
            if lam is not None:
                state['lam'][np.asarray(idx)] = lam
            return 3.0 * len(idx)

    monkeypatch.setattr(_hip, 'Context', Ctx)
    monkeypatch.setattr(_session.WORLD, 'enabled', False)
    _session.drop_sessions()
    exp = Experiment([np.zeros((q, T)) for _ in range(R)], 10.0)
    params = {'C': np.zeros((q, p)), 'd': np.zeros(q), 'tau': np.ones(p) * 0.1}
    assert inference.DUAL_SOLVER == 'fixedpoint'
    try:
        # everybody settles: nothing crosses the bus
        infRes, nll, vlb, opt = inference.dualVariational(exp, params)
        fp = calls['fp']
        assert np.array_equal(fp['idx'], np.arange(R)) and fp['rho0'] is None and not fp['warm'] and not fp['resident']
        assert not fp['want_rho'] and not fp['want_lam'] and fp['max_outer'] == inference.DUAL_FP_MAX_PASSES and fp['tol'] == inference.DUAL_FP_TOL
        assert calls['finalize'][1] is None and 'lbfgs' not in calls and 'lambda_reads' not in calls
        assert isinstance(opt, _session.DeviceDualOptimRes) and len(opt) == R
        assert np.array_equal(opt[3], state['lam'][3]) and [list(a) for a in calls['lambda_reads']] == [[3]]       # one trial read, one download
        assert np.array_equal(infRes.dual_iterations, [5] * R) and abs(vlb + 2.0) < 1e-12 and abs(nll + 3.0) < 1e-12
        # handed back on the same trials: a resident warm start
        infRes, nll, vlb, opt2 = inference.dualVariational(exp, params, prevOptimRes=opt)
        assert calls['fp']['resident'] and calls['fp']['warm'] and calls['fp']['rho0'] is None
        # a plain list of arrays (what the reference passes around): uploaded as log lambda, flagged warm
        infRes, nll, vlb, opt3 = inference.dualVariational(exp, params, prevOptimRes=[state['lam'][r] for r in range(R)])
        assert not calls['fp']['resident'] and calls['fp']['warm'] and np.allclose(calls['fp']['rho0'], np.log(state['lam']))
        # the log-lambda variant returns rho
        infRes, nll, vlb, opt4 = inference.dualVariational(exp, params, optimizeLogLambda=True)
        assert np.allclose(opt4[1], np.log(state['lam'][1]))
        # two trials handed back
        state['status'] = np.array([0, 2, 0, 1, 0], np.int32)
        before = state['lam'].copy()
        infRes, nll, vlb, opt5 = inference.dualVariational(exp, params)
        bad_idx, bad_rho = calls['lbfgs']
        assert np.array_equal(bad_idx, [1, 3]) and np.allclose(bad_rho, np.log(before[[1, 3]]))
        merged = before.copy()
        merged[[1, 3]] *= 2.0
        assert np.allclose(calls['finalize'][1], merged) and np.allclose(opt5[1], merged[1]) and np.allclose(opt5[0], merged[0])
        assert np.array_equal(infRes.dual_iterations, [5, 75, 5, 75, 5])
        assert abs(vlb - np.mean([0.0, -100.0, -2.0, -100.0, -4.0])) < 1e-12
    finally:
        _session.drop_sessions()


def test_committed_bench_lines_carry_the_contract_keys():
    """The driver reads ONE JSON line from bench.py; the newest committed lines of the three BASELINE workloads (profiles/rNN_bench_*.json, produced
    by tools/measure_round.sh on the GPU box) must carry every key of that contract - the throughput block, `roofline` of the dominant kernel
    (bound, achieved, peak, unit, frac, traffic) and `cpu_baseline` (value, unit, cores, kind, sample) - and consistent numbers."""
    import glob
    import json

    def newest(pattern):
        files = sorted(glob.glob(os.path.join(ROOT, 'profiles', pattern)))
        assert files, pattern
        return json.loads(open(files[-1]).read().strip().splitlines()[-1]), os.path.basename(files[-1])
    for pattern in ('r*_bench_c3_driver_protocol.json', 'r*_bench_c4_online.json', 'r*_bench_c5_dual_estep_mixed.json'):
        line, name = newest(pattern)
        for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config',
                  'roofline', 'cpu_baseline'):
            assert k in line, (name, k)
        assert line['n_gpus'] == 1 and line['higher_is_better'] is True and line['data'] == 'synthetic' and line['vs_baseline'] is None
        assert 'workload' in line['config'] and 'model' not in line['config']
        roof = line['roofline']
        for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
            assert k in roof, (name, k)
        assert roof['bound'] in ('hbm', 'mfma') and roof['unit'] in ('GB/s', 'TFLOP/s')
        assert abs(roof['frac'] - roof['achieved'] / roof['peak']) <= 1e-9 and 0.0 < roof['frac'] < 1.0
        cpu = line['cpu_baseline']
        for k in ('value', 'unit', 'cores', 'kind', 'sample'):
            assert k in cpu, (name, k)
        assert cpu['kind'] in ('port', 'reference') and cpu['value'] > 0 and cpu['cores'] >= 1
        if name.startswith('r06_bench_c3'):
            # value = 1024-trial EM iterations per second over the timed steps
            assert abs(line['value'] - 1e3 / line['ms_per_step']) <= 1e-6 * line['value'] and line['dtype'] == 'f64'
