"""The PRODUCT's multi-rank bookkeeping on CPU: funs.inference / funs.learning / funs.engine (trial slicing per rank,
minibatch draws from the shared RNG stream, the all-reduced E-step objective, per-trial stamps, the replicated M-step
drivers) run under 2 gloo ranks against a test double of the C-ABI context - the contract of include/pgpfa.h restated
with the oracle, its in-library RCCL all-reduce replaced by torch.distributed(gloo) - and must reproduce the 1-rank run.
What cannot run here (the HIP kernels and RCCL themselves) is covered by the -m gpu tests."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

_WORKER = r'''
import os, sys, json
import numpy as np
ROOT = {root!r}
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'poisson-gpfa_amd'))
import torch, torch.distributed as dist
from oracle import pgpfa_oracle as orc
import funs
from funs import _hip, _session, inference, learning, util, engine

rank, size = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
if size > 1:
    dist.init_process_group('gloo', rank=rank, world_size=size)


def allreduce(a):
    a = np.array(a, dtype=np.float64)
    if size > 1:
        t = torch.from_numpy(a.reshape(-1).copy())
        dist.all_reduce(t)
        return t.numpy().reshape(a.shape)
    return a


class FakeContext:
    """Test double of funs._hip.Context: same methods and return conventions, arithmetic by the oracle, the library's
    internal all-reduce of the M-step sums (include/pgpfa.h: pgpfa_mstep_cd_costgrad, pgpfa_mstep_precomp) by gloo."""
    def __init__(self, q, p, T, R, bin_ms, device=0):
        self.q, self.p, self.T, self.R, self.bin = q, p, T, R, bin_ms
        self.n = p * T
        self.pm, self.pv, self.pg = {{}}, {{}}, {{}}
        self._info = {{'last_pcg_iterations': 0.0, 'last_loo_unconverged': 0.0}}
        self.calls = []
        self.fail_trial = None                     # a trial whose mode search reports status 3 (test of the collective verdict)
    def close(self): pass
    def set_option(self, k, v): pass
    def info(self, k): return self._info.get(k, 0.0)
    def upload_counts(self, Y): self.Y = np.asarray(Y, dtype=np.float64)
    def set_params(self, C, d, tau): self.params = {{'C': np.array(C), 'd': np.array(d).reshape(-1), 'tau': np.array(tau).reshape(-1)}}
    def comm_init(self, uid, rank, nranks): pass
    def allreduce_host(self, arr): return allreduce(arr)
    def comm_describe(self): return "rank %d/%d device cpu pci none comm gloo" % (rank, size)
    def estep_laplace(self, idx=None, warm_start=False):
        idx = np.arange(self.R) if idx is None else np.asarray(idx)
        assert len(set(idx.tolist())) == len(idx)
        res, nll, _ = orc.laplace([self.Y[i] for i in idx], self.params, self.bin, mode='exact', return_cov=False)
        for j, i in enumerate(idx):
            self.pm[int(i)], self.pv[int(i)], self.pg[int(i)] = res['post_mean'][j], res['post_vsm'][j], res['post_vsmGP'][j]
        self.last = [int(i) for i in idx]
        self.calls.append(('estep', self.last))
        status = np.zeros(len(idx), np.int32)
        if self.fail_trial is not None:
            status[idx == self.fail_trial] = 3
        return -nll * len(idx), np.ones(len(idx), np.int32), status
    def count_moments(self, idx=None):
        idx = np.arange(self.R) if idx is None else np.asarray(idx)
        ras = np.concatenate([self.Y[i] for i in idx], axis=1).astype(np.int64) if len(idx) else np.zeros((self.q, 0), np.int64)
        return ras.sum(axis=1), ras @ ras.T, ras.shape[1]
    def post_mean(self, idx=None): return np.stack([self.pm[int(i)] for i in idx])
    def post_vsm(self, idx=None): return np.stack([self.pv[int(i)] for i in idx])
    def post_vsmgp(self, idx=None): return np.stack([self.pg[int(i)] for i in idx])
    def _last(self): return [self.Y[i] for i in self.last], [self.pm[i] for i in self.last], [self.pv[i] for i in self.last]
    def mstep_cd_costgrad(self, vec, prior_center=None, inv_s2=0.0):
        Ys, pm, pv = self._last()
        f, dC, dd = orc.mstep_cd_terms(np.asarray(vec), Ys, pm, pv, self.p, self.q) if Ys else (0.0, np.zeros((self.q, self.p)), np.zeros(self.q))
        buf = allreduce(np.concatenate([[f, len(Ys)], orc.cd_to_vec(dC, dd)]))
        Rtot = buf[1]
        cost, grad = -buf[0] / Rtot, -buf[2:] / Rtot
        if prior_center is not None:
            dv = np.asarray(vec) - np.asarray(prior_center)
            cost += 0.5 * inv_s2 * dv @ dv
            grad = grad + inv_s2 * dv
        return cost, grad
    def mstep_precomp(self):
        P = np.zeros((self.p, self.T, self.T))
        for i in self.last:
            for k in range(self.p):
                P[k] += self.pg[i][:, :, k] + np.outer(self.pm[i][k], self.pm[i][k])
        buf = allreduce(np.concatenate([[len(self.last)], P.ravel()]))
        self.Ntot, self.P = buf[0], buf[1:].reshape(P.shape)
        return self.Ntot
    def pautosum(self): return self.P
    def mstep_tau_costgrad_multi(self, logp):
        logp = np.asarray(logp).reshape(-1, self.p)
        cost = np.array([[orc.tau_cost(np.array([pv]), self.P[k], self.Ntot) for k, pv in enumerate(row)] for row in logp])
        grad = np.array([[float(np.asarray(orc.tau_grad(np.array([pv]), self.P[k], self.Ntot)).reshape(-1)[0]) for k, pv in enumerate(row)] for row in logp])
        return cost, grad
    def mstep_tau_costgrad_batch(self, logp):
        c, g = self.mstep_tau_costgrad_multi(np.asarray(logp).reshape(1, -1))
        return c[0], g[0]


_hip.Context = FakeContext
_session.WORLD.rank, _session.WORLD.size, _session.WORLD.local_rank, _session.WORLD.enabled = rank, size, rank, size > 1
_session.WORLD.exchange_unique_id = lambda: (b'0' * 128, '/nonexistent')

g = np.load(os.path.join(ROOT, 'tests', 'golden', 'c1_dataset.npz'))
R = int(os.environ.get('NTRIALS', '7'))                # odd on purpose: the ranks' slices differ in length
LIGHT = os.environ.get('LIGHT', '0') == '1'            # many ranks: batch EM and the failure verdict only
class Exp:
    pass
exp = Exp()
exp.data = [{{'Y': g['Y'][r].astype(float)}} for r in range(R)]
exp.ydim, exp.T = exp.data[0]['Y'].shape
exp.binSize, exp.trialDur, exp.numTrials = 10.0, 1000.0, R
init = {{'C': g['init_C'].copy(), 'd': g['init_d'].copy(), 'tau': g['init_tau'].copy()}}
out = {{}}

# batch EM, 2 iterations: every rank must end with the same parameters, equal to the 1-rank run
fit = engine.PPGPFAfit(exp, initParams={{k: v.copy() for k, v in init.items()}}, inferenceMethod='laplace', EMmode='Batch', maxEMiter=2,
                       CdOptimMethod='TNC', quiet=True)
sess, _ = _session.session_for(exp, 3)
lo, hi = sess.local_slice(R)
out['batch_slice'] = [lo, hi]
out['batch_estep_trials'] = [c[1] for c in sess.ctx.calls if c[0] == 'estep'][:2]
out['batch_nll'] = [float(v) for v in fit.posteriorLikelihood]
out['batch_C'] = np.asarray(fit.optimParams['C']).tolist()
out['batch_tau'] = np.asarray(fit.optimParams['tau']).tolist()
out['sample_mean_counts'] = np.asarray(fit.sampleMeanSpikeCounts).tolist()
out['infres_trials'] = fit.infRes.trial_idx.tolist()
out['infres_len'] = len(fit.infRes['post_mean'])

# a trial that fails on ONE rank (status 3) must fail the E-step on EVERY rank, with the same message, and leave the
# ranks able to go on with collectives (nobody is left behind in the all-reduce of the objective)
sess.ctx.fail_trial = R - 1                            # owned by the last rank
try:
    inference.laplace(exp, dict(fit.optimParams))
    out['fail_message'] = None
except _hip.HipBackendError as exc:
    out['fail_message'] = str(exc)
sess.ctx.fail_trial = None
out['after_fail_sum'] = float(sess.allreduce(np.array([1.0]))[0])
_, nll_again = inference.laplace(exp, dict(fit.optimParams), returnOptimRes=False)
out['after_fail_nll'] = float(nll_again)
if LIGHT:
    # config 4's form in small: stochastic EM with the minibatch split over ALL ranks (8 ranks x 1 trial of a minibatch of 8)
    np.random.seed(1)
    n0 = len(sess.ctx.calls)
    fit2 = engine.PPGPFAfit(exp, initParams={{k: v.copy() for k, v in init.items()}}, inferenceMethod='laplace', EMmode='Online', maxEMiter=2,
                            batchSize=8, onlineParamUpdateMethod='diag', quiet=True)
    out['online_estep_trials'] = [c[1] for c in sess.ctx.calls[n0:] if c[0] == 'estep'][:2]
    out['online_nll'] = [float(v) for v in fit2.posteriorLikelihood]
    out['online_C'] = np.asarray(fit2.optimParams['C']).tolist()
    out['online_tau'] = np.asarray(fit2.optimParams['tau']).tolist()
    with open(os.environ['OUT'] + '.%d' % rank, 'w') as fh:
        json.dump(out, fh)
    if size > 1:
        dist.destroy_process_group()
    sys.exit(0)

# online 'diag' EM, 3 minibatches of 4: the draws come from the global RNG (same on every rank), each rank takes its slice
np.random.seed(1)
n0 = len(sess.ctx.calls)
fit2 = engine.PPGPFAfit(exp, initParams={{k: v.copy() for k, v in init.items()}}, inferenceMethod='laplace', EMmode='Online', maxEMiter=3,
                        batchSize=4, onlineParamUpdateMethod='diag', quiet=True)
out['online_estep_trials'] = [c[1] for c in sess.ctx.calls[n0:] if c[0] == 'estep'][:3]
out['online_nll'] = [float(v) for v in fit2.posteriorLikelihood]
out['online_C'] = np.asarray(fit2.optimParams['C']).tolist()
out['online_tau'] = np.asarray(fit2.optimParams['tau']).tolist()

# stamps: the last minibatch's infRes is readable; after one more E-step over the same trials its unread entries are not
infRes = fit2.infRes
first = infRes['post_mean'][0]
res_again, _, _ = inference.laplace(exp, dict(fit2.optimParams))
# (a rank overwrites only the trials of ITS slice of the full list: entries of other trials are still the minibatch's)
out['stale_expected'] = [bool(lo <= int(t) < hi) for t in infRes.trial_idx]
out['stale_raises'] = []
for j in range(len(infRes.trial_idx)):
    try:
        infRes['post_vsm'][j]
        out['stale_raises'].append(False)
    except _hip.HipBackendError:
        out['stale_raises'].append(True)
out['cached_read_survives'] = bool(np.array_equal(infRes['post_mean'][0], first))
# the same minibatches with the timescale update as a lockstep root of the reference's gradient (round 6: tauOptimMethod='lockstep'): replicated
# on identical all-reduced sums, it must leave every rank with the same bits - and land where the TNC runs above stop (1e-3)
np.random.seed(1)
fit3 = engine.PPGPFAfit(exp, initParams={{k: v.copy() for k, v in init.items()}}, inferenceMethod='laplace', EMmode='Online', maxEMiter=3,
                        batchSize=4, onlineParamUpdateMethod='diag', tauOptimMethod='lockstep', quiet=True)
out['online_lockstep_tau'] = np.asarray(fit3.optimParams['tau']).tolist()
out['online_lockstep_C'] = np.asarray(fit3.optimParams['C']).tolist()
out['online_lockstep_nll'] = [float(v) for v in fit3.posteriorLikelihood]

with open(os.environ['OUT'] + '.%d' % rank, 'w') as fh:
    json.dump(out, fh)
if size > 1:
    dist.destroy_process_group()
'''


def _run(tmp_path, nranks, port, **extra):
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER.format(root=ROOT))
    out = tmp_path / ('out%d_%s' % (nranks, extra.get('NTRIALS', '7')))
    env = dict(os.environ, OUT=str(out), OMP_NUM_THREADS='1' if nranks > 2 else '2', **extra)
    if nranks == 1:
        env.update(RANK='0', WORLD_SIZE='1')
        subprocess.run([sys.executable, str(script)], check=True, env=env, timeout=900)
    else:
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=%d' % nranks, '--master-addr', '127.0.0.1',
               '--master-port', str(port), str(script)]
        subprocess.run(cmd, check=True, env=env, timeout=900)
    import json
    return [json.load(open(str(out) + '.%d' % r)) for r in range(nranks)]


@pytest.mark.timeout(1500)
def test_two_rank_product_host_layer_matches_single_rank(tmp_path):
    one = _run(tmp_path, 1, 0)[0]
    two = _run(tmp_path, 2, 29641)
    # slices partition the trial list; every rank ran its E-step on exactly its slice
    assert two[0]['batch_slice'] == [0, 4] and two[1]['batch_slice'] == [4, 7]
    assert two[0]['batch_estep_trials'] == [[0, 1, 2, 3]] * 2 and two[1]['batch_estep_trials'] == [[4, 5, 6]] * 2
    assert two[0]['infres_trials'] == [0, 1, 2, 3] and two[0]['infres_len'] == 4 and two[1]['infres_len'] == 3
    for r in range(2):
        # replicated state: both ranks hold the all-reduced objective and the same parameters, equal to the 1-rank run
        # (before the first M-step the sums differ by summation order only; after it scipy's TNC, which stops on an f-tolerance,
        # amplifies that rounding-level difference to ~1e-5 in the parameters - a dropped or doubled trial would show as O(0.1))
        assert abs(two[r]['batch_nll'][0] - one['batch_nll'][0]) <= 1e-11 * abs(one['batch_nll'][0])
        assert np.allclose(two[r]['batch_nll'], one['batch_nll'], rtol=1e-6, atol=0)
        assert np.allclose(two[r]['batch_C'], one['batch_C'], rtol=0, atol=1e-3)
        assert np.allclose(two[r]['batch_tau'], one['batch_tau'], rtol=1e-3, atol=0)
        assert abs(two[r]['online_nll'][0] - one['online_nll'][0]) <= 1e-11 * abs(one['online_nll'][0])
        assert np.allclose(two[r]['online_nll'], one['online_nll'], rtol=1e-6, atol=0)
        assert np.allclose(two[r]['online_C'], one['online_C'], rtol=0, atol=1e-3)
        assert np.allclose(two[r]['online_tau'], one['online_tau'], rtol=1e-3, atol=0)
        assert two[r]['batch_C'] == two[0]['batch_C'] and two[r]['online_tau'] == two[0]['online_tau']     # replicas agree bit for bit
        # the lockstep timescale update with prior: same bits on every rank, the 1-rank run's numbers, and where TNC stops
        assert two[r]['online_lockstep_tau'] == two[0]['online_lockstep_tau'] and two[r]['online_lockstep_C'] == two[0]['online_lockstep_C']
        assert np.allclose(two[r]['online_lockstep_tau'], one['online_lockstep_tau'], rtol=1e-6, atol=0)
        assert np.allclose(two[r]['online_lockstep_tau'], one['online_tau'], rtol=1e-3, atol=0)
        assert np.allclose(two[r]['online_lockstep_nll'], one['online_nll'], rtol=1e-5, atol=0)
        assert np.allclose(two[r]['sample_mean_counts'], one['sample_mean_counts'], rtol=1e-14, atol=0)      # all-reduced integer moments
        assert two[r]['stale_raises'] == two[r]['stale_expected'] and two[r]['cached_read_survives']
        # the failing trial lives on rank 1; both ranks raise the same verdict and the next collectives still line up
        assert two[r]['fail_message'] is not None and 'trial 6 on rank 1' in two[r]['fail_message'] and 'positive definite' in two[r]['fail_message']
        assert two[r]['fail_message'] == two[0]['fail_message']
        assert two[r]['after_fail_sum'] == 2.0 and abs(two[r]['after_fail_nll'] - one['after_fail_nll']) <= 1e-6 * abs(one['after_fail_nll'])
        assert two[r]['after_fail_nll'] == two[0]['after_fail_nll']
    # minibatches: the same draw on both ranks, split in order (first half / second half of the reference's index list)
    for it in range(3):
        whole = one['online_estep_trials'][it]
        assert two[0]['online_estep_trials'][it] + two[1]['online_estep_trials'][it] == whole
    np.random.seed(1)
    assert one['online_estep_trials'][0] == np.random.choice(7, 4, replace=False).tolist()
    assert one['stale_raises'] == [True] * 4 and one['cached_read_survives']
    assert one['fail_message'] is not None and 'trial 6' in one['fail_message'] and 'rank' not in one['fail_message']


@pytest.mark.timeout(1500)
def test_eight_ranks_uneven_slices(tmp_path):
    """13 trials over 8 ranks (slices of 2,2,2,2,2,1,1,1): batch EM and the collective failure verdict against the 1-rank run."""
    one = _run(tmp_path, 1, 0, NTRIALS='13', LIGHT='1')[0]
    many = _run(tmp_path, 8, 29655, NTRIALS='13', LIGHT='1')
    bounds = [r['batch_slice'] for r in many]
    assert bounds == [[0, 2], [2, 4], [4, 6], [6, 8], [8, 10], [10, 11], [11, 12], [12, 13]]
    for r in range(8):
        lo, hi = bounds[r]
        assert many[r]['batch_estep_trials'] == [list(range(lo, hi))] * 2
        assert abs(many[r]['batch_nll'][0] - one['batch_nll'][0]) <= 1e-11 * abs(one['batch_nll'][0])
        assert np.allclose(many[r]['batch_nll'], one['batch_nll'], rtol=1e-6, atol=0)
        assert np.allclose(many[r]['batch_C'], one['batch_C'], rtol=0, atol=1e-3)
        assert np.allclose(many[r]['batch_tau'], one['batch_tau'], rtol=1e-3, atol=0)
        assert many[r]['batch_C'] == many[0]['batch_C']
        assert many[r]['fail_message'] == many[0]['fail_message'] and 'trial 12 on rank 7' in many[r]['fail_message']
        assert many[r]['after_fail_sum'] == 8.0 and abs(many[r]['after_fail_nll'] - one['after_fail_nll']) <= 1e-6 * abs(one['after_fail_nll'])
        assert many[r]['after_fail_nll'] == many[0]['after_fail_nll']
        # the minibatch of 8 split over the 8 ranks (config 4's form): one trial each, the reference's draw, the same parameters everywhere
        assert abs(many[r]['online_nll'][0] - one['online_nll'][0]) <= 1e-11 * abs(one['online_nll'][0])
        assert np.allclose(many[r]['online_C'], one['online_C'], rtol=0, atol=1e-3) and np.allclose(many[r]['online_tau'], one['online_tau'], rtol=1e-3, atol=0)
        assert many[r]['online_C'] == many[0]['online_C'] and many[r]['online_tau'] == many[0]['online_tau']
    for it in range(2):
        whole = one['online_estep_trials'][it]
        assert sum((many[r]['online_estep_trials'][it] for r in range(8)), []) == whole and all(len(many[r]['online_estep_trials'][it]) == 1 for r in range(8))
    np.random.seed(1)
    assert one['online_estep_trials'][0] == np.random.choice(13, 8, replace=False).tolist()
