import sys, time, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'poisson-gpfa_amd')); sys.path.insert(0, ROOT)
import bench, funs
from funs import learning, inference, util, _session
q, p, T, R = bench.CONFIGS['c3']
true_params, Ys = bench.synth_shard(q, p, T, R, 12, 0)
exp = bench.Shard(Ys, 10.0)
sess, _ = _session.session_for(exp, p)
np.random.seed(0)
init = funs.util.initializeParams(p, q, exp)
params = {k: np.real(np.asarray(v)).astype(np.float64) for k, v in init.items()}
optim = None
orig_eval = sess.ctx.mstep_tau_costgrad_batch
cnt = {'tau': 0, 'tau_ms': 0.0}
def counted(pv):
    t0 = time.time(); r = orig_eval(pv); cnt['tau_ms'] += (time.time() - t0) * 1e3; cnt['tau'] += 1
    if os.environ.get('TAU_TRACE'): print('   tau eval', cnt['tau'], 'p0 %.12f' % pv[0], 'max|g| %.3e' % np.max(np.abs(r[1])), 'g0 %.3e' % r[1][0])
    return r
sess.ctx.mstep_tau_costgrad_batch = counted
orig_pass = sess.ctx.mstep_cd_newton_pass
def cpass(*a):
    t0 = time.time(); r = orig_pass(*a); cnt['cd_ms'] = cnt.get('cd_ms', 0) + (time.time() - t0) * 1e3; cnt['cd'] = cnt.get('cd', 0) + 1; return r
sess.ctx.mstep_cd_newton_pass = cpass
orig_chord = sess.ctx.mstep_cd_chord_pass
def cchord(*a):
    t0 = time.time(); r = orig_chord(*a); cnt['cd_ms'] = cnt.get('cd_ms', 0) + (time.time() - t0) * 1e3; cnt['chord'] = cnt.get('chord', 0) + 1; return r
sess.ctx.mstep_cd_chord_pass = cchord
for it in range(5):
    infRes, nll, optim = funs.inference.laplace(exp, params, prevOptimRes=optim)
    cnt.update(tau=0, tau_ms=0.0, cd=0, cd_ms=0.0, chord=0)
    t0 = time.time()
    C, d, _ = learning.learnLTparams(params, infRes, exp, 'newton')
    t1 = time.time()
    tau, det = learning.learnGPparams(params, infRes, exp)
    t2 = time.time()
    params = {'C': C, 'd': d, 'tau': tau}
    print('it', it, 'cd %.1f ms (%d full + %d chord passes, %.1f ms in passes)' % ((t1 - t0) * 1e3, cnt['cd'], cnt['chord'], cnt['cd_ms']),
          'tau total %.1f ms (%d evals, %.1f ms in evals)' % ((t2 - t1) * 1e3, cnt['tau'], cnt['tau_ms']), flush=True)
