import os, sys, time
import numpy as np
ROOT='/root/repo'
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'poisson-gpfa_amd'))
import bench, funs
from funs import _session
q,p,T,R=200,10,500,1024
true_params, Ys = bench.synth_shard(q, p, T, R, 0, 0)
exp = bench.Shard(Ys, 10.0)
sess,_=_session.session_for(exp,p); c=sess.ctx
np.random.seed(0)
params = {k: np.real(np.asarray(v)).astype(np.float64) for k, v in funs.util.initializeParams(p, q, exp).items()}
sess.set_params(params)
one=np.array([0],dtype=np.int32)
for i in range(5):
    t=time.time(); c.estep_laplace(one); print('1 trial call %d: %.1f ms (library %.1f) plans %d plan_ms %.1f chunk %d lowrank %d'%(i,(time.time()-t)*1e3,c.info('last_estep_ms'),c.info('plans'),c.info('plan_ms_total'),c.info('chunk_trials'),c.info('plan_lowrank')),flush=True)
for i in range(3):
    t=time.time(); c.estep_laplace(None); print('all call %d: %.1f ms (library %.1f) plans %d plan_ms %.1f chunk %d'%(i,(time.time()-t)*1e3,c.info('last_estep_ms'),c.info('plans'),c.info('plan_ms_total'),c.info('chunk_trials')),flush=True)
