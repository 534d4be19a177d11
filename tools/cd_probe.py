"""Timing probe of the (C,d) cost/gradient sweep at config-3 dimensions: E-step on R trials, then repeated sweeps under the
kernel's debug switches (option cd_debug: bit 0 no exp, 1 no second product, 2 no first product, 3 no staging)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'poisson-gpfa_amd')); sys.path.insert(0, ROOT)
import numpy as np
import bench
from funs import _hip
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
q, p, T, _ = bench.CONFIGS['c3']
true, Ys = bench.synth_shard(q, p, T, R, 12, 0)
ctx = _hip.Context(q, p, T, R, 10.0)
ctx.upload_counts(np.stack(Ys))
par = {'C': true['C'], 'd': true['d'], 'tau': np.linspace(0.1, 0.5, p)}
ctx.set_params(par['C'], par['d'], par['tau'])
ctx.estep_laplace()
v = np.concatenate([par['C'].T.ravel(), par['d']])
ref = None
for mode, dbg in (('vector kernel', -1), ('mfma', 0), ('no exp', 1), ('no 2nd product', 2), ('no 1st product', 4), ('no staging', 8), ('no products', 6), ('nothing', 15)):
    ctx.set_option('cd_mfma', 0 if dbg < 0 else 1)
    ctx.set_option('cd_debug', max(dbg, 0))
    c0, g0 = ctx.mstep_cd_costgrad(v)
    t0 = time.time()
    for _ in range(20):
        ctx.mstep_cd_costgrad(v)
    dt = (time.time() - t0) / 20
    if ref is None:
        ref = (c0, g0)
    print('%-16s %.3f ms  cost %.12g  max|dgrad| vs vector %.2e' % (mode, dt * 1e3, c0, np.max(np.abs(g0 - ref[1]))), flush=True)
# the Newton pass (cost + gradient + per-neuron Hessians): two-stage matrix-core form against the vector kernel
ctx.set_option('cd_mfma', 1); ctx.set_option('cd_debug', 0)
refn = None
for name, form, dbg in (('hess vector', 0, 0), ('hess mfma', 1, 0)):
    ctx.set_option('cd_hess_mfma', form)
    ctx.set_option('cd_debug', dbg)
    out = ctx.mstep_cd_newton_pass(v)
    t0 = time.time()
    for _ in range(10):
        ctx.mstep_cd_newton_pass(v)
    dt = (time.time() - t0) / 10
    if refn is None:
        refn = out
    print('%-16s %.3f ms  sum cost_n %.12g  max|ddelta| vs vector %.2e  max|ddec| %.2e' % (name, dt * 1e3, out[0].sum(), np.max(np.abs(out[1] - refn[1])), np.max(np.abs(out[2] - refn[2]))), flush=True)
