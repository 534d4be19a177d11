"""Dual-variational evaluation at config-5 dimensions (500 neurons, 20 latents, 1000 bins): batched cost + gradient of R trials
through the low-rank engine, FP64 vs mixed precision (option dual_f32): time per evaluation and agreement."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'poisson-gpfa_amd')); sys.path.insert(0, ROOT)
import numpy as np
import bench
from funs import _hip
R = int(sys.argv[1]) if len(sys.argv) > 1 else 16
q, p, T = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (500, 20, 1000)
true, Ys = bench.synth_shard(q, p, T, R, 12, 0)
Y = np.stack(Ys)
rng = np.random.default_rng(12)
tau = np.linspace(0.1, 0.5, p)
lam = np.exp(true['d'])[None, :, None] * (0.5 + rng.random((R, q, T)))
lam = lam.reshape(R, -1)
idx = np.arange(R, dtype=np.int32)
out = {}
for name, f32 in (('f64', 0), ('mixed f32', 1)):
    ctx = _hip.Context(q, p, T, R, 10.0)
    ctx.upload_counts(Y)
    ctx.set_option('cov_mode', 2); ctx.set_option('dual_lowrank', 1); ctx.set_option('dual_f32', f32)
    ctx.set_params(true['C'], true['d'], tau)
    cost, grad = ctx.dual_costgrad_batch(idx, lam)
    ctx.set_option('profile', 1)
    t0 = time.time()
    reps = 3
    for _ in range(reps):
        cost, grad = ctx.dual_costgrad_batch(idx, lam)
    dt = (time.time() - t0) / reps
    prof = {k: round(ctx.info('prof_%s_ms' % k) / reps, 1) for k in ('gemm', 'potrf', 'vsm', 'assemble')}
    print('%-10s rank %d: %.1f ms per batched evaluation of %d trials (%.2f ms per trial); kernel ms %s' % (
        name, int(ctx.info('lowrank_rtot')), dt * 1e3, R, dt * 1e3 / R, prof), flush=True)
    out[name] = (cost, grad, ctx.post_vsm(idx[:1]))
    ctx.close()
c0, g0, v0 = out['f64']; c1, g1, v1 = out['mixed f32']
print('cost rel diff', np.max(np.abs(c1 - c0) / np.abs(c0)), ' grad rel diff', np.max(np.abs(g1 - g0)) / np.max(np.abs(g0)),
      ' post_vsm rel diff', np.max(np.abs(v1 - v0)) / np.max(np.abs(v0)))
