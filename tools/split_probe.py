"""Accuracy of the split accumulation of sum_r Y~Y~^T (csrc/split.h) against the full-width FP64 product of the same engine, at config-3
dimensions with the generating parameters (short timescales: the largest eps ||Wt|| this workload produces) and with louder populations.
usage: python tools/split_probe.py [trials]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'poisson-gpfa_amd'))
import bench
from funs import _hip

R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
q, p, T = 200, 10, 500
true, Ys = bench.synth_shard(q, p, T, R, 12, 0)
Y = np.stack(Ys)
rel = lambda a, b: np.max(np.abs(a - b)) / np.max(np.abs(b))
for name, d_off, tau in (('generating parameters', 0.0, true['tau']), ('generating, rates x 3', 1.1, true['tau']), ('generating, rates x 8', 2.1, true['tau']),
                         ('tau 0.1..0.5 s', 0.0, np.linspace(0.1, 0.5, p))):
    out = []
    for split in (0, 1):
        ctx = _hip.Context(q, p, T, R, 10.0)
        ctx.set_option('cov_mode', 2); ctx.set_option('split_cov', split); ctx.set_option('split_max_norm', 100.0); ctx.set_option('measure_mix', 1)
        d = true['d'] + d_off
        if d_off:
            rng = np.random.default_rng(7)
            ctx.upload_counts(np.minimum(rng.poisson(np.exp(d)[None, :, None] * np.ones((R, 1, T))), 60000).astype(np.uint16))
        else:
            ctx.upload_counts(Y)
        ctx.set_params(true['C'], d, tau)
        obj, _, st = ctx.estep_laplace()
        ctx.mstep_precomp()
        out.append((ctx.pautosum(), (ctx.info('last_eps_wt_norm'), ctx.info('last_eps_wt_rms')), ctx.info('last_split_cov'), int(ctx.info('lowrank_rtot')), ctx.info('last_estep_ms')))
        ctx.close()
    print('%-26s rank %4d  eps||Wt|| max %.3e rms %.3e  split used %d  PautoSum split vs FP64 product: %.2e   E-step %.0f -> %.0f ms'
          % (name, out[1][3], out[1][1][0], out[1][1][1], out[1][2], rel(out[1][0], out[0][0]), out[0][4], out[1][4]))
