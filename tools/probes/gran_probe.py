"""rank_gran 4 against 16 on one of the small test shapes: statuses, dense retries, modes / post_vsm / PautoSum differences.
usage: python tools/probes/gran_probe.py q p T R"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'poisson-gpfa_amd'))
from funs import _hip
from oracle import pgpfa_oracle as orc
q, p, T, R = (int(a) for a in sys.argv[1:5])
rng = np.random.default_rng(q * 1000 + p)
_, Ys, _ = orc.synth_dataset(q, p, T, R, seed=p, dOffset=0.0)
Y = np.stack(Ys).astype(np.uint8)
par = {'C': 0.3 * rng.standard_normal((q, p)) / np.sqrt(max(1, p / 4)), 'd': np.log(Y.mean(axis=(0, 2)) + 0.1), 'tau': 0.05 + 0.3 * rng.random(p)}
if os.environ.get('PROBE_TAU'):
    par['tau'] = np.full(p, float(os.environ['PROBE_TAU']))          # (every timescale the same: PROBE_TAU=0.004 at 10-ms bins is full rank)
out = {}
for g in (16, 4):
    ctx = _hip.Context(q, p, T, R, 10.0)
    ctx.upload_counts(Y)
    ctx.set_option('cov_mode', 2); ctx.set_option('pcg_fused', 2); ctx.set_option('rank_gran', g)
    for kv in sys.argv[5:]:
        ctx.set_option(kv.split('=')[0], float(kv.split('=')[1]))
    ctx.set_params(par['C'], par['d'], par['tau'])
    try:
        obj, it, st = ctx.estep_laplace()
        print('gran %d: rank %d, plan_lowrank %d, retries %d (no-descent %d, search %d, cap %d), pcg %d, status %s' % (
            g, ctx.info('lowrank_rtot'), ctx.info('plan_lowrank'), ctx.info('last_dense_retries'), ctx.info('last_fallback_no_descent'),
            ctx.info('last_fallback_line_search'), ctx.info('last_fallback_outer_cap'), ctx.info('last_pcg_iterations'), st.tolist()))
        ctx.mstep_precomp()
        out[g] = (obj, ctx.post_mean().copy(), ctx.post_vsm().copy(), ctx.pautosum().copy())
    except Exception as exc:
        print('gran %d: %s' % (g, exc))
    ctx.close()
if 16 in out and 4 in out:
    a, b = out[4], out[16]
    print('obj rel diff %.2e, modes %.2e, vsm rel %.2e, PautoSum rel %.2e' % (abs(a[0] - b[0]) / abs(b[0]), np.max(np.abs(a[1] - b[1])),
          np.max(np.abs(a[2] - b[2])) / np.max(np.abs(b[2])), np.max(np.abs(a[3] - b[3])) / np.max(np.abs(b[3]))))
