"""Does a closed context give its device memory back?  Free HBM before / inside / after a config-5-sized variational E-step of 64 trials, with the
workspace arena mapped through HIP's virtual-memory calls (default) and as plain allocations.  usage: python tools/leak_probe.py [trials]"""
import gc, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'poisson-gpfa_amd'))
from funs import _hip

R = int(sys.argv[1]) if len(sys.argv) > 1 else 64
q, p, T = 500, 20, 1000


def free_gb():
    c = _hip.Context(2, 1, 4, 1, 10.0)
    try:
        return c.info('hbm_bytes_free') / 1e9
    finally:
        c.close()


rng = np.random.default_rng(0)
Y = rng.poisson(0.2, (R, q, T)).astype(np.uint8)
C, d, tau = rng.random((q, p)) - 0.5, -2.0 - rng.random(q), np.linspace(0.1, 0.5, p)
print('start: %.1f GB free' % free_gb())
for vmm in (1, 0, 1):
    ctx = _hip.Context(q, p, T, R, 10.0)
    ctx.set_option('workspace_vmm', vmm)
    ctx.upload_counts(Y)
    ctx.set_params(C, d, tau)
    ctx.set_option('dual_lowrank', 1)
    idx = np.arange(R, dtype=np.int32)
    ctx.dual_fixed_point(idx, None, want_rho=False, max_outer=2)
    ctx.dual_finalize(idx, None)
    inside = ctx.info('hbm_bytes_free') / 1e9
    own, arena = ctx.info('hbm_bytes_allocated') / 1e9, ctx.info('arena_bytes') / 1e9
    ctx.close()
    gc.collect()
    print('workspace_vmm %d: inside %.1f GB free (context %.1f GB, arena %.1f GB); after close %.1f GB free' % (vmm, inside, own, arena, free_gb()))
