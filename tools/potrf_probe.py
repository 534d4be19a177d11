"""Phase timings of potrf_diag_kernel (128 x 128 diagonal block: Cholesky steps + triangular inverse) for the batch sizes it runs at:
1 (shared preconditioner), 40 (timescale candidates), 1024 (per-trial r x r systems)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'poisson-gpfa_amd'))
from funs import _hip
ctx = _hip.Context(8, 2, 16, 2, 10.0)
for batch in (1, 40, 256, 1024):
    t = [ctx.bench_potrf_diag(batch, 50, ph) for ph in (0, 1, 3, 5, 7)]     # (+ 4: the round-1 form of the Cholesky steps, a row x 32 columns per thread)
    print('batch %4d: load/store %.1f us, + Cholesky steps %.1f us, + inverse %.1f us   (round-1 steps: %.1f, %.1f)' % (batch, *t), flush=True)
ctx.close()
