"""Two warm Laplace E-steps at config-3 dimensions (tau 0.1..0.5 s: rank ~620) - the workload of tools/pmc_passes.sh."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'poisson-gpfa_amd'))
import bench
from funs import _hip
R = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
q, p, T = 200, 10, 500
true, Ys = bench.synth_shard(q, p, T, R, 12, 0)
ctx = _hip.Context(q, p, T, R, 10.0)
ctx.upload_counts(np.stack(Ys))
ctx.set_params(true['C'], true['d'], np.linspace(0.1, 0.5, p))
for _ in range(2):
    ctx.estep_laplace()
ctx.close()
