"""Bring up a 1-rank RCCL communicator through the C-ABI and all-reduce once (bootstrap diagnostics)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'poisson-gpfa_amd'))
os.environ.setdefault('PGPFA_FORCE_COMM', '1')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1'); os.environ.setdefault('LOCAL_RANK', '0')
import numpy as np
t0 = time.time()
from funs import _hip, _session
ctx = _hip.Context(4, 2, 16, 2, 10.0)
print('context %.2fs' % (time.time() - t0), flush=True)
uid = _hip.comm_unique_id()
print('unique id %.2fs' % (time.time() - t0), flush=True)
ctx.comm_init(uid, 0, 1)
print('comm init %.2fs' % (time.time() - t0), flush=True)
print('allreduce', ctx.allreduce_host(np.ones(3)), '%.2fs' % (time.time() - t0), flush=True)
