#!/usr/bin/env python3
"""Throughput of leave-one-neuron-out prediction on one GPU, with the oracle's faithful CPU
restatement of util.leaveOneOutPrediction timed beside it on a few searches.  Prints one JSON line.
    python tools/loo_bench.py --config c2 [--trials N] [--cpu-searches K]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'poisson-gpfa_amd'))
sys.path.insert(0, ROOT)
import bench            # noqa: E402
import funs             # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--config', default='c2', choices=sorted(bench.CONFIGS))
    ap.add_argument('--trials', type=int, default=0)
    ap.add_argument('--cpu-searches', type=int, default=2)
    args = ap.parse_args()
    q, p, T, R = bench.CONFIGS[args.config]
    if args.trials > 0:
        R = args.trials
    true_params, Ys = bench.synth_shard(q, p, T, R, 12, 0)
    exp = bench.Shard(Ys, 10.0)
    params = {k: np.asarray(v, dtype=np.float64) for k, v in true_params.items()}
    # warm-up on the same device session (context creation, workspace allocation, code objects): one trial's searches
    from funs import _session
    sess, _ = _session.session_for(exp, p)
    sess.set_params(params)
    sess.ctx.loo_predict(np.array([0], dtype=np.int32))
    t0 = time.time()
    y_pred, err = funs.util.leaveOneOutPrediction(params, exp)
    dt = time.time() - t0
    out = {'metric': 'leave-one-neuron-out mode searches/s', 'config': args.config, 'neurons': q, 'latents': p, 'bins': T, 'trials': R,
           'searches': R * q, 'seconds': dt, 'value': R * q / dt, 'pred_err_mode': err}
    if args.cpu_searches > 0:
        from oracle import pgpfa_oracle as orc
        k = args.cpu_searches
        # k searches = neurons 0..k-1 of trial 0 (the oracle loops neurons inside a trial; time per search is uniform)
        Ysub = [np.asarray(Ys[0], dtype=np.float64)]
        t0 = time.time()
        C, d = params['C'], params['d']
        K = orc.make_K(params['tau'], T, 10.0)
        K_bigInv = np.linalg.inv(orc.make_K_big(K))
        import scipy.optimize as op
        worst = 0.0
        for n in range(k):
            Cw, dw, Yw = np.delete(C, n, 0), np.delete(d, n, 0), np.delete(Ysub[0], n, 0)
            C_big, d_big = orc.make_Cd_big(Cw, dw, T)
            x = op.fmin_ncg(orc.nlp_big, np.zeros(p * T), fprime=orc.nlp_big_grad, fhess=orc.nlp_big_hess,
                            args=(Yw.reshape(-1), C_big, d_big, K_bigInv), disp=False)
            yp = np.exp(C[n] @ x.reshape(p, T) + d[n])
            worst = max(worst, float(np.max(np.abs(yp - y_pred[0, n]) / yp)))
        cpu = (time.time() - t0) / k
        out['cpu_baseline'] = {'kind': 'port', 'seconds_per_search': cpu, 'value': 1.0 / cpu, 'sample': '%d searches of trial 0' % k,
                               'cores': os.cpu_count(), 'max_rel_diff_of_predictions': worst}
        out['speedup_vs_cpu_baseline'] = out['value'] * cpu
    print(json.dumps(out))


if __name__ == '__main__':
    main()
