#!/usr/bin/env python3
"""Golden fixture at config 3's dimensions (200 neurons, 10 latents, 500 bins), captured by IMPORTING the real reference:
one trial of util.dataset(seed 12) pushed through inference.laplace (inference.py:67-185; ~5 minutes and ~9 GB on 8 cores:
15 dense (5000 x 100000)(100000 x 5000) Hessian products), then polished by Newton steps on the reference's own callbacks
(negLogPosteriorUnNorm_grad / _hess, inference.py:34-65) and the covariance blocks re-sliced from the inverse of the
reference Hessian at the polished mode (inference.py:130-131, 164-172).

    python tests/golden/make_golden_c3.py        # writes tests/golden/c3_spot.npz (~0.3 MB)
"""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                # noqa: E402

np, util, inference = mg.np, mg.util, mg.inference


def blocks(cov, p, T):
    vsm = np.stack([cov[t::T, t::T] for t in range(T)])                                  # inference.py:169-172
    gp_diag = np.stack([np.diag(cov[k * T:(k + 1) * T, k * T:(k + 1) * T]) for k in range(p)])
    return vsm, gp_diag


def main():
    q, p, T = 200, 10, 500
    t0 = time.time()
    with mg.quiet():
        ds = util.dataset(trialDur=10 * T, binSize=10, numTrials=1, xdim=p, ydim=q, seed=12)
        np.random.seed(0)
        init = util.initializeParams(p, q, ds)
    init = {k: np.real(np.asarray(v)).astype(np.float64) for k, v in init.items()}
    Y = mg.stack_Y(ds)
    assert Y.max() < 256 and Y.min() >= 0
    print('dataset + init: %.1f s' % (time.time() - t0), flush=True)
    t0 = time.time()
    res, nll, opt = inference.laplace(ds, dict(init))
    print('inference.laplace: %.1f s, nll %.6f' % (time.time() - t0, nll), flush=True)
    raw_vsm, raw_gp_diag = res['post_vsm'][0], np.stack([np.diag(res['post_vsmGP'][0][:, :, k]) for k in range(p)])
    raw_mean = res['post_mean'][0]
    gp_rows = res['post_vsmGP'][0][::50, :, :].copy()                    # every 50th row of each T x T block
    del res
    # polish on the reference's callbacks
    C_big, d_big = util.makeCd_big(init, T)
    K_big, K = util.makeK_big(dict(init), ds.trialDur, ds.binSize)
    K_bigInv = np.linalg.inv(K_big)
    yb = Y[0].reshape(-1).astype(float)
    x = np.array(opt[0], dtype=np.float64)
    steps = []
    H = None
    for it in range(6):
        t0 = time.time()
        g = inference.negLogPosteriorUnNorm_grad(x, yb, C_big, d_big, K_bigInv, p, q)
        H = inference.negLogPosteriorUnNorm_hess(x, yb, C_big, d_big, K_bigInv, p, q)
        dx = np.linalg.solve(H, g)
        x = x - dx
        steps.append(float(np.max(np.abs(dx))))
        print('polish %d: max|dx| %.3e (%.1f s)' % (it, steps[-1], time.time() - t0), flush=True)
        if steps[-1] < 1e-11:
            break
    f_pol = float(inference.negLogPosteriorUnNorm(x, yb, C_big, d_big, K_bigInv, p, q))
    g_pol = inference.negLogPosteriorUnNorm_grad(x, yb, C_big, d_big, K_bigInv, p, q)
    H = inference.negLogPosteriorUnNorm_hess(x, yb, C_big, d_big, K_bigInv, p, q)
    cov = np.linalg.inv(H)
    pol_vsm, pol_gp_diag = blocks(cov, p, T)
    pol_gp_rows = np.stack([cov[k * T:(k + 1) * T, k * T:(k + 1) * T][::50, :] for k in range(p)], axis=2)   # (10, T, p)
    np.savez_compressed(
        os.path.join(HERE, 'c3_spot.npz'), Y=Y.astype(np.uint8), init_C=init['C'], init_d=init['d'], init_tau=init['tau'],
        binSize=ds.binSize, nll=nll, post_mean=raw_mean, post_vsm=raw_vsm, post_vsmGP_diag=raw_gp_diag, post_vsmGP_rows=gp_rows,
        polished=x, polished_f=f_pol, polished_grad_max=float(np.max(np.abs(g_pol))), polish_steps=np.array(steps),
        polished_vsm=pol_vsm, polished_vsmGP_diag=pol_gp_diag, polished_vsmGP_rows=pol_gp_rows)
    print('c3_spot.npz written; raw vs polished mode max|dx| = %.3e' % np.max(np.abs(raw_mean.reshape(-1) - x)))


if __name__ == '__main__':
    main()
