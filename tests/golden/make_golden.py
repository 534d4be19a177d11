#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by IMPORTING the real reference.

Runs only in the build container, where the reference is mounted read-only at
/root/reference; the fixtures (inputs + expected outputs, data only) are committed,
the reference never is.  Accommodations:
  * statsmodels is not installed but is imported at module top by the reference
    (learning.py:9-10, util.py:11): a stand-in module is written to a temp dir.
  * cwd must be the reference root (funs/__init__.py:9), MPLBACKEND=Agg.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz
"""
import io
import os
import sys
import tempfile
import contextlib

REF = os.environ.get('PGPFA_REFERENCE', '/root/reference')
OUT = os.path.dirname(os.path.abspath(__file__))

os.environ.setdefault('MPLBACKEND', 'Agg')
sys.dont_write_bytecode = True

_stub = tempfile.mkdtemp(prefix='sm_stub_')
os.makedirs(os.path.join(_stub, 'statsmodels', 'tools'))
open(os.path.join(_stub, 'statsmodels', '__init__.py'), 'w').close()
open(os.path.join(_stub, 'statsmodels', 'tools', '__init__.py'), 'w').close()
with open(os.path.join(_stub, 'statsmodels', 'tools', 'numdiff.py'), 'w') as fh:
    fh.write(
        "import numpy as np\n"
        "EPS = np.finfo(float).eps\n"
        "def _get_epsilon(x, s, epsilon, n):\n"
        "    if epsilon is None:\n"
        "        return EPS ** (1.0 / s) * np.maximum(np.abs(x), 0.1)\n"
        "    return epsilon\n"
        "def approx_fprime(*a, **k):\n    raise NotImplementedError\n"
        "def approx_hess(*a, **k):\n    raise NotImplementedError\n")
sys.path.insert(0, _stub)
os.chdir(REF)
sys.path.insert(0, REF)

import numpy as np                      # noqa: E402
import funs.util as util                # noqa: E402
import funs.engine as engine            # noqa: E402
import inference                        # noqa: E402  (bare copies are what engine calls)
import learning                         # noqa: E402
import scipy.optimize as op             # noqa: E402


@contextlib.contextmanager
def quiet():
    with contextlib.redirect_stdout(io.StringIO()):
        yield


def stack_Y(ds):
    return np.stack([tr['Y'] for tr in ds.data])


def param_seq_arrays(seq):
    return (np.stack([s['C'] for s in seq]), np.stack([s['d'] for s in seq]),
            np.stack([np.asarray(s['tau']).reshape(-1) for s in seq]))


def main():
    # ---------------- config 1: util.dataset() defaults == (30 n, 3 lat, T=100, 20 trials) -------
    with quiet():
        ds = util.dataset()                         # seed 12, util.py:659-671
        np.random.seed(0)
        init = util.initializeParams(3, 30, ds)
    Y = stack_Y(ds)
    assert Y.max() < 256 and Y.min() >= 0
    q, p, T, R = 30, 3, 100, 20
    true = ds.params
    np.savez_compressed(
        os.path.join(OUT, 'c1_dataset.npz'),
        Y=Y.astype(np.uint8), X=np.stack([tr['X'] for tr in ds.data]),
        true_C=true['C'], true_d=true['d'], true_tau=np.asarray(true['tau']).reshape(-1),
        init_C=init['C'], init_d=init['d'], init_tau=init['tau'],
        binSize=ds.binSize, trialDur=ds.trialDur)

    # ---------------- a1/a2: builders ------------------------------------------------------------
    K_big, K = util.makeK_big(dict(init), ds.trialDur, ds.binSize)
    C_big, d_big = util.makeCd_big(init, T)
    K_bigInv = np.linalg.inv(K_big)

    # ---------------- a4-a6: callbacks at a random point -----------------------------------------
    rng = np.random.RandomState(7)
    xprobe = 0.3 * rng.randn(p * T)
    ybar = Y[0].reshape(-1).astype(float)
    f = inference.negLogPosteriorUnNorm(xprobe, ybar, C_big, d_big, K_bigInv, p, q)
    g = inference.negLogPosteriorUnNorm_grad(xprobe, ybar, C_big, d_big, K_bigInv, p, q)
    H = inference.negLogPosteriorUnNorm_hess(xprobe, ybar, C_big, d_big, K_bigInv, p, q)
    np.savez_compressed(os.path.join(OUT, 'c1_callbacks.npz'),
                        K=K, xprobe=xprobe, f=f, g=g, H=H, vecCd0=util.CdtoVecCd(init['C'], init['d']))

    # ---------------- a7: Laplace E-step at the initial parameters -------------------------------
    infRes, nll, lapOpt = inference.laplace(ds, dict(init))
    # polished modes: Newton on the reference's own callbacks until the step is ~1e-13
    polished = []
    for r in range(R):
        yb = Y[r].reshape(-1).astype(float)
        x = lapOpt[r].copy()
        for _ in range(50):
            gg = inference.negLogPosteriorUnNorm_grad(x, yb, C_big, d_big, K_bigInv, p, q)
            HH = inference.negLogPosteriorUnNorm_hess(x, yb, C_big, d_big, K_bigInv, p, q)
            dx = np.linalg.solve(HH, gg)
            x = x - dx
            if np.max(np.abs(dx)) < 1e-13:
                break
        polished.append(x)
    precomp = learning.makePrecomp(infRes)
    np.savez_compressed(
        os.path.join(OUT, 'c1_laplace.npz'),
        nll=nll, post_mean=np.stack(infRes['post_mean']), post_vsm=np.stack(infRes['post_vsm']),
        post_vsmGP_diag=np.stack([np.stack([np.diag(G[:, :, k]) for k in range(p)]) for G in infRes['post_vsmGP']]),
        post_vsmGP_trial0=infRes['post_vsmGP'][0],
        post_cov_trial0=infRes['post_cov'][0],
        PautoSum=np.stack([pc['PautoSum'] for pc in precomp]),
        polished=np.stack(polished))

    # ---------------- a9/a10: (C,d) M-step --------------------------------------------------------
    v0 = util.CdtoVecCd(init['C'], init['d'])
    v1 = v0 + 0.05 * rng.randn(v0.size)
    cost0 = learning.MStepObservationCost(v0, p, q, ds, infRes)
    grad0 = learning.MStepObservationCost_grad(v0, p, q, ds, infRes)
    cost1 = learning.MStepObservationCost(v1, p, q, ds, infRes)
    grad1 = learning.MStepObservationCost_grad(v1, p, q, ds, infRes)
    inv_prior = -np.eye(v0.size) / 0.7 ** 2
    costp = learning.MStepObservationCostWithPrior(v1, init, p, q, ds, infRes, inv_prior)
    gradp = learning.MStepObservationCostWithPrior_grad(v1, init, p, q, ds, infRes, inv_prior)
    newC, newd, cdcost = learning.learnLTparams(dict(init), infRes, ds, 'TNC')
    # tight optimum of the same cost (L-BFGS-B then BFGS to |grad| ~ 1e-7), SURVEY 8c
    tight = op.minimize(learning.MStepObservationCost, util.CdtoVecCd(newC, newd), args=(p, q, ds, infRes),
                        jac=learning.MStepObservationCost_grad, method='L-BFGS-B',
                        options={'maxiter': 5000, 'ftol': 1e-15, 'gtol': 1e-10})
    tight = op.minimize(learning.MStepObservationCost, tight.x, args=(p, q, ds, infRes),
                        jac=learning.MStepObservationCost_grad, method='BFGS', options={'gtol': 1e-9})

    # ---------------- a11: tau M-step --------------------------------------------------------------
    pprobe = np.array([np.log(1.0 / 20.0 ** 2), np.log(1.0 / 45.0 ** 2), -5.0])
    tcost = np.array([[learning.MStepGPtimescaleCost(pp, precomp[k], 0.001) for pp in pprobe] for k in range(p)])
    tgrad = np.array([[float(np.asarray(learning.MStepGPtimescaleCost_grad(pp, precomp[k], 0.001)).reshape(-1)[0])
                       for pp in pprobe] for k in range(p)])
    tcostp = np.array([[learning.MStepGPtimescaleCostWithPrior(pp, precomp[k], 0.001, ds.binSize, init['tau'][k], 0.6)
                        for pp in pprobe] for k in range(p)])
    tgradp = np.array([[float(np.asarray(learning.MStepGPtimescaleCostWithPrior_grad(
        pp, precomp[k], 0.001, ds.binSize, init['tau'][k], 0.6)).reshape(-1)[0]) for pp in pprobe] for k in range(p)])
    newTau, taudet = learning.learnGPparams(dict(init), infRes, ds)
    np.savez_compressed(
        os.path.join(OUT, 'c1_mstep.npz'),
        v0=v0, v1=v1, cost0=cost0, grad0=grad0, cost1=cost1, grad1=grad1, costp=costp, gradp=gradp,
        prior_step=0.7, newC=newC, newd=newd, cdcost=cdcost, tight_vec=tight.x, tight_cost=tight.fun,
        pprobe=pprobe, tcost=tcost, tgrad=tgrad, tcostp=tcostp, tgradp=tgradp, tau_prior_step=0.6,
        newTau=newTau, tau_logp=np.array([float(np.asarray(dd.x).reshape(-1)[0]) for dd in taudet]))

    # ---------------- full batch EM, 5 iterations ---------------------------------------------------
    with quiet():
        fit = engine.PPGPFAfit(ds, initParams={k: np.array(v) for k, v in init.items()},
                               inferenceMethod='laplace', EMmode='Batch', maxEMiter=5)
    sC, sd, st = param_seq_arrays(fit.paramSeq)
    np.savez_compressed(os.path.join(OUT, 'c1_em_batch.npz'),
                        nll=np.asarray(fit.posteriorLikelihood), seq_C=sC, seq_d=sd, seq_tau=st,
                        final_post_mean=np.stack(fit.infRes['post_mean']))

    # ---------------- online 'diag' EM (a13/a15) -----------------------------------------------------
    np.random.seed(1)
    with quiet():
        fit = engine.PPGPFAfit(ds, initParams={k: np.array(v) for k, v in init.items()},
                               inferenceMethod='laplace', EMmode='Online', maxEMiter=4, batchSize=5,
                               onlineParamUpdateMethod='diag')
    np.random.seed(1)
    idx = np.stack([np.random.choice(R, 5, replace=False) for _ in range(4)])
    sC, sd, st = param_seq_arrays(fit.paramSeq)
    np.savez_compressed(os.path.join(OUT, 'c1_em_online.npz'),
                        nll=np.asarray(fit.posteriorLikelihood), seq_C=sC, seq_d=sd, seq_tau=st, batchTrIdx=idx)

    # ---------------- variational toy (a8): 20 n / 2 lat / T=50 / 10 trials, seed 5 ------------------
    with quiet():
        dv = util.dataset(trialDur=500, binSize=10, numTrials=10, xdim=2, ydim=20, seed=5)
        np.random.seed(0)
        initv = util.initializeParams(2, 20, dv)
    Yv = stack_Y(dv)
    out = {}
    for name, loglam, nit in (('bounded', False, 3), ('loglam', True, 2)):
        with quiet():
            fv = engine.PPGPFAfit(dv, initParams={k: np.array(v) for k, v in initv.items()},
                                  inferenceMethod='variational', EMmode='Batch', maxEMiter=nit,
                                  optimLogLamb=loglam)
        sC, sd, st = param_seq_arrays(fv.paramSeq)
        out.update({name + '_nll': np.asarray(fv.posteriorLikelihood), name + '_vlb': np.asarray(fv.variationalLowerBound),
                    name + '_seq_C': sC, name + '_seq_d': sd, name + '_seq_tau': st})
    # one E-step at the initial parameters, with the dual callbacks probed
    vres, vnll, vvlb, vopt = inference.dualVariational(dv, dict(initv))
    Cb, db = util.makeCd_big(initv, 50)
    Kb, _ = util.makeK_big(dict(initv), dv.trialDur, dv.binSize)
    Kbi = np.linalg.inv(Kb)
    lam_probe = 0.1 + rng.rand(20 * 50)
    yb = Yv[0].reshape(-1).astype(float)
    out.update(dict(
        Y=Yv.astype(np.uint8), init_C=initv['C'], init_d=initv['d'], init_tau=initv['tau'],
        binSize=dv.binSize, estep_nll=vnll, estep_vlb=vvlb, estep_lambda=np.stack(vopt),
        estep_post_mean=np.stack(vres['post_mean']), estep_post_vsm=np.stack(vres['post_vsm']),
        lam_probe=lam_probe, dual_cost=inference.dualProblem(lam_probe, yb, Cb, Kb, Kbi, db),
        dual_grad=inference.dualProblem_grad(lam_probe, yb, Cb, Kb, Kbi, db)))
    np.savez_compressed(os.path.join(OUT, 'var_toy.npz'), **out)

    # ---------------- config-2-size spot check: 100 n / 5 lat / T=200, 2 trials -----------------------
    with quiet():
        d2 = util.dataset(trialDur=2000, binSize=10, numTrials=2, xdim=5, ydim=100, seed=12)
        np.random.seed(0)
        init2 = util.initializeParams(5, 100, d2)
    r2, nll2, opt2 = inference.laplace(d2, dict(init2))
    np.savez_compressed(
        os.path.join(OUT, 'c2_spot.npz'), Y=stack_Y(d2).astype(np.uint8),
        init_C=init2['C'], init_d=init2['d'], init_tau=init2['tau'], binSize=d2.binSize,
        nll=nll2, post_mean=np.stack(r2['post_mean']), post_vsm=np.stack(r2['post_vsm']),
        post_vsmGP_diag=np.stack([np.stack([np.diag(G[:, :, k]) for k in range(5)]) for G in r2['post_vsmGP']]))
    print('golden fixtures written to', OUT)


if __name__ == '__main__':
    main()
