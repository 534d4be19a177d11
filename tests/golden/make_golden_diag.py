#!/usr/bin/env python3
"""Golden fixture for the post-fit summaries PPGPFAfit always computes: Poisson-PCA
initialiser moments, processParamResults / performSpikeCountAnalysis attributes (engine.py:487-597) and the
orthonormalised trajectories (engine.py:515-521), captured from the real reference on config 1 (3 batch EM iterations).

    python tests/golden/make_golden_diag.py        # writes tests/golden/c1_diag.npz
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                # noqa: E402

np, util, engine = mg.np, mg.util, mg.engine


def main():
    with mg.quiet():
        ds = util.dataset()                         # config 1 (seed 12)
        np.random.seed(0)
        init = util.initializeParams(3, 30, ds)
        fit = engine.PPGPFAfit(ds, initParams=dict(init), inferenceMethod='laplace', EMmode='Batch', maxEMiter=3,
                               extractAllTraj=True)
        fit.orthonormalizeTrajectories()
    out = {'init_C': init['C'], 'init_d': init['d'], 'init_tau': np.asarray(init['tau']),
           'seq_C': np.stack([p['C'] for p in fit.paramSeq]), 'seq_d': np.stack([p['d'] for p in fit.paramSeq]),
           'seq_tau': np.stack([np.asarray(p['tau']).reshape(-1) for p in fit.paramSeq]),
           'optim_C': fit.optimParams['C'], 'optim_d': fit.optimParams['d'], 'optim_tau': np.asarray(fit.optimParams['tau']).reshape(-1),
           'post_mean_all': np.stack(fit.infRes['post_mean']), 'x_tilde': fit.x_tilde}
    names = ['tauSeq', 'expectedSpikeCountsEst', 'expectedSpikeCountsEstVar', 'sampleMeanSpikeCounts', 'sampleMeanSpikeCountsVar',
             'expectedSpikeCountsTrue', 'expectedSpikeCountsTrueVar', 'varESpkCountTrue_Ratios', 'varESpkCountSampleMean_Ratios',
             'meanSquaredErrorOverTrueVariance_SM', 'subspaceAngleC', 'CabsoluteValue',
             'E_y_init_params', 'E_yy_init_params', 'E_y_optim_params', 'E_yy_optim_params', 'E_y_obs', 'E_yy_obs',
             'E_y_true_params', 'E_yy_true_params', 'mean_err_optim_true', 'mean_err_init_true', 'cov_err_optim_true', 'cov_err_init_true',
             'JSdiv_cov_optim_true', 'JSdiv_cov_init_true', 'mean_err_optim_obs', 'mean_err_init_obs', 'cov_err_optim_obs',
             'cov_err_init_obs', 'JSdiv_cov_optim_obs', 'JSdiv_cov_init_obs']
    for nm in names:
        out['attr_' + nm] = np.asarray(getattr(fit, nm))
    out['attr_names'] = np.array(names)
    np.savez_compressed(os.path.join(HERE, 'c1_diag.npz'), **out)
    print('c1_diag.npz written:', len(names), 'attributes; paramSeq length', len(fit.paramSeq))


if __name__ == '__main__':
    main()
