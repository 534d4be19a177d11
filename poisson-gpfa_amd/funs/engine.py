"""EM orchestration with the reference's constructor surface (funs/engine.py:107-128 of
mackelab/poisson-gpfa): `PPGPFAfit(experiment, initParams, ...)` runs batch or online (stochastic) EM,
E-step and M-step evaluations on the GPU, and leaves the reference's result attributes
(engine.py:453-474) on the object.  Plotting methods are not part of the hot path."""
import copy
import json
import os
import time

import numpy as np

from . import inference
from . import learning
from . import util


def _banner(rows):
    print('+-------------------- Fit Options --------------------+')
    for label, value in rows:
        print('| ' + label + (str(value) + ' |').rjust(53 - len(label)))
    print('+-----------------------------------------------------+')


def _copy_params(params):
    return {'C': np.array(params['C'], dtype=np.float64), 'd': np.array(params['d'], dtype=np.float64).reshape(-1),
            'tau': np.array(params['tau'], dtype=np.float64).reshape(-1)}


class PPGPFAfit():
    """Poisson-GPFA fit of binned population spike counts by (online) EM.

    Arguments and defaults are those of the reference (engine.py:107-128).  Result attributes:
    optimParams, paramSeq, infRes (of the last batch processed), posteriorLikelihood,
    variationalLowerBound, learningDetails, inferenceTime, learningTime, tauSeq, ... (engine.py:453-474).
    """

    def __init__(self, experiment, initParams=None, xdim=2, inferenceMethod='laplace', maxEMiter=50, optimLogLamb=False,
                 CdOptimMethod='TNC', tauOptimMethod='TNC', verbose=False, EMmode='Online', batchSize=5,
                 onlineParamUpdateMethod='diag', hessTol=None, stepPow=0.75, updateCdJointly=True, fullyUpdateTau=False,
                 extractAllTraj=False, extractAllTraj_trueParams=False, getPredictionErr=False, CdMaxIter=None,
                 tauMaxIter=None, *, quiet=False, onlineWarmStart=True):
        if EMmode not in ('Batch', 'Online'):
            raise ValueError("EMmode must be 'Batch' or 'Online'")
        if inferenceMethod not in ('laplace', 'variational'):
            raise ValueError("inferenceMethod must be 'laplace' or 'variational'")
        self.experiment = experiment
        ydim, T = np.shape(experiment.data[0]['Y'])
        numTrials = len(experiment.data)
        if initParams is None:
            initParams = util.initializeParams(xdim, ydim, experiment)
        else:
            xdim = np.shape(initParams['C'])[1]

        posteriorLikelihood, variationalLowerBound, learningDetails = [], [], []
        params = initParams
        paramSeq = [initParams]
        learningTime, inferenceTime = [], []
        rows = [('Dimensionality of Latent State: ', xdim), ('Dimensionality of Observed State (# neurons): ', ydim),
                ('EM mode: ', EMmode), ('Max EM iterations: ', maxEMiter), ('Inference Method: ', inferenceMethod)]

        def e_step(exp_, params_, prev):
            if inferenceMethod == 'laplace':
                infRes_, nll_, opt_ = inference.laplace(experiment=exp_, params=params_, prevOptimRes=prev, verbose=verbose)
                return infRes_, nll_, None, opt_
            infRes_, nll_, vlb_, opt_ = inference.dualVariational(experiment=exp_, params=params_, optimizeLogLambda=optimLogLamb,
                                                                 prevOptimRes=prev, verbose=verbose)
            return infRes_, nll_, vlb_, opt_

        log_path = os.environ.get('PGPFA_LOG_JSONL')          # one JSON line per EM iteration (metrics / logging)

        def report(i, nll, vlb):
            if log_path:
                rec = {'iteration': i + 1, 'of': maxEMiter, 'em_mode': EMmode, 'inference': inferenceMethod, 'nPLL': float(nll),
                       'VLB': None if vlb is None else float(vlb), 'estep_s': inferenceTime[-1], 'mstep_s': learningTime[-1]}
                sess = getattr(infRes, 'session', None)
                if sess is not None:
                    for key in ('last_pcg_iterations', 'last_newton_factorizations', 'last_newton_max_iter', 'last_dense_retries',
                                'lowrank_rtot', 'chunk_trials', 'plan_lowrank', 'last_dual_evaluations'):
                        try:
                            rec[key] = sess.ctx.info(key)
                        except Exception:
                            pass
                with open(log_path, 'a') as fh:
                    fh.write(json.dumps(rec) + '\n')
            if quiet:
                return
            if vlb is None:
                util.Printer('Iteration: %3d of %3d, nPLL: = %.4f' % (i + 1, maxEMiter, nll))
            else:
                util.Printer('Iteration: %3d of %3d, nPLL: = %.4f, VLB = %.4f' % (i + 1, maxEMiter, nll, vlb))

        infRes = None
        if EMmode == 'Batch':                                   # reference engine.py:154-240
            if not quiet:
                _banner(rows)
            optimRes = None
            for i in range(maxEMiter):
                before = time.time()
                infRes, nll, vlb, optimRes = e_step(experiment, params, optimRes)      # warm start after iteration 0
                posteriorLikelihood.append(nll)
                if vlb is not None:
                    variationalLowerBound.append(vlb)
                inferenceTime.append(time.time() - before)
                before = time.time()
                params, learnDet = learning.updateParams(oldParams=params, infRes=infRes, experiment=experiment,
                                                         CdOptimMethod=CdOptimMethod)
                learningTime.append(time.time() - before)
                learningDetails.append(learnDet)
                paramSeq.append(params)
                report(i, nll, vlb)

        if EMmode == 'Online':                                  # reference engine.py:243-449
            if not quiet:
                _banner(rows + [('Online Param Update Method: ', '`' + str(onlineParamUpdateMethod) + '`'),
                                ('Batch size (trials): ', batchSize)])
            gamma = np.linspace(0, 1, maxEMiter)
            step_cd = 1 / (np.arange(maxEMiter) + 1) ** stepPow
            step_tau = 1 / (np.arange(maxEMiter) + 1) ** stepPow
            size = xdim * ydim + ydim if updateCdJointly else xdim * ydim
            self.invPriorCovs = [np.diag(np.ones(size))]
            self.cumHess = [np.diag(np.ones(size))]
            for n in range(maxEMiter):
                sub = util.subsampleTrials(experiment, batchSize)
                before = time.time()
                # the reference starts every minibatch trial from zero (engine.py:298-301); with onlineWarmStart the
                # trials a previous minibatch already visited start from the mode it left on the device (same fixed point)
                prev = 'resident' if (onlineWarmStart and inferenceMethod == 'laplace') else None
                infRes, nll, vlb, _ = e_step(sub, params, prev)
                posteriorLikelihood.append(nll)
                if vlb is not None:
                    variationalLowerBound.append(vlb)
                inferenceTime.append(time.time() - before)
                before = time.time()
                if onlineParamUpdateMethod in ('balancingGamma', 'sequentialAverage', 'fullyUpdateAll'):
                    newParams, learnDet = learning.updateParams(oldParams=params, infRes=infRes, experiment=sub,
                                                                CdOptimMethod=CdOptimMethod, CdMaxIter=CdMaxIter,
                                                                tauMaxIter=None, verbose=verbose)
                    nextParams = newParams                                           # aliasing as in engine.py:325-341
                    if onlineParamUpdateMethod == 'balancingGamma':
                        for key in ('C', 'd', 'tau'):
                            nextParams[key] = gamma[n] * params[key] + (1 - gamma[n]) * newParams[key]
                    elif onlineParamUpdateMethod == 'sequentialAverage':
                        for key in ('C', 'd', 'tau'):
                            nextParams[key] = (params[key] + newParams[key]) / 2
                elif onlineParamUpdateMethod == 'diag':
                    newParams, learnDet, priorCov = learning.updateParamsWithPrior(
                        oldParams=params, infRes=infRes, experiment=sub, CdOptimMethod=CdOptimMethod,
                        tauOptimMethod=tauOptimMethod, regularizer_stepsize_Cd=step_cd[n],
                        regularizer_stepsize_tau=step_tau[n], prevInvPriorCov=self.invPriorCovs[-1], covOpts='useDiag',
                        verbose=verbose, updateCdJointly=updateCdJointly, hessTol=hessTol)
                    nextParams = newParams
                    self.invPriorCovs.append(priorCov)
                elif onlineParamUpdateMethod == 'hess':                     # reference engine.py:354-368
                    newParams, learnDet, priorCov = learning.updateParamsWithPrior(
                        oldParams=params, infRes=infRes, experiment=sub, CdOptimMethod=CdOptimMethod,
                        tauOptimMethod=tauOptimMethod, regularizer_stepsize_Cd=step_cd[n],
                        regularizer_stepsize_tau=step_tau[n], prevInvPriorCov=self.invPriorCovs[-1], covOpts='useHessian',
                        verbose=verbose, updateCdJointly=updateCdJointly, hessTol=hessTol)
                    nextParams = newParams
                    self.invPriorCovs.append(priorCov)
                elif onlineParamUpdateMethod == 'grad':                     # reference engine.py:384-397
                    newParams, learnDet, hess = learning.updateParamsWithGradDescent(
                        oldParams=params, infRes=infRes, experiment=sub, stepSize=step_cd[n], cumHess=self.cumHess[-1],
                        regularizer_stepsize_tau=step_tau[n], tauOptimMethod=tauOptimMethod, verbose=verbose,
                        updateCdJointly=updateCdJointly, hessTol=hessTol)
                    self.cumHess.append(self.cumHess[-1] + hess)
                    nextParams = newParams
                else:
                    raise ValueError("unknown onlineParamUpdateMethod '%s'" % onlineParamUpdateMethod)
                learningTime.append(time.time() - before)
                if fullyUpdateTau:
                    nextParams['tau'] = newParams['tau']
                report(n, nll, vlb)
                learningDetails.append(learnDet)
                params = nextParams
                paramSeq.append(params)
            self.onlineParamUpdateMethod = onlineParamUpdateMethod

        self.xdim, self.ydim, self.T = xdim, ydim, T
        self.trialDur, self.binSize, self.numTrials = experiment.trialDur, experiment.binSize, numTrials
        self.maxEMiter, self.EMmode, self.inferenceMethod = maxEMiter, EMmode, inferenceMethod
        self.initParams, self.paramSeq, self.optimParams = initParams, paramSeq, params
        self.posteriorLikelihood, self.variationalLowerBound = posteriorLikelihood, variationalLowerBound
        self.learningDetails, self.infRes = learningDetails, infRes
        self.learningTime, self.inferenceTime = np.asarray(learningTime), np.asarray(inferenceTime)
        self.CdOptimMethod, self.optimLogLamb = CdOptimMethod, optimLogLamb
        self.processParamResults()
        self.performSpikeCountAnalysis()
        if not quiet:
            print()
        if extractAllTraj:
            self.extractTrajectories(method=inferenceMethod)
        if extractAllTraj_trueParams:
            self.extractTrajWithTrueParams(method=inferenceMethod)
        if getPredictionErr:
            self.leaveOneOutPrediction()

    # -- summaries of the parameter path (reference engine.py:541-597) -------------------------------------------
    def processParamResults(self):
        n = self.maxEMiter
        seq = self.paramSeq
        self.tauSeq = np.zeros([self.xdim, n])
        self.expectedSpikeCountsEst = np.zeros([self.ydim, n])
        self.CabsoluteValue = np.zeros(n)
        for i in range(n):
            C = np.asarray(seq[i]['C'], dtype=np.float64)
            self.tauSeq[:, i] = np.asarray(seq[i]['tau']).reshape(-1)
            self.expectedSpikeCountsEst[:, i] = self.T * np.exp(0.5 * np.sum(C * C, axis=1) + np.asarray(seq[i]['d']).reshape(-1))
            self.CabsoluteValue[i] = float(np.sum(C ** 2))
        self.expectedSpikeCountsEstVar = np.var(self.expectedSpikeCountsEst, axis=0)
        # per-neuron total counts / number of trials: from the device's integer sums, not from a host raster
        _, _, totals, _ = util.countMoments(self.experiment, self.xdim)
        self._count_totals = totals
        self.sampleMeanSpikeCounts = totals / self.numTrials
        self.sampleMeanSpikeCountsVar = np.var(self.sampleMeanSpikeCounts)
        if hasattr(self.experiment, 'params'):
            Ct = np.asarray(self.experiment.params['C'], dtype=np.float64)
            self.expectedSpikeCountsTrue = self.T * np.exp(0.5 * np.sum(Ct * Ct, axis=1) + np.asarray(self.experiment.params['d']).reshape(-1))
            self.expectedSpikeCountsTrueVar = np.var(self.expectedSpikeCountsTrue)
            self.varESpkCountTrue_Ratios = self.expectedSpikeCountsEstVar / self.expectedSpikeCountsTrueVar
            self.subspaceAngleC = [util.subspaceAngle(self.experiment.params['C'], seq[i]['C']) for i in range(n)]
        self.varESpkCountSampleMean_Ratios = self.expectedSpikeCountsEstVar / self.sampleMeanSpikeCountsVar
        dev = self.expectedSpikeCountsEst - self.sampleMeanSpikeCounts[:, None]
        self.meanSquaredErrorOverTrueVariance_SM = list(np.sum(dev * dev, axis=0) / self.numTrials / self.sampleMeanSpikeCountsVar)

    # -- spike-count moment diagnostics (reference engine.py:484-513) --------------------------------------------
    def performSpikeCountAnalysis(self):
        E_y_obs, E_yy_obs, _, _ = util.countMoments(self.experiment, self.xdim)
        self.E_y_obs, self.E_yy_obs = E_y_obs, E_yy_obs
        self.E_y_init_params, self.E_yy_init_params = util.getMeanCovYfromParams(self.initParams)
        self.E_y_optim_params, self.E_yy_optim_params = util.getMeanCovYfromParams(self.optimParams)
        norm_obs = np.linalg.norm(E_yy_obs)

        def scores(E_ref, E_yy_ref, tag):
            for which, E_y, E_yy in (('optim', self.E_y_optim_params, self.E_yy_optim_params),
                                     ('init', self.E_y_init_params, self.E_yy_init_params)):
                dm = E_ref - E_y
                setattr(self, 'mean_err_%s_%s' % (which, tag), float(dm @ dm / np.var(E_ref) / self.numTrials))
                setattr(self, 'cov_err_%s_%s' % (which, tag), float(np.linalg.norm(E_yy_ref - E_yy) / norm_obs))
                setattr(self, 'JSdiv_cov_%s_%s' % (which, tag), float(util.JSLogdetDiv(E_yy, E_yy_ref)))
        if hasattr(self.experiment, 'params'):
            self.E_y_true_params, self.E_yy_true_params = util.getMeanCovYfromParams(self.experiment.params)
            scores(self.E_y_true_params, self.E_yy_true_params, 'true')
        scores(E_y_obs, E_yy_obs, 'obs')

    def orthonormalizeTrajectories(self):
        """reference engine.py:515-521: x_tilde[trial] = diag(D) V.T post_mean[trial] with U, D, V = scipy.linalg.svd(C).
        (scipy returns V^T in that slot, so this is diag(D) times the matrix of right singular vectors, not its
        transpose; the same routine is called here because the result depends on its sign conventions.)"""
        import scipy.linalg
        _, D, Vh = scipy.linalg.svd(np.asarray(self.optimParams['C'], dtype=np.float64))
        mix = np.diag(D) @ Vh.T
        self.x_tilde = np.asarray([mix @ np.asarray(self.infRes['post_mean'][tr]) for tr in range(self.numTrials)])

    def leaveOneOutPrediction(self):
        """reference engine.py:599-644: y_pred_mode[numTrials][ydim][T] and pred_err_mode with the fitted parameters."""
        self._keep_inf_res()                # (the held-out searches re-plan the chunk workspace; per-trial state stays)
        self.y_pred_mode, self.pred_err_mode = util.leaveOneOutPrediction(self.optimParams, self.experiment)

    def extractTrajectories(self, method='laplace'):
        """One more E-step over all trials with the fitted parameters (reference engine.py:523-532)."""
        self._keep_inf_res()
        if method == 'laplace':
            self.infRes, self.nll_all_traj, _ = inference.laplace(self.experiment, copy.copy(self.optimParams))
        else:
            self.infRes, self.nll_all_traj, self.vlb_all_traj, _ = inference.dualVariational(
                self.experiment, copy.copy(self.optimParams), optimizeLogLambda=self.optimLogLamb)

    def _keep_inf_res(self, limit_bytes=4 << 30):
        """self.infRes is a view of device state that the next E-step on the same trials overwrites; the reference's is a
        list of host arrays that stays.  Before such an E-step, copy post_mean / post_vsm to the host (up to limit_bytes;
        beyond that the entries stay lazy and reading a superseded one raises)."""
        res = getattr(self, 'infRes', None)
        if res is not None and hasattr(res, 'materialize') and res.host_bytes() <= limit_bytes:
            try:
                res.materialize()
            except Exception:                       # already superseded: the lazy entries will say so when read
                pass

    def extractTrajWithTrueParams(self, method='laplace'):
        self._keep_inf_res()
        if method == 'laplace':
            self.infRes_trueParams, self.nll_trueParams_all_traj, _ = inference.laplace(self.experiment, copy.copy(self.experiment.params))
        else:
            (self.infRes_trueParams, self.nll_trueParams_all_traj, self.vlb_trueParams_all_traj, _) = inference.dualVariational(
                self.experiment, copy.copy(self.experiment.params), optimizeLogLambda=self.optimLogLamb)

