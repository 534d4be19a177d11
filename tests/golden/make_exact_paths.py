#!/usr/bin/env python3
"""Golden EM paths of the ORACLE in exact mode (polished Newton E-step, same scipy M-step drivers).

The reference's own EM path (c1_em_batch.npz) carries its early-stopping slack (scipy Newton-CG
xtol, TNC f-tolerance): it sits ~3.5e-3 in nPLL from the exactly-converged path after 4 iterations.
The HIP path converges its E-step tightly, so it is compared with THIS path at tight tolerance and
with the reference's path at the reference's slack.  Needs only the oracle (no reference import).

    python tests/golden/make_exact_paths.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pgpfa_oracle as orc  # noqa: E402


def main():
    d = np.load(os.path.join(HERE, 'c1_dataset.npz'))
    Ys = [d['Y'][r].astype(float) for r in range(d['Y'].shape[0])]
    init = {'C': d['init_C'], 'd': d['init_d'], 'tau': d['init_tau']}
    fit = orc.fit_batch(Ys, init, float(d['binSize']), 5, 'TNC', mode='exact')
    np.random.seed(1)
    on = orc.fit_online_diag(Ys, init, float(d['binSize']), 4, 5, 'TNC', 'TNC', mode='exact')
    np.savez_compressed(
        os.path.join(HERE, 'c1_em_exact.npz'),
        nll=np.asarray(fit['nll']), seq_C=np.stack([s['C'] for s in fit['paramSeq']]),
        seq_d=np.stack([s['d'] for s in fit['paramSeq']]), seq_tau=np.stack([s['tau'] for s in fit['paramSeq']]),
        online_nll=np.asarray(on['nll']), online_seq_C=np.stack([s['C'] for s in on['paramSeq']]),
        online_seq_d=np.stack([s['d'] for s in on['paramSeq']]), online_seq_tau=np.stack([s['tau'] for s in on['paramSeq']]))
    print('written c1_em_exact.npz')


if __name__ == '__main__':
    main()
