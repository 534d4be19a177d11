"""ctypes binding of libpgpfa_hip.so (C-ABI declared in include/pgpfa.h).

The product path has NO CPU fallback: if the shared library is missing, or no MI355X is
visible, every compute entry point raises.  Build the library with
``python -c "import __graft_entry__ as g; g.build()"`` (hipcc --offload-arch=gfx950).
"""
import ctypes as ct
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), 'libpgpfa_hip.so')

c_double_p = ct.POINTER(ct.c_double)
c_int32_p = ct.POINTER(ct.c_int32)
c_uint8_p = ct.POINTER(ct.c_uint8)
c_int64_p = ct.POINTER(ct.c_int64)

# name -> (argtypes); every function returns int except where noted
_SIGNATURES = {
    'pgpfa_version': [],
    'pgpfa_device_count': [ct.POINTER(ct.c_int)],
    'pgpfa_create': [ct.POINTER(ct.c_void_p), ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_int, ct.c_double],
    'pgpfa_destroy': [ct.c_void_p],
    'pgpfa_set_option': [ct.c_void_p, ct.c_char_p, ct.c_double],
    'pgpfa_get_info': [ct.c_void_p, ct.c_char_p, c_double_p],
    'pgpfa_upload_counts_f64': [ct.c_void_p, c_double_p],
    'pgpfa_upload_counts_u8': [ct.c_void_p, c_uint8_p],
    'pgpfa_upload_counts_u16': [ct.c_void_p, ct.POINTER(ct.c_uint16)],
    'pgpfa_get_counts_u16': [ct.c_void_p, ct.c_int, c_int32_p, ct.POINTER(ct.c_uint16)],
    'pgpfa_set_params': [ct.c_void_p, c_double_p, c_double_p, c_double_p],
    'pgpfa_get_gram': [ct.c_void_p, c_double_p],
    'pgpfa_get_gram_inverse': [ct.c_void_p, c_double_p],
    'pgpfa_laplace_eval': [ct.c_void_p, ct.c_int, c_int32_p, c_double_p, c_double_p, c_double_p],
    'pgpfa_laplace_hessian': [ct.c_void_p, ct.c_int, c_double_p, c_double_p],
    'pgpfa_estep_laplace': [ct.c_void_p, ct.c_int, c_int32_p, ct.c_int, c_double_p, c_int32_p, c_int32_p],
    'pgpfa_set_modes': [ct.c_void_p, ct.c_int, c_int32_p, c_double_p],
    'pgpfa_get_post_mean': [ct.c_void_p, ct.c_int, c_int32_p, c_double_p],
    'pgpfa_get_post_vsm': [ct.c_void_p, ct.c_int, c_int32_p, c_double_p],
    'pgpfa_get_post_vsmgp': [ct.c_void_p, ct.c_int, c_int32_p, c_double_p],
    'pgpfa_get_post_cov': [ct.c_void_p, ct.c_int, c_double_p],
    'pgpfa_set_posterior': [ct.c_void_p, ct.c_int, c_int32_p, c_double_p, c_double_p, c_double_p],
    'pgpfa_mstep_cd_costgrad': [ct.c_void_p, c_double_p, c_double_p, ct.c_double, c_double_p, c_double_p],
    'pgpfa_mstep_cd_newton_pass': [ct.c_void_p, c_double_p, c_double_p, ct.c_double, c_double_p, c_double_p, c_double_p],
    'pgpfa_mstep_cd_chord_pass': [ct.c_void_p, c_double_p, c_double_p, ct.c_double, c_double_p, c_double_p, c_double_p],
    'pgpfa_mstep_cd_cost_per_neuron': [ct.c_void_p, c_double_p, c_double_p, ct.c_double, c_double_p],
    'pgpfa_mstep_precomp': [ct.c_void_p, c_double_p],
    'pgpfa_get_pautosum': [ct.c_void_p, c_double_p],
    'pgpfa_mstep_tau_costgrad': [ct.c_void_p, ct.c_int, ct.c_double, c_double_p, c_double_p],
    'pgpfa_mstep_tau_costgrad_batch': [ct.c_void_p, c_double_p, c_double_p, c_double_p],
    'pgpfa_mstep_tau_costgrad_multi': [ct.c_void_p, ct.c_int, c_double_p, c_double_p, c_double_p],
    'pgpfa_mstep_tau_costgrad_multi_begin': [ct.c_void_p, ct.c_int, c_double_p],
    'pgpfa_mstep_tau_costgrad_multi_end': [ct.c_void_p, c_double_p, c_double_p],
    'pgpfa_generate': [ct.c_void_p, ct.c_ulonglong, ct.c_int, c_int32_p, c_double_p, c_uint8_p],
    'pgpfa_loo_predict': [ct.c_void_p, ct.c_int, c_int32_p, c_double_p, c_double_p],
    'pgpfa_count_moments': [ct.c_void_p, ct.c_int, c_int32_p, c_int64_p, c_int64_p, c_int64_p],
    'pgpfa_dual_costgrad': [ct.c_void_p, ct.c_int, c_double_p, c_double_p, c_double_p],
    'pgpfa_dual_costgrad_batch': [ct.c_void_p, ct.c_int, c_int32_p, c_double_p, c_double_p, c_double_p],
    'pgpfa_dual_lbfgs': [ct.c_void_p, ct.c_int, c_int32_p, c_double_p, ct.c_int, ct.c_double, ct.c_double, c_double_p, c_int32_p],
    'pgpfa_get_dual_lambda': [ct.c_void_p, ct.c_int, c_int32_p, c_double_p],
    'pgpfa_dual_fixed_point': [ct.c_void_p, ct.c_int, c_int32_p, c_double_p, ct.c_int, ct.c_int, ct.c_double, c_double_p, c_int32_p, c_int32_p, c_double_p],
    'pgpfa_dual_finalize': [ct.c_void_p, ct.c_int, c_int32_p, c_double_p, c_double_p],
    'pgpfa_dual_post_mean': [ct.c_void_p, ct.c_int, c_double_p, c_double_p],
    'pgpfa_dual_post_cov': [ct.c_void_p, ct.c_int, c_double_p, c_double_p, c_double_p],
    'pgpfa_comm_unique_id': [ct.c_char_p],
    'pgpfa_comm_init': [ct.c_void_p, ct.c_char_p, ct.c_int, ct.c_int],
    'pgpfa_comm_allreduce_host': [ct.c_void_p, c_double_p, ct.c_int],
    'pgpfa_comm_describe': [ct.c_void_p, ct.c_char_p, ct.c_int],
    'pgpfa_test_potrf': [ct.c_void_p, ct.c_int, ct.c_int, c_double_p, c_double_p, c_double_p],
    'pgpfa_test_gemm_nt': [ct.c_void_p, ct.c_int, ct.c_int, ct.c_int, ct.c_double, c_double_p, c_double_p, ct.c_double, c_double_p],
    'pgpfa_test_gemm_nn': [ct.c_void_p, ct.c_int, ct.c_int, ct.c_int, ct.c_double, c_double_p, c_double_p, ct.c_double, c_double_p],
    'pgpfa_test_gemm_nt_f32': [ct.c_void_p, ct.c_int, ct.c_int, ct.c_int, ct.c_double, c_double_p, c_double_p, ct.c_double, c_double_p],
    'pgpfa_test_gemm_nn_f32': [ct.c_void_p, ct.c_int, ct.c_int, ct.c_int, ct.c_double, c_double_p, c_double_p, ct.c_double, c_double_p],
    'pgpfa_bench_syrk': [ct.c_void_p, ct.c_int, ct.c_int, ct.c_int, ct.c_int, c_double_p, c_double_p],
    'pgpfa_bench_potrf_diag': [ct.c_void_p, ct.c_int, ct.c_int, ct.c_int, c_double_p],
    'pgpfa_gemm_shape_report': [ct.c_void_p, ct.c_char_p, ct.c_int],
    'pgpfa_bench_mfma_peak': [ct.c_void_p, ct.c_int, c_double_p],
}
EXPORTED_SYMBOLS = sorted(list(_SIGNATURES) + ['pgpfa_last_error'])

_lib = None


class HipBackendError(RuntimeError):
    """Raised when the HIP library is missing, no GPU is visible, or a C-ABI call fails."""


def load_library():
    """Load libpgpfa_hip.so and attach prototypes.  Works without a GPU (symbols only)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipBackendError(
            'libpgpfa_hip.so not found at %s - build it with __graft_entry__.build(); '
            'this package has no CPU fallback' % LIB_PATH)
    lib = ct.CDLL(LIB_PATH)
    for name, argtypes in _SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = ct.c_int
    lib.pgpfa_last_error.argtypes = []
    lib.pgpfa_last_error.restype = ct.c_char_p
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise HipBackendError(load_library().pgpfa_last_error().decode('utf-8', 'replace'))


def device_count():
    n = ct.c_int(0)
    rc = load_library().pgpfa_device_count(ct.byref(n))
    return n.value if rc == 0 else 0


def dptr(a):
    return a.ctypes.data_as(c_double_p)


def iptr(a):
    return None if a is None else a.ctypes.data_as(c_int32_p)


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def as_idx(idx):
    if idx is None:
        return None
    return np.ascontiguousarray(idx, dtype=np.int32)


class Context:
    """Thin RAII wrapper over pgpfa_ctx."""

    def __init__(self, q, p, T, R, bin_ms, device=0):
        lib = load_library()
        if device_count() < 1:
            raise HipBackendError('no HIP device visible: the Poisson-GPFA hot path runs on MI355X only '
                                  '(there is no CPU fallback)')
        self.lib = lib
        self.q, self.p, self.T, self.R = int(q), int(p), int(T), int(R)
        self.n = self.p * self.T
        h = ct.c_void_p()
        check(lib.pgpfa_create(ct.byref(h), int(device), self.q, self.p, self.T, self.R, float(bin_ms)))
        self.h = h

    def close(self):
        if getattr(self, 'h', None):
            self.lib.pgpfa_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- options / info --------------------------------------------------------------
    def set_option(self, key, value):
        check(self.lib.pgpfa_set_option(self.h, key.encode(), float(value)))

    def info(self, key):
        v = ct.c_double(0.0)
        check(self.lib.pgpfa_get_info(self.h, key.encode(), ct.byref(v)))
        return v.value

    # -- data ---------------------------------------------------------------------------
    def upload_counts(self, Y):
        Y = np.asarray(Y)
        if Y.shape != (self.R, self.q, self.T):
            raise ValueError('counts must have shape (R,q,T)=%s, got %s' % ((self.R, self.q, self.T), Y.shape))
        if Y.dtype == np.uint8:
            Y = np.ascontiguousarray(Y)
            check(self.lib.pgpfa_upload_counts_u8(self.h, Y.ctypes.data_as(c_uint8_p)))
        elif Y.dtype == np.uint16:
            Y = np.ascontiguousarray(Y)
            check(self.lib.pgpfa_upload_counts_u16(self.h, Y.ctypes.data_as(ct.POINTER(ct.c_uint16))))
        else:
            Y = as_f64(Y)
            check(self.lib.pgpfa_upload_counts_f64(self.h, dptr(Y)))

    def set_params(self, C, d, tau):
        C, d, tau = as_f64(C), as_f64(d).reshape(-1), as_f64(tau).reshape(-1)
        if C.shape != (self.q, self.p) or d.shape != (self.q,) or tau.shape != (self.p,):
            raise ValueError('parameter shapes do not match the context (q=%d, p=%d)' % (self.q, self.p))
        check(self.lib.pgpfa_set_params(self.h, dptr(C), dptr(d), dptr(tau)))

    def gram(self):
        K = np.empty((self.p, self.T, self.T))
        check(self.lib.pgpfa_get_gram(self.h, dptr(K)))
        return K

    def gram_inverse(self):
        K = np.empty((self.p, self.T, self.T))
        check(self.lib.pgpfa_get_gram_inverse(self.h, dptr(K)))
        return K

    # -- Laplace ---------------------------------------------------------------------------
    def _n_idx(self, idx):
        return (self.R, None) if idx is None else (len(idx), as_idx(idx))

    def laplace_eval(self, idx, X, want_grad=True):
        n, ii = self._n_idx(idx)
        X = as_f64(X).reshape(n, self.n)
        f = np.empty(n)
        g = np.empty((n, self.p, self.T)) if want_grad else None
        check(self.lib.pgpfa_laplace_eval(self.h, n, iptr(ii), dptr(X), dptr(f), dptr(g) if want_grad else None))
        return f, g

    def laplace_hessian(self, trial, X):
        X = as_f64(X).reshape(self.n)
        H = np.empty((self.n, self.n))
        check(self.lib.pgpfa_laplace_hessian(self.h, int(trial), dptr(X), dptr(H)))
        return H

    def estep_laplace(self, idx=None, warm_start=False):
        n, ii = self._n_idx(idx)
        obj = ct.c_double(0.0)
        iters = np.zeros(n, dtype=np.int32)
        status = np.zeros(n, dtype=np.int32)
        warm = 2 if warm_start == 'resident' else (1 if warm_start else 0)
        check(self.lib.pgpfa_estep_laplace(self.h, n, iptr(ii), warm, ct.byref(obj), iptr(iters), iptr(status)))
        return obj.value, iters, status

    def count_moments(self, idx=None):
        """Exact integer moments of the resident counts over the listed trials:
        (sum[q], cross[q][q], n_samples) with sum_i = sum y_i, cross_ij = sum y_i y_j over all (trial, bin) samples."""
        n, ii = self._n_idx(idx)
        s = np.zeros(self.q, dtype=np.int64)
        S = np.zeros((self.q, self.q), dtype=np.int64)
        ns = ct.c_int64(0)
        check(self.lib.pgpfa_count_moments(self.h, n, iptr(ii), s.ctypes.data_as(c_int64_p), S.ctypes.data_as(c_int64_p), ct.byref(ns)))
        return s, S, int(ns.value)

    def generate(self, seed, idx=None, want_x=True, want_y=True):
        """Sample latents and counts of the listed trials on the device (they replace the resident counts) -> (X, Y) copies."""
        n, ii = self._n_idx(idx)
        X = np.empty((n, self.p, self.T)) if want_x else None
        check(self.lib.pgpfa_generate(self.h, int(seed), n, iptr(ii), dptr(X) if want_x else None, None))
        Y = self.counts(idx) if want_y else None
        return X, Y

    def counts(self, idx=None):
        """Resident counts of the listed trials: uint8 [n][q][T], or uint16 when some resident count exceeds 255."""
        n, ii = self._n_idx(idx)
        Y = np.empty((n, self.q, self.T), dtype=np.uint16)
        check(self.lib.pgpfa_get_counts_u16(self.h, n, iptr(ii), Y.ctypes.data_as(ct.POINTER(ct.c_uint16))))
        return Y if self.info('counts_two_bytes') else Y.astype(np.uint8)

    def loo_predict(self, idx=None):
        """Leave-one-neuron-out prediction for the listed trials -> (y_pred[n][q][T], summed squared error)."""
        n, ii = self._n_idx(idx)
        out = np.empty((n, self.q, self.T))
        err = ct.c_double(0.0)
        check(self.lib.pgpfa_loo_predict(self.h, n, iptr(ii), dptr(out), ct.byref(err)))
        return out, err.value

    def set_modes(self, idx, X):
        n, ii = self._n_idx(idx)
        X = as_f64(X).reshape(n, self.n)
        check(self.lib.pgpfa_set_modes(self.h, n, iptr(ii), dptr(X)))

    def post_mean(self, idx=None):
        n, ii = self._n_idx(idx)
        out = np.empty((n, self.p, self.T))
        check(self.lib.pgpfa_get_post_mean(self.h, n, iptr(ii), dptr(out)))
        return out

    def post_vsm(self, idx=None):
        n, ii = self._n_idx(idx)
        out = np.empty((n, self.T, self.p, self.p))
        check(self.lib.pgpfa_get_post_vsm(self.h, n, iptr(ii), dptr(out)))
        return out

    def post_vsmgp(self, idx=None):
        n, ii = self._n_idx(idx)
        out = np.empty((n, self.T, self.T, self.p))
        check(self.lib.pgpfa_get_post_vsmgp(self.h, n, iptr(ii), dptr(out)))
        return out

    def post_cov(self, trial):
        out = np.empty((self.n, self.n))
        check(self.lib.pgpfa_get_post_cov(self.h, int(trial), dptr(out)))
        return out

    def set_posterior(self, idx, post_mean, post_vsm, post_vsmgp=None):
        n, ii = self._n_idx(idx)
        m = as_f64(post_mean).reshape(n, self.n)
        v = as_f64(post_vsm).reshape(n, self.T * self.p * self.p)
        g = None if post_vsmgp is None else as_f64(post_vsmgp).reshape(n, self.T * self.T * self.p)
        check(self.lib.pgpfa_set_posterior(self.h, n, iptr(ii), dptr(m), dptr(v), None if g is None else dptr(g)))

    # -- dual variational ---------------------------------------------------------------------
    def dual_costgrad(self, trial, lam, want_grad=True):
        lam = as_f64(lam).reshape(-1)
        cost = ct.c_double(0.0)
        grad = np.empty(self.q * self.T) if want_grad else None
        check(self.lib.pgpfa_dual_costgrad(self.h, int(trial), dptr(lam), ct.byref(cost), dptr(grad) if want_grad else None))
        return cost.value, grad

    def dual_post_mean(self, trial, lam):
        lam = as_f64(lam).reshape(-1)
        out = np.empty(self.n)
        check(self.lib.pgpfa_dual_post_mean(self.h, int(trial), dptr(lam), dptr(out)))
        return out

    def dual_post_cov(self, trial, lam, want_prec=True):
        lam = as_f64(lam).reshape(-1)
        cov = np.empty((self.n, self.n))
        prec = np.empty((self.n, self.n)) if want_prec else None
        check(self.lib.pgpfa_dual_post_cov(self.h, int(trial), dptr(lam), dptr(cov), dptr(prec) if want_prec else None))
        return cov, prec

    def dual_costgrad_batch(self, idx, lam, want_grad=True):
        """Dual cost (and gradient) of the listed distinct trials, each at its own lambda: lam[n][q*T] -> cost[n], grad[n][q*T]."""
        n, ii = self._n_idx(idx)
        lam = as_f64(lam).reshape(n, self.q * self.T)
        cost = np.empty(n)
        grad = np.empty((n, self.q * self.T)) if want_grad else None
        check(self.lib.pgpfa_dual_costgrad_batch(self.h, n, iptr(ii), dptr(lam), dptr(cost), dptr(grad) if want_grad else None))
        return cost, grad

    def dual_lbfgs(self, idx, rho0, max_iter=15000, factr=1e7, pgtol=1e-5):
        """Device L-BFGS on the dual in rho = log(lambda) for the listed trials -> (rho_opt[n][q*T], dual optimum[n], iterations[n])."""
        n, ii = self._n_idx(idx)
        rho = np.array(as_f64(rho0).reshape(n, self.q * self.T), copy=True)
        fopt = np.empty(n)
        iters = np.zeros(n, dtype=np.int32)
        check(self.lib.pgpfa_dual_lbfgs(self.h, n, iptr(ii), dptr(rho), int(max_iter), float(factr), float(pgtol), dptr(fopt), iptr(iters)))
        return rho, fopt, iters

    def dual_fixed_point(self, idx, rho0=None, max_outer=40, tol=1e-8, warm=False, want_lam=False, want_rho=True, resident=False):
        """Optimum of the dual by the variance fixed point (pgpfa_dual_fixed_point) -> (rho_opt[n][q*T] or None, dual optimum[n], passes[n],
        status[n]: 0 converged, 1 pass cap, 2 not contracting[, lambda_opt[n][q*T] with want_lam]).  rho0 = None: the reference's cold start
        lambda = 0.5; warm: rho0 is a previous optimum; resident: the start is the optimum resident on the device (nothing uploaded).  The
        optimum stays on the device: dual_finalize(idx, None) takes it from there, dual_lambda(idx) reads it."""
        n, ii = self._n_idx(idx)
        m = self.q * self.T
        if resident:
            rho, start = (np.empty((n, m)) if want_rho else None), 3
        elif rho0 is None:
            rho, start = (np.empty((n, m)) if want_rho else None), 0
        else:
            rho, start = np.array(as_f64(rho0).reshape(n, m), copy=True), (2 if warm else 1)
        fopt = np.empty(n)
        outer = np.zeros(n, dtype=np.int32)
        status = np.zeros(n, dtype=np.int32)
        lam = np.empty((n, m)) if want_lam else None
        check(self.lib.pgpfa_dual_fixed_point(self.h, n, iptr(ii), dptr(rho) if rho is not None else None, start, int(max_outer), float(tol), dptr(fopt),
                                              iptr(outer), iptr(status), dptr(lam) if want_lam else None))
        return (rho, fopt, outer, status, lam) if want_lam else (rho, fopt, outer, status)

    def dual_lambda(self, idx=None):
        """The dual variables resident for the listed trials: [n][q*T]."""
        n, ii = self._n_idx(idx)
        out = np.empty((n, self.q * self.T))
        check(self.lib.pgpfa_get_dual_lambda(self.h, n, iptr(ii), dptr(out)))
        return out

    def dual_finalize(self, idx, lam):
        """lam = None: the optimum the last dual_fixed_point left on the device for these trials."""
        n, ii = self._n_idx(idx)
        tot = ct.c_double(0.0)
        if lam is None:
            check(self.lib.pgpfa_dual_finalize(self.h, n, iptr(ii), None, ct.byref(tot)))
            return tot.value
        lam = as_f64(lam).reshape(n, self.q * self.T)
        check(self.lib.pgpfa_dual_finalize(self.h, n, iptr(ii), dptr(lam), ct.byref(tot)))
        return tot.value

    # -- M-step --------------------------------------------------------------------------------
    def mstep_cd_costgrad(self, vec, prior_center=None, inv_s2=0.0):
        vec = as_f64(vec).reshape(-1)
        cost = ct.c_double(0.0)
        grad = np.empty(self.q * (self.p + 1))
        pc = None if prior_center is None else as_f64(prior_center).reshape(-1)
        check(self.lib.pgpfa_mstep_cd_costgrad(self.h, dptr(vec), None if pc is None else dptr(pc), float(inv_s2),
                                               ct.byref(cost), dptr(grad)))
        return cost.value, grad

    def mstep_cd_newton_pass(self, vec, prior_center=None, inv_s2=0.0):
        vec = as_f64(vec).reshape(-1)
        pc = None if prior_center is None else as_f64(prior_center).reshape(-1)
        cost_n, dec = np.empty(self.q), np.empty(self.q)
        delta = np.empty(self.q * (self.p + 1))
        check(self.lib.pgpfa_mstep_cd_newton_pass(self.h, dptr(vec), None if pc is None else dptr(pc), float(inv_s2),
                                                  dptr(cost_n), dptr(delta), dptr(dec)))
        return cost_n, delta, dec

    def mstep_cd_chord_pass(self, vec, prior_center=None, inv_s2=0.0):
        """cost_n, Newton-like step and decrement at vec with the Hessians of the last mstep_cd_newton_pass."""
        vec = as_f64(vec).reshape(-1)
        pc = None if prior_center is None else as_f64(prior_center).reshape(-1)
        cost_n, dec = np.empty(self.q), np.empty(self.q)
        delta = np.empty(self.q * (self.p + 1))
        check(self.lib.pgpfa_mstep_cd_chord_pass(self.h, dptr(vec), None if pc is None else dptr(pc), float(inv_s2),
                                                 dptr(cost_n), dptr(delta), dptr(dec)))
        return cost_n, delta, dec

    def mstep_cd_cost_per_neuron(self, vec, prior_center=None, inv_s2=0.0):
        vec = as_f64(vec).reshape(-1)
        pc = None if prior_center is None else as_f64(prior_center).reshape(-1)
        cost_n = np.empty(self.q)
        check(self.lib.pgpfa_mstep_cd_cost_per_neuron(self.h, dptr(vec), None if pc is None else dptr(pc), float(inv_s2), dptr(cost_n)))
        return cost_n

    def mstep_precomp(self):
        n = ct.c_double(0.0)
        check(self.lib.pgpfa_mstep_precomp(self.h, ct.byref(n)))
        return n.value

    def pautosum(self):
        out = np.empty((self.p, self.T, self.T))
        check(self.lib.pgpfa_get_pautosum(self.h, dptr(out)))
        return out

    def mstep_tau_costgrad(self, k, logp):
        cost, grad = ct.c_double(0.0), ct.c_double(0.0)
        check(self.lib.pgpfa_mstep_tau_costgrad(self.h, int(k), float(logp), ct.byref(cost), ct.byref(grad)))
        return cost.value, grad.value

    def mstep_tau_costgrad_batch(self, logp):
        logp = as_f64(logp).reshape(-1)
        cost, grad = np.empty(self.p), np.empty(self.p)
        check(self.lib.pgpfa_mstep_tau_costgrad_batch(self.h, dptr(logp), dptr(cost), dptr(grad)))
        return cost, grad

    def mstep_tau_costgrad_multi(self, logp):
        """logp[m][p] (m <= 4 candidate points per latent) -> cost[m][p], grad[m][p] in one batched pass."""
        logp = as_f64(logp).reshape(-1, self.p)
        m = logp.shape[0]
        cost, grad = np.empty((m, self.p)), np.empty((m, self.p))
        check(self.lib.pgpfa_mstep_tau_costgrad_multi(self.h, int(m), dptr(logp), dptr(cost), dptr(grad)))
        return cost, grad

    def mstep_tau_costgrad_multi_begin(self, logp):
        """Enqueue the batched pass on the side stream and return; collect with mstep_tau_costgrad_multi_end()."""
        logp = as_f64(logp).reshape(-1, self.p)
        check(self.lib.pgpfa_mstep_tau_costgrad_multi_begin(self.h, int(logp.shape[0]), dptr(logp)))
        self._tau_m = logp.shape[0]

    def mstep_tau_costgrad_multi_end(self):
        m = self._tau_m
        cost, grad = np.empty((m, self.p)), np.empty((m, self.p))
        check(self.lib.pgpfa_mstep_tau_costgrad_multi_end(self.h, dptr(cost), dptr(grad)))
        return cost, grad

    # -- comm --------------------------------------------------------------------------------------
    def comm_init(self, uid, rank, nranks):
        warm_rccl_image()
        check(self.lib.pgpfa_comm_init(self.h, uid, int(rank), int(nranks)))

    def comm_describe(self):
        """One line about this rank's communicator (device, PCI bus id, the rank count RCCL itself reports)."""
        buf = ct.create_string_buffer(256)
        check(self.lib.pgpfa_comm_describe(self.h, buf, 256))
        return buf.value.decode('utf-8', 'replace')

    def allreduce_host(self, arr):
        a = as_f64(arr).reshape(-1).copy()
        check(self.lib.pgpfa_comm_allreduce_host(self.h, dptr(a), a.size))
        return a.reshape(np.shape(arr))

    # -- test hooks ----------------------------------------------------------------------------------
    def test_potrf(self, A, want_inverse=True):
        A = as_f64(A)
        if A.ndim == 2:
            A = A[None]
        b, n, _ = A.shape
        L = np.empty_like(A)
        inv = np.empty_like(A) if want_inverse else None
        check(self.lib.pgpfa_test_potrf(self.h, b, n, dptr(A), dptr(L), dptr(inv) if want_inverse else None))
        return L, inv

    def test_gemm_nt(self, A, B, C=None, alpha=1.0, beta=0.0, f32=False):
        """C = alpha*A@B.T + beta*C with A (M,K), B (N,K) given row-major; passed column-major.  f32: single-precision MFMA kernel."""
        A, B = as_f64(A), as_f64(B)
        M, K = A.shape
        N = B.shape[0]
        Ccm = np.zeros((N, M)) if C is None else as_f64(np.asarray(C).T)     # column-major (M,N) == row-major (N,M)
        Acm, Bcm = as_f64(A.T), as_f64(B.T)                                   # column-major (M,K) == row-major (K,M)
        fn = self.lib.pgpfa_test_gemm_nt_f32 if f32 else self.lib.pgpfa_test_gemm_nt
        check(fn(self.h, M, N, K, float(alpha), dptr(Acm), dptr(Bcm), float(beta), dptr(Ccm)))
        return Ccm.T.copy()

    def test_gemm_nn(self, A, B, C=None, alpha=1.0, beta=0.0, f32=False):
        """C = alpha*A@B + beta*C with A (M,K), B (K,N) row-major in; B is passed K x N column-major."""
        A, B = as_f64(A), as_f64(B)
        M, K = A.shape
        N = B.shape[1]
        Ccm = np.zeros((N, M)) if C is None else as_f64(np.asarray(C).T)
        Acm, Bcm = as_f64(A.T), as_f64(B.T)                # (K,N) column-major == (N,K) row-major buffer
        fn = self.lib.pgpfa_test_gemm_nn_f32 if f32 else self.lib.pgpfa_test_gemm_nn
        check(fn(self.h, M, N, K, float(alpha), dptr(Acm), dptr(Bcm), float(beta), dptr(Ccm)))
        return Ccm.T.copy()

    def bench_mfma_peak(self, iters=20000):
        tf = ct.c_double(0.0)
        check(self.lib.pgpfa_bench_mfma_peak(self.h, int(iters), ct.byref(tf)))
        return tf.value

    def gemm_shape_report(self):
        """Text table of the GEMM launches timed since set_option('profile', 1 | 2), grouped by operand shape."""
        need = self.lib.pgpfa_gemm_shape_report(self.h, None, 0)
        if need < 0:
            check(1)
        buf = ct.create_string_buffer(max(need, 1))
        self.lib.pgpfa_gemm_shape_report(self.h, buf, need)
        return buf.value.decode()

    def bench_potrf_diag(self, batch, reps=50, phases=3):
        us = ct.c_double()
        check(self.lib.pgpfa_bench_potrf_diag(self.h, int(batch), int(reps), int(phases), ct.byref(us)))
        return us.value

    def bench_syrk(self, batch, n, k, reps):
        ms, fl = ct.c_double(0.0), ct.c_double(0.0)
        check(self.lib.pgpfa_bench_syrk(self.h, int(batch), int(n), int(k), int(reps), ct.byref(ms), ct.byref(fl)))
        return ms.value, fl.value


_rccl_warmed = False


def warm_rccl_image():
    """RCCL's shared object carries ~0.5 GB of device code that ncclCommInitRank loads through page faults; on a
    machine whose image has not been read yet that takes minutes.  One sequential read beforehand makes the first
    communicator come up in seconds (no-op cost when the file is already cached)."""
    global _rccl_warmed
    if _rccl_warmed:
        return
    _rccl_warmed = True
    try:
        path = None
        with open('/proc/self/maps') as fh:
            for line in fh:
                if 'librccl.so' in line:
                    path = line.split()[-1]
                    break
        if path is None:
            return
        with open(path, 'rb', buffering=0) as fh:
            while fh.read(16 << 20):
                pass
    except OSError:
        pass


def comm_unique_id():
    load_library()
    warm_rccl_image()
    buf = ct.create_string_buffer(128)
    check(load_library().pgpfa_comm_unique_id(buf))
    return buf.raw
