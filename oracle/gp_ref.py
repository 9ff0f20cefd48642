"""Oracle (TEST INFRASTRUCTURE): closed-form GP quantities by a route that shares nothing with the product's
composition (no Cholesky of an augmented matrix, no triangular solves): dense `slogdet` + `solve`, explicit
Nystrom matrices for the sparse bound.  Used to pin both the oracle engine and the HIP path at small n.

Formulas restated (the reference reaches them inside stheno):
  exact GP   log N(y; 0, S),  S = K + diag(noise) + eps I;  mean* = K_*x S^-1 y;  cov* = K_** - K_*x S^-1 K_x*
             (Rasmussen & Williams 2006, eq. 2.23-2.24, 2.30)
  VFE        Q = K_xz K_zz^-1 K_zx;  bound = log N(y; 0, Q + D) - 1/2 tr(D^-1 (K_xx - Q))     (Titsias 2009, eq. 9)
             posterior mean* = K_*z (K_zz + K_zx D^-1 K_xz)^-1 K_zx D^-1 y                     (Titsias 2009, eq. 10 + 6)
             posterior cov*  = K_** - K_*z K_zz^-1 K_z* + K_*z (K_zz + K_zx D^-1 K_xz)^-1 K_z*
"""
import numpy as np

from . import kernels as ok

__all__ = ["logpdf", "posterior", "vfe_bound", "vfe_posterior"]

_LOG_2PI = np.log(2.0 * np.pi)


def _mvn_logpdf(S, r):
    sign, logdet = np.linalg.slogdet(S)
    assert sign > 0
    return -0.5 * (logdet + len(r) * _LOG_2PI + r @ np.linalg.solve(S, r))


def logpdf(spec, x, y, noise, eps=1e-12):
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    S = ok.gram(spec, x, None, noise_diag=np.broadcast_to(noise, y.shape), jitter=eps)
    return _mvn_logpdf(S, y)


def posterior(spec, x, y, noise, xs, eps=1e-12):
    """(mean*, cov*) of the latent function at xs."""
    x, xs = np.asarray(x, dtype=np.float64), np.asarray(xs, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    S = ok.gram(spec, x, None, noise_diag=np.broadcast_to(noise, y.shape), jitter=eps)
    Ksx = ok.gram(spec, xs, x)
    mean = Ksx @ np.linalg.solve(S, y)
    cov = ok.gram(spec, xs) - Ksx @ np.linalg.solve(S, Ksx.T)
    return mean, cov


def vfe_bound(spec, x, y, noise, z, eps=1e-12):
    x, z = np.asarray(x, dtype=np.float64), np.asarray(z, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    d = np.broadcast_to(noise, y.shape).astype(np.float64)
    Kzz = ok.gram(spec, z, None, jitter=eps)
    Kxz = ok.gram(spec, x, z)
    Q = Kxz @ np.linalg.solve(Kzz, Kxz.T)
    kdiag = ok.gram_diag(spec, x)
    return _mvn_logpdf(Q + np.diag(d), y) - 0.5 * np.sum((kdiag - np.diag(Q)) / d)


def vfe_posterior(spec, x, y, noise, z, xs, eps=1e-12):
    x, z, xs = (np.asarray(a, dtype=np.float64) for a in (x, z, xs))
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    d = np.broadcast_to(noise, y.shape).astype(np.float64)
    Kzz = ok.gram(spec, z, None, jitter=eps)
    Kxz = ok.gram(spec, x, z)
    Ksz = ok.gram(spec, xs, z)
    Sigma = Kzz + Kxz.T @ (Kxz / d[:, None])
    mean = Ksz @ np.linalg.solve(Sigma, Kxz.T @ (y / d))
    cov = ok.gram(spec, xs) - Ksz @ np.linalg.solve(Kzz, Ksz.T) + Ksz @ np.linalg.solve(Sigma, Ksz.T)
    return mean, cov


class Process:
    """A Gaussian process given by explicit callables `mean(x) -> (n,)` and `k(a, b) -> (na, nb)`; conditioning returns
    a new Process by the dense closed forms above (no factorisations are shared with the product).  Used to pin
    observations OF A POSTERIOR - `PseudoObs` / `Obs` built on `f | obs`, conditioning twice - for every combination
    of exact and inducing-point observations (what stheno supports through its measure algebra; the reference reaches
    it in `GPARRegressor.logpdf(..., posterior=True)`, /root/reference/gpar/regression.py:493-499)."""

    def __init__(self, mean, k):
        self.mean, self.k = mean, k

    @classmethod
    def prior(cls, spec):
        return cls(lambda x: np.zeros(np.asarray(x).shape[0]), lambda a, b: ok.gram(spec, np.asarray(a, float), np.asarray(b, float)))

    def logpdf(self, x, y, noise, eps=1e-12):
        y = np.asarray(y, dtype=np.float64).reshape(-1)
        S = self.k(x, x) + np.diag(np.broadcast_to(noise, y.shape) + eps)
        return _mvn_logpdf(S, y - self.mean(x))

    def condition(self, x, y, noise, eps=1e-12):
        y = np.asarray(y, dtype=np.float64).reshape(-1)
        S = self.k(x, x) + np.diag(np.broadcast_to(noise, y.shape) + eps)
        a = np.linalg.solve(S, y - self.mean(x))
        mean = lambda xs: self.mean(xs) + self.k(xs, x) @ a
        k = lambda p, q: self.k(p, q) - self.k(p, x) @ np.linalg.solve(S, self.k(x, q))
        return Process(mean, k)

    def vfe_bound(self, x, y, noise, z, eps=1e-12, method="vfe"):
        """Titsias' bound (method "vfe"); the DTC / FITC log-densities log N(y; m, Q + D [+ diag(K - Q)])."""
        y = np.asarray(y, dtype=np.float64).reshape(-1)
        d = np.broadcast_to(noise, y.shape).astype(np.float64)
        Kzz = self.k(z, z) + eps * np.eye(np.asarray(z).shape[0])
        Kxz = self.k(x, z)
        Q = Kxz @ np.linalg.solve(Kzz, Kxz.T)
        kdiag = np.diag(self.k(x, x))
        if method == "fitc":
            return _mvn_logpdf(Q + np.diag(d + kdiag - np.diag(Q)), y - self.mean(x))
        if method == "dtc":
            return _mvn_logpdf(Q + np.diag(d), y - self.mean(x))
        return _mvn_logpdf(Q + np.diag(d), y - self.mean(x)) - 0.5 * np.sum((kdiag - np.diag(Q)) / d)

    def condition_sparse(self, x, y, noise, z, eps=1e-12, method="vfe"):
        y = np.asarray(y, dtype=np.float64).reshape(-1)
        d = np.broadcast_to(noise, y.shape).astype(np.float64)
        Kzz = self.k(z, z) + eps * np.eye(np.asarray(z).shape[0])
        Kxz = self.k(x, z)
        if method == "fitc":
            d = d + np.diag(self.k(x, x)) - np.sum(Kxz * np.linalg.solve(Kzz, Kxz.T).T, axis=1)
        Sigma = Kzz + Kxz.T @ (Kxz / d[:, None])
        b = np.linalg.solve(Sigma, Kxz.T @ ((y - self.mean(x)) / d))
        mean = lambda xs: self.mean(xs) + self.k(xs, z) @ b
        k = lambda p, q: (self.k(p, q) - self.k(p, z) @ np.linalg.solve(Kzz, self.k(z, q))
                          + self.k(p, z) @ np.linalg.solve(Sigma, self.k(z, q)))
        return Process(mean, k)
