"""CPU oracle for the GPAR per-layer GP inference hot path.

THIS PACKAGE IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import it.
``gpar_amd`` never imports it and has no CPU fallback: without the HIP library the product
raises.

PARITY UNPINNED (against stheno): the arithmetic of the reference lives in un-vendored,
un-pinned third-party packages (``stheno>=1.1``, ``backends>=1`` a.k.a. ``lab``,
``backends-matrix>=1``, ``mlkernels`` via stheno, ``varz>=0.6``; /root/reference/setup.py:3-12)
that are not installed in the build container and cannot be installed (no network), and the
reference's own tests hold no golden numbers for this path — only identities
(/root/reference/tests/test_model.py:118-149,152-218,244-272;
/root/reference/tests/test_regression.py:92-137).  The oracle therefore restates the
published textbook algorithms those packages implement (Rasmussen & Williams 2006 alg. 2.1 for
exact GP regression; Titsias 2009 for the VFE inducing-point bound; mlkernels' EQ / RQ /
Linear / stretch / periodic / select definitions) and is pinned by
  (i)   every literal / closed-form identity the reference tests do hold (tests/test_oracle*.py),
  (ii)  an O(n^3)-free-of-Cholesky second formula (slogdet + solve) at small n,
  (iii) central finite differences for every analytic gradient,
  (iv)  an independent published implementation of the same definitions: scikit-learn's
        GaussianProcessRegressor with RBF / RationalQuadratic / DotProduct / ExpSineSquared kernels
        (Gram matrices, log marginal likelihood, posterior mean and covariance agree to 1e-10;
        tests/test_oracle_sklearn.py).  This is not the reference, so the header above stands.
"""
