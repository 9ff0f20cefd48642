/*
 * gpar_hip.h — C ABI of libgpar_hip.so: the MI355X (gfx950) implementation of GPAR's
 * per-layer GP inference hot path.
 *
 * What this boundary replaces.  The reference (wesselb/gpar 0.3.2) contains no native code: every
 * Gram matrix, Cholesky factorisation and triangular solve is executed inside third-party Python
 * packages (stheno / mlkernels / matrix / lab -> torch CPU fp64 -> LAPACK), reached from
 *   - gpar/model.py:226      f.measure.logpdf(obs)            (Gram + potrf + trsv, per layer)
 *   - gpar/model.py:298-301  (f | obs).mean(x_)               (posterior mean)
 *   - gpar/model.py:264,270  f(x[, noise]).sample()           (posterior covariance + potrf + sample)
 *   - gpar/model.py:286-289  Obs / PseudoObs construction     (dense / inducing-point observations)
 *   - gpar/regression.py:92-180  kernel algebra of one layer  (which Gram matrix is built)
 *   - gpar/regression.py:459 minimise_l_bfgs_b(objective...)  (objective + gradient)
 * Each entry point below cites the call site whose arithmetic it performs.  A binding a maintainer of
 * the reference would add (ctypes) is shown in INTEGRATION.md.
 *
 * Conventions
 *   - All matrices are ROW-MAJOR IEEE fp64 in device (HBM) memory with an explicit leading dimension
 *     (elements between consecutive rows).  For symmetric matrices the LOWER triangle is authoritative;
 *     the strict upper triangle is never read; it may hold anything on entry and is SCRATCH: gpar_potrf keeps the
 *     hand-off words of its persistent panel kernels there (rows 0 .. 4, columns 8 .. 63 of every 64 x 64 diagonal tile:
 *     progress words of a panel's team, inverses of 16 x 16 diagonal blocks, and - where several panels share a launch -
 *     per-row-block progress words and tile counters), zeroed by one launch at the start of a factorisation.
 *   - Plain pointers and sizes only.  The caller owns every buffer, including workspace
 *     (gpar_workspace_doubles gives the sizes); the library never allocates or frees device MEMORY.  Every compute call
 *     only enqueues work on `stream` (a hipStream_t passed as void*) and returns; nothing synchronises the host except
 *     gpar_profile_read (a measurement aid).  What the library does create, lazily and once per device: one low-priority
 *     side stream per caller stream and a ring of events (gpar_potrf's look-ahead), the events of the profile hook, and the
 *     per-kernel dynamic-LDS attributes - so the first call on a device must not be made inside a stream capture.
 *   - Return value: 0 = launched OK; < 0 = -(hipError_t) or -1000-x for argument errors.  Numerical
 *     failure (non-positive pivot) is reported LAPACK-style through a device-side `info` word
 *     (1-based index of the first bad pivot, 0 = success) so that no host sync is forced.
 *   - gpar_potrf may use one internal low-priority stream per caller stream (look-ahead) and joins it back before
 *     it returns control of `stream`; callers see one in-order stream.  Process-global state (those streams, an event
 *     ring, the profile hook) sits behind one mutex: entry points only enqueue, so calls from several host threads
 *     (each with its own stream) are safe and simply serialise their enqueueing.  For the duration of a call the
 *     device that owns `stream` is made current (and the caller's current device restored on return), so a thread need
 *     not have called hipSetDevice; a null `stream` means the current device's default stream.
 *   - Pointers should be 16-byte aligned and leading dimensions even for the vectorised paths; other
 *     values are accepted and take a slower scalar path.
 */
#ifndef GPAR_HIP_H
#define GPAR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPAR_ABI_VERSION 7

/* ---- kernel specification -------------------------------------------------------------------
 * A GPAR layer kernel (gpar/regression.py:92-180) is a sum of products of elementary kernels applied
 * to selected, rescaled (and possibly periodically embedded) input columns.  It is handed over in two
 * flat structs:
 *   gpar_fspec_t : how a raw design row x (m inputs + previous outputs, gpar/model.py:320) becomes a
 *                  feature row z:  z[q] = embed_q(x[col[q]]) * inv_scale[q]
 *                  embed 0: identity   1: sin(freq*x)   2: cos(freq*x)   (mlkernels .periodic()).
 *   gpar_kspec_t : k(z, z') = sum_t coef[t] * prod_{f in term t} phi_f(z[off:off+nd], z'[off:off+nd])
 *                  phi EQ: exp(-r2/2); RQ: (1 + r2/(2 alpha))^-alpha; LINEAR: <z, z'>; r2 = |z - z'|^2.
 *                  A term without factors is the constant kernel `coef`.
 */
#define GPAR_MAX_DIMS 96
#define GPAR_MAX_FACTORS 12
#define GPAR_MAX_TERMS 8

#define GPAR_EMBED_ID 0
#define GPAR_EMBED_SIN 1
#define GPAR_EMBED_COS 2

#define GPAR_K_EQ 0
#define GPAR_K_RQ 1
#define GPAR_K_LINEAR 2

typedef struct {
    int32_t dz;                      /* number of feature dims (<= GPAR_MAX_DIMS) */
    int32_t pad_;
    int32_t col[GPAR_MAX_DIMS];      /* source column of x */
    int32_t embed[GPAR_MAX_DIMS];    /* GPAR_EMBED_* */
    double inv_scale[GPAR_MAX_DIMS]; /* 1 / length scale */
    double freq[GPAR_MAX_DIMS];      /* 2*pi / period (periodic dims only) */
} gpar_fspec_t;

typedef struct {
    int32_t type; /* GPAR_K_* */
    int32_t term; /* index of the product term this factor multiplies into */
    int32_t off;  /* first feature dim */
    int32_t nd;   /* number of feature dims (0 allowed: EQ/RQ -> 1, LINEAR -> 0) */
    double alpha; /* RQ shape */
} gpar_factor_t;

typedef struct {
    int32_t nterms;
    int32_t nfactors;
    double coef[GPAR_MAX_TERMS];
    gpar_factor_t factor[GPAR_MAX_FACTORS];
} gpar_kspec_t;

/* flags */
#define GPAR_GRAM_LOWER 1  /* symmetric Gram (z1 == z2): write only tiles that touch the lower triangle */
#define GPAR_GEMM_C_LOWER 1 /* C is square-aligned: compute/store only elements with col <= row */
#define GPAR_GEMM_A_LOWER 2 /* treat op(A) as lower triangular (entries with k > m are zero) */
#define GPAR_GEMM_K_FROM_ROW 4 /* op(A)[m][k] is stored as zero for k < m (with C_LOWER also op(B)^T[n][k] for k < row): k starts at the tile's first row */
#define GPAR_GEMM_K_TO_COL 8 /* op(B)[k][n] is stored as zero for k > n (upper-triangular op(B)): k ends with the tile's last column */

/* ---- run-time specialisation (ABI v4) -----------------------------------------------------------
 * A layer's kernel STRUCTURE is fixed when the model is built (gpar/regression.py:92-180 decides the term list once per layer);
 * for large problems gpar_gram compiles a kernel for that structure with hiprtc on first use (values - coefficients, RQ shapes -
 * stay arguments: training never recompiles) and caches it per device; small problems and any compilation failure use the
 * ahead-of-time interpreter, which computes the same bits.  Environment: GPAR_GRAM_JIT_MIN_ENTRIES (default 2^26 entries per
 * launch - where a training run repays the 0.3-0.6 s a structure costs; 0 = always, negative = never).  Structures with more
 * than 16 feature dims have no generated Gram kernel (they are bound by their arithmetic either way).
 *   gpar_jit_compile_check  compiles (does not load) the kernel of `kind` for `ks` / `dz` and architecture `arch` (e.g. "gfx950"):
 *                           returns the code-object size, or -1 with the compiler's log in `log`; needs no GPU.  kind 0: Gram
 *                           (an argument error for a structure with more than 16 feature dims);
 *                           1 / 11: parameter-gradient pass with symmetric / rectangular weights (21 / 31: with frequency
 *                           derivatives of periodic features); 2 / 12: input-gradient pass, symmetric / rectangular weights.
 *                           (gpar_gram_grad* / gpar_gram_input_grad use generated kernels from GPAR_GRAD_JIT_MIN_ENTRIES weight
 *                           entries on, default 2^24; summation order differs from the interpreter's: agreement to rounding.)
 *   gpar_jit_stats          kernels compiled / compilations failed / structures cached so far in this process. */
int gpar_jit_compile_check(int kind, const gpar_kspec_t* ks, int dz, const char* arch, char* log, int log_len);
/* Compile (if not cached yet) and load the kernel of `kind` (codes as above) for this structure on the device that owns `stream`,
 * on the calling thread and outside the library's lock: call it for all layers of a model from several host threads at once and
 * no compilation happens inside an evaluation.  0: ready; 1: compilation failed (the interpreter will serve); < 0: argument error.
 * [a GPARRegressor's layer structures are known when the model is built: gpar/regression.py:92-180] */
int gpar_jit_prepare(int kind, const gpar_kspec_t* ks, int dz, void* stream);
/* Creates NOW what the library otherwise creates at first use on the device that owns `stream`: the look-ahead side stream paired
 * with `stream`, the event ring, the kernels' dynamic-LDS attributes (not the profile hook's events: profiling is not capturable).  After it no entry point called
 * on `stream` creates a HIP object, so a sequence of calls can be captured into a hipGraph (a run-time compiled
 * kernel structure must have been launched once before the capture).  Optional: everything still initialises lazily without it. */
int gpar_init(void* stream);
int gpar_jit_stats(int* compiled, int* failures, int* cached);
/* Kernels compiled at BUILD time (ABI v5).  gpar_jit_compile compiles the kernel of `kind` (codes as above) for a structure and
 * architecture and returns its code object (code_out / capacity; the return value is the size, -1 on a compilation failure with the
 * compiler's log in `log`) and the key under which an archive holds it (key_out); needs no GPU.  The library's build step collects
 * the common layer structures into gpar_aot_<arch>.bin next to the library; a structure found there is loaded instead of compiled -
 * and, costing nothing, is then used at every problem size (GPAR_AOT=0: ignore the archive).  gpar_aot_stats: entries the archive of
 * the current device's architecture holds (0 before the first lookup) / kernels loaded from it so far.
 * [the structures are those gpar/regression.py:92-180 builds for the reference's keyword combinations] */
long long gpar_jit_compile(int kind, const gpar_kspec_t* ks, int dz, const char* arch, void* code_out, long long capacity, char* key_out,
                           int key_len, char* log, int log_len);
int gpar_aot_stats(int* entries, int* loaded);
/* (ABI v6) What ties an archive to the library that may use it: a hash of the sources this library generates for a probe structure
 * covering every factor type and kernel kind, and of GPAR_ABI_VERSION.  The build step writes it into the archive's header; the
 * library ignores an archive whose ABI version or fingerprint is not its own (stale code objects could have another argument
 * layout or other arithmetic) and compiles at run time instead.  Needs no GPU.  [no reference counterpart: build hygiene] */
unsigned long long gpar_aot_fingerprint(void);

int gpar_abi_version(void);
size_t gpar_sizeof_fspec(void);
size_t gpar_sizeof_kspec(void);

/* z = features(x).  x: n x (>= max col+1) ldx; z: n x dz ldz.   [mlkernels stretch/periodic/select,
 * reached from gpar/regression.py:110,127-129,138,146,166,178] */
int gpar_featurize(const gpar_fspec_t* fs, const double* x, int n, int ldx, double* z, int ldz, void* stream);

/* K[a][b] = row_scale[a] * k(z1[a], z2[b]) (+ diag_add[a] + diag_const if a == b and z1 == z2).
 * row_scale may be NULL (= 1): the inducing-point path builds D^-1/2 K_xz directly (gpar/model.py:286-287), so that the
 * scaled cross-Gram never needs a pass of its own.
 * [mlkernels K(x, y); f(x, noise / w) adds diag(noise / w): gpar/model.py:287-289; lab's B.epsilon jitter] */
int gpar_gram(const gpar_kspec_t* ks, const double* z1, int n1, int ldz1, const double* z2, int n2, int ldz2,
              int dz, double* K, int ldk, int flags, const double* diag_add, double diag_const, const double* row_scale,
              void* stream);
/* `batch` symmetric Gram matrices of one size in one launch: K_b = k(z_b, z_b) + diag(diag_add) + diag_const I with input set b at
 * z + b * stride_z and its matrix at K + b * stride_k (elements); diag_add (n values or null) is shared by the batch.
 * [the prior covariances K(x_s, x_s) of all posterior samples of a layer, gpar/model.py:264,270 via regression.py:559-563] */
int gpar_gram_batch(const gpar_kspec_t* ks, const double* z, int n, int ldz, long long stride_z, int dz, double* K, int ldk,
                    long long stride_k, int flags, const double* diag_add, double diag_const, int batch, void* stream);

/* out[a] = k(z[a], z[a])   [kernel diagonal; VFE trace term, posterior marginal variances] */
int gpar_gram_diag(const gpar_kspec_t* ks, const double* z, int n, int ldz, int dz, double* out, void* stream);

/* zd[q] = d z[q] / d freq[q] (zero for non-periodic features): the feature derivative the period gradients need. */
int gpar_featurize_dfreq(const gpar_fspec_t* fs, const double* x, int n, int ldx, double* zd, int ldz, void* stream);

/* Gradient moment sums of  1/2 sum_ab W_ab dK_ab/dtheta  for every kernel parameter, one fused pass over the lower
 * triangle of the symmetric n x n matrix W (= alpha alpha^T - K^-1 for the log marginal likelihood).
 * out[GPAR_GRAD_NACC]: [0, MAX_TERMS) C_t; then [MAX_FACTORS) Al_f; then [MAX_DIMS) A_q; then [MAX_DIMS) P_q
 * (definitions: gpar_amd/csrc/gram.h).  workspace: nblocks * GPAR_GRAD_NACC doubles; nblocks = grid size (<= tiles).
 * zd may be NULL when the kernel has no periodic features.  At most 4 factors per product term.
 * [replaces torch autograd through exp / pw_dists2 / cholesky / solve_triangular inside
 *  varz.minimise_l_bfgs_b, gpar/regression.py:459] */
#define GPAR_GRAD_NACC (GPAR_MAX_TERMS + GPAR_MAX_FACTORS + 2 * GPAR_MAX_DIMS)
int gpar_grad_nacc(void);
int gpar_gram_grad(const gpar_kspec_t* ks, const double* z, const double* zd, int n, int ldz, int dz, const double* W,
                   int ldw, double* workspace, int nblocks, double* out, void* stream);
/* One dense layer's training objective AND the ingredients of its gradient in one call (ABI v5):
 *   out[0] = log N(y; 0, k(x, x) + diag(noise_diag) + jitter I),  out[1] = log det,  out[2 .. 2 + GPAR_GRAD_NACC) = the moment sums of
 *   1/2 sum_ab W_ab dK_ab/dtheta (as gpar_gram_grad),  half_diag[a] = 1/2 W_aa (the derivative with respect to noise_diag[a]),
 * with W = alpha alpha^T - (K + D)^-1.  The same launches the separate entry points make, in the same order (same bits):
 * gpar_featurize (+ gpar_featurize_dfreq when zd is non-null), gpar_gram, the augmented factorisation, gpar_chol_inverse,
 * gpar_trmv_upper of the inverse's workspace X = L^-T on the row L^-1 y (alpha; a gpar_trsm_rln until ABI v6), the rank-1 update of W, gpar_gram_grad.  Workspaces: z (and zd) n x dz (ldz); A (n + 1) x (n + 1);
 * X and W n x n; alpha n doubles; workspace nblocks * GPAR_GRAD_NACC doubles.  What it removes is the caller's side: a Python host
 * spends ~1 ms per evaluation on ~30 launches and three synchronisations around them, which IS the evaluation below n ~ 1000 - the
 * size the reference's own examples train at.
 * [objective + gradient of one layer inside varz.minimise_l_bfgs_b, gpar/regression.py:434-459] */
int gpar_logpdf_dense_grad(const gpar_fspec_t* fs, const gpar_kspec_t* ks, const double* x, int n, int ldx, const double* y, long incy,
                           const double* noise_diag, double jitter, double* z, double* zd, int ldz, double* A, int lda, double* X, int ldxw,
                           double* W, int ldw, double* alpha, double* workspace, int nblocks, double* out, double* half_diag, int* info,
                           int potrf_flags, void* stream);
/* The same evaluation for a factor that ALREADY EXISTS (ABI v7) - the second half of gpar_logpdf_dense_grad: the value from the corner
 * of A and `logdet`, K^-1 from L, alpha, W, the weighted-sum pass, 1/2 diag W (gpar_featurize_dfreq first when zd is non-null).  For
 * callers that factor several layers' matrices TOGETHER: gpar_logpdf_dense_build per layer into the slots of one buffer, ONE
 * gpar_potrf_batch over the slots, then this call per layer on a stream of its own (the lock-step training rendezvous of
 * gpar_amd/fastfit.py: the p independent layer optimisations of fit(fix=True), gpar/regression.py:418-446, evaluate in rounds).
 * `logdet` / `info`: the layer's words of the batch (device); out[1] <- logdet[0], info_out[0] <- info[0] (either of the last two may
 * be NULL), so that one device-to-host copy of `out` .. `info_out` carries everything the host reads.  A failed factorisation
 * (info != 0) leaves garbage in out / half_diag, as gpar_logpdf_dense_grad does.
 * [objective + gradient of one layer inside varz.minimise_l_bfgs_b, gpar/regression.py:434-459] */
int gpar_logpdf_dense_grad_finish(const gpar_fspec_t* fs, const gpar_kspec_t* ks, const double* x, int n, int ldx, double* z, double* zd,
                                  int ldz, double* A, int lda, const double* logdet, const int* info, double* X, int ldxw, double* W, int ldw,
                                  double* alpha, double* workspace, int nblocks, double* out, double* half_diag, int* info_out, void* stream);

/* The same moment sums of  sum W dK/dtheta  for the other weight shapes the inducing-point (VFE) bound needs
 * [gradient of the PseudoObs elbo, gpar/model.py:226,286-287 under varz's optimiser]:
 *   GPAR_GRAD_SYM   z2 == z1: W symmetric n1 x n1, lower triangle read, sum over all pairs (what gpar_gram_grad does);
 *   GPAR_GRAD_RECT  W a full n1 x n2 matrix of cross-Gram weights K(x1_a, x2_j);
 *   GPAR_GRAD_DIAG  z2 == z1: W a vector of n1 weights of the prior variances k(x_a, x_a).
 * Sums are FULL sums (no factor 1/2).  zd1 / zd2 may both be NULL.  nblocks <= number of 64 x 64 tiles of the mode. */
#define GPAR_GRAD_SYM 0
#define GPAR_GRAD_RECT 1
#define GPAR_GRAD_DIAG 2
int gpar_gram_grad_cross(const gpar_kspec_t* ks, const double* z1, const double* zd1, int n1, int ldz1, const double* z2,
                         const double* zd2, int n2, int ldz2, int dz, const double* W, int ldw, int mode, double* workspace,
                         int nblocks, double* out, void* stream);

/* Gradient with respect to the inputs, in feature space:  out[a][q] = sum_b W(a, b) d k(z1_a, z2_b) / d z1_a[q]  for the
 * pair weights W (GPAR_GRAD_RECT: n1 x n2;  GPAR_GRAD_SYM: z2 == z1, W symmetric, lower triangle read).  out: n1 x dz (ldo);
 * workspace: gpar_workspace_doubles(GPAR_WS_INPUT_GRAD, n1, dz, nsplit) doubles; nsplit >= 1 column splits (summed in order).
 * The caller chains d z / d x (1 / scale, and the derivative of the periodic embedding) back to the design-matrix columns.
 * [torch autograd through mlkernels' pairwise distances when a layer's inputs are posterior means of earlier layers -
 *  fit(fix=False) with replace / impute / inducing points: gpar/regression.py:447-456 through gpar/model.py:291-322 - and
 *  the derivative with respect to inducing-point locations (the reference's todo.tasks:5)] */
int gpar_gram_input_grad(const gpar_kspec_t* ks, const double* z1, int n1, int ldz1, const double* z2, int n2, int ldz2, int dz,
                         const double* W, int ldw, int mode, int nsplit, double* workspace, double* out, int ldo, void* stream);

/* Partial right-looking blocked Cholesky of the leading `nf` columns of the symmetric N x N matrix A
 * (lower triangle).  On exit A[:, :nf] holds L (N x nf, lower trapezoid) and A[nf:, nf:] holds the Schur
 * complement A22 - L21 L21^T.  nf == N is the ordinary potrf.  logdet (device, optional) += 2*sum log L_jj;
 * info (device, optional) = first non-positive pivot (1-based), untouched on success (zero it first).
 * Appending rows below K turns this one routine into the whole exact-GP computation:
 *   row  [y^T, 0]      -> z^T = (L^-1 y)^T and -|z|^2          (log marginal likelihood, gpar/model.py:226)
 *   rows [K_*x, K_**]  -> V^T = K_*x L^-T and K_** - V^T V      (posterior covariance, gpar/model.py:264,270)
 * Its panel kernel hands tiles between workgroups of one launch; a workgroup only ever waits for workgroups with a LOWER
 * block index (dispatched before it), so nothing depends on all of them being co-resident and several factorisations may
 * run at once from different streams; every wait is bounded all the same, and a timeout is reported as info = -77
 * (results invalid: retry with GPAR_POTRF_UNFUSED).
 * [matrix.cholesky -> torch.linalg.cholesky (LAPACK dpotrf)] */
int gpar_potrf(double* A, int N, int nf, int lda, double* logdet, int* info, void* stream);
/* The same with hints.  GPAR_POTRF_NO_LOOKAHEAD: the caller keeps three or more factorisations in flight on streams of its
 * own (independent layers, gpar/model.py:221-243 run concurrently), so the internal side stream is not used. */
#define GPAR_POTRF_NO_LOOKAHEAD 1
/* GPAR_POTRF_UNFUSED: panels by separate diagonal / strip / update kernels (no persistent kernel, no in-launch hand-offs):
 * the caller's retry path should a hand-off ever time out (info = -77). */
#define GPAR_POTRF_UNFUSED 2
int gpar_potrf_ex(double* A, int N, int nf, int lda, double* logdet, int* info, int flags, void* stream);

/* One dense layer's log marginal likelihood, fused (features, Gram + noise_diag + jitter, observations into the augmented row,
 * partial factorisation, value):  value[0] = log N(y; 0, k(x, x) + diag(noise_diag) + jitter I).  x: n x width (ldx); y: n values
 * with stride incy; noise_diag: n values (or null); z: n x dz workspace (ldz); A: (n + 1) x (n + 1) workspace (lda), holds the
 * factor, L^-1 y in row n and -|L^-1 y|^2 in the corner on exit; logdet / info / value: one word each, written (not accumulated).
 * potrf_flags as for gpar_potrf_ex.  [f.measure.logpdf(Obs(f(x, noise / w), y)) for a prior process, gpar/model.py:226,289] */
int gpar_logpdf_dense(const gpar_fspec_t* fs, const gpar_kspec_t* ks, const double* x, int n, int ldx, const double* y, long incy,
                      const double* noise_diag, double jitter, double* z, int ldz, double* A, int lda, double* logdet, int* info,
                      double* value, int potrf_flags, void* stream);

/* The same in three steps, for `batch` layers that do not feed one another and have the same number of rows (complete data:
 * the design matrix of layer i is [x, y_<i], known up front - gpar/model.py:221-243 visits them in a loop all the same):
 *   gpar_logpdf_dense_build   per layer: features, Gram + noise_diag + jitter and the observations into its (n + 1) x (n + 1)
 *                             matrix A_b = A + b * stride_a, its logdet[b] / info[b] zeroed;
 *   gpar_potrf_batch          the `batch` partial factorisations in LOCK-STEP - one panel launch and one batched trailing update
 *                             per 512 columns.  At the sizes where a factorisation is a chain of latency-bound panel kernels
 *                             (n <= ~4608) several of them on separate streams contend for compute-unit slots; in lock-step the
 *                             chain is paid once and every launch carries `batch` times the parallel work;
 *   gpar_logpdf_dense_finish  value[b] from the corner of A_b and logdet[b].
 * Arguments as for gpar_logpdf_dense / gpar_potrf_ex; stride_a (elements, even) separates consecutive matrices; logdet / info /
 * value hold `batch` words.  [sum over layers of f.measure.logpdf(obs), gpar/model.py:221-243] */
int gpar_logpdf_dense_build(const gpar_fspec_t* fs, const gpar_kspec_t* ks, const double* x, int n, int ldx, const double* y, long incy,
                            const double* noise_diag, double jitter, double* z, int ldz, double* A, int lda, double* logdet, int* info,
                            void* stream);
int gpar_potrf_batch(double* A, int batch, long long stride_a, int N, int nf, int lda, double* logdet, int* info, int flags, void* stream);
int gpar_logpdf_dense_finish(const double* A, int batch, long long stride_a, int n, int lda, const double* logdet, double* value,
                             void* stream);

/* A whole lock-step evaluation in ONE call (ABI v5): the log marginal likelihoods of `batch` layers over one set of n rows, and
 * their sum.  Layer b is described by layers[b] (HOST memory, read during the call): its feature map and kernel, its observation-
 * noise variance, and the column y_col of y (and of w) that holds its observations.  x: n x width, the widest design matrix -
 * for a GPAR [inputs, y_0 .. y_(p-2)] - from which every layer's feature map selects its own columns (gpar/model.py:320: layer i
 * sees [x, y_<i]); y: n x (>= max y_col + 1), ldy; w: weights laid out as y (ldw) or NULL for unit weights - the noise diagonal
 * of layer b is noise_b / w[:, y_col], an IEEE division per row.  Workspaces: z: batch * n rows of ldz >= max dz doubles;
 * nd: batch * n doubles (only read when w != NULL); A / lda / stride_a / logdet / info as for gpar_potrf_batch ((n + 1) x (n + 1)
 * matrices); value: `batch` words; total: one word or NULL, = ((0 + value[0]) + value[1]) + ... in layer order.
 * The same kernels, in the same order per matrix, as gpar_logpdf_dense_build / gpar_potrf_batch / gpar_logpdf_dense_finish: the
 * results are bit-identical; what it removes is the caller's side - a Python host spends ~0.1 ms per layer on tensor bookkeeping
 * and ~25 small launches per evaluation around those calls, which is most of an evaluation at n <= 1024.
 * [sum over layers of f.measure.logpdf(Obs(f(x_i, noise_i / w_i), y_i)), gpar/model.py:221-243 with :286-289, complete data] */
typedef struct {
    const gpar_fspec_t* fs;
    const gpar_kspec_t* ks;
    double noise;
    int32_t y_col;
    int32_t pad_;
} gpar_layer_t;
size_t gpar_sizeof_layer(void);
int gpar_logpdf_lockstep(const gpar_layer_t* layers, int batch, const double* x, int n, int ldx, const double* y, int ldy, const double* w,
                         int ldw, double jitter, double* z, int ldz, double* nd, double* A, int lda, long long stride_a, double* logdet,
                         int* info, double* value, double* total, int potrf_flags, void* stream);

/* B <- B L^-T  (right side, lower, transposed: forward substitution on the rows of B; B is nrows x n).
 * [solve_triangular inside matrix.iqf_diag / PosteriorKernel, reached from gpar/model.py:226,264,298] */
int gpar_trsm_rlt(const double* L, int n, int ldl, double* B, int nrows, int ldb, void* stream);
/* The same solve, PREDICATED on a device word (ABI v5): it happens iff (flag[0] != 0) == (run_if != 0), decided on the device when
 * the kernels start - flag is written by an earlier kernel of the stream (gpar_chol_spread), and no host synchronisation is
 * involved; a solve that is skipped costs its (empty) launches, B is left untouched.  flag == NULL: unconditional.
 * gpar_chol_spread: spread[0] <- (max L_jj / min L_jj)^2, a lower bound of cond(L L^T) that the factor holds for free;
 * flag[0] <- 1 if the spread exceeds `limit` or a pivot is not positive, else 0 (either pointer may be NULL).
 * [The inducing-point bound (stheno PseudoObs, gpar/model.py:286-287) needs A = I + L_z^-1 K_zx D^-1 K_xz L_z^-T.  Solving the
 *  n x M cross-Gram against L_z first (M^2 n flops) is backward stable whatever cond(K_zz); forming K_zx D^-1 K_xz first and
 *  solving the M x M result from both sides (2 M^3 flops) loses a factor ~cond(K_zz) and is only taken when the pivot spread of
 *  L_z says that this is harmless - each of the two orders is predicated on the same word, one of them runs.] */
int gpar_trsm_rlt_if(const double* L, int n, int ldl, double* B, int nrows, int ldb, const int* flag, int run_if, void* stream);
int gpar_chol_spread(const double* L, int n, int ldl, double limit, double* spread, int* flag, void* stream);
/* The scalar side of the inducing-point bound in two calls instead of ~25 tensor operations per layer (ABI v5):
 *   gpar_vfe_assemble  A <- [[G + diag_add I, .], [c^T, 0]] ((M + 1) x (M + 1): what gpar_potrf factors next; lower triangle of G read),
 *                      logdet / info zeroed, scal (256 doubles: 64 partial sums of each) <- sum ys^2, sum kdiag / d, sum log d (n terms), tr G;
 *   gpar_vfe_value     out[0] <- -1/2 (with_trace (S1 - S3) + S2 + n log 2 pi + logdet + S0 + A[M][M]) once A is factored (its corner then
 *                      holds -|L_A^-1 c|^2), S_q the sums of the parts in order.
 * [the elbo of stheno's PseudoObs, gpar/model.py:226 with :286-287: trace term, log-determinants and quadratic form] */
int gpar_vfe_assemble(const double* G, int M, int ldg, const double* c, const double* ys, const double* kdiag, const double* d, int n,
                      double diag_add, double* A, int lda, double* scal, double* logdet, int* info, void* stream);
int gpar_vfe_value(const double* scal, const double* logdet, const double* A, int lda, int M, int n, int with_trace, double* out, void* stream);
/* B <- B L^-1  (right side, lower, not transposed: backward substitution on the rows of B). */
int gpar_trsm_rln(const double* L, int n, int ldl, double* B, int nrows, int ldb, void* stream);

/* Kinv (lower triangle) <- (L L^T)^-1 given the Cholesky factor L; X is an n x n workspace (its upper triangle holds L^-T on
 * exit; below the diagonal it is scratch).
 * Triangular-aware: 2 n^3 / 3 flops.  Kinv doubles as scratch while L^-T is formed (recursive blocked inversion for n a
 * multiple of 512, otherwise the solve of the identity).  [the K^-1 that the analytic gradient
 * 1/2 tr((aa^T - K^-1) dK) needs; replaces autograd through cholesky / solve_triangular, gpar/regression.py:459] */
int gpar_chol_inverse(const double* L, int n, int ldl, double* X, int ldx, double* Kinv, int ldk, void* stream);

/* C <- alpha * op(A) op(B) + beta * C with op(A) m x k, op(B) k x n.  ta: A is stored k x m (transposed);
 * tb: B is stored n x k (transposed).  fp64 on the matrix cores (v_mfma_f64_16x16x4).
 * [B.matmul in lab: K_*x alpha, V^T V, chol(var) z] */
int gpar_gemm(int ta, int tb, int m, int n, int k, double alpha, const double* A, int lda, const double* B,
              int ldb, double beta, double* C, int ldc, int flags, void* stream);

/* `batch` products of one shape in one launch: problem b reads A + b * stride_a, B + b * stride_b and writes C + b * stride_c
 * (elements).  [the per-sample downdates K(x_s, x_s) - V_s V_s^T of ancestral sampling, gpar/regression.py:559-563 run for all
 * samples of a layer at once] */
int gpar_gemm_batch(int ta, int tb, int m, int n, int k, double alpha, const double* A, int lda, long long stride_a, const double* B,
                    int ldb, long long stride_b, double beta, double* C, int ldc, long long stride_c, int flags, int batch, void* stream);

/* As gpar_gemm, with the K range cut into `splits` slices whose partial products go to `workspace`
 * (splits * m * n doubles) and are summed in slice order by a second kernel: for outputs with few 128 x 128 tiles and
 * a very long K (the inducing-point matrix B D^-1 B^T: M x M from K = n).  Deterministic.  A_LOWER / K_FROM_ROW / K_TO_COL
 * are not meaningful here. */
int gpar_gemm_splitk(int ta, int tb, int m, int n, int k, double alpha, const double* A, int lda, const double* B,
                     int ldb, double beta, double* C, int ldc, int flags, int splits, double* workspace, void* stream);

/* Small device-side reductions used by the fused paths (all asynchronous, all deterministic). */
/* out[0] (+)= sum_i x[i*incx]*y[i*incy] */
int gpar_dot(const double* x, int incx, const double* y, int incy, int n, double* out, int accumulate, void* stream);
/* out[j] = sum_i A[i][j] * v[i], j < cols, for a tall rows x cols matrix (two passes: per-128-row partial sums into
 * `workspace` - gpar_workspace_doubles(GPAR_WS_GEMV_T, rows, cols, 0) doubles - then an in-order sum).
 * [c = B D^-1 y of the inducing-point bound; stheno PseudoObs, gpar/model.py:286-287] */
int gpar_gemv_t(const double* A, int rows, int cols, int lda, const double* v, double* out, double* workspace, void* stream);
/* out[i] = sum_j A[i][j]^2, i < rows.  [marginal posterior variances k(x*, x*) - |V_i|^2, V = K_*x L^-T] */
int gpar_rownorm2(const double* A, int rows, int cols, int lda, double* out, void* stream);

/* Lower triangle (diagonal included) of the row-major n x n matrix A <-> packed storage of n (n + 1) / 2 doubles (row r at
 * offset r (r + 1) / 2).  [the exchange format of Cholesky factors between the GPUs of a node: layer-parallel conditioning
 * all-gathers packed factors, half the bytes of the padded square buffers] */
int gpar_pack_lower(const double* A, int n, int lda, double* out, void* stream);
int gpar_unpack_lower(const double* in, int n, double* A, int lda, void* stream);

/* Workspace sizes in doubles (the caller allocates every workspace; -1 for an unknown `op`):
 *   GPAR_WS_GEMM_SPLITK  (m, n, splits)   gpar_gemm_splitk
 *   GPAR_WS_GEMV_T       (rows, cols, -)  gpar_gemv_t
 *   GPAR_WS_GRAM_GRAD    (nblocks, -, -)  gpar_gram_grad / gpar_gram_grad_cross
 *   GPAR_WS_CHOL_INVERSE (n, ldx, -)      the X matrix of gpar_chol_inverse */
#define GPAR_WS_GEMM_SPLITK 1
#define GPAR_WS_GEMV_T 2
#define GPAR_WS_GRAM_GRAD 3
#define GPAR_WS_CHOL_INVERSE 4
#define GPAR_WS_INPUT_GRAD 5   /* (n1, dz, nsplit)  gpar_gram_input_grad */
long long gpar_workspace_doubles(int op, int a, int b, int c);
/* Standard normals from Philox-4x32-10 + Box-Muller: out[r][c], element index = r*cols + c in the stream
 * identified by (seed, offset).   [B.randn in Normal.sample] */
int gpar_randn(uint64_t seed, uint64_t offset, double* out, int rows, int cols, int ldo, void* stream);

/* y[i*incy] = sum_{j<=i} L[i][j] * x[j*incx], i < n: lower-triangular matrix times ONE vector (the strict upper triangle
 * is not read).  [the chol(cov) * z product of Normal.sample for a single draw; a 128-wide GEMM tile for one column is
 * all latency] */
int gpar_trmv_lower(const double* L, int n, int ldl, const double* x, int incx, double* y, int incy, void* stream);
/* y[i*incy] = sum_{j>=i} U[i][j] * x[j*incx], i < n: UPPER-triangular matrix times one vector (ABI v7; the strict lower triangle is
 * not read).  With U = X = L^-T - the workspace gpar_chol_inverse leaves behind - and x = L^-1 y (row n of an augmented factor) it is
 * alpha = (L L^T)^-1 y, the vector of the gradient's weights W = alpha alpha^T - K^-1, at memory speed instead of a backward
 * substitution.  [the alpha of the analytic gradient inside varz.minimise_l_bfgs_b, gpar/regression.py:459] */
int gpar_trmv_upper(const double* U, int n, int ldu, const double* x, int incx, double* y, int incy, void* stream);
/* y[i*incy] = alpha * sum_j A[i][j] * x[j*incx], i < rows: a general row-major matrix times ONE vector (ABI v5; one wave per row).
 * [posterior means K(x*, X) alpha and K(x*, Z) v: gpar/model.py:298-301 - matrix-vector products that a 128-wide GEMM tile serves badly] */
int gpar_gemv(const double* A, int rows, int cols, int lda, const double* x, int incx, double alpha, double* y, int incy, void* stream);
/* The same for `batch` lower-triangular matrices with one vector each, in one launch: y_b = L_b x_b (+ add_b if add is non-null),
 * matrix / vector b at L + b * stride_l, x + b * stride_x, add + b * stride_add, y + b * stride_y (elements).  [the draws of all
 * posterior samples of a layer: mean_s + chol(cov_s) z_s, gpar/regression.py:559-563] */
int gpar_trmv_lower_batch(const double* L, int batch, long long stride_l, int n, int ldl, const double* x, int incx, long long stride_x,
                          const double* add, int inca, long long stride_add, double* y, int incy, long long stride_y, void* stream);

/* Monte-Carlo reduction of predict [gpar/regression.py:589-595: np.mean / np.percentile over the sample axis]:
 * samples[s * stride + e], s < S, e < count.  mean[e] = (sum over s, in order) / S.  If lo / hi are non-null they
 * receive the order-statistic interpolations  v[k] + g (v[k+1] - v[k])  (numpy's "linear" method, evaluated with
 * numpy's _lerp so the result is bit-identical to np.percentile for the same (k, g)); the caller derives
 * (k_lo, g_lo), (k_hi, g_hi) from the percentiles exactly as numpy does.  Selection is by rank counting, O(S^2)
 * per element, S <= 65536.  An element with a NaN among its samples gets NaN bounds (and a NaN mean), as numpy gives. */
int gpar_sample_stats(const double* samples, int S, long long count, long long stride, int k_lo, double g_lo, int k_hi,
                      double g_hi, double* mean, double* lo, double* hi, void* stream);

/* Profiling hook for bench.py: when enabled, every trailing-update SYRK launched by gpar_potrf is
 * bracketed by hipEvents on its own stream; the accumulated (launches, sum of their durations in ms, union of
 * their intervals in ms - launches from different caller streams may overlap -, flops) can be read back (this
 * call synchronises the recorded events). */
int gpar_profile_enable(int on);
int gpar_profile_read(int* launches, double* ms, double* busy_ms, double* flops, int reset);

#ifdef __cplusplus
}
#endif
#endif /* GPAR_HIP_H */
