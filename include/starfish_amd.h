/*
 * starfish_amd.h -- C-ABI of the MI355X (gfx950) implementation of the Starfish per-MCMC-step
 * log-likelihood path.
 *
 * The reference (Starfish-develop/Starfish v0.4.2) is pure Python and has NO FFI / plugin layer
 * for this path: its boundary is the Python object API (SpectrumModel / Emulator and the free
 * functions of Starfish.transforms / Starfish.models.kernels).  This header is therefore the
 * boundary a maintainer would bind with ctypes; every entry point cites the
 * reference function it replaces (paths relative to the reference repository root).
 *
 * Conventions
 *   - plain C, extern "C"; no torch / C++ types in any signature;
 *   - every `d_*` argument is a DEVICE pointer to C-contiguous fp64 (or int32) owned by the
 *     caller (e.g. torch.Tensor.data_ptr()); `h_*` arguments are HOST pointers;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls only ENQUEUE work,
 *     they never synchronise -- the caller synchronises the stream before reading results;
 *   - no allocation inside hot calls: scratch comes from the caller through the *_workspace_bytes
 *     queries;
 *   - return value: 0 = ok, <0 = SF_E* (bad argument / HIP failure).  Per-item numerical status is
 *     reported LAPACK-style in `d_info[b]`: 0 ok, k>0 = order of the first non-positive pivot,
 *     <0 = SF_INFO_* (parameter outside the emulator grid, non-positive vsini, ...).
 *   - all matrices are row-major; the Cholesky factor is the LOWER triangle (A = L L^T), the strict
 *     upper triangle is not referenced.
 *   - the batched Cholesky forks part of its launches onto side streams (lookahead, slab groups) and
 *     joins them back to `stream` with events before returning: from the caller's point of view all
 *     work is ordered on `stream`.  Those streams and their event pool belong to the CONTEXT (to the
 *     calling thread for the context-free entry points): different contexts may be driven from
 *     different host threads, and live on different devices, concurrently; ONE context is not
 *     re-entrant (like the reference's model object).  Every context call makes the context's device
 *     current for the calling thread.  Only the bench timing hooks (sf_profile_*) are process-global
 *     (mutex-protected).
 *   - the library reads NO environment variable.  (The tuning / timing switches named in starfish_amd/csrc -- SF_CHOL_*,
 *     SF_DF_*, SF_WIDE_*, ... -- exist only in the separate development build `make -C starfish_amd/csrc TUNING=1`
 *     (-DSF_TUNING -> libstarfish_amd_tuning.so), which tools/ and two GPU tests load on purpose; the shipped
 *     libstarfish_amd.so contains no getenv and none of their names: tests/test_host_logic.py checks the binary.)
 */
#ifndef STARFISH_AMD_H
#define STARFISH_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SF_OK 0
#define SF_EINVAL (-1)  /* bad argument */
#define SF_ENOMEM (-2)  /* workspace too small / allocation failure */
#define SF_EHIP (-3)    /* a HIP runtime call failed (sf_last_error() has the text) */
#define SF_ENODEV (-4)  /* no gfx950 device visible */

/* per-item d_info codes (<0) */
#define SF_INFO_OUT_OF_GRID (-1) /* emulator queried outside its grid: emulator.py:377-378 */
#define SF_INFO_BAD_VSINI (-2)   /* vsini <= 0: transforms.py:121-122 */
#define SF_INFO_BAD_WEIGHT_COV (-3) /* Sigma_w not positive definite: spectrum_model.py:334 */
#define SF_INFO_BANDWIDTH (-4)   /* banded solver only: covariance support wider than the given half-width */
#define SF_INFO_INTERNAL (-5)    /* a bounded wait between workgroups / waves of ONE launch gave up: the banded sweep's wave
                                    synchronisation, or the persistent-kernel Cholesky (the default sequence of small batches)
                                    -- there EVERY matrix of the call carries the code and no result of the call is valid.
                                    Never seen on a device the process has to itself (the schedule is live by construction);
                                    on a device SHARED with other processes the launch stalls when workgroups that hold claimed
                                    tasks are kept from running, and is given up after 25 ms without a single completed task
                                    (or 4 s in any one wait): see sf_persistent_potrf / sf_persistent_potrf_status.  Callers
                                    recover by sf_persistent_potrf(0) and re-running the batch (starfish_amd/_device.py does,
                                    with a warning); it is never to be treated as a rejected walker */

#define SF_INFO_NAN (-6)         /* the likelihood came out NaN although every stage reported success: lnl = -inf, but
                                    distinguishable from a legitimately rejected walker */

#define SF_JITTER 1e-10 /* spectrum_model.py:399 */

const char* sf_version(void);
const char* sf_last_error(void);
/* number of visible HIP devices (0 when none); never fails */
int sf_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Stand-alone stage entry points (used by the drop-in free functions and stage parity tests)
 * ---------------------------------------------------------------------------------------- */

/* Starfish/models/kernels.py:7-41  global_covariance_matrix(wave, amplitude, lengthscale)
 * d_out[n*n] = Hann-tapered Matern-3/2 in velocity distance (overwritten, full matrix). */
int sf_global_cov(const double* d_wave, int n, double amplitude, double lengthscale,
                  double* d_out, void* stream);

/* Starfish/models/kernels.py:44-81  local_covariance_matrix(wave, amplitude, mu, sigma)
 * accumulate != 0 adds into d_out (the sum over kernels of spectrum_model.py:353-360). */
int sf_local_cov(const double* d_wave, int n, double amplitude, double mu, double sigma,
                 int accumulate, double* d_out, void* stream);

/* Starfish/transforms.py:93-134  rotational_broaden(wave, flux, vsini); dv = calculate_dv(wave)
 * (Starfish/utils.py:8-22).  d_flux / d_out: rows x nf, nf a power of two.
 * d_work: sf_fft_workspace_bytes(rows, nf). */
int sf_rotational_broaden(const double* d_flux, int rows, int nf, double dv, double vsini,
                          double* d_out, void* d_work, size_t work_bytes, void* stream);

/* Starfish/transforms.py:45-90  instrumental_broaden(wave, flux, fwhm) */
int sf_instrumental_broaden(const double* d_flux, int rows, int nf, double dv, double fwhm,
                            double* d_out, void* d_work, size_t work_bytes, void* stream);

size_t sf_fft_workspace_bytes(int rows, int nf);

/* Starfish/transforms.py:11-42  resample(wave, flux, new_wave): interpolating k=5 spline
 * (FITPACK knots x[0]x6, x[3:-3], x[-1]x6) per row, evaluated at new_wave.
 * h_wave[n] is a HOST pointer (the collocation factor is built on the host once per grid);
 * d_flux rows x n, d_new_wave[nq], d_out rows x nq.  d_work: sf_resample_workspace_bytes. */
int sf_resample(const double* h_wave, int n, const double* d_flux, int rows,
                const double* d_new_wave, int nq, double* d_out, void* d_work,
                size_t work_bytes, void* stream);
size_t sf_resample_workspace_bytes(int n, int rows);

/* Starfish/transforms.py:271-304  chebyshev_correct(wave, flux, coeffs): flux * chebval(
 * wave / wave_max, coeffs); h_coeffs[ncoef] is a HOST pointer (c0 must be 1 for 1-D use). */
int sf_chebyshev_correct(const double* d_wave, int n, double wave_max, const double* d_flux,
                         int rows, const double* h_coeffs, int ncoef, double* d_out,
                         void* stream);

/* Starfish/transforms.py:161-206  extinct(wave, flux, Av, Rv, law="ccm89"): flux * 10^(-0.4 A_lambda),
 * A_lambda = Av (a(x) + b(x)/Rv) from Cardelli, Clayton & Mathis (1989).  PARITY UNPINNED (the
 * reference calls the third-party `extinction` package); only the default law is provided. */
int sf_extinct_ccm89(const double* d_wave, int n, const double* d_flux, int rows, double Av, double Rv,
                     double* d_out, void* stream);
/* same with a choice of law: 0 = ccm89, 1 = odonnell94 (O'Donnell 1994: CCM89 with new optical/NIR
 * coefficients), 2 = calzetti00 (Calzetti et al. 2000, eq. 4), 3 = fitzpatrick99 (Fitzpatrick 1999: natural cubic
 * spline through Rv-dependent anchors + FM90 ultraviolet curve), 4 = fm07 (Fitzpatrick & Massa 2007, Rv = 3.1 only).
 * All PARITY UNPINNED (literature formulas).  Laws 3 and 4 synchronise the stream (a small table is uploaded). */
int sf_extinct(const double* d_wave, int n, const double* d_flux, int rows, double Av, double Rv, int law,
               double* d_out, void* stream);

/* scipy.linalg.cho_factor call site Starfish/models/spectrum_model.py:400 (LAPACK dpotrf).
 * In-place batched lower Cholesky of `batch` matrices, matrix b at d_A + b*stride (doubles),
 * n must be a multiple of 64 (callers pad with an identity block), row stride lda.
 * d_work: sf_potrf_workspace_bytes(n, batch).  d_info[batch]: 0, or the 1-based index of the first non-positive pivot. */
int sf_potrf_batch(double* d_A, int n, int lda, int64_t stride, int batch, int* d_info,
                   void* d_work, size_t work_bytes, void* stream);
size_t sf_potrf_workspace_bytes(int n, int batch);

/* Starfish/models/spectrum_model.py:401-404: logdet = 2 sum log L_ii and sqmah = R^T C^-1 R
 * = |L^-1 R|^2 (one forward substitution; the reference's cho_solve does two).
 * d_L as left by sf_potrf_batch; d_R: batch x ldr (ldr >= n, padded entries zero);
 * d_work: sf_potrf_workspace_bytes(n, batch) (only touched when n*8 bytes exceed the LDS). */
int sf_logdet_sqmah_batch(const double* d_L, int n, int lda, int64_t stride, int batch,
                          const double* d_R, int ldr, void* d_work, size_t work_bytes,
                          double* d_logdet, double* d_sqmah, void* stream);

/* ------------------------------------------------------------------------------------------
 * Per-order context: static data resident in HBM for the whole chain
 * ---------------------------------------------------------------------------------------- */
typedef struct sf_ctx sf_ctx;

/* All pointers are HOST pointers; everything is copied.  Mirrors what
 * SpectrumModel.__init__ (Starfish/models/spectrum_model.py:126-181) and Emulator.__init__
 * (Starfish/emulator/emulator.py:69-131) hold. */
typedef struct sf_order_desc {
    int32_t n;       /* masked data pixels N                     (spectrum.py:41-58)        */
    int32_t nf;      /* length of min_dv_wave, a power of two    (utils.py:79-84)           */
    int32_t m;       /* eigenspectra                             (emulator.py:104)          */
    int32_t n_grid;  /* emulator parameter dimensions                                       */
    int32_t M;       /* library grid points                                                  */
    int32_t reserved;
    const double* wave;         /* [n]   data.wave                                           */
    const double* flux;         /* [n]   data.flux                                           */
    const double* sigma;        /* [n]   data.sigma                                          */
    const double* min_dv_wave;  /* [nf]  spectrum_model.py:151-153                           */
    const double* bulk_fluxes;  /* [(m+2)*nf] eigenspectra; flux_mean; flux_std resampled on
                                   min_dv_wave (spectrum_model.py:154-156)                   */
    const double* grid_points;  /* [M*n_grid]                                                */
    const double* variances;    /* [m]                                                       */
    const double* lengthscales; /* [m*n_grid]                                                */
    const double* v11;          /* [(m*M)^2]  emulator.py:123-128                            */
    const double* w_hat;        /* [m*M]      component-major (emulator/_utils.py:19-21)     */
    /* optional (both or neither): the factor of the constant v11, so that callers holding many orders of
     * one emulator factor it once (LAPACK) instead of once per context.  linv = inverse of the LOWER
     * Cholesky factor of v11, row-major with zeros above the diagonal; alpha = v11^-1 w_hat.
     * NULL: computed by sf_ctx_create on the host (scalar code: fine up to m*M of a few hundred). */
    const double* linv;         /* [(m*M)^2] or NULL                                         */
    const double* alpha;        /* [m*M] or NULL                                             */
} sf_order_desc;

sf_ctx* sf_ctx_create(const sf_order_desc* desc, int device, int* err);
void sf_ctx_destroy(sf_ctx* ctx);
int sf_ctx_npad(const sf_ctx* ctx); /* N rounded up to the Cholesky leaf (64)              */
int sf_ctx_lda(const sf_ctx* ctx);  /* row stride used for the covariance matrices          */

/* Which optional parameters the model carries (the `"x" in self.params` tests of
 * SpectrumModel.__call__, spectrum_model.py:290-363) and the layout of one parameter row:
 *   [0] vsini  [1] vz  [2] log_scale  [3] norm factor (1 unless norm=True)
 *   [4] global log_amp  [5] global log_ls
 *   [6 .. 6+n_grid)             emulator grid parameters
 *   [.. +n_cheb)                c1, c2, ...   (c0 == 1, spectrum_model.py:301-304)
 *   [.. +3*n_local)             (mu, log_amp, log_sigma) per local kernel
 *   [.. +1)                     Av, only when has_av (extinct(), Starfish/transforms.py:161-206 as
 *                               called at spectrum_model.py:298-299: law ccm89, Rv = 3.1)
 * Unused slots are ignored.  Row stride = sf_param_stride(). */
typedef struct sf_model_desc {
    int32_t has_vsini;
    int32_t has_vz;
    int32_t has_log_scale; /* 0: renormalise by the integrated-flux ratio (spectrum_model.py:322-326) */
    int32_t has_global;
    int32_t n_local;
    int32_t n_cheb;
    int32_t use_sigma_w; /* 0 (default, parity): C_emu = X^T Sigma_w^-1 X as the code does
                            (spectrum_model.py:334-335); 1: X^T Sigma_w X as the paper says  */
    int32_t has_av;      /* 1: multiply every row by 10^(-0.4 A_lambda), A_lambda from the published
                            CCM89 law (Cardelli, Clayton & Mathis 1989) with Rv = 3.1.  PARITY
                            UNPINNED: the reference delegates this to the third-party `extinction`
                            C extension, which is not available to generate vectors from.          */
} sf_model_desc;

int sf_param_stride(const sf_ctx* ctx, const sf_model_desc* model);

/* scratch needed by the batched calls below for B walkers (covariance matrices included) */
size_t sf_workspace_bytes(const sf_ctx* ctx, const sf_model_desc* model, int B);

/* Emulator.__call__ (Starfish/emulator/emulator.py:330-394) for B parameter rows:
 * d_mu[B*m], d_cov[B*m*m].  d_params: B x stride rows laid out as above. */
int sf_emulator_query_batch(sf_ctx* ctx, const sf_model_desc* model, int B,
                            const double* d_params, double* d_mu, double* d_cov, int* d_info,
                            void* d_work, size_t work_bytes, void* stream);

/* Emulator.__call__ with several parameter rows (emulator.py:382-389): the JOINT conditional of the
 * B*m weights, component-major (index i*B + b): d_mu[B*m], d_cov[(B*m)^2]. */
int sf_emulator_joint_batch(sf_ctx* ctx, const sf_model_desc* model, int B, const double* d_params,
                            double* d_mu, double* d_cov, int* d_info, void* d_work,
                            size_t work_bytes, void* stream);

/* Transform chain of SpectrumModel.__call__ (spectrum_model.py:287-332): rotational_broaden,
 * doppler_shift, resample, chebyshev_correct, eigenspectrum reconstruction and (re)scaling.
 * Outputs: d_flux[B*n] scaled model flux, d_X[B*m*n] scaled eig*std rows, d_resid[B*n] =
 * flux - data.flux, d_log_scale[B] (the `_log_scale` attribute).  Any output may be NULL. */
int sf_transform_batch(sf_ctx* ctx, const sf_model_desc* model, int B, const double* d_params,
                       double* d_flux, double* d_X, double* d_resid, double* d_log_scale,
                       int* d_info, void* d_work, size_t work_bytes, void* stream);

/* SpectrumModel.__call__ (spectrum_model.py:277-365): d_flux[B*n], d_cov[B*n*n] (full symmetric
 * matrices, WITHOUT the 1e-10 jitter, exactly what the reference returns). */
int sf_forward_batch(sf_ctx* ctx, const sf_model_desc* model, int B, const double* d_params,
                     double* d_flux, double* d_cov, double* d_log_scale, int* d_info,
                     void* d_work, size_t work_bytes, void* stream);

/* The fused covariance fill on its own: C = X^T Sigma_w^-1 X (spectrum_model.py:334-338) + sigma^2 on
 * the diagonal (:338) + global Matern-3/2 kernel + local Gaussian kernels (models/kernels.py:7-81), and -- with
 * add_jitter -- the 1e-10 the likelihood adds before factorising (spectrum_model.py:399), in the reference's order of
 * additions, written in ONE pass into the caller's array: matrix b at d_cov + b * stride, row stride ld >= n.
 * lower_only: tiles above the diagonal are skipped (entries above the diagonal INSIDE a diagonal tile may be written; a
 * factorisation never reads them).  (sf_forward_batch = this with ld = n, both triangles, no
 * jitter, plus the flux; the likelihood entry points run the same kernel on the workspace matrices.) */
int sf_cov_fill_batch(sf_ctx* ctx, const sf_model_desc* model, int B, const double* d_params, double* d_cov, int ld,
                      int64_t stride, int lower_only, int add_jitter, int* d_info, void* d_work, size_t work_bytes,
                      void* stream);

/* SpectrumModel.log_likelihood without the prior term (spectrum_model.py:397-405):
 * d_lnl[B] = -(logdet + sqmah)/2; optional d_logdet[B], d_sqmah[B], d_resid[B*n],
 * d_log_scale[B].  Items with d_info[b] != 0 get lnl = -inf. */
int sf_loglike_batch(sf_ctx* ctx, const sf_model_desc* model, int B, const double* d_params,
                     double* d_lnl, double* d_logdet, double* d_sqmah, double* d_resid,
                     double* d_log_scale, int* d_info, void* d_work, size_t work_bytes,
                     void* stream);

/* ---- multi-order batches (reference: the multi-order container
 * Starfish/spectrum.py:96-115, orders independent docs/intro.rst:71-73, EchelleModel stub
 * Starfish/models/echelle_model.py:1-2) -------------------------------------------------------------
 * Evaluates the (order x walker) units of several orders in ONE enqueue: segment i is an order
 * context with its own B_i parameter rows; every order runs its transform chain and covariance fill
 * into its slice of a common covariance array (orders shorter than the longest one of the group are
 * padded with an identity block, which changes neither logdet nor the quadratic form) and all
 * sum(B_i) matrices share one batched Cholesky.  Outputs are concatenated in segment order
 * (unit = sum_{j<i} B_j + b); the caller sums over orders.  All contexts must live on the same
 * device and agree in the number of eigenspectra and grid dimensions; one model descriptor describes
 * the parameter rows of every segment.  Same values as nseg calls of sf_loglike_batch. */
typedef struct sf_segment {
    sf_ctx* ctx;
    const double* d_params; /* DEVICE: B x sf_param_stride() rows of this order */
    int32_t B;
    int32_t reserved;
} sf_segment;
size_t sf_multi_workspace_bytes(const sf_segment* segs, int nseg, const sf_model_desc* model);
int sf_loglike_multi_batch(const sf_segment* segs, int nseg, const sf_model_desc* model,
                           double* d_lnl, double* d_logdet, double* d_sqmah, double* d_log_scale,
                           int* d_info, void* d_work, size_t work_bytes, void* stream);

/* ---- structure-exploiting solver -----------------------------------
 * Same value as sf_loglike_batch, computed without ever forming the N x N matrix: the covariance of
 * spectrum_model.py:334-363 is  C = Bd + Y^T Y  with Bd = sigma^2 + K_global + K_local + jitter
 * banded (the `r <= r0` masks of models/kernels.py:33,78 give it a half-width of ~24 ls/dv pixels)
 * and Y^T Y = X^T Sigma_w^-1 X of rank m, so a banded Cholesky plus the m x m Woodbury capacitance
 * matrix give logdet C and R^T C^-1 R in O(N W^2) flops.  `halfwidth` is the caller's bound W on
 * max |i-j| over the non-zero entries of Bd (in pixels, over the whole batch); walkers whose
 * support is wider get info = SF_INFO_BANDWIDTH and lnl = -inf and must be re-run through
 * sf_loglike_batch.  Requires a strictly increasing wavelength grid and
 * halfwidth <= sf_banded_max_halfwidth(ctx).  Up to sf_banded_window_halfwidth(ctx) (144 px for
 * m <= 15) the band is swept through an LDS-resident window; wider bands are factorised as bordered
 * band matrices on the 128 x 128 tile kernels of the dense factorisation with K loops limited to the
 * band -- their workspace is as large as the dense path's (cost grows with the half-width: group
 * walkers by width).  Results
 * agree with the dense path to rounding (different summation order), not bit for bit. */
int sf_banded_max_halfwidth(const sf_ctx* ctx);
int sf_banded_window_halfwidth(const sf_ctx* ctx);
size_t sf_banded_workspace_bytes(const sf_ctx* ctx, const sf_model_desc* model, int B, int halfwidth);
int sf_loglike_banded_batch(sf_ctx* ctx, const sf_model_desc* model, int B, const double* d_params,
                            int halfwidth, double* d_lnl, double* d_logdet, double* d_sqmah,
                            double* d_resid, double* d_log_scale, int* d_info, void* d_work,
                            size_t work_bytes, void* stream);

/* Stand-alone banded kernel: for `batch` symmetric positive definite band matrices in lower band
 * storage d_band[b*stride + i*ldb + d] = A[i][i-d] (0 <= d <= halfwidth) and nrhs (<= 48) right-hand
 * sides d_rhs[b*rhs_stride + r*ldr + i], returns d_logdet[b] = log det A and
 * d_gram[b][r][c] = rhs_r^T A^-1 rhs_c (nrhs x nrhs, row-major).  d_info[b] (zeroed by the call) =
 * 1-based column of the first non-positive pivot.  (scipy.linalg.cholesky_banded + solves.) */
int sf_band_logdet_gram_batch(const double* d_band, int n, int halfwidth, int ldb, int64_t stride,
                              int batch, const double* d_rhs, int nrhs, int ldr, int64_t rhs_stride,
                              double* d_logdet, double* d_gram, int* d_info, void* stream);

/* Timing hooks for bench.py (process-global, not thread-safe): when enabled, the batched calls
 * record HIP events on the caller's stream around each stage and around every MFMA update launch.
 * sf_profile_read synchronises those events, returns the milliseconds accumulated since the last
 * read in ms_by_stage[6] = {transforms, fill, k_gemm_nt launches (overlapping launches of the two
 * Cholesky streams merged: union of their intervals), whole potrf, solve, k_gemm_nt launches (plain
 * sum of launch durations)}, the algorithmic flops and launch count of k_gemm_nt, and the number
 * of sf_loglike_batch calls; then it resets the counters. */
int sf_profile_enable(int on);
int sf_profile_read(double* ms_by_stage, double* gemm_flops, long* gemm_launches, long* calls);

/* Emulator training likelihood (Starfish/emulator/emulator.py:602-619): the matrix
 * v11 = iPhiPhi / lambda_xi + blockdiag_c(variance_c RBF_c(grid, grid)) (emulator.py:126-128, kernels.py:5-49) built on
 * the device into the padded layout of sf_potrf_batch: d_A[npad][lda], identity block from n = m M to npad.
 * d_grid[M*P], d_hyper = {lambda_xi, variances[m], lengthscales[m*P]}, d_iphiphi[n*n].  Emulator.train calls it once per
 * objective evaluation, followed by sf_potrf_batch + sf_logdet_sqmah_batch with the right-hand side w_hat. */
int sf_emulator_v11_build(const double* d_grid, int M, int P, int m, const double* d_hyper, const double* d_iphiphi,
                          double* d_A, int npad, int lda, void* stream);

/* Process-global switch of the persistent-kernel ("dataflow") Cholesky sequence: enable = 0 makes every later factorisation
 * take a launch sequence (kernels without waits inside), 1 restores the default choice, < 0 only queries.  Returns the
 * previous setting.  The recovery path after SF_INFO_INTERNAL.
 * The persistent kernel's workgroups wait for each other inside ONE launch: it assumes the process has the device to itself
 * (one process per GPU).  Several processes oversubscribing one device can keep each other's workgroups from becoming
 * running; the launch is then given up FAST -- as soon as no task of it has completed for 25 ms while a workgroup was waiting
 * (the longest task of the largest matrix the kernel takes, N = 16384, runs ~5 ms), or after 4 s inside one wait --
 * (SF_INFO_INTERNAL for the batch) and the caller falls back: a shared device costs the first call ~25 ms + one factorisation
 * on the launch sequences.  A process that knows it shares its device should switch the sequence off up front. */
int sf_persistent_potrf(int enable);

/* The process's record of persistent-kernel launches, for the caller's warning and for tests (host memory, no
 * synchronisation: read it after the stream that carried the aborted call has been synchronised):
 * h_out8 = {aborted launches so far, reason of the last abort (1 = a wait reached its 4-s bound, 2 = no task completed for
 * 25 ms), workgroups of that launch that had started, its grid size (fewer started than launched = the grid was not
 * co-resident: the device is shared or partitioned), 100 MHz ticks the reporting wait had lasted, tasks the launch had
 * completed, persistent launches enqueued by this process so far, 1 if the sequence is enabled}. */
int sf_persistent_potrf_status(long long* h_out8);

/* Tuning / test aid (process-global): the batched Cholesky has three launch sequences -- the fused panel kernel
 * (128-column panels, two workgroups per CU; medium batches), the unfused one (256-column panels, separate panel-solve and
 * diagonal-update launches; kept selectable) and the wide one (pairs of panels, one 16-wave workgroup per CU keeps a 128 x 256 tile: a third less HBM
 * traffic; taken when batch x 128-row slabs >= 3400 and the matrices have 2048 or more rows).
 * mode -1 = choose by batch and matrix size (default), 0 = always fused, 1 = always unfused, 2 = always wide,
 * 3 = wide for the first half of the panels, then fused (test aid: exercises the hand-over between the two),
 * 4 = dataflow: the whole factorisation as ONE persistent launch whose workgroups draw tasks and wait on exactly the tasks
 * they depend on (the default while batch x 128-column panels <= 2048, batch <= 128 and N <= 8192 -- every scalar evaluation,
 * the half-ensembles of a sampler, the per-GPU batches of a strong split; forced, it takes up to 128 panels, N = 16384, beyond
 * that the fused sequence runs instead).
 * The fused and wide sequences factorise matrices of 64 mod 128 rows in a frame shifted by 64 virtual identity rows
 * (addressing only: nothing moves in memory, the caller's layout and the pivot index reported in d_info are unchanged).
 * Same results to rounding. */
int sf_debug_cholesky_sequence(int mode);

/* Tuning aid: one wave spins for `wall_ticks_100mhz` ticks of the 100 MHz wall clock on `stream` and
 * writes {shader-clock ticks, wall ticks} to d_out2[2] -> sustained shader clock under load. */
int sf_debug_clock_probe(long long* d_out2, long long wall_ticks_100mhz, void* stream);

/* Measurement aid (bench.py's fill leg): a plain streaming write of `count` doubles (16-byte stores, one pass, nothing
 * read) -- the HBM write rate this box sustains, next to which the write-only covariance fill is priced. */
int sf_debug_stream_write(double* d_dst, size_t count, double value, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* STARFISH_AMD_H */
