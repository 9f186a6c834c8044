// Argument blocks of the transform / emulator kernels (internal).
#pragma once
#include "sf_common.h"

#define SF_KB 5      // sub/super-diagonals stored for the quintic collocation LU
#define SF_MAX_M 32  // eigenspectra handled by the per-pixel rank-m factor kernel
#define SF_IW 64     // half-width of the truncated inverse of the collocation matrix (decay ~0.43^k: < 1e-23)

struct sf_broaden_args {
    const double* in;      // rows x nf real input (free functions) or NULL
    const double2* spec;   // precomputed half spectra rows x (nf/2+1) (context path) when in == NULL
    int B, rows, nf;
    const double2* tw;     // exp(-2 pi i k / nf), k < nf/2
    double dv;
    int kind;              // 0 none, 1 rotational, 2 instrumental
    const double* params;  // per-item parameter rows (NULL -> scalar_param)
    int pstride, poff;
    double scalar_param;
    double* out;
    int64_t ob, orow, oelem;  // strides of the output: item, row, element
    double2* gscratch;     // B*rows*nf complex when nf does not fit the LDS
    double* mult;          // optional B x (nf/2+1) scratch (context path): per-walker multiplier table ->
                           // half-size FFT kernel
    int* info;
};
size_t sf_fft_scratch_bytes(int rows_total, int nf);
size_t sf_fft_half_scratch_bytes(int rows_total, int nf);
int sf_launch_broaden(const sf_broaden_args& a, hipStream_t s);
int sf_launch_rfft_rows(const double* in, int rows, int nf, const double2* tw, double2* spec,
                        double2* gscratch, hipStream_t s);
int sf_launch_spline_solve(double* data, int B, int rows, int64_t bstride, int64_t rstride,
                           int64_t estride, int n, const double* Lf, const double* Uf, const double* rdiag,
                           hipStream_t s);
// c = A^-1 y with the truncated (banded) inverse: y, c laid out [B][n][rows]; band[(2*SF_IW+1)][n]
int sf_launch_spline_apply(const double* y, double* c, int B, int rows, int n, const double* band, hipStream_t s);
int sf_launch_spline_eval(const double* coef, int rows, int ncoef, const double* t, const double* xq, int nq,
                          double* out, hipStream_t s);

struct sf_eval_args {
    const double* wave;    // [n] data wavelengths
    const double* knots;   // [nf + 6]
    const double* coef;    // [B?][nf][m+2] spline coefficients, row index fastest
    int coef_batched;      // 0: one static coefficient set shared by all walkers
    const double* params;
    const double* mu;      // [B][m] emulator weights
    double* X;             // [B][m][ldx]  unscaled eig*std rows
    double* flux;          // [B][ldx]     unscaled reconstruction
    const int* info;
    int n, nf, m, ldx, pstride, has_vz, n_cheb, off_cheb;
    int has_av, off_av;
    double wave_max;
};
int sf_launch_eval_rows(const sf_eval_args& a, int B, hipStream_t s);

struct sf_scale_args {
    const double* wave;
    const double* dflux;
    const double* flux;    // [B][ldx]
    const double* params;
    double* scale;         // [B]
    double* log_scale_out; // [B] or NULL
    int n, ldx, pstride, has_log_scale;
};
int sf_launch_scale(const sf_scale_args& a, int B, hipStream_t s);

struct sf_resid_args {
    const double* dflux;
    const double* flux;    // [B][ldx] unscaled
    const double* X;       // [B][m][ldx] unscaled
    const double* scale;   // [B]
    const double* Lw;      // [B][m][m] lower Cholesky factor of Sigma_w
    const int* info;
    double* resid;         // [B][ldx] or NULL
    double* Y;             // [B][mpad][ldy] or NULL
    double* flux_out;      // [B][n] or NULL
    double* X_out;         // [B][m][n] or NULL
    int n, m, mpad, ldx, ldy, use_sigma_w;
};
int sf_launch_resid_y(const sf_resid_args& a, int B, hipStream_t s);
// eval_rows + scale (given log_scale: spectrum_model.py:316-318) + resid_y in one pass: X and the flux never go through memory
// (e.X / e.flux / r.X / r.flux / r.scale are not used; `scale_out` [B] and `log_scale_out` [B] or NULL are written)
int sf_launch_eval_resid_y(const sf_eval_args& e, const sf_resid_args& r, double* scale_out, double* log_scale_out, int B,
                           hipStream_t s);

int sf_launch_extinct_rows(const double* wave, int n, const double* flux, int rows, double Av, double Rv, int law,
                           double* out, hipStream_t s);  // law: 0 ccm89, 1 odonnell94, 2 calzetti00
int sf_launch_extinct_spline_rows(const double* wave, int n, const double* flux, int rows, double Av, double Rv,
                                  const double* d_table, double* out, hipStream_t s);  // law 3 / 4: table built by sf_extinct
int sf_launch_cheb_rows(const double* wave, int n, double wave_max, const double* flux, int rows,
                        const double* d_coeffs, int ncoef, double* out, hipStream_t s);

struct sf_emu_args {
    const double* params;
    int pstride, off_grid;
    int m, M, P;
    const double* grid;          // [M][P]
    const double* variances;     // [m]
    const double* lengthscales;  // [m][P]
    const double* gmin;          // [P]
    const double* gmax;          // [P]
    const double* alpha;         // [mM]   v11^-1 w_hat
    const double* LinvT;         // [mM][mM] TRANSPOSE of the inverse of the lower Cholesky factor of v11
    double* zscratch;            // [B][mM][m]
    double* kbuf;                // [B][mM] v12 blocks of every walker
    double* mu;                  // [B][m]
    double* cov;                 // [B][m][m] or NULL
    double* Lw;                  // [B][m][m] or NULL
    int* info;
};
int sf_launch_emulator(const sf_emu_args& a, int B, hipStream_t s);
int sf_launch_emu_joint(const sf_emu_args& a, int B, const double* mu_pts, double* mu, double* cov, hipStream_t s);
int sf_launch_finish(int B, const double* logdet, const double* sqmah, const int* info, const int* info2,
                     double* lnl, int* info_out, hipStream_t s);
int sf_launch_v11_build(const double* grid, int M, int P, int m, const double* hyper, const double* iphiphi, double* A, int npad,
                        int lda, hipStream_t s);
