// Fused covariance fill for gfx950: one pass writes
//     C[b] = Y_b^T Y_b  +  diag(sigma^2)  +  K_global  +  sum_k K_local,k  (+ 1e-10 I)
// replacing the dense temporaries of the reference:
//   Starfish/models/spectrum_model.py:334-363  (X^T Sigma_w^-1 X, fill_diagonal, cov += ...)
//   Starfish/models/kernels.py:7-41            (global_covariance_matrix)
//   Starfish/models/kernels.py:44-81           (local_covariance_matrix)
//   Starfish/models/spectrum_model.py:399      (jitter)
// The rank-m emulator term runs on v_mfma_f64_16x16x4_f64 (the MFMA row index is mapped to matrix
// COLUMNS so that every lane owns 4 consecutive columns -> 32-byte stores, full 128-B lines per
// 4 lanes); the banded/patch kernels are evaluated only in wave sub-tiles that intersect their
// support.  Compiled with -ffp-contract=off: element formulas keep the reference's operation order.
#include "sf_common.h"

#define FT 64  // tile edge per workgroup (4 waves, 32 x 32 each)

__device__ __forceinline__ int sf_xcd_remap_f(int bid, int nblk) {
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

// kernels.py:27-40 with wx = wave[col], wy = wave[row]
__device__ __forceinline__ double sf_matern_elem(double w_row, double w_col, double amp, double ls,
                                                 double r0) {
    const double r = SF_C_KMS / 2 * fabs((w_col - w_row) / (w_col + w_row));
    if (!(r <= r0)) return 0.0;
    const double taper = 0.5 + 0.5 * cos(M_PI * r / r0);
    const double s3 = 1.7320508075688772;  // numpy.sqrt(3)
    return taper * amp * (1 + s3 * r / ls) * exp(-s3 * r / ls);
}

// kernels.py:68-80 with x = met[col], y = met[row]
__device__ __forceinline__ double sf_local_elem(double d_row, double d_col, double amp, double sigma,
                                                double r0) {
    const double r_tap = fmax(d_col, d_row);
    if (!(r_tap <= r0)) return 0.0;
    const double r2 = d_col * d_col + d_row * d_row;
    const double taper = 0.5 + 0.5 * cos(M_PI * r_tap / r0);
    return taper * amp * exp(-0.5 * r2 / (sigma * sigma));
}

__device__ __forceinline__ double sf_local_metric(double w, double mu) {
    return SF_C_KMS / mu * fabs(w - mu);  // kernels.py:69
}

// The tile bodies evaluate the two element formulas for 16 entries per lane: inlined (fp64 cos with its argument reduction, exp:
// ~1 KB of code each) the structured tile body is 66-70 KB of straight-line code per tile -- more than the 64 KB instruction
// cache of a CU pair holds.  As real calls (-DSF_FILL_CALL_ELEMS) the kernels are 15 KB, 143 instead of 163 VGPRs, same bits
// -- and no faster where it counts (round 6, same box: dense fill of cfg 2 3.39 -> 3.43 ms, N = 3000 with ld = N 2.40 ->
// 2.27, ld = 3008 2.15 -> 2.09, the likelihood's tile-list fill 0.495 both: profiles/r06_d_fill_called_elements_ab.txt):
// sequential code streams through the instruction prefetch, the kernel is bound by the latency of its fp64 chains.  Inlined.
#ifndef SF_FILL_CALL_ELEMS
#define SF_ELEM_CALL __forceinline__
#else
#define SF_ELEM_CALL __attribute__((noinline))
#endif
__device__ SF_ELEM_CALL double sf_matern_elem_t(double w_row, double w_col, double amp, double ls, double r0) {
    return sf_matern_elem(w_row, w_col, amp, ls, r0);
}
__device__ SF_ELEM_CALL double sf_local_elem_t(double d_row, double d_col, double amp, double sigma, double r0) {
    return sf_local_elem(d_row, d_col, amp, sigma, r0);
}

#define SF_MAX_LOCAL 32

// Which 128 x 128 tiles of the lower triangle carry anything besides the rank-m term (diagonal
// SF_NB blocks: sigma^2 / jitter / identity padding; Matern band; local patches)?  Only those are
// materialised for the factorisation; the MFMA update kernel generates the others from Y on the fly.
__global__ __launch_bounds__(256) void k_tile_map(sf_fill_args a) {
    const int nt = a.nt128;
    const int e = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (e >= nt * nt) return;
    const int tm = e / nt, tn = e - tm * nt;
    unsigned char flag = 0;
    if (tn <= tm) {
        // tile (tm, tn) of the factorisation's frame = rows 128 tm - fp .. of the matrix (a.fp leading virtual rows)
        const int vr = tm * 128, vc = tn * 128;
        const int rlo = max(vr - a.fp, 0), clo = max(vc - a.fp, 0);
        if (vr / SF_NB == vc / SF_NB || rlo >= a.n) {
            flag = 1;  // diagonal block (or pure padding rows)
        } else {
            const int rhi = min(vr - a.fp + 127, a.n - 1), chi = min(vc - a.fp + 127, a.n - 1);
            const double* __restrict__ P = a.params + (int64_t)b * a.pstride;
            if (a.has_global) {
                if (!a.monotonic) flag = 1;
                else {
                    const double wr = a.wave[rlo], wc = a.wave[chi];  // closest pair: rows are below cols
                    const double rmin = SF_C_KMS / 2 * fabs((wc - wr) / (wc + wr));
                    if (rmin <= 6 * exp(P[a.off_global + 1]) * (1 + 1e-9)) flag = 1;
                }
            }
            for (int k = 0; k < a.n_local && !flag; ++k) {
                if (!a.monotonic) { flag = 1; break; }
                const double mu = P[a.off_local + 3 * k];
                const double r0 = 4 * exp(P[a.off_local + 3 * k + 2]);
                auto dmin = [&](int lo, int hi) {
                    const double wl = a.wave[lo], wh = a.wave[hi];
                    if (wl <= mu && mu <= wh) return 0.0;
                    return fmin(sf_local_metric(wl, mu), sf_local_metric(wh, mu));
                };
                if (dmin(rlo, rhi) <= r0 * (1 + 1e-9) && dmin(clo, chi) <= r0 * (1 + 1e-9)) flag = 1;
            }
        }
    }
    a.tilemap[(int64_t)b * nt * nt + e] = flag;
    if (flag && a.tilelist) {
        const int idx = atomicAdd(&a.tilecount[b], 1);  // (the order of the list does not matter: tiles are independent)
        if (idx < a.list_cap) a.tilelist[(int64_t)b * a.list_cap + idx] = (unsigned short)((tm << 8) | tn);
    }
}

// Which structured kernels can reach the block rows [rlo, rhi] x columns [clo, chi] (indices < n)?  Conservative: the
// closest (row, column) pair in wavelength against the kernel's cut-off radius (kernels.py:29,73), with a 1e-9 margin.
// `g_r0` = 6 exp(log_ls) of the global kernel (unused without one).  Shared by the fill tiles (32 x 32 wave sub-tiles)
// and the 64 x 64 support map of the dense fill: a tile the map leaves out has no sub-tile that is reached.
__device__ __forceinline__ void sf_block_support(const sf_fill_args& a, const double* __restrict__ P, int rlo, int rhi,
                                                 int clo, int chi, double g_r0, bool& do_glob, unsigned& lmask) {
    const bool on_diag = !(rlo > chi || clo > rhi);
    do_glob = false;
    lmask = 0;
    if (a.has_global) {
        do_glob = true;
        if (a.monotonic && !on_diag) {
            // closest (row, col) pair of the block in wavelength
            double wr, wc;
            if (rlo > chi) { wr = a.wave[rlo]; wc = a.wave[chi]; }
            else { wr = a.wave[rhi]; wc = a.wave[clo]; }
            const double rmin = SF_C_KMS / 2 * fabs((wc - wr) / (wc + wr));
            do_glob = rmin <= g_r0 * (1 + 1e-9);
        }
    }
    for (int k = 0; k < a.n_local; ++k) {
        bool hit = true;
        if (a.monotonic) {
            const double mu = P[a.off_local + 3 * k];
            const double r0 = 4 * exp(P[a.off_local + 3 * k + 2]);  // kernels.py:73
            auto dmin = [&](int lo, int hi) {  // smallest metric over an index range
                const double wl = a.wave[lo], wh = a.wave[hi];
                if (wl <= mu && mu <= wh) return 0.0;
                return fmin(sf_local_metric(wl, mu), sf_local_metric(wh, mu));
            };
            hit = (dmin(rlo, rhi) <= r0 * (1 + 1e-9)) && (dmin(clo, chi) <= r0 * (1 + 1e-9));
        }
        if (hit) lmask |= 1u << k;
    }
}

// Every stored tile in ONE pass: rank-m term on MFMA + sigma^2 on the diagonal + identity padding, and
// (BAND) in the 32 x 32 sub-tiles that intersect the support of a structured kernel: + K_global, then
// + (0 + K_local,0 + K_local,1 ...), then the jitter -- the reference's order of additions
// (spectrum_model.py:338, 348, 353-363, 399).  Write-only: the pass is HBM-write bound (a separate band
// pass used to read-modify-write the same tiles: 1.15 -> 0.5 ms at cfg 2).
// Second half of a tile: the accumulators hold the rank-m term of the wave's 32 x 32 sub-tile at (R0, C0);
// acc[ti][tj] element r = (row R0+ti*16+gam, column C0+tj*16+4q+r).
template <bool BAND>
__device__ __forceinline__ void sf_tile_finish(const sf_fill_args& a, int b, int R0, int C0, const sf_d4 (&acc)[2][2],
                                               bool mirror = false) {
    const int lane = threadIdx.x & 63;
    const int gam = lane & 15, q = lane >> 4;
    const int nout = sf_fill_extent(a);
    double* __restrict__ Cb = a.C + (int64_t)b * a.stride;
    // which structured kernels reach this 32 x 32 sub-tile (wave-uniform)
    bool do_glob = false;
    double g_amp = 0, g_ls = 1, g_r0 = 0;
    unsigned lmask = 0;
    const double* __restrict__ P = a.params + (int64_t)b * a.pstride;
    if (BAND && R0 < a.n && C0 < a.n) {
        if (a.has_global) {
            g_amp = exp(P[a.off_global]);      // spectrum_model.py:343
            g_ls = exp(P[a.off_global + 1]);   // spectrum_model.py:344
            g_r0 = 6 * g_ls;                   // kernels.py:29
        }
        sf_block_support(a, P, R0, min(R0 + 31, a.n - 1), C0, min(C0 + 31, a.n - 1), g_r0, do_glob, lmask);
    }
    const bool structured = do_glob || lmask;

    const bool vec_ok = ((a.lda | a.stride) & 1) == 0;  // 16-byte stores need even row and matrix strides
#pragma unroll
    for (int ti = 0; ti < 2; ++ti) {
        const int row = R0 + ti * 16 + gam;
        if (row >= nout) continue;
        const double w_row = (BAND && row < a.n) ? a.wave[row] : 1.0;
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            const int col0 = C0 + tj * 16 + 4 * q;
            if (col0 >= nout) continue;
            double v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = col0 + r;
                double val = acc[ti][tj][r];
                if (row < a.n && col < a.n) {
                    if (row == col) {
                        const double sg = a.sigma[row];
                        val = val + sg * sg;                                     // spectrum_model.py:338
                    }
                } else {
                    val = (row == col) ? 1.0 : 0.0;  // identity padding up to the Cholesky leaf
                }
                v[r] = val;
            }
            if (BAND && structured && row < a.n) {
                double w_col[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) w_col[r] = (col0 + r < a.n) ? a.wave[col0 + r] : 1.0;
                if (do_glob && a.gtab) {  // log-uniform grid: one value per diagonal (see k_band_gtab)
                    const double* gt = a.gtab + (int64_t)b * a.n;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col0 + r < a.n) v[r] = v[r] + gt[abs(row - (col0 + r))];
                } else if (do_glob) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col0 + r < a.n) v[r] = v[r] + sf_matern_elem_t(w_row, w_col[r], g_amp, g_ls, g_r0);
                }
                if (lmask) {
                    double loc[4] = {0.0, 0.0, 0.0, 0.0};
                    for (int k = 0; k < a.n_local; ++k) {
                        if (!((lmask >> k) & 1)) continue;
                        const double mu = P[a.off_local + 3 * k];
                        const double amp = exp(P[a.off_local + 3 * k + 1]);  // spectrum_model.py:356
                        const double sig = exp(P[a.off_local + 3 * k + 2]);  // spectrum_model.py:357
                        const double d_row = sf_local_metric(w_row, mu);
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            loc[r] = loc[r] + sf_local_elem_t(d_row, sf_local_metric(w_col[r], mu), amp, sig, 4 * sig);
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col0 + r < a.n) v[r] = v[r] + loc[r];
                }
            }
            if (a.add_jitter && row < a.n) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (col0 + r == row) v[r] = v[r] + SF_JITTER;  // spectrum_model.py:399
            }
            double* dst = Cb + (int64_t)row * a.lda + col0;
            if (vec_ok && col0 + 3 < nout) {
                *(double2*)dst = make_double2(v[0], v[1]);
                *(double2*)(dst + 2) = make_double2(v[2], v[3]);
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (col0 + r < nout) dst[r] = v[r];
            }
            if (mirror) {
                // C is symmetric bit for bit (every term's formula is symmetric in (row, column), the MFMA sums over k in
                // the same order): the dense fill evaluates the structured tiles below the diagonal only and writes
                // each one a second time transposed -- 16 lanes cover 128 contiguous bytes of a row of the mirror tile
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (col0 + r < nout) Cb[(int64_t)(col0 + r) * a.lda + row] = v[r];
            }
        }
    }
}

template <bool BAND>
__device__ __forceinline__ void sf_fill_tile(const sf_fill_args& a, int b, int tm, int tn) {
    if (a.lower_only && tn > tm) return;
    if (a.tilemap) {  // (tm, tn) count 64-row tiles of the MATRIX; the map is indexed in the factorisation's frame
        const int fs = a.fp >> 6;
        if (!a.tilemap[(int64_t)b * a.nt128 * a.nt128 + ((tm + fs) >> 1) * a.nt128 + ((tn + fs) >> 1)]) return;
    }

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int R0 = tm * FT + (w >> 1) * 32, C0 = tn * FT + (w & 1) * 32;
    const int nout = sf_fill_extent(a);  // extent of the stored matrix
    if (R0 >= nout || C0 >= nout) return;
    if (a.lower_only && C0 > R0 + 31) return;
    const int gam = lane & 15, q = lane >> 4;

    const double* __restrict__ Yb = a.Y + (int64_t)b * a.mpad * a.ldy;
    double* __restrict__ Cb = a.C + (int64_t)b * a.stride;

    // acc[ti][tj] element (row R0+ti*16+gam, cols C0+tj*16+4q+r)
    sf_d4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (sf_d4){0.0, 0.0, 0.0, 0.0};
    const int colperm = 4 * (gam & 3) + (gam >> 2);
    for (int kk = 0; kk < a.mpad; kk += 4) {
        const double* yk = Yb + (int64_t)(kk + q) * a.ldy;
        double brow[2], acol[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            brow[i] = yk[R0 + i * 16 + gam];
            acol[i] = yk[C0 + i * 16 + colperm];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(acol[j], brow[i], acc[i][j], 0, 0, 0);
    }

    sf_tile_finish<BAND>(a, b, R0, C0, acc);
}

template <bool BAND>
__global__ __launch_bounds__(256, BAND ? 2 : 4) void k_fill_tiles(sf_fill_args a, int nt) {
    const int id = sf_xcd_remap_f(blockIdx.x, gridDim.x);
    const int tiles = nt * nt;
    const int b = id / tiles;
    const int t = id - b * tiles;
    sf_fill_tile<BAND>(a, b, t / nt, t - (t / nt) * nt);
}
// The likelihood path: G workgroups per walker walk the walker's list of materialised 128 x 128 tiles (four 64 x 64 tiles
// each).  The one-workgroup-per-tile grid above is 524 288 workgroups at cfg 2 of which nine in ten leave at once:
// dispatch-bound (0.70 ms for 0.9 GB written).
template <bool BAND>
__global__ __launch_bounds__(256, BAND ? 2 : 4) void k_fill_tiles_list(sf_fill_args a, int G) {
    const int b = blockIdx.x / G, g = blockIdx.x - b * G;
    const int cnt = min(a.tilecount[b], a.list_cap) * 4;
    const unsigned short* __restrict__ list = a.tilelist + (int64_t)b * a.list_cap;
    const int fs = a.fp >> 6;  // the list holds tiles of the factorisation's frame: a.fp / 64 virtual 64-row tiles in front
    for (int li = g; li < cnt; li += G) {
        const int e = list[li >> 2];
        const int tm = 2 * (e >> 8) + ((li >> 1) & 1) - fs, tn = 2 * (e & 255) + (li & 1) - fs;
        if (tm >= 0 && tn >= 0) sf_fill_tile<BAND>(a, b, tm, tn);
    }
}

__global__ void k_band_gtab(sf_fill_args a, double* __restrict__ gtab, int ws);

int sf_launch_fill(const sf_fill_args& a, int B, hipStream_t s) {
    if (a.n_local > SF_MAX_LOCAL) {
        sf_set_error("at most %d local kernels are supported", SF_MAX_LOCAL);
        return SF_EINVAL;
    }
    if (a.fp != 0 && (a.fp != 64 || !a.tilemap || !a.lower_only)) {
        sf_set_error("fill: a shifted tile frame needs fp = 64, a tile map and lower_only");
        return SF_EINVAL;
    }
    const int nout = sf_fill_extent(a);
    const int nt = (nout + FT - 1) / FT;
    const long long nblk = (long long)nt * nt * B;
    if (nblk > 0x7fffffffLL) {
        sf_set_error("fill grid too large");
        return SF_EINVAL;
    }
    const bool listed = a.tilemap && a.tilelist && a.tilecount && a.lower_only;
    if (a.tilemap) {
        if (listed) SF_HIP(hipMemsetAsync(a.tilecount, 0, sizeof(int) * (size_t)B, s));
        hipLaunchKernelGGL(k_tile_map, dim3((a.nt128 * a.nt128 + 255) / 256, B), dim3(256), 0, s, a);
        SF_LAUNCH_CHECK();
    }
    const int structured = a.has_global || a.n_local > 0;
    sf_fill_args a2 = a;
    if (!(a.gtab && a.has_global && a.loguniform && a.lower_only)) a2.gtab = nullptr;
    if (a2.gtab) {
        hipLaunchKernelGGL(k_band_gtab, dim3((a.n + 255) / 256, B), dim3(256), 0, s, a, a2.gtab, a.n - 1);
        SF_LAUNCH_CHECK();
    }
    if (listed) {
        const int G = 64;
        if (structured) hipLaunchKernelGGL(k_fill_tiles_list<true>, dim3((unsigned)B * G), dim3(256), 0, s, a2, G);
        else hipLaunchKernelGGL(k_fill_tiles_list<false>, dim3((unsigned)B * G), dim3(256), 0, s, a2, G);
    } else if (structured) hipLaunchKernelGGL(k_fill_tiles<true>, dim3((unsigned)nblk), dim3(256), 0, s, a2, nt);
    else hipLaunchKernelGGL(k_fill_tiles<false>, dim3((unsigned)nblk), dim3(256), 0, s, a2, nt);
    SF_LAUNCH_CHECK();
    return SF_OK;
}

// Band storage of Bd = diag(sigma^2) + K_global + sum K_local + jitter for the structure-exploiting
// solver (sf_band.hip): band[i*ldb + d] = Bd[i][i-d], d in [0, ws).  The element formulas and their
// order of additions are those of k_fill_tiles.  Diagonals d > hw (the caller's half-width) are stored as
// zeros; the thread on diagonal hw also probes diagonal hw + 1: a non-zero there means the caller's
// half-width is too small for this walker -> info = SF_INFO_BANDWIDTH (the result would silently drop
// covariance otherwise) -- independently of how many diagonals the storage happens to hold.
// The element formulas are those of sf_matern_elem / sf_local_elem with the per-walker divisions
// hoisted into reciprocals and cos(pi x) evaluated as cospi(x) (differences ~1e-16 relative, far inside
// the 1e-10 covariance tolerance; the dense fill keeps the reference's exact operation order).
// On a log-uniform wavelength grid (lambda_i = lambda_0 e^(i delta): every synthetic order, rectified
// spectra) the metric of the global kernel depends on the offset only, (l_i - l_j)/(l_i + l_j) =
// tanh((i-j) delta/2), so K_global is one value per diagonal: tabulated here per walker from a pair in
// the middle of the order (gtab[b][d], d <= ws; the extra entry feeds the bandwidth probe).  Differences
// to the per-entry evaluation are at the level of the rounding of the grid itself (~3e-11 relative in r).
__global__ __launch_bounds__(256) void k_band_gtab(sf_fill_args a, double* __restrict__ gtab, int ws) {
    const int b = blockIdx.y, d = blockIdx.x * 256 + threadIdx.x;
    if (d > ws) return;
    const double* __restrict__ P = a.params + (int64_t)b * a.pstride;
    const double amp = exp(P[a.off_global]), ls = exp(P[a.off_global + 1]);
    const int i = min(a.n - 1, a.n / 2 + d / 2), j = i - d;
    double v = 0.0;
    if (j >= 0) {
        const double r0 = 6 * ls;
        const double r = SF_C_KMS / 2 * fabs((a.wave[j] - a.wave[i]) / (a.wave[j] + a.wave[i]));
        if (r <= r0) {
            const double t = 1.7320508075688772 / ls * r;
            v = (0.5 + 0.5 * cospi(r / r0)) * amp * (1 + t) * exp(-t);
        }
    }
    gtab[(int64_t)b * (ws + 1) + d] = v;
}

#define SF_BF_ROWS 32
// tile_wt < 0: compact band storage band[i * ldb + d].  tile_wt >= 0: the same values straight into the lower
// 128 x 128 tiles of a dense-strided array (row stride ldb) that meet the band -- element (i, i - d), d < ws =
// 128 (tile_wt + 1), as far left as the first tile column (i / 128 - tile_wt) of the row (sf_launch_potrf_band).
__global__ __launch_bounds__(256) void k_band_fill(sf_fill_args a, double* __restrict__ band, int ws, int hw, int ldb,
                                                   int64_t sband, int* __restrict__ info,
                                                   const double* __restrict__ gtab, int tile_wt) {
    // per-walker constants once per block: exp() of the hyper-parameters (spectrum_model.py:343-357)
    __shared__ double s_glob[4];                 // amp, r0, 1/r0, sqrt(3)/ls
    __shared__ double s_loc[SF_MAX_LOCAL][6];    // mu, amp, r0, 1/r0, -0.5/sigma^2, c/mu
    const int b = blockIdx.y, tid = threadIdx.x;
    const double* __restrict__ P = a.params + (int64_t)b * a.pstride;
    if (tid == 0 && a.has_global) {
        const double amp = exp(P[a.off_global]), ls = exp(P[a.off_global + 1]);
        s_glob[0] = amp;
        s_glob[1] = 6 * ls;
        s_glob[2] = 1.0 / (6 * ls);
        s_glob[3] = 1.7320508075688772 / ls;
    }
    if (tid >= 64 && tid < 64 + a.n_local) {
        const int k = tid - 64;
        const double sig = exp(P[a.off_local + 3 * k + 2]);
        s_loc[k][0] = P[a.off_local + 3 * k];
        s_loc[k][1] = exp(P[a.off_local + 3 * k + 1]);
        s_loc[k][2] = 4 * sig;
        s_loc[k][3] = 1.0 / (4 * sig);
        s_loc[k][4] = -0.5 / (sig * sig);
        s_loc[k][5] = SF_C_KMS / s_loc[k][0];
    }
    __syncthreads();
    // SF_BF_ROWS rows per block (the exp() prologue is amortised), one wave per row at a time, 64 lanes
    // along the diagonals of the row (coalesced stores, no index division)
    const int lane = tid & 63;
    const double* __restrict__ gt = gtab ? gtab + (int64_t)b * (ws + 1) : nullptr;
    for (int i = blockIdx.x * SF_BF_ROWS + (tid >> 6); i < min(a.npad, (int)(blockIdx.x + 1) * SF_BF_ROWS); i += 4) {
        const bool tiled = tile_wt >= 0;
        double* __restrict__ dst = band + (int64_t)b * sband + (int64_t)i * ldb + (tiled ? i : 0);
        const int dmax = tiled ? i - max((i >> 7) - tile_wt, 0) * 128 : ws - 1;  // last stored diagonal of this row
        const int dstep = tiled ? -1 : 1;
        if (i >= a.n) {
            for (int d = lane; d <= min(dmax, ws - 1); d += 64) dst[dstep * d] = (d == 0) ? 1.0 : 0.0;  // identity padding
            continue;
        }
        const double w_row = a.wave[i];
        auto structured = [&](int col, bool& any) {
            const double w_col = a.wave[col];
            double acc = 0.0;
            if (a.has_global && gt) {
                acc = gt[i - col];
                any = any || acc != 0.0;
            } else if (a.has_global) {
                const double r = SF_C_KMS / 2 * fabs((w_col - w_row) / (w_col + w_row));
                if (r <= s_glob[1]) {
                    const double t = s_glob[3] * r;
                    acc = (0.5 + 0.5 * cospi(r * s_glob[2])) * s_glob[0] * (1 + t) * exp(-t);
                    any = true;
                }
            }
            for (int k = 0; k < a.n_local; ++k) {
                const double mu = s_loc[k][0], cm = s_loc[k][5];
                const double d_row = cm * fabs(w_row - mu), d_col = cm * fabs(w_col - mu);
                const double r_tap = fmax(d_row, d_col);
                if (r_tap <= s_loc[k][2]) {
                    acc += (0.5 + 0.5 * cospi(r_tap * s_loc[k][3])) * s_loc[k][1] *
                           exp((d_col * d_col + d_row * d_row) * s_loc[k][4]);
                    any = true;
                }
            }
            return acc;
        };
        for (int d = lane; d <= min(dmax, ws - 1); d += 64) {
            const int j = i - d;
            double v = 0.0;
            if (j >= 0 && d <= hw) {  // diagonals past the caller's half-width are stored as zeros, never as data
                bool any = false;
                const double k = structured(j, any);
                if (d == 0) {
                    const double sg = a.sigma[i];
                    v = sg * sg;
                    v = v + k;
                    if (a.add_jitter) v = v + SF_JITTER;
                } else {
                    v = k;
                }
                if (d == hw && j >= 1) {
                    // first diagonal past the caller's half-width (whatever the storage width): non-zero -> too small
                    bool outside = false;
                    (void)structured(j - 1, outside);
                    if (outside) atomicCAS(info + b, 0, SF_INFO_BANDWIDTH);
                }
            }
            dst[dstep * d] = v;
        }
    }
}

int sf_launch_band_fill(const sf_fill_args& a, int B, double* band, int ws, int halfwidth, int ldb, int64_t sband,
                        int* info, double* gtab, hipStream_t s, int tile_wt) {
    if (halfwidth < 0 || halfwidth >= ws) {
        sf_set_error("band fill: half-width %d does not fit the %d stored diagonals", halfwidth, ws);
        return SF_EINVAL;
    }
    if (a.n_local > SF_MAX_LOCAL) {
        sf_set_error("at most %d local kernels are supported", SF_MAX_LOCAL);
        return SF_EINVAL;
    }
    if (!a.monotonic) {
        sf_set_error("the banded solver needs a strictly increasing wavelength grid");
        return SF_EINVAL;
    }
    const bool table = gtab && a.has_global && a.loguniform && 2 * ws < a.n;
    if (table) {
        hipLaunchKernelGGL(k_band_gtab, dim3((ws + 256) / 256, B), dim3(256), 0, s, a, gtab, ws);
        SF_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_band_fill, dim3((unsigned)((a.npad + SF_BF_ROWS - 1) / SF_BF_ROWS), B), dim3(256), 0, s, a, band, ws, halfwidth, ldb, sband,
                       info, table ? (const double*)gtab : nullptr, tile_wt);
    SF_LAUNCH_CHECK();
    return SF_OK;
}

// ------------------------------------------------------------ stand-alone kernels (free functions)
__global__ __launch_bounds__(256) void k_global_cov(const double* __restrict__ wave, int n, double amp,
                                                    double ls, double* __restrict__ out) {
    const int col = blockIdx.x * 256 + threadIdx.x, row = blockIdx.y;
    if (col >= n) return;
    out[(int64_t)row * n + col] = sf_matern_elem(wave[row], wave[col], amp, ls, 6 * ls);
}

__global__ __launch_bounds__(256) void k_local_cov(const double* __restrict__ wave, int n, double amp,
                                                   double mu, double sigma, int accumulate,
                                                   double* __restrict__ out) {
    const int col = blockIdx.x * 256 + threadIdx.x, row = blockIdx.y;
    if (col >= n) return;
    const double v = sf_local_elem(sf_local_metric(wave[row], mu), sf_local_metric(wave[col], mu), amp,
                                   sigma, 4 * sigma);
    double* o = out + (int64_t)row * n + col;
    *o = accumulate ? (*o + v) : v;
}

int sf_launch_global_cov(const double* wave, int n, double amp, double ls, double* out, hipStream_t s) {
    if (n <= 0) return SF_OK;
    hipLaunchKernelGGL(k_global_cov, dim3((n + 255) / 256, n), dim3(256), 0, s, wave, n, amp, ls, out);
    SF_LAUNCH_CHECK();
    return SF_OK;
}

int sf_launch_local_cov(const double* wave, int n, double amp, double mu, double sigma, int accumulate,
                        double* out, hipStream_t s) {
    if (n <= 0) return SF_OK;
    hipLaunchKernelGGL(k_local_cov, dim3((n + 255) / 256, n), dim3(256), 0, s, wave, n, amp, mu, sigma,
                       accumulate, out);
    SF_LAUNCH_CHECK();
    return SF_OK;
}

// ---- dense (both triangles) fill of caller matrices: sf_forward_batch / sf_cov_fill_batch ------------------------
// One workgroup per 64 x 64 tile (k_fill_tiles) is 524 288 short workgroups at cfg 2, each with its own latency chain
// (parameters -> exp -> support tests -> Y fragments -> MFMA -> stores) at 3 waves per SIMD (the structured code path
// needs 149 VGPRs): 3.7 TB/s.  Here the tiles are split by what they need:
//   k_dense_map         per walker a byte per 64 x 64 tile: can a structured kernel reach it?  + compact list of those
//   k_fill_dense_plain  everything else (nine tiles in ten): rank-m term on the matrix cores (+ sigma^2 / jitter on the
//                       diagonal when no global kernel flags it), the lean <false> body (54 VGPRs, 8 waves per SIMD);
//                       a workgroup walks `span` column tiles of one 64-row strip, starting at a strip-dependent offset so
//                       that the strips of a round do not all write the same column range (row stride N = a power of two)
//   k_fill_dense_band   the listed tiles through the <true> body
// Same element values as k_fill_tiles (the same tile bodies; a tile the map leaves out has no sub-tile that a
// structured kernel reaches).
__global__ __launch_bounds__(256) void k_dense_map(sf_fill_args a, int nt, unsigned char* __restrict__ smap,
                                                   unsigned short* __restrict__ list, int* __restrict__ count) {
    const int e = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (e >= nt * nt) return;
    const int tm = e / nt, tn = e - tm * nt;
    const double* __restrict__ P = a.params + (int64_t)b * a.pstride;
    // (the flag of a tile is evaluated on the tile BELOW the diagonal of the pair {(tm, tn), (tn, tm)}: the map is symmetric by
    // construction, which the mirror writes of k_fill_dense_band and the skips of k_fill_dense_plain rely on)
    const int rlo = max(tm, tn) * FT, clo = min(tm, tn) * FT;
    bool g = false;
    unsigned lm = 0;
    if (rlo < a.n && clo < a.n) {
        const double g_r0 = a.has_global ? 6 * exp(P[a.off_global + 1]) : 0.0;
        sf_block_support(a, P, rlo, min(rlo + FT - 1, a.n - 1), clo, min(clo + FT - 1, a.n - 1), g_r0, g, lm);
    }
    const unsigned char flag = (g || lm) ? 1 : 0;
    smap[(int64_t)b * nt * nt + e] = flag;
    // (the list holds the tiles on and below the diagonal: k_fill_dense_band writes their mirror images as well)
    if (flag && tm >= tn) list[(int64_t)b * nt * nt + atomicAdd(&count[b], 1)] = (unsigned short)((tm << 8) | tn);
}

// KK = mpad / 4 MFMA K steps; a workgroup owns SPAN column tiles of one 64-row strip.  ALL Y fragments of the strip
// segment (and the segment's bytes of the support map) are requested up front, one round trip; after that a wave only
// issues MFMAs and stores.  With one tile per workgroup (k_fill_tiles) every workgroup's life was a load round trip
// through a memory system saturated with writes: 3.9 ms for 17.2 GB where the bare store pattern takes 2.7
// (tools/probes/write_pattern.hip); a rolling prefetch inside the loop does not help either, hipcc's wait counts then
// include the previous tile's stores (loads and stores share vmcnt).  Same MFMA sequence per accumulator as
// sf_fill_tile: same bits.
template <int KK, int SPAN>
__global__ __launch_bounds__(256, 4) void k_fill_dense_plain(sf_fill_args a, int nt, const unsigned char* __restrict__ smap) {
    const int nch = (nt + SPAN - 1) / SPAN;
    const int id = sf_xcd_remap_f(blockIdx.x, gridDim.x);
    const int b = id / (nt * nch);
    const int r = id - b * nt * nch;
    const int tm = r / nch, ch = r - tm * nch;
    const int t0 = ch * SPAN, cnt = min(SPAN, nt - t0);
    const unsigned char* __restrict__ row = smap ? smap + ((int64_t)b * nt + tm) * nt : nullptr;

    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int gam = lane & 15, q = lane >> 4;
    const int nout = sf_fill_extent(a);
    const int R0 = tm * FT + (w >> 1) * 32;
    if (R0 >= nout) return;
    const double* __restrict__ Yb = a.Y + (int64_t)b * a.mpad * a.ldy;
    const int colperm = 4 * (gam & 3) + (gam >> 2);
    // (Y holds npad >= 64 nt columns per row: the fragment loads of a partial last tile stay inside the walker's slice)
    double brow[KK][2];
#pragma unroll
    for (int k = 0; k < KK; ++k)
#pragma unroll
        for (int i = 0; i < 2; ++i) brow[k][i] = Yb[(int64_t)(4 * k + q) * a.ldy + R0 + i * 16 + gam];
    // tile t of the segment in staggered order (the strips of a round start at different columns)
    const int i0 = tm % cnt;
    double acol[SPAN][KK][2];
#pragma unroll
    for (int t = 0; t < SPAN; ++t) {
        const int tn = t0 + (i0 + t < cnt ? i0 + t : i0 + t - cnt);
        const int C0 = (t < cnt ? tn : t0) * FT + (w & 1) * 32;
#pragma unroll
        for (int k = 0; k < KK; ++k)
#pragma unroll
            for (int j = 0; j < 2; ++j) acol[t][k][j] = Yb[(int64_t)(4 * k + q) * a.ldy + C0 + j * 16 + colperm];
    }
    // support-map bytes of the segment: lane t reads tile t's byte, one ballot -> a wave-uniform mask (one round trip,
    // in flight together with the fragment loads)
    unsigned char fb = 0;
    if (row && lane < cnt) fb = row[t0 + (i0 + lane < cnt ? i0 + lane : i0 + lane - cnt)];
    const unsigned long long skip = __ballot(fb != 0);
    const bool vec_ok = ((a.lda | a.stride) & 1) == 0;
    double* __restrict__ Cb = a.C + (int64_t)b * a.stride;
#pragma unroll
    for (int t = 0; t < SPAN; ++t) {
        const int tn = t0 + (i0 + t < cnt ? i0 + t : i0 + t - cnt);
        const int C0 = tn * FT + (w & 1) * 32;
        if (t >= cnt || ((skip >> t) & 1) || C0 >= nout) continue;
        sf_d4 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = (sf_d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < KK; ++k)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(acol[t][k][j], brow[k][i], acc[i][j], 0, 0, 0);
        const bool interior = vec_ok && (R0 + 32 <= a.n) && (C0 + 32 <= a.n) && (R0 >= C0 + 32 || C0 >= R0 + 32);
        if (interior) {  // no diagonal entry, no padding: the accumulators are the values
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) {
                    double* dst = Cb + (int64_t)(R0 + ti * 16 + gam) * a.lda + C0 + tj * 16 + 4 * q;
                    *(double2*)dst = make_double2(acc[ti][tj][0], acc[ti][tj][1]);
                    *(double2*)(dst + 2) = make_double2(acc[ti][tj][2], acc[ti][tj][3]);
                }
        } else {
            sf_tile_finish<false>(a, b, R0, C0, acc);
        }
    }
}

template <int OCC>
__global__ __launch_bounds__(256, OCC) void k_fill_dense_band(sf_fill_args a, int nt, int G,
                                                              const unsigned short* __restrict__ list,
                                                              const int* __restrict__ count) {
    const int b = blockIdx.x / G, g = blockIdx.x - b * G;
    const int cnt = count[b];
    const unsigned short* __restrict__ l = list + (int64_t)b * nt * nt;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int gam = lane & 15, q = lane >> 4;
    const int nout = sf_fill_extent(a);
    const double* __restrict__ Yb = a.Y + (int64_t)b * a.mpad * a.ldy;
    const int colperm = 4 * (gam & 3) + (gam >> 2);
    for (int li = g; li < cnt; li += G) {
        const int e = l[li];
        const int tm = e >> 8, tn = e & 255;
        const int R0 = tm * FT + (w >> 1) * 32, C0 = tn * FT + (w & 1) * 32;
        if (R0 >= nout || C0 >= nout) continue;
        sf_d4 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = (sf_d4){0.0, 0.0, 0.0, 0.0};
        for (int kk = 0; kk < a.mpad; kk += 4) {  // (the MFMA sequence of sf_fill_tile)
            const double* yk = Yb + (int64_t)(kk + q) * a.ldy;
            double brow[2], acol[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                brow[i] = yk[R0 + i * 16 + gam];
                acol[i] = yk[C0 + i * 16 + colperm];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(acol[j], brow[i], acc[i][j], 0, 0, 0);
        }
        sf_tile_finish<true>(a, b, R0, C0, acc, tm > tn);
    }
}

size_t sf_fill_dense_map_tiles(int n) {
    const size_t nt = (size_t)(n + FT - 1) / FT;
    return nt * nt;
}

int sf_launch_fill_dense(const sf_fill_args& a, int B, unsigned char* smap, unsigned short* list, int* count, hipStream_t s,
                         sf_exec* ex) {
    const int nout = sf_fill_extent(a);
    const int nt = (nout + FT - 1) / FT;
    static const int old_env = SF_TUNE_INT("SF_FILL_OLD", 0);
    const bool all_structured = (a.has_global || a.n_local > 0) && !a.monotonic;  // (unsorted wavelengths: no culling)
    if (a.lower_only || a.tilemap || nt > 256 || !smap || !list || !count || old_env || all_structured || a.mpad > 16 || (a.mpad & 3))
        return sf_launch_fill(a, B, s);
    if (a.n_local > SF_MAX_LOCAL) {
        sf_set_error("at most %d local kernels are supported", SF_MAX_LOCAL);
        return SF_EINVAL;
    }
    sf_fill_args a2 = a;
    a2.gtab = nullptr;  // (the dense matrices keep the per-entry formula of the global kernel)
    const int structured = a.has_global || a.n_local > 0;
    if (structured) {
        SF_HIP(hipMemsetAsync(count, 0, sizeof(int) * (size_t)B, s));
        hipLaunchKernelGGL(k_dense_map, dim3((nt * nt + 255) / 256, B), dim3(256), 0, s, a2, nt, smap, list, count);
        SF_LAUNCH_CHECK();
    }
    // The structured tiles are bound by fp64 VALU work (exp / cos per entry: 0.85 ms at cfg 2), the plain ones by the HBM
    // write rate: with a context's auxiliary stream the two kernels run side by side (disjoint tiles, both write-only).
    static const int two_streams = SF_TUNE_INT("SF_FILL_TWO_STREAMS", 1);
    const bool fork = structured && ex && two_streams;
    hipStream_t sb = s;
    hipEvent_t e_map = nullptr, e_band = nullptr;
    if (fork) {
        SF_CHECK(sf_exec_event(ex, &e_map));
        SF_CHECK(sf_exec_event(ex, &e_band));
        sb = ex->aux;
        SF_HIP(hipEventRecord(e_map, s));
        SF_HIP(hipStreamWaitEvent(sb, e_map, 0));
        static const int G = SF_TUNE_INT("SF_FILL_BAND_G", 32);
        static const int occ = SF_TUNE_INT("SF_FILL_BAND_OCC", 2);
        if (occ == 4) hipLaunchKernelGGL(k_fill_dense_band<4>, dim3((unsigned)B * G), dim3(256), 0, sb, a2, nt, G, list, count);
        else if (occ == 3) hipLaunchKernelGGL(k_fill_dense_band<3>, dim3((unsigned)B * G), dim3(256), 0, sb, a2, nt, G, list, count);
        else hipLaunchKernelGGL(k_fill_dense_band<2>, dim3((unsigned)B * G), dim3(256), 0, sb, a2, nt, G, list, count);
        SF_LAUNCH_CHECK();
        SF_HIP(hipEventRecord(e_band, sb));
    }
    const unsigned char* pm = structured ? smap : nullptr;
    const int KK = a.mpad / 4;
    // (Y fragments of a whole segment live in registers; cfg 2, rank-m part alone: 4 tiles 2.97 ms, 8 tiles 3.12 ms)
    static const int span8 = SF_TUNE_INT("SF_FILL_SPAN8", 0);
    const int span = (KK <= 2 && span8) ? 8 : 4;
    const long long nblk = (long long)nt * ((nt + span - 1) / span) * B;
    if (nblk > 0x7fffffffLL) {
        sf_set_error("fill grid too large");
        return SF_EINVAL;
    }
    switch (KK) {
        case 1:
            if (span == 8) hipLaunchKernelGGL((k_fill_dense_plain<1, 8>), dim3((unsigned)nblk), dim3(256), 0, s, a2, nt, pm);
            else hipLaunchKernelGGL((k_fill_dense_plain<1, 4>), dim3((unsigned)nblk), dim3(256), 0, s, a2, nt, pm);
            break;
        case 2:
            if (span == 8) hipLaunchKernelGGL((k_fill_dense_plain<2, 8>), dim3((unsigned)nblk), dim3(256), 0, s, a2, nt, pm);
            else hipLaunchKernelGGL((k_fill_dense_plain<2, 4>), dim3((unsigned)nblk), dim3(256), 0, s, a2, nt, pm);
            break;
        case 3: hipLaunchKernelGGL((k_fill_dense_plain<3, 4>), dim3((unsigned)nblk), dim3(256), 0, s, a2, nt, pm); break;
        default: hipLaunchKernelGGL((k_fill_dense_plain<4, 4>), dim3((unsigned)nblk), dim3(256), 0, s, a2, nt, pm); break;
    }
    SF_LAUNCH_CHECK();
    if (fork) {
        SF_HIP(hipStreamWaitEvent(s, e_band, 0));
    } else if (structured) {
        const int G = 32;
        hipLaunchKernelGGL(k_fill_dense_band<2>, dim3((unsigned)B * G), dim3(256), 0, s, a2, nt, G, list, count);
        SF_LAUNCH_CHECK();
    }
    return SF_OK;
}


// Streaming-write probe (sf_debug_stream_write): every lane stores 16 bytes per iteration, a workgroup covers a contiguous
// 64 KB chunk per iteration (the write pattern of a bandwidth test, no reads).
__global__ __launch_bounds__(256) void k_stream_write(double* __restrict__ dst, size_t count2, double v) {
    double2* __restrict__ d2 = (double2*)dst;
    const double2 val = make_double2(v, v);
    const size_t chunk = 4096;  // double2 per workgroup and iteration
    for (size_t base = (size_t)blockIdx.x * chunk; base < count2; base += (size_t)gridDim.x * chunk) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const size_t j = base + (size_t)i * 256 + threadIdx.x;
            if (j < count2) d2[j] = val;
        }
    }
}
int sf_launch_stream_write(double* dst, size_t count, double v, hipStream_t s) {
    if (((uintptr_t)dst & 15) != 0 || (count & 1)) {
        sf_set_error("stream write probe: 16-byte aligned destination and an even count");
        return SF_EINVAL;
    }
    const size_t count2 = count / 2;
    if (!count2) return SF_OK;
    const unsigned grid = (unsigned)std::min<size_t>((count2 + 4095) / 4096, 256 * 32);
    hipLaunchKernelGGL(k_stream_write, dim3(grid), dim3(256), 0, s, dst, count2, v);
    SF_LAUNCH_CHECK();
    return SF_OK;
}
