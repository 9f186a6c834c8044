// Per-spectrum transform kernels for gfx950 (all fp64; compiled with -ffp-contract=off so element
// formulas keep the reference's operation order):
//   k_broaden        rfft -> kernel multiply -> irfft     Starfish/transforms.py:45-90, 93-134
//   k_spline_solve   banded (LU) solve of the k=5 B-spline collocation system  (transforms.py:39-42,
//                    FITPACK curfit with s=0; knots x[0]x6, x[3:-3], x[-1]x6)
//   k_eval_rows      doppler_shift + spline evaluation + chebyshev_correct + eigenspectrum
//                    reconstruction      transforms.py:137-158, 271-304; spectrum_model.py:293-313
//   k_scale, k_resid_y   rescale / renorm, residual and the rank-m factor Y   spectrum_model.py:316-335
//   k_emu_prep/z/post GP conditional of the PCA weights     Starfish/emulator/emulator.py:330-394
#include "sf_common.h"
#include "sf_transform.h"
typedef double sf_d4x __attribute__((ext_vector_type(4)));

// --------------------------------------------------------------------------------------- FFT
// In-place radix-2 decimation-in-time FFT on `buf` (LDS or global), input already bit-reversed.
// tw[k] = exp(-2 pi i k / nf), k < nf/2; this transform has length L = nf >> shift... (L == nf here)
__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// `twmul`: the table holds exp(-2 pi i k / (twmul * L)) (a table made for a longer transform).
__device__ void sf_fft_inplace(double2* buf, int L, const double2* __restrict__ tw, bool inverse, int twmul = 1) {
    const int tid = threadIdx.x, nth = blockDim.x;
    int ls = 0;  // log2(half-size)
    for (int s = 1; s < L; s <<= 1, ++ls) {
        const int twstep = L / (2 * s) * twmul;
        for (int idx = tid; idx < L / 2; idx += nth) {
            const int j = idx & (s - 1);
            const int i0 = ((idx >> ls) << (ls + 1)) + j;
            const int i1 = i0 + s;
            double2 wv = tw[j * twstep];
            if (inverse) wv.y = -wv.y;
            const double2 u = buf[i0];
            const double2 v = cmul(wv, buf[i1]);
            buf[i0] = make_double2(u.x + v.x, u.y + v.y);
            buf[i1] = make_double2(u.x - v.x, u.y - v.y);
        }
        __syncthreads();
    }
}

// Same transform with two radix-2 stages fused per pass (radix-2^2): half the LDS sweeps and barriers.
// Input bit-reversed (radix-2 order), output natural, exactly the butterflies of sf_fft_inplace.
__device__ void sf_fft_inplace_r4(double2* buf, int L, const double2* __restrict__ tw, bool inverse, int twmul = 1) {
    const int tid = threadIdx.x, nth = blockDim.x;
    int ls = 0, s = 1;
    int nst = 0;
    while ((1 << nst) < L) ++nst;
    if (nst & 1) {  // odd number of stages: one plain radix-2 stage first (half-size 1, twiddle 1)
        for (int idx = tid; idx < L / 2; idx += nth) {
            const double2 u = buf[2 * idx], v = buf[2 * idx + 1];
            buf[2 * idx] = make_double2(u.x + v.x, u.y + v.y);
            buf[2 * idx + 1] = make_double2(u.x - v.x, u.y - v.y);
        }
        __syncthreads();
        s = 2;
        ls = 1;
    }
    for (; s < L; s <<= 2, ls += 2) {
        const int stepA = L / (2 * s) * twmul, stepB = L / (4 * s) * twmul;
        for (int idx = tid; idx < L / 4; idx += nth) {
            const int j = idx & (s - 1);
            const int i0 = ((idx >> ls) << (ls + 2)) + j;
            double2 wA = tw[j * stepA], wB0 = tw[j * stepB], wB1 = tw[(j + s) * stepB];
            if (inverse) {
                wA.y = -wA.y;
                wB0.y = -wB0.y;
                wB1.y = -wB1.y;
            }
            const double2 x0 = buf[i0], x1 = buf[i0 + s], x2 = buf[i0 + 2 * s], x3 = buf[i0 + 3 * s];
            const double2 t1 = cmul(wA, x1), t3 = cmul(wA, x3);
            const double2 a0 = make_double2(x0.x + t1.x, x0.y + t1.y), a1 = make_double2(x0.x - t1.x, x0.y - t1.y);
            const double2 a2 = make_double2(x2.x + t3.x, x2.y + t3.y), a3 = make_double2(x2.x - t3.x, x2.y - t3.y);
            const double2 u2 = cmul(wB0, a2), u3 = cmul(wB1, a3);
            buf[i0] = make_double2(a0.x + u2.x, a0.y + u2.y);
            buf[i0 + 2 * s] = make_double2(a0.x - u2.x, a0.y - u2.y);
            buf[i0 + s] = make_double2(a1.x + u3.x, a1.y + u3.y);
            buf[i0 + 3 * s] = make_double2(a1.x - u3.x, a1.y - u3.y);
        }
        __syncthreads();
    }
}

// Decimation-in-frequency forward FFT: natural-order input, BIT-REVERSED output (so the product
// with a real symmetric multiplier feeds the DIT inverse above without any permutation pass).
__device__ void sf_fft_dif_forward(double2* buf, int L, const double2* __restrict__ tw) {
    const int tid = threadIdx.x, nth = blockDim.x;
    int ls = 0;
    while ((2 << ls) < L) ++ls;  // log2(L/2)
    for (int s = L >> 1; s >= 1; s >>= 1, --ls) {
        const int twstep = L / (2 * s);
        for (int idx = tid; idx < L / 2; idx += nth) {
            const int j = idx & (s - 1);
            const int i0 = ((idx >> ls) << (ls + 1)) + j;
            const int i1 = i0 + s;
            const double2 wv = tw[j * twstep];
            const double2 u = buf[i0], v = buf[i1];
            buf[i0] = make_double2(u.x + v.x, u.y + v.y);
            buf[i1] = cmul(wv, make_double2(u.x - v.x, u.y - v.y));
        }
        __syncthreads();
    }
}

__device__ __forceinline__ unsigned sf_bitrev(unsigned x, int bits) { return __brev(x) >> (32 - bits); }

// Gray (2005) rotational kernel, transforms.py:129-131, and the Gaussian profile, transforms.py:84-85
__device__ __forceinline__ double sf_rot_mult(int k, double val, double vsini) {
    if (k == 0) return 1.0;
    const double freq = k * val;
    const double ub = 2.0 * M_PI * vsini * freq;
    return j1(ub) / ub - 3 * cos(ub) / (2 * (ub * ub)) + 3.0 * sin(ub) / (2 * (ub * ub * ub));
}
__device__ __forceinline__ double sf_inst_mult(int k, double val, double fwhm) {
    const double freq = k * val;
    const double sigma = fwhm / 2.355;
    const double a = M_PI * sigma * freq;
    return exp(-2 * (a * a));
}

// One workgroup per spectrum row.
//   FWD   : true  -> the row is real input `in` (rows x nf) and is transformed first;
//           false -> `spec` holds the precomputed half spectrum (rows_static x (nf/2+1)).
//   kind  : 0 none (multiplier 1), 1 rotational (param = vsini), 2 instrumental (param = fwhm)
// Output element j of row r of item b is written at out[b*ob + r*orow + j*oelem].
template <bool FWD, bool USE_LDS>
__global__ __launch_bounds__(256) void k_broaden(const double* __restrict__ in,
                                                 const double2* __restrict__ spec, int rows, int nf,
                                                 const double2* __restrict__ tw, double dv, int kind,
                                                 const double* __restrict__ params, int pstride,
                                                 int poff, double scalar_param, double* __restrict__ out,
                                                 int64_t ob, int64_t orow, int64_t oelem,
                                                 double2* __restrict__ gscratch, int* __restrict__ info) {
    extern __shared__ __attribute__((aligned(16))) double2 lbuf[];
    const int row = blockIdx.x, b = blockIdx.y;
    double2* buf = USE_LDS ? lbuf : gscratch + ((int64_t)b * rows + row) * nf;
    const int tid = threadIdx.x;
    int bits = 0;
    while ((1 << bits) < nf) ++bits;
    const int nh = nf / 2;

    double param = scalar_param;
    if (params) param = params[(int64_t)b * pstride + poff];
    if (kind == 1 && !(param > 0.0)) {  // transforms.py:121-122
        if (tid == 0 && info) atomicCAS(&info[b], 0, SF_INFO_BAD_VSINI);
        return;
    }
    const double val = 1.0 / (nf * dv);  // numpy.fft.rfftfreq

    if (FWD) {
        const double* x = in + ((int64_t)b * rows + row) * nf;
        for (int j = tid; j < nf; j += 256) buf[j] = make_double2(x[j], 0.0);
        __syncthreads();
        sf_fft_dif_forward(buf, nf, tw);
        // position p holds frequency k = bitrev(p); the multiplier is real and even in k
        for (int p = tid; p < nf; p += 256) {
            const int k = (int)sf_bitrev((unsigned)p, bits);
            const int kk = (k <= nh) ? k : nf - k;
            double mult = 1.0;
            if (kind == 1) mult = sf_rot_mult(kk, val, param);
            else if (kind == 2) mult = sf_inst_mult(kk, val, param);
            double2 X = buf[p];
            X.x *= mult;
            X.y *= mult;
            if (kk == 0 || kk == nh) X.y = 0.0;  // c2r ignores the imaginary part of DC / Nyquist
            buf[p] = X;
        }
    } else {
        // X_k (k <= nf/2) and conj(X_{nf-k}) go straight to their bit-reversed slots
        for (int k = tid; k <= nh; k += 256) {
            double2 X = spec[(int64_t)row * (nh + 1) + k];
            double mult = 1.0;
            if (kind == 1) mult = sf_rot_mult(k, val, param);
            else if (kind == 2) mult = sf_inst_mult(k, val, param);
            X.x *= mult;
            X.y *= mult;
            if (k == 0 || k == nh) X.y = 0.0;
            buf[sf_bitrev(k, bits)] = X;
            if (k != 0 && k != nh) buf[sf_bitrev(nf - k, bits)] = make_double2(X.x, -X.y);
        }
    }
    __syncthreads();
    sf_fft_inplace(buf, nf, tw, true);
    const double inv_n = 1.0 / nf;
    double* o = out + (int64_t)b * ob + (int64_t)row * orow;
    for (int j = tid; j < nf; j += 256) o[(int64_t)j * oelem] = buf[j].x * inv_n;
}

// Hot-path variant (precomputed half spectra): the kernel multiplier depends on the walker only, so it is
// tabulated once per walker (k_kernel_mult) instead of once per row, and the real inverse transform runs
// as a HALF-size complex FFT:  Z_k = (X_k + conj X_{L-k}) + i e^{+2 pi i k/nf} (X_k - conj X_{L-k}),
// L = nf/2;  z = IDFT_L(Z)  =>  x_{2m} = Re z_m, x_{2m+1} = Im z_m.  64 KiB of LDS at nf = 8192.
__global__ __launch_bounds__(256) void k_kernel_mult(double* __restrict__ mult, int nh1, double val, int kind,
                                                     const double* __restrict__ params, int pstride, int poff,
                                                     double scalar_param, int* __restrict__ info) {
    const int b = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    double param = scalar_param;
    if (params) param = params[(int64_t)b * pstride + poff];
    if (kind == 1 && !(param > 0.0)) {  // transforms.py:121-122
        if (k == 0 && info) atomicCAS(&info[b], 0, SF_INFO_BAD_VSINI);
        return;
    }
    if (k >= nh1) return;
    double m = 1.0;
    if (kind == 1) m = sf_rot_mult(k, val, param);
    else if (kind == 2) m = sf_inst_mult(k, val, param);
    mult[(int64_t)b * nh1 + k] = m;
}

template <bool USE_LDS>
__global__ __launch_bounds__(256) void k_broaden_half(const double2* __restrict__ spec,
                                                      const double* __restrict__ mult, int rows, int nf,
                                                      const double2* __restrict__ tw, int kind,
                                                      const double* __restrict__ params, int pstride, int poff,
                                                      double scalar_param, double* __restrict__ out, int64_t ob,
                                                      int64_t orow, int64_t oelem, double2* __restrict__ gscratch) {
    extern __shared__ __attribute__((aligned(16))) double2 lbuf[];
    const int row = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int L = nf / 2;
    double2* buf = USE_LDS ? lbuf : gscratch + ((int64_t)b * rows + row) * nf;
    double param = scalar_param;
    if (params) param = params[(int64_t)b * pstride + poff];
    if (kind == 1 && !(param > 0.0)) return;  // flagged by k_kernel_mult
    int bits = 0;
    while ((1 << bits) < L) ++bits;
    const double2* X = spec + (int64_t)row * (L + 1);
    const double* mb = mult + (int64_t)b * (L + 1);
    for (int k = tid; k < L; k += 256) {
        double2 a = X[k], c = X[L - k];
        const double ma = mb[k], mc = mb[L - k];
        a.x *= ma;
        a.y *= ma;
        c.x *= mc;
        c.y *= mc;
        if (k == 0) a.y = 0.0, c.y = 0.0;  // c2r ignores the imaginary part of DC / Nyquist
        // conj(X_{L-k}) = (c.x, -c.y)
        const double2 E = make_double2(a.x + c.x, a.y - c.y);
        const double2 D = make_double2(a.x - c.x, a.y + c.y);
        const double2 w = tw[k];  // exp(-2 pi i k / nf); we need its conjugate
        const double2 O = make_double2(w.x * D.x + w.y * D.y, w.x * D.y - w.y * D.x);
        buf[sf_bitrev((unsigned)k, bits)] = make_double2(E.x - O.y, E.y + O.x);
    }
    __syncthreads();
    sf_fft_inplace_r4(buf, L, tw, true, 2);
    const double inv_n = 1.0 / nf;
    double* o = out + (int64_t)b * ob + (int64_t)row * orow;
    if (oelem == 1) {  // contiguous row: one 16-byte store per thread
        double2* o2 = (double2*)o;
        for (int m = tid; m < L; m += 256) {
            const double2 z = buf[m];
            o2[m] = make_double2(z.x * inv_n, z.y * inv_n);
        }
    } else {
        for (int m = tid; m < L; m += 256) {
            const double2 z = buf[m];
            o[(int64_t)(2 * m) * oelem] = z.x * inv_n;
            o[(int64_t)(2 * m + 1) * oelem] = z.y * inv_n;
        }
    }
}

// Forward half spectrum of static rows (context creation): spec[row][k], k <= nf/2.
template <bool USE_LDS>
__global__ __launch_bounds__(256) void k_rfft_rows(const double* __restrict__ in, int nf,
                                                   const double2* __restrict__ tw,
                                                   double2* __restrict__ spec,
                                                   double2* __restrict__ gscratch) {
    extern __shared__ __attribute__((aligned(16))) double2 lbuf[];
    const int row = blockIdx.x, tid = threadIdx.x;
    double2* buf = USE_LDS ? lbuf : gscratch + (int64_t)row * nf;
    int bits = 0;
    while ((1 << bits) < nf) ++bits;
    const double* x = in + (int64_t)row * nf;
    for (int j = tid; j < nf; j += 256) buf[sf_bitrev(j, bits)] = make_double2(x[j], 0.0);
    __syncthreads();
    sf_fft_inplace(buf, nf, tw, false);
    for (int k = tid; k <= nf / 2; k += 256) spec[(int64_t)row * (nf / 2 + 1) + k] = buf[k];
}

// ------------------------------------------------------------------------- banded spline solve
// One lane per right-hand side; element j of system s lives at data[s_base(s) + j*estride].
// Systems are grouped: s = b*rows + r -> base = b*bstride + r*rstride.
// The recurrences are sequential in j and only ~20 waves exist (B*(m+2)/64), so the kernel is pure
// latency: every lane keeps the NEXT chunk of SCH rows (and lane l the factor row j0+l) in flight in
// registers while the current chunk is eliminated; factor rows are broadcast through LDS.
#define SCH 64
__global__ __launch_bounds__(64) void k_spline_solve(double* __restrict__ data, int nsys, int rows,
                                                     int64_t bstride, int64_t rstride, int64_t estride,
                                                     int n, const double* __restrict__ Lf,
                                                     const double* __restrict__ Uf,
                                                     const double* __restrict__ rdiag) {
    __shared__ double fac[SCH * (SF_KB + 1)];
    const int lane = threadIdx.x;
    int s = blockIdx.x * 64 + lane;
    const bool live = s < nsys;
    if (!live) s = nsys - 1;  // keep the wave converged; results of dead lanes are not stored
    const int b = s / rows, r = s - b * rows;
    double* x = data + (int64_t)b * bstride + (int64_t)r * rstride;
    const int nch = (n + SCH - 1) / SCH;

    double cur[SCH], nxt[SCH];
    double cf[SF_KB + 1], nf[SF_KB + 1];
    auto load_chunk = [&](int ch, double* v) {
        const int j0 = ch * SCH;
#pragma unroll
        for (int jj = 0; jj < SCH; ++jj) v[jj] = (j0 + jj < n) ? x[(int64_t)(j0 + jj) * estride] : 0.0;
    };
    auto store_chunk = [&](int ch, const double* v) {
        const int j0 = ch * SCH;
        if (!live) return;
#pragma unroll
        for (int jj = 0; jj < SCH; ++jj)
            if (j0 + jj < n) x[(int64_t)(j0 + jj) * estride] = v[jj];
    };
    auto load_fac = [&](int ch, const double* __restrict__ F, bool with_diag, double* f) {
        const int j = ch * SCH + lane;
#pragma unroll
        for (int k = 0; k < SF_KB; ++k) f[k] = (j < n) ? F[(int64_t)j * SF_KB + k] : 0.0;
        f[SF_KB] = (with_diag && j < n) ? rdiag[j] : 0.0;
    };
    auto publish_fac = [&](const double* f) {
        __syncthreads();  // everyone finished reading the previous chunk's factors
#pragma unroll
        for (int k = 0; k <= SF_KB; ++k) fac[lane * (SF_KB + 1) + k] = f[k];
        __syncthreads();
    };

    // ---------------- forward: y_j = b_j - sum_{k=1..KB} L[j][k] y_{j-k}
    double y1 = 0, y2 = 0, y3 = 0, y4 = 0, y5 = 0;
    load_fac(0, Lf, false, cf);
    load_chunk(0, cur);
    for (int ch = 0; ch < nch; ++ch) {
        publish_fac(cf);
        if (ch + 1 < nch) {
            load_fac(ch + 1, Lf, false, nf);
            load_chunk(ch + 1, nxt);
        }
#pragma unroll
        for (int jj = 0; jj < SCH; ++jj) {
            const double* l = &fac[jj * (SF_KB + 1)];
            // older terms first (off the critical path); the dependent step is a single fma
            const double part = cur[jj] - ((l[4] * y5 + l[3] * y4) + (l[2] * y3 + l[1] * y2));
            const double v = fma(-l[0], y1, part);
            cur[jj] = v;
            y5 = y4; y4 = y3; y3 = y2; y2 = y1; y1 = v;
        }
        store_chunk(ch, cur);
#pragma unroll
        for (int jj = 0; jj < SCH; ++jj) cur[jj] = nxt[jj];
#pragma unroll
        for (int k = 0; k <= SF_KB; ++k) cf[k] = nf[k];
    }
    // ---------------- backward: c_j = (y_j - sum_{k=1..KB} U[j][k] c_{j+k}) / U[j][j]
    double c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0;
    __threadfence_block();
    load_fac(nch - 1, Uf, true, cf);
    load_chunk(nch - 1, cur);
    for (int ch = nch - 1; ch >= 0; --ch) {
        publish_fac(cf);
        if (ch > 0) {
            load_fac(ch - 1, Uf, true, nf);
            load_chunk(ch - 1, nxt);
        }
        const int j0 = ch * SCH;
#pragma unroll
        for (int jj = SCH - 1; jj >= 0; --jj) {
            if (j0 + jj < n) {
                const double* u = &fac[jj * (SF_KB + 1)];
                const double part = (cur[jj] - ((u[4] * c5 + u[3] * c4) + (u[2] * c3 + u[1] * c2))) * u[SF_KB];
                const double v = fma(-(u[0] * u[SF_KB]), c1, part);
                cur[jj] = v;
                c5 = c4; c4 = c3; c3 = c2; c2 = c1; c1 = v;
            }
        }
        store_chunk(ch, cur);
#pragma unroll
        for (int jj = 0; jj < SCH; ++jj) cur[jj] = nxt[jj];
#pragma unroll
        for (int k = 0; k <= SF_KB; ++k) cf[k] = nf[k];
    }
}

// Fully parallel variant for the per-walker path: the collocation matrix of the fixed log-lambda grid is
// well conditioned (cond ~ 15) and its inverse decays like 0.43^|i-j|, so c_i = sum_{|d| <= SF_IW}
// Ainv[i][i+d] y_{i+d} with the band precomputed at context creation (truncation < 1e-23 relative).
// That is a block-banded matrix product and runs on v_mfma_f64_16x16x4_f64:
//   C[16 i's][rows] = sum over the 9 input blocks kb of  T[ib][kb] (16 x 16)  x  Y[16 k's][rows]
// One wave owns one block of 16 outputs and keeps its 9 T blocks in registers (36 A fragments) while it
// loops over a chunk of walkers, so the 9.4 MB table is read B/chunk times, not B times; the B operand
// (lane (k, r) <- y[b][r][k], every row contiguous as the FFT kernel writes it) and the result are addressed
// straight in HBM/L2: no LDS.
// The walker loop is software pipelined: the fragments of walker b+1 are in flight while the matrix
// core works on walker b.
// y is [B][rows][n], c is [B][n][rows] (what k_eval_rows reads); tblk is [n/16][SF_IBLK][16][16] (zero outside
// the band / the matrix).
#define SF_IBLK (2 * (SF_IW / 16) + 1)
template <int NCB>
__global__ __launch_bounds__(256) void k_spline_apply(const double* __restrict__ y, double* __restrict__ c,
                                                      int rows, int n, const double* __restrict__ tblk, int B,
                                                      int wchunk) {
    const int lane = threadIdx.x & 63, l15 = lane & 15, lq = lane >> 4;
    const int ib = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ib * 16 >= n) return;
    double a[SF_IBLK][4];
#pragma unroll
    for (int kb = 0; kb < SF_IBLK; ++kb)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            a[kb][kk] = tblk[(((int64_t)ib * SF_IBLK + kb) * 16 + l15) * 16 + 4 * lq + kk];  // K slice lq of MFMA kk <-> k = 4 lq + kk
    const int nblk16 = n / 16;
    const int b0 = blockIdx.y * wchunk, b1 = min(B, b0 + wchunk);
    double bA[SF_IBLK][4][NCB], bB[SF_IBLK][4][NCB];
    // lane (r = l15, lq) takes the four CONTIGUOUS inputs 4 lq .. 4 lq + 3 of row r of a block (the same
    // permutation of the summation index as in the coefficient fragments): one 32-byte load per block
    auto fetch = [&](int b, double (&dst)[SF_IBLK][4][NCB]) {
        const double* yb = y + (int64_t)b * n * rows;
#pragma unroll
        for (int kb = 0; kb < SF_IBLK; ++kb) {
            int kblk = ib - SF_IW / 16 + kb;  // blocks outside the matrix carry zero coefficients
            kblk = kblk < 0 ? 0 : (kblk >= nblk16 ? nblk16 - 1 : kblk);
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const int r = cb * 16 + l15;
                const double2* p = (const double2*)(yb + (int64_t)(r < rows ? r : 0) * n + kblk * 16 + 4 * lq);
                const double2 lo = p[0], hi = p[1];  // rows >= `rows` are never stored
                dst[kb][0][cb] = lo.x;
                dst[kb][1][cb] = lo.y;
                dst[kb][2][cb] = hi.x;
                dst[kb][3][cb] = hi.y;
            }
        }
    };
    auto compute = [&](int b, const double (&bv)[SF_IBLK][4][NCB]) {
        sf_d4x acc[NCB];
#pragma unroll
        for (int cb = 0; cb < NCB; ++cb) acc[cb] = (sf_d4x){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kb = 0; kb < SF_IBLK; ++kb)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                for (int cb = 0; cb < NCB; ++cb)
                    acc[cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kb][kk], bv[kb][kk][cb], acc[cb], 0, 0, 0);
        double* cb_ = c + (int64_t)b * n * rows;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int i = ib * 16 + lq + 4 * r4;
#pragma unroll
            for (int cb = 0; cb < NCB; ++cb) {
                const int r = cb * 16 + l15;
                if (r < rows && i < n) cb_[(int64_t)i * rows + r] = acc[cb][r4];
            }
        }
    };
    if (b0 < b1) fetch(b0, bA);
    for (int b = b0; b < b1; b += 2) {  // ping-pong: the fragments of the next walker are in flight
        if (b + 1 < b1) fetch(b + 1, bB);
        compute(b, bA);
        if (b + 1 < b1) {
            if (b + 2 < b1) fetch(b + 2, bA);
            compute(b + 1, bB);
        }
    }
}

// ------------------------------------------------------------------------ spline evaluation
// FITPACK fpbspl: the six non-zero quintic B-splines on [t[ell], t[ell+1]) at x, knots scaled by s.
__device__ __forceinline__ void sf_bspl6(const double* __restrict__ t, double s, int ell, double x,
                                         double h[6]) {
    double tk[12];  // t[ell-5 .. ell+6] scaled
#pragma unroll
    for (int i = 0; i < 12; ++i) tk[i] = t[ell - 5 + i] * s;
    double hh[5];
    h[0] = 1.0;
#pragma unroll
    for (int j = 1; j <= 5; ++j) {
#pragma unroll
        for (int i = 0; i < j; ++i) hh[i] = h[i];
        h[0] = 0.0;
#pragma unroll
        for (int i = 1; i <= j; ++i) {
            // li = ell + i, lj = li - j  -> tk index = (.) - (ell - 5)
            const double tli = tk[5 + i], tlj = tk[5 + i - j];
            const double f = hh[i - 1] / (tli - tlj);
            h[i - 1] = h[i - 1] + f * (tli - x);
            h[i] = f * (x - tlj);
        }
    }
}

// splev interval search: largest ell in [5, ncoef-1] with t[ell]*s <= x
__device__ __forceinline__ int sf_find_interval(const double* __restrict__ t, double s, int ncoef, double x) {
    int lo = 5, hi = ncoef - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (t[mid] * s <= x) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// numpy.polynomial.chebyshev.chebval (Clenshaw) with coefficient vector [1, c1, c2, ...]
__device__ __forceinline__ double sf_chebval(double x, const double* __restrict__ c, int nc /* incl. c0 */,
                                             double c0first) {
    auto coef = [&](int i) { return i == 0 ? c0first : c[i - 1]; };
    double a0, a1;
    if (nc == 1) { a0 = coef(0); a1 = 0.0; }
    else if (nc == 2) { a0 = coef(0); a1 = coef(1); }
    else {
        const double x2 = 2 * x;
        a0 = coef(nc - 2);
        a1 = coef(nc - 1);
        for (int i = 3; i <= nc; ++i) {
            const double tmp = a0;
            a0 = coef(nc - i) - a1;
            a1 = tmp + a1 * x2;
        }
    }
    return a0 + a1 * x;
}

// Cardelli, Clayton & Mathis (1989) extinction law A(lambda)/A(V) = a(x) + b(x)/Rv, x = 1/lambda[um]
// (their eqs. 2a-5b).  Reference call site: extinct() Starfish/transforms.py:161-206 -> third-party
// `extinction.ccm89`; PARITY UNPINNED (that package is not available), checked against the paper's
// Table 3 only.  Returns the flux multiplier 10^(-0.4 Av (a + b/Rv)).
__device__ __forceinline__ double sf_ccm89_mult(double wave_A, double Av, double Rv) {
    const double x = 1e4 / wave_A;
    double a, b;
    if (x < 1.1) {
        const double p = pow(x, 1.61);
        a = 0.574 * p;
        b = -0.527 * p;
    } else if (x <= 3.3) {
        const double y = x - 1.82;
        a = 1 + y * (0.17699 + y * (-0.50447 + y * (-0.02427 + y * (0.72085 + y * (0.01979 + y * (-0.77530 + y * 0.32999))))));
        b = y * (1.41338 + y * (2.28305 + y * (1.07233 + y * (-5.38434 + y * (-0.62251 + y * (5.30260 + y * -2.09002))))));
    } else if (x <= 8.0) {
        double fa = 0.0, fb = 0.0;
        if (x >= 5.9) {
            const double d = x - 5.9;
            fa = -0.04473 * d * d - 0.009779 * d * d * d;
            fb = 0.2130 * d * d + 0.1207 * d * d * d;
        }
        a = 1.752 - 0.316 * x - 0.104 / ((x - 4.67) * (x - 4.67) + 0.341) + fa;
        b = -3.090 + 1.825 * x + 1.206 / ((x - 4.62) * (x - 4.62) + 0.263) + fb;
    } else {
        const double d = x - 8.0;
        a = -1.073 - 0.628 * d + 0.137 * d * d - 0.070 * d * d * d;
        b = 13.670 + 4.257 * d - 0.420 * d * d + 0.374 * d * d * d;
    }
    return pow(10.0, -0.4 * (Av * (a + b / Rv)));
}

// O'Donnell (1994, ApJ 422, 158): CCM89 with re-derived optical/NIR coefficients for 1.1 <= x <= 3.3 um^-1
// (continuous with the CCM infrared branch at x = 1.1: a = 0.6689, b = -0.6126); other ranges as CCM89.
// Calzetti et al. (2000, ApJ 533, 682), eq. 4: k(lambda) = 2.659 (-1.857 + 1.040/l) + Rv for
// 0.63 um <= l <= 2.2 um and 2.659 (-2.156 + 1.509/l - 0.198/l^2 + 0.011/l^3) + Rv for 0.12 um <= l < 0.63 um,
// A_lambda = Av k / Rv (k(0.55 um) = Rv).  Outside 0.12 - 2.2 um the nearer branch is extrapolated.
// Both PARITY UNPINNED like ccm89 (literature formulas; the reference's `extinction` package is unavailable).
__device__ __forceinline__ double sf_extinct_mult(double wave_A, double Av, double Rv, int law) {
    if (law == 1) {
        const double x = 1e4 / wave_A;
        if (x >= 1.1 && x <= 3.3) {
            const double y = x - 1.82;
            const double a = 1 + y * (0.104 + y * (-0.609 + y * (0.701 + y * (1.137 + y * (-1.718 + y * (-0.827 + y * (1.647 + y * -0.505)))))));
            const double b = y * (1.952 + y * (2.908 + y * (-3.989 + y * (-7.985 + y * (11.102 + y * (5.491 + y * (-10.805 + y * 3.347)))))));
            return pow(10.0, -0.4 * (Av * (a + b / Rv)));
        }
        return sf_ccm89_mult(wave_A, Av, Rv);
    }
    if (law == 2) {
        const double l = wave_A * 1e-4;  // micron
        const double il = 1.0 / l;
        const double k = (l >= 0.63) ? 2.659 * (-1.857 + 1.040 * il) + Rv
                                     : 2.659 * (-2.156 + il * (1.509 + il * (-0.198 + il * 0.011))) + Rv;
        return pow(10.0, -0.4 * (Av * k / Rv));
    }
    return sf_ccm89_mult(wave_A, Av, Rv);
}

__global__ __launch_bounds__(256) void k_extinct_rows(const double* __restrict__ wave, int n,
                                                      const double* __restrict__ flux, int rows, double Av,
                                                      double Rv, int law, double* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double mlt = sf_extinct_mult(wave[i], Av, Rv, law);
    for (int r = 0; r < rows; ++r) out[(int64_t)r * n + i] = flux[(int64_t)r * n + i] * mlt;
}

// Spline-based laws: Fitzpatrick (1999, PASP 111, 63) and Fitzpatrick & Massa (2007, ApJ 663, 320).  k(x) =
// E(lambda - V)/E(B - V) is a natural cubic spline through a handful of anchor points in x = 1/lambda [um^-1] up to
// 1e4/2700 and the Fitzpatrick-Massa ultraviolet parametrisation beyond; A_lambda = Av (1 + k / Rv).  The anchors,
// their second derivatives (host: sf_extinct) and the UV constants arrive in `p`:
//   p[0] = number of knots nk, p[1..7] = c1, c2, c3, c4, c5, x0^2, gamma^2, p[8] = 1 for the F99 far-UV term
//   (0.5392 y^2 + 0.05644 y^3) / 0 for FM07's y^2, then xk[nk], yk[nk], y2[nk].   PARITY UNPINNED like the others.
__global__ __launch_bounds__(256) void k_extinct_spline_rows(const double* __restrict__ wave, int n,
                                                             const double* __restrict__ flux, int rows, double Av,
                                                             double Rv, const double* __restrict__ p,
                                                             double* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int nk = (int)p[0];
    const double* xk = p + 9;
    const double* yk = xk + nk;
    const double* y2 = yk + nk;
    const double x = 1e4 / wave[i];
    double k;
    if (x >= xk[nk - 2]) {  // ultraviolet: lambda <= 2700 A (the last two knots are UV points themselves)
        const double x2 = x * x;
        const double d = x2 / ((x2 - p[6]) * (x2 - p[6]) + x2 * p[7]);
        k = p[1] + p[2] * x + p[3] * d;
        if (x >= p[5]) {
            const double y = x - p[5];
            k += p[8] != 0.0 ? p[4] * (0.5392 * y * y + 0.05644 * y * y * y) : p[4] * y * y;
        }
    } else {
        int lo = 0;
        while (lo + 2 < nk && x >= xk[lo + 1]) ++lo;
        const double h = xk[lo + 1] - xk[lo];
        const double a = (xk[lo + 1] - x) / h, b = (x - xk[lo]) / h;
        k = a * yk[lo] + b * yk[lo + 1] + ((a * a * a - a) * y2[lo] + (b * b * b - b) * y2[lo + 1]) * (h * h) / 6.0;
    }
    const double mlt = pow(10.0, -0.4 * (Av * (1.0 + k / Rv)));
    for (int r = 0; r < rows; ++r) out[(int64_t)r * n + i] = flux[(int64_t)r * n + i] * mlt;
}

// Generic resample (free function): out[r][q] = spline_r(xq[q]); coefficients coef[r][j] row-major.
__global__ __launch_bounds__(256) void k_spline_eval(const double* __restrict__ coef, int rows, int ncoef,
                                                     const double* __restrict__ t,
                                                     const double* __restrict__ xq, int nq,
                                                     double* __restrict__ out) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    const double x = xq[q];
    const int ell = sf_find_interval(t, 1.0, ncoef, x);
    double h[6];
    sf_bspl6(t, 1.0, ell, x, h);
    for (int r = 0; r < rows; ++r) {
        const double* c = coef + (int64_t)r * ncoef + ell - 5;
        double sp = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) sp = sp + c[j] * h[j];
        out[(int64_t)r * nq + q] = sp;
    }
}

// Fused: Doppler-scaled spline evaluation of the m+2 rows, Chebyshev multiply, reconstruction.
// the m + 2 rows at pixel i: xk[k] = eig_k * std (spectrum_model.py:312), returns the reconstruction sum_k w_k xk + mean
__device__ __forceinline__ double sf_eval_pixel(const sf_eval_args& a, int b, int i, double* xk) {
    const double* __restrict__ P = a.params + (int64_t)b * a.pstride;
    const double x = a.wave[i];
    double s = 1.0;
    if (a.has_vz) {
        const double vz = P[1];
        s = sqrt((SF_C_KMS + vz) / (SF_C_KMS - vz));  // transforms.py:157
    }
    const int ell = sf_find_interval(a.knots, s, a.nf, x);
    double h[6];
    sf_bspl6(a.knots, s, ell, x, h);
    const int rows = a.m + 2;
    const double* __restrict__ cf =
        (a.coef_batched ? a.coef + (int64_t)b * a.nf * rows : a.coef) + (int64_t)(ell - 5) * rows;
    double p = 1.0;
    if (a.n_cheb > 0) p = sf_chebval(x / a.wave_max, P + a.off_cheb, a.n_cheb + 1, 1.0);  // transforms.py:302-304
    double ext = 1.0;
    if (a.has_av) ext = sf_ccm89_mult(x, P[a.off_av], 3.1);  // spectrum_model.py:298-299 (Rv never passed)
    auto rowval = [&](int r) {
        double sp = 0.0;
#pragma unroll
        for (int j = 0; j < 6; ++j) sp = sp + cf[(int64_t)j * rows + r] * h[j];
        if (a.has_av) sp = sp * ext;  // extinct before the Chebyshev correction, as the reference orders them
        return a.n_cheb > 0 ? sp * p : sp;
    };
    const double mean = rowval(a.m), std = rowval(a.m + 1);
    const double* __restrict__ wmu = a.mu + (int64_t)b * a.m;
    double flux = 0.0;
    for (int k = 0; k < a.m; ++k) {
        xk[k] = rowval(k) * std;            // spectrum_model.py:312
        flux = flux + wmu[k] * xk[k];       // spectrum_model.py:313
    }
    return flux + mean;
}
// rank-m factor row at one pixel: xs (scaled X column) -> Y column, zero padded   (k_resid_y, k_eval_resid_y)
__device__ __forceinline__ void sf_y_column(const sf_resid_args& a, int b, int i, double* xs, double* __restrict__ Yb) {
    const double* __restrict__ Lw = a.Lw + (int64_t)b * a.m * a.m;
    if (!a.use_sigma_w) {
        // forward substitution Lw y = x  ->  y^T y = x^T Sigma_w^-1 x   (spectrum_model.py:334-335)
        for (int k = 0; k < a.m; ++k) {
            double v = xs[k];
            for (int j = 0; j < k; ++j) v -= Lw[k * a.m + j] * xs[j];
            xs[k] = v / Lw[k * a.m + k];
            Yb[(int64_t)k * a.ldy + i] = xs[k];
        }
    } else {
        // y = Lw^T x  ->  y^T y = x^T Sigma_w x   (the form printed in the paper / docs)
        for (int k = 0; k < a.m; ++k) {
            double v = 0.0;
            for (int j = k; j < a.m; ++j) v += Lw[j * a.m + k] * xs[j];
            Yb[(int64_t)k * a.ldy + i] = v;
        }
    }
    for (int k = a.m; k < a.mpad; ++k) Yb[(int64_t)k * a.ldy + i] = 0.0;
}

__global__ __launch_bounds__(256) void k_eval_rows(sf_eval_args a) {
    const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (i >= a.n) return;
    if (a.info && a.info[b] != 0) return;
    double xk[SF_MAX_M];
    const double flux = sf_eval_pixel(a, b, i, xk);
    double* __restrict__ Xb = a.X + (int64_t)b * a.m * a.ldx;
    for (int k = 0; k < a.m; ++k) Xb[(int64_t)k * a.ldx + i] = xk[k];
    a.flux[(int64_t)b * a.ldx + i] = flux;
}

__device__ __forceinline__ double sf_block_sum(double v, double* red) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double tot = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) tot += red[w];
    return tot;
}

// One workgroup per walker: scale factor Omega (spectrum_model.py:316-329, transforms.py:265-268)
__global__ __launch_bounds__(256) void k_scale(sf_scale_args a) {
    __shared__ double red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const double* __restrict__ P = a.params + (int64_t)b * a.pstride;
    const double norm = P[3];
    double scale, lscale;
    if (a.has_log_scale) {
        lscale = P[2];
        scale = exp(lscale) * norm;
    } else {
        const double* __restrict__ f = a.flux + (int64_t)b * a.ldx;
        double sd = 0.0, sm = 0.0;
        for (int i = tid; i + 1 < a.n; i += 256) {
            const double d = a.wave[i + 1] - a.wave[i];
            sd += d * (a.dflux[i + 1] + a.dflux[i]) / 2.0;
            sm += d * (f[i + 1] * norm + f[i] * norm) / 2.0;
        }
        sd = sf_block_sum(sd, red);
        sm = sf_block_sum(sm, red);
        scale = sd / sm;
        lscale = log(scale);
        scale = scale * norm;
    }
    if (tid == 0) {
        a.scale[b] = scale;
        if (a.log_scale_out) a.log_scale_out[b] = lscale;
    }
}

// Elementwise: rescale flux and X, residual, and Y = Lw^-1 X (or Lw^T X), zero padded.
__global__ __launch_bounds__(256) void k_resid_y(sf_resid_args a) {
    const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (i >= a.ldy) return;
    double* __restrict__ Yb = a.Y ? a.Y + (int64_t)b * a.mpad * a.ldy : nullptr;
    if (i >= a.n || (a.info && a.info[b] != 0)) {
        if (Yb)
            for (int k = 0; k < a.mpad; ++k) Yb[(int64_t)k * a.ldy + i] = 0.0;
        if (i < a.ldx && a.resid) a.resid[(int64_t)b * a.ldx + i] = 0.0;
        return;
    }
    const double sc = a.scale[b];
    const double f = a.flux[(int64_t)b * a.ldx + i] * sc;  // transforms.py:231
    if (a.flux_out) a.flux_out[(int64_t)b * a.n + i] = f;
    if (a.resid) a.resid[(int64_t)b * a.ldx + i] = f - a.dflux[i];  // spectrum_model.py:402
    const double* __restrict__ Xb = a.X + (int64_t)b * a.m * a.ldx;
    double xs[SF_MAX_M];
    for (int k = 0; k < a.m; ++k) {
        xs[k] = Xb[(int64_t)k * a.ldx + i] * sc;
        if (a.X_out) a.X_out[((int64_t)b * a.m + k) * a.n + i] = xs[k];
    }
    if (!Yb) return;
    sf_y_column(a, b, i, xs, Yb);
}

// k_eval_rows + k_scale (log_scale given) + k_resid_y in one pass over the pixels: the same operations in the same order, X and
// the unscaled flux stay in registers (banded step, B = 128: 75 + 6 + 50 us of launches -> one)
__global__ __launch_bounds__(256) void k_eval_resid_y(sf_eval_args e, sf_resid_args a, double* __restrict__ scale_out,
                                                      double* __restrict__ log_scale_out) {
    const int i = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    const double* __restrict__ P = e.params + (int64_t)b * e.pstride;
    const double lscale = P[2], sc = exp(lscale) * P[3];  // (k_scale: scale = exp(log_scale) * norm)
    if (i == 0) {
        scale_out[b] = sc;
        if (log_scale_out) log_scale_out[b] = lscale;
    }
    if (i >= a.ldy) return;
    double* __restrict__ Yb = a.Y ? a.Y + (int64_t)b * a.mpad * a.ldy : nullptr;
    if (i >= a.n || (e.info && e.info[b] != 0)) {
        if (Yb)
            for (int k = 0; k < a.mpad; ++k) Yb[(int64_t)k * a.ldy + i] = 0.0;
        if (i < a.ldx && a.resid) a.resid[(int64_t)b * a.ldx + i] = 0.0;
        return;
    }
    double xs[SF_MAX_M];
    const double f = sf_eval_pixel(e, b, i, xs) * sc;  // transforms.py:231
    if (a.flux_out) a.flux_out[(int64_t)b * a.n + i] = f;
    if (a.resid) a.resid[(int64_t)b * a.ldx + i] = f - a.dflux[i];  // spectrum_model.py:402
    for (int k = 0; k < a.m; ++k) {
        xs[k] = xs[k] * sc;
        if (a.X_out) a.X_out[((int64_t)b * a.m + k) * a.n + i] = xs[k];
    }
    if (!Yb) return;
    sf_y_column(a, b, i, xs, Yb);
}

// Chebyshev free function
__global__ __launch_bounds__(256) void k_cheb_rows(const double* __restrict__ wave, int n, double wave_max,
                                                   const double* __restrict__ flux, int rows,
                                                   const double* __restrict__ coeffs, int ncoef,
                                                   double* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double p = sf_chebval(wave[i] / wave_max, coeffs + 1, ncoef, coeffs[0]);
    for (int r = 0; r < rows; ++r) out[(int64_t)r * n + i] = flux[(int64_t)r * n + i] * p;
}

// ------------------------------------------------------------------------------------ emulator
// emulator.py:376-388 with the constant v11 factored once:
//   v11 = Lc Lc^T, alpha = v11^-1 w_hat, Linv = Lc^-1 (lower);  z = Linv v12;
//   mu = v12^T alpha;  cov = v22 - z^T z.   Also returns Lw = chol(cov) for the rank-m factor.
// Three launches for the whole batch:
//   k_emu_prep   per walker: range check, the v12 blocks k_i[j] (kernels.py:25-26) -> kbuf, mu
//   k_emu_z      z[b][r][i] = sum_j Linv[r][i M + j] k_i[b][j] as a TILED product: a workgroup owns 256 rows r of one
//                component i and 8 walkers; Linv^T is stored (row index fastest) so that the lanes read it coalesced,
//                every element loaded once serves 8 walkers from registers, the k_i of the 8 walkers sit in LDS.
//                (One workgroup per walker streaming its own copy of Linv -- 13.9 MB at the reference's worked
//                example m = 4, M = 330 -- took 0.5 ms per 128 walkers.)
//   k_emu_post   per walker: cov = v22 - z^T z, chol(cov)
#define EMU_WCHUNK 8
__device__ __forceinline__ double sf_wave_sum_t(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}
__global__ __launch_bounds__(256) void k_emu_prep(sf_emu_args a) {
    __shared__ int bad;
    __shared__ double red[4];
    const int b = blockIdx.x, tid = threadIdx.x;
    const double* __restrict__ P = a.params + (int64_t)b * a.pstride + a.off_grid;
    const int mM = a.m * a.M;
    if (tid == 0) {
        bad = 0;
        for (int d = 0; d < a.P; ++d)
            if (P[d] < a.gmin[d] || P[d] > a.gmax[d]) bad = 1;  // emulator.py:377-378
    }
    __syncthreads();
    double* __restrict__ kv = a.kbuf + (int64_t)b * mM;
    if (bad) {
        if (tid == 0 && a.info) a.info[b] = SF_INFO_OUT_OF_GRID;
        for (int e = tid; e < mM; e += 256) kv[e] = 0.0;  // keeps the batched product finite
        return;
    }
    // v12 blocks: kernels.py:25-26 (cdist of X/l and Z/l, sqeuclidean)
    for (int e = tid; e < mM; e += 256) {
        const int i = e / a.M, j = e - i * a.M;
        double d2 = 0.0;
        for (int d = 0; d < a.P; ++d) {
            const double l = a.lengthscales[i * a.P + d];
            const double df = a.grid[j * a.P + d] / l - P[d] / l;
            d2 = d2 + df * df;
        }
        kv[e] = a.variances[i] * exp(-0.5 * d2);
    }
    __syncthreads();
    for (int i = 0; i < a.m; ++i) {
        double acc = 0.0;
        for (int j = tid; j < a.M; j += 256) acc += kv[i * a.M + j] * a.alpha[i * a.M + j];
        acc = sf_block_sum(acc, red);
        if (tid == 0) a.mu[(int64_t)b * a.m + i] = acc;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_emu_z(sf_emu_args a, int B) {
    extern __shared__ double ksm[];  // EMU_WCHUNK x M: k_i of this chunk's walkers
    const int mM = a.m * a.M;
    const int i = blockIdx.y, b0 = blockIdx.z * EMU_WCHUNK;
    const int r = blockIdx.x * 256 + threadIdx.x;
    for (int e = threadIdx.x; e < EMU_WCHUNK * a.M; e += 256) {
        const int w = e / a.M, j = e - w * a.M;
        ksm[e] = (b0 + w < B) ? a.kbuf[(int64_t)(b0 + w) * mM + i * a.M + j] : 0.0;
    }
    __syncthreads();
    if (r >= mM) return;
    double acc[EMU_WCHUNK];
#pragma unroll
    for (int w = 0; w < EMU_WCHUNK; ++w) acc[w] = 0.0;
    // columns beyond r are zero in Linv (lower triangular): j <= r - i M
    const int jmax = min(a.M, r - i * a.M + 1);
    const double* __restrict__ lt = a.LinvT + (int64_t)i * a.M * mM + r;  // LinvT[c][r] = Linv[r][c]
    for (int j = 0; j < jmax; ++j) {
        const double l = lt[(int64_t)j * mM];
#pragma unroll
        for (int w = 0; w < EMU_WCHUNK; ++w) acc[w] += l * ksm[w * a.M + j];
    }
#pragma unroll
    for (int w = 0; w < EMU_WCHUNK; ++w)
        if (b0 + w < B) a.zscratch[((int64_t)(b0 + w) * mM + r) * a.m + i] = acc[w];
}

__global__ __launch_bounds__(256) void k_emu_post(sf_emu_args a) {
    extern __shared__ double esm[];
    double* covs = esm;  // m x m
    const int b = blockIdx.x, tid = threadIdx.x;
    if (a.info && a.info[b] != 0) return;
    const int mM = a.m * a.M;
    const double* __restrict__ z = a.zscratch + (int64_t)b * mM * a.m;
    // z^T z over the m M rows: every thread takes rows tid, tid + 256, ... and keeps a chunk of up to EMU_PAIRS
    // (i, j <= i) partial sums in registers; waves fold with shuffles, the four wave sums meet in LDS (fixed order)
    constexpr int EMU_PAIRS = 36;
    __shared__ double wsum[4][EMU_PAIRS];
    const int npairs = a.m * (a.m + 1) / 2;
    for (int p0 = 0; p0 < npairs; p0 += EMU_PAIRS) {
        const int np = min(EMU_PAIRS, npairs - p0);
        double acc[EMU_PAIRS];
#pragma unroll
        for (int q = 0; q < EMU_PAIRS; ++q) acc[q] = 0.0;
        // first pair of the chunk -> (i0, j0), row-major over the lower triangle
        int i0 = 0;
        while ((i0 + 1) * (i0 + 2) / 2 <= p0) ++i0;
        const int j0 = p0 - i0 * (i0 + 1) / 2;
        for (int r = tid; r < mM; r += 256) {
            const double* zr = z + (int64_t)r * a.m;
            int i = i0, j = j0;
#pragma unroll
            for (int q = 0; q < EMU_PAIRS; ++q) {
                if (q < np) {
                    acc[q] += zr[i] * zr[j];
                    if (++j > i) {
                        ++i;
                        j = 0;
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < EMU_PAIRS; ++q) {
            const double v = sf_wave_sum_t(acc[q]);
            if ((tid & 63) == 0) wsum[tid >> 6][q] = v;
        }
        __syncthreads();
        if (tid < np) {
            int i = i0, j = j0;
            for (int q = 0; q < tid; ++q)
                if (++j > i) {
                    ++i;
                    j = 0;
                }
            const double tot = wsum[0][tid] + wsum[1][tid] + wsum[2][tid] + wsum[3][tid];
            const double v = ((i == j) ? a.variances[i] : 0.0) - tot;  // v22 is diag(variances) at a single point
            covs[i * a.m + j] = v;
            covs[j * a.m + i] = v;
        }
        __syncthreads();
    }
    if (a.cov)
        for (int e = tid; e < a.m * a.m; e += 256) a.cov[(int64_t)b * a.m * a.m + e] = covs[e];
    __syncthreads();
    if (a.Lw) {
        // small dense Cholesky of Sigma_w (spectrum_model.py:334 cho_factor(weights_cov)): factor in LDS (in
        // place in `covs`, lower triangle), one thread, then a parallel copy-out with the upper part zeroed
        __shared__ int fail;
        if (tid == 0) {
            fail = 0;
            const int m = a.m;
            for (int j = 0; j < m; ++j) {
                double d = covs[j * m + j];
                for (int k = 0; k < j; ++k) d -= covs[j * m + k] * covs[j * m + k];
                if (!(d > 0.0)) { fail = 1; d = 1.0; }
                const double dj = sqrt(d), rj = 1.0 / dj;
                covs[j * m + j] = dj;
                for (int i = j + 1; i < m; ++i) {
                    double v = covs[i * m + j];
                    for (int k = 0; k < j; ++k) v -= covs[i * m + k] * covs[j * m + k];
                    covs[i * m + j] = v * rj;
                }
            }
            if (fail && a.info) a.info[b] = SF_INFO_BAD_WEIGHT_COV;
        }
        __syncthreads();
        double* L = a.Lw + (int64_t)b * a.m * a.m;
        for (int e = tid; e < a.m * a.m; e += 256) {
            const int i = e / a.m, j = e - i * a.m;
            L[e] = j <= i ? covs[e] : 0.0;
        }
    }
}

// Joint GP conditional over B query points (Emulator.__call__ with several parameter rows,
// emulator.py:382-389): with z_b = Linv v12_b left in zscratch by k_emu_z,
//   cov[(i,a),(j,b)] = delta_ij var_i exp(-1/2 |(p_a - p_b)/l_i|^2) - z_a[:, i] . z_b[:, j],
// indices component-major (i*B + a) as produced by the reference's block-diagonal batch_kernel.
__global__ __launch_bounds__(256) void k_emu_joint(sf_emu_args a, int B, const double* __restrict__ mu_pts,
                                                   double* __restrict__ mu, double* __restrict__ cov) {
    const int n = a.m * B, mM = a.m * a.M;
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e < n) {
        const int i = (int)(e / B), pa = (int)(e - (int64_t)i * B);
        mu[e] = mu_pts[pa * a.m + i];
    }
    if (e >= (int64_t)n * n) return;
    const int I = (int)(e / n), J = (int)(e - (int64_t)I * n);
    const int i = I / B, pa = I - i * B, j = J / B, pb = J - j * B;
    const double* za = a.zscratch + (int64_t)pa * mM * a.m + i;
    const double* zb = a.zscratch + (int64_t)pb * mM * a.m + j;
    double acc = 0.0;
    for (int r = 0; r < mM; ++r) acc += za[(int64_t)r * a.m] * zb[(int64_t)r * a.m];
    double v22 = 0.0;
    if (i == j) {
        const double* Pa = a.params + (int64_t)pa * a.pstride + a.off_grid;
        const double* Pb = a.params + (int64_t)pb * a.pstride + a.off_grid;
        double d2 = 0.0;
        for (int d = 0; d < a.P; ++d) {
            const double l = a.lengthscales[i * a.P + d];
            const double df = Pa[d] / l - Pb[d] / l;
            d2 += df * df;
        }
        v22 = a.variances[i] * exp(-0.5 * d2);
    }
    cov[e] = v22 - acc;
}

int sf_launch_emu_joint(const sf_emu_args& a, int B, const double* mu_pts, double* mu, double* cov, hipStream_t s) {
    const int64_t n = (int64_t)a.m * B, total = n * n;
    hipLaunchKernelGGL(k_emu_joint, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a, B, mu_pts, mu, cov);
    SF_LAUNCH_CHECK();
    return SF_OK;
}

// d_lnl = -(logdet + sqmah)/2, -inf where info != 0   (spectrum_model.py:405)
__global__ void k_finish(int B, const double* __restrict__ logdet, const double* __restrict__ sqmah,
                         const int* __restrict__ info, const int* __restrict__ info2,
                         double* __restrict__ lnl, int* __restrict__ info_out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int code = info ? info[b] : 0;
    if (code == 0 && info2) code = info2[b];
    double v = -(logdet[b] + sqmah[b]) / 2;
    if (code == 0 && !(v == v)) code = SF_INFO_NAN;
    if (code != 0) v = -INFINITY;
    lnl[b] = v;
    if (info_out) info_out[b] = code;
}

// ------------------------------------------------------------------------------------ launchers
static const size_t kLdsFftMax = 8192;  // complex points that fit the 160 KiB LDS (128 KiB)

size_t sf_fft_scratch_bytes(int rows_total, int nf) {  // full-size transform (free functions, set-up)
    return (size_t)nf > kLdsFftMax ? sizeof(double2) * (size_t)rows_total * nf : 0;
}
size_t sf_fft_half_scratch_bytes(int rows_total, int nf) {  // half-size transform of the hot path
    return (size_t)(nf / 2) > kLdsFftMax ? sizeof(double2) * (size_t)rows_total * nf : 0;
}

template <bool FWD>
static int launch_broaden_t(const sf_broaden_args& a, hipStream_t s) {
    const bool lds = (size_t)a.nf <= kLdsFftMax;
    const size_t shm = lds ? sizeof(double2) * (size_t)a.nf : 0;
    dim3 grid(a.rows, a.B);
    if (lds) {
        static sf_dev_once attr_once;  // devices whose function attributes are set
        SF_CHECK(sf_once_per_device(&attr_once, []() -> int {
            SF_HIP(hipFuncSetAttribute((const void*)k_broaden<true, true>,
            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            SF_HIP(hipFuncSetAttribute((const void*)k_broaden<false, true>,
            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            return SF_OK;
        }));
        hipLaunchKernelGGL((k_broaden<FWD, true>), grid, dim3(256), shm, s, a.in, a.spec, a.rows, a.nf, a.tw,
                           a.dv, a.kind, a.params, a.pstride, a.poff, a.scalar_param, a.out, a.ob, a.orow,
                           a.oelem, (double2*)nullptr, a.info);
    } else {
        if (!a.gscratch) {
            sf_set_error("broaden: nf=%d needs a global FFT scratch buffer", a.nf);
            return SF_ENOMEM;
        }
        hipLaunchKernelGGL((k_broaden<FWD, false>), grid, dim3(256), 0, s, a.in, a.spec, a.rows, a.nf, a.tw,
                           a.dv, a.kind, a.params, a.pstride, a.poff, a.scalar_param, a.out, a.ob, a.orow,
                           a.oelem, a.gscratch, a.info);
    }
    SF_LAUNCH_CHECK();
    return SF_OK;
}

static int launch_broaden_half(const sf_broaden_args& a, hipStream_t s) {
    const int nh1 = a.nf / 2 + 1;
    const double val = 1.0 / (a.nf * a.dv);  // numpy.fft.rfftfreq
    hipLaunchKernelGGL(k_kernel_mult, dim3((nh1 + 255) / 256, a.B), dim3(256), 0, s, a.mult, nh1, val, a.kind,
                       a.params, a.pstride, a.poff, a.scalar_param, a.info);
    SF_LAUNCH_CHECK();
    const bool lds = (size_t)(a.nf / 2) <= kLdsFftMax;
    dim3 grid(a.rows, a.B);
    if (lds) {
        static sf_dev_once attr_once;  // devices whose function attributes are set
        SF_CHECK(sf_once_per_device(&attr_once, []() -> int {
            SF_HIP(hipFuncSetAttribute((const void*)k_broaden_half<true>,
            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            return SF_OK;
        }));
        hipLaunchKernelGGL(k_broaden_half<true>, grid, dim3(256), sizeof(double2) * (size_t)(a.nf / 2), s, a.spec,
                           a.mult, a.rows, a.nf, a.tw, a.kind, a.params, a.pstride, a.poff, a.scalar_param, a.out,
                           a.ob, a.orow, a.oelem, (double2*)nullptr);
    } else {
        if (!a.gscratch) {
            sf_set_error("broaden: nf=%d needs a global FFT scratch buffer", a.nf);
            return SF_ENOMEM;
        }
        hipLaunchKernelGGL(k_broaden_half<false>, grid, dim3(256), 0, s, a.spec, a.mult, a.rows, a.nf, a.tw, a.kind,
                           a.params, a.pstride, a.poff, a.scalar_param, a.out, a.ob, a.orow, a.oelem, a.gscratch);
    }
    SF_LAUNCH_CHECK();
    return SF_OK;
}

int sf_launch_broaden(const sf_broaden_args& a, hipStream_t s) {
    if (a.nf < 4 || (a.nf & (a.nf - 1)) || a.nf > 65536) {
        sf_set_error("broaden: nf=%d must be a power of two in [4, 65536]", a.nf);
        return SF_EINVAL;
    }
    if (!a.in && a.mult) return launch_broaden_half(a, s);
    return a.in ? launch_broaden_t<true>(a, s) : launch_broaden_t<false>(a, s);
}

int sf_launch_rfft_rows(const double* in, int rows, int nf, const double2* tw, double2* spec,
                        double2* gscratch, hipStream_t s) {
    const bool lds = (size_t)nf <= kLdsFftMax;
    if (lds) {
        static sf_dev_once attr_once;  // devices whose function attributes are set
        SF_CHECK(sf_once_per_device(&attr_once, []() -> int {
            SF_HIP(hipFuncSetAttribute((const void*)k_rfft_rows<true>,
            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            return SF_OK;
        }));
        hipLaunchKernelGGL(k_rfft_rows<true>, dim3(rows), dim3(256), sizeof(double2) * (size_t)nf, s, in, nf, tw,
                           spec, (double2*)nullptr);
    } else {
        hipLaunchKernelGGL(k_rfft_rows<false>, dim3(rows), dim3(256), 0, s, in, nf, tw, spec, gscratch);
    }
    SF_LAUNCH_CHECK();
    return SF_OK;
}

int sf_launch_spline_solve(double* data, int B, int rows, int64_t bstride, int64_t rstride,
                           int64_t estride, int n, const double* Lf, const double* Uf, const double* rdiag,
                           hipStream_t s) {
    const int nsys = B * rows;
    hipLaunchKernelGGL(k_spline_solve, dim3((nsys + 63) / 64), dim3(64), 0, s, data, nsys, rows, bstride,
                       rstride, estride, n, Lf, Uf, rdiag);
    SF_LAUNCH_CHECK();
    return SF_OK;
}

int sf_launch_spline_apply(const double* y, double* c, int B, int rows, int n, const double* tblk, hipStream_t s) {
    const int ncb = (rows + 15) / 16;
    if (ncb > 2 || n % 16) {
        sf_set_error("spline_apply: rows=%d n=%d not supported", rows, n);
        return SF_EINVAL;
    }
    // enough waves to fill the chip (n/16 output blocks x walker chunks), long enough chunks to amortise
    // the register-resident coefficient blocks
    int wchunk = 32;
    while (wchunk > 4 && (int64_t)(n / 16) * ((B + wchunk - 1) / wchunk) < 2048) wchunk >>= 1;
    const dim3 grid((n / 16 + 3) / 4, (B + wchunk - 1) / wchunk);
    if (ncb == 1) hipLaunchKernelGGL(k_spline_apply<1>, grid, dim3(256), 0, s, y, c, rows, n, tblk, B, wchunk);
    else hipLaunchKernelGGL(k_spline_apply<2>, grid, dim3(256), 0, s, y, c, rows, n, tblk, B, wchunk);
    SF_LAUNCH_CHECK();
    return SF_OK;
}

int sf_launch_spline_eval(const double* coef, int rows, int ncoef, const double* t, const double* xq, int nq,
                          double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_spline_eval, dim3((nq + 255) / 256), dim3(256), 0, s, coef, rows, ncoef, t, xq, nq,
                       out);
    SF_LAUNCH_CHECK();
    return SF_OK;
}

int sf_launch_eval_rows(const sf_eval_args& a, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_eval_rows, dim3((a.n + 255) / 256, B), dim3(256), 0, s, a);
    SF_LAUNCH_CHECK();
    return SF_OK;
}

int sf_launch_scale(const sf_scale_args& a, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_scale, dim3(B), dim3(256), 0, s, a);
    SF_LAUNCH_CHECK();
    return SF_OK;
}

int sf_launch_resid_y(const sf_resid_args& a, int B, hipStream_t s) {
    if (a.m > SF_MAX_M) {
        sf_set_error("at most %d eigenspectra are supported", SF_MAX_M);
        return SF_EINVAL;
    }
    hipLaunchKernelGGL(k_resid_y, dim3((a.ldy + 255) / 256, B), dim3(256), 0, s, a);
    SF_LAUNCH_CHECK();
    return SF_OK;
}

int sf_launch_eval_resid_y(const sf_eval_args& e, const sf_resid_args& r, double* scale_out, double* log_scale_out, int B,
                           hipStream_t s) {
    if (r.m > SF_MAX_M) {
        sf_set_error("at most %d eigenspectra are supported", SF_MAX_M);
        return SF_EINVAL;
    }
    hipLaunchKernelGGL(k_eval_resid_y, dim3((r.ldy + 255) / 256, B), dim3(256), 0, s, e, r, scale_out, log_scale_out);
    SF_LAUNCH_CHECK();
    return SF_OK;
}

int sf_launch_extinct_rows(const double* wave, int n, const double* flux, int rows, double Av, double Rv, int law,
                           double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_extinct_rows, dim3((n + 255) / 256), dim3(256), 0, s, wave, n, flux, rows, Av, Rv, law, out);
    SF_LAUNCH_CHECK();
    return SF_OK;
}

int sf_launch_extinct_spline_rows(const double* wave, int n, const double* flux, int rows, double Av, double Rv,
                                  const double* d_table, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_extinct_spline_rows, dim3((n + 255) / 256), dim3(256), 0, s, wave, n, flux, rows, Av, Rv, d_table, out);
    SF_LAUNCH_CHECK();
    return SF_OK;
}

int sf_launch_cheb_rows(const double* wave, int n, double wave_max, const double* flux, int rows,
                        const double* d_coeffs, int ncoef, double* out, hipStream_t s) {
    hipLaunchKernelGGL(k_cheb_rows, dim3((n + 255) / 256), dim3(256), 0, s, wave, n, wave_max, flux, rows,
                       d_coeffs, ncoef, out);
    SF_LAUNCH_CHECK();
    return SF_OK;
}

int sf_launch_emulator(const sf_emu_args& a, int B, hipStream_t s) {
    const size_t shm_z = sizeof(double) * (size_t)EMU_WCHUNK * a.M;
    if (shm_z > 64 * 1024 || a.m > SF_MAX_M) {
        sf_set_error("emulator: M=%d / m=%d too large", a.M, a.m);
        return SF_EINVAL;
    }
    if (!a.kbuf) {
        sf_set_error("emulator: the v12 scratch is missing");
        return SF_EINVAL;
    }
    const int mM = a.m * a.M;
    hipLaunchKernelGGL(k_emu_prep, dim3(B), dim3(256), 0, s, a);
    SF_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_emu_z, dim3((mM + 255) / 256, a.m, (B + EMU_WCHUNK - 1) / EMU_WCHUNK), dim3(256), shm_z, s, a, B);
    SF_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_emu_post, dim3(B), dim3(256), sizeof(double) * (size_t)a.m * a.m, s, a);
    SF_LAUNCH_CHECK();
    return SF_OK;
}

int sf_launch_finish(int B, const double* logdet, const double* sqmah, const int* info, const int* info2,
                     double* lnl, int* info_out, hipStream_t s) {
    hipLaunchKernelGGL(k_finish, dim3((B + 255) / 256), dim3(256), 0, s, B, logdet, sqmah, info, info2, lnl,
                       info_out);
    SF_LAUNCH_CHECK();
    return SF_OK;
}


// ---------------------------------------------------------------------------------------------
// v11 = iPhiPhi / lambda_xi + blockdiag_c( variance_c exp(-1/2 |(x_i - x_j) / lengthscale_c|^2) )  of the emulator's
// training likelihood (Starfish/emulator/emulator.py:126-128,569-571; kernels.py:5-49), built ON the device into the
// padded layout the batched Cholesky takes (identity block from n = m M to npad): Emulator.train evaluates it once per
// objective call, and the host build + upload of the 1320 x 1320 matrix of the worked example cost 5x the
// factorisation.  hyper = [lambda_xi, variances[m], lengthscales[m][P]] (device).  Operation order of the reference:
// (x / l) differences squared and summed over the parameters in order, -0.5 * d2, exp, times the variance, added to
// iPhiPhi / lambda_xi.
__global__ __launch_bounds__(256) void k_v11_build(const double* __restrict__ grid, int M, int P, int m,
                                                   const double* __restrict__ hyper, const double* __restrict__ iphiphi,
                                                   double* __restrict__ A, int npad, int lda) {
    const int j = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
    if (j >= npad) return;
    const int n = m * M;
    double v;
    if (i < n && j < n) {
        v = iphiphi[(int64_t)i * n + j] / hyper[0];
        const int ci = i / M, cj = j / M;
        if (ci == cj) {
            const double* gi = grid + (int64_t)(i - ci * M) * P;
            const double* gj = grid + (int64_t)(j - cj * M) * P;
            const double* ls = hyper + 1 + m + ci * P;
            double d2 = 0.0;
            for (int p = 0; p < P; ++p) {
                const double d = gi[p] / ls[p] - gj[p] / ls[p];
                d2 = d2 + d * d;
            }
            v = v + hyper[1 + ci] * exp(-0.5 * d2);
        }
    } else {
        v = (i == j) ? 1.0 : 0.0;
    }
    A[(int64_t)i * lda + j] = v;
}
int sf_launch_v11_build(const double* grid, int M, int P, int m, const double* hyper, const double* iphiphi, double* A, int npad,
                        int lda, hipStream_t s) {
    if (!grid || !hyper || !iphiphi || !A || M <= 0 || P <= 0 || m <= 0 || npad < m * M || lda < npad) {
        sf_set_error("v11_build: bad arguments");
        return SF_EINVAL;
    }
    hipLaunchKernelGGL(k_v11_build, dim3((npad + 255) / 256, npad), dim3(256), 0, s, grid, M, P, m, hyper, iphiphi, A, npad, lda);
    SF_LAUNCH_CHECK();
    return SF_OK;
}
