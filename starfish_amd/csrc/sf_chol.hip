// Batched fp64 Cholesky (lower, row-major, in place) + logdet / Mahalanobis solve for gfx950.
//
// Replaces the LAPACK calls of the reference: scipy.linalg.cho_factor / cho_solve at
// Starfish/models/spectrum_model.py:400-404 (dpotrf + dpotrs on the N x N covariance).
//
// Structure (one launch sequence serves the whole batch; the batch supplies the parallelism):
//   outer LEFT-looking panels of SF_NB columns:   panel -= L[:, :k] L[k-block, :k]^T   (k_gemm_nt,
//     v_mfma_f64_16x16x4_f64 tiles, the >90 % flops part, long K so C is read/written once)
//   inside a panel, 64-column steps:  k_potrf_leaf (64x64 in LDS)  ->  k_trsm_leaf (row-per-lane
//     substitution, x in registers, L^T broadcast from LDS)  ->  k_gemm_nt with K = 64.
//   k_trsv_logdet: one forward substitution L z = R per matrix (sqmah = z.z) and 2 sum log L_ii.
#include "sf_common.h"

#define GT 128  // C tile edge of the MFMA kernel
#define GK 16   // K slab staged in LDS per step
#define GLD 18  // LDS row stride (doubles): conflict-free ds_read_b64 fragments, 16-B aligned rows

// Logical block id such that ids adjacent in work space run on the same XCD (block b is observed on
// XCD b % 8; each XCD has its own L2).  Bijective for any grid size; placement only affects speed.
__device__ __forceinline__ int sf_xcd_remap(int bid, int nblk) {
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

// C[r0.., c0..] -= Mx[r0.., k0..k0+K) * Mx[c0.., k0..k0+K)^T  on the same row-major matrix.
// M x Nc block, K a multiple of GK.  tri != 0: the block is diagonal-aligned (r0 == c0) and tiles
// lying entirely above the diagonal are skipped.
__global__ __launch_bounds__(256, 2) void k_gemm_nt(double* __restrict__ base, int lda, int64_t stride,
                                                     int r0, int c0, int k0, int M, int Nc, int K,
                                                     int tri, int mt, int nt) {
    __shared__ __attribute__((aligned(16))) double As[2][GT * GLD];
    __shared__ __attribute__((aligned(16))) double Bs[2][GT * GLD];

    const int id = sf_xcd_remap(blockIdx.x, gridDim.x);
    const int tiles = mt * nt;
    const int b = id / tiles;
    const int t = id - b * tiles;
    const int tm = t / nt, tn = t - tm * nt;
    const int row0 = r0 + tm * GT, col0 = c0 + tn * GT;
    if (tri && col0 > row0 + GT - 1) return;

    double* __restrict__ Mx = base + (int64_t)b * stride;
    const int rows_here = min(GT, r0 + M - row0);
    const int cols_here = min(GT, c0 + Nc - col0);

    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;
    const int l15 = lane & 15, lq = lane >> 4;

    // ---- accumulators start as the C tile (D = A*B + C with A negated)
    sf_d4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int col = wn * 64 + ni * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * 64 + mi * 16 + lq + 4 * r;
                double v = 0.0;
                if (row < rows_here && col < cols_here)
                    v = Mx[(int64_t)(row0 + row) * lda + col0 + col];
                acc[mi][ni][r] = v;
            }
        }

    // ---- global -> register -> LDS staging: thread covers rows lr+32p, two doubles at column lc
    const int lr = tid >> 3, lc = (tid & 7) * 2;
    const double* Ag = Mx + (int64_t)(row0 + lr) * lda + k0 + lc;
    const double* Bg = Mx + (int64_t)(col0 + lr) * lda + k0 + lc;
    double2 ra[4], rb[4];

    auto gload = [&](int kt) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int rr = lr + 32 * p;
            ra[p] = (rr < rows_here) ? *(const double2*)(Ag + (int64_t)(32 * p) * lda + kt * GK)
                                     : make_double2(0.0, 0.0);
            rb[p] = (rr < cols_here) ? *(const double2*)(Bg + (int64_t)(32 * p) * lda + kt * GK)
                                     : make_double2(0.0, 0.0);
        }
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            *(double2*)&As[buf][(lr + 32 * p) * GLD + lc] = ra[p];
            *(double2*)&Bs[buf][(lr + 32 * p) * GLD + lc] = rb[p];
        }
    };

    const int nk = K / GK;
    gload(0);
    lstore(0);
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const double* Ab = &As[cur][(wm * 64 + l15) * GLD + lq];
        const double* Bb = &Bs[cur][(wn * 64 + l15) * GLD + lq];
#pragma unroll
        for (int ks = 0; ks < GK / 4; ++ks) {
            double a[4], bb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = -Ab[i * 16 * GLD + ks * 4];
                bb[i] = Bb[i * 16 * GLD + ks * 4];
            }
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], bb[ni], acc[mi][ni], 0, 0, 0);
        }
        if (kt + 1 < nk) lstore(cur ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int col = wn * 64 + ni * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * 64 + mi * 16 + lq + 4 * r;
                if (row < rows_here && col < cols_here)
                    Mx[(int64_t)(row0 + row) * lda + col0 + col] = acc[mi][ni][r];
            }
        }
}

__device__ __forceinline__ double sf_readlane_d(double v, int srclane) {
    union { double d; int i[2]; } u;
    u.d = v;
    u.i[0] = __builtin_amdgcn_readlane(u.i[0], srclane);
    u.i[1] = __builtin_amdgcn_readlane(u.i[1], srclane);
    return u.d;
}

// 64 x 64 diagonal block at (c, c), ONE wave per matrix: lane r keeps row r in registers and the
// unblocked right-looking Cholesky (dpotf2 order: pivot sqrt, column scale by the reciprocal pivot,
// rank-1 update) broadcasts column entries with v_readlane -- no LDS, no barriers.
// Besides L (written in place) it leaves Lt[k][j] = L[j][k] (j > k), Lt[k][k] = 1 / L[k][k] in the
// read-only side buffer `ltbuf` that k_trsm_leaf fetches through the scalar cache.
// With a right-hand side (rhs != NULL) the forward substitution L z = R rides along: lane r carries
// R[c + r]; after column k is final, z_k = R_k / L_kk is broadcast and R_r -= L_rk z_k (r > k).
#define SF_LTB (SF_LEAF * SF_LEAF + SF_LEAF)  // doubles per matrix in the side buffer: Lt + z
__global__ __launch_bounds__(64) void k_potrf_leaf(double* __restrict__ base, int lda, int64_t stride,
                                                   int c, int* __restrict__ info,
                                                   double* __restrict__ ltbuf, double* __restrict__ rhs,
                                                   int ldr) {
    const int b = blockIdx.x, r = threadIdx.x;
    double* D = base + (int64_t)b * stride + (int64_t)c * lda + c;
    double* prow = D + (int64_t)r * lda;
    double a[SF_LEAF];
#pragma unroll
    for (int j = 0; j < SF_LEAF; j += 2) {
        const double2 v = *(const double2*)(prow + j);
        a[j] = v.x;
        a[j + 1] = v.y;
    }
    int bad = 0;
    double* lt = ltbuf + (int64_t)b * SF_LTB;
    double rv = rhs ? rhs[(int64_t)b * ldr + c + r] : 0.0;
#pragma unroll
    for (int k = 0; k < SF_LEAF; ++k) {
        const double akk = sf_readlane_d(a[k], k);
        if (!(akk > 0.0) && !bad) bad = c + k + 1;
        const double d = sqrt(akk);
        const double inv = 1.0 / d;
        a[k] = (r > k) ? a[k] * inv : ((r == k) ? d : 0.0);
        lt[k * SF_LEAF + r] = (r == k) ? inv : a[k];
        const double zk = sf_readlane_d(rv, k) * inv;
        rv = (r > k) ? rv - a[k] * zk : ((r == k) ? zk : rv);
#pragma unroll
        for (int j = k + 1; j < SF_LEAF; ++j) {
            const double ljk = sf_readlane_d(a[k], j);
            a[j] -= a[k] * ljk;
        }
    }
    // lower triangle back in place (entries above the diagonal of row r are left untouched)
#pragma unroll
    for (int j = 0; j < SF_LEAF; ++j)
        if (j <= r) prow[j] = a[j];
    if (rhs) {
        rhs[(int64_t)b * ldr + c + r] = rv;  // z of this block
        lt[SF_LEAF * SF_LEAF + r] = rv;
    }
    if (r == 0 && bad && info[b] == 0) info[b] = bad;
}

// Rows below a factored 64 x 64 block:  X L^T = A  solved in place.  One wave handles 64 rows: the
// 64 x 64 slab is fetched with coalesced 16-byte loads (two full rows per instruction), transposed
// through LDS so that lane r owns row r in registers, eliminated with one v_fma_f64 per element whose
// L^T operand comes from the read-only side buffer through the scalar cache, and written back the
// same way.  With a right-hand side, the row's entry is updated right-looking: R[row] -= x . z_block.
__global__ __launch_bounds__(64) void k_trsm_leaf(double* __restrict__ base, int lda, int64_t stride,
                                                  int c, int n, const double* __restrict__ ltbuf,
                                                  double* __restrict__ rhs, int ldr) {
    __shared__ double tile[SF_LEAF * (SF_LEAF + 1)];
    const int b = blockIdx.y, lane = threadIdx.x;
    const double* __restrict__ Lt = ltbuf + (int64_t)b * SF_LTB;
    const int row0 = c + SF_LEAF + blockIdx.x * SF_LEAF;
    const int nvalid = min(SF_LEAF, n - row0);
    double* slab = base + (int64_t)b * stride + (int64_t)row0 * lda + c;
    const int half = lane >> 5, col2 = (lane & 31) * 2;
#pragma unroll 8
    for (int i = 0; i < SF_LEAF / 2; ++i) {
        const int rr = 2 * i + half;
        if (rr < nvalid) {
            const double2 v = *(const double2*)(slab + (int64_t)rr * lda + col2);
            tile[rr * 65 + col2] = v.x;
            tile[rr * 65 + col2 + 1] = v.y;
        }
    }
    __syncthreads();
    double x[SF_LEAF];
#pragma unroll
    for (int j = 0; j < SF_LEAF; ++j) x[j] = tile[lane * 65 + j];
#pragma unroll
    for (int k = 0; k < SF_LEAF; ++k) {
        x[k] *= Lt[k * SF_LEAF + k];
        const double xk = x[k];
#pragma unroll
        for (int j = k + 1; j < SF_LEAF; ++j) x[j] -= xk * Lt[k * SF_LEAF + j];
    }
#pragma unroll
    for (int j = 0; j < SF_LEAF; ++j) tile[lane * 65 + j] = x[j];
    if (rhs && lane < nvalid) {
        const double* __restrict__ z = Lt + SF_LEAF * SF_LEAF;
        double acc = rhs[(int64_t)b * ldr + row0 + lane];
#pragma unroll
        for (int j = 0; j < SF_LEAF; ++j) acc -= x[j] * z[j];
        rhs[(int64_t)b * ldr + row0 + lane] = acc;
    }
    __syncthreads();
#pragma unroll 8
    for (int i = 0; i < SF_LEAF / 2; ++i) {
        const int rr = 2 * i + half;
        if (rr < nvalid)
            *(double2*)(slab + (int64_t)rr * lda + col2) = make_double2(tile[rr * 65 + col2], tile[rr * 65 + col2 + 1]);
    }
}

__device__ __forceinline__ double sf_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// One workgroup per matrix: forward substitution L z = R by 64-row blocks (left-looking: the
// row block is read once, coalesced), then logdet = 2 sum log L_ii and sqmah = z.z.
template <bool ZGLOBAL>
__global__ __launch_bounds__(256) void k_trsv_logdet(const double* __restrict__ base, int n, int lda,
                                                     int64_t stride, const double* __restrict__ R,
                                                     int ldr, double* __restrict__ zscratch,
                                                     double* __restrict__ logdet,
                                                     double* __restrict__ sqmah) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    double* Ts = sm;                     // 64 x 65 diagonal block
    double* tv = Ts + SF_LEAF * 65;      // 64 right-hand sides of the block
    double* red = tv + SF_LEAF;          // 8 reduction slots
    double* z = ZGLOBAL ? zscratch + (int64_t)blockIdx.x * n : red + 8;

    const int b = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const double* Mx = base + (int64_t)b * stride;
    const double* Rb = R + (int64_t)b * ldr;

    for (int c = 0; c < n; c += SF_LEAF) {
        for (int e = tid; e < SF_LEAF * SF_LEAF; e += 256) {
            const int i = e >> 6, j = e & 63;
            Ts[i * 65 + j] = Mx[(int64_t)(c + i) * lda + c + j];
        }
        // 16 rows per wave, all 16 row streams in flight together
        double s[16];
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) s[rr] = 0.0;
        const double* prow = Mx + (int64_t)(c + w * 16) * lda;
        for (int k = lane; k < c; k += 64) {
            const double zk = z[k];
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) s[rr] += prow[(int64_t)rr * lda + k] * zk;
        }
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {
            const double tot = sf_wave_sum(s[rr]);
            if (lane == 0) tv[w * 16 + rr] = Rb[c + w * 16 + rr] - tot;
        }
        __syncthreads();
        if (w == 0) {
            double tval = tv[lane];
#pragma unroll 8
            for (int k = 0; k < SF_LEAF; ++k) {
                const double zk = __shfl(tval, k) / Ts[k * 65 + k];
                if (lane > k)
                    tval -= Ts[lane * 65 + k] * zk;
                else if (lane == k)
                    tval = zk;
            }
            z[c + lane] = tval;
        }
        __syncthreads();
    }
    double slog = 0.0, ssq = 0.0;
    for (int i = tid; i < n; i += 256) {
        slog += log(Mx[(int64_t)i * lda + i]);
        const double zi = z[i];
        ssq += zi * zi;
    }
    slog = sf_wave_sum(slog);
    ssq = sf_wave_sum(ssq);
    if (lane == 0) {
        red[w] = slog;
        red[4 + w] = ssq;
    }
    __syncthreads();
    if (tid == 0) {
        logdet[b] = 2.0 * (red[0] + red[1] + red[2] + red[3]);
        sqmah[b] = red[4] + red[5] + red[6] + red[7];
    }
}

// ------------------------------------------------------------------------------------ launchers
static int launch_gemm(double* A, int lda, int64_t stride, int batch, int r0, int c0, int k0, int M,
                       int Nc, int K, int tri, hipStream_t s) {
    if (M <= 0 || Nc <= 0 || K <= 0) return SF_OK;
    const int mt = (M + GT - 1) / GT, nt = (Nc + GT - 1) / GT;
    const long long nblk = (long long)mt * nt * batch;
    if (nblk > 0x7fffffffLL) {
        sf_set_error("gemm grid too large");
        return SF_EINVAL;
    }
    // algorithmic flops: only entries on/below the diagonal of a diagonal-aligned block count
    const double useful = tri ? ((double)M * Nc - 0.5 * (double)Nc * (Nc - 1)) : (double)M * Nc;
    void* tok;
    sf_prof_gemm_begin(s, 2.0 * K * useful * batch, &tok);
    hipLaunchKernelGGL(k_gemm_nt, dim3((unsigned)nblk), dim3(256), 0, s, A, lda, stride, r0, c0, k0, M,
                       Nc, K, tri, mt, nt);
    sf_prof_gemm_end(tok);
    SF_LAUNCH_CHECK();
    return SF_OK;
}

// One workgroup per matrix: logdet = 2 sum log L_ii and sqmah = |z|^2 where z = L^-1 R was produced
// in place of R by the factorisation (k_potrf_leaf / k_trsm_leaf with a right-hand side).
__global__ __launch_bounds__(256) void k_logdet_z(const double* __restrict__ base, int n, int lda,
                                                  int64_t stride, const double* __restrict__ zbuf, int ldr,
                                                  double* __restrict__ logdet,
                                                  double* __restrict__ sqmah) {
    __shared__ double red[8];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const double* Mx = base + (int64_t)b * stride;
    const double* z = zbuf + (int64_t)b * ldr;
    double slog = 0.0, ssq = 0.0;
    for (int i = tid; i < n; i += 256) {
        slog += log(Mx[(int64_t)i * lda + i]);
        const double zi = z[i];
        ssq += zi * zi;
    }
    slog = sf_wave_sum(slog);
    ssq = sf_wave_sum(ssq);
    if (lane == 0) {
        red[w] = slog;
        red[4 + w] = ssq;
    }
    __syncthreads();
    if (tid == 0) {
        logdet[b] = 2.0 * (red[0] + red[1] + red[2] + red[3]);
        sqmah[b] = red[4] + red[5] + red[6] + red[7];
    }
}

int sf_launch_logdet_z(const double* L, int n, int lda, int64_t stride, int batch, const double* z, int ldr,
                       double* logdet, double* sqmah, hipStream_t s) {
    hipLaunchKernelGGL(k_logdet_z, dim3(batch), dim3(256), 0, s, L, n, lda, stride, z, ldr, logdet, sqmah);
    SF_LAUNCH_CHECK();
    return SF_OK;
}

// Factor each n x n matrix in place; with rhs != NULL (batch x ldr) the forward substitution
// L z = rhs is fused into the leaf kernels and z overwrites rhs.
int sf_launch_potrf(double* A, int n, int lda, int64_t stride, int batch, int* info, double* ltbuf,
                    double* rhs, int ldr, hipStream_t s) {
    const int nrows = n;
    if (n % SF_LEAF != 0 || lda < n || batch <= 0 || (lda & 1) || !ltbuf) {
        sf_set_error("potrf: n must be a positive multiple of %d, lda >= n and even, workspace required", SF_LEAF);
        return SF_EINVAL;
    }
    SF_HIP(hipMemsetAsync(info, 0, sizeof(int) * (size_t)batch, s));
    for (int k0 = 0; k0 < n; k0 += SF_NB) {
        const int k1 = (k0 + SF_NB < n) ? k0 + SF_NB : n;
        if (k0 > 0) {
            int rc = launch_gemm(A, lda, stride, batch, k0, k0, 0, nrows - k0, k1 - k0, k0, 1, s);
            if (rc) return rc;
        }
        for (int c = k0; c < k1; c += SF_LEAF) {
            hipLaunchKernelGGL(k_potrf_leaf, dim3(batch), dim3(64), 0, s, A, lda, stride, c, info, ltbuf, rhs, ldr);
            SF_LAUNCH_CHECK();
            const int below = nrows - (c + SF_LEAF);
            if (below > 0) {
                hipLaunchKernelGGL(k_trsm_leaf, dim3((below + 63) / 64, batch), dim3(64), 0, s, A,
                                   lda, stride, c, nrows, (const double*)ltbuf, rhs, ldr);
                SF_LAUNCH_CHECK();
            }
            if (c + SF_LEAF < k1) {
                int rc = launch_gemm(A, lda, stride, batch, c + SF_LEAF, c + SF_LEAF, c, below,
                                     k1 - (c + SF_LEAF), SF_LEAF, 1, s);
                if (rc) return rc;
            }
        }
    }
    return SF_OK;
}

int sf_launch_logdet_sqmah(const double* L, int n, int lda, int64_t stride, int batch, const double* R,
                           int ldr, double* zscratch, double* logdet, double* sqmah, hipStream_t s) {
    if (n % SF_LEAF != 0 || batch <= 0) {
        sf_set_error("logdet_sqmah: n must be a multiple of %d", SF_LEAF);
        return SF_EINVAL;
    }
    const size_t fixed = sizeof(double) * (SF_LEAF * 65 + SF_LEAF + 8);
    const size_t with_z = fixed + sizeof(double) * (size_t)n;
    if (with_z <= 160 * 1024) {
        static bool attr_set = false;
        if (!attr_set) {
            SF_HIP(hipFuncSetAttribute((const void*)k_trsv_logdet<false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set = true;
        }
        hipLaunchKernelGGL(k_trsv_logdet<false>, dim3(batch), dim3(256), with_z, s, L, n, lda, stride, R,
                           ldr, (double*)nullptr, logdet, sqmah);
    } else {
        if (!zscratch) {
            sf_set_error("logdet_sqmah: n=%d needs a z scratch buffer", n);
            return SF_ENOMEM;
        }
        hipLaunchKernelGGL(k_trsv_logdet<true>, dim3(batch), dim3(256), fixed, s, L, n, lda, stride, R, ldr,
                           zscratch, logdet, sqmah);
    }
    SF_LAUNCH_CHECK();
    return SF_OK;
}
