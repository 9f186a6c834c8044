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
#ifndef GLD
#define GLD 17  // LDS row stride (doubles), odd: the 16 rows of a fragment hit 16 distinct bank pairs for
                // ds_read_b64 (64 banks) and ds_read2_b64 (32 banks) alike
#endif

// Logical block id such that ids adjacent in work space run on the same XCD (block b is observed on
// XCD b % 8; each XCD has its own L2).  Bijective for any grid size; placement only affects speed.
__device__ __forceinline__ int sf_xcd_remap(int bid, int nblk) {
#ifdef SF_NO_XCD_REMAP
    return bid;
#endif
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

// Batched MFMA update  Cout = Cin -/+ A * B^T  on 128 x 128 tiles (v_mfma_f64_16x16x4_f64, 4 waves,
// each 64 x 64 = 4 x 4 MFMA tiles; K staged through LDS in slabs of GK with register prefetch).
// All operands are row-major blocks addressed from their own origin (the host passes pointers already
// offset to the block): A is M x K, B is Nc x K, C is M x Nc.
struct sf_gemm_args {
    const double* A;
    const double* B;
    const double* Cin;  // NULL: start from zero
    double* Cout;
    int64_t sA, sB, sCin, sCout;  // batch strides (doubles)
    int lda, ldb, ldcin, ldcout;
    int M, Nc, K;
    int tri;    // block is diagonal-aligned: skip tiles lying entirely above the diagonal
    int btri;   // B[c][k] == 0 for k > c: column tile tn only needs k < (tn + 1) * GT
    int remap_after, remap_shift;  // output row i >= remap_after is stored at row i + remap_shift
    // fused left-looking right-hand-side update, done by the tiles with tm == tn while B streams by:
    //   rhs[c] -= sum_k B[c][k] * z[k]
    double* rhs;
    const double* z;
    int64_t srhs, sz;
    int mt, nt;
};

// Occupancy note (measured on MI355X, tools/probes/mfma_clock.hip): ONE wave issues a
// v_mfma_f64_16x16x4_f64 only every ~140 cycles even with independent accumulators, two waves per
// SIMD reach one per ~100 cycles, four waves per SIMD saturate the 64-cycle pipe.  The kernel is
// therefore built for 4 waves/SIMD: 512 threads (8 waves, each 32 x 64 of the 128 x 128 tile = 2 x 4
// MFMA tiles = 64 accumulator VGPRs), <= 128 VGPRs, two workgroups per CU.
template <bool NEG, bool RHS>
__global__ __launch_bounds__(512, 4) void k_gemm_nt(sf_gemm_args g) {
    __shared__ __attribute__((aligned(16))) double As[2][GT * GLD];
    __shared__ __attribute__((aligned(16))) double Bs[2][GT * GLD];

    const int id = sf_xcd_remap(blockIdx.x, gridDim.x);
    const int tiles = g.mt * g.nt;
    const int b = id / tiles;
    const int t = id - b * tiles;
    const int tm = t / g.nt, tn = t - tm * g.nt;
    const int row0 = tm * GT, col0 = tn * GT;
    if (g.tri && col0 > row0 + GT - 1) return;

    const int rows_here = min(GT, g.M - row0);
    const int cols_here = min(GT, g.Nc - col0);
    const int Kt = g.btri ? min(g.K, col0 + GT) : g.K;

    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int wm = w >> 1, wn = w & 1;  // 4 x 2 waves: rows wm*32.., cols wn*64..
    const int l15 = lane & 15, lq = lane >> 4;

    // ---- accumulators start as the C tile
    sf_d4 acc[2][4];
    const double* Cin = g.Cin ? g.Cin + (int64_t)b * g.sCin + (int64_t)row0 * g.ldcin + col0 : nullptr;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int col = wn * 64 + ni * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * 32 + mi * 16 + lq + 4 * r;
                double v = 0.0;
                if (Cin && row < rows_here && col < cols_here) v = Cin[(int64_t)row * g.ldcin + col];
                acc[mi][ni][r] = v;
            }
        }

    // ---- global -> register -> LDS staging: thread covers rows lr+64p, two doubles at column lc
    const int lr = tid >> 3, lc = (tid & 7) * 2;
    const double* Ag = g.A + (int64_t)b * g.sA + (int64_t)(row0 + lr) * g.lda + lc;
    const double* Bg = g.B + (int64_t)b * g.sB + (int64_t)(col0 + lr) * g.ldb + lc;
    double2 ra[2], rb[2];
    const bool do_rhs = RHS && g.rhs && (tm == tn);
    const double* zg = do_rhs ? g.z + (int64_t)b * g.sz + lc : nullptr;
    double2 zv = make_double2(0.0, 0.0);
    double part[2] = {0.0, 0.0};

    auto gload = [&](int kt) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const int rr = lr + 64 * p;
            ra[p] = (rr < rows_here) ? *(const double2*)(Ag + (int64_t)(64 * p) * g.lda + kt * GK)
                                     : make_double2(0.0, 0.0);
            rb[p] = (rr < cols_here) ? *(const double2*)(Bg + (int64_t)(64 * p) * g.ldb + kt * GK)
                                     : make_double2(0.0, 0.0);
        }
        if (RHS && do_rhs) zv = *(const double2*)(zg + kt * GK);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            double* pa = &As[buf][(lr + 64 * p) * GLD + lc];
            double* pb = &Bs[buf][(lr + 64 * p) * GLD + lc];
            pa[0] = ra[p].x;
            pa[1] = ra[p].y;
            pb[0] = rb[p].x;
            pb[1] = rb[p].y;
        }
        if (RHS && do_rhs) {
#pragma unroll
            for (int p = 0; p < 2; ++p) part[p] += rb[p].x * zv.x + rb[p].y * zv.y;
        }
    };

    const int nk = Kt / GK;
    if (nk > 0) {
        gload(0);
        lstore(0);
    }
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) gload(kt + 1);
        const double* Ab = &As[cur][(wm * 32 + l15) * GLD + lq];
        const double* Bb = &Bs[cur][(wn * 64 + l15) * GLD + lq];
#pragma unroll
        for (int ks = 0; ks < GK / 4; ++ks) {
            double a[2], bb[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[i] = NEG ? -Ab[i * 16 * GLD + ks * 4] : Ab[i * 16 * GLD + ks * 4];
#pragma unroll
            for (int i = 0; i < 4; ++i) bb[i] = Bb[i * 16 * GLD + ks * 4];
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], bb[ni], acc[mi][ni], 0, 0, 0);
        }
        if (kt + 1 < nk) lstore(cur ^ 1);
        __syncthreads();
    }

    double* Cout = g.Cout + (int64_t)b * g.sCout + col0;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
            const int col = wn * 64 + ni * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * 32 + mi * 16 + lq + 4 * r;
                if (row < rows_here && col < cols_here) {
                    int orow = row0 + row;
                    if (orow >= g.remap_after) orow += g.remap_shift;
                    Cout[(int64_t)orow * g.ldcout + col] = acc[mi][ni][r];
                }
            }
        }

    if (RHS && do_rhs) {
        // the 8 threads sharing lr cover the 16 k-columns of a slab: fold them, one of them commits
        double* rhs = g.rhs + (int64_t)b * g.srhs + col0;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            double v = part[p];
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            const int rr = lr + 64 * p;
            if ((tid & 7) == 0 && rr < cols_here) rhs[rr] -= v;
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
                                                   int c, int* __restrict__ info, int info_off,
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
        if (!(akk > 0.0) && !bad) bad = info_off + c + k + 1;
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
                                                  double* __restrict__ rhs, int ldr, int rhs_rows) {
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
    if (rhs && lane < nvalid && row0 + lane < rhs_rows) {
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
static int launch_gemm(sf_gemm_args g, int batch, bool neg, double flops, hipStream_t s) {
    if (g.M <= 0 || g.Nc <= 0) return SF_OK;
    g.mt = (g.M + GT - 1) / GT;
    g.nt = (g.Nc + GT - 1) / GT;
    const long long nblk = (long long)g.mt * g.nt * batch;
    if (nblk > 0x7fffffffLL) {
        sf_set_error("gemm grid too large");
        return SF_EINVAL;
    }
    void* tok;
    sf_prof_gemm_begin(s, flops, &tok);
    if (g.rhs)
        hipLaunchKernelGGL((k_gemm_nt<true, true>), dim3((unsigned)nblk), dim3(512), 0, s, g);
    else if (neg)
        hipLaunchKernelGGL((k_gemm_nt<true, false>), dim3((unsigned)nblk), dim3(512), 0, s, g);
    else
        hipLaunchKernelGGL((k_gemm_nt<false, false>), dim3((unsigned)nblk), dim3(512), 0, s, g);
    sf_prof_gemm_end(tok);
    SF_LAUNCH_CHECK();
    return SF_OK;
}

// One workgroup per matrix: logdet = 2 sum log L_ii and sqmah = |z|^2 where z = L^-1 R was produced
// in place of R by the factorisation.
__global__ __launch_bounds__(256) void k_logdet_z(const double* __restrict__ base, int n, int lda,
                                                  int64_t stride, const double* __restrict__ zbuf, int ldr,
                                                  double* __restrict__ logdet,
                                                  double* __restrict__ sqmah) {
    __shared__ double red[8];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const double* Mx = base + (int64_t)b * stride;
    const double* z = zbuf ? zbuf + (int64_t)b * ldr : nullptr;
    double slog = 0.0, ssq = 0.0;
    for (int i = tid; i < n; i += 256) {
        slog += log(Mx[(int64_t)i * lda + i]);
        const double zi = z ? z[i] : 0.0;
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

// Panel scratch T (per matrix, row stride SF_LDT): rows [0, pw) the updated diagonal block,
// rows [pw, 2pw) an identity block that turns into W = L_kk^-T while the diagonal block is factored,
// rows [2pw, ...) the updated rows below the diagonal block.
__global__ __launch_bounds__(256) void k_set_identity(double* __restrict__ T, int64_t sT, int pw) {
    double* I = T + (int64_t)blockIdx.y * sT + (int64_t)pw * SF_LDT;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < pw * pw; e += gridDim.x * 256) {
        const int i = e / pw, j = e - i * pw;
        I[(int64_t)i * SF_LDT + j] = (i == j) ? 1.0 : 0.0;
    }
}

// After the diagonal block is factored inside T: copy L_kk back into the matrix and store
// Wt[c][j] = W[j][c] = (L_kk^-1)[c][j] (lower triangular, zero above) for the panel solve GEMM.
__global__ __launch_bounds__(256) void k_panel_finish(const double* __restrict__ T, int64_t sT, int pw,
                                                      double* __restrict__ Cdiag, int ldc, int64_t sC,
                                                      double* __restrict__ Wt, int64_t sW) {
    __shared__ double tile[32][33];
    const int b = blockIdx.z;
    const double* Tb = T + (int64_t)b * sT;
    double* Cb = Cdiag + (int64_t)b * sC;
    double* Wb = Wt + (int64_t)b * sW;
    const int bi = blockIdx.y * 32, bj = blockIdx.x * 32;  // 32 x 32 block (rows bi.., cols bj..)
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8 threads
    for (int r = ty; r < 32; r += 8) {
        const int i = bi + r, j = bj + tx;
        if (i < pw && j < pw) {
            if (j <= i) Cb[(int64_t)i * ldc + j] = Tb[(int64_t)i * SF_LDT + j];
            tile[r][tx] = Tb[(int64_t)(pw + i) * SF_LDT + j];  // W[i][j]
        }
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = bj + r, j = bi + tx;  // Wt[c][j] = W[j][c]
        if (c < pw && j < pw) Wb[(int64_t)c * SF_LDT + j] = (j <= c) ? tile[tx][r] : 0.0;
    }
}

size_t sf_potrf_work_doubles(int n, int batch) {
    const size_t b = (size_t)batch;
    return b * SF_LTB_DOUBLES + b * (size_t)(n + SF_NB) * SF_LDT + b * (size_t)SF_NB * SF_LDT + 64;
}

// Factor each n x n matrix in place (lower).  Left-looking over panels of SF_NB columns:
//   U  T <- C[k0:, k0:k1] - L[k0:, :k0] L[k0:k1, :k0]^T        one MFMA launch, C read once, long K
//   D  factor T's diagonal block together with an identity block -> L_kk and W = L_kk^-T
//      (64-column leaf steps on the small (2 pw) x pw problem)
//   G  C[k1:, k0:k1] <- T[below] W                               one MFMA launch (triangular B)
// With rhs != NULL (batch x ldr) the forward substitution L z = rhs is fused: U applies the
// left-looking update of rhs[k0:k1] while B streams through LDS, D finishes it; z overwrites rhs.
int sf_launch_potrf(double* A, int n, int lda, int64_t stride, int batch, int* info, double* work,
                    double* rhs, int ldr, hipStream_t s) {
    if (n % SF_LEAF != 0 || lda < n || batch <= 0 || (lda & 1) || !work) {
        sf_set_error("potrf: n must be a positive multiple of %d, lda >= n and even, workspace required", SF_LEAF);
        return SF_EINVAL;
    }
    double* ltbuf = work;
    double* T = ltbuf + (size_t)batch * SF_LTB_DOUBLES;
    const int64_t sT = (int64_t)(n + SF_NB) * SF_LDT;
    double* Wt = T + (size_t)batch * sT;
    const int64_t sW = (int64_t)SF_NB * SF_LDT;
    SF_HIP(hipMemsetAsync(info, 0, sizeof(int) * (size_t)batch, s));
    for (int k0 = 0; k0 < n; k0 += SF_NB) {
        const int k1 = (k0 + SF_NB < n) ? k0 + SF_NB : n;
        const int pw = k1 - k0;
        // ---- U
        {
            sf_gemm_args g = {};
            g.A = A + (int64_t)k0 * lda;
            g.B = A + (int64_t)k0 * lda;
            g.Cin = A + (int64_t)k0 * lda + k0;
            g.Cout = T;
            g.sA = g.sB = g.sCin = stride;
            g.sCout = sT;
            g.lda = g.ldb = g.ldcin = lda;
            g.ldcout = SF_LDT;
            g.M = n - k0;
            g.Nc = pw;
            g.K = k0;
            g.tri = 1;
            g.remap_after = pw;
            g.remap_shift = pw;
            if (rhs && k0 > 0) {
                g.rhs = rhs + k0;
                g.z = rhs;
                g.srhs = g.sz = ldr;
            }
            const double useful = (double)g.M * pw - 0.5 * (double)pw * (pw - 1);
            int rc = launch_gemm(g, batch, true, 2.0 * k0 * useful * batch, s);
            if (rc) return rc;
        }
        // ---- D
        hipLaunchKernelGGL(k_set_identity, dim3(16, batch), dim3(256), 0, s, T, sT, pw);
        SF_LAUNCH_CHECK();
        for (int c = 0; c < pw; c += SF_LEAF) {
            hipLaunchKernelGGL(k_potrf_leaf, dim3(batch), dim3(64), 0, s, T, SF_LDT, sT, c, info, k0, ltbuf,
                               rhs ? rhs + k0 : nullptr, ldr);
            SF_LAUNCH_CHECK();
            const int below = 2 * pw - (c + SF_LEAF);
            hipLaunchKernelGGL(k_trsm_leaf, dim3((below + 63) / 64, batch), dim3(64), 0, s, T, SF_LDT, sT, c,
                               2 * pw, (const double*)ltbuf, rhs ? rhs + k0 : nullptr, ldr, pw);
            SF_LAUNCH_CHECK();
            if (c + SF_LEAF < pw) {
                sf_gemm_args g = {};
                const int o = c + SF_LEAF;
                g.A = T + (int64_t)o * SF_LDT + c;
                g.B = g.A;
                g.Cin = T + (int64_t)o * SF_LDT + o;
                g.Cout = T + (int64_t)o * SF_LDT + o;
                g.sA = g.sB = g.sCin = g.sCout = sT;
                g.lda = g.ldb = g.ldcin = g.ldcout = SF_LDT;
                g.M = below;
                g.Nc = pw - o;
                g.K = SF_LEAF;
                g.tri = 1;
                g.remap_after = 0x7fffffff;
                const double useful = (double)g.M * g.Nc - 0.5 * (double)g.Nc * (g.Nc - 1);
                int rc = launch_gemm(g, batch, true, 2.0 * SF_LEAF * useful * batch, s);
                if (rc) return rc;
            }
        }
        hipLaunchKernelGGL(k_panel_finish, dim3((pw + 31) / 32, (pw + 31) / 32, batch), dim3(256), 0, s,
                           (const double*)T, sT, pw, A + (int64_t)k0 * lda + k0, lda, stride, Wt, sW);
        SF_LAUNCH_CHECK();
        // ---- G
        if (n > k1) {
            sf_gemm_args g = {};
            g.A = T + (int64_t)(2 * pw) * SF_LDT;
            g.B = Wt;
            g.Cin = nullptr;
            g.Cout = A + (int64_t)k1 * lda + k0;
            g.sA = sT;
            g.sB = sW;
            g.sCout = stride;
            g.lda = g.ldb = SF_LDT;
            g.ldcout = lda;
            g.M = n - k1;
            g.Nc = pw;
            g.K = pw;
            g.btri = 1;
            g.remap_after = 0x7fffffff;
            int rc = launch_gemm(g, batch, false, (double)g.M * pw * pw * batch, s);
            if (rc) return rc;
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
