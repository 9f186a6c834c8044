// Batched fp64 Cholesky (lower, row-major, in place) + logdet / Mahalanobis solve for gfx950.
//
// Replaces the LAPACK calls of the reference: scipy.linalg.cho_factor / cho_solve at
// Starfish/models/spectrum_model.py:400-404 (dpotrf + dpotrs on the N x N covariance).
//
// Structure (one launch sequence serves the whole batch; the batch supplies the parallelism):
//   default (>= 28 matrices): LEFT-looking panels of 128 columns with the FUSED panel kernel k_chol_panel -- one
//     workgroup per 128-row slab does the long-K update (v_mfma_f64_16x16x4_f64, > 80 % of the flops), the
//     triangular solve against the explicit inverse of the diagonal tile, the in-place store of L, the forward
//     substitution of the right-hand side and the rank-128 update of its own diagonal tile; k_diag_mfma factors
//     the 128 x 128 diagonal tile (L_kk, L_kk^-1, z_k) on a side stream one panel ahead (lookahead); launches
//     that cannot fill the chip are split along K (partial sums + deterministic reduce).  sf_launch_potrf_v2.
//   small batches: the unfused sequence of round 1 (panels of SF_NB = 256 columns; k_gemm_nt long-K update into a
//     panel scratch, k_diag_mfma on 256 x 256 blocks, separate panel-solve and diagonal-update launches): half as
//     many sequential long-K steps.  sf_launch_potrf_v1.
//   k_logdet_z: logdet = 2 sum log L_ii and sqmah = |z|^2 (z = L^-1 R is produced inside the factorisation);
//   k_trsv_logdet: the stand-alone forward substitution of sf_logdet_sqmah_batch.
#include <atomic>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "sf_common.h"
#include <stdio.h>

#ifdef SF_TUNING
#define SF_PANEL_SKIPS(g, bit) ((g).skip & (bit))  // phase switched off (wrong results, timing only)
#else
#define SF_PANEL_SKIPS(g, bit) (false)
#endif
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
    // block-diagonal mode (diag_blocks > 0): the launch updates diag_blocks independent SF_NB x SF_NB
    // diagonal blocks; block j takes A/B at +j*dA and C at +j*dC (M = Nc = total rows covered)
    int diag_blocks;
    int64_t dA, dC;
    // matrix-free start: if tilemap says this 128 x 128 tile was never materialised, its initial value is
    // Y^T Y (rank-mpad product of the rows/columns of Y) instead of Cin
    const double* genY;
    const unsigned char* tilemap;
    int64_t sY;
    int ldy, mpad, nt128, tm_off, tn_off;
    int mt, nt;
    int no_syrk;  // tuning aid: diagonal tiles through the generic path
};

// Diagonal 128 x 128 tile of a symmetric update C -= P P^T (block-diagonal launches): only the 36 MFMA
// blocks on or below the diagonal are computed, dealt to the 8 waves in equal shares (rows p and 7-p of
// the 8 x 8 block grid hold 9 blocks; one wave takes 5 of them, its partner 4 plus a spare), and the single
// operand P is staged once instead of twice.  40 block products per slab instead of 64.
// Blocks above the diagonal are neither read nor written (nothing references them).
template <bool RHS, int NTH>
__device__ __forceinline__ void sf_syrk_diag_tile(const sf_gemm_args& g, int b, int row0, double (*As)[GT * GLD]) {
    static_assert(NTH == 512, "8 waves");
    constexpr int NP = 1024 / NTH, RPP = NTH / 8;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int p = w >> 1, h = w & 1;
    int bi[5], bj[5];
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        if (h == 0) {
            bi[q] = 7 - p;
            bj[q] = q;
        } else {
            const int n_hi = 3 - p;  // blocks 5 .. 7-p of row 7-p, then blocks 0 .. p of row p
            const int qq = q < 4 ? q : 0;
            bi[q] = qq < n_hi ? 7 - p : p;
            bj[q] = qq < n_hi ? 5 + qq : qq - n_hi;
        }
    }
    const int nstore = h == 0 ? 5 : 4;

    const int lr = tid >> 3, lc = (tid & 7) * 2;
    const double* Ap[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) Ap[q] = g.A + (int64_t)b * g.sA + (int64_t)(row0 + lr + RPP * q) * g.lda + lc;
    double2 ra[NP];
    const bool do_rhs = RHS && g.rhs;
    const double* zg = do_rhs ? g.z + (int64_t)b * g.sz + lc : nullptr;
    double2 zv = make_double2(0.0, 0.0);
    double part[NP];
#pragma unroll
    for (int q = 0; q < NP; ++q) part[q] = 0.0;
    auto gload = [&](int kt) {
#pragma unroll
        for (int q = 0; q < NP; ++q) ra[q] = *(const double2*)(Ap[q] + kt * GK);
        if (RHS && do_rhs) zv = *(const double2*)(zg + kt * GK);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            double* pa = &As[buf][(lr + RPP * q) * GLD + lc];
            pa[0] = ra[q].x;
            pa[1] = ra[q].y;
        }
        if (RHS && do_rhs) {
#pragma unroll
            for (int q = 0; q < NP; ++q) part[q] += ra[q].x * zv.x + ra[q].y * zv.y;
        }
    };
    const int nk = g.K / GK;
    if (nk > 0) gload(0);
    sf_d4 acc[5];
    const double* Cin = g.Cin + (int64_t)b * g.sCin + (int64_t)row0 * g.ldcin + row0;
#pragma unroll
    for (int q = 0; q < 5; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            acc[q][r] = Cin[(int64_t)(16 * bi[q] + lq + 4 * r) * g.ldcin + 16 * bj[q] + l15];
    if (nk > 0) lstore(0);
    __syncthreads();
    auto compute = [&](int cur) {
        const double* S = &As[cur][l15 * GLD + lq];
#pragma unroll
        for (int ks = 0; ks < GK / 4; ++ks) {
#pragma unroll
            for (int q = 0; q < 5; ++q)
                acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(S[bi[q] * 16 * GLD + ks * 4], S[bj[q] * 16 * GLD + ks * 4],
                                                              acc[q], 0, 0, 1);  // blgp 1 = neg:[1,0,0]: -A B + C
        }
    };
    for (int kt = 0; kt + 1 < nk; ++kt) {
        gload(kt + 1);
        compute(kt & 1);
        lstore((kt & 1) ^ 1);
        __syncthreads();
    }
    if (nk > 0) compute((nk - 1) & 1);
    double* Cout = g.Cout + (int64_t)b * g.sCout + (int64_t)row0 * g.ldcout + row0;
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        if (q >= nstore) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            Cout[(int64_t)(16 * bi[q] + lq + 4 * r) * g.ldcout + 16 * bj[q] + l15] = acc[q][r];
    }
    if (RHS && do_rhs) {
        double* rhs = g.rhs + (int64_t)b * g.srhs + row0;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            double v = part[q];
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            if ((tid & 7) == 0) rhs[lr + RPP * q] -= v;
        }
    }
}

// Occupancy note (measured on MI355X, tools/probes/mfma_clock.hip): ONE wave issues a
// v_mfma_f64_16x16x4_f64 only every ~140 cycles even with independent accumulators, two waves per
// SIMD reach one per ~100 cycles, four waves per SIMD saturate the 64-cycle pipe.  The kernel is
// therefore built for 4 waves/SIMD: 512 threads (8 waves, each 32 x 64 of the 128 x 128 tile = 2 x 4
// MFMA tiles = 64 accumulator VGPRs), <= 128 VGPRs, two workgroups per CU.
// Wave layout: WN waves across the 128 columns, 128/(16*TM) ... the tile is always 128 x 128;
// NTH = 512 -> 8 waves of 32 x 64 (TM=2, TN=4);  NTH = 1024 -> 16 waves of 32 x 32 (TM=2, TN=2).
template <bool NEG, bool RHS, int NTH>
__global__ __launch_bounds__(NTH, (NTH == 256) ? 2 : NTH / 128) void k_gemm_nt(sf_gemm_args g) {
    constexpr int TM = (NTH == 256) ? 4 : 2;
    constexpr int TN = (NTH == 1024) ? 2 : 4;
    constexpr int WN = 128 / (16 * TN);   // waves across columns
    constexpr int NP = 1024 / NTH;        // staging passes of NTH/8 rows each
    constexpr int RPP = NTH / 8;          // rows per staging pass
    __shared__ __attribute__((aligned(16))) double As[2][GT * GLD];
    __shared__ __attribute__((aligned(16))) double Bs[2][GT * GLD];

    const int id = sf_xcd_remap(blockIdx.x, gridDim.x);
    const int tiles = g.mt * g.nt;
    const int b = id / tiles;
    const int t = id - b * tiles;
    int tm, tn, rows_here, cols_here;
    if (g.diag_blocks) {
        // 2 x 2 tiles per SF_NB block, the upper-right one is never needed
        const int jb = t >> 2;
        tm = (t >> 1) & 1;
        tn = t & 1;
        if (tn > tm) return;
        const int blk = min(SF_NB, g.M - jb * SF_NB);  // the last block may be narrower
        rows_here = min(GT, blk - tm * GT);
        cols_here = min(GT, blk - tn * GT);
        if (rows_here <= 0 || cols_here <= 0) return;
        g.A += jb * g.dA;
        g.B += jb * g.dA;
        if (g.Cin) g.Cin += jb * g.dC;
        g.Cout += jb * g.dC;
        if (RHS && g.rhs) g.rhs += jb * SF_NB;
        if (NTH == 512 && NEG && tm == tn && rows_here == GT && g.K > 0 && g.A == g.B && g.Cin && !g.no_syrk) {
            sf_syrk_diag_tile<RHS, 512>(g, b, tm * GT, As);
            return;
        }
    } else {
        tm = t / g.nt;
        tn = t - tm * g.nt;
        if (g.tri && tn * GT > tm * GT + GT - 1) return;
        rows_here = min(GT, g.M - tm * GT);
        cols_here = min(GT, g.Nc - tn * GT);
    }
    const int row0 = tm * GT, col0 = tn * GT;
    const int Kt = g.btri ? min(g.K, col0 + GT) : g.K;

    const int tid = threadIdx.x;
    const int lane = tid & 63, w = tid >> 6;
    const int wm = w / WN, wn = w % WN;  // rows wm*32.., cols wn*(16*TN)..
    const int l15 = lane & 15, lq = lane >> 4;

    // ---- global -> register -> LDS staging: thread covers rows lr+64p, two doubles at column lc.
    // Rows past the block edge are CLAMPED to the last valid row instead of being predicated: the
    // duplicated data only feeds accumulator rows / columns that are never stored, and the loads stay
    // branch-free (a predicated load makes hipcc wait for the whole vm queue).
    const int lr = tid >> 3, lc = (tid & 7) * 2;
    const double* Ap[NP];
    const double* Bp[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        Ap[p] = g.A + (int64_t)b * g.sA + (int64_t)(row0 + min(lr + RPP * p, rows_here - 1)) * g.lda + lc;
        Bp[p] = g.B + (int64_t)b * g.sB + (int64_t)(col0 + min(lr + RPP * p, cols_here - 1)) * g.ldb + lc;
    }
    double2 ra[NP], rb[NP];
    const bool do_rhs = RHS && g.rhs && (tm == tn);
    const double* zg = do_rhs ? g.z + (int64_t)b * g.sz + lc : nullptr;
    double2 zv = make_double2(0.0, 0.0);
    double part[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) part[p] = 0.0;

    auto gload = [&](int kt) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            ra[p] = *(const double2*)(Ap[p] + kt * GK);
            rb[p] = *(const double2*)(Bp[p] + kt * GK);
        }
        if (RHS && do_rhs) zv = *(const double2*)(zg + kt * GK);
    };
    auto lstore = [&](int buf) {
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            double* pa = &As[buf][(lr + RPP * p) * GLD + lc];
            double* pb = &Bs[buf][(lr + RPP * p) * GLD + lc];
            pa[0] = ra[p].x;
            pa[1] = ra[p].y;
            pb[0] = rb[p].x;
            pb[1] = rb[p].y;
        }
        if (RHS && do_rhs) {
#pragma unroll
            for (int p = 0; p < NP; ++p) part[p] += rb[p].x * zv.x + rb[p].y * zv.y;
        }
    };

    // the first operand slab is requested before the accumulators are initialised so that both
    // latencies overlap (matters for the short-K launches)
    const int nk = Kt / GK;
    if (nk > 0) gload(0);

    // ---- accumulators start as the C tile (read, or generated from Y when it was never materialised)
    sf_d4 acc[TM][TN];
    const double* Cin = g.Cin ? g.Cin + (int64_t)b * g.sCin + (int64_t)row0 * g.ldcin + col0 : nullptr;
    bool generate = false;
    if (g.tilemap && !g.diag_blocks)
        generate = !g.tilemap[(int64_t)b * g.nt128 * g.nt128 + (g.tm_off + tm) * g.nt128 + (g.tn_off + tn)];
    if (generate) {
        const double* Yb = g.genY + (int64_t)b * g.sY;
        // global pixel index of this lane's row / column (clamped: Y has ldy columns; rows past the
        // matrix edge are never stored)
        const int gr = (g.tm_off + tm) * GT + wm * (16 * TM) + l15;
        const int gc = (g.tn_off + tn) * GT + wn * (16 * TN) + l15;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = (sf_d4){0.0, 0.0, 0.0, 0.0};
        for (int kk = 0; kk < g.mpad; kk += 4) {
            const double* yk = Yb + (int64_t)(kk + lq) * g.ldy;
            double ya[TM], yb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) ya[i] = yk[min(gr + i * 16, g.ldy - 1)];
#pragma unroll
            for (int i = 0; i < TN; ++i) yb[i] = yk[min(gc + i * 16, g.ldy - 1)];
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[mi], yb[ni], acc[mi][ni], 0, 0, 0);
        }
    } else {
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int col = wn * (16 * TN) + ni * 16 + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wm * (16 * TM) + mi * 16 + lq + 4 * r;
                    double v = 0.0;
                    if (Cin && row < rows_here && col < cols_here) v = Cin[(int64_t)row * g.ldcin + col];
                    acc[mi][ni][r] = v;
                }
            }
    }

    if (nk > 0) lstore(0);
    __syncthreads();

    auto compute = [&](int cur) {
        const double* Ab = &As[cur][(wm * (16 * TM) + l15) * GLD + lq];
        const double* Bb = &Bs[cur][(wn * (16 * TN) + l15) * GLD + lq];
#pragma unroll
        for (int ks = 0; ks < GK / 4; ++ks) {
            double a[TM], bb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = Ab[i * 16 * GLD + ks * 4];
#pragma unroll
            for (int i = 0; i < TN; ++i) bb[i] = Bb[i * 16 * GLD + ks * 4];
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], bb[ni], acc[mi][ni], 0, 0, NEG ? 1 : 0);  // the f64 MFMA's blgp bits negate: neg:[1,0,0]
        }
    };
    // steady state is ONE basic block: issue the next slab's global loads, run this slab's MFMAs from
    // LDS, then park the loaded slab in the other LDS buffer; the last slab is peeled
    for (int kt = 0; kt + 1 < nk; ++kt) {
        gload(kt + 1);
        compute(kt & 1);
        lstore((kt & 1) ^ 1);
        __syncthreads();
    }
    if (nk > 0) compute((nk - 1) & 1);

    double* Cout = g.Cout + (int64_t)b * g.sCout + col0;
#pragma unroll
    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int col = wn * (16 * TN) + ni * 16 + l15;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm * (16 * TM) + mi * 16 + lq + 4 * r;
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
        for (int p = 0; p < NP; ++p) {
            double v = part[p];
            v += __shfl_xor(v, 1);
            v += __shfl_xor(v, 2);
            v += __shfl_xor(v, 4);
            const int rr = lr + RPP * p;
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
    if (g.diag_blocks) {  // 4 tile slots per block
        g.mt = g.diag_blocks;
        g.nt = 4;
    }
    const long long nblk = (long long)g.mt * g.nt * batch;
    if (nblk > 0x7fffffffLL) {
        sf_set_error("gemm grid too large");
        return SF_EINVAL;
    }
    static const bool no_syrk = SF_TUNE_FLAG("SF_NO_SYRK");
    g.no_syrk = no_syrk;
    void* tok;
    sf_prof_gemm_begin(s, flops, &tok);
    static const bool big = SF_TUNE_FLAG("SF_GEMM_1024");  // tuning aid: 16 waves of 32 x 32
    static const bool small = SF_TUNE_FLAG("SF_GEMM_256");  // tuning aid: 4 waves of 64 x 64
    if (small) {
        if (g.rhs)
            hipLaunchKernelGGL((k_gemm_nt<true, true, 256>), dim3((unsigned)nblk), dim3(256), 0, s, g);
        else if (neg)
            hipLaunchKernelGGL((k_gemm_nt<true, false, 256>), dim3((unsigned)nblk), dim3(256), 0, s, g);
        else
            hipLaunchKernelGGL((k_gemm_nt<false, false, 256>), dim3((unsigned)nblk), dim3(256), 0, s, g);
    } else if (big) {
        if (g.rhs)
            hipLaunchKernelGGL((k_gemm_nt<true, true, 1024>), dim3((unsigned)nblk), dim3(1024), 0, s, g);
        else if (neg)
            hipLaunchKernelGGL((k_gemm_nt<true, false, 1024>), dim3((unsigned)nblk), dim3(1024), 0, s, g);
        else
            hipLaunchKernelGGL((k_gemm_nt<false, false, 1024>), dim3((unsigned)nblk), dim3(1024), 0, s, g);
    } else {
        if (g.rhs)
            hipLaunchKernelGGL((k_gemm_nt<true, true, 512>), dim3((unsigned)nblk), dim3(512), 0, s, g);
        else if (neg)
            hipLaunchKernelGGL((k_gemm_nt<true, false, 512>), dim3((unsigned)nblk), dim3(512), 0, s, g);
        else
            hipLaunchKernelGGL((k_gemm_nt<false, false, 512>), dim3((unsigned)nblk), dim3(512), 0, s, g);
    }
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

// ---------------------------------------------------------------------------------------------
// Diagonal-block step of one panel as ONE launch on the matrix cores (one workgroup of 16 waves per
// matrix, one 16 x 16 block of the current block column per wave): L_kk and its inverse for the pw x pw block (pw <= 256), LEFT-looking over 16-column block
// columns so that nothing is read-modify-written in memory:
//   U  every wave accumulates its blocks of column k in registers:  M(i,k) - sum_{j<k} L(i,j) L(k,j)^T
//      for the rows of the matrix block and  - sum_{e<=j<k} X(e,j) L(k,j)^T  for the rows of the "identity
//      block" E (whose solved rows X = rows of W = L_kk^-T).  A operands stream from L2, the row L(k,:)
//      shared by the whole column is staged in LDS once;
//   P  wave 0, which owns M(k,k), factorises it and inverts the factor in the MFMA accumulator layout
//      (column j of the symmetric block is register j/4 of quarter j%4 = a K-slice of the MFMA operands,
//      so every rank-1 elimination is one MFMA without data movement; pivots from scalars so that the
//      rsqrt chain overlaps the matrix core);
//   X  every wave solves the blocks it still holds as a product with the 16 x 16 inverse F and writes
//      them out: L to the matrix (and in place, as operand of later columns), W transposed to Wt.
// The identity block is implicit (row block e of E starts at column e with X = F^T).  Finally
// z_k = L_kk^-1 r_k as a product with the explicit inverse.  Replaces k_set_identity + 4 x (k_potrf_leaf,
// k_trsm_leaf, K=64 k_gemm_nt) + k_panel_finish: one scheduling wait on the contended chip instead of 13.
#define DBS (16 * 17)
#define DLD 17
__device__ __forceinline__ double sfd_rsqrt(double p) {
    double y = __builtin_amdgcn_rsq(p);
    const double h = 0.5 * p;
    double e = __builtin_fma(-h * y, y, 0.5);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-h * y, y, 0.5);
    y = __builtin_fma(y, e, y);
    return y;
}

template <int NT>
__global__ __launch_bounds__(NT) void k_diag_mfma(double* __restrict__ T, int64_t sT, int pw,
                                                      int* __restrict__ info, int info_off,
                                                      double* __restrict__ rhs, int ldr,
                                                      double* __restrict__ Cdiag, int ldc, int64_t sC,
                                                      double* __restrict__ Wt, int64_t sW) {
    __shared__ double LK[(NT / 64 - 1) * DBS];  // L(k, j), j < k: the B operand of the whole block column
    __shared__ double ST[(NT / 64) * DBS];      // per-wave staging block (accumulator layout -> operand layout)
    __shared__ double Fb[DBS];       // inverse of the current 16 x 16 diagonal factor
    __shared__ double rz[256];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int nb = pw >> 4;
    double* Tb = T + (int64_t)b * sT;
    double* Eb = Tb + (int64_t)pw * SF_LDT;
    double* Cb = Cdiag + (int64_t)b * sC;
    double* Wb = Wt + (int64_t)b * sW;
    double* st = ST + wave * DBS;

    // Wt is lower triangular: zero the blocks above the diagonal (the buffer alternates between panels)
    for (int e = tid; e < nb * nb * 256; e += NT) {
        const int blk = e >> 8, bc = blk / nb, be = blk - bc * nb;
        if (be > bc) Wb[(int64_t)(bc * 16 + ((e >> 4) & 15)) * SF_LDT + be * 16 + (e & 15)] = 0.0;
    }
    int bad = 0;
    for (int k = 0; k < nb; ++k) {
        const int m = nb - 1 - k;  // matrix row blocks below the diagonal block
        // ---- stage L(k, 0..k-1) (final since the previous columns) in LDS
        for (int e = tid; e < k * 256; e += NT) {
            const int j = e >> 8, r = (e >> 4) & 15, cc = e & 15;
            LK[j * DBS + r * DLD + cc] = Tb[(int64_t)(16 * k + r) * SF_LDT + 16 * j + cc];
        }
        __syncthreads();
        // ---- U: block of this wave: t = 0 -> M(k,k), 1..m -> M(k+t,k), then E(e,k)
        sf_d4 acc[1];
        int kind[1];  // 0 none, 1 matrix row block, 2 inverse row block
        int ibk[1];
        {
            constexpr int u = 0;
            const int t = wave;
            kind[u] = 0;
            ibk[u] = 0;
            acc[u] = (sf_d4){0.0, 0.0, 0.0, 0.0};
            if (t <= m + k) {
                const bool isM = t <= m;
                const int ib = isM ? k + t : t - m - 1;
                kind[u] = isM ? 1 : 2;
                ibk[u] = ib;
                const double* rowp = (isM ? Tb : Eb) + (int64_t)(16 * ib) * SF_LDT;
                if (isM) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = lq + 4 * r;
                        // the diagonal block is read symmetrically from its lower triangle
                        acc[u][r] = (t == 0) ? rowp[(int64_t)max(row, l15) * SF_LDT + 16 * k + min(row, l15)]
                                             : rowp[(int64_t)row * SF_LDT + 16 * k + l15];
                    }
                }
                const int j0 = isM ? 0 : ib;  // X(e, j) exists for j >= e
                // K is a summation index: lane (l15, lq) takes the four CONTIGUOUS columns 4 lq .. 4 lq + 3
                // of its row (two 16-byte loads, full 128-B lines per 4 lanes) and MFMA kk uses element kk, i.e.
                // slice lq of instruction kk stands for k = 4 lq + kk -- in both operands.
                const double2* ap = (const double2*)(rowp + (int64_t)l15 * SF_LDT + 4 * lq);
                // A fragments stream from L2: four block columns in flight (clamped loads past the end)
                double2 av[4][2];
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int jj = min(j0 + d, max(k - 1, 0));
                    av[d][0] = ap[8 * jj];
                    av[d][1] = ap[8 * jj + 1];
                }
                for (int j = j0; j < k; ++j) {
                    // rotating register window: block column j is consumed, j + 4 is requested
                    const double* lk = LK + j * DBS + l15 * DLD + 4 * lq;
                    const double a4[4] = {av[0][0].x, av[0][0].y, av[0][1].x, av[0][1].y};
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
                        acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[kk], lk[kk], acc[u], 0, 0, 1);  // neg:[1,0,0]
#pragma unroll
                    for (int d = 0; d < 3; ++d) {
                        av[d][0] = av[d + 1][0];
                        av[d][1] = av[d + 1][1];
                    }
                    const int jj = min(j + 4, max(k - 1, 0));
                    av[3][0] = ap[8 * jj];
                    av[3][1] = ap[8 * jj + 1];
                }
            }
        }
        // ---- P: wave 0 holds the updated diagonal block in acc[0]
        if (wave == 0) {
            sf_d4 a0 = acc[0], f, lt;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f[r] = (lq + 4 * r) == l15 ? 1.0 : 0.0;
                lt[r] = 0.0;
            }
            double p = sf_readlane_d(a0[0], 0);
            double pkeep = 1.0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int qj = j & 3, rj = j >> 2;
                pkeep = lane == j ? p : pkeep;
                const double rs = sfd_rsqrt(p);
                const bool in_q = lq == qj;
                const double v = (in_q && l15 > j) ? a0[rj] * rs : 0.0;  // l_ij, i = l15 > j
                const double g = in_q ? f[rj] * rs : 0.0;                // row j of F, scaled
                if (in_q) {
                    f[rj] = g;
                    lt[rj] = l15 == j ? p * rs : v;  // L^T[j][i]
                }
                if (j + 1 < 16) {
                    const double an = sf_readlane_d(a0[(j + 1) >> 2], ((j + 1) & 3) * 16 + j + 1);
                    const double vn = sf_readlane_d(v, qj * 16 + j + 1);
                    p = __builtin_fma(-vn, vn, an);
                }
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, a0, 0, 0, 1);  // neg:[1,0,0]
                f = __builtin_amdgcn_mfma_f64_16x16x4f64(v, g, f, 0, 0, 1);
            }
            const unsigned long long neg = __ballot(lane < 16 && !(pkeep > 0.0));
            if (neg && !bad) bad = 16 * k + __ffsll((long long)neg);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = lq + 4 * r;
                Fb[row * DLD + l15] = f[r];
                Wb[(int64_t)(16 * k + row) * SF_LDT + 16 * k + l15] = f[r];  // diagonal block of L_kk^-1
                // X of the identity row block k is F^T: operand of later columns
                Eb[(int64_t)(16 * k + l15) * SF_LDT + 16 * k + row] = f[r];
                if (l15 >= row) {
                    Cb[(int64_t)(16 * k + l15) * ldc + 16 * k + row] = lt[r];          // L[i][j] -> matrix
                    Tb[(int64_t)(16 * k + l15) * SF_LDT + 16 * k + row] = lt[r];        // and in place
                }
            }
            kind[0] = 0;
        }
        __syncthreads();
        // ---- X: solve the blocks still held in registers, write them out
#pragma unroll
        for (int u = 0; u < 1; ++u) {
            if (kind[u] == 0) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) st[(lq + 4 * r) * DLD + l15] = acc[u][r];
            sf_d4 x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk)
                x = __builtin_amdgcn_mfma_f64_16x16x4f64(st[l15 * DLD + kk * 4 + lq], Fb[l15 * DLD + kk * 4 + lq], x,
                                                         0, 0, 0);
            const int ib = ibk[u];
            double* rowp = (kind[u] == 1 ? Tb : Eb) + (int64_t)(16 * ib) * SF_LDT + 16 * k;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = lq + 4 * r;
                rowp[(int64_t)row * SF_LDT + l15] = x[r];  // in place: operand of the later columns
                if (kind[u] == 1) Cb[(int64_t)(16 * ib + row) * ldc + 16 * k + l15] = x[r];  // L
                else Wb[(int64_t)(16 * k + l15) * SF_LDT + 16 * ib + row] = x[r];             // (L^-T)^T
            }
        }
        __syncthreads();
    }
    if (tid == 0 && bad && info && info[b] == 0) info[b] = info_off + bad;
    // ---- z_k = L_kk^-1 r_k with the explicit inverse
    if (rhs) {
        double* rb = rhs + (int64_t)b * ldr;
        for (int i = tid; i < pw; i += NT) rz[i] = rb[i];
        __syncthreads();
        for (int i = tid; i < pw; i += NT) {
            const double* wrow = Wb + (int64_t)i * SF_LDT;
            double zacc = 0.0;
            for (int j = 0; j <= i; ++j) zacc = __builtin_fma(wrow[j], rz[j], zacc);
            rb[i] = zacc;
        }
    }
}


// The same diagonal-block step for the 128-column panels of the fused sequence, with the tile and its growing inverse
// RESIDENT IN LDS: k_diag_mfma keeps them in the L2-backed scratch, so every one of its 8 block columns pays two global
// round trips (stage L(k,:), write X / read it back as an operand of the next column), 55-65 us per tile when the chip
// is idle and three times that beside the panel launches.  Here the tile is read once, the eight columns run out of LDS
// (no staging buffers: an accumulator block is written to its own destination block and read back in operand layout
// by the same wave), and L / L^-1 / z are written once.
// The inverse grows IN PLACE of the factor (36 blocks of 16 x 17 doubles + one: 79.6 KB, two workgroups per CU or one
// beside a panel workgroup; with a second set of blocks for the inverse it was 157 KB = a whole CU): block column k of
// W = L^-T is row k of L^-1, and row k of L is read for the last time at step k -- by the waves that update block column
// k and by the waves that accumulate row k of the inverse, all before that step's first barrier -- so after it the
// inverse waves drop X(e, k) into the slot of L(k, e).  L leaves for global memory block by block as it becomes final.
// (body shared by the kernel below and by the diagonal-tile tasks of k_potrf_dataflow; dsm: SF_DIAG_LDS_BYTES of LDS)
__device__ __forceinline__ void sf_diag_lds_body(const double* __restrict__ T, int64_t sT, int pw, int* __restrict__ info,
                                                 int info_off, double* __restrict__ rhs, int ldr,
                                                 double* __restrict__ Cdiag, int ldc, int64_t sC,
                                                 double* __restrict__ Wt, int64_t sW, int fp0, const int b,
                                                 double* __restrict__ dsm, const int tid, long long* stamps = nullptr) {
#ifdef SF_TUNING
#define SF_D_STAMP(i) do { if (stamps && tid == 0) stamps[i] = wall_clock64(); } while (0)
#else
#define SF_D_STAMP(i)
#endif
    SF_D_STAMP(0);
    // fp0: the first fp0 rows / columns of the tile are virtual (identity in T; Cdiag and rhs point fp0 elements BEFORE
    // the matrix there: never stored, read as zero) -- the first tile of a shifted frame, see sf_potrf_front_pad
    double* Tl = dsm;              // lower blocks (bi >= bj) of the tile at (bi (bi + 1) / 2 + bj) * DBS
    double* El = Tl;               // blocks X(e, j), e <= j, of W = L_kk^-T: in the slot of L(j, e) once row j of L is dead
    double* Fb = Tl + 36 * DBS;    // inverse of the current 16 x 16 diagonal factor
    double* rz = Fb + DBS;         // [128]
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lq = lane >> 4;
    const int nb = pw >> 4;
    const double* Tb = T + (int64_t)b * sT;
    double* Cb = Cdiag + (int64_t)b * sC;
    double* Wb = Wt + (int64_t)b * sW;
    auto tb = [](int bi, int bj) { return (bi * (bi + 1) / 2 + bj) * DBS; };
    auto eb = [](int e, int j) { return (j * (j + 1) / 2 + e) * DBS; };  // = tb(j, e)

    {
        // the lower blocks only, 16 bytes per load, ALL of a thread's loads in flight before the first LDS store (nine per
        // thread for a full tile: one round trip instead of thirty-two short ones; 5.5 -> ~3 us of the tile's 41)
        const int cnt = nb * (nb + 1) / 2 * 128;
        double2 v[9];
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const int e = min(tid + 512 * u, cnt - 1);
            const int blk = e >> 7, r = (e >> 3) & 15, c2 = e & 7;
            int bi = 0;
            while ((bi + 1) * (bi + 2) / 2 <= blk) ++bi;
            const int bj = blk - bi * (bi + 1) / 2;
            v[u] = *(const double2*)(Tb + (int64_t)(16 * bi + r) * SF_LDT + 16 * bj + 2 * c2);
        }
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            const int e = tid + 512 * u;
            if (e < cnt) {
                const int r = (e >> 3) & 15, c2 = e & 7;
                double* d = Tl + (e >> 7) * DBS + r * DLD + 2 * c2;  // (tb(bi, bj) = blk * DBS: the same enumeration)
                d[0] = v[u].x;
                d[1] = v[u].y;
            }
        }
    }
    __syncthreads();
    SF_D_STAMP(1);
    int bad = 0;
    const int oF = l15 * DLD + lq;  // operand fragment: row l15, K slice lq of instruction kk stands for k = 4 kk + lq
    for (int k = 0; k < nb; ++k) {
        const int m = nb - 1 - k;
        if (k == 7) SF_D_STAMP(8);
        // ---- U: wave t <= m holds M(k + t, k), wave t > m the block X(t - m - 1, k) of the inverse
        const int t = wave;
        const bool has = t <= m + k, isM = t <= m;
        const int ib = isM ? k + t : t - m - 1;
        double* own = has ? (isM ? Tl + tb(ib, k) : El + eb(ib, k)) : Fb;
        sf_d4 acc = {0.0, 0.0, 0.0, 0.0};
        if (has) {
            if (isM) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = lq + 4 * r;  // (the diagonal block is read symmetrically from its lower triangle)
                    acc[r] = t == 0 ? own[max(row, l15) * DLD + min(row, l15)] : own[row * DLD + l15];
                }
            }
            for (int j = isM ? 0 : ib; j < k; ++j) {
                const double* ap = (isM ? Tl + tb(ib, j) : El + eb(ib, j)) + oF;
                const double* bp = Tl + tb(k, j) + oF;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[4 * kk], bp[4 * kk], acc, 0, 0, 1);  // neg:[1,0,0]
            }
        }
        // ---- P: wave 0 factorises the diagonal block and inverts the factor in the accumulator layout
        if (k == 7) SF_D_STAMP(9);
        if (wave == 0) {
            sf_d4 a0 = acc, f, lt;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                f[r] = (lq + 4 * r) == l15 ? 1.0 : 0.0;
                lt[r] = 0.0;
            }
            double p = sf_readlane_d(a0[0], 0);
            double pkeep = 1.0;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int qj = j & 3, rj = j >> 2;
                pkeep = lane == j ? p : pkeep;
                const double rs = sfd_rsqrt(p);
                const bool in_q = lq == qj;
                const double v = (in_q && l15 > j) ? a0[rj] * rs : 0.0;  // l_ij, i = l15 > j
                const double g = in_q ? f[rj] * rs : 0.0;                // row j of F, scaled
                if (in_q) {
                    f[rj] = g;
                    lt[rj] = l15 == j ? p * rs : v;  // L^T[j][i]
                }
                if (j + 1 < 16) {
                    const double an = sf_readlane_d(a0[(j + 1) >> 2], ((j + 1) & 3) * 16 + j + 1);
                    const double vn = sf_readlane_d(v, qj * 16 + j + 1);
                    p = __builtin_fma(-vn, vn, an);
                }
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, a0, 0, 0, 1);  // neg:[1,0,0]
                f = __builtin_amdgcn_mfma_f64_16x16x4f64(v, g, f, 0, 0, 1);
            }
            const unsigned long long neg = __ballot(lane < 16 && !(pkeep > 0.0));
            if (neg && !bad) bad = 16 * k + __ffsll((long long)neg);
            double* Ekk = El + eb(k, k);  // (= own: the diagonal block of the tile has been consumed)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = lq + 4 * r;
                Fb[row * DLD + l15] = f[r];
                Ekk[l15 * DLD + row] = f[r];                      // X of the identity row block k is F^T
                // L[i][j], lower triangle of the diagonal block: straight to the matrix
                if (l15 >= row && 16 * k + row >= fp0) Cb[(int64_t)(16 * k + l15) * ldc + 16 * k + row] = lt[r];
            }
        }
        if (k == 7) SF_D_STAMP(10);
        __syncthreads();
        if (k == 7) SF_D_STAMP(11);
        // ---- X: the other blocks times F^T, through their own destination block (accumulator -> operand layout)
        if (has && wave != 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) own[(lq + 4 * r) * DLD + l15] = acc[r];
            double a[4], f4[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                a[kk] = own[oF + 4 * kk];
                f4[kk] = Fb[oF + 4 * kk];
            }
            sf_d4 x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) x = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], f4[kk], x, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) own[(lq + 4 * r) * DLD + l15] = x[r];
            if (isM && 16 * k + l15 >= fp0) {  // block (k + t, k) of L is final
#pragma unroll
                for (int r = 0; r < 4; ++r) Cb[(int64_t)(16 * ib + lq + 4 * r) * ldc + 16 * k + l15] = x[r];
            }
        }
        __syncthreads();
    }
    SF_D_STAMP(2);
    if (tid == 0 && bad && info && info[b] == 0) info[b] = info_off + bad;
    // ---- Wt[c][j] = (L_kk^-1)[c][j] (block (cb, jb) = X(jb, cb)^T, zero above)
    for (int idx = tid; idx < nb * nb * 128; idx += 512) {  // (16-byte stores: half as many store instructions per thread)
        const int blk = idx >> 7, bi = blk / nb, bj = blk - bi * nb, r = (idx >> 3) & 15, c = (idx & 7) * 2;
        const double* e = El + eb(bj, bi) + c * DLD + r;
        *(double2*)(Wb + (int64_t)(16 * bi + r) * SF_LDT + 16 * bj + c) = bj <= bi ? make_double2(e[0], e[DLD]) : make_double2(0.0, 0.0);
    }
    // ---- z_k = L_kk^-1 r_k with the explicit inverse
    if (rhs) {
        double* rb = rhs + (int64_t)b * ldr;
        if (tid < pw) rz[tid] = tid >= fp0 ? rb[tid] : 0.0;
        __syncthreads();
        {
            // four lanes per row (j = p, p + 4, ... <= i each), added by two shuffles: the 128-term chain of one lane per row
            // was 3.9 us of the tile's 41
            const int i = tid >> 2, p = tid & 3, ibk = i >> 4, ir = i & 15;
            double zacc = 0.0;
            if (i < pw)
                for (int j = p; j <= i; j += 4) zacc = __builtin_fma(El[eb(j >> 4, ibk) + (j & 15) * DLD + ir], rz[j], zacc);
            zacc += __shfl_xor(zacc, 1);
            zacc += __shfl_xor(zacc, 2);
            if (p == 0 && i < pw && i >= fp0) rb[i] = zacc;
        }
    }
#ifdef SF_TUNING
    if (stamps) {
        SF_D_STAMP(3);
        __syncthreads();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SF_D_STAMP(4);
    }
#endif
}
__global__ __launch_bounds__(512) void k_diag_lds(const double* __restrict__ T, int64_t sT, int pw, int* __restrict__ info,
                                                  int info_off, double* __restrict__ rhs, int ldr,
                                                  double* __restrict__ Cdiag, int ldc, int64_t sC,
                                                  double* __restrict__ Wt, int64_t sW, int fp0, int prio, long long* stamps) {
    extern __shared__ double dsm[];
    if (prio) __builtin_amdgcn_s_setprio(2);
    sf_diag_lds_body(T, sT, pw, info, info_off, rhs, ldr, Cdiag, ldc, sC, Wt, sW, fp0, blockIdx.x, dsm, threadIdx.x,
                     blockIdx.x == 0 ? stamps : nullptr);
}
static const int chain_prio = SF_TUNE_INT("SF_CHAIN_PRIO", 1);  // tuning aid: 0 = the chain's workgroups at normal wave priority
#define SF_DIAG_LDS_BYTES ((37 * DBS + 128) * sizeof(double))
static int sf_launch_diag128(double* T, int64_t sT, int pw, int* info, int info_off, double* rhs, int ldr, double* Cdiag,
                             int ldc, int64_t sC, double* Wt, int64_t sW, int batch, hipStream_t s, int fp0 = 0) {
    static const bool scratch = SF_TUNE_FLAG("SF_DIAG_SCRATCH");  // tuning aid: the L2-resident k_diag_mfma<512>
    if (scratch && fp0 == 0) {
        hipLaunchKernelGGL(k_diag_mfma<512>, dim3(batch), dim3(512), 0, s, T, sT, pw, info, info_off, rhs, ldr, Cdiag, ldc, sC, Wt, sW);
    } else {
        static sf_dev_once attr_once;  // devices whose function attributes are set
        SF_CHECK(sf_once_per_device(&attr_once, []() -> int {
            SF_HIP(hipFuncSetAttribute((const void*)k_diag_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            return SF_OK;
        }));
        long long* stamps = nullptr;
#ifdef SF_TUNING
        static int printed = 0;
        static long long* hs = nullptr;
        if (SF_TUNE_FLAG("SF_DIAG_STAMPS") && printed < 6) {
            if (!hs) SF_HIP(hipHostMalloc((void**)&hs, 16 * sizeof(long long)));
            for (int i = 0; i < 16; ++i) hs[i] = 0;
            stamps = hs;
        }
#endif
        hipLaunchKernelGGL(k_diag_lds, dim3(batch), dim3(512), SF_DIAG_LDS_BYTES, s, T, sT, pw, info, info_off, rhs, ldr, Cdiag, ldc,
                           sC, Wt, sW, fp0, chain_prio, stamps);
#ifdef SF_TUNING
        if (stamps) {  // (synchronises) phases of workgroup 0, us
            (void)hipStreamSynchronize(s);
            ++printed;
            fprintf(stderr, "k_diag_lds batch %d: tile load %.1f | 8 block columns %.1f (last column: U %.1f, P %.1f, barrier %.1f, X + barrier %.1f) | W store %.1f + z %.1f | drain %.1f | total %.1f us\n",
                    batch, (hs[1] - hs[0]) / 100.0, (hs[2] - hs[1]) / 100.0, (hs[9] - hs[8]) / 100.0, (hs[10] - hs[9]) / 100.0, (hs[11] - hs[10]) / 100.0,
                    (hs[2] - hs[11]) / 100.0, 0.0, (hs[3] - hs[2]) / 100.0, (hs[4] - hs[3]) / 100.0, (hs[4] - hs[0]) / 100.0);
        }
#endif
    }
    SF_LAUNCH_CHECK();
    return SF_OK;
}

// ---------------------------------------------------------------------------------------------
// Fused left-looking panel step (panel width = tile edge = 128).  One workgroup owns a 128-row slab
// of the panel [k0, k0 + pw) and does, without leaving the CU:
//   1  T  = C[slab, panel] - L[slab, :k0] L[panel rows, :k0]^T          long K, the k_gemm_nt main loop
//   2  L  = T W,  W = L_kk^-T (Wt = L_kk^-1 from k_diag_mfma)           K = pw, triangular
//   3  L -> C[slab, panel] in place;  rhs[slab] -= L z_k                (forward substitution rides along)
//   4  S  = C[slab, slab] - L L^T                                       K = pw, lower triangle only
// Steps 2 and 4 take their A operand from the accumulators through LDS (32-column chunks): the panel
// scratch T of the unfused scheme is never written or read back, and the short-K launches G and R
// (0.3 of peak, all tiles of a round in the same memory phase) are gone.  In step 2 the chunks are
// visited in DESCENDING k order: L block columns need exactly the chunks up to their own, so a wave
// dumps a T block at the moment its registers become the accumulators of the L block -- no second
// accumulator set.  Step 4 uses the 36-blocks-on-8-waves layout of sf_syrk_diag_tile.
// pw == 0: nothing but the copy of the diagonal tile to Sout (start of the factorisation).
#define CLD 33  // row stride (doubles) of the 128 x 32 chunk buffer
struct sf_panel_args {
    double* C;
    int64_t sC;
    int lda, n;
    int k0, pw;       // panel columns [k0, k0 + pw), pw in {0, 64, 128}
    int row0, nslab;  // nslab slabs of 128 rows, the first at row0 (multiple of 128); the last one may be shorter
    int slab_step;    // distance between consecutive slabs of this launch, in slabs (slab groups are interleaved)
#ifdef SF_TUNING
    int skip;         // tuning builds only (wrong results, timing only): 1 no solve, 2 no rank-pw update, 4 no main loop
#endif
    // split-K for launches that cannot fill the chip (late panels, small batches): mode 1 = ksplit workgroups per
    // slab each accumulate kchunk K-slabs and park their 128 x 128 partial sum in `part`; mode 2 = one workgroup
    // per slab adds the partial sums in fixed order (deterministic) and runs steps 2-4; mode 0 = everything at once
    int ksplit, kchunk;
    double* part;     // [tiles * ksplit][128 * 128]
    const double* Wt; // [batch] x sW: Wt[c][k] = (L_kk^-1)[c][k], row stride SF_LDT
    int64_t sW;
    double* rhs;      // [batch] x ldr or NULL
    int ldr;
    double* Sout;     // updated diagonal tile goes here (row stride ldS) instead of in place when non-NULL
    int64_t sS;
    int ldS;
    const double* genY;  // matrix-free start (see sf_gemm_args)
    const unsigned char* tilemap;
    int64_t sY;
    int ldy, mpad, nt128;
    // bordered band matrices (sf_launch_potrf_band): rows < nband have no entries further than kband columns left of
    // the diagonal, so the K loop of a slab starts at its first non-zero column; rows >= nband (the border: dense
    // rows that ride along) form one extra slab at xrow0, the last of the launch.  All 0 for dense matrices.
    int kband, nband, xrow0;
    // shifted frame (sf_potrf_front_pad): C, rhs and genY point fp (lda + 1) / fp / fp elements BEFORE the data, n / k0 /
    // row0 / the tile map count in that frame.  Rows and columns < fp are virtual (identity): every K loop starts at
    // column fp, the panel-0 accesses that would touch a virtual column are predicated.  0 for unshifted matrices.
    int fp;
    int prio;  // wave priority (s_setprio) of the whole workgroup: the chain's launches share their SIMDs with bulk workgroups
    // dataflow sequence (k_potrf_dataflow): the K range of a partial-sum task ends at K slab kstop (0: at the panel); MODE 3
    // (partial sums added, then the K slabs [ktail, panel) in the same workgroup) starts its own loop at ktail; before the
    // triangular solve the workgroup waits until *wflag >= wval (the counter the diagonal-tile task publishes)
    int kstop, ktail;
    const int* wflag;
    int wval;
    int* abort_flag;
};

// The fields of a step that differ from task to task inside k_potrf_dataflow (everything else of sf_panel_args is constant
// over a factorisation and stays in the kernel arguments: a per-task copy of the whole structure does not fit the SGPRs)
struct sf_panel_task {
    int k0, pw, row0, nslab, slab_step;
    int ksplit, kchunk, kstop, ktail;
    double* part;
    const double* Wt;
    int64_t sW;
    double* Sout;
    const int* wflag;
    int wval;
    int* abort_flag;
    int* lds_int;  // one int of LDS for the wait's broadcast
    int* top_flag; // dataflow chain / front task: counter set to top_val as soon as L is stored (before step 4)
    int top_val;
    const int* sflag;  // ... and the counter (>= sval) that says the slab's diagonal tile is ready for step 4
    int sval;
    long long* stamps;  // tuning builds: wall-clock stamps {K work done, diagonal tile there, L published, step 4 may start}
    int prio;
};
__device__ __forceinline__ sf_panel_task sf_task_of(const sf_panel_args& g) {
    sf_panel_task q;
    q.k0 = g.k0;
    q.pw = g.pw;
    q.row0 = g.row0;
    q.nslab = g.nslab;
    q.slab_step = g.slab_step;
    q.ksplit = g.ksplit;
    q.kchunk = g.kchunk;
    q.kstop = g.kstop;
    q.ktail = g.ktail;
    q.part = g.part;
    q.Wt = g.Wt;
    q.sW = g.sW;
    q.Sout = g.Sout;
    q.wflag = g.wflag;
    q.wval = g.wval;
    q.abort_flag = g.abort_flag;
    q.lds_int = nullptr;
    q.top_flag = nullptr;
    q.top_val = 0;
    q.sflag = nullptr;
    q.sval = 0;
    q.stamps = nullptr;
    q.prio = g.prio;
    return q;
}

// a pointer the compiler must treat as wave-uniform (an SGPR pair): the operand base of the direct-to-LDS loads
__device__ __forceinline__ const double* sf_uniform_ptr(const double* p) {
    const unsigned long long v = (unsigned long long)p;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return (const double*)(((unsigned long long)hi << 32) | lo);
}

// granule swizzle of the main loop's LDS image (see k_chol_panel)
__device__ __forceinline__ int sf_swz(int row) {
    const int t = (row >> 1) & 7;
    return t ^ ((((t >> 1) ^ (t >> 2)) & 1) << 1);
}

// one 16-wide K block of the triangular solve for the 16-column blocks ni >= NI_LO of a wave
template <int NI_LO>
__device__ __forceinline__ void sf_solve_step(sf_d4 (&acc)[2][4], const double* Ab, const double* Bb) {
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        double a[2], bb[4];
#pragma unroll
        for (int i = 0; i < 2; ++i) a[i] = Ab[i * 16 * CLD + ks * 4];
#pragma unroll
        for (int i = NI_LO; i < 4; ++i) bb[i] = Bb[i * 16 * GLD + ks * 4];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = NI_LO; ni < 4; ++ni)
                acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], bb[ni], acc[mi][ni], 0, 0, 0);
    }
}

// ---- dataflow sequence (k_potrf_dataflow): dependencies between workgroups of ONE launch ------------------------
// A producer finishes its global stores, __syncthreads(), then ONE lane: agent-scope release (write-back of the XCD's L2),
// s_waitcnt by hand (the compiler may drop its own when the wave's scoreboard is provably empty), relaxed agent-scope
// store / add on a monotone counter.  A consumer: ONE lane polls the counter with relaxed agent-scope loads (L2-served,
// s_sleep between polls), then ONE agent-scope acquire (invalidates this CU's L1), __syncthreads(), plain loads.
// (/opt/skills/guides/MI355X_MICROARCH.md, "Valid forms".)  Every wait is bounded: after SF_DF_TIMEOUT_TICKS of the 100 MHz
// wall clock the waiter raises the launch's abort flag, which every other wait and the task dispenser observe.
#define SF_DF_TIMEOUT_TICKS 400000000LL  // 4 s
// ... and the launch is also aborted when NO task of the launch has completed for SF_DF_STALL_TICKS while a workgroup was
// waiting (round 6): every task end bumps a progress counter (abort_flag[5]); the longest task of the largest matrix the tables
// hold (N = 16384: one slab's 1024 K slabs) runs ~5 ms, so 25 ms without a single completion chip-wide means the workgroups
// that hold the claimed tasks are not running -- a device shared with other processes (profiles/r05_g_shared_gpu_abort.txt: the
// stall begins mid-launch, an arrival gate at the head of the kernel would not see it).  The caller's fall-back then costs
// ~25 ms + one factorisation on the launch sequences instead of 4 s.  abort_flag[6] counts the workgroups that started (a
// diagnostic: grid not co-resident), abort_flag[7] != 0 replaces the bound (units of 2^16 ticks; tuning builds).
#define SF_DF_STALL_TICKS 2500000LL  // 25 ms
#define SF_DF_ABORT_TIMEOUT 1
#define SF_DF_ABORT_STALL 2
__device__ __forceinline__ int sf_df_load(const int* flag) {
    return __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the waiter that raises the abort flag leaves what it was waiting for behind it: abort_flag[1..] = {counter (offset from the abort
// flag, in ints), target, value} of the first counter that had not arrived (tuning builds print it)
// (abort_flag[8..9]: address of the process's abort record in host memory, sf_df_diag -- what the caller's warning quotes:
// {aborted launches, reason, workgroups that had started, grid, ticks the reporting wait had lasted, tasks completed})
__device__ __forceinline__ void sf_df_report(int* abort_flag, const int* f, int target, int reason = SF_DF_ABORT_TIMEOUT,
                                             long long waited = 0) {
    if (__hip_atomic_exchange(abort_flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        if (f) {
            abort_flag[1] = (int)(f - abort_flag);
            abort_flag[2] = target;
            abort_flag[3] = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        long long* diag = (long long*)__hip_atomic_load((long long*)(abort_flag + 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (diag) {
            diag[1] = reason;
            diag[2] = sf_df_load(abort_flag + 6);
            diag[3] = gridDim.x;
            diag[4] = waited;
            diag[5] = sf_df_load(abort_flag + 5);
            __hip_atomic_fetch_add(diag, 1LL, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__device__ __forceinline__ long long sf_df_stall_ticks(const int* abort_flag) {
    const int o = abort_flag[7];
    return o ? (long long)o << 16 : SF_DF_STALL_TICKS;
}
// Waits until *f1 >= t1 and *f2 >= t2 and *f3 >= t3 (NULL flags are skipped), then ONE acquire for all of them.  `probe`
// (optional) is only looked at, before the acquire: *probe_ok tells whether it had reached its target -- the data it guards
// is then covered by this acquire and needs no wait of its own later.  Returns false when the launch is being aborted.
// (s_okp: one int of LDS -- the kernels keep their LDS image at offset 0 of the workgroup's allocation, so no static __shared__
// variable may exist beside the dynamic buffer: with sm at offset 16 the direct-to-LDS operand loads lose their alignment)
// `rescue` (queued tasks' waits BEFORE their bodies only): a callable that looks for a ready chain / front task nobody has
// claimed and claims it; after SF_DF_RESCUE_TICKS inside one wait the polling lane calls it every ~50 us.  When it returns
// true the wait ends with SF_DF_DEFERRED: the workgroup sets its task aside, runs the chain task it has just claimed and
// comes back (k_potrf_dataflow).  This is what makes the schedule live BY CONSTRUCTION: chain and front tasks are claimed by
// whoever finds them ready at the dispenser, and a claim can be missed (see there); a workgroup that waits before a body
// holds nothing but its task number, and in-body waits only ever depend on tasks that are already running.  The normal path
// never gets here: waits that long mean the chip is starved of chain progress anyway.
#define SF_DF_RESCUE_TICKS 50000LL  // 500 us of the 100 MHz wall clock
#define SF_DF_DEFERRED 4
struct sf_df_no_rescue {
    __device__ __forceinline__ bool operator()() const { return false; }
};
// The rare part of a wait (every 32nd poll), out of line: the waits are inlined at a dozen sites of a kernel whose task loop is
// 100 KB of code -- with the abort record and the stall bound inlined as well every site grew, and launches of 32-64 matrices
// ran 1.2 % slower (same-box A/B, both orders: profiles/r06_b_dataflow_wait_code_size_ab.txt).
// Returns 0: keep polling; 1: give up (the launch is being aborted, by somebody else or by this call); 2: keep polling, and
// the wait has lasted long enough for the caller to look for an unclaimed chain task (SF_DF_RESCUE_TICKS).
struct sf_df_watch {
    long long t0, tp;  // start of the wait; when the launch's progress counter last moved, as seen from this wait
    int pg0;
};
__device__ __attribute__((noinline)) int sf_df_wait_slow(sf_df_watch& w, int* abort_flag, const int* f1, int t1, const int* f2, int t2,
                                                         const int* f3, int t3, const bool look_at_progress) {
    if (sf_df_load(abort_flag) != 0) return 1;
    const long long now = wall_clock64();
    const long long waited = now - w.t0;
    int reason = 0;
    if (look_at_progress) {  // (every ~0.3 ms: one more L2 round trip in the polling loop)
        const int pg = sf_df_load(abort_flag + 5);
        if (pg != w.pg0) {
            w.pg0 = pg;
            w.tp = now;
        } else if (now - w.tp > sf_df_stall_ticks(abort_flag)) {  // nothing completes any more: see SF_DF_STALL_TICKS
            reason = SF_DF_ABORT_STALL;
        }
    }
    // (abort_flag[4]: the bound in units of 2^20 ticks when the host asked for another one -- tuning builds)
    if (!reason && waited > SF_DF_TIMEOUT_TICKS && (abort_flag[4] == 0 || (waited >> 20) > abort_flag[4])) reason = SF_DF_ABORT_TIMEOUT;
    if (reason) {
        const bool m1 = f1 && sf_df_load(f1) < t1, m2 = f2 && sf_df_load(f2) < t2;
        sf_df_report(abort_flag, m1 ? f1 : (m2 ? f2 : f3), m1 ? t1 : (m2 ? t2 : t3), reason, waited);
        return 1;
    }
    return waited > SF_DF_RESCUE_TICKS ? 2 : 0;
}
template <class RESCUE>
__device__ __forceinline__ int sf_df_wait_r(const int* f1, int t1, const int* f2, int t2, const int* f3, int t3,
                                            const int* probe, int tprobe, bool* probe_ok, int* abort_flag, const int tid,
                                            int* s_okp, RESCUE&& rescue, const bool can_rescue) {
    if (tid == 0) {
        int ok = 1;
        // (short-circuit on purpose: a poller asks for the first counter that is missing only -- polls of all three, every
        // time, from a few hundred waiting workgroups slowed the launch by 2 %)
        auto ready = [&]() {
            return (!f1 || sf_df_load(f1) >= t1) && (!f2 || sf_df_load(f2) >= t2) && (!f3 || sf_df_load(f3) >= t3);
        };
        if (!ready()) {
            sf_df_watch w;
            w.t0 = w.tp = wall_clock64();
            w.pg0 = sf_df_load(abort_flag + 5);
            unsigned it = 0;
            for (;;) {
                __builtin_amdgcn_s_sleep(4);
                if (ready()) break;
                if ((++it & 31) == 0) {
                    const int r = sf_df_wait_slow(w, abort_flag, f1, t1, f2, t2, f3, t3, (it & 255) == 0);
                    if (r == 1) {
                        ok = 0;
                        break;
                    }
                    if (r == 2 && can_rescue && rescue()) {
                        ok = SF_DF_DEFERRED;
                        break;
                    }
                }
            }
        }
        if (probe && sf_df_load(probe) >= tprobe) ok |= 2;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *s_okp = ok;
    }
    __syncthreads();
    const int ok = __builtin_amdgcn_readfirstlane(*s_okp);
    __syncthreads();  // (s_ok is rewritten by the next wait)
    if (probe_ok) *probe_ok = (ok & 2) != 0;
    return ok & (1 | SF_DF_DEFERRED);
}
__device__ __forceinline__ bool sf_df_wait(const int* f1, int t1, const int* f2, int t2, const int* f3, int t3,
                                           const int* probe, int tprobe, bool* probe_ok, int* abort_flag, const int tid,
                                           int* s_okp) {
    return sf_df_wait_r(f1, t1, f2, t2, f3, t3, probe, tprobe, probe_ok, abort_flag, tid, s_okp, sf_df_no_rescue(), false) == 1;
}
__device__ __forceinline__ bool sf_df_wait(const int* flag, int target, int* abort_flag, const int tid, int* s_okp) {
    return sf_df_wait(flag, target, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, abort_flag, tid, s_okp);
}
// call after __syncthreads(): every wave's stores have been issued and waited for
__device__ __forceinline__ void sf_df_release() {
#ifndef SF_EXP_DF_NORELEASE  // (timing experiment, stale reads possible: what do the L2 write-backs of the releases cost?)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void sf_df_set(int* flag, int value) {
    __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sf_df_add(int* flag, int value) {
    __hip_atomic_fetch_add(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// MODE 0: the whole step; 1: split-K partial sums only; 2: partial sums added in split order + steps 2-4; 3: as 2, with the
// K slabs [g.ktail, panel) accumulated by this workgroup after the partial sums (dataflow sequence: the chain's step).
// `id` = tile (MODE 1: tile * ksplit + split) index; sm / red: 4 * GT * GLD + 2 * GT doubles of LDS.
// (GA: `const sf_panel_args`, or the same in the constant address space -- the kernel arguments of k_potrf_dataflow)
// (MODE 3 with ksplit = 0, ktail = 0 is MODE 0, and with ktail = the panel's K slab count it is MODE 2: k_potrf_dataflow runs
// every task type but the partial sums through ONE inlined copy of <3> -- see there.)
template <bool RHS, int MODE, class GA>
__device__ __forceinline__ void sf_panel_body(GA& g, const sf_panel_task& tk, const int id, double* __restrict__ sm,
                                              double (*red)[GT], const int tid) {
    constexpr int TM = 2, TN = 4;
    double(*As)[GT * GLD] = (double(*)[GT * GLD]) sm;
    double(*Bs)[GT * GLD] = (double(*)[GT * GLD])(sm + 2 * GT * GLD);
    double* Ach = sm;  // 128 x CLD chunk buffer of the epilogue (aliases As)

    // (integer division runs on the VALU: without the readfirstlane its wave-uniform results -- and every address and loop
    // bound derived from them -- would live in VGPRs)
    const int tile = __builtin_amdgcn_readfirstlane(MODE == 1 ? id / tk.ksplit : id);
    const int sp = __builtin_amdgcn_readfirstlane(MODE == 1 ? id - tile * tk.ksplit : 0);
    const int b = __builtin_amdgcn_readfirstlane(tile / tk.nslab);
    const int sl = tile - b * tk.nslab;
    const int row0 = (g.xrow0 && sl == tk.nslab - 1) ? g.xrow0 : tk.row0 + sl * tk.slab_step * GT;
    const int rows_here = min(GT, ((g.nband && row0 < g.nband) ? g.nband : g.n) - row0);
    const int pw = tk.pw, k0 = tk.k0;
    const int cfp = k0 == 0 ? g.fp : 0;  // panel columns below cfp are virtual (zero below the diagonal tile)
    if (tk.prio) __builtin_amdgcn_s_setprio(2);

    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // rows wm*32.., cols wn*64..; waves w and w + 4 share a SIMD: they get different column halves, because in
    // the triangular solve the two halves have different amounts of work
    const int wm = w >> 1, wn = (w ^ (w >> 2)) & 1;
    const int l15 = lane & 15, lq = lane >> 4;
    double* Cb = g.C + (int64_t)b * g.sC;

    sf_d4 acc[TM][TN];
    if (pw > 0) {
        // ---------------------------------------------------------------- 1: long-K update
        const int lr = tid >> 3, lc = (tid & 7) * 2;
        const double* Ap[2];
        const double* Bp[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            Ap[p] = Cb + (int64_t)(row0 + min(lr + 64 * p, rows_here - 1)) * g.lda + lc;
            Bp[p] = Cb + (int64_t)(k0 + min(lr + 64 * p, pw - 1)) * g.lda + lc;
        }
        // Operand staging: DIRECT global -> LDS loads (global_load_lds_dwordx4: no staging registers, no ds_write
        // pass).  A wave instruction deposits 64 consecutive 16-byte granules = 8 unpadded rows of a 16-double K slab;
        // bank conflicts are avoided by an XOR swizzle of the granule index with sf_swz(row), applied on the SOURCE
        // address here and on the fragment reads below (the LDS image itself is lane-linear).  The swizzle is made
        // for the lane groups of ds_read_b128 ({0-3,12-15,20-27}, {4-11,16-19,28-31}, ...: each holds all 16 rows of a
        // fragment once, rows 0-3 / 12-15 with one granule column and rows 4-11 with the column two further): with
        // t = (row >> 1) & 7, rows with t in {2,3,4,5} get t ^ 2, the others t -- 16 distinct 16-byte bank slots.
        const int grow = lane >> 3, gpos = lane & 7;  // row within the 8-row group, granule slot within the row
        // (addresses = a wave-uniform base in SGPRs, advanced along K by scalar adds, + a 32-bit lane offset: four VGPRs
        // instead of four 64-bit pointers advanced by VALU adds -- the kernel sits at the 128-VGPR limit, and a pointer that
        // spills is reloaded inside the K loop, where the wait for the scratch load also waits for the operand loads)
        unsigned Aoff[2], Boff[2];
#ifdef SF_EXP_AL2  // timing only: every slab streams the rows of the panel's first slab (L2-resident A operand)
        const double* Abase = sf_uniform_ptr(Cb + (int64_t)(k0 + (k0 + 2 * GT <= g.n ? GT : 0)) * g.lda);
#else
        const double* Abase = sf_uniform_ptr(Cb + (int64_t)row0 * g.lda);
#endif
        const double* Bbase = sf_uniform_ptr(Cb + (int64_t)k0 * g.lda);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int row = 16 * w + 8 * q + grow;
            const int c = gpos ^ sf_swz(row);
            Aoff[q] = (unsigned)(min(row, rows_here - 1) * g.lda + 2 * c) * 8u;
            Boff[q] = (unsigned)(min(row, pw - 1) * g.lda + 2 * c) * 8u;
        }
        typedef __attribute__((address_space(3))) void* lds_ptr;
        double* A2 = sm;                // [2][128 x 16]
        double* B2 = sm + 2 * GT * GK;  // [2][128 x 16]
        // (inline asm: hipcc drains vmcnt(0) before the next LDS read of ANY buffer when it sees the builtin in
        // flight; the loads are therefore hidden from it and waited for by hand right before the barrier)
        const unsigned ldsA = (unsigned)(size_t)(lds_ptr)A2, ldsB = (unsigned)(size_t)(lds_ptr)B2;
        auto glds16 = [&](const double* sbase, unsigned voff, unsigned lds_dst) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(voff), "s"(sbase), "s"(lds_dst)
                         : "memory");
        };
        auto gload = [&](int kt, int buf) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const unsigned off = (unsigned)(buf * GT * GK + (16 * w + 8 * q) * GK) * 8u;
                glds16(sf_uniform_ptr(Abase + kt * GK), Aoff[q], ldsA + off);
                glds16(sf_uniform_ptr(Bbase + kt * GK), Boff[q], ldsB + off);
            }
        };
        auto gwait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
        const int nk_all = SF_PANEL_SKIPS(g, 4) ? 0 : k0 / GK;
        // band: the K loop starts at the first column where both operands can be non-zero (a band slab's own rows;
        // for the dense border rows the panel's rows decide -- what lies left of that was never even written)
        const int klo = g.kband ? min(max((row0 < g.nband ? row0 : k0) - g.kband, 0) / GK, nk_all) : min(g.fp / GK, nk_all);
        const int nk_lim = (MODE == 1 && tk.kstop > 0) ? min(tk.kstop, nk_all) : nk_all;
        const int kbeg = MODE == 1 ? min(klo + sp * tk.kchunk, nk_lim) : (MODE == 3 ? min(max(tk.ktail, klo), nk_all) : klo);
        const int kend = MODE == 1 ? min(kbeg + tk.kchunk, nk_lim) : (MODE == 2 ? kbeg : nk_all);
        const int nk = kend - kbeg;
        if (nk > 0) gload(kbeg, 0);

        bool generate = false;
        if (g.tilemap) generate = !g.tilemap[(int64_t)b * g.nt128 * g.nt128 + (row0 / GT) * g.nt128 + k0 / GT];
        if (MODE == 2 || (MODE == 3 && tk.ksplit > 0)) {
            // the partial sums of the split-K workgroups, added in split order
            const double* P = tk.part + (int64_t)tile * tk.ksplit * (GT * GT);
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = (sf_d4){0.0, 0.0, 0.0, 0.0};
            // Partial tiles are stored in ACCUMULATOR order -- element (mi, ni, r) of thread t at (((mi TN + ni) 2 + r / 2) 512 + t) 2
            // + r % 2 -- so that a lane reads its values as 16-byte loads, a wave instruction covers 1 KB, and eight loads are
            // in flight per wait: in the tile's row-major layout hipcc (at the 128-VGPR limit, one temporary) waited for every
            // single 8-byte load -- 256 load latencies in series, 125-180 us of the chain task's ~250 at eight partial sums.
            const double2* P2 = (const double2*)P;
            for (int q = 0; q < tk.ksplit; ++q) {
                const double2* Pq = P2 + (int64_t)q * (GT * GT / 2) + tid;
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) {
                    double2 t[2 * TN];
#pragma unroll
                    for (int j = 0; j < 2 * TN; ++j) t[j] = Pq[(mi * 2 * TN + j) * 512];
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) {
                        acc[mi][ni][0] += t[2 * ni].x;
                        acc[mi][ni][1] += t[2 * ni].y;
                        acc[mi][ni][2] += t[2 * ni + 1].x;
                        acc[mi][ni][3] += t[2 * ni + 1].y;
                    }
                }
            }
        } else if (MODE == 1 && sp > 0) {
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = (sf_d4){0.0, 0.0, 0.0, 0.0};
        } else if (generate) {
            const double* Yb = g.genY + (int64_t)b * g.sY;
            const int gr = row0 + wm * (16 * TM) + l15;
            const int gc = k0 + wn * (16 * TN) + l15;
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) acc[mi][ni] = (sf_d4){0.0, 0.0, 0.0, 0.0};
            for (int kk = 0; kk < g.mpad; kk += 4) {
                const double* yk = Yb + (int64_t)(kk + lq) * g.ldy;
                double ya[TM], yb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) ya[i] = yk[min(gr + i * 16, g.ldy + g.fp - 1)];
#pragma unroll
                for (int i = 0; i < TN; ++i) yb[i] = gc + i * 16 >= cfp ? yk[min(gc + i * 16, g.ldy + g.fp - 1)] : 0.0;
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[mi], yb[ni], acc[mi][ni], 0, 0, 0);
            }
        } else {
            const double* Cin = Cb + (int64_t)row0 * g.lda + k0;
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) {
                    const int col = wn * (16 * TN) + ni * 16 + l15;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = wm * (16 * TM) + mi * 16 + lq + 4 * r;
                        double v = 0.0;
                        if (row < rows_here && col < pw && col >= cfp) v = Cin[(int64_t)row * g.lda + col];
                        acc[mi][ni][r] = v;
                    }
                }
        }
#ifdef SF_TUNING
        if (tk.stamps && tid == 0) tk.stamps[4] = wall_clock64();  // (issue point of the last partial-sum loads)
#endif
        gwait();
        __syncthreads();
        // (the accumulators come from compiler-counted loads: consume them here, so that hipcc places its own
        // vmcnt(0) for them BEFORE the loop and not inside it, where it would also drain the hand-counted prefetch)
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(acc[mi][ni][r]));
#ifdef SF_TUNING
        if (tk.stamps && tid == 0) tk.stamps[5] = wall_clock64();  // (partial sums added, first operand slab landed)
#endif
        // fragment reads: lane (l15, lq) takes the two granules 2 lq, 2 lq + 1 of its row = the four consecutive
        // k = 4 lq .. 4 lq + 3; MFMA j of a slab uses element j of every lane, i.e. slice lq of instruction j stands
        // for k = 4 lq + j -- in both operands (K is a summation index)
        // (a wave whose 32 rows lie beyond the matrix -- the last slab of an order that is not a multiple of 128,
        // e.g. 3008 = 23.5 slabs -- leaves the matrix core to the other waves: its tile is never stored.  cfg 3:
        // 5 % of the long-K MFMA work, 5650 -> 5940 order-evals/s)
        const bool wave_live = wm * (16 * TM) < rows_here;
        auto compute = [&](int cur) {
            const double* Ab = A2 + cur * GT * GK;
            const double* Bb = B2 + cur * GT * GK;
            if (!wave_live) return;
#ifdef SF_EXP_FRAGPF
            double2 a[2][TM], bb[2][TN];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int row = wm * (16 * TM) + i * 16 + l15;
                    a[h][i] = *(const double2*)(Ab + row * GK + 2 * ((2 * lq + h) ^ sf_swz(row)));
                }
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    const int row = wn * (16 * TN) + i * 16 + l15;
                    bb[h][i] = *(const double2*)(Bb + row * GK + 2 * ((2 * lq + h) ^ sf_swz(row)));
                }
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[h][mi].x, bb[h][ni].x, acc[mi][ni], 0, 0, 1);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[h][mi].y, bb[h][ni].y, acc[mi][ni], 0, 0, 1);
                    }
#else
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                double2 a[TM], bb[TN];
#ifdef SF_EXP_NOLDSREAD
#pragma unroll
                for (int i = 0; i < TM; ++i) asm volatile("" : "=v"(a[i].x), "=v"(a[i].y));
#pragma unroll
                for (int i = 0; i < TN; ++i) asm volatile("" : "=v"(bb[i].x), "=v"(bb[i].y));
#else
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int row = wm * (16 * TM) + i * 16 + l15;
                    a[i] = *(const double2*)(Ab + row * GK + 2 * ((2 * lq + h) ^ sf_swz(row)));
                }
#pragma unroll
                for (int i = 0; i < TN; ++i) {
                    const int row = wn * (16 * TN) + i * 16 + l15;
                    bb[i] = *(const double2*)(Bb + row * GK + 2 * ((2 * lq + h) ^ sf_swz(row)));
                }
#endif
#ifdef SF_EXP_SETPRIO
                __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi].x, bb[ni].x, acc[mi][ni], 0, 0, 1);  // neg:[1,0,0]
                        acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi].y, bb[ni].y, acc[mi][ni], 0, 0, 1);
                    }
#ifdef SF_EXP_SETPRIO
                __builtin_amdgcn_s_setprio(0);
#endif
            }
#endif
        };
        for (int kt = 0; kt + 1 < nk; ++kt) {
#ifndef SF_EXP_NOGLOAD
            gload(kbeg + kt + 1, (kt & 1) ^ 1);
#endif
            compute(kt & 1);
            gwait();
#ifndef SF_EXP_NOBARRIER
            __syncthreads();
#endif
        }
        if (nk > 0) compute((nk - 1) & 1);
        __syncthreads();  // the epilogue re-uses the LDS with its own layouts
        if (MODE == 1) {
            double2* P2 = (double2*)(tk.part + ((int64_t)tile * tk.ksplit + sp) * (GT * GT)) + tid;  // (accumulator order: see MODE 2)
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int ni = 0; ni < TN; ++ni) {
                    P2[((mi * TN + ni) * 2 + 0) * 512] = make_double2(acc[mi][ni][0], acc[mi][ni][1]);
                    P2[((mi * TN + ni) * 2 + 1) * 512] = make_double2(acc[mi][ni][2], acc[mi][ni][3]);
                }
            return;
        }

        // (dataflow sequence: the long-K loop above did not need the diagonal tile's factor; everything below does)
#ifdef SF_TUNING
        if (tk.stamps && tid == 0) tk.stamps[0] = wall_clock64();
#endif
        if (tk.wflag && !sf_df_wait(tk.wflag, tk.wval, tk.abort_flag, tid, tk.lds_int)) return;
#ifdef SF_TUNING
        if (tk.stamps && tid == 0) tk.stamps[1] = wall_clock64();
#endif
        // ---------------------------------------------------------------- 2: L = T W through LDS
        const int nsb = SF_PANEL_SKIPS(g, 1) ? 0 : pw >> 4;  // 16-column blocks of the panel (4 or 8)
        const double* Wp[2];
#pragma unroll
        for (int p = 0; p < 2; ++p)
            Wp[p] = tk.Wt + (int64_t)b * tk.sW + (int64_t)min(lr + 64 * p, pw - 1) * SF_LDT + lc;
        double2 rw[2];
        auto wload = [&](int sb) {
#pragma unroll
            for (int p = 0; p < 2; ++p) rw[p] = *(const double2*)(Wp[p] + sb * 16);
        };
        auto wstore = [&](int buf) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                double* pb = &Bs[buf][(lr + 64 * p) * GLD + lc];
                pb[0] = rw[p].x;
                pb[1] = rw[p].y;
            }
        };
        // dump the two 16-column blocks of chunk q that this wave owns (accumulator -> operand layout)
        auto dump = [&](int q, bool zero) {
            if (wn != (q >> 1)) return;
#pragma unroll
            for (int nn = 0; nn < 2; ++nn)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    if (half != (q & 1)) continue;
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            Ach[(wm * (16 * TM) + mi * 16 + lq + 4 * r) * CLD + nn * 16 + l15] = acc[mi][2 * half + nn][r];
                        if (zero) acc[mi][2 * half + nn] = (sf_d4){0.0, 0.0, 0.0, 0.0};
                    }
                }
        };
        if (nsb > 0) wload(nsb - 1);
        int buf = 0;
        // (fully unrolled: chunk and block indices are compile-time constants, only wave-uniform branches remain)
#pragma unroll
        for (int sbi = 0; sbi < GT / 16; ++sbi) {
            const int sb = GT / 16 - 1 - sbi;
            if (sb >= nsb) continue;  // narrow last panel
            if (sb & 1) {  // first block of chunk sb / 2 in descending order
                __syncthreads();  // everybody is done with the previous contents of the chunk buffer / As
                dump(sb >> 1, true);
            }
            wstore(buf);
            __syncthreads();
            if (sb > 0) wload(sb - 1);
            // W[k][c] = 0 for k > c: this wave's 64 columns need the blocks k <= 4 wn + 3 only (one wave-uniform
            // branch around a straight-line body; inside it the zero blocks of W are multiplied through, which
            // leaves the not-yet-dumped T blocks and the finished sums bit-for-bit unchanged.  Skipping block by
            // block -- a switch over four straight-line bodies -- makes hipcc spill ~250 VGPRs: measured, not kept)
            if (sb <= wn * TN + (TN - 1) && wave_live) {
                const double* Ab = &Ach[(wm * (16 * TM) + l15) * CLD + (sb & 1) * 16 + lq];
                const double* Bb = &Bs[buf][(wn * (16 * TN) + l15) * GLD + lq];
                sf_solve_step<0>(acc, Ab, Bb);
            }
            buf ^= 1;
        }

        // ---------------------------------------------------------------- 3: L in place, rhs -= L z
        double* Lout = Cb + (int64_t)row0 * g.lda + k0;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int col = wn * (16 * TN) + ni * 16 + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wm * (16 * TM) + mi * 16 + lq + 4 * r;
                    if (row < rows_here && col < pw && col >= cfp) Lout[(int64_t)row * g.lda + col] = acc[mi][ni][r];
                }
            }
        if (tk.top_flag) {
            // dataflow chain task: the slab's row is final HERE -- the next chain task's K work reads L, not the diagonal tile
            // that step 4 updates and parks for this workgroup's own D(k) -- so it is published before step 4, not after it
            __syncthreads();
            if (tid == 0) {
                sf_df_release();
                sf_df_set(tk.top_flag, tk.top_val);
#ifdef SF_TUNING
                if (tk.stamps) tk.stamps[2] = wall_clock64();
#endif
            }
        }
        if (RHS && g.rhs) {
            const double* z = g.rhs + (int64_t)b * g.ldr + k0;
            double zc[TN];
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int col = wn * (16 * TN) + ni * 16 + l15;
                zc[ni] = (col < pw && col >= cfp) ? z[col] : 0.0;
            }
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double v = 0.0;
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) v = __builtin_fma(acc[mi][ni][r], zc[ni], v);
                    v += __shfl_xor(v, 1);
                    v += __shfl_xor(v, 2);
                    v += __shfl_xor(v, 4);
                    v += __shfl_xor(v, 8);
                    if (l15 == 0) red[wn][wm * (16 * TM) + mi * 16 + lq + 4 * r] = v;
                }
        }
    }

    // -------------------------------------------------------------------- 4: S = C[slab, slab] - L L^T
    // The L slab just stored is read back (L2) through the ordinary operand staging -- the accumulators are free
    // by now, so the 36 lower blocks fit one pass of 5 + 4 blocks per wave pair (see sf_syrk_diag_tile); keeping
    // L in registers and dumping it chunk by chunk needed two passes and 16 barriers.
    {
        const int p = w >> 1, h = w & 1;
        int bi[5], bj[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            if (h == 0) {
                bi[q] = 7 - p;
                bj[q] = q;
            } else {
                const int n_hi = 3 - p;  // blocks 5 .. 7-p of row 7-p, then blocks 0 .. p of row p
                const int qq = q < 4 ? q : 0;
                bi[q] = qq < n_hi ? 7 - p : p;
                bj[q] = qq < n_hi ? 5 + qq : qq - n_hi;
            }
        }
        const int nstore = h == 0 ? 5 : 4;
        const int nk2 = SF_PANEL_SKIPS(g, 2) ? 0 : pw / GK;
        const int lr = tid >> 3, lc = (tid & 7) * 2;
        const double* Lp[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) Lp[q] = Cb + (int64_t)(row0 + min(lr + 64 * q, rows_here - 1)) * g.lda + k0 + lc;
        double2 rl[2];
        auto gload2 = [&](int kt) {
#pragma unroll
            for (int q = 0; q < 2; ++q) rl[q] = kt * GK + lc >= cfp ? *(const double2*)(Lp[q] + kt * GK) : make_double2(0.0, 0.0);
        };
        auto lstore2 = [&](int buf) {
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                double* pa = &As[buf][(lr + 64 * q) * GLD + lc];
                pa[0] = rl[q].x;
                pa[1] = rl[q].y;
            }
        };
        // (dataflow front tasks start on the slab's L rows; the slab's diagonal tile -- updated by the step of the previous
        // panel, possibly still running in another workgroup -- is only needed from here on)
        if (tk.sflag && !sf_df_wait(tk.sflag, tk.sval, tk.abort_flag, tid, tk.lds_int)) return;
#ifdef SF_TUNING
        if (tk.stamps && tid == 0) tk.stamps[3] = wall_clock64();
#endif
        __syncthreads();  // the L slab is visible to every wave of the workgroup; the LDS buffers are free
        if (nk2 > 0) gload2(0);
        const double* Sin = Cb + (int64_t)row0 * g.lda + row0;
        sf_d4 acc2[5];
        // (full slabs -- all but the last of a matrix whose order is not a multiple of 128 -- take straight-line loads and
        // stores: behind per-element predicates hipcc put every access into a block of its own and waited for it there,
        // twenty load and eighteen store latencies in series per task)
        const bool full_tile = rows_here == GT && row0 >= g.fp;
        if (full_tile) {
#pragma unroll
            for (int q = 0; q < 5; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * bi[q] + lq + 4 * r, col = 16 * bj[q] + l15;
                    acc2[q][r] = Sin[(int64_t)row * g.lda + col];  // (wave pairs with four blocks read a fifth one they never store)
                }
        } else {
#pragma unroll
            for (int q = 0; q < 5; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * bi[q] + lq + 4 * r, col = 16 * bj[q] + l15;
                    if (row0 + min(row, col) < g.fp)  // virtual rows / columns of the first tile: identity
                        acc2[q][r] = row == col ? 1.0 : 0.0;
                    else
                        acc2[q][r] = (q < nstore && row < rows_here && col < rows_here) ? Sin[(int64_t)row * g.lda + col] : 0.0;
                }
        }
        if (nk2 > 0) lstore2(0);
        __syncthreads();
        auto compute2 = [&](int cur) {
            const double* S = &As[cur][l15 * GLD + lq];
#pragma unroll
            for (int ks = 0; ks < GK / 4; ++ks) {
#pragma unroll
                for (int q = 0; q < 5; ++q)
                    acc2[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(S[bi[q] * 16 * GLD + ks * 4], S[bj[q] * 16 * GLD + ks * 4],
                                                                   acc2[q], 0, 0, 1);  // neg:[1,0,0]
            }
        };
        for (int kt = 0; kt + 1 < nk2; ++kt) {
            gload2(kt + 1);
            compute2(kt & 1);
            lstore2((kt & 1) ^ 1);
            __syncthreads();
        }
        if (nk2 > 0) compute2((nk2 - 1) & 1);
        const bool parked = tk.Sout && sl == 0;  // (only the first slab of a launch is the next diagonal tile)
        double* So = parked ? tk.Sout + (int64_t)b * g.sS : Cb + (int64_t)row0 * g.lda + row0;
        const int ldo = parked ? g.ldS : g.lda;
        if (full_tile) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * bi[q] + lq + 4 * r, col = 16 * bj[q] + l15;
                    So[(int64_t)row * ldo + col] = acc2[q][r];
                }
            if (nstore == 5) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * bi[4] + lq + 4 * r, col = 16 * bj[4] + l15;
                    So[(int64_t)row * ldo + col] = acc2[4][r];
                }
            }
        } else {
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                if (q >= nstore) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * bi[q] + lq + 4 * r, col = 16 * bj[q] + l15;
                    if (row < rows_here && col < rows_here) So[(int64_t)row * ldo + col] = acc2[q][r];
                }
            }
        }
    }
    if (RHS && g.rhs && pw > 0) {
        __syncthreads();
        if (tid < rows_here) g.rhs[(int64_t)b * g.ldr + row0 + tid] -= red[0][tid] + red[1][tid];
    }
}

template <bool RHS, int MODE>
__global__ __launch_bounds__(512, 4) void k_chol_panel(sf_panel_args g) {
    __shared__ __attribute__((aligned(16))) double sm[4 * GT * GLD];
    __shared__ double red[2][GT];
    sf_panel_body<RHS, MODE>(g, sf_task_of(g), sf_xcd_remap(blockIdx.x, gridDim.x), sm, red, threadIdx.x);
}

// ---------------------------------------------------------------------------------------------
// WIDE fused panel step: a PAIR of 128-column panels [k0, k0 + 256) per launch.  ONE workgroup of 16 waves (1024
// threads, one per CU: 148 KB of LDS, 4 waves per SIMD) owns a 128-row slab and keeps the 128 x 256 tile in its
// accumulators (wave = 32 rows x 64 columns: two 16-column blocks of each panel), so the slab's L[slab, :k0] -- the
// A operand, the stream that comes from HBM -- is read once per 256 columns instead of once per 128: half the HBM
// traffic of the long-K update, three quarters of the L2 -> LDS traffic, half the tile read-modify-writes and half
// the launches of k_chol_panel.  The factorisation is POWER-bound at these batch sizes (profiles/r03_*: the same
// instruction stream with the operands kept in L2 runs 5 % faster at a 5 % higher clock), so traffic is time.
//   1  T  = C[slab, pair] - L[slab, :k0] L[pair rows, :k0]^T      K slabs of 16 through a THREE-stage LDS ring filled
//         by direct global -> LDS loads; the fragments of the next half slab are read before the barrier (the data of
//         slab k+1 is complete one barrier earlier), so no wave waits for LDS after a barrier
//   2a L1 = T1 W_k            (W_k = L_kk^-T, explicit inverse from k_diag_lds; descending 32-column chunks as in
//                              k_chol_panel: a wave dumps its T blocks when its registers become the L accumulators)
//   2b T2 -= L1 L21^T         (L21 = L[panel k+1 rows, panel k columns], left in place by the chain's narrow step)
//   2c L2 = T2 W_k+1
//   3  L -> C in place, rhs[slab] -= L1 z_k + L2 z_k+1
//   4  S  = C[slab, slab] - L L^T (K = 256): L goes from the accumulators into one 128 x 128 LDS image per panel; the 36
//         lower blocks of the tile are spread over the 16 waves (9 per SIMD) and accumulate over both panels in registers
// Same arithmetic as two consecutive k_chol_panel steps; the summation order of 2b differs (natural k order instead
// of the K-permuted fragments), so results agree to rounding, not bit for bit.
#define WST (3 * GT * GK)  // doubles per LDS stage: A 128 x 16, B 256 x 16
#define SF_PANELW_LDS ((3 * WST + 4 * GT) * sizeof(double))
struct sf_panelw_args {
    double* C;
    int64_t sC;
    int lda, n;
    int k0;            // pair columns [k0, k0 + 256), both panels full
    int row0, nslab;   // nslab slabs of 128 rows, the first at row0; the last may be shorter
    int slab_step;     // distance between the slabs of this launch, in slabs (the two slab groups are interleaved)
    const double* Wt0; // [batch] x sW: (L_kk^-1)[c][k], row stride SF_LDT
    const double* Wt1; // ... of panel k + 1
    int64_t sW;
    double* rhs;
    int ldr;
    double* Sout;      // the first slab's updated diagonal tile goes here (next diagonal tile) when non-NULL
    int64_t sS;
    int ldS;
    const double* genY;
    const unsigned char* tilemap;
    int64_t sY;
    int ldy, mpad, nt128;
    int fp;            // shifted frame, see sf_panel_args
#ifdef SF_TUNING
    long long* stamps; // tuning builds (SF_WIDE_STAMPS): 100 MHz wall-clock stamps of the phases of workgroup gridDim.x / 2
#endif
};
#ifdef SF_TUNING
#define SF_W_STAMP(i) do { if (g.stamps && blockIdx.x == gridDim.x / 2 && threadIdx.x == 0) g.stamps[i] = wall_clock64(); } while (0)
#else
#define SF_W_STAMP(i)
#endif

template <bool RHS>
__device__ __forceinline__ void sf_panelw_body(const sf_panelw_args& g, const int id, double* __restrict__ smw, const int tid) {
    constexpr int TM = 2, TN = 4;
    double* red = smw + 3 * WST;  // [4][GT]

    const int b = id / g.nslab;
    const int sl = id - b * g.nslab;
    const int row0 = g.row0 + sl * g.slab_step * GT;
    const int rows_here = min(GT, g.n - row0);
    const int k0 = g.k0;
    const int cfp = k0 == 0 ? g.fp : 0;  // pair columns below cfp are virtual (zero below the diagonal tile)

    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the four waves of a SIMD (w, w + 4, w + 8, w + 12) share a row group and take the four column groups: the
    // triangular phases give the column groups different amounts of work, every SIMD gets the same total
    const int wm = w & 3, wn = w >> 2;
    const int l15 = lane & 15, lq = lane >> 4;
    double* Cb = g.C + (int64_t)b * g.sC;
    // block ni of this wave: columns bc(ni) .. + 16 of the pair (ni 0, 1: panel k; ni 2, 3: panel k + 1)
#define WBC(ni) ((((ni) >> 1) * GT) + wn * 32 + (((ni)&1) * 16))
    const bool wave_live = wm * 32 < rows_here;

    sf_d4 acc[TM][TN];
    SF_W_STAMP(0);
    // ---------------------------------------------------------------- 1: long-K update
    {
        typedef __attribute__((address_space(3))) void* lds_ptr;
        const unsigned lds0 = (unsigned)(size_t)(lds_ptr)smw;
        const int grow = lane >> 3, gpos = lane & 7;
        // 384 rows of 8 granules per stage = 48 groups of 8 rows, three per wave; groups 0-15 are A rows, 16-47 B rows
        // (a wave-uniform base advanced along K by scalar adds + 32-bit lane offsets: see k_chol_panel)
        unsigned soff[3];
        const double* sbase = sf_uniform_ptr(Cb + g.fp);  // (K starts at column fp)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int G = 3 * w + j;
            const bool isA = G < 16;
            const int r = (isA ? G : G - 16) * 8 + grow;
            const int c = gpos ^ sf_swz(r);
            soff[j] = (unsigned)(((int64_t)(isA ? row0 + min(r, rows_here - 1) : k0 + r) * g.lda + 2 * c) * 8);
        }
        auto glds16 = [&](const double* sb, unsigned voff, unsigned lds_dst) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(voff), "s"(sb), "s"(lds_dst)
                         : "memory");
        };
        auto gload = [&](int kt, int stage) {
#pragma unroll
            for (int j = 0; j < 3; ++j) glds16(sf_uniform_ptr(sbase + kt * GK), soff[j], lds0 + (unsigned)(stage * WST * 8 + (3 * w + j) * 1024));
        };
        auto gwait = [&]() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); };
        const int nk = max(k0 - g.fp, 0) / GK;
        if (nk > 0) gload(0, 0);
        if (nk > 1) gload(1, 1);

        // start of the tile: generated as Y^T Y (never materialised) or read from C, per 128-column half
        bool gen_half[2] = {false, false};
        if (g.tilemap) {
            const unsigned char* tm = g.tilemap + (int64_t)b * g.nt128 * g.nt128 + (row0 / GT) * g.nt128 + k0 / GT;
            gen_half[0] = !tm[0];
            gen_half[1] = !tm[1];
        }
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            if (gen_half[hf]) {
                const double* Yb = g.genY + (int64_t)b * g.sY;
                const int gr = row0 + wm * 32 + l15;
                const int gc = k0 + hf * GT + wn * 32 + l15;
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int nn = 0; nn < 2; ++nn) acc[mi][2 * hf + nn] = (sf_d4){0.0, 0.0, 0.0, 0.0};
                // (two K steps per round trip: mpad = 8 is one round of loads -- the workgroup has the CU to itself, every
                // dependent round trip of the prologue is exposed; the MFMA sequence per accumulator is unchanged)
                for (int kk = 0; kk < g.mpad; kk += 8) {
                    double ya[2][TM], yb[2][2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const double* yk = Yb + (int64_t)(min(kk + 4 * u, g.mpad - 4) + lq) * g.ldy;
#pragma unroll
                        for (int i = 0; i < TM; ++i) ya[u][i] = yk[min(gr + i * 16, g.ldy + g.fp - 1)];
#pragma unroll
                        for (int i = 0; i < 2; ++i) yb[u][i] = gc + i * 16 >= cfp ? yk[min(gc + i * 16, g.ldy + g.fp - 1)] : 0.0;
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        if (kk + 4 * u >= g.mpad) break;
#pragma unroll
                        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                            for (int nn = 0; nn < 2; ++nn)
                                acc[mi][2 * hf + nn] = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[u][mi], yb[u][nn], acc[mi][2 * hf + nn], 0, 0, 0);
                    }
                }
            } else {
                const double* Cin = Cb + (int64_t)row0 * g.lda + k0;
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int nn = 0; nn < 2; ++nn) {
                        const int col = WBC(2 * hf + nn) + l15;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = wm * 32 + mi * 16 + lq + 4 * r;
                            acc[mi][2 * hf + nn][r] = (row < rows_here && col >= cfp) ? Cin[(int64_t)row * g.lda + col] : 0.0;
                        }
                    }
            }
        }
        gwait();
        __syncthreads();
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(acc[mi][ni][r]));
        SF_W_STAMP(1);

        // fragment reads (layout and swizzle of k_chol_panel; sf_swz of a fragment row depends on l15 only)
        const int sw = sf_swz(l15);
        const int e0 = 2 * ((2 * lq) ^ sw), e1 = 2 * ((2 * lq + 1) ^ sw);
        const int arow = (wm * 32 + l15) * GK, brow = (GT + wn * 32 + l15) * GK;
        auto frag = [&](int stage, int h, double2(&a)[TM], double2(&bb)[TN]) {
            const double* S = smw + stage * WST + (h ? e1 : e0);
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *(const double2*)(S + arow + i * 16 * GK);
#pragma unroll
            for (int i = 0; i < TN; ++i) bb[i] = *(const double2*)(S + brow + ((i >> 1) * GT + (i & 1) * 16) * GK);
        };
        // (instructions lo .. hi - 1 of a 16-MFMA burst, in the order  x: (mi, ni) ...,  y: (mi, ni) ...)
        auto mfma_part = [&](const double2(&a)[TM], const double2(&bb)[TN], auto lo_t, auto hi_t) {
            constexpr int lo = decltype(lo_t)::value, hi = decltype(hi_t)::value;
#pragma unroll
            for (int i = lo; i < hi; ++i) {
                const int y = i >> 3, mi = (i >> 2) & 1, ni = i & 3;
                acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(y ? a[mi].y : a[mi].x, y ? bb[ni].y : bb[ni].x, acc[mi][ni], 0, 0, 1);  // neg:[1,0,0]
            }
        };
        typedef std::integral_constant<int, 0> I0;
        typedef std::integral_constant<int, 8> I8;
        typedef std::integral_constant<int, 16> I16;
        double2 a0[TM], b0[TN], a1[TM], b1[TN];
        int s0 = 0, s1 = 1, s2 = 2;  // stages of slab kt, kt + 1, kt + 2
        if (nk > 0 && wave_live) frag(0, 0, a0, b0);
        for (int kt = 0; kt < nk; ++kt) {
            // Order of a slab: the first eight MFMAs (their fragments were read before the barrier) go out BEFORE the slab's
            // loads and fragment reads.  All sixteen waves leave the barrier in step; whatever stands between it and a wave's first
            // MFMA -- three loads with their M0 moves, six LDS reads -- is time in which no wave of the CU feeds the matrix
            // pipes (one workgroup per CU: nobody else does).  Same box, potrf of cfg 2 (ms): loads first 47.1, after 4 / 8 / 12
            // MFMAs 46.7 / 46.5 / 46.55; loads after all sixteen 52.0 (then they no longer land within the slab);
            // profiles/r05_p_wide_k_loop_issue_order_ab.txt.  (s_setprio is a scheduling boundary for hipcc: the order holds.
            // The bursts run at raised priority: a wave with matrix work ready goes before the waves that are still issuing their
            // fragment reads -- cfg 2 48.77 -> 48.53 ms on the same box, three runs each.)
            if (wave_live) {
                __builtin_amdgcn_s_setprio(1);
                mfma_part(a0, b0, I0(), I8());
                __builtin_amdgcn_s_setprio(0);
            }
#ifndef SF_EXPW_NOGLOAD
            if (kt + 2 < nk) gload(kt + 2, s2);
#endif
            if (wave_live) {
                frag(s0, 1, a1, b1);
                __builtin_amdgcn_s_setprio(1);
                mfma_part(a0, b0, I8(), I16());
                __builtin_amdgcn_s_setprio(0);
                if (kt + 1 < nk) frag(s1, 0, a0, b0);  // complete since the previous barrier
                __builtin_amdgcn_s_setprio(1);
                mfma_part(a1, b1, I0(), I16());
                __builtin_amdgcn_s_setprio(0);
            }
            gwait();
#ifndef SF_EXPW_NOBARRIER
            __syncthreads();
#endif
            const int t = s0;
            s0 = s1;
            s1 = s2;
            s2 = t;
        }
    }

    // ---------------------------------------------------------------- 2: triangular solves through LDS
    // (round 6: the chunk buffers alternate -- a chunk is dumped while the previous one is still being read, so the barrier
    // that used to stand in front of every dump is gone: 12 of the epilogue's ~36 workgroup-wide barriers)
    double* Ach0 = smw;                         // [2][128][CLD] chunks of the A operand (accumulator -> operand layout)
    double* Bs = smw + 2 * GT * CLD;            // [2][128][GLD] 16-column blocks of W  /  [2][128][CLD] chunks of L21
    const int lr = tid >> 3, lc = (tid & 7) * 2;  // staging: 128 rows x 8 threads
    // L = T W on the blocks NB, NB + 1 of every wave (NB = 0: panel k, NB = 2: panel k + 1), K blocks in descending order
    // (rw: the K block 7 of W, requested by the caller ahead of the phase that precedes the solve: every global round trip of
    // the epilogue -- W, L21, z, the diagonal tile -- is in flight before the phase that needs it: with one workgroup per CU
    // nothing else hides them; 78 -> ~66 us of fixed cost per task, profiles/r05_e_wide_kernel_phases_*.txt)
    auto w_rows = [&](const double* Wt) { return Wt + (int64_t)b * g.sW + (int64_t)lr * SF_LDT + lc; };
    auto solve = [&](const double* Wt, double2 rw, auto nbtag) {
        constexpr int NB = decltype(nbtag)::value;
        const double* Wp = w_rows(Wt);
        int buf = 0;
#pragma unroll
        for (int sbi = 0; sbi < 8; ++sbi) {
            const int sb = 7 - sbi;
            double* Ach = Ach0 + ((sb >> 1) & 1) * (GT * CLD);  // (chunk sb / 2: its buffer was last read two chunks = four barriers ago)
            // (the phase before the second solve -- step 2b -- reads the same buffers: one barrier in front of its first dump)
            if (sb == 7 && NB != 0) __syncthreads();
            if (sb & 1) {  // first block of chunk sb / 2: its owner waves hand their T blocks over
                if (wn == (sb >> 1)) {
#pragma unroll
                    for (int nn = 0; nn < 2; ++nn)
#pragma unroll
                        for (int mi = 0; mi < TM; ++mi) {
#pragma unroll
                            for (int r = 0; r < 4; ++r)
                                Ach[(wm * 32 + mi * 16 + lq + 4 * r) * CLD + nn * 16 + l15] = acc[mi][NB + nn][r];
                            acc[mi][NB + nn] = (sf_d4){0.0, 0.0, 0.0, 0.0};
                        }
                }
            }
            {
                double* pb = Bs + buf * (GT * GLD) + lr * GLD + lc;
                pb[0] = rw.x;
                pb[1] = rw.y;
            }
            __syncthreads();
            if (sb > 0) rw = *(const double2*)(Wp + (sb - 1) * 16);
            // W[k][c] = 0 for k > c: the wave's columns (blocks 2 wn, 2 wn + 1 of the panel) need K blocks <= 2 wn + 1
            if (sb <= 2 * wn + 1 && wave_live) {
                const double* Ab = Ach + (wm * 32 + l15) * CLD + (sb & 1) * 16 + lq;
                const double* Bb = Bs + buf * (GT * GLD) + (wn * 32 + l15) * GLD + lq;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    double a[TM], bb[2];
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[i] = Ab[i * 16 * CLD + ks * 4];
#pragma unroll
                    for (int i = 0; i < 2; ++i) bb[i] = Bb[i * 16 * GLD + ks * 4];
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                        for (int nn = 0; nn < 2; ++nn)
                            acc[mi][NB + nn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], bb[nn], acc[mi][NB + nn], 0, 0, 0);
                }
            }
            buf ^= 1;
        }
    };
    // ---- requested now, used later: first rows of W_k
    double2 rw0 = *(const double2*)(w_rows(g.Wt0) + 7 * 16);
    __syncthreads();  // (the main loop's last reads of the ring are done)
    SF_W_STAMP(2);
    solve(g.Wt0, rw0, std::integral_constant<int, 0>());
    SF_W_STAMP(3);
    double2 rw1 = *(const double2*)(w_rows(g.Wt1) + 7 * 16);  // (in flight during step 2b)

    // 2b: T2 -= L1 L21^T, 32 columns of L1 at a time (chunk q = the blocks of the waves wn == q)
    {
        const double* L21 = Cb + (int64_t)(k0 + GT + lr) * g.lda + k0 + (tid & 7) * 4;
        auto l21 = [&](int q, double2& l0, double2& l1) {
            const bool real = (tid & 7) * 4 + q * 32 >= cfp;
            l0 = real ? *(const double2*)(L21 + q * 32) : make_double2(0.0, 0.0);
            l1 = real ? *(const double2*)(L21 + q * 32 + 2) : make_double2(0.0, 0.0);
        };
        double2 l0, l1, n0 = make_double2(0.0, 0.0), n1 = n0;
        l21(0, l0, l1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q + 1 < 4) l21(q + 1, n0, n1);  // (the next chunk is in flight under this chunk's MFMAs)
            double* Ach = Ach0 + (q & 1) * (GT * CLD);
            double* Bc = Bs + (q & 1) * (GT * CLD);  // [128][CLD]
            if (q == 0) __syncthreads();  // the first solve's last reads of the buffers are done
            if (wn == q) {
#pragma unroll
                for (int nn = 0; nn < 2; ++nn)
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            Ach[(wm * 32 + mi * 16 + lq + 4 * r) * CLD + nn * 16 + l15] = acc[mi][nn][r];
            }
            {
                double* pb = Bc + lr * CLD + (tid & 7) * 4;
                pb[0] = l0.x;
                pb[1] = l0.y;
                pb[2] = l1.x;
                pb[3] = l1.y;
            }
            __syncthreads();
            if (wave_live) {
                const double* Ab = Ach + (wm * 32 + l15) * CLD + lq;
                const double* Bb = Bc + (wn * 32 + l15) * CLD + lq;
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    double a[TM], bb[2];
#pragma unroll
                    for (int i = 0; i < TM; ++i) a[i] = Ab[i * 16 * CLD + ks * 4];
#pragma unroll
                    for (int i = 0; i < 2; ++i) bb[i] = Bb[i * 16 * CLD + ks * 4];
#pragma unroll
                    for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                        for (int nn = 0; nn < 2; ++nn)
                            acc[mi][2 + nn] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], bb[nn], acc[mi][2 + nn], 0, 0, 1);  // neg
                }
            }
            l0 = n0;
            l1 = n1;
        }
    }
    SF_W_STAMP(4);
    solve(g.Wt1, rw1, std::integral_constant<int, 2>());
    SF_W_STAMP(5);
    // ---- the slab's diagonal tile (step 4) is requested before the stores of step 3
    constexpr int TLD = 130;
    const int nsb = wn == 0 ? 3 : 2;
    int sbi[3], sbj[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int t = min(wm + 4 * (wn + 4 * j), 35);
        int bi = 0;
        while ((bi + 1) * (bi + 2) / 2 <= t) ++bi;
        sbi[j] = bi;
        sbj[j] = t - bi * (bi + 1) / 2;
    }
    sf_d4 acc2[3];
    {
        const double* Sin = Cb + (int64_t)row0 * g.lda + row0;
        if (rows_here == GT) {  // (full slab: straight-line loads -- see k_chol_panel, step 4)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * sbi[j] + lq + 4 * r, col = 16 * sbj[j] + l15;
                    acc2[j][r] = Sin[(int64_t)row * g.lda + col];  // (a wave with two blocks reads a third one it never stores)
                }
        } else {
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * sbi[j] + lq + 4 * r, col = 16 * sbj[j] + l15;
                    acc2[j][r] = (j < nsb && row < rows_here && col < rows_here) ? Sin[(int64_t)row * g.lda + col] : 0.0;
                }
        }
    }
    double zc[TN] = {0.0, 0.0, 0.0, 0.0};
    if (RHS && g.rhs) {
        const double* z = g.rhs + (int64_t)b * g.ldr + k0;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) zc[ni] = WBC(ni) + l15 >= cfp ? z[WBC(ni) + l15] : 0.0;
    }

    // ---------------------------------------------------------------- 3: L in place, rhs -= L z
    {
        double* Lout = Cb + (int64_t)row0 * g.lda + k0;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi)
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int col = WBC(ni) + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = wm * 32 + mi * 16 + lq + 4 * r;
                    if (row < rows_here && col >= cfp) Lout[(int64_t)row * g.lda + col] = acc[mi][ni][r];
                }
            }
        if (RHS && g.rhs) {
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double v = 0.0;
#pragma unroll
                    for (int ni = 0; ni < TN; ++ni) v = __builtin_fma(acc[mi][ni][r], zc[ni], v);
                    v += __shfl_xor(v, 1);
                    v += __shfl_xor(v, 2);
                    v += __shfl_xor(v, 4);
                    v += __shfl_xor(v, 8);
                    if (l15 == 0) red[wn * GT + wm * 32 + mi * 16 + lq + 4 * r] = v;
                }
        }
    }

    SF_W_STAMP(6);
    // ---------------------------------------------------------------- 4: S = C[slab, slab] - L L^T, K = 256
    // L is taken from the accumulators through ONE LDS image per panel (Ts, 128 x 128, row stride 130: the operand reads of
    // a wave instruction hit distinct 8-byte banks per half wave), not read back from global memory: the 36 lower blocks
    // of the tile are spread 9 per SIMD (3 + 2 + 2 + 2 over its waves: block t = wm + 4 (wn + 4 j) of the row-major
    // lower-triangular enumeration) and accumulate over both panels in registers -- four barriers, no K-slab staging loop,
    // no cross-wave reduction.
    {
        double* Ts = smw;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            __syncthreads();  // the previous contents of the LDS image are dead
#pragma unroll
            for (int nn = 0; nn < 2; ++nn)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        Ts[(wm * 32 + mi * 16 + lq + 4 * r) * TLD + wn * 32 + nn * 16 + l15] = acc[mi][2 * half + nn][r];
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (j >= nsb) continue;
                const double* Pa = Ts + (sbi[j] * 16 + l15) * TLD + lq;
                const double* Pb = Ts + (sbj[j] * 16 + l15) * TLD + lq;
#pragma unroll 8
                for (int ks = 0; ks < GT / 4; ++ks)
                    acc2[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(Pa[ks * 4], Pb[ks * 4], acc2[j], 0, 0, 1);  // neg:[1,0,0]
            }
        }
        SF_W_STAMP(7);
        const bool parked = g.Sout && sl == 0;  // (only the first slab of a launch is the next diagonal tile)
        double* So = parked ? g.Sout + (int64_t)b * g.sS : Cb + (int64_t)row0 * g.lda + row0;
        const int ldo = parked ? g.ldS : g.lda;
        if (rows_here == GT) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * sbi[j] + lq + 4 * r, col = 16 * sbj[j] + l15;
                    So[(int64_t)row * ldo + col] = acc2[j][r];
                }
            if (nsb == 3) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * sbi[2] + lq + 4 * r, col = 16 * sbj[2] + l15;
                    So[(int64_t)row * ldo + col] = acc2[2][r];
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (j >= nsb) continue;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * sbi[j] + lq + 4 * r, col = 16 * sbj[j] + l15;
                    if (row < rows_here && col < rows_here) So[(int64_t)row * ldo + col] = acc2[j][r];
                }
            }
        }
    }
    if (RHS && g.rhs) {
        // (red was written before the barriers of step 4)
        if (tid < rows_here)
            g.rhs[(int64_t)b * g.ldr + row0 + tid] -= (red[tid] + red[GT + tid]) + (red[2 * GT + tid] + red[3 * GT + tid]);
    }
#ifdef SF_TUNING
    if (g.stamps) {
        __syncthreads();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        SF_W_STAMP(8);
    }
#endif
#undef WBC
}
template <bool RHS>
__global__ __launch_bounds__(1024) void k_chol_panel_w(sf_panelw_args g) {
    extern __shared__ __attribute__((aligned(16))) double smw[];
    // (several tasks per workgroup -- the 5-15 us a CU needs to start a 16-wave workgroup with 148 KB of LDS amortised -- measured
    // without any gain at cfg 2 and cfg 3: profiles/r05_h_wide_tasks_per_workgroup_ab.txt)
    sf_panelw_body<RHS>(g, sf_xcd_remap(blockIdx.x, gridDim.x), smw, threadIdx.x);
}

// The per-matrix scratch strides are skewed by a few hundred bytes: with strides that are multiples of
// 32 KiB every workgroup of the batch touches the same HBM channel / L2 bank at the same time (measured:
// 3.6 us per dependent load in k_diag_mfma before the skew).
#define SF_TSKEW 40
// Split-K policy of the fused factorisation: a launch of `wgs` workgroups with `nk` K-slabs each is split
// `S` ways when it cannot fill the chip (512 resident workgroups): late panels and small batches, where the
// time of a launch is the time of ONE workgroup's K loop.  S is a power of two, every part keeps >= 8 slabs.
#define SF_CHIP_WGS 512
#define SF_SPLIT_MAX 8
static size_t sf_split_region_tiles(void) { return 2 * SF_CHIP_WGS; }  // partial-sum tiles per region
__device__ __forceinline__ size_t sf_split_region_tiles_dev(void) { return 2 * SF_CHIP_WGS; }
static int sf_split_policy(long long wgs, int nk) {
    static const int force = SF_TUNE_INT("SF_CHOL_SPLIT", -1);  // tuning aid
    int S = 1;
    // (a split launch stops at 384 of the 512 slots: two slab groups are in flight and the chain's launches need room --
    // N = 4096, cap 512 / 384 / 256 / 192: B = 16 11.22 / 11.18 / 11.26 / 11.71 ms, 32: 16.06 / 15.78 / 16.09 / 17.28,
    // 64: 27.42 / 27.05 / 26.89 / 28.7)
    static const int cap = SF_TUNE_INT("SF_SPLIT_CAP", 384);
    while (2 * S <= SF_SPLIT_MAX && wgs * 2 * S <= cap && nk / (2 * S) >= 8) S *= 2;
    if (force >= 1) {
        S = 1;
        while (2 * S <= force && 2 * S <= SF_SPLIT_MAX && wgs * 2 * S <= 2 * SF_CHIP_WGS && nk / (2 * S) >= 8) S *= 2;
    }
    return S;
}
// partial-sum tiles: one region for the chain (top) launches, one per slab group
size_t sf_potrf_work_doubles(int n, int batch) {
    const size_t b = (size_t)batch;
    return b * SF_LTB_DOUBLES + b * ((size_t)(n + SF_NB) * SF_LDT + SF_TSKEW) + 2 * b * ((size_t)SF_NB * SF_LDT + SF_TSKEW) + 64 +
           (size_t)(SF_EXEC_GROUPS + 1) * sf_split_region_tiles() * (GT * GT);
}

// ---- two-stream lookahead ---------------------------------------------------------------------
// The diagonal-block chain is a sequence of small latency-bound launches; it runs on the side stream of
// the caller's sf_exec (owned by the context or by the calling thread) concurrently with the big MFMA
// launches of the caller's stream.
#define SF_TRY(x)          \
    do {                   \
        int rc__ = (x);    \
        if (rc__) return rc__; \
    } while (0)

// Factor each n x n matrix in place (lower), panels of SF_NB columns:
//   Ur  T[below] <- C[k1:, k0:k1] - L[k1:, :k0] L[k0:k1, :k0]^T   LEFT-looking for everything below the
//                                                                diagonal block: C read once, long K
//   R   C[jj] -= L[j-rows, k0:k1] L[j-rows, k0:k1]^T for the future DIAGONAL blocks j > k (RIGHT-looking,
//       K = SF_NB): keeps the next diagonal block ready without a long-K launch of only a few tiles;
//       its diagonal tiles also apply rhs[j-rows] -= L[j-rows, k0:k1] z[k0:k1]
//   D   factor the diagonal block together with an identity block -> L_kk and W = L_kk^-T
//       (64-column leaf steps on the small (2 pw) x pw problem), L_kk -> matrix, W^T (F)
//   G   C[k1:, k0:k1] <- T[below] W                               MFMA (triangular B)
// With rhs != NULL (batch x ldr) the forward substitution L z = rhs is fused (R and D); z overwrites rhs.
//
// Lookahead (two streams): only the rows of the NEXT diagonal block are on the critical chain.
//   side:  D(k) F(k) | wait Ur(k) | Gt(k) Rnext(k -> k+1) | D(k+1) ...
//   main:  wait Gt(k-1) | Ur(k) | wait F(k) | Gr(k) Rrest(k) | ...
static int sf_launch_potrf_v1(double* A, int n, int lda, int64_t stride, int batch, int* info, double* work,
                              double* rhs, int ldr, hipStream_t s, const sf_gen_args* gen, sf_exec* ex) {
    if (n % SF_LEAF != 0 || lda < n || batch <= 0 || (lda & 1) || !work) {
        sf_set_error("potrf: n must be a positive multiple of %d, lda >= n and even, workspace required", SF_LEAF);
        return SF_EINVAL;
    }
    double* ltbuf = work;
    double* T = ltbuf + (size_t)batch * SF_LTB_DOUBLES;
    const int64_t sT = (int64_t)(n + SF_NB) * SF_LDT + SF_TSKEW;
    double* Wt2 = T + (size_t)batch * sT;  // two W^T buffers, alternating by panel parity
    const int64_t sW = (int64_t)SF_NB * SF_LDT + SF_TSKEW;
    SF_HIP(hipMemsetAsync(info, 0, sizeof(int) * (size_t)batch, s));

    SF_TRY(sf_exec_prepare(ex));
    hipStream_t c = ex->side;  // side ("critical chain") stream
    auto next_event = [&](hipEvent_t* e) { return sf_exec_event(ex, e); };
    static const bool no_lookahead = SF_TUNE_FLAG("SF_NO_LOOKAHEAD");  // tuning aid: single stream
    static const int rlazy = SF_TUNE_INT("SF_RLAZY", 1);  // tuning aid; measured: no gain for 2, 4, 8
    if (no_lookahead) c = s;
    hipEvent_t e_fork, e_gt_prev = nullptr;
    SF_TRY(next_event(&e_fork));
    SF_HIP(hipEventRecord(e_fork, s));
    SF_HIP(hipStreamWaitEvent(c, e_fork, 0));

    // R: right-looking update of `nblk` future diagonal blocks starting at row/col j0 with panel [k0,k1)
    auto launch_r = [&](int j0, int nrows, int k0, int pw, double* cout, int ldcout, int64_t scout,
                        int64_t dcout, hipStream_t st) -> int {
        sf_gemm_args g = {};
        g.A = g.B = A + (int64_t)j0 * lda + k0;
        g.Cin = A + (int64_t)j0 * lda + j0;
        g.Cout = cout;
        g.sA = g.sB = g.sCin = stride;
        g.sCout = scout;
        g.lda = g.ldb = g.ldcin = lda;
        g.ldcout = ldcout;
        g.M = g.Nc = nrows;
        g.K = pw;
        g.remap_after = 0x7fffffff;
        g.diag_blocks = (nrows + SF_NB - 1) / SF_NB;
        g.dA = (int64_t)SF_NB * lda;
        g.dC = dcout;
        if (rhs && pw > 0) {
            g.rhs = rhs + j0;
            g.z = rhs + k0;
            g.srhs = g.sz = ldr;
        }
        // algorithmic flops: lower triangle of every block
        double useful = 0.0;
        for (int r = 0; r < nrows; r += SF_NB) {
            const double bw = (nrows - r < SF_NB) ? nrows - r : SF_NB;
            useful += 0.5 * bw * (bw + 1);
        }
        return launch_gemm(g, batch, true, 2.0 * pw * useful * batch, st);
    };

    // diagonal block 0 goes to the panel scratch unchanged (K = 0: a copy)
    {
        const int pw0 = n < SF_NB ? n : SF_NB;
        SF_TRY(launch_r(0, pw0, 0, 0, T, SF_LDT, sT, 0, c));
    }
    int panel = 0;
    for (int k0 = 0; k0 < n; k0 += SF_NB, ++panel) {
        const int k1 = (k0 + SF_NB < n) ? k0 + SF_NB : n;
        const int pw = k1 - k0;
        const int nbelow = n - k1;
        const int ntop = nbelow < SF_NB ? nbelow : SF_NB;  // rows of the next diagonal block
        double* Wt = Wt2 + (size_t)(panel & 1) * batch * sW;
        hipEvent_t e_ur = nullptr, e_f, e_gt;
        // ---- Ur on the main stream: rows [k1, n) -> T rows [2pw, ...)
        if (nbelow > 0) {
            if (e_gt_prev) SF_HIP(hipStreamWaitEvent(s, e_gt_prev, 0));
            sf_gemm_args g = {};
            g.A = A + (int64_t)k1 * lda;
            g.B = A + (int64_t)k0 * lda;
            g.Cin = A + (int64_t)k1 * lda + k0;
            g.Cout = T + (int64_t)(2 * pw) * SF_LDT;
            g.sA = g.sB = g.sCin = stride;
            g.sCout = sT;
            g.lda = g.ldb = g.ldcin = lda;
            g.ldcout = SF_LDT;
            g.M = nbelow;
            g.Nc = pw;
            g.K = k0;
            g.remap_after = 0x7fffffff;
            if (gen) {
                g.genY = gen->Y;
                g.sY = (int64_t)gen->mpad * gen->ldy;
                g.ldy = gen->ldy;
                g.mpad = gen->mpad;
                g.tilemap = gen->tilemap;
                g.nt128 = gen->nt128;
                g.tm_off = k1 / GT;
                g.tn_off = k0 / GT;
            }
            SF_TRY(launch_gemm(g, batch, true, 2.0 * k0 * (double)nbelow * pw * batch, s));
            SF_TRY(next_event(&e_ur));
            SF_HIP(hipEventRecord(e_ur, s));
        }
        // ---- D + F on the side stream (T rows [0, pw) already hold the fully updated diagonal block)
        static const bool leaf_diag = SF_TUNE_FLAG("SF_LEAF_DIAG");  // tuning aid: the 13-launch chain
        if (!leaf_diag) {
            hipLaunchKernelGGL(k_diag_mfma<1024>, dim3(batch), dim3(1024), 0, c, T, sT, pw, info, k0,
                               rhs ? rhs + k0 : nullptr, ldr, A + (int64_t)k0 * lda + k0, lda, stride, Wt, sW);
            SF_LAUNCH_CHECK();
        } else {
            hipLaunchKernelGGL(k_set_identity, dim3(64, batch), dim3(256), 0, c, T, sT, pw);
            SF_LAUNCH_CHECK();
            for (int cc = 0; cc < pw; cc += SF_LEAF) {
                hipLaunchKernelGGL(k_potrf_leaf, dim3(batch), dim3(64), 0, c, T, SF_LDT, sT, cc, info, k0, ltbuf,
                                   rhs ? rhs + k0 : nullptr, ldr);
                SF_LAUNCH_CHECK();
                const int below = 2 * pw - (cc + SF_LEAF);
                hipLaunchKernelGGL(k_trsm_leaf, dim3((below + 63) / 64, batch), dim3(64), 0, c, T, SF_LDT, sT,
                                   cc, 2 * pw, (const double*)ltbuf, rhs ? rhs + k0 : nullptr, ldr, pw);
                SF_LAUNCH_CHECK();
                if (cc + SF_LEAF < pw) {
                    sf_gemm_args g = {};
                    const int o = cc + SF_LEAF;
                    g.A = g.B = T + (int64_t)o * SF_LDT + cc;
                    g.Cin = g.Cout = T + (int64_t)o * SF_LDT + o;
                    g.sA = g.sB = g.sCin = g.sCout = sT;
                    g.lda = g.ldb = g.ldcin = g.ldcout = SF_LDT;
                    g.M = below;
                    g.Nc = pw - o;
                    g.K = SF_LEAF;
                    g.tri = 1;
                    g.remap_after = 0x7fffffff;
                    const double useful = (double)g.M * g.Nc - 0.5 * (double)g.Nc * (g.Nc - 1);
                    SF_TRY(launch_gemm(g, batch, true, 2.0 * SF_LEAF * useful * batch, c));
                }
            }
            hipLaunchKernelGGL(k_panel_finish, dim3((pw + 31) / 32, (pw + 31) / 32, batch), dim3(256), 0, c,
                               (const double*)T, sT, pw, A + (int64_t)k0 * lda + k0, lda, stride, Wt, sW);
            SF_LAUNCH_CHECK();
        }
        if (nbelow <= 0) break;
        SF_TRY(next_event(&e_f));
        SF_HIP(hipEventRecord(e_f, c));
        // ---- G: T[below] W.  Top rows (next diagonal block) on the side stream, the rest on main.
        auto launch_g = [&](int row_lo, int nrows, hipStream_t st) -> int {
            sf_gemm_args g = {};
            g.A = T + (int64_t)(2 * pw + row_lo) * SF_LDT;
            g.B = Wt;
            g.Cout = A + (int64_t)(k1 + row_lo) * lda + k0;
            g.sA = sT;
            g.sB = sW;
            g.sCout = stride;
            g.lda = g.ldb = SF_LDT;
            g.ldcout = lda;
            g.M = nrows;
            g.Nc = pw;
            g.K = pw;
            g.btri = 1;
            g.remap_after = 0x7fffffff;
            return launch_gemm(g, batch, false, (double)nrows * pw * pw * batch, st);
        };
        SF_HIP(hipStreamWaitEvent(c, e_ur, 0));
        SF_TRY(launch_g(0, ntop, c));
        SF_TRY(next_event(&e_gt));
        SF_HIP(hipEventRecord(e_gt, c));
        e_gt_prev = e_gt;
        // next diagonal block: apply this panel's columns and park it in the panel scratch
        // (the diagonal blocks after the next one are brought up to date only every `rlazy` panels, with a
        // correspondingly longer K: their read-modify-write traffic is what bounds those launches; the next
        // block therefore may still miss the last few panels)
        const int pend0 = (panel / rlazy) * rlazy * SF_NB;  // first panel column not yet applied to block k+1
        SF_TRY(launch_r(k1, ntop, pend0, k1 - pend0, T, SF_LDT, sT, 0, c));
        if (nbelow > ntop) {
            SF_HIP(hipStreamWaitEvent(s, e_f, 0));
            SF_TRY(launch_g(ntop, nbelow - ntop, s));
            // the diagonal blocks after the next one are updated in place, every `rlazy` panels
            const int j0 = k1 + ntop;
            if ((panel + 1) % rlazy == 0 && j0 < n)
                SF_TRY(launch_r(j0, n - j0, pend0, k1 - pend0, A + (int64_t)j0 * lda + j0, lda, stride,
                                (int64_t)SF_NB * lda + SF_NB, s));
        }
    }
    // join: the caller's stream continues only after the side chain is done
    hipEvent_t e_join;
    SF_TRY(next_event(&e_join));
    SF_HIP(hipEventRecord(e_join, c));
    SF_HIP(hipStreamWaitEvent(s, e_join, 0));
    return SF_OK;
}

// The fused sequences work in the frame of sf_potrf_front_pad: from here on A / rhs point fp (lda + 1) / fp elements before
// the data and n counts the fp virtual leading rows too (the scratch layout above is sized with the real n).
#define SF_SHIFT_FRAME()                                                              \
    do {                                                                              \
        if (fp != 0 && (fp != 64 || n % GT != 64)) {                                  \
            sf_set_error("potrf: front pad %d does not fit n = %d", fp, n);           \
            return SF_EINVAL;                                                         \
        }                                                                             \
        A -= (int64_t)fp * (lda + 1);                                                 \
        if (rhs) rhs -= fp;                                                           \
        n += fp;                                                                      \
    } while (0)
static std::atomic<int> g_chol_sequence{-1};
int sf_set_cholesky_sequence(int mode) {
    if (mode < -1 || mode > 4) {
        sf_set_error("cholesky sequence: -1 automatic, 0 fused panel kernel, 1 unfused, 2 wide (panel pairs), 3 wide then narrow (test aid), 4 dataflow");
        return SF_EINVAL;
    }
    g_chol_sequence.store(mode);
    return SF_OK;
}

// Factorisation with the fused panel kernel (default).  Panels of 128 columns; per panel k
//   D(k)      k_diag_mfma on the updated diagonal tile (parked in the scratch T): L_kk, L_kk^-1, z_k
//   top(k)    k_chol_panel for the slab of the NEXT diagonal tile (rows k1 .. k1+128): its updated tile goes to T
//   rest(k)   k_chol_panel for all slabs below, as G launches on G streams: slab i belongs to group i mod G
// Lookahead: the chain  D(k) -> [wait group of slab k+1] top(k) -> D(k+1) ...  runs on the side stream;
// group g only needs D(k) (which ran beside rest(k-1)) and its own previous launch (a slab stays in its
// group), so there is no chip-wide barrier between panels: while one group's launch drains its last
// workgroups the other groups keep the CUs full (one launch per panel left 0.25-0.75 of a round of 512
// workgroups idle at every panel boundary).
static int sf_launch_potrf_v2(double* A, int n, int lda, int64_t stride, int batch, int* info, double* work,
                              double* rhs, int ldr, hipStream_t s, const sf_gen_args* gen, sf_exec* ex, int fp) {
    if (n % SF_LEAF != 0 || lda < n || batch <= 0 || (lda & 1) || !work) {
        sf_set_error("potrf: n must be a positive multiple of %d, lda >= n and even, workspace required", SF_LEAF);
        return SF_EINVAL;
    }
    double* T = work + (size_t)batch * SF_LTB_DOUBLES;
    const int64_t sT = (int64_t)(n + SF_NB) * SF_LDT + SF_TSKEW;  // (layout shared with the unfused path)
    double* Wt2 = T + (size_t)batch * sT;
    const int64_t sW = (int64_t)SF_NB * SF_LDT + SF_TSKEW;
    SF_SHIFT_FRAME();
    SF_HIP(hipMemsetAsync(info, 0, sizeof(int) * (size_t)batch, s));
    SF_TRY(sf_exec_prepare(ex));
    static const bool no_lookahead = SF_TUNE_FLAG("SF_NO_LOOKAHEAD");
    static const int ngroups_env = SF_TUNE_INT("SF_CHOL_GROUPS", 2);
    const int G = no_lookahead ? 1 : (ngroups_env < 1 ? 1 : (ngroups_env > SF_EXEC_GROUPS ? SF_EXEC_GROUPS : ngroups_env));
    hipStream_t c = no_lookahead ? s : ex->side;
    hipStream_t gs[SF_EXEC_GROUPS];
    for (int g = 0; g < G; ++g) gs[g] = g == 0 ? s : ex->grp[g - 1];
    hipEvent_t e_fork;
    SF_TRY(sf_exec_event(ex, &e_fork));
    SF_HIP(hipEventRecord(e_fork, s));
    if (c != s) SF_HIP(hipStreamWaitEvent(c, e_fork, 0));
    for (int g = 0; g < G; ++g)
        if (gs[g] != s) SF_HIP(hipStreamWaitEvent(gs[g], e_fork, 0));

    double* part = Wt2 + 2 * (size_t)batch * sW + 64;  // split-K partial sums: region 0 = chain, 1 + g = group g
    const int nt = (n + GT - 1) / GT;
    // phase 0: the whole step; 1 / 2: only the split-K partial sums / only what follows them (the chain runs the partial
    // sums of top(k), which do not need D(k), beside D(k) on another stream); split_of() tells whether the step is split
    auto split_of = [&](int k0, int pw, long long nblk) { return pw > 0 ? sf_split_policy(nblk, (k0 > fp ? k0 - fp : 0) / GK) : 1; };
    auto launch_panel = [&](int k0, int pw, int row0, int nslab, int step, const double* Wt, bool to_scratch,
                            hipStream_t st, int region, int phase) -> int {
        sf_panel_args g = {};
        g.C = A;
        g.sC = stride;
        g.lda = lda;
        g.n = n;
        g.k0 = k0;
        g.pw = pw;
        g.row0 = row0;
        g.nslab = nslab;
        g.slab_step = step;
#ifdef SF_TUNING
        static const int skip = SF_TUNE_INT("SF_PANEL_SKIP", 0);
        g.skip = skip;
#endif
        g.Wt = Wt;
        g.sW = sW;
        g.rhs = rhs;
        g.ldr = ldr;
        if (to_scratch) {
            g.Sout = T;
            g.sS = sT;
            g.ldS = SF_LDT;
            g.prio = chain_prio;
        }
        g.fp = fp;
        if (gen) {
            g.genY = gen->Y - fp;
            g.sY = (int64_t)gen->mpad * gen->ldy;
            g.ldy = gen->ldy;
            g.mpad = gen->mpad;
            g.tilemap = gen->tilemap;
            g.nt128 = gen->nt128;
        }
        const long long nblk = (long long)nslab * batch;
        if (nblk > 0x7fffffffLL) {
            sf_set_error("panel grid too large");
            return SF_EINVAL;
        }
        // algorithmic flops: update 2 k0 rows pw, solve rows pw^2, symmetric rank-pw update of the lower tiles
        double rows = 0.0;
        for (int i = 0; i < nslab; ++i) {
            const int r0 = row0 + i * step * GT;
            rows += (n - r0 < GT) ? n - r0 : GT;
        }
        const double flops_main = 2.0 * (k0 > fp ? k0 - fp : 0) * rows * pw * batch;
        const double flops_epi = (rows * pw * (double)pw + (double)GT * rows * pw) * batch;
        const int nk = (k0 > fp ? k0 - fp : 0) / GK;
        const int S = split_of(k0, pw, nblk);
        void* tok;  // (every kernel launch is one profiled launch: what rocprofv3 --stats counts)
        if (S > 1) {
            g.ksplit = S;
            g.kchunk = (nk + S - 1) / S;
            g.part = part + (size_t)region * sf_split_region_tiles() * (GT * GT);
            if (phase != 2) {
                sf_prof_gemm_begin(st, flops_main, &tok);
                hipLaunchKernelGGL((k_chol_panel<false, 1>), dim3((unsigned)(nblk * S)), dim3(512), 0, st, g);
                sf_prof_gemm_end(tok);
            }
            if (phase != 1) {
                sf_prof_gemm_begin(st, flops_epi, &tok);
                if (rhs)
                    hipLaunchKernelGGL((k_chol_panel<true, 2>), dim3((unsigned)nblk), dim3(512), 0, st, g);
                else
                    hipLaunchKernelGGL((k_chol_panel<false, 2>), dim3((unsigned)nblk), dim3(512), 0, st, g);
                sf_prof_gemm_end(tok);
            }
        } else {
            sf_prof_gemm_begin(st, flops_main + flops_epi, &tok);
            if (rhs)
                hipLaunchKernelGGL((k_chol_panel<true, 0>), dim3((unsigned)nblk), dim3(512), 0, st, g);
            else
                hipLaunchKernelGGL((k_chol_panel<false, 0>), dim3((unsigned)nblk), dim3(512), 0, st, g);
            sf_prof_gemm_end(tok);
        }
        SF_LAUNCH_CHECK();
        return SF_OK;
    };

    // diagonal tile 0 goes to the scratch unchanged
    SF_TRY(launch_panel(0, 0, 0, 1, 1, nullptr, true, c, 0, 0));
    static const bool part_on_chain = SF_TUNE_FLAG("SF_PART_ON_CHAIN");  // tuning aid: the partial sums of top(k) after D(k) on the chain
    hipEvent_t e_epi = nullptr;                   // end of top(k-1) on the chain
    hipEvent_t e_rest[SF_EXEC_GROUPS] = {};       // last launch of every group
    hipEvent_t e_rest_prev[SF_EXEC_GROUPS] = {};  // ... one panel earlier (their readers of Wt[panel & 1])
    for (int k = 0; k < nt; ++k) {
        const int k0 = k * GT;
        const int pw = (n - k0 < GT) ? n - k0 : GT;
        double* Wt = Wt2 + (size_t)(k & 1) * batch * sW;
        // D(k) overwrites the W buffer of panel k-2: every group must be done reading it
        if (c != s || G > 1)
            for (int g = 0; g < G; ++g)
                if (e_rest_prev[g] && (c != gs[g])) SF_HIP(hipStreamWaitEvent(c, e_rest_prev[g], 0));
        SF_TRY(sf_launch_diag128(T, sT, pw, info, k0 - fp, rhs ? rhs + k0 : nullptr, ldr, A + (int64_t)k0 * lda + k0, lda, stride, Wt, sW,
                                 batch, c, k == 0 ? fp : 0));
        if (k + 1 >= nt) break;
        hipEvent_t e_d;
        SF_TRY(sf_exec_event(ex, &e_d));
        SF_HIP(hipEventRecord(e_d, c));
        // top(k): the slab of the next diagonal tile, on the chain; its row was finished by the group of slab k+1.
        // Its long-K part (split-K partial sums) needs the rows of slabs k and k+1 left of the panel, not D(k): when the
        // step is split it runs BESIDE D(k), on the stream of the group of slab k+1 (whose last launch it waits for
        // anyway) -- the chain is D(k) | partial sums -> reduce + solve + diagonal tile -> D(k+1).
        {
            hipStream_t gk = gs[(k + 1) % G];
            hipEvent_t dep = e_rest[(k + 1) % G];
            if (c != gk && !part_on_chain && split_of(k0, pw, batch) > 1) {
                if (e_epi) SF_HIP(hipStreamWaitEvent(gk, e_epi, 0));  // row k's columns of panel k-1; the partial-sum region
                SF_TRY(launch_panel(k0, pw, (k + 1) * GT, 1, 1, Wt, true, gk, 0, 1));
                hipEvent_t e_part;
                SF_TRY(sf_exec_event(ex, &e_part));
                SF_HIP(hipEventRecord(e_part, gk));
                SF_HIP(hipStreamWaitEvent(c, e_part, 0));
                SF_TRY(launch_panel(k0, pw, (k + 1) * GT, 1, 1, Wt, true, c, 0, 2));
            } else {
                if (dep && c != gk) SF_HIP(hipStreamWaitEvent(c, dep, 0));
                SF_TRY(launch_panel(k0, pw, (k + 1) * GT, 1, 1, Wt, true, c, 0, 0));
            }
            if (c != s) {
                SF_TRY(sf_exec_event(ex, &e_epi));
                SF_HIP(hipEventRecord(e_epi, c));
            }
        }
        // rest(k): slabs k+2 .. nt-1, slab i on the stream of group i mod G
        for (int g = 0; g < G; ++g) e_rest_prev[g] = e_rest[g];
        for (int g = 0; g < G; ++g) {
            int first = k + 2;
            while (first % G != g) ++first;
            if (first >= nt) continue;
            const int cnt = (nt - 1 - first) / G + 1;
            if (gs[g] != c) SF_HIP(hipStreamWaitEvent(gs[g], e_d, 0));
            SF_TRY(launch_panel(k0, pw, first * GT, cnt, G, Wt, false, gs[g], 1 + g, 0));
            SF_TRY(sf_exec_event(ex, &e_rest[g]));
            SF_HIP(hipEventRecord(e_rest[g], gs[g]));
        }
    }
    // join: the caller's stream continues only after the chain and every group are done
    hipEvent_t e_join;
    if (c != s) {
        SF_TRY(sf_exec_event(ex, &e_join));
        SF_HIP(hipEventRecord(e_join, c));
        SF_HIP(hipStreamWaitEvent(s, e_join, 0));
    }
    for (int g = 0; g < G; ++g)
        if (e_rest[g] && gs[g] != s) SF_HIP(hipStreamWaitEvent(s, e_rest[g], 0));
    return SF_OK;
}

// Factorisation with the WIDE panel kernel: pairs of panels.  Per pair p (panels k = 2p, k + 1; columns [k0, k0 + 256)):
//   chain(p)  on the side stream:  D(k) -> top(k): narrow k_chol_panel for slab k+1 (gives L21, parks tile (k+1, k+1))
//             -> D(k+1);  depends on A(p-1) only
//   A(p)      k_chol_panel_w for the slabs k+2, k+3 (the rows of the NEXT pair's diagonal block; parks tile (k+2, k+2)):
//             one round of workgroups on its own stream, so that chain(p+1) runs beside B(p)
//   B(p)      k_chol_panel_w for the slabs k+4 .. on the caller's stream
// A trailing single panel (odd number of panels) and pairs without rows below them are narrow steps of the chain.
// The four most recent inverse tiles W(k) live in the two 256-row buffers of the narrow sequence (slot k & 3).
// tail_rounds: the pairs whose wide launches have at most this many rounds of workgroups left (and everything after them)
// are single narrow steps; -1 = wide to the end; -2 = switch half-way (test aid: exercises the hand-over on any size).
static int sf_launch_potrf_v3(double* A, int n, int lda, int64_t stride, int batch, int* info, double* work,
                              double* rhs, int ldr, hipStream_t s, const sf_gen_args* gen, sf_exec* ex, int tail_rounds, int fp) {
    if (n % SF_LEAF != 0 || lda < n || batch <= 0 || (lda & 1) || !work) {
        sf_set_error("potrf: n must be a positive multiple of %d, lda >= n and even, workspace required", SF_LEAF);
        return SF_EINVAL;
    }
    static sf_dev_once attr_once;
    SF_CHECK(sf_once_per_device(&attr_once, []() -> int {
        SF_HIP(hipFuncSetAttribute((const void*)k_chol_panel_w<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SF_HIP(hipFuncSetAttribute((const void*)k_chol_panel_w<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        return SF_OK;
    }));
    double* T = work + (size_t)batch * SF_LTB_DOUBLES;
    const int64_t sT = (int64_t)(n + SF_NB) * SF_LDT + SF_TSKEW;
    double* Wt2 = T + (size_t)batch * sT;
    const int64_t sW = (int64_t)SF_NB * SF_LDT + SF_TSKEW;
    auto Wslot = [&](int k) { return Wt2 + (size_t)((k >> 1) & 1) * batch * sW + (size_t)(k & 1) * GT * SF_LDT; };
    double* part = Wt2 + 2 * (size_t)batch * sW + 64;
    SF_SHIFT_FRAME();
    SF_HIP(hipMemsetAsync(info, 0, sizeof(int) * (size_t)batch, s));
    SF_TRY(sf_exec_prepare(ex));
    // A(p) sits between chain(p) and chain(p+1) anyway: it shares the chain's stream.  A stream of its own made a cfg-2
    // step 3 % slower (48.4 -> 50.0 ms: every additional ACTIVE stream costs dispatch latency on all of them -- the
    // transform chain ahead of the factorisation went from 0.40 to 0.70 ms); at cfg 3, where an A launch is ten rounds of
    // workgroups, a separate stream measured the same (265.1 / 267.0 vs 266.8 / 265.8 ms).
    hipStream_t c = ex->side, xa = ex->side;
    hipEvent_t e_fork;
    SF_TRY(sf_exec_event(ex, &e_fork));
    SF_HIP(hipEventRecord(e_fork, s));
    SF_HIP(hipStreamWaitEvent(c, e_fork, 0));
    SF_HIP(hipStreamWaitEvent(xa, e_fork, 0));
    SF_HIP(hipStreamWaitEvent(ex->grp[0], e_fork, 0));
    const int nt = (n + GT - 1) / GT;

    auto narrow = [&](int k0, int pw, int row0, int nslab, int step, const double* Wt, bool to_scratch, hipStream_t st,
                      int region) -> int {
        sf_panel_args g = {};
        g.C = A;
        g.sC = stride;
        g.lda = lda;
        g.n = n;
        g.k0 = k0;
        g.pw = pw;
        g.row0 = row0;
        g.nslab = nslab;
        g.slab_step = step;
        g.Wt = Wt;
        g.sW = sW;
        g.rhs = rhs;
        g.ldr = ldr;
        if (to_scratch) {
            g.Sout = T;
            g.sS = sT;
            g.ldS = SF_LDT;
            g.prio = chain_prio;
        }
        g.fp = fp;
        if (gen) {
            g.genY = gen->Y - fp;
            g.sY = (int64_t)gen->mpad * gen->ldy;
            g.ldy = gen->ldy;
            g.mpad = gen->mpad;
            g.tilemap = gen->tilemap;
            g.nt128 = gen->nt128;
        }
        const long long nblk = (long long)nslab * batch;
        double rows = 0.0;
        for (int i = 0; i < nslab; ++i) rows += (n - (row0 + i * step * GT) < GT) ? n - (row0 + i * step * GT) : GT;
        const double flops_main = 2.0 * (k0 > fp ? k0 - fp : 0) * rows * pw * batch;
        const double flops_epi = (rows * pw * (double)pw + (double)GT * rows * pw) * batch;
        const int nk = (k0 > fp ? k0 - fp : 0) / GK;
        const int S = pw > 0 ? sf_split_policy(nblk, nk) : 1;
        void* tok;
        if (S > 1) {
            g.ksplit = S;
            g.kchunk = (nk + S - 1) / S;
            g.part = part + (size_t)region * sf_split_region_tiles() * (GT * GT);
            sf_prof_gemm_begin(st, flops_main, &tok);
            hipLaunchKernelGGL((k_chol_panel<false, 1>), dim3((unsigned)(nblk * S)), dim3(512), 0, st, g);
            sf_prof_gemm_end(tok);
            sf_prof_gemm_begin(st, flops_epi, &tok);
            if (rhs)
                hipLaunchKernelGGL((k_chol_panel<true, 2>), dim3((unsigned)nblk), dim3(512), 0, st, g);
            else
                hipLaunchKernelGGL((k_chol_panel<false, 2>), dim3((unsigned)nblk), dim3(512), 0, st, g);
            sf_prof_gemm_end(tok);
        } else {
            sf_prof_gemm_begin(st, flops_main + flops_epi, &tok);
            if (rhs)
                hipLaunchKernelGGL((k_chol_panel<true, 0>), dim3((unsigned)nblk), dim3(512), 0, st, g);
            else
                hipLaunchKernelGGL((k_chol_panel<false, 0>), dim3((unsigned)nblk), dim3(512), 0, st, g);
            sf_prof_gemm_end(tok);
        }
        SF_LAUNCH_CHECK();
        return SF_OK;
    };
#ifdef SF_TUNING
    long long* wstamps = nullptr;
    int wstamp_n = 0, wstamp_k[64];
    if (SF_TUNE_FLAG("SF_WIDE_STAMPS")) {
        SF_HIP(hipHostMalloc((void**)&wstamps, sizeof(long long) * 16 * 64));
        for (int i = 0; i < 16 * 64; ++i) wstamps[i] = 0;
    }
#endif
    auto wide = [&](int k, int slab0, int nslab, int step, bool park, hipStream_t st) -> int {
        sf_panelw_args g = {};
        g.C = A;
        g.sC = stride;
        g.lda = lda;
        g.n = n;
        g.k0 = k * GT;
        g.row0 = slab0 * GT;
        g.nslab = nslab;
        g.slab_step = step;
        g.Wt0 = Wslot(k);
        g.Wt1 = Wslot(k + 1);
        g.sW = sW;
        g.rhs = rhs;
        g.ldr = ldr;
        if (park) {
            g.Sout = T;
            g.sS = sT;
            g.ldS = SF_LDT;
        }
        g.fp = fp;
        if (gen) {
            g.genY = gen->Y - fp;
            g.sY = (int64_t)gen->mpad * gen->ldy;
            g.ldy = gen->ldy;
            g.mpad = gen->mpad;
            g.tilemap = gen->tilemap;
            g.nt128 = gen->nt128;
        }
        const long long nblk = (long long)nslab * batch;
        if (nblk > 0x7fffffffLL) {
            sf_set_error("panel grid too large");
            return SF_EINVAL;
        }
#ifdef SF_TUNING
        if (wstamps && wstamp_n < 64) {
            wstamp_k[wstamp_n] = k * 1000 + nslab;
            g.stamps = wstamps + 16 * wstamp_n++;
        }
#endif
        double rows = 0.0;
        for (int i = 0; i < nslab; ++i) rows += (n - (slab0 + i * step) * GT < GT) ? n - (slab0 + i * step) * GT : GT;
        // algorithmic flops of the two panel steps it replaces: update 2 k0 rows 128 (+ 128 more K for the second panel),
        // solves rows 128^2 each, symmetric rank-128 updates of the lower tiles
        const double kk = g.k0 > fp ? g.k0 - fp : 0;
        const double flops = (2.0 * kk * rows * GT + 2.0 * (kk + GT) * rows * GT + 2.0 * (rows * GT * (double)GT + (double)GT * rows * GT)) * batch;
        void* tok;
        sf_prof_gemm_begin(st, flops, &tok);
        if (rhs)
            hipLaunchKernelGGL(k_chol_panel_w<true>, dim3((unsigned)nblk), dim3(1024), SF_PANELW_LDS, st, g);
        else
            hipLaunchKernelGGL(k_chol_panel_w<false>, dim3((unsigned)nblk), dim3(1024), SF_PANELW_LDS, st, g);
        sf_prof_gemm_end(tok);
        SF_LAUNCH_CHECK();
        return SF_OK;
    };
    auto diag = [&](int k) -> int {
        const int k0 = k * GT;
        const int pw = (n - k0 < GT) ? n - k0 : GT;
        return sf_launch_diag128(T, sT, pw, info, k0 - fp, rhs ? rhs + k0 : nullptr, ldr, A + (int64_t)k0 * lda + k0, lda, stride, Wslot(k), sW,
                                 batch, c, k == 0 ? fp : 0);
    };

    SF_TRY(narrow(0, 0, 0, 1, 1, nullptr, true, c, 0));  // diagonal tile 0 goes to the scratch unchanged
    // B(p) runs as two interleaved slab groups on two streams (like the narrow sequence): a group's next launch only
    // needs its own previous one, so the last, partly filled round of one group overlaps the other group's work.
    // Group g = slabs of parity g (k even: k+4+g, k+6+g, ...), in the wide pairs and in the narrow tail alike.
    hipStream_t bs[2] = {s, ex->grp[0]};
    hipEvent_t e_A = nullptr;
    hipEvent_t e_last[2] = {nullptr, nullptr};        // last launch of either group
    std::vector<hipEvent_t> readers[4];               // launches that read W slot j (a D step may only overwrite it after them)
    auto wait_readers = [&](int slot) -> int {
        for (hipEvent_t e : readers[slot]) SF_HIP(hipStreamWaitEvent(c, e, 0));
        readers[slot].clear();
        return SF_OK;
    };
    // The narrow loop below finishes what the pairs leave (a trailing single panel, the last diagonal block) and can
    // take over earlier (`tail_rounds`): the timeline suggested that the last pairs -- few rounds of ~1 ms workgroups,
    // every dependency of the chain costs a round -- would be better off as narrow steps, the measurement says no
    // (cfg 2: wide to the end 49.3 ms, hand-over with 2 / 5 / 8 / 12 rounds left 50.0 / 50.4 / 51.0 / 51.8, narrow 51.5).
    // One narrow step of the chain + both slab groups: panel k as D(k), top(k), rest(k) -- the steps before the first pair
    // (`head`) and after the last one.
    auto narrow_step = [&](int k) -> int {
        const int k0 = k * GT;
        const int pw = (n - k0 < GT) ? n - k0 : GT;
        if (e_A) {  // the tile parked by the last wide A launch, and the rows of its two slabs
            SF_HIP(hipStreamWaitEvent(c, e_A, 0));
            for (int g = 0; g < 2; ++g) SF_HIP(hipStreamWaitEvent(bs[g], e_A, 0));
            e_A = nullptr;
        }
        SF_TRY(wait_readers(k & 3));
        SF_TRY(diag(k));
        if (k + 1 >= nt) return SF_OK;
        hipEvent_t e_d;
        SF_TRY(sf_exec_event(ex, &e_d));
        SF_HIP(hipEventRecord(e_d, c));
        // top(k): the slab of the next diagonal tile, on the chain; its rows were finished by the group of its parity
        if (e_last[(k + 1) & 1]) SF_HIP(hipStreamWaitEvent(c, e_last[(k + 1) & 1], 0));
        SF_TRY(narrow(k0, pw, (k + 1) * GT, 1, 1, Wslot(k), true, c, 0));
        for (int g = 0; g < 2; ++g) {
            int first = k + 2;
            if ((first & 1) != g) ++first;
            if (first >= nt) continue;
            const int cnt = (nt - 1 - first) / 2 + 1;
            SF_HIP(hipStreamWaitEvent(bs[g], e_d, 0));
            SF_TRY(narrow(k0, pw, first * GT, cnt, 2, Wslot(k), false, bs[g], 1 + g));
            SF_TRY(sf_exec_event(ex, &e_last[g]));
            SF_HIP(hipEventRecord(e_last[g], bs[g]));
            readers[k & 3].push_back(e_last[g]);
        }
        return SF_OK;
    };
    // The first pairs have short K loops: a wide workgroup (one per CU) is then mostly its epilogue -- tile in, two
    // triangular solves, tile out, one after the other with nothing beside it on the CU (B = 128: pair 0 runs at 0.44 of
    // the matrix peak, pair 1 at 0.65, pair 2 at 0.70; the pairs from K = 1024 on at 0.81-0.88,
    // profiles/r05_d_wide_per_pair_b128.txt).  The panels left of `head` are narrow steps (two workgroups per CU overlap
    // one's memory phases with the other's solves) -- in tuning builds only: measured without gain for head = 2 ... 12
    // (profiles/r05_d_wide_narrow_head_sweep.txt), the release library always starts with pair 0.
    static const int head_env = SF_TUNE_INT("SF_WIDE_HEAD", 0);
    const int head = std::min(nt, std::max(head_env, 0) & ~1);
    int k = 0;
    for (; k < head; ++k) SF_TRY(narrow_step(k));
    bool handover = head > 0;
    for (; k < nt; k += 2) {
        if (k + 2 >= nt) break;  // no rows below the pair: the narrow loop finishes the diagonal block
        const long long rounds_left = (long long)batch * (nt - (k + 4) > 0 ? nt - (k + 4) : 0) / 256;
        if (tail_rounds >= 0 && rounds_left <= tail_rounds) break;  // -> narrow tail from panel k
        if (tail_rounds == -2 && k >= (nt / 4) * 2 && k > 0) break;
        // chain(p): needs the tile parked by A(p-1) and the rows of slab k+1 (A(p-1))
        if (e_A) SF_HIP(hipStreamWaitEvent(c, e_A, 0));
        if (handover) {  // (after narrow steps: the rows of slab k+1 come from the slab group of its parity)
            for (int g = 0; g < 2; ++g)
                if (e_last[g]) SF_HIP(hipStreamWaitEvent(c, e_last[g], 0));
            handover = false;
        }
        SF_TRY(wait_readers(k & 3));
        SF_TRY(diag(k));
        SF_TRY(narrow(k * GT, GT, (k + 1) * GT, 1, 1, Wslot(k), true, c, 0));  // (rows below the pair exist: panel k is full)
        SF_TRY(wait_readers((k + 1) & 3));
        SF_TRY(diag(k + 1));
        hipEvent_t e_chain;
        SF_TRY(sf_exec_event(ex, &e_chain));
        SF_HIP(hipEventRecord(e_chain, c));
        // A(p): slabs k+2, k+3 -- needs chain(p) and the rows B(p-1) finished (the first slab of either group)
        const int na = (nt - (k + 2) < 2) ? nt - (k + 2) : 2;
        SF_HIP(hipStreamWaitEvent(xa, e_chain, 0));
        for (int g = 0; g < 2; ++g)
            if (e_last[g]) SF_HIP(hipStreamWaitEvent(xa, e_last[g], 0));
        SF_TRY(wide(k, k + 2, na, 1, true, xa));
        SF_TRY(sf_exec_event(ex, &e_A));
        SF_HIP(hipEventRecord(e_A, xa));
        readers[k & 3].push_back(e_A);
        readers[(k + 1) & 3].push_back(e_A);
        // B(p): slabs k+4 .., slab k+4+g, k+6+g, ... in group g
        static const int ngrp = SF_TUNE_INT("SF_WIDE_GROUPS", 2);  // tuning aid: 1 = one launch per pair on the caller's stream
        for (int g = 0; g < ngrp; ++g) {
            const int first = k + 4 + g;
            if (first >= nt) continue;
            const int cnt = (nt - 1 - first) / ngrp + 1;
            SF_HIP(hipStreamWaitEvent(bs[g], e_chain, 0));
            SF_TRY(wide(k, first, cnt, ngrp, false, bs[g]));
            SF_TRY(sf_exec_event(ex, &e_last[g]));
            SF_HIP(hipEventRecord(e_last[g], bs[g]));
            readers[k & 3].push_back(e_last[g]);
            readers[(k + 1) & 3].push_back(e_last[g]);
        }
    }
    // narrow tail (also: a trailing single panel, pairs without rows below them)
    for (; k < nt; ++k) SF_TRY(narrow_step(k));
    hipEvent_t e_join;
    SF_TRY(sf_exec_event(ex, &e_join));
    SF_HIP(hipEventRecord(e_join, c));
    SF_HIP(hipStreamWaitEvent(s, e_join, 0));
    if (e_A) SF_HIP(hipStreamWaitEvent(s, e_A, 0));
    if (e_last[1]) SF_HIP(hipStreamWaitEvent(s, e_last[1], 0));
#ifdef SF_TUNING
    if (wstamps) {  // (synchronises: phases of one workgroup per wide launch, us)
        (void)hipStreamSynchronize(s);
        fprintf(stderr, "wide launches, workgroup grid/2: k nslab | prologue | K loop | solve 1 | 2b | solve 2 | store + rhs | S load + step 4 | S store + drain | total (us)\n");
        for (int i = 0; i < wstamp_n; ++i) {
            const long long* t = wstamps + 16 * i;
            fprintf(stderr, "%2d %2d |", wstamp_k[i] / 1000, wstamp_k[i] % 1000);
            const int seg[8][2] = {{0, 1}, {1, 2}, {2, 3}, {3, 4}, {4, 5}, {5, 6}, {6, 7}, {7, 8}};
            for (auto& sg : seg) fprintf(stderr, " %7.1f |", (t[sg[1]] - t[sg[0]]) / 100.0);
            fprintf(stderr, " %7.1f\n", (t[8] - t[0]) / 100.0);
        }
        (void)hipHostFree(wstamps);
    }
#endif
    return SF_OK;
}

// ---------------------------------------------------------------------------------------------
// Bordered band matrices on the fused panel kernel (the structure-exploiting solver for half-widths beyond the
// LDS window of k_band_forms):
//
//        [ Bd   .  ]      Bd: nband x nband, zero further than `halfwidth` from the diagonal (128 x 128 tiles of a
//    A = [         ]          dense-strided array; only the tiles that meet the band are ever touched)
//        [ R    G  ]      R:  the 1 + m right-hand sides as 64 extra ROWS,  G = 0
//
// Left-looking panels exactly as in sf_launch_potrf_v2, but rest(k) covers only the slabs that meet the band plus the
// border slab, and every slab's K loop starts at its first non-zero column: O(n W^2) flops on kernels that run at
// the dense path's rate, spread over the whole chip (round 1's in-place sweep kept one matrix on one CU and streamed
// its operands from L2: 7.5 / 10.3 / 30.1 ms at W = 241 / 361 / 724 against 5.0 / 6.0 / 11.3 here).  The border rows come out as Z = R L^-T, their diagonal tile as -Z Z^T: the Gram matrix the
// Woodbury step needs; L_band's diagonal gives logdet(Bd).  The diagonal tile of the border is never factorised.
// border rows: row 0 <- rhs0 (the residual), rows 1 .. nrhs-1 <- rhs rows, everything else (and the border's own
// diagonal tile) zero
__global__ __launch_bounds__(256) void k_band_border_rows(const double* __restrict__ rhs0, int64_t srhs0, const double* __restrict__ rhs,
                                                          int64_t srhs, int ldr, int nrhs, int n, int nband, double* __restrict__ A,
                                                          int64_t sA, int lda) {
    const int b = blockIdx.z, r = blockIdx.y, col = blockIdx.x * 256 + threadIdx.x;
    if (col >= nband + 64) return;
    double v = 0.0;
    if (r < nrhs && col < n) v = r == 0 ? rhs0[(int64_t)b * srhs0 + col] : rhs[(int64_t)b * srhs + (int64_t)(r - 1) * ldr + col];
    A[(int64_t)b * sA + (int64_t)(nband + r) * lda + col] = v;
}
__global__ __launch_bounds__(256) void k_band_tiles_finish(const double* __restrict__ A, int64_t sA, int lda, int nband, int nrhs,
                                                           double* __restrict__ logdet, double* __restrict__ gram) {
    __shared__ double red[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const double* Ab = A + (int64_t)b * sA;
    double acc = 0.0;
    for (int i = tid; i < nband; i += 256) acc += log(Ab[(int64_t)i * lda + i]);
    red[tid] = acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (tid < st) red[tid] += red[tid + st];
        __syncthreads();
    }
    if (tid == 0) logdet[b] = 2.0 * red[0];
    for (int e = tid; e < nrhs * nrhs; e += 256) {
        const int r = e / nrhs, c = e - r * nrhs;
        gram[(int64_t)b * nrhs * nrhs + e] = -Ab[(int64_t)(nband + max(r, c)) * lda + nband + min(r, c)];
    }
}

int sf_band_tiles_lda(int nband) { return nband + 64 + 16; }
int sf_band_tiles_wt(int halfwidth) { return (halfwidth + GT - 1) / GT; }
size_t sf_band_tiles_doubles(int nband, int batch) {  // the dense-strided array + the factorisation's scratch
    return (size_t)batch * (nband + 64) * sf_band_tiles_lda(nband) + sf_potrf_work_doubles(nband + 64, batch) + 64;
}

// The lower 128 x 128 tiles that meet the band are in place at the start of `tiles` (k_band_fill's tile mode: row
// stride sf_band_tiles_lda(nband), nband + 64 rows per matrix, zeros where the band ends inside a tile, identity
// padding from n to nband = n rounded up to 64); `tiles` holds sf_band_tiles_doubles(nband, batch) doubles.  rhs0 /
// rhs: the right-hand sides (row 0 separate, as in sf_launch_band_forms).  Outputs logdet(Bd) and the nrhs x nrhs
// Gram matrix of the solved right-hand sides; info[b] (cleared by the caller) gets the first non-positive pivot.
int sf_launch_potrf_band(int n, int nband, int halfwidth, int batch, const double* rhs0, int64_t srhs0, const double* rhs,
                         int nrhs, int ldr, int64_t srhs, double* logdet, double* gram, int* info, double* tiles,
                         hipStream_t s) {
    if (nband % SF_LEAF != 0 || nband < n || batch <= 0 || nrhs < 1 || nrhs > 64 || halfwidth < 0 || !tiles) {
        sf_set_error("potrf_band: bad arguments (n=%d nband=%d halfwidth=%d nrhs=%d)", n, nband, halfwidth, nrhs);
        return SF_EINVAL;
    }
    const int next = nband + 64, lda = sf_band_tiles_lda(nband);
    const int64_t sA = (int64_t)next * lda;
    double* A = tiles;
    double* work = tiles + (size_t)batch * sA;
    const int nt = (nband + GT - 1) / GT;
    const int wt = sf_band_tiles_wt(halfwidth);

    hipLaunchKernelGGL(k_band_border_rows, dim3((next + 255) / 256, 64, batch), dim3(256), 0, s, rhs0, srhs0, rhs, srhs, ldr, nrhs, n,
                       nband, A, sA, lda);
    SF_LAUNCH_CHECK();

    double* T = work + (size_t)batch * SF_LTB_DOUBLES;
    const int64_t sT = (int64_t)(next + SF_NB) * SF_LDT + SF_TSKEW;
    double* Wt = T + (size_t)batch * sT;
    const int64_t sW = (int64_t)SF_NB * SF_LDT + SF_TSKEW;
    // (info is NOT cleared here: the band fill may have flagged a half-width that is too small; a non-zero entry stays)
    // One stream, two launches per panel: the launches are short (a few slabs, K <= halfwidth + 128), so the
    // lookahead of the dense sequence has nothing to hide behind -- measured with the chain on a side stream:
    // 13.4 ms against 9.4 at W = 361 (cross-stream waits cost more than the kernels they overlap).
    auto launch_panel = [&](int k0, int pw, int row0, int nslab, bool border) -> int {
        sf_panel_args g = {};
        g.C = A;
        g.sC = sA;
        g.lda = lda;
        g.n = next;
        g.k0 = k0;
        g.pw = pw;
        g.row0 = row0;
        g.nslab = nslab + (border ? 1 : 0);
        g.slab_step = 1;
        g.Wt = Wt;
        g.sW = sW;
        g.kband = halfwidth > 0 ? halfwidth : 1;
        g.nband = nband;
        g.xrow0 = border ? nband : 0;
        if (nslab > 0) {  // the first slab is the next diagonal tile: its update is parked in the scratch for D(k+1)
            g.Sout = T;
            g.sS = sT;
            g.ldS = SF_LDT;
        }
        const long long nblk = (long long)g.nslab * batch;
        if (nblk <= 0) return SF_OK;
        void* tok;
        sf_prof_gemm_begin(s, 2.0 * (double)min(k0, halfwidth + GT) * GT * pw * (double)nblk, &tok);
        hipLaunchKernelGGL((k_chol_panel<false, 0>), dim3((unsigned)nblk), dim3(512), 0, s, g);
        sf_prof_gemm_end(tok);
        SF_LAUNCH_CHECK();
        return SF_OK;
    };
    SF_TRY(launch_panel(0, 0, 0, 1, false));  // diagonal tile 0 goes to the scratch unchanged
    for (int k = 0; k < nt; ++k) {
        const int k0 = k * GT;
        const int pw = (nband - k0 < GT) ? nband - k0 : GT;
        SF_TRY(sf_launch_diag128(T, sT, pw, info, k0, nullptr, 0, A + (int64_t)k0 * lda + k0, lda, sA, Wt, sW, batch, s));
        // the slabs k+1 .. k+wt that meet the band, and the border
        const int last = (k + wt < nt - 1) ? k + wt : nt - 1;
        SF_TRY(launch_panel(k0, pw, (k + 1) * GT, last - k, true));
    }
    hipLaunchKernelGGL(k_band_tiles_finish, dim3(batch), dim3(256), 0, s, A, sA, lda, nband, nrhs, logdet, gram);
    SF_LAUNCH_CHECK();
    return SF_OK;
}


// =====================================================================================================================
// DATAFLOW sequence (round 4): the whole factorisation of a batch as ONE persistent launch.
//
// The launch sequences above are bound by their panel boundaries once the batch no longer fills the chip many times over
// (cfg 2 split over 2 / 4 / 8 GPUs: 64 / 32 / 16 matrices): the chain D(k) -> top(k) -> D(k+1) waits for workgroup slots
// behind bulk workgroups that start and end together, the bulk launches wait for the chain's events, every launch fills
// and drains the chip on its own (timelines in profiles/r04_*: the three streams 85-90 % busy, the matrix cores 0.49-0.74).
// Here 512 workgroups (two per CU) stay resident and draw TASKS; a task waits for exactly the tasks whose results it reads
// (monotone counters in global memory, agent scope) -- nothing else orders the work.  Per panel k and matrix b:
//   C(b,k)      chain task: the step of slab k for panel k-1 -- the partial sums FP(b,k-1,1,.) added in split order, then the K
//               tail over the last 128 columns, solve, L in place, tile (k,k) parked -- and the diagonal tile D(b,k) right
//               behind it in the SAME workgroup.  The K work needs the FIRST half of the previous chain task (row k-1
//               final), only the solve its second half (D(b,k-1)): it runs beside that diagonal tile, in another workgroup.
//               Chain tasks are not queued: whichever workgroup finds one READY at the dispenser claims it (compare-and-swap
//               on the matrix's chain counter) -- in a queue it waited until a workgroup had worked its way to it.
//   FP(b,k,d,s) partial sums of the FRONT slabs k+d, d = 1..3, for panel k over the columns LEFT of panel k-1: they depend on
//               tasks two stages back, so they run long before row k is final
//   FR(b,k,d)   d = 2, 3: partial sums added + K tail + solve + L in place + own diagonal tile for slab k+d.  With the front
//               three slabs wide the rows the chain needs next are one reduce-and-epilogue behind it (~150 us), not one
//               long-K task: a lead slab as an ordinary task held the chain of 16 matrices at ~400 us per panel
//   R(b,i,k)    the fused panel step of the slabs i >= k+4 (MODE 0: the K loop starts as soon as row k is final, only the
//               triangular solve waits for D(b,k)), or, while a stage has fewer tasks than its XCD has workgroup slots,
//               RP(b,i,k,s) partial sums + RR(b,i,k) reduce + epilogue
// Queued tasks are drawn in an order in which every dependency precedes its dependants (stage k: FP(.,k+1,.,.), FR(.,k,.),
// R / RP(.,.,k), RR(.,.,k)); a workgroup holds at most one task, only claimed tasks are waited for, a chain task is claimed
// only when its K work can start: the schedule cannot deadlock whatever the residency or placement of the workgroups.  The
// inverse tiles W_k of ALL panels are kept (one per panel, in the scratch the unfused sequence uses for its panel): no
// buffer of the chain is ever recycled.  Same arithmetic as the fused sequence (the same kernels' bodies); the summation
// order differs where the split differs.
struct sf_df_stage {
    int off;      // first task of the stage's segment
    int St;       // split of FP(., k+1, ., .): 0 = the front tasks of panel k+1 run their whole K loops themselves
    int Sr;       // split of the ordinary rest tasks (1 = unsplit MODE 0)
    int thr_pt;   // FP(b, k+1, d, .) arrivals the front task of (b, k+1, d) waits for (cumulative over the panels of that parity)
    int thr_rp;   // RP(b, i, k, .) arrivals RR(b, i, k) waits for (cumulative)
    int dep;      // RP of this stage re-uses the partial-sum region of stage `dep` (same parity, split): wait for its reduces
    int fw;       // front width of this panel (slabs k+1 .. k+F are front slabs) | front slabs of panel k+1 that exist << 8
};
// ... and as it travels in the kernel arguments (12 bytes: two tables of 128 stages stay below the 4 KB of a kernel's arguments;
// N = 16384 has 127 stages)
struct sf_df_stage_packed {
    int off;
    unsigned short thr_pt, thr_rp, fw;
    unsigned char split;  // St | Sr << 4
    signed char dep;
};
static_assert(sizeof(sf_df_stage_packed) == 12, "sf_df_stage_packed");
template <class S>
__host__ __device__ __forceinline__ sf_df_stage sf_df_stage_of(S& x) {  // (copy out of the constant address space)
    sf_df_stage r;
    r.off = x.off;
    const int sp = x.split;
    r.St = sp & 15;
    r.Sr = sp >> 4;
    r.thr_pt = x.thr_pt;
    r.thr_rp = x.thr_rp;
    r.dep = x.dep;
    r.fw = x.fw;
    return r;
}
static inline sf_df_stage_packed sf_df_pack(const sf_df_stage& x) {
    sf_df_stage_packed r;
    r.off = x.off;
    r.thr_pt = (unsigned short)x.thr_pt;
    r.thr_rp = (unsigned short)x.thr_rp;
    r.fw = (unsigned short)x.fw;
    r.split = (unsigned char)(x.St | (x.Sr << 4));
    r.dep = (signed char)x.dep;
    return r;
}
// One task queue per XCD: matrix b belongs to queue b % 8 (its slabs share the B operand L[panel rows, :k0] through that
// XCD's L2 -- with ONE queue for the chip the operand was fetched by every XCD: L2 hit rate 0.14 instead of 0.38, 1.5 x the
// HBM reads); a workgroup serves the queue of the XCD it runs on and, once that is exhausted, the others in turn.  Queues
// with the same number of matrices share a task table (at most two sizes).
#define SF_DF_QUEUES 8
#define SF_DF_MAX_STAGES 127  // (two tables of 12-byte entries in the kernel arguments: < 4 KB; N = 16384 = 128 panels)
#define SF_DF_FRONT_MAX 6    // slabs k+1 .. k+front of panel k are front slabs (front <= 6, chosen by the batch size)
#define SF_DF_FRONT_WIDEST 3 // ... and the widest front a release build chooses: the stride of the front's partial sums and counters
#define SF_DF_QTILES (2 * SF_CHIP_WGS / SF_DF_QUEUES)  // partial-sum tiles per queue and stage parity
struct sf_df_args {
    sf_panel_args p;  // matrix, right-hand side, generator, frame: the per-task fields are filled in by the kernel
    int nt, batch, front;  // front: the LARGEST front width (the width of panel k is st[.][k].fw & 255: it grows towards the end)
    int fstart[SF_DF_FRONT_MAX];        // first panel whose front is d slabs wide (index d - 1): chain_next[.][d - 1] counts from there
    int thr_base[2][2][SF_DF_FRONT_MAX];  // [table][panel parity][d - 1]: partial-sum arrivals of that parity before distance d existed
    int fp_pos;       // position of the front partial sums inside a stage's segment, in 1/256 of its rest tasks
    int bq[2], ntasks[2];  // table v serves the queues with bq[v] matrices
    int pt_cap;       // largest split of the front partial sums: a (matrix, front slab) owns pt_cap tiles per panel parity in region 2
    int *head, *abort_flag, *done_top, *done_D, *done_row, *row_L, *fp_cnt, *rp_cnt, *stage_done;
    int* chain_next;  // [batch][3]: the next chain task (d = 1) / front task (d = 2, 3) of every matrix (claimed by compare-and-swap once ready)
    double* T;        // per matrix: parked diagonal tile [GT x SF_LDT], then W_k for every panel
    int64_t sT;
    double* part;     // three regions of sf_split_region_tiles() tiles: rest partial sums by stage parity, front partial sums
    int* info;
    int qbal;         // 1: with fewer matrices than queues the XCDs are dealt to the non-empty queues round-robin
    long long* diag;  // the process's abort record in host memory (sf_df_diag), or NULL
    long long* dbg;   // tuning builds: per workgroup {ticks waiting, ticks in task bodies, tasks, ticks by type} (100 MHz)
    int miss_claims;  // tuning builds (SF_DF_MISS_CLAIMS): 1 = the dispenser leaves chain / front tasks to the waits' rescue while its queues hold tasks; 2 = a claimed chain task is never run (forces the stall bound)
    long long* trace; // tuning builds (SF_DF_TRACE_FILE): [0] = records written, then {type | k << 8 | i << 16 | b << 24 | workgroup << 40, claimed, body start, end}
    long long trace_cap;
    sf_df_stage_packed st[2][SF_DF_MAX_STAGES];
};
static_assert(sizeof(sf_df_args) <= 4096, "kernel arguments of k_potrf_dataflow");
#define SF_DF_LDS_DOUBLES ((37 * DBS + 128) > (4 * GT * GLD + 2 * GT) ? (37 * DBS + 128) : (4 * GT * GLD + 2 * GT))
#define SF_DF_LDS_BYTES ((SF_DF_LDS_DOUBLES + 4) * sizeof(double))

typedef const __attribute__((address_space(4))) sf_df_args sf_df_kargs;
// The dispenser's scans are real function calls (one lane, once per task): inlined at their three sites they pushed the
// register allocation of the whole task loop over the edge (a spill reload inside a K loop, tools/check_isa.py).
#ifdef SF_EXP_HELPER_INLINE
#define SF_DF_HELPER __forceinline__
#else
#define SF_DF_HELPER __attribute__((noinline))
#endif
#ifdef SF_EXP_NOPROGADD  // (timing experiment: what do the progress counter's adds cost?  The stall bound then fires on any long wait)
#define SF_DF_PROGRESS()
#else
#define SF_DF_PROGRESS() sf_df_add(a.abort_flag + 5, 1)
#endif
#ifdef SF_TUNING
#define SF_DF_MISS_CLAIMS(x) ((a.miss_claims & 1) && (x))
#else
#define SF_DF_MISS_CLAIMS(x) (false)
#endif
// A ready chain (d = 1) / front (d >= 2) task among the matrices of queue qx that nobody has claimed?  Claims it by
// compare-and-swap on the matrix's counter: cb = matrix, ck = chain task index (d = 1) or panel (d >= 2), cd = d.  One lane.
__device__ SF_DF_HELPER bool sf_df_try_chain(sf_df_kargs& a, const int qx, int& cb, int& ck, int& cd) {
    const int nt = a.nt, F = a.front;
    const int Bq = (a.batch - qx + SF_DF_QUEUES - 1) / SF_DF_QUEUES;
    const int vq = Bq == a.bq[0] ? 0 : 1;
    for (int dd = 1; dd <= F; ++dd) {  // (the chain itself first)
        for (int j = 0; j < Bq; ++j) {
            const int b1 = qx + SF_DF_QUEUES * j;
            int* ctr = a.chain_next + SF_DF_FRONT_MAX * b1 + dd - 1;
            const int k1 = sf_df_load(ctr);  // d = 1: chain task index (panel k1 - 1); d >= 2: panel - fstart
            const int kp = dd == 1 ? k1 - 1 : k1 + a.fstart[dd - 1];
            if (dd == 1 ? k1 >= nt : kp + dd > nt - 1) continue;
            bool ready = true;
            if (kp >= 0) {
                ready = sf_df_load(a.done_top + b1) >= kp;
                if (ready && kp >= 1) {
                    ready = sf_df_load(a.row_L + (size_t)b1 * nt + kp + dd) >= kp;
                    const int St = a.st[vq][kp - 1].split & 15;
                    if (ready && St > 0)
                        ready = sf_df_load(a.fp_cnt + 2 * SF_DF_FRONT_MAX * b1 + SF_DF_FRONT_MAX * (kp & 1) + dd - 1) >=
                                a.st[vq][kp - 1].thr_pt - a.thr_base[vq][kp & 1][dd - 1];
                }
            }
            if (!ready) continue;
            int expect = k1;
            if (__hip_atomic_compare_exchange_strong(ctr, &expect, k1 + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                     __HIP_MEMORY_SCOPE_AGENT)) {
                cb = b1;
                ck = dd == 1 ? k1 : kp;
                cd = dd;
                return true;
            }
        }
    }
    return false;
}

template <bool RHS>
__global__ __launch_bounds__(512, 4) void k_potrf_dataflow(const sf_df_args a_in) {
    extern __shared__ __attribute__((aligned(16))) double dsm[];
    double* sm = dsm;
    double(*red)[GT] = (double(*)[GT])(dsm + 4 * GT * GLD);
    int* s_ints = (int*)(dsm + SF_DF_LDS_DOUBLES);  // [0] task id, [1] wait result, [2] [3] chain task: behind the kernels' LDS image, which starts at 0
    const size_t region = sf_split_region_tiles_dev() * (size_t)(GT * GT);
    // The arguments are read through the kernel-argument segment pointer inside the task loop, and everything derived from
    // the thread index is recomputed per task (the index is laundered through an empty asm): otherwise hipcc hoists the
    // lane-dependent invariants of all the inlined task bodies out of the loop and spills them (600 bytes of scratch per
    // lane, scratch loads inside the MFMA loops).
    sf_df_kargs* ap = (sf_df_kargs*)__builtin_amdgcn_kernarg_segment_ptr();
    (void)a_in;
    // (workgroup b of a launch runs on XCD b % 8 -- observed, not promised; placement is a speed matter only here: any
    // workgroup may serve any queue)
    // Workgroups are dealt to the queues in proportion to the MATRICES a queue holds (round 6): every matrix gets 512 / batch
    // workgroup slots.  A batch that is not a multiple of 8 leaves the first batch % 8 queues one matrix more than the others
    // (with fewer than 8 matrices: the others empty): the XCDs of the smaller queues keep round(matrices x 512 / batch) of
    // their 64 workgroups and send the rest to the larger queues, round-robin.  Before, such workgroups only moved on when
    // their own queue was exhausted -- with an empty own queue all of them to queue 0, whose one matrix then had five XCDs
    // (limited by its chain) while the others had one each (limited by throughput): N = 16384, 4 matrices 180 ms against 109
    // for the launch sequence; 12 matrices cost what 16 cost.  (8 % batch == 0: whole XCDs, the matrix's operands stay in
    // one L2.)
    int qcur = (int)(blockIdx.x & (SF_DF_QUEUES - 1));
    {
        const int nb = ap->batch, big = nb % SF_DF_QUEUES;  // queues 0 .. big - 1 hold one matrix more
        if (ap->qbal && big != 0) {
            if (nb < SF_DF_QUEUES && SF_DF_QUEUES % nb == 0) {
                qcur = qcur % nb;
            } else {
                const int mine = (nb - qcur + SF_DF_QUEUES - 1) / SF_DF_QUEUES;  // matrices of this XCD's own queue
                const int slot = (int)(blockIdx.x >> 3);
                const int keep = (int)(((long long)mine * gridDim.x + nb / 2) / nb);  // its share of the grid's workgroups
                if (qcur >= big && slot >= keep) qcur = (slot - keep + qcur) % big;
            }
        }
    }
    int visited = 0;
    int kst = 0;  // stage hint: a workgroup draws the tasks of a queue in increasing order
    if (threadIdx.x == 0) {
        s_ints[5] = 0;  // (idle spell of the end-of-launch phase, see the dispenser)
        s_ints[6] = 0;
        sf_df_add(ap->abort_flag + 6, 1);  // workgroups of the launch that have started (diagnostic of an aborted launch)
        if (blockIdx.x == 0)  // (where sf_df_report finds the abort record: the waits only carry the abort flag's address)
            __hip_atomic_store((long long*)(ap->abort_flag + 8), (long long)ap->diag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // A queued task whose wait was interrupted to run a chain / front task nobody had claimed (sf_df_wait_r): resume = 1 the
    // claimed chain task is in s_ints[0..4] already; pend_t >= 0: that queued task is taken up again instead of a new one.
    int pend_t = -1, resume = 0;
    for (;;) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        sf_df_kargs& a = *ap;
        const int nt = a.nt, F = a.front;
        const int n = a.p.n, fp = a.p.fp;
        const int B = (a.batch - qcur + SF_DF_QUEUES - 1) / SF_DF_QUEUES;  // matrices of this queue: qcur, qcur + 8, ...
        const int v = B == a.bq[0] ? 0 : 1;
        const int ntasks = B > 0 ? a.ntasks[v] : 0;
        // ---- dispenser.  Chain tasks first, then the queue of this workgroup's XCD, then the other queues.
        // (A claim can be missed: two workgroups that complete the last two dependencies of a chain task within a store's flight
        // time of each other may both read the other's counter too early and both find the task not ready.  The next
        // workgroup of the queue that passes here claims it, a few us later; if every workgroup of the queue sits in a wait
        // by then, the workgroups of the other queues do at the end of the launch, when each looks at every chain.  Measured
        // and not taken: waiting for this workgroup's counter stores to be acknowledged before the scan (2 % of a launch),
        // one lane per candidate instead of one lane walking the matrices (claims cost 20 instead of 28 us, launches of 8-32
        // matrices ran 2-5 % slower), waits that give up after 20 us to serve the chains and come back (3-14 % slower).
        // Since round 5 the window is closed where it matters: a queued task's wait that has lasted 500 us looks at the chains
        // itself, sf_df_wait_r.)
        if (resume) {  // s_ints[0..4] = the chain task claimed inside the interrupted wait
            resume = 0;
        } else if (pend_t >= 0) {  // back to the task that was set aside
            if (tid == 0) s_ints[0] = pend_t;
            pend_t = -1;
        } else if (tid == 0) {
            int t = -1, cb = 0, ck = 0, cd = 1;
            if (sf_df_load(a.abort_flag) == 0) {
                t = -2;
                auto try_chain = [&](int qx) { return SF_DF_MISS_CLAIMS(visited < SF_DF_QUEUES) ? false : sf_df_try_chain(a, qx, cb, ck, cd); };
                if (try_chain(qcur)) {
                    t = -3;
                } else if (visited < SF_DF_QUEUES) {
                    t = ntasks > 0 ? __hip_atomic_fetch_add(a.head + qcur, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ntasks;
                    if (t >= ntasks) t = -4;  // this queue is exhausted
                } else {
                    // every queue is exhausted: help the chains that are still running, leave when none is
                    bool live = false;
                    for (int qx = 0; qx < SF_DF_QUEUES && t == -2; ++qx)
                        if (try_chain(qx)) {
                            t = -3;
                            s_ints[5] = 0;
                        }
                    if (t == -2) {
                        for (int b1 = 0; b1 < a.batch; ++b1) {
                            live = live || sf_df_load(a.chain_next + SF_DF_FRONT_MAX * b1) < nt;
                            for (int dd = 2; dd <= F; ++dd)
                                live = live || sf_df_load(a.chain_next + SF_DF_FRONT_MAX * b1 + dd - 1) + a.fstart[dd - 1] + dd <= nt - 1;
                        }
                        if (!live) {
                            t = -5;
                        } else {
                            // (bounded like every wait: chains that stay open with nothing left to run them would spin here for ever)
                            // (s_ints[5]: the 10.5 ms unit of the wall clock at which this idle spell began -- or the launch's
                            // progress counter, s_ints[6], last moved --, + 1; 0 = none.  No task completed for three units, 21-31
                            // ms: SF_DF_STALL_TICKS)
                            const int now = (int)((wall_clock64() >> 20) & 0x3fffffff) + 1;
                            const int pg = sf_df_load(a.abort_flag + 5);
                            if (s_ints[5] == 0 || pg != s_ints[6]) {
                                s_ints[5] = now;
                                s_ints[6] = pg;
                            }
                            const int idle = (now - s_ints[5]) & 0x3fffffff;
                            const int stall = (int)(sf_df_stall_ticks(a.abort_flag) >> 20);
                            if (idle > stall) {
                                sf_df_report(a.abort_flag, a.chain_next, nt, SF_DF_ABORT_STALL, (long long)idle << 20);
                                t = -1;
                            }
                            __builtin_amdgcn_s_sleep(64);
                        }
                    }
                }
            }
            s_ints[0] = t;
            s_ints[2] = cb;
            s_ints[3] = ck;
            s_ints[4] = cd;
        }
        __syncthreads();
        const int t = __builtin_amdgcn_readfirstlane(s_ints[0]);  // (wave-uniform: everything decoded from it lives in SGPRs)
        const int chain_b = __builtin_amdgcn_readfirstlane(s_ints[2]), chain_k = __builtin_amdgcn_readfirstlane(s_ints[3]);
        const int chain_d = __builtin_amdgcn_readfirstlane(s_ints[4]);
        __syncthreads();
        if (t == -1) {  // a wait timed out somewhere: nothing of this launch can be trusted
            if (a.info)
                for (int bb = tid; bb < a.batch; bb += 512) a.info[bb] = SF_INFO_INTERNAL;
            return;
        }
        if (t == -5) return;
        if (t == -2) continue;
        if (t == -4) {  // this queue is exhausted: the next one
            ++visited;
            qcur = (qcur + 1) & (SF_DF_QUEUES - 1);
            kst = 0;
            continue;
        }

        // ---- decode: k = panel, i = slab, d = i - k for front tasks
        enum { T_C, T_FP, T_FR, T_R, T_RP, T_RR };
        int type, bl = 0, k, i = 0, sp = 0, S = 1, d = 0;
        int bchain = -1;
        if (t == -3) {
            type = chain_d == 1 ? T_C : T_FR;
            bchain = chain_b;
            k = chain_k;
            d = chain_d;
            i = k + d;
        } else {
            while (kst + 1 < nt - 1 && t >= a.st[v][kst + 1].off) ++kst;
            const sf_df_stage st = sf_df_stage_of(a.st[v][kst]);
            // front slabs of panel kst + 1 that exist, front slabs d >= 2 of this panel, ordinary slabs of this panel
            const int Fk = st.fw & 255, nF = st.fw >> 8;
            const int nord = max(0, nt - kst - 1 - Fk);
            const int n_fp = B * nF * st.St;
            const int n_r1 = B * nord * (st.Sr > 1 ? st.Sr : 1);
            // segment: fp_pos/256 of the rest tasks, the front partial sums of the NEXT panel, the other rest tasks, the reduces.
            // (FP tasks at the very front of the segment are claimed while the two rows they read are still being finished by
            // tasks of the previous stage: with many matrices per queue -- the chain is not what the rest waits for -- they
            // come later: 1.1 of 1.95 ms of waiting per workgroup at 32 matrices was theirs)
            const int n_r0 = (int)(((long long)n_r1 * a.fp_pos) >> 8);
            int u = t - st.off;
            if (u >= n_r0 && u < n_r0 + n_fp) {
                u -= n_r0;
                type = T_FP;
                k = kst + 1;
                S = st.St;
                d = 1 + u / (B * S);  // (d = 1 first: the chain's own partial sums)
                u -= (d - 1) * B * S;
                bl = u / S;
                sp = u - bl * S;
                i = k + d;
            } else if (u < n_r1 + n_fp) {
                if (u >= n_r0) u -= n_fp;
                k = kst;
                S = st.Sr;
                type = S > 1 ? T_RP : T_R;
                const int tile = u / S;  // ordinary slabs matrix by matrix: tasks side by side on an XCD stream the same B operand
                sp = u - tile * S;
                bl = tile / nord;
                i = k + 1 + Fk + (tile - bl * nord);
            } else {
                u -= n_r1 + n_fp;
                k = kst;
                S = st.Sr;
                type = T_RR;
                bl = u / nord;
                i = k + 1 + Fk + (u - bl * nord);
            }
        }
        bl = __builtin_amdgcn_readfirstlane(bl);  // (the divisions above ran on the VALU)
        i = __builtin_amdgcn_readfirstlane(i);
        sp = __builtin_amdgcn_readfirstlane(sp);
        k = __builtin_amdgcn_readfirstlane(k);
        S = __builtin_amdgcn_readfirstlane(S);
        d = __builtin_amdgcn_readfirstlane(d);
        type = __builtin_amdgcn_readfirstlane(type);
        const int b = bchain >= 0 ? bchain : qcur + SF_DF_QUEUES * bl;  // the matrix
#ifdef SF_TUNING
        // test aid (SF_DF_MISS_CLAIMS=2): the workgroup that claimed the chain task of panel 2 of matrix 0 never runs it --
        // what a workgroup that is kept from running looks like to the others: everything downstream waits, no task
        // completes any more, the stall bound gives the launch up (tests/test_gpu_recovery.py)
        if ((a.miss_claims & 2) && type == T_C && b == 0 && k == 2) {
            if (tid == 0)
                while (sf_df_load(a.abort_flag) == 0) __builtin_amdgcn_s_sleep(64);
            __syncthreads();
            continue;
        }
#endif
        const int vb = bchain >= 0 ? (((a.batch - (b & (SF_DF_QUEUES - 1)) + SF_DF_QUEUES - 1) / SF_DF_QUEUES) == a.bq[0] ? 0 : 1) : v;  // its queue's table
#ifdef SF_TUNING
        const long long dbg_t0 = wall_clock64();
        long long dbg_t1 = dbg_t0, dbg_top = 0;
#define SF_DF_MARK() dbg_t1 = wall_clock64()
#else
#define SF_DF_MARK()
#endif

        // ---- the per-task fields of the panel step (a.p holds what is constant over the factorisation)
        const auto& g = a.p;
        sf_panel_task q = {};
        q.nslab = 1;
        q.slab_step = 1;
        q.abort_flag = a.abort_flag;
        q.lds_int = s_ints + 1;
        q.sW = a.sT;
        auto Wof = [&](int kk) { return a.T + (size_t)(1 + kk) * GT * SF_LDT; };
        // front partial sums of (panel parity, front slab d, matrix): pt_cap tiles each in region 2; the body indexes them with
        // the matrix number b and the split S of the panel: base = slot of (parity, d, b) minus b S
        auto fpart = [&](int kk, int dd, int SS) {
            return a.part + 2 * region + (((int64_t)((kk & 1) * F + dd - 1) * a.batch + b) * a.pt_cap - (int64_t)b * SS) * (GT * GT);
        };
        int* fcnt = a.fp_cnt + 2 * SF_DF_FRONT_MAX * b;
        bool ok = true;
        int mode = 3, bid = b;  // which body runs: 1 = partial sums (bid = b S + split), 3 = everything else
        if (type == T_C || type == T_FR) {
            // the step of the front slab k+d (chain: of slab k for panel kp = k - 1) for panel kp: partial sums, K tail, epilogue
            const int kp = type == T_C ? k - 1 : k;
            const int slab = kp + d;
            if (type == T_C) {
                __builtin_amdgcn_s_setprio(2);  // the chain's waves share their SIMDs with rest tasks issuing MFMAs back to back
                q.Sout = a.T;                   // (g.sS = a.sT, g.ldS = SF_LDT)
            }
            if (kp < 0) {
                // start of the factorisation: diagonal tile 0 goes to the scratch unchanged (pw = 0, mode 0)
                if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __syncthreads();
            } else {
                const sf_df_stage stp = sf_df_stage_of(a.st[vb][kp > 0 ? kp - 1 : 0]);  // (FP(., kp, ., .) belongs to stage kp - 1)
                const int St = kp >= 1 ? stp.St : 0;
                q.k0 = kp * GT;
                q.pw = min(GT, n - q.k0);
                q.row0 = slab * GT;
                q.Wt = Wof(kp);
                q.ksplit = St;
                q.ktail = St > 0 ? (kp - 1) * (GT / GK) : 0;
                q.part = fpart(kp, d, St);
                // the slab's own row left of the tail; row kp final = the FIRST half of the chain task C(b,kp) (only the solve
                // needs its second half, the diagonal tile: this task's K work runs beside it); the partial sums
                bool dready = false;
                // (own row: its L blocks left of the tail -- row_L; the slab's diagonal tile, updated by the previous panel's step
                // for this slab, is waited for inside the body, right before step 4)
                ok = sf_df_wait(kp >= 1 ? a.row_L + (size_t)b * nt + slab : nullptr, kp, a.done_top + b, kp,
                                St > 0 ? fcnt + SF_DF_FRONT_MAX * (kp & 1) + d - 1 : nullptr, stp.thr_pt - a.thr_base[vb][kp & 1][d - 1], a.done_D + b, kp + 1, &dready, a.abort_flag,
                                tid, s_ints + 1);
                if (!dready) {
                    q.wflag = a.done_D + b;
                    q.wval = kp + 1;
                }
                if (kp >= 1) {
                    q.sflag = a.done_row + (size_t)b * nt + slab;
                    q.sval = kp;
                }
                // (the row is published from inside the body, as soon as L is stored: the chain's row counter / the slab's row_L)
                q.top_flag = type == T_C ? a.done_top + b : a.row_L + (size_t)b * nt + slab;
                q.top_val = type == T_C ? k : kp + 1;
#ifdef SF_TUNING
                if (a.dbg && b == 0 && type == T_C && k < 64) q.stamps = a.dbg + 16 * SF_CHIP_WGS + 16 * 64 + 8 * k;
#endif
            }
        } else {
            // ---- queued tasks: ONE wait site for the four types (it carries the chain rescue, see sf_df_wait_r)
            const int k0 = k * GT;
            const int nk = (k0 > fp ? k0 - fp : 0) / GK;
            q.k0 = k0;
            q.pw = min(GT, n - k0);
            q.Wt = Wof(k);
            q.row0 = i * GT;
            const sf_df_stage st = sf_df_stage_of(a.st[vb][k]);
            const int Fk = st.fw & 255;
            const int nord = nt - k - 1 - Fk;
            int* rowflag = a.done_row + (size_t)b * nt + i;
            int* sdone = a.stage_done + (size_t)qcur * nt;
            const int *f1, *f2, *f3 = nullptr, *probe = nullptr;
            int t1, t2, t3 = 0;
            q.ksplit = S;
            if (type == T_FP) {
                // slab k+d, panel k, K slabs [fp / GK, (k - 1) 8): rows k and k+d through panel k-2; the slots' previous user (the
                // front task of (b, k-2, d)) must have read them: the chain's second half for d = 1, the row counter otherwise
                const int cnt = (k - 1) * (GT / GK) - fp / GK;
                q.kchunk = (cnt + S - 1) / S;
                q.kstop = (k - 1) * (GT / GK);
                q.part = fpart(k, d, S);
                f1 = a.done_row + (size_t)b * nt + k;
                t1 = k - 1;
                f2 = rowflag;
                t2 = k - 1;
                f3 = d == 1 ? a.done_D + b : a.done_row + (size_t)b * nt + i - 2;
                t3 = d == 1 ? k : k - 1;
            } else {
                q.kchunk = (nk + S - 1) / S;
                // (the body indexes the partial sums with the matrix number b: slot of (local matrix, slab) minus b S)
                q.part = a.part + (size_t)(k & 1) * region +
                         ((int64_t)qcur * SF_DF_QTILES + ((int64_t)bl * nord + (i - k - 1 - Fk) - b) * S) * (GT * GT);
                if (type == T_RR) {
                    f1 = a.rp_cnt + (size_t)b * nt + i;
                    t1 = st.thr_rp;
                    f2 = a.done_D + b;
                    t2 = k + 1;
                } else {
                    // K loop: row k through panel k-1 (the chain task's first half), the slab's own row through panel k-1; only
                    // the solve needs the diagonal tile -- if that is there already, this acquire covers it (T_R: probe)
                    f1 = k >= 1 ? a.done_top + b : nullptr;
                    t1 = k;
                    f2 = rowflag;
                    t2 = k;
                    if (type == T_R) {
                        probe = a.done_D + b;
                    } else if (st.dep >= 0) {  // T_RP re-uses the partial-sum slots of stage dep: its reduces must have read them
                        f3 = sdone + st.dep;
                        t3 = B * (nt - st.dep - 1 - (a.st[v][st.dep].fw & 255));
                    }
                }
            }
            bool dready = false;
            const int wr = sf_df_wait_r(f1, t1, f2, t2, f3, t3, probe, k + 1, &dready, a.abort_flag, tid, s_ints + 1,
                                        [&]() {  // (one lane) a ready chain / front task that nobody has claimed, on any queue
#ifdef SF_EXP_NORESCUE
                                            return false;
#endif
                                            int cb = 0, ck = 0, cd = 1;
                                            for (int x = 0; x < SF_DF_QUEUES; ++x) {
                                                const int qx = (qcur + x) & (SF_DF_QUEUES - 1);
                                                if (sf_df_try_chain(a, qx, cb, ck, cd)) {
                                                    s_ints[0] = -3;
                                                    s_ints[2] = cb;
                                                    s_ints[3] = ck;
                                                    s_ints[4] = cd;
                                                    return true;
                                                }
                                            }
                                            return false;
                                        },
                                        true);
            if (wr == SF_DF_DEFERRED) {  // a chain task first (claimed in the wait), then this task again
                pend_t = t;
                resume = 1;
                continue;
            }
            ok = wr == 1;
            if (type == T_FP || type == T_RP) {
                mode = 1;
                bid = b * S + sp;
            } else if (type == T_R) {  // the whole step: mode 3 without partial sums, K loop from the start
                q.ksplit = 0;
                q.ktail = 0;
                if (!dready) {
                    q.wflag = a.done_D + b;
                    q.wval = k + 1;
                }
            } else {  // T_RR: mode 3 with an empty K loop (the partial sums cover all of it)
                q.ktail = k0 / GK;
            }
        }
        // ---- the bodies: TWO inlined copies for the six task types -- the partial sums (FP, RP), and <3> for everything else: the
        // chain / front step as it is, the whole step (R) as <3> without partial sums, the reduce (RR) as <3> with an empty K
        // loop, the copy of diagonal tile 0 as <3> with pw = 0.  With one copy per type the kernel was 113 KB of code; the
        // instruction cache is 64 KB and shared by two CUs whose four workgroups run different task types (now: ~60 KB).
        SF_DF_MARK();
        if (ok) {
            if (mode == 1)
                sf_panel_body<RHS, 1>(g, q, bid, sm, red, tid);
            else
                sf_panel_body<RHS, 3>(g, q, bid, sm, red, tid);
        }
        if (type == T_C || type == T_FR) {
            const int kp = type == T_C ? k - 1 : k;
            const int slab = kp + d;
            if (ok && type == T_FR) {
                __syncthreads();
                if (tid == 0) {
                    sf_df_release();
                    sf_df_set(a.done_row + (size_t)b * nt + slab, kp + 1);
                    SF_DF_PROGRESS();  // (progress of the launch: see SF_DF_STALL_TICKS)
                }
            }
            if (ok && type == T_C) {
                __syncthreads();
#ifdef SF_TUNING
                dbg_top = wall_clock64();
#endif
                if (tid == 0) {  // the parked tile must be re-read through the L2 (k = 0: nothing was published from the body)
                    if (k == 0) {
                        sf_df_release();
                        sf_df_set(a.done_top + b, k);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
                __syncthreads();
                const int k0 = k * GT;
                const int pw = min(GT, n - k0);
                sf_diag_lds_body(a.T, a.sT, pw, a.info, k0 - fp, g.rhs ? g.rhs + k0 : nullptr, g.ldr,
                                 g.C + (int64_t)k0 * g.lda + k0, g.lda, g.sC, Wof(k), a.sT, k == 0 ? fp : 0, b, dsm, tid);
                __syncthreads();
                if (tid == 0) {
                    sf_df_release();
                    sf_df_set(a.done_D + b, k + 1);
                    SF_DF_PROGRESS();
                }
            }
            if (type == T_C) __builtin_amdgcn_s_setprio(0);
        } else {
            int* rowflag = a.done_row + (size_t)b * nt + i;
            int* sdone = a.stage_done + (size_t)qcur * nt;
            // (a workgroup that left the body on a timed-out wait finds the abort flag at the dispenser)
            __syncthreads();
            if (ok && tid == 0) {
                sf_df_release();
                if (type == T_FP) {
                    sf_df_add(fcnt + SF_DF_FRONT_MAX * (k & 1) + d - 1, 1);
                } else if (type == T_RP) {
                    sf_df_add(a.rp_cnt + (size_t)b * nt + i, 1);
                } else {
                    sf_df_set(a.row_L + (size_t)b * nt + i, k + 1);
                    sf_df_set(rowflag, k + 1);
                    if (type == T_RR) sf_df_add(sdone + k, 1);
                }
                SF_DF_PROGRESS();
            }
        }
        if (!ok) continue;  // (timed out: the dispenser sees the abort flag and flags every matrix)
        __syncthreads();  // the next task re-uses the LDS
#ifdef SF_TUNING
        if (a.dbg && tid == 0) {
            const long long t2 = wall_clock64();
            long long* dd = a.dbg + 16 * (size_t)blockIdx.x;
            dd[0] += dbg_t1 - dbg_t0;
            dd[1] += t2 - dbg_t1;
            dd[2] += 1;
            dd[3 + type] += t2 - dbg_t0;
            dd[9 + type] += dbg_t1 - dbg_t0;
            if (a.trace) {
                const long long slot = (long long)atomicAdd((unsigned long long*)a.trace, 1ull);
                if (slot < a.trace_cap) {
                    long long* r = a.trace + 4 + 4 * slot;
                    r[0] = (long long)type | ((long long)k << 8) | ((long long)i << 16) | ((long long)b << 24) | ((long long)blockIdx.x << 40);
                    r[1] = dbg_t0;
                    r[2] = dbg_t1;
                    r[3] = t2;
                }
            }
            if (b == 0 && k < 64) {  // timeline of matrix 0: chain task, its partial sums, the front slab d = 2
                long long* tr = a.dbg + 16 * SF_CHIP_WGS + 16 * k;
                const int slot = type == T_C ? 0 : (type == T_FP && d == 1 && sp == 0) ? 3 : (type == T_FP && d == 2 && sp == 0) ? 6 : (type == T_FR && d == 2) ? 9 : -1;
                if (slot >= 0) {
                    tr[slot] = dbg_t0;
                    tr[slot + 1] = dbg_t1;
                    tr[slot + 2] = t2;
                    if (type == T_C) tr[12] = dbg_top;
                }
            }
        }
#endif
    }
}

// split factor of a stage's tasks: the largest power of two that keeps the stage within the workgroup slots of its queue's
// XCD and every K chunk at 8 slabs or more
static int sf_df_split(long long tasks, int nk, int smax, int cap) {
    int S = 1;
    while (2 * S <= smax && tasks * 2 * S <= cap && nk / (2 * S) >= 8) S *= 2;
    return S;
}

static int g_df_enabled_query(void);
// The abort record of the process: six long longs of pinned host memory that the workgroup which aborts a persistent launch
// fills in (sf_df_report) and sf_persistent_potrf_status() hands to the caller's warning -- the status itself travels in
// d_info like every other (SF_INFO_INTERNAL).  Allocated on the first persistent launch; visible to every device.
static long long* g_df_diag = nullptr;
static std::atomic<long long> g_df_launches{0};
static long long* sf_df_diag(void) {
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (!g_df_diag) {
        void* p = nullptr;
        if (hipHostMalloc(&p, 8 * sizeof(long long), hipHostMallocPortable | hipHostMallocMapped) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;  // (no record then: the launch itself does not depend on it)
        }
        for (int i = 0; i < 8; ++i) ((volatile long long*)p)[i] = 0;
        g_df_diag = (long long*)p;
    }
    return g_df_diag;
}
int sf_persistent_potrf_read_status(long long* out8) {
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    const volatile long long* d = g_df_diag;
    for (int i = 0; i < 6; ++i) out8[i] = d ? d[i] : 0;
    out8[6] = g_df_launches.load();
    out8[7] = g_df_enabled_query();
    return SF_OK;
}

static int sf_launch_potrf_v4(double* A, int n, int lda, int64_t stride, int batch, int* info, double* work,
                              double* rhs, int ldr, hipStream_t s, const sf_gen_args* gen, int fp) {
    if (n % SF_LEAF != 0 || lda < n || batch <= 0 || (lda & 1) || !work) {
        sf_set_error("potrf: n must be a positive multiple of %d, lda >= n and even, workspace required", SF_LEAF);
        return SF_EINVAL;
    }
    static sf_dev_once attr_once;
    SF_CHECK(sf_once_per_device(&attr_once, []() -> int {
        SF_HIP(hipFuncSetAttribute((const void*)k_potrf_dataflow<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SF_DF_LDS_BYTES));
        SF_HIP(hipFuncSetAttribute((const void*)k_potrf_dataflow<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)SF_DF_LDS_BYTES));
        return SF_OK;
    }));
    double* T = work + (size_t)batch * SF_LTB_DOUBLES;
    const int64_t sT = (int64_t)(n + SF_NB) * SF_LDT + SF_TSKEW;  // parked tile + one inverse tile per panel: (n + 64 + 128) rows
    double* Wt2 = T + (size_t)batch * sT;
    const int64_t sW = (int64_t)SF_NB * SF_LDT + SF_TSKEW;
    double* part = Wt2 + 2 * (size_t)batch * sW + 64;
    SF_SHIFT_FRAME();
    const int nt = (n + GT - 1) / GT;
    // counters: in the two inverse-tile buffers of the launch sequences (2 x batch x sW doubles), which this sequence does not use
    int* flags = (int*)Wt2;
    const size_t ndbg = 2 * (16 * SF_CHIP_WGS + 16 * 64 + 8 * 64);
    const size_t nflags = 64 + (size_t)batch * (3 * nt + 2 + 3 * SF_DF_FRONT_MAX) + (size_t)SF_DF_QUEUES * nt + 8 + ndbg;
    // (region 2 of `part` holds 2 x front x batch x pt_cap tiles with pt_cap >= 1: batches beyond what it holds at the widest
    // front are refused here -- the automatic choice stops at 128 matrices, a forced sequence 4 falls back in sf_launch_potrf)
    if (nt - 1 > SF_DF_MAX_STAGES || nflags * sizeof(int) > 2 * (size_t)batch * sW * sizeof(double) ||
        2 * (size_t)SF_DF_FRONT_WIDEST * batch > sf_split_region_tiles()) {
        sf_set_error("potrf: dataflow sequence: %d panels / %d matrices do not fit its tables", nt, batch);
        return SF_EINVAL;
    }
    sf_df_args a = {};
    a.head = flags;  // [SF_DF_QUEUES]
    a.abort_flag = flags + 32;
    a.done_top = flags + 64;
    a.done_D = a.done_top + batch;
    a.chain_next = a.done_D + batch;
    a.fp_cnt = a.chain_next + SF_DF_FRONT_MAX * batch;  // [batch][2][SF_DF_FRONT_MAX]
    a.done_row = a.fp_cnt + 2 * SF_DF_FRONT_MAX * batch;      // [batch][nt]
    a.row_L = a.done_row + (size_t)batch * nt;
    a.rp_cnt = a.row_L + (size_t)batch * nt;
    a.stage_done = a.rp_cnt + (size_t)batch * nt;  // [SF_DF_QUEUES][nt]
    SF_HIP(hipMemsetAsync(flags, 0, nflags * sizeof(int), s));
    SF_HIP(hipMemsetAsync(info, 0, sizeof(int) * (size_t)batch, s));
#ifdef SF_TUNING
    // test aids: a launch that finds its abort flag raised (every matrix comes back SF_INFO_INTERNAL: the callers' recovery
    // path); a dispenser that leaves every chain / front task to the rescue of the waits (sf_df_wait_r)
    if (SF_TUNE_FLAG("SF_DF_FORCE_ABORT")) SF_HIP(hipMemsetAsync(a.abort_flag, 1, 1, s));
    static const int timeout_s = SF_TUNE_INT("SF_DF_TIMEOUT_S", 0);  // (bound of the waits in seconds instead of 4)
    if (timeout_s > 0) SF_HIP(hipMemsetD32Async((hipDeviceptr_t)(a.abort_flag + 4), timeout_s * 95, 1, s));
    // (no-progress bound in ms instead of 25; with SF_DF_TIMEOUT_S alone the stall bound follows it: the experiment that showed
    // the shared-device deadlock to be one -- a launch still stuck after 60 s -- stays reproducible)
    static const int stall_ms = SF_TUNE_INT("SF_DF_STALL_MS", 0);
    const long long stall_units = stall_ms > 0 ? ((long long)stall_ms * 100000) >> 16 : (timeout_s > 0 ? ((long long)timeout_s * 100000000) >> 16 : 0);
    if (stall_units > 0) SF_HIP(hipMemsetD32Async((hipDeviceptr_t)(a.abort_flag + 7), (int)std::min<long long>(stall_units, 0x7fffffff), 1, s));
    a.miss_claims = SF_TUNE_INT("SF_DF_MISS_CLAIMS", 0);
#endif

    sf_panel_args& g = a.p;
    g.C = A;
    g.sC = stride;
    g.lda = lda;
    g.n = n;
    g.rhs = rhs;
    g.ldr = ldr;
    g.fp = fp;
    g.sS = sT;  // (the parked diagonal tile of matrix b: T + b sT, row stride SF_LDT)
    g.ldS = SF_LDT;
#ifdef SF_TUNING
    static const int skip = SF_TUNE_INT("SF_PANEL_SKIP", 0);
    g.skip = skip;
#endif
    if (gen) {
        g.genY = gen->Y - fp;
        g.sY = (int64_t)gen->mpad * gen->ldy;
        g.ldy = gen->ldy;
        g.mpad = gen->mpad;
        g.tilemap = gen->tilemap;
        g.nt128 = gen->nt128;
    }
    // front width: the rows the chain needs next must be a reduce-and-epilogue behind it, and the first ORDINARY slab of a
    // stage (a long-K task, or partial sums + reduce) gets `front` chain periods before the front needs its row.  Front tasks
    // cost more than ordinary ones (partial sums written and read back).  N = 4096, front 1 / 2 / 3 / 4 / 6: B = 16 8.1 / 8.0 /
    // 7.87 / 7.84 / 7.85 ms, B = 32 13.7 / 13.8 / 13.7 / 13.9 / 14.6, B = 64 25.45 / 25.7 / 26.3 / 26.85 / 28.0
    static const int front_env = SF_TUNE_INT("SF_DF_FRONT", 0);
    const int F0 = std::max(1, std::min(SF_DF_FRONT_MAX, front_env > 0 ? front_env : (batch <= 20 ? 3 : 1)));
    // ... and for 21-48 matrices the front widens to three slabs for the last panels: where a stage has fewer tasks than the
    // chip has workgroup slots (batch x slabs left <= 400) AND its K loops are long (2048 columns or more: the front keeps
    // long-K tasks out of the chain's way, its partial sums cost a round trip through memory).  Same-box, wide front from
    // that panel on / never (`tools/knobs_potrf.sh`): N = 4096: B = 24 10.55 / 11.2 ms, 32: 13.42 / 13.65, 40: 16.7 / 16.93,
    // 48: 19.75 / 19.8, 64: 25.7 / 25.5 (not taken from 49 matrices on); N = 3008, B = 32: 6.2 / 6.25; a wide front over the
    // short K loops of N = 2048 loses 3-5 %.
    static const int tail_env = SF_TUNE_INT("SF_DF_TAIL", -1);  // (tuning aid: panels of wide front, 0 = none)
    static const int tailw_env = SF_TUNE_INT("SF_DF_TAIL_FRONT", SF_DF_FRONT_WIDEST);
    const int Ftail = std::max(F0, std::min(SF_DF_FRONT_MAX, tailw_env));
    int kT = nt;  // first panel of the wide front (nt: none)
    if (tail_env >= 0) kT = std::max(0, nt - 1 - tail_env);
    else if (batch > 20 && batch <= 48) kT = std::max(2048 / GT, nt - 400 / batch);
    auto Fof = [&](int k) { return k >= kT ? Ftail : F0; };
    const int F = Ftail;  // (the largest width: strides of the front's partial sums and counters)
    if (2 * (size_t)F * batch > sf_split_region_tiles()) {
        sf_set_error("potrf: dataflow sequence: front %d x %d matrices exceed the partial-sum region", F, batch);
        return SF_EINVAL;
    }
    a.front = F;
    for (int d = 1; d <= SF_DF_FRONT_MAX; ++d) a.fstart[d - 1] = d <= F0 ? 0 : kT;
    static const int fp_pos_env = SF_TUNE_INT("SF_DF_FP_POS", -1);
    a.fp_pos = fp_pos_env >= 0 ? fp_pos_env : 256;  // (0 / 64 / 128 / 192 / 256: B = 16 7.75 / 7.7 / 7.6 / 7.7 / 7.55 ms, B = 32 13.8 / 13.9 / 13.75 / 13.7 / 13.65)
    a.nt = nt;
    a.batch = batch;
    a.T = T;
    a.sT = sT;
    a.part = part;
    a.info = info;
    a.diag = sf_df_diag();
    static const int qbal_env = SF_TUNE_INT("SF_DF_QBAL", 1);
    a.qbal = qbal_env;
#ifdef SF_TUNING
    if (SF_TUNE_FLAG("SF_DF_VERBOSE")) a.dbg = (long long*)(flags + ((nflags - ndbg + 1) & ~(size_t)1));
    static const char* trace_file = SF_TUNE_STR("SF_DF_TRACE_FILE");  // every task's {what, claimed, body start, end} as text
    if (trace_file && a.dbg) {
        a.trace_cap = 1 << 18;
        SF_HIP(hipMalloc((void**)&a.trace, sizeof(long long) * (4 + 4 * (size_t)a.trace_cap)));
        SF_HIP(hipMemsetAsync(a.trace, 0, sizeof(long long) * 4, s));
    }
#endif

    // ---- the task tables: one per queue size (ceil and floor of batch / 8)
    static const int cap_env = SF_TUNE_INT("SF_DF_CAP", 0);
    // workgroup slots a queue can count on: those of one XCD -- of 8 / batch XCDs when there are fewer matrices than queues
    // (bounded by the partial-sum tiles a queue owns)
    auto cap_of = [&](int Bq) {
        if (cap_env > 0) return cap_env;
        if (!a.qbal || batch % SF_DF_QUEUES == 0) return SF_CHIP_WGS / SF_DF_QUEUES;
        return std::max(16, std::min<int>(SF_DF_QTILES, (int)((long long)Bq * SF_CHIP_WGS / batch)));
    };
    // front partial-sum tasks per queue, panel and front slab (64 / 32 / 16 / 8 with a one-slab front: B = 32 14.9 / 14.8 / 14.55 /
    // 14.45 ms, B = 48 20.8 / 20.2 / 20.2 / 20.6)
    static const int pt_tasks = SF_TUNE_INT("SF_DF_PT_TASKS", 16);
    const int kpb = GT / GK;
    const int st_cap = (int)std::min<size_t>(SF_SPLIT_MAX, std::max<size_t>(1, sf_split_region_tiles() / (2 * (size_t)F * (size_t)batch)));
    a.pt_cap = st_cap;
    a.bq[0] = (batch + SF_DF_QUEUES - 1) / SF_DF_QUEUES;
    a.bq[1] = batch / SF_DF_QUEUES;
    static thread_local sf_df_stage tab[2][SF_DF_MAX_STAGES];
    for (int v = 0; v < 2; ++v) {
        const int B = a.bq[v];
        if (B <= 0 || (v == 1 && a.bq[1] == a.bq[0])) {
            a.ntasks[v] = v == 1 ? a.ntasks[0] : 0;
            continue;
        }
        const int cap = cap_of(B);
        int off = 0, thr_pt[2] = {0, 0}, thr_rp = 0, last_split[2] = {-1, -1};
        bool seen[2][SF_DF_FRONT_MAX] = {};
        for (int k = 0; k + 1 < nt; ++k) {
            sf_df_stage& st = tab[v][k];
            st.off = off;
            // FP(., k+1, ., .): K slabs [fp / GK, k 8) of panel k+1 (everything left of panel k)
            const int nF = std::max(0, std::min(Fof(k + 1), nt - 1 - (k + 1)));
            const int nord = std::max(0, nt - k - 1 - Fof(k));
            st.fw = Fof(k) | (nF << 8);
            const int cnt_pt = nF > 0 ? k * kpb - fp / GK : 0;
            st.St = cnt_pt >= 8 ? sf_df_split(B, cnt_pt, st_cap, pt_tasks) : 0;
            for (int d = 1; d <= nF; ++d)  // (a distance that appears with the wide front has missed the arrivals counted so far)
                if (!seen[(k + 1) & 1][d - 1]) {
                    seen[(k + 1) & 1][d - 1] = true;
                    a.thr_base[v][(k + 1) & 1][d - 1] = thr_pt[(k + 1) & 1];
                }
            thr_pt[(k + 1) & 1] += st.St;
            st.thr_pt = thr_pt[(k + 1) & 1];
            const int nk = (k * GT > fp ? k * GT - fp : 0) / GK;
            st.Sr = nord > 0 ? sf_df_split((long long)B * nord, nk, SF_SPLIT_MAX, cap) : 1;
            while (st.Sr > 1 && (size_t)B * nord * st.Sr > SF_DF_QTILES) st.Sr /= 2;
            st.dep = -1;
            if (st.Sr > 1) {
                thr_rp += st.Sr;
                st.dep = last_split[k & 1];
                last_split[k & 1] = k;
            }
            st.thr_rp = thr_rp;
            off += B * nF * st.St + B * nord * (st.Sr > 1 ? st.Sr + 1 : 1);
        }
        a.ntasks[v] = off;
    }
    if (a.bq[1] == a.bq[0]) {
        for (int k = 0; k + 1 < nt; ++k) tab[1][k] = tab[0][k];
        for (int par = 0; par < 2; ++par)
            for (int d = 0; d < SF_DF_FRONT_MAX; ++d) a.thr_base[1][par][d] = a.thr_base[0][par][d];
    }
    for (int v = 0; v < 2; ++v)
        for (int k = 0; k + 1 < nt; ++k) a.st[v][k] = sf_df_pack(tab[v][k]);
    // algorithmic flops (as the launch sequences count them): update, solve, diagonal-tile update of every panel
    double flops = 0.0;
    for (int k = 0; k + 1 < nt; ++k) {
        const int k0 = k * GT, pw = std::min(GT, n - k0);
        const double rows = (double)(n - (k + 1) * GT);
        flops += (2.0 * (k0 > fp ? k0 - fp : 0) * rows * pw + rows * pw * (double)pw + (double)GT * rows * pw) * batch;
    }
    long long total = (long long)batch * nt * F;  // the chain and front tasks
    for (int qx = 0; qx < SF_DF_QUEUES; ++qx) {
        const int B = (batch - qx + SF_DF_QUEUES - 1) / SF_DF_QUEUES;
        if (B > 0) total += a.ntasks[B == a.bq[0] ? 0 : 1];
    }
    static const int grid_env = SF_TUNE_INT("SF_DF_GRID", SF_CHIP_WGS);
    const int grid = (int)std::min<long long>(total, grid_env);
#ifdef SF_TUNING
    if (SF_TUNE_FLAG("SF_DF_VERBOSE")) {
        fprintf(stderr, "dataflow: n=%d nt=%d batch=%d tasks=%lld grid=%d lds=%zu bq=%d/%d St/Sr:", n, nt, batch, total, grid,
                (size_t)SF_DF_LDS_BYTES, a.bq[0], a.bq[1]);
        for (int k = 0; k + 1 < nt; ++k) fprintf(stderr, " %d/%d%s", tab[0][k].St, tab[0][k].Sr, k == kT && Ftail > F0 ? "|" : "");
        fprintf(stderr, "\n");
    }
#endif
    void* tok;
    sf_prof_gemm_begin(s, flops, &tok);
    if (rhs)
        hipLaunchKernelGGL(k_potrf_dataflow<true>, dim3(grid), dim3(512), SF_DF_LDS_BYTES, s, a);
    else
        hipLaunchKernelGGL(k_potrf_dataflow<false>, dim3(grid), dim3(512), SF_DF_LDS_BYTES, s, a);
    g_df_launches.fetch_add(1);
    sf_prof_gemm_end(tok);
    SF_LAUNCH_CHECK();
#ifdef SF_TUNING
    if (SF_TUNE_FLAG("SF_DF_CHECK")) {  // which wait timed out?  (synchronises)
        int ab[4];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(ab, a.abort_flag, sizeof(ab), hipMemcpyDeviceToHost);
        if (ab[0]) {
            const long off = ab[1] + 32;  // offset from `flags`
            const long o_top = 64, o_D = o_top + batch, o_cn = o_D + batch, o_fp = o_cn + SF_DF_FRONT_MAX * batch, o_row = o_fp + 2 * SF_DF_FRONT_MAX * batch,
                       o_L = o_row + (long)batch * nt, o_rp = o_L + (long)batch * nt, o_sd = o_rp + (long)batch * nt;
            const char* what = off >= o_sd ? "stage_done[q][k]" : off >= o_rp ? "rp_cnt[b][i]" : off >= o_L ? "row_L[b][i]" : off >= o_row ? "done_row[b][i]" :
                               off >= o_fp ? "fp_cnt[b][parity][d]" : off >= o_cn ? "chain_next" : off >= o_D ? "done_D[b]" : "done_top[b]";
            const long base = off >= o_sd ? o_sd : off >= o_rp ? o_rp : off >= o_L ? o_L : off >= o_row ? o_row : off >= o_fp ? o_fp : off >= o_cn ? o_cn : off >= o_D ? o_D : o_top;
            const long rel = off - base, per = off >= o_sd ? nt : off >= o_rp ? nt : off >= o_row ? nt : off >= o_fp ? 2 * SF_DF_FRONT_MAX : 1;
            fprintf(stderr, "dataflow ABORTED (n=%d batch=%d front=%d): a wait for %s index %ld / %ld (target %d, value %d) timed out\n", n, batch, F, what,
                    rel / per, rel % per, ab[2], ab[3]);
            std::vector<int> fl(nflags - ndbg);
            (void)hipMemcpy(fl.data(), flags, sizeof(int) * fl.size(), hipMemcpyDeviceToHost);
            for (int bb = 0; bb < batch; ++bb) {
                fprintf(stderr, "  b=%d: done_top %d done_D %d chain_next", bb, fl[o_top + bb], fl[o_D + bb]);
                for (int d = 0; d < F; ++d) fprintf(stderr, " %d", fl[o_cn + SF_DF_FRONT_MAX * bb + d]);
                fprintf(stderr, " | row_L:");
                for (int i = 0; i < nt; ++i) fprintf(stderr, " %d", fl[o_L + (long)bb * nt + i]);
                fprintf(stderr, " | done_row:");
                for (int i = 0; i < nt; ++i) fprintf(stderr, " %d", fl[o_row + (long)bb * nt + i]);
                fprintf(stderr, "\n");
            }
            fprintf(stderr, "  queue heads:");
            for (int qx = 0; qx < SF_DF_QUEUES; ++qx) fprintf(stderr, " %d", fl[qx]);
            fprintf(stderr, " of %d / %d tasks\n", a.ntasks[0], a.ntasks[1]);
        }
    }
    if (a.dbg) {
        static long long host[16 * SF_CHIP_WGS + 16 * 64 + 8 * 64];
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(host, a.dbg, sizeof(host), hipMemcpyDeviceToHost);
        double w = 0, bd = 0, nn = 0, ty[6] = {0, 0, 0, 0, 0, 0}, tw[6] = {0, 0, 0, 0, 0, 0};
        long long wmax = 0, bmax = 0;
        for (int i = 0; i < grid; ++i) {
            w += host[16 * i];
            bd += host[16 * i + 1];
            nn += host[16 * i + 2];
            for (int j = 0; j < 6; ++j) ty[j] += host[16 * i + 3 + j];
            for (int j = 0; j < 6; ++j) tw[j] += host[16 * i + 9 + j];
            wmax = std::max(wmax, host[16 * i]);
            bmax = std::max(bmax, host[16 * i] + host[16 * i + 1]);
        }
        if (SF_TUNE_FLAG("SF_DF_TRACE")) {
            const long long* tr = host + 16 * SF_CHIP_WGS;
            long long t0 = tr[0];
            fprintf(stderr, "matrix 0, us since its first task: k | C claim start end | FP(k,1) claim start end | FP(k,2) claim start end | FR(k,2) claim start end\n");
            for (int k = 1; k < nt && k < 64; ++k) {  // (k = 0 has no panel part)
                fprintf(stderr, "%2d |", k);
                for (int j = 0; j < 12; ++j) fprintf(stderr, "%s%8.1f", j % 3 == 0 && j ? " |" : "", tr[16 * k + j] ? (tr[16 * k + j] - t0) / 100.0 : 0.0);
                const long long* st4 = host + 16 * SF_CHIP_WGS + 16 * 64 + 8 * k;
                fprintf(stderr, " | C: reduce %6.1f + tail %6.1f", (st4[5] - tr[16 * k + 1]) / 100.0, (st4[0] - st4[5]) / 100.0);
                fprintf(stderr, " | C: K work %6.1f, wait D %6.1f, solve+store %6.1f, wait S %6.1f, step 4 %6.1f, D %6.1f\n",
                        (st4[0] - tr[16 * k + 1]) / 100.0, (st4[1] - st4[0]) / 100.0, (st4[2] - st4[1]) / 100.0, (st4[3] - st4[2]) / 100.0,
                        (tr[16 * k + 12] - st4[3]) / 100.0, (tr[16 * k + 2] - tr[16 * k + 12]) / 100.0);
            }
        }
        fprintf(stderr, "dataflow per workgroup: waiting %.2f ms (max %.2f), bodies %.2f ms, busy max %.2f ms, %.0f tasks; by type C %.2f FP %.2f FR %.2f R %.2f RP %.2f RR %.2f ms\n",
                w / grid / 1e5, wmax / 1e5, bd / grid / 1e5, bmax / 1e5, nn / grid, ty[0] / grid / 1e5, ty[1] / grid / 1e5, ty[2] / grid / 1e5,
                ty[3] / grid / 1e5, ty[4] / grid / 1e5, ty[5] / grid / 1e5);
        if (a.trace) {
            std::vector<long long> tr(4 + 4 * (size_t)a.trace_cap);
            (void)hipMemcpy(tr.data(), a.trace, sizeof(long long) * tr.size(), hipMemcpyDeviceToHost);
            (void)hipFree(a.trace);
            if (FILE* f = fopen(trace_file, "w")) {
                const long long nrec = std::min<long long>(tr[0], a.trace_cap);
                fprintf(f, "# n=%d batch=%d nt=%d front=%d grid=%d: type(C FP FR R RP RR) k i b workgroup claimed start end (10 ns ticks)\n", n, batch, nt, F, grid);
                for (long long r = 0; r < nrec; ++r) {
                    const long long* e = &tr[4 + 4 * r];
                    fprintf(f, "%lld %lld %lld %lld %lld %lld %lld %lld\n", e[0] & 255, (e[0] >> 8) & 255, (e[0] >> 16) & 255, (e[0] >> 24) & 65535, e[0] >> 40, e[1], e[2], e[3]);
                }
                fclose(f);
            }
        }
        fprintf(stderr, "dataflow waiting by type: C %.2f FP %.2f FR %.2f R %.2f RP %.2f RR %.2f ms\n", tw[0] / grid / 1e5, tw[1] / grid / 1e5,
                tw[2] / grid / 1e5, tw[3] / grid / 1e5, tw[4] / grid / 1e5, tw[5] / grid / 1e5);
    }
#endif
    return SF_OK;
}

// Small batches are bound by the number of sequential long-K steps; the unfused sequence has half as many (256-column
// panels).  Measured at N = 4096, fused (partial sums of top(k) beside D(k), chain at raised wave priority) / unfused:
// B = 12: 10.0 / 9.05-9.6 ms, 16: 10.97 / 11.1, 20: 12.2 / 12.6, 24: 13.3 / 13.9, 32: 15.4 / 16.9, 64: 26.1 / 29.5.
#define SF_UNFUSED_BELOW 16
// The dataflow sequence (one persistent launch) wins wherever the launch sequences cannot keep the chip full between their
// panel boundaries.  Measured (tools/bench_potrf.py, same box, launch sequences (fused; wide where it is their choice) /
// dataflow): N = 4096: B = 8 7.4 / 5.1 ms, 16: 9.6 / 7.85, 32: 14.2 / 13.7, 48: 20.0 / 19.9, 64: 25.9 / 25.45, 80: 31.9 / 32.1,
// 96: 37.5 / 38.3, 112: 42.4 / 44.4, 128: 47.1 / 50.7; N = 3008: B = 16 6.15 / 4.7, 64: 11.8 / 11.5, 96: 16.7 / 16.9; N = 2048:
// B = 16 3.36 / 2.5, 128: 8.15 / 8.1; N = 1024: B = 16 1.29 / 0.88, 256: 2.8 / 3.2 (32 matrices per queue: the dispenser's scan
// of their chain counters shows)  ->  taken while batch x panels <= 2048 and batch <= 128.
// sf_persistent_potrf(0): the callers' recovery after a launch that came back SF_INFO_INTERNAL -- from then on every
// factorisation of the process takes a launch sequence (no waits inside kernels), forced sequence 4 included.
static std::atomic<int> g_df_enabled{1};
int sf_set_persistent_potrf(int enable) {
    return enable < 0 ? g_df_enabled.load() : g_df_enabled.exchange(enable ? 1 : 0);
}
static int g_df_enabled_query(void) { return g_df_enabled.load(); }
static bool sf_potrf_dataflow_fits(int n, int batch) {
    // panels: n is a multiple of 64; an order of 64 mod 128 rows has (n + 64) / 128 of them in either frame (shifted by 64
    // virtual rows, sf_potrf_front_pad, or not) -- N = 16384 is 128 panels = 127 stages, the tables' limit (the round-5 check
    // added the 64 rows unconditionally: 129 panels, so N = 16384 never took the persistent kernel, forced or not)
    const int nt = (n + GT - 1) / GT;
    return g_df_enabled.load() && nt - 1 <= SF_DF_MAX_STAGES && 2 * (size_t)SF_DF_FRONT_WIDEST * batch <= sf_split_region_tiles();
}
static bool sf_potrf_dataflow_auto(int n, int batch) {
    static const int lim = SF_TUNE_INT("SF_DF_BELOW", 2048);
    // (round 6: the panel count is the true one -- N = 4096: up to 64 matrices, the half-ensemble of a 128-walker sampler; same
    // box, persistent kernel / fused sequence there: 25.5 / 26.0 ms.  Rounds 4-5 counted 64 virtual rows more: 62 matrices.)
    const int nt = (n + GT - 1) / GT;
    // (... and stops at N = 8192: the kernel FITS up to N = 16384 -- forced sequence 4, tests -- but was only ever measured to
    // win up to 65 panels; at N = 16384 the tasks are milliseconds long and the launch sequences keep the chip as full:
    // profiles/r06_a_dataflow_n16384.txt)
    static const int nt_max = SF_TUNE_INT("SF_DF_NT_MAX", 65);
    return sf_potrf_dataflow_fits(n, batch) && (long long)batch * nt <= lim && batch <= 128 && nt <= nt_max;
}
int sf_potrf_front_pad(int n, int batch) {
    static const bool off = SF_TUNE_FLAG("SF_NO_FRONT_PAD");  // tuning aid: A/B of the shifted frame
    static const char* force = SF_TUNE_STR("SF_CHOL_UNFUSED");
    const int sel = g_chol_sequence.load();
    // (a matrix with more panels than the dataflow tables hold takes the fused sequence when the dataflow one is forced)
    const bool df_fits = sf_potrf_dataflow_fits(n, batch);
    const bool df = sel >= 0 ? (sel == 4 && df_fits) : (!force && sf_potrf_dataflow_auto(n, batch));
    const bool v1 = !df && (sel >= 0 ? sel == 1 : (force ? force[0] == '1' : batch < SF_UNFUSED_BELOW));  // (as in sf_launch_potrf)
    if (off || v1 || n % GT != 64 || n < 2 * GT) return 0;
    return 64;
}

int sf_launch_potrf(double* A, int n, int lda, int64_t stride, int batch, int* info, double* work,
                    double* rhs, int ldr, hipStream_t s, const sf_gen_args* gen, sf_exec* ex) {
    // The fused panel kernel (128-column panels) is the faster sequence once the batch fills the chip (SF_UNFUSED_BELOW).
    static const char* force = SF_TUNE_STR("SF_CHOL_UNFUSED");  // tuning aid: "1" always unfused, "0" always fused
    const int sel = g_chol_sequence.load();                 // sf_debug_cholesky_sequence(): tests drive both
    const bool df_fits = sf_potrf_dataflow_fits(n, batch);
    const bool df = sel >= 0 ? (sel == 4 && df_fits) : (!force && sf_potrf_dataflow_auto(n, batch));
    const bool v1 = !df && (sel >= 0 ? sel == 1 : (force ? force[0] == '1' : batch < SF_UNFUSED_BELOW));
    // The wide sequence (panel pairs, one 16-wave workgroup per CU) halves the A-operand stream and a third of all HBM
    // traffic of the factorisation, but one workgroup per CU has nothing to overlap its epilogue and barriers with: it pays
    // once its launches are many rounds of workgroups.  Measured (bench.py, same box, fused / wide): N = 4096: B = 48
    // 21.45 / 23.1 ms, 64: 27.34 / 27.4, 80: 33.0 / 33.4, 96: 38.2 / 38.6, 112: 44.6 / 43.9, 128: 49.7 / 48.4; N = 16384,
    // B = 32 (cfg 5): 707.7 / 696.6; 1600 units of N = 3008 (cfg 3): 277.3 / 273.2 -> taken from batch x slabs >= 3400.
    const bool wide_auto = n >= 2048 && (long long)batch * ((n + GT - 1) / GT) >= 3400;
    const bool v3 = !df && (sel >= 0 ? sel == 2 : (force ? force[0] == '2' : wide_auto));
    if (!ex) ex = sf_exec_thread_local();
    static const int tail_env = SF_TUNE_INT("SF_WIDE_TAIL_ROUNDS", -1);  // measured at cfg 2: -1 (wide to the end) 49.3 ms, 2: 50.0, 5: 50.4, 8: 51.0, 12: 51.8 (narrow: 51.5)
    const bool v3h = sel == 3;  // (test aid) wide pairs for the first half of the panels, narrow steps after
    // frame of the fused sequences: the caller's (whose tile map was built in it) or this call's own
    const int fp = gen ? gen->fp : sf_potrf_front_pad(n, batch);
    if (df) return sf_launch_potrf_v4(A, n, lda, stride, batch, info, work, rhs, ldr, s, gen, fp);
    if (v3 || v3h)
        return sf_launch_potrf_v3(A, n, lda, stride, batch, info, work, rhs, ldr, s, gen, ex, v3h ? -2 : (sel == 2 ? -1 : tail_env), fp);
    if (v1 && fp == 0) return sf_launch_potrf_v1(A, n, lda, stride, batch, info, work, rhs, ldr, s, gen, ex);
    return sf_launch_potrf_v2(A, n, lda, stride, batch, info, work, rhs, ldr, s, gen, ex, fp);
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
        static sf_dev_once attr_once;  // devices whose function attributes are set
        SF_CHECK(sf_once_per_device(&attr_once, []() -> int {
            SF_HIP(hipFuncSetAttribute((const void*)k_trsv_logdet<false>,
            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            return SF_OK;
        }));
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

// Debug aid for the tuning scripts: one wave spins for `wall_ticks` ticks of the 100 MHz wall clock and
// reports how many shader-clock ticks (s_memtime) elapsed -> sustained shader clock while other
// streams are busy.  out[0] = s_memtime ticks, out[1] = wall ticks.
__global__ void k_clock_probe(long long* out, long long wall_ticks) {
    const long long w0 = wall_clock64();
    const long long t0 = __builtin_amdgcn_s_memtime();
    long long w1 = w0;
    while (w1 - w0 < wall_ticks) {
        __builtin_amdgcn_s_sleep(32);
        w1 = wall_clock64();
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) {
        out[0] = t1 - t0;
        out[1] = w1 - w0;
    }
}
int sf_launch_clock_probe(long long* out, long long wall_ticks, hipStream_t s) {
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, s, out, wall_ticks);
    SF_LAUNCH_CHECK();
    return SF_OK;
}
