// Structure-exploiting solver for the likelihood:
//
//     C  =  Bd  +  Y^T Y ,      Bd = diag(sigma^2) + K_global + sum K_local + 1e-10 I
//
// The reference builds C dense (Starfish/models/spectrum_model.py:334-363) although K_global and
// K_local are compactly supported (the `r <= r0` masks of Starfish/models/kernels.py:33,78), so Bd is
// a band matrix of half-width W pixels (24 ls/dv for the global kernel, whose metric is a quarter of
// the velocity separation) and Y^T Y = X^T Sigma_w^-1 X has rank m.  With
//     Bd = L L^T (banded),  z = L^-1 R,  Z = L^-1 Y^T,   S = I_m + Z^T Z = L_S L_S^T
//     logdet C = logdet Bd + logdet S
//     R^T C^-1 R = z^T z - | L_S^-1 Z^T z |^2                        (Woodbury)
// only forward substitutions are needed and the cost drops from N^3/3 to O(N W^2) flops per walker.
//
// k_band_forms: one workgroup per matrix sweeps the band in 16-column steps over a sliding window of
// (W+16) rows held in LDS.  The window is stored as 16 x 16 blocks addressed by the UNORDERED pair of
// ring slots {row block % nbr, column block % nbr}: every live block of the lower triangle has its own
// slot, nothing is ever shifted, and the footprint is nbr(nbr+1)/2 blocks instead of nbr^2.
// Everything runs on v_mfma_f64_16x16x4_f64.  Per step k (columns 16k .. 16k+15):
//   wave 0   runs the sequential chain by itself, free of barriers: it factorises the diagonal block k AND
//            inverts its factor in the MFMA accumulator layout (one rank-1 MFMA per column each, pivots from
//            scalars so that the rsqrt chain overlaps the matrix core), publishes F_k = L_kk^-1, solves the
//            next sub-diagonal block X_{k+1,k} = A_{k+1,k} F_k^T, applies it to the next diagonal block and
//            goes on with k+1;
//   waves 1+ ("workers") trail by one column: they solve the rest of block column k as products with F_k,
//            apply column k to the window (software pipelined: operands of the next block are read under
//            the MFMAs of the current one), fetch the next 16 band rows / right-hand-side columns from HBM
//            into registers and drop them into the slots column k retires.
// LDS flags order the two sides; the workers meet at two LDS-counter barriers per column.  Block addresses
// depend on the ring phase k % nbr only: each wave tabulates its operations once (lane = phase) and reads
// them back with one v_readlane per operation.
// The right-hand sides (residual + the m rows of Y) ride along as extra ROWS of the matrix, so their
// forward substitution is the same solve/update, and their Gram matrix [z Z]^T [z Z] accumulates in
// LDS.  L is never written anywhere: the outputs are logdet(Bd) and the (1+m)^2 Gram matrix.
// k_woodbury then does the m x m capacitance solve.
// Measured on MI355X (cfg 2, W = 141 px, 16 waves): 1.27 ms for 128 matrices = one CU each (11.5k cycles
// per column against 4.7k of MFMA time: 296 MFMAs x 64 cycles over four SIMDs).  While 2*batch workgroups
// fit the chip the sweep is therefore "twisted": one workgroup eliminates the first half of the columns
// top-down, a second one the last half bottom-up (same code on the index-reversed matrix), both dump what
// is left of a (W rounded up to 16)-row separator, k_band_merge adds the two Schur contributions and a
// short third sweep finishes the separator: 0.74 ms.
#include "sf_common.h"
#include <type_traits>

#define BB 16
#define BS (BB * (BB + 1))  // doubles per stored block (row stride 17: conflict-free fragment reads)
#define BLD (BB + 1)
#define SFB_PF 4            // prefetch registers per thread
#define SFB_PT 8            // update pairs per wave (table row; the last entry is the count)

struct sf_band_args {
    const double* band;  // [batch] x sband : band[i*ldb + d] = A[i][i-d], d in [0, halfwidth]
    int64_t sband;
    int ldb;
    int halfwidth;
    const double* rhs;   // [batch] x srhs : row r of the right-hand sides at rhs + r*ldr (length n)
    int64_t srhs;
    int ldr;
    int nrhs;
    // optional separate source for row 0 (the residual lives in its own buffer in the fused path);
    // when given, rows 1.. come from rhs rows 0..
    const double* rhs0;
    int64_t srhs0;
    int n;      // matrix order
    int nbr;    // window blocks per side
    double* logdet;  // [batch]
    double* gram;    // [batch][nrhs*nrhs]
    int* info;       // [batch] first non-positive pivot (1-based), left untouched otherwise
    // optional additive terms of a final pass over a merged separator block (twisted factorisation)
    const double* logdet_add;  // [batch]
    const double* gram_add;    // [batch][nrhs*nrhs]
    int info_off;              // added to the reported pivot column (separator pass)
    // Two-sided ("twisted") sweep: grid = 2 * batch, workgroup 2b eliminates block columns [0, kend0)
    // top-down, workgroup 2b+1 eliminates the last kend1 block columns bottom-up (the same algorithm on
    // the index-reversed matrix); both stop at the nm-row separator and dump what is left of it.
    int nhalf;                 // 1 = ordinary single sweep
    int kend0, kend1;
    double* dumpM;             // [batch][2][nm*nm]   separator block, lower triangle, local row-major
    double* dumpR;             // [batch][2][NR*nm]   right-hand-side rows over the separator columns
    double* dumpG;             // [batch][2][nrhs*nrhs] partial Gram matrices
    double* dumpL;             // [batch][2]          partial log-determinants
};

__device__ __forceinline__ double sfb_readlane(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// 1/sqrt(p): hardware estimate + two Newton steps (full double precision for p > 0)
__device__ __forceinline__ double sfb_rsqrt(double p) {
    double y = __builtin_amdgcn_rsq(p);
    const double h = 0.5 * p;
    double e = __builtin_fma(-h * y, y, 0.5);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-h * y, y, 0.5);
    y = __builtin_fma(y, e, y);
    return y;
}

// Workgroup barrier for LDS data only: the HBM prefetch loads stay in flight across it (the compiler
// waits for them where their registers are first used).
__device__ __forceinline__ void sfb_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ int sfb_pair(int a, int b) {  // slot of the unordered pair {a, b}
    return a >= b ? a * (a + 1) / 2 + b : b * (b + 1) / 2 + a;
}

// One element of block row gb for the initial window fill (generic, off the critical path).
// e < nbr*256: band blocks (gb, gb-nbr+1 .. gb) row-major inside each block; then NR*16 rhs entries.
// `rev` >= 0: the sweep runs on the index-reversed matrix, i -> rev - i (rev = padded order - 1).
__device__ __noinline__ double sfb_fetch(const double* band, const double* rhs, const double* rhs0, int gb,
                                         int e, int nbr, int n, int halfwidth, int ldb, int nrhs, int ldr,
                                         int rev) {
    const int row_elems = nbr * BB * BB;
    if (e < row_elems) {
        const int blk = e >> 8, r = (e >> 4) & 15, cc = e & 15;
        const int gc = gb - nbr + 1 + blk;
        const int i = gb * BB + r, d = i - (gc * BB + cc);
        if (gc < 0 || d < 0) return 0.0;
        const int io = rev >= 0 ? rev - i + d : i;  // larger original index of the pair
        if (io < n) return d <= halfwidth ? band[(int64_t)io * ldb + d] : 0.0;
        return d == 0 ? 1.0 : 0.0;  // identity padding up to a multiple of 16
    }
    const int q = e - row_elems, r = q >> 4, cc = q & 15;
    const int i = gb * BB + cc;
    const int io = rev >= 0 ? rev - i : i;
    if (io >= n || io < 0 || r >= nrhs) return 0.0;
    if (rhs0) return r == 0 ? rhs0[io] : rhs[(int64_t)(r - 1) * ldr + io];
    return rhs[(int64_t)r * ldr + io];
}

// Thread roles (blockDim = 512, 768 or 1024):
//   wave 0       diagonal-block factorisation + inverse (S3), its share of the MFMA work elsewhere
//   waves 1..3   prefetch of the right-hand-side columns, log(pivot), MFMA work
//   waves 4..    prefetch of the band rows (groups of 256 threads = one 16 x 16 block each), MFMA work
template <int NRB>
__global__ __launch_bounds__(1024) void k_band_forms(sf_band_args a) {
    extern __shared__ double lds[];
    const int b = a.nhalf == 2 ? blockIdx.x >> 1 : blockIdx.x;
    const int half = a.nhalf == 2 ? blockIdx.x & 1 : 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nthreads = blockDim.x, nwaves = nthreads >> 6;
    const int nbr = a.nbr;
    constexpr int NR = NRB * BB, LDG = NR + 1;
    double* Wb = lds;                                  // nbr(nbr+1)/2 band blocks
    double* RH = Wb + (nbr * (nbr + 1) / 2) * BS;      // [NRB][nbr] right-hand-side blocks (column ring)
    double* Fb2 = RH + NRB * nbr * BS;                 // [2] inverse of the diagonal factor, L_kk^-1 (by parity of k)
    double* G = Fb2 + 2 * BS;                          // [NR][LDG]   Gram accumulator
    double* pv = G + NR * LDG;                         // [2][16] pivots (for the log-determinant)
    double* red = pv + 2 * BB;                         // [16] reduction scratch
    typedef __attribute__((address_space(3))) int lds_int_t;  // ds_read / ds_add, not flat accesses
    // progress flags of the software pipeline (all monotonic):
    //   [0] F_k published (k+1)          [1] waves that placed block 0 of the newest row (4 per step)
    //   [2] X_{k+1,k} published (k+1)    [3] wave 0 done with column k (k+1)
    //   [4] workers done with column k (k+1)   [5] worker barrier arrivals   [7] a spin timed out
    volatile lds_int_t* sync = (volatile lds_int_t*)(red + BB);
    volatile lds_int_t* ptab = sync + 16;              // [16 waves][SFB_PT] (I << 8 | J) update pairs, [..][SFB_PT-1] = count

    const int n = a.n, W = a.halfwidth;
    const int nblk = (n + BB - 1) / BB;
    const int rev = half ? nblk * BB - 1 : -1;                            // reversed indexing (bottom-up half)
    const int kend = a.nhalf == 2 ? (half ? a.kend1 : a.kend0) : nblk;   // block columns to eliminate
    const double* __restrict__ band = a.band + (int64_t)b * a.sband;
    const double* __restrict__ rhs = a.rhs + (int64_t)b * a.srhs;
    const double* __restrict__ rhs0 = a.rhs0 ? a.rhs0 + (int64_t)b * a.srhs0 : nullptr;
    const int row_elems = nbr * BB * BB, all_elems = row_elems + NR * BB;
    auto wrap = [&](int slot) { return slot >= nbr ? slot - nbr : slot; };  // slot in [0, 2 nbr)

    // ---- initial window: block rows 0 .. nbr-1
    for (int gb = 0; gb < nbr; ++gb)
        for (int e = tid; e < all_elems; e += nthreads) {
            const double v = sfb_fetch(band, rhs, rhs0, gb, e, nbr, n, W, a.ldb, a.nrhs, a.ldr, rev);
            if (e < row_elems) {
                const int blk = e >> 8, gc = gb - nbr + 1 + blk;
                if (gc >= 0) Wb[sfb_pair(gb, gc) * BS + ((e >> 4) & 15) * BLD + (e & 15)] = v;
            } else {
                const int q = e - row_elems, r = q >> 4;
                RH[((r >> 4) * nbr + gb) * BS + (r & 15) * BLD + (q & 15)] = v;
            }
        }
    for (int e = tid; e < NR * LDG; e += nthreads) G[e] = 0.0;
    if (tid < 16) sync[tid] = 0;

    const int l15 = lane & 15, lq = lane >> 4;
    // Wave 0: Cholesky of the 16 x 16 diagonal block AND the inverse of its factor, both kept in the
    // MFMA accumulator layout (lane (lq, l15), register r <-> element (lq + 4r, l15)).  Column j of the
    // symmetric block is also its row j = register j/4 of the 16 lanes of quarter j%4, which is exactly
    // where a K-slice of the MFMA operands lives: the rank-1 elimination  A -= v v^T  and the update of
    // F = L^-1 (F -= v g^T) are one v_mfma_f64_16x16x4_f64 each, with no data movement at all.
    int bad = 0;
    auto potrf16 = [&](int kb, int ks) {  // ks = kb % nbr
        const double* D = Wb + sfb_pair(ks, ks) * BS;
        sf_d4 acc, f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = lq + 4 * r;
            acc[r] = D[max(row, l15) * BLD + min(row, l15)];  // only the lower triangle is maintained
            f[r] = row == l15 ? 1.0 : 0.0;
        }
        double p = sfb_readlane(acc[0], 0);
        double pkeep = 1.0;  // lane j keeps pivot j: one LDS store and one sign test after the loop
#pragma unroll
        for (int j = 0; j < BB; ++j) {
            const int qj = j & 3, rj = j >> 2;
            pkeep = lane == j ? p : pkeep;
            const double rs = sfb_rsqrt(p);
            const bool in_q = lq == qj;
            const double v = (in_q && l15 > j) ? acc[rj] * rs : 0.0;  // l_ij, i = l15 > j
            const double g = in_q ? f[rj] * rs : 0.0;                 // row j of F, scaled
            if (in_q) f[rj] = g;
            if (j + 1 < BB) {
                // next pivot a_{j+1,j+1} - l_{j+1,j}^2 from scalars, so that its rsqrt chain runs while
                // the matrix core applies this column's rank-1 update
                const double an = sfb_readlane(acc[(j + 1) >> 2], ((j + 1) & 3) * 16 + j + 1);
                const double vn = sfb_readlane(v, qj * 16 + j + 1);
                p = __builtin_fma(-vn, vn, an);
            }
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(v, v, acc, 0, 0, 1);  // blgp = neg:[1,0,0]: -A B + C
            f = __builtin_amdgcn_mfma_f64_16x16x4f64(v, g, f, 0, 0, 1);
        }
        if (lane < BB) pv[(kb & 1) * BB + lane] = pkeep;
        const unsigned long long neg = __ballot(lane < BB && !(pkeep > 0.0));
        if (neg && !bad) bad = kb * BB + __ffsll((long long)neg);
#pragma unroll
        for (int r = 0; r < 4; ++r) Fb2[(kb & 1) * BS + (lq + 4 * r) * BLD + l15] = f[r];
    };

    // Prefetch roles, fixed per thread: waves >= 4 load the band (groups of 256 threads = one 16 x 16
    // block each, element (pr, pc) of it), waves 1..3 the right-hand-side columns.  Per slot: a running
    // source pointer (one block row further every step) and the LDS offset of the element inside its
    // destination block; whether a slot loads at all is a per-thread constant, whether its row is still
    // inside the matrix is one comparison per step (top-down sweeps only).
    const int ngroups = (nwaves - 4) >> 2;             // 256-thread groups of band loaders (1..3)
    const int pg = wave >= 4 ? (wave - 4) >> 2 : 0;    // this wave's group (uniform)
    const bool band_loader = wave >= 4;
    typedef const double __attribute__((address_space(1))) * gptr_t;  // keeps the prefetch on global_load
    gptr_t pptr[SFB_PF];
    int poff[SFB_PF];
    int pmask = 0;                                     // bit q: slot q loads; bit 4+q: identity padding when out of range
    int prow;                                          // padded-matrix row of this thread's elements (next block row)
    const int pstep = (rev >= 0 ? -BB : BB) * (band_loader ? a.ldb : 1);  // doubles per step
    {
        const int lt = tid - 256, pr = (lt >> 4) & 15, pc = lt & 15;
        const int rt = tid - 64;
        prow = nbr * BB + (band_loader ? pr : (rt & 15));
#pragma unroll
        for (int q = 0; q < SFB_PF; ++q) {
            if (band_loader) {
                const int blk = pg + q * ngroups;
                const int d = (nbr - 1 - blk) * BB + pr - pc;
                const int io = rev >= 0 ? rev - prow + d : prow;  // larger original index of the pair
                const bool okd = blk < nbr && d >= 0 && d <= W;
                pmask |= (okd ? 1 : 0) << q;
                pmask |= ((blk < nbr && d == 0) ? 16 : 0) << q;
                pptr[q] = (gptr_t)(band + (int64_t)io * a.ldb + (okd ? d : 0));
                poff[q] = pr * BLD + pc;
            } else {
                const int e = rt + q * 192, r = e >> 4;
                const int io = rev >= 0 ? rev - prow : prow;
                const bool okr = wave >= 1 && e < NR * BB && r < a.nrhs;
                pmask |= (okr ? 1 : 0) << q;
                const int rr = rhs0 ? (r > 0 ? r - 1 : 0) : r;
                pptr[q] = (gptr_t)(((rhs0 && r == 0) ? rhs0 : rhs + (int64_t)rr * a.ldr) + io);
                poff[q] = (wave >= 1 && e < NR * BB) ? (e >> 8) * nbr * BS + ((e >> 4) & 15) * BLD + (e & 15) : -1;
            }
        }
    }
    double pf[SFB_PF];                                 // prefetched values (band or rhs), final when they land
    double ld_acc = 0.0;                               // wave `logwave`, lanes 0..15: partial sums of log(pivot)
    __syncthreads();

    const int nb1 = nbr - 1, RB = nb1 + NRB;
    auto spin_until = [&](int which, int target) {
        int guard = 0;
        while (sync[which] < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++guard > (1 << 22)) {  // cannot happen; keeps a logic error from hanging the GPU
                sync[7] = 1;
                break;
            }
        }
        asm volatile("" ::: "memory");
    };
    auto publish = [&](int which, int value) {  // after this wave's LDS writes have landed
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) sync[which] = value;
    };
    // ---- block addressing.  Every 16 x 16 block lives at lds + index * BS: window blocks by ring-slot pair,
    //      then the right-hand-side ring (index NBW + rb * nbr + slot).  Which blocks an operation touches
    //      depends on the ring phase ks = k % nbr only, so each wave tabulates its operations ONCE, lane ks
    //      of a VGPR holding the packed block indices for that phase; per step one v_readlane replaces
    //      the scalar index arithmetic (pair(), wrap(), the band/rhs/Gram case split), which otherwise
    //      costs more issue time than the MFMAs it feeds.
    const int NBW = nbr * (nbr + 1) / 2;
    auto idxP = [&](int ks_, int I) {  // row block I below the diagonal block of column ks_
        return I < nb1 ? sfb_pair(wrap(ks_ + 1 + I), ks_) : NBW + (I - nb1) * nbr + ks_;
    };
    auto pack_update = [&](int ks_, int I, int J) {  // A | B << 8 | C << 16 | gram << 24
        int c, gram = 0;
        if (J >= nb1) {
            c = (I - nb1) * 4 + (J - nb1);
            gram = 1;
        } else if (I < nb1) {
            c = sfb_pair(wrap(ks_ + 1 + I), wrap(ks_ + 1 + J));
        } else {
            c = NBW + (I - nb1) * nbr + wrap(ks_ + 1 + J);
        }
        return idxP(ks_, I) | (idxP(ks_, J) << 8) | (c << 16) | (gram << 24);
    };
    const int tl = lane < nbr ? lane : 0;  // ring phase this lane tabulates
    const int oAB = l15 * BLD + 4 * lq;   // operand fragment (K slice lq = columns 4 lq .. 4 lq + 3 of row l15)
    // X = P F^T in place (B operand = rows of F = L_kk^-1; K slice lq of MFMA kk stands for k = 4 lq + kk
    // in both operands: contiguous, paired LDS reads)
    auto xsolve = [&](int pidx, const double* Fb) {
        double* P = lds + pidx * BS;
        sf_d4 acc = {0.0, 0.0, 0.0, 0.0};
        const double* Ap = P + oAB;
        const double* Bp = Fb + oAB;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(Ap[kk], Bp[kk], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) P[(lq + 4 * r) * BLD + l15] = acc[r];
    };
    // Trailing update  C -= P_I P_J^T  of block (I, J), I >= J, of [band rows below ; rhs rows] x [same]
    // (Gram blocks accumulate with the same sign: G = -Z Z^T, negated on output), split into its operand
    // reads and its MFMAs + write-back so that the next block's operands can be read under the MFMAs.
    auto load_ops = [&](int w, sf_d4& c, sf_d4& av, sf_d4& bv, int& ldc) -> double* {
        const double* Ap = lds + (w & 255) * BS + oAB;
        const double* Bp = lds + ((w >> 8) & 255) * BS + oAB;
        const int ci = (w >> 16) & 255;
        double* Cb;
        if (w >> 24) {
            ldc = LDG;
            Cb = G + ((ci >> 2) * BB) * LDG + (ci & 3) * BB + lq * LDG + l15;
        } else {
            ldc = BLD;
            Cb = lds + ci * BS + lq * BLD + l15;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r] = Cb[4 * r * ldc];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            av[kk] = Ap[kk];
            bv[kk] = Bp[kk];
        }
        return Cb;
    };
    auto mma_store = [&](double* Cb, int ldc, sf_d4 c, const sf_d4& av, const sf_d4& bv) {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) c = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bv[kk], c, 0, 0, 1);  // neg:[1,0,0]
#pragma unroll
        for (int r = 0; r < 4; ++r) Cb[4 * r * ldc] = c[r];
    };

    // ---- division of labour.  Wave 0 runs the sequential chain by itself and never meets a barrier:
    //          potrf(k) -> F_k | X_{k+1,k} = A_{k+1,k} F_k^T | D_{k+1} -= X X^T | potrf(k+1) ...
    //      The other waves ("workers") trail by one column: they solve the rest of block column k against
    //      F_k, apply column k to the window (every block pair except (0,0)), fetch the next 16 rows from
    //      HBM and place them into the slots column k retires.  Flags in LDS order the two.
    const int nworkers = nwaves - 1;
    const int logwave = nwaves - 4;                    // accumulates log(pivot): no solves, the smallest share of updates
    const int nprim = nwaves - (nwaves >> 2);          // workers that do not share wave 0's SIMD
    // position of this wave among the workers, those on wave 0's SIMD last
    const int wpos = wave == 0 ? -1 : (wave & 3) ? wave - 1 - (wave >> 2) : nprim + (wave >> 2) - 1;
    if (tid >= 1 && tid < nwaves) {
        // pair list of wave `tid`: round robin over the workers, those on wave 0's SIMD last (the
        // remainder goes to the others)
        const int w = tid, wp = (w & 3) ? w - 1 - (w >> 2) : nprim + (w >> 2) - 1;
        const int npairs = RB * (RB + 1) / 2;
        int cnt = 0;
        for (int t = 1 + wp; t < npairs; t += nworkers) {
            int I = 0;
            while ((I + 1) * (I + 2) / 2 <= t) ++I;
            if (cnt < SFB_PT - 1) ptab[w * SFB_PT + cnt] = (I << 8) | (t - I * (I + 1) / 2);
            ++cnt;
        }
        ptab[w * SFB_PT + SFB_PT - 1] = cnt < SFB_PT ? cnt : SFB_PT - 1;
        if (cnt >= SFB_PT) sync[7] = 1;  // cannot happen for the window sizes the launcher admits
    }
    __syncthreads();

    if (wave == 0) {
        // lane ks: P0 = block (k+1, k) | D_{k+1} << 8
        const int tab0 = idxP(tl, 0) | (sfb_pair(wrap(tl + 1), wrap(tl + 1)) << 8);
        __builtin_amdgcn_s_setprio(3);  // the other waves of this SIMD do bulk MFMA work
        int ks = 0;
        for (int kb = 0; kb < kend; ++kb, ks = wrap(ks + 1)) {
            potrf16(kb, ks);
            publish(0, kb + 1);
            const int w0 = __builtin_amdgcn_readlane(tab0, ks);
            const int p0 = w0 & 255;
            spin_until(4, kb);                       // column kb-1 fully applied by the workers
            xsolve(p0, Fb2 + (kb & 1) * BS);
            publish(2, kb + 1);
            {                                        // the next diagonal block
                sf_d4 c, av, bv;
                int ldc;
                double* Cb = load_ops(p0 | (p0 << 8) | ((w0 >> 8) << 16), c, av, bv, ldc);
                mma_store(Cb, ldc, c, av, bv);
            }
            publish(3, kb + 1);
        }
        __builtin_amdgcn_s_setprio(0);
    } else {
        const int mycnt = __builtin_amdgcn_readfirstlane(ptab[wave * SFB_PT + SFB_PT - 1]);
        // operation tables (lane = ring phase)
        int tabu[SFB_PT - 1];
#pragma unroll
        for (int p = 0; p < SFB_PT - 1; ++p) {
            const int e = __builtin_amdgcn_readfirstlane(ptab[wave * SFB_PT + (p < mycnt ? p : 0)]);
            tabu[p] = p < mycnt ? pack_update(tl, e >> 8, e & 255) : 0;
        }
        const int xI0 = 1 + wpos, xI1 = 1 + wpos + nworkers;  // row blocks this wave solves (at most two)
        const int tabx = (xI0 < RB ? idxP(tl, xI0) : 0) | ((xI1 < RB ? idxP(tl, xI1) : 0) << 8);
        int tabp = 0;  // destination block of prefetch slot q when column slot `lane` retires
#pragma unroll
        for (int q = 0; q < SFB_PF; ++q) {
            const int blk = pg + q * ngroups;
            tabp |= (band_loader ? (blk < nbr ? sfb_pair(tl, wrap(tl + 1 + blk)) : 0) : NBW + tl) << (8 * q);
            if (band_loader && blk >= nbr) poff[q] = -1;
        }
        int arrivals = 0;
        auto worker_barrier = [&]() {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            arrivals += nworkers;
            if (lane == 0) __hip_atomic_fetch_add((lds_int_t*)&sync[5], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            spin_until(5, arrivals);
        };
        int ks = 0;
        for (int kb = 0; kb < kend; ++kb, ks = wrap(ks + 1)) {
            // ---- solve block column kb (row block 0 is wave 0's)
            const int wx = __builtin_amdgcn_readlane(tabx, ks);
            spin_until(0, kb + 1);
            if (xI0 < RB) {
                if (xI0 == nb1 - 1 && kb > 0) spin_until(1, 4 * kb);  // the newest row: placed by four waves
                xsolve(wx & 255, Fb2 + (kb & 1) * BS);
            }
            if (xI1 < RB) {
                if (xI1 == nb1 - 1 && kb > 0) spin_until(1, 4 * kb);
                xsolve((wx >> 8) & 255, Fb2 + (kb & 1) * BS);
            }
            worker_barrier();
            // ---- HBM prefetch of block row kb + nbr (lands under the updates)
            if (kb + 1 < kend) {
                // straight-line code: all loads are issued back to back, nothing here reads their results.
                // Rows beyond the matrix (padding up to a multiple of 16) become identity rows; a bottom-up
                // sweep never leaves the matrix (its order is a multiple of 16).
                const bool inr = rev >= 0 || prow < n;
                const int live = inr ? pmask : 0;
#pragma unroll
                for (int q = 0; q < SFB_PF; ++q) {
                    pf[q] = (!inr && ((pmask >> (4 + q)) & 1)) ? 1.0 : 0.0;
                    if ((live >> q) & 1) pf[q] = *pptr[q];
                    pptr[q] += pstep;
                }
                prow += BB;
            }
            // ---- apply column kb to the window (operands of the next block are read while the matrix
            //      core works on the current one)
            spin_until(2, kb + 1);
            {
                sf_d4 cA, aA, bA, cB, aB, bB;
                double *pA = nullptr, *pB = nullptr;
                int ldA = BLD, ldB = BLD;
                if (mycnt > 0) pA = load_ops(__builtin_amdgcn_readlane(tabu[0], ks), cA, aA, bA, ldA);
#pragma unroll
                for (int p = 0; p < SFB_PT - 1; p += 2) {
                    if (p + 1 < SFB_PT - 1 && p + 1 < mycnt) pB = load_ops(__builtin_amdgcn_readlane(tabu[p + 1 < SFB_PT - 1 ? p + 1 : 0], ks), cB, aB, bB, ldB);
                    if (p < mycnt) mma_store(pA, ldA, cA, aA, bA);
                    if (p + 2 < SFB_PT - 1 && p + 2 < mycnt) pA = load_ops(__builtin_amdgcn_readlane(tabu[p + 2 < SFB_PT - 1 ? p + 2 : 0], ks), cA, aA, bA, ldA);
                    if (p + 1 < SFB_PT - 1 && p + 1 < mycnt) mma_store(pB, ldB, cB, aB, bB);
                }
            }
            if (wave == logwave && lane < BB) ld_acc += log(pv[(kb & 1) * BB + lane]);  // (its share of updates is the smallest)
            worker_barrier();
            if (wave == 1) publish(4, kb + 1);
            // ---- the prefetched rows land in the slots of column kb
            if (kb + 1 < kend) {
                spin_until(3, kb + 1);
                const int wp = __builtin_amdgcn_readlane(tabp, ks);
#pragma unroll
                for (int q = 0; q < SFB_PF; ++q)
                    if (poff[q] >= 0) lds[((wp >> (8 * q)) & 255) * BS + poff[q]] = pf[q];
                if (wave >= 4 && wave < 8) {  // group 0 holds block 0 = (new row, column kb+1) in its first register
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_fetch_add((lds_int_t*)&sync[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        }
    }
    __syncthreads();

    // ---- outputs
    if (wave == logwave && lane < BB) red[lane] = ld_acc;
    __syncthreads();
    double ldsum = 0.0;
    if (tid == 0) {
        for (int i = 0; i < BB; ++i) ldsum += red[i];
        if (sync[7] && a.info) a.info[b] = SF_INFO_INTERNAL;  // a pipeline flag wait timed out
        else if (bad && a.info && a.info[b] == 0) {
            // bottom-up half: report the column in the original numbering
            a.info[b] = half ? nblk * BB - bad + 1 : bad + a.info_off;
        }
    }
    if (a.nhalf == 2) {
        // what is left of the separator (block rows/columns kend .. kend+nb1-1): A_MM minus this half's
        // Schur contribution, its right-hand-side columns, the partial Gram matrix and log-determinant
        const int nm = nb1 * BB;
        const int64_t slot2 = (int64_t)b * 2 + half;
        double* dM = a.dumpM + slot2 * nm * nm;
        for (int e = tid; e < nm * nm; e += nthreads) {
            const int i = e / nm, j = e - i * nm;
            if (j > i) continue;
            const int rs = (kend + (i >> 4)) % nbr, cs = (kend + (j >> 4)) % nbr;
            dM[e] = Wb[sfb_pair(rs, cs) * BS + (i & 15) * BLD + (j & 15)];
        }
        double* dR = a.dumpR + slot2 * NR * nm;
        for (int e = tid; e < NR * nm; e += nthreads) {
            const int r = e / nm, j = e - r * nm;
            dR[e] = RH[((r >> 4) * nbr + (kend + (j >> 4)) % nbr) * BS + (r & 15) * BLD + (j & 15)];
        }
        double* dG = a.dumpG + slot2 * a.nrhs * a.nrhs;
        for (int e = tid; e < a.nrhs * a.nrhs; e += nthreads) {
            const int r = e / a.nrhs, c = e - r * a.nrhs;
            dG[e] = -((c <= r) ? G[r * LDG + c] : G[c * LDG + r]);
        }
        if (tid == 0) a.dumpL[slot2] = ldsum;
        return;
    }
    if (tid == 0) a.logdet[b] = ldsum + (a.logdet_add ? a.logdet_add[b] : 0.0);
    double* out = a.gram + (int64_t)b * a.nrhs * a.nrhs;
    const double* gadd = a.gram_add ? a.gram_add + (int64_t)b * a.nrhs * a.nrhs : nullptr;
    for (int e = tid; e < a.nrhs * a.nrhs; e += nthreads) {
        const int r = e / a.nrhs, c = e - r * a.nrhs;
        out[e] = -((c <= r) ? G[r * LDG + c] : G[c * LDG + r]) + (gadd ? gadd[e] : 0.0);
    }
}

// Merge of the two half sweeps: separator = top + flip(bottom) - A_MM (A_MM is contained in both dumps),
// written in band storage (half-width nm-1) so that the ordinary single sweep finishes the job.
__global__ __launch_bounds__(256) void k_band_merge(sf_band_args a, int nm, int NR, int row0, double* mband,
                                                    double* mrhs, double* gadd, double* ladd) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const double* band = a.band + (int64_t)b * a.sband;
    const double* top = a.dumpM + ((int64_t)b * 2) * nm * nm;
    const double* bot = top + (int64_t)nm * nm;
    double* mb = mband + (int64_t)b * nm * nm;
    for (int e = tid; e < nm * nm; e += 256) {
        const int i = e / nm, j = e - i * nm;
        if (j > i) continue;
        const int d = i - j;
        const double aij = d <= a.halfwidth ? band[(int64_t)(row0 + i) * a.ldb + d] : 0.0;
        mb[i * nm + d] = top[i * nm + j] + bot[(nm - 1 - j) * nm + (nm - 1 - i)] - aij;
    }
    const double* rt = a.dumpR + ((int64_t)b * 2) * NR * nm;
    const double* rb = rt + (int64_t)NR * nm;
    const double* rhs = a.rhs + (int64_t)b * a.srhs;
    const double* rhs0 = a.rhs0 ? a.rhs0 + (int64_t)b * a.srhs0 : nullptr;
    double* mr = mrhs + (int64_t)b * NR * nm;
    for (int e = tid; e < NR * nm; e += 256) {
        const int r = e / nm, j = e - r * nm;
        double orig = 0.0;
        if (r < a.nrhs) {
            if (rhs0) orig = r == 0 ? rhs0[row0 + j] : rhs[(int64_t)(r - 1) * a.ldr + row0 + j];
            else orig = rhs[(int64_t)r * a.ldr + row0 + j];
        }
        mr[e] = rt[e] + rb[r * nm + (nm - 1 - j)] - orig;
    }
    const int ng = a.nrhs * a.nrhs;
    const double* gt = a.dumpG + ((int64_t)b * 2) * ng;
    for (int e = tid; e < ng; e += 256) gadd[(int64_t)b * ng + e] = gt[e] + gt[ng + e];
    if (tid == 0) ladd[b] = a.dumpL[2 * b] + a.dumpL[2 * b + 1];
}

// Capacitance step of the Woodbury identity, one wave per walker:
//   S = I + gram[1:,1:],  S = L_S L_S^T,  u = L_S^-1 gram[1:,0]
//   logdet = logdet_band + sum log diag(S-pivots),  sqmah = gram[0][0] - u.u
__global__ __launch_bounds__(64) void k_woodbury(const double* __restrict__ gram, int nrhs,
                                                 const double* __restrict__ logdet_band,
                                                 double* __restrict__ logdet, double* __restrict__ sqmah,
                                                 int* __restrict__ info) {
    __shared__ double S[33 * 33];
    __shared__ double u[33];
    const int b = blockIdx.x, lane = threadIdx.x, m = nrhs - 1;
    const double* g = gram + (int64_t)b * nrhs * nrhs;
    for (int e = lane; e < m * m; e += 64) {
        const int r = e / m, c = e - r * m;
        S[r * 33 + c] = g[(r + 1) * nrhs + (c + 1)] + (r == c ? 1.0 : 0.0);
    }
    if (lane < m) u[lane] = g[(lane + 1) * nrhs];
    __syncthreads();
    double ld = 0.0;
    int bad = 0;
    for (int j = 0; j < m; ++j) {
        const double p = S[j * 33 + j];
        if (!(p > 0.0) && !bad) bad = 1;
        const double rs = 1.0 / sqrt(p);
        ld += log(p);
        __syncthreads();
        double lij = 0.0;
        if (lane > j && lane < m) {
            lij = S[lane * 33 + j] * rs;
            S[lane * 33 + j] = lij;
        }
        if (lane == j) u[j] *= rs;  // forward substitution rides along: u_j = (v_j - ...)/l_jj
        __syncthreads();
        if (lane > j && lane < m) {
            for (int c = j + 1; c <= lane; ++c) S[lane * 33 + c] -= lij * S[c * 33 + j];
            u[lane] -= lij * u[j];
        }
        __syncthreads();
    }
    if (lane == 0) {
        double uu = 0.0;
        for (int j = 0; j < m; ++j) uu += u[j] * u[j];
        logdet[b] = logdet_band[b] + ld;
        sqmah[b] = g[0] - uu;
        if (bad && info && info[b] == 0) info[b] = SF_INFO_BAD_WEIGHT_COV;
    }
}

// ------------------------------------------------------------------------------------ launchers
static size_t band_lds_bytes(int nbr, int nrb) {
    const size_t nr = (size_t)nrb * BB;
    return sizeof(double) * ((size_t)nbr * (nbr + 1) / 2 * BS + (size_t)nrb * nbr * BS + 2 * BS + nr * (nr + 1) + 4 * BB +
                             16 * SFB_PT / 2);  // window, rhs ring, two F, Gram, pivots/flags, pair table
}

static int band_nbr(int halfwidth) {  // window rows >= W + 16, and at least three blocks (see S2)
    const int nbr = (halfwidth + 2 * BB - 1) / BB;
    return nbr < 3 ? 3 : nbr;
}

int sf_band_max_halfwidth(int nrhs) {
    const int nrb = (nrhs + BB - 1) / BB;
    int best = -1;
    for (int w = 0; w <= 4096; w += BB) {
        if (band_lds_bytes(band_nbr(w), nrb) <= 160 * 1024) best = w;
        else break;
    }
    return best;
}

static int launch_forms_kernel(const sf_band_args& a, int nrb, int nblocks, hipStream_t s);

int sf_launch_band_forms(const double* band, int n, int halfwidth, int ldb, int64_t sband, int batch,
                         const double* rhs0, int64_t srhs0, const double* rhs, int nrhs, int ldr,
                         int64_t srhs, double* logdet, double* gram, int* info, hipStream_t s,
                         const double* logdet_add, const double* gram_add, int info_off) {
    const int nrb = (nrhs + BB - 1) / BB;
    if (n <= 0 || batch <= 0 || halfwidth < 0 || ldb < halfwidth + 1 || nrhs < 1 || nrb > 3) {
        sf_set_error("band_forms: bad arguments (n=%d halfwidth=%d ldb=%d nrhs=%d)", n, halfwidth, ldb, nrhs);
        return SF_EINVAL;
    }
    const int nbr = band_nbr(halfwidth);
    const size_t shm = band_lds_bytes(nbr, nrb);
    if (shm > 160 * 1024) {
        sf_set_error("band_forms: half-width %d with %d right-hand sides exceeds the LDS window (max %d)",
                     halfwidth, nrhs, sf_band_max_halfwidth(nrhs));
        return SF_EINVAL;
    }
    sf_band_args a;
    a.band = band;
    a.sband = sband;
    a.ldb = ldb;
    a.halfwidth = halfwidth;
    a.rhs = rhs;
    a.srhs = srhs;
    a.ldr = ldr;
    a.nrhs = nrhs;
    a.rhs0 = rhs0;
    a.srhs0 = srhs0;
    a.n = n;
    a.nbr = nbr;
    a.logdet = logdet;
    a.gram = gram;
    a.info = info;
    a.logdet_add = logdet_add;
    a.gram_add = gram_add;
    a.info_off = info_off;
    a.nhalf = 1;
    a.kend0 = a.kend1 = 0;
    a.dumpM = a.dumpR = a.dumpG = a.dumpL = nullptr;
    return launch_forms_kernel(a, nrb, batch, s);
}

static int launch_forms_kernel(const sf_band_args& a, int nrb, int nblocks, hipStream_t s) {
    const int nbr = a.nbr;
    const size_t shm = band_lds_bytes(nbr, nrb);
    static sf_dev_once attr_once;  // devices whose function attributes are set
    SF_CHECK(sf_once_per_device(&attr_once, []() -> int {
        SF_HIP(hipFuncSetAttribute((const void*)k_band_forms<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SF_HIP(hipFuncSetAttribute((const void*)k_band_forms<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SF_HIP(hipFuncSetAttribute((const void*)k_band_forms<3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        return SF_OK;
    }));
    // waves 4.. prefetch the band rows in groups of 256 threads (one 16 x 16 block per group and
    // register): enough groups for SFB_PF registers to cover the nbr blocks of a block row
    int nwaves = nbr <= SFB_PF ? 8 : nbr <= 2 * SFB_PF ? 12 : 16;
    static const int force_waves = SF_TUNE_INT("SF_BAND_WAVES", 0);
    if (force_waves) nwaves = force_waves;
    if (nwaves < 8 || nwaves > 16 || (nwaves & 3) || ((nwaves - 4) / 4) * SFB_PF < nbr ||
        nwaves * 64 < (nbr - 1 + nrb) * BB) {
        sf_set_error("band_forms: window too large for one workgroup");
        return SF_EINVAL;
    }
    const dim3 blk(nwaves * 64);
    if (nrb == 1) hipLaunchKernelGGL(k_band_forms<1>, dim3(nblocks), blk, shm, s, a);
    else if (nrb == 2) hipLaunchKernelGGL(k_band_forms<2>, dim3(nblocks), blk, shm, s, a);
    else hipLaunchKernelGGL(k_band_forms<3>, dim3(nblocks), blk, shm, s, a);
    SF_LAUNCH_CHECK();
    return SF_OK;
}

// Two workgroups per matrix: top-down and bottom-up half sweeps meet at a separator of nm = 16*(nbr-1)
// rows, which a third (short) sweep finishes.  Pays off while 2*batch workgroups still fit the chip (the
// sweep is a sequential chain, so halving it halves the latency) and whenever it fills the rounds better.  `work` needs
// sf_band_twisted_work_doubles(...) doubles.
size_t sf_band_twisted_work_doubles(int halfwidth, int nrhs, int batch) {
    const size_t nbr = band_nbr(halfwidth), nm = (nbr - 1) * BB, nr = (size_t)((nrhs + BB - 1) / BB) * BB;
    const size_t per = 2 * (nm * nm + nr * nm + (size_t)nrhs * nrhs + 1)  // dumps of the two halves
                       + nm * nm + nr * nm + (size_t)nrhs * nrhs + 1;     // merged separator, rhs, partial sums
    return per * batch + 64;
}

bool sf_band_twisted_applicable(int n, int halfwidth, int batch) {
    static int ncu = 0;
    if (!ncu) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
            ncu = 1;
    }
    static const bool off = SF_TUNE_FLAG("SF_BAND_NO_TWIST");
    const int nbr = band_nbr(halfwidth), nblk = (n + BB - 1) / BB;
    if (off || n % BB != 0 || nblk < 6 * nbr) return false;
    // One workgroup occupies a CU for the whole sweep, so the launch takes ceil(workgroups / CUs) rounds.
    // Two half sweeps per matrix are half as long each but twice as many, plus merge and separator sweep
    // (measured ~12 % of a sweep per round of matrices): take them when that means less time -- batch <=
    // CUs/2, and again wherever the last round of single sweeps would be mostly empty (CUs < batch <=
    // 1.5 CUs, ...).  Measured at cfg 2: B = 300: 2.37 -> 1.96 ms, 384: 2.34 -> 1.98, 640: 3.55 -> 3.25;
    // B = 1600 (cfg 3 shape) stays with single sweeps (6.13 vs 6.39 ms).
    const int rounds1 = (batch + ncu - 1) / ncu, rounds2 = (2 * batch + ncu - 1) / ncu;
    return 0.5 * rounds2 + 0.12 * rounds1 < (double)rounds1;
}

int sf_launch_band_forms_twisted(const double* band, int n, int halfwidth, int ldb, int64_t sband, int batch,
                                 const double* rhs0, int64_t srhs0, const double* rhs, int nrhs, int ldr,
                                 int64_t srhs, double* logdet, double* gram, int* info, double* work,
                                 hipStream_t s) {
    const int nrb = (nrhs + BB - 1) / BB, NR = nrb * BB;
    const int nbr = band_nbr(halfwidth), nb1 = nbr - 1, nm = nb1 * BB, nblk = n / BB;
    if (n % BB || nblk < 2 * nbr + nb1 || band_lds_bytes(nbr, nrb) > 160 * 1024 || ldb < halfwidth + 1) {
        sf_set_error("band_forms_twisted: bad arguments");
        return SF_EINVAL;
    }
    sf_band_args a;
    a.band = band;
    a.sband = sband;
    a.ldb = ldb;
    a.halfwidth = halfwidth;
    a.rhs = rhs;
    a.srhs = srhs;
    a.ldr = ldr;
    a.nrhs = nrhs;
    a.rhs0 = rhs0;
    a.srhs0 = srhs0;
    a.n = n;
    a.nbr = nbr;
    a.logdet = nullptr;
    a.gram = nullptr;
    a.info = info;
    a.logdet_add = nullptr;
    a.gram_add = nullptr;
    a.info_off = 0;
    a.nhalf = 2;
    a.kend0 = (nblk - nb1) / 2;
    a.kend1 = nblk - nb1 - a.kend0;
    const size_t b = (size_t)batch;
    double* w = work;
    a.dumpM = w; w += 2 * b * nm * nm;
    a.dumpR = w; w += 2 * b * NR * nm;
    a.dumpG = w; w += 2 * b * nrhs * nrhs;
    a.dumpL = w; w += 2 * b;
    double* mband = w; w += b * nm * nm;
    double* mrhs = w; w += b * NR * nm;
    double* gadd = w; w += b * nrhs * nrhs;
    double* ladd = w;
    int rc = launch_forms_kernel(a, nrb, 2 * batch, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_band_merge, dim3(batch), dim3(256), 0, s, a, nm, NR, a.kend0 * BB, mband, mrhs, gadd, ladd);
    SF_LAUNCH_CHECK();
    // the separator: a dense nm x nm matrix in band storage (half-width nm-1 -> the same window size)
    return sf_launch_band_forms(mband, nm, nm - 1, nm, (int64_t)nm * nm, batch, nullptr, 0, mrhs, nrhs, nm,
                                (int64_t)NR * nm, logdet, gram, info, s, ladd, gadd, a.kend0 * BB);
}

int sf_launch_woodbury(const double* gram, int nrhs, int batch, const double* logdet_band, double* logdet,
                       double* sqmah, int* info, hipStream_t s) {
    if (nrhs < 1 || nrhs > 33) {
        sf_set_error("woodbury: 0..32 low-rank rows supported");
        return SF_EINVAL;
    }
    hipLaunchKernelGGL(k_woodbury, dim3(batch), dim3(64), 0, s, gram, nrhs, logdet_band, logdet, sqmah, info);
    SF_LAUNCH_CHECK();
    return SF_OK;
}
