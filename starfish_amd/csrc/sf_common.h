// Shared declarations for the gfx950 kernels and the C-ABI host layer (internal; not installed).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include "../../include/starfish_amd.h"

#define SF_LEAF 64        // Cholesky leaf block (potrf / trsm granularity); matrices padded to it
#define SF_NB 256         // outer left-looking panel width
#define SF_C_KMS 2.99792458e5

typedef double sf_d4 __attribute__((ext_vector_type(4)));

void sf_set_error(const char* fmt, ...);

#define SF_HIP(call)                                                                     \
    do {                                                                                 \
        hipError_t e_ = (call);                                                          \
        if (e_ != hipSuccess) {                                                          \
            sf_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
            return SF_EHIP;                                                              \
        }                                                                                \
    } while (0)

#define SF_LAUNCH_CHECK()                                                                \
    do {                                                                                 \
        hipError_t e_ = hipGetLastError();                                               \
        if (e_ != hipSuccess) {                                                          \
            sf_set_error("%s:%d launch -> %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return SF_EHIP;                                                              \
        }                                                                                \
    } while (0)

static inline size_t sf_align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// Function attributes (dynamic LDS limit) are per device, and contexts may be driven from different host threads:
// a call site owns one sf_dev_once; the set-up of a device runs once, under the site's mutex, and the device is
// marked only AFTER it succeeded (a second thread never launches before the attribute is in place).
struct sf_dev_once {
    std::atomic<unsigned long long> done{0};  // bitmask of the devices that are set up
    std::mutex mu;
};
template <class F>
static inline int sf_once_per_device(sf_dev_once* once, F&& setup) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return setup();
    const unsigned long long bit = 1ull << dev;
    if (once->done.load(std::memory_order_acquire) & bit) return SF_OK;
    std::lock_guard<std::mutex> lk(once->mu);
    if (once->done.load(std::memory_order_relaxed) & bit) return SF_OK;
    const int rc = setup();
    if (rc == SF_OK) once->done.fetch_or(bit, std::memory_order_release);
    return rc;
}
#define SF_CHECK(expr)              \
    do {                            \
        const int rc_ = (expr);     \
        if (rc_ != SF_OK) return rc_; \
    } while (0)

// Tuning / test switches read from the environment exist ONLY in -DSF_TUNING builds (`make TUNING=1` ->
// libstarfish_amd_tuning.so, loaded by tools/ through SF_LIB_PATH).  The release library has no environment-dependent
// behaviour: the macros collapse to their defaults and the switch names do not appear in the binary.
#ifdef SF_TUNING
#define SF_TUNE_FLAG(name) (getenv(name) != nullptr)
#define SF_TUNE_INT(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#define SF_TUNE_STR(name) (static_cast<const char*>(getenv(name)))
#else
#define SF_TUNE_FLAG(name) (false)
#define SF_TUNE_INT(name, dflt) (dflt)
#define SF_TUNE_STR(name) (static_cast<const char*>(nullptr))
#endif

// Streams and events a launch sequence needs besides the caller's stream.  Owned by a context (sf_ctx) or,
// for the context-free entry points, by the calling thread -- the library keeps no process-global stream state,
// so contexts can be driven from different host threads (and devices) concurrently.
#ifndef SF_EXEC_GROUPS
#define SF_EXEC_GROUPS 2
#endif
static_assert(SF_EXEC_GROUPS >= 2, "the wide sequence and the multi-order lanes use grp[0]: at least two slab groups");
// streams the slab groups of the fused Cholesky are spread over (the caller's + 1; measured: 1 -> 54.8, 2 -> 52.9, 3 -> 53.8 ms at cfg 2)
struct sf_exec {
    int device = -1;
    hipStream_t side = nullptr;  // highest priority: the diagonal-block chain of the Cholesky
    hipStream_t grp[SF_EXEC_GROUPS - 1] = {};  // slab groups 1.. of the fused Cholesky (group 0 = caller's stream)
    hipStream_t aux = nullptr;   // banded path: band fill beside the transforms; multi-order calls: the fills
    hipEvent_t fork = nullptr, join = nullptr;
    hipEvent_t* pool = nullptr;
    size_t pool_size = 0, pool_cap = 0, used = 0;
};
int sf_exec_prepare(sf_exec* ex);               // streams of the CURRENT device (created on first use), pool rewound
int sf_exec_event(sf_exec* ex, hipEvent_t* e);  // next pooled event (timing disabled)
void sf_exec_release(sf_exec* ex);
sf_exec* sf_exec_thread_local(void);

// profiling hooks (sf_abi.cpp)
void sf_prof_gemm_begin(hipStream_t s, double flops, void** tok);
void sf_prof_gemm_end(void* tok);

// ---- launchers implemented in the .hip files (all enqueue on `s`, never synchronise) ----------
// sf_chol.hip
#define SF_LTB_DOUBLES (SF_LEAF * SF_LEAF + SF_LEAF)  // side buffer per matrix: L^T of the leaf + its z
#define SF_LDT (SF_NB + 16)                            // row stride of the panel scratch
size_t sf_potrf_work_doubles(int n, int batch);        // doubles of scratch sf_launch_potrf needs
// Optional "matrix-free" start of the factorisation: 128x128 tiles whose tilemap byte is 0 were never
// written to A; their initial value is the rank-m product Y^T Y and is generated in the MFMA kernel.
struct sf_gen_args {
    const double* Y;  // [batch][mpad][ldy]
    int mpad, ldy;
    const unsigned char* tilemap;  // [batch][nt128 * nt128]
    int nt128;
    int fp;  // front pad the tile map was built for (sf_potrf_front_pad): tiles are those of the SHIFTED frame
};
// Orders whose padded size is an odd multiple of 64 (3008 = 23.5 slabs of 128 rows) are factorised in a frame shifted by
// `fp` = 64 virtual identity rows / columns IN FRONT: the half-empty slab becomes the first one, whose workgroups have
// the shortest K loops, instead of the last one with the longest (cfg 3: 6 % of the chip's time).  Nothing moves in
// memory: the kernels address the matrix from a base pointer fp (lda + 1) elements earlier, skip the fp leading columns
// of every K loop (they hold zeros below the diagonal block) and predicate the few accesses of the first tile row /
// column that would touch the virtual part.  0 = frame not shifted (n a multiple of 128, or the unfused sequence).
int sf_potrf_front_pad(int n, int batch);
int sf_launch_potrf(double* A, int n, int lda, int64_t stride, int batch, int* info, double* work,
                    double* rhs, int ldr, hipStream_t s, const sf_gen_args* gen = nullptr, sf_exec* ex = nullptr);
int sf_band_tiles_lda(int nband);
size_t sf_band_tiles_doubles(int nband, int batch);
int sf_band_tiles_wt(int halfwidth);  // tile (i, j) of a bordered band matrix meets the band iff i - j <= wt
int sf_launch_potrf_band(int n, int nband, int halfwidth, int batch, const double* rhs0, int64_t srhs0, const double* rhs,
                         int nrhs, int ldr, int64_t srhs, double* logdet, double* gram, int* info, double* tiles,
                         hipStream_t s);
int sf_set_persistent_potrf(int enable);  // 0 / 1: the dataflow sequence may be chosen; < 0: query.  Returns the previous value
int sf_persistent_potrf_read_status(long long* out8);  // sf_persistent_potrf_status (include/starfish_amd.h)
int sf_set_cholesky_sequence(int mode);  // -1 automatic (by batch size), 0 fused panel kernel, 1 unfused
int sf_launch_logdet_z(const double* L, int n, int lda, int64_t stride, int batch, const double* z, int ldr,
                       double* logdet, double* sqmah, hipStream_t s);
int sf_launch_logdet_sqmah(const double* L, int n, int lda, int64_t stride, int batch,
                           const double* R, int ldr, double* zscratch, double* logdet, double* sqmah, hipStream_t s);

int sf_launch_clock_probe(long long* out, long long wall_ticks, hipStream_t s);

// sf_fill.hip
struct sf_fill_args {
    const double* wave;    // [n]
    const double* sigma;   // [n]
    const double* Y;       // [B][mpad][ldy]  (rank-m factor rows, zero padded to mpad)
    const double* params;  // [B][pstride]
    double* C;             // [B] x stride
    int n, npad, lda, mpad, ldy, pstride;
    int64_t stride;
    int has_global, n_local, off_global, off_local;
    int lower_only;        // 1: only tiles touching the lower triangle, identity padding written
    int add_jitter;        // 1: + SF_JITTER on the diagonal
    int nout;              // rows / columns of C that are written; 0: npad with lower_only (workspace matrices: identity
                           // padding included), n otherwise.  Caller-owned matrices of n rows set it to n
    int monotonic;         // wave sorted ascending -> band culling allowed
    int loguniform;        // wave_i = wave_0 e^(i delta) to rounding -> K_global depends on i-j only
    unsigned char* tilemap; // optional [B][nt128*nt128]: 1 = the 128x128 tile is materialised in C
    int nt128;
    int fp;                // 0 or 64: the tile map / list index the tiles of the factorisation's shifted frame (sf_potrf_front_pad)
    // optional compact work list of the materialised tiles (likelihood path): k_tile_map appends (tm << 8 | tn) per flagged
    // 128 x 128 tile, the fill then launches a few workgroups per walker that walk the list instead of one (mostly empty)
    // workgroup per 64 x 64 tile of the whole matrix
    unsigned short* tilelist;  // [B][list_cap]
    int* tilecount;            // [B]
    int list_cap;
    double* gtab;          // optional [B][n] scratch: K_global per diagonal (log-uniform grids, likelihood path)
};
static inline __host__ __device__ int sf_fill_extent(const sf_fill_args& a) {
    return a.nout > 0 ? a.nout : (a.lower_only ? a.npad : a.n);
}
int sf_launch_fill(const sf_fill_args& a, int B, hipStream_t s);
// dense (both triangles) matrices of the caller: plain / structured tiles split (smap, list: sf_fill_dense_map_tiles(n) per matrix)
size_t sf_fill_dense_map_tiles(int n);
// (ex: prepared executor whose auxiliary stream takes the structured tiles beside the plain ones, or NULL)
int sf_launch_fill_dense(const sf_fill_args& a, int B, unsigned char* smap, unsigned short* list, int* count, hipStream_t s,
                         sf_exec* ex);
int sf_launch_stream_write(double* dst, size_t count, double v, hipStream_t s);
// band storage of the structured part of C (sf_band.hip consumes it); a.npad = rows written (>= a.n)
int sf_launch_band_fill(const sf_fill_args& a, int B, double* band, int ws, int halfwidth, int ldb, int64_t sband,
                        int* info, double* gtab, hipStream_t s, int tile_wt = -1);  // ws stored diagonals > halfwidth; gtab: B x (ws+1) or NULL
int sf_launch_global_cov(const double* wave, int n, double amp, double ls, double* out, hipStream_t s);
int sf_launch_local_cov(const double* wave, int n, double amp, double mu, double sigma, int accumulate,
                        double* out, hipStream_t s);

// sf_band.hip: banded Cholesky + forward substitution of (1 + m) right-hand sides -> logdet, Gram matrix
int sf_band_max_halfwidth(int nrhs);
int sf_launch_band_forms(const double* band, int n, int halfwidth, int ldb, int64_t sband, int batch,
                         const double* rhs0, int64_t srhs0, const double* rhs, int nrhs, int ldr,
                         int64_t srhs, double* logdet, double* gram, int* info, hipStream_t s,
                         const double* logdet_add = nullptr, const double* gram_add = nullptr, int info_off = 0);
// two half sweeps + separator (halves the sequential chain when 2*batch workgroups fit the chip)
size_t sf_band_twisted_work_doubles(int halfwidth, int nrhs, int batch);
bool sf_band_twisted_applicable(int n, int halfwidth, int batch);
int sf_launch_band_forms_twisted(const double* band, int n, int halfwidth, int ldb, int64_t sband, int batch,
                                 const double* rhs0, int64_t srhs0, const double* rhs, int nrhs, int ldr,
                                 int64_t srhs, double* logdet, double* gram, int* info, double* work,
                                 hipStream_t s);
int sf_launch_woodbury(const double* gram, int nrhs, int batch, const double* logdet_band, double* logdet,
                       double* sqmah, int* info, hipStream_t s);
// wide bands (beyond the LDS window): bordered band matrices on the fused panel kernel, sf_launch_potrf_band
#define SF_BAND_TILES_MAX_HALFWIDTH 768
