// C-ABI host layer (include/starfish_amd.h): context creation, workspace carving and the launch
// sequences.  Host code only prepares constants that the reference recomputes on every call but that
// do not depend on the walker (collocation factor of the fixed log-lambda grid, factor of the
// constant v11); all per-walker arithmetic runs in the HIP kernels.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <utility>
#include <vector>

#include "sf_common.h"
#include "sf_transform.h"

// ----------------------------------------------------------------------------------- errors
static thread_local char g_err[512] = "";
void sf_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* sf_last_error(void) { return g_err; }
extern "C" const char* sf_version(void) { return "starfish_amd 0.1 (gfx950)"; }
extern "C" int sf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ----------------------------------------------------------------------------------- profiling
// Process-global timing hooks for bench.py (HIP events on the launch streams), serialised by a mutex.
// Nothing on the data path reads them.
enum { PS_TRANSFORM = 0, PS_FILL, PS_GEMM, PS_POTRF, PS_SOLVE, PS_COUNT };
struct ProfSpan {
    hipEvent_t a, b;
    int stage;
};
static struct {
    int on = 0;
    std::vector<ProfSpan> spans;
    std::vector<hipEvent_t> pool;
    double gemm_flops = 0.0;
    long gemm_launches = 0;
    long calls = 0;
    hipEvent_t ref = nullptr;  // common time origin for merging overlapping launch intervals
} g_prof;
static std::mutex g_prof_mu;

static hipEvent_t prof_event() {
    hipEvent_t e;
    if (!g_prof.pool.empty()) {
        e = g_prof.pool.back();
        g_prof.pool.pop_back();
    } else {
        (void)hipEventCreate(&e);
    }
    return e;
}
struct ProfScope {
    hipStream_t s;
    ProfSpan sp;
    bool live;
    ProfScope(hipStream_t st, int stage) : s(st), live(g_prof.on != 0) {
        if (!live) return;
        std::lock_guard<std::mutex> lk(g_prof_mu);
        sp.stage = stage;
        sp.a = prof_event();
        sp.b = prof_event();
        (void)hipEventRecord(sp.a, s);
    }
    ~ProfScope() {
        if (!live) return;
        std::lock_guard<std::mutex> lk(g_prof_mu);
        (void)hipEventRecord(sp.b, s);
        g_prof.spans.push_back(sp);
    }
};
// called from sf_chol.hip around every k_gemm_nt launch
void sf_prof_gemm_begin(hipStream_t s, double flops, void** tok) {
    *tok = nullptr;
    if (!g_prof.on) return;
    ProfScope* p = new ProfScope(s, PS_GEMM);
    {
        std::lock_guard<std::mutex> lk(g_prof_mu);
        g_prof.gemm_flops += flops;
        g_prof.gemm_launches += 1;
    }
    *tok = p;
}
void sf_prof_gemm_end(void* tok) {
    if (tok) delete (ProfScope*)tok;
}

static void prof_count_call() {
    if (!g_prof.on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.calls += 1;
}

// ----------------------------------------------------------------------------------- sf_exec
int sf_exec_prepare(sf_exec* ex) {
    int dev = 0;
    SF_HIP(hipGetDevice(&dev));
    if (ex->side == nullptr || ex->device != dev) {
        if (ex->side) sf_exec_release(ex);
        // highest priority: the small launches of the diagonal-block chain must win freed CU slots against
        // the thousands of pending MFMA workgroups of the caller's stream, otherwise the chain starves
        int prio_lo = 0, prio_hi = 0;
        SF_HIP(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        SF_HIP(hipStreamCreateWithPriority(&ex->side, hipStreamNonBlocking, prio_hi));
        // (high priority too: in a multi-order call the next chunk's transform chains and fills -- dozens of small launches
        // per order -- run on it beside the factorisation, whose workgroups take a CU's whole register file; at normal
        // priority they only got CUs when a big launch drained: 16 of the 21 orders of cfg 3's second chunk were filled
        // AFTER the first chunk's factorisation, 16.7 ms of a 276 ms step with nothing else running)
        SF_HIP(hipStreamCreateWithPriority(&ex->aux, hipStreamNonBlocking, prio_hi));
        for (int g = 0; g < SF_EXEC_GROUPS - 1; ++g) SF_HIP(hipStreamCreateWithFlags(&ex->grp[g], hipStreamNonBlocking));
        SF_HIP(hipEventCreateWithFlags(&ex->fork, hipEventDisableTiming));
        SF_HIP(hipEventCreateWithFlags(&ex->join, hipEventDisableTiming));
        ex->device = dev;
    }
    ex->used = 0;
    return SF_OK;
}
int sf_exec_event(sf_exec* ex, hipEvent_t* e) {
    if (ex->used == ex->pool_size) {
        if (ex->pool_size == ex->pool_cap) {
            const size_t cap = ex->pool_cap ? 2 * ex->pool_cap : 256;
            hipEvent_t* np = (hipEvent_t*)realloc(ex->pool, cap * sizeof(hipEvent_t));
            if (!np) {
                sf_set_error("out of host memory (event pool)");
                return SF_ENOMEM;
            }
            ex->pool = np;
            ex->pool_cap = cap;
        }
        hipEvent_t ne;
        SF_HIP(hipEventCreateWithFlags(&ne, hipEventDisableTiming));
        ex->pool[ex->pool_size++] = ne;
    }
    *e = ex->pool[ex->used++];
    return SF_OK;
}
void sf_exec_release(sf_exec* ex) {
    if (!ex) return;
    for (size_t i = 0; i < ex->pool_size; ++i) (void)hipEventDestroy(ex->pool[i]);
    free(ex->pool);
    ex->pool = nullptr;
    ex->pool_size = ex->pool_cap = ex->used = 0;
    if (ex->fork) (void)hipEventDestroy(ex->fork);
    if (ex->join) (void)hipEventDestroy(ex->join);
    if (ex->side) (void)hipStreamDestroy(ex->side);
    if (ex->aux) (void)hipStreamDestroy(ex->aux);
    for (int g = 0; g < SF_EXEC_GROUPS - 1; ++g) {
        if (ex->grp[g]) (void)hipStreamDestroy(ex->grp[g]);
        ex->grp[g] = nullptr;
    }
    ex->fork = ex->join = nullptr;
    ex->side = ex->aux = nullptr;
    ex->device = -1;
}
// context-free entry points (sf_potrf_batch, ...): one sf_exec per calling thread and device
sf_exec* sf_exec_thread_local(void) {
    static thread_local sf_exec per_device[64];
    int dev = 0;
    (void)hipGetDevice(&dev);
    return &per_device[(dev >= 0 && dev < 64) ? dev : 0];
}

extern "C" int sf_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.on = on;
    if (on) {
        if (!g_prof.ref) SF_HIP(hipEventCreate(&g_prof.ref));
        SF_HIP(hipEventRecord(g_prof.ref, 0));
    }
    return SF_OK;
}
extern "C" int sf_profile_read(double* ms_by_stage, double* gemm_flops, long* gemm_launches, long* calls) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    double acc[PS_COUNT + 1] = {0, 0, 0, 0, 0, 0};
    std::vector<std::pair<double, double>> gemm_iv;  // [start, end) of every MFMA launch, ms since ref
    for (auto& sp : g_prof.spans) {
        float ms = 0.f;
        SF_HIP(hipEventSynchronize(sp.b));
        SF_HIP(hipEventElapsedTime(&ms, sp.a, sp.b));
        acc[sp.stage] += ms;
        if (sp.stage == PS_GEMM && g_prof.ref) {
            float ta = 0.f;
            if (hipEventElapsedTime(&ta, g_prof.ref, sp.a) == hipSuccess) gemm_iv.emplace_back(ta, ta + ms);
        }
        g_prof.pool.push_back(sp.a);
        g_prof.pool.push_back(sp.b);
    }
    g_prof.spans.clear();
    // launches on the two streams of the Cholesky overlap: merge the intervals so that concurrent
    // launches are not counted twice (slot 2 = union, slot 5 = plain sum of launch durations)
    acc[PS_COUNT] = acc[PS_GEMM];
    if (!gemm_iv.empty()) {
        std::sort(gemm_iv.begin(), gemm_iv.end());
        double uni = 0.0, lo = gemm_iv[0].first, hi = gemm_iv[0].second;
        for (size_t i = 1; i < gemm_iv.size(); ++i) {
            if (gemm_iv[i].first <= hi) {
                if (gemm_iv[i].second > hi) hi = gemm_iv[i].second;
            } else {
                uni += hi - lo;
                lo = gemm_iv[i].first;
                hi = gemm_iv[i].second;
            }
        }
        uni += hi - lo;
        acc[PS_GEMM] = uni;
    }
    if (ms_by_stage)
        for (int i = 0; i < PS_COUNT + 1; ++i) ms_by_stage[i] = acc[i];
    if (gemm_flops) *gemm_flops = g_prof.gemm_flops;
    if (gemm_launches) *gemm_launches = g_prof.gemm_launches;
    if (calls) *calls = g_prof.calls;
    g_prof.gemm_flops = 0.0;
    g_prof.gemm_launches = 0;
    g_prof.calls = 0;
    return SF_OK;
}

// ------------------------------------------------------------------- host-side spline set-up
// FITPACK knots of an interpolating k=5 spline through x[0..n): x0 x6, x[3:-3], x[n-1] x6.
static void quintic_knots(const double* x, int n, std::vector<double>& t) {
    t.resize((size_t)n + 6);
    for (int i = 0; i < 6; ++i) t[i] = x[0];
    for (int j = 3; j <= n - 4; ++j) t[j + 3] = x[j];
    for (int i = 0; i < 6; ++i) t[n + i] = x[n - 1];
}
static void bspl6_host(const double* t, int ell, double x, double h[6]) {
    double hh[5];
    h[0] = 1.0;
    for (int j = 1; j <= 5; ++j) {
        for (int i = 0; i < j; ++i) hh[i] = h[i];
        h[0] = 0.0;
        for (int i = 1; i <= j; ++i) {
            const int li = ell + i, lj = li - j;
            const double f = hh[i - 1] / (t[li] - t[lj]);
            h[i - 1] = h[i - 1] + f * (t[li] - x);
            h[i] = f * (x - t[lj]);
        }
    }
}
// Band LU (no pivoting; B-spline collocation matrices are totally positive) of A[i][j] = B_j(x_i).
// Outputs, per row j: Lf[j][k-1] = L[j][j-k], Uf[j][k-1] = U[j][j+k] (k = 1..SF_KB), rdiag[j] = 1/U[j][j].
static int quintic_collocation_lu(const double* x, int n, std::vector<double>& t, std::vector<double>& Lf,
                                  std::vector<double>& Uf, std::vector<double>& rdiag) {
    if (n < 6) {
        sf_set_error("resample needs at least 6 points, got %d", n);
        return SF_EINVAL;
    }
    for (int i = 1; i < n; ++i)
        if (!(x[i] > x[i - 1])) {
            sf_set_error("resample: the source grid must be strictly increasing");
            return SF_EINVAL;
        }
    quintic_knots(x, n, t);
    const int W = 2 * SF_KB + 1;
    std::vector<double> ab((size_t)n * W, 0.0);  // ab[i][col - i + KB]
    int ell = 5;
    for (int i = 0; i < n; ++i) {
        while (ell < n - 1 && t[ell + 1] <= x[i]) ++ell;
        double h[6];
        bspl6_host(t.data(), ell, x[i], h);
        for (int q = 0; q < 6; ++q) {
            const int col = ell - 5 + q;
            const int d = col - i + SF_KB;
            if (h[q] != 0.0) {
                if (d < 0 || d >= W) {
                    sf_set_error("collocation bandwidth exceeded at row %d", i);
                    return SF_EINVAL;
                }
                ab[(size_t)i * W + d] = h[q];
            }
        }
    }
    for (int k = 0; k < n; ++k) {
        const double piv = ab[(size_t)k * W + SF_KB];
        if (!(std::fabs(piv) > 0.0)) {
            sf_set_error("singular spline collocation matrix at row %d", k);
            return SF_EINVAL;
        }
        const int imax = (k + SF_KB < n - 1) ? k + SF_KB : n - 1;
        for (int i = k + 1; i <= imax; ++i) {
            double& lik = ab[(size_t)i * W + (k - i + SF_KB)];
            if (lik == 0.0) continue;
            lik /= piv;
            for (int j = k + 1; j <= imax; ++j) {
                const double ukj = ab[(size_t)k * W + (j - k + SF_KB)];
                if (ukj != 0.0) ab[(size_t)i * W + (j - i + SF_KB)] -= lik * ukj;
            }
        }
    }
    Lf.assign((size_t)n * SF_KB, 0.0);
    Uf.assign((size_t)n * SF_KB, 0.0);
    rdiag.resize(n);
    for (int j = 0; j < n; ++j) {
        rdiag[j] = 1.0 / ab[(size_t)j * W + SF_KB];
        for (int k = 1; k <= SF_KB; ++k) {
            if (j - k >= 0) Lf[(size_t)j * SF_KB + k - 1] = ab[(size_t)j * W + (SF_KB - k)];
            if (j + k < n) Uf[(size_t)j * SF_KB + k - 1] = ab[(size_t)j * W + (SF_KB + k)];
        }
    }
    return SF_OK;
}

// Truncated inverse of the collocation matrix from its band LU, by windowed column solves: column j of
// A^-1 is obtained with a forward sweep over [j, j+WF] and a backward sweep over [j-WF, j+WF] (entries
// further out are < 1e-30 of the peak).  band[(j - i + SF_IW) * n + i] = Ainv[i][j] for |i - j| <= SF_IW.
static void truncated_inverse_band(int n, const std::vector<double>& Lf, const std::vector<double>& Uf,
                                   const std::vector<double>& rdiag, std::vector<double>& band) {
    const int W = SF_IW, WF = SF_IW + 40;
    band.assign((size_t)(2 * W + 1) * n, 0.0);
    std::vector<double> yv(WF + 1), xv(2 * WF + 1);
    for (int j = 0; j < n; ++j) {
        const int hi = (j + WF < n - 1) ? j + WF : n - 1;
        const int lo = (j - WF > 0) ? j - WF : 0;
        yv[0] = 1.0;
        for (int i = j + 1; i <= hi; ++i) {
            double v = 0.0;
            for (int k = 1; k <= SF_KB && i - k >= j; ++k) v -= Lf[(size_t)i * SF_KB + k - 1] * yv[i - k - j];
            yv[i - j] = v;
        }
        // xv index: i - lo
        for (int i = hi; i >= lo; --i) {
            double v = (i >= j) ? yv[i - j] : 0.0;
            for (int k = 1; k <= SF_KB && i + k <= hi; ++k) v -= Uf[(size_t)i * SF_KB + k - 1] * xv[i + k - lo];
            xv[i - lo] = v * rdiag[i];
        }
        const int ilo = (j - W > 0) ? j - W : 0, ihi = (j + W < n - 1) ? j + W : n - 1;
        for (int i = ilo; i <= ihi; ++i) band[(size_t)(j - i + W) * n + i] = xv[i - lo];
    }
}

static void make_twiddles(int nf, std::vector<double>& tw) {
    tw.resize((size_t)nf);  // nf/2 complex values
    for (int k = 0; k < nf / 2; ++k) {
        const long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)k / (long double)nf;
        tw[2 * k] = (double)cosl(ang);
        tw[2 * k + 1] = (double)sinl(ang);
    }
}

// ----------------------------------------------------------------------------------- context
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    int alloc(size_t bytes) {
        SF_HIP(hipMalloc(&p, bytes ? bytes : 8));
        return SF_OK;
    }
    int upload(const void* src, size_t bytes) {
        int rc = alloc(bytes);
        if (rc) return rc;
        SF_HIP(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
        return SF_OK;
    }
    template <typename T>
    T* as() const {
        return (T*)p;
    }
};

struct sf_ctx {
    int device = 0;
    int n = 0, nf = 0, m = 0, P = 0, M = 0, npad = 0, lda = 0, mpad = 0, rows = 0;
    int monotonic = 1;
    int loguniform = 0;  // wave_i = wave_0 e^(i delta) to the rounding of the grid
    double dv = 0.0, wave_max = 0.0;
    DevBuf wave, flux, sigma, knots, spec, tw, Lf, Uf, rdiag, coef_static, inv_band;
    DevBuf grid, variances, lengthscales, gmin, gmax, alpha, Linv;  // (Linv holds the TRANSPOSE of Lc^-1)
    sf_exec exec;        // side / auxiliary streams and the event pool of this context's launch sequences
    sf_exec exec_potrf;  // multi-order calls: the factorisation's own streams / events (exec pipelines the fills)
    ~sf_ctx() {
        sf_exec_release(&exec);
        sf_exec_release(&exec_potrf);
    }
};

static double min_dv(const double* w, int n) {  // Starfish/utils.py:22
    double best = INFINITY;
    for (int i = 0; i + 1 < n; ++i) {
        const double v = (w[i + 1] - w[i]) / w[i];
        if (v < best) best = v;
    }
    return SF_C_KMS * best;
}

// Cholesky of v11 and the constants derived from it (emulator.py:387-388 solves with the constant
// v11 on every call; here the factor is built once).
static int emulator_constants(const double* v11, const double* w_hat, int N, std::vector<double>& alpha,
                              std::vector<double>& Linv) {
    std::vector<double> L((size_t)N * N, 0.0);
    for (int i = 0; i < N; ++i) {
        const double* ai = v11 + (size_t)i * N;
        double* li = &L[(size_t)i * N];
        for (int j = 0; j <= i; ++j) {
            const double* lj = &L[(size_t)j * N];
            double s = ai[j];
            for (int k = 0; k < j; ++k) s -= li[k] * lj[k];
            if (i == j) {
                if (!(s > 0.0)) {
                    sf_set_error("emulator v11 is not positive definite (row %d)", i);
                    return SF_EINVAL;
                }
                li[j] = std::sqrt(s);
            } else {
                li[j] = s / lj[j];
            }
        }
    }
    // W = Linv^T (row-major W[j][i] = Linv[i][j]) so the inner products run over contiguous memory
    std::vector<double> W((size_t)N * N, 0.0);
    for (int j = 0; j < N; ++j) {
        double* wj = &W[(size_t)j * N];
        for (int i = j; i < N; ++i) {
            const double* li = &L[(size_t)i * N];
            double s = (i == j) ? 1.0 : 0.0;
            for (int k = j; k < i; ++k) s -= li[k] * wj[k];
            wj[i] = s / li[i];
        }
    }
    Linv.assign((size_t)N * N, 0.0);
    for (int i = 0; i < N; ++i)
        for (int j = 0; j <= i; ++j) Linv[(size_t)i * N + j] = W[(size_t)j * N + i];
    // alpha = Linv^T (Linv w_hat)
    std::vector<double> y(N, 0.0);
    for (int i = 0; i < N; ++i) {
        double s = 0.0;
        for (int j = 0; j <= i; ++j) s += Linv[(size_t)i * N + j] * w_hat[j];
        y[i] = s;
    }
    alpha.assign(N, 0.0);
    for (int j = 0; j < N; ++j) {
        double s = 0.0;
        for (int i = j; i < N; ++i) s += W[(size_t)j * N + i] * y[i];
        alpha[j] = s;
    }
    return SF_OK;
}

extern "C" sf_ctx* sf_ctx_create(const sf_order_desc* d, int device, int* err) {
    int rc_dummy = 0;
    int& rc = err ? *err : rc_dummy;
    rc = SF_OK;
    auto fail = [&](int code) -> sf_ctx* {
        rc = code;
        return nullptr;
    };
    // n == 0 builds an emulator-only context (Emulator.__call__ without a SpectrumModel)
    const bool order_ok = d && (d->n == 0 || (d->n >= 2 && d->nf >= 8 && !(d->nf & (d->nf - 1)) && d->wave &&
                                               d->flux && d->sigma && d->min_dv_wave && d->bulk_fluxes));
    if (!d || !order_ok || d->m < 1 || d->m > SF_MAX_M || d->n_grid < 1 || d->M < 1 || !d->grid_points ||
        !d->variances || !d->lengthscales || !d->v11 || !d->w_hat) {
        sf_set_error("sf_ctx_create: bad descriptor");
        return fail(SF_EINVAL);
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= device) {
        sf_set_error("sf_ctx_create: no HIP device %d", device);
        return fail(SF_ENODEV);
    }
    if (hipSetDevice(device) != hipSuccess) {
        sf_set_error("hipSetDevice(%d) failed", device);
        return fail(SF_EHIP);
    }
    sf_ctx* c = new sf_ctx();
    c->device = device;
    c->n = d->n;
    c->nf = d->nf;
    c->m = d->m;
    c->P = d->n_grid;
    c->M = d->M;
    c->rows = d->m + 2;
    c->npad = (int)sf_align_up((size_t)d->n, SF_LEAF);
    c->lda = c->npad + 16;  // breaks the power-of-two row stride (HBM channel camping)
    c->mpad = (int)sf_align_up((size_t)d->m, 4);
    const bool has_order = d->n > 0;
    c->dv = has_order ? min_dv(d->min_dv_wave, d->nf) : 0.0;
    c->wave_max = has_order ? d->wave[0] : 0.0;
    for (int i = 0; i < d->n; ++i) {
        if (d->wave[i] > c->wave_max) c->wave_max = d->wave[i];
        if (i && !(d->wave[i] > d->wave[i - 1])) c->monotonic = 0;
    }
    if (c->monotonic && d->n > 2) {
        // log-uniform grid?  (w_i - w_{i-1}) / (w_i + w_{i-1}) = tanh(delta/2) for every i, to the
        // rounding of the wavelengths themselves (relative spread ~ ulp(w)/dw, e.g. 3e-11 at 5000 A, dv = 2)
        double qmin = 1e300, qmax = 0.0;
        for (int i = 1; i < d->n; ++i) {
            const double q = (d->wave[i] - d->wave[i - 1]) / (d->wave[i] + d->wave[i - 1]);
            qmin = q < qmin ? q : qmin;
            qmax = q > qmax ? q : qmax;
        }
        c->loguniform = (qmax - qmin) <= 2e-10 * qmax;
    }
#define TRY(x)              \
    do {                    \
        int rc__ = (x);     \
        if (rc__) {         \
            delete c;       \
            return fail(rc__); \
        }                   \
    } while (0)
    const size_t nb = sizeof(double) * (size_t)d->n;
    if (has_order) {
    TRY(c->wave.upload(d->wave, nb));
    TRY(c->flux.upload(d->flux, nb));
    TRY(c->sigma.upload(d->sigma, nb));

    std::vector<double> t, Lf, Uf, rdiag, tw;
    TRY(quintic_collocation_lu(d->min_dv_wave, d->nf, t, Lf, Uf, rdiag));
    TRY(c->knots.upload(t.data(), sizeof(double) * t.size()));
    TRY(c->Lf.upload(Lf.data(), sizeof(double) * Lf.size()));
    TRY(c->Uf.upload(Uf.data(), sizeof(double) * Uf.size()));
    TRY(c->rdiag.upload(rdiag.data(), sizeof(double) * rdiag.size()));
    {
        std::vector<double> band;
        truncated_inverse_band(d->nf, Lf, Uf, rdiag, band);
        // 16 x 16 coefficient blocks for the MFMA band product (k_spline_apply): output block ib uses the
        // input blocks ib-4 .. ib+4
        const int nfb = d->nf / 16, nkb = 2 * (SF_IW / 16) + 1, n = d->nf;
        std::vector<double> tblk((size_t)nfb * nkb * 256, 0.0);
        for (int ib = 0; ib < nfb; ++ib)
            for (int kb = 0; kb < nkb; ++kb)
                for (int r = 0; r < 16; ++r)
                    for (int cc = 0; cc < 16; ++cc) {
                        const int i = ib * 16 + r, k = (ib - SF_IW / 16 + kb) * 16 + cc;
                        if (k < 0 || k >= n || k - i > SF_IW || i - k > SF_IW) continue;
                        tblk[(((size_t)ib * nkb + kb) * 16 + r) * 16 + cc] = band[(size_t)(k - i + SF_IW) * n + i];
                    }
        TRY(c->inv_band.upload(tblk.data(), sizeof(double) * tblk.size()));
    }
    make_twiddles(d->nf, tw);
    TRY(c->tw.upload(tw.data(), sizeof(double) * tw.size()));

    // static spline coefficients of the un-broadened rows, stored [nf][rows]
    {
        std::vector<double> ct((size_t)d->nf * c->rows);
        for (int r = 0; r < c->rows; ++r)
            for (int j = 0; j < d->nf; ++j) ct[(size_t)j * c->rows + r] = d->bulk_fluxes[(size_t)r * d->nf + j];
        TRY(c->coef_static.upload(ct.data(), sizeof(double) * ct.size()));
        TRY(sf_launch_spline_solve(c->coef_static.as<double>(), 1, c->rows, 0, 1, c->rows, d->nf,
                                   c->Lf.as<double>(), c->Uf.as<double>(), c->rdiag.as<double>(), 0));
    }
    // half spectra of the static rows (rfft once; every walker only multiplies and inverts)
    {
        DevBuf bulk, scratch;
        TRY(bulk.upload(d->bulk_fluxes, sizeof(double) * (size_t)c->rows * d->nf));
        TRY(c->spec.alloc(sizeof(double2) * (size_t)c->rows * (d->nf / 2 + 1)));
        const size_t sb = sf_fft_scratch_bytes(c->rows, d->nf);
        if (sb) TRY(scratch.alloc(sb));
        TRY(sf_launch_rfft_rows(bulk.as<double>(), c->rows, d->nf, c->tw.as<double2>(), c->spec.as<double2>(),
                                scratch.as<double2>(), 0));
        if (hipDeviceSynchronize() != hipSuccess) {
            sf_set_error("context set-up kernels failed: %s", hipGetErrorString(hipGetLastError()));
            delete c;
            return fail(SF_EHIP);
        }
    }
    }  // has_order
    // emulator constants
    {
        const int N = d->m * d->M;
        std::vector<double> alpha, Linv, gmin(d->n_grid), gmax(d->n_grid);
        if ((d->linv != nullptr) != (d->alpha != nullptr)) {
            sf_set_error("sf_ctx_create: linv and alpha must be given together");
            delete c;
            return fail(SF_EINVAL);
        }
        if (d->linv) {
            Linv.assign(d->linv, d->linv + (size_t)N * N);
            alpha.assign(d->alpha, d->alpha + N);
        } else {
            TRY(emulator_constants(d->v11, d->w_hat, N, alpha, Linv));
        }
        for (int p = 0; p < d->n_grid; ++p) {
            gmin[p] = gmax[p] = d->grid_points[p];
            for (int j = 1; j < d->M; ++j) {
                const double v = d->grid_points[(size_t)j * d->n_grid + p];
                if (v < gmin[p]) gmin[p] = v;
                if (v > gmax[p]) gmax[p] = v;
            }
        }
        TRY(c->grid.upload(d->grid_points, sizeof(double) * (size_t)d->M * d->n_grid));
        TRY(c->variances.upload(d->variances, sizeof(double) * d->m));
        TRY(c->lengthscales.upload(d->lengthscales, sizeof(double) * (size_t)d->m * d->n_grid));
        TRY(c->gmin.upload(gmin.data(), sizeof(double) * d->n_grid));
        TRY(c->gmax.upload(gmax.data(), sizeof(double) * d->n_grid));
        TRY(c->alpha.upload(alpha.data(), sizeof(double) * N));
        {
            // the batched product reads Linv by columns: store the transpose (row index fastest)
            std::vector<double> LinvT((size_t)N * N);
            for (int i = 0; i < N; ++i)
                for (int j = 0; j < N; ++j) LinvT[(size_t)j * N + i] = Linv[(size_t)i * N + j];
            TRY(c->Linv.upload(LinvT.data(), sizeof(double) * (size_t)N * N));
        }
    }
#undef TRY
    return c;
}

extern "C" void sf_ctx_destroy(sf_ctx* c) { delete c; }
extern "C" int sf_ctx_npad(const sf_ctx* c) { return c ? c->npad : SF_EINVAL; }
extern "C" int sf_ctx_lda(const sf_ctx* c) { return c ? c->lda : SF_EINVAL; }

static int model_ok(const sf_ctx* c, const sf_model_desc* mdl) {
    if (!c || !mdl || mdl->n_local < 0 || mdl->n_cheb < 0) {
        sf_set_error("bad context / model descriptor");
        return SF_EINVAL;
    }
    return SF_OK;
}
// the calling thread's current device becomes the context's (HIP's current device is per thread)
static int use_device(const sf_ctx* c) {
    SF_HIP(hipSetDevice(c->device));
    return SF_OK;
}
extern "C" int sf_param_stride(const sf_ctx* c, const sf_model_desc* mdl) {
    if (model_ok(c, mdl)) return SF_EINVAL;
    return 6 + c->P + mdl->n_cheb + 3 * mdl->n_local + (mdl->has_av ? 1 : 0);
}

// ----------------------------------------------------------------------------------- workspace
struct Carve {
    char* base;
    size_t off = 0, cap;
    Carve(void* p, size_t bytes) : base((char*)p), cap(bytes) {}
    template <typename T>
    T* take(size_t count) {
        off = sf_align_up(off, 256);
        T* r = base ? (T*)(base + off) : nullptr;
        off += sizeof(T) * count;
        return r;
    }
};
// Row strides / per-unit sizes of the batched buffers.  A single-order call uses the order's own padding; a
// multi-order call (sf_loglike_multi_batch) pads every order to the largest one of the group so that all
// units share ONE batched factorisation.
struct Layout {
    int m, mpad, M, nf, rows, npad, lda;
};
static size_t tilemap_bytes(const Layout& L) {  // per unit: one byte per 128 x 128 tile, as the kernels index it
    const size_t nt128 = (size_t)(L.npad + 127) / 128;  // (the same count in the frame shifted by 64: npad = 64 mod 128 there)
    return nt128 * nt128;
}
static Layout layout_of(const sf_ctx* c) { return Layout{c->m, c->mpad, c->M, c->nf, c->rows, c->npad, c->lda}; }
struct Work {
    double *mu, *Lw, *zs, *kv, *scale, *logdet, *sqmah, *coef, *ybro, *Xraw, *fraw, *resid, *Y, *C, *ztrsv, *ltbuf, *mult, *gtab;
    double2* fft;
    int *info_e, *info_c;
    unsigned char* tilemap;
    unsigned short* tilelist;  // compact list of the materialised tiles (see sf_fill_args)
    int* tilecount;
    unsigned char* dmap;       // dense fill of caller matrices: structured-support map / list of 64 x 64 tiles
    unsigned short* dlist;
    int* dcount;
    size_t bytes, ltbuf_stride;
    Layout L;
    int trans_bt = 0;      // walkers one set of transient buffers is sized for
    size_t fft_set = 0;    // double2 per set
};
// B units of per-unit buffers; the transient buffers of the transform chain (used by one launch sequence at a
// time, stream ordered) are sized for Bt walkers
// (trans_sets > 1: that many independent sets of the transient buffers, for transform chains running side by side)
static Work carve(const Layout& L, const sf_model_desc* mdl, int B, int Bt, void* p, size_t cap, bool need_C,
                  int potrf_units = 0, int potrf_slots = 1, int trans_sets = 1) {
    Carve k(p, cap);
    Work w;
    w.L = L;
    if (potrf_units <= 0) potrf_units = B;
    const size_t b = (size_t)B, bt = (size_t)Bt * (size_t)trans_sets;
    w.mu = k.take<double>(b * L.m);
    w.Lw = k.take<double>(b * L.m * L.m);
    w.zs = k.take<double>(b * L.m * L.M * L.m);
    w.kv = k.take<double>(b * L.m * L.M);
    w.scale = k.take<double>(b);
    w.logdet = k.take<double>(b);
    w.sqmah = k.take<double>(b);
    w.info_e = k.take<int>(b);
    w.info_c = k.take<int>(b);
    w.coef = mdl->has_vsini ? k.take<double>(bt * L.nf * L.rows) : nullptr;
    w.ybro = mdl->has_vsini ? k.take<double>(bt * L.nf * L.rows) : nullptr;  // broadened rows before the fit
    w.mult = mdl->has_vsini ? k.take<double>(bt * (L.nf / 2 + 1)) : nullptr;  // broadening kernel per walker
    const size_t fb = mdl->has_vsini ? sf_fft_half_scratch_bytes(Bt * L.rows, L.nf) : 0;
    w.fft = fb ? k.take<double2>(fb / sizeof(double2) * (size_t)trans_sets) : nullptr;
    w.trans_bt = Bt;
    w.fft_set = fb / sizeof(double2);
    w.Xraw = k.take<double>(b * L.m * L.npad);
    w.fraw = k.take<double>(b * L.npad);
    w.resid = k.take<double>(b * L.npad);
    w.Y = k.take<double>(b * L.mpad * L.npad);
    w.ztrsv = k.take<double>(b * L.npad);
    w.ltbuf_stride = sf_align_up(sf_potrf_work_doubles(L.npad, potrf_units), 32);
    w.ltbuf = need_C ? k.take<double>(w.ltbuf_stride * potrf_slots) : nullptr;  // Cholesky scratch (per concurrent call)
    w.tilemap = need_C ? k.take<unsigned char>(b * tilemap_bytes(L)) : nullptr;
    w.tilelist = need_C ? k.take<unsigned short>(b * tilemap_bytes(L)) : nullptr;  // (capacity: every tile)
    w.tilecount = need_C ? k.take<int>(b) : nullptr;
    w.gtab = need_C ? k.take<double>(b * (size_t)L.npad) : nullptr;
    w.dmap = k.take<unsigned char>(b * sf_fill_dense_map_tiles(L.npad));
    w.dlist = k.take<unsigned short>(b * sf_fill_dense_map_tiles(L.npad));
    w.dcount = k.take<int>(b);
    w.C = need_C ? k.take<double>(b * (size_t)L.npad * L.lda) : nullptr;
    w.bytes = sf_align_up(k.off, 256);
    return w;
}
static Work carve(const sf_ctx* c, const sf_model_desc* mdl, int B, void* p, size_t cap, bool need_C) {
    return carve(layout_of(c), mdl, B, B, p, cap, need_C);
}
// the buffers of the units [u0, ...) of a multi-order workspace (transient buffers are shared)
static Work slice(const Work& w, int u0) {
    Work s = w;
    const Layout& L = w.L;
    const size_t u = (size_t)u0;
    s.mu += u * L.m;
    s.Lw += u * L.m * L.m;
    s.zs += u * L.m * L.M * L.m;
    s.kv += u * L.m * L.M;
    s.scale += u;
    s.logdet += u;
    s.sqmah += u;
    s.info_e += u;
    s.info_c += u;
    s.Xraw += u * L.m * L.npad;
    s.fraw += u * L.npad;
    s.resid += u * L.npad;
    s.Y += u * L.mpad * L.npad;
    s.ztrsv += u * L.npad;
    if (s.tilemap) s.tilemap += u * tilemap_bytes(L);
    if (s.tilelist) s.tilelist += u * tilemap_bytes(L);
    if (s.tilecount) s.tilecount += u;
    if (s.gtab) s.gtab += u * (size_t)L.npad;
    if (s.dmap) s.dmap += u * sf_fill_dense_map_tiles(L.npad);
    if (s.dlist) s.dlist += u * sf_fill_dense_map_tiles(L.npad);
    if (s.dcount) s.dcount += u;
    if (s.C) s.C += u * (size_t)L.npad * L.lda;
    return s;
}
// set `set` of the transient buffers (multi-order calls run several orders' transform chains side by side)
static Work with_trans_set(const Work& w, int set) {
    Work s = w;
    const Layout& L = w.L;
    const size_t o = (size_t)set * (size_t)w.trans_bt;
    if (s.coef) s.coef += o * L.nf * L.rows;
    if (s.ybro) s.ybro += o * L.nf * L.rows;
    if (s.mult) s.mult += o * (L.nf / 2 + 1);
    if (s.fft) s.fft += (size_t)set * w.fft_set;
    return s;
}
extern "C" size_t sf_workspace_bytes(const sf_ctx* c, const sf_model_desc* mdl, int B) {
    if (model_ok(c, mdl) || B <= 0) return 0;
    return carve(c, mdl, B, nullptr, 0, true).bytes;
}

// ----------------------------------------------------------------------------------- stages
static sf_emu_args emu_args(sf_ctx* c, const sf_model_desc* mdl, const double* d_params, const Work& w,
                            double* d_mu, double* d_cov, double* d_Lw, int* d_info);
static int run_emulator(sf_ctx* c, const sf_model_desc* mdl, int B, const double* d_params, const Work& w,
                        double* d_mu, double* d_cov, double* d_Lw, int* d_info, hipStream_t s) {
    return sf_launch_emulator(emu_args(c, mdl, d_params, w, d_mu, d_cov, d_Lw, d_info), B, s);
}
static sf_emu_args emu_args(sf_ctx* c, const sf_model_desc* mdl, const double* d_params, const Work& w,
                            double* d_mu, double* d_cov, double* d_Lw, int* d_info) {
    sf_emu_args e;
    e.params = d_params;
    e.pstride = sf_param_stride(c, mdl);
    e.off_grid = 6;
    e.m = c->m;
    e.M = c->M;
    e.P = c->P;
    e.grid = c->grid.as<double>();
    e.variances = c->variances.as<double>();
    e.lengthscales = c->lengthscales.as<double>();
    e.gmin = c->gmin.as<double>();
    e.gmax = c->gmax.as<double>();
    e.alpha = c->alpha.as<double>();
    e.LinvT = c->Linv.as<double>();
    e.zscratch = w.zs;
    e.kbuf = w.kv;
    e.mu = d_mu;
    e.cov = d_cov;
    e.Lw = d_Lw;
    e.info = d_info;
    return e;
}

// emulator + transform chain -> unscaled X / flux, scale, then residual / Y
static int run_transforms(sf_ctx* c, const sf_model_desc* mdl, int B, const double* d_params, const Work& w,
                          double* d_flux_out, double* d_X_out, double* d_resid_out, double* d_log_scale,
                          bool want_Y, hipStream_t s) {
    // (the broadening + spline-fit launches depend on vsini only, not on the emulator; forked onto a stream of their own beside
    // the emulator's launches and the band fill they do not shorten the banded step: every one of these launches fills the chip
    // by itself -- with three streams each simply takes longer, profiles/r05_q_banded_step_timelines.txt)
    const int pstride = sf_param_stride(c, mdl);
    SF_HIP(hipMemsetAsync(w.info_e, 0, sizeof(int) * (size_t)B, s));
    int rc = run_emulator(c, mdl, B, d_params, w, w.mu, nullptr, w.Lw, w.info_e, s);
    if (rc) return rc;
    const double* coef = c->coef_static.as<double>();
    if (mdl->has_vsini) {
        sf_broaden_args a;
        a.in = nullptr;
        a.spec = c->spec.as<double2>();
        a.B = B;
        a.rows = c->rows;
        a.nf = c->nf;
        a.tw = c->tw.as<double2>();
        a.dv = c->dv;
        a.kind = 1;
        a.params = d_params;
        a.pstride = pstride;
        a.poff = 0;
        a.scalar_param = 0.0;
        a.out = w.ybro;  // [B][rows][nf]: every row contiguous (coalesced stores)
        a.ob = (int64_t)c->nf * c->rows;
        a.orow = c->nf;
        a.oelem = 1;
        a.gscratch = w.fft;
        a.mult = w.mult;
        a.info = w.info_e;
        rc = sf_launch_broaden(a, s);
        if (rc) return rc;
        // spline coefficients of the broadened rows: truncated-inverse band product (fully parallel)
        rc = sf_launch_spline_apply(w.ybro, w.coef, B, c->rows, c->nf, c->inv_band.as<double>(), s);
        if (rc) return rc;
        coef = w.coef;
    }
    sf_eval_args ev;
    ev.wave = c->wave.as<double>();
    ev.knots = c->knots.as<double>();
    ev.coef = coef;
    ev.coef_batched = mdl->has_vsini ? 1 : 0;
    ev.params = d_params;
    ev.mu = w.mu;
    ev.X = w.Xraw;
    ev.flux = w.fraw;
    ev.info = w.info_e;
    ev.n = c->n;
    ev.nf = c->nf;
    ev.m = c->m;
    ev.ldx = w.L.npad;
    ev.pstride = pstride;
    ev.has_vz = mdl->has_vz;
    ev.n_cheb = mdl->n_cheb;
    ev.off_cheb = 6 + c->P;
    ev.has_av = mdl->has_av;
    ev.off_av = 6 + c->P + mdl->n_cheb + 3 * mdl->n_local;
    ev.wave_max = c->wave_max;
    sf_resid_args r;
    r.dflux = c->flux.as<double>();
    r.flux = w.fraw;
    r.X = w.Xraw;
    r.scale = w.scale;
    r.Lw = w.Lw;
    r.info = w.info_e;
    r.resid = w.resid;
    r.Y = want_Y ? w.Y : nullptr;
    r.flux_out = d_flux_out;
    r.X_out = d_X_out;
    r.n = c->n;
    r.m = c->m;
    r.mpad = c->mpad;
    r.ldx = w.L.npad;
    r.ldy = w.L.npad;
    r.use_sigma_w = mdl->use_sigma_w;
    if (mdl->has_log_scale) {
        // the scale factor does not depend on the flux: rows, scale and residual / Y in one pass, X never stored
        static const bool unfused = SF_TUNE_FLAG("SF_TRANSFORM_UNFUSED");  // (tests: the three launches give the same bits)
        if (!unfused) {
            rc = sf_launch_eval_resid_y(ev, r, w.scale, d_log_scale, B, s);
            if (rc) return rc;
            if (d_resid_out)
                SF_HIP(hipMemcpy2DAsync(d_resid_out, sizeof(double) * c->n, w.resid, sizeof(double) * w.L.npad,
                                        sizeof(double) * c->n, B, hipMemcpyDeviceToDevice, s));
            return SF_OK;
        }
    }
    rc = sf_launch_eval_rows(ev, B, s);
    if (rc) return rc;

    sf_scale_args sc;
    sc.wave = c->wave.as<double>();
    sc.dflux = c->flux.as<double>();
    sc.flux = w.fraw;
    sc.params = d_params;
    sc.scale = w.scale;
    sc.log_scale_out = d_log_scale;
    sc.n = c->n;
    sc.ldx = w.L.npad;
    sc.pstride = pstride;
    sc.has_log_scale = mdl->has_log_scale;
    rc = sf_launch_scale(sc, B, s);
    if (rc) return rc;

    rc = sf_launch_resid_y(r, B, s);
    if (rc) return rc;
    if (d_resid_out)
        SF_HIP(hipMemcpy2DAsync(d_resid_out, sizeof(double) * c->n, w.resid, sizeof(double) * w.L.npad,
                                sizeof(double) * c->n, B, hipMemcpyDeviceToDevice, s));
    return SF_OK;
}

static sf_fill_args fill_args(sf_ctx* c, const sf_model_desc* mdl, const double* d_params, const Work& w) {
    sf_fill_args f;
    f.wave = c->wave.as<double>();
    f.sigma = c->sigma.as<double>();
    f.Y = w.Y;
    f.params = d_params;
    f.n = c->n;
    f.npad = w.L.npad;
    f.mpad = c->mpad;
    f.ldy = w.L.npad;
    f.pstride = sf_param_stride(c, mdl);
    f.has_global = mdl->has_global;
    f.n_local = mdl->n_local;
    f.off_global = 4;
    f.off_local = 6 + c->P + mdl->n_cheb;
    f.monotonic = c->monotonic;
    f.loguniform = c->loguniform;
    f.nout = 0;
    f.tilemap = nullptr;
    f.nt128 = 0;
    f.fp = 0;
    f.tilelist = nullptr;
    f.tilecount = nullptr;
    f.list_cap = 0;
    f.gtab = nullptr;
    return f;
}

static int check_work(const sf_ctx* c, const sf_model_desc* mdl, int B, const void* d_work, size_t have,
                      bool need_C) {
    if (model_ok(c, mdl)) return SF_EINVAL;
    if (B <= 0 || !d_work) {
        sf_set_error("bad batch size / workspace");
        return SF_EINVAL;
    }
    const size_t need = carve(c, mdl, B, nullptr, 0, need_C).bytes;
    if (have < need) {
        sf_set_error("workspace too small: have %zu, need %zu", have, need);
        return SF_ENOMEM;
    }
    return use_device(c);
}

extern "C" int sf_emulator_query_batch(sf_ctx* c, const sf_model_desc* mdl, int B, const double* d_params,
                                       double* d_mu, double* d_cov, int* d_info, void* d_work,
                                       size_t work_bytes, void* stream) {
    int rc = check_work(c, mdl, B, d_work, work_bytes, false);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    Work w = carve(c, mdl, B, d_work, work_bytes, false);
    int* info = d_info ? d_info : w.info_e;
    SF_HIP(hipMemsetAsync(info, 0, sizeof(int) * (size_t)B, s));
    return run_emulator(c, mdl, B, d_params, w, d_mu ? d_mu : w.mu, d_cov, w.Lw, info, s);
}

extern "C" int sf_emulator_joint_batch(sf_ctx* c, const sf_model_desc* mdl, int B, const double* d_params,
                                       double* d_mu, double* d_cov, int* d_info, void* d_work,
                                       size_t work_bytes, void* stream) {
    int rc = check_work(c, mdl, B, d_work, work_bytes, false);
    if (rc) return rc;
    if (!d_mu || !d_cov) {
        sf_set_error("sf_emulator_joint_batch: d_mu and d_cov are required");
        return SF_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    Work w = carve(c, mdl, B, d_work, work_bytes, false);
    int* info = d_info ? d_info : w.info_e;
    SF_HIP(hipMemsetAsync(info, 0, sizeof(int) * (size_t)B, s));
    rc = run_emulator(c, mdl, B, d_params, w, w.mu, nullptr, nullptr, info, s);
    if (rc) return rc;
    return sf_launch_emu_joint(emu_args(c, mdl, d_params, w, w.mu, nullptr, nullptr, info), B, w.mu, d_mu, d_cov, s);
}

extern "C" int sf_transform_batch(sf_ctx* c, const sf_model_desc* mdl, int B, const double* d_params,
                                  double* d_flux, double* d_X, double* d_resid, double* d_log_scale,
                                  int* d_info, void* d_work, size_t work_bytes, void* stream) {
    int rc = check_work(c, mdl, B, d_work, work_bytes, false);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    Work w = carve(c, mdl, B, d_work, work_bytes, false);
    rc = run_transforms(c, mdl, B, d_params, w, d_flux, d_X, d_resid, d_log_scale, false, s);
    if (rc) return rc;
    if (d_info) SF_HIP(hipMemcpyAsync(d_info, w.info_e, sizeof(int) * (size_t)B, hipMemcpyDeviceToDevice, s));
    return SF_OK;
}

extern "C" int sf_forward_batch(sf_ctx* c, const sf_model_desc* mdl, int B, const double* d_params,
                                double* d_flux, double* d_cov, double* d_log_scale, int* d_info,
                                void* d_work, size_t work_bytes, void* stream) {
    int rc = check_work(c, mdl, B, d_work, work_bytes, false);
    if (rc) return rc;
    if (!d_cov) {
        sf_set_error("sf_forward_batch: d_cov is required");
        return SF_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    Work w = carve(c, mdl, B, d_work, work_bytes, false);
    rc = run_transforms(c, mdl, B, d_params, w, d_flux, nullptr, nullptr, d_log_scale, true, s);
    if (rc) return rc;
    sf_fill_args f = fill_args(c, mdl, d_params, w);
    f.C = d_cov;
    f.lda = c->n;
    f.stride = (int64_t)c->n * c->n;
    f.lower_only = 0;
    f.add_jitter = 0;
    rc = sf_exec_prepare(&c->exec);
    if (rc) return rc;
    rc = sf_launch_fill_dense(f, B, w.dmap, w.dlist, w.dcount, s, &c->exec);
    if (rc) return rc;
    if (d_info) SF_HIP(hipMemcpyAsync(d_info, w.info_e, sizeof(int) * (size_t)B, hipMemcpyDeviceToDevice, s));
    return SF_OK;
}

extern "C" int sf_cov_fill_batch(sf_ctx* c, const sf_model_desc* mdl, int B, const double* d_params, double* d_cov, int ld,
                                 int64_t stride, int lower_only, int add_jitter, int* d_info, void* d_work, size_t work_bytes,
                                 void* stream) {
    int rc = check_work(c, mdl, B, d_work, work_bytes, false);
    if (rc) return rc;
    if (!d_cov || ld < c->n || stride < (int64_t)c->n * ld) {
        sf_set_error("sf_cov_fill_batch: d_cov required, ld >= n (%d), stride >= n * ld", c->n);
        return SF_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    Work w = carve(c, mdl, B, d_work, work_bytes, false);
    prof_count_call();
    {
        // (the rank-m term needs Y = L_w^-1 (Omega X): the transform chain and the emulator query run first)
        ProfScope ps(s, PS_TRANSFORM);
        rc = run_transforms(c, mdl, B, d_params, w, nullptr, nullptr, nullptr, nullptr, true, s);
        if (rc) return rc;
    }
    {
        ProfScope ps(s, PS_FILL);
        sf_fill_args f = fill_args(c, mdl, d_params, w);
        f.C = d_cov;
        f.lda = ld;
        f.stride = stride;
        f.lower_only = lower_only ? 1 : 0;
        f.add_jitter = add_jitter ? 1 : 0;
        f.nout = c->n;  // the caller's matrices have n rows: no identity padding (it belongs to the workspace layout only)
        rc = sf_exec_prepare(&c->exec);
        if (rc) return rc;
        rc = sf_launch_fill_dense(f, B, w.dmap, w.dlist, w.dcount, s, &c->exec);  // (lower_only: the tile grid of sf_launch_fill)
        if (rc) return rc;
    }
    if (d_info) SF_HIP(hipMemcpyAsync(d_info, w.info_e, sizeof(int) * (size_t)B, hipMemcpyDeviceToDevice, s));
    return SF_OK;
}

extern "C" int sf_loglike_batch(sf_ctx* c, const sf_model_desc* mdl, int B, const double* d_params,
                                double* d_lnl, double* d_logdet, double* d_sqmah, double* d_resid,
                                double* d_log_scale, int* d_info, void* d_work, size_t work_bytes,
                                void* stream) {
    int rc = check_work(c, mdl, B, d_work, work_bytes, true);
    if (rc) return rc;
    if (!d_lnl) {
        sf_set_error("sf_loglike_batch: d_lnl is required");
        return SF_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    Work w = carve(c, mdl, B, d_work, work_bytes, true);
    prof_count_call();
    const int64_t stride = (int64_t)c->npad * c->lda;
    int call_fp = 0;
    {
        ProfScope ps(s, PS_TRANSFORM);
        rc = run_transforms(c, mdl, B, d_params, w, nullptr, nullptr, d_resid, d_log_scale, true, s);
        if (rc) return rc;
    }
    {
        ProfScope ps(s, PS_FILL);
        sf_fill_args f = fill_args(c, mdl, d_params, w);
        f.C = w.C;
        f.lda = c->lda;
        f.stride = stride;
        f.lower_only = 1;
        f.add_jitter = 1;
        f.gtab = w.gtab;        // per-diagonal table of the global kernel (used on log-uniform grids only)
        f.tilemap = w.tilemap;  // only tiles that carry more than the rank-m term are materialised
        f.tilelist = w.tilelist;
        f.tilecount = w.tilecount;
        f.list_cap = (int)tilemap_bytes(layout_of(c));
        // (the tiles of the factorisation's frame: see sf_potrf_front_pad.  Evaluated ONCE per call and handed on to the
        // factorisation below: it depends on the process-global persistent-kernel switch, which a recovery on another host
        // thread may flip between the two stages -- tile map and factorisation must agree on the frame)
        f.fp = sf_potrf_front_pad(c->npad, B);
        f.nt128 = (c->npad + f.fp + 127) / 128;
        call_fp = f.fp;
        rc = sf_launch_fill(f, B, s);
        if (rc) return rc;
    }
    {
        ProfScope ps(s, PS_POTRF);
        // the residual rides through the factorisation: w.resid is overwritten with z = L^-1 R
        sf_gen_args gen;
        gen.Y = w.Y;
        gen.mpad = c->mpad;
        gen.ldy = c->npad;
        gen.tilemap = w.tilemap;
        gen.fp = call_fp;
        gen.nt128 = (c->npad + gen.fp + 127) / 128;
        rc = sf_launch_potrf(w.C, c->npad, c->lda, stride, B, w.info_c, w.ltbuf, w.resid, c->npad, s, &gen, &c->exec);
        if (rc) return rc;
    }
    {
        ProfScope ps(s, PS_SOLVE);
        rc = sf_launch_logdet_z(w.C, c->npad, c->lda, stride, B, w.resid, c->npad, w.logdet, w.sqmah, s);
        if (rc) return rc;
        rc = sf_launch_finish(B, w.logdet, w.sqmah, w.info_e, w.info_c, d_lnl, d_info, s);
        if (rc) return rc;
    }
    if (d_logdet) SF_HIP(hipMemcpyAsync(d_logdet, w.logdet, sizeof(double) * (size_t)B, hipMemcpyDeviceToDevice, s));
    if (d_sqmah) SF_HIP(hipMemcpyAsync(d_sqmah, w.sqmah, sizeof(double) * (size_t)B, hipMemcpyDeviceToDevice, s));
    return SF_OK;
}

// ------------------------------------------------------------------- multi-order batches
// The units of several orders (the reference's multi-order spectra, Starfish/spectrum.py:96-115; orders are
// independent, docs/intro.rst:71-73) share ONE batched factorisation: every order runs its own transform chain
// and covariance fill into its slice of a common [units][npad][lda] array, padded (identity block) to the largest
// order of the group; the Cholesky, which is where the time goes, then sees sum(B_i) matrices in one launch
// sequence instead of nseg half-filled ones, and the host synchronises once.
static int multi_layout(const sf_segment* segs, int nseg, const sf_model_desc* mdl, Layout* L, int* units, int* bmax) {
    if (!segs || nseg <= 0 || !mdl) {
        sf_set_error("multi-order call: bad segment list / model descriptor");
        return SF_EINVAL;
    }
    const sf_ctx* c0 = segs[0].ctx;
    if (model_ok(c0, mdl)) return SF_EINVAL;
    *L = layout_of(c0);
    long long U = 0;
    int bm = 0;
    for (int i = 0; i < nseg; ++i) {
        const sf_ctx* c = segs[i].ctx;
        if (!c || !c->n || segs[i].B <= 0 || !segs[i].d_params) {
            sf_set_error("multi-order call: segment %d has no order context / batch / parameters", i);
            return SF_EINVAL;
        }
        if (c->device != c0->device || c->m != c0->m || c->P != c0->P) {
            sf_set_error("multi-order call: segment %d differs from segment 0 in device, eigenspectra or grid dimensions", i);
            return SF_EINVAL;
        }
        L->M = std::max(L->M, c->M);
        L->nf = std::max(L->nf, c->nf);
        L->npad = std::max(L->npad, c->npad);
        U += segs[i].B;
        bm = std::max(bm, (int)segs[i].B);
    }
    L->lda = L->npad + 16;
    if (U > 0x3fffffffLL) {
        sf_set_error("multi-order call: too many units");
        return SF_EINVAL;
    }
    *units = (int)U;
    *bmax = bm;
    return SF_OK;
}
// Chunks of a multi-order call (whole segments): a SMALL first chunk (at least 256 units: enough matrices to keep a
// factorisation's launches full) and the rest as the second -- only the first chunk's fills are exposed, the others
// run behind the first factorisation.  (Equal chunks: 1, 2, 3, 4 of them gave 283.1, 282.1, 282.7, 283.9 ms at cfg 3.)
// Orders whose transform chains + fills run side by side (own stream and own set of transient buffers each): a chain is
// ~14 small dependent launches, latency-bound -- alone it takes ~1 ms per order with the chip idle around it.
#define SF_MULTI_LANES 3  // (4, 6 and 8 lanes measured: no further gain)
static int multi_first_units(int U) {
    static const int first = std::max(1, SF_TUNE_INT("SF_MULTI_FIRST", 256));  // tuning aid
    return std::min(U, first);
}
static int multi_chunk_cap(int U, int bmax) { return std::min(U, std::max(U - multi_first_units(U), multi_first_units(U) + bmax)); }
extern "C" size_t sf_multi_workspace_bytes(const sf_segment* segs, int nseg, const sf_model_desc* mdl) {
    Layout L;
    int U = 0, bmax = 0;
    if (multi_layout(segs, nseg, mdl, &L, &U, &bmax)) return 0;
    return carve(L, mdl, U, bmax, nullptr, 0, true, multi_chunk_cap(U, bmax), 1, SF_MULTI_LANES).bytes;
}
extern "C" int sf_loglike_multi_batch(const sf_segment* segs, int nseg, const sf_model_desc* mdl, double* d_lnl,
                                      double* d_logdet, double* d_sqmah, double* d_log_scale, int* d_info,
                                      void* d_work, size_t work_bytes, void* stream) {
    Layout L;
    int U = 0, bmax = 0;
    int rc = multi_layout(segs, nseg, mdl, &L, &U, &bmax);
    if (rc) return rc;
    if (!d_lnl || !d_work) {
        sf_set_error("sf_loglike_multi_batch: d_lnl and a workspace are required");
        return SF_EINVAL;
    }
    const size_t need = carve(L, mdl, U, bmax, nullptr, 0, true, multi_chunk_cap(U, bmax), 1, SF_MULTI_LANES).bytes;
    if (work_bytes < need) {
        sf_set_error("workspace too small: have %zu, need %zu", work_bytes, need);
        return SF_ENOMEM;
    }
    sf_ctx* c0 = segs[0].ctx;
    if (use_device(c0)) return SF_EHIP;
    hipStream_t s = (hipStream_t)stream;
    Work W = carve(L, mdl, U, bmax, d_work, work_bytes, true, multi_chunk_cap(U, bmax), 1, SF_MULTI_LANES);
    prof_count_call();
    const int64_t stride = (int64_t)L.npad * L.lda;
    // (one frame for every chunk: the fills run before the chunk sizes are known; a chunk too small for the fused
    // sequences is factorised by them all the same -- sf_launch_potrf honours the frame of the tile map)
    const int fp = sf_potrf_front_pad(L.npad, 1 << 20);
    const int nt128 = (L.npad + fp + 127) / 128;
    // Pipeline: the per-order transform chains and fills (many small launches, a few per cent of the step) run on
    // the context's auxiliary stream one chunk of orders ahead of the factorisation on the caller's stream, so all
    // but the first chunk's are hidden behind the Cholesky of the previous chunk (see multi_first_units).
    sf_exec* ex = &c0->exec;
    rc = sf_exec_prepare(ex);
    if (rc) return rc;
    static const bool no_pipe = SF_TUNE_FLAG("SF_MULTI_NO_PIPELINE");  // tuning aid
    static const int lanes_env = SF_TUNE_INT("SF_MULTI_LANES_USED", SF_MULTI_LANES);
    const int nlanes = no_pipe ? 1 : std::max(1, std::min(lanes_env, SF_MULTI_LANES));
    // (the factorisation has its own executor, exec_potrf: all four streams of `ex` are free for the chains)
    // (no stream is created for the lanes: every additional ACTIVE stream costs dispatch latency on all of them --
    // one more for the wide sequence's A launches made a cfg-2 step 3 % slower)
    hipStream_t lane_stream[SF_MULTI_LANES] = {no_pipe ? s : ex->aux, ex->side, ex->grp[0]};
    const int first_units = multi_first_units(U);
    if (!no_pipe) {
        SF_HIP(hipEventRecord(ex->fork, s));
        for (int l = 0; l < nlanes; ++l) SF_HIP(hipStreamWaitEvent(lane_stream[l], ex->fork, 0));
    }
    struct Chunk {
        int u0, units;
        hipEvent_t filled[SF_MULTI_LANES];
    };
    std::vector<Chunk> chunks;
    bool lane_used[SF_MULTI_LANES] = {};
    int u0 = 0, cu0 = 0;
    for (int i = 0; i < nseg; ++i) {
        sf_ctx* c = segs[i].ctx;
        const int B = segs[i].B;
        const int lane = i % nlanes;
        hipStream_t sp = lane_stream[lane];
        lane_used[lane] = true;
        Work w = with_trans_set(slice(W, u0), lane);
        {
            ProfScope ps(sp, PS_TRANSFORM);
            rc = run_transforms(c, mdl, B, segs[i].d_params, w, nullptr, nullptr, nullptr,
                                d_log_scale ? d_log_scale + u0 : nullptr, true, sp);
            if (rc) return rc;
        }
        {
            ProfScope ps(sp, PS_FILL);
            sf_fill_args f = fill_args(c, mdl, segs[i].d_params, w);
            f.C = w.C;
            f.lda = L.lda;
            f.stride = stride;
            f.lower_only = 1;
            f.add_jitter = 1;
            f.gtab = w.gtab;
            f.tilemap = w.tilemap;
            f.tilelist = w.tilelist;
            f.tilecount = w.tilecount;
            f.list_cap = (int)tilemap_bytes(L);
            f.nt128 = nt128;
            f.fp = fp;
            rc = sf_launch_fill(f, B, sp);
            if (rc) return rc;
        }
        u0 += B;
        if ((chunks.empty() && u0 - cu0 >= first_units) || i == nseg - 1) {
            Chunk ch{cu0, u0 - cu0, {}};
            for (int l = 0; l < nlanes; ++l) {
                if (!lane_used[l] || lane_stream[l] == s) continue;
                rc = sf_exec_event(ex, &ch.filled[l]);
                if (rc) return rc;
                SF_HIP(hipEventRecord(ch.filled[l], lane_stream[l]));
                lane_used[l] = false;
            }
            chunks.push_back(ch);
            cu0 = u0;
        }
    }
    // (sf_launch_potrf rewinds the event pool of the executor it is given: the factorisation uses its own.
    // Two factorisations in flight on two streams, to hide one's under-filled last panels behind the other, were
    // measured slower: 306 vs 291 ms at cfg 3.)
    for (const Chunk& ch : chunks) {
        for (int l = 0; l < SF_MULTI_LANES; ++l)
            if (ch.filled[l]) SF_HIP(hipStreamWaitEvent(s, ch.filled[l], 0));
        Work w = slice(W, ch.u0);
        {
            ProfScope ps(s, PS_POTRF);
            sf_gen_args gen;
            gen.Y = w.Y;
            gen.mpad = L.mpad;
            gen.ldy = L.npad;
            gen.tilemap = w.tilemap;
            gen.nt128 = nt128;
            gen.fp = fp;
            rc = sf_launch_potrf(w.C, L.npad, L.lda, stride, ch.units, w.info_c, W.ltbuf, w.resid, L.npad, s, &gen,
                                 &c0->exec_potrf);
            if (rc) return rc;
        }
        {
            ProfScope ps(s, PS_SOLVE);
            rc = sf_launch_logdet_z(w.C, L.npad, L.lda, stride, ch.units, w.resid, L.npad, w.logdet, w.sqmah, s);
            if (rc) return rc;
            rc = sf_launch_finish(ch.units, w.logdet, w.sqmah, w.info_e, w.info_c, d_lnl + ch.u0,
                                  d_info ? d_info + ch.u0 : nullptr, s);
            if (rc) return rc;
        }
    }
    if (d_logdet) SF_HIP(hipMemcpyAsync(d_logdet, W.logdet, sizeof(double) * (size_t)U, hipMemcpyDeviceToDevice, s));
    if (d_sqmah) SF_HIP(hipMemcpyAsync(d_sqmah, W.sqmah, sizeof(double) * (size_t)U, hipMemcpyDeviceToDevice, s));
    return SF_OK;
}

// ------------------------------------------------------------------- structure-exploiting solver
struct BandWork {
    double *band, *gram, *logdet_band, *twist, *gtab, *tiles;
    int ldb;
    size_t bytes;
};
static BandWork carve_band(const sf_ctx* c, const sf_model_desc* mdl, int B, int halfwidth, void* p, size_t cap,
                           size_t base_bytes) {
    Carve k(p, cap);
    k.off = base_bytes;
    BandWork w;
    // Half-widths beyond the LDS window are factorised as bordered band matrices on the tile kernels of the dense
    // path (sf_launch_potrf_band); the band fill writes those tiles directly.
    const bool tiles = halfwidth > sf_band_max_halfwidth(c->m + 1);
    w.ldb = tiles ? 128 * (sf_band_tiles_wt(halfwidth) + 1) : ((halfwidth + 2) & ~1);
    w.band = tiles ? nullptr : k.take<double>((size_t)B * c->npad * w.ldb);  // (the tiles are filled directly)
    w.gram = k.take<double>((size_t)B * (c->m + 1) * (c->m + 1));
    w.logdet_band = k.take<double>((size_t)B);
    w.twist = tiles ? nullptr : k.take<double>(sf_band_twisted_work_doubles(halfwidth, c->m + 1, B));
    w.gtab = k.take<double>((size_t)B * (w.ldb + 2));
    w.tiles = tiles ? k.take<double>(sf_band_tiles_doubles(c->npad, B)) : nullptr;
    w.bytes = sf_align_up(k.off, 256);
    return w;
}
extern "C" int sf_banded_max_halfwidth(const sf_ctx* c) {
    if (!c || !c->n) return SF_EINVAL;
    if (!c->monotonic) return -1;
    const int lds_max = sf_band_max_halfwidth(c->m + 1), wide_max = SF_BAND_TILES_MAX_HALFWIDTH;
    return c->m + 1 <= 48 ? (wide_max > lds_max ? wide_max : lds_max) : lds_max;
}
extern "C" int sf_banded_window_halfwidth(const sf_ctx* c) {
    if (!c || !c->n) return SF_EINVAL;
    return c->monotonic ? sf_band_max_halfwidth(c->m + 1) : -1;
}
extern "C" size_t sf_banded_workspace_bytes(const sf_ctx* c, const sf_model_desc* mdl, int B, int halfwidth) {
    if (model_ok(c, mdl) || B <= 0 || halfwidth < 0) return 0;
    const size_t base = carve(c, mdl, B, nullptr, 0, false).bytes;
    return carve_band(c, mdl, B, halfwidth, nullptr, 0, base).bytes;
}
extern "C" int sf_loglike_banded_batch(sf_ctx* c, const sf_model_desc* mdl, int B, const double* d_params,
                                       int halfwidth, double* d_lnl, double* d_logdet, double* d_sqmah,
                                       double* d_resid, double* d_log_scale, int* d_info, void* d_work,
                                       size_t work_bytes, void* stream) {
    if (model_ok(c, mdl)) return SF_EINVAL;
    if (B <= 0 || !d_work || !d_lnl) {
        sf_set_error("sf_loglike_banded_batch: bad batch size / workspace / d_lnl");
        return SF_EINVAL;
    }
    const int wmax = sf_banded_max_halfwidth(c);
    if (halfwidth < 0 || halfwidth > wmax) {
        sf_set_error("sf_loglike_banded_batch: half-width %d outside [0, %d] (use sf_loglike_batch)", halfwidth, wmax);
        return SF_EINVAL;
    }
    const size_t base = carve(c, mdl, B, nullptr, 0, false).bytes;
    const size_t need = carve_band(c, mdl, B, halfwidth, nullptr, 0, base).bytes;
    if (work_bytes < need) {
        sf_set_error("workspace too small: have %zu, need %zu", work_bytes, need);
        return SF_ENOMEM;
    }
    hipStream_t s = (hipStream_t)stream;
    if (use_device(c)) return SF_EHIP;
    Work w = carve(c, mdl, B, d_work, work_bytes, false);
    BandWork bw = carve_band(c, mdl, B, halfwidth, d_work, work_bytes, base);
    prof_count_call();
    int rc;
    // The band fill depends on the covariance hyper-parameters only, the transforms on the stellar ones:
    // the two run side by side (fill on a library-owned auxiliary stream, joined before the sweep).
    sf_exec* aux = &c->exec;
    rc = sf_exec_prepare(aux);
    if (rc) return rc;
    hipStream_t sf = aux->aux;
    if (sf != s) {
        SF_HIP(hipEventRecord(aux->fork, s));
        SF_HIP(hipStreamWaitEvent(sf, aux->fork, 0));
    }
    const int64_t sband = (int64_t)c->npad * bw.ldb;
    {
        ProfScope ps(sf, PS_FILL);
        SF_HIP(hipMemsetAsync(w.info_c, 0, sizeof(int) * (size_t)B, sf));
        sf_fill_args f = fill_args(c, mdl, d_params, w);
        f.C = nullptr;
        f.lda = 0;
        f.stride = 0;
        f.lower_only = 1;
        f.add_jitter = 1;
        f.npad = (c->n + 15) / 16 * 16;
        if (bw.tiles) {  // straight into the 128 x 128 tiles of the bordered band matrix
            static const bool poison = SF_TUNE_FLAG("SF_BAND_TILES_POISON");  // test aid: NaN wherever a tile is read before it is written
            if (poison) SF_HIP(hipMemsetAsync(bw.tiles, 0xff, sizeof(double) * sf_band_tiles_doubles(c->npad, B), sf));
            f.npad = c->npad;
            const int lda_t = sf_band_tiles_lda(c->npad);
            rc = sf_launch_band_fill(f, B, bw.tiles, bw.ldb, halfwidth, lda_t, (int64_t)(c->npad + 64) * lda_t, w.info_c, bw.gtab, sf,
                                     sf_band_tiles_wt(halfwidth));
        } else
            rc = sf_launch_band_fill(f, B, bw.band, halfwidth + 1, halfwidth, bw.ldb, sband, w.info_c, bw.gtab, sf);
        if (rc) return rc;
    }
    if (sf != s) SF_HIP(hipEventRecord(aux->join, sf));
    {
        ProfScope ps(s, PS_TRANSFORM);
        rc = run_transforms(c, mdl, B, d_params, w, nullptr, nullptr, d_resid, d_log_scale, true, s);
        if (rc) return rc;
    }
    if (sf != s) SF_HIP(hipStreamWaitEvent(s, aux->join, 0));
    {
        ProfScope ps(s, PS_POTRF);
        const int n16 = (c->n + 15) / 16 * 16;
        if (bw.tiles)
            rc = sf_launch_potrf_band(c->n, c->npad, halfwidth, B, w.resid, c->npad, w.Y, c->m + 1, c->npad,
                                      (int64_t)c->mpad * c->npad, bw.logdet_band, bw.gram, w.info_c, bw.tiles, s);
        else if (sf_band_twisted_applicable(n16, halfwidth, B))
            rc = sf_launch_band_forms_twisted(bw.band, n16, halfwidth, bw.ldb, sband, B, w.resid, c->npad, w.Y,
                                              c->m + 1, c->npad, (int64_t)c->mpad * c->npad, bw.logdet_band,
                                              bw.gram, w.info_c, bw.twist, s);
        else
            rc = sf_launch_band_forms(bw.band, n16, halfwidth, bw.ldb, sband, B, w.resid, c->npad, w.Y, c->m + 1,
                                      c->npad, (int64_t)c->mpad * c->npad, bw.logdet_band, bw.gram, w.info_c, s);
        if (rc) return rc;
    }
    {
        ProfScope ps(s, PS_SOLVE);
        rc = sf_launch_woodbury(bw.gram, c->m + 1, B, bw.logdet_band, w.logdet, w.sqmah, w.info_c, s);
        if (rc) return rc;
        rc = sf_launch_finish(B, w.logdet, w.sqmah, w.info_e, w.info_c, d_lnl, d_info, s);
        if (rc) return rc;
    }
    if (d_logdet) SF_HIP(hipMemcpyAsync(d_logdet, w.logdet, sizeof(double) * (size_t)B, hipMemcpyDeviceToDevice, s));
    if (d_sqmah) SF_HIP(hipMemcpyAsync(d_sqmah, w.sqmah, sizeof(double) * (size_t)B, hipMemcpyDeviceToDevice, s));
    return SF_OK;
}

extern "C" int sf_band_logdet_gram_batch(const double* d_band, int n, int halfwidth, int ldb, int64_t stride,
                                         int batch, const double* d_rhs, int nrhs, int ldr, int64_t rhs_stride,
                                         double* d_logdet, double* d_gram, int* d_info, void* stream) {
    if (!d_band || !d_rhs || !d_logdet || !d_gram || !d_info) {
        sf_set_error("sf_band_logdet_gram_batch: null pointer");
        return SF_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    SF_HIP(hipMemsetAsync(d_info, 0, sizeof(int) * (size_t)batch, s));
    return sf_launch_band_forms(d_band, n, halfwidth, ldb, stride, batch, nullptr, 0, d_rhs, nrhs, ldr, rhs_stride,
                                d_logdet, d_gram, d_info, s);
}

// --------------------------------------------------------------------- stand-alone entry points
extern "C" int sf_global_cov(const double* d_wave, int n, double amplitude, double lengthscale, double* d_out,
                             void* stream) {
    if (!d_wave || !d_out || n < 0) {
        sf_set_error("sf_global_cov: bad argument");
        return SF_EINVAL;
    }
    return sf_launch_global_cov(d_wave, n, amplitude, lengthscale, d_out, (hipStream_t)stream);
}
extern "C" int sf_local_cov(const double* d_wave, int n, double amplitude, double mu, double sigma,
                            int accumulate, double* d_out, void* stream) {
    if (!d_wave || !d_out || n < 0) {
        sf_set_error("sf_local_cov: bad argument");
        return SF_EINVAL;
    }
    return sf_launch_local_cov(d_wave, n, amplitude, mu, sigma, accumulate, d_out, (hipStream_t)stream);
}

extern "C" size_t sf_fft_workspace_bytes(int rows, int nf) {
    if (rows <= 0 || nf <= 0) return 0;
    return sf_align_up(sizeof(double) * (size_t)nf, 256) + sf_fft_scratch_bytes(rows, nf) + 256;
}
static int broaden_free(const double* d_flux, int rows, int nf, double dv, int kind, double param,
                        double* d_out, void* d_work, size_t work_bytes, hipStream_t s) {
    if (!d_flux || !d_out || rows <= 0 || !d_work || work_bytes < sf_fft_workspace_bytes(rows, nf)) {
        sf_set_error("broaden: bad argument or workspace");
        return SF_EINVAL;
    }
    if (nf < 2 || (nf & (nf - 1))) {
        sf_set_error("broaden: nf=%d must be a power of two", nf);
        return SF_EINVAL;
    }
    Carve k(d_work, work_bytes);
    double* twd = k.take<double>((size_t)nf);
    const size_t fb = sf_fft_scratch_bytes(rows, nf);
    double2* scratch = fb ? k.take<double2>(fb / sizeof(double2)) : nullptr;
    std::vector<double> tw;
    make_twiddles(nf, tw);
    // pageable host -> device copy: synchronous w.r.t. the host buffer, safe to free afterwards
    SF_HIP(hipMemcpyAsync(twd, tw.data(), sizeof(double) * (size_t)nf, hipMemcpyHostToDevice, s));
    SF_HIP(hipStreamSynchronize(s));
    sf_broaden_args a;
    a.in = d_flux;
    a.spec = nullptr;
    a.B = 1;
    a.rows = rows;
    a.nf = nf;
    a.tw = (const double2*)twd;
    a.dv = dv;
    a.kind = kind;
    a.params = nullptr;
    a.pstride = 0;
    a.poff = 0;
    a.scalar_param = param;
    a.out = d_out;
    a.ob = 0;
    a.orow = nf;
    a.oelem = 1;
    a.gscratch = scratch;
    a.mult = nullptr;
    a.info = nullptr;
    return sf_launch_broaden(a, s);
}
extern "C" int sf_rotational_broaden(const double* d_flux, int rows, int nf, double dv, double vsini,
                                     double* d_out, void* d_work, size_t work_bytes, void* stream) {
    if (!(vsini > 0.0)) {
        sf_set_error("vsini must be positive");  // transforms.py:121-122
        return SF_EINVAL;
    }
    return broaden_free(d_flux, rows, nf, dv, 1, vsini, d_out, d_work, work_bytes, (hipStream_t)stream);
}
extern "C" int sf_instrumental_broaden(const double* d_flux, int rows, int nf, double dv, double fwhm,
                                       double* d_out, void* d_work, size_t work_bytes, void* stream) {
    if (fwhm < 0.0) {
        sf_set_error("FWHM must be non-negative");  // transforms.py:78-79
        return SF_EINVAL;
    }
    return broaden_free(d_flux, rows, nf, dv, 2, fwhm, d_out, d_work, work_bytes, (hipStream_t)stream);
}

extern "C" size_t sf_resample_workspace_bytes(int n, int rows) {
    if (n <= 0 || rows <= 0) return 0;
    // knots, Lf, Uf, rdiag, coefficient rows
    return sf_align_up(sizeof(double) * ((size_t)n + 6), 256) + 2 * sf_align_up(sizeof(double) * (size_t)n * SF_KB, 256) +
           sf_align_up(sizeof(double) * (size_t)n, 256) + sf_align_up(sizeof(double) * (size_t)n * rows, 256) + 1024;
}
extern "C" int sf_resample(const double* h_wave, int n, const double* d_flux, int rows, const double* d_new_wave,
                           int nq, double* d_out, void* d_work, size_t work_bytes, void* stream) {
    if (!h_wave || !d_flux || !d_new_wave || !d_out || rows <= 0 || nq < 0 || !d_work ||
        work_bytes < sf_resample_workspace_bytes(n, rows)) {
        sf_set_error("sf_resample: bad argument or workspace");
        return SF_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    std::vector<double> t, Lf, Uf, rdiag;
    int rc = quintic_collocation_lu(h_wave, n, t, Lf, Uf, rdiag);
    if (rc) return rc;
    Carve k(d_work, work_bytes);
    double* dt = k.take<double>(t.size());
    double* dL = k.take<double>(Lf.size());
    double* dU = k.take<double>(Uf.size());
    double* dr = k.take<double>(rdiag.size());
    double* dc = k.take<double>((size_t)n * rows);
    SF_HIP(hipMemcpyAsync(dt, t.data(), sizeof(double) * t.size(), hipMemcpyHostToDevice, s));
    SF_HIP(hipMemcpyAsync(dL, Lf.data(), sizeof(double) * Lf.size(), hipMemcpyHostToDevice, s));
    SF_HIP(hipMemcpyAsync(dU, Uf.data(), sizeof(double) * Uf.size(), hipMemcpyHostToDevice, s));
    SF_HIP(hipMemcpyAsync(dr, rdiag.data(), sizeof(double) * rdiag.size(), hipMemcpyHostToDevice, s));
    SF_HIP(hipMemcpyAsync(dc, d_flux, sizeof(double) * (size_t)n * rows, hipMemcpyDeviceToDevice, s));
    SF_HIP(hipStreamSynchronize(s));  // the host vectors go out of scope below
    rc = sf_launch_spline_solve(dc, 1, rows, 0, n, 1, n, dL, dU, dr, s);
    if (rc) return rc;
    if (nq == 0) return SF_OK;
    return sf_launch_spline_eval(dc, rows, n, dt, d_new_wave, nq, d_out, s);
}

extern "C" int sf_chebyshev_correct(const double* d_wave, int n, double wave_max, const double* d_flux, int rows,
                                    const double* h_coeffs, int ncoef, double* d_out, void* stream) {
    if (!d_wave || !d_flux || !h_coeffs || !d_out || n < 0 || rows <= 0 || ncoef < 1 || ncoef > 64) {
        sf_set_error("sf_chebyshev_correct: bad argument");
        return SF_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    // coefficients ride in a small device buffer owned by this call (stream-ordered alloc/free)
    double* dco = nullptr;
    SF_HIP(hipMalloc((void**)&dco, sizeof(double) * 64));
    hipError_t e = hipMemcpyAsync(dco, h_coeffs, sizeof(double) * ncoef, hipMemcpyHostToDevice, s);
    int rc = SF_OK;
    if (e != hipSuccess) rc = SF_EHIP;
    if (!rc) rc = sf_launch_cheb_rows(d_wave, n, wave_max, d_flux, rows, dco, ncoef, d_out, s);
    (void)hipStreamSynchronize(s);
    (void)hipFree(dco);
    return rc;
}

extern "C" int sf_extinct_ccm89(const double* d_wave, int n, const double* d_flux, int rows, double Av, double Rv,
                                double* d_out, void* stream) {
    if (!d_wave || !d_flux || !d_out || n < 0 || rows <= 0 || !(Rv > 0.0)) {
        sf_set_error("sf_extinct_ccm89: bad argument");
        return SF_EINVAL;
    }
    return sf_launch_extinct_rows(d_wave, n, d_flux, rows, Av, Rv, 0, d_out, (hipStream_t)stream);
}
// Anchor points of the spline-based laws (k = E(lambda - V)/E(B - V) at x = 1/lambda [um^-1]) and the second
// derivatives of the NATURAL cubic spline through them.  fitzpatrick99: Fitzpatrick (1999) section 5 / table 4 as
// coded in his FM_UNRED: optical anchors as polynomials in Rv, infrared ones scaled by Rv/3.1, two ultraviolet
// anchors from the FM90 curve with c2 = -0.824 + 4.717/Rv, c1 = 2.030 - 3.007 c2.  fm07: Fitzpatrick & Massa (2007)
// mean curve, defined for Rv = 3.1 only.  PARITY UNPINNED (see the header).
static int extinct_spline_table(int law, double Rv, std::vector<double>& tab) {
    std::vector<double> xk, yk;
    double c1, c2, c3, c4, c5, x0, gam, f99;
    auto uv = [&](double x) {
        const double x2 = x * x;
        double k = c1 + c2 * x + c3 * x2 / ((x2 - x0 * x0) * (x2 - x0 * x0) + x2 * gam * gam);
        if (x >= c5) {
            const double y = x - c5;
            k += f99 != 0.0 ? c4 * (0.5392 * y * y + 0.05644 * y * y * y) : c4 * y * y;
        }
        return k;
    };
    if (law == 3) {
        x0 = 4.596, gam = 0.99, c3 = 3.23, c4 = 0.41, c5 = 5.9, f99 = 1.0;
        c2 = -0.824 + 4.717 / Rv;
        c1 = 2.030 - 3.007 * c2;
        xk = {0.0, 1e4 / 26500.0, 1e4 / 12200.0, 1e4 / 6000.0, 1e4 / 5470.0, 1e4 / 4670.0, 1e4 / 4110.0, 1e4 / 2700.0, 1e4 / 2600.0};
        const double r2 = Rv * Rv, r3 = r2 * Rv, r4 = r3 * Rv;
        yk = {-Rv,
              0.26469 * Rv / 3.1 - Rv,
              0.82925 * Rv / 3.1 - Rv,
              -4.22809e-01 + 1.00270 * Rv + 2.13572e-04 * r2 - Rv,
              -5.13540e-02 + 1.00216 * Rv - 7.35778e-05 * r2 - Rv,
              7.00127e-01 + 1.00184 * Rv - 3.32598e-05 * r2 - Rv,
              1.19456 + 1.01707 * Rv - 5.46959e-03 * r2 + 7.97809e-04 * r3 - 4.45636e-05 * r4 - Rv,
              uv(1e4 / 2700.0),
              uv(1e4 / 2600.0)};
    } else {
        if (std::fabs(Rv - 3.1) > 1e-12) {
            sf_set_error("fm07 is defined for Rv = 3.1 only");
            return SF_EINVAL;
        }
        x0 = 4.592, gam = 0.922, c1 = -0.175, c2 = 0.807, c3 = 2.991, c4 = 0.319, c5 = 6.097, f99 = 0.0;
        xk = {0.0, 0.25, 0.50, 0.75, 1.0, 1e4 / 5530.0, 1e4 / 4000.0, 1e4 / 3300.0, 1e4 / 2700.0, 1e4 / 2600.0};
        yk.resize(xk.size());
        for (int i = 0; i < 5; ++i) yk[i] = (-0.83 + 0.63 * Rv) * std::pow(xk[i], 1.84) - Rv;
        yk[5] = 0.0;
        yk[6] = 1.322;
        yk[7] = 2.055;
        yk[8] = uv(xk[8]);
        yk[9] = uv(xk[9]);
    }
    const int nk = (int)xk.size();
    // natural cubic spline: tridiagonal system for the second derivatives (y2[0] = y2[nk-1] = 0)
    std::vector<double> y2(nk, 0.0), u(nk, 0.0);
    for (int i = 1; i < nk - 1; ++i) {
        const double sig = (xk[i] - xk[i - 1]) / (xk[i + 1] - xk[i - 1]);
        const double pp = sig * y2[i - 1] + 2.0;
        y2[i] = (sig - 1.0) / pp;
        const double dd = (yk[i + 1] - yk[i]) / (xk[i + 1] - xk[i]) - (yk[i] - yk[i - 1]) / (xk[i] - xk[i - 1]);
        u[i] = (6.0 * dd / (xk[i + 1] - xk[i - 1]) - sig * u[i - 1]) / pp;
    }
    for (int i = nk - 2; i >= 1; --i) y2[i] = y2[i] * y2[i + 1] + u[i];
    tab = {(double)nk, c1, c2, c3, c4, c5, x0 * x0, gam * gam, f99};
    tab.insert(tab.end(), xk.begin(), xk.end());
    tab.insert(tab.end(), yk.begin(), yk.end());
    tab.insert(tab.end(), y2.begin(), y2.end());
    return SF_OK;
}
extern "C" int sf_extinct(const double* d_wave, int n, const double* d_flux, int rows, double Av, double Rv, int law,
                          double* d_out, void* stream) {
    if (!d_wave || !d_flux || !d_out || n < 0 || rows <= 0 || !(Rv > 0.0) || law < 0 || law > 4) {
        sf_set_error("sf_extinct: bad argument");
        return SF_EINVAL;
    }
    if (law <= 2) return sf_launch_extinct_rows(d_wave, n, d_flux, rows, Av, Rv, law, d_out, (hipStream_t)stream);
    std::vector<double> tab;
    int rc = extinct_spline_table(law, Rv, tab);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)stream;
    // the table rides in a small device buffer owned by this call (like sf_chebyshev_correct's coefficients)
    double* dtab = nullptr;
    SF_HIP(hipMalloc((void**)&dtab, sizeof(double) * tab.size()));
    if (hipMemcpyAsync(dtab, tab.data(), sizeof(double) * tab.size(), hipMemcpyHostToDevice, s) != hipSuccess) rc = SF_EHIP;
    if (!rc) rc = sf_launch_extinct_spline_rows(d_wave, n, d_flux, rows, Av, Rv, dtab, d_out, s);
    (void)hipStreamSynchronize(s);
    (void)hipFree(dtab);
    return rc;
}

extern "C" size_t sf_potrf_workspace_bytes(int n, int batch) {
    if (n <= 0 || batch <= 0) return 0;
    // z scratch of the stand-alone solve + the transposed leaf factor read by the panel solves
    return sf_align_up(sizeof(double) * (size_t)n * batch, 256) +
           sf_align_up(sizeof(double) * sf_potrf_work_doubles(n, batch), 256) + 256;
}
extern "C" int sf_potrf_batch(double* d_A, int n, int lda, int64_t stride, int batch, int* d_info, void* d_work,
                              size_t work_bytes, void* stream) {
    if (!d_A || !d_info || !d_work || work_bytes < sf_potrf_workspace_bytes(n, batch)) {
        sf_set_error("sf_potrf_batch: bad argument or workspace");
        return SF_EINVAL;
    }
    double* ltbuf = (double*)((char*)d_work + sf_align_up(sizeof(double) * (size_t)n * batch, 256));
    ProfScope ps((hipStream_t)stream, PS_POTRF);
    return sf_launch_potrf(d_A, n, lda, stride, batch, d_info, ltbuf, nullptr, 0, (hipStream_t)stream);
}
extern "C" int sf_logdet_sqmah_batch(const double* d_L, int n, int lda, int64_t stride, int batch,
                                     const double* d_R, int ldr, void* d_work, size_t work_bytes,
                                     double* d_logdet, double* d_sqmah, void* stream) {
    if (!d_L || !d_R || !d_logdet || !d_sqmah || ldr < n) {
        sf_set_error("sf_logdet_sqmah_batch: bad argument");
        return SF_EINVAL;
    }
    double* z = nullptr;
    if (d_work && work_bytes >= sf_potrf_workspace_bytes(n, batch)) z = (double*)d_work;
    ProfScope ps((hipStream_t)stream, PS_SOLVE);
    return sf_launch_logdet_sqmah(d_L, n, lda, stride, batch, d_R, ldr, z, d_logdet, d_sqmah,
                                  (hipStream_t)stream);
}

extern "C" int sf_emulator_v11_build(const double* d_grid, int M, int P, int m, const double* d_hyper, const double* d_iphiphi,
                                     double* d_A, int npad, int lda, void* stream) {
    return sf_launch_v11_build(d_grid, M, P, m, d_hyper, d_iphiphi, d_A, npad, lda, (hipStream_t)stream);
}

// Recovery switch of the callers (process-global): after a batch came back SF_INFO_INTERNAL the host layer turns the
// persistent-kernel sequence off and re-runs the batch on a launch sequence (starfish_amd/_device.py).
extern "C" int sf_persistent_potrf(int enable) { return sf_set_persistent_potrf(enable); }
extern "C" int sf_persistent_potrf_status(long long* h_out8) {
    if (!h_out8) {
        sf_set_error("sf_persistent_potrf_status: h_out8 is required");
        return SF_EINVAL;
    }
    return sf_persistent_potrf_read_status(h_out8);
}

// Tuning / test aid: pin the launch sequence of the batched Cholesky (process-global).
extern "C" int sf_debug_cholesky_sequence(int mode) { return sf_set_cholesky_sequence(mode); }

// Tuning aid (not part of the Starfish surface): sustained shader clock while other streams are busy.
extern "C" int sf_debug_stream_write(double* d_dst, size_t count, double value, void* stream) {
    if (!d_dst) {
        sf_set_error("sf_debug_stream_write: d_dst is required");
        return SF_EINVAL;
    }
    return sf_launch_stream_write(d_dst, count, value, (hipStream_t)stream);
}

extern "C" int sf_debug_clock_probe(long long* d_out2, long long wall_ticks_100mhz, void* stream) {
    return sf_launch_clock_probe(d_out2, wall_ticks_100mhz, (hipStream_t)stream);
}
