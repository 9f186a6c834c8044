// Probe 6: the in-register 16x16 Cholesky + inverse of the banded solver (MFMA accumulator layout), one wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double rl(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
__device__ __forceinline__ double rsq(double p) {
    double y = __builtin_amdgcn_rsq(p);
    const double h = 0.5 * p;
    double e = __builtin_fma(-h * y, y, 0.5);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-h * y, y, 0.5);
    y = __builtin_fma(y, e, y);
    return y;
}
template <int MODE>
__global__ void k(const double* A, double* out, long long* cyc, int reps) {
    __shared__ double pv[16];
    const int lane = threadIdx.x, l15 = lane & 15, lq = lane >> 4;
    long long t0 = __builtin_readcyclecounter();
    d4 f;
    for (int rep = 0; rep < reps; ++rep) {
        d4 acc;
        for (int r = 0; r < 4; ++r) {
            const int row = lq + 4 * r;
            acc[r] = A[max(row, l15) * 16 + min(row, l15)] + rep * 1e-9;
            f[r] = row == l15 ? 1.0 : 0.0;
        }
        double p = rl(acc[0], 0);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int qj = j & 3, rj = j >> 2;
            const double rs = rsq(p);
            if (MODE & 1) if (lane == 0) pv[j] = p;
            const bool in_q = lq == qj;
            const double v = (in_q && l15 > j) ? acc[rj] * rs : 0.0;
            const double g = in_q ? f[rj] * rs : 0.0;
            if (in_q) f[rj] = g;
            if (j + 1 < 16) {
                const double an = rl(acc[(j + 1) >> 2], ((j + 1) & 3) * 16 + j + 1);
                const double vn = rl(v, qj * 16 + j + 1);
                p = __builtin_fma(-vn, vn, an);
            }
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-v, v, acc, 0, 0, 0);
            if (!(MODE & 2)) f = __builtin_amdgcn_mfma_f64_16x16x4f64(-v, g, f, 0, 0, 0);
        }
    }
    long long t1 = __builtin_readcyclecounter();
    for (int r = 0; r < 4; ++r) out[(lq + 4 * r) * 16 + l15] = f[r];
    if (lane == 0) cyc[0] = (t1 - t0) / reps;
    if (lane == 1) out[256] = pv[3];
}
template <int MODE> void run(const char* name, const double* dA) {
    double* out; long long* cyc; (void)hipMalloc(&out, 300 * 8); (void)hipMalloc(&cyc, 8);
    k<MODE><<<1, 64>>>(dA, out, cyc, 2);
    k<MODE><<<1, 64>>>(dA, out, cyc, 200);
    long long h; (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-40s %lld cycles per 16x16 block (%.0f per column)\n", name, h, h / 16.0);
}
int main() {
    double hA[256];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) hA[i * 16 + j] = (i == j) ? 20.0 : 1.0 / (1 + abs(i - j));
    double* dA; (void)hipMalloc(&dA, sizeof(hA)); (void)hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice);
    run<0>("potrf16 + inverse (2 MFMA/column)", dA);
    run<1>("... + pivot store to LDS by lane 0", dA);
    run<2>("potrf16 only (1 MFMA/column)", dA);
    return 0;
}
