// Probe 5: latency of dependent instruction chains in ONE wave (what bounds the in-register
// diagonal-block factorisation of the banded solver): f64 fma, v_rsq_f64, readlane->VALU, f64 MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double rl(double v, int lane) {
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
template <int MODE>
__global__ void k(double* out, long long* cyc, int iters) {
    double x = 1.0 + threadIdx.x * 1e-3, y = 0.999;
    d4 acc = {1, 2, 3, 4};
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (MODE == 0) x = __builtin_fma(x, y, 1e-9);                       // dependent f64 fma
            if (MODE == 1) x = __builtin_amdgcn_rsq(x) + 1.0;                    // rsq + add
            if (MODE == 2) x = __builtin_fma(rl(x, u), y, 1e-9);                 // readlane -> fma
            if (MODE == 3) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0); }  // dependent MFMA
            if (MODE == 4) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0); x = acc[0] * y; }  // MFMA -> VALU -> MFMA
            if (MODE == 5) x = x * y;                                            // dependent f64 mul
            if (MODE == 6) { float f = (float)x; f = __builtin_amdgcn_rsqf(f); x = (double)f + 1.0; }  // f32 rsq round trip
        }
    }
    long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x + acc[0] + acc[1] + acc[2] + acc[3];
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
template <int MODE> void run(const char* name) {
    double* out; long long* cyc; hipMalloc(&out, 64 * 8); hipMalloc(&cyc, 8);
    const int iters = 1000;
    k<MODE><<<1, 64>>>(out, cyc, 10);
    k<MODE><<<1, 64>>>(out, cyc, iters);
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-28s %.1f cycles per step\n", name, (double)h / (iters * 16));
}
int main() {
    run<0>("dependent v_fma_f64"); run<5>("dependent v_mul_f64"); run<1>("v_rsq_f64 + add"); run<2>("readlane(x2) -> v_fma_f64");
    run<3>("dependent mfma f64 16x16x4"); run<4>("mfma -> v_mul -> mfma"); run<6>("cvt + v_rsq_f32 + cvt + add");
    return 0;
}
