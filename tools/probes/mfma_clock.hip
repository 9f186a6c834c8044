// Probe 2: cycles per v_mfma_f64_16x16x4_f64 (s_memtime = shader clock) and the sustained shader
// clock (s_memtime ticks / wall_clock64 ticks @100 MHz) under a pure-MFMA load.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(256) void k(double* out, long long* tm, int iters) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 0.5;
    long long w0 = wall_clock64();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    long long w1 = wall_clock64();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { tm[0] = t1 - t0; tm[1] = w1 - w0; }
}
template <int NACC>
void run(int nblk, int iters, const char* tag) {
    double* out; long long* tm; long long h[2];
    hipMalloc(&out, sizeof(double) * nblk * 256); hipMalloc(&tm, 16);
    k<NACC><<<nblk, 256>>>(out, tm, iters);
    hipDeviceSynchronize();
    hipMemcpy(h, tm, 16, hipMemcpyDeviceToHost);
    printf("%s: NACC=%d blocks=%d: %.1f memtime-ticks/MFMA, memtime/wall = %.3f (x100MHz => %.0f MHz)\n", tag, NACC, nblk,
           (double)h[0] / ((double)iters * NACC), (double)h[0] / h[1], 100.0 * h[0] / h[1]);
    hipFree(out); hipFree(tm);
}
int main() {
    run<8>(1, 20000, "one WG on the chip");
    run<4>(256 * 1, 20000, "1 wave/SIMD");
    run<4>(256 * 2, 20000, "2 waves/SIMD");
    run<4>(256 * 3, 20000, "3 waves/SIMD");
    run<4>(256 * 4, 20000, "4 waves/SIMD");
    run<4>(256 * 6, 20000, "6 waves/SIMD");
    run<4>(256 * 8, 20000, "8 waves/SIMD");
    run<2>(256 * 8, 20000, "8 waves/SIMD, 2 acc");
    run<1>(256 * 8, 20000, "8 waves/SIMD, 1 acc (dependent chain)");
    run<1>(256 * 1, 20000, "1 wave/SIMD, 1 acc (dependent chain)");
    return 0;
}
