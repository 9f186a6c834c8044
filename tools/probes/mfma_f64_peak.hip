// Micro-probe: sustained v_mfma_f64_16x16x4_f64 rate and HBM stream-write rate on this MI355X.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_f64_peak.hip -o gpurun_out/mfma_probe && ./gpurun_out/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters, double a0, double b0) {
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_write(double2* p, size_t n2, double v) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n2; i += stride) p[i] = make_double2(v, v);
}
__global__ __launch_bounds__(256) void k_copy(const double2* a, double2* p, size_t n2) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    for (; i < n2; i += stride) p[i] = a[i];
}

template <int NACC>
void run_mfma(int wg_per_cu, int iters) {
    double* out;
    const int nblk = 256 * wg_per_cu;
    hipMalloc(&out, sizeof(double) * nblk * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k_mfma<NACC><<<nblk, 256>>>(out, 10, 1.0, 1.0);
    hipEventRecord(e0);
    k_mfma<NACC><<<nblk, 256>>>(out, iters, 1.0, 1.0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)nblk * 4 * iters * NACC * 2.0 * 16 * 16 * 4;
    printf("mfma_f64 16x16x4: %d acc/wave, %d WG(4 waves)/CU, %d iters: %.3f ms -> %.1f TFLOP/s\n", NACC,
           wg_per_cu, iters, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    run_mfma<4>(1, 20000);
    run_mfma<4>(2, 20000);
    run_mfma<16>(1, 5000);
    run_mfma<16>(2, 5000);
    run_mfma<16>(2, 50000);  // long run: sustained clock
    const size_t bytes = (size_t)8 << 30;
    double2 *p, *q;
    hipMalloc(&p, bytes);
    hipMalloc(&q, bytes);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        k_write<<<2048, 256>>>(p, bytes / 16, 1.0);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("stream write 8 GiB: %.3f ms -> %.2f TB/s\n", ms, bytes / ms / 1e9);
        hipEventRecord(e0);
        k_copy<<<2048, 256>>>(p, q, bytes / 16);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        printf("stream copy 8 GiB (r+w 16 GiB): %.3f ms -> %.2f TB/s\n", ms, 2.0 * bytes / ms / 1e9);
    }
    return 0;
}
