// Probe 4: can fp64 MFMA and fp64 VALU FMA streams run concurrently on a CU (separate pipes)?
// Blocks of 512 threads: waves [0, nm) run v_mfma_f64_16x16x4_f64 loops, the rest run v_fma_f64 loops.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k(double* out, int iters, int nm) {
    const int w = threadIdx.x >> 6;
    double s = 0;
    if (w < nm) {
        d4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = (d4){0, 0, 0, 0};
        double a = 1.0 + threadIdx.x * 1e-9, b = 0.5;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        double acc[32];
        for (int i = 0; i < 32; ++i) acc[i] = i;
        double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9;
        // 8 MFMA = 8*2048 flops per wave-iter; match flops: 32 v_fma (wave64: 128 flops each) x 4 = 16384 flops
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 32; ++i) acc[i] = __builtin_fma(acc[i], a, b);
        }
        for (int i = 0; i < 32; ++i) s += acc[i];
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
void run(int nm, int blocks_per_cu, int iters) {
    double* out; const int nblk = 256 * blocks_per_cu;
    hipMalloc(&out, sizeof(double) * nblk * 512);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<nblk, 512>>>(out, 10, nm);
    hipEventRecord(e0); k<<<nblk, 512>>>(out, iters, nm); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double fl = (double)nblk * 8 * iters * 16384.0;
    printf("MFMA waves %d / VALU waves %d per block, %d blocks/CU: %.3f ms -> %.1f TFLOP/s total (MFMA part %.1f, VALU part %.1f)\n",
           nm, 8 - nm, blocks_per_cu, ms, fl / ms / 1e9, fl / ms / 1e9 * nm / 8.0, fl / ms / 1e9 * (8 - nm) / 8.0);
    hipFree(out);
}
int main() {
    run(8, 2, 4000); run(0, 2, 4000); run(4, 2, 4000); run(6, 2, 4000); run(4, 4, 4000); run(5, 4, 4000); run(6,4,4000);
    return 0;
}
