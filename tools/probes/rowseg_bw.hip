// Probe 3: HBM/L2 bandwidth of the GEMM operand access pattern: every workgroup streams a 128-row
// slab of a row-major matrix (row stride ~32.9 KB) in K-slabs of SEG bytes per row.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int SEGD>  // doubles per row per step (16 -> 128 B, 32 -> 256 B, 64 -> 512 B)
__global__ __launch_bounds__(512) void k(const double* __restrict__ A, long lda, long slab_stride, int K, double* out) {
    const double* base = A + (long)blockIdx.x * slab_stride;
    constexpr int TPR = SEGD / 2;          // threads per row (16 B each)
    constexpr int RPP = 512 / TPR;         // rows per pass
    constexpr int PASSES = 128 / RPP;
    const int lr = threadIdx.x / TPR, lc = (threadIdx.x % TPR) * 2;
    double s = 0;
    for (int k0 = 0; k0 < K; k0 += SEGD) {
        double2 v[PASSES];
#pragma unroll
        for (int p = 0; p < PASSES; ++p) v[p] = *(const double2*)(base + (long)(lr + RPP * p) * lda + k0 + lc);
#pragma unroll
        for (int p = 0; p < PASSES; ++p) s += v[p].x + v[p].y;
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}
template <int SEGD>
void run(const double* A, long lda, int nslab, int K, double* out, const char* tag) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<SEGD><<<nslab, 512>>>(A, lda, 128 * lda, K, out);
    hipEventRecord(e0);
    k<SEGD><<<nslab, 512>>>(A, lda, 128 * lda, K, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%s: seg %4d B, %d slabs x 128 rows x K=%d: %.3f ms -> %.2f TB/s\n", tag, SEGD * 8, nslab, K, ms,
           (double)nslab * 128 * K * 8 / ms / 1e9);
}
int main() {
    const long lda = 4112; const int nslab = 4096;  // 4096 slabs x 128 rows = 524288 rows ~ 17 GB
    double *A, *out;
    hipMalloc(&A, (size_t)nslab * 128 * lda * 8); hipMalloc(&out, (size_t)nslab * 512 * 8);
    hipMemset(A, 0, (size_t)nslab * 128 * lda * 8);
    for (int K : {1024, 4096}) {
        run<16>(A, lda, nslab, K, out, "row-major");
        run<32>(A, lda, nslab, K, out, "row-major");
        run<64>(A, lda, nslab, K, out, "row-major");
        run<128>(A, lda, nslab, K, out, "row-major");
    }
    return 0;
}
