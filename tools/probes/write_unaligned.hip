// Write-pattern probe, second part: rows that are NOT 128-byte aligned (N = 3000, row stride N: every other row starts 64
// bytes into a cache line).  What does the dense fill's store pattern cost there, and which re-arrangement helps?
//   mode 0  the fill's pattern: a wave instruction = 16 rows x 16 columns, lane (row g, quarter q) stores 32 B as two 16-B
//           stores (each instruction writes HALF of every 128-B segment; on odd rows the segment straddles two lines)
//   mode 1  the same, the column window of ODD rows shifted by 8 columns (64 B): every 128-B segment is one line
//   mode 2  a wave instruction = 4 rows x 64 columns (512 B of a row per instruction), windows as mode 0
//   mode 3  mode 2 with the odd rows' windows shifted by 8 columns
//   mode 4  a wave instruction = 1 row x 256 columns (2 KB contiguous), window of odd rows shifted by 8 columns
// Workgroup = 64 x 256 tile strip (as k_fill_dense_plain with SPAN = 4), 256 threads.  No reads, no arithmetic.
//   hipcc --offload-arch=gfx950 -O3 write_unaligned.hip -o write_unaligned && ./write_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

__device__ __forceinline__ void st32(double* p, int col, int n) {  // 32 bytes at columns col .. col + 3 (clipped to the row)
    const double2 v = make_double2(1.0, 2.0);
    if (col >= 0 && col + 3 < n) {
        *(double2*)p = v;
        *(double2*)(p + 2) = v;
    } else {
        for (int r = 0; r < 4; ++r)
            if (col + r >= 0 && col + r < n) p[r] = 1.0;
    }
}

__global__ __launch_bounds__(256) void k_write(double* __restrict__ dst, int n, int ld, int mode) {
    const int tr = (n + 63) / 64, tc = (n + 255) / 256 + 1;  // (+1 strip: the shifted windows of the last columns)
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    const int b = id / (tr * tc);
    const int r = id - b * tr * tc;
    const int tm = r / tc, tn = r - tm * tc;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double* base = dst + (size_t)b * n * ld;
    const int R0 = tm * 64, C0 = tn * 256;
    const bool shift = mode == 1 || mode == 3 || mode == 4;
    if (mode <= 1) {
        // wave w: 32 x 32 sub-tiles of four 64 x 64 tiles, as the fill: rows (w >> 1) * 32 + ti * 16 + g, cols (w & 1) * 32 + tj * 16 + 4 q
        const int g = lane & 15, q = lane >> 4;
        for (int t = 0; t < 4; ++t)
            for (int ti = 0; ti < 2; ++ti)
                for (int tj = 0; tj < 2; ++tj) {
                    const int row = R0 + (w >> 1) * 32 + ti * 16 + g;
                    int col = C0 + t * 64 + (w & 1) * 32 + tj * 16 + 4 * q;
                    if (shift && (row & 1)) col -= 8;  // (window of odd rows: 8 columns to the left, the extra strip covers the end)
                    if (row < n) st32(base + (size_t)row * ld + col, col, n);
                }
    } else if (mode <= 3) {
        const int lr = lane >> 4, lc = (lane & 15) * 4;  // 4 rows x 64 columns per instruction
        for (int k = w; k < 16 * 4; k += 4) {             // 16 row groups x 4 column blocks, along the row first
            const int br = k >> 2, bc = k & 3;
            const int row = R0 + br * 4 + lr;
            int col = C0 + bc * 64 + lc;
            if (shift && (row & 1)) col -= 8;
            if (row < n) st32(base + (size_t)row * ld + col, col, n);
        }
    } else {
        for (int k = w; k < 64; k += 4) {  // one row x 256 columns per instruction
            const int row = R0 + k;
            int col = C0 + lane * 4;
            if (row & 1) col -= 8;
            if (row < n) st32(base + (size_t)row * ld + col, col, n);
        }
    }
}

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 128;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int n : {4096, 3000, 3008}) {
        const int ld = n;
        const size_t bytes = (size_t)B * n * ld * 8;
        double* d;
        if (hipMalloc(&d, bytes + 4096) != hipSuccess) return 1;
        for (int mode = 0; mode < 5; ++mode) {
            const int tr = (n + 63) / 64, tc = (n + 255) / 256 + 1;
            const unsigned grid = (unsigned)((size_t)B * tr * tc);
            hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, d, n, ld, mode);
            hipEventRecord(e0, 0);
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_write, dim3(grid), dim3(256), 0, 0, d, n, ld, mode);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            ms /= 3;
            printf("{\"n\": %d, \"ld\": %d, \"mode\": %d, \"ms\": %.3f, \"GBs\": %.0f}\n", n, ld, mode, ms, bytes / ms / 1e6);
        }
        hipFree(d);
    }
    return 0;
}
