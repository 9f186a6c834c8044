// Write-pattern probe for the dense covariance fill: how fast can [B][N][N] doubles be written when a workgroup owns a
// TR x TC tile (row stride N) and a wave instruction covers IR rows x IC columns (IR * IC = 256 doubles, 32 B per lane)?
// No reads, no arithmetic.  hipcc --offload-arch=gfx950 -O3 write_pattern.hip -o write_pattern && ./write_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int xcd = bid & 7, slot = bid >> 3;
    const int q = nblk >> 3, r = nblk & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

// grid: B * (N/TR) * (N/TC) workgroups of 256 threads; tile order: column tiles fastest (stagger: start column depends on row tile)
template <int IR>
__global__ __launch_bounds__(256) void k_write(double* __restrict__ dst, int N, int TR, int TC, int remap, int stagger, int nt_store) {
    constexpr int IC = 256 / IR;  // columns per wave instruction
    const int tiles_r = N / TR, tiles_c = N / TC;
    int id = remap ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int b = id / (tiles_r * tiles_c);
    int r = id - b * tiles_r * tiles_c;
    const int tm = r / tiles_c;
    int tn = r - tm * tiles_c;
    if (stagger) tn = (tn + tm) % tiles_c;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int lr = lane / (IC / 4), lc = (lane % (IC / 4)) * 4;  // lane's row / first column inside an instruction block
    double* base = dst + ((size_t)b * N + (size_t)tm * TR) * N + (size_t)tn * TC;
    const int blocks_c = TC / IC, blocks_r = TR / IR, nblocks = blocks_c * blocks_r;
    const double2 v = make_double2(1.0, 2.0);
    for (int k = w; k < nblocks; k += 4) {  // consecutive blocks of a wave: along the row first
        const int br = k / blocks_c, bc = k - br * blocks_c;
        double* p = base + (size_t)(br * IR + lr) * N + bc * IC + lc;
        if (nt_store) {
            typedef double d2 __attribute__((ext_vector_type(2)));
            const d2 vv = {1.0, 2.0};
            __builtin_nontemporal_store(vv, (d2*)p);
            __builtin_nontemporal_store(vv, (d2*)(p + 2));
        } else {
            *(double2*)p = v;
            *(double2*)(p + 2) = v;
        }
    }
}

int main(int argc, char** argv) {
    const int N = 4096, B = argc > 1 ? atoi(argv[1]) : 128;
    const size_t bytes = (size_t)B * N * N * 8;
    double* d;
    if (hipMalloc(&d, bytes) != hipSuccess) return 1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    struct Cfg { int TR, TC, IR, remap, stagger, nt; };
    std::vector<Cfg> cfgs;
    for (int nt = 0; nt < 2; ++nt) {
        for (int ir : {16, 4, 1}) {
            cfgs.push_back({64, 64, ir, 1, 0, nt});
            cfgs.push_back({64, 256, ir, 1, 0, nt});
            cfgs.push_back({16, 1024, ir, 1, 0, nt});
            cfgs.push_back({16, 4096, ir, 1, 0, nt});
        }
        cfgs.push_back({64, 64, 16, 0, 0, nt});
        cfgs.push_back({64, 64, 16, 1, 1, nt});
        cfgs.push_back({64, 1024, 16, 1, 1, nt});
        cfgs.push_back({64, 1024, 16, 1, 0, nt});
        cfgs.push_back({32, 512, 16, 1, 0, nt});
        cfgs.push_back({128, 128, 16, 1, 0, nt});
        cfgs.push_back({4, 4096, 4, 1, 0, nt});
        cfgs.push_back({1, 4096, 1, 1, 0, nt});
    }
    for (const Cfg& c : cfgs) {
        const unsigned grid = (unsigned)((size_t)B * (N / c.TR) * (N / c.TC));
        auto launch = [&]() {
            if (c.IR == 16) hipLaunchKernelGGL(k_write<16>, dim3(grid), dim3(256), 0, 0, d, N, c.TR, c.TC, c.remap, c.stagger, c.nt);
            else if (c.IR == 4) hipLaunchKernelGGL(k_write<4>, dim3(grid), dim3(256), 0, 0, d, N, c.TR, c.TC, c.remap, c.stagger, c.nt);
            else hipLaunchKernelGGL(k_write<1>, dim3(grid), dim3(256), 0, 0, d, N, c.TR, c.TC, c.remap, c.stagger, c.nt);
        };
        if (c.TR % c.IR || c.TC % (256 / c.IR)) continue;
        launch();
        hipEventRecord(e0, 0);
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 3;
        printf("{\"TR\": %d, \"TC\": %d, \"IR\": %d, \"IC\": %d, \"remap\": %d, \"stagger\": %d, \"nt\": %d, \"wgs\": %u, \"ms\": %.3f, \"GBs\": %.0f}\n",
               c.TR, c.TC, c.IR, 256 / c.IR, c.remap, c.stagger, c.nt, grid, ms, bytes / ms / 1e6);
    }
    return 0;
}
