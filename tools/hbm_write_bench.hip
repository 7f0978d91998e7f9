// Microbenchmark: HBM write throughput (pure fill, and fill + read at 1:3 / 1:1), 1.6 GB per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// MODE 0: f32x4 per lane contiguous; 1: dword per lane, 32 lanes = one 128-B row, two rows per instruction (accumulator layout)
template <int MODE, int NT>
__global__ __launch_bounds__(256) void fill_kernel(float* __restrict__ y, size_t n4, const float* __restrict__ x, int reads_per_write) {
    const size_t stride = (size_t)gridDim.x * 256;
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        for (int r = 0; r < reads_per_write; ++r) { const f32x4 v = ((const f32x4*)x)[(i + (size_t)r * n4) % (n4 * 2)]; s += v.x + v.w; }
        if (MODE == 0) {
            const f32x4 v = {s, 1.f, 2.f, 3.f};
            if (NT) __builtin_nontemporal_store(v, (f32x4*)y + i); else ((f32x4*)y)[i] = v;
        } else {
            // the wave's 64 float4 slots = 16 rows of 32 floats... write as 4 dword stores covering the same 1 KB
            const size_t wbase = (i & ~(size_t)63) * 4;
            const int lane = threadIdx.x & 63;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (NT) __builtin_nontemporal_store(s + q, y + wbase + q * 64 + lane); else y[wbase + q * 64 + lane] = s + q;
            }
        }
    }
}

template <int MODE, int NT>
void run(float* y, size_t n4, const float* x, int rpw, int grid, const char* tag) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int z = 0; z < 2; ++z) hipLaunchKernelGGL((fill_kernel<MODE, NT>), dim3(grid), dim3(256), 0, 0, y, n4, x, rpw);
    CK(hipEventRecord(e0));
    for (int z = 0; z < 10; ++z) hipLaunchKernelGGL((fill_kernel<MODE, NT>), dim3(grid), dim3(256), 0, 0, y, n4, x, rpw);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
    const double bytes = (double)n4 * 16 * (1 + rpw);
    printf("%-26s mode %d nt %d reads/write %d grid %5d: %7.1f us  %5.2f TB/s total (%.2f write)\n", tag, MODE, NT, rpw, grid, ms * 1e3,
           bytes / (ms * 1e-3) / 1e12, (double)n4 * 16 / (ms * 1e-3) / 1e12);
}

int main() {
    const size_t n4 = (size_t)100 << 20;           // 1.6 GB of float4
    float *y, *x;
    CK(hipMalloc(&y, n4 * 16)); CK(hipMalloc(&x, n4 * 32)); CK(hipMemset(x, 0, n4 * 32));
    for (int grid : {2048, 8192, 65536}) {
        run<0, 0>(y, n4, x, 0, grid, "fill f32x4");
        run<0, 1>(y, n4, x, 0, grid, "fill f32x4 nt");
        run<1, 0>(y, n4, x, 0, grid, "fill dword rows");
        run<1, 1>(y, n4, x, 0, grid, "fill dword rows nt");
        run<0, 0>(y, n4, x, 1, grid, "copy 1:1");
        run<0, 1>(y, n4, x, 1, grid, "copy 1:1 nt");
    }
    return 0;
}
