// Microbenchmark: HBM throughput of the K-streaming tile read pattern of a 1x1-conv GEMM.
// A is M x K fp32 row-major; a workgroup owns ROWS rows and walks K in chunks of BKF floats,
// DEPTH chunks in flight (register prefetch), REP workgroups read the same tile (the N/BN
// column tiles of the GEMM).  Prints TB/s of unique bytes per variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int NT>
__device__ __forceinline__ f32x4 ld(const float* p) {
    if constexpr (NT) return __builtin_nontemporal_load((const f32x4*)p);
    else return *(const f32x4*)p;
}

// SAMEXCD: the rep workgroups of a tile get block ids 8 apart (same XCD, adjacent in its dispatch order)
template <int ROWS, int BKF, int DEPTH, int MINB, int NT, int SAMEXCD, int WLOAD>
__global__ __launch_bounds__(256, MINB) void pat_kernel(const float* __restrict__ a, float* __restrict__ out, int M, int K, int rep,
                                                        const float* __restrict__ w) {
    constexpr int QPR = BKF / 4, RPP = 256 / QPR, LD = ROWS / RPP;   // float4 per row, rows per pass, loads per thread per chunk
    int tile = blockIdx.x / rep;
    if constexpr (SAMEXCD) { const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3; tile = (slot / rep) * 8 + xcd; }
    const int ntile = SAMEXCD ? ((blockIdx.x >> 3) % rep) : (blockIdx.x % rep);
    const int tid = threadIdx.x, lc = tid % QPR, lr = tid / QPR;
    const float* base = a + (size_t)tile * ROWS * K + (size_t)lr * K + lc * 4;
    f32x4 r[DEPTH][LD];
    const int nk = K / BKF;
    f32x4 s = {0, 0, 0, 0};
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
#pragma unroll
        for (int i = 0; i < LD; ++i)
            r[d][i] = d < nk ? ld<NT>(base + (size_t)i * RPP * K + d * BKF) : (f32x4){0, 0, 0, 0};
    for (int kc = 0; kc < nk; kc += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
            for (int i = 0; i < LD; ++i) s += r[d][i];
            if constexpr (WLOAD) {   // 128 x BKF fp16 weights of this n-tile: BKF * 256 B per chunk, L2-resident
                const float* wp = w + ((size_t)ntile * 128 * K + (size_t)(kc + d) * BKF * 128) / 2;
#pragma unroll
                for (int i = 0; i < BKF / 16; ++i) s += *(const f32x4*)(wp + (i * 256 + tid) * 4);
            }
            const int kn = kc + d + DEPTH;
            if (kn < nk) {
#pragma unroll
                for (int i = 0; i < LD; ++i) r[d][i] = ld<NT>(base + (size_t)i * RPP * K + kn * BKF);
            }
        }
    }
    if (s.x + s.y + s.z + s.w == 12345.678f) out[blockIdx.x] = s.x;
}

template <int ROWS, int BKF, int DEPTH, int MINB, int NT = 1, int SAMEXCD = 0, int WLOAD = 0>
void run(const float* a, float* out, int M, int K, int rep, const char* tag, const float* w = nullptr) {
    const int grid = M / ROWS * rep;
    if (SAMEXCD && (M / ROWS) % 8) { printf("bad tiles\n"); exit(1); }
    if (M % ROWS || K % BKF || (size_t)M * K > (size_t)200704 * 1024 || grid > (1 << 22)) { printf("bad shape\n"); exit(1); }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int z = 0; z < 2; ++z) hipLaunchKernelGGL((pat_kernel<ROWS, BKF, DEPTH, MINB, NT, SAMEXCD, WLOAD>), dim3(grid), dim3(256), 0, 0, a, out, M, K, rep, w);
    CK(hipEventRecord(e0));
    const int it = 10;
    for (int z = 0; z < it; ++z) hipLaunchKernelGGL((pat_kernel<ROWS, BKF, DEPTH, MINB, NT, SAMEXCD, WLOAD>), dim3(grid), dim3(256), 0, 0, a, out, M, K, rep, w);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= it;
    printf("%-28s nt %d samexcd %d w %d rows %3d bk %3d depth %d minb %d rep %d K %4d: %7.1f us  %5.2f TB/s unique\n", tag, NT, SAMEXCD, WLOAD, ROWS, BKF, DEPTH, MINB, rep, K, ms * 1e3,
           (double)M * K * 4 / (ms * 1e-3) / 1e12);
}

int main() {
    const int M = 200704;
    float *a, *out, *w;
    CK(hipMalloc(&a, (size_t)M * 1024 * 4 + (1 << 20))); CK(hipMalloc(&out, 1 << 24)); CK(hipMalloc(&w, 8 << 20));
    CK(hipMemset(a, 0, (size_t)M * 1024 * 4)); CK(hipMemset(w, 0, 8 << 20));
    for (int K : {1024, 256}) {
        const int Mk = K == 1024 ? M : M * 4;   // same bytes
        const int reps[2] = {1, K == 1024 ? 2 : 8};
        for (int rep : reps) {
            run<128, 32, 2, 3, 1, 0, 0>(a, out, Mk, K, rep, "nt, round robin", w);
            run<128, 32, 2, 3, 0, 0, 0>(a, out, Mk, K, rep, "plain, round robin", w);
            run<128, 32, 2, 3, 0, 1, 0>(a, out, Mk, K, rep, "plain, same xcd", w);
            run<128, 32, 2, 3, 1, 1, 0>(a, out, Mk, K, rep, "nt, same xcd", w);
            run<128, 32, 2, 3, 0, 1, 1>(a, out, Mk, K, rep, "plain, same xcd, +W", w);
            run<128, 64, 2, 3, 0, 1, 1>(a, out, Mk, K, rep, "plain, same xcd, +W", w);
            run<64, 64, 2, 4, 0, 1, 1>(a, out, Mk, K, rep, "plain, same xcd, +W", w);
        }
    }
    return 0;
}
