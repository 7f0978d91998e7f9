// The GEMM-shaped parts of the debiasing-adapter step (final_main.py:160-174 forward, its backward, 455-466 step body)
// for the reference's shapes: hidden width H = 128 (final_main.py:241), D = 512 / 768 / 1024, a batch of 4 ... 8192 rows.
//
// At B = 256 the whole step is 0.34 GFLOP and 7 MB: nothing in it is throughput-bound, the step is latency- and launch-
// bound.  Through the general implicit-GEMM kernel its five products ran as grids of 8 - 64 tiles with K loops of up to
// 64 barrier-separated chunks: 95 of the step's 135 us (rocprofv3, tools/bench_adapter_step.py).  These kernels cut every
// product so that >= 64 workgroups work on it, fuse what is elementwise into the product that consumes it, and keep
// every sum in a fixed order (no atomics): results do not depend on the launch geometry's timing and the one-call step
// equals the autograd path bit for bit (both run these kernels).
//
//   fc1_partial      hpart[ks][b][:] = x[b][128 ks : 128 ks + 128] . W1[:, same]^T     grid (B / 32, 2, D / 128); the partials
//                    live in the z buffer (B x D floats = D / 128 slices of B x 128), which is written later
//   bn_stats         h = b1 + sum_ks hpart (fixed order), column mean / biased variance over the batch (two passes),
//                    running statistics with the unbiased variance, num_batches_tracked          grid H / 4
//   fc2              r = relu(bn(h)) (stored for the backward), z = r . W2^T + b2                grid (B / 32, D / 64)
//   bwd2             dW2 = dz^T r, db2 = colsum dz (blocks < D / 32) and drpart[ks] = dz[:, slice ks] . W2[slice ks]
//                    (the other blocks) in ONE launch: both only need dz, r and W2
//   bn_bwd           dr = sum_ks drpart, dhn = dr * (bn(h) > 0), dbeta / dgamma column sums, dh                  grid H / 4
//   bwd1             dW1 = dh^T x, db1 = colsum dh                                                 grid D / 32
//
// All fp32 on the vector ALUs (exact fp32 products, like the fp32-input MFMA the general kernel uses): a 32 x 128 output
// tile per workgroup, 4 x 4 register tiles, operands through LDS rows padded to 132 floats (16-B aligned, conflict-free
// ds_read_b128 for 16 consecutive rows).  Bound: launch latency (1.2 us of HBM time at B = 256).
#include "common.h"

namespace {

constexpr int LP = 132;                    // LDS row pitch in floats

constexpr int TB = 16, TI = TB / 8;        // rows of a tile product's output tile, rows per thread

// acc[i][j] += sum_k a[row_i][k] * w[col_j][k] over 128 k, both operands as [rows][LP] in LDS; rows ty + 8 i, cols tx + 32 j
template <int RI, int CJ>
__device__ __forceinline__ void tile_kk(const float* __restrict__ as, const float* __restrict__ ws, int tx, int ty, float (&acc)[RI][CJ]) {
#pragma unroll 4
    for (int k4 = 0; k4 < 32; ++k4) {
        f32x4 a[RI], w[CJ];
#pragma unroll
        for (int i = 0; i < RI; ++i) a[i] = *(const f32x4*)(as + (ty + 8 * i) * LP + 4 * k4);
#pragma unroll
        for (int j = 0; j < CJ; ++j) w[j] = *(const f32x4*)(ws + (tx + 32 * j) * LP + 4 * k4);
#pragma unroll
        for (int i = 0; i < RI; ++i)
#pragma unroll
            for (int j = 0; j < CJ; ++j) {
                acc[i][j] = fmaf(a[i][0], w[j][0], acc[i][j]);
                acc[i][j] = fmaf(a[i][1], w[j][1], acc[i][j]);
                acc[i][j] = fmaf(a[i][2], w[j][2], acc[i][j]);
                acc[i][j] = fmaf(a[i][3], w[j][3], acc[i][j]);
            }
    }
}

// rows [r0, r0 + TB) x columns [c0, c0 + 128) of a row-major [n_rows][ld] matrix -> LDS [TB][LP] (rows past n_rows: zeros)
__device__ __forceinline__ void load_tileTB(const float* __restrict__ src, long long ld, int r0, int n_rows, int c0, float* __restrict__ dst, int tid) {
#pragma unroll
    for (int i = 0; i < TB / 8; ++i) {
        const int q = tid + 256 * i, r = q >> 5, c = q & 31;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r0 + r < n_rows) v = *(const f32x4*)(src + (long long)(r0 + r) * ld + c0 + 4 * c);
        *(f32x4*)(dst + r * LP + 4 * c) = v;
    }
}
// rows [r0, r0 + 32) x columns [c0, c0 + 128) of a row-major [n_rows][ld] matrix -> LDS [32][LP] (rows past n_rows: zeros)
__device__ __forceinline__ void load_tile32(const float* __restrict__ src, long long ld, int r0, int n_rows, int c0, float* __restrict__ dst, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = tid + 256 * i, r = q >> 5, c = q & 31;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (r0 + r < n_rows) v = *(const f32x4*)(src + (long long)(r0 + r) * ld + c0 + 4 * c);
        *(f32x4*)(dst + r * LP + 4 * c) = v;
    }
}
// rows [r0, r0 + 64) x columns [c0, c0 + 128) -> LDS [64][LP]
__device__ __forceinline__ void load_tile64(const float* __restrict__ src, long long ld, int r0, int c0, float* __restrict__ dst, int tid) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = tid + 256 * i, r = q >> 5, c = q & 31;
        *(f32x4*)(dst + r * LP + 4 * c) = *(const f32x4*)(src + (long long)(r0 + r) * ld + c0 + 4 * c);
    }
}
// rows [r0, r0 + 128) x columns [c0, c0 + 128) -> LDS [128][LP]
__device__ __forceinline__ void load_tile128(const float* __restrict__ src, long long ld, int r0, int c0, float* __restrict__ dst, int tid) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int q = tid + 256 * i, r = q >> 5, c = q & 31;
        *(f32x4*)(dst + r * LP + 4 * c) = *(const f32x4*)(src + (long long)(r0 + r) * ld + c0 + 4 * c);
    }
}

__global__ __launch_bounds__(256) void fc1_partial_kernel(const float* __restrict__ x, const float* __restrict__ w1, float* __restrict__ part,
                                                          int B, int D) {
    // tile = 32 rows x 64 hidden units over a 128-deep K slice (24 KB + 32 KB of operands per workgroup)
    __shared__ __attribute__((aligned(16))) float as[32 * LP];
    __shared__ __attribute__((aligned(16))) float ws[64 * LP];
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
    const int b0 = blockIdx.x * 32, h0 = blockIdx.y * 64, ks = blockIdx.z;
    load_tile32(x, D, b0, B, ks * 128, as, tid);
    load_tile64(w1, D, h0, ks * 128, ws, tid);
    __syncthreads();
    float acc[4][2] = {};
    tile_kk<4, 2>(as, ws, tx, ty, acc);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int b = b0 + ty + 8 * i;
        if (b < B)
#pragma unroll
            for (int j = 0; j < 2; ++j) part[((long long)ks * B + b) * 128 + h0 + tx + 32 * j] = acc[i][j];
    }
}

// 4 columns x 64 row-lanes per workgroup (thread = column tid & 3, row-lane tid >> 2); sums over the batch: per row-lane
// sequentially, then a fixed butterfly over the 16 row-lanes of a wave (lane bits 2 .. 5), then the 4 waves in order
__device__ __forceinline__ float lanes64_sum(float (*red)[4], int c, int rl, float v) {
#pragma unroll
    for (int o = 4; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) < 4) red[threadIdx.x >> 6][c] = v;
    __syncthreads();
    (void)rl;
    return ((red[0][c] + red[1][c]) + red[2][c]) + red[3][c];
}

// v[u] += sum_ks part[ks][b + 64 u][j] (ks in order) for the four rows of a row-lane: the loads of eight slices x four rows are
// issued together (a loop that adds each load before issuing the next pays one memory round trip per slice)
__device__ __forceinline__ void ksum4(const float* __restrict__ part, int KS, int B, int b, int j, float (&v)[4]) {
    long long off[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) off[u] = (long long)(b + 64 * u < B ? b + 64 * u : B - 1) * 128 + j;       // clamped: branch-free loads
    for (int k0 = 0; k0 < KS; k0 += 8) {
        float t[8][4];
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            const long long base = (long long)(k0 + kk < KS ? k0 + kk : KS - 1) * B * 128;
#pragma unroll
            for (int u = 0; u < 4; ++u) t[kk][u] = part[base + off[u]];
        }
#pragma unroll
        for (int kk = 0; kk < 8; ++kk)
            if (k0 + kk < KS)
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] += t[kk][u];
    }
}

__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ part, int KS, const float* __restrict__ b1, float* __restrict__ h,
                                                       int B, float eps, float momentum, float* __restrict__ mean_o, float* __restrict__ invstd_o,
                                                       float* __restrict__ rmean, float* __restrict__ rvar, long long* __restrict__ nbt) {
    __shared__ float red[64][4];
    const int c = threadIdx.x & 3, rl = threadIdx.x >> 2, j = blockIdx.x * 4 + c;
    const float bj = b1[j];
    float s = 0.f;
    float v[4];
    // four rows of this row-lane at a time: 4 KS independent loads in flight (the loop is latency-bound, not bandwidth-bound)
    for (int b = rl; b < B; b += 256) {
        v[0] = v[1] = v[2] = v[3] = bj;
        ksum4(part, KS, B, b, j, v);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (b + 64 * u < B) { h[(long long)(b + 64 * u) * 128 + j] = v[u]; s += v[u]; }
    }
    const float mean = lanes64_sum(red, c, rl, s) / (float)B;
    float q = 0.f;
    if (B <= 256) {                                          // the only row group is still in registers
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (rl + 64 * u < B) { const float d = v[u] - mean; q = fmaf(d, d, q); }
    } else {
#pragma unroll 4
        for (int b = rl; b < B; b += 64) { const float d = h[(long long)b * 128 + j] - mean; q = fmaf(d, d, q); }
    }
    const float var = lanes64_sum(red, c, rl, q) / (float)B;
    if (rl == 0) {
        mean_o[j] = mean;
        invstd_o[j] = rsqrtf(var + eps);
        if (rmean) rmean[j] = (1.f - momentum) * rmean[j] + momentum * mean;
        if (rvar) rvar[j] = (1.f - momentum) * rvar[j] + momentum * (var * (float)B / (float)(B - 1));
    }
    if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;
}

// eval mode: h = b1 + sum_ks part (no statistics)
__global__ __launch_bounds__(256) void fc1_reduce_kernel(const float* __restrict__ part, int KS, const float* __restrict__ b1, float* __restrict__ h,
                                                         long long total, int B) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i & 127);
    const long long b = i >> 7;
    float v = b1[j];
    for (int ks = 0; ks < KS; ++ks) v += part[((long long)ks * B + b) * 128 + j];
    h[i] = v;
}

__global__ __launch_bounds__(256) void fc2_kernel(const float* __restrict__ h, const float* __restrict__ mean, const float* __restrict__ invstd,
                                                  int var_mode, float eps, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                  const float* __restrict__ w2, const float* __restrict__ b2, float* __restrict__ r,
                                                  float* __restrict__ z, int B, int D) {
    // tile = 32 rows x 64 output columns, K = H = 128 (the whole reduction)
    __shared__ __attribute__((aligned(16))) float as[32 * LP];
    __shared__ __attribute__((aligned(16))) float ws[64 * LP];
    const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
    const int b0 = blockIdx.x * 32, d0 = blockIdx.y * 64;
    // r tile = relu(bn(h tile)): the expression of the stand-alone BatchNorm + ReLU kernel (bn1d_relu_kernel)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int q = tid + 256 * i, rr = q >> 5, c = q & 31;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (b0 + rr < B) {
            f32x4 is = ((const f32x4*)invstd)[c];
            if (var_mode) {
#pragma unroll
                for (int k = 0; k < 4; ++k) is[k] = rsqrtf(is[k] + eps);
            }
            v = (*(const f32x4*)(h + (long long)(b0 + rr) * 128 + 4 * c) - ((const f32x4*)mean)[c]) * is * ((const f32x4*)gamma)[c] + ((const f32x4*)beta)[c];
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
            if (blockIdx.y == 0) *(f32x4*)(r + (long long)(b0 + rr) * 128 + 4 * c) = v;
        }
        *(f32x4*)(as + rr * LP + 4 * c) = v;
    }
    load_tile64(w2, 128, d0, 0, ws, tid);                  // W2 [D][128]: rows d0 .. d0 + 63
    __syncthreads();
    float acc[4][2] = {};
    tile_kk<4, 2>(as, ws, tx, ty, acc);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int b = b0 + ty + 8 * i;
        if (b < B)
#pragma unroll
            for (int j = 0; j < 2; ++j) z[(long long)b * D + d0 + tx + 32 * j] = acc[i][j] + b2[d0 + tx + 32 * j];
    }
}

// blocks [0, NS D / 32): split s = bid / (D / 32) of the batch: dw2part[s][d0 .. d0 + 31][:] = sum_{b in split} dz[b][d] r[b][:],
//                         db2part[s][d] = sum_{b in split} dz[b][d]           (summed over s, in order, by grad_reduce blocks)
// the other blocks:       drpart[ks][b0 .. b0 + 31][:] = dz[b][128 ks ..] . W2[128 ks ..][:]
// one more block, when loss_mean is given: the batch mean of the per-row losses (mean_reduce_kernel's statements)
__global__ __launch_bounds__(256) void bwd2_kernel(const float* __restrict__ dz, const float* __restrict__ r, const float* __restrict__ w2,
                                                   float* __restrict__ dw2part, float* __restrict__ db2part, float* __restrict__ drpart, int B, int D,
                                                   int NS, int RB, const float* __restrict__ loss_rows, float* __restrict__ loss_mean) {
    __shared__ __attribute__((aligned(16))) float s0[32 * LP];
    __shared__ __attribute__((aligned(16))) float s1[128 * LP];
    const int tid = threadIdx.x;
    const int nd = D / 32;
    if (loss_mean && blockIdx.x == gridDim.x - 1) {
        float s = 0.f;
        for (int i = threadIdx.x; i < B; i += 256) s += loss_rows[i];
        s = wave_sum(s);
        if ((threadIdx.x & 63) == 0) s0[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) *loss_mean = ((s0[0] + s0[1]) + (s0[2] + s0[3])) / (float)B;
        return;
    }
    if ((int)blockIdx.x < nd * NS) {
        const int sp = blockIdx.x / nd, d0 = (blockIdx.x % nd) * 32, td = tid & 7, th = tid >> 3;   // register tile: d = 4 td .., h = 4 th ..
        const int bb = sp * RB, be = bb + RB < B ? bb + RB : B;
        float acc[4][4] = {};
        f32x4 cs = {0.f, 0.f, 0.f, 0.f};
        float* dzs = s1;                                                         // [32 b][36]
        for (int bc = bb; bc < be; bc += 32) {
            __syncthreads();
            {   // dz chunk [32 b][32 d]: one float4 per thread
                const int rr = tid >> 3, c = tid & 7;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (bc + rr < be) v = *(const f32x4*)(dz + (long long)(bc + rr) * D + d0 + 4 * c);
                *(f32x4*)(dzs + rr * 36 + 4 * c) = v;
            }
            load_tile32(r, 128, bc, be, 0, s0, tid);                             // r chunk [32 b][128 h]
            __syncthreads();
#pragma unroll 8
            for (int b = 0; b < 32; ++b) {
                const f32x4 dv = *(const f32x4*)(dzs + b * 36 + 4 * td), rv = *(const f32x4*)(s0 + b * LP + 4 * th);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(dv[i], rv[j], acc[i][j]);
                cs += dv;
            }
        }
        float* o = dw2part + (long long)sp * D * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            *(f32x4*)(o + (long long)(d0 + 4 * td + i) * 128 + 4 * th) = (f32x4){acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
        if (th == 0) *(f32x4*)(db2part + (long long)sp * D + d0 + 4 * td) = cs;
    } else {
        const int bid = blockIdx.x - nd * NS, nbt = (B + TB - 1) / TB, ks = bid / nbt, b0 = (bid % nbt) * TB;
        const int tx = tid & 31, ty = tid >> 5;
        load_tileTB(dz, D, b0, B, ks * 128, s0, tid);                            // dz[b][d slice]
        load_tile128(w2, 128, ks * 128, 0, s1, tid);                             // W2[d slice][h]: rows are the REDUCTION index
        __syncthreads();
        // dr[b][h] = sum_d dz[b][d] W2[d][h]; rows ty + 8 i, columns 4 tx .. 4 tx + 3
        float acc[TI][4] = {};
#pragma unroll 4
        for (int k4 = 0; k4 < 32; ++k4) {
            f32x4 a[TI], w[4];
#pragma unroll
            for (int i = 0; i < TI; ++i) a[i] = *(const f32x4*)(s0 + (ty + 8 * i) * LP + 4 * k4);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) w[kk] = *(const f32x4*)(s1 + (4 * k4 + kk) * LP + 4 * tx);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i][kk], w[kk][j], acc[i][j]);
        }
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            const int b = b0 + ty + 8 * i;
            if (b < B) *(f32x4*)(drpart + ((long long)ks * B + b) * 128 + 4 * tx) = (f32x4){acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
        }
    }
}

// train-mode BatchNorm backward, 4 columns x 64 row-lanes per workgroup:
// dr = sum_ks drpart; dhn = dr * (gamma xhat + beta > 0); dbeta = sum_b dhn; dgamma = sum_b dhn xhat;
// dh = gamma invstd (dhn - dbeta / B - xhat dgamma / B)
// (blocks >= 32 of the same launch sum the batch splits of dW2 / db2, which only bwd2 precedes)
__global__ __launch_bounds__(256) void bn_bwd_kernel(const float* __restrict__ drpart, int KS, const float* __restrict__ h, const float* __restrict__ mean,
                                                     const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dh, int B,
                                                     const float* __restrict__ dw2part, float* __restrict__ dw2, long long nw4,
                                                     const float* __restrict__ db2part, float* __restrict__ db2, long long nb4, int NS) {
    __shared__ float red[64][4];
    if (blockIdx.x >= 32) {
        const long long i = (long long)(blockIdx.x - 32) * 256 + threadIdx.x;
        const float* p; float* o; long long n4, k;
        if (i < nw4) { p = dw2part; o = dw2; n4 = nw4; k = i; }
        else if (i < nw4 + nb4) { p = db2part; o = db2; n4 = nb4; k = i - nw4; }
        else return;
        f32x4 v = ((const f32x4*)p)[k];
        for (int sidx = 1; sidx < NS; ++sidx) v += ((const f32x4*)p)[(long long)sidx * n4 + k];
        ((f32x4*)o)[k] = v;
        return;
    }
    const int c = threadIdx.x & 3, rl = threadIdx.x >> 2, j = blockIdx.x * 4 + c;
    const float mu = mean[j], is = invstd[j], ga = gamma[j], be = beta[j];
    float sb = 0.f, sg = 0.f;
    float dn[4] = {0.f, 0.f, 0.f, 0.f}, xh[4] = {0.f, 0.f, 0.f, 0.f};
    for (int b = rl; b < B; b += 256) {
        float d[4] = {0.f, 0.f, 0.f, 0.f}, hv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (b + 64 * u < B) hv[u] = h[(long long)(b + 64 * u) * 128 + j];
        ksum4(drpart, KS, B, b, j, d);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (b + 64 * u < B) {
                xh[u] = (hv[u] - mu) * is;
                dn[u] = (fmaf(ga, xh[u], be) > 0.f) ? d[u] : 0.f;
                if (B > 256) dh[(long long)(b + 64 * u) * 128 + j] = dn[u];   // dhn for now; finished below by the same thread
                sb += dn[u];
                sg = fmaf(dn[u], xh[u], sg);
            }
    }
    const float db = lanes64_sum(red, c, rl, sb), dg = lanes64_sum(red, c, rl, sg);
    if (rl == 0) { dbeta[j] = db; dgamma[j] = dg; }
    const float invB = 1.f / (float)B;
    if (B <= 256) {                                          // the only row group is still in registers
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (rl + 64 * u < B) dh[(long long)(rl + 64 * u) * 128 + j] = ga * is * (dn[u] - db * invB - xh[u] * dg * invB);
    } else {
        for (int b = rl; b < B; b += 64) {
            const float xhb = (h[(long long)b * 128 + j] - mu) * is;
            const float d = dh[(long long)b * 128 + j];
            dh[(long long)b * 128 + j] = ga * is * (d - db * invB - xhb * dg * invB);
        }
    }
}

// split s = blockIdx.y of the batch: dw1part[s][:, d0 .. d0 + 31] = sum_{b in split} dh[b][:]^T x[b][d0 ..]; blockIdx.x == 0 also
// db1part[s] = sum_{b in split} dh[b][:]
__global__ __launch_bounds__(256) void bwd1_kernel(const float* __restrict__ dh, const float* __restrict__ x, float* __restrict__ dw1part,
                                                   float* __restrict__ db1part, int B, int D, int RB) {
    __shared__ __attribute__((aligned(16))) float dhs[32 * LP];
    __shared__ __attribute__((aligned(16))) float xs[32 * 36];
    const int tid = threadIdx.x, td = tid & 7, th = tid >> 3;                    // register tile: h = 4 th .., d = 4 td ..
    const int d0 = blockIdx.x * 32, sp = blockIdx.y;
    const int bb = sp * RB, be = bb + RB < B ? bb + RB : B;
    float acc[4][4] = {};
    f32x4 cs = {0.f, 0.f, 0.f, 0.f};
    for (int bc = bb; bc < be; bc += 32) {
        __syncthreads();
        {
            const int rr = tid >> 3, c = tid & 7;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (bc + rr < be) v = *(const f32x4*)(x + (long long)(bc + rr) * D + d0 + 4 * c);
            *(f32x4*)(xs + rr * 36 + 4 * c) = v;
        }
        load_tile32(dh, 128, bc, be, 0, dhs, tid);
        __syncthreads();
#pragma unroll 8
        for (int b = 0; b < 32; ++b) {
            const f32x4 hv = *(const f32x4*)(dhs + b * LP + 4 * th), xv = *(const f32x4*)(xs + b * 36 + 4 * td);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(hv[i], xv[j], acc[i][j]);
            cs += hv;
        }
    }
    float* o = dw1part + (long long)sp * 128 * D;
#pragma unroll
    for (int i = 0; i < 4; ++i)
        *(f32x4*)(o + (long long)(4 * th + i) * D + d0 + 4 * td) = (f32x4){acc[i][0], acc[i][1], acc[i][2], acc[i][3]};
    if (blockIdx.x == 0 && td == 0) *(f32x4*)(db1part + (long long)sp * 128 + 4 * th) = cs;
}

// out[i] = sum_s part[s][i] in the order s = 0, 1, ...: two tensors per launch (a weight gradient and its bias gradient)
__global__ __launch_bounds__(256) void grad_reduce_kernel(const float* __restrict__ pa, float* __restrict__ oa, long long na4,
                                                          const float* __restrict__ pb, float* __restrict__ ob, long long nb4, int NS) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const float* p; float* o; long long n4, k;
    if (i < na4) { p = pa; o = oa; n4 = na4; k = i; }
    else if (i < na4 + nb4) { p = pb; o = ob; n4 = nb4; k = i - na4; }
    else return;
    f32x4 v = ((const f32x4*)p)[k];
    for (int sidx = 1; sidx < NS; ++sidx) v += ((const f32x4*)p)[(long long)sidx * n4 + k];
    ((f32x4*)o)[k] = v;
}

}  // namespace

// ---- launchers (adapter_ops.hip composes dbmm_adapter_fwd / dbmm_adapter_bwd from them) -----------------------------------
bool dbmm_adapter_fast_shape(int64_t B, int64_t D, int64_t H) { return H == 128 && D >= 128 && (D % 128) == 0 && B >= 2 && B <= (1 << 20); }

// forward: h (pre-BatchNorm), mean / invstd (train), r, z.  `z` doubles as the scratch of the K-split partial sums.
int dbmm_adapter_fwd_fast(const float* x, const float* w1, const float* b1, const float* gamma, const float* beta, float* running_mean,
                          float* running_var, int64_t* nbt, const float* w2, const float* b2, float* h, float* mean, float* invstd, float* r,
                          float* z, int64_t B, int64_t D, int train, float eps, float momentum, hipStream_t s) {
    const int KS = (int)(D / 128), nb = (int)((B + 31) / 32);
    hipLaunchKernelGGL(fc1_partial_kernel, dim3(nb, 2, KS), dim3(256), 0, s, x, w1, z, (int)B, (int)D);
    DBMM_CHECK_LAUNCH();
    if (train) {
        hipLaunchKernelGGL(bn_stats_kernel, dim3(32), dim3(256), 0, s, (const float*)z, KS, b1, h, (int)B, eps, momentum, mean, invstd, running_mean,
                           running_var, (long long*)nbt);
    } else {
        const long long total = B * 128;
        hipLaunchKernelGGL(fc1_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)z, KS, b1, h, total, (int)B);
    }
    DBMM_CHECK_LAUNCH();
    hipLaunchKernelGGL(fc2_kernel, dim3(nb, (unsigned)(D / 64)), dim3(256), 0, s, (const float*)h, train ? (const float*)mean : (const float*)running_mean,
                       train ? (const float*)invstd : (const float*)running_var, train ? 0 : 1, eps, gamma, beta, w2, b2, r, z, (int)B, (int)D);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

// batch splits of the two weight-gradient products: at most 16, each a multiple of 32 rows
static void bwd_split(int64_t B, int* NS, int* RB) {
    const int nb = (int)((B + 31) / 32);
    *NS = nb < 16 ? nb : 16;
    *RB = ((nb + *NS - 1) / *NS) * 32;
    *NS = (int)((B + *RB - 1) / *RB);
}
size_t dbmm_adapter_bwd_fast_floats(int64_t B, int64_t D) {
    int NS, RB; bwd_split(B, &NS, &RB);
    return (size_t)B * D + (size_t)NS * (2 * 128 * D + D + 128);
}

// backward: dW2, db2, dgamma, dbeta, dW1, db1; dh is left in `dh`; `scratch` = dbmm_adapter_bwd_fast_floats(B, D) floats:
// K-split partial sums of dr [D / 128][B][128] | batch-split partial sums of dW2, db2, dW1, db1
int dbmm_adapter_bwd_fast(const float* x, const float* dz, const float* h, const float* mean, const float* invstd, const float* r, const float* gamma,
                          const float* beta, const float* w2, float* dw1, float* db1, float* dgamma, float* dbeta, float* dw2, float* db2,
                          float* dh, float* scratch, int64_t B, int64_t D, hipStream_t s, const float** dw1part_o, const float** db1part_o,
                          int* nsplit_o, const float* loss_rows, float* loss_mean) {
    const int KS = (int)(D / 128), nb = (int)((B + TB - 1) / TB);
    int NS, RB; bwd_split(B, &NS, &RB);
    float* drpart = scratch;
    float* dw2part = drpart + B * D;
    float* db2part = dw2part + (size_t)NS * D * 128;
    float* dw1part = db2part + (size_t)NS * D;
    float* db1part = dw1part + (size_t)NS * 128 * D;
    hipLaunchKernelGGL(bwd2_kernel, dim3((unsigned)(D / 32 * NS + nb * KS + (loss_mean ? 1 : 0))), dim3(256), 0, s, dz, r, w2, dw2part, db2part, drpart,
                       (int)B, (int)D, NS, RB, loss_rows, loss_mean);
    DBMM_CHECK_LAUNCH();
    const long long nw4 = D * 128 / 4, nb4 = D / 4;
    hipLaunchKernelGGL(bn_bwd_kernel, dim3((unsigned)(32 + (nw4 + nb4 + 255) / 256)), dim3(256), 0, s, (const float*)drpart, KS, h, mean, invstd, gamma,
                       beta, dgamma, dbeta, dh, (int)B, (const float*)dw2part, dw2, nw4, (const float*)db2part, db2, nb4, NS);
    DBMM_CHECK_LAUNCH();
    hipLaunchKernelGGL(bwd1_kernel, dim3((unsigned)(D / 32), NS), dim3(256), 0, s, (const float*)dh, x, dw1part, db1part, (int)B, (int)D, RB);
    DBMM_CHECK_LAUNCH();
    if (dw1part_o) {       // the caller (the one-call step) sums the batch splits of dW1 / db1 inside its SGD launch, in the same order
        *dw1part_o = dw1part; *db1part_o = db1part; *nsplit_o = NS;
        return DBMM_OK;
    }
    hipLaunchKernelGGL(grad_reduce_kernel, dim3((unsigned)((nw4 + 32 + 255) / 256)), dim3(256), 0, s, (const float*)dw1part, dw1, nw4,
                       (const float*)db1part, db1, 32LL, NS);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}
