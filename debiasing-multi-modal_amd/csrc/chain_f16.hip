// fp16 mode: conv3 + residual of one bottleneck block CHAINED with conv1 of the next block in one launch
// (/root/reference/clip/model.py:42-55, two consecutive Bottleneck.forward bodies on the reference's GPU path):
//
//   x'  = relu( (y2 @ W3^T) * sc3 + b3 + x )         f16 [M][N]   written to HBM (it is the next residual)
//   y1' = relu( (x' @ W1^T) * sc1 + b1 )             f16 [M][P]   written to HBM
//
// Why: in the fp16 mode the 1x1 convs of layers 1-2 are HBM-bound and conv1 of the NEXT block re-reads the 256 / 512-channel tensor
// conv3 has just written: 1.6 of the 5.7 GB the two launches move per layer-1 block boundary at B = 1024.  As in bottleneck_chain_kernel
// (the parity mode's chain) one workgroup owns 128 pixel rows for ALL N output channels of conv3, walks them in 64-channel slabs, and
// every finished slab is at once the next 64-deep K chunk of conv1': the wide tensor is written once and not read back.
//
// Arithmetic: one fp16 MFMA per product, fp32 accumulation, BatchNorm scale / bias, residual and ReLU on the fp32 accumulator -- the
// arithmetic of conv1x1_f16_kernel.  The slab x' is ROUNDED TO fp16 before it feeds conv1', exactly what the two launches do (conv1'
// reads the stored fp16 x'), so the chain equals them bit for bit.
//
// Data movement: the y2 fragments (K <= 128) are loaded once per tile straight into registers; weights (W3 slab [64][K], W1 chunk
// [P][64]) are prefetched one slab ahead into registers and staged in LDS; residual loads and x' / y1' stores go straight from / to the
// accumulator layout as packed dwords (the W rows are staged interleaved: block j, column c <-> channel 2 c + j, so a lane holds two
// ADJACENT channels and 32 lanes make a 128-B row segment); the next slab's residual is in flight during this slab's MFMAs.  Each wave
// owns 32 rows through both GEMMs, so the fp16 slab goes through a wave-private LDS slab and needs no barrier.
// Bound: HBM.  Algorithmic bytes per pixel row: 2 * (K + 2 N + P).
#include <stdlib.h>
#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOR = 0x80000000u;
constexpr long long EXT_LIM = 0x7FFFFFF0LL;

struct ChainHP {
    const u16* a; const u16* w3; const float* sc3; const float* b3; const u16* res; u16* x;
    const u16* w1; const float* sc1; const float* b1; u16* y1;
    const u16* a2; const u16* w2; const float* ratio;              // DUAL: the block's downsample branch (64 channels deep) instead of a residual
    u16* xp; int Ho, Wo;                                            // POOL: AvgPool2d(2) of x' [M / 4][N]; M = B * Ho * Wo pixels
    int M, N;
};

// pixel (standard order) of tile row m: identity, or 2x2-window-major (m = 4 * pooled pixel + dy * 2 + dx)
template <int POOL>
__device__ __forceinline__ int row_pixel(const ChainHP& p, int m) {
    if constexpr (!POOL) return m;
    const int mp = m >> 2, q = m & 3, wp2 = p.Wo >> 1, hwp = (p.Ho >> 1) * wp2;
    const int n = mp / hwp, rem = mp - n * hwp, hp = rem / wp2;
    return (n * p.Ho + 2 * hp + (q >> 1)) * p.Wo + 2 * (rem - hp * wp2) + (q & 1);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t desc(const void* base, long long total, long long shift) {
    long long ext = total - shift;
    ext = ext < 0 ? 0 : (ext > EXT_LIM ? EXT_LIM : ext);
    return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)base + shift), 0, (int)ext, 0x00020000);
}
__device__ __forceinline__ unsigned pack2(float a, float b) {
    const f16x2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(unsigned, v);
}
// LDS rows of RB bytes: XOR of the 16-B chunk index with row bits keeps the lanes of a ds_read_b128 group on distinct bank quads
template <int RB>
__device__ __forceinline__ int swz(int row) { return RB == 128 ? ((row >> 1) & 7) : (row & 15); }

constexpr int BM = 128, BNS = 64;

// (two workgroups per CU; three -- layer 1's K = P = 64 variant squeezed into 168 registers, 10 of them spilled -- measured 934 against 921 us)
// DUAL (the first block of layer 1): x' = relu((y2 @ W3^T + ((xp @ Wd^T) * ratio)) * sc3 + b) -- the downsample branch's 64-deep GEMM
// runs first, its sums are multiplied per output channel by ratio = sc_d / sc3 and conv3 continues on the same accumulators: the
// order and arithmetic of conv1x1_f16_kernel<.., TWO = 1> (dbmm_conv1x1_dual_stream_f16), so again bit-equal to the two launches.
// POOL (a stage seam): the tile's rows walk the pixels in 2x2-window-major order, so the four registers (r & 3) of an accumulator group are
// one pooling window and AvgPool2d(2) of x' -- the next stage's downsample input -- is a sum inside the lane (fp32 sum of the ROUNDED fp16
// values in (dy, dx) order, times 0.25: avgpool2_f16_kernel's arithmetic, bit-equal).  POOL = 2: only the pooled copy is written; conv1' has
// consumed the un-pooled x' here and nobody else reads it.  M % 4 == 0, Ho and Wo even.
template <int K, int P, int DUAL = 0, int POOL = 0>
__global__ __launch_bounds__(256, 2) void chain_f16_kernel(const ChainHP p) {
    static_assert(!(DUAL && POOL), "a stage's first block is not its last");
    static_assert(K == 64 || K == 128, "conv3 reduction depth: 64 (layer 1) or 128 (layer 2)");
    static_assert(P == 64 || P == 128, "conv1' width");
    constexpr int KS = K / 16, TN1 = P / 32;
    constexpr int W3_BYTES = BNS * K * 2, W1_BYTES = P * BNS * 2, XS_BYTES = 4 * 32 * 128, W2_BYTES = DUAL ? BNS * 64 * 2 : 0;
    __shared__ __attribute__((aligned(256))) unsigned char lds[W3_BYTES + W1_BYTES + XS_BYTES + W2_BYTES];
    unsigned char* W3b = lds;                                       // [64 rows (block j, column c)][K]
    unsigned char* W1b = lds + W3_BYTES;                            // [P rows (block, column)][64 k]
    unsigned char* W2b = lds + W3_BYTES + W1_BYTES + XS_BYTES;      // DUAL: [64 rows (block j, column c)][64]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned char* XS = lds + W3_BYTES + W1_BYTES + wave * (32 * 128);   // this wave's x' slab: [32 rows][64 k] fp16
    const int fr = lane & 31, fh = lane >> 5;
    const int m0 = xcd_remap(blockIdx.x, gridDim.x) * BM;
    const int NT = p.N / BNS;
    const long long Mll = p.M;
    const int g0 = row_pixel<POOL>(p, m0);                                              // descriptors rebased to the tile's first pixel
    const __amdgpu_buffer_rsrc_t rsA = desc(p.a, Mll * K * 2, (long long)g0 * K * 2);
    const __amdgpu_buffer_rsrc_t rsR = DUAL ? desc(p.a2, Mll * 64 * 2, (long long)g0 * 64 * 2) : desc(p.res, Mll * p.N * 2, (long long)g0 * p.N * 2);
    const __amdgpu_buffer_rsrc_t rsX = desc(p.x, POOL == 2 ? 0 : Mll * p.N * 2, (long long)g0 * p.N * 2);
    const __amdgpu_buffer_rsrc_t rsY = desc(p.y1, Mll * P * 2, (long long)g0 * P * 2);
    __amdgpu_buffer_rsrc_t rsXP = rsX;
    if constexpr (POOL) rsXP = desc(p.xp, (Mll >> 2) * p.N * 2, (long long)(m0 >> 2) * p.N * 2);

    // accumulator rows of this lane: u = (r & 3) + 8 (r >> 2) + 4 fh of the wave's 32; valid iff m0 + 32 wave + u < M.  The 16 rows are
    // 4 groups (t = r >> 2) of 4 consecutive tile rows: the group's first pixel goes into the vector offset, the step inside the group
    // (wave-uniform: q, or (dy Wo + dx) of a 2x2 window) into the scalar offset.
    const int row_lim = p.M - m0 - wave * 32 - 4 * fh;                                  // row u is valid iff (r & 3) + 8 (r >> 2) < row_lim
    unsigned gx[POOL ? 4 : 1], gy[POOL ? 4 : 1];                                        // (standard order: one offset, all steps scalar)
#pragma unroll
    for (int t = 0; t < (POOL ? 4 : 1); ++t) {
        const int dp = row_pixel<POOL>(p, m0 + wave * 32 + 8 * t + 4 * fh) - g0;
        gx[t] = (unsigned)(dp * p.N + 2 * fr) * 2u;                                     // + slab * 128 B
        gy[t] = (unsigned)(dp * P + 2 * fr) * 2u;
    }
    auto vrow = [&](const unsigned (&g)[POOL ? 4 : 1], int r) { return g[POOL ? (r >> 2) : 0]; };
    auto srow = [&](int r) { return POOL ? (r >> 1 & 1) * p.Wo + (r & 1) : (r & 3) + 8 * (r >> 2); };     // pixel step of register r's row

    // ---- weight slabs through registers: W3 slab [64 n][K], W1 chunk [P][64 k]; 16-B chunks dealt over the 256 threads ----
    constexpr int CPR3 = K / 8, RPP3 = 256 / CPR3, W3LD = BNS * CPR3 / 256;             // chunks per row, rows per pass, loads per thread
    constexpr int CPR1 = BNS / 8, RPP1 = 256 / CPR1, W1LD = P * CPR1 / 256;
    const int wc3 = tid % CPR3, wr3 = tid / CPR3, wc1 = tid % CPR1, wr1 = tid / CPR1;
    u32x4 w3r[W3LD], w1r[W1LD], w2r[DUAL ? 2 : 1];
    auto load_w = [&](int nt) {
        if (DUAL) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int lr = wr1 + RPP1 * j, ch = 2 * (lr & 31) + (lr >> 5);
                w2r[j] = *(const u32x4*)(p.w2 + (size_t)(nt * BNS + ch) * 64 + wc1 * 8);
            }
        }
#pragma unroll
        for (int j = 0; j < W3LD; ++j) {
            const int lr = wr3 + RPP3 * j, ch = 2 * (lr & 31) + (lr >> 5);              // LDS row (block lr >> 5, column lr & 31) <-> channel
            w3r[j] = *(const u32x4*)(p.w3 + (size_t)(nt * BNS + ch) * K + wc3 * 8);
        }
#pragma unroll
        for (int j = 0; j < W1LD; ++j) {
            const int lr = wr1 + RPP1 * j, ch = (lr >> 6) * 64 + 2 * (lr & 31) + ((lr >> 5) & 1);   // output channel of LDS row lr
            w1r[j] = *(const u32x4*)(p.w1 + (size_t)ch * p.N + nt * BNS + wc1 * 8);
        }
    };
    auto store_w = [&]() {
        if (DUAL) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int lr = wr1 + RPP1 * j;
                *(u32x4*)(W2b + lr * 128 + ((wc1 ^ swz<128>(lr)) << 4)) = w2r[j];
            }
        }
#pragma unroll
        for (int j = 0; j < W3LD; ++j) {
            const int lr = wr3 + RPP3 * j;
            *(u32x4*)(W3b + lr * (K * 2) + ((wc3 ^ swz<K * 2>(lr)) << 4)) = w3r[j];
        }
#pragma unroll
        for (int j = 0; j < W1LD; ++j) {
            const int lr = wr1 + RPP1 * j;
            *(u32x4*)(W1b + lr * 128 + ((wc1 ^ swz<128>(lr)) << 4)) = w1r[j];
        }
    };
    unsigned rv[16];
    auto load_res = [&](int nt) {
        if (DUAL) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int u = (r & 3) + 8 * (r >> 2);
            rv[r] = __builtin_amdgcn_raw_buffer_load_b32(rsR, u < row_lim ? vrow(gx, r) + nt * (BNS * 2) : OOR, (unsigned)(srow(r) * p.N * 2), 0);
        }
    };

    // ---- prologue: first weight slabs and residual slab in flight; the y2 fragments of the wave's 32 rows into registers ----
    load_w(0);
    load_res(0);
    u32x4 ay[KS], ay2[DUAL ? 4 : 1];
    {
        const int m = m0 + wave * 32 + fr;
        const unsigned va = m < p.M ? (unsigned)((row_pixel<POOL>(p, m) - g0) * K + 8 * fh) * 2u : OOR;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) ay[ks] = __builtin_amdgcn_raw_buffer_load_b128(rsA, va, (unsigned)(ks * 32), 0);
        if (DUAL) {
            const unsigned va2 = m < p.M ? (unsigned)((row_pixel<POOL>(p, m) - g0) * 64 + 8 * fh) * 2u : OOR;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) ay2[ks] = __builtin_amdgcn_raw_buffer_load_b128(rsR, va2, (unsigned)(ks * 32), 0);
        }
    }
    f32x16 acc1[TN1];
#pragma unroll
    for (int j = 0; j < TN1; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[j][r] = 0.f;
    store_w();
    __syncthreads();

    for (int nt = 0; nt < NT; ++nt) {
        if (nt + 1 < NT) load_w(nt + 1);
        // ---- conv3: this slab's 64 channels (two blocks: even / odd channels) of the wave's 32 rows ----
        f32x16 acc3[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc3[j][r] = 0.f;
        const int n = nt * BNS + 2 * fr;
        if (DUAL) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int lr = j * 32 + fr;
                    const u32x4 wf = *(const u32x4*)(W2b + lr * 128 + (((2 * ks + fh) ^ swz<128>(lr)) << 4));
                    acc3[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ay2[ks]), __builtin_bit_cast(f16x8, wf), acc3[j], 0, 0, 0);
                }
            const float r0 = p.ratio[n], r1 = p.ratio[n + 1];
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc3[0][r] *= r0; acc3[1][r] *= r1; }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int lr = j * 32 + fr;
                const u32x4 wf = *(const u32x4*)(W3b + lr * (K * 2) + (((2 * ks + fh) ^ swz<K * 2>(lr)) << 4));
                acc3[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ay[ks]), __builtin_bit_cast(f16x8, wf), acc3[j], 0, 0, 0);
            }
        // ---- x' = relu(acc * sc3 + b3 + residual): packed fp16 to HBM and to the wave's LDS slab ----
        const float s0 = p.sc3 ? p.sc3[n] : 1.f, s1 = p.sc3 ? p.sc3[n + 1] : 1.f, c0 = p.b3 ? p.b3[n] : 0.f, c1 = p.b3 ? p.b3[n + 1] : 0.f;
        unsigned xv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v0 = fmaf(acc3[0][r], s0, c0), v1 = fmaf(acc3[1][r], s1, c1);
            if (!DUAL) {
                const f16x2 rh = __builtin_bit_cast(f16x2, rv[r]);
                v0 += (float)rh[0]; v1 += (float)rh[1];
            }
            xv[r] = pack2(fmaxf(v0, 0.f), fmaxf(v1, 0.f));
        }
        if (nt + 1 < NT) load_res(nt + 1);                          // (rv is free: the next slab's residual flies during the stores and conv1')
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int u = (r & 3) + 8 * (r >> 2);
            if constexpr (POOL != 2)
                __builtin_amdgcn_raw_buffer_store_b32(xv[r], rsX, u < row_lim ? vrow(gx, r) + nt * (BNS * 2) : OOR, (unsigned)(srow(r) * p.N * 2), 0);
            const int row = u + 4 * fh;
            *(unsigned*)(XS + row * 128 + (((fr >> 2) ^ swz<128>(row)) << 4) + (fr & 3) * 4) = xv[r];
        }
        if constexpr (POOL) {
            // rows 4 i .. 4 i + 3 of the wave are one 2x2 window = registers 4 t .. 4 t + 3 of this lane (i = 2 t + fh)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float s0 = 0.f, s1 = 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f16x2 h = __builtin_bit_cast(f16x2, xv[4 * t + q]);
                    s0 += (float)h[0]; s1 += (float)h[1];
                }
                const int mp = wave * 8 + 2 * t + fh;                               // pooled row within the tile
                __builtin_amdgcn_raw_buffer_store_b32(pack2(s0 * 0.25f, s1 * 0.25f), rsXP,
                                                      8 * t < row_lim ? (unsigned)(mp * p.N + 2 * fr) * 2u + nt * (BNS * 2) : OOR, 0, 0);
            }
        }
        // ---- conv1': the slab is its K chunk [64 nt, 64 nt + 64) ----
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const u32x4 af = *(const u32x4*)(XS + fr * 128 + (((2 * ks + fh) ^ swz<128>(fr)) << 4));
#pragma unroll
            for (int j = 0; j < TN1; ++j) {
                const int lr = j * 32 + fr;
                const u32x4 wf = *(const u32x4*)(W1b + lr * 128 + (((2 * ks + fh) ^ swz<128>(lr)) << 4));
                acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af), __builtin_bit_cast(f16x8, wf), acc1[j], 0, 0, 0);
            }
        }
        __syncthreads();                                            // every wave is done with this slab's weights
        if (nt + 1 < NT) store_w();
        __syncthreads();
    }
    // ---- y1' = relu(acc1 * sc1 + b1): blocks (2 h, 2 h + 1) hold the even / odd channels of the h-th 64 ----
#pragma unroll
    for (int h = 0; h < TN1 / 2; ++h) {
        const int n = h * 64 + 2 * fr;
        const float s0 = p.sc1 ? p.sc1[n] : 1.f, s1 = p.sc1 ? p.sc1[n + 1] : 1.f, c0 = p.b1 ? p.b1[n] : 0.f, c1 = p.b1 ? p.b1[n + 1] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int u = (r & 3) + 8 * (r >> 2);
            const float v0 = fmaxf(fmaf(acc1[2 * h][r], s0, c0), 0.f), v1 = fmaxf(fmaf(acc1[2 * h + 1][r], s1, c1), 0.f);
            __builtin_amdgcn_raw_buffer_store_b32(pack2(v0, v1), rsY, u < row_lim ? vrow(gy, r) + h * 128 : OOR, (unsigned)(srow(r) * P * 2), 0);
        }
    }
}

}  // namespace

static int chain_launch(const void* y2, const void* w3, const float* scale3, const float* bias3, const void* residual, const void* downsample_in,
                        const void* wd, const float* ratio, void* x_out, void* x_pooled, int64_t Ho, int64_t Wo, const void* w1, const float* scale1,
                        const float* bias1, void* y1_out, int64_t M, int64_t K, int64_t K2, int64_t N, int64_t P, void* stream) {
    const void* second = downsample_in ? downsample_in : residual;
    if (!y2 || !w3 || !second || (!x_out && !x_pooled) || !w1 || !y1_out || (downsample_in && (!wd || !ratio))) return DBMM_E_ARG;
    if (M <= 0 || N <= 0 || M > INT32_MAX - 1024) return DBMM_E_SHAPE;
    if (!(K == 64 || K == 128) || !(P == 64 || P == 128) || (N % 64) || N < 64) return DBMM_E_UNSUPPORTED;
    if (downsample_in && (K != 64 || P != 64 || K2 != 64 || x_pooled)) return DBMM_E_UNSUPPORTED;
    if (x_pooled && ((Ho & 1) || (Wo & 1) || (M & 3) || M > (INT32_MAX >> 1))) return DBMM_E_UNSUPPORTED;
    if (!dbmm_aligned16(y2) || !dbmm_aligned16(w3) || !dbmm_aligned16(second) || (x_out && !dbmm_aligned16(x_out)) || !dbmm_aligned16(w1) ||
        !dbmm_aligned16(y1_out) || (wd && !dbmm_aligned16(wd)) || (x_pooled && !dbmm_aligned16(x_pooled)))
        return DBMM_E_ALIGN;
    if ((132LL + (x_pooled ? 2 * Wo : 0)) * N * 2 >= EXT_LIM) return DBMM_E_UNSUPPORTED;      // a tile's pixel span under its rebased descriptors
    ChainHP p{};
    p.a = (const u16*)y2; p.w3 = (const u16*)w3; p.sc3 = scale3; p.b3 = bias3; p.res = (const u16*)residual; p.x = (u16*)x_out;
    p.w1 = (const u16*)w1; p.sc1 = scale1; p.b1 = bias1; p.y1 = (u16*)y1_out; p.M = (int)M; p.N = (int)N;
    p.a2 = (const u16*)downsample_in; p.w2 = (const u16*)wd; p.ratio = ratio;
    p.xp = (u16*)x_pooled; p.Ho = (int)Ho; p.Wo = (int)Wo;
    const dim3 grid((unsigned)((M + BM - 1) / BM));
    hipStream_t s = (hipStream_t)stream;
#define CHAIN_LAUNCH(KK, PP, DD, PL) hipLaunchKernelGGL((chain_f16_kernel<KK, PP, DD, PL>), grid, dim3(256), 0, s, p)
#define CHAIN_SHAPES(PL)                                   \
    do {                                                   \
        if (K == 64 && P == 64) CHAIN_LAUNCH(64, 64, 0, PL);        \
        else if (K == 64 && P == 128) CHAIN_LAUNCH(64, 128, 0, PL); \
        else if (K == 128 && P == 64) CHAIN_LAUNCH(128, 64, 0, PL); \
        else CHAIN_LAUNCH(128, 128, 0, PL);                         \
    } while (0)
    if (downsample_in) CHAIN_LAUNCH(64, 64, 1, 0);
    else if (!x_pooled) CHAIN_SHAPES(0);
    else if (x_out) CHAIN_SHAPES(1);
    else CHAIN_SHAPES(2);
#undef CHAIN_SHAPES
#undef CHAIN_LAUNCH
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

// see include/dbmm.h
extern "C" int dbmm_bottleneck_chain_f16(const void* y2, const void* w3, const float* scale3, const float* bias3, const void* residual, void* x_out,
                                         const void* w1, const float* scale1, const float* bias1, void* y1_out, int64_t M, int64_t K, int64_t N,
                                         int64_t P, void* stream) {
    if (!x_out) return DBMM_E_ARG;
    return chain_launch(y2, w3, scale3, bias3, residual, nullptr, nullptr, nullptr, x_out, nullptr, 0, 0, w1, scale1, bias1, y1_out, M, K, 0, N, P,
                        stream);
}

// see include/dbmm.h
extern "C" int dbmm_bottleneck_chain_pool_f16(const void* y2, const void* w3, const float* scale3, const float* bias3, const void* residual,
                                              void* x_out, void* x_pooled, const void* w1, const float* scale1, const float* bias1, void* y1_out,
                                              int64_t B, int64_t Ho, int64_t Wo, int64_t K, int64_t N, int64_t P, void* stream) {
    if (!x_pooled) return DBMM_E_ARG;
    if (B <= 0 || Ho <= 0 || Wo <= 0) return DBMM_E_SHAPE;
    return chain_launch(y2, w3, scale3, bias3, residual, nullptr, nullptr, nullptr, x_out, x_pooled, Ho, Wo, w1, scale1, bias1, y1_out, B * Ho * Wo,
                        K, 0, N, P, stream);
}

// see include/dbmm.h
extern "C" int dbmm_bottleneck_chain_dual_f16(const void* y2, const void* w3, const float* scale3, const void* xp, const void* wd, const float* ratio,
                                              const float* bias, void* x_out, const void* w1, const float* scale1, const float* bias1, void* y1_out,
                                              int64_t M, int64_t K, int64_t K2, int64_t N, int64_t P, void* stream) {
    if (!xp || !x_out) return DBMM_E_ARG;
    return chain_launch(y2, w3, scale3, bias, nullptr, xp, wd, ratio, x_out, nullptr, 0, 0, w1, scale1, bias1, y1_out, M, K, K2, N, P, stream);
}
