// 3x3 / stride 1 / pad 1 convolution over 32 input channels -- the two wide stem convs of ModifiedResNet
// (clip/model.py:108-116: conv2 32 -> 32 and conv3 32 -> 64 on 112 x 112 maps, the latter followed by AvgPool2d(2)) --
// with eval-mode BatchNorm scale / bias, ReLU and the optional 2x2 average pool in the epilogue.
//
// Why a kernel of its own: with K = 9 * 32 = 288 a 128-pixel tile of the implicit-GEMM kernels has a nine-step K loop;
// at B = 1024 the two convs ran 1.3 + 2.2 ms for 0.24 + 0.47 TFLOP and 3.3 + 2.5 GB -- 15-20 % of either roof.  The
// time went into latencies that a nine-step loop cannot hide (three dependent operand round trips and an epilogue per
// tile, nothing prefetched across tiles).  Here
//   * the whole weight (N x 288 fp16, 19 / 37 KB) is loaded into LDS once per workgroup, which is persistent;
//   * a tile is 4 rows x 28 columns of output (112 of the 128 MFMA rows; 28 pooling windows = 2 x 14 pooled pixels),
//     its 6 x 30-pixel input patch is fetched ONCE and all nine taps are formed from it by shifted LDS reads
//     (1.6 input reads per output pixel instead of 9 through L1);
//   * the next tile's patch is in flight (registers) while the current tile computes and stores.
// Arithmetic = the fp16-pair path of igemm_f32.hip: x = (hi + lo) * 2^-s with s from the producer's device maximum,
// weights exact in one fp16 plane, two MFMA products, fp32 accumulation, exact power-of-two rescale.
//
// Output goes straight from the MFMA accumulator layout: one register = two 128-B row segments per wave instruction
// (N = 32: one), the full-rate store shape; a pooling window is the four registers (r & 3) of one lane.
// Bound: HBM (input read once + 0.6 halo from L2, output written once).
#include <stdlib.h>
#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOR = 0x80000000u;
constexpr long long EXT_LIM = 0x7FFFFFF0LL;
constexpr int TR = 4, TC = 28;                        // output tile: rows x columns (112 pixels)
constexpr int PR = TR + 2, PC = TC + 2, PPX = PR * PC; // input patch: 6 x 30 = 180 pixels
constexpr int CIN = 32, KTOT = 9 * CIN;               // 288
constexpr int WROW = KTOT + 8;                        // LDS pitch of a weight row in halves (592 B: conflict-free b128 reads)

struct PatchP {
    const float* x; const float* x_absmax;
    const u16* w; int w_exp; const float* sc; const float* bias;
    float* y; float* y_absmax;
    int B, H, W, tiles_w, tiles_h, n_tiles;
};

__device__ __forceinline__ int scale_exp(float amax) {
    const unsigned b = __float_as_uint(amax) & 0x7fffffffu;
    int s = b ? 13 - ((int)(b >> 23) - 127) : 0;
    return s < -60 ? -60 : (s > 60 ? 60 : s);
}
__device__ __forceinline__ float pow2f(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }

__device__ __forceinline__ void split2h_pair(float x0, float x1, float sc, unsigned& hi, unsigned& lo) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(x0), "v"(sc));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(x1), "v"(sc));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(x0), "v"(sc), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(x1), "v"(sc), "v"(hi));
#else
    (void)x0; (void)x1; (void)sc; hi = lo = 0;
#endif
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t desc(const void* base, long long total, long long shift) {
    long long ext = total - shift;
    ext = ext < 0 ? 0 : (ext > EXT_LIM ? EXT_LIM : ext);
    return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)base + shift), 0, (int)ext, 0x00020000);
}
// patch rows of 32 halves (64 B): four rows per 256-B bank sweep
__device__ __forceinline__ int pswz(int row) { return (row >> 2) & 3; }

template <int N, int POOL>
__global__ __launch_bounds__(256, N == 32 ? 3 : 2) void conv3x3_c32_kernel(const PatchP p) {
    constexpr int TN = N / 32, NLD = (PPX * 8 + 255) / 256;        // 6 patch loads (16 B) per thread
    // LDS pitch of a patch row, chosen with the MFMA-row -> pixel map so that the 16 lanes of every ds_read_b128 group
    // (lane sets {0-3,12-15,20-27}, {4-11,16-19,28-31}, MI355X_MICROARCH.md) read patch rows that are distinct modulo 16
    // = distinct bank quads under the XOR swizzle.  Un-pooled: a wave takes ONE tile row (28 pixels + 4 idle lanes), its
    // rows are consecutive, any pitch works.  Pooled: lanes 4o .. 4o+3 are the 2x2 window o, i.e. rows p, p+1, p+pitch,
    // p+pitch+1 -- pitch 40 (= 8 mod 16) keeps the two image rows of a group apart (pitch 30 measured 30 % conflict cycles).
    constexpr int PCL = POOL ? 40 : PC, PLN = PR * PCL * 32;       // halves per plane
    __shared__ __attribute__((aligned(16))) u16 lds[N * WROW + 2 * PLN];
    u16* Wl = lds;                                                 // [N][WROW]
    u16* Pl = lds + N * WROW;                                      // [2 planes][6 x PCL][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;

    // ---- the whole weight, once -------------------------------------------------------------------------------------
    for (int i = tid; i < N * (KTOT / 8); i += 256) {
        const int n = i / (KTOT / 8), c = i - n * (KTOT / 8);
        *(u32x4*)(Wl + n * WROW + c * 8) = *(const u32x4*)(p.w + (size_t)n * KTOT + c * 8);
    }
    const int s_a = scale_exp(*p.x_absmax);
    const float a_sc = pow2f(s_a), acc_scale = pow2f(-s_a - p.w_exp);
    float sv[TN], bv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { sv[j] = p.sc[j * 32 + fr] * acc_scale; bv[j] = p.bias[j * 32 + fr]; }

    // this lane's MFMA row (A operand): tile pixel (dy, c) -> patch row of tap (0, 0); rows >= 112 are padding
    const int ml = wave * 32 + fr;
    int a_dy, a_c;
    bool a_ok;
    if (POOL) { const int o = ml >> 2, q = ml & 3; a_dy = 2 * (o / 14) + (q >> 1); a_c = 2 * (o % 14) + (q & 1); a_ok = ml < TR * TC; }
    else { a_dy = wave; a_c = fr; a_ok = fr < TC; }
    const int a_row0 = a_ok ? a_dy * PCL + a_c : 0;

    const long long img_bytes = (long long)p.H * p.W * CIN * 4;
    f32x4 pre[NLD];
    auto tile_coords = [&](int tile, int& b, int& h0, int& w0) {
        const int tw = tile % p.tiles_w, r = tile / p.tiles_w;
        b = r / p.tiles_h; h0 = (r - b * p.tiles_h) * TR; w0 = tw * TC;
    };
    auto load_patch = [&](int tile) {
        int b, h0, w0;
        tile_coords(tile, b, h0, w0);
        const __amdgpu_buffer_rsrc_t rs = desc(p.x, (long long)p.B * img_bytes, (long long)b * img_bytes);
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int idx = tid + 256 * k, i = idx >> 3, quad = idx & 7;
            const int py = i / PC, px = i - py * PC, h = h0 - 1 + py, w = w0 - 1 + px;
            const bool ok = i < PPX && h >= 0 && h < p.H && w >= 0 && w < p.W;
            pre[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                rs, ok ? (unsigned)((h * p.W + w) * (CIN * 4) + quad * 16) : OOR, 0, 0));
        }
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int idx = tid + 256 * k, i = idx >> 3, quad = idx & 7;
            if (i < PPX) {
                unsigned hp[2], lp[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) split2h_pair(pre[k][2 * j], pre[k][2 * j + 1], a_sc, hp[j], lp[j]);
                const int py = i / PC, row = py * PCL + (i - py * PC);
                const int off = row * 32 + (((quad >> 1) ^ pswz(row)) << 3) + ((quad & 1) << 2);
                *(u32x2*)(Pl + off) = (u32x2){hp[0], hp[1]};
                *(u32x2*)(Pl + PLN + off) = (u32x2){lp[0], lp[1]};
            }
        }
    };

    float amax = 0.f;
    int tile = blockIdx.x;
    if (tile < p.n_tiles) load_patch(tile);
    for (; tile < p.n_tiles; tile += gridDim.x) {
        store_patch();
        __syncthreads();                                           // patch (and, the first time, the weights) visible
        if (tile + (int)gridDim.x < p.n_tiles) load_patch(tile + gridDim.x);   // in flight during the MFMAs and stores below

        f32x16 acc[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int row = a_row0 + (tap / 3) * PCL + (tap % 3);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int aoff = row * 32 + (((2 * ks + fh) ^ pswz(row)) << 3);
                const u32x4 ah = *(const u32x4*)(Pl + aoff), al = *(const u32x4*)(Pl + PLN + aoff);
                u32x4 wf[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) wf[j] = *(const u32x4*)(Wl + (j * 32 + fr) * WROW + tap * 32 + (2 * ks + fh) * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al), __builtin_bit_cast(f16x8, wf[j]), acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, wf[j]), acc[j], 0, 0, 0);
            }
        }
        // ---- epilogue: BatchNorm scale / bias, ReLU, (2x2 average), stores from the accumulator layout -------------------
        int b, h0, w0;
        tile_coords(tile, b, h0, w0);
        if constexpr (POOL) {
            const int Hp = p.H >> 1, Wp = p.W >> 1;
            const long long tot = (long long)p.B * Hp * Wp * N * 4;
            const long long base_px = ((long long)b * Hp + (h0 >> 1)) * Wp + (w0 >> 1);
            const __amdgpu_buffer_rsrc_t rs = desc(p.y, tot, base_px * N * 4);
#pragma unroll
            for (int t = 0; t < 4; ++t) {                          // window o = registers 4t .. 4t+3 of this lane
                const int o = wave * 8 + 2 * t + fh;
                const unsigned off = o < 28 ? (unsigned)(((o / 14) * Wp + (o % 14)) * (N * 4) + fr * 4) : OOR;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaxf(fmaf(acc[j][4 * t + q], sv[j], bv[j]), 0.f);
                    const float s = (((v[0] + v[1]) + v[2]) + v[3]) * 0.25f;      // (dy, dx) order, like avgpool_kernel
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, s), rs, off, (unsigned)(j * 128), 0);
                    if (o < 28) amax = fmaxf(amax, fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));   // bounds the pooled values too
                }
            }
        } else {
            const long long tot = (long long)p.B * p.H * p.W * N * 4;
            const long long base_px = ((long long)b * p.H + h0) * p.W + w0;
            const __amdgpu_buffer_rsrc_t rs = desc(p.y, tot, base_px * N * 4);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = (r & 3) + 8 * (r >> 2) + 4 * fh, dy = wave;      // accumulator row = the wave's tile row, column c
                const bool m_ok = c < TC;
                const unsigned off = m_ok ? (unsigned)((dy * p.W + c) * (N * 4) + fr * 4) : OOR;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float v = fmaxf(fmaf(acc[j][r], sv[j], bv[j]), 0.f);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, off, (unsigned)(j * 128), 0);
                    if (m_ok) amax = fmaxf(amax, v);
                }
            }
        }
        __syncthreads();                                           // every wave is done reading this patch
    }
    if (p.y_absmax) {
        amax = wave_max(amax);
        float* red = (float*)Pl;
        if (lane == 0) red[wave] = amax;
        __syncthreads();
        if (tid == 0) {
            const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            if (m > *(volatile const float*)p.y_absmax) atomicMax((unsigned*)p.y_absmax, __float_as_uint(m));
        }
    }
}

}  // namespace

// see include/dbmm.h
extern "C" int dbmm_conv3x3_c32_bn_relu_x2(const float* x, const float* x_absmax, const void* w_plane_f16, int w_exp,
                                           const float* scale, const float* bias, float* y, float* y_absmax, int64_t B, int64_t H,
                                           int64_t W, int64_t Cin, int64_t Cout, int pool, void* stream) {
    if (!x || !x_absmax || !w_plane_f16 || !scale || !bias || !y) return DBMM_E_ARG;
    if (B <= 0 || H <= 0 || W <= 0 || (pool != 0 && pool != 2)) return DBMM_E_SHAPE;
    if (Cin != CIN || (Cout != 32 && Cout != 64) || (H % TR) || (W % TC) || w_exp < -40 || w_exp > 40) return DBMM_E_UNSUPPORTED;
    if (H * W * CIN * 4 >= EXT_LIM || B * (H / TR) * (W / TC) > INT32_MAX) return DBMM_E_UNSUPPORTED;
    if (!dbmm_aligned16(x) || !dbmm_aligned16(w_plane_f16) || !dbmm_aligned16(y)) return DBMM_E_ALIGN;
    PatchP p{};
    p.x = x; p.x_absmax = x_absmax; p.w = (const u16*)w_plane_f16; p.w_exp = w_exp; p.sc = scale; p.bias = bias;
    p.y = y; p.y_absmax = y_absmax; p.B = (int)B; p.H = (int)H; p.W = (int)W;
    p.tiles_w = (int)(W / TC); p.tiles_h = (int)(H / TR); p.n_tiles = (int)(B * p.tiles_h * p.tiles_w);
    const int per_cu = Cout == 32 ? 3 : 2;                        // 42 / 61 KB of LDS per workgroup
    const int grid = p.n_tiles < 256 * per_cu ? p.n_tiles : 256 * per_cu;
    hipStream_t s = (hipStream_t)stream;
    if (Cout == 32) {
        if (pool) hipLaunchKernelGGL((conv3x3_c32_kernel<32, 1>), dim3(grid), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv3x3_c32_kernel<32, 0>), dim3(grid), dim3(256), 0, s, p);
    } else {
        if (pool) hipLaunchKernelGGL((conv3x3_c32_kernel<64, 1>), dim3(grid), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv3x3_c32_kernel<64, 0>), dim3(grid), dim3(256), 0, s, p);
    }
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}
