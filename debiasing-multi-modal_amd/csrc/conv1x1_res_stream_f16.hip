// fp16 mode: conv3 + BatchNorm + residual + ReLU of a bottleneck block (clip/model.py:50-54) for K = 256 into many channels (layer 3:
// 256 -> 1024; K = 128 for layer 2's last block, pooled variant only), the fp16 twin of conv1x1_res_stream.hip:
//
//   y = relu( (a @ W^T) * scale + bias + residual )        a f16 [M][K], W f16 [N][K], residual / y f16 [M][N]
//   (POOL: also AvgPool2d(2) of y, f16 [M / 4][N], for the next stage's downsample branch -- the stage's last block)
//
// One workgroup owns 128 pixel rows for a RANGE of 64-channel slabs: the a fragments are loaded once straight into registers (64 of them)
// and every slab is 32 MFMAs per wave between a residual load issued two slabs earlier and packed-fp16 stores straight from the accumulator
// layout (the W rows are staged interleaved -- block j, column c <-> channel 2 c + j -- so a lane holds two ADJACENT channels and 32 lanes
// make a 128-B row segment).  The weight slab [64][256] is prefetched one slab ahead into registers and lands in a ring of two LDS buffers:
// one barrier per slab.  (tile, slab) units are dealt as ONE range per workgroup (layer 3 at B = 1024: 1568 tiles x 16 slabs over 512
// workgroup slots = 49 units each).
//
// Arithmetic: one fp16 MFMA per product, fp32 accumulation over the 16 K steps in order, scale / bias / residual / ReLU on the fp32
// accumulator, one rounding to fp16 -- conv1x1_f16_kernel's, bit for bit; the pooled copy is avgpool2_f16_kernel's fp32 sum of the ROUNDED
// values in (dy, dx) order.  POOL walks 2x2-window-major tile rows (the four registers (r & 3) of an accumulator group are one window).
// Bound: HBM, 2 * (K + 2 N) bytes per pixel row (+ N / 2 pooled).
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOR = 0x80000000u;
constexpr long long EXT_LIM = 0x7FFFFFF0LL;

struct ResStreamHP {
    const u16* a; const u16* w; const float* sc; const float* b; const u16* res; u16* y; u16* yp;
    int M, N, n_tiles, Ho, Wo;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t desc(const void* base, long long total, long long shift) {
    long long ext = total - shift;
    ext = ext < 0 ? 0 : (ext > EXT_LIM ? EXT_LIM : ext);
    return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)base + shift), 0, (int)ext, 0x00020000);
}
__device__ __forceinline__ unsigned pack2(float a, float b) {
    const f16x2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(unsigned, v);
}
// pixel (standard order) of tile row m: identity, or 2x2-window-major (m = 4 * pooled pixel + dy * 2 + dx)
template <int POOL>
__device__ __forceinline__ int row_pixel(const ResStreamHP& p, int m) {
    if constexpr (!POOL) return m;
    const int mp = m >> 2, q = m & 3, wp2 = p.Wo >> 1, hwp = (p.Ho >> 1) * wp2;
    const int n = mp / hwp, rem = mp - n * hwp, hp = rem / wp2;
    return (n * p.Ho + 2 * hp + (q >> 1)) * p.Wo + 2 * (rem - hp * wp2) + (q & 1);
}

constexpr int BM = 128, BNS = 64;

template <int K, int POOL>
__global__ __launch_bounds__(256, 2) void conv1x1_res_stream_f16_kernel(const ResStreamHP p) {
    static_assert(K == 128 || K == 256, "reduction depth: layer 2 / layer 3");
    constexpr int KS = K / 16;
    __shared__ __attribute__((aligned(256))) unsigned char lds[2 * BNS * K * 2];      // ring of two weight slabs [64 rows (block j, column c)][K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int NT = p.N / BNS;
    const long long U = (long long)p.n_tiles * NT;
    long long u = U * blockIdx.x / gridDim.x;
    const long long u_end = U * (blockIdx.x + 1) / gridDim.x;
    const long long Mll = p.M;
    // weight slab: 16-B chunks dealt over the 256 threads (K = 256: 32 chunks per row, 8 rows per pass, 8 loads per thread)
    constexpr int CPR = K / 8, RPP = 256 / CPR, WLD = BNS * CPR / 256;
    const int wc = tid % CPR, wr = tid / CPR;
    u32x4 wreg[WLD];

    while (u < u_end) {
        const int tile = (int)(u / NT), nt0 = (int)(u - (long long)tile * NT);
        const int nt1 = (long long)(NT - nt0) < u_end - u ? NT : nt0 + (int)(u_end - u);
        u += nt1 - nt0;
        const int m0 = tile * BM;
        const int g0 = row_pixel<POOL>(p, m0);                     // descriptors rebased to the tile's first pixel
        const __amdgpu_buffer_rsrc_t rsA = desc(p.a, Mll * K * 2, (long long)g0 * K * 2);
        const __amdgpu_buffer_rsrc_t rsR = desc(p.res, Mll * p.N * 2, (long long)g0 * p.N * 2);
        const __amdgpu_buffer_rsrc_t rsY = desc(p.y, Mll * p.N * 2, (long long)g0 * p.N * 2);
        __amdgpu_buffer_rsrc_t rsP = rsY;
        if constexpr (POOL) rsP = desc(p.yp, (Mll >> 2) * p.N * 2, (long long)(m0 >> 2) * p.N * 2);
        // this lane's 16 accumulator rows: 4 groups (t = r >> 2) of 4 consecutive tile rows 8 t + 4 fh + (r & 3); valid iff m0 + 32 wave + row < M
        const int row_lim = p.M - m0 - wave * 32 - 4 * fh;          // row u = (r & 3) + 8 (r >> 2) is valid iff u < row_lim
        unsigned gx[POOL ? 4 : 1];
#pragma unroll
        for (int t = 0; t < (POOL ? 4 : 1); ++t)
            gx[t] = (unsigned)((row_pixel<POOL>(p, m0 + wave * 32 + 8 * t + 4 * fh) - g0) * p.N + 2 * fr) * 2u;      // + slab * 128 B
        auto vrow = [&](int r) { return gx[POOL ? (r >> 2) : 0]; };
        auto srow = [&](int r) { return POOL ? (r >> 1 & 1) * p.Wo + (r & 1) : (r & 3) + 8 * (r >> 2); };          // pixel step of register r's row
        auto load_w = [&](int nt) {
#pragma unroll
            for (int j = 0; j < WLD; ++j) {
                const int lr = wr + RPP * j, ch = 2 * (lr & 31) + (lr >> 5);     // LDS row (block lr >> 5, column lr & 31) <-> channel
                wreg[j] = *(const u32x4*)(p.w + (size_t)(nt * BNS + ch) * K + wc * 8);
            }
        };
        auto store_w = [&](int slot) {
#pragma unroll
            for (int j = 0; j < WLD; ++j) {
                const int lr = wr + RPP * j;
                *(u32x4*)(lds + slot * (BNS * K * 2) + lr * (K * 2) + ((wc ^ (lr & 15)) << 4)) = wreg[j];
            }
        };
        unsigned rv[2][16];
        auto load_res = [&](int nt, auto slot_c) {
            constexpr int S = decltype(slot_c)::value;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ru = (r & 3) + 8 * (r >> 2);
                rv[S][r] = __builtin_amdgcn_raw_buffer_load_b32(rsR, ru < row_lim ? vrow(r) + nt * (BNS * 2) : OOR, (unsigned)(srow(r) * p.N * 2), 0);
            }
        };
        // ---- segment prologue ----
        load_w(nt0);
        load_res(nt0, std::integral_constant<int, 0>());
        u32x4 ay[KS];
        {
            const int m = m0 + wave * 32 + fr;
            const unsigned va = m < p.M ? (unsigned)((row_pixel<POOL>(p, m) - g0) * K + 8 * fh) * 2u : OOR;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) ay[ks] = __builtin_amdgcn_raw_buffer_load_b128(rsA, va, (unsigned)(ks * 32), 0);
        }
        if (nt0 + 1 < nt1) load_res(nt0 + 1, std::integral_constant<int, 1>());
        store_w(0);                                                // (slot 0: its last readers left through the previous segment's last barrier)
        __syncthreads();
        // ---- slabs ----
        auto slab = [&](int nt, auto slot_c) {
            constexpr int S = decltype(slot_c)::value;
            if (nt + 1 < nt1) load_w(nt + 1);                      // lands during this slab's MFMAs
            f32x16 acc[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
            const unsigned char* Wb = lds + S * (BNS * K * 2);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int lr = j * 32 + fr;
                    const u32x4 wf = *(const u32x4*)(Wb + lr * (K * 2) + (((2 * ks + fh) ^ (lr & 15)) << 4));
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ay[ks]), __builtin_bit_cast(f16x8, wf), acc[j], 0, 0, 0);
                }
            const int n = nt * BNS + 2 * fr;
            const float s0 = p.sc ? p.sc[n] : 1.f, s1 = p.sc ? p.sc[n + 1] : 1.f, c0 = p.b ? p.b[n] : 0.f, c1 = p.b ? p.b[n + 1] : 0.f;
            unsigned xv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const f16x2 rh = __builtin_bit_cast(f16x2, rv[S][r]);
                const float v0 = fmaxf(fmaf(acc[0][r], s0, c0) + (float)rh[0], 0.f), v1 = fmaxf(fmaf(acc[1][r], s1, c1) + (float)rh[1], 0.f);
                xv[r] = pack2(v0, v1);
                const int ru = (r & 3) + 8 * (r >> 2);
                __builtin_amdgcn_raw_buffer_store_b32(xv[r], rsY, ru < row_lim ? vrow(r) + nt * (BNS * 2) : OOR, (unsigned)(srow(r) * p.N * 2), 0);
            }
            if constexpr (POOL) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {                      // window = registers 4 t .. 4 t + 3 of this lane
                    float t0 = 0.f, t1 = 0.f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f16x2 h = __builtin_bit_cast(f16x2, xv[4 * t + q]);
                        t0 += (float)h[0]; t1 += (float)h[1];
                    }
                    const int mp = wave * 8 + 2 * t + fh;          // pooled row within the tile
                    __builtin_amdgcn_raw_buffer_store_b32(pack2(t0 * 0.25f, t1 * 0.25f), rsP,
                                                          8 * t < row_lim ? (unsigned)(mp * p.N + 2 * fr) * 2u + nt * (BNS * 2) : OOR, 0, 0);
                }
            }
            if (nt + 2 < nt1) load_res(nt + 2, slot_c);            // two slabs ahead, into the slot just consumed
            if (nt + 1 < nt1) store_w(S ^ 1);                      // (that slot's readers left through the previous slab's barrier)
            __syncthreads();
        };
        for (int nt = nt0; nt < nt1; nt += 2) {
            slab(nt, std::integral_constant<int, 0>());
            if (nt + 1 < nt1) slab(nt + 1, std::integral_constant<int, 1>());
        }
    }
}

}  // namespace

// see include/dbmm.h (dbmm_conv1x1_res_pool_f16) and common.h
int dbmm_conv1x1_res_stream_f16(const void* x, const void* w, const float* scale, const float* bias, const void* residual, void* y, void* y_pooled,
                                int64_t M, int64_t Ho, int64_t Wo, int64_t Cin, int64_t Cout, void* stream) {
    if (!x || !w || !residual || !y) return DBMM_E_ARG;
    if (M <= 0 || Cout <= 0 || M > (INT32_MAX >> 1)) return DBMM_E_SHAPE;
    if ((Cin != 256 && !(Cin == 128 && y_pooled)) || (Cout % BNS)) return DBMM_E_UNSUPPORTED;     // (K = 128: only the pooled variant pays)
    if (y_pooled && (Ho <= 0 || Wo <= 0 || (Ho & 1) || (Wo & 1) || (M & 3) || M % (Ho * Wo))) return DBMM_E_UNSUPPORTED;
    if ((132LL + (y_pooled ? 2 * Wo : 0)) * Cout * 2 >= EXT_LIM) return DBMM_E_UNSUPPORTED;       // a tile's pixel span under its rebased descriptors
    if (!dbmm_aligned16(x) || !dbmm_aligned16(w) || !dbmm_aligned16(residual) || !dbmm_aligned16(y) || (y_pooled && !dbmm_aligned16(y_pooled)))
        return DBMM_E_ALIGN;
    ResStreamHP p{};
    p.a = (const u16*)x; p.w = (const u16*)w; p.sc = scale; p.b = bias; p.res = (const u16*)residual; p.y = (u16*)y; p.yp = (u16*)y_pooled;
    p.M = (int)M; p.N = (int)Cout; p.n_tiles = (int)((M + BM - 1) / BM); p.Ho = (int)Ho; p.Wo = (int)Wo;
    const long long units = (long long)p.n_tiles * (Cout / BNS);
    const int grid = (int)(units < 512 ? units : 512);              // two workgroups per CU
    hipStream_t s = (hipStream_t)stream;
    if (Cin == 128) hipLaunchKernelGGL((conv1x1_res_stream_f16_kernel<128, 1>), dim3(grid), dim3(256), 0, s, p);
    else if (y_pooled) hipLaunchKernelGGL((conv1x1_res_stream_f16_kernel<256, 1>), dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((conv1x1_res_stream_f16_kernel<256, 0>), dim3(grid), dim3(256), 0, s, p);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

// see include/dbmm.h
extern "C" int dbmm_conv1x1_res_pool_f16(const void* x, const void* w, const float* scale, const float* bias, const void* residual, void* y,
                                         void* y_pooled, int64_t B, int64_t Ho, int64_t Wo, int64_t Cin, int64_t Cout, void* stream) {
    if (!y_pooled) return DBMM_E_ARG;
    if (B <= 0 || Ho <= 0 || Wo <= 0) return DBMM_E_SHAPE;
    if (!dbmm_opt(OPT_CONV1X1_RES_STREAM)) return DBMM_E_UNSUPPORTED;
    return dbmm_conv1x1_res_stream_f16(x, w, scale, bias, residual, y, y_pooled, B * Ho * Wo, Ho, Wo, Cin, Cout, stream);
}
