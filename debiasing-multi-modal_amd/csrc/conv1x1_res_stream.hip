// conv3 + BatchNorm + residual + ReLU of a bottleneck block (clip/model.py:50-54) for a SHORT reduction into MANY channels
// (layer 3: K = 256 -> N = 1024), parity mode:
//
//   y = relu( (a @ W^T) * scale + bias + residual )        a [M][K], W [N][K], residual / y [M][N], all fp32
//
// Why a kernel of its own.  On 128 x 128 tiles this family is the largest share of the headline step (9 launches, 3.9 of 28 ms) and
// HBM-bound at 4.5 - 4.7 TB/s: a tile's eight-step K loop, then its residual round trip, then its stores -- three workgroups per CU
// overlap them only partly, and the A panel is fetched once per 128 output columns.  Here, as in the first half of
// bottleneck_chain_kernel, one workgroup owns 128 pixel rows for a RANGE of 32-channel slabs: the a tile is split once into (hi, lo)
// fp16 A fragments that stay in registers (K = 256: 128 registers), and every slab is 32 MFMAs per wave between a residual load issued
// two slabs earlier and stores straight from the accumulator layout (one register = two 128-B row segments per wave instruction).
// The weight slab [32][K] (exact fp16 plane) is prefetched one slab ahead into registers and lands in a ring of two LDS buffers: one
// barrier per slab.
//
// Balance.  M / 128 tiles x N / 32 slabs are dealt as ONE range of (tile, slab) units per workgroup (layer 3 at B = 1024: 1568 tiles x
// 32 slabs over 512 workgroup slots = 98 units = 3.06 tiles each; whole tiles would leave a fourth round 6 % full); a workgroup that
// enters a tile in the middle pays that tile's prologue (the a tile: 12 % of the tile's bytes) again.
//
// Arithmetic = the fp16-pair path (fp32 value = fp16 hi + lo under an exact power-of-two scale from the producer's device maximum,
// weights exact in one fp16 plane, (lo, w) then (hi, w) per K step, fp32 accumulate).  Bound: HBM, 4 * (K + 2 N) bytes per pixel row.
//
// MEASURED (tools/bench_res_stream.py, same box, layer 3 at B = 1024: 1.85 GB per launch): 0.405 -> 0.373 ms (4.57 -> 4.96 TB/s), B = 2048
// 4.73 -> 5.04 TB/s, B = 512 level (0.205 ms): routed from M >= 131,072 rows (igemm_f32.hip, option conv1x1_res_stream).
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOR = 0x80000000u;
constexpr long long EXT_LIM = 0x7FFFFFF0LL;

struct ResStreamP {
    const float* a; const float* a_absmax;
    const u16* w; int w_exp; const float* sc; const float* b;
    const float* res; float* y; float* y_absmax;
    int M, N, n_tiles;
    float* yp; int Ho, Wo;                                 // POOL: AvgPool2d(2) of y [M / 4][N]; M = B * Ho * Wo pixels
};

// pixel (standard order) of tile row m: identity, or 2x2-window-major (m = 4 * pooled pixel + dy * 2 + dx)
template <int POOL>
__device__ __forceinline__ int row_pixel(const ResStreamP& p, int m) {
    if constexpr (!POOL) return m;
    const int mp = m >> 2, q = m & 3, wp2 = p.Wo >> 1, hwp = (p.Ho >> 1) * wp2;
    const int n = mp / hwp, rem = mp - n * hwp, hp = rem / wp2;
    return (n * p.Ho + 2 * hp + (q >> 1)) * p.Wo + 2 * (rem - hp * wp2) + (q & 1);
}

__device__ __forceinline__ int scale_exp(float amax) {      // s with amax * 2^s in [2^13, 2^14)
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
// LDS rows of RH halves: XOR of the 16-B chunk index with row bits keeps the 16 lanes of a ds_read_b128 group on distinct bank quads
template <int RH>
__device__ __forceinline__ int swz(int row) { return RH == 64 ? ((row >> 1) & 7) : (row & 15); }

constexpr int BM = 128, BNS = 32;

// POOL (the last block of a stage): the tile's rows walk the pixels in 2x2-window-major order, the four registers (r & 3) of an accumulator
// group are one pooling window, and the launch also writes AvgPool2d(2) of y (summed in (dy, dx) order like avgpool_kernel) for the next
// stage's downsample branch -- dbmm_conv_bn_act_x2 with pool = 2 and y_full.
template <int K, int POOL = 0>
__global__ __launch_bounds__(256, 2) void conv1x1_res_stream_kernel(const ResStreamP p) {
    static_assert(K == 256, "reduction depth (K = 128 compiles and is correct, but measured level with the tile kernel at M = 802,816 and 4 % behind it at 200,704: not instantiated)");
    constexpr int KS = K / 16;
    constexpr int AY_BYTES = 2 * BM * 64 * 2, W_BYTES = BNS * K * 2;
    __shared__ __attribute__((aligned(256))) unsigned char lds_raw[AY_BYTES + 2 * W_BYTES];
    u16* Ay = (u16*)lds_raw;                                       // [2 planes][128][64]: one 64-wide k pass of the a tile
    u16* Wr = (u16*)(lds_raw + AY_BYTES);                          // ring of two weight slabs [32][K]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int NT = p.N / BNS;
    const long long U = (long long)p.n_tiles * NT;
    long long u = U * blockIdx.x / gridDim.x;
    const long long u_end = U * (blockIdx.x + 1) / gridDim.x;
    const int s_a = scale_exp(*p.a_absmax);
    const float a_sc = pow2f(s_a), acc_scale = pow2f(-s_a - p.w_exp);
    const long long Mll = p.M;

    // weight slab [32 n][K]: 16-B chunks dealt over the 256 threads
    constexpr int CPR = K / 8, RPP = 256 / CPR, WLD = BNS * CPR / 256;
    const int wc = tid % CPR, wr = tid / CPR;
    u32x4 w3r[WLD];
    float y_amax = 0.f;

    while (u < u_end) {
        const int tile = (int)(u / NT), nt0 = (int)(u - (long long)tile * NT);
        const int nt1 = (long long)(NT - nt0) < u_end - u ? NT : nt0 + (int)(u_end - u);
        u += nt1 - nt0;
        const int m0 = tile * BM;
        const int g0 = row_pixel<POOL>(p, m0);                     // descriptors rebased to the tile's first pixel
        const __amdgpu_buffer_rsrc_t rsA = desc(p.a, Mll * K * 4, (long long)g0 * K * 4);
        const __amdgpu_buffer_rsrc_t rsR = desc(p.res, Mll * p.N * 4, (long long)g0 * p.N * 4);
        const __amdgpu_buffer_rsrc_t rsY = desc(p.y, Mll * p.N * 4, (long long)g0 * p.N * 4);
        __amdgpu_buffer_rsrc_t rsP = rsY;
        if constexpr (POOL) rsP = desc(p.yp, (Mll >> 2) * p.N * 4, (long long)(m0 >> 2) * p.N * 4);
        // this lane's 16 accumulator rows: 4 groups (t = r >> 2) of 4 consecutive tile rows 8 t + 4 fh + (r & 3); M % 4 == 0.  The group's
        // first pixel goes into the vector offset, the step inside the group (wave-uniform: q, or (dy Wo + dx) of a window) into the scalar one
        unsigned gx[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int row = wave * 32 + 8 * t + 4 * fh;
            gx[t] = m0 + row < p.M ? (unsigned)(row_pixel<POOL>(p, m0 + row) - g0) * (unsigned)(p.N * 4) + (unsigned)(fr * 4) : OOR;
        }
        auto qpix = [&](int q) { return POOL ? (q >> 1) * p.Wo + (q & 1) : q; };
        auto load_w = [&](int nt) {
#pragma unroll
            for (int j = 0; j < WLD; ++j) w3r[j] = *(const u32x4*)(p.w + (size_t)(nt * BNS + wr + RPP * j) * K + wc * 8);
        };
        auto store_w = [&](int slot) {
#pragma unroll
            for (int j = 0; j < WLD; ++j) {
                const int row = wr + RPP * j;
                *(u32x4*)(Wr + slot * (BNS * K) + row * K + ((wc ^ swz<K>(row)) << 3)) = w3r[j];
            }
        };
        float rr[2][16];
        auto load_res = [&](int nt, auto slot_c) {
            constexpr int S = decltype(slot_c)::value;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                rr[S][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, gx[r >> 2], (unsigned)((nt * BNS) * 4 + qpix(r & 3) * p.N * 4), 0));
        };
        // ---- segment prologue: first weight slab and residual slab in flight; the a tile -> fp16 planes in LDS -> A fragments ----
        load_w(nt0);
        load_res(nt0, std::integral_constant<int, 0>());
        u32x4 af[KS][2];
        {
            const int lc = tid & 15, lr = tid >> 4;
#pragma unroll
            for (int kp = 0; kp < K / 64; ++kp) {
                f32x4 q[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = lr + 16 * i;
                    const unsigned off = m0 + row < p.M ? (unsigned)(row_pixel<POOL>(p, m0 + row) - g0) * (unsigned)(K * 4) + lc * 16u : OOR;
                    q[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, off, (unsigned)(kp * 64 * 4), 0));
                }
                __syncthreads();                                  // the previous pass's fragments (the previous segment's last slab) have been read
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int row = lr + 16 * i;
                    unsigned hp[2], lp[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) split2h_pair(q[i][2 * j], q[i][2 * j + 1], a_sc, hp[j], lp[j]);
                    const int off = row * 64 + (((lc >> 1) ^ swz<64>(row)) << 3) + ((lc & 1) << 2);
                    *(u32x2*)(Ay + off) = (u32x2){hp[0], hp[1]};
                    *(u32x2*)(Ay + BM * 64 + off) = (u32x2){lp[0], lp[1]};
                }
                if (kp == 0) store_w(0);                          // (ring slot 0: its last readers left through the barrier above)
                __syncthreads();
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int row = wave * 32 + fr;
                    const int off = row * 64 + (((2 * ks + fh) ^ swz<64>(row)) << 3);
                    af[4 * kp + ks][0] = *(const u32x4*)(Ay + off);
                    af[4 * kp + ks][1] = *(const u32x4*)(Ay + BM * 64 + off);
                }
            }
        }
        if (nt0 + 1 < nt1) load_res(nt0 + 1, std::integral_constant<int, 1>());   // (after the staging: its 32 load registers are free again)
        // ---- slabs ----
        auto slab = [&](int nt, auto slot_c) {
            constexpr int S = decltype(slot_c)::value;
            if (nt + 1 < nt1) load_w(nt + 1);                      // lands during this slab's MFMAs
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const u16* Wb = Wr + S * (BNS * K);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const u32x4 wf = *(const u32x4*)(Wb + fr * K + (((2 * ks + fh) ^ swz<K>(fr)) << 3));
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[ks][1]), __builtin_bit_cast(f16x8, wf), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[ks][0]), __builtin_bit_cast(f16x8, wf), acc, 0, 0, 0);
            }
            const int n = nt * BNS + fr;
            const float sv = p.sc[n] * acc_scale, bv = p.b ? p.b[n] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = fmaxf(fmaf(acc[r], sv, bv) + rr[S][r], 0.f);
                acc[r] = v;
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsY, gx[r >> 2], (unsigned)((nt * BNS) * 4 + qpix(r & 3) * p.N * 4), 0);
                if (gx[r >> 2] != OOR) y_amax = fmaxf(y_amax, v);
            }
            if constexpr (POOL) {
#pragma unroll
                for (int t = 0; t < 4; ++t) {                      // window = registers 4 t .. 4 t + 3 of this lane
                    const float sm = (((acc[4 * t] + acc[4 * t + 1]) + acc[4 * t + 2]) + acc[4 * t + 3]) * 0.25f;
                    const int mp = wave * 8 + 2 * t + fh;          // pooled row within the tile
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sm), rsP,
                                                          gx[t] != OOR ? (unsigned)mp * (unsigned)(p.N * 4) + (unsigned)(fr * 4) : OOR, (unsigned)((nt * BNS) * 4), 0);
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
    // ---- max |y| for the consumer's fp16 scale: one filtered atomic per workgroup ----
    if (p.y_absmax) {
        y_amax = wave_max(y_amax);
        float* red = (float*)lds_raw;                              // (past the last barrier)
        if (lane == 0) red[wave] = y_amax;
        __syncthreads();
        if (tid == 0) {
            const float m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
            if (m > *(volatile const float*)p.y_absmax) atomicMax((unsigned*)p.y_absmax, __float_as_uint(m));
        }
    }
}

}  // namespace

// common.h: conv3 + BatchNorm + residual + ReLU with K = 256 (y_pooled: the rows are pixels of [B][Ho][Wo] maps and AvgPool2d(2) of y is
// written too); DBMM_E_UNSUPPORTED (nothing launched) for other shapes
int dbmm_conv1x1_res_stream(const float* a, const float* a_absmax, const void* w_plane_f16, int w_exp, const float* scale, const float* bias,
                            const float* residual, float* y, float* y_pooled, float* y_absmax, int64_t M, int64_t Ho, int64_t Wo, int64_t K,
                            int64_t N, void* stream) {
    if (!a || !a_absmax || !w_plane_f16 || !scale || !residual || !y) return DBMM_E_ARG;
    if (M <= 0 || N <= 0 || M > (INT32_MAX >> 1)) return DBMM_E_SHAPE;
    if (K != 256 || (N % BNS) || (M & 3) || w_exp < -40 || w_exp > 40) return DBMM_E_UNSUPPORTED;
    if (y_pooled && (Ho <= 0 || Wo <= 0 || (Ho & 1) || (Wo & 1) || M % (Ho * Wo))) return DBMM_E_UNSUPPORTED;
    if ((132LL + (y_pooled ? 2 * Wo : 0)) * N * 4 >= EXT_LIM) return DBMM_E_UNSUPPORTED;          // a tile's pixel span under its rebased descriptors
    if (!dbmm_aligned16(a) || !dbmm_aligned16(w_plane_f16) || !dbmm_aligned16(residual) || !dbmm_aligned16(y) || (y_pooled && !dbmm_aligned16(y_pooled)))
        return DBMM_E_ALIGN;
    ResStreamP p{};
    p.a = a; p.a_absmax = a_absmax; p.w = (const u16*)w_plane_f16; p.w_exp = w_exp; p.sc = scale; p.b = bias; p.res = residual; p.y = y;
    p.y_absmax = y_absmax; p.M = (int)M; p.N = (int)N; p.n_tiles = (int)((M + BM - 1) / BM);
    p.yp = y_pooled; p.Ho = (int)Ho; p.Wo = (int)Wo;
    const long long units = (long long)p.n_tiles * (N / BNS);
    const int grid = (int)(units < 512 ? units : 512);              // two workgroups per CU
    hipStream_t s = (hipStream_t)stream;
    if (y_pooled) hipLaunchKernelGGL((conv1x1_res_stream_kernel<256, 1>), dim3(grid), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((conv1x1_res_stream_kernel<256, 0>), dim3(grid), dim3(256), 0, s, p);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}
