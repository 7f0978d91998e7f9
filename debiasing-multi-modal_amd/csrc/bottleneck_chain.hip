// conv3 + residual of one bottleneck block CHAINED with conv1 of the next block in one launch.
//
//   x' = relu( (y2 @ W3^T) * sc3 + b3 + x )          [M][N]   written to HBM (it is the next residual)
//   y1' = relu( (x' @ W1^T) * sc1 + b1 )             [M][P]   written to HBM
//   (POOL: also avgpool2x2(x') [M/4][N] for the downsample branch of the next stage's first block; POOL = 2: ONLY the pooled
//    copy -- at a stage seam nothing else reads the un-pooled x', conv1' has just consumed it here)
//
// Why: at the headline batch the 1x1 convs of layers 1-2 are HBM-bound and conv1 of the NEXT block re-reads
// the 256/512-channel tensor conv3 has just written (clip/model.py:42-55 back to back): 3.3 GB of the 13 GB a
// layer-1 block moves at B = 1024.  Here one workgroup owns 128 pixel rows for ALL N output channels of conv3,
// walks them in 64-channel slabs, and every finished slab is at once the next 64-deep K chunk of conv1': the
// wide tensor is written once and never read back for conv1.
//
// Arithmetic = the fp16-pair path of igemm_f32.hip (fp32 value = fp16 hi + lo with an exact power-of-two scale,
// weights exact in one fp16 plane, fp32 accumulate).  conv3 takes its scale from the producer's device scalar
// (a_absmax); the slab's scale is WAVE-LOCAL: every wave owns 32 pixel rows through both GEMMs (4x1 wave layout),
// so it can use the running maximum of its own finished slabs and rescale its conv1' accumulators (exactly, by a
// power of two) when that maximum crosses a binade -- no a-priori bound, no grid-wide reduction.
//
// Data movement.  y2 tile -> LDS as fp16 planes (coalesced 16-B loads, all threads) -> A fragments live in
// registers for the whole tile.  Weights (W3 slab [64][K], W1 chunk [P][64]) are prefetched one slab ahead into
// registers and staged in LDS.  Residual loads and x' / y1' stores go straight from / to the MFMA accumulator
// layout: one register of a 32x32 accumulator is two 128-B row segments per wave instruction, which streams at
// full rate (MI355X_MICROARCH.md, "access shape per wave-instruction"); the next slab's residual is in flight
// during the current slab's conv1' MFMAs.  The finished slab goes to a wave-private fp32 LDS slab and is split
// into hi / lo when the conv1' A fragments are read.
//
// Bound: HBM.  Algorithmic bytes per pixel row: 4 * (K + 2 N + P) (+ N for the pooled copy).
//
// Tried and dropped (same-box A/B, B = 1024): the layer-3 shapes (K = 256, P = 256: 128 registers of A fragments + 128
// of conv1' accumulators = the whole 512-register file, one workgroup per CU) ran 1.15 ms against 0.68 ms for the two
// separate launches, and the layer 2 -> 3 seam (K = 128, P = 256) 1.71 against 1.37: with one wave per SIMD nothing
// hides the slab's residual round trip.  The chain serves the shapes that fit 2 workgroups per CU.
#include <stdlib.h>
#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOR = 0x80000000u;
constexpr long long EXT_LIM = 0x7FFFFFF0LL;

struct ChainP {
    const float* a; const float* a_absmax;                 // y2 [M][K] in standard pixel order
    const u16* w3; int w3_exp; const float* sc3; const float* b3;      // [N][K] fp16 plane of W3 * 2^w3_exp
    const float* res; float* x; float* xp; float* x_absmax;            // residual [M][N], x' [M][N], pooled [M/4][N]
    const u16* w1; int w1_exp; const float* sc1; const float* b1;      // [P][N] fp16 plane of W1 * 2^w1_exp
    float* y1; float* y1_absmax;                            // [M][P]
    int M, N;
    int Ho, Wo;                                             // POOL: map of the M = B * Ho * Wo pixels
    // DUAL (first block of a stage, clip/model.py:36-38,52): instead of adding a residual the block adds its
    // downsample branch bn_d(conv_d(a2)).  Like dbmm_gemm_dual_bn_act_x2 the branch's K2 = 64 chunk is accumulated
    // first, the accumulators are multiplied by ratio[n] * 2^(s - s2) and the main pair continues in them.
    const float* a2; const float* a2_absmax; const u16* wd; const float* ratio;     // a2 [M][64], wd [N][64] fp16 plane
    // CONV2: the launch starts one conv earlier -- `a` is y1 [M][K] (conv2's input, NHWC, M = B * H * W pixels in standard
    // order) and the y2 tile is computed here by the 3x3 / pad 1 conv2 + BatchNorm + ReLU instead of being read from HBM.
    const u16* w2; int w2_exp; const float* sc2; const float* b2;      // [K][(cin/32, kh, kw, 32)] fp16 plane of W2 * 2^w2_exp
    int H, W;
};

__device__ __forceinline__ int scale_exp(float amax) {      // s with amax * 2^s in [2^13, 2^14)
    const unsigned b = __float_as_uint(amax) & 0x7fffffffu;
    int s = b ? 13 - ((int)(b >> 23) - 127) : 0;
    return s < -60 ? -60 : (s > 60 ? 60 : s);
}
__device__ __forceinline__ float pow2f(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }

// (hi, lo) fp16 pairs of x0 * sc and x1 * sc, packed {x0 | x1 << 16}
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

// LDS rows of RH halves (2 * RH bytes): XOR of the 16-B chunk index with row bits keeps the 16 lanes of a ds_read_b128
// group on distinct bank quads -- 64-B rows: 4 rows per 256-B bank sweep; 128-B rows: 2; 256-B rows: every row starts a sweep
template <int RH>
__device__ __forceinline__ int swz(int row) { return RH == 32 ? ((row >> 2) & 3) : (RH == 64 ? ((row >> 1) & 7) : (row & 15)); }

constexpr int BM = 128;                      // pixel rows per workgroup

// K = 64 (layer 1): 64-channel slabs.  K = 128 (layer 2): the A fragments take 64 registers, so the slab (conv3
// accumulators, residual prefetch, weight prefetch) is halved to 32 channels to stay inside 256 VGPRs at 2 workgroups / CU.
template <int K, int P>
struct ChainGeo {
    static constexpr int BNS = K == 64 ? 64 : 32;                 // channels per slab
    static constexpr int SLROW = BNS + 4;                         // slab row pitch in floats (272 / 144 B: conflict-free b128 rows)
    static constexpr int R1_BYTES = BM * SLROW * 4 > 2 * BM * 64 * 2 ? BM * SLROW * 4 : 2 * BM * 64 * 2;   // slab | y2 planes (64 k at a time)
    static constexpr int W3_BYTES = BNS * K * 2;
    static constexpr int W1_BYTES = P * BNS * 2;
    static constexpr int LDS_BYTES = R1_BYTES + W3_BYTES + W1_BYTES;
};

// pixel (standard order) of tile row m: identity, or 2x2-window-major (m = 4 * pooled pixel + dy * 2 + dx)
template <int POOL>
__device__ __forceinline__ int row_pixel(const ChainP& p, int m) {
    if constexpr (!POOL) return m;
    const int mp = m >> 2, q = m & 3, wp2 = p.Wo >> 1, hwp = (p.Ho >> 1) * wp2;
    const int n = mp / hwp, rem = mp - n * hwp, hp = rem / wp2;
    return (n * p.Ho + 2 * hp + (q >> 1)) * p.Wo + 2 * (rem - hp * wp2) + (q & 1);
}

// CONV2 phase geometry: the activation strip of one (32-channel slab, kh) group -- 130 pixels [m0 - 1, m0 + 128] shifted by
// (kh - 1) image rows, as fp16 (hi, lo) planes -- and the group's three kw taps of W2
constexpr int C2_SROWS = 132, C2_PLANE = C2_SROWS * 32;          // strip rows (130 used), halves per plane (8448 B = 33 x 256)
constexpr int C2_W2P = 104;                                      // LDS pitch of a W2 group row in halves (96 used; 208 B rows: conflict-free)
template <int K>
struct Conv2Geo {
    static constexpr int W2_OFF = 2 * C2_PLANE;                   // halves
    static constexpr int GROUP_BYTES = (2 * C2_PLANE * 2 + K * C2_W2P * 2 + 255) / 256 * 256;   // strip planes + W2 group
    // K = 64: two group stages (60 KB with the zero line, still two workgroups per CU): the next group is split and stored
    // while this one computes, one barrier per group.  K = 128 would need 87 KB: one stage, two barriers per group.
    static constexpr int NST = K == 64 ? 2 : 1;
    static constexpr int Z_OFF = NST * GROUP_BYTES;               // bytes: 256-B zero line
    static constexpr int BYTES = Z_OFF + 256;
    static constexpr int STAGE_BYTES = BM * 68 * 4;               // y2 staging, 64 channels at a time
};
constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <int K, int P, int POOL, int DUAL = 0, int CONV2 = 0>
__global__ __launch_bounds__(256, 2) void bottleneck_chain_kernel(const ChainP p) {
    static_assert(!DUAL || K == 64, "dual-source variant: layer-1 geometry (K = K2 = 64)");
    static_assert(!CONV2 || !POOL, "the conv2 phase walks pixels in standard order");
    static_assert(K == 64 || K == 128, "conv3 reduction depth: 64 (layer 1) or 128 (layer 2)");
    static_assert(P == 64 || P == 128, "conv1' width");
    using G = ChainGeo<K, P>;
    constexpr int BNS = G::BNS, SLROW = G::SLROW;
    constexpr int KS = K / 16, TN3 = BNS / 32, TN1 = P / 32;
    constexpr int LDS_TOTAL = cmax(G::LDS_BYTES + (DUAL ? BNS * 64 * 2 : 0), CONV2 ? cmax(Conv2Geo<K>::BYTES, Conv2Geo<K>::STAGE_BYTES) : 0);
    __shared__ __attribute__((aligned(256))) unsigned char lds_raw[LDS_TOTAL];
    u16* Ay = (u16*)lds_raw;                                       // [2 planes][128][64] (prologue only)
    float* slab = (float*)lds_raw;                                 // [4 waves][32][SLROW]   (aliases Ay)
    u16* W3b = (u16*)(lds_raw + G::R1_BYTES);                      // [BNS][K]
    u16* W1b = (u16*)(lds_raw + G::R1_BYTES + G::W3_BYTES);        // [P][BNS]
    u16* Wdb = (u16*)(lds_raw + G::R1_BYTES + G::W3_BYTES + G::W1_BYTES);   // DUAL: [BNS][64]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int m0 = xcd_remap(blockIdx.x, gridDim.x) * BM;
    const int NT = p.N / BNS;

    // ---- descriptors rebased to the tile's first pixel (tensors may exceed 2 GiB) ---------------------------
    const int g0 = row_pixel<POOL>(p, m0);
    const long long Mll = p.M;
    const __amdgpu_buffer_rsrc_t rsA = desc(p.a, Mll * K * 4, (long long)g0 * K * 4);
    const __amdgpu_buffer_rsrc_t rsR = DUAL ? desc(p.a2, Mll * 64 * 4, (long long)g0 * 64 * 4)      // DUAL: the branch input
                                            : desc(p.res, Mll * p.N * 4, (long long)g0 * p.N * 4);
    const __amdgpu_buffer_rsrc_t rsX = desc(p.x, POOL == 2 ? 0 : Mll * p.N * 4, (long long)g0 * p.N * 4);
    const __amdgpu_buffer_rsrc_t rsY = desc(p.y1, Mll * P * 4, (long long)g0 * P * 4);
    __amdgpu_buffer_rsrc_t rsXP = rsX;
    if constexpr (POOL) rsXP = desc(p.xp, (Mll >> 2) * p.N * 4, (long long)(m0 >> 2) * p.N * 4);

    // This lane's 16 accumulator rows are 4 groups (t = r >> 2) of 4 consecutive tile rows 8t + 4fh + q (q = r & 3).
    // M is a multiple of 4, so a group is valid or not as a whole; its 4 pixels are the group's first pixel plus a
    // WAVE-UNIFORM step (q, or (dy * Wo + dx) of a 2x2 window), which goes into the scalar offset of every access:
    // no per-access address arithmetic.  Byte offsets relative to g0, + this lane's column; OOR = past M.
    unsigned gx[4], gy[4];                              // into the [.][N] tensors / the [.][P] tensor
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int m = m0 + wave * 32 + 8 * t + 4 * fh;
        const int dp = row_pixel<POOL>(p, m) - g0;
        gx[t] = m < p.M ? (unsigned)dp * (unsigned)(p.N * 4) + (unsigned)(fr * 4) : OOR;
        gy[t] = m < p.M ? (unsigned)dp * (unsigned)(P * 4) + (unsigned)(fr * 4) : OOR;
    }
    auto qpix = [&](int q) { return POOL ? (q >> 1) * p.Wo + (q & 1) : q; };       // wave-uniform pixel step of row q

    // ---- prologue: first weight slabs and residual slab in flight; y2 tile -> fp16 planes in LDS -> A fragments ----
    // weight slabs: W3 [BNS n][K] and the W1 chunk [P][BNS k], 16-B chunks dealt over the 256 threads
    constexpr int CPR3 = K / 8, RPP3 = 256 / CPR3, W3LD = BNS * CPR3 / 256;        // chunks per row, rows per pass, loads per thread
    constexpr int CPR1 = BNS / 8, RPP1 = 256 / CPR1, W1LD = P * CPR1 / 256;
    static_assert(W3LD >= 1 && W1LD >= 1, "weight slabs smaller than one pass of the workgroup");
    const int wc3 = tid % CPR3, wr3 = tid / CPR3, wc1 = tid % CPR1, wr1 = tid / CPR1;
    u32x4 w3r[W3LD], w1r[W1LD], wdr[DUAL ? W3LD : 1];
    auto load_w = [&](int nt) {
#pragma unroll
        for (int j = 0; j < W3LD; ++j) {
            w3r[j] = *(const u32x4*)(p.w3 + (size_t)(nt * BNS + wr3 + RPP3 * j) * K + wc3 * 8);
            if constexpr (DUAL) wdr[j] = *(const u32x4*)(p.wd + (size_t)(nt * BNS + wr3 + RPP3 * j) * 64 + wc3 * 8);
        }
#pragma unroll
        for (int j = 0; j < W1LD; ++j)
            w1r[j] = *(const u32x4*)(p.w1 + (size_t)(wr1 + RPP1 * j) * p.N + nt * BNS + wc1 * 8);
    };
    auto store_w = [&]() {
#pragma unroll
        for (int j = 0; j < W3LD; ++j) {
            const int row = wr3 + RPP3 * j;
            *(u32x4*)(W3b + row * K + ((wc3 ^ swz<K>(row)) << 3)) = w3r[j];
            if constexpr (DUAL) *(u32x4*)(Wdb + row * 64 + ((wc3 ^ swz<64>(row)) << 3)) = wdr[j];
        }
#pragma unroll
        for (int j = 0; j < W1LD; ++j) {
            const int row = wr1 + RPP1 * j;
            *(u32x4*)(W1b + row * BNS + ((wc1 ^ swz<BNS>(row)) << 3)) = w1r[j];
        }
    };
    load_w(0);
    // residual slab nt: this lane's 16 rows x TN3 column blocks, element (row r, col j*32 + fr)
    float rr[DUAL ? 1 : TN3][16];
    auto load_res = [&](int nt) {
        if constexpr (DUAL) return;
#pragma unroll
        for (int j = 0; j < TN3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                rr[j][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                    rsR, gx[r >> 2], (unsigned)((nt * BNS + j * 32) * 4 + qpix(r & 3) * p.N * 4), 0));
    };
    load_res(0);
    // y2, 64 k at a time: 128 rows x 16 quads, thread (lc = tid & 15, lr = tid >> 4) loads rows lr + 16 i, splits them
    // into (hi, lo) fp16 planes in LDS; then every wave reads the A fragments of its 32 rows into registers, where they
    // stay for the whole tile: lane (row fr, k half fh) holds 8 k values per 16-deep step
    u32x4 af[KS][2], af2[DUAL ? 4 : 1][2];
    bool first_pass = true;
    // one 64-wide k pass of an A operand (row pitch `ldk` floats, k offset `k0`) -> 4 fragment steps
    auto stage_issue = [&](const __amdgpu_buffer_rsrc_t& rs, int ldk, int k0, f32x4 (&q)[8]) {
        const int lc = tid & 15, lr = tid >> 4;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = lr + 16 * i, m = m0 + row;
            const unsigned off = m < p.M ? (unsigned)(row_pixel<POOL>(p, m) - g0) * (unsigned)(ldk * 4) + lc * 16u : OOR;
            q[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, (unsigned)(k0 * 4), 0));
        }
    };
    auto stage_finish = [&](const f32x4 (&q)[8], float sc, u32x4 (*dst)[2]) {
        const int lc = tid & 15, lr = tid >> 4;
        if (!first_pass) __syncthreads();                 // the previous pass's fragments have been read
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = lr + 16 * i;
            unsigned hp[2], lp[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) split2h_pair(q[i][2 * j], q[i][2 * j + 1], sc, hp[j], lp[j]);
            const int off = row * 64 + (((lc >> 1) ^ swz<64>(row)) << 3) + ((lc & 1) << 2);
            *(u32x2*)(Ay + off) = (u32x2){hp[0], hp[1]};
            *(u32x2*)(Ay + BM * 64 + off) = (u32x2){lp[0], lp[1]};
        }
        if (first_pass) store_w();
        first_pass = false;
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int row = wave * 32 + fr;
            const int off = row * 64 + (((2 * ks + fh) ^ swz<64>(row)) << 3);
            dst[ks][0] = *(const u32x4*)(Ay + off);
            dst[ks][1] = *(const u32x4*)(Ay + BM * 64 + off);
        }
    };
    auto stage_pass = [&](const __amdgpu_buffer_rsrc_t& rs, int ldk, int k0, float sc, u32x4 (*dst)[2]) {
        f32x4 q[8];
        stage_issue(rs, ldk, k0, q);
        stage_finish(q, sc, dst);
    };
    // DUAL + CONV2: the branch input's loads are issued BEFORE the conv2 phase and consumed after it (their round trip
    // would otherwise sit exposed between the two phases of every tile)
    f32x4 q2[(DUAL && CONV2) ? 8 : 1];
    if constexpr (DUAL && CONV2) stage_issue(rsR, 64, 0, q2);
    int s_a;                                            // exponent of the scale the A fragments carry
    if constexpr (!CONV2) {
        s_a = scale_exp(*p.a_absmax);
#pragma unroll
        for (int kp = 0; kp < K / 64; ++kp) stage_pass(rsA, K, kp * 64, pow2f(s_a), af + kp * 4);
    } else {
        // ---- conv2 (3x3, pad 1) + BatchNorm + ReLU of this tile: y2 never goes to HBM --------------------------------
        // K-loop groups g = (32-channel slab, kh): the 130-pixel strip [m0 - 1, m0 + 128] + (kh - 1) * W serves the three
        // kw taps by LDS row shifts (igemm_halo_kernel's idea), the group's W2 columns are 96 contiguous halves per output
        // channel.  One LDS stage, the next group in flight in registers, two barriers per group.  Border taps read a
        // 256-B zero line at their own offset modulo 256 B.
        using C2 = Conv2Geo<K>;
        constexpr int TN2 = K / 32, NG = (K / 32) * 3, W2LD = (K * 12 + 255) / 256;
        u16* Sp = (u16*)lds_raw;                                     // per stage: [2 planes][132][32] then [K][C2_W2P]
        constexpr int NST = C2::NST, STH = C2::GROUP_BYTES / 2;      // stages, halves per stage
        constexpr int ZLH = C2::Z_OFF / 2;                           // zero line, halves from the LDS base
        if (tid < 16) *(u32x4*)(lds_raw + C2::Z_OFF + tid * 16) = (u32x4){0u, 0u, 0u, 0u};
        const int s_y1 = scale_exp(*p.a_absmax);
        const float y1_sc = pow2f(s_y1);
        const int px0 = m0 - 1 - p.W > 0 ? m0 - 1 - p.W : 0;
        const __amdgpu_buffer_rsrc_t rsS = desc(p.a, Mll * K * 4, (long long)px0 * K * 4);
        const int lc = tid & 7, lr = tid >> 3;                       // strip: k-quad lc of rows lr + 32 i
        unsigned s_off[5];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int j = lr + 32 * i;
            s_off[i] = j < 130 ? (unsigned)((m0 - 1 + j - px0) * (K * 4)) + lc * 16u : OOR;
        }
        // this lane's output pixel (A-operand row 32 wave + fr): tap validity, bit kh * 3 + kw
        unsigned fmask = 0;
        {
            const int m = m0 + wave * 32 + fr;
            if (m < p.M) {
                const int hw = p.H * p.W, rem = m - (m / hw) * hw, ho = rem / p.W, wo = rem - ho * p.W;
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
                        if (ho + kh - 1 >= 0 && ho + kh - 1 < p.H && wo + kw - 1 >= 0 && wo + kw - 1 < p.W) fmask |= 1u << (kh * 3 + kw);
            }
        }
        f32x4 sq[5];
        u32x4 w2q[W2LD];
        auto load_group = [&](int g) {
            const int sl = g / 3, kh = g - sl * 3;
            const unsigned delta = (unsigned)(((kh - 1) * p.W * K + sl * 32) * 4);
#pragma unroll
            for (int i = 0; i < 5; ++i)
                sq[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsS, s_off[i] == OOR ? OOR : s_off[i] + delta, 0, 0));
#pragma unroll
            for (int i = 0; i < W2LD; ++i) {
                const int idx = tid + 256 * i, row = idx / 12, c = idx - row * 12;
                w2q[i] = idx < K * 12 ? *(const u32x4*)(p.w2 + (size_t)row * (9 * K) + (size_t)g * 96 + c * 8) : (u32x4){0u, 0u, 0u, 0u};
            }
        };
        auto store_group = [&](int st) {
            u16* Sps = Sp + st * STH;
            u16* W2b = Sps + C2::W2_OFF;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int row = lr + 32 * i;
                if (row < 130) {
                    unsigned hp[2], lp[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) split2h_pair(sq[i][2 * j], sq[i][2 * j + 1], y1_sc, hp[j], lp[j]);
                    const int off = row * 32 + (((lc >> 1) ^ swz<32>(row)) << 3) + ((lc & 1) << 2);
                    *(u32x2*)(Sps + off) = (u32x2){hp[0], hp[1]};
                    *(u32x2*)(Sps + C2_PLANE + off) = (u32x2){lp[0], lp[1]};
                }
            }
#pragma unroll
            for (int i = 0; i < W2LD; ++i) {
                const int idx = tid + 256 * i, row = idx / 12, c = idx - row * 12;
                if (idx < K * 12) *(u32x4*)(W2b + row * C2_W2P + c * 8) = w2q[i];
            }
        };
        f32x16 acc2[TN2];
#pragma unroll
        for (int j = 0; j < TN2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[j][r] = 0.f;
        auto compute_group = [&](int g, int st) {
            const u16* Sps = Sp + st * STH;
            const u16* W2b = Sps + C2::W2_OFF;
            const int zl = ZLH - st * STH;                           // the zero line relative to this stage
            const int kh = g % 3;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const bool ok = (fmask >> (kh * 3 + kw)) & 1u;
                const int row = wave * 32 + fr + kw;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const int addr = row * 32 + (((2 * ks + fh) ^ swz<32>(row)) << 3);
                    const int a0 = ok ? addr : zl + (addr & 127), a1 = ok ? addr + C2_PLANE : zl + (addr & 127);
                    const u32x4 ah = *(const u32x4*)(Sps + a0), al = *(const u32x4*)(Sps + a1);
                    u32x4 wf[TN2];
#pragma unroll
                    for (int j = 0; j < TN2; ++j) wf[j] = *(const u32x4*)(W2b + (j * 32 + fr) * C2_W2P + kw * 32 + (2 * ks + fh) * 8);
#pragma unroll
                    for (int j = 0; j < TN2; ++j)
                        acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al), __builtin_bit_cast(f16x8, wf[j]), acc2[j], 0, 0, 0);
#pragma unroll
                    for (int j = 0; j < TN2; ++j)
                        acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, wf[j]), acc2[j], 0, 0, 0);
                }
            }
        };
        load_group(0);
        if constexpr (NST == 2) {
            // two stages: group g + 1 is fetched (registers), split and stored into the other stage while group g computes;
            // the stage it overwrites was last read in iteration g - 1, which every wave left through that iteration's barrier
            store_group(0);
            __syncthreads();
            for (int g = 0; g < NG; ++g) {
                if (g + 1 < NG) load_group(g + 1);
                compute_group(g, g & 1);
                if (g + 1 < NG) store_group((g + 1) & 1);
                __syncthreads();
            }
        } else {
            for (int g = 0; g < NG; ++g) {
                store_group(0);
                __syncthreads();
                if (g + 1 < NG) load_group(g + 1);
                compute_group(g, 0);
                __syncthreads();
            }
        }
        // y2 = relu(bn2(.)) in the accumulator layout; its fp16 scale is the wave's own maximum (the wave's 32 rows feed only
        // this wave's conv3); through a wave-private fp32 staging block, 64 channels at a time, into A fragments
        const float c2_scale = pow2f(-s_y1 - p.w2_exp);
        float ymax = 0.f;
#pragma unroll
        for (int j = 0; j < TN2; ++j) {
            const float sv = p.sc2[j * 32 + fr] * c2_scale, bv = p.b2[j * 32 + fr];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = fmaxf(fmaf(acc2[j][r], sv, bv), 0.f);
                if (gx[r >> 2] == OOR) v = 0.f;
                acc2[j][r] = v;
                ymax = fmaxf(ymax, v);
            }
        }
        s_a = scale_exp(wave_max(ymax));
        const float y2_sc = pow2f(s_a);
        float* St = (float*)lds_raw + wave * (32 * 68);
#pragma unroll
        for (int h = 0; h < K / 64; ++h) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int r = 0; r < 16; ++r) St[((r & 3) + 8 * (r >> 2) + 4 * fh) * 68 + jj * 32 + fr] = acc2[2 * h + jj][r];
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const f32x4 x0 = *(const f32x4*)(St + fr * 68 + 16 * ks + 8 * fh), x1 = *(const f32x4*)(St + fr * 68 + 16 * ks + 8 * fh + 4);
                unsigned hh[4], ll[4];
                split2h_pair(x0[0], x0[1], y2_sc, hh[0], ll[0]); split2h_pair(x0[2], x0[3], y2_sc, hh[1], ll[1]);
                split2h_pair(x1[0], x1[1], y2_sc, hh[2], ll[2]); split2h_pair(x1[2], x1[3], y2_sc, hh[3], ll[3]);
                af[4 * h + ks][0] = (u32x4){hh[0], hh[1], hh[2], hh[3]};
                af[4 * h + ks][1] = (u32x4){ll[0], ll[1], ll[2], ll[3]};
            }
        }
        __syncthreads();                                // staging done in every wave: the chain's buffers may be filled
        if constexpr (!DUAL) store_w();
    }
    float dual_dyn = 1.f;
    if constexpr (DUAL) {
        const int s_a2 = scale_exp(*p.a2_absmax);
        if constexpr (CONV2) stage_finish(q2, pow2f(s_a2), af2);
        else stage_pass(rsR, 64, 0, pow2f(s_a2), af2);
        dual_dyn = pow2f(s_a - s_a2);
    }
    __syncthreads();                                    // the y2 planes are dead: the region becomes the slab

    float* Ls = slab + wave * (32 * SLROW);
    const float acc3_scale = pow2f(-s_a - p.w3_exp);
    f32x16 acc1[TN1];
#pragma unroll
    for (int j = 0; j < TN1; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[j][r] = 0.f;
    float run_max = 0.f, x_amax = 0.f;
    int s_cur = 0;                                      // exponent the conv1' accumulators are scaled by so far

    for (int nt = 0; nt < NT; ++nt) {
        const int n0 = nt * BNS;
        if (nt + 1 < NT) load_w(nt + 1);                // lands during this slab's MFMAs
        // ---- conv3 slab: acc3[32 x 64] = y2 rows . W3[n0 .. n0+63]^T ----------------------------------------
        f32x16 acc3[TN3];
#pragma unroll
        for (int j = 0; j < TN3; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc3[j][r] = 0.f;
        if constexpr (DUAL) {                            // downsample branch first, then into the main pair's units
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                u32x4 wf[TN3];
#pragma unroll
                for (int j = 0; j < TN3; ++j) {
                    const int row = j * 32 + fr;
                    wf[j] = *(const u32x4*)(Wdb + row * 64 + (((2 * ks + fh) ^ swz<64>(row)) << 3));
                }
#pragma unroll
                for (int pl = 1; pl >= 0; --pl)
#pragma unroll
                    for (int j = 0; j < TN3; ++j)
                        acc3[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af2[ks][pl]),
                                                                         __builtin_bit_cast(f16x8, wf[j]), acc3[j], 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < TN3; ++j) {
                const float rt = p.ratio[n0 + j * 32 + fr] * dual_dyn;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc3[j][r] *= rt;
            }
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            u32x4 wf[TN3];
#pragma unroll
            for (int j = 0; j < TN3; ++j) {
                const int row = j * 32 + fr;
                wf[j] = *(const u32x4*)(W3b + row * K + (((2 * ks + fh) ^ swz<K>(row)) << 3));
            }
#pragma unroll
            for (int pl = 1; pl >= 0; --pl)              // (lo, w) first, then (hi, w)
#pragma unroll
                for (int j = 0; j < TN3; ++j)
                    acc3[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[ks][pl]),
                                                                     __builtin_bit_cast(f16x8, wf[j]), acc3[j], 0, 0, 0);
        }
        // ---- epilogue: BatchNorm scale / bias, residual, ReLU; store x'; slab; (pooled copy) ---------------------
        float tmax = 0.f;
#pragma unroll
        for (int j = 0; j < TN3; ++j) {
            const int n = n0 + j * 32 + fr;
            const float sv = p.sc3[n] * acc3_scale, bv = p.b3[n];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = fmaxf(fmaf(acc3[j][r], sv, bv) + (DUAL ? 0.f : rr[j][r]), 0.f);
                if (gx[r >> 2] == OOR) v = 0.f;          // rows past M: keep the slab clean (their stores are dropped)
                acc3[j][r] = v;
                tmax = fmaxf(tmax, v);
                if constexpr (POOL != 2)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsX, gx[r >> 2],
                                                          (unsigned)((n0 + j * 32) * 4 + qpix(r & 3) * p.N * 4), 0);
                Ls[((r & 3) + 8 * (r >> 2) + 4 * fh) * SLROW + j * 32 + fr] = v;
            }
            if constexpr (POOL) {
                // rows 4i .. 4i+3 of the wave are one 2x2 window = registers 4t .. 4t+3 of this lane (i = 2t + fh);
                // summed in (dy, dx) order like avgpool_kernel
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const float s = (((acc3[j][4 * t] + acc3[j][4 * t + 1]) + acc3[j][4 * t + 2]) + acc3[j][4 * t + 3]) * 0.25f;
                    const int mp = wave * 8 + 2 * t + fh;                    // pooled row within the tile
                    const unsigned off = gx[t] != OOR ? (unsigned)mp * (unsigned)(p.N * 4) + (unsigned)(fr * 4) : OOR;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, s), rsXP, off, (unsigned)((n0 + j * 32) * 4), 0);
                }
            }
        }
        if (nt + 1 < NT) load_res(nt + 1);              // in flight during the conv1' MFMAs below
        // ---- conv1' chunk: acc1[32 x P] += slab[32 x 64] . W1[:, n0 .. n0+63]^T, slab scale = wave-local ---------
        tmax = wave_max(tmax);
        x_amax = fmaxf(x_amax, tmax);
        if (tmax > run_max) {
            run_max = tmax;
            const int s_new = scale_exp(run_max);
            if (s_new != s_cur) {                        // wave-uniform: the running maximum crossed a binade
                const float f = pow2f(s_new - s_cur);
#pragma unroll
                for (int j = 0; j < TN1; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc1[j][r] *= f;
                s_cur = s_new;
            }
        }
        const float t_sc = pow2f(s_cur);
#pragma unroll
        for (int ks = 0; ks < BNS / 16; ++ks) {
            const f32x4 x0 = *(const f32x4*)(Ls + fr * SLROW + 16 * ks + 8 * fh);
            const f32x4 x1 = *(const f32x4*)(Ls + fr * SLROW + 16 * ks + 8 * fh + 4);
            unsigned h[4], l[4];
            split2h_pair(x0[0], x0[1], t_sc, h[0], l[0]); split2h_pair(x0[2], x0[3], t_sc, h[1], l[1]);
            split2h_pair(x1[0], x1[1], t_sc, h[2], l[2]); split2h_pair(x1[2], x1[3], t_sc, h[3], l[3]);
            const u32x4 ah = {h[0], h[1], h[2], h[3]}, al = {l[0], l[1], l[2], l[3]};
            u32x4 wf[TN1];
#pragma unroll
            for (int j = 0; j < TN1; ++j) {
                const int row = j * 32 + fr;
                wf[j] = *(const u32x4*)(W1b + row * BNS + (((2 * ks + fh) ^ swz<BNS>(row)) << 3));
            }
#pragma unroll
            for (int j = 0; j < TN1; ++j)
                acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al), __builtin_bit_cast(f16x8, wf[j]),
                                                                 acc1[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TN1; ++j)
                acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, wf[j]),
                                                                 acc1[j], 0, 0, 0);
        }
        __syncthreads();                                // every wave is done with this slab's weights
        if (nt + 1 < NT) store_w();
        __syncthreads();
    }

    // ---- y1' = relu(bn1(acc1)) ---------------------------------------------------------------------------------
    float y_amax = 0.f;
    const float acc1_scale = pow2f(-s_cur - p.w1_exp);
#pragma unroll
    for (int j = 0; j < TN1; ++j) {
        const int n = j * 32 + fr;
        const float sv = p.sc1[n] * acc1_scale, bv = p.b1[n];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = fmaxf(fmaf(acc1[j][r], sv, bv), 0.f);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsY, gy[r >> 2],
                                                  (unsigned)(j * 32 * 4 + qpix(r & 3) * P * 4), 0);
            if (gy[r >> 2] != OOR) y_amax = fmaxf(y_amax, v);
        }
    }
    // ---- maxima for the consumers' fp16 scales: one filtered atomic per workgroup and tensor ------------------
    y_amax = wave_max(y_amax);
    float* red = (float*)W3b;                           // (past the last barrier: the weight slabs are dead)
    if (lane == 0) { red[wave] = x_amax; red[4 + wave] = y_amax; }
    __syncthreads();
    if (tid == 0) {
        const float xm = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        const float ym = fmaxf(fmaxf(red[4], red[5]), fmaxf(red[6], red[7]));
        if (p.x_absmax && xm > *(volatile const float*)p.x_absmax) atomicMax((unsigned*)p.x_absmax, __float_as_uint(xm));
        if (p.y1_absmax && ym > *(volatile const float*)p.y1_absmax) atomicMax((unsigned*)p.y1_absmax, __float_as_uint(ym));
    }
}

}  // namespace

// bottleneck_chain8.hip: the layer-3 geometry (K = P = 256) on eight waves
int dbmm_chain8_launch(const float* y2, const float* y2_absmax, const void* w3_plane_f16, int w3_exp, const float* scale3,
                       const float* bias3, const float* residual, float* x_out, float* x_absmax, const void* w1_plane_f16, int w1_exp,
                       const float* scale1, const float* bias1, float* y1_out, float* y1_absmax, int64_t M, int64_t N, void* stream);

// see include/dbmm.h
extern "C" int dbmm_bottleneck_block_chain_x2(const float* y1, const float* y1_absmax, const void* w2_plane_f16, int w2_exp,
                                              const float* scale2, const float* bias2, const void* w3_plane_f16, int w3_exp,
                                              const float* scale3, const float* bias3, const float* residual, const float* a2,
                                              const float* a2_absmax, const void* wd_plane_f16, const float* ratio, float* x_out,
                                              float* x_absmax, const void* w1_plane_f16, int w1_exp, const float* scale1,
                                              const float* bias1, float* y1_out, float* y1_out_absmax, int64_t B, int64_t H,
                                              int64_t W, int64_t K, int64_t N, int64_t P, void* stream) {
    const bool dual = a2 != nullptr;
    if (!y1 || !y1_absmax || !w2_plane_f16 || !scale2 || !bias2 || !w3_plane_f16 || !scale3 || !bias3 || !x_out || !w1_plane_f16 ||
        !scale1 || !bias1 || !y1_out)
        return DBMM_E_ARG;
    if (dual ? (!a2_absmax || !wd_plane_f16 || !ratio || residual) : (!residual || a2_absmax || wd_plane_f16 || ratio)) return DBMM_E_ARG;
    if (B <= 0 || H <= 0 || W <= 0 || K <= 0 || N <= 0 || P <= 0) return DBMM_E_SHAPE;
    const int64_t M = B * H * W;
    if (M > (INT32_MAX >> 1)) return DBMM_E_SHAPE;
    const bool served = dual ? (K == 64 && P == 64) : ((K == 64 && P == 64) || (K == 128 && P == 128));
    if (!served || (N % 64) != 0 || (M & 3)) return DBMM_E_UNSUPPORTED;
    if (w2_exp < -40 || w2_exp > 40 || w3_exp < -40 || w3_exp > 40 || w1_exp < -40 || w1_exp > 40) return DBMM_E_UNSUPPORTED;
    if (!dbmm_aligned16(y1) || !dbmm_aligned16(w2_plane_f16) || !dbmm_aligned16(w3_plane_f16) || !dbmm_aligned16(w1_plane_f16) ||
        !dbmm_aligned16(x_out) || !dbmm_aligned16(y1_out) || (residual && !dbmm_aligned16(residual)) || (a2 && !dbmm_aligned16(a2)) ||
        (wd_plane_f16 && !dbmm_aligned16(wd_plane_f16)))
        return DBMM_E_ALIGN;
    ChainP p{};
    p.a = y1; p.a_absmax = y1_absmax;
    p.w2 = (const u16*)w2_plane_f16; p.w2_exp = w2_exp; p.sc2 = scale2; p.b2 = bias2; p.H = (int)H; p.W = (int)W;
    p.w3 = (const u16*)w3_plane_f16; p.w3_exp = w3_exp; p.sc3 = scale3; p.b3 = bias3;
    p.res = residual; p.a2 = a2; p.a2_absmax = a2_absmax; p.wd = (const u16*)wd_plane_f16; p.ratio = ratio;
    p.x = x_out; p.x_absmax = x_absmax;
    p.w1 = (const u16*)w1_plane_f16; p.w1_exp = w1_exp; p.sc1 = scale1; p.b1 = bias1;
    p.y1 = y1_out; p.y1_absmax = y1_out_absmax;
    p.M = (int)M; p.N = (int)N; p.Ho = (int)H; p.Wo = (int)W;
    const dim3 grid((unsigned)((M + BM - 1) / BM));
    hipStream_t s = (hipStream_t)stream;
    if (dual) hipLaunchKernelGGL((bottleneck_chain_kernel<64, 64, 0, 1, 1>), grid, dim3(256), 0, s, p);
    else if (K == 64) hipLaunchKernelGGL((bottleneck_chain_kernel<64, 64, 0, 0, 1>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((bottleneck_chain_kernel<128, 128, 0, 0, 1>), grid, dim3(256), 0, s, p);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

extern "C" int dbmm_bottleneck_chain_dual_x2(const float* y2, const float* y2_absmax, const void* w3_plane_f16, int w3_exp,
                                             const float* scale3, const float* bias, const float* a2, const float* a2_absmax,
                                             const void* wd_plane_f16, const float* ratio, float* x_out, float* x_absmax,
                                             const void* w1_plane_f16, int w1_exp, const float* scale1, const float* bias1,
                                             float* y1_out, float* y1_absmax, int64_t M, int64_t K, int64_t K2, int64_t N,
                                             int64_t P, void* stream) {
    if (!y2 || !y2_absmax || !w3_plane_f16 || !scale3 || !bias || !a2 || !a2_absmax || !wd_plane_f16 || !ratio || !x_out ||
        !w1_plane_f16 || !scale1 || !bias1 || !y1_out)
        return DBMM_E_ARG;
    if (M <= 0 || K <= 0 || K2 <= 0 || N <= 0 || P <= 0 || M > (INT32_MAX >> 1)) return DBMM_E_SHAPE;
    if (K != 64 || K2 != 64 || (N % 64) != 0 || (P != 64 && P != 128) || (M & 3)) return DBMM_E_UNSUPPORTED;
    if (w3_exp < -40 || w3_exp > 40 || w1_exp < -40 || w1_exp > 40) return DBMM_E_UNSUPPORTED;
    if (!dbmm_aligned16(y2) || !dbmm_aligned16(a2) || !dbmm_aligned16(w3_plane_f16) || !dbmm_aligned16(wd_plane_f16) ||
        !dbmm_aligned16(w1_plane_f16) || !dbmm_aligned16(x_out) || !dbmm_aligned16(y1_out))
        return DBMM_E_ALIGN;
    ChainP p{};
    p.a = y2; p.a_absmax = y2_absmax;
    p.w3 = (const u16*)w3_plane_f16; p.w3_exp = w3_exp; p.sc3 = scale3; p.b3 = bias;
    p.a2 = a2; p.a2_absmax = a2_absmax; p.wd = (const u16*)wd_plane_f16; p.ratio = ratio;
    p.x = x_out; p.x_absmax = x_absmax;
    p.w1 = (const u16*)w1_plane_f16; p.w1_exp = w1_exp; p.sc1 = scale1; p.b1 = bias1;
    p.y1 = y1_out; p.y1_absmax = y1_absmax;
    p.M = (int)M; p.N = (int)N; p.Ho = 1; p.Wo = (int)M;
    const dim3 grid((unsigned)((M + BM - 1) / BM));
    hipStream_t s = (hipStream_t)stream;
    if (P == 64) hipLaunchKernelGGL((bottleneck_chain_kernel<64, 64, 0, 1>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((bottleneck_chain_kernel<64, 128, 0, 1>), grid, dim3(256), 0, s, p);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

extern "C" int dbmm_bottleneck_chain_x2(const float* y2, const float* y2_absmax, const void* w3_plane_f16, int w3_exp,
                                        const float* scale3, const float* bias3, const float* residual, float* x_out,
                                        float* x_pooled, float* x_absmax, const void* w1_plane_f16, int w1_exp,
                                        const float* scale1, const float* bias1, float* y1_out, float* y1_absmax,
                                        int64_t B, int64_t Ho, int64_t Wo, int64_t K, int64_t N, int64_t P, void* stream) {
    if (!y2 || !y2_absmax || !w3_plane_f16 || !scale3 || !bias3 || !residual || !w1_plane_f16 || !scale1 || !bias1 || !y1_out)
        return DBMM_E_ARG;
    if (!x_out && !x_pooled) return DBMM_E_ARG;         // x_out may be NULL only when the pooled copy is what the caller keeps
    if (B <= 0 || Ho <= 0 || Wo <= 0 || K <= 0 || N <= 0 || P <= 0) return DBMM_E_SHAPE;
    const int64_t M = B * Ho * Wo;
    if (M > (INT32_MAX >> 1)) return DBMM_E_SHAPE;
    if (M & 3) return DBMM_E_UNSUPPORTED;               // the kernel validates rows in groups of 4
    if (K == 256 && P == 256 && !x_pooled && dbmm_opt(OPT_CHAIN8)) {
        if (w3_exp < -40 || w3_exp > 40 || w1_exp < -40 || w1_exp > 40) return DBMM_E_UNSUPPORTED;
        if (!dbmm_aligned16(y2) || !dbmm_aligned16(w3_plane_f16) || !dbmm_aligned16(w1_plane_f16) || !dbmm_aligned16(residual) ||
            !dbmm_aligned16(x_out) || !dbmm_aligned16(y1_out))
            return DBMM_E_ALIGN;
        return dbmm_chain8_launch(y2, y2_absmax, w3_plane_f16, w3_exp, scale3, bias3, residual, x_out, x_absmax, w1_plane_f16, w1_exp,
                                  scale1, bias1, y1_out, y1_absmax, M, N, stream);
    }
    if ((K != 64 && K != 128) || (N % 64) != 0 || (P != 64 && P != 128)) return DBMM_E_UNSUPPORTED;
    if (w3_exp < -40 || w3_exp > 40 || w1_exp < -40 || w1_exp > 40) return DBMM_E_UNSUPPORTED;
    if (x_pooled && ((Ho & 1) || (Wo & 1))) return DBMM_E_UNSUPPORTED;
    if (!dbmm_aligned16(y2) || !dbmm_aligned16(w3_plane_f16) || !dbmm_aligned16(w1_plane_f16) || !dbmm_aligned16(residual) ||
        (x_out && !dbmm_aligned16(x_out)) || !dbmm_aligned16(y1_out) || (x_pooled && !dbmm_aligned16(x_pooled)))
        return DBMM_E_ALIGN;
    ChainP p{};
    p.a = y2; p.a_absmax = y2_absmax;
    p.w3 = (const u16*)w3_plane_f16; p.w3_exp = w3_exp; p.sc3 = scale3; p.b3 = bias3;
    p.res = residual; p.x = x_out; p.xp = x_pooled; p.x_absmax = x_absmax;
    p.w1 = (const u16*)w1_plane_f16; p.w1_exp = w1_exp; p.sc1 = scale1; p.b1 = bias1;
    p.y1 = y1_out; p.y1_absmax = y1_absmax;
    p.M = (int)M; p.N = (int)N; p.Ho = (int)Ho; p.Wo = (int)Wo;
    const dim3 grid((unsigned)((M + BM - 1) / BM));
    hipStream_t s = (hipStream_t)stream;
#define CHAIN_LAUNCH(KK, PP, PL) hipLaunchKernelGGL((bottleneck_chain_kernel<KK, PP, PL>), grid, dim3(256), 0, s, p)
    if (K == 64) {
        if (!x_out)        { if (P == 64) CHAIN_LAUNCH(64, 64, 2); else CHAIN_LAUNCH(64, 128, 2); }
        else if (x_pooled) { if (P == 64) CHAIN_LAUNCH(64, 64, 1); else CHAIN_LAUNCH(64, 128, 1); }
        else               { if (P == 64) CHAIN_LAUNCH(64, 64, 0); else CHAIN_LAUNCH(64, 128, 0); }
    } else {
        if (!x_out)        { if (P == 64) CHAIN_LAUNCH(128, 64, 2); else CHAIN_LAUNCH(128, 128, 2); }
        else if (x_pooled) { if (P == 64) CHAIN_LAUNCH(128, 64, 1); else CHAIN_LAUNCH(128, 128, 1); }
        else               { if (P == 64) CHAIN_LAUNCH(128, 64, 0); else CHAIN_LAUNCH(128, 128, 0); }
    }
#undef CHAIN_LAUNCH
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}
