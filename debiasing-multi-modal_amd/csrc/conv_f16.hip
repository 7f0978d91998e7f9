// fp16 mode of the ModifiedResNet towers (the reference's GPU path: clip/model.py:146 casts the image to the conv weights'
// dtype, convert_weights :375-396 leaves every conv weight in fp16, BatchNorm parameters stay fp32): activations are fp16
// NHWC in HBM, every product is ONE fp16 MFMA (v_mfma_f32_32x32x16_f16) with fp32 accumulation; eval-mode BatchNorm
// (per-channel scale / bias in fp32), ReLU, the residual add and the 2x2 average pools are applied to the fp32
// accumulator and the result is rounded to fp16 ONCE when stored (the reference rounds after the conv, after the
// BatchNorm and after the pool: this path is at least as accurate; tests pin it to the reference's own fp16 outputs).
//
//   conv3x3_f16_kernel<WM, WN, TN, POOL>   3x3 / stride 1 / pad 1 conv + BN + ReLU (+ AvgPool2d(2)), implicit GEMM:
//       * a wave owns 128 pixels x 32 TN channels (4 x TN MFMA tiles: 4 + TN fragment reads per 4 TN MFMAs -- the LDS
//         bandwidth a 64 x 64 wave tile spends per MFMA is what caps the 128 x 128 two-barrier kernels at half the peak);
//         workgroup = 4 waves = 256 x 128 (WM, WN = 2, 2) or 512 x 64 / 512 x 32 pixels x channels (4, 1);
//       * K order (cin / 32, kh, kw, 32).  For one (32-channel slab, kh) GROUP the three kw taps read the SAME BM + 2
//         consecutive pixels shifted by 0 / 1 / 2 rows, so the strip goes to LDS once per group and the taps are LDS
//         row shifts; border taps (image edges, rows past M) are masked when the fragment is read: the lane's address is
//         redirected to a 256-B zero line at the same offset modulo 256 B (bank-neutral);
//       * both operands reach LDS by LDS-DMA (buffer_load ... lds, 1 KB per wave instruction, the XOR swizzle that
//         keeps ds_read_b128 conflict-free applied to the SOURCE chunk): no staging registers, no ds_write.  A strips
//         are double-buffered per group, the per-tap W blocks (BN x 32) sit in a ring of 3-4 slots staged 2-3 taps
//         ahead; waits are counted (s_waitcnt vmcnt(N) never drains the queue inside the loop); one barrier per tap =
//         per 16 MFMAs of a wave; two workgroups per CU cover each other's barriers;
//       * POOL: tile rows run 2x2-window-major (row = 4 * pooled pixel + dy * 2 + dx) so that a pooling window is the
//         four registers r & 3 of one lane; the strip of a group becomes two strips (one per window row dy) STRIP rows
//         apart, STRIP = 136 / 264 chosen (brute force) so that every 16-lane fragment read stays on 16 bank quads;
//       * epilogue straight from the accumulators: the W rows of a wave's two 32-column blocks are staged interleaved
//         (block j, column c <-> channel 2 c + j), so a lane holds two ADJACENT channels = one packed dword, 32 lanes =
//         one 128-B row segment; rows >= M are masked by lane.
//   stem_s2_f16_kernel      3x3 / stride 2 conv on the NCHW image (fp32 or fp16; rounded to fp16 like the reference's cast),
//                           folded BatchNorm + ReLU, fp16 NHWC out
//   avgpool2_f16_kernel     AvgPool2d(2) on fp16 NHWC (the downsample branches; fp32 sum in (dy, dx) order)
//
// Bounds: conv3x3 MFMA (2500 TFLOP/s dense fp16); stem / pool HBM.
#include <stdlib.h>
#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOR = 0x80000000u;
constexpr long long EXT_LIM = 0x7FFFFFF0LL;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t desc(const void* base, long long total, long long shift) {
    long long ext = total - shift;
    ext = ext < 0 ? 0 : (ext > EXT_LIM ? EXT_LIM : ext);
    return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)base + shift), 0, (int)ext, 0x00020000);
}
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t r, unsigned char* lds_dst, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
#else
    (void)r; (void)lds_dst; (void)voff; (void)soff;
#endif
}
__device__ __forceinline__ unsigned pack2(float a, float b) {
    const f16x2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(unsigned, v);
}

struct ConvHP {
    const u16* x; const u16* w; const float* scale; const float* bias; u16* y;
    long long x_total, w_total, y_total;        // bytes
    int B, H, W, Cin, N, M;                     // M = B * H * W conv output pixels
    int tiles_n, n_tiles;
    const u16* res; int act;                    // 1x1 kernel: residual [M][N] added before the activation; act 0 | 1 (ReLU)
    // 1x1 kernel, TWO = 1 (conv3 + downsample branch of a stage's first block as one launch): a second operand pair x2 [M][Cin2], w2 [N][Cin2]
    // whose chunks run first; then the accumulators are multiplied per output channel by ratio[n] and the main pair continues
    const u16* x2; const u16* w2; const float* ratio; long long x2_total, w2_total; int Cin2;
};

// standard-order pixel of the corner of pooled pixel mp (2x2 windows of an H x W map)
__device__ __forceinline__ int pool_corner(const ConvHP& p, int mp) {
    const int wp2 = p.W >> 1, hwp = (p.H >> 1) * wp2;
    const int n = mp / hwp, rem = mp - n * hwp, hp = rem / wp2;
    return (n * p.H + 2 * hp) * p.W + 2 * (rem - hp * wp2);
}

template <int BM> struct StripGeo { static constexpr int STRIP = BM == 256 ? 136 : 264; };

template <int WM, int WN, int TN, int POOL>
__global__ __launch_bounds__(256, 2) void conv3x3_f16_kernel(const ConvHP p) {
    static_assert(WM * WN == 4 && (TN == 1 || TN == 2), "4 waves; a wave tile is 128 rows x 32 TN columns");
    constexpr int TM = 4, BM = WM * 128, BN = WN * TN * 32;
    constexpr int STRIP = StripGeo<BM>::STRIP;                   // POOL: LDS row of the dy = 1 strip
    constexpr int NWIN = BM / 4;                                 // POOL: windows per tile; a strip holds 2 NWIN + 2 pixels
    constexpr int NROWS = POOL ? 2 * STRIP : BM + 2;             // LDS rows (64 B: 32 channels) of an A stage
    constexpr int NIA = (NROWS + 15) / 16, NA = (NIA + 3) / 4;   // wave-DMA instructions per A stage: all / per wave
    constexpr int A_BYTES = NIA * 1024;
    constexpr int NWT = BN / 16, NWI = (NWT + 3) / 4;            // wave-DMA instructions per W tap block: all / per wave
    constexpr int W_BYTES = BN * 64;
    constexpr int NWS = BM == 512 ? 3 : 4, D = NWS - 1;          // W ring slots, staging distance in taps
    constexpr int ZOFF = 2 * A_BYTES, DUMP = ZOFF + 256, WOFF = DUMP + 1024;
    constexpr int LDS_BYTES = WOFF + NWS * W_BYTES;
    // vmcnt bookkeeping (per wave, in issue order).  Step t issues W(t + D), then -- on a group's first tap -- A(g + 1).
    // At the top of step t everything up to W(t) (and A(g) on a group's first tap) must have landed; what may stay in flight:
    constexpr int ALLOW0 = (D - 1) * NWI, ALLOW12 = (D - 1) * NWI + NA;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave / WN, wn = wave % WN;
    const int fr = lane & 31, fh = lane >> 5;
    const int tile = xcd_remap(blockIdx.x, p.n_tiles);
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;

    // ---- descriptors rebased to the first pixel / channel the tile touches (tensors may exceed 2 GiB) --------------
    const int pxf = POOL ? pool_corner(p, m0 >> 2) : m0;
    const int px0 = pxf - 1 - p.W > 0 ? pxf - 1 - p.W : 0;
    const __amdgpu_buffer_rsrc_t rsA = desc(p.x, p.x_total, (long long)px0 * p.Cin * 2);
    const __amdgpu_buffer_rsrc_t rsW = desc(p.w, p.w_total, (long long)n0 * (9 * p.Cin) * 2);
    // (stages past the end of K are issued too -- every wave issues the same number of DMA instructions, the waits are counted --
    //  and fetch nothing: out of range BY LANE, voffset = OOR with a zero scalar offset on the real descriptor.  A zero-extent
    //  descriptor with a non-zero scalar offset is not relied on: the range check is offset >= num_records - soffset.)

    // ---- stager: wave-DMA instruction k of a block covers LDS rows 16 k .. 16 k + 15; lane -> (row 16 k + lane / 4, 16-B
    //      slot lane % 4), which receives SOURCE chunk slot ^ swz(row).  Instructions k = wave + 4 i; the ones past the
    //      block's end (every wave issues the same number: the waits are counted) fetch nothing and land in a dump KB.
    unsigned fa_off[NA];        // byte offset of the group (slab 0, kh = 1) pixel from px0, + chunk; OOR = no such row
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int k = wave + 4 * i, row = 16 * k + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
        int pix = 0;
        bool have = false;                                       // a row the stage uses (its pixel may still be < 0 or >= M: zeros)
        if (k < NIA) {
            if constexpr (POOL) {
                const int dy = row >= STRIP, cc = row - dy * STRIP;                  // strip column: 0 / 2 NWIN + 1 = the halo pixels
                if (cc < 2 * NWIN + 2) {
                    const int wl = cc == 0 ? 0 : (cc == 2 * NWIN + 1 ? NWIN - 1 : (cc - 1) >> 1);
                    const int dxo = cc == 0 ? -1 : (cc == 2 * NWIN + 1 ? 2 : (cc - 1) & 1);
                    const int mp = (m0 >> 2) + wl;
                    if (4 * mp < p.M) { pix = pool_corner(p, mp) + dy * p.W + dxo; have = true; }
                }
            } else if (row < BM + 2) {
                pix = m0 - 1 + row; have = true;
            }
        }
        // (a pixel index below px0 -- only pixel -1 of the first tile -- wraps past the extent: zeros; such a row only feeds masked taps)
        fa_off[i] = have ? (unsigned)((pix - px0) * p.Cin) * 2u + c * 16u : OOR;
    }
    unsigned fw_off[NWI];
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
        const int k = wave + 4 * i, row = 16 * k + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
        const int wq = row / (TN * 32), within = row % (TN * 32), j = within >> 5, cc = within & 31;
        const int nrel = wq * TN * 32 + (TN == 2 ? 2 * cc + j : cc);              // block j, column cc <-> channel 2 cc + j
        fw_off[i] = (k < NWT && n0 + nrel < p.N) ? (unsigned)nrel * (unsigned)(9 * p.Cin * 2) + c * 16u : OOR;
    }
    auto issue_a = [&](int g, int stage, bool valid) {          // group g = (slab, kh): pixel shift (kh - 1) * W, channels slab * 32 ..
        const int slab = g / 3, kh = g - slab * 3;
        const unsigned delta = (unsigned)(((kh - 1) * p.W * p.Cin + slab * 32) * 2);
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int k = wave + 4 * i;
            unsigned char* dst = k < NIA ? lds + stage * A_BYTES + k * 1024 : lds + DUMP;
            glds16(rsA, dst, (!valid || fa_off[i] == OOR) ? OOR : fa_off[i] + delta, 0u);
        }
    };
    auto issue_w = [&](int t, int slot, bool valid) {           // tap step t: K columns [32 t, 32 t + 32)
#pragma unroll
        for (int i = 0; i < NWI; ++i) {
            const int k = wave + 4 * i;
            unsigned char* dst = k < NWT ? lds + WOFF + slot * W_BYTES + k * 1024 : lds + DUMP;
            glds16(rsW, dst, valid ? fw_off[i] : OOR, valid ? (unsigned)t * 64u : 0u);
        }
    };

    // ---- fragment addresses (bytes from the stage / slot base) and the tap-validity masks of this lane's four A rows ----
    int faddr[TM][3];           // row (lrow + kw), k-step 0; k-step 1 = ^ 32
    unsigned fmask[TM];         // bit kh * 3 + kw
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * 128 + i * 32 + fr, m = m0 + r;
        unsigned msk = 0;
        if (m < p.M) {
            int ho, wo;
            if constexpr (POOL) {
                const int q = m & 3, mp = m >> 2, wp2 = p.W >> 1, hwp = (p.H >> 1) * wp2;
                const int rem = mp % hwp, hp = rem / wp2;
                ho = 2 * hp + (q >> 1); wo = 2 * (rem - hp * wp2) + (q & 1);
            } else {
                const int hw = p.H * p.W, rem = m % hw;
                ho = rem / p.W; wo = rem - ho * p.W;
            }
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
                    if (ho + kh - 1 >= 0 && ho + kh - 1 < p.H && wo + kw - 1 >= 0 && wo + kw - 1 < p.W) msk |= 1u << (kh * 3 + kw);
        }
        fmask[i] = msk;
        const int lrow = POOL ? ((r >> 1) & 1) * STRIP + (r >> 2) * 2 + (r & 1) : r;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) faddr[i][kw] = (lrow + kw) * 64 + ((fh ^ (((lrow + kw) >> 2) & 3)) << 4);
    }
    int waddr[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int r = wn * TN * 32 + j * 32 + fr;
        waddr[j] = r * 64 + ((fh ^ ((r >> 2) & 3)) << 4);
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (tid < 16) *(u32x4*)(lds + ZOFF + tid * 16) = (u32x4){0u, 0u, 0u, 0u};      // the zero line (never staged over)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // ... written before this wave reaches the first barrier

    const int G = (p.Cin >> 5) * 3, T = 3 * G;
    issue_a(0, 0, true);
#pragma unroll
    for (int t = 0; t < D; ++t) issue_w(t, t, t < T);

    int wslot = 0;                                               // slot of the current tap = t % NWS
    u32x4 afc[2][TM], afn[2][TM];                                // A fragments of the current / next tap, both k-steps
    for (int g = 0; g < G; ++g) {
        const int kh = g % 3, astage = g & 1;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int t = 3 * g + kw;
            if (kw == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ALLOW0) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ALLOW12) : "memory");
            __builtin_amdgcn_s_barrier();                        // every wave's share of W(t) / A(g) has landed; step t - 1 is read out
            {
                int ns = wslot + D; ns = ns >= NWS ? ns - NWS : ns;
                issue_w(t + D, ns, t + D < T);
            }
            if (kw == 0) issue_a(g + 1, astage ^ 1, g + 1 < G);
            const unsigned char* Wb = lds + WOFF + wslot * W_BYTES;
            const int tap = kh * 3 + kw;
            // A fragments of taps kw = 1, 2 were read during the previous tap's MFMAs (same A stage, no barrier in between);
            // tap 0 of a group reads them now -- its strip was only guaranteed by the barrier above
            auto read_a = [&](int kwx, u32x4 (&dst)[2][TM]) {
                const int tp = kh * 3 + kwx;
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const bool ok = (fmask[i] >> tp) & 1u;
                    const int ao = ok ? astage * A_BYTES + faddr[i][kwx] : ZOFF + (faddr[i][kwx] & 255);
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) dst[ks][i] = *(const u32x4*)(lds + (ao ^ (ks * 32)));
                }
            };
            if (kw == 0) read_a(0, afc);
            u32x4 wf[2][TN];
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < TN; ++j) wf[ks][j] = *(const u32x4*)(Wb + (waddr[j] ^ (ks * 32)));
            if (kw < 2) read_a(kw + 1, afn);
            (void)tap;                                            // (s_setprio(1) around the MFMAs below measured 3 - 15 % slower)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, afc[ks][i]), __builtin_bit_cast(f16x8, wf[ks][j]),
                                                                           acc[i][j], 0, 0, 0);
            if (kw < 2) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < TM; ++i) afc[ks][i] = afn[ks][i];
            }
            wslot = wslot + 1 == NWS ? 0 : wslot + 1;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // the (zero-extent) tail DMAs

    // ---- epilogue: BatchNorm scale / bias, ReLU, (2x2 average), packed fp16 stores from the accumulator layout -----------
    float sv[TN], bv[TN];
    const int ncol = n0 + wn * TN * 32 + (TN == 2 ? 2 * fr : fr);        // this lane's first channel
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = ncol + j;
        sv[j] = (p.scale && n < p.N) ? p.scale[n] : 1.f;
        bv[j] = (p.bias && n < p.N) ? p.bias[n] : 0.f;
    }
    const bool col_ok = ncol < p.N;
    if constexpr (POOL) {
        // register group t = r >> 2 of row block i = window (wm * 32 + i * 8 + 2 t + fh) of the tile; its four rows are r & 3
        const __amdgpu_buffer_rsrc_t rsY = desc(p.y, p.y_total, (long long)(m0 >> 2) * p.N * 2);
        const int wins_left = (p.M >> 2) - (m0 >> 2);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int wl = wm * 32 + i * 8 + 2 * t + fh;
                const unsigned voff = (col_ok && wl < wins_left) ? (unsigned)(wl * p.N + ncol) * 2u : OOR;
                float s[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaxf(fmaf(acc[i][j][4 * t + q], sv[j], bv[j]), 0.f);
                    s[j] = (((v[0] + v[1]) + v[2]) + v[3]) * 0.25f;                  // (dy, dx) order, like the pool kernel
                }
                if constexpr (TN == 2) __builtin_amdgcn_raw_buffer_store_b32(pack2(s[0], s[1]), rsY, voff, 0u, 0);
                else __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (_Float16)s[0]), rsY, voff, 0u, 0);
            }
    } else {
        const __amdgpu_buffer_rsrc_t rsY = desc(p.y, p.y_total, (long long)m0 * p.N * 2);
        const int rows_left = p.M - m0;
        const int row_lim = (rows_left < BM ? rows_left : BM) - (wm * 128 + 4 * fh);   // row u of this lane is valid iff u < row_lim
        const unsigned vbase = col_ok ? (unsigned)((wm * 128 + 4 * fh) * p.N + ncol) * 2u : OOR;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ru = i * 32 + (r & 3) + 8 * (r >> 2);
                const unsigned voff = ru < row_lim ? vbase : OOR;                    // rows >= M: masked by lane
                const float v0 = fmaxf(fmaf(acc[i][0][r], sv[0], bv[0]), 0.f);
                if constexpr (TN == 2) {
                    const float v1 = fmaxf(fmaf(acc[i][1][r], sv[1], bv[1]), 0.f);
                    __builtin_amdgcn_raw_buffer_store_b32(pack2(v0, v1), rsY, voff, (unsigned)(ru * p.N * 2), 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (_Float16)v0), rsY, voff, (unsigned)(ru * p.N * 2), 0);
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 1x1 conv + BatchNorm (+ residual) (+ ReLU) for the HBM-bound shapes (short K or narrow N: layers 1 - 2 of an RN50, every
// downsample conv): the same machinery with one tap per 32-channel slab and nothing to mask.  A ring of THREE slots, each
// a K chunk of both operands (BM x 32 of A, BN x 32 of W: 20 - 24 KB), staged two chunks ahead by LDS-DMA: two
// workgroups per CU keep ~100 KB in flight per CU, which is what it takes to stream at the HBM rate with a ~2 us round trip
// (the register-staged 128 x 128 GEMM holds 3.3 TB/s on these shapes).  Wave tile 32 TM x 64 (TM = 4: 256 x 128 tiles;
// TM = 2: 256 x 64 for Cout <= 64).  Epilogue from the accumulator layout; the residual is loaded in the same shape
// (one packed dword = two channels per lane, 128-B row segments), a row block's 16 loads in flight together.
// ---------------------------------------------------------------------------------------------------------------
template <int WM, int WN, int TM, int RES, int TWO = 0>
__global__ __launch_bounds__(256, 2) void conv1x1_f16_kernel(const ConvHP p) {
    static_assert(WM * WN == 4 && (TM == 2 || TM == 4), "4 waves; a wave tile is 32 TM rows x 64 columns");
    constexpr int TN = 2, BM = WM * TM * 32, BN = WN * 64;
    constexpr int NA = BM / 64;                                  // wave-DMA instructions per wave and A chunk (16 rows each)
    constexpr int NWT = BN / 16, NWI = (NWT + 3) / 4;            // ... per W chunk: all / per wave
    constexpr int A_BYTES = BM * 64, W_BYTES = BN * 64, SLOT = A_BYTES + W_BYTES;
    constexpr int DUMP = 3 * SLOT, LDS_BYTES = DUMP + 1024;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave / WN, wn = wave % WN;
    const int fr = lane & 31, fh = lane >> 5;
    const int tile = xcd_remap(blockIdx.x, p.n_tiles);
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
    const __amdgpu_buffer_rsrc_t rsA = desc(p.x, p.x_total, (long long)m0 * p.Cin * 2);
    const __amdgpu_buffer_rsrc_t rsW = desc(p.w, p.w_total, (long long)n0 * p.Cin * 2);
    const __amdgpu_buffer_rsrc_t rsA2 = TWO ? desc(p.x2, p.x2_total, (long long)m0 * p.Cin2 * 2) : rsA;
    const __amdgpu_buffer_rsrc_t rsW2 = TWO ? desc(p.w2, p.w2_total, (long long)n0 * p.Cin2 * 2) : rsW;
    const int G2 = TWO ? p.Cin2 >> 5 : 0;                        // chunks of the second pair (they run first)

    unsigned fa_off[NA], fw_off[NWI], fa_off2[TWO ? NA : 1], fw_off2[TWO ? NWI : 1];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int k = wave + 4 * i, row = 16 * k + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
        fa_off[i] = m0 + row < p.M ? (unsigned)row * (unsigned)(p.Cin * 2) + c * 16u : OOR;
        if constexpr (TWO) fa_off2[i] = m0 + row < p.M ? (unsigned)row * (unsigned)(p.Cin2 * 2) + c * 16u : OOR;
    }
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
        const int k = wave + 4 * i, row = 16 * k + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
        const int wq = row >> 6, within = row & 63, j = within >> 5, cc = within & 31;
        const int nrel = wq * 64 + 2 * cc + j;                                       // block j, column cc <-> channel 2 cc + j
        fw_off[i] = (k < NWT && n0 + nrel < p.N) ? (unsigned)nrel * (unsigned)(p.Cin * 2) + c * 16u : OOR;
        if constexpr (TWO) fw_off2[i] = (k < NWT && n0 + nrel < p.N) ? (unsigned)nrel * (unsigned)(p.Cin2 * 2) + c * 16u : OOR;
    }
    auto issue = [&](int t, int slot, bool valid) {              // K chunk t = channels [32 t, 32 t + 32) of both operands
        const bool second = TWO && t < G2;
        const unsigned so = valid ? (unsigned)(second ? t : t - G2) * 64u : 0u;      // a chunk past K: out of range by lane (voffset = OOR), zero scalar offset
#pragma unroll
        for (int i = 0; i < NA; ++i)
            glds16(second ? rsA2 : rsA, lds + slot * SLOT + (wave + 4 * i) * 1024, valid ? (second ? fa_off2[TWO ? i : 0] : fa_off[i]) : OOR, so);
#pragma unroll
        for (int i = 0; i < NWI; ++i) {
            const int k = wave + 4 * i;
            glds16(second ? rsW2 : rsW, k < NWT ? lds + slot * SLOT + A_BYTES + k * 1024 : lds + DUMP, valid ? (second ? fw_off2[TWO ? i : 0] : fw_off[i]) : OOR, so);
        }
    };
    int faddr[TM], waddr[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm * (TM * 32) + i * 32 + fr;
        faddr[i] = r * 64 + ((fh ^ ((r >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int r = wn * 64 + j * 32 + fr;
        waddr[j] = A_BYTES + r * 64 + ((fh ^ ((r >> 2) & 3)) << 4);
    }
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int G = G2 + (p.Cin >> 5);
    issue(0, 0, true);
    issue(1, 1, 1 < G);
    int slot = 0;
    for (int t = 0; t < G; ++t) {
        if (TWO && t == G2) {                                    // the second pair's sums -> the main pair's scale
            const int nc = n0 + wn * 64 + 2 * fr;
            const float r0 = nc < p.N ? p.ratio[nc] : 1.f, r1 = nc + 1 < p.N ? p.ratio[nc + 1] : 1.f;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[i][0][r] *= r0; acc[i][1][r] *= r1; }
        }
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NWI) : "memory");       // chunk t has landed; chunk t + 1 may be in flight
        __builtin_amdgcn_s_barrier();
        {
            int ns = slot + 2; ns = ns >= 3 ? ns - 3 : ns;                     // the slot chunk t - 1 was read from
            issue(t + 2, ns, t + 2 < G);
        }
        const unsigned char* Sb = lds + slot * SLOT;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 af[TM], wf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *(const u32x4*)(Sb + (faddr[i] ^ (ks * 32)));
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[j] = *(const u32x4*)(Sb + (waddr[j] ^ (ks * 32)));
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i]), __builtin_bit_cast(f16x8, wf[j]),
                                                                       acc[i][j], 0, 0, 0);
        }
        slot = slot + 1 == 3 ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue ----------------------------------------------------------------------------------------------------
    const int ncol = n0 + wn * 64 + 2 * fr;
    const bool col_ok = ncol < p.N;
    float sv[2], bv[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        sv[j] = (p.scale && ncol + j < p.N) ? p.scale[ncol + j] : 1.f;
        bv[j] = (p.bias && ncol + j < p.N) ? p.bias[ncol + j] : 0.f;
    }
    const __amdgpu_buffer_rsrc_t rsY = desc(p.y, p.y_total, (long long)m0 * p.N * 2);
    const __amdgpu_buffer_rsrc_t rsR = RES ? desc(p.res, p.y_total, (long long)m0 * p.N * 2) : rsY;
    const int rows_left = p.M - m0;
    const int row_lim = (rows_left < BM ? rows_left : BM) - (wm * (TM * 32) + 4 * fh);
    const unsigned vbase = col_ok ? (unsigned)((wm * (TM * 32) + 4 * fh) * p.N + ncol) * 2u : OOR;
    const bool relu = p.act == DBMM_ACT_RELU;
    unsigned rv[RES ? 2 : 1][16];
    auto load_res = [&](int i, int set) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ru = i * 32 + (r & 3) + 8 * (r >> 2);
            rv[set][r] = __builtin_amdgcn_raw_buffer_load_b32(rsR, ru < row_lim ? vbase : OOR, (unsigned)(ru * p.N * 2), 0);
        }
    };
    if constexpr (RES) load_res(0, 0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        if constexpr (RES) { if (i + 1 < TM) load_res(i + 1, (i + 1) & 1); }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ru = i * 32 + (r & 3) + 8 * (r >> 2);
            float v0 = fmaf(acc[i][0][r], sv[0], bv[0]), v1 = fmaf(acc[i][1][r], sv[1], bv[1]);
            if constexpr (RES) {
                const f16x2 rh = __builtin_bit_cast(f16x2, rv[i & 1][r]);
                v0 += (float)rh[0]; v1 += (float)rh[1];
            }
            if (relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
            __builtin_amdgcn_raw_buffer_store_b32(pack2(v0, v1), rsY, ru < row_lim ? vbase : OOR, (unsigned)(ru * p.N * 2), 0);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// stem conv1: 3x3, stride 2, pad 1, Cin = 3, NCHW image -> fp16 NHWC, folded BatchNorm + ReLU.  One thread per output pixel
// (27 taps in registers, weights broadcast from LDS, 8 output channels at a time); the workgroup's 256 pixels leave
// through an LDS staging tile as one contiguous run of 16-B lane stores.  The image is rounded to fp16 first, as the
// reference's `x.type(self.conv1.weight.dtype)` does (clip/model.py:146).
// ---------------------------------------------------------------------------------------------------------------
template <int COUT, typename TIN>
__global__ __launch_bounds__(256) void stem_s2_f16_kernel(const TIN* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
                                                          const float* __restrict__ bias, u16* __restrict__ y, int B, int H, int W, int Ho, int Wo) {
    constexpr int PITCH = COUT + 8;                             // staging row pitch in halves (16-B aligned rows, shifted banks)
    __shared__ __attribute__((aligned(16))) float sw[29 * COUT];
    __shared__ __attribute__((aligned(16))) u16 stage[256 * PITCH];
    for (int i = threadIdx.x; i < 27 * COUT; i += 256) sw[i] = w[i];
    for (int i = threadIdx.x; i < COUT; i += 256) { sw[27 * COUT + i] = bias ? bias[i] : 0.f; sw[28 * COUT + i] = scale ? scale[i] : 1.f; }
    __syncthreads();
    const long long m0 = (long long)blockIdx.x * 256, m = m0 + threadIdx.x;
    const long long M = (long long)B * Ho * Wo;
    if (m < M) {
        const int wo = (int)(m % Wo), ho = (int)((m / Wo) % Ho), n = (int)(m / ((long long)Wo * Ho));
        float xin[27];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int hi = ho * 2 - 1 + kh, wi = wo * 2 - 1 + kw;
                const bool ok = hi >= 0 && hi < H && wi >= 0 && wi < W;
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    xin[(kh * 3 + kw) * 3 + c] = ok ? (float)(_Float16)(float)x[(((long long)n * 3 + c) * H + hi) * W + wi] : 0.f;
            }
        u16* so = stage + threadIdx.x * PITCH;
        for (int c0 = 0; c0 < COUT; c0 += 8) {
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
            for (int t = 0; t < 27; ++t) {
                const f32x4 w0 = *(const f32x4*)(sw + t * COUT + c0);
                const f32x4 w1 = *(const f32x4*)(sw + t * COUT + c0 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[j] = fmaf(xin[t], w0[j], acc[j]);
                    acc[4 + j] = fmaf(xin[t], w1[j], acc[4 + j]);
                }
            }
            float o[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = fmaxf(fmaf(acc[j], sw[28 * COUT + c0 + j], sw[27 * COUT + c0 + j]), 0.f);
            *(u32x4*)(so + c0) = (u32x4){pack2(o[0], o[1]), pack2(o[2], o[3]), pack2(o[4], o[5]), pack2(o[6], o[7])};
        }
    }
    __syncthreads();
    constexpr int QPP = COUT / 8;                               // 16-B chunks per pixel
    const long long run = (M - m0 < 256 ? M - m0 : 256) * QPP;
    u32x4* yo = (u32x4*)(y + m0 * COUT);
#pragma unroll 2
    for (int q = threadIdx.x; q < 256 * QPP; q += 256)
        if (q < run) yo[q] = *(const u32x4*)(stage + (q / QPP) * PITCH + (q % QPP) * 8);
}

// The same conv as an MFMA gather (resnet_ops.hip's stem_s2_mfma_kernel at one fp16 product): a wave owns blocks of 32 output pixels; lane
// (pixel fr, K half fh) gathers its 16 of the 32 K slots (27 taps + 5 zeros) straight from the NCHW image through a bounds-checked
// descriptor, rounds them to fp16 (the reference's cast) and two 32x32x16 MFMAs per 32 channels replace 864 FMAs per pixel; the weights
// are B fragments held in registers for the whole launch.  The block's 32 x Cout fp16 outputs go through a wave-private LDS tile and
// leave as one contiguous run of 16-B lane stores.  The next block's taps are in flight during the MFMAs and stores.
// The weights are rounded to fp16 here -- exact for a model whose conv1 is stored in fp16, which is what fp16 mode means -- so BatchNorm
// comes as a separate per-channel scale / bias on the fp32 accumulator (dbmm_conv_stem_s2_bn_f16), not folded into the weights.
// fp32 accumulation inside the MFMA instead of 27 sequential FMAs: another order of summation of the same exact products.
// Bound: HBM (0.6 MB in as fp32 + 0.8 MB out per image at 224 px).
template <int NB, typename TIN>                                       // NB = Cout / 32
__global__ __launch_bounds__(256, 3) void stem_s2_f16_mfma_kernel(const TIN* __restrict__ x, const float* __restrict__ w, const float* __restrict__ scale,
                                                                  const float* __restrict__ bias, u16* __restrict__ y, int B, int H, int W, int Ho, int Wo,
                                                                  int n_blocks) {
    constexpr int Cout = NB * 32, PITCH = Cout + 8, ES = (int)sizeof(TIN);
    __shared__ __attribute__((aligned(16))) u16 stage_all[4 * 32 * PITCH];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u16* stage = stage_all + wave * (32 * PITCH);
    const int fr = lane & 31, fh = lane >> 5;
    const long long M = (long long)B * Ho * Wo, img = 3LL * H * W;
    // K slot (fh, t = ks * 8 + i) -> tap: t < 9: (kw, c) = (t / 3, t % 3), kh = fh;  t >= 9: kh = 2, (kw, c) pair t - 9 (fh = 0) or t - 2 (fh = 1: t = 9, 10)
    auto slot_tap = [](int f, int t, int& kh, int& kw, int& c) -> bool {
        int pr;
        if (t < 9) { kh = f; pr = t; }
        else { kh = 2; pr = f ? t - 2 : t - 9; if (pr > 8) { kh = kw = c = 0; return false; } }
        kw = pr / 3; c = pr - kw * 3;
        return true;
    };
    unsigned wq[NB][2][4];
    float bv[NB], sv[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        float wv[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            int kh0, kw0, c0, kh1, kw1, c1;
            const bool v0 = slot_tap(0, t, kh0, kw0, c0), v1 = slot_tap(1, t, kh1, kw1, c1);
            const int k = fh ? (kh1 * 3 + kw1) * 3 + c1 : (kh0 * 3 + kw0) * 3 + c0;
            wv[t] = (fh ? v1 : v0) ? w[k * Cout + j * 32 + fr] : 0.f;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int q = 0; q < 4; ++q) wq[j][ks][q] = pack2(wv[ks * 8 + 2 * q], wv[ks * 8 + 2 * q + 1]);
        bv[j] = bias ? bias[j * 32 + fr] : 0.f;
        sv[j] = scale ? scale[j * 32 + fr] : 1.f;
    }
    const int n_waves = gridDim.x * 4;
    float xv[16];
    auto gather = [&](int blk) {
        const long long mb = (long long)blk * 32, m = mb + fr;
        const long long img0 = mb / ((long long)Ho * Wo);                 // image of the block's first pixel (descriptor base)
        const long long left = (long long)B - img0;
        const long long span = 31 / ((long long)Ho * Wo) + 2;             // images a block of 32 pixels can touch (two, unless an image has < 31 of them)
        const long long ext = (left < span ? left : span) * img * ES;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(x + img0 * img), 0, (int)ext, 0x00020000);
        unsigned base = OOR;
        int mask = 0;
        if (m < M) {
            const int wo = (int)(m % Wo), ho = (int)((m / Wo) % Ho);
            const int n = (int)(m / ((long long)Wo * Ho) - img0);
            const int hi0 = 2 * ho - 1, wi0 = 2 * wo - 1;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                if (hi0 + d >= 0 && hi0 + d < H) mask |= 1 << d;
                if (wi0 + d >= 0 && wi0 + d < W) mask |= 8 << d;
            }
            base = (unsigned)((long long)n * img + (long long)hi0 * W + wi0);      // may wrap below zero: only used with valid taps
        }
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            int kh0, kw0, c0, kh1, kw1, c1;
            const bool v0 = slot_tap(0, t, kh0, kw0, c0), v1 = slot_tap(1, t, kh1, kw1, c1);
            const int off0 = (c0 * H + kh0) * W + kw0, off1 = (c1 * H + kh1) * W + kw1;          // uniform
            const int bit0 = v0 ? (1 << kh0) | (8 << kw0) : 64, bit1 = v1 ? (1 << kh1) | (8 << kw1) : 64;   // bit 6 is never in a mask
            const int bit = fh ? bit1 : bit0;
            const bool ok = (mask & bit) == bit;
            const unsigned off = ok ? (base + (unsigned)(fh ? off1 : off0)) * (unsigned)ES : OOR;
            if constexpr (ES == 4) xv[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
            else xv[t] = (float)__builtin_bit_cast(_Float16, __builtin_amdgcn_raw_buffer_load_b16(rs, off, 0, 0));
        }
    };
    int blk = blockIdx.x * 4 + wave;
    if (blk < n_blocks) gather(blk);
    for (; blk < n_blocks; blk += n_waves) {
        unsigned a[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int q = 0; q < 4; ++q) a[ks][q] = pack2(xv[ks * 8 + 2 * q], xv[ks * 8 + 2 * q + 1]);     // the image rounded to fp16
        const long long mb = (long long)blk * 32;
        if (blk + n_waves < n_blocks) gather(blk + n_waves);          // in flight during the MFMAs and stores
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, (u32x4){a[ks][0], a[ks][1], a[ks][2], a[ks][3]}),
                                                             __builtin_bit_cast(f16x8, (u32x4){wq[j][ks][0], wq[j][ks][1], wq[j][ks][2], wq[j][ks][3]}), acc, 0, 0, 0);
            // accumulator register r = pixel row (r & 3) + 8 (r >> 2) + 4 fh, column = channel j * 32 + fr
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * fh;
                const _Float16 hv = (_Float16)fmaxf(fmaf(acc[r], sv[j], bv[j]), 0.f);
                stage[row * PITCH + j * 32 + fr] = __builtin_bit_cast(u16, hv);
            }
        }
        // the block's 32 x Cout outputs are one contiguous run: 16-B chunks lane by lane (rows past M are cut by the descriptor)
        const long long rows_left = M - mb;
        const long long bytes = (rows_left < 32 ? rows_left : 32) * Cout * 2;
        const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(y + mb * Cout), 0, (int)bytes, 0x00020000);
        constexpr int QPP = Cout / 8;
#pragma unroll
        for (int i = 0; i < 32 * QPP / 64; ++i) {
            const int c = lane + 64 * i;
            const u32x4 v = *(const u32x4*)(stage + (c / QPP) * PITCH + (c % QPP) * 8);
            __builtin_amdgcn_raw_buffer_store_b128(v, ry, (unsigned)c * 16u, 0, 0);
        }
    }
}

// AvgPool2d(2) on fp16 NHWC: a thread owns 8 channels of one output pixel
__global__ __launch_bounds__(256) void avgpool2_f16_kernel(const u16* __restrict__ x, u16* __restrict__ y, int H, int W, int C, long long n_out) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;       // chunk index: output pixel * (C / 8) + chunk
    if (q >= n_out) return;
    const int cpp = C >> 3, Wo = W >> 1, Ho = H >> 1;
    const long long op = q / cpp;
    const int ch = (int)(q - op * cpp) * 8, wo = (int)(op % Wo), ho = (int)((op / Wo) % Ho);
    const long long n = op / ((long long)Wo * Ho);
    const u16* base = x + (((n * H + 2 * ho) * W + 2 * wo) * (long long)C + ch);
    float s[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = 0.f;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const f16x8 v = __builtin_bit_cast(f16x8, *(const u32x4*)(base + ((long long)dy * W + dx) * C));
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] += (float)v[e];
        }
    *(u32x4*)(y + op * C + ch) = (u32x4){pack2(s[0] * 0.25f, s[1] * 0.25f), pack2(s[2] * 0.25f, s[3] * 0.25f),
                                         pack2(s[4] * 0.25f, s[5] * 0.25f), pack2(s[6] * 0.25f, s[7] * 0.25f)};
}

// ---------------------------------------------------------------------------------------------------------------
// The 3x3 conv for Cout % 256 == 0 (layers 3 / 4) on the eight-phase structure of gemm_f16_8ph_kernel / conv3x3_halo8_kernel: 512 threads =
// 4 x 2 waves of 64 x 128 outputs, tile 256 pixels x 256 channels, two wave groups a barrier apart, persistent workgroups.  A tap (one
// 32-channel K tile) is TWO phases of 8 MFMAs -- q0: both row blocks x the first 64 of the wave's 128 columns, q1: x the second 64 -- with
// six fragment reads each (q0: four W fragments + row block 1 of the tap; q1: four W fragments + row block 0 of the NEXT tap).  Everything
// reaches LDS by LDS-DMA: the tap's W tile (256 x 32 fp16 = 16 KB) as two half-tiles into a ring of three (first half issued at q1 three taps
// ahead, second at the following q0), the activation strip of a (32-channel slab, kh) group (258 pixels; POOL: two strips of 130) into a
// ring of three, issued at the first phase of the group two groups ahead; waits are counted.  A loop trip = three groups = one slab: W
// buffer = kw, strip buffer = kh, mask bit = 3 kh + kw are compile-time.  Same arithmetic and order of accumulation as conv3x3_f16_kernel.
// ---------------------------------------------------------------------------------------------------------------
constexpr int E_AROWS = 384, E_ABUF = E_AROWS * 64;              // strip rows a group's three DMA instructions per thread cover; bytes
constexpr int E_WBUF = 256 * 64, E_WHALF = 128 * 64;
constexpr int E_OFF_A = 3 * E_WBUF;                               // W ring first: fragment reads reach buffers / blocks through ds_read immediates
constexpr int E_LDS = E_OFF_A + 3 * E_ABUF;                       // 49,152 + 73,728 = 122,880
constexpr int E_STRIP1 = 136, E_ZROW = 268;                       // POOL: row of the dy = 1 strip; the zero line (rows 268..271, never a strip row)

template <int POOL>
__global__ __launch_bounds__(512, 1) void conv3x3_f16_8ph_kernel(const ConvHP p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[E_LDS];
    constexpr int NLD = POOL ? 2 * E_STRIP1 : 258;                // strip rows in use (POOL: rows 130..135 of the first strip's block stay unused)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wid >> 1, wc = wid & 1, grp = wid >> 2;
    const int fr = lane & 31, fh = lane >> 5;
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot_in_xcd = blockIdx.x >> 3, wg_per_xcd = (nwg - xcd + 7) >> 3;
    const int tq = p.n_tiles >> 3, trm = p.n_tiles & 7;
    const int t_lo = xcd < trm ? xcd * (tq + 1) : trm * (tq + 1) + (xcd - trm) * tq, t_hi = t_lo + tq + (xcd < trm ? 1 : 0);
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, 0, 0x00020000), rsW = rsA;
    int m0 = 0, n0 = 0;
    const int G = 3 * (p.Cin >> 5);

    // W half-tile h by LDS-DMA: thread -> LDS row rl = tid >> 2 of the half (wave column rl >> 6, block j = (rl >> 5) & 1, column cc = rl & 31 <->
    // channel 2 cc + j of the wave's h-th 64: a lane's two accumulator blocks of a half are ADJACENT channels), slot tid & 3
    unsigned voffW[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int rl = tid >> 2, nrel = (rl >> 6) * 128 + h * 64 + 2 * (rl & 31) + ((rl >> 5) & 1);
        voffW[h] = (unsigned)nrel * (unsigned)(9 * p.Cin * 2) + (((tid & 3) ^ ((rl >> 2) & 3)) << 4);
    }
    // fragment addresses: A row block 0 at tap kw (k-step 1 = ^ 32, row block 1 = + A_BLK, buffer = + E_ABUF); W block 0 of a half (+ 2048: block 1)
    constexpr int A_BLK = POOL ? 1024 : 2048;
    int faddr[3], boff;
    {
        const int r = wr * 64 + fr;
        const int lrow = POOL ? ((r >> 1) & 1) * E_STRIP1 + (r >> 2) * 2 + (r & 1) : r;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) faddr[kw] = E_OFF_A + (lrow + kw) * 64 + ((fh ^ (((lrow + kw) >> 2) & 3)) << 4);
        const int rw = wc * 64 + fr;
        boff = rw * 64 + ((fh ^ ((rw >> 2) & 3)) << 4);
    }
    unsigned fa_off[3], fmask[2];
    auto set_tile = [&](int tile) {
        m0 = (tile / p.tiles_n) * 256; n0 = (tile % p.tiles_n) * 256;
        const int pxf = POOL ? pool_corner(p, m0 >> 2) : m0;
        const int px0 = pxf - 1 - p.W > 0 ? pxf - 1 - p.W : 0;
        rsA = desc(p.x, p.x_total, (long long)px0 * p.Cin * 2);
        rsW = desc(p.w, p.w_total, (long long)n0 * (9 * p.Cin) * 2);
#pragma unroll
        for (int i = 0; i < 3; ++i) {                             // DMA instruction k = wid + 8 i covers strip rows 16 k .. 16 k + 15
            const int row = 16 * (wid + 8 * i) + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
            int pix = 0;
            bool have = false;
            if constexpr (POOL) {
                const int dy = row >= E_STRIP1, cc = row - dy * E_STRIP1;
                if (row < NLD && cc < 130) {
                    const int wl = cc == 0 ? 0 : (cc == 129 ? 63 : (cc - 1) >> 1), dxo = cc == 0 ? -1 : (cc == 129 ? 2 : (cc - 1) & 1);
                    const int mp = (m0 >> 2) + wl;
                    if (4 * mp < p.M) { pix = pool_corner(p, mp) + dy * p.W + dxo; have = true; }
                }
            } else if (row < NLD) {
                pix = m0 - 1 + row; have = true;
            }
            // offset of the row's kh = 0 pixel from the descriptor base; "negative" pixels wrap past the extent (zeros) until a group's shift brings them back
            fa_off[i] = have ? (unsigned)((pix - p.W - px0) * p.Cin) * 2u + c * 16u : OOR;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = m0 + wr * 64 + i * 32 + fr;
            unsigned msk = 0;
            if (m < p.M) {
                int ho, wo;
                if constexpr (POOL) {
                    const int q = m & 3, mp = m >> 2, wp2 = p.W >> 1, hwp = (p.H >> 1) * wp2;
                    const int rem = mp % hwp, hp = rem / wp2;
                    ho = 2 * hp + (q >> 1); wo = 2 * (rem - hp * wp2) + (q & 1);
                } else {
                    const int hw = p.H * p.W, rem = m % hw;
                    ho = rem / p.W; wo = rem - ho * p.W;
                }
#pragma unroll
                for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 3; ++kw)
                        if (ho + kh - 1 >= 0 && ho + kh - 1 < p.H && wo + kw - 1 >= 0 && wo + kw - 1 < p.W) msk |= 1u << (kh * 3 + kw);
            }
            fmask[i] = msk;
        }
    };
    auto dma_a = [&](int g, int buf) {                            // group g = (slab, kh) -> strip buffer `buf` (groups past the last: never multiplied)
        const int slab = g / 3, kh = g - 3 * slab;
        const unsigned delta = (unsigned)((kh * p.W * p.Cin + slab * 32) * 2);
#pragma unroll
        for (int i = 0; i < 3; ++i) glds16(rsA, lds + E_OFF_A + buf * E_ABUF + (wid + 8 * i) * 1024, fa_off[i] + delta, 0u);
    };
    auto dma_w = [&](int h, int t, int buf) { glds16(rsW, lds + buf * E_WBUF + h * E_WHALF + wid * 1024, voffW[h], (unsigned)t * 64u); };

    f32x16 acc[2][4];                                            // [row block][half * 2 + block]
    u32x4 fa0[2], fa0n[2], fa1[2], fb[2][2];                     // A: [ks]; W: [block][ks] of the half in use
    auto a_addr = [&](int i, int buf, int kw, int tapbit) -> int {
        const bool ok = (fmask[i] >> tapbit) & 1u;
        const int f = faddr[kw];
        return ok ? f + buf * E_ABUF + i * A_BLK : (E_OFF_A + buf * E_ABUF + E_ZROW * 64 + (f & 255));
    };
    // One phase.  j = phase within the trip (static, 0..17): group gi = j / 6 (= kh), tap kw = (j % 6) / 2, q = j & 1; K tile t = t0 + 3 gi + kw.
    //   q0: reads W half 0 + row block 1;  DMA of W half 1 two taps on;  first tap of a group: the strip of the group two on;  vmcnt retires half 1 of THIS tap
    //   q1: reads W half 1 + row block 0 of the next tap;  DMA of W half 0 three taps on;  vmcnt retires half 0 of the NEXT tap
    // (instructions issued after the DMA being retired: four half-tiles, + the three strip instructions unless they are older: kw = 2)
    auto phase = [&](int j, int t0, int g0) {
        const int gi = j / 6, kw = (j - 6 * gi) >> 1, q = j & 1, t = t0 + 3 * gi + kw;
        const unsigned char* wb = lds + kw * E_WBUF + q * E_WHALF;
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fb[b][ks] = *(const u32x4*)(wb + b * 2048 + (ks ? boff ^ 32 : boff));
        if (q == 0) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fa0[ks] = fa0n[ks];
            const int a1 = a_addr(1, gi, kw, 3 * gi + kw);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fa1[ks] = *(const u32x4*)(lds + (ks ? a1 ^ 32 : a1));
            __builtin_amdgcn_sched_barrier(0);
            if (kw == 0) dma_a(g0 + gi + 2, (gi + 2) % 3);
            dma_w(1, t + 2, (kw + 2) % 3);
        } else {
            const int a0 = kw < 2 ? a_addr(0, gi, kw + 1, 3 * gi + kw + 1) : a_addr(0, (gi + 1) % 3, 0, 3 * ((gi + 1) % 3));
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fa0n[ks] = *(const u32x4*)(lds + (ks ? a0 ^ 32 : a0));
            __builtin_amdgcn_sched_barrier(0);
            dma_w(0, t + 3, kw);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kw == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[i][2 * q + b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, i ? fa1[ks] : fa0[ks]), __builtin_bit_cast(f16x8, fb[b][ks]),
                                                                                acc[i][2 * q + b], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };

    // the zero line of the three strip buffers (the strip DMAs' out-of-range rows may or may not rewrite it -- with zeros)
    if (tid < 48) *(u32x4*)(lds + E_OFF_A + (tid >> 4) * E_ABUF + E_ZROW * 64 + (tid & 15) * 16) = (u32x4){0u, 0u, 0u, 0u};
    __syncthreads();

    const int n_items = t_lo + slot_in_xcd < t_hi ? (t_hi - t_lo - slot_in_xcd + wg_per_xcd - 1) / wg_per_xcd : 0;
    // what a tile needs before its first phases: both halves of taps 0 and 1, half 0 of tap 2; the strips of groups 0 and 1
    auto prologue = [&]() {
        dma_w(0, 0, 0); dma_w(1, 0, 0); dma_w(0, 1, 1); dma_w(1, 1, 1); dma_w(0, 2, 2);
        dma_a(0, 0); dma_a(1, 1);
    };
    if (n_items > 0) { set_tile(t_lo + slot_in_xcd); prologue(); }
    for (int k = 0; k < n_items; ++k) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the prologue (and the previous tile's stores)
        __builtin_amdgcn_s_barrier();
        {
            const int a0 = a_addr(0, 0, 0, 0);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fa0n[ks] = *(const u32x4*)(lds + (ks ? a0 ^ 32 : a0));
        }
        if (grp == 1) __builtin_amdgcn_s_barrier();               // the second wave group runs one barrier behind
        for (int g0 = 0; g0 < G; g0 += 3) {
#pragma unroll
            for (int j = 0; j < 18; ++j) phase(j, 3 * g0, g0);
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the look-ahead DMAs past the last tap
        __builtin_amdgcn_s_barrier();
        const int em0 = m0, en0 = n0;
        if (k + 1 < n_items) { set_tile(t_lo + slot_in_xcd + (k + 1) * wg_per_xcd); prologue(); }   // in flight during the epilogue below

        // epilogue: BatchNorm scale / bias, ReLU, (2x2 average), packed fp16 stores straight from the accumulators (as conv3x3_f16_kernel)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ncol = en0 + wc * 128 + h * 64 + 2 * fr;
            const float s0 = p.scale ? p.scale[ncol] : 1.f, s1 = p.scale ? p.scale[ncol + 1] : 1.f;
            const float b0 = p.bias ? p.bias[ncol] : 0.f, b1 = p.bias ? p.bias[ncol + 1] : 0.f;
            if constexpr (POOL) {
                const __amdgpu_buffer_rsrc_t rsY = desc(p.y, p.y_total, (long long)(em0 >> 2) * p.N * 2);
                const int wins_left = (p.M >> 2) - (em0 >> 2);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int t4 = 0; t4 < 4; ++t4) {
                        const int wl = wr * 16 + i * 8 + 2 * t4 + fh;
                        float v[2][4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            v[0][q] = fmaxf(fmaf(acc[i][2 * h][4 * t4 + q], s0, b0), 0.f);
                            v[1][q] = fmaxf(fmaf(acc[i][2 * h + 1][4 * t4 + q], s1, b1), 0.f);
                        }
                        const float o0 = (((v[0][0] + v[0][1]) + v[0][2]) + v[0][3]) * 0.25f, o1 = (((v[1][0] + v[1][1]) + v[1][2]) + v[1][3]) * 0.25f;
                        __builtin_amdgcn_raw_buffer_store_b32(pack2(o0, o1), rsY, wl < wins_left ? (unsigned)(wl * p.N + ncol) * 2u : OOR, 0u, 0);
                    }
            } else {
                const __amdgpu_buffer_rsrc_t rsY = desc(p.y, p.y_total, (long long)em0 * p.N * 2);
                const int rows_left = p.M - em0;
                const int row_lim = (rows_left < 256 ? rows_left : 256) - (wr * 64 + 4 * fh);
                const unsigned vbase = (unsigned)((wr * 64 + 4 * fh) * p.N + ncol) * 2u;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ru = i * 32 + (r & 3) + 8 * (r >> 2);
                        const float v0 = fmaxf(fmaf(acc[i][2 * h][r], s0, b0), 0.f), v1 = fmaxf(fmaf(acc[i][2 * h + 1][r], s1, b1), 0.f);
                        __builtin_amdgcn_raw_buffer_store_b32(pack2(v0, v1), rsY, ru < row_lim ? vbase : OOR, (unsigned)(ru * p.N * 2), 0);
                    }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// The two 32-channel stem convs (conv2 32 -> 32, conv3 32 -> 64 + AvgPool2d(2) on 112 x 112 maps) as in the parity mode's conv3x3_c32_kernel
// (conv_patch.hip), at one fp16 product: persistent workgroups, the whole weight (N x 288 fp16) in LDS once, a tile = 4 rows x 28 columns of
// output whose 6 x 30-pixel input patch is fetched ONCE (16-B chunks through registers, the next tile's in flight during this tile's MFMAs)
// and serves all nine taps by shifted LDS reads.  With K = 288 the generic kernel's tiles are a nine-step loop whose latencies nothing hides
// (0.55 + 0.67 ms for 1.6 + 1.2 GB at B = 1024).  Outputs: BatchNorm scale / bias, ReLU, (2x2 average: the four registers (r & 3) of a lane
// are one window) on the fp32 accumulator, rounded to fp16 into an LDS tile, then whole rows as 16-B lane stores.
// Same products and K order (kh, kw, 32 channels) as conv3x3_f16_kernel.  Bound: HBM.
// ---------------------------------------------------------------------------------------------------------------
constexpr int C_TR = 4, C_TC = 28, C_PR = 6, C_PC = 30, C_PPX = C_PR * C_PC, C_KTOT = 288, C_WROW = C_KTOT + 8;
struct C32P {
    const u16* x; const u16* w; const float* scale; const float* bias; u16* y;
    int B, H, W, tiles_w, tiles_h, n_tiles;
};

template <int N, int POOL>
__global__ __launch_bounds__(256, N == 32 ? 4 : 2) void conv3x3_c32_f16_kernel(const C32P p) {
    constexpr int TN = N / 32, NLD = (C_PPX * 4 + 255) / 256;      // 3 patch loads (16 B = 8 channels) per thread
    constexpr int PCL = POOL ? 40 : C_PC, PLN = C_PR * PCL * 32;    // patch row pitch (conv_patch.hip: pooled lanes of a read group stay on distinct rows mod 16)
    constexpr int SPITCH = N + 8, SROWS = POOL ? 28 : C_TR * C_TC;    // output staging: [pooled window | tile pixel][N]
    // N = 64: the lane's 36 weight fragments live in REGISTERS for the whole launch (144 of them; two workgroups per CU either way) -- from
    // LDS three fragment reads fed two MFMAs and the LDS pipe, not the matrix pipe, set the pace.  N = 32: in LDS (four workgroups per CU).
    constexpr bool WREG = N == 64;
    constexpr int WL_HALVES = WREG ? 0 : N * C_WROW;
    __shared__ __attribute__((aligned(16))) u16 lds[WL_HALVES + PLN + SROWS * SPITCH];
    u16* Wl = lds;                                                 // [N][C_WROW]
    u16* Pl = lds + WL_HALVES;                                     // [6 x PCL][32]
    u16* St = Pl + PLN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    u32x4 wreg[WREG ? 18 : 1][TN];
    if constexpr (WREG) {
#pragma unroll
        for (int t = 0; t < 18; ++t)
#pragma unroll
            for (int j = 0; j < TN; ++j) wreg[t][j] = *(const u32x4*)(p.w + (size_t)(j * 32 + fr) * C_KTOT + (t >> 1) * 32 + (2 * (t & 1) + fh) * 8);
    } else {
        for (int i = tid; i < N * (C_KTOT / 8); i += 256) {
            const int n = i / (C_KTOT / 8), c = i - n * (C_KTOT / 8);
            *(u32x4*)(Wl + n * C_WROW + c * 8) = *(const u32x4*)(p.w + (size_t)n * C_KTOT + c * 8);
        }
    }
    float sv[TN], bv[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { sv[j] = p.scale ? p.scale[j * 32 + fr] : 1.f; bv[j] = p.bias ? p.bias[j * 32 + fr] : 0.f; }
    // this lane's MFMA row (A operand): tile pixel (dy, c) -> patch row of tap (0, 0); rows >= 112 are padding
    const int ml = wave * 32 + fr;
    int a_dy, a_c;
    bool a_ok;
    if (POOL) { const int o = ml >> 2, q = ml & 3; a_dy = 2 * (o / 14) + (q >> 1); a_c = 2 * (o % 14) + (q & 1); a_ok = ml < C_TR * C_TC; }
    else { a_dy = wave; a_c = fr; a_ok = fr < C_TC; }
    const int a_row0 = a_ok ? a_dy * PCL + a_c : 0;
    const long long img_bytes = (long long)p.H * p.W * 64;
    u32x4 pre[NLD];
    auto tile_coords = [&](int tile, int& b, int& h0, int& w0) {
        const int tw = tile % p.tiles_w, r = tile / p.tiles_w;
        b = r / p.tiles_h; h0 = (r - b * p.tiles_h) * C_TR; w0 = tw * C_TC;
    };
    auto load_patch = [&](int tile) {
        int b, h0, w0;
        tile_coords(tile, b, h0, w0);
        const __amdgpu_buffer_rsrc_t rs = desc(p.x, (long long)p.B * img_bytes, (long long)b * img_bytes);
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int idx = tid + 256 * k, i = idx >> 2, chunk = idx & 3;
            const int py = i / C_PC, px = i - py * C_PC, h = h0 - 1 + py, w = w0 - 1 + px;
            const bool ok = i < C_PPX && h >= 0 && h < p.H && w >= 0 && w < p.W;
            pre[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, ok ? (unsigned)((h * p.W + w) * 64 + chunk * 16) : OOR, 0, 0);
        }
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int idx = tid + 256 * k, i = idx >> 2, chunk = idx & 3;
            if (i < C_PPX) {
                const int py = i / C_PC, row = py * PCL + (i - py * C_PC);
                *(u32x4*)(Pl + row * 32 + ((chunk ^ ((row >> 2) & 3)) << 3)) = pre[k];
            }
        }
    };
    int tile = blockIdx.x;
    if (tile < p.n_tiles) load_patch(tile);
    for (; tile < p.n_tiles; tile += gridDim.x) {
        store_patch();
        __syncthreads();                                           // patch (and, the first time, the weights) visible; the previous tile's staged outputs are out
        if (tile + (int)gridDim.x < p.n_tiles) load_patch(tile + gridDim.x);
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
                const u32x4 af = *(const u32x4*)(Pl + row * 32 + (((2 * ks + fh) ^ ((row >> 2) & 3)) << 3));
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    u32x4 wf;
                    if constexpr (WREG) wf = wreg[2 * tap + ks][j];
                    else wf = *(const u32x4*)(Wl + (j * 32 + fr) * C_WROW + tap * 32 + (2 * ks + fh) * 8);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af), __builtin_bit_cast(f16x8, wf), acc[j], 0, 0, 0);
                }
            }
        }
        // ---- BatchNorm scale / bias, ReLU, (2x2 average) -> fp16 staging tile ----
        if constexpr (POOL) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {                          // window o = registers 4 t .. 4 t + 3 of this lane
                const int o = wave * 8 + 2 * t + fh;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    float v[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = fmaxf(fmaf(acc[j][4 * t + q], sv[j], bv[j]), 0.f);
                    const _Float16 hv = (_Float16)((((v[0] + v[1]) + v[2]) + v[3]) * 0.25f);
                    if (o < 28) St[o * SPITCH + j * 32 + fr] = __builtin_bit_cast(u16, hv);
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = (r & 3) + 8 * (r >> 2) + 4 * fh;     // accumulator row = the wave's tile row, column c
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const _Float16 hv = (_Float16)fmaxf(fmaf(acc[j][r], sv[j], bv[j]), 0.f);
                    if (c < C_TC) St[(wave * C_TC + c) * SPITCH + j * 32 + fr] = __builtin_bit_cast(u16, hv);
                }
            }
        }
        __syncthreads();                                           // every wave is done reading this patch; the staged tile is complete
        // ---- whole output rows as 16-B lane stores ----
        int b, h0, w0;
        tile_coords(tile, b, h0, w0);
        constexpr int QPP = N / 8;                                 // 16-B chunks per pixel
        if constexpr (POOL) {
            const int Hp = p.H >> 1, Wp = p.W >> 1;
            const long long base_px = ((long long)b * Hp + (h0 >> 1)) * Wp + (w0 >> 1);
            const __amdgpu_buffer_rsrc_t rs = desc(p.y, (long long)p.B * Hp * Wp * N * 2, base_px * N * 2);
            for (int q = tid; q < 28 * QPP; q += 256) {
                const int o = q / QPP, ch = q - o * QPP;
                const u32x4 v = *(const u32x4*)(St + o * SPITCH + ch * 8);
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, (unsigned)(((o / 14) * Wp + (o % 14)) * (N * 2) + ch * 16), 0, 0);
            }
        } else {
            const long long base_px = ((long long)b * p.H + h0) * p.W + w0;
            const __amdgpu_buffer_rsrc_t rs = desc(p.y, (long long)p.B * p.H * p.W * N * 2, base_px * N * 2);
            for (int q = tid; q < 4 * C_TC * QPP; q += 256) {
                const int pix = q / QPP, ch = q - pix * QPP, dy = pix / C_TC, c = pix - dy * C_TC;
                const u32x4 v = *(const u32x4*)(St + pix * SPITCH + ch * 8);
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, (unsigned)((dy * p.W + c) * (N * 2) + ch * 16), 0, 0);
            }
        }
    }
}

template <int WM, int WN, int TN>
int launch_conv(ConvHP& p, int pool, hipStream_t s) {
    constexpr int BM = WM * 128, BN = WN * TN * 32;
    p.tiles_n = (p.N + BN - 1) / BN;
    p.n_tiles = ((p.M + BM - 1) / BM) * p.tiles_n;
    if (pool) hipLaunchKernelGGL((conv3x3_f16_kernel<WM, WN, TN, 1>), dim3(p.n_tiles), dim3(256), 0, s, p);
    else hipLaunchKernelGGL((conv3x3_f16_kernel<WM, WN, TN, 0>), dim3(p.n_tiles), dim3(256), 0, s, p);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

}  // namespace

// see include/dbmm.h
extern "C" int dbmm_conv3x3_bn_relu_f16(const void* x, const void* w, const float* scale, const float* bias, void* y, int64_t B, int64_t H,
                                        int64_t W, int64_t Cin, int64_t Cout, int pool, void* stream) {
    if (!x || !w || !y) return DBMM_E_ARG;
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (pool != 0 && pool != 2)) return DBMM_E_SHAPE;
    const long long M = (long long)B * H * W;
    if (M > INT32_MAX - 1024) return DBMM_E_SHAPE;
    if ((Cin % 32) || (Cout % 8)) return DBMM_E_UNSUPPORTED;
    if (pool && ((H & 1) || (W & 1))) return DBMM_E_UNSUPPORTED;
    if (!dbmm_aligned16(x) || !dbmm_aligned16(w) || !dbmm_aligned16(y)) return DBMM_E_ALIGN;
    const long long wb = Cout * 9 * Cin * 2;
    if (wb >= EXT_LIM || (long long)(1024 + 2 * W + 8) * Cin * 2 >= EXT_LIM || 1024LL * Cout * 2 >= EXT_LIM) return DBMM_E_UNSUPPORTED;
    ConvHP p{};
    p.x = (const u16*)x; p.w = (const u16*)w; p.scale = scale; p.bias = bias; p.y = (u16*)y;
    p.x_total = M * Cin * 2; p.w_total = wb; p.y_total = (pool ? M / 4 : M) * Cout * 2;
    p.B = (int)B; p.H = (int)H; p.W = (int)W; p.Cin = (int)Cin; p.N = (int)Cout; p.M = (int)M;
    hipStream_t s = (hipStream_t)stream;
    // the 32-channel stem convs on 4 x 28 output tiles: the persistent patch kernel (option conv_patch)
    if (dbmm_opt(OPT_CONV_PATCH) && Cin == 32 && (Cout == 32 || Cout == 64) && (H % C_TR) == 0 && (W % C_TC) == 0 && H * W * 64 < EXT_LIM &&
        B * (H / C_TR) * (W / C_TC) <= INT32_MAX) {
        C32P q{};
        q.x = (const u16*)x; q.w = (const u16*)w; q.scale = scale; q.bias = bias; q.y = (u16*)y; q.B = (int)B; q.H = (int)H; q.W = (int)W;
        q.tiles_w = (int)(W / C_TC); q.tiles_h = (int)(H / C_TR); q.n_tiles = (int)(B * q.tiles_h * q.tiles_w);
        const int per_cu = Cout == 32 ? 4 : 2;
        const int grid = q.n_tiles < 256 * per_cu ? q.n_tiles : 256 * per_cu;
        if (Cout == 32) {
            if (pool) hipLaunchKernelGGL((conv3x3_c32_f16_kernel<32, 1>), dim3(grid), dim3(256), 0, s, q);
            else hipLaunchKernelGGL((conv3x3_c32_f16_kernel<32, 0>), dim3(grid), dim3(256), 0, s, q);
        } else {
            if (pool) hipLaunchKernelGGL((conv3x3_c32_f16_kernel<64, 1>), dim3(grid), dim3(256), 0, s, q);
            else hipLaunchKernelGGL((conv3x3_c32_f16_kernel<64, 0>), dim3(grid), dim3(256), 0, s, q);
        }
        DBMM_CHECK_LAUNCH();
        return DBMM_OK;
    }
    // Cout % 256 == 0 (layers 3 / 4) at a size that fills the chip: the eight-phase 256 x 256 kernel (option f16_conv_8ph)
    if (dbmm_opt(OPT_F16_CONV_8PH) && (Cout % 256) == 0 && M >= 16384) {
        p.tiles_n = (int)(Cout / 256);
        p.n_tiles = (int)((M + 255) / 256) * p.tiles_n;
        const int grid = p.n_tiles < 256 ? p.n_tiles : 256;
        if (pool) hipLaunchKernelGGL((conv3x3_f16_8ph_kernel<1>), dim3(grid), dim3(512), 0, s, p);
        else hipLaunchKernelGGL((conv3x3_f16_8ph_kernel<0>), dim3(grid), dim3(512), 0, s, p);
        DBMM_CHECK_LAUNCH();
        return DBMM_OK;
    }
    if (Cout <= 32) return launch_conv<4, 1, 1>(p, pool, s);
    if (Cout <= 64) return launch_conv<4, 1, 2>(p, pool, s);
    return launch_conv<2, 2, 2>(p, pool, s);
}

// the streaming 1x1 kernel (see dbmm_conv1x1_bn_act_f16 in f16_ops.hip, which routes the HBM-bound shapes here)
int dbmm_conv1x1_stream_f16(const void* x, const void* w, const float* scale, const float* bias, const void* residual, void* y, int64_t M,
                            int64_t Cin, int64_t Cout, int act, void* stream) {
    if (!x || !w || !y) return DBMM_E_ARG;
    if (M <= 0 || Cin <= 0 || Cout <= 0 || M > INT32_MAX - 1024) return DBMM_E_SHAPE;
    if (act != DBMM_ACT_NONE && act != DBMM_ACT_RELU) return DBMM_E_UNSUPPORTED;
    if ((Cin % 32) || (Cout % 8)) return DBMM_E_UNSUPPORTED;
    if (!dbmm_aligned16(x) || !dbmm_aligned16(w) || !dbmm_aligned16(y) || (residual && !dbmm_aligned16(residual))) return DBMM_E_ALIGN;
    if (Cout * Cin * 2 >= EXT_LIM || 512LL * Cin * 2 >= EXT_LIM || 512LL * Cout * 2 >= EXT_LIM) return DBMM_E_UNSUPPORTED;
    ConvHP p{};
    p.x = (const u16*)x; p.w = (const u16*)w; p.scale = scale; p.bias = bias; p.y = (u16*)y; p.res = (const u16*)residual; p.act = act;
    p.x_total = M * Cin * 2; p.w_total = Cout * Cin * 2; p.y_total = M * Cout * 2;
    p.Cin = (int)Cin; p.N = (int)Cout; p.M = (int)M;
    hipStream_t s = (hipStream_t)stream;
    if (Cout <= 64) {
        p.tiles_n = 1; p.n_tiles = (int)((M + 255) / 256);
        if (residual) hipLaunchKernelGGL((conv1x1_f16_kernel<4, 1, 2, 1>), dim3(p.n_tiles), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv1x1_f16_kernel<4, 1, 2, 0>), dim3(p.n_tiles), dim3(256), 0, s, p);
    } else {
        p.tiles_n = (int)((Cout + 127) / 128); p.n_tiles = (int)((M + 255) / 256) * p.tiles_n;
        if (residual) hipLaunchKernelGGL((conv1x1_f16_kernel<2, 2, 4, 1>), dim3(p.n_tiles), dim3(256), 0, s, p);
        else hipLaunchKernelGGL((conv1x1_f16_kernel<2, 2, 4, 0>), dim3(p.n_tiles), dim3(256), 0, s, p);
    }
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

// the streaming kernel with a second operand pair (dbmm_conv1x1_dual_bn_act_f16 in f16_ops.hip routes here what its GEMM kernel does not take)
int dbmm_conv1x1_dual_stream_f16(const void* y2, const void* w3, const float* scale3, const void* xp, const void* wd, const float* ratio,
                                 const float* bias, void* out, int64_t M, int64_t K, int64_t K2, int64_t Cout, int act, void* stream) {
    if (M <= 0 || M > INT32_MAX - 1024) return DBMM_E_SHAPE;
    if ((K % 32) || (K2 % 32) || (Cout % 8) || Cout <= 64) return DBMM_E_UNSUPPORTED;
    if (Cout * (K > K2 ? K : K2) * 2 >= EXT_LIM || 512LL * (K > K2 ? K : K2) * 2 >= EXT_LIM || 512LL * Cout * 2 >= EXT_LIM) return DBMM_E_UNSUPPORTED;
    ConvHP p{};
    p.x = (const u16*)y2; p.w = (const u16*)w3; p.scale = scale3; p.bias = bias; p.y = (u16*)out; p.res = nullptr; p.act = act;
    p.x_total = M * K * 2; p.w_total = Cout * K * 2; p.y_total = M * Cout * 2;
    p.x2 = (const u16*)xp; p.w2 = (const u16*)wd; p.ratio = ratio; p.x2_total = M * K2 * 2; p.w2_total = Cout * K2 * 2; p.Cin2 = (int)K2;
    p.Cin = (int)K; p.N = (int)Cout; p.M = (int)M;
    p.tiles_n = (int)((Cout + 127) / 128); p.n_tiles = (int)((M + 255) / 256) * p.tiles_n;
    hipLaunchKernelGGL((conv1x1_f16_kernel<2, 2, 4, 0, 1>), dim3(p.n_tiles), dim3(256), 0, (hipStream_t)stream, p);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

static int stem_f16_launch(const void* x_nchw, int x_is_f16, const float* w, const float* scale, const float* bias, void* y_nhwc, int64_t B,
                           int64_t H, int64_t W, int64_t Cout, bool mfma, void* stream) {
    if (!x_nchw || !w || !y_nhwc) return DBMM_E_ARG;
    if (B <= 0 || H <= 0 || W <= 0) return DBMM_E_SHAPE;
    if (Cout != 32 && Cout != 64) return DBMM_E_UNSUPPORTED;
    if (!dbmm_aligned16(y_nhwc) || !dbmm_aligned16(w)) return DBMM_E_ALIGN;
    const int Ho = (int)((H - 1) / 2 + 1), Wo = (int)((W - 1) / 2 + 1);
    const long long M = (long long)B * Ho * Wo;
    const dim3 g((unsigned)((M + 255) / 256));
    hipStream_t s = (hipStream_t)stream;
    if (mfma && 3 * H * W * 8 < 0x7FFFFFF0LL && (M + 31) / 32 <= INT32_MAX) {
        const int n_blocks = (int)((M + 31) / 32);
        const int wgs = (n_blocks + 3) / 4 < 256 * 8 ? (n_blocks + 3) / 4 : 256 * 8;
#define DBMM_STEMM(NB, T) hipLaunchKernelGGL((stem_s2_f16_mfma_kernel<NB, T>), dim3(wgs), dim3(256), 0, s, (const T*)x_nchw, w, scale, bias, (u16*)y_nhwc, (int)B, (int)H, (int)W, Ho, Wo, n_blocks)
        if (Cout == 32) { if (x_is_f16) DBMM_STEMM(1, _Float16); else DBMM_STEMM(1, float); }
        else { if (x_is_f16) DBMM_STEMM(2, _Float16); else DBMM_STEMM(2, float); }
#undef DBMM_STEMM
        DBMM_CHECK_LAUNCH();
        return DBMM_OK;
    }
#define DBMM_STEMH(C, T) hipLaunchKernelGGL((stem_s2_f16_kernel<C, T>), g, dim3(256), 0, s, (const T*)x_nchw, w, scale, bias, (u16*)y_nhwc, (int)B, (int)H, (int)W, Ho, Wo)
    if (Cout == 32) { if (x_is_f16) DBMM_STEMH(32, _Float16); else DBMM_STEMH(32, float); }
    else { if (x_is_f16) DBMM_STEMH(64, _Float16); else DBMM_STEMH(64, float); }
#undef DBMM_STEMH
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

// see include/dbmm.h: fp32 weights (e.g. with BatchNorm folded in) used as they are -- the FMA kernel
extern "C" int dbmm_conv_stem_s2_f16(const void* x_nchw, int x_is_f16, const float* w, const float* bias, void* y_nhwc, int64_t B, int64_t H,
                                     int64_t W, int64_t Cout, void* stream) {
    return stem_f16_launch(x_nchw, x_is_f16, w, nullptr, bias, y_nhwc, B, H, W, Cout, false, stream);
}

// see include/dbmm.h: fp16-exact weights + BatchNorm scale / bias -- the MFMA gather kernel (option stem_mfma = 0: the FMA kernel)
extern "C" int dbmm_conv_stem_s2_bn_f16(const void* x_nchw, int x_is_f16, const float* w, const float* scale, const float* bias, void* y_nhwc,
                                        int64_t B, int64_t H, int64_t W, int64_t Cout, void* stream) {
    if (!scale) return DBMM_E_ARG;
    return stem_f16_launch(x_nchw, x_is_f16, w, scale, bias, y_nhwc, B, H, W, Cout, dbmm_opt(OPT_STEM_MFMA) != 0, stream);
}

extern "C" int dbmm_avgpool2_f16(const void* x, void* y, int64_t B, int64_t H, int64_t W, int64_t C, void* stream) {
    if (!x || !y) return DBMM_E_ARG;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (H & 1) || (W & 1)) return DBMM_E_SHAPE;
    if (C % 8) return DBMM_E_UNSUPPORTED;
    if (!dbmm_aligned16(x) || !dbmm_aligned16(y)) return DBMM_E_ALIGN;
    const long long n_out = B * (H / 2) * (W / 2) * (C / 8);
    hipLaunchKernelGGL(avgpool2_f16_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const u16*)x, (u16*)y,
                       (int)H, (int)W, (int)C, n_out);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}
