// fp32 implicit-GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32, exact f32).
//
//   C[M][N] = act( alpha * (A[M][K] . W[N][K]^T + bias[N]) + R[M][N] )
//
// One kernel family serves every dense contraction of the hot path:
//   AMODE 0  A row-major [M][lda]              (1x1 convs on NHWC, nn.Linear inputs)
//   AMODE 1  A gathered from an NHWC image     (KHxKW conv, zero padding; K = (kh,kw,cin))
//   AMODE 2  A K-major [K][lda]                (weight gradients: reduction over the batch)
//   WMODE 0  W [N][ldw] (nn.Linear / packed conv weight)      WMODE 1  W [K][ldw]
//
// Tiling: 256 threads = 4 wave64; block tile BM x BN x BK; each wave owns TM x TN MFMA tiles
// of 32x32.  Global -> registers -> LDS (double buffered, one barrier per K chunk, next
// chunk's global loads issued before the current chunk's MFMAs).  LDS rows hold BK floats
// with the 16-B chunk index XOR-swizzled by row bits, which makes the ds_read_b128 operand
// fetches bank-conflict free (16 lanes x 16 B cover all 64 banks).
//
// K is consumed in a permuted order that is identical for A and W: per 8-wide group g the
// lane half h = lane>>5 reads k = 8g+4h .. 8g+4h+3 as one 16-B LDS read and MFMA step j uses
// element j of both halves (k pair {8g+j, 8g+4+j}).  Any fixed K order is a valid fp32 sum.
//
// Operand fetch (FAST = 1, the production path): `buffer_load_dwordx4` through a buffer
// descriptor.  Out-of-tile rows, out-of-image halo taps and rows past M use an out-of-range
// offset, for which the hardware returns zeros -- no branches, no selects.  Row-major
// operands keep a per-thread row offset in a VGPR for the whole K loop and advance along K
// through the scalar offset operand (zero VALU instructions per load); the conv gather adds
// one wave-uniform tap offset per load and tests one bit of a per-row tap-validity mask.
// (An ablation with the loads removed ran at 149 TFLOP/s against 119 for the first,
// branchy flat-load loader: address arithmetic + exec-mask branches, not latency, were the
// loss.)  FAST = 0 is the general fallback (K tails, K-major operands, Cin % BK != 0,
// operands >= 2 GiB).
//
// DMA = 1 (default with FAST): the same buffer loads write LDS directly (`buffer_load ... lds`):
// no staging VGPRs, no ds_write pass.  The LDS-DMA destination is wave-uniform base + lane*16 B,
// so the swizzle moves to the per-lane SOURCE chunk (lane l of a row loads chunk lc ^ swz(row)
// and lands on physical chunk lc), the rule-21 "linear dest + swizzled source + swizzled read".
//
// Work distribution: one tile per workgroup, or -- when the tile count does not divide over
// the 256 CUs (e.g. 784 tiles: every CU waits for the 16 that got a 4th tile) -- stream-K:
// a resident grid of 256 x MINB workgroups each takes an equal contiguous share of all
// (tile, K-chunk) units; tiles cut by a share boundary leave raw partial accumulators in a
// caller-provided workspace and a small second kernel sums them and runs the epilogue.
// No inter-workgroup synchronisation inside a launch; results do not depend on scheduling.
//
// Roofline: MFMA-bound. 32x32x2 = 4096 FLOP / 64 cycles / SIMD -> 157.3 TFLOP/s chip peak.
#include <stdlib.h>
#include "common.h"

namespace {

struct IgemmP {
    const float* a;
    const float* w;
    const float* bias;
    const float* res;
    float* c;
    long long lda, ldw, ldr, ldc;
    int M, N, K;
    int H, W, Cin, Ho, Wo, KH, KW, stride, pad;  // AMODE 1 only
    int wl;                             // AMODE 1: K order of the packed weights (DBMM_WL_*)
    int slab;                           // AMODE 1: 0 = tap-major K, else channels per slab of a chunk-major K (16 / 32)
    int act;
    float alpha;
    int tiles_n, n_tiles;
    long long sa, sw, sbias, sres, sc;  // per-batch element strides (grid.y = batch index)
    unsigned a_bytes, w_bytes;          // FAST loader: eligibility (0 = not eligible) / W buffer extent (< 2 GiB)
    long long a_total, a2_total;        // bytes of the A operands.  They may exceed 2 GiB (RN50 layer 1 from B = 668):
                                        // every tile builds its buffer descriptor on a 64-bit base of its own (a_desc)
                                        // so the 32-bit offsets only ever span one tile's rows
    int sk_blocks;                      // stream-K: resident grid size (0 = one tile per block)
    float* sk_ws;                       // stream-K: [sk_blocks][2][BM*BN] partial accumulators
    // second operand pair of the fused "conv3 + downsample branch" GEMM (TWO = 1): its K2 / 32 chunks
    // run first, then the accumulators are multiplied per output channel by ratio[n] * 2^(s - s2) and the
    // main pair (a, wh) continues in the same accumulators
    const float* a2; long long lda2; unsigned a2_bytes; int K2;
    const unsigned short* wh2; unsigned wh2_bytes; long long ldw2;
    const float* a2_absmax; const float* ratio;
    int sk_nk;                          // stream-K: work units per tile when they are not K / BK chunks (0 = chunks)
    const unsigned short* w3;           // split path: W as three bf16 planes [3][N][ldw]
    unsigned w3_bytes;
    const unsigned short* wh;           // fp16-pair path: W * 2^w_exp as two fp16 planes [2][N][ldw]
    unsigned wh_bytes;
    int w_exp;
    int pool2;                          // epilogue averages 2x2 output windows (rows walked window-major)
    float* c_full;                      // pool2 only, optional: the un-pooled output [M][ldc], standard row order
    int nw;                             // fp16-pair path: W planes present (2 = hi + lo, 1 = W exact in fp16)
    const float* oscale;                // optional per-output-channel scale applied to the accumulator (unfolded BN)
    const float* a_absmax;              // fp16-pair path: device scalar >= max|A| (null: path not in use)
    float* absmax_out;                  // optional device scalar: atomic max of |C| over the written outputs
    int epi_direct;                     // epilogue straight from the accumulator layout (igemm_epilogue.inc); DBMM_IGEMM_EPI_DIRECT=0: staged
};

inline int epi_direct_env() { return dbmm_opt(OPT_IGEMM_EPI_DIRECT); }

// fp16-pair path: A is scaled by 2^s so that max|A| * 2^s lies in [2^13, 2^14) (fp16 tops out at
// 65504; the fp32 accumulator is rescaled by 2^-(s + w_exp) in the epilogue -- powers of two,
// so the scaling itself is exact).
__device__ __forceinline__ int a_scale_exp(const float* a_absmax) {
    const unsigned b = __float_as_uint(*a_absmax) & 0x7fffffffu;
    int s = b ? 13 - ((int)(b >> 23) - 127) : 0;
    return s < -60 ? -60 : (s > 60 ? 60 : s);
}
__device__ __forceinline__ float pow2f(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }
// pool2 row order: row m = 4 * (pooled pixel, standard order) + (dy * 2 + dx).  Returns the
// standard-order pixel index of the window's top-left corner (q = 0); q adds (q >> 1) * Wo + (q & 1).
__device__ __forceinline__ int pool2_base_pixel(const IgemmP& p, int mp) {
    const int wp2 = p.Wo >> 1, hwp = (p.Ho >> 1) * wp2;
    const int n = mp / hwp, rem = mp - n * hwp, hp = rem / wp2;
    return (n * p.Ho + 2 * hp) * p.Wo + 2 * (rem - hp * wp2);
}

__device__ __forceinline__ float igemm_acc_scale(const IgemmP& p) {
    return p.a_absmax ? pow2f(-a_scale_exp(p.a_absmax) - p.w_exp) : 1.f;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOR = 0x80000000u;   // >= any accepted extent, and OOR + (K offset) cannot wrap
constexpr long long EXT_LIM = 0x7FFFFFF0LL;

// Buffer descriptor of an A operand rebased to `shift` bytes (16-B aligned, wave-uniform) past its start: extent =
// what is left of the tensor, capped below 2 GiB.  Offsets are then relative to the tile's first row / image, a few
// MB at most, while the tensor itself may be any size.  zero = the zero-extent twin (loads return 0).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t a_desc(const float* base, long long total, long long shift, bool zero = false) {
    long long ext = total - shift;
    ext = ext < 0 ? 0 : (ext > EXT_LIM ? EXT_LIM : ext);
    return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)base + shift), 0, zero ? 0 : (int)ext, 0x00020000);
}

__device__ __forceinline__ f32x4 buf_load16(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// LDS-DMA: 16 B per lane from a buffer straight into LDS at (wave-uniform dst) + 16 * lane.
// (The builtin only exists in the device pass, hence the guard; the host pass never calls it.)
__device__ __forceinline__ void buf_load16_lds(__amdgpu_buffer_rsrc_t r, float* lds_dst, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
#else
    (void)r; (void)lds_dst; (void)voff; (void)soff;
#endif
}

// LDS rows hold BK floats; the 16-B chunk index is XOR-swizzled with row bits so that the 16
// lanes of a ds_read_b128 group (16 x 16 B = all 64 banks) never collide:
//   BK = 32 (128-B rows, 2 rows per bank sweep): chunk ^= (row >> 1) & 7
//   BK = 16 ( 64-B rows, 4 rows per bank sweep): chunk ^= (row >> 2) & 3
template <int BK>
__device__ __forceinline__ int lds_swz(int row) {
    return BK == 32 ? ((row >> 1) & 7) : ((row >> 2) & 3);
}
template <int BK>
__device__ __forceinline__ int lds_off(int row, int kchunk) {  // float index of a 16-B chunk
    return row * BK + ((kchunk ^ lds_swz<BK>(row)) << 2);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int BK>
struct Geo {
    static constexpr int TM = BM / WAVES_M / 32, TN = BN / WAVES_N / 32;
    static constexpr int WTN = TN * 32;          // wave tile width
    static constexpr int LROW = WTN + 4;         // padded epilogue-staging row (16-B aligned)
    static constexpr int EPI_FLOATS = 4 * 32 * LROW;
    static constexpr int LDS_FLOATS = 2 * (BM + BN) * BK > EPI_FLOATS ? 2 * (BM + BN) * BK : EPI_FLOATS;
};

// ---------------------------------------------------------------------------------------------
// One output tile over K chunks [kb, ke).  partial == nullptr: full epilogue; otherwise the raw
// accumulators are written to `partial` ([TM*TN*16][256] floats, thread-contiguous).
// FIXUP = 1 (second stream-K kernel): no K loop; the accumulators are the sum of the partials of
// workgroups fix_c0, fix_c0+1, ... whose shares intersect the tile, then the epilogue runs.
// (The epilogue is deliberately part of this function body: hipcc allocated ~250 VGPRs and
// spilled ~950 at the 128-register budget when it was a separate inlined function.)
//
// Epilogue.  C/D layout of the 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
// Fast path (N, ldc, ldr multiples of 4): each wave transposes 32 output rows at a time through
// its own slice of the (now idle) staging LDS so that every lane owns 4 consecutive columns:
// bias/residual loads and stores are 16 B per lane, a whole 128/256-B row segment per 8/16
// lanes, and all residual loads of a half tile are in flight together.  (The per-element path
// below it was latency-bound at ~1 TB/s on the conv3 + residual layers.)
// ---------------------------------------------------------------------------------------------
template <int BM, int BN, int WAVES_M, int WAVES_N, int AMODE, int WMODE, int BK, int FAST, int FIXUP, int DMA>
__device__ __forceinline__ void igemm_tile(const IgemmP& p, float* lds, int tile, int kb, int ke, float* partial,
                                           int fix_c0) {
    using G = Geo<BM, BN, WAVES_M, WAVES_N, BK>;
    constexpr int TM = G::TM, TN = G::TN, WTN = G::WTN, LROW = G::LROW;
    constexpr int CPR = BK / 4;                  // 16-B chunks per LDS row
    constexpr int RPP = 256 / CPR;               // rows covered by one pass of the 256 threads
    constexpr int ALD = (BM + RPP - 1) / RPP, WLD = (BN + RPP - 1) / RPP;  // float4 loads per thread per chunk
    // (a tile narrower than one pass of the 256 threads leaves the surplus threads idle: the
    //  `< BM` / `< BN` / `< BK` guards below are compile-time true otherwise)
    float* As = lds;                 // [2][BM][BK]
    float* Ws = lds + 2 * BM * BK;   // [2][BN][BK]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
    const int wm0 = (wave / WAVES_N) * (TM * 32), wn0 = (wave % WAVES_N) * (TN * 32);

    // ---- per-thread global-load bookkeeping (fallback loader) -----------------------------------
    // row-major / conv: thread covers 16-B chunk (tid % CPR) of rows tid / CPR + RPP*i
    const int lc = tid % CPR, lr = tid / CPR;
    long long a_base[ALD];
    int a_hi0[ALD], a_wi0[ALD];
    if constexpr (!FAST && AMODE == 0) {
#pragma unroll
        for (int i = 0; i < ALD; ++i) {
            const int m = m0 + lr + RPP * i;
            a_base[i] = (m < p.M && lr + RPP * i < BM) ? (long long)m * p.lda : -1;
        }
    } else if constexpr (!FAST && AMODE == 1) {
#pragma unroll
        for (int i = 0; i < ALD; ++i) {
            const int m = m0 + lr + RPP * i;
            if (m < p.M && lr + RPP * i < BM) {
                const int hw = p.Ho * p.Wo;
                const int n = m / hw, rem = m - n * hw;
                const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
                a_base[i] = (long long)n * p.H * p.W * p.Cin;
                a_hi0[i] = ho * p.stride - p.pad;
                a_wi0[i] = wo * p.stride - p.pad;
            } else {
                a_base[i] = 0; a_hi0[i] = -(1 << 28); a_wi0[i] = 0;
            }
        }
    }
    // K-major operands: thread covers m-quad (tid % (BX/4)) of k rows tid/(BX/4) + step*i
    constexpr int AQ = BM / 4, WQ = BN / 4;

    // ---- FAST loader state ---------------------------------------------------------------------
    unsigned fa_off[ALD], fa_mask[ALD], fw_off[WLD];
    __amdgpu_buffer_rsrc_t rsA, rsW;
    int f_ci0 = 0, f_kh = 0, f_kw = 0, f_tap = 0;     // conv: wave-uniform position along K
    if constexpr (FAST) {
        static_assert(!FAST || (AMODE != 2 && WMODE == 0), "FAST loader: K-contiguous operands only");
        // A descriptor rebased to the tile: row m0 (row-major) / the image of row m0 (conv gather)
        int fn0 = 0;
        if constexpr (AMODE == 1) fn0 = m0 / (p.Ho * p.Wo);
        const long long a_shift = AMODE == 0 ? (long long)m0 * p.lda * 4 : (long long)fn0 * p.H * p.W * p.Cin * 4;
        rsA = a_desc(p.a, p.a_total, a_shift);
        rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, (int)p.w_bytes, 0x00020000);
#pragma unroll
        for (int i = 0; i < WLD; ++i) {
            const int n = n0 + lr + RPP * i;
            const unsigned cw = DMA ? (unsigned)(lc ^ lds_swz<BK>(lr + RPP * i)) : (unsigned)lc;   // source chunk
            fw_off[i] = (lr + RPP * i < BN && n < p.N) ? (unsigned)n * (unsigned)p.ldw * 4u + cw * 16u : OOR;
        }
#pragma unroll
        for (int i = 0; i < ALD; ++i) {
            const int m = m0 + lr + RPP * i;
            const bool rv = (lr + RPP * i < BM) && (m < p.M);
            const unsigned ca = DMA ? (unsigned)(lc ^ lds_swz<BK>(lr + RPP * i)) : (unsigned)lc;   // source chunk
            if constexpr (AMODE == 0) {
                fa_off[i] = rv ? (unsigned)(m - m0) * (unsigned)p.lda * 4u + ca * 16u : OOR;
                fa_mask[i] = 0;
            } else {
                const int hw = p.Ho * p.Wo;
                const int n = m / hw, rem = m - n * hw;
                const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
                const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
                // modular 32-bit arithmetic: the halo start may lie "before" the tile's first image, the tap
                // offset added per load brings every valid tap back into range
                fa_off[i] = ((unsigned)((n - fn0) * p.H + hi0) * (unsigned)p.W + (unsigned)wi0) * (unsigned)p.Cin * 4u + ca * 16u;
                unsigned msk = 0;
                for (int kh = 0; kh < p.KH; ++kh)
                    for (int kw = 0; kw < p.KW; ++kw)
                        if (hi0 + kh >= 0 && hi0 + kh < p.H && wi0 + kw >= 0 && wi0 + kw < p.W)
                            msk |= 1u << (kh * p.KW + kw);
                fa_mask[i] = rv ? msk : 0u;
            }
        }
        if constexpr (AMODE == 1) {   // position of chunk kb along K
            if (p.slab) {                             // K = (cin/slab, kh, kw, slab): taps cycle fastest
                const int taps = p.KH * p.KW, cpc = p.slab / BK > 0 ? p.slab / BK : 1;   // chunks per (slab, tap) cell
                const int cell = kb / cpc;
                f_tap = cell % taps; f_ci0 = (cell / taps) * p.slab + (kb % cpc) * BK;
            } else {                                  // K = (kh, kw, cin)
                const int k = kb * BK;
                f_tap = k / p.Cin; f_ci0 = k - f_tap * p.Cin;
            }
            f_kh = f_tap / p.KW; f_kw = f_tap - f_kh * p.KW;
        }
    }

    f32x4 a_reg[ALD], w_reg[WLD];

    // LDS-DMA: chunk k0 -> staging buffer `buf`; lane l of a wave lands on base + 16*l, i.e. on
    // row (wave's first row) + l / CPR, physical chunk l % CPR
    const int wrow0 = (tid & ~63) / CPR;          // wave-uniform first row of this wave's pass
    auto dma_chunk = [&](int k0, int buf) {
        if constexpr (FAST && DMA) {
            float* Ab = As + buf * BM * BK;
            float* Wb = Ws + buf * BN * BK;
            unsigned delta = 0, bit = 0;
            if constexpr (AMODE == 1) {
                delta = ((unsigned)(f_kh * p.W + f_kw) * (unsigned)p.Cin + (unsigned)f_ci0) * 4u;
                bit = 1u << f_tap;
            }
#pragma unroll
            for (int i = 0; i < ALD; ++i) {
                if (wrow0 + RPP * i < BM) {           // wave-uniform
                    unsigned voff, soff;
                    if constexpr (AMODE == 0) { voff = fa_off[i]; soff = (unsigned)k0 * 4u; }
                    else { voff = (fa_mask[i] & bit) ? fa_off[i] + delta : OOR; soff = 0u; }
                    buf_load16_lds(rsA, Ab + (wrow0 + RPP * i) * BK, voff, soff);
                }
            }
#pragma unroll
            for (int i = 0; i < WLD; ++i)
                if (wrow0 + RPP * i < BN)
                    buf_load16_lds(rsW, Wb + (wrow0 + RPP * i) * BK, fw_off[i], (unsigned)k0 * 4u);
            if constexpr (AMODE == 1) {
                if (p.slab) {
                    f_ci0 += BK;
                    if ((f_ci0 & (p.slab - 1)) == 0) {
                        f_ci0 -= p.slab; ++f_tap;
                        if (++f_kw == p.KW) { f_kw = 0; ++f_kh; }
                        if (f_tap == p.KH * p.KW) { f_tap = 0; f_kh = 0; f_kw = 0; f_ci0 += p.slab; }
                    }
                } else {
                    f_ci0 += BK;
                    if (f_ci0 == p.Cin) {
                        f_ci0 = 0; ++f_tap;
                        if (++f_kw == p.KW) { f_kw = 0; ++f_kh; }
                    }
                }
            }
        }
    };

    auto load_a = [&](int k0) {
        if constexpr (FAST && AMODE == 0) {
#pragma unroll
            for (int i = 0; i < ALD; ++i) a_reg[i] = buf_load16(rsA, fa_off[i], (unsigned)k0 * 4u);
        } else if constexpr (FAST && AMODE == 1) {
            // chunk k0 lies inside tap (f_kh, f_kw) at channel f_ci0 (Cin % BK == 0)
            const unsigned delta = ((unsigned)(f_kh * p.W + f_kw) * (unsigned)p.Cin + (unsigned)f_ci0) * 4u;
            const unsigned bit = 1u << f_tap;
#pragma unroll
            for (int i = 0; i < ALD; ++i)
                a_reg[i] = buf_load16(rsA, (fa_mask[i] & bit) ? fa_off[i] + delta : OOR, 0u);
            if (p.slab) {
                // the KH*KW taps of one channel slab are consecutive K chunks: the 9 re-reads of
                // an input pixel happen back to back and are served by L1/L2 instead of the fabric
                f_ci0 += BK;
                if ((f_ci0 & (p.slab - 1)) == 0) {
                    f_ci0 -= p.slab; ++f_tap;
                    if (++f_kw == p.KW) { f_kw = 0; ++f_kh; }
                    if (f_tap == p.KH * p.KW) { f_tap = 0; f_kh = 0; f_kw = 0; f_ci0 += p.slab; }
                }
            } else {
                f_ci0 += BK;
                if (f_ci0 == p.Cin) {
                    f_ci0 = 0; ++f_tap;
                    if (++f_kw == p.KW) { f_kw = 0; ++f_kh; }
                }
            }
        } else if constexpr (AMODE == 0) {
            const int k = k0 + lc * 4;
#pragma unroll
            for (int i = 0; i < ALD; ++i) {
                a_reg[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (a_base[i] >= 0 && k < p.K) a_reg[i] = *(const f32x4*)(p.a + a_base[i] + k);
            }
        } else if constexpr (AMODE == 1) {
            const int k = k0 + lc * 4;
            int tap, ci;
            if (p.slab) {
                const int cell = k / p.slab, taps = p.KH * p.KW;
                tap = cell % taps; ci = (cell / taps) * p.slab + (k & (p.slab - 1));
            } else {
                tap = k / p.Cin; ci = k - tap * p.Cin;
            }
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
            const bool kin = k < p.K;
#pragma unroll
            for (int i = 0; i < ALD; ++i) {
                const int hi = a_hi0[i] + kh, wi = a_wi0[i] + kw;
                a_reg[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (kin && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W)
                    a_reg[i] = *(const f32x4*)(p.a + a_base[i] + ((long long)hi * p.W + wi) * p.Cin + ci);
            }
        } else {
            const int mq = tid % AQ;
#pragma unroll
            for (int i = 0; i < ALD; ++i) {
                const int kk = tid / AQ + (256 / AQ) * i;
                const int k = k0 + kk, m = m0 + mq * 4;
                a_reg[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (kk < BK && k < p.K && m < p.M) a_reg[i] = *(const f32x4*)(p.a + (long long)k * p.lda + m);
            }
        }
    };
    auto load_w = [&](int k0) {
        if constexpr (FAST) {
#pragma unroll
            for (int i = 0; i < WLD; ++i) w_reg[i] = buf_load16(rsW, fw_off[i], (unsigned)k0 * 4u);
        } else if constexpr (WMODE == 0) {
            const int k = k0 + lc * 4;
#pragma unroll
            for (int i = 0; i < WLD; ++i) {
                const int n = n0 + lr + RPP * i;
                w_reg[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (lr + RPP * i < BN && n < p.N && k < p.K) w_reg[i] = *(const f32x4*)(p.w + (long long)n * p.ldw + k);
            }
        } else {
            const int nq = tid % WQ;
#pragma unroll
            for (int i = 0; i < WLD; ++i) {
                const int kk = tid / WQ + (256 / WQ) * i;
                const int k = k0 + kk, n = n0 + nq * 4;
                w_reg[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (kk < BK && k < p.K && n < p.N) w_reg[i] = *(const f32x4*)(p.w + (long long)k * p.ldw + n);
            }
        }
    };
    auto store_lds = [&](int buf) {
        float* Ab = As + buf * BM * BK;
        float* Wb = Ws + buf * BN * BK;
        if constexpr (AMODE != 2) {
#pragma unroll
            for (int i = 0; i < ALD; ++i)
                if (lr + RPP * i < BM) *(f32x4*)(Ab + lds_off<BK>(lr + RPP * i, lc)) = a_reg[i];
        } else {
            const int mq = tid % AQ;
#pragma unroll
            for (int i = 0; i < ALD; ++i) {
                const int kk = tid / AQ + (256 / AQ) * i;
                if (kk < BK)
#pragma unroll
                for (int j = 0; j < 4; ++j) Ab[lds_off<BK>(mq * 4 + j, kk >> 2) + (kk & 3)] = a_reg[i][j];
            }
        }
        if constexpr (WMODE == 0) {
#pragma unroll
            for (int i = 0; i < WLD; ++i)
                if (lr + RPP * i < BN) *(f32x4*)(Wb + lds_off<BK>(lr + RPP * i, lc)) = w_reg[i];
        } else {
            const int nq = tid % WQ;
#pragma unroll
            for (int i = 0; i < WLD; ++i) {
                const int kk = tid / WQ + (256 / WQ) * i;
                if (kk < BK)
#pragma unroll
                for (int j = 0; j < 4; ++j) Wb[lds_off<BK>(nq * 4 + j, kk >> 2) + (kk & 3)] = w_reg[i][j];
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fr = lane & 31, fh = lane >> 5, fsw = lds_swz<BK>(fr);

    if constexpr (FIXUP) {
        const int nk = p.sk_nk ? p.sk_nk : (p.K + BK - 1) / BK;
        const long long U = (long long)p.n_tiles * nk, t = tile;
        for (int c = fix_c0; c < p.sk_blocks; ++c) {
            const long long bc = U * c / p.sk_blocks;
            if (bc >= (t + 1) * nk) break;
            if (U * (c + 1) / p.sk_blocks == bc) continue;   // empty share: wrote nothing
            // slot 0 = a share that starts inside the tile, slot 1 = one that started earlier
            const float* part = p.sk_ws + ((size_t)c * 2 + (bc >= t * nk ? 0 : 1)) * (size_t)(BM * BN);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += part[((i * TN + j) * 16 + r) * 256 + tid];
        }
    } else {
    __syncthreads();                      // a previous tile of this workgroup may still use the LDS
    if constexpr (FAST && DMA) {
        dma_chunk(kb * BK, 0);
    } else {
        load_a(kb * BK); load_w(kb * BK);
        store_lds(0);
    }
    __syncthreads();
    }

    for (int kc = kb; !FIXUP && kc < ke; ++kc) {
        const int buf = (kc - kb) & 1;
        if (kc + 1 < ke) {
            if constexpr (FAST && DMA) dma_chunk((kc + 1) * BK, buf ^ 1);
            else { load_a((kc + 1) * BK); load_w((kc + 1) * BK); }
        }
        const float* Ab = As + buf * BM * BK;
        const float* Wb = Ws + buf * BN * BK;
#pragma unroll
        for (int g = 0; g < BK / 8; ++g) {
            const int ch = ((2 * g + fh) ^ fsw) << 2;
            f32x4 af[TM], wf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *(const f32x4*)(Ab + (wm0 + i * 32 + fr) * BK + ch);
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[j] = *(const f32x4*)(Wb + (wn0 + j * 32 + fr) * BK + ch);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], wf[j][s], acc[i][j], 0, 0, 0);
        }
        if constexpr (!(FAST && DMA)) {
            if (kc + 1 < ke) store_lds(buf ^ 1);
        }
        __syncthreads();
    }

    if (partial) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) partial[((i * TN + j) * 16 + r) * 256 + tid] = acc[i][j][r];
        return;
    }
    const float acc_scale = igemm_acc_scale(p);   // 1 unless this is the fix-up of an fp16-pair launch
    {
constexpr bool EPI_DIRECT = false;        // fp32-input kernels: small register budgets, small share of the time
#include "igemm_epilogue.inc"
    }
}

// ---------------------------------------------------------------------------------------------
// Split-precision tile ("x3"): every fp32 operand value is the exact sum of three bf16 values
// (hi + mid + lo, 8 + 8 + 8 mantissa bits); a product a*b is accumulated as the six bf16 x bf16
// partial products down to 2^-16 relative size (hh, hm, mh, mm, hl, lh -- each exact in the
// fp32 accumulator) on v_mfma_f32_32x32x16_bf16.  Six bf16 MFMAs cover 16 k values in 6 x 32
// cycles where the fp32-input MFMA needs 8 x 64: 2.67x the fp32-MFMA rate at fp32-level
// accuracy (the dropped terms are <= 2^-24 relative, the size of an fp32 rounding).
// A (activations) stays fp32 in HBM and is split by the loader on its way into LDS; W comes
// pre-split (dbmm_split_weight_planes).  LDS per stage: 3 planes x (BM + BN) rows x 16 bf16,
// 16-B chunk index XOR (row >> 3) & 1 -> conflict-free ds_read_b128 fragment reads.
// ---------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split3(float x, u16& hi, u16& mid, u16& lo) {
    const __bf16 h = (__bf16)x;
    const float r1 = x - (float)h;
    const __bf16 m = (__bf16)r1;
    const float r2 = r1 - (float)m;
    const __bf16 l = (__bf16)r2;
    hi = __builtin_bit_cast(u16, h); mid = __builtin_bit_cast(u16, m); lo = __builtin_bit_cast(u16, l);
}

// fp16-pair variant (NP = 2): with per-tensor power-of-two scaling an fp32 value is hi + lo of
// two fp16 values to 2^-22 relative (abs floor 2^-39 of the tensor maximum), so THREE products
// (hl, lh, hh) on v_mfma_f32_32x32x16_f16 reach the accuracy the six bf16 products do -- half
// the MFMA work, a third less LDS traffic and half the split arithmetic.  It needs an upper
// bound of max|A| on the device (p.a_absmax, written by the producing launch's epilogue).
__device__ __forceinline__ void split2h(float xs, u16& hi, u16& lo) {
    const _Float16 h = (_Float16)xs;
    const _Float16 l = (_Float16)(xs - (float)h);
    hi = __builtin_bit_cast(u16, h); lo = __builtin_bit_cast(u16, l);
}

// (hi, lo) fp16 pairs of x0 * sc and x1 * sc, packed {x0 | x1 << 16}, on v_fma_mix{lo,hi}_f16
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

// instruction-order fences of the split kernels' K loop (see kloop): X3_SCHED 0 = none (compiler's order),
// 1 = loads first, all MFMAs, then the split; 2 = loads first, the split interleaves with the second half of the MFMAs
#ifndef X3_SCHED
#define X3_SCHED 2
#endif
#if X3_SCHED == 0
#define X3_FENCE_TOP()
#define X3_FENCE_MID()
#define X3_SPLIT_KS (BK / 16)
#elif X3_SCHED == 1
#define X3_FENCE_TOP() __builtin_amdgcn_sched_barrier(0)
#define X3_FENCE_MID() __builtin_amdgcn_sched_barrier(0)
#define X3_SPLIT_KS (BK / 16)
#else
#define X3_FENCE_TOP() __builtin_amdgcn_sched_barrier(0)
#define X3_FENCE_MID() __builtin_amdgcn_sched_barrier(0)
#define X3_SPLIT_KS (BK / 32)
#endif

// timing ablations for developer builds (-DX3_ABL=bits; results are wrong by construction): 1 no MFMAs,
// 2 activation loads return zeros without touching memory, 4 same for the weight loads, 8 no split / LDS store of A
#ifndef X3_ABL
#define X3_ABL 0
#endif

template <int BM, int BN, int NP, int NW, int BK>
struct GeoX3 {   // LDS floats for the split path (NP / NW 16-bit planes of A / W, two stages) vs the epilogue staging
    static constexpr int STAGE_U16 = (NP * BM + NW * BN) * BK;
    static constexpr int TILE_FLOATS = 2 * STAGE_U16 / 2;
};

// LDS rows hold BK 16-bit values (32 or 64 B); XOR of the 16-B chunk index with row bits keeps
// the 16 lanes of a ds_read_b128 group on distinct bank quads (same geometry as lds_swz above)
template <int BK>
__device__ __forceinline__ int x3_swz(int row) { return BK == 16 ? ((row >> 3) & 1) : ((row >> 2) & 3); }

template <int BM, int BN, int WAVES_M, int WAVES_N, int AMODE, int NP, int NW, int BK, int TWO = 0, int EPID = 1>
__device__ __forceinline__ void igemm_tile_x3(const IgemmP& p, float* lds, int tile, int kb, int ke, float* partial) {
    static_assert(BK == 16 || BK == 32, "BK");
    static_assert((NP == 3 && NW == 3) || (NP == 2 && (NW == 2 || NW == 1)), "planes");
    static_assert(!TWO || (AMODE == 0 && NP == 2 && NW == 1 && BK == 32), "dual-source variant: plain GEMM, one weight plane");
    using G = Geo<BM, BN, WAVES_M, WAVES_N, BK>;
    constexpr int TM = G::TM, TN = G::TN, WTN = G::WTN, LROW = G::LROW;
    constexpr int QPR = BK / 4, RPA = 256 / QPR;   // A: k-quads (4 fp32) per row, rows per pass of the 256 threads
    constexpr int ALD = BM / RPA;                  // float4 loads per thread per chunk
    constexpr int CPW = BK / 8, RPW = 256 / CPW;   // W: 16-B chunks (8 halves) per plane row, rows per pass
    constexpr int WLD = (BN + RPW - 1) / RPW;      // b128 loads per thread per plane per chunk
    constexpr int STAGE = GeoX3<BM, BN, NP, NW, BK>::STAGE_U16;
    u16* Ap = (u16*)lds;                           // per stage: [NP][BM][BK] then [NW][BN][BK]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
    const int wm0 = (wave / WAVES_N) * (TM * 32), wn0 = (wave % WAVES_N) * (TN * 32);
    const int lc = tid & (QPR - 1), lr = tid / QPR;   // A: k-quad lc of rows lr + RPA*i
    const int wc = tid & (CPW - 1), wr = tid / CPW;   // W: chunk wc of rows wr + RPW*j, all planes

    unsigned fa_off[ALD], fa_mask[ALD], fw_off[WLD];
    int f_ci0 = 0, f_kh = 0, f_kw = 0, f_tap = 0;
    // A descriptor rebased to the tile (see a_desc): row-major -> the tile's first row (pool2: the first window's
    // corner pixel, window corners grow with the pooled index); conv gather -> the image of the tile's first row
    int fr0 = 0, fn0 = 0;
    if constexpr (AMODE == 0) fr0 = p.pool2 ? pool2_base_pixel(p, m0 >> 2) : m0;
    else fn0 = p.pool2 ? (m0 >> 2) / ((p.Ho >> 1) * (p.Wo >> 1)) : m0 / (p.Ho * p.Wo);
    const long long a_shift = AMODE == 0 ? (long long)fr0 * p.lda * 4 : (long long)fn0 * p.H * p.W * p.Cin * 4;
    __amdgpu_buffer_rsrc_t rsA = a_desc(p.a, p.a_total, a_shift);
    __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)(NP == 3 ? p.w3 : p.wh), 0,
                                                                   (int)(NP == 3 ? p.w3_bytes : p.wh_bytes), 0x00020000);
    // zero-extent twins: a load through them returns zeros without touching memory.  The chunk
    // loads past the end of the K range go through these instead of being branched around -- a
    // conditional load makes the compiler's vmcnt bookkeeping assume the worst path and wait for
    // the NEWEST loads before the split, which silently turned the two-chunk prefetch into one.
    __amdgpu_buffer_rsrc_t rsA0 = a_desc(p.a, p.a_total, a_shift, true);
    __amdgpu_buffer_rsrc_t rsW0 = __builtin_amdgcn_make_buffer_rsrc((void*)(NP == 3 ? p.w3 : p.wh), 0, 0, 0x00020000);
    float a_sc = 1.f;
    if constexpr (NP == 2) a_sc = pow2f(a_scale_exp(p.a_absmax));
    const unsigned plane_bytes = (unsigned)p.N * (unsigned)p.ldw * 2u;
#pragma unroll
    for (int j = 0; j < WLD; ++j) {
        const int row = wr + RPW * j;
        fw_off[j] = (row < BN && n0 + row < p.N) ? ((unsigned)(n0 + row) * (unsigned)p.ldw + wc * 8u) * 2u : OOR;
    }
    // TWO: switch the loader to the second operand pair (a2, wh2) and back
    auto use_second = [&]() {
        rsA = a_desc(p.a2, p.a2_total, (long long)m0 * p.lda2 * 4);
        rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.wh2, 0, (int)p.wh2_bytes, 0x00020000);
        rsA0 = a_desc(p.a2, p.a2_total, (long long)m0 * p.lda2 * 4, true);
        rsW0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.wh2, 0, 0, 0x00020000);
        a_sc = pow2f(a_scale_exp(p.a2_absmax));
#pragma unroll
        for (int j = 0; j < WLD; ++j) {
            const int row = wr + RPW * j;
            fw_off[j] = (row < BN && n0 + row < p.N) ? ((unsigned)(n0 + row) * (unsigned)p.ldw2 + wc * 8u) * 2u : OOR;
        }
#pragma unroll
        for (int i = 0; i < ALD; ++i) {
            const int m = m0 + lr + RPA * i;
            fa_off[i] = m < p.M ? (unsigned)(m - m0) * (unsigned)p.lda2 * 4u + lc * 16u : OOR;
        }
    };
    auto use_first = [&]() {
        rsA = a_desc(p.a, p.a_total, a_shift);
        rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.wh, 0, (int)p.wh_bytes, 0x00020000);
        rsA0 = a_desc(p.a, p.a_total, a_shift, true);
        rsW0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.wh, 0, 0, 0x00020000);
        a_sc = pow2f(a_scale_exp(p.a_absmax));
#pragma unroll
        for (int j = 0; j < WLD; ++j) {
            const int row = wr + RPW * j;
            fw_off[j] = (row < BN && n0 + row < p.N) ? ((unsigned)(n0 + row) * (unsigned)p.ldw + wc * 8u) * 2u : OOR;
        }
#pragma unroll
        for (int i = 0; i < ALD; ++i) {
            const int m = m0 + lr + RPA * i;
            fa_off[i] = m < p.M ? (unsigned)(m - m0) * (unsigned)p.lda * 4u + lc * 16u : OOR;
        }
    };
#pragma unroll
    for (int i = 0; i < ALD; ++i) {
        const int m = m0 + lr + RPA * i;
        const bool rv = m < p.M;
        if constexpr (AMODE == 0) {
            const int row = p.pool2 ? pool2_base_pixel(p, m >> 2) + ((m & 3) >> 1) * p.Wo + (m & 1) : m;
            fa_off[i] = rv ? (unsigned)(row - fr0) * (unsigned)p.lda * 4u + lc * 16u : OOR;
            fa_mask[i] = 0;
        } else {
            int n, ho, wo;
            if (p.pool2) {       // rows in 2x2-window-major order: m = (pooled pixel) * 4 + (dy * 2 + dx)
                const int q = m & 3, mp = m >> 2, wp2 = p.Wo >> 1, hwp = (p.Ho >> 1) * wp2;
                n = mp / hwp;
                const int rem = mp - n * hwp, hp = rem / wp2;
                ho = 2 * hp + (q >> 1); wo = 2 * (rem - hp * wp2) + (q & 1);
            } else {
                const int hw = p.Ho * p.Wo;
                n = m / hw;
                const int rem = m - n * hw;
                ho = rem / p.Wo; wo = rem - ho * p.Wo;
            }
            const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
            fa_off[i] = ((unsigned)((n - fn0) * p.H + hi0) * (unsigned)p.W + (unsigned)wi0) * (unsigned)p.Cin * 4u + lc * 16u;
            unsigned msk = 0;
            for (int kh = 0; kh < p.KH; ++kh)
                for (int kw = 0; kw < p.KW; ++kw)
                    if (hi0 + kh >= 0 && hi0 + kh < p.H && wi0 + kw >= 0 && wi0 + kw < p.W)
                        msk |= 1u << (kh * p.KW + kw);
            fa_mask[i] = rv ? msk : 0u;
        }
    }
    if constexpr (AMODE == 1) {
        if (p.slab == BK) {                                   // K = (cin/BK, kh, kw, BK): taps cycle fastest
            const int taps = p.KH * p.KW;
            f_tap = kb % taps; f_ci0 = (kb / taps) * BK;
        } else {
            const int k = kb * BK;
            f_tap = k / p.Cin; f_ci0 = k - f_tap * p.Cin;
        }
        f_kh = f_tap / p.KW; f_kw = f_tap - f_kh * p.KW;
    }

    // two register sets: the loads of chunk k+2 are issued while chunk k computes and chunk k+1
    // (already landed) is split into LDS -- one chunk of lead did not cover the L2 / Infinity-
    // Cache latency (ablation on the bf16 kernel: 31 % of the time)
    f32x4 a_r0[ALD], a_r1[ALD];
    u32x4 w_r0[NW * WLD], w_r1[NW * WLD];
    auto load_chunk = [&](int k0, f32x4 (&a_reg)[ALD], u32x4 (&w_reg)[NW * WLD], bool valid) {
        const __amdgpu_buffer_rsrc_t ra = (valid && !(X3_ABL & 2)) ? rsA : rsA0, rw = (valid && !(X3_ABL & 4)) ? rsW : rsW0;   // scalar selects
        if constexpr (AMODE == 0) {
#pragma unroll
            for (int i = 0; i < ALD; ++i) a_reg[i] = buf_load16(ra, fa_off[i], (unsigned)k0 * 4u);
        } else {
            const unsigned delta = ((unsigned)(f_kh * p.W + f_kw) * (unsigned)p.Cin + (unsigned)f_ci0) * 4u;
            const unsigned bit = 1u << f_tap;
#pragma unroll
            for (int i = 0; i < ALD; ++i)
                a_reg[i] = buf_load16(ra, (fa_mask[i] & bit) ? fa_off[i] + delta : OOR, 0u);
            if (p.slab == BK) {
                ++f_tap;
                if (++f_kw == p.KW) { f_kw = 0; ++f_kh; }
                if (f_tap == p.KH * p.KW) { f_tap = 0; f_kh = 0; f_kw = 0; f_ci0 += BK; }
            } else {
                f_ci0 += BK;
                if (f_ci0 == p.Cin) {
                    f_ci0 = 0; ++f_tap;
                    if (++f_kw == p.KW) { f_kw = 0; ++f_kh; }
                }
            }
        }
#pragma unroll
        for (int pl = 0; pl < NW; ++pl)
#pragma unroll
            for (int j = 0; j < WLD; ++j)
                w_reg[pl * WLD + j] = __builtin_amdgcn_raw_buffer_load_b128(rw, fw_off[j], pl * plane_bytes + (unsigned)k0 * 2u, 0);
    };
    auto store_chunk = [&](int stage, const f32x4 (&a_reg)[ALD], const u32x4 (&w_reg)[NW * WLD]) {
        u16* Ab = Ap + stage * STAGE;              // planes [NP][BM][BK]
        u16* Wb = Ab + NP * BM * BK;               // planes [NW][BN][BK]
#pragma unroll
        for (int i = 0; i < ALD; ++i) {
            const int row = lr + RPA * i;
            const int off = row * BK + ((((lc >> 1) ^ x3_swz<BK>(row))) << 3) + ((lc & 1) << 2);
            if constexpr (NP == 3) {
                u16 h[4], m[4], l[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) split3(a_reg[i][j], h[j], m[j], l[j]);
                *(u32x2*)(Ab + 0 * BM * BK + off) = (u32x2){(unsigned)h[0] | ((unsigned)h[1] << 16), (unsigned)h[2] | ((unsigned)h[3] << 16)};
                *(u32x2*)(Ab + 1 * BM * BK + off) = (u32x2){(unsigned)m[0] | ((unsigned)m[1] << 16), (unsigned)m[2] | ((unsigned)m[3] << 16)};
                *(u32x2*)(Ab + 2 * BM * BK + off) = (u32x2){(unsigned)l[0] | ((unsigned)l[1] << 16), (unsigned)l[2] | ((unsigned)l[3] << 16)};
            } else if constexpr (X3_ABL & 8) {
                asm volatile("" ::"v"(a_reg[i]));
            } else {
                // two packed fp16 results per pair of elements, one mixed-precision FMA each:
                // hi = f16(x * sc), lo = f16(x * sc - hi) (the fp32 difference is exact), written
                // straight into the low / high half of the packed register.  What the compiler made
                // of the same arithmetic in C was 4.4 VALU per element; this is 2.
                unsigned hp[2], lp[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) split2h_pair(a_reg[i][2 * j], a_reg[i][2 * j + 1], a_sc, hp[j], lp[j]);
                *(u32x2*)(Ab + 0 * BM * BK + off) = (u32x2){hp[0], hp[1]};
                *(u32x2*)(Ab + 1 * BM * BK + off) = (u32x2){lp[0], lp[1]};
            }
        }
#pragma unroll
        for (int j = 0; j < WLD; ++j) {
            const int row = wr + RPW * j;
            if (row < BN) {
                const int off = row * BK + ((wc ^ x3_swz<BK>(row)) << 3);
#pragma unroll
                for (int pl = 0; pl < NW; ++pl) *(u32x4*)(Wb + pl * BN * BK + off) = w_reg[pl * WLD + j];
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fr = lane & 31, fh = lane >> 5;

    auto compute = [&](int stage, int ks_lo = 0, int ks_hi = BK / 16) {
        const u16* Ab = Ap + stage * STAGE;
        const u16* Wb = Ab + NP * BM * BK;
#pragma unroll
        for (int ks = ks_lo; ks < ks_hi; ++ks) {
            // this lane's 8 k values of sub-step ks: chunk 2*ks + fh, swizzled (tile row offsets are
            // multiples of 32, so the swizzle depends on fr only)
            const int fch = ((2 * ks + fh) ^ x3_swz<BK>(fr)) << 3;
            u32x4 af[TM][NP], wf[TN][NW];
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                for (int i = 0; i < TM; ++i) af[i][pl] = *(const u32x4*)(Ab + pl * BM * BK + (wm0 + i * 32 + fr) * BK + fch);
#pragma unroll
            for (int pl = 0; pl < NW; ++pl)
#pragma unroll
                for (int j = 0; j < TN; ++j) wf[j][pl] = *(const u32x4*)(Wb + pl * BN * BK + (wn0 + j * 32 + fr) * BK + fch);
            // smallest partial products first: bf16 (h,l) (l,h) (m,m) (h,m) (m,h) (h,h); fp16 (h,l) (l,h) (h,h);
            // fp16 with W exact in one plane: (l,w) (h,w)
            constexpr int NQ = NP == 3 ? 6 : (NW == 2 ? 3 : 2);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                constexpr int PA3[6] = {0, 2, 1, 0, 1, 0}, PB3[6] = {2, 0, 1, 1, 0, 0};
                constexpr int PA2[3] = {0, 1, 0}, PB2[3] = {1, 0, 0};
                constexpr int PA1[2] = {1, 0}, PB1[2] = {0, 0};
                const int pa = NP == 3 ? PA3[q] : (NW == 2 ? PA2[q] : PA1[q]);
                const int pb = NP == 3 ? PB3[q] : (NW == 2 ? PB2[q] : PB1[q]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if constexpr (X3_ABL & 1) {
                            asm volatile("" ::"v"(af[i][pa]), "v"(wf[j][pb]));
                        } else if constexpr (NP == 3)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[i][pa]),
                                                                                __builtin_bit_cast(bf16x8, wf[j][pb]), acc[i][j], 0, 0, 0);
                        else
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i][pa]),
                                                                               __builtin_bit_cast(f16x8, wf[j][pb]), acc[i][j], 0, 0, 0);
                    }
            }
        }
    };

    auto kloop = [&](int kb, int ke) {
        __syncthreads();
        load_chunk(kb * BK, a_r0, w_r0, true);
        load_chunk((kb + 1) * BK, a_r1, w_r1, kb + 1 < ke);
        store_chunk(0, a_r0, w_r0);
        __syncthreads();

        // chunk kc lives in LDS stage (kc-kb)&1; set r1 holds chunk kc+1 on even steps, r0 on odd ones
        for (int kc = kb; kc < ke; kc += 2) {
            // (no branches inside: a chunk past the end arrives as zeros through the zero-extent
            //  descriptors and adds nothing; an odd chunk count costs one phantom chunk)
            // The scheduling fences keep the order written here.  Left alone the compiler sinks the
            // loads below half of the MFMAs (their destination registers double as fragment
            // registers) and hoists the split of the OTHER set to the top of the next half step:
            // issue -> use shrinks from two half steps to a quarter of one and every wave sits in
            // s_waitcnt vmcnt for an HBM round trip per chunk.
            load_chunk((kc + 2) * BK, a_r0, w_r0, kc + 2 < ke);
            X3_FENCE_TOP();
            compute(0, 0, X3_SPLIT_KS);
            X3_FENCE_MID();
            compute(0, X3_SPLIT_KS, BK / 16);
            store_chunk(1, a_r1, w_r1);
            __syncthreads();
            load_chunk((kc + 3) * BK, a_r1, w_r1, kc + 3 < ke);
            X3_FENCE_TOP();
            compute(1, 0, X3_SPLIT_KS);
            X3_FENCE_MID();
            compute(1, X3_SPLIT_KS, BK / 16);
            store_chunk(0, a_r0, w_r0);
            __syncthreads();
        }
    };
    if constexpr (!TWO) {
        kloop(kb, ke);
    } else {
        // chunks [0, nk2) belong to the second operand pair (the downsample branch), the rest to the
        // main pair.  A share that holds second-pair chunks rescales what it accumulated into the
        // main pair's units before continuing (or before it is written out as a stream-K partial),
        // so partial sums of different shares add up.
        const int nk2 = p.K2 / BK;
        if (kb < nk2) {
            use_second();
            kloop(kb, ke < nk2 ? ke : nk2);
            const float dyn = pow2f(a_scale_exp(p.a_absmax) - a_scale_exp(p.a2_absmax));
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn0 + j * 32 + (lane & 31);
                const float rt = (n < p.N ? p.ratio[n] : 0.f) * dyn;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] *= rt;
            }
            use_first();
        }
        if (ke > nk2) kloop((kb > nk2 ? kb : nk2) - nk2, ke - nk2);
    }

    if (partial) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) partial[((i * TN + j) * 16 + r) * 256 + tid] = acc[i][j][r];
        return;
    }
    const float acc_scale = NP == 2 ? igemm_acc_scale(p) : 1.f;
    {
constexpr bool EPI_DIRECT = NP == 2 && EPID && TM * TN <= 4;   // fp16-pair kernels with <= 64 accumulator registers, not the stream-K builds (register
                                                                 // budget); on the 128 x 256 GEMM tile it measured -1.7 % (ViT-B/32 parity): 28 registers spill around it
#include "igemm_epilogue.inc"
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int AMODE, int MINB, int SK, int NP, int NW, int BK, int TWO = 0>
__global__ __launch_bounds__(256, MINB) void igemm_x3_kernel(const IgemmP p) {
    using G = Geo<BM, BN, WAVES_M, WAVES_N, BK>;
    constexpr int LDSF = GeoX3<BM, BN, NP, NW, BK>::TILE_FLOATS > G::EPI_FLOATS ? GeoX3<BM, BN, NP, NW, BK>::TILE_FLOATS : G::EPI_FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[LDSF];
    const int nk = p.K / BK + (TWO ? p.K2 / BK : 0);
    if constexpr (!SK) {
        igemm_tile_x3<BM, BN, WAVES_M, WAVES_N, AMODE, NP, NW, BK, TWO, 1>(p, lds, xcd_remap(blockIdx.x, p.n_tiles), 0, nk, nullptr);
    } else {
        const long long U = (long long)p.n_tiles * nk;
        long long u = U * blockIdx.x / p.sk_blocks;
        const long long u1 = U * (blockIdx.x + 1) / p.sk_blocks;
        for (int seg = 0; u < u1; ++seg) {
            const int tile = (int)(u / nk), kb = (int)(u - (long long)tile * nk);
            const int ke = (int)((u1 - u < nk - kb) ? kb + (u1 - u) : nk);
            float* partial = (kb == 0 && ke == nk)
                                 ? nullptr
                                 : p.sk_ws + ((size_t)blockIdx.x * 2 + (seg ? 1 : 0)) * (size_t)(BM * BN);
            igemm_tile_x3<BM, BN, WAVES_M, WAVES_N, AMODE, NP, NW, BK, TWO, 0>(p, lds, tile, kb, ke, partial);
            u += ke - kb;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolution with the activation tile REUSED across the three kw taps
// ("halo" variant of the fp16-pair kernel with one exact weight plane; K order (cin/32, kh, kw, 32)).
// Ablation of igemm_tile_x3 showed the operand loads through the CU's vector-memory path to be the
// bound (16 KB of fp32 activations + 8 KB of weights per 128x128x32 chunk).  With the standard row
// order consecutive tile rows are consecutive pixels, so for one (32-channel slab, kh) the three
// kw taps read the SAME 130 pixels [m0 - 1, m0 + 128] shifted by 0 / 1 / 2 rows: they are fetched,
// split and stored to LDS once per group of three K steps instead of three times.  What a tap may
// not use (image borders: left / right column wrap, top / bottom rows, rows past M) is masked when
// the A fragment is READ: the lane's LDS address is redirected to a zero row.
// A work unit is 6 K steps (two (slab, kh) groups): two A stages alternate per group, two W stages
// per step, all indices static, no branches in the loop.  Needs Cin % 64 == 0.
// ---------------------------------------------------------------------------------------------
template <int BN, int POOL = 0>
struct GeoHalo {
    // LDS rows per A plane.  Standard row order: 136 (8704 B = 34 x 256): 130 used + a 256-B zero line in rows 132..135.
    // POOL (2x2-window-major rows, see igemm_tile_halo): two strips of 66 pixels at rows 0 and 72, zero line in rows 140..143.
    static constexpr int AR = POOL ? 144 : 136;
    static constexpr int ZROW = POOL ? 140 : 132;         // first row of the zero line
    static constexpr int STRIP = 72;                      // POOL: LDS row of the dy = 1 strip (72 keeps the fragment reads conflict-free)
    static constexpr int A_STAGE = 2 * AR * 32;           // u16: [2 planes][AR][32]
    static constexpr int W_STAGE = BN * 32;               // u16: [BN][32]
    static constexpr int TILE_FLOATS = (2 * A_STAGE + 2 * W_STAGE) / 2;
};

// POOL = 1: the conv feeds a fused 2x2 average pool, tile rows run window-major (row m = 4 * pooled pixel + dy * 2 + dx,
// like the pool2 mode of igemm_tile_x3).  The 128 rows are 32 windows = two image rows x 64 columns per (slab, kh)
// group, held as two strips of 66 pixels (left halo, 32 x (dx 0, dx 1), right halo): fragment row (w, dy, dx) of tap kw
// reads LDS row dy * 72 + 2 w + dx + kw.  Windows that wrap to the next pooled row / image load from wherever they
// live; a strip neighbour that is not the image neighbour is exactly a border tap and masked like every border tap.
template <int BN, int WAVES_M, int WAVES_N, int POOL = 0, int EPID = 1>
__device__ __forceinline__ void igemm_tile_halo(const IgemmP& p, float* lds, int tile, int ub, int ue, float* partial) {
    constexpr int BM = 128, BK = 32;
    using G = Geo<BM, BN, WAVES_M, WAVES_N, BK>;
    using H = GeoHalo<BN, POOL>;
    constexpr int TM = G::TM, TN = G::TN, WTN = G::WTN, LROW = G::LROW;
    // Masked taps read zeros from the 256-B line at rows 132..135, at the SAME offset modulo 256 B as the address they
    // replace: a redirected lane then sits on the bank quad it would have used anyway, so the redirect adds no bank
    // conflict (one shared zero row did: PMC showed 18 % conflict cycles on the 14x14 / 7x7 maps, where most 16-lane
    // groups hold a border pixel)
    constexpr int AR = H::AR, ZB = H::ZROW * 32;
    constexpr int ALD = 5;                                // 130 (POOL: 132) rows x 8 k-quads / 256 threads, passes of 32 rows
    constexpr int NLD = POOL ? 132 : 130;                 // rows the loader fills
    constexpr int RPW = 64, WLD = (BN + RPW - 1) / RPW;   // W: 4 chunks per row, 64 rows per pass
    u16* As = (u16*)lds;                                  // [2 stages][2 planes][AR][32]
    u16* Ws = As + 2 * H::A_STAGE;                        // [2 stages][BN][32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
    const int wm0 = (wave / WAVES_N) * (TM * 32), wn0 = (wave % WAVES_N) * (TN * 32);
    const int lc = tid & 7, lr = tid >> 3;                // A: k-quad lc of LDS rows lr + 32 i
    const int wc = tid & 3, wr = tid >> 2;                // W: chunk wc of rows wr + 64 j
    const int fr = lane & 31, fh = lane >> 5;

    // A descriptor rebased to the first pixel the tile can touch, m0 - 1 - W (see a_desc); for the tiles at the very
    // start of the tensor the base stays 0 and "negative" pixels wrap past the extent = zeros, as before
    const int pxf = POOL ? pool2_base_pixel(p, m0 >> 2) : m0;     // first output pixel of the tile
    const int px0 = pxf - 1 - p.W > 0 ? pxf - 1 - p.W : 0;
    const long long a_shift = (long long)px0 * p.Cin * 4;
    __amdgpu_buffer_rsrc_t rsA = a_desc(p.a, p.a_total, a_shift);
    __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.wh, 0, (int)p.wh_bytes, 0x00020000);
    __amdgpu_buffer_rsrc_t rsA0 = a_desc(p.a, p.a_total, a_shift, true);
    __amdgpu_buffer_rsrc_t rsW0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.wh, 0, 0, 0x00020000);
    const float a_sc = pow2f(a_scale_exp(p.a_absmax));

    // A: LDS row j holds pixel (m0 - 1 + j) + (kh - 1) * W of the current slab.  Byte offset of the
    // kh = 1 pixel; a negative pixel index wraps to >= 2^31 and is out of range = zeros.
    unsigned fa_off[ALD];
#pragma unroll
    for (int i = 0; i < ALD; ++i) {
        const int j = lr + 32 * i;
        if constexpr (POOL) {
            // loader row j = strip dy = j / 66, strip column c = j % 66: c = 0 / 65 are the halo pixels left of the first
            // and right of the last window, c = 1 + 2 w + dx the window pixels
            const int dy = j >= 66, c = j - 66 * dy;
            const int wl = c == 0 ? 0 : (c == 65 ? 31 : (c - 1) >> 1), dxo = c == 0 ? -1 : (c == 65 ? 2 : (c - 1) & 1);
            const int mp = (m0 >> 2) + wl;
            fa_off[i] = (j < NLD && 4 * mp < p.M)
                            ? (unsigned)((pool2_base_pixel(p, mp) + dy * p.W + dxo - px0) * p.Cin) * 4u + lc * 16u : OOR;
        } else {
            fa_off[i] = j < NLD ? (unsigned)((m0 - 1 + j - px0) * p.Cin) * 4u + lc * 16u : OOR;
        }
    }
    unsigned fw_off[WLD];
#pragma unroll
    for (int j = 0; j < WLD; ++j) {
        const int row = wr + RPW * j;
        fw_off[j] = (row < BN && n0 + row < p.N) ? ((unsigned)(n0 + row) * (unsigned)p.ldw + wc * 8u) * 2u : OOR;
    }
    // tap-validity masks of this lane's two FRAGMENT rows (output pixels), bit kh * 3 + kw
    unsigned fmask[TM];
    int faddr[TM][3][2];                                  // u16 index of (row + kw, k sub-step) in plane 0
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int r = wm0 + i * 32 + fr, m = m0 + r;
        unsigned msk = 0;
        if (m < p.M) {
            int ho, wo;
            if constexpr (POOL) {
                const int q = m & 3, mp = m >> 2, wp2 = p.Wo >> 1, hwp = (p.Ho >> 1) * wp2;
                const int rem = mp % hwp, hp = rem / wp2;
                ho = 2 * hp + (q >> 1); wo = 2 * (rem - hp * wp2) + (q & 1);
            } else {
                const int hw = p.Ho * p.Wo, n = m / hw, rem = m - n * hw;
                ho = rem / p.Wo; wo = rem - ho * p.Wo;
            }
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw)
                    if (ho + kh - 1 >= 0 && ho + kh - 1 < p.H && wo + kw - 1 >= 0 && wo + kw - 1 < p.W)
                        msk |= 1u << (kh * 3 + kw);
        }
        fmask[i] = msk;
        const int lrow = POOL ? ((r >> 1) & 1) * H::STRIP + (r >> 2) * 2 + (r & 1) : r;   // LDS row of the kw = 0 tap
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                faddr[i][kw][ks] = (lrow + kw) * 32 + (((2 * ks + fh) ^ x3_swz<32>(lrow + kw)) << 3);
    }
    int wfaddr[TN][2];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int r = wn0 + j * 32 + fr;
            wfaddr[j][ks] = r * 32 + (((2 * ks + fh) ^ x3_swz<32>(r)) << 3);
        }

    f32x4 a_r[ALD];
    u32x4 w_r0[WLD], w_r1[WLD];
    // group g = (slab, kh): pixel shift (kh - 1) * W, channel offset slab * 32
    auto load_a = [&](int g, bool valid) {
        const int slab = g / 3, kh = g - slab * 3;
        const unsigned delta = (unsigned)(((kh - 1) * p.W * p.Cin + slab * 32) * 4);
        const __amdgpu_buffer_rsrc_t ra = valid ? rsA : rsA0;
#pragma unroll
        for (int i = 0; i < ALD; ++i) a_r[i] = buf_load16(ra, fa_off[i] == OOR ? OOR : fa_off[i] + delta, 0u);
    };
    auto store_a = [&](int stage) {
        u16* Ab = As + stage * H::A_STAGE;
#pragma unroll
        for (int i = 0; i < ALD; ++i) {
            const int j = lr + 32 * i;
            const int row = (POOL && j >= 66) ? j + (H::STRIP - 66) : j;
            if (j < NLD) {
                unsigned hp[2], lp[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) split2h_pair(a_r[i][2 * j], a_r[i][2 * j + 1], a_sc, hp[j], lp[j]);
                const int off = row * 32 + ((((lc >> 1) ^ x3_swz<32>(row))) << 3) + ((lc & 1) << 2);
                *(u32x2*)(Ab + off) = (u32x2){hp[0], hp[1]};
                *(u32x2*)(Ab + AR * 32 + off) = (u32x2){lp[0], lp[1]};
            }
        }
    };
    auto load_w = [&](int t, u32x4 (&w_reg)[WLD], bool valid) {
        const __amdgpu_buffer_rsrc_t rw = valid ? rsW : rsW0;
#pragma unroll
        for (int j = 0; j < WLD; ++j) w_reg[j] = __builtin_amdgcn_raw_buffer_load_b128(rw, fw_off[j], (unsigned)t * 64u, 0);
    };
    auto store_w = [&](int stage, const u32x4 (&w_reg)[WLD]) {
        u16* Wb = Ws + stage * H::W_STAGE;
#pragma unroll
        for (int j = 0; j < WLD; ++j) {
            const int row = wr + RPW * j;
            if (row < BN) *(u32x4*)(Wb + row * 32 + ((wc ^ x3_swz<32>(row)) << 3)) = w_reg[j];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // tap = kh * 3 + kw of the step (bit of fmask); astage / wstage / kw are compile-time at every call
    auto compute = [&](int astage, int wstage, int kw, int tap) {
        const u16* Ab = As + astage * H::A_STAGE;
        const u16* Wb = Ws + wstage * H::W_STAGE;
        int aoff[TM][2];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const bool ok = (fmask[i] >> tap) & 1u;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) aoff[i][ks] = ok ? faddr[i][kw][ks] : ZB + (faddr[i][kw][ks] & 127);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 af[TM][2], wf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                af[i][0] = *(const u32x4*)(Ab + aoff[i][ks]);
                af[i][1] = *(const u32x4*)(Ab + AR * 32 + aoff[i][ks]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[j] = *(const u32x4*)(Wb + wfaddr[j][ks]);
#pragma unroll
            for (int pl = 1; pl >= 0; --pl)              // (lo, w) first, then (hi, w)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i][pl]),
                                                                           __builtin_bit_cast(f16x8, wf[j]), acc[i][j], 0, 0, 0);
        }
    };

    // (an odd number of groups -- Cin = 32 -- leaves half a phantom unit: loads past the end are zeros)
    const int nk_all = p.K / 32;
    const int t_end = ue * 6 < nk_all ? ue * 6 : nk_all, g_end = ue * 2 < nk_all / 3 ? ue * 2 : nk_all / 3;
    __syncthreads();                                      // a previous tile of this workgroup may still use the LDS
    // the 256-B zero line (rows ZROW .. ZROW + 3) of both planes of both A stages (never written by the loader)
    if (tid < 2 * 2 * 4 * 4) {                            // 2 stages x 2 planes x 4 rows x 4 chunks of 16 B
        const int ch = tid & 3, row = H::ZROW + ((tid >> 2) & 3), pl = (tid >> 4) & 1, st = tid >> 5;
        *(u32x4*)(As + st * H::A_STAGE + pl * AR * 32 + row * 32 + ch * 8) = (u32x4){0u, 0u, 0u, 0u};
    }
    load_a(2 * ub, true);
    load_w(6 * ub, w_r0, true);
    load_w(6 * ub + 1, w_r1, 6 * ub + 1 < t_end);
    store_a(0);
    store_w(0, w_r0);
    load_a(2 * ub + 1, 2 * ub + 1 < g_end);
    __syncthreads();

    for (int u = ub; u < ue; ++u) {
        const int t = 6 * u, g = 2 * u;
        const int tap0 = (g % 3) * 3, tap1 = ((g + 1) % 3) * 3;      // kh * 3 of the unit's two groups
        // step 0: group g, kw 0
        load_w(t + 2, w_r0, t + 2 < t_end);
        compute(0, 0, 0, tap0);
        store_w(1, w_r1);
        __syncthreads();
        // step 1: kw 1; the other A stage (last read in the previous unit) receives group g + 1
        load_w(t + 3, w_r1, t + 3 < t_end);
        compute(0, 1, 1, tap0 + 1);
        store_w(0, w_r0);
        store_a(1);
        load_a(g + 2, g + 2 < g_end);
        __syncthreads();
        // step 2: kw 2
        load_w(t + 4, w_r0, t + 4 < t_end);
        compute(0, 0, 2, tap0 + 2);
        store_w(1, w_r1);
        __syncthreads();
        // step 3: group g + 1, kw 0
        load_w(t + 5, w_r1, t + 5 < t_end);
        compute(1, 1, 0, tap1);
        store_w(0, w_r0);
        __syncthreads();
        // step 4: kw 1; A stage 0 (last read at step 2) receives group g + 2
        load_w(t + 6, w_r0, t + 6 < t_end);
        compute(1, 0, 1, tap1 + 1);
        store_w(1, w_r1);
        store_a(0);
        load_a(g + 3, g + 3 < g_end);
        __syncthreads();
        // step 5: kw 2
        load_w(t + 7, w_r1, t + 7 < t_end);
        compute(1, 1, 2, tap1 + 2);
        store_w(0, w_r0);
        __syncthreads();
    }

    if (partial) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) partial[((i * TN + j) * 16 + r) * 256 + tid] = acc[i][j][r];
        return;
    }
    const float acc_scale = igemm_acc_scale(p);
    {
constexpr bool EPI_DIRECT = EPID && TM * TN <= 4;
#include "igemm_epilogue.inc"
    }
}

template <int BN, int WAVES_M, int WAVES_N, int MINB, int SK, int POOL = 0>
__global__ __launch_bounds__(256, MINB) void igemm_halo_kernel(const IgemmP p) {
    using G = Geo<128, BN, WAVES_M, WAVES_N, 32>;
    constexpr int LDSF = GeoHalo<BN, POOL>::TILE_FLOATS > G::EPI_FLOATS ? GeoHalo<BN, POOL>::TILE_FLOATS : G::EPI_FLOATS;
    __shared__ __attribute__((aligned(16))) float lds[LDSF];
    const int nu = (p.K / 32 + 5) / 6;                    // units of 6 K steps per tile (stream-K: K % 192 == 0)
    if constexpr (!SK) {
        igemm_tile_halo<BN, WAVES_M, WAVES_N, POOL, 1>(p, lds, xcd_remap(blockIdx.x, p.n_tiles), 0, nu, nullptr);
    } else {
        const long long U = (long long)p.n_tiles * nu;
        long long u = U * blockIdx.x / p.sk_blocks;
        const long long u1 = U * (blockIdx.x + 1) / p.sk_blocks;
        for (int seg = 0; u < u1; ++seg) {
            const int tile = (int)(u / nu), ub = (int)(u - (long long)tile * nu);
            const int ue = (int)((u1 - u < nu - ub) ? ub + (u1 - u) : nu);
            float* partial = (ub == 0 && ue == nu)
                                 ? nullptr
                                 : p.sk_ws + ((size_t)blockIdx.x * 2 + (seg ? 1 : 0)) * (size_t)(128 * BN);
            igemm_tile_halo<BN, WAVES_M, WAVES_N, POOL, 0>(p, lds, tile, ub, ue, partial);
            u += ue - ub;
        }
    }
}

// fp32 [N][K] -> three bf16 planes [3][N][K] (hi, mid, lo)
__global__ __launch_bounds__(256) void split_planes_kernel(const float* __restrict__ w, u16* __restrict__ out,
                                                           long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        u16 h, m, l;
        split3(w[i], h, m, l);
        out[i] = h; out[n + i] = m; out[2 * n + i] = l;
    }
}

// fp32 [N][K] * sc -> two fp16 planes [2][N][K] (hi, lo)
__global__ __launch_bounds__(256) void split_planes_h_kernel(const float* __restrict__ w, u16* __restrict__ out,
                                                             long long n, float sc) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        u16 h, l;
        split2h(w[i] * sc, h, l);
        out[i] = h; out[n + i] = l;
    }
}

// MINB = resident workgroups per CU the register allocator must leave room for (hipcc sizes
// its VGPR budget from a 64-KB LDS model otherwise and drops the 128x128 tile to 1 block/CU).
template <int BM, int BN, int WAVES_M, int WAVES_N, int AMODE, int WMODE, int BK, int MINB, int FAST, int SK, int DMA>
__global__ __launch_bounds__(256, MINB) void igemm_f32_kernel(const IgemmP p_in) {
    static_assert(WAVES_M * WAVES_N == 4, "4 waves");
    using G = Geo<BM, BN, WAVES_M, WAVES_N, BK>;
    __shared__ __attribute__((aligned(16))) float lds[G::LDS_FLOATS];
    IgemmP p = p_in;
    if (blockIdx.y) {   // batched GEMM: independent problems along grid.y
        const long long b = blockIdx.y;
        p.a += b * p.sa; p.w += b * p.sw; p.c += b * p.sc;
        if (p.bias) p.bias += b * p.sbias;
        if (p.res) p.res += b * p.sres;
    }
    const int nk = (p.K + BK - 1) / BK;
    if constexpr (!SK) {
        igemm_tile<BM, BN, WAVES_M, WAVES_N, AMODE, WMODE, BK, FAST, 0, DMA>(p, lds, xcd_remap(blockIdx.x, p.n_tiles), 0,
                                                                        nk, nullptr, 0);
    } else {
        // stream-K: this workgroup's contiguous share [u, u1) of the n_tiles * nk (tile, chunk) units
        const long long U = (long long)p.n_tiles * nk;
        long long u = U * blockIdx.x / p.sk_blocks;
        const long long u1 = U * (blockIdx.x + 1) / p.sk_blocks;
        for (int seg = 0; u < u1; ++seg) {
            const int tile = (int)(u / nk), kb = (int)(u - (long long)tile * nk);
            const int ke = (int)((u1 - u < nk - kb) ? kb + (u1 - u) : nk);
            float* partial = (kb == 0 && ke == nk)
                                 ? nullptr
                                 : p.sk_ws + ((size_t)blockIdx.x * 2 + (seg ? 1 : 0)) * (size_t)(BM * BN);
            igemm_tile<BM, BN, WAVES_M, WAVES_N, AMODE, WMODE, BK, FAST, 0, DMA>(p, lds, tile, kb, ke, partial, 0);
            u += ke - kb;
        }
    }
}

// Second stream-K kernel: workgroup g (1 <= g < sk_blocks) owns share boundary g.  If that
// boundary cuts tile t and is the first boundary inside t, it sums every partial of t (the
// workgroups whose shares intersect t, recomputed from the same arithmetic) and runs the
// epilogue.  Slot 0 = a share that starts inside t, slot 1 = a share that started earlier.
template <int BM, int BN, int WAVES_M, int WAVES_N, int BK>
__global__ __launch_bounds__(256) void igemm_fixup_kernel(const IgemmP p) {
    using G = Geo<BM, BN, WAVES_M, WAVES_N, BK>;
    __shared__ __attribute__((aligned(16))) float lds[G::EPI_FLOATS];
    const int nk = p.sk_nk ? p.sk_nk : (p.K + BK - 1) / BK;
    const long long U = (long long)p.n_tiles * nk;
    const int g = blockIdx.x + 1, nb = p.sk_blocks;
    const long long b = U * g / nb;
    if (b % nk == 0) return;                                  // boundary on a tile edge: nothing split
    const long long t = b / nk;
    const long long bp = U * (g - 1) / nb;
    if (bp % nk != 0 && bp / nk == t) return;                 // an earlier boundary owns this tile
    igemm_tile<BM, BN, WAVES_M, WAVES_N, 0, 0, BK, 0, 1, 0>(p, lds, (int)t, 0, 0, nullptr, g - 1);
}

// FAST loader eligibility (see the header comment); `DBMM_IGEMM_FAST=0` forces the fallback.
template <int AMODE, int WMODE, int BK>
bool fast_ok(const IgemmP& p) {
    if (!dbmm_opt(OPT_IGEMM_FAST) || AMODE == 2 || WMODE != 0 || (p.K % BK) != 0) return false;
    if (AMODE == 1 && ((p.Cin % BK) != 0 || p.KH * p.KW > 32)) return false;
    if (AMODE == 1 && p.slab && BK > p.slab) return false;   // a chunk must not straddle taps
    return p.a_bytes != 0 && p.w_bytes != 0;
}

constexpr int NUM_CUS = 256;

// template arguments of this thread's most recent igemm launch (profiling aid: lets bench.py
// name the exact instantiation rocprofv3 reports); not used by any compute path
thread_local int g_last_cfg[11] = {0};

// resident workgroups per CU of the fp16-pair kernels: the 128x128 tile with two W planes is held
// to 2 by its 64 KB of LDS; with one W plane (48 KB, <= 168 VGPRs) and for the narrower tiles it is 3
inline int sk_mode_env() { return dbmm_opt(OPT_IGEMM_STREAMK); }
// The split-precision kernels keep 2-4 workgroups resident per CU, which already smooths the tile
// quantisation the one-round model predicts: a CU that runs a leftover tile alone runs it ~3x
// faster.  Same-box A/B at B = 512: stream-K was +4..5 % on two layer shapes and -6..-52 % on
// three (layer3/4 conv1: partial sums, extra pipeline fills, a fix-up launch).  It is kept for
// grids that cannot fill every resident slot once (small batches); DBMM_IGEMM_STREAMK=2 forces it.
inline bool sk_skip(int n_tiles, int resident_per_cu) {
    return sk_mode_env() != 2 && n_tiles >= NUM_CUS * resident_per_cu;
}

constexpr int X2_MINB(int BN, int NW) { return (BN == 128 && NW == 2) ? 2 : 3; }

template <int BM, int BN, int WM, int WN, int AMODE, int WMODE, int BK, int MINB>
int launch_cfg(IgemmP& p, hipStream_t s, int nbatch, void* ws, size_t ws_bytes) {
    const int tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    p.n_tiles = tiles_m * p.tiles_n;
    p.sk_blocks = 0; p.sk_ws = nullptr;
    // stream-K when whole tiles leave >7 % of the chip idle in the last round and the tile's K
    // loop is long enough to be worth cutting (DBMM_IGEMM_STREAMK=0 disables, =2 forces)
    const int sk_mode = sk_mode_env();
    const int nk = (p.K + BK - 1) / BK;
    const int grid_sk = NUM_CUS * MINB;
    const size_t need = (size_t)grid_sk * 2 * BM * BN * sizeof(float);
    if (sk_mode && nbatch == 1 && ws && ws_bytes >= need && dbmm_aligned16(ws) && nk >= 8) {
        const double per_cu = (double)p.n_tiles / NUM_CUS;
        const double eff = per_cu / (double)((p.n_tiles + NUM_CUS - 1) / NUM_CUS);
        if (sk_mode == 2 || (eff < 0.93 && (long long)p.n_tiles * nk >= 4LL * grid_sk)) {
            p.sk_blocks = grid_sk; p.sk_ws = (float*)ws;
        }
    }
    const dim3 grid(p.sk_blocks ? p.sk_blocks : p.n_tiles, nbatch);
    bool launched = false;
    {
        const int c[11] = {BM, BN, WM, WN, AMODE, WMODE, BK, MINB, 0, p.sk_blocks ? 1 : 0, 0};
        for (int i = 0; i < 11; ++i) g_last_cfg[i] = c[i];
    }
    if constexpr ((AMODE == 0 || AMODE == 1) && WMODE == 0 && BK == 16 && BM == 128 && (BN == 128 || BN == 64 || BN == 32)) {
        // split-precision path: needs pre-split weights and the FAST loader's preconditions
        const int x3_allow = dbmm_opt(OPT_IGEMM_X3), x2_allow = dbmm_opt(OPT_IGEMM_X2), x2_bk = dbmm_opt(OPT_IGEMM_X2_BK);
        // 32-deep K chunks (half the barriers) when a chunk never straddles a filter tap
        const bool bk32 = x2_bk == 32 && (p.K % 32) == 0 &&
                          (AMODE == 0 || ((p.Cin % 32) == 0 && (p.slab == 0 || p.slab == 32)));
        const bool slab_ok16 = AMODE == 0 || p.slab == 0 || p.slab == 16;     // the 16-deep split kernels: one tap per chunk
        const bool pool_ok = !p.pool2 || ((p.N & 3) == 0 && (p.ldc & 3) == 0 && (!p.res || (p.ldr & 3) == 0));
        if (x2_allow && p.wh && p.a_absmax && nbatch == 1 && fast_ok<AMODE, WMODE, BK>(p) && (p.nw == 2 || bk32) && (bk32 || slab_ok16) && pool_ok) {
            constexpr int MB2 = X2_MINB(BN, 2), MB1 = X2_MINB(BN, 1);
            const int MB = p.nw == 1 ? MB1 : MB2;
            const int nkx = p.K / (bk32 ? 32 : 16);
            if (p.sk_blocks) {
                p.sk_blocks = NUM_CUS * MB;
                if ((long long)p.n_tiles * nkx < 4LL * p.sk_blocks || nkx < 8 || sk_skip(p.n_tiles, MB)) p.sk_blocks = 0;
            }
            const dim3 g3(p.sk_blocks ? p.sk_blocks : p.n_tiles, 1);
            if (p.nw == 1) {          // W exact in one fp16 plane: two partial products (bk32 guaranteed above)
                if (p.sk_blocks)
                    hipLaunchKernelGGL((igemm_x3_kernel<BM, BN, WM, WN, AMODE, MB1, 1, 2, 1, 32>), g3, dim3(256), 0, s, p);
                else
                    hipLaunchKernelGGL((igemm_x3_kernel<BM, BN, WM, WN, AMODE, MB1, 0, 2, 1, 32>), g3, dim3(256), 0, s, p);
            } else if (bk32) {
                if (p.sk_blocks)
                    hipLaunchKernelGGL((igemm_x3_kernel<BM, BN, WM, WN, AMODE, MB2, 1, 2, 2, 32>), g3, dim3(256), 0, s, p);
                else
                    hipLaunchKernelGGL((igemm_x3_kernel<BM, BN, WM, WN, AMODE, MB2, 0, 2, 2, 32>), g3, dim3(256), 0, s, p);
            } else {
                if (p.sk_blocks)
                    hipLaunchKernelGGL((igemm_x3_kernel<BM, BN, WM, WN, AMODE, MB2, 1, 2, 2, 16>), g3, dim3(256), 0, s, p);
                else
                    hipLaunchKernelGGL((igemm_x3_kernel<BM, BN, WM, WN, AMODE, MB2, 0, 2, 2, 16>), g3, dim3(256), 0, s, p);
            }
            g_last_cfg[6] = bk32 ? 32 : 16; g_last_cfg[7] = MB; g_last_cfg[8] = 2; g_last_cfg[9] = p.sk_blocks ? 1 : 0;
            g_last_cfg[10] = p.nw;
            DBMM_CHECK_LAUNCH();
            if (p.sk_blocks) {   // the fix-up needs a_absmax / w_exp too: its epilogue rescales the sums
                if (bk32)
                    hipLaunchKernelGGL((igemm_fixup_kernel<BM, BN, WM, WN, 32>), dim3(p.sk_blocks - 1), dim3(256), 0, s, p);
                else
                    hipLaunchKernelGGL((igemm_fixup_kernel<BM, BN, WM, WN, 16>), dim3(p.sk_blocks - 1), dim3(256), 0, s, p);
                DBMM_CHECK_LAUNCH();
            }
            return DBMM_OK;
        }
        if (p.pool2) return DBMM_E_UNSUPPORTED;   // only the fp16-pair kernels walk rows window-major
        p.a_absmax = nullptr;   // every other kernel takes A unscaled
        if constexpr (BN != 32)
        if (x3_allow && p.w3 && nbatch == 1 && fast_ok<AMODE, WMODE, BK>(p) && slab_ok16) {
            constexpr int MB = BN == 128 ? 2 : 3;   // register budget for the two prefetch sets (no spills)
            if (p.sk_blocks) {   // the resident grid is sized for this kernel's occupancy
                p.sk_blocks = NUM_CUS * MB;
                if ((long long)p.n_tiles * (p.K / 16) < 4LL * p.sk_blocks) p.sk_blocks = 0;
            }
            const dim3 g3(p.sk_blocks ? p.sk_blocks : p.n_tiles, 1);
            if (p.sk_blocks)
                hipLaunchKernelGGL((igemm_x3_kernel<BM, BN, WM, WN, AMODE, MB, 1, 3, 3, 16>), g3, dim3(256), 0, s, p);
            else
                hipLaunchKernelGGL((igemm_x3_kernel<BM, BN, WM, WN, AMODE, MB, 0, 3, 3, 16>), g3, dim3(256), 0, s, p);
            g_last_cfg[7] = MB; g_last_cfg[8] = 3; g_last_cfg[9] = p.sk_blocks ? 1 : 0; g_last_cfg[10] = 3;
            DBMM_CHECK_LAUNCH();
            if (p.sk_blocks) {
                hipLaunchKernelGGL((igemm_fixup_kernel<BM, BN, WM, WN, BK>), dim3(p.sk_blocks - 1), dim3(256), 0, s, p);
                DBMM_CHECK_LAUNCH();
            }
            return DBMM_OK;
        }
    }

    if (p.pool2) return DBMM_E_UNSUPPORTED;
    p.a_absmax = nullptr;
    if constexpr (AMODE != 2 && WMODE == 0) {
        if (fast_ok<AMODE, WMODE, BK>(p)) {
            // operands reach LDS by `buffer_load ... lds` (the register-staged variant measured the same and is gone)
            g_last_cfg[8] = 1; g_last_cfg[10] = 1;
            if (p.sk_blocks)
                hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM, WN, AMODE, WMODE, BK, MINB, 1, 1, 1>), grid, dim3(256), 0, s, p);
            else
                hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM, WN, AMODE, WMODE, BK, MINB, 1, 0, 1>), grid, dim3(256), 0, s, p);
            launched = true;
        }
    }
    if (!launched) {
        if (p.sk_blocks)
            hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM, WN, AMODE, WMODE, BK, MINB, 0, 1, 0>), grid, dim3(256), 0, s, p);
        else
            hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM, WN, AMODE, WMODE, BK, MINB, 0, 0, 0>), grid, dim3(256), 0, s, p);
    }
    DBMM_CHECK_LAUNCH();
    if (p.sk_blocks) {
        hipLaunchKernelGGL((igemm_fixup_kernel<BM, BN, WM, WN, BK>), dim3(p.sk_blocks - 1), dim3(256), 0, s, p);
        DBMM_CHECK_LAUNCH();
    }
    return DBMM_OK;
}

// developer knob: DBMM_IGEMM_BK=32 forces the 64-KB / 2-workgroups-per-CU variants
inline int forced_bk() { return dbmm_opt(OPT_IGEMM_BK); }

// 3x3 / stride 1 / pad 1 convs on the halo kernel (see igemm_tile_halo).  Returns 1 when it launched.
template <int BN, int POOL = 0>
int launch_halo(IgemmP& p, hipStream_t s, void* ws, size_t ws_bytes, int* rc) {
    constexpr int BM = 128, MB = BN == 32 ? 4 : 3;   // 37 / 42 / 50 KB of LDS per workgroup (POOL: + 2 KB)
    const int tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    p.n_tiles = tiles_m * p.tiles_n;
    if (p.n_tiles < 192) return 0;                  // small problems: the 64x64 tiles fill the chip better
    const int nu = (p.K / 32 + 5) / 6;
    const bool whole_units = (p.K % 192) == 0;           // stream-K cuts at unit boundaries
    p.sk_blocks = 0; p.sk_ws = nullptr; p.sk_nk = 0;
    const int sk_mode = sk_mode_env();
    const int grid_sk = NUM_CUS * MB;
    const size_t need = (size_t)grid_sk * 2 * BM * BN * sizeof(float);
    if (!POOL && sk_mode && whole_units && ws && ws_bytes >= need && dbmm_aligned16(ws) && nu >= 4) {
        const double per_slot = (double)p.n_tiles / NUM_CUS;
        const double eff = per_slot / (double)((p.n_tiles + NUM_CUS - 1) / NUM_CUS);
        if (sk_mode == 2 || (eff < 0.93 && (long long)p.n_tiles * nu >= 4LL * grid_sk && !sk_skip(p.n_tiles, MB))) {
            p.sk_blocks = grid_sk; p.sk_ws = (float*)ws; p.sk_nk = nu;
        }
    }
    const dim3 g(p.sk_blocks ? p.sk_blocks : p.n_tiles, 1);
#ifndef HALO64_4X1
#define HALO64_4X1 1
#endif
    // 64-column tiles with the waves 4 x 1 like the 32-column ones: a wave owns 32 rows x all 64 columns, so no A fragment
    // is read (from LDS) by two waves and a k-step is 4 reads for 4 MFMAs instead of 6
    constexpr int WM = (BN == 32 || (BN == 64 && HALO64_4X1)) ? 4 : 2, WN = (BN == 32 || (BN == 64 && HALO64_4X1)) ? 1 : 2;
    if constexpr (POOL) {
        hipLaunchKernelGGL((igemm_halo_kernel<BN, WM, WN, MB, 0, 1>), g, dim3(256), 0, s, p);
    } else {
        if (p.sk_blocks)
            hipLaunchKernelGGL((igemm_halo_kernel<BN, WM, WN, MB, 1>), g, dim3(256), 0, s, p);
        else
            hipLaunchKernelGGL((igemm_halo_kernel<BN, WM, WN, MB, 0>), g, dim3(256), 0, s, p);
    }
    {
        const int c[11] = {BM, BN, WM, WN, 1, POOL, 32, MB, 4, p.sk_blocks ? 1 : 0, 1};   // [8] = 4: halo kernel, [5] = POOL
        for (int i = 0; i < 11; ++i) g_last_cfg[i] = c[i];
    }
    *rc = (int)hipGetLastError();
    if (*rc == 0 && p.sk_blocks) {
        hipLaunchKernelGGL((igemm_fixup_kernel<BM, BN, WM, WN, 32>), dim3(p.sk_blocks - 1), dim3(256), 0, s, p);
        *rc = (int)hipGetLastError();
    }
    return 1;
}

// DBMM_IGEMM_HALO_POOL: pooled (2x2-window-major) 3x3 convs on the halo kernel: 0 never, 1 where the 128 x 256
// per-tap tile does not apply (Cout % 256 != 0: layer 2's first block), 2 (default) every pooled 3x3 conv.
// RN50, B = 1024, same box: 32.44 k / 32.78 k / 33.34 k images/s for 0 / 1 / 2.
inline int halo_pool() { return dbmm_opt(OPT_IGEMM_HALO_POOL); }

template <int AMODE, int WMODE>
int launch_modes(IgemmP& p, hipStream_t s, int nbatch = 1, void* ws = nullptr, size_t wsb = 0) {
    // tile choice: widest N tile that N fills; drop to 64x64 when the 128-wide grid would
    // leave most of the 256 CUs idle (small-M projections).
    const long long t128 = (long long)((p.M + 127) / 128) * ((p.N + 127) / 128) * nbatch;
    if constexpr (AMODE == 1 && WMODE == 0) {
        // option igemm_halo = 0 selects the per-tap kernel
        const int use_halo = dbmm_opt(OPT_IGEMM_HALO);
        if (use_halo && nbatch == 1 && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.slab == 32 &&
            p.wh && p.nw == 1 && p.a_absmax && (p.Cin % 32) == 0 && p.a_bytes && p.wh_bytes && (p.N & 3) == 0 &&
            p.pool2 && p.N > 32 && (p.M & 3) == 0 && (p.ldc & 3) == 0 && (!p.res || (p.ldr & 3) == 0) &&
            (halo_pool() == 2 || (halo_pool() == 1 && (p.N % 256) != 0))) {
            int rc = 0;
            if (p.N <= 64 ? launch_halo<64, 1>(p, s, ws, wsb, &rc) : launch_halo<128, 1>(p, s, ws, wsb, &rc)) return rc;
        }
        if (use_halo && nbatch == 1 && p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1 && p.slab == 32 &&
            p.wh && p.nw == 1 && p.a_absmax && !p.pool2 && (p.Cin % 32) == 0 && p.a_bytes && p.wh_bytes &&
            (p.N & 3) == 0) {
            int rc = 0;
            if (p.N <= 32 ? launch_halo<32>(p, s, ws, wsb, &rc)
                          : (p.N <= 64 ? launch_halo<64>(p, s, ws, wsb, &rc) : launch_halo<128>(p, s, ws, wsb, &rc)))
                return rc;
        }
    }
    if (forced_bk() == 32) {
        if (p.N <= 32) return launch_cfg<128, 32, 4, 1, AMODE, WMODE, 32, 3>(p, s, nbatch, ws, wsb);
        if (p.N <= 64) return launch_cfg<128, 64, 2, 2, AMODE, WMODE, 32, 3>(p, s, nbatch, ws, wsb);
        if (t128 < 192) return launch_cfg<64, 64, 2, 2, AMODE, WMODE, 32, 4>(p, s, nbatch, ws, wsb);
        return launch_cfg<128, 128, 2, 2, AMODE, WMODE, 32, 2>(p, s, nbatch, ws, wsb);
    }
    // default: 16-deep K chunks -> 18-35 KB of LDS per workgroup, 4-6 resident workgroups per CU
    // (one wave of each on every SIMD) so a wave parked at a barrier, a global load or in its
    // epilogue always leaves others feeding the matrix pipe.
    // batched GEMMs with at most 32 rows per problem (attention pool: 32 heads x tokens per image):
    // a 32x128 tile wastes nothing along M where the 128-row tiles idle three quarters of the MFMA rows
    if (p.M <= 32 && nbatch > 1 && p.N > 32) return launch_cfg<32, 128, 1, 4, AMODE, WMODE, 16, 6>(p, s, nbatch, ws, wsb);
    if (p.N <= 32) return launch_cfg<128, 32, 4, 1, AMODE, WMODE, 16, 6>(p, s, nbatch, ws, wsb);
    if (p.N <= 64) return launch_cfg<128, 64, 2, 2, AMODE, WMODE, 16, 5>(p, s, nbatch, ws, wsb);
    if (t128 < 192) return launch_cfg<64, 64, 2, 2, AMODE, WMODE, 16, 6>(p, s, nbatch, ws, wsb);
    return launch_cfg<128, 128, 2, 2, AMODE, WMODE, 16, 4>(p, s, nbatch, ws, wsb);
}

// buffer extents for the FAST loader.  A may be any size (tiles rebase their descriptors, see a_desc); a weight
// operand of 2 GiB or more is not eligible (0)
inline void set_extents(IgemmP& p, long long a_bytes, long long w_bytes) {
    const long long lim = EXT_LIM;
    p.a_total = a_bytes > 0 ? a_bytes : 0;
    p.a_bytes = a_bytes > 0 ? (unsigned)(a_bytes < lim ? a_bytes : lim) : 0u;
    p.w_bytes = (w_bytes > 0 && w_bytes < lim) ? (unsigned)w_bytes : 0u;
}

// Optional operands of the split-precision paths.
//   w3: the weight as three bf16 planes (dbmm_split_weight_planes);
//   wh / w_exp / a_absmax: the weight * 2^w_exp as two fp16 planes (dbmm_split_weight_planes_f16)
//       and a device scalar >= max|A| -- both needed for the fp16-pair kernel;
//   absmax_out: device scalar that receives (atomic max) max|C| of this launch.
struct SplitArgs {
    const void* w3 = nullptr;
    const void* wh = nullptr;
    int w_exp = 0;
    int nw = 2;                        // planes in wh: 2 = hi + lo, 1 = the scaled weight is exact in fp16
    const float* a_absmax = nullptr;
    float* absmax_out = nullptr;
    const float* oscale = nullptr;     // per-output-channel scale of the accumulator (BatchNorm kept out of the weights)
    int pool = 0;                      // 2: average 2x2 output windows in the epilogue
    float* c_full = nullptr;           // pool == 2: also store the un-pooled output here
    bool no_8ph = false;               // the tail rows of a split 1x1 conv: 128 x 128 tiles only
};

inline void set_planes(IgemmP& p, const SplitArgs& sx, long long N, long long ldw) {
    const long long b3 = 3 * N * ldw * 2, b2 = (long long)sx.nw * N * ldw * 2;
    p.w3 = (sx.w3 && b3 < 0x7FFFFFF0LL && dbmm_aligned16(sx.w3)) ? (const unsigned short*)sx.w3 : nullptr;
    p.w3_bytes = p.w3 ? (unsigned)b3 : 0u;
    const bool h_ok = sx.wh && sx.a_absmax && (sx.nw == 1 || sx.nw == 2) && b2 < 0x7FFFFFF0LL && dbmm_aligned16(sx.wh) &&
                      sx.w_exp >= -40 && sx.w_exp <= 40;
    p.wh = h_ok ? (const unsigned short*)sx.wh : nullptr;
    p.wh_bytes = h_ok ? (unsigned)b2 : 0u;
    p.w_exp = h_ok ? sx.w_exp : 0;
    p.nw = h_ok ? sx.nw : 2;
    p.a_absmax = h_ok ? sx.a_absmax : nullptr;   // launch_cfg clears it again when another kernel runs
}

int gemm_impl(const float* a, int64_t lda, int trans_a, const float* w, int64_t ldw, int trans_w, const float* bias,
              const float* residual, int64_t ldr, float* c, int64_t ldc, int64_t M, int64_t N, int64_t K, float alpha,
              int act, void* ws, size_t wsb, void* stream, const SplitArgs& sx = SplitArgs()) {
    if (!a || !w || !c) return DBMM_E_ARG;
    if (M <= 0 || N <= 0 || K <= 0 || M > INT32_MAX || N > INT32_MAX || K > INT32_MAX) return DBMM_E_SHAPE;
    if (act < 0 || act > 2) return DBMM_E_ARG;
    if ((lda & 3) || (ldw & 3)) return DBMM_E_ALIGN;
    if (!dbmm_aligned16(a) || !dbmm_aligned16(w)) return DBMM_E_ALIGN;
    if (!trans_a && (K & 3)) return DBMM_E_SHAPE;   // 16-B chunks along K
    if (trans_a && (M & 3)) return DBMM_E_SHAPE;    // 16-B chunks along M
    if (!trans_w && (K & 3)) return DBMM_E_SHAPE;
    if (trans_w && (N & 3)) return DBMM_E_SHAPE;
    IgemmP p{};
    p.epi_direct = epi_direct_env();
    p.a = a; p.w = w; p.bias = bias; p.res = residual; p.c = c;
    p.lda = lda; p.ldw = ldw; p.ldr = ldr; p.ldc = ldc;
    p.M = (int)M; p.N = (int)N; p.K = (int)K; p.act = act; p.alpha = alpha;
    set_extents(p, trans_a ? 0 : ((M - 1) * lda + K) * 4, trans_w ? 0 : ((N - 1) * ldw + K) * 4);
    if (!trans_a && !trans_w) set_planes(p, sx, N, ldw);
    p.absmax_out = sx.absmax_out; p.oscale = sx.oscale;
    hipStream_t s = (hipStream_t)stream;
    if (!trans_a && !trans_w) {
        // wide GEMMs (transformer projections): a 128 x 256 tile halves the per-FLOP cost of fetching and splitting the fp32
        // activations (the A tile is shared by twice as many output columns); 128 accumulator registers, 2 workgroups per CU.
        // DBMM_IGEMM_BN256=0 disables.
        // The deep-pipelined 256 x 256 kernel (gemm_pair_8ph.hip).  Round 4, with the tiles of a short last round cut along K (same box,
        // tools/bench_gemm_pair.py, profiles/r04_ab_gemm_pair_8ph.log, TF fp32-equivalent): ViT-B/32 at 512 images qkv 353 -> 431, c_fc 349 -> 397,
        // c_proj 406 -> 477; ViT-L/14@336 at 256 images 449 -> 522 / 433 -> 470 / 505 -> 559, out_proj (N = K = 1024) 425 -> 455; text tower
        // qkv 310 -> 385.  It still loses where N is narrow AND K short -- the out projections of the 768- and 512-wide towers (296 -> 274,
        // 303 -> 262): 12 / 8 K tiles per 256 x 256 tile are prologue and epilogue.
        // gemm_8ph = 0 never, 1 (default) by that rule (K >= 1024 or N >= 1536), 2 wherever the kernel applies (the tests run all three).
        {
            const int m8 = dbmm_opt(OPT_GEMM_8PH);
            const bool pays = m8 == 2 || (m8 == 1 && (K >= 1024 || N >= 1536));
            if (pays && p.wh && p.nw == 1 && p.a_absmax && (N % 256) == 0 && (K % 64) == 0 && M >= 16384 && (lda & 3) == 0 &&
                (ldw & 7) == 0 && dbmm_aligned16(c) && (!residual || dbmm_aligned16(residual)) && 256 * (lda > ldc ? lda : ldc) * 4 < 0x7FFFFFF0LL &&
                p.wh_bytes) {
                const bool cut = dbmm_opt(OPT_TAIL_SPLIT) != 0;                       // a short last round's tiles cut along K (needs the workspace)
                const int rc = dbmm_gemm_pair_8ph_ws(a, lda, sx.a_absmax, p.wh, p.w_exp, ldw, sx.oscale, bias, residual, ldr, c, ldc, sx.absmax_out,
                                                     M, N, K, alpha, act, cut ? ws : nullptr, cut ? wsb : 0, stream);
                if (rc == DBMM_OK) {
                    const int cfg[11] = {256, 256, 2, 4, 0, 0, 32, 1, 6, 0, 1};      // [8] = 6: gemm_pair_8ph_kernel
                    for (int i = 0; i < 11; ++i) g_last_cfg[i] = cfg[i];
                }
                if (rc != DBMM_E_UNSUPPORTED) return rc;
            }
        }
        const int bn256 = dbmm_opt(OPT_IGEMM_BN256);
        if (bn256 && p.wh && p.nw == 1 && p.a_absmax && p.a_bytes && (K % 32) == 0 && N >= 768 && (N % 256) == 0 && M >= 8192 &&
            (ldc & 3) == 0 && (!residual || (ldr & 3) == 0)) {
            p.tiles_n = (int)(N / 256);
            p.n_tiles = (int)((M + 127) / 128) * p.tiles_n;
            hipLaunchKernelGGL((igemm_x3_kernel<128, 256, 2, 2, 0, 2, 0, 2, 1, 32>), dim3(p.n_tiles), dim3(256), 0, s, p);
            const int c[11] = {128, 256, 2, 2, 0, 0, 32, 2, 2, 0, 1};
            for (int i = 0; i < 11; ++i) g_last_cfg[i] = c[i];
            DBMM_CHECK_LAUNCH();
            return DBMM_OK;
        }
        return launch_modes<0, 0>(p, s, 1, ws, wsb);
    }
    if (!trans_a && trans_w) return launch_modes<0, 1>(p, s, 1, ws, wsb);
    if (trans_a && !trans_w) return launch_modes<2, 0>(p, s, 1, ws, wsb);
    return launch_modes<2, 1>(p, s, 1, ws, wsb);
}

int conv_impl(const float* x, const float* w, const float* bias, const float* residual, float* y, int64_t B, int64_t H,
              int64_t W, int64_t Cin, int64_t Cout, int64_t KH, int64_t KW, int64_t stride, int64_t pad, int act,
              int w_layout, void* ws, size_t wsb, void* stream, const SplitArgs& sx = SplitArgs()) {
    if (w_layout != DBMM_WL_TAP_MAJOR && w_layout != DBMM_WL_CHUNK_MAJOR && w_layout != DBMM_WL_CHUNK32_MAJOR)
        return DBMM_E_ARG;
    if (w_layout == DBMM_WL_CHUNK_MAJOR && (Cin & 15)) return DBMM_E_SHAPE;
    if (w_layout == DBMM_WL_CHUNK32_MAJOR && (Cin & 31)) return DBMM_E_SHAPE;
    if (!x || !w || !y) return DBMM_E_ARG;
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0)
        return DBMM_E_SHAPE;
    if (Cin & 3) return DBMM_E_SHAPE;
    if (act < 0 || act > 2) return DBMM_E_ARG;
    if (!dbmm_aligned16(x) || !dbmm_aligned16(w)) return DBMM_E_ALIGN;
    const int64_t Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return DBMM_E_SHAPE;
    const int64_t M = B * Ho * Wo, K = KH * KW * Cin;
    if (M > INT32_MAX || K > INT32_MAX) return DBMM_E_SHAPE;
    IgemmP p{};
    p.epi_direct = epi_direct_env();
    p.a = x; p.w = w; p.bias = bias; p.res = residual; p.c = y;
    p.lda = Cin; p.ldw = K; p.ldr = Cout; p.ldc = Cout;
    p.M = (int)M; p.N = (int)Cout; p.K = (int)K; p.act = act; p.alpha = 1.f;
    p.H = (int)H; p.W = (int)W; p.Cin = (int)Cin; p.Ho = (int)Ho; p.Wo = (int)Wo;
    p.KH = (int)KH; p.KW = (int)KW; p.stride = (int)stride; p.pad = (int)pad; p.wl = w_layout;
    p.slab = w_layout == DBMM_WL_CHUNK_MAJOR ? 16 : (w_layout == DBMM_WL_CHUNK32_MAJOR ? 32 : 0);
    set_extents(p, B * H * W * Cin * 4, Cout * K * 4);
    set_planes(p, sx, Cout, K);   // planes carry the same K order as `w`
    p.absmax_out = sx.absmax_out; p.oscale = sx.oscale;
    if (sx.pool != 0 && sx.pool != 2) return DBMM_E_ARG;
    if (sx.pool == 2) {
        if ((Ho & 1) || (Wo & 1) || M > (INT32_MAX >> 1)) return DBMM_E_UNSUPPORTED;
        if (sx.c_full && !dbmm_aligned16(sx.c_full)) return DBMM_E_ALIGN;
        p.pool2 = 1; p.c_full = sx.c_full;
    } else if (sx.c_full) {
        return DBMM_E_ARG;
    }
    hipStream_t s = (hipStream_t)stream;
    if (KH == 1 && KW == 1 && stride == 1 && pad == 0) {                                           // plain GEMM
        // conv3 + residual + ReLU with a short reduction into many channels (layer 3: 256 -> 1024): rows owned by one workgroup for a range
        // of 32-channel slabs, A fragments in registers, residual two slabs ahead (conv1x1_res_stream.hip)
        // (pool = 2 with the un-pooled output kept -- the last block of a stage -- is its window-major variant: y = the pooled tensor there)
        if (dbmm_opt(OPT_CONV1X1_RES_STREAM) && residual && act == DBMM_ACT_RELU && p.wh && p.nw == 1 && p.a_absmax && (!p.pool2 || p.c_full) &&
            sx.oscale && Cin == 256 && Cout >= 4 * Cin && M >= 131072 && p.wh_bytes) {   // (level with the tile kernel below that size and for K = 128)
            const int rc = p.pool2 ? dbmm_conv1x1_res_stream(x, sx.a_absmax, p.wh, p.w_exp, sx.oscale, bias, residual, p.c_full, y, sx.absmax_out, M, Ho,
                                                             Wo, Cin, Cout, stream)
                                   : dbmm_conv1x1_res_stream(x, sx.a_absmax, p.wh, p.w_exp, sx.oscale, bias, residual, y, nullptr, sx.absmax_out, M, 0, 0,
                                                             Cin, Cout, stream);
            if (rc == DBMM_OK) {
                const int cfg[11] = {(int)Cin, 32, 4, 1, 0, p.pool2, 32, 2, 9, 0, 1};   // [8] = 9: conv1x1_res_stream_kernel<K, POOL>
                for (int i = 0; i < 11; ++i) g_last_cfg[i] = cfg[i];
            }
            if (rc != DBMM_E_UNSUPPORTED) return rc;
        }
        // (the 128 x 256 tile of gemm_impl was measured here too, on RN50 layers 3-4 at B = 1024: neutral, not kept)
        // Also measured and not kept (round 3, profiles/r03_pair_stream_ab_layers.log): an LDS-DMA streaming variant (fp32 A tile by
        // DMA into a 3-slot ring, split into (hi, lo) when the fragments are read): 7 - 34 % SLOWER on every layer-2/3/4 shape
        // at B = 1024 -- the split then runs once per wave column and N tile instead of once per tile, on fp32 fragment reads.
        {
            // The deep-pipelined 256 x 256 kernel (gemm_pair_8ph.hip) where it measured ahead of the 128 x 128 tile on the RN50 shapes at
            // B = 1024 (same box, profiles/r04_ab_1x1_8ph.log): long K into <= 512 channels -- layer 3's first conv1 (M 802,816, K 512,
            // N 256) 0.635 -> 0.530 ms, layer 4's conv1s (K 1024 / 2048, N 512) 0.535 -> 0.467 and 0.262 -> 0.239 ms.  It loses on the
            // K <= 256 conv3 shapes (8 K tiles: the tile is prologue and epilogue; 0.399 -> 0.428 ms) and where 256-row tiles quantise
            // badly over the 256 CUs (layer 3's conv1: 784 tiles = 3.06 per CU, 0.287 -> 0.306 ms).
            // conv1x1_8ph = 0 never, 1 (default) by that rule, 2 wherever the kernel applies.
            const int m8 = sx.no_8ph ? 0 : dbmm_opt(OPT_CONV1X1_8PH);
            const long long tn8 = Cout / 256, mt8 = (M + 255) / 256, t8 = mt8 * tn8;
            // Tile quantisation (option tail_split): the tiles of a short last round (at most 128) are cut along K over the idle CUs inside
            // dbmm_gemm_pair_8ph_ws -- layer 3's conv1 (784 tiles = 3.06 rounds) then costs 3 rounds + 16 x 16 slices instead of 4 rounds.
            const long long rem8 = t8 % NUM_CUS;
            const bool cut = dbmm_opt(OPT_TAIL_SPLIT) && ws && t8 > NUM_CUS && rem8 != 0 && rem8 <= NUM_CUS / 2;
            const bool pays = m8 == 2 || (m8 == 1 && Cin >= 512 && Cout <= 512 && (Cout == 512 || t8 >= 8 * NUM_CUS || cut));
            if (pays && p.wh && p.nw == 1 && p.a_absmax && !p.pool2 && (Cout % 256) == 0 && (Cin % 64) == 0 && M >= 16384 && p.wh_bytes &&
                dbmm_aligned16(y) && (!residual || dbmm_aligned16(residual)) && 256 * (Cin > Cout ? Cin : Cout) * 4 < 0x7FFFFFF0LL) {
                const int rc = dbmm_gemm_pair_8ph_ws(x, Cin, sx.a_absmax, p.wh, p.w_exp, K, sx.oscale, bias, residual, Cout, y, Cout, sx.absmax_out,
                                                     M, Cout, K, 1.f, act, cut ? ws : nullptr, cut ? wsb : 0, stream);
                if (rc == DBMM_OK) {
                    const int cfg[11] = {256, 256, 2, 4, 0, 0, 32, 1, 6, 0, 1};      // [8] = 6: gemm_pair_8ph_kernel
                    for (int i = 0; i < 11; ++i) g_last_cfg[i] = cfg[i];
                }
                if (rc != DBMM_E_UNSUPPORTED) return rc;
            }
        }
        if (dbmm_opt(OPT_CONV1X1_BN256) && p.wh && p.nw == 1 && p.a_absmax && p.a_bytes && !p.pool2 && (K % 32) == 0 && (Cout % 256) == 0 &&
            M >= 8192) {
            p.tiles_n = (int)(Cout / 256);
            p.n_tiles = (int)((M + 127) / 128) * p.tiles_n;
            hipLaunchKernelGGL((igemm_x3_kernel<128, 256, 2, 2, 0, 2, 0, 2, 1, 32>), dim3(p.n_tiles), dim3(256), 0, s, p);
            const int c[11] = {128, 256, 2, 2, 0, 0, 32, 2, 2, 0, 1};
            for (int i = 0; i < 11; ++i) g_last_cfg[i] = c[i];
            DBMM_CHECK_LAUNCH();
            return DBMM_OK;
        }
        return launch_modes<0, 0>(p, s, 1, ws, wsb);
    }
    {
        // KxK convs that the halo kernel does not take (pooled 3x3 convs under DBMM_IGEMM_HALO_POOL < 2, strided or
        // larger windows) with >= 256 output channels: the 128 x 256 tile halves the per-FLOP cost of the gather + split
        // of the activations, which is what bounds the per-tap kernel.  DBMM_IGEMM_BN256_KXK=0 disables.
        const int bn256 = dbmm_opt(OPT_IGEMM_BN256_KXK);
        const bool halo_takes_it = dbmm_opt(OPT_IGEMM_HALO) && KH == 3 && KW == 3 && stride == 1 && pad == 1 &&
                                   (!p.pool2 || halo_pool() == 2);
        const bool pool_ok = !p.pool2 || ((Cout & 3) == 0 && (!residual || true));
        if (bn256 && !halo_takes_it && p.wh && p.nw == 1 && p.a_absmax && p.a_bytes && p.slab == 32 && (Cin % 32) == 0 && KH * KW <= 32 &&
            (Cout % 256) == 0 && M >= 8192 && pool_ok) {
            p.tiles_n = (int)(Cout / 256);
            p.n_tiles = (int)((M + 127) / 128) * p.tiles_n;
            hipLaunchKernelGGL((igemm_x3_kernel<128, 256, 2, 2, 1, 2, 0, 2, 1, 32>), dim3(p.n_tiles), dim3(256), 0, s, p);
            const int c[11] = {128, 256, 2, 2, 1, 0, 32, 2, 2, 0, 1};
            for (int i = 0; i < 11; ++i) g_last_cfg[i] = c[i];
            DBMM_CHECK_LAUNCH();
            return DBMM_OK;
        }
    }
    // 3x3 / stride 1 / pad 1 with Cout % 128 == 0 (layers 2 / 3 / 4): the eight-phase halo kernels (conv3x3_halo8.hip: 256 x 256 tiles, 256 x 128
    // where Cout % 256 != 0); option
    // tail_split lets it cut the tiles of a short last round along K.  Same-box A/B at B = 1024: profiles/r04_ab_halo8.log.
    // halo8 = 0 never, 1 (default) / 2 wherever the kernel applies.
    if (dbmm_opt(OPT_IGEMM_HALO) && dbmm_opt(OPT_HALO8) && KH == 3 && KW == 3 && stride == 1 && pad == 1 && p.slab == 32 && p.wh &&
        p.nw == 1 && p.a_absmax && (Cin % 64) == 0 && (Cout % 128) == 0 && p.wh_bytes && !residual && !p.c_full && M >= 16384 &&
        (act == DBMM_ACT_NONE || act == DBMM_ACT_RELU)) {
        const int rc = dbmm_conv3x3_halo8(x, sx.a_absmax, p.wh, p.w_exp, sx.oscale, bias, y, sx.absmax_out, B, H, W, Cin, Cout, act, p.pool2 ? 2 : 0,
                                          dbmm_opt(OPT_TAIL_SPLIT), ws, wsb, stream);
        if (rc == DBMM_OK) {
            const int c[11] = {256, (Cout % 256) ? 128 : 256, 4, 2, 1, p.pool2, 32, 1, 7, 0, 1};   // [8] = 7: conv3x3_halo8(n)_kernel, [5] = POOL
            for (int i = 0; i < 11; ++i) g_last_cfg[i] = c[i];
        }
        if (rc != DBMM_E_UNSUPPORTED) return rc;
    }
    return launch_modes<1, 0>(p, s, 1, ws, wsb);
}

}  // namespace

extern "C" void dbmm_debug_last_igemm(int* out11) {
    for (int i = 0; i < 11; ++i) out11[i] = g_last_cfg[i];
}

extern "C" size_t dbmm_workspace_bytes_igemm(void) {
    // stream-K partial accumulators: (256 CUs x 6 resident workgroups) x 2 slots x 128x128 fp32
    return (size_t)NUM_CUS * 6 * 2 * 128 * 128 * sizeof(float);
}

extern "C" int dbmm_gemm_bias_act(const float* a, int64_t lda, int trans_a, const float* w, int64_t ldw,
                                  int trans_w, const float* bias, const float* residual, int64_t ldr,
                                  float* c, int64_t ldc, int64_t M, int64_t N, int64_t K, float alpha,
                                  int act, void* stream) {
    return gemm_impl(a, lda, trans_a, w, ldw, trans_w, bias, residual, ldr, c, ldc, M, N, K, alpha, act, nullptr, 0,
                     stream);
}

extern "C" int dbmm_gemm_bias_act_ws(const float* a, int64_t lda, int trans_a, const float* w, int64_t ldw,
                                     int trans_w, const float* bias, const float* residual, int64_t ldr, float* c,
                                     int64_t ldc, int64_t M, int64_t N, int64_t K, float alpha, int act,
                                     void* workspace, size_t workspace_bytes, void* stream) {
    return gemm_impl(a, lda, trans_a, w, ldw, trans_w, bias, residual, ldr, c, ldc, M, N, K, alpha, act, workspace,
                     workspace_bytes, stream);
}

extern "C" int dbmm_gemm_batched(const float* a, int64_t lda, int64_t stride_a, int trans_a, const float* w,
                                 int64_t ldw, int64_t stride_w, int trans_w, const float* bias, int64_t stride_bias,
                                 float* c, int64_t ldc, int64_t stride_c, int64_t M, int64_t N, int64_t K,
                                 int64_t batch, float alpha, int act, void* stream) {
    if (!a || !w || !c) return DBMM_E_ARG;
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0 || batch > 65535 || M > INT32_MAX || N > INT32_MAX || K > INT32_MAX)
        return DBMM_E_SHAPE;
    if (act < 0 || act > 2) return DBMM_E_ARG;
    if ((lda & 3) || (ldw & 3) || (stride_a & 3) || (stride_w & 3)) return DBMM_E_ALIGN;
    if (!dbmm_aligned16(a) || !dbmm_aligned16(w)) return DBMM_E_ALIGN;
    if ((!trans_a && (K & 3)) || (trans_a && (M & 3)) || (!trans_w && (K & 3)) || (trans_w && (N & 3)))
        return DBMM_E_SHAPE;
    IgemmP p{};
    p.epi_direct = epi_direct_env();
    p.a = a; p.w = w; p.bias = bias; p.res = nullptr; p.c = c;
    p.lda = lda; p.ldw = ldw; p.ldr = 0; p.ldc = ldc;
    p.sa = stride_a; p.sw = stride_w; p.sbias = stride_bias; p.sc = stride_c;
    p.M = (int)M; p.N = (int)N; p.K = (int)K; p.act = act; p.alpha = alpha;
    set_extents(p, trans_a ? 0 : ((M - 1) * lda + K) * 4, trans_w ? 0 : ((N - 1) * ldw + K) * 4);
    hipStream_t s = (hipStream_t)stream;
    const int nb = (int)batch;
    if (!trans_a && !trans_w) return launch_modes<0, 0>(p, s, nb);
    if (!trans_a && trans_w) return launch_modes<0, 1>(p, s, nb);
    if (trans_a && !trans_w) return launch_modes<2, 0>(p, s, nb);
    return launch_modes<2, 1>(p, s, nb);
}

extern "C" int dbmm_conv_bn_act(const float* x, const float* w, const float* bias, const float* residual,
                                float* y, int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                                int64_t KH, int64_t KW, int64_t stride, int64_t pad, int act, void* stream) {
    return conv_impl(x, w, bias, residual, y, B, H, W, Cin, Cout, KH, KW, stride, pad, act, DBMM_WL_TAP_MAJOR, nullptr, 0,
                     stream);
}

extern "C" int dbmm_conv_bn_act_ws(const float* x, const float* w, const float* bias, const float* residual,
                                   float* y, int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t KH,
                                   int64_t KW, int64_t stride, int64_t pad, int act, int w_layout, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    return conv_impl(x, w, bias, residual, y, B, H, W, Cin, Cout, KH, KW, stride, pad, act, w_layout, workspace,
                     workspace_bytes, stream);
}

extern "C" size_t dbmm_split_planes_bytes(int64_t N, int64_t K) { return (size_t)(3 * N * K) * 2; }

extern "C" int dbmm_split_weight_planes(const float* w, void* planes, int64_t N, int64_t K, void* stream) {
    if (!w || !planes) return DBMM_E_ARG;
    if (N <= 0 || K <= 0) return DBMM_E_SHAPE;
    const long long n = (long long)N * K;
    const long long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0,
                       (hipStream_t)stream, w, (u16*)planes, n);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

extern "C" int dbmm_conv_bn_act_x3(const float* x, const float* w, const void* w_planes, const float* bias,
                                   const float* residual, float* y, int64_t B, int64_t H, int64_t W, int64_t Cin,
                                   int64_t Cout, int64_t KH, int64_t KW, int64_t stride, int64_t pad, int act,
                                   int w_layout, void* workspace, size_t workspace_bytes, void* stream) {
    SplitArgs sx; sx.w3 = w_planes;
    return conv_impl(x, w, bias, residual, y, B, H, W, Cin, Cout, KH, KW, stride, pad, act, w_layout, workspace,
                     workspace_bytes, stream, sx);
}

extern "C" size_t dbmm_split_planes_f16_bytes(int64_t N, int64_t K) { return (size_t)(2 * N * K) * 2; }

extern "C" int dbmm_split_weight_planes_f16(const float* w, void* planes, int64_t N, int64_t K, int w_exp, void* stream) {
    if (!w || !planes) return DBMM_E_ARG;
    if (N <= 0 || K <= 0 || w_exp < -40 || w_exp > 40) return DBMM_E_SHAPE;
    const long long n = (long long)N * K;
    const long long blocks = (n + 255) / 256;
    hipLaunchKernelGGL(split_planes_h_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0,
                       (hipStream_t)stream, w, (u16*)planes, n, ldexpf(1.f, w_exp));
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

extern "C" int dbmm_conv_bn_act_x2(const float* x, const float* x_absmax, const float* w, const void* w_planes_f16,
                                   int w_planes, int w_exp, const float* out_scale, const float* bias,
                                   const float* residual, float* y, float* y_full, float* y_absmax,
                                   int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t KH, int64_t KW,
                                   int64_t stride, int64_t pad, int act, int pool, int w_layout, void* workspace,
                                   size_t workspace_bytes, void* stream) {
    if (w_planes_f16 && w_planes != 1 && w_planes != 2) return DBMM_E_ARG;
    SplitArgs sx; sx.wh = w_planes_f16; sx.nw = w_planes_f16 ? w_planes : 2; sx.w_exp = w_exp; sx.a_absmax = x_absmax;
    sx.absmax_out = y_absmax; sx.oscale = out_scale; sx.pool = pool; sx.c_full = y_full;
    return conv_impl(x, w, bias, residual, y, B, H, W, Cin, Cout, KH, KW, stride, pad, act, w_layout, workspace,
                     workspace_bytes, stream, sx);
}

extern "C" int dbmm_gemm_bias_act_x3(const float* a, int64_t lda, const float* w, const void* w_planes, int64_t ldw,
                                     const float* bias, const float* residual, int64_t ldr, float* c, int64_t ldc,
                                     int64_t M, int64_t N, int64_t K, float alpha, int act, void* workspace,
                                     size_t workspace_bytes, void* stream) {
    SplitArgs sx; sx.w3 = w_planes;
    return gemm_impl(a, lda, 0, w, ldw, 0, bias, residual, ldr, c, ldc, M, N, K, alpha, act, workspace, workspace_bytes,
                     stream, sx);
}

extern "C" int dbmm_gemm_bias_act_x2(const float* a, int64_t lda, const float* a_absmax, const float* w,
                                     const void* w_planes_f16, int w_planes, int w_exp, int64_t ldw,
                                     const float* out_scale, const float* bias, const float* residual, int64_t ldr,
                                     float* c, int64_t ldc, float* c_absmax, int64_t M, int64_t N, int64_t K,
                                     float alpha, int act, void* workspace, size_t workspace_bytes, void* stream) {
    if (w_planes_f16 && w_planes != 1 && w_planes != 2) return DBMM_E_ARG;
    SplitArgs sx; sx.wh = w_planes_f16; sx.nw = w_planes_f16 ? w_planes : 2; sx.w_exp = w_exp; sx.a_absmax = a_absmax;
    sx.absmax_out = c_absmax; sx.oscale = out_scale;
    return gemm_impl(a, lda, 0, w, ldw, 0, bias, residual, ldr, c, ldc, M, N, K, alpha, act, workspace, workspace_bytes,
                     stream, sx);
}

// conv3 + downsample branch of a bottleneck as ONE GEMM launch (see dbmm.h)
extern "C" int dbmm_gemm_dual_bn_act_x2(const float* a, int64_t lda, const float* a_absmax, const void* w_plane_f16,
                                        int w_exp, int64_t ldw, int64_t K, const float* out_scale, const float* a2,
                                        int64_t lda2, const float* a2_absmax, const void* w2_plane_f16, int64_t ldw2,
                                        int64_t K2, const float* ratio, const float* bias, float* c, int64_t ldc,
                                        float* c_absmax, int64_t M, int64_t N, int act, void* workspace,
                                        size_t workspace_bytes, void* stream) {
    if (!a || !a2 || !a_absmax || !a2_absmax || !w_plane_f16 || !w2_plane_f16 || !ratio || !c) return DBMM_E_ARG;
    if (act < 0 || act > 2) return DBMM_E_ARG;
    if (M <= 0 || N <= 0 || K <= 0 || K2 <= 0 || M > INT32_MAX || N > INT32_MAX) return DBMM_E_SHAPE;
    if ((lda & 3) || (lda2 & 3) || (ldw & 7) || (ldw2 & 7) || (ldc & 3) || !dbmm_aligned16(a) || !dbmm_aligned16(a2) ||
        !dbmm_aligned16(w_plane_f16) || !dbmm_aligned16(w2_plane_f16) || !dbmm_aligned16(c))
        return DBMM_E_ALIGN;
    if ((K % 32) || (K2 % 32) || (N & 3) || w_exp < -40 || w_exp > 40) return DBMM_E_UNSUPPORTED;
    const long long ab = ((M - 1) * lda + K) * 4, ab2 = ((M - 1) * lda2 + K2) * 4, wb = ((N - 1) * ldw + K) * 2,
                    wb2 = ((N - 1) * ldw2 + K2) * 2, lim = 0x7FFFFFF0LL;
    if (wb >= lim || wb2 >= lim) return DBMM_E_UNSUPPORTED;      // (activations may exceed 2 GiB: tiles rebase, a_desc)
    // the eight-phase 256 x 256 kernel (gemm_pair_8ph.hip, TWO = 1).  dual_8ph = 0 never, 1 where it measured ahead, 2 wherever it applies.
    if (dbmm_opt(OPT_DUAL_8PH)) {
        const int rc = dbmm_gemm_dual_pair_8ph(a, lda, a_absmax, w_plane_f16, w_exp, ldw, K, out_scale, a2, lda2, a2_absmax, w2_plane_f16, ldw2, K2, ratio, bias,
                                               c, ldc, c_absmax, M, N, act, stream);
        if (rc == DBMM_OK) {
            const int cfg[11] = {256, 256, 4, 2, 0, 0, 32, 1, 8, 0, 1};             // [8] = 8: gemm_pair_8ph_kernel, TWO = 1
            for (int i = 0; i < 11; ++i) g_last_cfg[i] = cfg[i];
        }
        if (rc != DBMM_E_UNSUPPORTED) return rc;
    }
    constexpr int BM = 128, BN = 128, MB = 3;
    IgemmP p{};
    p.epi_direct = epi_direct_env();
    p.a = a; p.lda = lda; p.a_bytes = (unsigned)(ab < lim ? ab : lim); p.a_total = ab; p.a_absmax = a_absmax;
    p.wh = (const unsigned short*)w_plane_f16; p.wh_bytes = (unsigned)wb; p.ldw = ldw; p.w_exp = w_exp; p.nw = 1;
    p.a2 = a2; p.lda2 = lda2; p.a2_bytes = (unsigned)(ab2 < lim ? ab2 : lim); p.a2_total = ab2; p.a2_absmax = a2_absmax; p.K2 = (int)K2;
    p.wh2 = (const unsigned short*)w2_plane_f16; p.wh2_bytes = (unsigned)wb2; p.ldw2 = ldw2; p.ratio = ratio;
    p.oscale = out_scale; p.bias = bias; p.c = c; p.ldc = ldc; p.absmax_out = c_absmax;
    p.M = (int)M; p.N = (int)N; p.K = (int)K; p.act = act; p.alpha = 1.f;
    p.tiles_n = (p.N + BN - 1) / BN;
    p.n_tiles = ((p.M + BM - 1) / BM) * p.tiles_n;
    if (p.n_tiles < 192) return DBMM_E_UNSUPPORTED;
    const int nk = (int)(K / 32 + K2 / 32);
    const int sk_mode = sk_mode_env();
    const int grid_sk = NUM_CUS * MB;
    const size_t need = (size_t)grid_sk * 2 * BM * BN * sizeof(float);
    if (sk_mode && workspace && workspace_bytes >= need && dbmm_aligned16(workspace) && nk >= 8) {
        const double per_cu = (double)p.n_tiles / NUM_CUS;
        const double eff = per_cu / (double)((p.n_tiles + NUM_CUS - 1) / NUM_CUS);
        if (sk_mode == 2 || (eff < 0.93 && (long long)p.n_tiles * nk >= 4LL * grid_sk && !sk_skip(p.n_tiles, MB))) {
            p.sk_blocks = grid_sk; p.sk_ws = (float*)workspace; p.sk_nk = nk;
        }
    }
    hipStream_t s = (hipStream_t)stream;
    const dim3 g(p.sk_blocks ? p.sk_blocks : p.n_tiles, 1);
    if (p.sk_blocks)
        hipLaunchKernelGGL((igemm_x3_kernel<BM, BN, 2, 2, 0, MB, 1, 2, 1, 32, 1>), g, dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((igemm_x3_kernel<BM, BN, 2, 2, 0, MB, 0, 2, 1, 32, 1>), g, dim3(256), 0, s, p);
    {
        const int cfg[11] = {BM, BN, 2, 2, 0, 0, 32, MB, 5, p.sk_blocks ? 1 : 0, 1};   // [8] = 5: dual-source GEMM
        for (int i = 0; i < 11; ++i) g_last_cfg[i] = cfg[i];
    }
    DBMM_CHECK_LAUNCH();
    if (p.sk_blocks) {
        hipLaunchKernelGGL((igemm_fixup_kernel<BM, BN, 2, 2, 32>), dim3(p.sk_blocks - 1), dim3(256), 0, s, p);
        DBMM_CHECK_LAUNCH();
    }
    return DBMM_OK;
}

extern "C" int dbmm_conv1x1_bn_act(const float* x, const float* w, const float* bias, const float* residual,
                                   float* y, int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                                   int act, void* stream) {
    return dbmm_conv_bn_act(x, w, bias, residual, y, B, H, W, Cin, Cout, 1, 1, 1, 0, act, stream);
}

extern "C" int dbmm_conv3x3_bn_act(const float* x, const float* w, const float* bias, const float* residual,
                                   float* y, int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                                   int act, void* stream) {
    return dbmm_conv_bn_act(x, w, bias, residual, y, B, H, W, Cin, Cout, 3, 3, 1, 1, act, stream);
}
