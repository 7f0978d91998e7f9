// Parity-mode 3x3 / stride 1 / pad 1 conv (+ folded BatchNorm scale / bias, ReLU, optional fused 2x2 average pool) on the
// deep-pipelined 256 x 256 structure of gemm_pair_8ph_kernel -- the conv2 of the layer-3 / layer-4 bottlenecks
// (/root/reference/clip/model.py:24-26, 44-45: conv2 -> bn2 -> relu -> avgpool(stride)), Cout % 256 == 0.
//
//   y[pixel][n] = act(scale[n] * sum_{kh,kw,c} x[pixel + (kh-1) W + (kw-1)][c] * w[n][(c/32, kh, kw, c%32)] + bias[n])
//
// at fp32 accuracy by fp16-pair products: x = (hi + lo) * 2^-s, w exact in ONE fp16 plane (w * 2^w_exp), two MFMA products
// per fp32 product, fp32 accumulation -- the arithmetic, K order and accumulation order of igemm_halo_kernel (igemm_f32.hip),
// so the two kernels agree bit for bit.
//
// What is taken from the halo kernel: a (32-channel slab, kh) "group" of the activations is split ONCE into fp16 (hi, lo)
// planes and kept in LDS as a strip of 258 pixels (256 output pixels + one halo pixel each side, flattened pixel order);
// its three kw taps read the same strip one LDS row apart.  Border taps (and strip neighbours that are not image
// neighbours: exactly the border taps) are redirected, per fragment row, to a 256-B line of zeros at the same offset
// modulo 256 B (bank-neutral).  POOL: tile rows run 2x2-window-major (row = 4 * window + 2 dy + dx), the strip becomes two
// strips of 130 pixels (one per window row) and the epilogue averages the four accumulator registers of a window.
//
// What is taken from the eight-phase GEMM: 512 threads = 4 x 2 waves of 64 x 128 outputs, two wave groups one barrier
// apart so that one group's fragment reads / staging run under the other group's MFMAs; a tap = one 32-deep K tile = four
// phases of 8 MFMAs (quadrants (A0,B0) (A0,B1) (A1,B1) (A1,B0)); the weight tile of a tap (256 x 32 fp16 = 16 KB) comes by
// LDS-DMA as two half-tiles into a ring of two buffers, five phases ahead, retired by counted vmcnt; persistent
// workgroups, the next tile's weight prologue and first activation group in flight during the epilogue.
// What differs: the activations are staged once per GROUP (12 phases) instead of once per K tile -- 5 loads, 40 split
// instructions and 6 LDS stores per thread and group against 12 / 96 / 12 per thread and 3 K tiles of the GEMM -- and the
// four LDS fragment reads per phase are spread evenly (B0 | B1 | A1 | next tap's A0) instead of 8 | 4 | 4 | 0.
// LDS: 2 A buffers x 2 planes x 272 rows x 64 B + 2 W buffers x 16 KB = 100 KB.  Needs Cin % 64 == 0, Cout % 256 == 0.
#include <stdlib.h>
#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr unsigned OOR = 0x80000000u;
constexpr long long EXT_LIM = 0x7FFFFFF0LL;
constexpr int AR = 272, A_PLANE = AR * 64, A_BUF = 2 * A_PLANE;   // rows of a plane (64 B each); bytes of a plane / of a buffer (hi, lo)
constexpr int W_HALF = 128 * 64, W_BUF = 2 * W_HALF;
constexpr int OFF_W = 0, OFF_A = 2 * W_BUF;                       // W first: every fragment read then reaches its buffer / plane / block through the
constexpr int LDS_BYTES = 2 * A_BUF + 2 * W_BUF;                  // 16-bit immediate offset of ds_read (102,400 B in all)
constexpr int STRIP1 = 136;                                       // POOL: LDS row of the dy = 1 strip (136 = 72 mod 64: conflict-free like the halo kernel's 72)
constexpr int ZROW = 268;                                         // the zero line: rows 268..271 (byte 17,152 = 67 x 256 of a plane)

struct Halo8P {
    const float* a; const float* a_absmax; const unsigned short* w; const float* oscale; const float* bias;
    float* c; float* c_absmax;
    long long a_total, w_total, ldw, ldc;
    int M, N, Cin, H, W, w_exp, tiles_n, n_tiles;
    // the tiles of a short last round, cut along K (see the launcher): tiles [0, n_full) are computed whole; tile n_full + l (l < n_cut) is
    // computed by n_slices workgroups over a share of the loop trips each, which leave their accumulators in ws[(l * n_slices + s)][32][512][4]
    // for conv3x3_halo8_fixup_kernel
    int n_full, n_cut, n_slices;
    float* ws;
};

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
// standard-order pixel of the top-left corner of pooled pixel mp (rows run window-major: m = 4 * mp + 2 dy + dx)
__device__ __forceinline__ int pool_base_pixel(const Halo8P& p, int mp) {
    const int wp2 = p.W >> 1, hwp = (p.H >> 1) * wp2;
    const int n = mp / hwp, rem = mp - n * hwp, hp = rem / wp2;
    return (n * p.H + 2 * hp) * p.W + 2 * (rem - hp * wp2);
}

// one (filtered) atomic per workgroup: all launches of a layer hit ONE address
__device__ __forceinline__ void halo8_amax(const Halo8P& p, float out_amax, float* red, int tid) {
    if (!p.c_absmax) return;
    out_amax = wave_max(out_amax);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = out_amax;
    __syncthreads();
    if (tid == 0) {
        float m = red[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) m = fmaxf(m, red[i]);
        if (m > *(volatile const float*)p.c_absmax) atomicMax((unsigned*)p.c_absmax, __float_as_uint(m));
    }
}

template <int POOL, int ACT>
__global__ __launch_bounds__(512, 1) void conv3x3_halo8_kernel(const Halo8P p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];
    constexpr int NLD = POOL ? 260 : 258;                             // strip rows the loader fills
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wid >> 1, wc = wid & 1, grp = wid >> 2;   // 4 x 2 waves, two groups of four
    const int fr = lane & 31, fh = lane >> 5;
    // this workgroup's tiles: its XCD's contiguous range, walked with the stride of the XCD's workgroups (tiles that share
    // an activation strip -- the N tiles of one row block -- are neighbours in that range)
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot_in_xcd = blockIdx.x >> 3, wg_per_xcd = (nwg - xcd + 7) >> 3;
    const int tq = p.n_full >> 3, trm = p.n_full & 7;
    const int t_lo = xcd < trm ? xcd * (tq + 1) : trm * (tq + 1) + (xcd - trm) * tq, t_hi = t_lo + tq + (xcd < trm ? 1 : 0);
    const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, 0, 0x00020000);   // zero extent: loads return 0
    __amdgpu_buffer_rsrc_t rsA = rs0, rsW = rs0;
    int m0 = 0, n0 = 0;
    const int G = 3 * (p.Cin >> 5);                                   // groups (slab, kh); three taps = K tiles of 32 each
    const int s_a = scale_exp(*p.a_absmax);
    const float a_sc = pow2f(s_a), acc_scale = pow2f(-s_a - p.w_exp);

    // --- activation loader: pass i = 0, 1 covers strip rows (tid >> 2) + 128 i, thread -> 8 channels (32 B of fp32) of its row; pass 2 is the
    //     2 (POOL: 4) rows past 256, 16 B per thread, and only the first 16 (32) threads have a piece.  LDS: 64-B rows per plane, 16-B chunk ^ ((row >> 2) & 3).
    const int lq = tid & 3;
    int st_off[3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int j = (tid >> 2) + 128 * i;
        const int row = POOL ? (j < 130 ? j : j + (STRIP1 - 130)) : j;
        st_off[i] = row * 64 + ((lq ^ ((row >> 2) & 3)) << 4);
    }
    // pass 2: strip row 256 + (tid >> 3), 4 channels (16 B of fp32 -> 8 B per plane) per thread
    const bool has2 = 256 + (tid >> 3) < NLD;
    {
        const int j = 256 + (tid >> 3), row = POOL ? j + (STRIP1 - 130) : j, pc = tid & 7;
        st_off[2] = row * 64 + (((pc >> 1) ^ ((row >> 2) & 3)) << 4) + (pc & 1) * 8;
    }
    // --- weight half-tile h by LDS-DMA (as gemm_pair_8ph_kernel): chunk tid = local row tid >> 2, slot tid & 3; local row lr is tile
    //     column (lr >> 6) * 128 + h * 64 + (lr & 63)
    unsigned voffW[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int lr = tid >> 2;
        voffW[h] = (unsigned)((lr >> 6) * 128 + h * 64 + (lr & 63)) * (unsigned)(p.ldw * 2) + (((tid & 3) ^ ((lr >> 2) & 3)) << 4);
    }
    // --- fragment addresses.  A: tile row r = wr * 64 + 32 i + fr lives in strip row lrow(r) + kw for tap kw; W: half-tile row wc * 64 + 32 cb + fr
    //     Only the ks = 0 address of row block 0 / column block 0 is kept: ks = 1 is the chunk index ^ 2 = address ^ 32; the second block is
    //     32 rows on = a constant number of bytes (A: 2048, POOL 16 strip rows = 1024; W: 2048) with the same swizzle bits.
    constexpr int A_BLK = POOL ? 1024 : 2048, W_BLK = 2048;
    int faddr[3], boff;
    {
        const int r = wr * 64 + fr;
        const int lrow = POOL ? ((r >> 1) & 1) * STRIP1 + (r >> 2) * 2 + (r & 1) : r;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) faddr[kw] = OFF_A + (lrow + kw) * 64 + ((fh ^ (((lrow + kw) >> 2) & 3)) << 4);
        const int rw = wc * 64 + fr;
        boff = rw * 64 + ((fh ^ ((rw >> 2) & 3)) << 4);
    }

    unsigned fa_off[3];                                               // byte offset of the loader rows' kh = 0 pixel from the tile's descriptor base
    unsigned fmask[2];                                                // tap validity of this lane's two fragment rows, bit kh * 3 + kw
    auto set_tile = [&](int tile) {
        m0 = (tile / p.tiles_n) * 256; n0 = (tile % p.tiles_n) * 256;
        // descriptor rebased to the first pixel the tile can touch (tiles at the very start: base 0, "negative" pixels wrap past the
        // extent = zeros); offsets are relative to it, so the tensor itself may be any size
        const int pxf = POOL ? pool_base_pixel(p, m0 >> 2) : m0;
        const int px0 = pxf - 1 - p.W > 0 ? pxf - 1 - p.W : 0;
        rsA = desc(p.a, p.a_total, (long long)px0 * p.Cin * 4);
        rsW = desc(p.w, p.w_total, (long long)n0 * p.ldw * 2);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int j = i < 2 ? (tid >> 2) + 128 * i : 256 + (tid >> 3);
            const unsigned piece = i < 2 ? lq * 32u : (tid & 7) * 16u;
            if constexpr (POOL) {
                // strip dy = j / 130, column c = j % 130: c = 0 / 129 the halo pixels left of the first / right of the last window
                const int dy = j >= 130, c = j - 130 * dy;
                const int wl = c == 0 ? 0 : (c == 129 ? 63 : (c - 1) >> 1), dxo = c == 0 ? -1 : (c == 129 ? 2 : (c - 1) & 1);
                const int mp = (m0 >> 2) + wl;
                fa_off[i] = (j < NLD && 4 * mp < p.M)
                                ? (unsigned)((pool_base_pixel(p, mp) + dy * p.W + dxo - p.W - px0) * p.Cin) * 4u + piece : OOR;
            } else {
                fa_off[i] = j < NLD ? (unsigned)((m0 - 1 + j - p.W - px0) * p.Cin) * 4u + piece : OOR;
            }
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

    // activations of the next group to load, in flight in registers: passes 0 / 1 (2 x 16 B each), pass 2 (16 B)
    f32x4 ar[2][2], ar2;
    int ld_kh = 0, ld_slab = 0;                                       // the group the next load_a fetches
    auto load_a = [&]() {
        const unsigned delta = (unsigned)((ld_kh * p.W * p.Cin + ld_slab * 32) * 4);
        // (groups past the last one -- the final trip's look-ahead -- fetch whatever follows: split, stored, never multiplied)
        const __amdgpu_buffer_rsrc_t rs = rsA;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                ar[i][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, fa_off[i] + delta, 16u * h, 0));
        ar2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, fa_off[2] + delta, 0u, 0));
        if (++ld_kh == 3) { ld_kh = 0; ++ld_slab; }
    };
    // split a pass's values ONCE, store them into the (hi, lo) planes of A buffer `buf`
    auto convert_a = [&](int i, int buf) {
        if (i < 2) {
            unsigned hi[4], lo[4];
            split2h_pair(ar[i][0][0], ar[i][0][1], a_sc, hi[0], lo[0]);
            split2h_pair(ar[i][0][2], ar[i][0][3], a_sc, hi[1], lo[1]);
            split2h_pair(ar[i][1][0], ar[i][1][1], a_sc, hi[2], lo[2]);
            split2h_pair(ar[i][1][2], ar[i][1][3], a_sc, hi[3], lo[3]);
            unsigned char* slot = lds + OFF_A + buf * A_BUF + st_off[i];
            *(u32x4*)slot = (u32x4){hi[0], hi[1], hi[2], hi[3]};
            *(u32x4*)(slot + A_PLANE) = (u32x4){lo[0], lo[1], lo[2], lo[3]};
        } else {
            unsigned hi[2], lo[2];
            split2h_pair(ar2[0], ar2[1], a_sc, hi[0], lo[0]);
            split2h_pair(ar2[2], ar2[3], a_sc, hi[1], lo[1]);
            unsigned char* slot = lds + OFF_A + buf * A_BUF + st_off[2];
            if (has2) {
                *(u32x2*)slot = (u32x2){hi[0], hi[1]};
                *(u32x2*)(slot + A_PLANE) = (u32x2){lo[0], lo[1]};
            }
        }
    };
    // (K tiles past the last one land in buffers no phase reads again before the next tile's prologue refills them)
    auto dma_w = [&](int h, int t, int buf) {
        glds16(rsW, lds + OFF_W + buf * W_BUF + h * W_HALF + wid * 1024, voffW[h], (unsigned)t * 64u);
    };
    f32x16 acc[2][4];                                                 // [A row block (32 rows)][W half * 2 + column block]
    u32x4 fah[2][2], fal[2][2];                                       // [row block][ks]
    u32x4 fb0[2][2], fb1[2][2];                                       // [column block][ks]
    auto read_fa = [&](int i, int buf, int kw, int tapbit) {
        const bool ok = (fmask[i] >> tapbit) & 1u;
        const int f = faddr[kw];
        const int a0 = ok ? f + i * A_BLK : (OFF_A + ZROW * 64 + (f & 255));
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int a = ks ? a0 ^ 32 : a0;
            fah[i][ks] = *(const u32x4*)(lds + buf * A_BUF + a);
            fal[i][ks] = *(const u32x4*)(lds + buf * A_BUF + A_PLANE + a);
        }
    };

    // One phase.  j = phase within the loop trip (static, 0..23): a trip is two groups = six taps; tap tt = j >> 2 of the trip is
    // K tile t = t0 + tt (t0 = 3 * first group, even), lives in W buffer tt & 1 and reads A buffer tt / 3; kh0 = kh of the trip's
    // first group (the second has kh0 + 1, the next trip's first kh0 + 2, all mod 3).  Besides its four fragment reads a phase does:
    //   p0: DMA Bh1(t + 1)                      p3: DMA Bh0(t + 2); vmcnt retires this tap's p0 DMA
    //   p2: vmcnt retires the last p3's DMA     kw = 0, p1: the five loads of the NEXT group's strip
    //   kw = 1, p3 / kw = 2, p0 / kw = 2, p1: split + store pass 0 / 1 / 2 of that strip into the other A buffer (it is read from
    //   kw = 2's p3 on: two phases after the last store, see gemm_pair_8ph_kernel)
    // vmcnt: instructions issued after the DMA being retired = 1 (the other half-tile's DMA), + 5 in the tap with the loads.
    auto phase = [&](int j, int t0, int kh0) {
        const int ph = j & 3, tt = j >> 2, gi = tt / 3, kw = tt - 3 * gi, wb = tt & 1, t = t0 + tt;
        const int kh = gi ? (kh0 == 2 ? 0 : kh0 + 1) : kh0;
        const unsigned char* wbase = lds + OFF_W + wb * W_BUF;
        if (ph == 0) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) fb0[cb][ks] = *(const u32x4*)(wbase + cb * W_BLK + (ks ? boff ^ 32 : boff));
            if (kw == 2) convert_a(1, gi ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            dma_w(1, t + 1, wb ^ 1);
        } else if (ph == 1) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) fb1[cb][ks] = *(const u32x4*)(wbase + W_HALF + cb * W_BLK + (ks ? boff ^ 32 : boff));
            if (kw == 2) convert_a(2, gi ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            if (kw == 0) load_a();
        } else if (ph == 2) {
            read_fa(1, gi, kw, kh * 3 + kw);
        } else {
            if (kw < 2) read_fa(0, gi, kw + 1, kh * 3 + kw + 1);
            else read_fa(0, gi ^ 1, 0, (kh == 2 ? 0 : kh + 1) * 3);
            if (kw == 1) convert_a(0, gi ^ 1);
            __builtin_amdgcn_sched_barrier(0);
            dma_w(0, t + 2, wb);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ph >= 2) {
            if (kw == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        const int ai = (ph >= 2) ? 1 : 0, bj = (ph == 1 || ph == 2) ? 1 : 0;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)                            // (lo, w) first, then (hi, w)
                acc[ai][2 * bj + cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fal[ai][ks]), __builtin_bit_cast(f16x8, bj ? fb1[cb][ks] : fb0[cb][ks]),
                                                                              acc[ai][2 * bj + cb], 0, 0, 0);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
                acc[ai][2 * bj + cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fah[ai][ks]), __builtin_bit_cast(f16x8, bj ? fb1[cb][ks] : fb0[cb][ks]),
                                                                              acc[ai][2 * bj + cb], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };
    float out_amax = 0.f;

    // the zero line of both planes of both A buffers (never written by the loader)
    if (tid < 64) *(u32x4*)(lds + OFF_A + (tid >> 5) * A_BUF + ((tid >> 4) & 1) * A_PLANE + ZROW * 64 + (tid & 15) * 16) = (u32x4){0u, 0u, 0u, 0u};

    // Work items of this workgroup: its whole tiles (all groups), then its share of the n_cut * n_slices slices (slice i = cut tile i / S,
    // loop trips [s T / S, (s + 1) T / S) with s = i % S), dealt round-robin over the workgroups.
    const int n_whole = t_lo + slot_in_xcd < t_hi ? (t_hi - t_lo - slot_in_xcd + wg_per_xcd - 1) / wg_per_xcd : 0;
    const int n_sl_all = p.n_cut * p.n_slices;                        // slices: number blockIdx.x + j * gridDim.x goes to this workgroup
    const int n_sl = (int)blockIdx.x < n_sl_all ? (n_sl_all - 1 - (int)blockIdx.x) / nwg + 1 : 0;
    const int n_items = n_whole + n_sl;
    int gb = 0, ge = G, slice_id = 0;                                 // the current item's groups [gb, ge), both even; its slice number
    auto set_item = [&](int k) {
        if (k < n_whole) { gb = 0; ge = G; set_tile(t_lo + slot_in_xcd + k * wg_per_xcd); }
        else {
            slice_id = blockIdx.x + (k - n_whole) * nwg;
            const int l = slice_id / p.n_slices, sl = slice_id - l * p.n_slices, T = G >> 1;
            gb = 2 * (sl * T / p.n_slices); ge = 2 * ((sl + 1) * T / p.n_slices);
            set_tile(p.n_full + l);
        }
    };
    // what an item needs before its first phases: Bh0, Bh1 of its first tap into W buffer 0, Bh0 of the second into buffer 1, its first
    // group's strip in registers
    auto prologue = [&]() {
        dma_w(0, 3 * gb, 0); dma_w(1, 3 * gb, 0); dma_w(0, 3 * gb + 1, 1);
        ld_slab = gb / 3; ld_kh = gb - 3 * ld_slab;
        load_a();
    };
    if (n_items > 0) { set_item(0); prologue(); }
    for (int k = 0; k < n_items; ++k) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        // the first group's strip: the wait its conversion needs also retires everything older (the W prologue, the previous epilogue's stores)
        convert_a(0, 0); convert_a(1, 0); convert_a(2, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int kh0 = gb % 3;
        read_fa(0, 0, 0, kh0 * 3);                                    // the first tap's A0 fragments (every later tap's are read in the p3 before it)
        if (grp == 1) __builtin_amdgcn_s_barrier();                   // the second wave group runs one barrier behind
        for (int g2 = gb; g2 < ge; g2 += 2) {
#pragma unroll
            for (int j = 0; j < 24; ++j) phase(j, 3 * g2, kh0);
            kh0 = kh0 == 0 ? 2 : kh0 - 1;                             // (kh0 + 2) mod 3
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the look-ahead loads / DMA past this item's last tap
        __builtin_amdgcn_s_barrier();                                 // every wave is done with the LDS
        const int em0 = m0, en0 = n0, e_slice = slice_id;
        const bool whole = k < n_whole;
        if (k + 1 < n_items) { set_item(k + 1); prologue(); }         // in flight during the epilogue below
        if (whole) {
            constexpr int EP_NB = 4, EP_WCOLS = 128;
#include "conv3x3_halo8_epilogue.inc"
        } else {                                                      // a slice: the raw accumulators, 16 B per thread and store (8 KB per workgroup instruction)
            // (buffer stores with the register's offset in soffset: 128 flat addresses would be hoisted out of the item loop and spilled)
            const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ws + (size_t)e_slice * (128 * 512)), 0, 128 * 512 * 4, 0x00020000);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {                     // four registers per store: [register quad][thread][4]
                        const u32x4 v = {__float_as_uint(acc[i][j][4 * q]), __float_as_uint(acc[i][j][4 * q + 1]), __float_as_uint(acc[i][j][4 * q + 2]),
                                         __float_as_uint(acc[i][j][4 * q + 3])};
                        __builtin_amdgcn_raw_buffer_store_b128(v, rsP, (unsigned)tid * 16u, (unsigned)(((i * 4 + j) * 4 + q) * 8192), 0);
                    }
        }
    }
    __syncthreads();                                                  // all DMA drained (loop exit), the LDS is free
    halo8_amax(p, out_amax, (float*)lds, tid);
}

// The cut tiles: sum the slices' accumulators in slice order (same thread <-> element mapping as the main kernel), then the epilogue of
// conv3x3_halo8_epilogue.inc for ONE 32 x 32 block (i, j) of every wave's 64 x 128 per workgroup -- grid (n_cut, 8): with one workgroup per
// tile the n_cut x n_slices x 256 KB were read by n_cut CUs only (layer 3: 50 MB through 16 CUs = 55 us, as much as the cut saved).
template <int POOL, int ACT>
__global__ __launch_bounds__(512) void conv3x3_halo8_fixup_kernel(const Halo8P p) {
    __shared__ float red[8];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 1, wc = wid & 1, fr = lane & 31, fh = lane >> 5;
    const int tile = p.n_full + blockIdx.x, m0 = (tile / p.tiles_n) * 256, n0 = (tile % p.tiles_n) * 256;
    const int bi = blockIdx.y >> 2, bj = blockIdx.y & 3;
    f32x16 a;
    const f32x4* src = (const f32x4*)(p.ws + (size_t)blockIdx.x * p.n_slices * (128 * 512)) + (size_t)blockIdx.y * 4 * 512 + tid;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = src[q * 512];
#pragma unroll
        for (int e = 0; e < 4; ++e) a[4 * q + e] = v[e];
    }
    for (int sl = 1; sl < p.n_slices; ++sl) {
        src += 32 * 512;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = src[q * 512];
#pragma unroll
            for (int e = 0; e < 4; ++e) a[4 * q + e] += v[e];
        }
    }
    const int s_a = scale_exp(*p.a_absmax);
    const int n = n0 + wc * 128 + 32 * bj + fr;
    const float sv = (p.oscale ? p.oscale[n] : 1.f) * pow2f(-s_a - p.w_exp), bv = p.bias ? p.bias[n] : 0.f;
    float out_amax = 0.f;
    if constexpr (POOL) {
        const int mpw = (m0 + wr * 64) >> 2;
        const long long rows_left = (long long)(p.M >> 2) - mpw;
        const __amdgpu_buffer_rsrc_t rsC = desc(p.c, rows_left > 0 ? ((rows_left - 1) * p.ldc + p.N) * 4 + (long long)mpw * p.ldc * 4 : 0,
                                                (long long)mpw * p.ldc * 4);
        const int row_lim = (int)(rows_left < 1024 ? rows_left : 1024) - fh;
        const unsigned vc = (unsigned)((fh * p.ldc + n) * 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float sum = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v = a[4 * k + q] * sv + bv;
                if (ACT == DBMM_ACT_RELU) v = fmaxf(v, 0.f);
                sum += v;
            }
            sum *= 0.25f;
            const int u = 8 * bi + 2 * k;
            const bool valid = u < row_lim;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, sum), rsC, valid ? vc : OOR, (unsigned)(u * p.ldc * 4), 0);
            if (valid) out_amax = fmaxf(out_amax, fabsf(sum));
        }
    } else {
        const int mw = m0 + wr * 64;
        const long long rows_left = (long long)p.M - mw;
        const __amdgpu_buffer_rsrc_t rsC = desc(p.c, rows_left > 0 ? ((rows_left - 1) * p.ldc + p.N) * 4 + (long long)mw * p.ldc * 4 : 0,
                                                (long long)mw * p.ldc * 4);
        const int row_lim = (int)(rows_left < 1024 ? rows_left : 1024) - 4 * fh;
        const unsigned vc = (unsigned)((4 * fh * p.ldc + n) * 4);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = a[r] * sv + bv;
            if (ACT == DBMM_ACT_RELU) v = fmaxf(v, 0.f);
            const int u = 32 * bi + (r & 3) + 8 * (r >> 2);
            const bool valid = u < row_lim;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsC, valid ? vc : OOR, (unsigned)(u * p.ldc * 4), 0);
            if (valid) out_amax = fmaxf(out_amax, fabsf(v));
        }
    }
    halo8_amax(p, out_amax, red, tid);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same conv for Cout % 128 == 0 (layer 2: 128 channels): tile 256 pixels x 128 channels, 4 x 2 waves of 64 x 64 outputs (64 accumulator
// registers).  A tap is TWO phases of 8 MFMAs -- q0: row block A0 x both column blocks, q1: A1 x both -- with six fragment reads each
// (q0: the tap's four W fragments + A1's first k-step; q1: A1's second k-step + the NEXT tap's A0), one 8-KB weight tile per tap by LDS-DMA
// into a ring of three (issued at q1, three taps = five phases ahead of its first read), and THREE activation buffers: group g's strip is
// loaded at the first phase of group g - 2 and split / stored in that group's last two phases (the registers are free again before the next
// load; the ring's third buffer is what lets the store wait that long).  A loop trip = three groups = one 32-channel slab: weight buffer =
// kw, activation buffer = kh, mask bit = 3 kh + kw are all compile-time.  No K cut of short last rounds (layer 2 has 49+ rounds).
constexpr int N_W_BUF = 128 * 64;                                 // a tap's weight tile
constexpr int N_OFF_A = 3 * N_W_BUF;
constexpr int N_LDS_BYTES = N_OFF_A + 3 * A_BUF;                  // 24,576 + 104,448 = 129,024

template <int POOL, int ACT>
__global__ __launch_bounds__(512, 1) void conv3x3_halo8n_kernel(const Halo8P p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[N_LDS_BYTES];
    constexpr int NLD = POOL ? 260 : 258;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wid >> 1, wc = wid & 1, grp = wid >> 2;
    const int fr = lane & 31, fh = lane >> 5;
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot_in_xcd = blockIdx.x >> 3, wg_per_xcd = (nwg - xcd + 7) >> 3;
    const int tq = p.n_tiles >> 3, trm = p.n_tiles & 7;
    const int t_lo = xcd < trm ? xcd * (tq + 1) : trm * (tq + 1) + (xcd - trm) * tq, t_hi = t_lo + tq + (xcd < trm ? 1 : 0);
    __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, 0, 0x00020000), rsW = rsA;
    int m0 = 0, n0 = 0;
    const int G = 3 * (p.Cin >> 5);
    const int s_a = scale_exp(*p.a_absmax);
    const float a_sc = pow2f(s_a), acc_scale = pow2f(-s_a - p.w_exp);

    // activation loader and fragment rows: as conv3x3_halo8_kernel
    const int lq = tid & 3;
    int st_off[3];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int j = (tid >> 2) + 128 * i;
        const int row = POOL ? (j < 130 ? j : j + (STRIP1 - 130)) : j;
        st_off[i] = row * 64 + ((lq ^ ((row >> 2) & 3)) << 4);
    }
    const bool has2 = 256 + (tid >> 3) < NLD;
    {
        const int j = 256 + (tid >> 3), row = POOL ? j + (STRIP1 - 130) : j, pc = tid & 7;
        st_off[2] = row * 64 + (((pc >> 1) ^ ((row >> 2) & 3)) << 4) + (pc & 1) * 8;
    }
    // weight tile by LDS-DMA: chunk tid = tile column tid >> 2, slot tid & 3 of its 64-B row
    const unsigned voffW = (unsigned)(tid >> 2) * (unsigned)(p.ldw * 2) + (((tid & 3) ^ ((tid >> 4) & 3)) << 4);
    constexpr int A_BLK = POOL ? 1024 : 2048, W_BLK = 2048;
    int faddr[3], boff;
    {
        const int r = wr * 64 + fr;
        const int lrow = POOL ? ((r >> 1) & 1) * STRIP1 + (r >> 2) * 2 + (r & 1) : r;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) faddr[kw] = N_OFF_A + (lrow + kw) * 64 + ((fh ^ (((lrow + kw) >> 2) & 3)) << 4);
        const int rw = wc * 64 + fr;
        boff = rw * 64 + ((fh ^ ((rw >> 2) & 3)) << 4);
    }
    unsigned fa_off[3], fmask[2];
    auto set_tile = [&](int tile) {
        m0 = (tile / p.tiles_n) * 256; n0 = (tile % p.tiles_n) * 128;
        const int pxf = POOL ? pool_base_pixel(p, m0 >> 2) : m0;
        const int px0 = pxf - 1 - p.W > 0 ? pxf - 1 - p.W : 0;
        rsA = desc(p.a, p.a_total, (long long)px0 * p.Cin * 4);
        rsW = desc(p.w, p.w_total, (long long)n0 * p.ldw * 2);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int j = i < 2 ? (tid >> 2) + 128 * i : 256 + (tid >> 3);
            const unsigned piece = i < 2 ? lq * 32u : (tid & 7) * 16u;
            if constexpr (POOL) {
                const int dy = j >= 130, c = j - 130 * dy;
                const int wl = c == 0 ? 0 : (c == 129 ? 63 : (c - 1) >> 1), dxo = c == 0 ? -1 : (c == 129 ? 2 : (c - 1) & 1);
                const int mp = (m0 >> 2) + wl;
                fa_off[i] = (j < NLD && 4 * mp < p.M)
                                ? (unsigned)((pool_base_pixel(p, mp) + dy * p.W + dxo - p.W - px0) * p.Cin) * 4u + piece : OOR;
            } else {
                fa_off[i] = j < NLD ? (unsigned)((m0 - 1 + j - p.W - px0) * p.Cin) * 4u + piece : OOR;
            }
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

    f32x4 ar[2][2], ar2;                                              // the strip in flight (group g + 2 while group g computes)
    f32x4 br[2][2], br2;                                              // item start only: the item's second group
    auto load_strip = [&](f32x4 (&r)[2][2], f32x4& r2, int g) {
        const int slab = g / 3, kh = g - 3 * slab;
        const unsigned delta = (unsigned)((kh * p.W * p.Cin + slab * 32) * 4);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                r[i][h] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, fa_off[i] + delta, 16u * h, 0));
        r2 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, fa_off[2] + delta, 0u, 0));
    };
    auto convert = [&](const f32x4 (&r)[2][2], const f32x4& r2, int i, int buf) {
        if (i < 2) {
            unsigned hi[4], lo[4];
            split2h_pair(r[i][0][0], r[i][0][1], a_sc, hi[0], lo[0]);
            split2h_pair(r[i][0][2], r[i][0][3], a_sc, hi[1], lo[1]);
            split2h_pair(r[i][1][0], r[i][1][1], a_sc, hi[2], lo[2]);
            split2h_pair(r[i][1][2], r[i][1][3], a_sc, hi[3], lo[3]);
            unsigned char* slot = lds + N_OFF_A + buf * A_BUF + st_off[i];
            *(u32x4*)slot = (u32x4){hi[0], hi[1], hi[2], hi[3]};
            *(u32x4*)(slot + A_PLANE) = (u32x4){lo[0], lo[1], lo[2], lo[3]};
        } else {
            unsigned hi[2], lo[2];
            split2h_pair(r2[0], r2[1], a_sc, hi[0], lo[0]);
            split2h_pair(r2[2], r2[3], a_sc, hi[1], lo[1]);
            unsigned char* slot = lds + N_OFF_A + buf * A_BUF + st_off[2];
            if (has2) {
                *(u32x2*)slot = (u32x2){hi[0], hi[1]};
                *(u32x2*)(slot + A_PLANE) = (u32x2){lo[0], lo[1]};
            }
        }
    };
    auto dma_w = [&](int t, int buf) { glds16(rsW, lds + buf * N_W_BUF + wid * 1024, voffW, (unsigned)t * 64u); };
    f32x16 acc[2][2];
    u32x4 fah[2][2], fal[2][2], fb[2][2];                             // A: [row block][ks]; W: [column block][ks]
    auto a_addr = [&](int i, int buf, int kw, int tapbit) -> int {
        const bool ok = (fmask[i] >> tapbit) & 1u;
        const int f = faddr[kw] + buf * A_BUF;
        return ok ? f + i * A_BLK : (N_OFF_A + buf * A_BUF + ZROW * 64 + (f & 255));
    };
    auto read_fa = [&](int i, int ks, int a0) {
        const int a = ks ? a0 ^ 32 : a0;
        fah[i][ks] = *(const u32x4*)(lds + a);
        fal[i][ks] = *(const u32x4*)(lds + A_PLANE + a);
    };

    // One phase.  j = phase within the trip (static, 0..17): group gi = j / 6 of the trip (= kh), tap kw = (j % 6) / 2, q = j & 1; the tap is
    // K tile t = t0 + 3 gi + kw (t0 = 9 * slab).  Staging besides the six fragment reads:
    //   (kw 0, q0): the five loads of group g + 2's strip        (kw 2, q0) / (kw 2, q1): split + store pass 0 / passes 1, 2 into buffer (kh + 2) % 3
    //   q1 of every tap: DMA of the weight tile three taps on into this tap's buffer; vmcnt retires the tile of the NEXT tap
    //   (instructions issued after that tile's DMA: the two later tiles', + the five loads unless they are older: kw 2)
    auto phase = [&](int j, int t0, int g0) {
        const int gi = j / 6, kw = (j - 6 * gi) >> 1, q = j & 1, t = t0 + 3 * gi + kw;
        if (q == 0) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) fb[cb][ks] = *(const u32x4*)(lds + kw * N_W_BUF + cb * W_BLK + (ks ? boff ^ 32 : boff));
            read_fa(1, 0, a_addr(1, gi, kw, 3 * gi + kw));
            if (kw == 2) convert(ar, ar2, 0, (gi + 2) % 3);
            __builtin_amdgcn_sched_barrier(0);
            if (kw == 0) load_strip(ar, ar2, g0 + gi + 2);
        } else {
            read_fa(1, 1, a_addr(1, gi, kw, 3 * gi + kw));
            const int a0 = kw < 2 ? a_addr(0, gi, kw + 1, 3 * gi + kw + 1) : a_addr(0, (gi + 1) % 3, 0, 3 * ((gi + 1) % 3));
            read_fa(0, 0, a0); read_fa(0, 1, a0);
            if (kw == 2) { convert(ar, ar2, 1, (gi + 2) % 3); convert(ar, ar2, 2, (gi + 2) % 3); }
            __builtin_amdgcn_sched_barrier(0);
            dma_w(t + 3, kw);
            __builtin_amdgcn_sched_barrier(0);
            if (kw == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)                            // (lo, w) first, then (hi, w)
                acc[q][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fal[q][ks]), __builtin_bit_cast(f16x8, fb[cb][ks]), acc[q][cb], 0, 0, 0);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
                acc[q][cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fah[q][ks]), __builtin_bit_cast(f16x8, fb[cb][ks]), acc[q][cb], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };
    float out_amax = 0.f;

    // the zero line of both planes of the three A buffers
    if (tid < 96) *(u32x4*)(lds + N_OFF_A + (tid >> 5) * A_BUF + ((tid >> 4) & 1) * A_PLANE + ZROW * 64 + (tid & 15) * 16) = (u32x4){0u, 0u, 0u, 0u};

    const int n_items = t_lo + slot_in_xcd < t_hi ? (t_hi - t_lo - slot_in_xcd + wg_per_xcd - 1) / wg_per_xcd : 0;
    // what a tile needs before its first phases: the weight tiles of its first three taps, its first two groups' strips in registers
    auto prologue = [&]() {
        dma_w(0, 0); dma_w(1, 1); dma_w(2, 2);
        load_strip(ar, ar2, 0); load_strip(br, br2, 1);
    };
    if (n_items > 0) { set_tile(t_lo + slot_in_xcd); prologue(); }
    for (int k = 0; k < n_items; ++k) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        convert(ar, ar2, 0, 0); convert(ar, ar2, 1, 0); convert(ar, ar2, 2, 0);
        convert(br, br2, 0, 1); convert(br, br2, 1, 1); convert(br, br2, 2, 1);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        { const int a0 = a_addr(0, 0, 0, 0); read_fa(0, 0, a0); read_fa(0, 1, a0); }
        if (grp == 1) __builtin_amdgcn_s_barrier();
        for (int g0 = 0; g0 < G; g0 += 3) {
#pragma unroll
            for (int j = 0; j < 18; ++j) phase(j, 3 * g0, g0);
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        const int em0 = m0, en0 = n0;
        if (k + 1 < n_items) { set_tile(t_lo + slot_in_xcd + (k + 1) * wg_per_xcd); prologue(); }
        {
            constexpr int EP_NB = 2, EP_WCOLS = 64;
#include "conv3x3_halo8_epilogue.inc"
        }
    }
    __syncthreads();
    halo8_amax(p, out_amax, (float*)lds, tid);
}

}  // namespace

// see common.h.  DBMM_E_UNSUPPORTED: the caller falls back to igemm_halo_kernel.
// Tile quantisation: the persistent grid works in rounds of 256 tiles (layer 3 at B = 1024: 784 tiles = 3.06 rounds = the time of 4).
// With a workspace and split != 0 the tiles of a short last round are cut along K into S slices (dbmm_cut_slices, common.h) dealt over the
// workgroups, and a small second launch sums the slices and runs the epilogue: the last round then costs ~1 / S of a round plus the slices'
// fixed costs and 2 x tiles x S x 256 KB of traffic.
int dbmm_conv3x3_halo8(const float* x, const float* x_absmax, const void* w_plane_f16, int w_exp, const float* out_scale, const float* bias,
                       float* y, float* y_absmax, int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int act, int pool, int split,
                       void* workspace, size_t workspace_bytes, void* stream) {
    if (!x || !x_absmax || !w_plane_f16 || !y) return DBMM_E_ARG;
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return DBMM_E_SHAPE;
    if (act != DBMM_ACT_NONE && act != DBMM_ACT_RELU) return DBMM_E_UNSUPPORTED;
    if (pool != 0 && pool != 2) return DBMM_E_ARG;
    const int64_t M = B * H * W, K = 9 * Cin;
    if ((Cout % 128) || (Cin % 64) || w_exp < -40 || w_exp > 40 || M > (INT32_MAX >> 1)) return DBMM_E_UNSUPPORTED;
    if (pool && ((H & 1) || (W & 1))) return DBMM_E_UNSUPPORTED;
    if (!dbmm_aligned16(x) || !dbmm_aligned16(w_plane_f16) || !dbmm_aligned16(y)) return DBMM_E_ALIGN;
    const long long wb = Cout * K * 2;
    // a tile's offsets from its rebased descriptors: the strip plus an image row either side / 256 output rows
    if (wb >= EXT_LIM || (260 + 4 * W) * Cin * 4 >= EXT_LIM || 256 * Cout * 4 >= EXT_LIM) return DBMM_E_UNSUPPORTED;
    Halo8P p{};
    p.a = x; p.a_absmax = x_absmax; p.w = (const unsigned short*)w_plane_f16; p.oscale = out_scale; p.bias = bias; p.c = y; p.c_absmax = y_absmax;
    p.a_total = M * Cin * 4; p.w_total = wb; p.ldw = K; p.ldc = Cout;
    p.M = (int)M; p.N = (int)Cout; p.Cin = (int)Cin; p.H = (int)H; p.W = (int)W; p.w_exp = w_exp;
    hipStream_t s = (hipStream_t)stream;
    if (Cout % 256) {                                                 // 128-channel tiles (layer 2)
        p.tiles_n = (int)(Cout / 128);
        p.n_tiles = (int)((M + 255) / 256) * p.tiles_n;
        p.n_full = p.n_tiles; p.n_cut = 0; p.n_slices = 1; p.ws = nullptr;
        const int gridn = p.n_tiles < 256 ? p.n_tiles : 256;
#define DBMM_H8N(P, A) hipLaunchKernelGGL((conv3x3_halo8n_kernel<P, A>), dim3(gridn), dim3(512), 0, s, p)
        if (pool) { if (act == DBMM_ACT_RELU) DBMM_H8N(1, 1); else DBMM_H8N(1, 0); }
        else { if (act == DBMM_ACT_RELU) DBMM_H8N(0, 1); else DBMM_H8N(0, 0); }
#undef DBMM_H8N
        DBMM_CHECK_LAUNCH();
        return DBMM_OK;
    }
    p.tiles_n = (int)(Cout / 256);
    p.n_tiles = (int)((M + 255) / 256) * p.tiles_n;
    p.n_full = p.n_tiles; p.n_cut = 0; p.n_slices = 1; p.ws = nullptr;
    const int rem = p.n_tiles % 256, trips = (int)(3 * (Cin / 32) / 2);
    if (split && workspace && dbmm_aligned16(workspace) && p.n_tiles > 256 && rem != 0) {
        const int S = dbmm_cut_slices(rem, trips);
        if (S >= 2 && (size_t)rem * S * (128 * 512 * sizeof(float)) <= workspace_bytes) {
            p.n_full = p.n_tiles - rem; p.n_cut = rem; p.n_slices = S; p.ws = (float*)workspace;
        }
    }
    const int grid = p.n_tiles < 256 ? p.n_tiles : 256;               // persistent: one workgroup per CU
#define DBMM_H8(P, A) hipLaunchKernelGGL((conv3x3_halo8_kernel<P, A>), dim3(grid), dim3(512), 0, s, p)
    if (pool) { if (act == DBMM_ACT_RELU) DBMM_H8(1, 1); else DBMM_H8(1, 0); }
    else { if (act == DBMM_ACT_RELU) DBMM_H8(0, 1); else DBMM_H8(0, 0); }
#undef DBMM_H8
    DBMM_CHECK_LAUNCH();
    if (p.n_cut) {
#define DBMM_H8F(P, A) hipLaunchKernelGGL((conv3x3_halo8_fixup_kernel<P, A>), dim3(p.n_cut, 8), dim3(512), 0, s, p)
        if (pool) { if (act == DBMM_ACT_RELU) DBMM_H8F(1, 1); else DBMM_H8F(1, 0); }
        else { if (act == DBMM_ACT_RELU) DBMM_H8F(0, 1); else DBMM_H8F(0, 0); }
#undef DBMM_H8F
        DBMM_CHECK_LAUNCH();
    }
    return DBMM_OK;
}
