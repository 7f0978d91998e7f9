// Parity-mode GEMM of the transformer towers on the deep-pipelined structure of gemm_f16_8ph_kernel (f16_ops.hip):
//   c = act(alpha * ((a @ w^T) * out_scale + bias) + residual),  a fp32 [M][K],  w exact in ONE fp16 plane (w * 2^w_exp),
// at fp32 accuracy: a = (hi + lo) * 2^-s, two fp16 MFMA products per fp32 product, fp32 accumulation (the fp16-pair
// arithmetic of igemm_f32.hip; clip/model.py:171-240 are the GEMMs it serves in the parity mode).
//
// What differs from the fp16 kernel: the activations are fp32 in HBM, so LDS-DMA brings them in as fp32 and the split into
// (hi, lo) happens when a wave READS its fragments -- 16 VALU per 32 x 16 fragment, placed in the "load" half of a phase,
// i.e. while the other wave group of the SIMD owns the matrix pipe (VALU and MFMA of different waves overlap).  Every
// wave that shares an A row block repeats that split, so the waves are laid out 4 x 2 (64 x 128 of output each; with
// 2 x 4 the four waves of a row quadrupled the split and the kernel ran BELOW the two-barrier one: 337 vs 506 TF-eq at
// K = 4096): two waves per row block = as many split instructions as splitting once on the way into LDS.  To keep two
// buffers of four half-tiles inside LDS a K tile is 32 deep: A half-tile 128 rows x 32 k fp32 = 16 KB (2 DMA instructions
// per thread), W half-tile 128 rows x 32 k fp16 = 8 KB (1), 96 KB in all.  A phase is one 32 x 64 quadrant x 32 k =
// 2 column blocks x 2 k-steps x 2 products = 8 MFMAs, as in the fp16 kernel, and the stage / wait / read distances are the same
// (stage five phases ahead, retire three stages back, read one phase after the wait); only the vmcnt constants differ
// because the stages issue 2 / 1 / 1 / 2 instructions: outstanding after the wait = the last three stages = 5 / 4 / 4 / 5.
// Persistent workgroups, next tile's first five half-tiles issued before the epilogue, epilogue straight from the
// accumulator layout (one register = two 128-B fp32 row segments), branch-free buffer accesses, activation / residual
// compile-time.  Needs N % 256 == 0, K % 64 == 0.
#include <stdlib.h>
#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOR = 0x80000000u;
constexpr long long EXT_LIM = 0x7FFFFFF0LL;
constexpr int A_HALF = 128 * 32 * 4, W_HALF = 128 * 32 * 2;        // bytes of the half-tiles
constexpr int BUF = 2 * A_HALF + 2 * W_HALF;                       // one buffer: Ah0, Bh0, Bh1, Ah1
constexpr int OFF_AH0 = 0, OFF_BH0 = A_HALF, OFF_BH1 = A_HALF + W_HALF, OFF_AH1 = A_HALF + 2 * W_HALF;

struct PairP {
    const float* a; const float* a_absmax; const unsigned short* w; const float* oscale; const float* bias; const float* res;
    float* c; float* c_absmax;
    long long lda, ldw, ldr, ldc, a_total, w_total;
    int M, N, K, w_exp, tiles_n, n_tiles;
    float alpha;
    // TWO: a second operand pair whose K2 / 32 tiles run FIRST (conv3 + downsample branch of a stage's first block as one GEMM, see
    // dbmm_gemm_dual_bn_act_x2): then the accumulators are multiplied per output channel by ratio[n] * 2^(s - s2) and the main pair continues
    const float* a2; const float* a2_absmax; const unsigned short* w2; const float* ratio;
    long long lda2, ldw2, a2_total, w2_total;
    int K2;
    // the tiles of a short last round, cut along K (see pair_8ph_launch): tiles [0, n_full) are computed whole; tile n_full + l (l < n_cut) by
    // n_slices workgroups over a share of the loop trips each, which leave their accumulators in ws[(l * n_slices + s)][32][512][4] for
    // gemm_pair_8ph_fixup_kernel.  Not with TWO.
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
__device__ __forceinline__ f16x8 frag(const unsigned (&v)[4]) { return __builtin_bit_cast(f16x8, (u32x4){v[0], v[1], v[2], v[3]}); }

template <int ACT, int RES, int TWO = 0>
__global__ __launch_bounds__(512, 1) void gemm_pair_8ph_kernel(const PairP p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * BUF + (TWO ? 8192 : 0)];   // 96 KB (+ TWO: this thread's four column ratios)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wid >> 1, wc = wid & 1, grp = wid >> 2;   // 4 x 2 waves, two groups of four
    const int fr = lane & 31, fh = lane >> 5;
    // this workgroup's tiles: its XCD's contiguous range (xcd_remap's split), walked with the stride of the XCD's workgroups
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot_in_xcd = blockIdx.x >> 3, wg_per_xcd = (nwg - xcd + 7) >> 3;
    const int tq = p.n_full >> 3, trm = p.n_full & 7;
    const int t_lo = xcd < trm ? xcd * (tq + 1) : trm * (tq + 1) + (xcd - trm) * tq, t_hi = t_lo + tq + (xcd < trm ? 1 : 0);
    const __amdgpu_buffer_rsrc_t rsA0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0, 0x00020000);
    __amdgpu_buffer_rsrc_t rsA = rsA0, rsW = rsW0, rsA2 = rsA0, rsW2 = rsW0;
    int m0 = 0, n0 = 0;
    // A half-tile h (128 rows x 32 k): thread -> (local row tid >> 2, k group tid & 3 of 8 k = 32 B of fp32).  Local row lr is
    // tile row (lr >> 5) * 64 + h * 32 + (lr & 31): the h-th 32 rows of every wave row's 64.  W half-tile h by LDS-DMA: chunk
    // tid = local row tid >> 2, slot tid & 3 of a 64-B row, source chunk slot ^ ((row >> 2) & 3); local row lr is tile column
    // (lr >> 6) * 128 + h * 64 + (lr & 63).
    // (TWO: the per-lane offsets of BOTH operand pairs are recomputed where they are used and the ratios wait in LDS -- kept in registers they
    //  pushed the loop over the budget, and every spill reload inside it is a vector-memory load the counted waits then drain behind)
    unsigned voffA[2], voffW[2];
    const int nT2 = TWO ? p.K2 / 32 : 0, nT = nT2 + p.K / 32;         // K tiles of the second pair (they run first) / in all
    const int s_a = scale_exp(*p.a_absmax), s_a2 = TWO ? scale_exp(*p.a2_absmax) : 0;
    const float a_sc = pow2f(s_a), acc_scale = pow2f(-s_a - p.w_exp), a_sc2 = pow2f(s_a2);
    auto set_tile = [&](int tile) {
        m0 = (tile / p.tiles_n) * 256; n0 = (tile % p.tiles_n) * 256;
        rsA = desc(p.a, p.a_total, (long long)m0 * p.lda * 4);
        rsW = desc(p.w, p.w_total, (long long)n0 * p.ldw * 2);
        if constexpr (TWO) {
            rsA2 = desc(p.a2, p.a2_total, (long long)m0 * p.lda2 * 4);
            rsW2 = desc(p.w2, p.w2_total, (long long)n0 * p.ldw2 * 2);
        }
        const int lr = tid >> 2;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int ra = (lr >> 5) * 64 + h * 32 + (lr & 31);
            const unsigned wrow = (unsigned)((lr >> 6) * 128 + h * 64 + (lr & 63)), wsl = (unsigned)(((tid & 3) ^ ((lr >> 2) & 3)) << 4);
            voffA[h] = m0 + ra < p.M ? (unsigned)ra * (unsigned)(p.lda * 4) + (tid & 3) * 32u : OOR;
            voffW[h] = wrow * (unsigned)(p.ldw * 2) + wsl;
        }
        if constexpr (TWO) {                                          // ratio[n] * 2^(s - s2) of this lane's four columns
            f32x4 rr;
#pragma unroll
            for (int j = 0; j < 4; ++j) rr[j] = p.ratio[n0 + (wid & 1) * 128 + 32 * j + (lane & 31)] * pow2f(s_a - s_a2);
            *(f32x4*)(lds + 2 * BUF + tid * 16) = rr;
        }
    };
    auto voff_a = [&](int h, bool second) -> unsigned {               // TWO: the A offset of load_a, from scratch
        const int lr = tid >> 2, ra = (lr >> 5) * 64 + h * 32 + (lr & 31);
        const unsigned ld4 = (unsigned)((second ? p.lda2 : p.lda) * 4);
        return m0 + ra < p.M ? (unsigned)ra * ld4 + (tid & 3) * 32u : OOR;
    };
    auto voff_w = [&](int h, bool second) -> unsigned {
        const int lr = tid >> 2;
        const unsigned ld2 = (unsigned)((second ? p.ldw2 : p.ldw) * 2);
        return (unsigned)((lr >> 6) * 128 + h * 64 + (lr & 63)) * ld2 + (((tid & 3) ^ ((lr >> 2) & 3)) << 4);
    };
    // this thread's 16-B slot inside an fp16 plane of an A half-tile (its 8 k values), and the fragment addresses: A rows
    // wr * 32 + fr, W rows wc * 64 + cb * 32 + fr, chunk 2 ks + fh, 64-B rows swizzled by (row >> 2) & 3
    const int cv_off = (tid >> 2) * 64 + (((tid & 3) ^ (((tid >> 2) >> 2) & 3)) << 4);
    int aoff[2], boff[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int r = wr * 32 + fr;
        aoff[ks] = r * 64 + (((2 * ks + fh) ^ ((r >> 2) & 3)) << 4);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int rw = wc * 64 + cb * 32 + fr;
            boff[cb][ks] = rw * 64 + (((2 * ks + fh) ^ ((rw >> 2) & 3)) << 4);
        }
    }
    // A in flight in registers: [half][tile parity][2 x 16 B]
    f32x4 ar[2][2][2];
    auto load_a = [&](int h, int t, int set) {
        const bool second = TWO && t < nT2;                           // (selects, not branches: the phases stay straight-line code)
        // (K tiles past the last one -- the final trip's look-ahead -- fetch whatever follows: split, stored, never multiplied)
        const __amdgpu_buffer_rsrc_t rs = second ? rsA2 : rsA;
        const unsigned vo = TWO ? voff_a(h, second) : voffA[h], so = (unsigned)(second ? t : t - nT2) * 128u;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            ar[h][set][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, vo, so + i * 16u, 0));
    };
    // split this thread's 8 values ONCE and store them into the (hi, lo) planes of the half-tile's slot
    auto convert_a = [&](int h, int set, int buf, int t = 0) {         // t: the K tile the values belong to (TWO: which operand's scale)
        unsigned hi[4], lo[4];
        const float sc = (TWO && t < nT2) ? a_sc2 : a_sc;
        split2h_pair(ar[h][set][0][0], ar[h][set][0][1], sc, hi[0], lo[0]);
        split2h_pair(ar[h][set][0][2], ar[h][set][0][3], sc, hi[1], lo[1]);
        split2h_pair(ar[h][set][1][0], ar[h][set][1][1], sc, hi[2], lo[2]);
        split2h_pair(ar[h][set][1][2], ar[h][set][1][3], sc, hi[3], lo[3]);
        unsigned char* slot = lds + buf * BUF + (h ? OFF_AH1 : OFF_AH0) + cv_off;
        *(u32x4*)slot = (u32x4){hi[0], hi[1], hi[2], hi[3]};
        *(u32x4*)(slot + A_HALF / 2) = (u32x4){lo[0], lo[1], lo[2], lo[3]};
    };
    auto dma_w = [&](int h, int t, int buf) {
        const bool second = TWO && t < nT2;
        glds16(second ? rsW2 : rsW, lds + buf * BUF + (h ? OFF_BH1 : OFF_BH0) + wid * 1024, TWO ? voff_w(h, second) : voffW[h],
               (unsigned)(second ? t : t - nT2) * 64u);
    };
    f32x16 acc[2][4];                                                 // [A half (32 rows)][W half * 2 + column block]
    u32x4 fah[2], fal[2];                                             // [ks]
    u32x4 fb0[2][2], fb1[2][2];                                       // [column block][ks]

    // One phase.  j = phase within the loop trip (static), t2 = the trip's first K tile (even).  Tile t = t2 + (j >> 2) lives
    // in buffer (j >> 2) & 1; the wave walks its 64 x 128 block as quadrants (A0,B0) (A0,B1) (A1,B1) (A1,B0).  Besides its
    // fragment reads a phase does one piece of staging for later tiles:
    //   p0: convert Ah1(t) -> LDS (read at p2);  DMA Bh1(t + 1) (read at its tile's p1, five phases on)
    //   p1: load Ah0(t + 2) into registers (converted at p2 of t + 1)
    //   p2: convert Ah0(t + 1) -> LDS (read at its p0);  load Ah1(t + 2) into registers (converted at p0 of t + 2);
    //       vmcnt(5) retires the DMA of the last p3
    //   p3: DMA Bh0(t + 2);  vmcnt(5) retires the DMA of this tile's p0
    // (5 = the vector-memory instructions issued after the DMA being retired: 1 + 2 + 2 or 2 + 2 + 1.)  Converted data is
    // read TWO phases after it is stored: the storing wave's lgkmcnt(0) sits behind its first barrier, and the other wave
    // group runs one barrier behind (storing one phase ahead and waiting before the barrier stretched every other interval).
    auto phase = [&](int j, int t2) {
        const int ph = j & 3, buf = (j >> 2) & 1, t = t2 + (j >> 2);
        const unsigned char* base = lds + buf * BUF;
        if (ph == 0) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) fb0[cb][ks] = *(const u32x4*)(base + OFF_BH0 + boff[cb][ks]);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                fah[ks] = *(const u32x4*)(base + OFF_AH0 + aoff[ks]);
                fal[ks] = *(const u32x4*)(base + OFF_AH0 + A_HALF / 2 + aoff[ks]);
            }
            convert_a(1, buf, buf, t);
            __builtin_amdgcn_sched_barrier(0);
            dma_w(1, t + 1, buf ^ 1);
        } else if (ph == 1) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) fb1[cb][ks] = *(const u32x4*)(base + OFF_BH1 + boff[cb][ks]);
            __builtin_amdgcn_sched_barrier(0);
            load_a(0, t + 2, buf);
        } else if (ph == 2) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                fah[ks] = *(const u32x4*)(base + OFF_AH1 + aoff[ks]);
                fal[ks] = *(const u32x4*)(base + OFF_AH1 + A_HALF / 2 + aoff[ks]);
            }
            convert_a(0, buf ^ 1, buf ^ 1, t + 1);
            __builtin_amdgcn_sched_barrier(0);
            load_a(1, t + 2, buf);
        } else {
            dma_w(0, t + 2, buf);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (ph >= 2) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        const int ai = (ph >= 2) ? 1 : 0, bj = (ph == 1 || ph == 2) ? 1 : 0;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)                            // (lo, w) first, then (hi, w)
                acc[ai][2 * bj + cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fal[ks]), __builtin_bit_cast(f16x8, bj ? fb1[cb][ks] : fb0[cb][ks]),
                                                                              acc[ai][2 * bj + cb], 0, 0, 0);
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
                acc[ai][2 * bj + cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fah[ks]), __builtin_bit_cast(f16x8, bj ? fb1[cb][ks] : fb0[cb][ks]),
                                                                              acc[ai][2 * bj + cb], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };
    const int wm0 = wr * 64, wn0 = wc * 128;
    float out_amax = 0.f;

    // Work items of this workgroup: its whole tiles (all K tiles), then -- if there are cut tiles and this workgroup's index is below
    // n_cut * n_slices -- one slice: a cut tile's loop trips [s T / S, (s + 1) T / S)
    const int n_whole = t_lo + slot_in_xcd < t_hi ? (t_hi - t_lo - slot_in_xcd + wg_per_xcd - 1) / wg_per_xcd : 0;
    const int n_sl_all = TWO ? 0 : p.n_cut * p.n_slices;              // slices: number blockIdx.x + j * gridDim.x goes to this workgroup
    const int n_sl = (int)blockIdx.x < n_sl_all ? (n_sl_all - 1 - (int)blockIdx.x) / nwg + 1 : 0;
    const int n_items = n_whole + n_sl;
    int tb = 0, te = nT, slice_id = 0;                                // the current item's K tiles [tb, te), both even; its slice number
    auto set_item = [&](int k) {
        if (k < n_whole) { tb = 0; te = nT; set_tile(t_lo + slot_in_xcd + k * wg_per_xcd); }
        else {
            slice_id = blockIdx.x + (k - n_whole) * nwg;
            const int l = slice_id / p.n_slices, sl = slice_id - l * p.n_slices, T = nT >> 1;
            tb = 2 * (sl * T / p.n_slices); te = 2 * ((sl + 1) * T / p.n_slices);
            set_tile(p.n_full + l);
        }
    };
    // the W half-tiles an item needs before its first phases: Bh0, Bh1 of its first K tile into buffer 0, Bh0 of the second into buffer 1
    auto prologue_w = [&]() { dma_w(0, tb, 0); dma_w(1, tb, 0); dma_w(0, tb + 1, 1); };
    if (n_items > 0) { set_item(0); prologue_w(); }
    for (int k = 0; k < n_items; ++k) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        // the activations of the first K tiles go through registers: Ah0 of the first is converted here, its Ah1 and the second tile's halves
        // wait in their register sets for the phases that convert them (p0 / p2 of the first tile, p0 of the second)
        load_a(0, tb, 0); load_a(1, tb, 0); load_a(0, tb + 1, 1); load_a(1, tb + 1, 1);
        convert_a(0, 0, 0, tb);
        // everything older than those loads (the W prologue issued before the previous item's epilogue, that epilogue's loads and
        // stores) has been retired by the wait the conversion needed; the LDS stores are waited for before the barrier
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (grp == 1) __builtin_amdgcn_s_barrier();                   // the second wave group runs one barrier behind
        for (int t2 = tb; t2 < te; t2 += 2) {
            if (TWO && t2 == nT2) {                                   // the second pair's sums -> the main pair's scale (both K2 / 32 and t2 are even)
                const f32x4 rr = *(const f32x4*)(lds + 2 * BUF + tid * 16);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[i][j][r] *= rr[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) phase(j, t2);
        }
        if (grp == 0) __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the look-ahead loads / DMA past this item's last K tile
        __builtin_amdgcn_s_barrier();                                 // every wave is done with the ring
        const int em0 = m0, en0 = n0, e_slice = slice_id;
        const bool whole = k < n_whole;
        if (k + 1 < n_items) { set_item(k + 1); prologue_w(); }       // in flight during the epilogue below
        if (!whole) {                                                 // a slice: the raw accumulators, 16 B per thread and store
            const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ws + (size_t)e_slice * (128 * 512)), 0, 128 * 512 * 4, 0x00020000);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const u32x4 v = {__float_as_uint(acc[i][j][4 * q]), __float_as_uint(acc[i][j][4 * q + 1]), __float_as_uint(acc[i][j][4 * q + 2]),
                                         __float_as_uint(acc[i][j][4 * q + 3])};
                        __builtin_amdgcn_raw_buffer_store_b128(v, rsP, (unsigned)tid * 16u, (unsigned)(((i * 4 + j) * 4 + q) * 8192), 0);
                    }
            continue;
        }

        // epilogue straight from the accumulators: lane (fr, fh) holds column 32 j + fr of the wave's 128 (blocks j = 0..3), rows
        // 32 i + (r & 3) + 8 (r >> 2) + 4 fh: one register = two 128-B row segments per wave instruction.  Rows >= M fall off
        // the descriptors' extents; residual in units of (row block, column-block pair), the next unit's in flight behind
        // the current one.
        const int mw = em0 + wm0;
        const long long rows_left = (long long)p.M - mw;
        const __amdgpu_buffer_rsrc_t rsC = desc(p.c, rows_left > 0 ? ((rows_left - 1) * p.ldc + p.N) * 4 + (long long)mw * p.ldc * 4 : 0,
                                                (long long)mw * p.ldc * 4);
        const __amdgpu_buffer_rsrc_t rsR = RES ? desc(p.res, rows_left > 0 ? ((rows_left - 1) * p.ldr + p.N) * 4 + (long long)mw * p.ldr * 4 : 0,
                                                      (long long)mw * p.ldr * 4)
                                               : rsW0;
        unsigned vc[4], vr[4];
        float sv[4], bv[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = en0 + wn0 + 32 * j + fr;
            vc[j] = (unsigned)((4 * fh * p.ldc + n) * 4); vr[j] = (unsigned)((4 * fh * p.ldr + n) * 4);
            sv[j] = (p.oscale ? p.oscale[n] : 1.f) * acc_scale;
            bv[j] = p.bias ? p.bias[n] : 0.f;
        }
        float rv[RES ? 2 : 1][2][16];
        // Rows >= M of a ragged last tile are masked BY LANE (an out-of-range voffset), not left to the descriptor's extent:
        // their row offset travels in soffset, which can exceed num_records there, and the range check is
        // `offset >= num_records - soffset`.  A full tile (FULL) pays nothing for it.
        const int row_lim = (int)(rows_left < 1024 ? rows_left : 1024) - 4 * fh;     // row u of this lane is valid iff u < row_lim
        auto epilogue = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            auto load_res = [&](int u, int slot) {                    // unit u = (i = u >> 1, column blocks 2 (u & 1), + 1)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ru = 32 * (u >> 1) + (r & 3) + 8 * (r >> 2);
                        const unsigned vo = (FULL || ru < row_lim) ? vr[2 * (u & 1) + jj] : 0x80000000u;
                        rv[slot][jj][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, vo, (unsigned)(ru * p.ldr * 4), 0));
                    }
            };
            if constexpr (RES) load_res(0, 0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if constexpr (RES) { if (u + 1 < 4) load_res(u + 1, (u + 1) & 1); }
                const int i = u >> 1;
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int j = 2 * (u & 1) + jj;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        float v = (acc[i][j][r] * sv[j] + bv[j]) * p.alpha;
                        if constexpr (RES) v += rv[u & 1][jj][r];
                        if (ACT == DBMM_ACT_RELU) v = fmaxf(v, 0.f);
                        else if (ACT == DBMM_ACT_QUICKGELU) v = v / (1.f + expf(-1.702f * v));
                        const int ru = 32 * i + (r & 3) + 8 * (r >> 2);
                        const bool valid = FULL || ru < row_lim;
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsC, valid ? vc[j] : 0x80000000u,
                                                              (unsigned)(ru * p.ldc * 4), 0);
                        if (valid) out_amax = fmaxf(out_amax, fabsf(v));
                    }
                }
            }
        };
        if (rows_left >= 64) epilogue(std::true_type{}); else epilogue(std::false_type{});
    }
    if (p.c_absmax) {                                                 // one (filtered) atomic per workgroup
        out_amax = wave_max(out_amax);
        __syncthreads();                                              // all DMA drained (loop exit), the ring is free
        float* red = (float*)lds;
        if (lane == 0) red[wid] = out_amax;
        __syncthreads();
        if (tid == 0) {
            float m = red[0];
#pragma unroll
            for (int i = 1; i < 8; ++i) m = fmaxf(m, red[i]);
            if (m > *(volatile const float*)p.c_absmax) atomicMax((unsigned*)p.c_absmax, __float_as_uint(m));
        }
    }
}

}  // namespace

namespace {
// The cut tiles: sum the slices' accumulators in slice order (same thread <-> element mapping as the main kernel), then the main kernel's
// epilogue for ONE 32 x 32 block (i, j) of every wave's 64 x 128 per workgroup -- grid (n_cut, 8).
template <int ACT, int RES>
__global__ __launch_bounds__(512) void gemm_pair_8ph_fixup_kernel(const PairP p) {
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
    const int n = n0 + wc * 128 + 32 * bj + fr, mw = m0 + wr * 64;
    const float sv = (p.oscale ? p.oscale[n] : 1.f) * pow2f(-s_a - p.w_exp), bv = p.bias ? p.bias[n] : 0.f;
    const long long rows_left = (long long)p.M - mw;
    const __amdgpu_buffer_rsrc_t rsC = desc(p.c, rows_left > 0 ? ((rows_left - 1) * p.ldc + p.N) * 4 + (long long)mw * p.ldc * 4 : 0, (long long)mw * p.ldc * 4);
    const __amdgpu_buffer_rsrc_t rsR = RES ? desc(p.res, rows_left > 0 ? ((rows_left - 1) * p.ldr + p.N) * 4 + (long long)mw * p.ldr * 4 : 0,
                                                  (long long)mw * p.ldr * 4)
                                           : __builtin_amdgcn_make_buffer_rsrc((void*)p.c, 0, 0, 0x00020000);
    const unsigned vc = (unsigned)((4 * fh * p.ldc + n) * 4), vr = (unsigned)((4 * fh * p.ldr + n) * 4);
    const int row_lim = (int)(rows_left < 1024 ? rows_left : 1024) - 4 * fh;
    float rv[16], out_amax = 0.f;
    if constexpr (RES) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ru = 32 * bi + (r & 3) + 8 * (r >> 2);
            rv[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsR, ru < row_lim ? vr : 0x80000000u, (unsigned)(ru * p.ldr * 4), 0));
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float v = (a[r] * sv + bv) * p.alpha;
        if constexpr (RES) v += rv[r];
        if (ACT == DBMM_ACT_RELU) v = fmaxf(v, 0.f);
        else if (ACT == DBMM_ACT_QUICKGELU) v = v / (1.f + expf(-1.702f * v));
        const int ru = 32 * bi + (r & 3) + 8 * (r >> 2);
        const bool valid = ru < row_lim;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsC, valid ? vc : 0x80000000u, (unsigned)(ru * p.ldc * 4), 0);
        if (valid) out_amax = fmaxf(out_amax, fabsf(v));
    }
    if (p.c_absmax) {
        out_amax = wave_max(out_amax);
        if (lane == 0) red[wid] = out_amax;
        __syncthreads();
        if (tid == 0) {
            float m = red[0];
#pragma unroll
            for (int i = 1; i < 8; ++i) m = fmaxf(m, red[i]);
            if (m > *(volatile const float*)p.c_absmax) atomicMax((unsigned*)p.c_absmax, __float_as_uint(m));
        }
    }
}

// Tile quantisation: the persistent grid works in rounds of 256 tiles.  With a workspace the tiles of a short last round are cut along K into
// S slices (dbmm_cut_slices, common.h) dealt over the workgroups, and a small second launch sums the slices and runs the epilogue (as
// conv3x3_halo8.hip does).
int pair_8ph_launch(PairP& p, int act, bool two, void* workspace, size_t workspace_bytes, void* stream) {
    p.tiles_n = p.N / 256;
    p.n_tiles = ((p.M + 255) / 256) * p.tiles_n;
    p.n_full = p.n_tiles; p.n_cut = 0; p.n_slices = 1; p.ws = nullptr;
    const int rem = p.n_tiles % 256, trips = p.K / 64;
    if (!two && workspace && dbmm_aligned16(workspace) && p.n_tiles > 256 && rem != 0) {
        const int S = dbmm_cut_slices(rem, trips);
        if (S >= 2 && (size_t)rem * S * (128 * 512 * sizeof(float)) <= workspace_bytes) {
            p.n_full = p.n_tiles - rem; p.n_cut = rem; p.n_slices = S; p.ws = (float*)workspace;
        }
    }
    const int grid = p.n_tiles < 256 ? p.n_tiles : 256;               // persistent: one workgroup per CU
    hipStream_t s = (hipStream_t)stream;
#define DBMM_P8(A, R, T) hipLaunchKernelGGL((gemm_pair_8ph_kernel<A, R, T>), dim3(grid), dim3(512), 0, s, p)
    if (two) { if (act == 0) DBMM_P8(0, 0, 1); else if (act == 1) DBMM_P8(1, 0, 1); else return DBMM_E_UNSUPPORTED; }
    else if (p.res) { if (act == 0) DBMM_P8(0, 1, 0); else if (act == 1) DBMM_P8(1, 1, 0); else DBMM_P8(2, 1, 0); }
    else { if (act == 0) DBMM_P8(0, 0, 0); else if (act == 1) DBMM_P8(1, 0, 0); else DBMM_P8(2, 0, 0); }
#undef DBMM_P8
    DBMM_CHECK_LAUNCH();
    if (p.n_cut) {
#define DBMM_P8F(A, R) hipLaunchKernelGGL((gemm_pair_8ph_fixup_kernel<A, R>), dim3(p.n_cut, 8), dim3(512), 0, s, p)
        if (p.res) { if (act == 0) DBMM_P8F(0, 1); else if (act == 1) DBMM_P8F(1, 1); else DBMM_P8F(2, 1); }
        else { if (act == 0) DBMM_P8F(0, 0); else if (act == 1) DBMM_P8F(1, 0); else DBMM_P8F(2, 0); }
#undef DBMM_P8F
        DBMM_CHECK_LAUNCH();
    }
    return DBMM_OK;
}
}  // namespace

// common.h: dbmm_gemm_pair_8ph with a workspace for the K cut of a short last round (null: one launch)
int dbmm_gemm_pair_8ph_ws(const float* a, int64_t lda, const float* a_absmax, const void* w_plane_f16, int w_exp, int64_t ldw,
                          const float* out_scale, const float* bias, const float* residual, int64_t ldr, float* c, int64_t ldc,
                          float* c_absmax, int64_t M, int64_t N, int64_t K, float alpha, int act, void* workspace, size_t workspace_bytes, void* stream) {
    if (!a || !a_absmax || !w_plane_f16 || !c) return DBMM_E_ARG;
    if (M <= 0 || N <= 0 || K <= 0 || M > INT32_MAX || N > INT32_MAX || K > INT32_MAX) return DBMM_E_SHAPE;
    if (act < 0 || act > 2) return DBMM_E_ARG;
    if ((N % 256) || (K % 64) || w_exp < -40 || w_exp > 40) return DBMM_E_UNSUPPORTED;
    if ((lda & 3) || (ldw & 7) || !dbmm_aligned16(a) || !dbmm_aligned16(w_plane_f16) || !dbmm_aligned16(c) ||
        (residual && !dbmm_aligned16(residual)))
        return DBMM_E_ALIGN;
    const long long wb = ((N - 1) * ldw + K) * 2;
    if (wb >= EXT_LIM || 256 * lda * 4 >= EXT_LIM || 256 * ldc * 4 >= EXT_LIM || (residual && 256 * ldr * 4 >= EXT_LIM)) return DBMM_E_UNSUPPORTED;
    PairP p{};
    p.a = a; p.a_absmax = a_absmax; p.w = (const unsigned short*)w_plane_f16; p.oscale = out_scale; p.bias = bias; p.res = residual;
    p.c = c; p.c_absmax = c_absmax;
    p.lda = lda; p.ldw = ldw; p.ldr = ldr; p.ldc = ldc; p.a_total = ((M - 1) * lda + K) * 4; p.w_total = wb;
    p.M = (int)M; p.N = (int)N; p.K = (int)K; p.w_exp = w_exp; p.alpha = alpha;
    return pair_8ph_launch(p, act, false, workspace, workspace_bytes, stream);
}

// see include/dbmm.h
extern "C" int dbmm_gemm_pair_8ph(const float* a, int64_t lda, const float* a_absmax, const void* w_plane_f16, int w_exp, int64_t ldw,
                                  const float* out_scale, const float* bias, const float* residual, int64_t ldr, float* c, int64_t ldc,
                                  float* c_absmax, int64_t M, int64_t N, int64_t K, float alpha, int act, void* stream) {
    return dbmm_gemm_pair_8ph_ws(a, lda, a_absmax, w_plane_f16, w_exp, ldw, out_scale, bias, residual, ldr, c, ldc, c_absmax, M, N, K, alpha, act, nullptr, 0,
                                 stream);
}

// common.h: the dual-source GEMM of dbmm_gemm_dual_bn_act_x2 on this kernel (TWO = 1).  DBMM_E_UNSUPPORTED: not this kernel's shape.
int dbmm_gemm_dual_pair_8ph(const float* a, int64_t lda, const float* a_absmax, const void* w_plane_f16, int w_exp, int64_t ldw, int64_t K,
                            const float* out_scale, const float* a2, int64_t lda2, const float* a2_absmax, const void* w2_plane_f16, int64_t ldw2,
                            int64_t K2, const float* ratio, const float* bias, float* c, int64_t ldc, float* c_absmax, int64_t M, int64_t N, int act,
                            void* stream) {
    if ((N % 256) || (K % 64) || (K2 % 64) || M < 16384 || (act != 0 && act != 1)) return DBMM_E_UNSUPPORTED;
    const long long wb = ((N - 1) * ldw + K) * 2, wb2 = ((N - 1) * ldw2 + K2) * 2;
    if (wb >= EXT_LIM || wb2 >= EXT_LIM || 256 * lda * 4 >= EXT_LIM || 256 * lda2 * 4 >= EXT_LIM || 256 * ldc * 4 >= EXT_LIM) return DBMM_E_UNSUPPORTED;
    PairP p{};
    p.a = a; p.a_absmax = a_absmax; p.w = (const unsigned short*)w_plane_f16; p.oscale = out_scale; p.bias = bias; p.res = nullptr;
    p.c = c; p.c_absmax = c_absmax;
    p.lda = lda; p.ldw = ldw; p.ldr = 0; p.ldc = ldc; p.a_total = ((M - 1) * lda + K) * 4; p.w_total = wb;
    p.M = (int)M; p.N = (int)N; p.K = (int)K; p.w_exp = w_exp; p.alpha = 1.f;
    p.a2 = a2; p.a2_absmax = a2_absmax; p.w2 = (const unsigned short*)w2_plane_f16; p.ratio = ratio;
    p.lda2 = lda2; p.ldw2 = ldw2; p.a2_total = ((M - 1) * lda2 + K2) * 4; p.w2_total = wb2; p.K2 = (int)K2;
    return pair_8ph_launch(p, act, true, nullptr, 0, stream);
}
