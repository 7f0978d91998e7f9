// conv3 + residual of one bottleneck block CHAINED with conv1 of the next block, layer-3 geometry (clip/model.py:42-55 back to back):
//
//   x'  = relu( (y2 @ W3^T) * sc3 + b3 + x )        y2 [M][256], W3 [N][256], x / x' [M][N] (N = 1024)
//   y1' = relu( (x' @ W1^T) * sc1 + b1 )            W1 [256][N], y1' [M][256]
//
// Why a second chain kernel.  bottleneck_chain_kernel keeps the y2 tile as (hi, lo) A fragments in registers for the whole tile and
// the conv1' accumulators of all P output channels beside them; at K = P = 256 that is 128 + 128 registers per lane: one wave per
// SIMD, and the launch ran 1.15 ms against 0.68 ms for the two separate launches (bottleneck_chain.hip, "tried and dropped").  The
// two separate launches move 2.88 GB per layer-3 block at B = 1024 (conv3: y2 + residual + x'; conv1: x' again + y1'), 0.82 GB of it
// the re-read of x' -- at 3.6 - 4.6 TB/s they are the top family of the step.
//
// Here the work of one 128-row tile is split over EIGHT waves (512 threads, two waves per SIMD) as 4 row groups x 2:
//   * conv3, K-split inside a wave pair: wave (rg, ch) holds the A fragments of rows rg*32.. for K half ch only (64 registers) and
//     accumulates a PARTIAL 32 x 64 slab over its 128 k values; the pair swaps the halves of the slab it does not finish through LDS
//     (wave ch finishes slab columns ch*32 .. +31: partial + partner's partial, BatchNorm, residual, ReLU, store, slab to LDS);
//   * conv1', column-split: wave (rg, ch) accumulates y1' rows rg*32.., columns ch*128 .. +127 (64 registers) over the whole slab.
// 64 + 64 + 32 + 16 registers of fragments / accumulators / residual leave room for two waves per SIMD, and every wave still issues
// 32 + 32 MFMAs per slab with 32 KB of fragment reads (the same MFMA : LDS ratio as the four-wave kernel).
//
// Schedule.  A slab is three phases per wave: C3 (conv3 partial, 32 MFMAs), EP (combine + epilogue: vector ALU, LDS, memory), C1
// (conv1', 32 MFMAs), each ended by a workgroup barrier.  With all eight waves in the same phase the matrix pipe idles through every
// EP and barrier (first version, measured with in-kernel stamps: 44 % of a slab in the two MFMA phases, 24 % in EP, 30 % in
// barriers; 0.81 ms against 0.69 ms for the two separate launches).  So the two waves of a SIMD run ONE PHASE APART: waves 0-3
// (row groups 0, 1; "group A") and waves 4-7 (row groups 2, 3; "group B", one tick behind) -- while one does EP the other owns the
// matrix pipe.  The groups share only the weights: W3 slabs [64][256] and W1 chunks [256][64] (fp16, exact planes) form ONE stream of
// 32-KB blocks (W3(0), W1(0), W3(1), ...) through a ring of THREE LDS buffers; group A fetches the W3 blocks, group B the W1 blocks,
// by LDS-DMA two ticks ahead of the first reader, each right after it has consumed its pending register loads (the compiler waits
// vmcnt(0) at the first use of a tracked load while a DMA is in flight), with counted waits.
// The pair's exchange goes through the slab itself (the partner's partial sits where the finished value will go); the slab's fp16
// scale is the running maximum of the row group's finished slabs, kept in LDS by ds_max.
// Residual loads / x' / y1' stores straight from / to the MFMA accumulator layout (one register = two 128-B row segments); the next
// slab's residual is in flight from one EP to the next.
//
// MEASURED (MI355X, B = 1024, layer-3 geometry; profiles/r04_chain8_*.log): correct (tests/test_gpu_kernels.py, test_gpu_headline.py)
// but NOT faster than the two launches it replaces: 0.754 ms against 0.686 ms (all waves in phase: 0.805 ms).  The in-kernel stamps
// say why: a slab costs ~8,600 cycles against 4,096 of MFMA; EP takes ~2,400 - 2,900 cycles whether or not the partner wave computes,
// and its cost is the ISSUE of vector-memory instructions -- 28 cycles per dword-per-lane load / store, 60 per 1-KB LDS-DMA piece, 84
// per dwordx4 load: proportional to bytes (a row-domain EP with quad-transposed dwordx4 accesses, 4 x fewer instructions, measured
// 0.782 ms).  Per slab a CU moves 64 KB of residual + x' AND 64 KB of weights: the weight stream from L2 (512 KB + 512 KB per 128
// rows, the same as the separate launches pay) doubles the bytes through the CU's vector-memory path, and chaining only removes the
// x' re-read = 18 % of that path's traffic while adding three barriers per slab.  The lever for this family is taller tiles (fewer
// weight bytes per row), not chaining.  Library option chain8 (default 0) routes the layer-3 shapes here.
//
// Arithmetic = the fp16-pair path (fp32 value = fp16 hi + lo under an exact power-of-two scale, weights exact in one fp16 plane,
// fp32 accumulate).  Bound: HBM, 4 * (K + 2 N + P) bytes per pixel row.
#include <stdlib.h>
#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOR = 0x80000000u;
constexpr long long EXT_LIM = 0x7FFFFFF0LL;

struct Chain8P {
    const float* a; const float* a_absmax;                              // y2 [M][256]
    const u16* w3; int w3_exp; const float* sc3; const float* b3;       // [N][256] fp16 plane of W3 * 2^w3_exp
    const float* res; float* x; float* x_absmax;                        // residual [M][N], x' [M][N]
    const u16* w1; int w1_exp; const float* sc1; const float* b1;       // [256][N] fp16 plane of W1 * 2^w1_exp
    float* y1; float* y1_absmax;                                        // [M][256]
    int M, N;
};

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
__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t r, unsigned char* lds_dst, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
#else
    (void)r; (void)lds_dst; (void)voff; (void)soff;
#endif
}
// one ds_max_u32 per lane on ONE LDS word (the hardware serialises the 64 lanes; atomicMax() would be expanded by the compiler into a
// 64-trip scan loop over the lanes first)
__device__ __forceinline__ void lds_max_u32(unsigned* word, unsigned v) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned a = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned*)word;
    asm volatile("ds_max_u32 %0, %1" :: "v"(a), "v"(v) : "memory");
#else
    (void)word; (void)v;
#endif
}
// workgroup barrier that leaves vector-memory operations (LDS-DMA, residual loads, stores) in flight: this wave's LDS accesses
// are complete (lgkmcnt), nothing else is waited for
__device__ __forceinline__ void bar() {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#endif
}

#ifdef CH8_STAMP
// developer build only (-DCH8_STAMP): cycles waves 0 (group A) and 4 (group B) of one workgroup spend in each tick and barrier
__device__ long long g_ch8_stamps[24];
#define STAMP(i) do { if (stamp_on) { const long long now__ = clock64(); st[i] += now__ - tprev; tprev = now__; } } while (0)
#define ESTAMP(i) do { if (stamp_on) { const long long now__ = clock64(); est[i] += now__ - eprev; eprev = now__; } } while (0)
#define ESTART() do { if (stamp_on) eprev = clock64(); } while (0)
#else
#define STAMP(i) do { } while (0)
#define ESTAMP(i) do { } while (0)
#define ESTART() do { } while (0)
#endif

constexpr int K = 256, P = 256, BM = 128, BNS = 64;
constexpr int SLROW = BNS + 4;                                   // slab row pitch in floats (272 B: conflict-free b128 rows)
constexpr int WB_BYTES = 32768;                                  // one weight block: W3 slab [64][256] or W1 chunk [256][64], fp16
constexpr int OFF_SLAB = 3 * WB_BYTES;                           // [4 row groups][32][68] fp32 (34,816 B); the prologue's y2 planes alias it
constexpr int SLAB_BYTES = 4 * 32 * SLROW * 4;
constexpr int OFF_MAX = OFF_SLAB + SLAB_BYTES;                   // running maximum per row group (4 words) + the final reduction
constexpr int LDS_TOTAL = OFF_MAX + 128;                         // 133,248 B

__global__ __launch_bounds__(512, 1) void bottleneck_chain8_kernel(const Chain8P p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_TOTAL];
    u16* Ay = (u16*)(lds + OFF_SLAB);                              // prologue: [2 planes][128][64]
    float* slab = (float*)(lds + OFF_SLAB);
    unsigned* maxw = (unsigned*)(lds + OFF_MAX);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), rg = wave >> 1, ch = wave & 1, grp = wave >> 2, gw = wave & 3;
    const int fr = lane & 31, fh = lane >> 5;
    const int m0 = xcd_remap(blockIdx.x, gridDim.x) * BM;
    const int NT = p.N / BNS;
    const long long Mll = p.M;

    // ---- descriptors rebased to the tile's first row (tensors may exceed 2 GiB) -----------------------------------------------
    const __amdgpu_buffer_rsrc_t rsA = desc(p.a, Mll * K * 4, (long long)m0 * K * 4);
    const __amdgpu_buffer_rsrc_t rsR = desc(p.res, Mll * p.N * 4, (long long)m0 * p.N * 4);
    const __amdgpu_buffer_rsrc_t rsX = desc(p.x, Mll * p.N * 4, (long long)m0 * p.N * 4);
    const __amdgpu_buffer_rsrc_t rsY = desc(p.y1, Mll * P * 4, (long long)m0 * P * 4);
    // group A stages the W3 blocks, group B the W1 blocks: one descriptor and one set of lane offsets per wave
    const __amdgpu_buffer_rsrc_t rsW = grp == 0 ? desc(p.w3, (long long)p.N * K * 2, 0) : desc(p.w1, (long long)P * p.N * 2, 0);

    // this lane's 16 accumulator rows = 4 groups (t = r >> 2) of 4 consecutive tile rows rg*32 + 8t + 4fh + q; M % 4 == 0, so a
    // group is valid or not as a whole; the row step q goes into the scalar offset of every access
    unsigned gx[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int row = rg * 32 + 8 * t + 4 * fh;
        gx[t] = m0 + row < p.M ? (unsigned)row * (unsigned)(p.N * 4) + (unsigned)(fr * 4) : OOR;
    }

    // ---- weight staging by LDS-DMA: a 32-KB block = 32 wave-instructions of 1 KB, 8 per wave of the staging group ---------------
    // W3 block (512-B rows, two per instruction): lane -> row 2q + lane / 32, 16-B slot lane % 32, source chunk slot ^ (row & 15)
    // W1 block (128-B rows, eight per instruction): lane -> row 8q + lane / 8, slot lane % 8, source chunk slot ^ ((row >> 1) & 7)
    // (the lane offsets are recomputed per block -- a few integer instructions per 1-KB instruction -- instead of living in registers)
    // block j of the stream: j even = W3 slab j / 2 (group A), j odd = W1 chunk j / 2 (group B); ring buffer j % 3
    auto dma_block = [&](int j) {
        const unsigned soff = grp == 0 ? (unsigned)((j >> 1) * BNS) * (unsigned)(K * 2) : (unsigned)((j >> 1) * BNS * 2);
        unsigned char* dst = lds + (j % 3) * WB_BYTES + gw * 8192;
        int ln;                                                      // (opaque copy of the lane id: keeps the offsets out of loop-invariant registers)
        asm volatile("v_mov_b32 %0, %1" : "=v"(ln) : "v"(lane));
        if (grp == 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r3 = 2 * (gw * 8 + i) + (ln >> 5), c3 = (ln & 31) ^ (r3 & 15);
                glds16(rsW, dst + i * 1024, (unsigned)r3 * (unsigned)(K * 2) + (unsigned)c3 * 16u, soff);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r1 = 8 * (gw * 8 + i) + (ln >> 3), c1 = (ln & 7) ^ ((r1 >> 1) & 7);
                glds16(rsW, dst + i * 1024, (unsigned)r1 * (unsigned)(p.N * 2) + (unsigned)c1 * 16u, soff);
            }
        }
    };
    dma_block(grp);                                                  // A: block 0 = W3(0); B: block 1 = W1(0)

    // ---- prologue: y2 tile -> (hi, lo) fp16 planes in LDS, 64 k at a time -> A fragments of this wave's K half ----------------
    // thread (lc = tid & 15, lr = tid >> 4) loads k quad lc of rows lr + 32 i
    u32x4 af[8][2];                                                  // [k step of this wave's K half][plane]
    {
        const int lc = tid & 15, lr = tid >> 4;
        const float a_sc = pow2f(scale_exp(*p.a_absmax));
        f32x4 q[4][4];
        auto load_pass = [&](int kp) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = lr + 32 * i;
                const unsigned off = m0 + row < p.M ? (unsigned)row * (unsigned)(K * 4) + lc * 16u : OOR;
                q[kp][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsA, off, (unsigned)(kp * 64 * 4), 0));
            }
        };
        load_pass(0); load_pass(1);
#pragma unroll
        for (int kp = 0; kp < 4; ++kp) {
            if (kp + 2 < 4) load_pass(kp + 2);                       // two passes in flight
            if (kp) bar();                                           // the previous pass's fragments have been read
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = lr + 32 * i;
                unsigned hp[2], lp[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) split2h_pair(q[kp][i][2 * j], q[kp][i][2 * j + 1], a_sc, hp[j], lp[j]);
                const int off = row * 64 + (((lc >> 1) ^ ((row >> 1) & 7)) << 3) + ((lc & 1) << 2);
                *(u32x2*)(Ay + off) = (u32x2){hp[0], hp[1]};
                *(u32x2*)(Ay + BM * 64 + off) = (u32x2){lp[0], lp[1]};
            }
            bar();
            if ((kp >> 1) == ch) {                                   // wave-uniform: this pass belongs to the wave's K half
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int row = rg * 32 + fr;
                    const int off = row * 64 + (((2 * ks + fh) ^ ((row >> 1) & 7)) << 3);
                    af[(kp & 1) * 4 + ks][0] = *(const u32x4*)(Ay + off);
                    af[(kp & 1) * 4 + ks][1] = *(const u32x4*)(Ay + BM * 64 + off);
                }
            }
        }
    }
    if (tid < 4) maxw[tid] = 0u;
    const int s_a = scale_exp(*p.a_absmax);
    const float acc3_scale = pow2f(-s_a - p.w3_exp);

    // residual of slab nt: this lane's 16 rows, slab columns ch*32 + fr
    float rr[16];
    auto load_res = [&](int nt) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            rr[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                rsR, gx[r >> 2], (unsigned)((nt * BNS + ch * 32) * 4 + (r & 3) * p.N * 4), 0));
    };
    load_res(0);

    f32x16 acc1[4], acc3[2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[j][r] = 0.f;
    float run_max = 0.f, sc3v = 0.f, b3v = 0.f;
    int s_cur = 0;                                                   // exponent the conv1' accumulators are scaled by so far
    float* Ls = slab + rg * (32 * SLROW);
#ifdef CH8_STAMP
    const bool stamp_on = blockIdx.x == 8 && gw == 0;
    long long st[6] = {0, 0, 0, 0, 0, 0}, est[5] = {0, 0, 0, 0, 0}, tprev = 0, eprev = 0;
#endif

    // ---- the three phases of a slab -------------------------------------------------------------------------------------------------
    // C3: partial conv3 over this wave's K half; tile 0 = the slab columns this wave finishes (ch*32 ..), tile 1 = the partner's, which
    // is handed over through the slab itself
    auto C3 = [&](int nt) {
        const int n0 = nt * BNS;
        sc3v = p.sc3[n0 + ch * 32 + fr]; b3v = p.b3[n0 + ch * 32 + fr];          // consumed by EP, one tick on
        const u16* W3b = (const u16*)(lds + ((2 * nt) % 3) * WB_BYTES);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc3[j][r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            u32x4 wf[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int row = ((j ^ ch) << 5) + fr;
                wf[j] = *(const u32x4*)(W3b + row * K + (((2 * (ch * 8 + ks) + fh) ^ (row & 15)) << 3));
            }
#pragma unroll
            for (int pl = 1; pl >= 0; --pl)                            // (lo, w) first, then (hi, w)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc3[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[ks][pl]), __builtin_bit_cast(f16x8, wf[j]),
                                                                     acc3[j], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) Ls[((r & 3) + 8 * (r >> 2) + 4 * fh) * SLROW + (ch ^ 1) * 32 + fr] = acc3[1][r];
    };
    // EP: sum of the two K halves, BatchNorm, residual, ReLU; store x'; finished values into the slab; row-group maximum; stage the
    // group's next weight block; next residual
    auto EP = [&](int nt) {
        const int n0 = nt * BNS;
        ESTART();
        // the residual and the BatchNorm parameters were loaded a tick or more ago; naming them HERE puts the compiler's wait for them
        // in front of the DMA issued below (it does not see hand-written waits and, with an LDS-DMA in flight, waits vmcnt(0) at the
        // first use of a tracked load)
#pragma unroll
        for (int r = 0; r < 16; ++r) asm volatile("" :: "v"(rr[r]));
        asm volatile("" :: "v"(sc3v), "v"(b3v));
        ESTAMP(0);
        if (nt + 1 < NT) dma_block(2 * nt + 2 + grp);               // A: W3(nt + 1), B: W1(nt + 1); first read two ticks on
        ESTAMP(1);
        const float sv = sc3v * acc3_scale, bv = b3v;
        float xp[16];                                                // the partner's partial of tile 0: all 16 reads in flight at once
#pragma unroll
        for (int r = 0; r < 16; ++r) xp[r] = Ls[((r & 3) + 8 * (r >> 2) + 4 * fh) * SLROW + ch * 32 + fr];
        float tmax = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float v = fmaxf(fmaf(acc3[0][r] + xp[r], sv, bv) + rr[r], 0.f);
            if (gx[r >> 2] == OOR) v = 0.f;                          // rows past M: keep the slab clean (their stores are dropped)
            tmax = fmaxf(tmax, v);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsX, gx[r >> 2],
                                                  (unsigned)((n0 + ch * 32) * 4 + (r & 3) * p.N * 4), 0);
            Ls[((r & 3) + 8 * (r >> 2) + 4 * fh) * SLROW + ch * 32 + fr] = v;
        }
        ESTAMP(2);
        lds_max_u32(maxw + rg, __float_as_uint(tmax));                // values are >= 0: their bit patterns order like the values
        ESTAMP(3);
        if (nt + 1 < NT) load_res(nt + 1);                            // in flight until the next EP
        ESTAMP(4);
    };
    // C1: acc1[32 x 128] += slab[32 x 64] . W1[ch*128 .., n0 .. n0+63]^T under the row group's running maximum as fp16 scale
    auto C1 = [&](int nt) {
        const u16* W1b = (const u16*)(lds + ((2 * nt + 1) % 3) * WB_BYTES);
        const float pm = __uint_as_float(__builtin_amdgcn_readfirstlane(maxw[rg]));
        if (pm > run_max) {
            run_max = pm;
            const int s_new = scale_exp(run_max);
            if (s_new != s_cur) {                                    // wave-uniform: the running maximum crossed a binade
                const float f = pow2f(s_new - s_cur);
#pragma unroll
                for (int j = 0; j < 4; ++j)
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
            u32x4 wf[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = ch * 128 + j * 32 + fr;
                wf[j] = *(const u32x4*)(W1b + row * BNS + (((2 * ks + fh) ^ ((row >> 1) & 7)) << 3));
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, al), __builtin_bit_cast(f16x8, wf[j]), acc1[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah), __builtin_bit_cast(f16x8, wf[j]), acc1[j], 0, 0, 0);
        }
    };

    // blocks 0 / 1 and the first residual have landed (this wave's share; the barrier makes it everybody's); the y2 planes are dead
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bar();
#ifdef CH8_STAMP
    tprev = clock64();
#endif
    // ---- ticks: group A runs C3 | EP | C1 of slab nt in ticks 3 nt, 3 nt + 1, 3 nt + 2; group B one tick later ---------------------
    // Block 2 m (W3(m), m >= 1) is issued by A in EP(m - 1) (tick 3 m - 2) and first read in tick 3 m: A waits for it at the end of
    // tick 3 m - 1, where only the 16 residual loads issued after the DMA may still be in flight (loads return in order).  Block
    // 2 m + 1 (W1(m)) is issued by B in EP(m - 1) (tick 3 m - 1) and first read by A in tick 3 m + 2: B waits at the end of tick
    // 3 m + 1, with its 16 residual loads and the two BatchNorm loads of its C3(m) behind the DMA.  A block's buffer was last read one
    // tick before the DMA that refills it is issued.
    // (the two groups run SEPARATE loops -- same barriers in the same order -- so that each path gets its own register allocation)
    if (grp == 0) {
        C3(0);
        bar();
        for (int nt = 0; nt < NT; ++nt) {
            EP(nt);
            STAMP(0);
            bar();
            STAMP(1);
            C1(nt);
            if (nt + 1 < NT) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");      // W3(nt + 1) landed
            STAMP(2);
            bar();
            STAMP(3);
            if (nt + 1 < NT) C3(nt + 1);
            STAMP(4);
            bar();
            STAMP(5);
        }
    } else {
        bar();
        for (int nt = 0; nt < NT; ++nt) {
            C3(nt);
            if (nt > 0) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");           // W1(nt) landed
            STAMP(0);
            bar();
            STAMP(1);
            EP(nt);
            STAMP(2);
            bar();
            STAMP(3);
            C1(nt);
            STAMP(4);
            bar();
            STAMP(5);
        }
    }
#ifdef CH8_STAMP
    if (stamp_on && lane == 0)
    {
        for (int i = 0; i < 6; ++i) g_ch8_stamps[grp * 8 + i] = st[i];
        if (grp == 0) for (int i = 0; i < 5; ++i) g_ch8_stamps[16 + i] = est[i];
    }
#endif

    // ---- y1' = relu(bn1(acc1)) ------------------------------------------------------------------------------------------------
    float y_amax = 0.f;
    const float acc1_scale = pow2f(-s_cur - p.w1_exp);
    unsigned gy[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) gy[t] = gx[t] != OOR ? (unsigned)(rg * 32 + 8 * t + 4 * fh) * (unsigned)(P * 4) + (unsigned)(fr * 4) : OOR;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = ch * 128 + j * 32 + fr;
        const float sv = p.sc1[n] * acc1_scale, bv = p.b1[n];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = fmaxf(fmaf(acc1[j][r], sv, bv), 0.f);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsY, gy[r >> 2],
                                                  (unsigned)((ch * 128 + j * 32) * 4 + (r & 3) * P * 4), 0);
            if (gy[r >> 2] != OOR) y_amax = fmaxf(y_amax, v);
        }
    }
    // ---- maxima for the consumers' fp16 scales: one filtered atomic per workgroup and tensor ----------------------------------------
    y_amax = wave_max(y_amax);
    float* red = (float*)(lds + OFF_MAX + 32);
    if (lane == 0) red[wave] = y_amax;
    bar();
    if (tid == 0) {
        float xm = fmaxf(fmaxf(__uint_as_float(maxw[0]), __uint_as_float(maxw[1])), fmaxf(__uint_as_float(maxw[2]), __uint_as_float(maxw[3])));
        float ym = red[0];
#pragma unroll
        for (int i = 1; i < 8; ++i) ym = fmaxf(ym, red[i]);
        if (p.x_absmax && xm > *(volatile const float*)p.x_absmax) atomicMax((unsigned*)p.x_absmax, __float_as_uint(xm));
        if (p.y1_absmax && ym > *(volatile const float*)p.y1_absmax) atomicMax((unsigned*)p.y1_absmax, __float_as_uint(ym));
    }
}

}  // namespace

#ifdef CH8_STAMP
extern "C" void dbmm_debug_chain8_stamps(long long* out8 /* [24] */) {
    (void)hipDeviceSynchronize();
    (void)hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_ch8_stamps), 24 * sizeof(long long));
}
#endif

// the layer-3 geometry of dbmm_bottleneck_chain_x2 (K = P = 256, no pooled copy); see include/dbmm.h
int dbmm_chain8_launch(const float* y2, const float* y2_absmax, const void* w3_plane_f16, int w3_exp, const float* scale3,
                       const float* bias3, const float* residual, float* x_out, float* x_absmax, const void* w1_plane_f16, int w1_exp,
                       const float* scale1, const float* bias1, float* y1_out, float* y1_absmax, int64_t M, int64_t N, void* stream) {
    if ((N % BNS) != 0 || (M & 3) || N * 512 >= EXT_LIM) return DBMM_E_UNSUPPORTED;
    Chain8P p{};
    p.a = y2; p.a_absmax = y2_absmax;
    p.w3 = (const u16*)w3_plane_f16; p.w3_exp = w3_exp; p.sc3 = scale3; p.b3 = bias3;
    p.res = residual; p.x = x_out; p.x_absmax = x_absmax;
    p.w1 = (const u16*)w1_plane_f16; p.w1_exp = w1_exp; p.sc1 = scale1; p.b1 = bias1;
    p.y1 = y1_out; p.y1_absmax = y1_absmax;
    p.M = (int)M; p.N = (int)N;
    const dim3 grid((unsigned)((M + BM - 1) / BM));
    hipLaunchKernelGGL(bottleneck_chain8_kernel, grid, dim3(512), 0, (hipStream_t)stream, p);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}
