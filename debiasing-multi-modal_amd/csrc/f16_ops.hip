// fp16 mode of the transformer towers (the reference's GPU path: convert_weights, clip/model.py:375-396, and
// `model.half()` semantics, clip/clip.py:139-141): activations are fp16 in HBM, every product is ONE fp16 MFMA
// (v_mfma_f32_32x32x16_f16) with fp32 accumulation, LayerNorm statistics / softmax / bias / QuickGELU / residual adds
// are computed in fp32 and rounded to fp16 once when stored (the reference's LayerNorm subclass also computes in
// fp32, clip/model.py:157-163).  This is the throughput mode BASELINE configs[4] names; the fp32-accurate path
// (igemm_f32.hip) stays the parity mode.
//
//   gemm_f16_kernel      C = act(A . W^T + bias) + R          128x128x64 tiles, register-staged double buffer
//   mha_f16_kernel       softmax(Q K^T / 8) V, head_dim 64     flash style, S^T = K Q^T so a query is a lane
//   layernorm_f16_kernel wave per row, fp32 statistics
//   im2col / tokens / embedding gather / EOT gather            layout kernels with fp16 output
//
// Bounds: GEMM and attention MFMA (2500 TFLOP/s dense fp16), the rest HBM.
#include <stdlib.h>
#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef unsigned short u16;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOR = 0x80000000u;
constexpr long long EXT_LIM = 0x7FFFFFF0LL;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t desc(const void* base, long long total, long long shift) {
    long long ext = total - shift;
    ext = ext < 0 ? 0 : (ext > EXT_LIM ? EXT_LIM : ext);
    return __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)base + shift), 0, (int)ext, 0x00020000);
}
// 128-B LDS rows (64 halves): XOR of the 16-B chunk index, two rows per 256-B bank sweep
__device__ __forceinline__ int swz64(int row) { return (row >> 1) & 7; }

__device__ __forceinline__ unsigned pack2(float a, float b) {
    const f16x2 v = {(_Float16)a, (_Float16)b};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float fast_exp2(float x) {       // v_exp_f32: 1 ulp, denormal results flush to zero
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_exp2f(x);
#else
    return exp2f(x);
#endif
}
__device__ __forceinline__ float act_f(float v, int act) {
    if (act == DBMM_ACT_RELU) return fmaxf(v, 0.f);
    // QuickGELU v * sigmoid(1.702 v) on the hardware exp2 / rcp (1 ulp each; the result is rounded to fp16 anyway).
    // v -> -inf: exp2 -> inf, rcp -> 0, v * 0 = -0; v -> +inf: exp2 -> 0, v * 1.
#if defined(__HIP_DEVICE_COMPILE__)
    if (act == DBMM_ACT_QUICKGELU) return v * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.4554669595930157f * v));
#else
    if (act == DBMM_ACT_QUICKGELU) return v / (1.f + expf(-1.702f * v));
#endif
    return v;
}

// ---------------------------------------------------------------------------------------------------------------
// GEMM
// ---------------------------------------------------------------------------------------------------------------
struct GemmHP {
    const u16* a; const u16* w; const float* bias; const u16* res; u16* c;
    long long lda, ldw, ldr, ldc, a_total, w_total;
    int M, N, K, act, tiles_n, n_tiles;
    const float* oscale;        // optional per-output-channel scale of the accumulator (eval-mode BatchNorm of a 1x1 conv)
    int res_first;              // 0: act(acc * s + b) + residual (transformer blocks); 1: act(acc * s + b + residual) (bottleneck conv3)
    // gemm_f16_8ph_kernel: the tiles of a short last round cut along K (gemm_f16_impl): tiles [0, n_full) whole; tile n_full + l (l < n_cut)
    // by n_slices workgroups over a share of the loop trips each, accumulators left in ws[(l * n_slices + s)][32][512][4] for
    // gemm_f16_8ph_fixup_kernel.  Kernels that do not cut ignore these (n_full is only read by the eight-phase kernel).
    int n_full, n_cut, n_slices;
    float* ws;
    // gemm_f16_8ph_kernel<.., TWO = 1>: a second operand pair (a2 [M][K2], w2 [N][K2]) whose K2 / 64 tiles run FIRST; then the accumulators
    // are multiplied per output column by ratio[n] and the main pair continues (conv3 + downsample branch as one GEMM, as gemm_pair_8ph.hip)
    const u16* a2; const u16* w2; const float* ratio;
    long long lda2, ldw2, a2_total, w2_total;
    int K2;
};

constexpr int GBM = 128;
#ifndef GF16_SCHED
#define GF16_SCHED 1
#endif
#if GF16_SCHED >= 1
#define GF16_FENCE() __builtin_amdgcn_sched_barrier(0)
#else
#define GF16_FENCE()
#endif
#if GF16_SCHED >= 2
#define GF16_FENCE2() __builtin_amdgcn_sched_barrier(0)
#else
#define GF16_FENCE2()
#endif

// GBK = K depth of a chunk: 64 (128-B LDS rows, 64 KB for two stages, 2 workgroups per CU) or 32 (64-B rows, 32 KB, 3 per CU)
template <int GBK>
__device__ __forceinline__ int gswz(int row) { return GBK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); }

// GBN = tile width: 128 (wave tile 64 x 64) or 256 (64 x 128: three LDS fragment reads per four MFMAs instead of four)
template <int GBK, int GBN, int MINB>
__global__ __launch_bounds__(256, MINB) void gemm_f16_kernel(const GemmHP p) {
    constexpr int WTN = GBN / 2, TN = WTN / 32;                   // wave tile width, 32-column MFMA tiles per wave
    constexpr int G_LROW = WTN + 4;                               // epilogue staging pitch (floats) of a wave's 32 x WTN block
    constexpr int G_STAGE = (GBM + GBN) * GBK;                    // halves per stage
    constexpr int EPI_HALVES = 4 * 32 * G_LROW * 2;               // the epilogue staging (fp32) aliases the stages
    constexpr int CPR = GBK / 8, RPP = 256 / CPR, NLD = GBM / RPP, NLW = GBN / RPP;   // 16-B chunks per row, rows per pass, loads per operand
    __shared__ __attribute__((aligned(16))) u16 lds[2 * G_STAGE > EPI_HALVES ? 2 * G_STAGE : EPI_HALVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int tile = xcd_remap(blockIdx.x, p.n_tiles);
    const int m0 = (tile / p.tiles_n) * GBM, n0 = (tile % p.tiles_n) * GBN;
    const int wm0 = (wave >> 1) * 64, wn0 = (wave & 1) * WTN;

    const __amdgpu_buffer_rsrc_t rsA = desc(p.a, p.a_total, (long long)m0 * p.lda * 2);
    const __amdgpu_buffer_rsrc_t rsW = desc(p.w, p.w_total, (long long)n0 * p.ldw * 2);
    // loader: 16-B chunk lc of rows lr + RPP i of both operands
    const int lc = tid % CPR, lr = tid / CPR;
    unsigned a_off[NLD], w_off[NLW];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int r = lr + RPP * i;
        a_off[i] = m0 + r < p.M ? (unsigned)r * (unsigned)(p.lda * 2) + lc * 16u : OOR;
    }
#pragma unroll
    for (int i = 0; i < NLW; ++i) {
        const int r = lr + RPP * i;
        w_off[i] = n0 + r < p.N ? (unsigned)r * (unsigned)(p.ldw * 2) + lc * 16u : OOR;
    }
    // zero-extent twins: chunk loads past the end of K go through them (hardware returns zeros, no memory access), so
    // the loop body has no branches and an odd chunk count costs one phantom chunk of zeros
    const __amdgpu_buffer_rsrc_t rsA0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0, 0x00020000);
    // two register sets: the loads of chunk k+2 are issued while chunk k computes and chunk k+1 (landed) is written to
    // LDS -- one chunk of lead (16 MFMAs = 0.25 us) does not cover an L2 / HBM round trip
    u32x4 a_r0[NLD], w_r0[NLW], a_r1[NLD], w_r1[NLW];
    const int nk = p.K / GBK;
    auto load_chunk = [&](int kc, u32x4 (&a_r)[NLD], u32x4 (&w_r)[NLW]) {
        const bool valid = kc < nk;
        const __amdgpu_buffer_rsrc_t ra = valid ? rsA : rsA0, rw = valid ? rsW : rsW0;
#pragma unroll
        for (int i = 0; i < NLD; ++i) a_r[i] = __builtin_amdgcn_raw_buffer_load_b128(ra, a_off[i], (unsigned)(kc * GBK * 2), 0);
#pragma unroll
        for (int i = 0; i < NLW; ++i) w_r[i] = __builtin_amdgcn_raw_buffer_load_b128(rw, w_off[i], (unsigned)(kc * GBK * 2), 0);
    };
    auto store_chunk = [&](int stage, const u32x4 (&a_r)[NLD], const u32x4 (&w_r)[NLW]) {
        u16* Ab = lds + stage * G_STAGE;
        u16* Wb = Ab + GBM * GBK;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int r = lr + RPP * i;
            *(u32x4*)(Ab + r * GBK + ((lc ^ gswz<GBK>(r)) << 3)) = a_r[i];
        }
#pragma unroll
        for (int i = 0; i < NLW; ++i) {
            const int r = lr + RPP * i;
            *(u32x4*)(Wb + r * GBK + ((lc ^ gswz<GBK>(r)) << 3)) = w_r[i];
        }
    };
    f32x16 acc[2][TN];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto compute = [&](int st) {
        const u16* Ab = lds + st * G_STAGE;
        const u16* Wb = Ab + GBM * GBK;
#pragma unroll
        for (int ks = 0; ks < GBK / 16; ++ks) {
            u32x4 af[2], wf[TN];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = wm0 + 32 * i + fr;
                af[i] = *(const u32x4*)(Ab + r * GBK + (((2 * ks + fh) ^ gswz<GBK>(r)) << 3));
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int r = wn0 + 32 * j + fr;
                wf[j] = *(const u32x4*)(Wb + r * GBK + (((2 * ks + fh) ^ gswz<GBK>(r)) << 3));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, af[i]), __builtin_bit_cast(f16x8, wf[j]),
                                                                       acc[i][j], 0, 0, 0);
        }
    };

    load_chunk(0, a_r0, w_r0);
    load_chunk(1, a_r1, w_r1);
    store_chunk(0, a_r0, w_r0);
    __syncthreads();
    // chunk kc lives in stage kc & 1; set r1 holds chunk kc+1 on even steps, r0 on odd ones
    // (the scheduling fences keep the loads in front of the MFMAs: left alone the compiler sinks them behind the LDS
    //  stores of the other set, whose s_waitcnt vmcnt(0) then drains the whole queue in the middle of the step)
    for (int kc = 0; kc < nk; kc += 2) {
        load_chunk(kc + 2, a_r0, w_r0);
        GF16_FENCE();
        compute(0);
        GF16_FENCE2();
        store_chunk(1, a_r1, w_r1);
        __syncthreads();
        load_chunk(kc + 3, a_r1, w_r1);
        GF16_FENCE();
        compute(1);                                       // (kc + 1 == nk: a chunk of zeros)
        GF16_FENCE2();
        store_chunk(0, a_r0, w_r0);
        __syncthreads();
    }

    // epilogue: a wave transposes its 64 x WTN block 32 rows at a time through LDS so that every lane owns 8 consecutive
    // columns: bias / residual / store are 16 B per lane, whole row segments per WTN / 8 lanes
    constexpr int LPR = WTN / 8, RPI = 64 / LPR, NIT = 32 / RPI;   // lanes per row, rows per wave instruction, instructions per half
    float* Ls = (float*)lds + wave * (32 * G_LROW);
    const int ec = (lane % LPR) * 8, er = lane / LPR;
    const int n = n0 + wn0 + ec;
    f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0, s0 = {1.f, 1.f, 1.f, 1.f}, s1 = s0;
    if (p.bias && n < p.N) { b0 = *(const f32x4*)(p.bias + n); b1 = *(const f32x4*)(p.bias + n + 4); }
    if (p.oscale && n < p.N) { s0 = *(const f32x4*)(p.oscale + n); s1 = *(const f32x4*)(p.oscale + n + 4); }
    const bool rf = p.res_first != 0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) Ls[((r & 3) + 8 * (r >> 2) + 4 * fh) * G_LROW + j * 32 + fr] = acc[i][j][r];
        __syncthreads();
#pragma unroll
        for (int t0 = 0; t0 < NIT; t0 += 4) {              // residual loads of four row groups in flight together
            u32x4 rv[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int m = m0 + wm0 + 32 * i + er + RPI * (t0 + t);
                rv[t] = (u32x4){0u, 0u, 0u, 0u};
                if (p.res && m < p.M && n < p.N) rv[t] = *(const u32x4*)(p.res + (long long)m * p.ldr + n);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int row = er + RPI * (t0 + t), m = m0 + wm0 + 32 * i + row;
                const f32x4 v0 = *(const f32x4*)(Ls + row * G_LROW + ec) * s0 + b0, v1 = *(const f32x4*)(Ls + row * G_LROW + ec + 4) * s1 + b1;
                const f16x8 rh = __builtin_bit_cast(f16x8, rv[t]);
                float o[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float r0 = (float)rh[q], r1 = (float)rh[4 + q];
                    o[q] = act_f(v0[q] + (rf ? r0 : 0.f), p.act) + (rf ? 0.f : r0);
                    o[4 + q] = act_f(v1[q] + (rf ? r1 : 0.f), p.act) + (rf ? 0.f : r1);
                }
                if (m < p.M && n < p.N)
                    *(u32x4*)(p.c + (long long)m * p.ldc + n) = (u32x4){pack2(o[0], o[1]), pack2(o[2], o[3]), pack2(o[4], o[5]), pack2(o[6], o[7])};
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// GEMM, deep-pipelined: 256 x 256 x 64 tiles, 8 waves (2 x 4, 128 x 64 of output each), one workgroup per CU.
// The structure of cdna_hip_programming.md section 5 ("256^2 8-phase"), re-derived for this kernel's 32x32x16 fp16 MFMA:
//   * operands reach LDS by LDS-DMA (buffer_load ... lds, 1 KB per wave instruction): no staging registers, no ds_write;
//     the XOR swizzle that keeps ds_read_b128 conflict-free is applied to the SOURCE chunk index;
//   * a K tile is four HALF-TILES of 128 rows x 64 k (16 KB): Ah0 / Ah1 = the first / second 64 rows of both wave rows'
//     A blocks, Bh0 / Bh1 = the first / second 32 columns of all four wave columns' W blocks.  A wave walks its 128 x 64
//     block as four 64 x 32 quadrants, one per PHASE: (A0,B0) (A0,B1) (A1,B1) (A1,B0), so a phase needs at most one new
//     half of each operand (12 / 4 / 8 / 0 ds_read_b128) and every half-tile has ONE phase in which it is first needed;
//   * each phase also stages one half-tile (2 DMA instructions per thread) for five phases later; s_waitcnt vmcnt(6)
//     (three half-tiles stay in flight -- the queue is never drained in the loop) retires the one staged three phases ago,
//     which is read one phase after that wait; a slot is restaged >= 3 phases after its last read.  Two LDS buffers
//     x four half-tiles = 128 KB;
//   * the wave rows run ONE BARRIER APART: while waves 0-3 issue their 8 MFMAs (256 cycles, s_setprio 1) waves 4-7 do
//     their LDS reads, DMA issue and waits, and vice versa -- each SIMD's two waves alternate on the matrix pipe.
//   * with one workgroup per CU nothing else hides a tile's first round trip or its epilogue, so the workgroups are
//     PERSISTENT (one per CU, each walking a contiguous tile range of its XCD) and the next tile's first five half-tiles
//     are issued BEFORE the current tile's epilogue, which needs no LDS: the W half-tiles are the even / odd columns,
//     so a lane's two accumulator blocks are adjacent columns and every register is one packed dword of a full row segment.
// Needs N % 256 == 0 and K % 128 == 0 (two K tiles per loop trip).  Epilogue as in gemm_f16_kernel.
// ---------------------------------------------------------------------------------------------------------------
constexpr int PH_HALF = 128 * 64 * 2;                              // bytes of a half-tile
#ifndef GF16_DEPHASE
#define GF16_DEPHASE 0
#endif
#ifndef GF16_ABL            // timing ablations (results wrong): 1 no C stores, 2 no epilogue arithmetic either, 8 no main loop
#define GF16_ABL 0
#endif

__device__ __forceinline__ void glds16(__amdgpu_buffer_rsrc_t r, unsigned char* lds_dst, unsigned voff, unsigned soff) {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_dst, 16, voff, soff, 0, 0);
#else
    (void)r; (void)lds_dst; (void)voff; (void)soff;
#endif
}

template <int ACT, int RES, int TWO = 0>                             // epilogue specialised: a run-time `act` costs 22 VALU per output pair
__global__ __launch_bounds__(512, 1) void gemm_f16_8ph_kernel(const GemmHP p) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[8 * PH_HALF];      // [buffer 2][Ah0, Bh0, Bh1, Ah1] = 128 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6), wr = wid >> 2, wc = wid & 3;
    const int fr = lane & 31, fh = lane >> 5;
    // this workgroup's tiles: the XCD's contiguous range (xcd_remap's split), walked with the stride of the XCD's workgroups
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, slot_in_xcd = blockIdx.x >> 3, wg_per_xcd = (nwg - xcd + 7) >> 3;
    const int tq = p.n_full >> 3, trm = p.n_full & 7;
    const int t_lo = xcd < trm ? xcd * (tq + 1) : trm * (tq + 1) + (xcd - trm) * tq, t_hi = t_lo + tq + (xcd < trm ? 1 : 0);
    const __amdgpu_buffer_rsrc_t rsA0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.a, 0, 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, 0, 0x00020000);
    __amdgpu_buffer_rsrc_t rsA = rsA0, rsW = rsW0, rsA2 = rsA0, rsW2 = rsW0;
    int m0 = 0, n0 = 0;
    float rr0 = 1.f, rr1 = 1.f;                                      // TWO: ratio of this lane's two output columns
    // stager: thread -> LDS chunk (tid + 512 i) of a half-tile = local row (tid >> 3) + 64 i, slot tid & 7; it fetches
    // source chunk slot ^ swz(row).  Half-tile kind k = 0..3 (Ah0, Bh0, Bh1, Ah1) -> operand rows:
    //   A half h: tile row (lr >> 6) * 128 + h * 64 + (lr & 63);   W half h: tile column (lr >> 5) * 64 + 2 * (lr & 31) + h
    //   (W half 0 = the EVEN columns of every wave column's 64, half 1 = the odd ones: accumulator blocks j = 0 / 1 of a lane
    //    are then two ADJACENT output columns -- one packed dword in the epilogue)
    unsigned voff[4][2];
    auto set_tile = [&](int tile) {
        m0 = (tile / p.tiles_n) * 256; n0 = (tile % p.tiles_n) * 256;
        rsA = desc(p.a, p.a_total, (long long)m0 * p.lda * 2);
        rsW = desc(p.w, p.w_total, (long long)n0 * p.ldw * 2);
        if constexpr (TWO) {
            rsA2 = desc(p.a2, p.a2_total, (long long)m0 * p.lda2 * 2);
            rsW2 = desc(p.w2, p.w2_total, (long long)n0 * p.ldw2 * 2);
            const int n = n0 + (wid & 3) * 64 + 2 * (lane & 31);
            rr0 = p.ratio[n]; rr1 = p.ratio[n + 1];
        }
        if constexpr (!TWO)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int lr = (tid >> 3) + 64 * i, c = (tid & 7) ^ ((lr >> 1) & 7);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int ra = (lr >> 6) * 128 + h * 64 + (lr & 63), rw = (lr >> 5) * 64 + 2 * (lr & 31) + h;
                voff[h ? 3 : 0][i] = m0 + ra < p.M ? (unsigned)ra * (unsigned)(p.lda * 2) + c * 16u : OOR;
                voff[h ? 2 : 1][i] = (unsigned)rw * (unsigned)(p.ldw * 2) + c * 16u;
            }
        }
    };
    const int nT2 = TWO ? p.K2 / 64 : 0, nT = nT2 + p.K / 64;       // K tiles of the second pair (they run first) / in all
    // stage number q: kind q & 3 of K tile q >> 2 into buffer (q >> 2) & 1; tiles past the end go through the zero-extent
    // descriptors so that every phase issues exactly two DMA instructions per wave (the vmcnt arithmetic relies on it)
    // (TWO: the second pair's lane offsets are recomputed here -- eight more registers spilled inside the loop --, and tiles past the end
    //  fetch whatever follows into buffers no phase reads again: two-way descriptor selects only)
    auto stage = [&](int kind, int buf, int t) {
        const bool isA = kind == 0 || kind == 3, valid = t < nT, second = TWO && t < nT2;
        unsigned char* slot = lds + (buf * 4 + kind) * PH_HALF + wid * 1024;
        if constexpr (TWO) {
            const __amdgpu_buffer_rsrc_t rs = isA ? (second ? rsA2 : rsA) : (second ? rsW2 : rsW);
            const int h = kind >> 1;                                  // kinds 0, 1 -> half 0; 2, 3 -> half 1
            int tid_o = tid;
            asm volatile("" : "+v"(tid_o));                           // opaque: the offsets below are loop-invariant and would be hoisted (and spilled)
            const unsigned ld2 = (unsigned)((isA ? (second ? p.lda2 : p.lda) : (second ? p.ldw2 : p.ldw)) * 2);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int lr = (tid_o >> 3) + 64 * i, c = (tid_o & 7) ^ ((lr >> 1) & 7);
                const int ra = (lr >> 6) * 128 + h * 64 + (lr & 63), rw = (lr >> 5) * 64 + 2 * (lr & 31) + h;
                const unsigned vo = isA ? (m0 + ra < p.M ? (unsigned)ra * ld2 + c * 16u : OOR) : (unsigned)rw * ld2 + c * 16u;
                glds16(rs, slot + i * 8192, vo, (unsigned)(second ? t : t - nT2) * 128u);
            }
        } else {
            const __amdgpu_buffer_rsrc_t rs = isA ? (valid ? rsA : rsA0) : (valid ? rsW : rsW0);
#pragma unroll
            for (int i = 0; i < 2; ++i) glds16(rs, slot + i * 8192, voff[kind][i], (unsigned)t * 128u);
        }
    };
    // fragment addresses inside a half-tile (bytes): A rows wr * 64 + blk * 32 + fr, W rows wc * 32 + fr, chunk (2 ks + fh) ^ swz
    int aoff[2][4], boff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int r = wr * 64 + b * 32 + fr;
            aoff[b][ks] = r * 128 + (((2 * ks + fh) ^ ((r >> 1) & 7)) << 4);
        }
        const int r = wc * 32 + fr;
        boff[ks] = r * 128 + (((2 * ks + fh) ^ ((r >> 1) & 7)) << 4);
    }
    f32x16 acc[4][2];
    u32x4 fa[2][4], fb0[4], fb1[4];

    // one phase: j = phase within the loop trip (static), t2 = first K tile of the trip
    // `first`: the tile's first trip.  Its prologue has staged both K tiles of the trip (stages 0 .. 7), so phases 0 - 2 stage
    // nothing and phases 0 - 4 wait for nothing: the first counted wait (phase 5) is where the previous tile's epilogue stores
    // -- older in the queue than every DMA of this tile -- have to be acknowledged, 2000+ cycles after they were issued,
    // instead of at the top of the tile.
    auto phase = [&](int j, int t2, bool first) {
        const int ph = j & 3, buf = (j >> 2) & 1;
        const unsigned char* base = lds + buf * 4 * PH_HALF;
        if (ph == 0) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fb0[ks] = *(const u32x4*)(base + 1 * PH_HALF + boff[ks]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) fa[b][ks] = *(const u32x4*)(base + 0 * PH_HALF + aoff[b][ks]);
        } else if (ph == 1) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fb1[ks] = *(const u32x4*)(base + 2 * PH_HALF + boff[ks]);
        } else if (ph == 2) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) fa[b][ks] = *(const u32x4*)(base + 3 * PH_HALF + aoff[b][ks]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(first && j < 3)) {
            const int q = j + 5;                                      // stage number relative to the trip's first tile
            stage(q & 3, (q >> 2) & 1, t2 + (q >> 2));
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!(first && j < 5)) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        const int ai = (ph >= 2) ? 2 : 0, bj = (ph == 1 || ph == 2) ? 1 : 0;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int b = 0; b < 2; ++b)
                acc[ai + b][bj] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[b][ks]),
                                                                         __builtin_bit_cast(f16x8, bj ? fb1[ks] : fb0[ks]), acc[ai + b][bj], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    };

    // Work items of this workgroup: its whole tiles (all K tiles), then -- if there are cut tiles and this workgroup's index is below
    // n_cut * n_slices -- one slice: a cut tile's loop trips [s T / S, (s + 1) T / S)
    const int n_whole = t_lo + slot_in_xcd < t_hi ? (t_hi - t_lo - slot_in_xcd + wg_per_xcd - 1) / wg_per_xcd : 0;
    const bool has_slice = !TWO && (int)blockIdx.x < p.n_cut * p.n_slices;
    const int n_items = n_whole + (has_slice ? 1 : 0);
    int tb = 0, te = nT;                                              // the current item's K tiles [tb, te), both even
    auto set_item = [&](int k) {
        if (k < n_whole) { tb = 0; te = nT; set_tile(t_lo + slot_in_xcd + k * wg_per_xcd); }
        else {
            const int l = blockIdx.x / p.n_slices, sl = blockIdx.x - l * p.n_slices, T = nT >> 1;
            tb = 2 * (sl * T / p.n_slices); te = 2 * ((sl + 1) * T / p.n_slices);
            set_tile(p.n_full + l);
        }
    };
    // prologue of an item: stages 0 .. 7 = its first two K tiles, both buffers
    auto prologue = [&]() {
#pragma unroll
        for (int q = 0; q < 8; ++q) stage(q & 3, (q >> 2) & 1, tb + (q >> 2));
    };
    const int wm0 = wr * 128, wn0 = wc * 64;

    bool first_tile = true;
    // De-phase the workgroups.  Every tile takes the same time, so persistent workgroups that start together reach their
    // epilogues together: 256 x (128 KB of C + residual + the next A tile) hit HBM in one burst while it idles during
    // the main loops -- measured as a FIXED 14-16 us per tile whatever K (K sweep, tools/bench_gemm_f16.py).  The groups
    // of workgroups that share an A tile (tiles_n consecutive slots of an XCD; kept in step for the L2) start spread
    // over one tile time instead.
    if (GF16_DEPHASE && (t_hi - t_lo) >= 3 * wg_per_xcd) {
        const int per = (wg_per_xcd + p.tiles_n - 1) / p.tiles_n, grp = (slot_in_xcd / p.tiles_n) + per * xcd, ngrp = per * 8;
        const int units = (int)((long long)grp * nT * 1400 / ngrp) >> 10;       // ~1400 cycles per K tile, s_sleep 16 = 1024 cycles
        for (int u = 0; u < units; ++u) __builtin_amdgcn_s_sleep(16);
    }
    if (n_items > 0) { set_item(0); prologue(); }
    for (int k = 0; k < n_items; ++k) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    // The prologue's 16 DMA instructions must have landed; the 64 epilogue stores issued AFTER them need not have been
    // acknowledged yet (4.6 us of a tile's fixed cost when they were waited for here): vmcnt(63) leaves at most 63 of the
    // 80 outstanding, i.e. retires the 16 DMAs (and one store).  The first tile has no stores behind its prologue.
    if (first_tile) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
    first_tile = false;
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();                        // the second wave row runs one barrier behind
#pragma unroll
    for (int j = 0; j < 8; ++j) phase(j, tb, true);
    for (int t2 = tb + 2; t2 < ((GF16_ABL & 8) ? tb + 2 : te); t2 += 2) {
        if (TWO && t2 == nT2) {                                       // the second pair's sums -> the main pair's scale (K2 / 64 and t2 are even)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[i][0][r] *= rr0; acc[i][1][r] *= rr1; }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) phase(j, t2, false);
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the look-ahead DMA past this item's last K tile
    __builtin_amdgcn_s_barrier();                                     // every wave is done with the ring
    const int em0 = m0, en0 = n0;
    const bool whole = k < n_whole;
    if (k + 1 < n_items) { set_item(k + 1); prologue(); }             // in flight during the epilogue below
    if (!whole) {                                                     // a slice: the raw accumulators, 16 B per thread and store
        const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc((void*)(p.ws + (size_t)blockIdx.x * (128 * 512)), 0, 128 * 512 * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u32x4 v = {__float_as_uint(acc[i][j][4 * q]), __float_as_uint(acc[i][j][4 * q + 1]), __float_as_uint(acc[i][j][4 * q + 2]),
                                     __float_as_uint(acc[i][j][4 * q + 3])};
                    __builtin_amdgcn_raw_buffer_store_b128(v, rsP, (unsigned)tid * 16u, (unsigned)(((i * 2 + j) * 4 + q) * 8192), 0);
                }
        continue;
    }

    // epilogue straight from the accumulators, no LDS: lane (fr, fh) holds output columns 2 fr and 2 fr + 1 (blocks j = 0 / 1,
    // see the stager's column map) of rows 32 i + (r & 3) + 8 (r >> 2) + 4 fh: one packed dword per register, 32 lanes = one
    // whole 128-B row segment, two rows per wave instruction -- the full-rate access shape, residual loads likewise.
    // (Through the LDS transpose of gemm_f16_kernel the 68 KB of fp32 staging writes cost as much as the MFMAs of two K
    //  tiles; with swapped MFMA operands -- a lane owning 4 columns of ITS row, 8-B accesses to 32 different lines per
    //  instruction -- the vector-memory path made it slower still: 848 -> 663 TF on the out-projection shape.)
    {
        const int n = en0 + wn0 + 2 * fr;
        float bn0 = 0.f, bn1 = 0.f, sn0 = 1.f, sn1 = 1.f;
        if (p.bias) { bn0 = p.bias[n]; bn1 = p.bias[n + 1]; }
        if (p.oscale) { sn0 = p.oscale[n]; sn1 = p.oscale[n + 1]; }
        const bool rf = p.res_first != 0;
        // descriptors rebased to the wave's first row: the extent ends with the last valid row, so rows past M are dropped
        // by the hardware (no branches); lane offset = its column pair + its half's 4-row step, the row of a register is a
        // wave-uniform scalar offset
        const int mw = em0 + wm0;
        const long long rows_left = (long long)p.M - mw;
        const __amdgpu_buffer_rsrc_t rsC = desc(p.c, rows_left > 0 ? ((rows_left - 1) * p.ldc + p.N) * 2 + (long long)mw * p.ldc * 2 : 0, (long long)mw * p.ldc * 2);
        const __amdgpu_buffer_rsrc_t rsR = p.res ? desc(p.res, rows_left > 0 ? ((rows_left - 1) * p.ldr + p.N) * 2 + (long long)mw * p.ldr * 2 : 0, (long long)mw * p.ldr * 2)
                                                 : __builtin_amdgcn_make_buffer_rsrc((void*)p.c, 0, 0, 0x00020000);
        const unsigned vc = (unsigned)((4 * fh * p.ldc + n) * 2), vr = (unsigned)((4 * fh * p.ldr + n) * 2);
        // Rows >= M of a ragged last tile are masked BY LANE (an out-of-range voffset) instead of being left to the extent:
        // their row offset travels in soffset, which can exceed num_records there (range check: offset >= num_records -
        // soffset).  A full tile pays nothing.
        const int row_lim = (int)(rows_left < 1024 ? rows_left : 1024) - 4 * fh;     // row u of this lane is valid iff u < row_lim
        auto epilogue = [&](auto full_tag) {
            constexpr bool FULL = decltype(full_tag)::value;
            // all 64 residual loads in flight together (the fragment registers are dead here): one round trip, not four
            unsigned rvv[RES ? 4 : 1][16];
            if constexpr (RES) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int ru = 32 * i + (r & 3) + 8 * (r >> 2);
                        rvv[i][r] = __builtin_amdgcn_raw_buffer_load_b32(rsR, (FULL || ru < row_lim) ? vr : OOR, (unsigned)(ru * p.ldr * 2), 0);
                    }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const f16x2 rh = __builtin_bit_cast(f16x2, RES ? rvv[i][r] : 0u);
                    const float r0 = RES ? (float)rh[0] : 0.f, r1 = RES ? (float)rh[1] : 0.f;
                    float o0 = act_f(fmaf(acc[i][0][r], sn0, bn0) + (rf ? r0 : 0.f), ACT), o1 = act_f(fmaf(acc[i][1][r], sn1, bn1) + (rf ? r1 : 0.f), ACT);
                    if (RES) { o0 += rf ? 0.f : r0; o1 += rf ? 0.f : r1; }
                    if (GF16_ABL & 2) { asm volatile("" ::"v"(acc[i][0][r]), "v"(acc[i][1][r]), "v"(rh)); continue; }
                    if (GF16_ABL & 1) { asm volatile("" ::"v"(pack2(o0, o1))); continue; }
                    const int ru = 32 * i + (r & 3) + 8 * (r >> 2);
                    __builtin_amdgcn_raw_buffer_store_b32(pack2(o0, o1), rsC, (FULL || ru < row_lim) ? vc : OOR, (unsigned)(ru * p.ldc * 2), 0);
                }
            }
        };
        if (rows_left >= 128) epilogue(std::true_type{}); else epilogue(std::false_type{});
    }
    }
}

// The cut tiles of gemm_f16_8ph_kernel: sum the slices' accumulators in slice order (same thread <-> element mapping), then that kernel's
// epilogue for ONE 32-row block i of every wave's 128 x 64 per workgroup -- grid (n_cut, 4).
template <int ACT, int RES>
__global__ __launch_bounds__(512) void gemm_f16_8ph_fixup_kernel(const GemmHP p) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, wr = wid >> 2, wc = wid & 3, fr = lane & 31, fh = lane >> 5;
    const int tile = p.n_full + blockIdx.x, m0 = (tile / p.tiles_n) * 256, n0 = (tile % p.tiles_n) * 256, bi = blockIdx.y;
    f32x16 a0, a1;
    const f32x4* src = (const f32x4*)(p.ws + (size_t)blockIdx.x * p.n_slices * (128 * 512)) + (size_t)bi * 8 * 512 + tid;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 v = src[q * 512], u = src[(4 + q) * 512];
#pragma unroll
        for (int e = 0; e < 4; ++e) { a0[4 * q + e] = v[e]; a1[4 * q + e] = u[e]; }
    }
    for (int sl = 1; sl < p.n_slices; ++sl) {
        src += 32 * 512;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = src[q * 512], u = src[(4 + q) * 512];
#pragma unroll
            for (int e = 0; e < 4; ++e) { a0[4 * q + e] += v[e]; a1[4 * q + e] += u[e]; }
        }
    }
    const int n = n0 + wc * 64 + 2 * fr, mw = m0 + wr * 128;
    float bn0 = 0.f, bn1 = 0.f, sn0 = 1.f, sn1 = 1.f;
    if (p.bias) { bn0 = p.bias[n]; bn1 = p.bias[n + 1]; }
    if (p.oscale) { sn0 = p.oscale[n]; sn1 = p.oscale[n + 1]; }
    const bool rf = p.res_first != 0;
    const long long rows_left = (long long)p.M - mw;
    const __amdgpu_buffer_rsrc_t rsC = desc(p.c, rows_left > 0 ? ((rows_left - 1) * p.ldc + p.N) * 2 + (long long)mw * p.ldc * 2 : 0, (long long)mw * p.ldc * 2);
    const __amdgpu_buffer_rsrc_t rsR = p.res ? desc(p.res, rows_left > 0 ? ((rows_left - 1) * p.ldr + p.N) * 2 + (long long)mw * p.ldr * 2 : 0, (long long)mw * p.ldr * 2)
                                             : __builtin_amdgcn_make_buffer_rsrc((void*)p.c, 0, 0, 0x00020000);
    const unsigned vc = (unsigned)((4 * fh * p.ldc + n) * 2), vr = (unsigned)((4 * fh * p.ldr + n) * 2);
    const int row_lim = (int)(rows_left < 1024 ? rows_left : 1024) - 4 * fh;
    unsigned rvv[16];
    if constexpr (RES) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ru = 32 * bi + (r & 3) + 8 * (r >> 2);
            rvv[r] = __builtin_amdgcn_raw_buffer_load_b32(rsR, ru < row_lim ? vr : OOR, (unsigned)(ru * p.ldr * 2), 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const f16x2 rh = __builtin_bit_cast(f16x2, RES ? rvv[r] : 0u);
        const float r0 = RES ? (float)rh[0] : 0.f, r1 = RES ? (float)rh[1] : 0.f;
        float o0 = act_f(fmaf(a0[r], sn0, bn0) + (rf ? r0 : 0.f), ACT), o1 = act_f(fmaf(a1[r], sn1, bn1) + (rf ? r1 : 0.f), ACT);
        if (RES) { o0 += rf ? 0.f : r0; o1 += rf ? 0.f : r1; }
        const int ru = 32 * bi + (r & 3) + 8 * (r >> 2);
        __builtin_amdgcn_raw_buffer_store_b32(pack2(o0, o1), rsC, ru < row_lim ? vc : OOR, (unsigned)(ru * p.ldc * 2), 0);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Attention core.  One workgroup = 4 waves = 128 queries of one (image, head); each wave owns 32 queries.  K / V tiles of
// 64 keys go through LDS: K row-major [key][64], V TRANSPOSED [d][key position] with the keys of every 16-group
// permuted so that the 8 keys a lane needs for one MFMA step are one 16-B read.  S^T = K Q^T puts a query in a lane
// column: row maximum / sum are register reductions plus one exchange with lane ^ 32, the O^T rescale is lane-local,
// and the probabilities in their accumulator layout ARE the B operand of O^T += V^T P^T (keys permuted to match).
// ---------------------------------------------------------------------------------------------------------------
constexpr int A_OROW = 72;                                       // O staging pitch in halves (144 B rows: conflict-free b64 writes)

// QT = 32-query tiles per wave: 1 (a workgroup covers 128 queries; the default) or 2 (256: every K / V fragment read from LDS
// and every staged K / V tile serves twice as many MFMAs -- kept behind DBMM_MHA_F16_QT2, it measured slower)
// the 8 halves of a V^T fragment = two transposed 4 x 16 blocks 8 rows (keys) apart
__device__ __forceinline__ u32x4 vt_frag(const unsigned char* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 8 * 128));
    const u32x2 ua = __builtin_bit_cast(u32x2, a), ub = __builtin_bit_cast(u32x2, b);
    return (u32x4){ua[0], ua[1], ub[0], ub[1]};
#else
    (void)p; return (u32x4){0u, 0u, 0u, 0u};
#endif
}

// NW = waves per workgroup: 4, or 2 for sequences of at most 64 tokens (ViT-B/32's 50: two of four waves multiplied clamped queries)
template <int QT, int NW = 4>
__global__ __launch_bounds__(64 * NW, 2) void mha_f16_kernel(const u16* __restrict__ qkv, u16* __restrict__ out, int L, int E,
                                                             int heads, int causal, float scale_log2e) {
    __shared__ __attribute__((aligned(16))) u16 Ks[64 * 64];
    __shared__ __attribute__((aligned(16))) u16 Vt[64 * 64];
    __shared__ __attribute__((aligned(16))) u16 Os[NW * 32 * A_OROW];
    constexpr int QB = 32 * NW * QT;                             // queries per workgroup
    constexpr int KPI = 8 * NW, NI = 64 / KPI;                   // key rows per staging pass, passes per 64-key tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int qb = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
    const long long row0 = (long long)b * L;
    const long long ld = 3LL * E;
    int q_idx[QT];                                               // this lane's queries
    // Q fragments (B operand of S^T): lane (query, k half fh) holds d = 16 s + 8 fh .. + 7
    u32x4 qf[QT][4];
    f32x16 o_acc[QT][2];
    float m_run[QT], l_run[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        q_idx[qt] = qb * QB + (wave * QT + qt) * 32 + fr;
        const int q_ld = q_idx[qt] < L ? q_idx[qt] : L - 1;
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[qt][s] = *(const u32x4*)(qkv + (row0 + q_ld) * ld + head * 64 + 16 * s + 8 * fh);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) o_acc[qt][j][r] = 0.f;
        m_run[qt] = -INFINITY; l_run[qt] = 0.f;
    }

    // staging: thread loads 16 B (8 d) of key rows (tid >> 3) + KPI i for K and V
    const int lc = tid & 7, lk = tid >> 3;
    u32x4 k_r[NI], v_r[NI];
    const int q_hi = qb * QB + QB - 1 < L - 1 ? qb * QB + QB - 1 : L - 1;
    const int n_keys = causal ? q_hi + 1 : L;                   // causal: keys past the block's last query never count
    const int T = (n_keys + 63) / 64;
    auto load_tile = [&](int t) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int key = t * 64 + lk + KPI * i;
            key = key < L ? key : L - 1;
            const u16* base = qkv + (row0 + key) * ld + head * 64 + lc * 8;
            k_r[i] = *(const u32x4*)(base + E);
            v_r[i] = *(const u32x4*)(base + 2 * E);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int key = lk + KPI * i;
            *(u32x4*)(Ks + key * 64 + ((lc ^ swz64(key)) << 3)) = k_r[i];
            // V stays ROW-major ([key][64 d], one 16-B store like K): the V^T fragments are formed by the hardware
            // transpose read below.  Chunk XOR 4 on rows 2, 3 (mod 4) keeps the four rows of a transposed block on
            // distinct bank groups.  (Staged transposed by hand this was 16 ds_write_b16 per thread and tile.)
            *(u32x4*)(Vt + key * 64 + ((lc ^ (((key >> 1) & 1) << 2)) << 3)) = v_r[i];
        }
    };
    // ds_read_b64_tr_b16 (cdna_hip_programming.md T10): per 16 lanes a block of 4 rows x 16 columns, lane 4q + p supplies the
    // address of row q, columns 4p .. 4p + 3, lane i receives column i of the four rows.  For the O^T += V^T P^T operand the
    // rows are keys and the columns d: a lane's 8 k values of step (kt, u) are keys kt*32 + 16u + 4fh + {0..3} and + 8, i.e.
    // two blocks; its column is d = 32 j + 16 g + (lane & 15).  Byte address of this lane's piece for j = 0 / 1, without the
    // (kt, u, block) row offset, which is a multiple of 4 rows and goes into the immediate:
    int vtr[2];
    {
        const int g = (lane >> 4) & 1, pq = lane & 15, q = pq >> 2, pp = pq & 3;
#pragma unroll
        for (int j = 0; j < 2; ++j) vtr[j] = (4 * fh + q) * 128 + 16 * (4 * (j ^ (q >> 1)) + 2 * g + (pp >> 1)) + 8 * (pp & 1);
    }
    load_tile(0);
    store_tile();
    __syncthreads();
    for (int t = 0; t < T; ++t) {
        if (t + 1 < T) load_tile(t + 1);
        // ---- S^T[key][query] for the tile's two 32-key halves: one K fragment read serves every query tile ---------------
        f32x16 s_acc[QT][2];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) s_acc[qt][kt][r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int row = kt * 32 + fr;
                const u32x4 kf = *(const u32x4*)(Ks + row * 64 + (((2 * s + fh) ^ swz64(row)) << 3));
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
                    s_acc[qt][kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, kf), __builtin_bit_cast(f16x8, qf[qt][s]),
                                                                           s_acc[qt][kt], 0, 0, 0);
            }
        // ---- online softmax: a lane holds keys (r&3) + 8 (r>>2) + 4 fh of each half for ITS query of every tile ----------
        // The softmax is what bounds this kernel (32 elements per lane and tile against 16 MFMAs per wave), so its
        // instruction count is kept down: masks only on tiles that can hold a masked key (the last one, and under the
        // causal mask the tiles that reach past the wave's first query) -- a wave-uniform branch; the 1/8 * log2(e) scale
        // folded into the exponent's FMA; the hardware exp2 (arguments <= 0, flushed denormals are zeros anyway); the
        // accumulator rescale skipped while no lane's maximum moved.
        u32x4 pf[QT][2][2];
        const bool tile_masked = (t + 1) * 64 > L || (causal && (t + 1) * 64 - 1 > qb * QB + wave * QT * 32);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            float mx = -INFINITY;
            if (tile_masked) {
#pragma unroll
                for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = t * 64 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                        if (key >= L || (causal && key > q_idx[qt])) s_acc[qt][kt][r] = -INFINITY;
                    }
            }
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s_acc[qt][kt][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * scale_log2e;     // (scale > 0: the maximum commutes with it)
            const float m_new = fmaxf(m_run[qt], mx);
            const float m_use = m_new == -INFINITY ? 0.f : m_new;   // a fully masked row so far: exp2(-inf - 0) = 0
            const float alpha = fast_exp2(m_run[qt] - m_use);
            float psum = 0.f;
#pragma unroll
            for (int kt = 0; kt < 2; ++kt) {
                float pv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) { pv[r] = fast_exp2(fmaf(s_acc[qt][kt][r], scale_log2e, -m_use)); psum += pv[r]; }
#pragma unroll
                for (int u = 0; u < 2; ++u)
                    pf[qt][kt][u] = (u32x4){pack2(pv[8 * u], pv[8 * u + 1]), pack2(pv[8 * u + 2], pv[8 * u + 3]),
                                            pack2(pv[8 * u + 4], pv[8 * u + 5]), pack2(pv[8 * u + 6], pv[8 * u + 7])};
            }
            l_run[qt] = l_run[qt] * alpha + psum;
            m_run[qt] = m_new;
            if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0ull) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o_acc[qt][j][r] *= alpha;
            }
        }
        // ---- O^T[d][query] += V^T[d][keys] P^T[keys][query]: one V^T fragment read serves every query tile ----------------
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const u32x4 vf = vt_frag((const unsigned char*)Vt + vtr[j] + (kt * 32 + 16 * u) * 128);
#pragma unroll
                    for (int qt = 0; qt < QT; ++qt)
                        o_acc[qt][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, vf), __builtin_bit_cast(f16x8, pf[qt][kt][u]),
                                                                              o_acc[qt][j], 0, 0, 0);
                }
        __syncthreads();
        if (t + 1 < T) store_tile();
        __syncthreads();
    }
    // ---- normalise; O^T -> rows through LDS (wave-private staging, one query tile at a time); 16-B stores ----------------
    u16* Ow = Os + wave * (32 * A_OROW);
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        const float l_tot = l_run[qt] + __shfl_xor(l_run[qt], 32, 64);
        const float inv = l_tot > 0.f ? 1.f / l_tot : 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = j * 32 + 8 * g + 4 * fh;
                *(u32x2*)(Ow + fr * A_OROW + d) = (u32x2){pack2(o_acc[qt][j][4 * g] * inv, o_acc[qt][j][4 * g + 1] * inv),
                                                           pack2(o_acc[qt][j][4 * g + 2] * inv, o_acc[qt][j][4 * g + 3] * inv)};
            }
        // (the wave's own LDS writes are ordered before its reads, and these reads before the next tile's writes)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (lane >> 3) + 8 * i, q = qb * QB + (wave * QT + qt) * 32 + row;
            if (q < L) *(u32x4*)(out + (row0 + q) * (long long)E + head * 64 + (lane & 7) * 8) = *(const u32x4*)(Ow + row * A_OROW + (lane & 7) * 8);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm, fp16 in / out, fp32 statistics (two pass, biased variance); wave per row, 8 halves per lane and pass
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_f16_kernel(const u16* __restrict__ x, long long ldx, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, u16* __restrict__ y, long long ldy,
                                                            int rows, int E8, float eps) {
    const int lane = threadIdx.x & 63;
    const float invE = 1.f / (float)(E8 * 8);
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
        const u32x4* xr = (const u32x4*)(x + (long long)row * ldx);
        u32x4* yr = (u32x4*)(y + (long long)row * ldy);
        constexpr int NV = 4;                                    // E <= 2048 held in registers; wider rows re-read
        float v[NV][8];
        float s = 0.f;
        if (E8 <= 64 * NV) {
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int i = lane + 64 * j;
                const f16x8 h = i < E8 ? __builtin_bit_cast(f16x8, xr[i]) : (f16x8){0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                for (int q = 0; q < 8; ++q) { v[j][q] = (float)h[q]; s += v[j][q]; }
            }
            const float mean = wave_sum(s) * invE;
            float qq = 0.f;
#pragma unroll
            for (int j = 0; j < NV; ++j)
                if (lane + 64 * j < E8)
#pragma unroll
                    for (int q = 0; q < 8; ++q) { const float d = v[j][q] - mean; qq += d * d; }
            const float rstd = rsqrtf(wave_sum(qq) * invE + eps);
#pragma unroll
            for (int j = 0; j < NV; ++j) {
                const int i = lane + 64 * j;
                if (i < E8) {
                    const f32x4 g0 = ((const f32x4*)gamma)[2 * i], g1 = ((const f32x4*)gamma)[2 * i + 1];
                    const f32x4 c0 = ((const f32x4*)beta)[2 * i], c1 = ((const f32x4*)beta)[2 * i + 1];
                    float o[8];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        o[q] = (v[j][q] - mean) * rstd * g0[q] + c0[q];
                        o[4 + q] = (v[j][4 + q] - mean) * rstd * g1[q] + c1[q];
                    }
                    yr[i] = (u32x4){pack2(o[0], o[1]), pack2(o[2], o[3]), pack2(o[4], o[5]), pack2(o[6], o[7])};
                }
            }
        } else {
            for (int i = lane; i < E8; i += 64) { const f16x8 h = __builtin_bit_cast(f16x8, xr[i]); for (int q = 0; q < 8; ++q) s += (float)h[q]; }
            const float mean = wave_sum(s) * invE;
            float qq = 0.f;
            for (int i = lane; i < E8; i += 64) {
                const f16x8 h = __builtin_bit_cast(f16x8, xr[i]);
                for (int q = 0; q < 8; ++q) { const float d = (float)h[q] - mean; qq += d * d; }
            }
            const float rstd = rsqrtf(wave_sum(qq) * invE + eps);
            for (int i = lane; i < E8; i += 64) {
                const f16x8 h = __builtin_bit_cast(f16x8, xr[i]);
                float o[8];
                for (int q = 0; q < 8; ++q) o[q] = ((float)h[q] - mean) * rstd * gamma[8 * i + q] + beta[8 * i + q];
                yr[i] = (u32x4){pack2(o[0], o[1]), pack2(o[2], o[3]), pack2(o[4], o[5]), pack2(o[6], o[7])};
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// layout kernels
// ---------------------------------------------------------------------------------------------------------------
// patch im2col: NCHW image (fp32 or fp16) -> [B * g * g][Kp] fp16, K = 3 * P * P zero-padded to Kp (a multiple of 64).
// Walks the IMAGE (round 4; the first version walked the output element by element with a div / mod chain and read 4-byte pieces:
// 0.87 TB/s): a thread takes VEC consecutive pixels of one image row -- consecutive threads read consecutive vectors of that row,
// P % VEC == 0 keeps a vector inside one patch row -- and writes them as one VEC-half store at (patch, channel, kh, kw).
template <typename TI, int VEC>
__global__ __launch_bounds__(256) void im2col_patch_f16_kernel(const TI* __restrict__ x, u16* __restrict__ out, int R, int P, int g,
                                                               int Kp, long long n_vec, long long n_pad) {
    const int K = 3 * P * P, RV = R / VEC;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (long long)gridDim.x * blockDim.x) {
        const int xv = (int)(i % RV);
        const long long row = i / RV;                                     // (b * 3 + c) * R + y
        const int y = (int)(row % R), c = (int)((row / R) % 3);
        const long long b = row / (3LL * R);
        const int x0 = xv * VEC, gx = x0 / P, kw = x0 - gx * P, gy = y / P, kh = y - gy * P;
        typedef TI vin_t __attribute__((ext_vector_type(VEC)));           // both sides are VEC-element aligned: R, P, Kp are multiples of VEC
        typedef _Float16 vout_t __attribute__((ext_vector_type(VEC)));
        const vin_t vi = *(const vin_t*)(x + row * R + x0);
        vout_t vo;
#pragma unroll
        for (int j = 0; j < VEC; ++j) vo[j] = (_Float16)(float)vi[j];
        *(vout_t*)((_Float16*)out + ((b * g + gy) * g + gx) * (long long)Kp + (c * P + kh) * P + kw) = vo;
    }
    const int pad = Kp - K;                                               // zero columns K .. Kp - 1 (ViT-L/14: 588 -> 640)
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_pad; i += (long long)gridDim.x * blockDim.x)
        ((_Float16*)out)[(i / pad) * Kp + K + (i % pad)] = (_Float16)0.f;
}

// class token + patches + positional embedding -> tokens [B][L][W] fp16 (sum in fp32, one rounding)
__global__ __launch_bounds__(256) void vit_tokens_f16_kernel(const u16* __restrict__ patches, const float* __restrict__ cls,
                                                             const float* __restrict__ pos, u16* __restrict__ out, int L, int W,
                                                             long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % W);
        const long long row = i / W;
        const int l = (int)(row % L);
        const long long b = row / L;
        const float v = l == 0 ? cls[c] : (float)((const _Float16*)patches)[(b * (L - 1) + (l - 1)) * W + c];
        ((_Float16*)out)[i] = (_Float16)(v + pos[(long long)l * W + c]);
    }
}

__global__ __launch_bounds__(256) void embed_gather_f16_kernel(const int32_t* __restrict__ tokens, const float* __restrict__ table,
                                                               const float* __restrict__ pos, u16* __restrict__ out, int L, int W,
                                                               int vocab, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % W);
        const long long row = i / W;
        const int l = (int)(row % L);
        int t = tokens[row];
        t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);
        ((_Float16*)out)[i] = (_Float16)(table[(long long)t * W + c] + pos[(long long)l * W + c]);
    }
}

__global__ __launch_bounds__(64) void gather_eot_f16_kernel(const int32_t* __restrict__ tokens, const u16* __restrict__ x,
                                                            u16* __restrict__ out, int L, int W) {
    const int n = blockIdx.x, lane = threadIdx.x;
    int best = INT32_MIN, bi = 0;                               // first index of the maximum token id (torch.argmax tie rule)
    for (int l = lane; l < L; l += 64) {
        const int t = tokens[(long long)n * L + l];
        if (t > best) { best = t; bi = l; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int ob = __shfl_xor(best, o, 64), oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    for (int c = lane; c < W; c += 64) out[(long long)n * W + c] = x[((long long)n * L + bi) * W + c];
}

__global__ __launch_bounds__(256) void cast_f32_f16_kernel(const float* __restrict__ x, u16* __restrict__ y, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        ((_Float16*)y)[i] = (_Float16)x[i];
}

inline unsigned grid_for(long long total) {
    const long long blocks = (total + 255) / 256;
    return (unsigned)(blocks < 16384 ? (blocks > 0 ? blocks : 1) : 16384);
}

}  // namespace

namespace {
int gemm_f16_impl(const void* a, int64_t lda, const void* w, int64_t ldw, const float* out_scale, const float* bias, const void* residual,
                  int64_t ldr, int res_first, void* c, int64_t ldc, int64_t M, int64_t N, int64_t K, int act, void* stream, int part,
                  void* workspace = nullptr, size_t workspace_bytes = 0);
}

extern "C" int dbmm_gemm_f16(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, const void* residual,
                             int64_t ldr, void* c, int64_t ldc, int64_t M, int64_t N, int64_t K, int act, void* stream) {
    return gemm_f16_impl(a, lda, w, ldw, nullptr, bias, residual, ldr, 0, c, ldc, M, N, K, act, stream, 0);
}

// see include/dbmm.h: dbmm_gemm_f16 with a workspace (the tiles of a short last round of the eight-phase kernel are cut along K)
extern "C" int dbmm_gemm_f16_ws(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, const void* residual,
                                int64_t ldr, void* c, int64_t ldc, int64_t M, int64_t N, int64_t K, int act, void* workspace,
                                size_t workspace_bytes, void* stream) {
    return gemm_f16_impl(a, lda, w, ldw, nullptr, bias, residual, ldr, 0, c, ldc, M, N, K, act, stream, 0, workspace, workspace_bytes);
}

// 1x1 conv + eval-mode BatchNorm (+ residual) + activation on fp16 NHWC maps = the same GEMM with a per-channel scale and the
// residual added BEFORE the activation (Bottleneck.forward, clip/model.py:42-55); see include/dbmm.h
int dbmm_conv1x1_stream_f16(const void* x, const void* w, const float* scale, const float* bias, const void* residual, void* y, int64_t M,
                            int64_t Cin, int64_t Cout, int act, void* stream);       // conv_f16.hip
int dbmm_conv1x1_dual_stream_f16(const void* y2, const void* w3, const float* scale3, const void* xp, const void* wd, const float* ratio,
                                 const float* bias, void* out, int64_t M, int64_t K, int64_t K2, int64_t Cout, int act, void* stream);   // conv_f16.hip

extern "C" int dbmm_conv1x1_bn_act_f16(const void* x, const void* w, const float* scale, const float* bias, const void* residual, void* y,
                                       int64_t M, int64_t Cin, int64_t Cout, int act, void* stream) {
    // option conv1x1_stream: 0 = always the GEMM kernels, 1 (default) = the streaming kernel where the GEMM is HBM-bound
    // (short K or narrow N), 2 = wherever it applies
    const int mode = dbmm_opt(OPT_CONV1X1_STREAM);
    // what dbmm_gemm_f16 gives the eight-phase kernel -- except the conv3 + residual shapes with K <= 256 (layers 2 / 3: two or four K tiles
    // per 256 x 256 tile, all prologue and epilogue): the streaming kernel is 6-7 % faster there (tools/bench_conv_f16.py, B = 1024:
    // 128 -> 512 + residual 397 -> 370 us, 256 -> 1024 + residual 246 -> 230 us)
    const bool gemm8 = (Cout % 256) == 0 && (Cin % 128) == 0 && M >= 16384 && !(residual && Cin <= 256 && mode != 3);
    if ((mode == 2 || ((mode == 1 || mode == 3) && !gemm8)) && (act == DBMM_ACT_NONE || act == DBMM_ACT_RELU) && (Cin % 32) == 0) {
        const int rc = dbmm_conv1x1_stream_f16(x, w, scale, bias, residual, y, M, Cin, Cout, act, stream);
        if (rc != DBMM_E_UNSUPPORTED) return rc;
    }
    return gemm_f16_impl(x, Cin, w, Cin, scale, bias, residual, Cout, 1, y, Cout, M, Cout, Cin, act, stream, 0);
}

// see include/dbmm.h: dbmm_conv1x1_bn_act_f16 with a workspace
extern "C" int dbmm_conv1x1_bn_act_f16_ws(const void* x, const void* w, const float* scale, const float* bias, const void* residual, void* y,
                                          int64_t M, int64_t Cin, int64_t Cout, int act, void* workspace, size_t workspace_bytes, void* stream) {
    // conv3 + residual + ReLU with K = 256 into >= 1024 channels (layer 3): rows owned by one workgroup for a range of 64-channel slabs
    // (conv1x1_res_stream_f16.hip; same arithmetic bit for bit)
    if (dbmm_opt(OPT_CONV1X1_RES_STREAM) && residual && act == DBMM_ACT_RELU && Cin == 256 && Cout >= 1024 && M >= 131072) {
        const int rc = dbmm_conv1x1_res_stream_f16(x, w, scale, bias, residual, y, nullptr, M, 0, 0, Cin, Cout, stream);
        if (rc != DBMM_E_UNSUPPORTED) return rc;
    }
    const int mode = dbmm_opt(OPT_CONV1X1_STREAM);
    const bool gemm8 = (Cout % 256) == 0 && (Cin % 128) == 0 && M >= 16384 && !(residual && Cin <= 256 && mode != 3);
    if ((mode == 2 || ((mode == 1 || mode == 3) && !gemm8)) && (act == DBMM_ACT_NONE || act == DBMM_ACT_RELU) && (Cin % 32) == 0) {
        const int rc = dbmm_conv1x1_stream_f16(x, w, scale, bias, residual, y, M, Cin, Cout, act, stream);
        if (rc != DBMM_E_UNSUPPORTED) return rc;
    }
    return gemm_f16_impl(x, Cin, w, Cin, scale, bias, residual, Cout, 1, y, Cout, M, Cout, Cin, act, stream, 0, workspace, workspace_bytes);
}

// see include/dbmm.h: conv3 + downsample branch of a stage's first block as ONE GEMM on the eight-phase kernel (TWO = 1)
extern "C" int dbmm_conv1x1_dual_bn_act_f16(const void* y2, const void* w3, const float* scale3, const void* xp, const void* wd, const float* ratio,
                                            const float* bias, void* out, int64_t M, int64_t K, int64_t K2, int64_t Cout, int act, void* stream) {
    if (!y2 || !w3 || !scale3 || !xp || !wd || !ratio || !out) return DBMM_E_ARG;
    if (M <= 0 || K <= 0 || K2 <= 0 || Cout <= 0 || M > INT32_MAX) return DBMM_E_SHAPE;
    if (act != DBMM_ACT_NONE && act != DBMM_ACT_RELU) return DBMM_E_ARG;
    if (!dbmm_aligned16(y2) || !dbmm_aligned16(w3) || !dbmm_aligned16(xp) || !dbmm_aligned16(wd) || !dbmm_aligned16(out) || !dbmm_aligned16(scale3) ||
        (bias && !dbmm_aligned16(bias)))
        return DBMM_E_ALIGN;
    // short K (layer 1: 64 + 64 channels) or narrow shapes: the streaming kernel's dual-source mode; the rest on the eight-phase GEMM
    if (!dbmm_opt(OPT_F16_8PH) || (Cout % 256) || (K % 128) || (K2 % 128) || M < 16384)
        return dbmm_conv1x1_dual_stream_f16(y2, w3, scale3, xp, wd, ratio, bias, out, M, K, K2, Cout, act, stream);
    GemmHP p{};
    p.a = (const u16*)y2; p.w = (const u16*)w3; p.bias = bias; p.res = nullptr; p.c = (u16*)out; p.oscale = scale3; p.res_first = 1;
    p.lda = K; p.ldw = K; p.ldr = 0; p.ldc = Cout; p.a_total = M * K * 2; p.w_total = Cout * K * 2;
    p.a2 = (const u16*)xp; p.w2 = (const u16*)wd; p.ratio = ratio; p.lda2 = K2; p.ldw2 = K2; p.a2_total = M * K2 * 2; p.w2_total = Cout * K2 * 2;
    p.K2 = (int)K2;
    p.M = (int)M; p.N = (int)Cout; p.K = (int)K; p.act = act;
    p.tiles_n = (int)(Cout / 256);
    p.n_tiles = (int)((M + 255) / 256) * p.tiles_n;
    p.n_full = p.n_tiles; p.n_cut = 0; p.n_slices = 1; p.ws = nullptr;
    const int grid = p.n_tiles < 256 ? p.n_tiles : 256;
    if (act == DBMM_ACT_RELU) hipLaunchKernelGGL((gemm_f16_8ph_kernel<1, 0, 1>), dim3(grid), dim3(512), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((gemm_f16_8ph_kernel<0, 0, 1>), dim3(grid), dim3(512), 0, (hipStream_t)stream, p);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

namespace {
int gemm_f16_impl(const void* a, int64_t lda, const void* w, int64_t ldw, const float* out_scale, const float* bias, const void* residual,
                  int64_t ldr, int res_first, void* c, int64_t ldc, int64_t M, int64_t N, int64_t K, int act, void* stream, int part,
                  void* workspace, size_t workspace_bytes) {
    if (!a || !w || !c) return DBMM_E_ARG;
    if (M <= 0 || N <= 0 || K <= 0 || M > INT32_MAX || N > INT32_MAX || K > INT32_MAX) return DBMM_E_SHAPE;
    if (act < 0 || act > 2) return DBMM_E_ARG;
    if ((K % 64) || (N & 7)) return DBMM_E_UNSUPPORTED;
    if ((lda & 7) || (ldw & 7) || (ldc & 7) || (residual && (ldr & 7)) || !dbmm_aligned16(a) || !dbmm_aligned16(w) || !dbmm_aligned16(c) ||
        (residual && !dbmm_aligned16(residual)) || (bias && !dbmm_aligned16(bias)) || (out_scale && !dbmm_aligned16(out_scale)))
        return DBMM_E_ALIGN;
    GemmHP p{};
    p.a = (const u16*)a; p.w = (const u16*)w; p.bias = bias; p.res = (const u16*)residual; p.c = (u16*)c;
    p.oscale = out_scale; p.res_first = res_first;
    p.lda = lda; p.ldw = ldw; p.ldr = ldr; p.ldc = ldc;
    p.a_total = ((M - 1) * lda + K) * 2; p.w_total = ((N - 1) * ldw + K) * 2;
    p.M = (int)M; p.N = (int)N; p.K = (int)K; p.act = act;
    // options: f16_bn256 = 0 | 1 (128 x 256 tile for wide GEMMs), f16_8ph = 0: without the deep-pipelined 256 x 256 kernel
    // (a 32-deep chunk at four workgroups per CU measured the same as the 64-deep one and is gone)
    const int bn256 = dbmm_opt(OPT_F16_BN256);
    {
        // part: 0 = the whole problem, 1 = the eight-phase share of a split problem, 2 = its tail (128 x 128 tiles)
        if (part != 2 && dbmm_opt(OPT_F16_8PH) && (N % 256) == 0 && (K % 128) == 0 && (M >= 16384 || part == 1)) {
            p.tiles_n = (int)(N / 256);
            p.n_tiles = (int)((M + 255) / 256) * p.tiles_n;
            // Tile quantisation: the persistent kernel runs ceil(tiles / 256) rounds, and a last round of a few tiles costs a whole
            // one (ViT-B/32 at 512 images: out-proj / c_proj have 100 x 3 = 300 tiles = 2 rounds for 1.17 rounds of work; RN50 layer 3:
            // 784 tiles = 4 rounds for 3.06).  Whole rounds stay here; the rows of the short last round go to the 128 x 128 kernel, a
            // launch of <= 512 tiles (one pass over the chip's 2 x 256 slots) that costs about 0.4 of an eight-phase round.
            // With a workspace the tiles of the short last round (at most 128) are instead cut along K into S = min(256 / tiles, trips)
            // slices, one workgroup each, and gemm_f16_8ph_fixup_kernel sums them and runs the epilogue (as gemm_pair_8ph.hip).
            p.n_full = p.n_tiles; p.n_cut = 0; p.n_slices = 1; p.ws = nullptr;
            if (part == 0 && dbmm_opt(OPT_TAIL_SPLIT) && p.n_tiles > 256 && (p.n_tiles % 256) != 0) {
                const int rem = p.n_tiles % 256, trips = (int)(K / 128);
                int S = rem <= 128 ? 256 / rem : 1;
                S = S < trips ? S : trips;
                const int mode = dbmm_opt(OPT_TAIL_SPLIT);            // 1: by rule, 2: the K cut wherever it applies, 3: the row split only
                // same box, tools/bench_tail_f16.py (profiles/r04_ab_tail_f16.log): the cut pays on long K only -- ViT-B/32's c_proj (K 3072, 300 /
                // 600 tiles) 160 -> 149 us / 261 -> 251 us; K <= 2048 shapes lose 1-10 % to the slices' fixed cost (prologue, 256 KB of fp32
                // partial sums per slice against a 128-KB fp16 tile, the second launch)
                const bool cut_pays = K >= 3072 && rem >= 32;
                if ((mode == 2 || (mode == 1 && cut_pays)) && workspace && dbmm_aligned16(workspace) && S >= 2 &&
                    (size_t)rem * S * (128 * 512 * sizeof(float)) <= workspace_bytes) {
                    p.n_full = p.n_tiles - rem; p.n_cut = rem; p.n_slices = S; p.ws = (float*)workspace;
                } else {
                const int64_t mt = (M + 255) / 256, full_rounds = p.n_tiles / 256, mt_full = full_rounds * 256 / p.tiles_n;
                const int64_t m_split = mt_full * 256, tail_rows = M - m_split;
                const int64_t tail_tiles = ((tail_rows + 127) / 128) * ((N + 127) / 128);
                // (the row split: +1-2 % on the 300-tile projections of ViT-B/32 at 512 images, a loss from 600 tiles on)
                if ((mode == 3 || (mode == 1 && p.n_tiles <= 512)) && mt_full >= 1 && mt_full < mt && tail_tiles <= 512 && (p.n_tiles % 256) <= 128) {
                    int rc = gemm_f16_impl(a, lda, w, ldw, out_scale, bias, residual, ldr, res_first, c, ldc, m_split, N, K, act, stream, 1, nullptr, 0);
                    if (rc != DBMM_OK) return rc;
                    return gemm_f16_impl((const u16*)a + m_split * lda, lda, w, ldw, out_scale, bias,
                                         residual ? (const void*)((const u16*)residual + m_split * ldr) : nullptr, ldr, res_first,
                                         (u16*)c + m_split * ldc, ldc, tail_rows, N, K, act, stream, 2, nullptr, 0);
                }
                }
            }
            const int grid = p.n_tiles < 256 ? p.n_tiles : 256;   // persistent: one workgroup per CU
            hipStream_t s8 = (hipStream_t)stream;
#define DBMM_8PH(A, R) hipLaunchKernelGGL((gemm_f16_8ph_kernel<A, R>), dim3(grid), dim3(512), 0, s8, p)
            if (residual) { if (act == 0) DBMM_8PH(0, 1); else if (act == 1) DBMM_8PH(1, 1); else DBMM_8PH(2, 1); }
            else { if (act == 0) DBMM_8PH(0, 0); else if (act == 1) DBMM_8PH(1, 0); else DBMM_8PH(2, 0); }
#undef DBMM_8PH
            DBMM_CHECK_LAUNCH();
            if (p.n_cut) {
#define DBMM_8PHF(A, R) hipLaunchKernelGGL((gemm_f16_8ph_fixup_kernel<A, R>), dim3(p.n_cut, 4), dim3(512), 0, s8, p)
                if (residual) { if (act == 0) DBMM_8PHF(0, 1); else if (act == 1) DBMM_8PHF(1, 1); else DBMM_8PHF(2, 1); }
                else { if (act == 0) DBMM_8PHF(0, 0); else if (act == 1) DBMM_8PHF(1, 0); else DBMM_8PHF(2, 0); }
#undef DBMM_8PHF
                DBMM_CHECK_LAUNCH();
            }
            return DBMM_OK;
        }
    }
    const bool wide = bn256 && N >= 768 && (N % 256) == 0 && M >= 32768;     // (ViT-B/32 at 25,600 rows measured 3 % slower on it)
    const int bn = wide ? 256 : 128;
    p.tiles_n = (int)((N + bn - 1) / bn);
    p.n_tiles = (int)((M + GBM - 1) / GBM) * p.tiles_n;
    if (wide) hipLaunchKernelGGL((gemm_f16_kernel<32, 256, 2>), dim3(p.n_tiles), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((gemm_f16_kernel<64, 128, 2>), dim3(p.n_tiles), dim3(256), 0, (hipStream_t)stream, p);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}
}  // namespace

extern "C" int dbmm_mha_core_f16(const void* qkv, void* out, int64_t B, int64_t L, int64_t E, int64_t heads, int causal,
                                 void* stream) {
    if (!qkv || !out) return DBMM_E_ARG;
    if (B <= 0 || L <= 0 || E <= 0 || heads <= 0 || B > 65535 || heads > 65535) return DBMM_E_SHAPE;
    if (E != heads * 64) return DBMM_E_UNSUPPORTED;              // head_dim 64 (every CLIP tower)
    if (!dbmm_aligned16(qkv) || !dbmm_aligned16(out)) return DBMM_E_ALIGN;
    // (64 queries per wave measured 8 % slower on ViT-L/14@336 -- 256 registers hold occupancy at 2 workgroups per CU where
    // this 32-query form, 161 registers, runs 3 -- and is not instantiated)
    if (L <= 64 && dbmm_opt(OPT_MHA_SHORT)) {
        hipLaunchKernelGGL((mha_f16_kernel<1, 2>), dim3(1, (unsigned)heads, (unsigned)B), dim3(128), 0, (hipStream_t)stream, (const u16*)qkv, (u16*)out,
                           (int)L, (int)E, (int)heads, causal ? 1 : 0, 0.125f * 1.4426950408889634f);
    } else {
        const dim3 grid((unsigned)((L + 127) / 128), (unsigned)heads, (unsigned)B);
        hipLaunchKernelGGL(mha_f16_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, (const u16*)qkv, (u16*)out, (int)L, (int)E,
                           (int)heads, causal ? 1 : 0, 0.125f * 1.4426950408889634f);
    }
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

extern "C" int dbmm_layernorm_f16(const void* x, int64_t ldx, const float* gamma, const float* beta, void* y, int64_t ldy,
                                  int64_t rows, int64_t E, float eps, void* stream) {
    if (!x || !gamma || !beta || !y) return DBMM_E_ARG;
    if (rows <= 0 || E <= 0 || rows > INT32_MAX) return DBMM_E_SHAPE;
    if ((E & 7) || (ldx & 7) || (ldy & 7)) return DBMM_E_SHAPE;
    if (!dbmm_aligned16(x) || !dbmm_aligned16(y) || !dbmm_aligned16(gamma) || !dbmm_aligned16(beta)) return DBMM_E_ALIGN;
    const long long blocks = (rows + 3) / 4;
    hipLaunchKernelGGL(layernorm_f16_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream,
                       (const u16*)x, (long long)ldx, gamma, beta, (u16*)y, (long long)ldy, (int)rows, (int)(E / 8), eps);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

extern "C" int dbmm_im2col_patch_f16(const void* x_nchw, int x_is_f16, void* out, int64_t B, int64_t R, int64_t P, int64_t Kp,
                                     void* stream) {
    if (!x_nchw || !out) return DBMM_E_ARG;
    if (B <= 0 || R <= 0 || P <= 0 || R % P || Kp < 3 * P * P) return DBMM_E_SHAPE;
    const int g = (int)(R / P);
    const int vec = (P % 4 == 0) ? 4 : ((P % 2 == 0) ? 2 : 1);
    const long long n_vec = (long long)B * 3 * R * (R / vec), n_pad = (long long)B * g * g * (Kp - 3 * P * P);
    const dim3 grid(grid_for(n_vec));
    hipStream_t s = (hipStream_t)stream;
#define DBMM_IM2COL(T, V) hipLaunchKernelGGL((im2col_patch_f16_kernel<T, V>), grid, dim3(256), 0, s, (const T*)x_nchw, (u16*)out, (int)R, \
                                             (int)P, g, (int)Kp, n_vec, n_pad)
    if (x_is_f16) { if (vec == 4) DBMM_IM2COL(_Float16, 4); else if (vec == 2) DBMM_IM2COL(_Float16, 2); else DBMM_IM2COL(_Float16, 1); }
    else { if (vec == 4) DBMM_IM2COL(float, 4); else if (vec == 2) DBMM_IM2COL(float, 2); else DBMM_IM2COL(float, 1); }
#undef DBMM_IM2COL
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

extern "C" int dbmm_vit_tokens_f16(const void* patches, const float* cls, const float* pos, void* out, int64_t B, int64_t L,
                                   int64_t W, void* stream) {
    if (!patches || !cls || !pos || !out) return DBMM_E_ARG;
    if (B <= 0 || L <= 1 || W <= 0) return DBMM_E_SHAPE;
    const long long total = (long long)B * L * W;
    hipLaunchKernelGGL(vit_tokens_f16_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const u16*)patches, cls, pos,
                       (u16*)out, (int)L, (int)W, total);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

extern "C" int dbmm_embed_gather_f16(const int32_t* tokens, const float* table, const float* pos, void* out, int64_t n, int64_t L,
                                     int64_t W, int64_t vocab, void* stream) {
    if (!tokens || !table || !pos || !out) return DBMM_E_ARG;
    if (n <= 0 || L <= 0 || W <= 0 || vocab <= 0) return DBMM_E_SHAPE;
    const long long total = (long long)n * L * W;
    hipLaunchKernelGGL(embed_gather_f16_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, tokens, table, pos, (u16*)out,
                       (int)L, (int)W, (int)vocab, total);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

extern "C" int dbmm_gather_eot_f16(const int32_t* tokens, const void* x, void* out, int64_t n, int64_t L, int64_t W, void* stream) {
    if (!tokens || !x || !out) return DBMM_E_ARG;
    if (n <= 0 || L <= 0 || W <= 0) return DBMM_E_SHAPE;
    hipLaunchKernelGGL(gather_eot_f16_kernel, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, tokens, (const u16*)x, (u16*)out, (int)L,
                       (int)W);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

extern "C" int dbmm_cast_f32_f16(const float* x, void* y, int64_t n, void* stream) {
    if (!x || !y) return DBMM_E_ARG;
    if (n <= 0) return DBMM_E_SHAPE;
    hipLaunchKernelGGL(cast_f32_f16_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, (u16*)y, (long long)n);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}
