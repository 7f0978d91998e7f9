// ModifiedResNet kernels that are not GEMM-shaped: stem conv (Cin = 3), average pooling,
// attention-pool token assembly and its one-query attention core.  All HBM-bound
// (coalesced, float4 where the layout allows); the heavy projections go through
// dbmm_gemm_bias_act.
#include <stdlib.h>
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------
// stem conv1: 3x3, stride 2, pad 1, Cin = 3, NCHW image -> NHWC, + bias (folded BN) + ReLU.
// One thread per output pixel: its 27 input taps live in registers, weights [27][Cout] are
// broadcast from LDS in groups of 8 output channels.  0.2 % of the network's FLOPs, but 1.6 GB of
// output at B = 1024: the results go through an LDS staging tile so that the workgroup's 256
// pixels x Cout channels leave as ONE contiguous run of 16-B lane stores (a thread storing its own
// pixel's channels writes 16 B into 64 different 128-B lines per instruction: 2.6 TB/s).
// ---------------------------------------------------------------------------------------
template <int COUT>
__global__ __launch_bounds__(256) void stem_s2_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ y,
                                                      float* __restrict__ y_absmax, int B, int H, int W, int Ho,
                                                      int Wo) {
    constexpr int Cout = COUT, PITCH = COUT + 4;                // staging row pitch (floats): 16-B aligned, rows shifted by 4 banks
    __shared__ __attribute__((aligned(16))) float sw[28 * COUT];        // [27][Cout] then bias[Cout]
    __shared__ __attribute__((aligned(16))) float stage[256 * PITCH];
    for (int i = threadIdx.x; i < 27 * Cout; i += blockDim.x) sw[i] = w[i];
    for (int i = threadIdx.x; i < Cout; i += blockDim.x) sw[27 * Cout + i] = bias ? bias[i] : 0.f;
    __syncthreads();
    const long long m0 = (long long)blockIdx.x * blockDim.x, m = m0 + threadIdx.x;
    const long long M = (long long)B * Ho * Wo;
    float omax = 0.f;                       // outputs are post-ReLU: max == max|y|
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
                xin[(kh * 3 + kw) * 3 + c] = ok ? x[(((long long)n * 3 + c) * H + hi) * W + wi] : 0.f;
        }
    float* so = stage + threadIdx.x * PITCH;
    for (int c0 = 0; c0 < Cout; c0 += 8) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            // (weights as scalar loads straight from memory, SGPR operands of the FMAs, measured 0.93 ms against
            //  0.67 ms through LDS at B = 1024: the SGPR file spills into VGPR lanes)
            const f32x4 w0 = *(const f32x4*)(sw + t * Cout + c0);
            const f32x4 w1 = *(const f32x4*)(sw + t * Cout + c0 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = fmaf(xin[t], w0[j], acc[j]);
                acc[4 + j] = fmaf(xin[t], w1[j], acc[4 + j]);
            }
        }
        f32x4 o0, o1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o0[j] = fmaxf(acc[j] + sw[27 * Cout + c0 + j], 0.f);
            o1[j] = fmaxf(acc[4 + j] + sw[27 * Cout + c0 + 4 + j], 0.f);
        }
        *(f32x4*)(so + c0) = o0;
        *(f32x4*)(so + c0 + 4) = o1;
#pragma unroll
        for (int j = 0; j < 4; ++j) omax = fmaxf(omax, fmaxf(o0[j], o1[j]));
    }
    }
    __syncthreads();
    // the workgroup's pixels m0 .. m0 + 255 are contiguous in y ([M][Cout]): 16-B chunk q of the run = pixel q / (Cout/4)
    constexpr int QPP = COUT / 4;
    const long long run = (M - m0 < 256 ? M - m0 : 256) * QPP;
    f32x4* yo = (f32x4*)(y + m0 * Cout);
#pragma unroll 4
    for (int q = threadIdx.x; q < 256 * QPP; q += 256)
        if (q < run) yo[q] = *(const f32x4*)(stage + (q / QPP) * PITCH + (q % QPP) * 4);
    if (y_absmax) {       // one (filtered) atomic per workgroup: every workgroup targets the same address
        __shared__ float wmax[4];
        omax = wave_max(omax);
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = omax;
        __syncthreads();
        if (threadIdx.x == 0) {
            omax = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
            if (omax > *(volatile const float*)y_absmax) atomicMax((unsigned*)y_absmax, __float_as_uint(omax));
        }
    }
}

// ---------------------------------------------------------------------------------------
// stem conv1 on the matrix cores (Cout = 32 or 64).  The 27-tap gather is the A operand of a
// 32 pixels x 32 (27 + 5 zero) K x Cout product: each lane fetches the 16 taps its fragment rows
// need straight from the NCHW image (masked taps through the descriptor's out-of-range zero),
// splits them into fp16 hi + lo under the WAVE's own exact power-of-two scale (a wave's outputs
// depend on its own 32 pixels only, so no tensor-wide maximum is needed), the weights -- folded
// BatchNorm, not fp16-exact -- are split the same way once per wave, and an fp32 product is the
// three MFMA products (lo,hi) (hi,lo) (hi,hi): 12 MFMAs per 32 pixels replace 864 FMAs per pixel,
// no LDS at all.  The accumulator layout (column = channel) stores two whole 128-B pixel rows per
// wave instruction.  Bound: HBM (0.15 MB in + 1.6 MB out per image at 224 px).
// ---------------------------------------------------------------------------------------
typedef _Float16 stem_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int stem_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ stem_f16x8 stem_frag(const unsigned (&v)[4]) { return __builtin_bit_cast(stem_f16x8, (stem_u32x4){v[0], v[1], v[2], v[3]}); }

__device__ __forceinline__ int stem_scale_exp(float amax) {          // 2^s * amax in [2^13, 2^14): fp16 hi + lo keep 22 bits
    const unsigned b = __float_as_uint(amax) & 0x7fffffffu;
    int s = b ? 13 - ((int)(b >> 23) - 127) : 0;
    return s < -60 ? -60 : (s > 60 ? 60 : s);
}
__device__ __forceinline__ float stem_pow2(int e) { return __uint_as_float((unsigned)(e + 127) << 23); }
__device__ __forceinline__ void stem_split_pair(float x0, float x1, float sc, unsigned& hi, unsigned& lo) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(hi) : "v"(x0), "v"(sc));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(hi) : "v"(x1), "v"(sc));
    asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(lo) : "v"(x0), "v"(sc), "v"(hi));
    asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(lo) : "v"(x1), "v"(sc), "v"(hi));
#else
    (void)x0; (void)x1; (void)sc; hi = lo = 0;
#endif
}

#ifndef STEM_PREFETCH
#define STEM_PREFETCH 1
#define STEM_MINB(NB) (NB == 1 ? 3 : 2)
#endif
template <int NB>                                                    // NB = Cout / 32
__global__ __launch_bounds__(256, STEM_MINB(NB)) void stem_s2_mfma_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ y,
                                                           float* __restrict__ y_absmax, int B, int H, int W, int Ho, int Wo,
                                                           int n_blocks) {
    constexpr int Cout = NB * 32;
    constexpr unsigned OOR = 0x80000000u;
    // (the wave index through readfirstlane: the block number and with it the buffer descriptors are then PROVABLY
    //  wave-uniform; derived from threadIdx alone the compiler wrapped each of the 48 buffer operations of a block in a
    //  waterfall loop -- 208 v_readfirstlane in the listing)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 31, fh = lane >> 5;
    const long long M = (long long)B * Ho * Wo, img = 3LL * H * W;
    // K slot (fh, t = ks * 8 + i) -> tap.  Any bijection works as long as both operands use it; this one keeps a slot's
    // two taps (fh = 0 / 1) compile-time constants, so offsets and validity bits are scalar selects instead of 32 VGPRs:
    //   t < 9: (kw, c) = (t / 3, t % 3), kh = fh;   t >= 9: kh = 2, (kw, c) pair t - 9 (fh = 0) or t - 2 (fh = 1, t = 9, 10)
    auto slot_tap = [](int f, int t, int& kh, int& kw, int& c) -> bool {
        int pr;
        if (t < 9) { kh = f; pr = t; }
        else { kh = 2; pr = f ? t - 2 : t - 9; if (pr > 8) { kh = kw = c = 0; return false; } }
        kw = pr / 3; c = pr - kw * 3;
        return true;
    };
    // weights: B operand, column n = channel j * 32 + fr, the same K indices; split under the wave's scale
    unsigned wh[NB][2][4], wlo[NB][2][4];
    float bv[NB];
    int e_w;
    {
        float wv[NB][16], wmax = 0.f;
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                int kh0, kw0, c0, kh1, kw1, c1;
                const bool v0 = slot_tap(0, t, kh0, kw0, c0), v1 = slot_tap(1, t, kh1, kw1, c1);
                const int k = fh ? (kh1 * 3 + kw1) * 3 + c1 : (kh0 * 3 + kw0) * 3 + c0;
                wv[j][t] = (fh ? v1 : v0) ? w[k * Cout + j * 32 + fr] : 0.f;
                wmax = fmaxf(wmax, fabsf(wv[j][t]));
            }
        e_w = stem_scale_exp(wave_max(wmax));
        const float sc = stem_pow2(e_w);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int q = 0; q < 4; ++q) stem_split_pair(wv[j][ks * 8 + 2 * q], wv[j][ks * 8 + 2 * q + 1], sc, wh[j][ks][q], wlo[j][ks][q]);
            bv[j] = bias ? bias[j * 32 + fr] : 0.f;
        }
    }
    const int n_waves = gridDim.x * 4;
    float xv[16];
    long long img0 = 0;                                              // image of the block's first pixel (descriptor base)
    auto gather = [&](int blk) {
        const long long m = (long long)blk * 32 + fr;
        const long long mb = (long long)blk * 32;
        img0 = mb / ((long long)Ho * Wo);
        const long long left = (long long)B - img0;
        const long long span = 31 / ((long long)Ho * Wo) + 2;         // images a block of 32 pixels can touch (two, unless an image has < 31 of them)
        const long long ext = (left < span ? left : span) * img * 4;
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
            xv[t] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, ok ? (base + (unsigned)(fh ? off1 : off0)) * 4u : OOR, 0, 0));
        }
    };
    float omax = 0.f;
    int blk = blockIdx.x * 4 + wave;
    if (STEM_PREFETCH && blk < n_blocks) gather(blk);
    for (; blk < n_blocks; blk += n_waves) {
        if (!STEM_PREFETCH) gather(blk);
        float amax = 0.f;
#pragma unroll
        for (int t = 0; t < 16; ++t) amax = fmaxf(amax, fabsf(xv[t]));
        const int e_a = stem_scale_exp(wave_max(amax));
        const float sc = stem_pow2(e_a);
        unsigned ah[2][4], al[2][4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int q = 0; q < 4; ++q) stem_split_pair(xv[ks * 8 + 2 * q], xv[ks * 8 + 2 * q + 1], sc, ah[ks][q], al[ks][q]);
        const long long mb = (long long)blk * 32;
        if (STEM_PREFETCH && blk + n_waves < n_blocks) gather(blk + n_waves);   // next block's taps in flight during the MFMAs and stores
        const float osc = stem_pow2(-e_a - e_w);
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(stem_frag(al[ks]), stem_frag(wh[j][ks]), acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(stem_frag(ah[ks]), stem_frag(wlo[j][ks]), acc, 0, 0, 0);
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(stem_frag(ah[ks]), stem_frag(wh[j][ks]), acc, 0, 0, 0);
            // accumulator register r = pixel row (r & 3) + 8 (r >> 2) + 4 fh, column = channel j * 32 + fr
            const long long rows_left = M - mb;
            const long long bytes = (rows_left < 32 ? rows_left : 32) * Cout * 4;
            const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc((void*)(y + mb * Cout), 0, (int)bytes, 0x00020000);
            const unsigned voff = (unsigned)((4 * fh * Cout + j * 32 + fr) * 4);
            float bmax = 0.f;
            // rows >= M (only a ragged last block has any) are masked by lane: their row offset sits in soffset and may
            // exceed num_records, which the range check `offset >= num_records - soffset` should not be asked to survive
            const int row_lim = (int)(rows_left < 32 ? rows_left : 32) - 4 * fh;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row0 = (r & 3) + 8 * (r >> 2);                          // + 4 fh
                const float v = fmaxf(fmaf(acc[r], osc, bv[j]), 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), ry, row0 < row_lim ? voff : OOR,
                                                      (unsigned)(row0 * Cout * 4), 0);
                bmax = fmaxf(bmax, v);
            }
            // rows past M exist only in the very last block: their lanes gathered zeros, so v = relu(bias) there
            if (rows_left >= 32) omax = fmaxf(omax, bmax);
            else
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if ((r & 3) + 8 * (r >> 2) + 4 * fh < rows_left) omax = fmaxf(omax, fmaxf(fmaf(acc[r], osc, bv[j]), 0.f));
        }
    }
    if (y_absmax) {
        __shared__ float wmax_s[4];
        omax = wave_max(omax);
        if (lane == 0) wmax_s[wave] = omax;
        __syncthreads();
        if (threadIdx.x == 0) {
            omax = fmaxf(fmaxf(wmax_s[0], wmax_s[1]), fmaxf(wmax_s[2], wmax_s[3]));
            if (omax > *(volatile const float*)y_absmax) atomicMax((unsigned*)y_absmax, __float_as_uint(omax));
        }
    }
}

// any Cout % 8 == 0 (tiny test geometries): every thread stores its own pixel's channels
__global__ __launch_bounds__(256) void stem_s2_generic_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ y,
                                                      float* __restrict__ y_absmax, int B, int H, int W, int Ho,
                                                      int Wo, int Cout) {
    extern __shared__ __attribute__((aligned(16))) float sw[];  // [27][Cout] then bias[Cout]
    for (int i = threadIdx.x; i < 27 * Cout; i += blockDim.x) sw[i] = w[i];
    for (int i = threadIdx.x; i < Cout; i += blockDim.x) sw[27 * Cout + i] = bias ? bias[i] : 0.f;
    __syncthreads();
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long M = (long long)B * Ho * Wo;
    float omax = 0.f;                       // outputs are post-ReLU: max == max|y|
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
                xin[(kh * 3 + kw) * 3 + c] = ok ? x[(((long long)n * 3 + c) * H + hi) * W + wi] : 0.f;
        }
    float* yo = y + m * Cout;
    for (int c0 = 0; c0 < Cout; c0 += 8) {
        float acc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
        for (int t = 0; t < 27; ++t) {
            const f32x4 w0 = *(const f32x4*)(sw + t * Cout + c0);
            const f32x4 w1 = *(const f32x4*)(sw + t * Cout + c0 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[j] = fmaf(xin[t], w0[j], acc[j]);
                acc[4 + j] = fmaf(xin[t], w1[j], acc[4 + j]);
            }
        }
        f32x4 o0, o1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o0[j] = fmaxf(acc[j] + sw[27 * Cout + c0 + j], 0.f);
            o1[j] = fmaxf(acc[4 + j] + sw[27 * Cout + c0 + 4 + j], 0.f);
        }
        *(f32x4*)(yo + c0) = o0;
        *(f32x4*)(yo + c0 + 4) = o1;
#pragma unroll
        for (int j = 0; j < 4; ++j) omax = fmaxf(omax, fmaxf(o0[j], o1[j]));
    }
    }
    if (y_absmax) {       // one (filtered) atomic per workgroup: every workgroup targets the same address
        __shared__ float wmax[4];
        omax = wave_max(omax);
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = omax;
        __syncthreads();
        if (threadIdx.x == 0) {
            omax = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
            if (omax > *(volatile const float*)y_absmax) atomicMax((unsigned*)y_absmax, __float_as_uint(omax));
        }
    }
}

// ---------------------------------------------------------------------------------------
// AvgPool2d(k), kernel = stride = k, NHWC, one float4 of channels per thread.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void avgpool_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                      int H, int W, int C4, int Ho, int Wo, int k,
                                                      long long total) {
    const float inv = 1.f / (float)(k * k);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        long long t = i / C4;
        const int wo = (int)(t % Wo); t /= Wo;
        const int ho = (int)(t % Ho);
        const long long n = t / Ho;
        const f32x4* src = (const f32x4*)x + ((n * H + (long long)ho * k) * W + (long long)wo * k) * C4 + c;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int a = 0; a < k; ++a)
            for (int b = 0; b < k; ++b) s += src[((long long)a * W + b) * C4];
        ((f32x4*)y)[i] = s * inv;
    }
}

// ---------------------------------------------------------------------------------------
// attention-pool tokens: t[b][0] = mean_j x[b][j] + pos[0]; t[b][1+j] = x[b][j] + pos[1+j]
// ---------------------------------------------------------------------------------------
// (token rows are padded with zero rows up to Lp = a multiple of 4 so that the token matrix can
//  be used as a K-major GEMM operand)
// XH = 1: x is fp16 (the fp16 mode's feature map, read directly instead of through a cast pass)
template <int XH>
__global__ __launch_bounds__(256) void attnpool_tokens_kernel(const void* __restrict__ x,
                                                              const float* __restrict__ pos,
                                                              float* __restrict__ t, int HW, int C4, int Lp) {
    const int b = blockIdx.y;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C4) return;
    const f32x4* pp = (const f32x4*)pos + c;
    f32x4* tb = (f32x4*)t + (long long)b * Lp * C4 + c;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < HW; ++j) {
        f32x4 v;
        if constexpr (XH) {
            typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
            const f16x4 h = ((const f16x4*)x)[((long long)b * HW + j) * C4 + c];
            v = (f32x4){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
        } else {
            v = ((const f32x4*)x)[((long long)b * HW + j) * C4 + c];
        }
        s += v;
        tb[(long long)(j + 1) * C4] = v + pp[(long long)(j + 1) * C4];
    }
    tb[0] = s * (1.f / (float)HW) + pp[0];
    for (int j = HW + 1; j < Lp; ++j) tb[(long long)j * C4] = (f32x4){0.f, 0.f, 0.f, 0.f};
}

// in-place softmax over the first L entries of rows of length Lp (padding entries -> 0);
// one wave per row
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ s, int rows, int L, int Lp) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float* r = s + (long long)row * Lp;
    float m = -INFINITY;
    for (int j = lane; j < L; j += 64) m = fmaxf(m, r[j]);
    m = wave_max(m);
    float sum = 0.f;
    for (int j = lane; j < L; j += 64) sum += expf(r[j] - m);
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int j = lane; j < Lp; j += 64) r[j] = (j < L) ? expf(r[j] - m) * inv : 0.f;
}

}  // namespace

extern "C" int dbmm_conv_stem_s2(const float* x_nchw, const float* w, const float* bias, float* y_nhwc,
                                 float* y_absmax, int64_t B, int64_t H, int64_t W, int64_t Cout, void* stream) {
    if (!x_nchw || !w || !y_nhwc) return DBMM_E_ARG;
    if (B <= 0 || H <= 0 || W <= 0 || Cout <= 0 || (Cout & 7) || Cout > 512) return DBMM_E_SHAPE;
    if (!dbmm_aligned16(y_nhwc)) return DBMM_E_ALIGN;
    const int64_t Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const int64_t M = B * Ho * Wo;
    if (M > INT32_MAX) return DBMM_E_SHAPE;
    const dim3 g((unsigned)((M + 255) / 256));
    hipStream_t s = (hipStream_t)stream;
#define DBMM_STEM(C) hipLaunchKernelGGL(stem_s2_kernel<C>, g, dim3(256), 0, s, x_nchw, w, bias, y_nhwc, y_absmax, (int)B, (int)H, (int)W, (int)Ho, (int)Wo)
    // option stem_mfma = 0: the FMA kernels
    if (dbmm_opt(OPT_STEM_MFMA) && (Cout == 32 || Cout == 64) && 3 * H * W * 8 < 0x7FFFFFF0LL && (M + 31) / 32 <= INT32_MAX) {
        const int n_blocks = (int)((M + 31) / 32);
        const int wgs = (n_blocks + 3) / 4 < 256 * 8 ? (n_blocks + 3) / 4 : 256 * 8;
        if (Cout == 32)
            hipLaunchKernelGGL(stem_s2_mfma_kernel<1>, dim3(wgs), dim3(256), 0, s, x_nchw, w, bias, y_nhwc, y_absmax, (int)B, (int)H,
                               (int)W, (int)Ho, (int)Wo, n_blocks);
        else
            hipLaunchKernelGGL(stem_s2_mfma_kernel<2>, dim3(wgs), dim3(256), 0, s, x_nchw, w, bias, y_nhwc, y_absmax, (int)B, (int)H,
                               (int)W, (int)Ho, (int)Wo, n_blocks);
        DBMM_CHECK_LAUNCH();
        return DBMM_OK;
    }
    if (Cout == 32) DBMM_STEM(32);            // RN50 / RN101
    else if (Cout == 40) DBMM_STEM(40);       // RN50x4
    else if (Cout == 48) DBMM_STEM(48);       // RN50x16
    else if (Cout == 64) DBMM_STEM(64);       // RN50x64
    else
        hipLaunchKernelGGL(stem_s2_generic_kernel, g, dim3(256), (size_t)(28 * Cout) * sizeof(float), s, x_nchw, w, bias, y_nhwc,
                           y_absmax, (int)B, (int)H, (int)W, (int)Ho, (int)Wo, (int)Cout);
#undef DBMM_STEM
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

extern "C" int dbmm_avgpool2d(const float* x, float* y, int64_t B, int64_t H, int64_t W, int64_t C, int64_t k,
                              void* stream) {
    if (!x || !y) return DBMM_E_ARG;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || k <= 0 || (C & 3) || H % k || W % k) return DBMM_E_SHAPE;
    if (!dbmm_aligned16(x) || !dbmm_aligned16(y)) return DBMM_E_ALIGN;
    const int64_t Ho = H / k, Wo = W / k, C4 = C / 4;
    const long long total = (long long)B * Ho * Wo * C4;
    const long long blocks = (total + 255) / 256;
    const unsigned grid = (unsigned)(blocks < 16384 ? blocks : 16384);
    hipLaunchKernelGGL(avgpool_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, y, (int)H, (int)W,
                       (int)C4, (int)Ho, (int)Wo, (int)k, total);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

extern "C" size_t dbmm_workspace_bytes_attnpool(int64_t B, int64_t HW, int64_t C) {
    // tokens [B][Lp][C] + q [B][C] + U/Sx [B][heads][C] + scores [B][heads][Lp] + o [B][C]
    const int64_t Lp = (HW + 1 + 3) / 4 * 4, heads = C / 64;
    return (size_t)(B * Lp * C + 2 * B * C + B * heads * C + B * heads * Lp) * sizeof(float);
}

// AttentionPool2d with the single query exploited algebraically (clip/model.py:68-91 computes
// k and v projections of all HW+1 tokens although only token 0 queries):
//   score_h[j] = q_h . (Wk_h t_j + bk_h) = (Wk_h^T q_h) . t_j + const   -> softmax drops the const
//   out_h      = sum_j p_h[j] (Wv_h t_j + bv_h) = Wv_h (sum_j p_h[j] t_j) + bv_h   (sum_j p = 1)
// i.e. 2*(HW+1)*C*C MACs per image become ~4*C*C.  All products are (batched) MFMA GEMMs.
extern "C" int dbmm_attnpool_x(const void* x, int x_is_f16, const float* pos, const float* wq, const float* bq,
                               const float* wkv, const float* bkv, const float* wc, const float* bc, float* out,
                               int64_t B, int64_t HW, int64_t C, int64_t heads, int64_t Dout, void* workspace,
                               size_t workspace_bytes, void* stream);

extern "C" int dbmm_attnpool(const float* x, const float* pos, const float* wq, const float* bq,
                             const float* wkv, const float* bkv, const float* wc, const float* bc, float* out,
                             int64_t B, int64_t HW, int64_t C, int64_t heads, int64_t Dout, void* workspace,
                             size_t workspace_bytes, void* stream) {
    return dbmm_attnpool_x(x, 0, pos, wq, bq, wkv, bkv, wc, bc, out, B, HW, C, heads, Dout, workspace, workspace_bytes, stream);
}

// see include/dbmm.h: the same with the feature map in fp32 (x_is_f16 = 0) or fp16 (1; every later product in fp32 as before)
extern "C" int dbmm_attnpool_x(const void* x, int x_is_f16, const float* pos, const float* wq, const float* bq,
                               const float* wkv, const float* bkv, const float* wc, const float* bc, float* out,
                               int64_t B, int64_t HW, int64_t C, int64_t heads, int64_t Dout, void* workspace,
                               size_t workspace_bytes, void* stream) {
    if (!x || !pos || !wq || !wkv || !bkv || !wc || !out || !workspace) return DBMM_E_ARG;
    if (B <= 0 || HW <= 0 || C <= 0 || heads <= 0 || C != heads * 64 || (C & 3) || Dout <= 0 || B > 65535)
        return DBMM_E_SHAPE;
    if (workspace_bytes < dbmm_workspace_bytes_attnpool(B, HW, C)) return DBMM_E_WORKSPACE;
    if (!dbmm_aligned16(x) || !dbmm_aligned16(pos) || !dbmm_aligned16(workspace)) return DBMM_E_ALIGN;
    const int64_t L = HW + 1, Lp = (L + 3) / 4 * 4;
    float* tok = (float*)workspace;          // [B][Lp][C]
    float* q = tok + B * Lp * C;             // [B][C]
    float* U = q + B * C;                    // [B][heads][C]   (re-used for Sx)
    float* S = U + B * heads * C;            // [B][heads][Lp]
    float* o = S + B * heads * Lp;           // [B][C]
    hipStream_t s = (hipStream_t)stream;
    const int C4 = (int)(C / 4);
    if (x_is_f16)
        hipLaunchKernelGGL(attnpool_tokens_kernel<1>, dim3((C4 + 255) / 256, (unsigned)B), dim3(256), 0, s, x, pos, tok, (int)HW, C4, (int)Lp);
    else
        hipLaunchKernelGGL(attnpool_tokens_kernel<0>, dim3((C4 + 255) / 256, (unsigned)B), dim3(256), 0, s, x, pos, tok, (int)HW, C4, (int)Lp);
    DBMM_CHECK_LAUNCH();
    int rc;
    // q = (t_0 Wq^T + bq) * head_dim^-0.5          (token 0 of every image: row stride Lp*C)
    rc = dbmm_gemm_bias_act(tok, Lp * C, 0, wq, C, 0, bq, nullptr, 0, q, C, B, C, C, 0.125f, DBMM_ACT_NONE, stream);
    if (rc) return rc;
    // U[b][h][:] = Wk_h^T q_h : per head, A = q[:, 64h:64h+64], W = Wk rows 64h.. as [K=64][N=C]
    rc = dbmm_gemm_batched(q, C, 64, 0, wkv, C, 64 * C, 1, nullptr, 0, U, heads * C, C, B, C, 64, heads, 1.f,
                           DBMM_ACT_NONE, stream);
    if (rc) return rc;
    // S[b][h][j] = U[b][h] . t[b][j] : per image, A = U[b] [heads][C], W = t[b] [Lp][C]
    rc = dbmm_gemm_batched(U, C, heads * C, 0, tok, C, Lp * C, 0, nullptr, 0, S, Lp, heads * Lp, heads, Lp, C, B, 1.f,
                           DBMM_ACT_NONE, stream);
    if (rc) return rc;
    const int64_t rows = B * heads;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, S, (int)rows, (int)L,
                       (int)Lp);
    DBMM_CHECK_LAUNCH();
    // Sx[b][h][:] = sum_j p[b][h][j] t[b][j][:] : per image, A = P[b] [heads][Lp], W = t[b] K-major [Lp][C]
    float* Sx = U;
    rc = dbmm_gemm_batched(S, Lp, heads * Lp, 0, tok, C, Lp * C, 1, nullptr, 0, Sx, C, heads * C, heads, C, Lp, B, 1.f,
                           DBMM_ACT_NONE, stream);
    if (rc) return rc;
    // o[:, 64h:64h+64] = Sx[:, h, :] Wv_h^T + bv_h : per head
    rc = dbmm_gemm_batched(Sx, heads * C, C, 0, wkv + C * C, C, 64 * C, 0, bkv + C, 64, o, C, 64, B, 64, C, heads, 1.f,
                           DBMM_ACT_NONE, stream);
    if (rc) return rc;
    return dbmm_gemm_bias_act(o, C, 0, wc, C, 0, bc, nullptr, 0, out, Dout, B, Dout, C, 1.f, DBMM_ACT_NONE, stream);
}
