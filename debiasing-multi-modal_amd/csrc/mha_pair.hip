// Attention core of the fp32 parity mode on the 16-bit matrix cores: softmax(Q K^T / 8) V, head_dim 64, with every
// product as fp16 (hi, lo) PAIRS -- q, k, v scaled by the exact power of two 2^s taken from the qkv GEMM's device
// maximum, p in [0, 1] scaled by 2^13 -- and three partial products per fp32 product (hi*lo, lo*hi, hi*hi; the dropped
// lo*lo is 2^-22 relative), fp32 accumulation: the same error model as the fp16-pair GEMMs around it.
// (nn.MultiheadAttention as called at clip/model.py:185-187; the fp32-input-MFMA kernel in transformer_ops.hip computes
// the same thing at 1/16 of the matrix rate and was 40 % of the ViT-L/14@336 step.)
//
// Structure = mha_f16_kernel (f16_ops.hip): a workgroup is 128 queries of one (image, head), a wave 32 queries; S^T =
// K Q^T puts a query in an accumulator column = a lane, so the row maximum / sum are register reductions plus one
// exchange with lane ^ 32 and the O^T rescale is lane-local; V is staged TRANSPOSED with the keys of every 16-group
// permuted (8a + 4h + i -> 8h + 4a + i) so that the probabilities in their accumulator layout are the B operand of
// O^T += V^T P^T.  Here K, V^T and P carry two planes each.
// Bound: MFMA (2500 TF / 3 products = 833 TF fp32-equivalent).
#include <stdlib.h>
#include "common.h"

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int scale_exp(float amax) {
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
__device__ __forceinline__ int swz64(int row) { return (row >> 1) & 7; }
__device__ __forceinline__ f32x16 mfma16(const u32x4& a, const u32x4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

constexpr int O_ROW = 68;                                        // O staging pitch in floats (272 B)
constexpr int PL = 64 * 64;                                      // halves per K / V plane

// the 8 halves of a V^T fragment = two hardware-transposed 4-key x 16-d blocks 8 keys apart (cdna_hip_programming.md T10)
__device__ __forceinline__ u32x4 vt_frag(const unsigned char* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef short s16x4 __attribute__((__vector_size__(4 * sizeof(short))));
    const s16x4 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    const s16x4 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 8 * 128));
    const u32x2 ua = __builtin_bit_cast(u32x2, a), ub = __builtin_bit_cast(u32x2, b);
    return (u32x4){ua[0], ua[1], ub[0], ub[1]};
#else
    (void)p; return (u32x4){0u, 0u, 0u, 0u};
#endif
}

// NW = waves per workgroup: 4 (128 queries; the default) or 2 (64 queries: sequences of at most 64 tokens -- ViT-B/32's 50 -- left two of
// four waves multiplying clamped queries; with two waves three workgroups fit a CU).
template <int NW>
__global__ __launch_bounds__(64 * NW, 2) void mha_pair_kernel(const float* __restrict__ qkv, const float* __restrict__ qkv_absmax,
                                                                             float* __restrict__ out, int L, int E, int causal) {
    constexpr int QB = 32 * NW, KPI = 4 * NW, NI = 64 / KPI;      // queries per workgroup; key rows per staging pass, passes per 64-key tile
    __shared__ __attribute__((aligned(16))) u16 Ks[2 * PL];      // [hi | lo][key][64 d]
    __shared__ __attribute__((aligned(16))) u16 Vt[2 * PL];      // [hi | lo][key][64 d]
    __shared__ __attribute__((aligned(16))) float Os[NW * 32 * O_ROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 31, fh = lane >> 5;
    const int qb = blockIdx.x, head = blockIdx.y, b = blockIdx.z;
    const long long row0 = (long long)b * L, ld = 3LL * E;
    const int q_idx = qb * QB + wave * 32 + fr, q_wave0 = qb * QB + wave * 32;     // this lane's query, the wave's first
    const int q_ld = q_idx < L ? q_idx : L - 1;
    const int s_x = scale_exp(*qkv_absmax);
    const float x_sc = pow2f(s_x);
    // scores: S~ = sum (q 2^s)(k 2^s) -> S / 8 * log2(e) = S~ * 2^(-2s) * 0.125 * log2(e)
    const float score_scale = pow2f(-2 * s_x) * (0.125f * 1.4426950408889634f);

    // Q fragments (B operand of S^T): lane (query, k half fh) holds d = 16 s + 8 fh .. + 7 as (hi, lo)
    u32x4 qh[4], ql[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const float* src = qkv + (row0 + q_ld) * ld + head * 64 + 16 * s + 8 * fh;
        const f32x4 x0 = *(const f32x4*)src, x1 = *(const f32x4*)(src + 4);
        unsigned h[4], l[4];
        split2h_pair(x0[0], x0[1], x_sc, h[0], l[0]); split2h_pair(x0[2], x0[3], x_sc, h[1], l[1]);
        split2h_pair(x1[0], x1[1], x_sc, h[2], l[2]); split2h_pair(x1[2], x1[3], x_sc, h[3], l[3]);
        qh[s] = (u32x4){h[0], h[1], h[2], h[3]}; ql[s] = (u32x4){l[0], l[1], l[2], l[3]};
    }
    f32x16 o_acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) o_acc[j][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    // staging: thread loads 16 B (4 d) of key rows (tid >> 4) + KPI i for K and V
    const int lc = tid & 15, lk = tid >> 4;
    f32x4 k_r[NI], v_r[NI];
    const int q_hi = qb * QB + QB - 1 < L - 1 ? qb * QB + QB - 1 : L - 1;
    const int n_keys = causal ? q_hi + 1 : L;
    const int T = (n_keys + 63) / 64;
    auto load_tile = [&](int t) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            int key = t * 64 + lk + KPI * i;
            key = key < L ? key : L - 1;
            const float* base = qkv + (row0 + key) * ld + head * 64 + lc * 4;
            k_r[i] = *(const f32x4*)(base + E);
            v_r[i] = *(const f32x4*)(base + 2 * E);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int key = lk + KPI * i;
            unsigned h[2], l[2];
            split2h_pair(k_r[i][0], k_r[i][1], x_sc, h[0], l[0]); split2h_pair(k_r[i][2], k_r[i][3], x_sc, h[1], l[1]);
            const int koff = key * 64 + (((lc >> 1) ^ swz64(key)) << 3) + ((lc & 1) << 2);
            *(u32x2*)(Ks + koff) = (u32x2){h[0], h[1]};
            *(u32x2*)(Ks + PL + koff) = (u32x2){l[0], l[1]};
            // V stays ROW-major ([key][64 d] per plane, 8-B stores like K): the V^T fragments come from the hardware transpose
            // read (vt_frag).  Chunk XOR 4 on rows 2, 3 (mod 4) keeps the four rows of a transposed block on distinct bank
            // groups.  (Staged transposed by hand this was 32 two-byte LDS stores per thread and tile.)
            split2h_pair(v_r[i][0], v_r[i][1], x_sc, h[0], l[0]); split2h_pair(v_r[i][2], v_r[i][3], x_sc, h[1], l[1]);
            const int voff = key * 64 + (((lc >> 1) ^ (((key >> 1) & 1) << 2)) << 3) + ((lc & 1) << 2);
            *(u32x2*)(Vt + voff) = (u32x2){h[0], h[1]};
            *(u32x2*)(Vt + PL + voff) = (u32x2){l[0], l[1]};
        }
    };
    // ds_read_b64_tr_b16 addressing as in mha_f16_kernel (f16_ops.hip): this lane's piece of a transposed 4-key x 16-d block
    // for d block j = 0 / 1, without the (kt, u, block) key offset
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
        // ---- S^T[key][query], three partial products, smallest first -------------------------------------------------------
        f32x16 s_acc[2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s_acc[kt][r] = 0.f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int row = kt * 32 + fr, off = row * 64 + (((2 * s + fh) ^ swz64(row)) << 3);
                const u32x4 kh = *(const u32x4*)(Ks + off), kl = *(const u32x4*)(Ks + PL + off);
                s_acc[kt] = mfma16(kh, ql[s], s_acc[kt]);
                s_acc[kt] = mfma16(kl, qh[s], s_acc[kt]);
                s_acc[kt] = mfma16(kh, qh[s], s_acc[kt]);
            }
        }
        // ---- online softmax (a lane holds keys (r&3) + 8 (r>>2) + 4 fh of each half for ITS query) ----------------------------
        // (masks only on tiles that can hold a masked key -- a wave-uniform branch --, the score scale folded into the exponent's
        //  FMA, the accumulator rescale skipped while no lane's maximum moved: the softmax, not the MFMAs, bounds this kernel)
        float mx = -INFINITY;
        if ((t + 1) * 64 > L || (causal && (t + 1) * 64 - 1 > q_wave0)) {
#pragma unroll
            for (int kt = 0; kt < 2; ++kt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = t * 64 + kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    if (key >= L || (causal && key > q_idx)) s_acc[kt][r] = -INFINITY;
                }
        }
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s_acc[kt][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64)) * score_scale;         // (scale > 0: the maximum commutes with it)
        const float m_new = fmaxf(m_run, mx);
        const float m_use = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = exp2f(m_run - m_use);
        float psum = 0.f;
        u32x4 ph[2][2], pl[2][2];
#pragma unroll
        for (int kt = 0; kt < 2; ++kt) {
            float pv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { pv[r] = exp2f(fmaf(s_acc[kt][r], score_scale, -m_use)); psum += pv[r]; }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                unsigned h[4], l[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) split2h_pair(pv[8 * u + 2 * e], pv[8 * u + 2 * e + 1], 8192.f, h[e], l[e]);
                ph[kt][u] = (u32x4){h[0], h[1], h[2], h[3]}; pl[kt][u] = (u32x4){l[0], l[1], l[2], l[3]};
            }
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0ull) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) o_acc[j][r] *= alpha;
        }
        // ---- O^T[d][query] += V^T[d][keys] P^T[keys][query] -----------------------------------------------------------------
#pragma unroll
        for (int kt = 0; kt < 2; ++kt)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const unsigned char* vp = (const unsigned char*)Vt + vtr[j] + (kt * 32 + 16 * u) * 128;
                    const u32x4 vh = vt_frag(vp), vl = vt_frag(vp + PL * 2);
                    o_acc[j] = mfma16(vh, pl[kt][u], o_acc[j]);
                    o_acc[j] = mfma16(vl, ph[kt][u], o_acc[j]);
                    o_acc[j] = mfma16(vh, ph[kt][u], o_acc[j]);
                }
        __syncthreads();
        if (t + 1 < T) store_tile();
        __syncthreads();
    }
    // ---- normalise (undo 2^s of v and 2^13 of p); O^T -> rows through LDS; 16-B stores -------------------------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = (l_tot > 0.f ? 1.f / l_tot : 0.f) * pow2f(-s_x - 13);
    float* Ow = Os + wave * (32 * O_ROW);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = j * 32 + 8 * g + 4 * fh;
            *(f32x4*)(Ow + fr * O_ROW + d) = (f32x4){o_acc[j][4 * g] * inv, o_acc[j][4 * g + 1] * inv, o_acc[j][4 * g + 2] * inv,
                                                      o_acc[j][4 * g + 3] * inv};
        }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = (lane >> 4) + 4 * i, q = qb * QB + wave * 32 + row;
        if (q < L) *(f32x4*)(out + (row0 + q) * (long long)E + head * 64 + (lane & 15) * 4) = *(const f32x4*)(Ow + row * O_ROW + (lane & 15) * 4);
    }
}

}  // namespace

// see include/dbmm.h
extern "C" int dbmm_mha_core_x2(const float* qkv, const float* qkv_absmax, float* out, int64_t B, int64_t L, int64_t E,
                                int64_t heads, int causal, void* stream) {
    if (!qkv || !qkv_absmax || !out) return DBMM_E_ARG;
    if (B <= 0 || L <= 0 || heads <= 0 || E != heads * 64 || B > 65535 || heads > 65535) return DBMM_E_SHAPE;
    if (!dbmm_aligned16(qkv) || !dbmm_aligned16(out)) return DBMM_E_ALIGN;
    if (L <= 64 && dbmm_opt(OPT_MHA_SHORT)) {
        hipLaunchKernelGGL(mha_pair_kernel<2>, dim3(1, (unsigned)heads, (unsigned)B), dim3(128), 0, (hipStream_t)stream, qkv, qkv_absmax, out, (int)L,
                           (int)E, causal ? 1 : 0);
    } else {
        const dim3 grid((unsigned)((L + 127) / 128), (unsigned)heads, (unsigned)B);
        hipLaunchKernelGGL(mha_pair_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, qkv, qkv_absmax, out, (int)L, (int)E, causal ? 1 : 0);
    }
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}
