// Transformer-side kernels of CLIP's ViT and text towers that are not GEMMs: LayerNorm,
// the attention core, token-embedding gather, patch im2col, class-token assembly and the
// EOT-row gather.  All fp32; the projections (QKV / out / MLP) go through
// dbmm_gemm_bias_act with bias / QuickGELU / residual epilogues.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------
// LayerNorm: one wave per row, float4 lanes, two-pass statistics in fp32 (biased variance),
// exactly the reference's LayerNorm subclass (fp32 compute) -- HBM-bound: 1 read + 1 write
// per element, re-reads hit L1/L2.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, long long ldx,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y,
                                                        long long ldy, int rows, int E4, float eps,
                                                        float* __restrict__ y_absmax) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const f32x4* xr = (const f32x4*)(x + (long long)row * ldx);
    const float invE = 1.f / (float)(E4 * 4);
    float s = 0.f;
    for (int i = lane; i < E4; i += 64) { const f32x4 v = xr[i]; s += (v[0] + v[1]) + (v[2] + v[3]); }
    const float mean = wave_sum(s) * invE;
    float q = 0.f;
    for (int i = lane; i < E4; i += 64) {
        const f32x4 v = xr[i] - mean;
        q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    const float rstd = rsqrtf(wave_sum(q) * invE + eps);
    f32x4* yr = (f32x4*)(y + (long long)row * ldy);
    float amax = 0.f;
    for (int i = lane; i < E4; i += 64) {
        const f32x4 v = (xr[i] - mean) * rstd * ((const f32x4*)gamma)[i] + ((const f32x4*)beta)[i];
        yr[i] = v;
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    if (y_absmax) {          // scale source of the fp16-pair GEMM that consumes y
        amax = wave_max(amax);
        if (lane == 0 && amax > *(volatile const float*)y_absmax) atomicMax((unsigned*)y_absmax, __float_as_uint(amax));
    }
}

// ---------------------------------------------------------------------------------------
// Attention core, head_dim 64.  One wave per (64-query tile, head, image); lane = query row
// (q and the output accumulator live in registers), K/V tiles of 64 keys are staged in LDS
// and read as wave-uniform (broadcast) float4s.  Online softmax with one rescale per group
// of 8 keys.  0.5 % (ViT-B/32) / 8.6 % (ViT-L/14) of the tower's FLOPs; VALU fp32.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void mha_core_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                      int L, int E, int causal) {
    __shared__ __attribute__((aligned(16))) float Ks[64 * 64];
    __shared__ __attribute__((aligned(16))) float Vs[64 * 64];
    const int lane = threadIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int qi = blockIdx.x * 64 + lane;
    const long long ld = 3LL * E;
    const float* base = qkv + (long long)b * L * ld + h * 64;
    float q[64], o[64];
    {
        const int qr = qi < L ? qi : L - 1;
        const f32x4* qp = (const f32x4*)(base + (long long)qr * ld);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const f32x4 v = qp[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) q[4 * i + j] = v[j] * 0.125f;
        }
    }
#pragma unroll
    for (int d = 0; d < 64; ++d) o[d] = 0.f;
    float m = -INFINITY, l = 0.f;
    const int kend = causal ? min(L, blockIdx.x * 64 + 64) : L;   // keys beyond the tile's last query are masked
    for (int k0 = 0; k0 < kend; k0 += 64) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 16; ++i) {  // 64 keys x 16 float4, coalesced: 16 lanes per key row
            const int idx = i * 64 + lane, kr = idx >> 4, c4 = idx & 15;
            const int kj = k0 + kr;
            f32x4 kvv = {0.f, 0.f, 0.f, 0.f}, vvv = {0.f, 0.f, 0.f, 0.f};
            if (kj < L) {
                kvv = *(const f32x4*)(base + (long long)kj * ld + E + c4 * 4);
                vvv = *(const f32x4*)(base + (long long)kj * ld + 2 * E + c4 * 4);
            }
            *(f32x4*)(Ks + kr * 64 + c4 * 4) = kvv;
            *(f32x4*)(Vs + kr * 64 + c4 * 4) = vvv;
        }
        __syncthreads();
        const int nkeys = min(64, kend - k0);
        for (int g = 0; g < nkeys; g += 8) {
            float s[8];
            float gm = -INFINITY;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int kr = g + u;               // < 64 always (tile rows beyond L hold zeros)
                const f32x4* kp = (const f32x4*)(Ks + kr * 64);
                float a0 = 0.f, a1 = 0.f;
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const f32x4 x0 = kp[i], x1 = kp[i + 1];
                    a0 = fmaf(q[4 * i + 0], x0[0], a0); a0 = fmaf(q[4 * i + 1], x0[1], a0);
                    a0 = fmaf(q[4 * i + 2], x0[2], a0); a0 = fmaf(q[4 * i + 3], x0[3], a0);
                    a1 = fmaf(q[4 * i + 4], x1[0], a1); a1 = fmaf(q[4 * i + 5], x1[1], a1);
                    a1 = fmaf(q[4 * i + 6], x1[2], a1); a1 = fmaf(q[4 * i + 7], x1[3], a1);
                }
                const int kj = k0 + kr;
                const bool ok = (kj < L) && (!causal || kj <= qi);
                s[u] = ok ? (a0 + a1) : -INFINITY;
                gm = fmaxf(gm, s[u]);
            }
            const float mn = fmaxf(m, gm);
            if (mn == -INFINITY) continue;           // whole group masked for this lane so far
            const float sc = expf(m - mn);
            l *= sc;
#pragma unroll
            for (int d = 0; d < 64; ++d) o[d] *= sc;
            m = mn;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float pj = expf(s[u] - mn);
                l += pj;
                const f32x4* vp = (const f32x4*)(Vs + (g + u) * 64);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const f32x4 vv = vp[i];
                    o[4 * i + 0] = fmaf(pj, vv[0], o[4 * i + 0]); o[4 * i + 1] = fmaf(pj, vv[1], o[4 * i + 1]);
                    o[4 * i + 2] = fmaf(pj, vv[2], o[4 * i + 2]); o[4 * i + 3] = fmaf(pj, vv[3], o[4 * i + 3]);
                }
            }
        }
    }
    if (qi < L) {
        const float inv = 1.f / l;
        f32x4* op = (f32x4*)(out + ((long long)b * L + qi) * E + h * 64);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            f32x4 v = {o[4 * i] * inv, o[4 * i + 1] * inv, o[4 * i + 2] * inv, o[4 * i + 3] * inv};
            op[i] = v;
        }
    }
}

__global__ __launch_bounds__(256) void embed_gather_kernel(const int32_t* __restrict__ tokens,
                                                           const float* __restrict__ table,
                                                           const float* __restrict__ pos, float* __restrict__ out,
                                                           int L, int W4, int vocab, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % W4);
        const long long row = i / W4;
        const int l = (int)(row % L);
        int t = tokens[row];
        t = t < 0 ? 0 : (t >= vocab ? vocab - 1 : t);   // never read outside the table
        ((f32x4*)out)[i] = ((const f32x4*)table)[(long long)t * W4 + c] + ((const f32x4*)pos)[(long long)l * W4 + c];
    }
}

__global__ __launch_bounds__(256) void im2col_patch_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                           int R, int P, int g, long long total) {
    const int K = 3 * P * P;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int k = (int)(i % K);
        const long long m = i / K;
        const int gx = (int)(m % g), gy = (int)((m / g) % g);
        const long long b = m / ((long long)g * g);
        const int kw = k % P, kh = (k / P) % P, c = k / (P * P);
        out[i] = x[((b * 3 + c) * R + (long long)gy * P + kh) * R + (long long)gx * P + kw];
    }
}

__global__ __launch_bounds__(256) void vit_tokens_kernel(const float* __restrict__ patches,
                                                         const float* __restrict__ cls,
                                                         const float* __restrict__ pos, float* __restrict__ out,
                                                         int L, int W4, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % W4);
        const long long row = i / W4;
        const int l = (int)(row % L);
        const long long b = row / L;
        const f32x4 v = (l == 0) ? ((const f32x4*)cls)[c]
                                 : ((const f32x4*)patches)[(b * (L - 1) + (l - 1)) * W4 + c];
        ((f32x4*)out)[i] = v + ((const f32x4*)pos)[(long long)l * W4 + c];
    }
}

__global__ __launch_bounds__(64) void gather_eot_kernel(const int32_t* __restrict__ tokens,
                                                        const float* __restrict__ x, float* __restrict__ out,
                                                        int L, int W) {
    const int n = blockIdx.x, lane = threadIdx.x;
    // first index of the maximum token id (torch.argmax tie rule)
    int best = INT32_MIN, bi = 0;
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

inline unsigned grid_for(long long total) {
    const long long blocks = (total + 255) / 256;
    return (unsigned)(blocks < 16384 ? (blocks > 0 ? blocks : 1) : 16384);
}

}  // namespace

extern "C" int dbmm_layernorm(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y,
                              int64_t ldy, int64_t rows, int64_t E, float eps, float* y_absmax, void* stream) {
    if (!x || !gamma || !beta || !y) return DBMM_E_ARG;
    if (rows <= 0 || E <= 0 || (E & 3) || rows > INT32_MAX) return DBMM_E_SHAPE;
    if ((ldx & 3) || (ldy & 3) || !dbmm_aligned16(x) || !dbmm_aligned16(y) || !dbmm_aligned16(gamma) ||
        !dbmm_aligned16(beta))
        return DBMM_E_ALIGN;
    hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x,
                       (long long)ldx, gamma, beta, y, (long long)ldy, (int)rows, (int)(E / 4), eps, y_absmax);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

extern "C" int dbmm_mha_core(const float* qkv, float* out, int64_t B, int64_t L, int64_t E, int64_t heads,
                             int causal, void* stream) {
    if (!qkv || !out) return DBMM_E_ARG;
    if (B <= 0 || L <= 0 || heads <= 0 || E != heads * 64 || B > 65535 || heads > 65535) return DBMM_E_SHAPE;
    if (!dbmm_aligned16(qkv) || !dbmm_aligned16(out)) return DBMM_E_ALIGN;
    hipLaunchKernelGGL(mha_core_kernel, dim3((unsigned)((L + 63) / 64), (unsigned)heads, (unsigned)B), dim3(64), 0,
                       (hipStream_t)stream, qkv, out, (int)L, (int)E, causal);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

extern "C" int dbmm_embed_gather(const int32_t* tokens, const float* table, const float* pos, float* out,
                                 int64_t n, int64_t L, int64_t W, int64_t vocab, void* stream) {
    if (!tokens || !table || !pos || !out) return DBMM_E_ARG;
    if (n <= 0 || L <= 0 || W <= 0 || (W & 3) || vocab <= 0) return DBMM_E_SHAPE;
    if (!dbmm_aligned16(table) || !dbmm_aligned16(pos) || !dbmm_aligned16(out)) return DBMM_E_ALIGN;
    const long long total = (long long)n * L * (W / 4);
    hipLaunchKernelGGL(embed_gather_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, tokens, table,
                       pos, out, (int)L, (int)(W / 4), (int)vocab, total);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

extern "C" int dbmm_im2col_patch(const float* x_nchw, float* out, int64_t B, int64_t R, int64_t P, void* stream) {
    if (!x_nchw || !out) return DBMM_E_ARG;
    if (B <= 0 || R <= 0 || P <= 0 || R % P) return DBMM_E_SHAPE;
    const int64_t g = R / P;
    const long long total = (long long)B * g * g * 3 * P * P;
    hipLaunchKernelGGL(im2col_patch_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x_nchw, out,
                       (int)R, (int)P, (int)g, total);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

extern "C" int dbmm_vit_tokens(const float* patches, const float* cls, const float* pos, float* out, int64_t B,
                               int64_t L, int64_t W, void* stream) {
    if (!patches || !cls || !pos || !out) return DBMM_E_ARG;
    if (B <= 0 || L <= 1 || W <= 0 || (W & 3)) return DBMM_E_SHAPE;
    if (!dbmm_aligned16(patches) || !dbmm_aligned16(cls) || !dbmm_aligned16(pos) || !dbmm_aligned16(out))
        return DBMM_E_ALIGN;
    const long long total = (long long)B * L * (W / 4);
    hipLaunchKernelGGL(vit_tokens_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, patches, cls, pos,
                       out, (int)L, (int)(W / 4), total);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

extern "C" int dbmm_gather_eot(const int32_t* tokens, const float* x, float* out, int64_t n, int64_t L, int64_t W,
                               void* stream) {
    if (!tokens || !x || !out) return DBMM_E_ARG;
    if (n <= 0 || L <= 0 || W <= 0) return DBMM_E_SHAPE;
    hipLaunchKernelGGL(gather_eot_kernel, dim3((unsigned)n), dim3(64), 0, (hipStream_t)stream, tokens, x, out, (int)L,
                       (int)W);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}
