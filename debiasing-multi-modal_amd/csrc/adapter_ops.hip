// Debiasing-adapter step kernels (final_main.py:53-174, 426-496; demo/util.py:118-136):
// BatchNorm1d batch statistics / apply / backward, fused row-L2-norm + image x text logits +
// cross-entropy forward and backward, column reductions for bias gradients, a multi-tensor
// SGD-momentum update and the per-group accuracy counters.  These are HBM/launch-bound
// (arithmetic intensity ~1 FLOP/B); the four GEMM-shaped products go through
// dbmm_gemm_bias_act (fp32 MFMA).
#include "common.h"

// adapter_step.hip: the products of the step for the reference's shapes (H = 128, D % 128 == 0)
bool dbmm_adapter_fast_shape(int64_t B, int64_t D, int64_t H);
size_t dbmm_adapter_bwd_fast_floats(int64_t B, int64_t D);
int dbmm_adapter_fwd_fast(const float* x, const float* w1, const float* b1, const float* gamma, const float* beta, float* running_mean,
                          float* running_var, int64_t* nbt, const float* w2, const float* b2, float* h, float* mean, float* invstd, float* r,
                          float* z, int64_t B, int64_t D, int train, float eps, float momentum, hipStream_t s);
int dbmm_adapter_bwd_fast(const float* x, const float* dz, const float* h, const float* mean, const float* invstd, const float* r, const float* gamma,
                          const float* beta, const float* w2, float* dw1, float* db1, float* dgamma, float* dbeta, float* dw2, float* db2,
                          float* dh, float* scratch, int64_t B, int64_t D, hipStream_t s, const float** dw1part = nullptr,
                          const float** db1part = nullptr, int* nsplit = nullptr, const float* loss_rows = nullptr, float* loss_mean = nullptr);

namespace {

// ---- BatchNorm1d(H) statistics over the batch: 8 columns x 32 row-lanes per block ---------
__global__ __launch_bounds__(256) void bn1d_stats_kernel(const float* __restrict__ h, int B, int H, float eps,
                                                         float momentum, float* __restrict__ mean_o,
                                                         float* __restrict__ invstd_o, float* __restrict__ rmean,
                                                         float* __restrict__ rvar, long long* __restrict__ nbt) {
    __shared__ float red[32][9];
    const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int j = blockIdx.x * 8 + cl;
    const bool ok = j < H;
    float s = 0.f;
    if (ok) for (int b = rl; b < B; b += 32) s += h[(long long)b * H + j];
    red[rl][cl] = s;
    __syncthreads();
    float mean = 0.f;
    for (int r = 0; r < 32; ++r) mean += red[r][cl];
    mean /= (float)B;
    __syncthreads();
    float q = 0.f;
    if (ok) for (int b = rl; b < B; b += 32) { const float d = h[(long long)b * H + j] - mean; q = fmaf(d, d, q); }
    red[rl][cl] = q;
    __syncthreads();
    if (rl == 0 && ok) {
        float var = 0.f;
        for (int r = 0; r < 32; ++r) var += red[r][cl];
        var /= (float)B;
        mean_o[j] = mean;
        invstd_o[j] = rsqrtf(var + eps);
        if (rmean) rmean[j] = (1.f - momentum) * rmean[j] + momentum * mean;
        if (rvar) rvar[j] = (1.f - momentum) * rvar[j] + momentum * (var * (float)B / (float)(B - 1));
    }
    if (nbt && blockIdx.x == 0 && threadIdx.x == 0) *nbt += 1;
}

__global__ __launch_bounds__(256) void bn1d_relu_kernel(const float* __restrict__ h, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ r,
                                                        int H4, int var_mode, float eps, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % H4);
        f32x4 is = ((const f32x4*)invstd)[c];
        if (var_mode) {
#pragma unroll
            for (int k = 0; k < 4; ++k) is[k] = rsqrtf(is[k] + eps);
        }
        f32x4 v = (((const f32x4*)h)[i] - ((const f32x4*)mean)[c]) * is * ((const f32x4*)gamma)[c] +
                  ((const f32x4*)beta)[c];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = fmaxf(v[k], 0.f);
        ((f32x4*)r)[i] = v;
    }
}

// dhn = dr * (hn > 0);  dbeta_j = sum_b dhn;  dgamma_j = sum_b dhn * xhat
__global__ __launch_bounds__(256) void bn1d_bwd_reduce_kernel(const float* __restrict__ dr,
                                                              const float* __restrict__ h,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ invstd,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, int B, int H,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float red[2][32][9];
    const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int j = blockIdx.x * 8 + cl;
    const bool ok = j < H;
    float sb = 0.f, sg = 0.f;
    if (ok) {
        const float mu = mean[j], is = invstd[j], ga = gamma[j], be = beta[j];
        for (int b = rl; b < B; b += 32) {
            const float xh = (h[(long long)b * H + j] - mu) * is;
            const float d = (fmaf(ga, xh, be) > 0.f) ? dr[(long long)b * H + j] : 0.f;
            sb += d;
            sg = fmaf(d, xh, sg);
        }
    }
    red[0][rl][cl] = sb; red[1][rl][cl] = sg;
    __syncthreads();
    if (rl == 0 && ok) {
        float a = 0.f, g = 0.f;
        for (int r = 0; r < 32; ++r) { a += red[0][r][cl]; g += red[1][r][cl]; }
        dbeta[j] = a; dgamma[j] = g;
    }
}

// dh = gamma * invstd * (dhn - dbeta/B - xhat * dgamma/B)      (train-mode BN backward)
__global__ __launch_bounds__(256) void bn1d_bwd_apply_kernel(const float* __restrict__ dr,
                                                             const float* __restrict__ h,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ invstd,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ dgamma,
                                                             const float* __restrict__ dbeta, float* __restrict__ dh,
                                                             int H, float invB, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i % H);
        const float is = invstd[j], ga = gamma[j];
        const float xh = (h[i] - mean[j]) * is;
        const float d = (fmaf(ga, xh, beta[j]) > 0.f) ? dr[i] : 0.f;
        dh[i] = ga * is * (d - dbeta[j] * invB - xh * dgamma[j] * invB);
    }
}

// out[j] = sum_b x[b][j]: 16 columns (4 float4 lanes) x 64 row-lanes per block, fixed summation
// order (row-lane partial sums, then an LDS tree) => deterministic.  N % 4 == 0.
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int B, int N,
                                                     float* __restrict__ out) {
    __shared__ f32x4 red[64][4];
    const int cg = threadIdx.x & 3, rl = threadIdx.x >> 2;
    const int j = blockIdx.x * 16 + cg * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    if (j < N) for (int b = rl; b < B; b += 64) s += *(const f32x4*)(x + (long long)b * N + j);
    red[rl][cg] = s;
    __syncthreads();
    for (int o = 32; o > 0; o >>= 1) {
        if (rl < o) red[rl][cg] += red[rl + o][cg];
        __syncthreads();
    }
    if (rl == 0 && j < N) *(f32x4*)(out + j) = red[0][cg];
}

// y[r][:] = x[r][:] / ||x[r][:]||: one wave per row, the row read twice (the second pass hits L2), no epsilon (clip/model.py:362-363)
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int D4) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    const f32x4* xr = (const f32x4*)x + (long long)row * D4;
    f32x4* yr = (f32x4*)y + (long long)row * D4;
    float s = 0.f;
    for (int i = lane; i < D4; i += 64) { const f32x4 v = xr[i]; s = fmaf(v.x, v.x, s); s = fmaf(v.y, v.y, s); s = fmaf(v.z, v.z, s); s = fmaf(v.w, v.w, s); }
    const float inv = 1.f / sqrtf(wave_sum(s));
    for (int i = lane; i < D4; i += 64) yr[i] = xr[i] * inv;
}

// tn[c][:] = text[:, c] / ||text[:, c]||
__global__ __launch_bounds__(256) void text_colnorm_kernel(const float* __restrict__ text, float* __restrict__ tn,
                                                           int D, int C) {
    __shared__ float red[4];
    const int c = blockIdx.x;
    float s = 0.f;
    for (int i = threadIdx.x; i < D; i += 256) { const float v = text[(long long)i * C + c]; s = fmaf(v, v, s); }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float inv = 1.f / sqrtf((red[0] + red[1]) + (red[2] + red[3]));
    for (int i = threadIdx.x; i < D; i += 256) tn[(long long)c * D + i] = text[(long long)i * C + c] * inv;
}

// ---- fused row L2-norm + logits + CE: one wave per row -------------------------------------
// forward of one row by one wave (every lane ends up with the row's logits lg[] and 1 / ||z||); lane 0 writes the outputs
template <int CMAX>
__device__ __forceinline__ void ce_fwd_row(const float* __restrict__ z, const float* __restrict__ z_old, float w_old, const float* __restrict__ tn,
                                           const long long* __restrict__ labels, float invT, float* __restrict__ logits,
                                           float* __restrict__ loss_rows, long long* __restrict__ pred, float* __restrict__ inv_norm, int row,
                                           int lane, int D4, int C, float (&lg)[CMAX], float& inv) {
    const f32x4* zr = (const f32x4*)z + (long long)row * D4;
    const f32x4* zo = z_old ? (const f32x4*)z_old + (long long)row * D4 : nullptr;
    float ss = 0.f, sso = 0.f, dot[CMAX], doto[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) { dot[c] = 0.f; doto[c] = 0.f; }
    for (int i = lane; i < D4; i += 64) {
        const f32x4 v = zr[i];
        ss += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        f32x4 vo = {0.f, 0.f, 0.f, 0.f};
        if (zo) { vo = zo[i]; sso += (vo[0] * vo[0] + vo[1] * vo[1]) + (vo[2] * vo[2] + vo[3] * vo[3]); }
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            if (c < C) {
                const f32x4 t = ((const f32x4*)tn)[(long long)c * D4 + i];
                dot[c] += (v[0] * t[0] + v[1] * t[1]) + (v[2] * t[2] + v[3] * t[3]);
                if (zo) doto[c] += (vo[0] * t[0] + vo[1] * t[1]) + (vo[2] * t[2] + vo[3] * t[3]);
            }
        }
    }
    ss = wave_sum(ss);
    inv = 1.f / sqrtf(ss);
    float invo = 0.f;
    if (zo) invo = 1.f / sqrtf(wave_sum(sso));
    float mx = -INFINITY;
    int am = 0;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) {
        lg[c] = -INFINITY;
        if (c < C) {
            const float d = wave_sum(dot[c]) * inv;
            float f = d;
            if (zo) f = w_old * (wave_sum(doto[c]) * invo) + (1.f - w_old) * d;
            lg[c] = f * invT;
            if (lg[c] > mx) { mx = lg[c]; am = c; }
        }
    }
    if (lane == 0) {
        if (inv_norm) inv_norm[row] = inv;
        if (logits) for (int c = 0; c < C; ++c) logits[(long long)row * C + c] = lg[c];
        if (pred) pred[row] = am;
        if (loss_rows && labels) {
            float se = 0.f;
#pragma unroll
            for (int c = 0; c < CMAX; ++c) if (c < C) se += expf(lg[c] - mx);
            const int y = (int)labels[row];
            float ly = 0.f;
#pragma unroll
            for (int c = 0; c < CMAX; ++c) if (c == y) ly = lg[c];
            loss_rows[row] = (mx + logf(se)) - ly;
        }
    }
}

template <int CMAX>
__global__ __launch_bounds__(256) void l2norm_sim_ce_fwd_kernel(
    const float* __restrict__ z, const float* __restrict__ z_old, float w_old, const float* __restrict__ tn,
    const long long* __restrict__ labels, float invT, float* __restrict__ logits, float* __restrict__ loss_rows,
    long long* __restrict__ pred, float* __restrict__ inv_norm, int B, int D4, int C) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    float lg[CMAX], inv;
    ce_fwd_row<CMAX>(z, z_old, w_old, tn, labels, invT, logits, loss_rows, pred, inv_norm, row, lane, D4, C, lg, inv);
}

__global__ __launch_bounds__(256) void mean_reduce_kernel(const float* __restrict__ x, int n, float* __restrict__ out) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += x[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *out = ((red[0] + red[1]) + (red[2] + red[3])) / (float)n;
}

// backward of one row by one wave: dz from the row's logits `lgv` (read back or still in registers), its label and 1 / ||z||
template <int CMAX>
__device__ __forceinline__ void ce_bwd_row(const float* __restrict__ z, float inv, float w_new, const float* __restrict__ tn, const float (&lgv)[CMAX],
                                           const long long* __restrict__ labels, const float* __restrict__ dlogits, float invT, float gscale,
                                           float* __restrict__ dz, int row, int lane, int D4, int C) {
    float dl[CMAX];
    if (dlogits) {   // upstream gradient given (autograd path): dl = dlogits / T * blend weight
#pragma unroll
        for (int c = 0; c < CMAX; ++c) dl[c] = (c < C) ? dlogits[(long long)row * C + c] * invT * w_new : 0.f;
    } else {
        float mx = -INFINITY;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) { dl[c] = (c < C) ? lgv[c] : -INFINITY; mx = fmaxf(mx, dl[c]); }
        float se = 0.f;
#pragma unroll
        for (int c = 0; c < CMAX; ++c) { dl[c] = (c < C) ? expf(dl[c] - mx) : 0.f; se += dl[c]; }
        const int y = (int)labels[row];
        const float k = gscale * invT * w_new;   // d loss / d (feat . tn[c]) incl. blend weight
#pragma unroll
        for (int c = 0; c < CMAX; ++c) dl[c] = (dl[c] / se - (c == y ? 1.f : 0.f)) * k;
    }
    const f32x4* zr = (const f32x4*)z + (long long)row * D4;
    float fd = 0.f;
    for (int i = lane; i < D4; i += 64) {
        f32x4 df = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < CMAX; ++c) if (c < C) df += dl[c] * ((const f32x4*)tn)[(long long)c * D4 + i];
        const f32x4 f = zr[i] * inv;
        fd += (f[0] * df[0] + f[1] * df[1]) + (f[2] * df[2] + f[3] * df[3]);
    }
    fd = wave_sum(fd);
    f32x4* dzr = (f32x4*)dz + (long long)row * D4;
    for (int i = lane; i < D4; i += 64) {
        f32x4 df = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < CMAX; ++c) if (c < C) df += dl[c] * ((const f32x4*)tn)[(long long)c * D4 + i];
        const f32x4 f = zr[i] * inv;
        dzr[i] = (df - f * fd) * inv;
    }
}

template <int CMAX>
__global__ __launch_bounds__(256) void l2norm_sim_ce_bwd_kernel(
    const float* __restrict__ z, const float* __restrict__ inv_norm, float w_new, const float* __restrict__ tn,
    const float* __restrict__ logits, const long long* __restrict__ labels, const float* __restrict__ dlogits,
    float invT, float gscale, float* __restrict__ dz, int B, int D4, int C) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    float lgv[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) lgv[c] = (logits && c < C) ? logits[(long long)row * C + c] : -INFINITY;
    ce_bwd_row<CMAX>(z, inv_norm[row], w_new, tn, lgv, labels, dlogits, invT, gscale, dz, row, lane, D4, C);
}

// the one-call step: forward and backward of a row in ONE launch (both are row-local; the logits stay in registers).  Same
// statements as the two kernels above, so the results are the same bits.
template <int CMAX>
__global__ __launch_bounds__(256) void l2norm_sim_ce_fwdbwd_kernel(
    const float* __restrict__ z, const float* __restrict__ z_old, float w_old, float w_new, const float* __restrict__ tn,
    const long long* __restrict__ labels, float invT, float gscale, float* __restrict__ logits, float* __restrict__ loss_rows,
    float* __restrict__ dz, int B, int D4, int C) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= B) return;
    float lg[CMAX], inv;
    ce_fwd_row<CMAX>(z, z_old, w_old, tn, labels, invT, logits, loss_rows, nullptr, nullptr, row, lane, D4, C, lg, inv);
    ce_bwd_row<CMAX>(z, inv, w_new, tn, lg, labels, nullptr, invT, gscale, dz, row, lane, D4, C);
}

// ---- multi-tensor SGD with momentum ---------------------------------------------------------
struct SgdArgs {
    float* p[16];
    const float* g[16];
    float* m[16];
    long long n[16];
    int ns[16];            // > 1: g[t] holds ns[t] partial gradients n[t] apart, summed here in order (grad_reduce_kernel's order)
};
__global__ __launch_bounds__(256) void sgd_kernel(const SgdArgs a, float lr, float mu, float wd, int first) {
    const int t = blockIdx.y;
    float* p = a.p[t]; const float* g = a.g[t]; float* m = a.m[t];
    const long long n = a.n[t];
    const int ns = a.ns[t];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float w = p[i];
        float gv = g[i];
        for (int sidx = 1; sidx < ns; ++sidx) gv += g[(long long)sidx * n + i];
        const float gi = fmaf(wd, w, gv);
        const float b = first ? gi : fmaf(mu, m[i], gi);
        m[i] = b;
        p[i] = w - lr * b;
    }
}

// ---- update_dict counters --------------------------------------------------------------------
__global__ __launch_bounds__(256) void group_count_kernel(const float* __restrict__ logits,
                                                          const long long* __restrict__ y,
                                                          const long long* __restrict__ g,
                                                          unsigned long long* __restrict__ counts, int B, int C, int G) {
    __shared__ unsigned int sc[64][2];
    if (threadIdx.x < 64) { sc[threadIdx.x][0] = 0; sc[threadIdx.x][1] = 0; }
    __syncthreads();
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) {
        int am = 0; float mx = logits[(long long)b * C];
        for (int c = 1; c < C; ++c) { const float v = logits[(long long)b * C + c]; if (v > mx) { mx = v; am = c; } }
        const int gi = (int)g[b];
        if (gi >= 0 && gi < G) {
            atomicAdd(&sc[gi][0], 1u);
            if ((long long)am == y[b]) atomicAdd(&sc[gi][1], 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x < G) {
        if (sc[threadIdx.x][0]) atomicAdd(&counts[threadIdx.x * 2], (unsigned long long)sc[threadIdx.x][0]);
        if (sc[threadIdx.x][1]) atomicAdd(&counts[threadIdx.x * 2 + 1], (unsigned long long)sc[threadIdx.x][1]);
    }
}

// sums[g] += sum of loss_rows over group g; single block, fixed order => deterministic
__global__ __launch_bounds__(256) void group_loss_sum_kernel(const float* __restrict__ loss,
                                                             const long long* __restrict__ g,
                                                             float* __restrict__ sums, int B, int G) {
    __shared__ float red[256];
    for (int gi = 0; gi < G; ++gi) {
        float s = 0.f;
        for (int b = threadIdx.x; b < B; b += 256) if ((int)g[b] == gi) s += loss[b];
        red[threadIdx.x] = s;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
        if (threadIdx.x == 0) sums[gi] += red[0];
        __syncthreads();
    }
}

// out[i][:] = table[idx[i]][:]  (float4 lanes; indices clamped into the table)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ table,
                                                          const long long* __restrict__ idx, float* __restrict__ out,
                                                          int D4, long long n_rows, long long total) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / D4;
        const int c = (int)(i - r * D4);
        long long src = idx[r];
        src = src < 0 ? 0 : (src >= n_rows ? n_rows - 1 : src);
        ((f32x4*)out)[i] = ((const f32x4*)table)[src * D4 + c];
    }
}

inline unsigned grid_for(long long total) {
    const long long blocks = (total + 255) / 256;
    return (unsigned)(blocks < 8192 ? (blocks > 0 ? blocks : 1) : 8192);
}

}  // namespace

extern "C" int dbmm_bn1d_stats(const float* h, int64_t B, int64_t H, float eps, float momentum, float* mean,
                               float* invstd, float* running_mean, float* running_var, int64_t* nbt, void* stream) {
    if (!h || !mean || !invstd) return DBMM_E_ARG;
    if (B < 2 || H <= 0 || B > INT32_MAX || H > INT32_MAX) return DBMM_E_SHAPE;  // torch raises for B == 1 too
    hipLaunchKernelGGL(bn1d_stats_kernel, dim3((unsigned)((H + 7) / 8)), dim3(256), 0, (hipStream_t)stream, h, (int)B,
                       (int)H, eps, momentum, mean, invstd, running_mean, running_var, (long long*)nbt);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

extern "C" int dbmm_bn1d_relu(const float* h, const float* mean, const float* invstd, const float* gamma,
                              const float* beta, float* r, int64_t B, int64_t H, int var_mode, float eps,
                              void* stream) {
    if (!h || !mean || !invstd || !gamma || !beta || !r) return DBMM_E_ARG;
    if (B <= 0 || H <= 0 || (H & 3)) return DBMM_E_SHAPE;
    if (!dbmm_aligned16(h) || !dbmm_aligned16(r) || !dbmm_aligned16(mean) || !dbmm_aligned16(invstd) ||
        !dbmm_aligned16(gamma) || !dbmm_aligned16(beta))
        return DBMM_E_ALIGN;
    const long long total = (long long)B * (H / 4);
    hipLaunchKernelGGL(bn1d_relu_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, h, mean, invstd,
                       gamma, beta, r, (int)(H / 4), var_mode, eps, total);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

extern "C" int dbmm_adapter_fwd(const float* x, const float* w1, const float* b1, const float* gamma,
                                const float* beta, float* running_mean, float* running_var, int64_t* nbt,
                                const float* w2, const float* b2, float* h, float* mean, float* invstd, float* r,
                                float* z, int64_t B, int64_t D, int64_t H, int train, float eps, float momentum,
                                void* stream) {
    if (!x || !w1 || !b1 || !gamma || !beta || !w2 || !b2 || !h || !r || !z) return DBMM_E_ARG;
    if (!running_mean || !running_var) return DBMM_E_ARG;
    if (train && (!mean || !invstd)) return DBMM_E_ARG;
    if (dbmm_adapter_fast_shape(B, D, H) && (!train || B >= 2) && dbmm_aligned16(x) && dbmm_aligned16(w1) && dbmm_aligned16(w2) && dbmm_aligned16(h) &&
        dbmm_aligned16(r) && dbmm_aligned16(z) && dbmm_aligned16(gamma) && dbmm_aligned16(beta) &&
        dbmm_aligned16(train ? mean : running_mean) && dbmm_aligned16(train ? invstd : running_var) && dbmm_opt(OPT_ADAPTER_STEP_FUSED))
        return dbmm_adapter_fwd_fast(x, w1, b1, gamma, beta, running_mean, running_var, nbt, w2, b2, h, mean, invstd, r, z, B, D, train, eps,
                                     momentum, (hipStream_t)stream);
    int rc = dbmm_gemm_bias_act(x, D, 0, w1, D, 0, b1, nullptr, 0, h, H, B, H, D, 1.f, DBMM_ACT_NONE, stream);
    if (rc) return rc;
    if (train) {
        rc = dbmm_bn1d_stats(h, B, H, eps, momentum, mean, invstd, running_mean, running_var, nbt, stream);
        if (rc) return rc;
        rc = dbmm_bn1d_relu(h, mean, invstd, gamma, beta, r, B, H, 0, eps, stream);
    } else {
        rc = dbmm_bn1d_relu(h, running_mean, running_var, gamma, beta, r, B, H, 1, eps, stream);
    }
    if (rc) return rc;
    return dbmm_gemm_bias_act(r, H, 0, w2, H, 0, b2, nullptr, 0, z, D, B, D, H, 1.f, DBMM_ACT_NONE, stream);
}

// dr [B][H] | dh [B][H] | the fast path's partial sums (adapter_step.hip)
extern "C" size_t dbmm_workspace_bytes_adapter_bwd(int64_t B, int64_t D, int64_t H) {
    return ((size_t)(2 * B * H) + (dbmm_adapter_fast_shape(B, D, H) ? dbmm_adapter_bwd_fast_floats(B, D) : 0)) * sizeof(float);
}

extern "C" int dbmm_adapter_bwd(const float* x, const float* dz, const float* h, const float* mean,
                                const float* invstd, const float* r, const float* gamma, const float* beta,
                                const float* w2, float* dw1, float* db1, float* dgamma, float* dbeta, float* dw2,
                                float* db2, int64_t B, int64_t D, int64_t H, void* workspace, size_t workspace_bytes,
                                void* stream) {
    if (!x || !dz || !h || !mean || !invstd || !r || !gamma || !beta || !w2 || !dw1 || !db1 || !dgamma || !dbeta ||
        !dw2 || !db2 || !workspace)
        return DBMM_E_ARG;
    if (B < 2 || D <= 0 || H <= 0 || (D & 3) || (H & 3)) return DBMM_E_SHAPE;
    if (workspace_bytes < dbmm_workspace_bytes_adapter_bwd(B, D, H)) return DBMM_E_WORKSPACE;
    if (!dbmm_aligned16(workspace)) return DBMM_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    float* dr = (float*)workspace;
    float* dh = dr + B * H;
    if (dbmm_adapter_fast_shape(B, D, H) && dbmm_aligned16(x) && dbmm_aligned16(dz) && dbmm_aligned16(r) && dbmm_aligned16(w2) && dbmm_aligned16(dw1) &&
        dbmm_aligned16(dw2) && dbmm_aligned16(db1) && dbmm_aligned16(db2) && dbmm_opt(OPT_ADAPTER_STEP_FUSED))
        return dbmm_adapter_bwd_fast(x, dz, h, mean, invstd, r, gamma, beta, w2, dw1, db1, dgamma, dbeta, dw2, db2, dh, dh + B * H, B, D, s);
    int rc;
    // dW2[D][H] = dz^T r   (reduction over the batch: both operands K-major)
    rc = dbmm_gemm_bias_act(dz, D, 1, r, H, 1, nullptr, nullptr, 0, dw2, H, D, H, B, 1.f, DBMM_ACT_NONE, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((D + 15) / 16)), dim3(256), 0, s, dz, (int)B, (int)D, db2);
    DBMM_CHECK_LAUNCH();
    // dr[B][H] = dz W2   (W2 is [D][H]: K-major weight operand)
    rc = dbmm_gemm_bias_act(dz, D, 0, w2, H, 1, nullptr, nullptr, 0, dr, H, B, H, D, 1.f, DBMM_ACT_NONE, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(bn1d_bwd_reduce_kernel, dim3((unsigned)((H + 7) / 8)), dim3(256), 0, s, dr, h, mean, invstd,
                       gamma, beta, (int)B, (int)H, dgamma, dbeta);
    DBMM_CHECK_LAUNCH();
    const long long total = (long long)B * H;
    hipLaunchKernelGGL(bn1d_bwd_apply_kernel, dim3(grid_for(total)), dim3(256), 0, s, dr, h, mean, invstd, gamma, beta,
                       dgamma, dbeta, dh, (int)H, 1.f / (float)B, total);
    DBMM_CHECK_LAUNCH();
    // dW1[H][D] = dh^T x
    rc = dbmm_gemm_bias_act(dh, H, 1, x, D, 1, nullptr, nullptr, 0, dw1, D, H, D, B, 1.f, DBMM_ACT_NONE, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((H + 15) / 16)), dim3(256), 0, s, dh, (int)B, (int)H, db1);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

extern "C" int dbmm_l2norm_rows(const float* x, float* y, int64_t B, int64_t D, void* stream) {
    if (!x || !y) return DBMM_E_ARG;
    if (B <= 0 || D <= 0 || (D & 3) || B > INT32_MAX) return DBMM_E_SHAPE;
    if (!dbmm_aligned16(x) || !dbmm_aligned16(y)) return DBMM_E_ALIGN;
    hipLaunchKernelGGL(l2norm_rows_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, y, (int)B, (int)(D / 4));
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

extern "C" int dbmm_colsum(const float* x, float* out, int64_t B, int64_t N, void* stream) {
    if (!x || !out) return DBMM_E_ARG;
    if (B <= 0 || N <= 0 || (N & 3) || B > INT32_MAX) return DBMM_E_SHAPE;
    if (!dbmm_aligned16(x) || !dbmm_aligned16(out)) return DBMM_E_ALIGN;
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((N + 15) / 16)), dim3(256), 0, (hipStream_t)stream, x, (int)B, (int)N, out);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

extern "C" int dbmm_text_colnorm(const float* text, float* tn, int64_t D, int64_t C, void* stream) {
    if (!text || !tn) return DBMM_E_ARG;
    if (D <= 0 || C <= 0 || C > 65535) return DBMM_E_SHAPE;
    hipLaunchKernelGGL(text_colnorm_kernel, dim3((unsigned)C), dim3(256), 0, (hipStream_t)stream, text, tn, (int)D, (int)C);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

extern "C" int dbmm_l2norm_sim_ce_fwd(const float* z, const float* z_old, float ebd_weight, const float* tn,
                                      const int64_t* labels, float temperature, float* logits, float* loss_rows,
                                      float* loss_mean, int64_t* pred, float* inv_norm, int64_t B, int64_t D,
                                      int64_t C, void* stream) {
    if (!z || !tn) return DBMM_E_ARG;
    if (B <= 0 || D <= 0 || (D & 3) || C <= 0 || C > 8 || B > INT32_MAX) return DBMM_E_SHAPE;
    if (loss_mean && !(loss_rows && labels)) return DBMM_E_ARG;
    if (!dbmm_aligned16(z) || !dbmm_aligned16(tn) || (z_old && !dbmm_aligned16(z_old))) return DBMM_E_ALIGN;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)((B + 3) / 4));
    if (C <= 4)
        hipLaunchKernelGGL(l2norm_sim_ce_fwd_kernel<4>, grid, dim3(256), 0, s, z, z_old, ebd_weight, tn,
                           (const long long*)labels, 1.f / temperature, logits, loss_rows, (long long*)pred, inv_norm,
                           (int)B, (int)(D / 4), (int)C);
    else
        hipLaunchKernelGGL(l2norm_sim_ce_fwd_kernel<8>, grid, dim3(256), 0, s, z, z_old, ebd_weight, tn,
                           (const long long*)labels, 1.f / temperature, logits, loss_rows, (long long*)pred, inv_norm,
                           (int)B, (int)(D / 4), (int)C);
    DBMM_CHECK_LAUNCH();
    if (loss_mean) {
        hipLaunchKernelGGL(mean_reduce_kernel, dim3(1), dim3(256), 0, s, loss_rows, (int)B, loss_mean);
        DBMM_CHECK_LAUNCH();
    }
    return DBMM_OK;
}

extern "C" int dbmm_l2norm_sim_ce_bwd(const float* z, const float* inv_norm, float ebd_weight, int blended,
                                      const float* tn, const float* logits, const int64_t* labels,
                                      const float* dlogits, float temperature, float grad_scale, float* dz,
                                      int64_t B, int64_t D, int64_t C, void* stream) {
    if (!z || !inv_norm || !tn || !dz) return DBMM_E_ARG;
    if (!dlogits && (!logits || !labels)) return DBMM_E_ARG;
    if (B <= 0 || D <= 0 || (D & 3) || C <= 0 || C > 8 || B > INT32_MAX) return DBMM_E_SHAPE;
    if (!dbmm_aligned16(z) || !dbmm_aligned16(tn) || !dbmm_aligned16(dz)) return DBMM_E_ALIGN;
    const float w_new = blended ? (1.f - ebd_weight) : 1.f;
    const float gs = grad_scale / (float)B;
    const dim3 grid((unsigned)((B + 3) / 4));
    hipStream_t s = (hipStream_t)stream;
    if (C <= 4)
        hipLaunchKernelGGL(l2norm_sim_ce_bwd_kernel<4>, grid, dim3(256), 0, s, z, inv_norm, w_new, tn, logits,
                           (const long long*)labels, dlogits, 1.f / temperature, gs, dz, (int)B, (int)(D / 4), (int)C);
    else
        hipLaunchKernelGGL(l2norm_sim_ce_bwd_kernel<8>, grid, dim3(256), 0, s, z, inv_norm, w_new, tn, logits,
                           (const long long*)labels, dlogits, 1.f / temperature, gs, dz, (int)B, (int)(D / 4), (int)C);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

// forward + backward of the cosine-logits / CE head in one launch (the one-call step): logits, per-row losses, dz
static int ce_fwdbwd(const float* z, const float* z_old, float ebd_weight, const float* tn, const int64_t* labels, float temperature,
                     float* logits, float* loss_rows, float* dz, int64_t B, int64_t D, int64_t C, void* stream) {
    const float w_new = z_old ? (1.f - ebd_weight) : 1.f;
    const float gs = 1.f / (float)B;
    const dim3 grid((unsigned)((B + 3) / 4));
    hipStream_t s = (hipStream_t)stream;
    if (C <= 4)
        hipLaunchKernelGGL(l2norm_sim_ce_fwdbwd_kernel<4>, grid, dim3(256), 0, s, z, z_old, ebd_weight, w_new, tn, (const long long*)labels,
                           1.f / temperature, gs, logits, loss_rows, dz, (int)B, (int)(D / 4), (int)C);
    else
        hipLaunchKernelGGL(l2norm_sim_ce_fwdbwd_kernel<8>, grid, dim3(256), 0, s, z, z_old, ebd_weight, w_new, tn, (const long long*)labels,
                           1.f / temperature, gs, logits, loss_rows, dz, (int)B, (int)(D / 4), (int)C);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

static int sgd_impl(int64_t n, float* const* params, const float* const* grads, float* const* bufs, const int64_t* sizes, const int* nsplit,
                    float lr, float momentum, float weight_decay, int first_step, void* stream);

extern "C" int dbmm_sgd_momentum(int64_t n, float* const* params, const float* const* grads, float* const* bufs,
                                 const int64_t* sizes, float lr, float momentum, float weight_decay, int first_step,
                                 void* stream) {
    return sgd_impl(n, params, grads, bufs, sizes, nullptr, lr, momentum, weight_decay, first_step, stream);
}

static int sgd_impl(int64_t n, float* const* params, const float* const* grads, float* const* bufs, const int64_t* sizes, const int* nsplit,
                    float lr, float momentum, float weight_decay, int first_step, void* stream) {
    if (!params || !grads || !bufs || !sizes) return DBMM_E_ARG;
    if (n <= 0 || n > 16) return DBMM_E_SHAPE;
    SgdArgs a{};
    long long mx = 0;
    for (int i = 0; i < n; ++i) {
        if (!params[i] || !grads[i] || !bufs[i] || sizes[i] <= 0) return DBMM_E_ARG;
        a.p[i] = params[i]; a.g[i] = grads[i]; a.m[i] = bufs[i]; a.n[i] = sizes[i]; a.ns[i] = nsplit ? nsplit[i] : 1;
        if (sizes[i] > mx) mx = sizes[i];
    }
    long long bx = (mx + 1023) / 1024;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)bx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, a, lr, momentum,
                       weight_decay, first_step);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

extern "C" int dbmm_group_count(const float* logits, const int64_t* y, const int64_t* g, int64_t* counts, int64_t B,
                                int64_t C, int64_t G, void* stream) {
    if (!logits || !y || !g || !counts) return DBMM_E_ARG;
    if (B <= 0 || C <= 0 || G <= 0 || G > 64 || B > INT32_MAX) return DBMM_E_SHAPE;
    hipLaunchKernelGGL(group_count_kernel, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, (hipStream_t)stream, logits,
                       (const long long*)y, (const long long*)g, (unsigned long long*)counts, (int)B, (int)C, (int)G);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

extern "C" int dbmm_group_loss_sum(const float* loss_rows, const int64_t* g, float* sums, int64_t B, int64_t G,
                                   void* stream) {
    if (!loss_rows || !g || !sums) return DBMM_E_ARG;
    if (B <= 0 || G <= 0 || G > 64 || B > INT32_MAX) return DBMM_E_SHAPE;
    hipLaunchKernelGGL(group_loss_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, loss_rows,
                       (const long long*)g, sums, (int)B, (int)G);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

extern "C" int dbmm_gather_rows(const float* table, const int64_t* idx, float* out, int64_t n_rows, int64_t n_idx,
                                int64_t D, void* stream) {
    if (!table || !idx || !out) return DBMM_E_ARG;
    if (n_rows <= 0 || n_idx <= 0 || D <= 0 || (D & 3)) return DBMM_E_SHAPE;
    if (!dbmm_aligned16(table) || !dbmm_aligned16(out)) return DBMM_E_ALIGN;
    const long long total = (long long)n_idx * (D / 4);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, table,
                       (const long long*)idx, out, (int)(D / 4), (long long)n_rows, total);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

// ---- one call = one training step body (final_main.py:455-466 / :610-623) ----------------------
// workspace layout (floats): h, r [B*H]x2 | z [B*D] | mean, invstd [H]x2 | (old: h, r, z, mean, invstd)
// | inv_norm [B] | dz [B*D] | dw1 [H*D] db1 dgamma dbeta [H]x3 dw2 [D*H] db2 [D] | bwd scratch [2*B*H + B*D]
static size_t train_step_floats(int64_t B, int64_t D, int64_t H, int with_old) {
    size_t n = 2 * B * H + B * D + 2 * H;
    if (with_old) n += 2 * B * H + B * D + 2 * H;
    n += B + B * D + (H * D + 3 * H + D * H + D) + dbmm_workspace_bytes_adapter_bwd(B, D, H) / sizeof(float);
    return (n + 3) / 4 * 4 + 64;
}

extern "C" size_t dbmm_workspace_bytes_adapter_train_step(int64_t B, int64_t D, int64_t H, int with_old) {
    return train_step_floats(B, D, H, with_old) * sizeof(float);
}

extern "C" int dbmm_adapter_train_step(const float* x, const int64_t* labels, float* w1, float* b1, float* gamma,
                                       float* beta, float* rmean, float* rvar, int64_t* nbt, float* w2, float* b2,
                                       float* m_w1, float* m_b1, float* m_gamma, float* m_beta, float* m_w2,
                                       float* m_b2, const float* o_w1, const float* o_b1, const float* o_gamma,
                                       const float* o_beta, float* o_rmean, float* o_rvar, int64_t* o_nbt,
                                       const float* o_w2, const float* o_b2, float ebd_weight, const float* tn,
                                       float temperature, float lr, float momentum, float weight_decay,
                                       int first_step, float* logits, float* loss_rows, float* loss_mean, int64_t B,
                                       int64_t D, int64_t H, int64_t C, void* workspace, size_t workspace_bytes,
                                       void* stream) {
    if (!x || !labels || !w1 || !b1 || !gamma || !beta || !rmean || !rvar || !w2 || !b2 || !m_w1 || !m_b1 ||
        !m_gamma || !m_beta || !m_w2 || !m_b2 || !tn || !logits || !loss_rows || !loss_mean || !workspace)
        return DBMM_E_ARG;
    const int with_old = o_w1 != nullptr;
    if (with_old && (!o_b1 || !o_gamma || !o_beta || !o_rmean || !o_rvar || !o_w2 || !o_b2)) return DBMM_E_ARG;
    if (B < 2 || D <= 0 || H <= 0 || (D & 3) || (H & 3) || C <= 0 || C > 8) return DBMM_E_SHAPE;
    if (workspace_bytes < dbmm_workspace_bytes_adapter_train_step(B, D, H, with_old)) return DBMM_E_WORKSPACE;
    if (!dbmm_aligned16(workspace)) return DBMM_E_ALIGN;
    auto up4 = [](size_t n) { return (n + 3) / 4 * 4; };
    float* f = (float*)workspace;
    float* h = f;            f += up4(B * H);
    float* r = f;            f += up4(B * H);
    float* z = f;            f += up4(B * D);
    float* mean = f;         f += up4(H);
    float* invstd = f;       f += up4(H);
    float *oh = nullptr, *orr = nullptr, *oz = nullptr, *omean = nullptr, *oinv = nullptr;
    if (with_old) {
        oh = f; f += up4(B * H); orr = f; f += up4(B * H); oz = f; f += up4(B * D);
        omean = f; f += up4(H); oinv = f; f += up4(H);
    }
    float* inv_norm = f;     f += up4(B);
    float* dz = f;           f += up4(B * D);
    float* dw1 = f;          f += up4(H * D);
    float* db1 = f;          f += up4(H);
    float* dgamma = f;       f += up4(H);
    float* dbeta = f;        f += up4(H);
    float* dw2 = f;          f += up4(D * H);
    float* db2 = f;          f += up4(D);
    float* bws = f;
    const float eps = 1e-5f, bn_momentum = 0.1f;
    int rc;
    rc = dbmm_adapter_fwd(x, w1, b1, gamma, beta, rmean, rvar, nbt, w2, b2, h, mean, invstd, r, z, B, D, H, 1, eps,
                          bn_momentum, stream);
    if (rc) return rc;
    if (with_old) {   // frozen branch: forward only, but BatchNorm still in train mode
        rc = dbmm_adapter_fwd(x, o_w1, o_b1, o_gamma, o_beta, o_rmean, o_rvar, o_nbt, o_w2, o_b2, oh, omean, oinv, orr,
                              oz, B, D, H, 1, eps, bn_momentum, stream);
        if (rc) return rc;
    }
    // one-call step on the purpose-built kernels: cosine logits + CE forward AND backward are one launch (row-local), the batch
    // mean of the losses rides in a spare block of the next launch, the sums of the batch-split dW1 / db1 partials in the SGD
    // launch -- the same statements as the stand-alone launches the autograd path uses, hence the same bits
    const bool fast = dbmm_adapter_fast_shape(B, D, H) && dbmm_opt(OPT_ADAPTER_STEP_FUSED) && dbmm_aligned16(x);
    if (fast) {
        if ((D & 3) || C <= 0 || C > 8) return DBMM_E_SHAPE;
        rc = ce_fwdbwd(z, oz, ebd_weight, tn, labels, temperature, logits, loss_rows, dz, B, D, C, stream);
        if (rc) return rc;
    } else {
        rc = dbmm_l2norm_sim_ce_fwd(z, oz, ebd_weight, tn, labels, temperature, logits, loss_rows, loss_mean, nullptr, inv_norm, B, D, C, stream);
        if (rc) return rc;
        rc = dbmm_l2norm_sim_ce_bwd(z, inv_norm, ebd_weight, with_old, tn, logits, labels, nullptr, temperature, 1.f, dz, B, D, C, stream);
        if (rc) return rc;
    }
    float* ps[6] = {w1, b1, gamma, beta, w2, b2};
    const float* gs[6] = {dw1, db1, dgamma, dbeta, dw2, db2};
    float* ms[6] = {m_w1, m_b1, m_gamma, m_beta, m_w2, m_b2};
    const int64_t ns[6] = {H * D, H, H, H, D * H, D};
    if (fast) {
        float* dh = bws + B * H;
        const float *dw1part = nullptr, *db1part = nullptr;
        int nsplit = 1;
        rc = dbmm_adapter_bwd_fast(x, dz, h, mean, invstd, r, gamma, beta, w2, dw1, db1, dgamma, dbeta, dw2, db2, dh, dh + B * H, B, D,
                                   (hipStream_t)stream, &dw1part, &db1part, &nsplit, loss_rows, loss_mean);
        if (rc) return rc;
        gs[0] = dw1part; gs[1] = db1part;
        const int nsp[6] = {nsplit, nsplit, 1, 1, 1, 1};
        return sgd_impl(6, ps, gs, ms, ns, nsp, lr, momentum, weight_decay, first_step, stream);
    }
    rc = dbmm_adapter_bwd(x, dz, h, mean, invstd, r, gamma, beta, w2, dw1, db1, dgamma, dbeta, dw2, db2, B, D, H, bws,
                          dbmm_workspace_bytes_adapter_bwd(B, D, H), stream);
    if (rc) return rc;
    return dbmm_sgd_momentum(6, ps, gs, ms, ns, lr, momentum, weight_decay, first_step, stream);
}

extern "C" int dbmm_version(void) { return 101; }

extern "C" const char* dbmm_error_string(int code) {
    switch (code) {
        case DBMM_OK: return "ok";
        case DBMM_E_SHAPE: return "unsupported or inconsistent dimensions";
        case DBMM_E_ALIGN: return "pointer or leading dimension not 16-byte aligned";
        case DBMM_E_WORKSPACE: return "workspace too small";
        case DBMM_E_ARG: return "null pointer or bad enum";
        case DBMM_E_UNSUPPORTED: return "no kernel for this (valid) request; use the unfused calls";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown dbmm error";
    }
}
