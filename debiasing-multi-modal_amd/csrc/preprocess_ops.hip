// Device-side image preprocessing: the reference's
// `_transform` (clip/clip.py:79-86) = Resize(n_px, BICUBIC) -> CenterCrop(n_px) -> ToTensor ->
// Normalize, for an RGB uint8 image already in HBM.  The resize reproduces Pillow's 8-bit
// resampler (the library the reference's torchvision transform calls) bit for bit: it is
// integer arithmetic -- coefficients in 22-bit fixed point (built on the host, see
// preprocess.py), a horizontal pass to uint8, a vertical pass to uint8 -- so the uint8 image
// is identical to PIL's and the fp32 output identical to ToTensor/Normalize of it.
// Only the R x R crop window is ever computed.  Byte-granular, HBM/L2-bound integer work.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "dbmm.h"

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;   // Pillow's Resample.c

__device__ __forceinline__ int clip8(int v) {
    v >>= PRECISION_BITS;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal pass: tmp[y - row0][x][c] for the crop's R columns and the source rows
// [row0, row0 + nrows) the vertical pass will read.  bounds/kk are indexed by crop column.
__global__ __launch_bounds__(256) void resize_h_kernel(const uint8_t* __restrict__ img, int W,
                                                       const int* __restrict__ bounds, const int* __restrict__ kk,
                                                       int ks, int row0, int nrows, int R, uint8_t* __restrict__ tmp,
                                                       long long img_stride, long long tmp_stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nrows * R) return;
    img += (long long)blockIdx.y * img_stride;               // blockIdx.y = image of a uniform-geometry batch
    tmp += (long long)blockIdx.y * tmp_stride;
    const int y = i / R, x = i - y * R;
    const int xmin = bounds[2 * x], xn = bounds[2 * x + 1];
    const int* k = kk + (long long)x * ks;
    const uint8_t* src = img + ((long long)(row0 + y) * W + xmin) * 3;
    int s0 = 1 << (PRECISION_BITS - 1), s1 = s0, s2 = s0;
    for (int t = 0; t < xn; ++t) {
        const int kv = k[t];
        s0 += (int)src[3 * t] * kv; s1 += (int)src[3 * t + 1] * kv; s2 += (int)src[3 * t + 2] * kv;
    }
    uint8_t* o = tmp + (long long)i * 3;
    o[0] = (uint8_t)clip8(s0); o[1] = (uint8_t)clip8(s1); o[2] = (uint8_t)clip8(s2);
}

// vertical pass + ToTensor + Normalize: out[c][y][x] = (u8 / 255 - mean[c]) / std[c], fp32.
// bounds are relative to row0 of tmp.
__global__ __launch_bounds__(256) void resize_v_norm_kernel(const uint8_t* __restrict__ tmp,
                                                            const int* __restrict__ bounds, const int* __restrict__ kk,
                                                            int ks, int R, float m0, float m1, float m2, float s0f,
                                                            float s1f, float s2f, float* __restrict__ out,
                                                            uint8_t* __restrict__ out_u8, long long tmp_stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= R * R) return;
    tmp += (long long)blockIdx.y * tmp_stride;
    out += (long long)blockIdx.y * 3 * R * R;
    if (out_u8) out_u8 += (long long)blockIdx.y * 3 * R * R;
    const int y = i / R, x = i - y * R;
    const int ymin = bounds[2 * y], yn = bounds[2 * y + 1];
    const int* k = kk + (long long)y * ks;
    const uint8_t* src = tmp + ((long long)ymin * R + x) * 3;
    int a0 = 1 << (PRECISION_BITS - 1), a1 = a0, a2 = a0;
    for (int t = 0; t < yn; ++t) {
        const int kv = k[t];
        const uint8_t* p = src + (long long)t * R * 3;
        a0 += (int)p[0] * kv; a1 += (int)p[1] * kv; a2 += (int)p[2] * kv;
    }
    const int u0 = clip8(a0), u1 = clip8(a1), u2 = clip8(a2);
    if (out_u8) {
        uint8_t* o = out_u8 + (long long)i * 3;
        o[0] = (uint8_t)u0; o[1] = (uint8_t)u1; o[2] = (uint8_t)u2;
    }
    const long long plane = (long long)R * R;
    out[i] = ((float)u0 / 255.0f - m0) / s0f;
    out[plane + i] = ((float)u1 / 255.0f - m1) / s1f;
    out[2 * plane + i] = ((float)u2 / 255.0f - m2) / s2f;
}

}  // namespace

extern "C" size_t dbmm_workspace_bytes_preprocess(int64_t nrows, int64_t R) { return (size_t)(nrows * R * 3); }

static int preprocess_launch(const uint8_t* img, int64_t B, int64_t H, int64_t W, const int32_t* h_bounds, const int32_t* h_coeffs, int64_t h_ksize,
                             const int32_t* v_bounds, const int32_t* v_coeffs, int64_t v_ksize, int64_t row0, int64_t nrows, int64_t R,
                             const float* mean3, const float* std3, float* out_chw, uint8_t* out_u8_hwc, void* workspace, size_t workspace_bytes,
                             void* stream) {
    if (!img || !h_bounds || !h_coeffs || !v_bounds || !v_coeffs || !mean3 || !std3 || !out_chw || !workspace)
        return DBMM_E_ARG;
    if (B <= 0 || B > 65535 || H <= 0 || W <= 0 || R <= 0 || h_ksize <= 0 || v_ksize <= 0 || nrows <= 0 || row0 < 0 || row0 + nrows > H ||
        R * R > INT32_MAX || nrows * R > INT32_MAX || H * W * 3 > INT32_MAX)
        return DBMM_E_SHAPE;
    if (workspace_bytes < B * dbmm_workspace_bytes_preprocess(nrows, R)) return DBMM_E_WORKSPACE;
    hipStream_t s = (hipStream_t)stream;
    uint8_t* tmp = (uint8_t*)workspace;
    const long long tmp_stride = nrows * R * 3;
    hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)((nrows * R + 255) / 256), (unsigned)B), dim3(256), 0, s, img, (int)W,
                       h_bounds, h_coeffs, (int)h_ksize, (int)row0, (int)nrows, (int)R, tmp, (long long)(H * W * 3), tmp_stride);
    DBMM_CHECK_LAUNCH();
    hipLaunchKernelGGL(resize_v_norm_kernel, dim3((unsigned)((R * R + 255) / 256), (unsigned)B), dim3(256), 0, s, tmp, v_bounds,
                       v_coeffs, (int)v_ksize, (int)R, mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2],
                       out_chw, out_u8_hwc, tmp_stride);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

extern "C" int dbmm_resize_crop_normalize_u8(const uint8_t* img_hwc, int64_t H, int64_t W, const int32_t* h_bounds,
                                             const int32_t* h_coeffs, int64_t h_ksize, const int32_t* v_bounds,
                                             const int32_t* v_coeffs, int64_t v_ksize, int64_t row0, int64_t nrows,
                                             int64_t R, const float* mean3, const float* std3, float* out_chw,
                                             uint8_t* out_u8_hwc, void* workspace, size_t workspace_bytes,
                                             void* stream) {
    return preprocess_launch(img_hwc, 1, H, W, h_bounds, h_coeffs, h_ksize, v_bounds, v_coeffs, v_ksize, row0, nrows, R, mean3, std3, out_chw,
                             out_u8_hwc, workspace, workspace_bytes, stream);
}

// a batch of B images of ONE geometry [B][H][W][3] (a dataset like CelebA: every image 218 x 178) in two launches;
// out_chw [B][3][R][R]; workspace >= B * dbmm_workspace_bytes_preprocess(nrows, R).  Same arithmetic as the single-image entry.
extern "C" int dbmm_resize_crop_normalize_u8_batch(const uint8_t* img_bhwc, int64_t B, int64_t H, int64_t W, const int32_t* h_bounds,
                                                   const int32_t* h_coeffs, int64_t h_ksize, const int32_t* v_bounds,
                                                   const int32_t* v_coeffs, int64_t v_ksize, int64_t row0, int64_t nrows,
                                                   int64_t R, const float* mean3, const float* std3, float* out_chw,
                                                   uint8_t* out_u8_hwc, void* workspace, size_t workspace_bytes, void* stream) {
    return preprocess_launch(img_bhwc, B, H, W, h_bounds, h_coeffs, h_ksize, v_bounds, v_coeffs, v_ksize, row0, nrows, R, mean3, std3, out_chw,
                             out_u8_hwc, workspace, workspace_bytes, stream);
}
