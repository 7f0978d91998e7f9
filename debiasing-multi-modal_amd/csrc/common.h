// Shared helpers for the gfx950 kernels of libdbmm_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "../../include/dbmm.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DBMM_CHECK_LAUNCH()                      \
    do {                                         \
        hipError_t e__ = hipGetLastError();      \
        if (e__ != hipSuccess) return (int)e__;  \
    } while (0)

// library options (csrc/options.hip; set by name through dbmm_set_option, seeded once from DBMM_<NAME> at load time)
enum DbmmOpt {
    OPT_IGEMM_EPI_DIRECT, OPT_IGEMM_FAST, OPT_IGEMM_STREAMK, OPT_IGEMM_X3, OPT_IGEMM_X2, OPT_IGEMM_X2_BK, OPT_IGEMM_BK,
    OPT_IGEMM_HALO, OPT_IGEMM_HALO_POOL, OPT_IGEMM_BN256, OPT_IGEMM_BN256_KXK, OPT_GEMM_8PH, OPT_F16_8PH, OPT_F16_BN256,
    OPT_STEM_MFMA, OPT_MHA_VALU, OPT_CONV_PATCH, OPT_MHA_X2, OPT_ADAPTER_STEP_FUSED, OPT_CONV1X1_STREAM, OPT_CONV1X1_8PH, OPT_CHAIN8, OPT_CONV1X1_BN256, OPT_TAIL_SPLIT, OPT_HALO8, OPT_DUAL_8PH, OPT_MHA_SHORT, OPT_F16_CONV_8PH, OPT_CONV1X1_RES_STREAM, DBMM_OPT_COUNT
};
int dbmm_opt(int id);

// conv3x3_halo8.hip: parity-mode 3x3 / stride 1 / pad 1 conv (+ scale / bias / ReLU, pool = 2: fused 2x2 average pool) on the eight-phase
// 256 x 256 structure; x NHWC fp32, w ONE fp16 plane [Cout][9 Cin] in the 32-channel-slab K order.  split != 0 + a workspace: the tiles of a short
// last round are cut along K over the idle CUs.  DBMM_E_UNSUPPORTED: not this kernel's shape.
int dbmm_conv3x3_halo8(const float* x, const float* x_absmax, const void* w_plane_f16, int w_exp, const float* out_scale, const float* bias,
                       float* y, float* y_absmax, int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int act, int pool, int split,
                       void* workspace, size_t workspace_bytes, void* stream);

// gemm_pair_8ph.hip: dbmm_gemm_pair_8ph (include/dbmm.h) with a workspace: the tiles of a short last round are cut along K over the idle CUs
int dbmm_gemm_pair_8ph_ws(const float* a, int64_t lda, const float* a_absmax, const void* w_plane_f16, int w_exp, int64_t ldw,
                          const float* out_scale, const float* bias, const float* residual, int64_t ldr, float* c, int64_t ldc,
                          float* c_absmax, int64_t M, int64_t N, int64_t K, float alpha, int act, void* workspace, size_t workspace_bytes, void* stream);
// gemm_pair_8ph.hip: dbmm_gemm_dual_bn_act_x2's GEMM on the eight-phase 256 x 256 kernel (arguments checked by the caller)
int dbmm_gemm_dual_pair_8ph(const float* a, int64_t lda, const float* a_absmax, const void* w_plane_f16, int w_exp, int64_t ldw, int64_t K,
                            const float* out_scale, const float* a2, int64_t lda2, const float* a2_absmax, const void* w2_plane_f16, int64_t ldw2,
                            int64_t K2, const float* ratio, const float* bias, float* c, int64_t ldc, float* c_absmax, int64_t M, int64_t N, int act,
                            void* stream);

// K cut of a short last round of the 256-tile persistent kernels: `rem` tiles left for 256 workgroups, `trips` loop trips per tile -> slices
// per tile (0: no cut).  One slice per workgroup at most: a slice costs ~0.3 of a layer-3 tile on top of its share of the loop (prologue,
// 256 KB of partial sums) and the summing launch reads rem x S x 256 KB, so a second slice per workgroup (rem > 128) never paid
// (HISTORY.md, round 4); with rem = 64 and S = 4 the cut is already neutral.
static inline int dbmm_cut_slices(int rem, int trips) {
    if (rem <= 0 || rem > 128) return 0;
    int S = 256 / rem;
    S = S < trips ? S : trips;
    S = S < 16 ? S : 16;
    return S >= 2 ? S : 0;
}

// conv1x1_res_stream.hip: y = relu((a @ W^T) * scale + bias + residual) (+ y_pooled = AvgPool2d(2) of y), K = 256, N % 32 == 0, M % 4 == 0
int dbmm_conv1x1_res_stream(const float* a, const float* a_absmax, const void* w_plane_f16, int w_exp, const float* scale, const float* bias,
                            const float* residual, float* y, float* y_pooled, float* y_absmax, int64_t M, int64_t Ho, int64_t Wo, int64_t K,
                            int64_t N, void* stream);

// conv1x1_res_stream_f16.hip: the fp16 twin (K = 256, Cout % 64 == 0; y_pooled: also AvgPool2d(2) of y, rows = pixels of [B][Ho][Wo] maps)
int dbmm_conv1x1_res_stream_f16(const void* x, const void* w, const float* scale, const float* bias, const void* residual, void* y, void* y_pooled,
                                int64_t M, int64_t Ho, int64_t Wo, int64_t Cin, int64_t Cout, void* stream);

static inline bool dbmm_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// wave64 butterfly sum / max (all lanes get the result)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Bijective XCD-aware remap of a linear workgroup id: blocks are dealt round-robin over the
// 8 XCDs, so give each XCD a contiguous range of tile ids (neighbouring tiles share operand
// panels in that XCD's L2).  Speed only, never correctness.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}
