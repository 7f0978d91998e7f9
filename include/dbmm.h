/*
 * dbmm.h -- C ABI of libdbmm_hip.so: the MI355X (gfx950) kernels under the reference's
 * CLIP-embedding + debiasing-adapter hot path.
 *
 * The reference (Lainshower/debiasing-multi-modal) is pure PyTorch: it has no native
 * boundary of its own.  Each entry point below replaces the ATen operator sequence at the
 * cited reference site (paths relative to /root/reference); the Python mirror of the
 * reference classes in debiasing-multi-modal_amd/ binds them through ctypes
 * (INTEGRATION.md shows the stub a maintainer would add).
 *
 * Conventions
 *   - plain pointers + int64 sizes; no torch / hip types in signatures (`stream` is a
 *     hipStream_t passed as void*; NULL = the null stream).
 *   - the caller owns every buffer, including workspaces; the library never allocates,
 *     frees, retains pointers or synchronises.  All work is enqueued on `stream`.
 *   - return 0 on success, a negative DBMM_E_* for shape/alignment violations, or the
 *     positive hipError_t of a failed launch.  Never throws, never aborts.
 *   - activations are fp32, channels-last (NHWC) inside the library; image input is the
 *     reference's NCHW and is converted by the first kernel that touches it.
 *   - all float pointers must be 16-byte aligned; leading dimensions multiples of 4.
 */
#ifndef DBMM_H
#define DBMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DBMM_OK 0
#define DBMM_E_SHAPE (-1)     /* unsupported / inconsistent dimensions            */
#define DBMM_E_ALIGN (-2)     /* pointer or leading dimension not 16-byte aligned */
#define DBMM_E_WORKSPACE (-3) /* workspace too small                              */
#define DBMM_E_ARG (-4)       /* null pointer / bad enum                          */
#define DBMM_E_UNSUPPORTED (-5) /* valid request this build has no kernel for (caller falls back) */

/* K order of a packed conv weight [Cout][K], K = KH*KW*Cin */
#define DBMM_WL_TAP_MAJOR 0   /* (kh, kw, cin)                                              */
#define DBMM_WL_CHUNK_MAJOR 1 /* (cin/16, kh, kw, 16): needs Cin % 16 == 0; the taps of one
                                 16-channel slab are adjacent along K, so the KH*KW re-reads
                                 of an input pixel hit L1/L2 instead of HBM                  */
#define DBMM_WL_CHUNK32_MAJOR 2 /* (cin/32, kh, kw, 32): the same with 32-channel slabs (Cin % 32
                                 == 0) -- the order the 32-deep fp16-pair kernels can step     */

#define DBMM_ACT_NONE 0
#define DBMM_ACT_RELU 1
#define DBMM_ACT_QUICKGELU 2 /* x * sigmoid(1.702 x), clip/model.py:166-168 */

int dbmm_version(void);
const char* dbmm_error_string(int code);

/* Library options: the switches the launchers consult (which kernel family serves a shape; every setting gives the same
 * results up to fp32 rounding).  Names are listed in csrc/options.hip ("igemm_halo", "gemm_8ph", "f16_8ph",
 * "adapter_step_fused", ...).  Defaults are the measured best; each option is seeded once, at load time, from the
 * environment variable DBMM_<NAME IN UPPER CASE> when that is set -- no launch path reads the environment.
 * Returns DBMM_E_ARG for an unknown name.  The reference has no counterpart (it has no native code). */
int dbmm_set_option(const char* name, int value);
int dbmm_get_option(const char* name, int* value);

/* ---------------------------------------------------------------------------------------
 * Dense contractions on fp32 MFMA (v_mfma_f32_32x32x2_f32), LDS-tiled implicit GEMM.
 * ------------------------------------------------------------------------------------ */

/* y[B,Ho,Wo,Cout] = act( conv(x[B,H,W,Cin], w) + bias + residual ), NHWC, BatchNorm already
 * folded into (w, bias) by the caller.  w is [Cout][KH][KW][Cin].  Cin % 4 == 0.
 * Replaces conv->bn->relu(->add) of Bottleneck.forward (clip/model.py:45-54) and the stem
 * convs 2/3 (clip/model.py:141-142).  residual may be NULL; it has y's shape. */
int dbmm_conv_bn_act(const float* x, const float* w, const float* bias, const float* residual,
                     float* y, int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                     int64_t KH, int64_t KW, int64_t stride, int64_t pad, int act, void* stream);
/* named specialisations of the above */
int dbmm_conv1x1_bn_act(const float* x, const float* w, const float* bias, const float* residual,
                        float* y, int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                        int act, void* stream);
int dbmm_conv3x3_bn_act(const float* x, const float* w, const float* bias, const float* residual,
                        float* y, int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                        int act, void* stream);

/* c[M,N] = act( alpha * (op(a) @ op(w)^T + bias) + residual )
 *   trans_a = 0: a is [M][lda] (K contiguous);  1: a is [K][lda] (M contiguous)
 *   trans_w = 0: w is [N][ldw] (K contiguous, i.e. nn.Linear.weight); 1: w is [K][ldw]
 * Replaces F.linear / addmm at clip/model.py:72-90 (attention-pool projections), :185-191
 * (QKV / out_proj / c_fc+QuickGELU / c_proj with residual), :238, :354 (projections) and the
 * adapter Linear layers and their weight/input gradients (final_main.py:167-172). */
int dbmm_gemm_bias_act(const float* a, int64_t lda, int trans_a, const float* w, int64_t ldw,
                       int trans_w, const float* bias, const float* residual, int64_t ldr,
                       float* c, int64_t ldc, int64_t M, int64_t N, int64_t K, float alpha,
                       int act, void* stream);

/* Variants with a caller-provided scratch buffer (dbmm_workspace_bytes_igemm() bytes, 16-B
 * aligned).  With it the library may pick the stream-K work split when whole tiles would
 * leave part of the 256 CUs idle (e.g. 784 tiles); results are deterministic either way. */
size_t dbmm_workspace_bytes_igemm(void);
int dbmm_conv_bn_act_ws(const float* x, const float* w, const float* bias, const float* residual,
                        float* y, int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                        int64_t KH, int64_t KW, int64_t stride, int64_t pad, int act, int w_layout,
                        void* workspace, size_t workspace_bytes, void* stream);
int dbmm_gemm_bias_act_ws(const float* a, int64_t lda, int trans_a, const float* w, int64_t ldw,
                          int trans_w, const float* bias, const float* residual, int64_t ldr,
                          float* c, int64_t ldc, int64_t M, int64_t N, int64_t K, float alpha,
                          int act, void* workspace, size_t workspace_bytes, void* stream);

/* Split-precision variants ("x3"): each fp32 operand value is the exact sum of three bf16 values
 * and a product is accumulated as its six largest bf16 x bf16 partial products on the bf16
 * matrix cores (2.67x the fp32-MFMA rate, dropped terms <= 2^-24 relative = fp32-level accuracy).
 * w_planes = the weight split once by dbmm_split_weight_planes ([3][N][K] bf16,
 * dbmm_split_planes_bytes(N, K) bytes); activations stay fp32 and are split on the fly.  `w`
 * (fp32, same K order as the planes, see w_layout) is still required: shapes the split kernel does not cover (K or Cin
 * not a multiple of 16, N <= 32, operands >= 2 GiB) run on the fp32-MFMA kernel. */
size_t dbmm_split_planes_bytes(int64_t N, int64_t K);
int dbmm_split_weight_planes(const float* w, void* planes, int64_t N, int64_t K, void* stream);
int dbmm_conv_bn_act_x3(const float* x, const float* w, const void* w_planes, const float* bias,
                        const float* residual, float* y, int64_t B, int64_t H, int64_t W, int64_t Cin,
                        int64_t Cout, int64_t KH, int64_t KW, int64_t stride, int64_t pad, int act,
                        int w_layout, void* workspace, size_t workspace_bytes, void* stream);
int dbmm_gemm_bias_act_x3(const float* a, int64_t lda, const float* w, const void* w_planes, int64_t ldw,
                          const float* bias, const float* residual, int64_t ldr, float* c, int64_t ldc,
                          int64_t M, int64_t N, int64_t K, float alpha, int act, void* workspace,
                          size_t workspace_bytes, void* stream);

/* fp16-pair variant ("x2"): with a per-tensor power-of-two scale an fp32 value is hi + lo of two
 * fp16 values to 2^-22 relative, so three fp16 x fp16 partial products (hl, lh, hh) give
 * fp32-level accuracy at HALF the matrix-core work of the bf16 triple.  The scale of the
 * activations comes from a device scalar x_absmax >= max|x| that the producing launch wrote
 * through its y_absmax argument (atomic max over |y|; the caller zeroes the scalar beforehand;
 * an average pool may reuse its input's scalar: any upper bound works, a bound 2^k too large
 * costs k bits of the 2^-39 absolute floor).  w_planes_f16 = w * 2^w_exp split by
 * dbmm_split_weight_planes_f16 ([2][N][K] fp16); choose w_exp so that max|w| * 2^w_exp < 2^15.
 * x_absmax or w_planes_f16 may be NULL (fp32-MFMA kernel, y_absmax still honoured).
 * w_planes = 2: hi and lo plane.  w_planes = 1: the scaled weight is EXACTLY representable in
 * fp16 (true for every conv / linear weight the reference's build_model loads: it stores them
 * in fp16, clip/model.py:375-396,433) and only that plane is given -> two partial products; the
 * 32-deep K chunk is required (K % 32 == 0, and Cin % 32 == 0 for KxK convs), otherwise the
 * fp32-MFMA kernel runs.  out_scale (optional, [Cout]) multiplies the accumulator per output
 * channel before the bias: y = act((acc * out_scale + bias) + residual) -- it carries the
 * BatchNorm scale so that the weights themselves can stay the stored fp16 values.
 * pool = 2 fuses the AvgPool2d(2) that follows the conv in the reference's stem and stride-2
 * bottlenecks (clip/model.py:25,48,117,145) into the epilogue: the kernel walks the output
 * pixels 2x2-window-major, averages each window after the activation and writes only the pooled
 * tensor y[B][Ho/2][Wo/2][Cout] (same summation order as dbmm_avgpool2d: bit-identical result).
 * With pool = 2, y_full (optional, [B][Ho][Wo][Cout]) additionally receives the un-pooled output:
 * the last conv3 of a stage needs both -- the next block's conv1 reads the full map, its
 * downsample branch the pooled one (clip/model.py:36-38).  The residual, if any, is un-pooled.
 * Needs the fp16-pair kernel and even Ho and Wo; otherwise DBMM_E_UNSUPPORTED is returned and
 * nothing is launched (run the conv and the pool separately).  y_full must be NULL when pool = 0. */
size_t dbmm_split_planes_f16_bytes(int64_t N, int64_t K);
int dbmm_split_weight_planes_f16(const float* w, void* planes, int64_t N, int64_t K, int w_exp, void* stream);
int dbmm_conv_bn_act_x2(const float* x, const float* x_absmax, const float* w, const void* w_planes_f16,
                        int w_planes, int w_exp, const float* out_scale, const float* bias,
                        const float* residual, float* y, float* y_full, float* y_absmax,
                        int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout, int64_t KH, int64_t KW,
                        int64_t stride, int64_t pad, int act, int pool, int w_layout, void* workspace,
                        size_t workspace_bytes, void* stream);

/* GEMM on the fp16-pair path (see dbmm_conv_bn_act_x2 for the operands): c = act(alpha *
 * (a @ w^T * out_scale + bias) + residual), a row-major [M][K] with a device scalar a_absmax >=
 * max|a|, w [N][K] plus its fp16 planes (w_planes = 1 when w * 2^w_exp is exact in fp16 -- every
 * nn.Linear / attention projection weight of a model loaded like the reference does; needs
 * K % 32 == 0), c_absmax (optional) receives max|c|.  The attention core's output is a convex
 * combination of V rows, so the qkv GEMM's c_absmax also bounds it. */
int dbmm_gemm_bias_act_x2(const float* a, int64_t lda, const float* a_absmax, const float* w,
                          const void* w_planes_f16, int w_planes, int w_exp, int64_t ldw,
                          const float* out_scale, const float* bias, const float* residual, int64_t ldr,
                          float* c, int64_t ldc, float* c_absmax, int64_t M, int64_t N, int64_t K, float alpha,
                          int act, void* workspace, size_t workspace_bytes, void* stream);

/* First block of a ResNet stage (clip/model.py:36-38,42-55): out = relu(bn3(conv3(y)) +
 * bn_d(conv_d(avgpool(x)))).  Both convs are 1x1, so the sum is ONE GEMM over the concatenated
 * K = K + K2 -- the downsample branch's output is never written or re-read:
 *   c = act((a @ w^T) * 2^-(s+w_exp) * out_scale[n] + (a2 @ w2^T) * 2^-(s2+w2_exp) * out_scale2[n] + bias[n])
 * a [M][K] (lda) = conv2 output, a2 [M][K2] (lda2) = (pooled) block input, both with device scalars
 * a_absmax / a2_absmax; w_plane_f16 / w2_plane_f16 = single exact fp16 planes [N][K] / [N][K2] of
 * the stored weights times 2^w_exp / 2^w2_exp.  The kernel accumulates the second pair first,
 * multiplies the accumulators by ratio[n] * 2^(s - s2) and continues with the first pair;
 * ratio[n] = out_scale2[n] / out_scale[n] * 2^(w_exp - w2_exp) is supplied by the caller (out_scale
 * must be non-zero), bias = both BatchNorm biases added.  K, K2 multiples of 32.
 * DBMM_E_UNSUPPORTED (nothing launched) when the shape does not suit the 128x128 fp16-pair tile. */
int dbmm_gemm_dual_bn_act_x2(const float* a, int64_t lda, const float* a_absmax, const void* w_plane_f16,
                             int w_exp, int64_t ldw, int64_t K, const float* out_scale, const float* a2,
                             int64_t lda2, const float* a2_absmax, const void* w2_plane_f16, int64_t ldw2,
                             int64_t K2, const float* ratio, const float* bias, float* c, int64_t ldc,
                             float* c_absmax, int64_t M, int64_t N, int act, void* workspace,
                             size_t workspace_bytes, void* stream);

/* The two wide stem convolutions of ModifiedResNet (clip/model.py:108-116): 3x3 / stride 1 / pad 1 over Cin = 32
 * channels, eval-mode BatchNorm scale / bias, ReLU, optional AvgPool2d(2) (pool = 2), one launch of a persistent kernel
 * that keeps the whole weight in LDS and forms the nine taps from one 6 x 30 input patch per 4 x 28 output tile.
 * x NHWC [B][H][W][32] with device scalar x_absmax; w_plane_f16 [Cout][kh][kw][32] = stored weight * 2^w_exp as one
 * exact fp16 plane; y NHWC [B][H][W][Cout] or, pooled, [B][H/2][W/2][Cout]; y_absmax optional.  Same fp16-pair
 * arithmetic as dbmm_conv_bn_act_x2.  Served: Cin = 32, Cout in {32, 64}, H % 4 == 0, W % 28 == 0; DBMM_E_UNSUPPORTED
 * otherwise (nothing launched). */
int dbmm_conv3x3_c32_bn_relu_x2(const float* x, const float* x_absmax, const void* w_plane_f16, int w_exp,
                                const float* scale, const float* bias, float* y, float* y_absmax, int64_t B, int64_t H,
                                int64_t W, int64_t Cin, int64_t Cout, int pool, void* stream);

/* conv3 + residual of one bottleneck block chained with conv1 of the NEXT block (clip/model.py:42-55, two
 * consecutive Bottleneck.forward bodies) in one launch:
 *   x_out  = relu((y2 @ w3^T) * scale3 + bias3 + residual)        [B*Ho*Wo][N]
 *   y1_out = relu((x_out @ w1^T) * scale1 + bias1)                [B*Ho*Wo][P]
 *   x_pooled (optional) = AvgPool2d(2) of x_out                   [B*Ho/2*Wo/2][N]  (next stage's downsample input)
 * With x_pooled given, x_out may be NULL: the un-pooled tensor is then not written at all (at a stage seam its only other reader is
 * conv1 of the next block, which this launch has already computed).
 * The wide tensor x_out is written once and not read back for conv1.  y2 [B*Ho*Wo][K] fp32 with its device scalar
 * y2_absmax >= max|y2|; w3_plane_f16 [N][K] / w1_plane_f16 [P][N] = the stored (fp16-exact) weights times
 * 2^w3_exp / 2^w1_exp as single fp16 planes; scale / bias = eval-mode BatchNorm per channel.  x_absmax / y1_absmax
 * (optional, zeroed by the caller) receive the maxima for the consumers' fp16 scales.  Same fp16-pair arithmetic
 * as dbmm_conv_bn_act_x2.  Shapes served: K in {64, 128}, N % 64 == 0, P in {64, 128}, B*Ho*Wo % 4 == 0; DBMM_E_UNSUPPORTED (nothing
 * launched) otherwise -- the caller then issues the two convs separately. */
int dbmm_bottleneck_chain_x2(const float* y2, const float* y2_absmax, const void* w3_plane_f16, int w3_exp,
                             const float* scale3, const float* bias3, const float* residual, float* x_out,
                             float* x_pooled, float* x_absmax, const void* w1_plane_f16, int w1_exp,
                             const float* scale1, const float* bias1, float* y1_out, float* y1_absmax,
                             int64_t B, int64_t Ho, int64_t Wo, int64_t K, int64_t N, int64_t P, void* stream);

/* The same chain for the FIRST block of a stage whose input has the output's resolution (layer 1): the block adds
 * its downsample branch instead of a residual (clip/model.py:36-38,52), as in dbmm_gemm_dual_bn_act_x2:
 *   x_out  = relu((y2 @ w3^T) * scale3 + (a2 @ wd^T) * scale_d + bias)   with ratio[n] = scale_d[n] / scale3[n] * 2^(w3_exp - wd_exp),
 *   y1_out = relu((x_out @ w1^T) * scale1 + bias1).
 * y2 [M][K], a2 [M][K2] (the block input) with device scalars; bias = both BatchNorm biases added.  Served: K = K2 = 64,
 * N % 64 == 0, P in {64, 128}, M % 4 == 0; DBMM_E_UNSUPPORTED otherwise. */
int dbmm_bottleneck_chain_dual_x2(const float* y2, const float* y2_absmax, const void* w3_plane_f16, int w3_exp,
                                  const float* scale3, const float* bias, const float* a2, const float* a2_absmax,
                                  const void* wd_plane_f16, const float* ratio, float* x_out, float* x_absmax,
                                  const void* w1_plane_f16, int w1_exp, const float* scale1, const float* bias1,
                                  float* y1_out, float* y1_absmax, int64_t M, int64_t K, int64_t K2, int64_t N,
                                  int64_t P, void* stream);

/* The same chains started one conv earlier: conv2 (3x3, pad 1) + BatchNorm + ReLU of the block is computed inside the
 * launch from the block's conv1 output y1 (NHWC [B][H][W][K], device scalar y1_absmax), so neither y2 nor -- for conv1' --
 * x_out is read back from HBM:  y1 -> conv2 -> conv3 + (residual | downsample branch) -> x_out -> next conv1 -> y1_out.
 * w2_plane_f16 [K][(cin/32, kh, kw, 32)] = the stored 3x3 weight times 2^w2_exp as one exact fp16 plane in the
 * DBMM_WL_CHUNK32_MAJOR K order.  Pass `residual` for an ordinary block, or (a2, a2_absmax, wd_plane_f16, ratio) with
 * residual = NULL for a stage's first block at unchanged resolution (see dbmm_bottleneck_chain_dual_x2).
 * Served: (K, P) = (64, 64) or (128, 128); dual: (64, 64); N % 64 == 0; B*H*W % 4 == 0.  DBMM_E_UNSUPPORTED otherwise. */
int dbmm_bottleneck_block_chain_x2(const float* y1, const float* y1_absmax, const void* w2_plane_f16, int w2_exp,
                                   const float* scale2, const float* bias2, const void* w3_plane_f16, int w3_exp,
                                   const float* scale3, const float* bias3, const float* residual, const float* a2,
                                   const float* a2_absmax, const void* wd_plane_f16, const float* ratio, float* x_out,
                                   float* x_absmax, const void* w1_plane_f16, int w1_exp, const float* scale1,
                                   const float* bias1, float* y1_out, float* y1_out_absmax, int64_t B, int64_t H,
                                   int64_t W, int64_t K, int64_t N, int64_t P, void* stream);

/* The same GEMM (w_planes = 1 only) on the deep-pipelined 256 x 256 kernel (csrc/gemm_pair_8ph.hip): fp32 activations by
 * LDS-DMA, split into fp16 (hi, lo) when the fragments are read, two wave groups one barrier apart.  N % 256 == 0,
 * K % 64 == 0.  dbmm_gemm_bias_act_x2 routes the projections here where this kernel measured ahead (N >= 3072 with
 * K >= 1024, or K >= 4096; DBMM_GEMM_8PH = 0 never / 2 wherever it applies). */
int dbmm_gemm_pair_8ph(const float* a, int64_t lda, const float* a_absmax, const void* w_plane_f16, int w_exp, int64_t ldw,
                       const float* out_scale, const float* bias, const float* residual, int64_t ldr, float* c, int64_t ldc,
                       float* c_absmax, int64_t M, int64_t N, int64_t K, float alpha, int act, void* stream);

/* ---------------------------------------------------------------------------------------
 * fp16 mode of the transformer towers -- the reference's GPU path (convert_weights, clip/model.py:375-396; fp16
 * activations, fp32 LayerNorm statistics, clip/model.py:157-163).  Tensors marked f16 are IEEE half in HBM; every
 * product is one fp16 MFMA with fp32 accumulation; bias / QuickGELU / residual / softmax run in fp32 and the result
 * is rounded to fp16 once.  Throughput mode (BASELINE configs[4]); the fp32-accurate entry points above stay the
 * parity mode.
 * --------------------------------------------------------------------------------------- */
/* c f16 [M][ldc] = act(a f16 [M][lda] @ w f16 [N][ldw]^T + bias f32 [N]) + residual f16 [M][ldr].  K % 64 == 0, N % 8 == 0. */
int dbmm_gemm_f16(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, const void* residual,
                  int64_t ldr, void* c, int64_t ldc, int64_t M, int64_t N, int64_t K, int act, void* stream);
/* The same with a workspace (16-B aligned, dbmm_workspace_bytes_igemm() bytes are enough): when the 256 x 256 tiles of the deep-pipelined
 * kernel leave a short last round on the 256 CUs, its tiles are cut along K over the idle CUs and summed by a second launch. */
int dbmm_gemm_f16_ws(const void* a, int64_t lda, const void* w, int64_t ldw, const float* bias, const void* residual,
                     int64_t ldr, void* c, int64_t ldc, int64_t M, int64_t N, int64_t K, int act, void* workspace,
                     size_t workspace_bytes, void* stream);
/* fp16 mode of the ModifiedResNet towers (clip/model.py:10-154 on the reference's GPU path: the image is cast to fp16, :146,
 * conv weights are fp16, BatchNorm parameters fp32).  Activations f16 NHWC; eval-mode BatchNorm as fp32 per-channel
 * scale / bias on the fp32 accumulator, residual add, ReLU and AvgPool2d(2) fused, ONE rounding to fp16 when stored.
 *   conv1x1: y f16 [M][Cout] = act(x f16 [M][Cin] @ w f16 [Cout][Cin]^T * scale + bias + residual f16 [M][Cout])
 *            (Bottleneck conv1 / conv3 / downsample conv).  Cin % 64 == 0, Cout % 8 == 0, else DBMM_E_UNSUPPORTED.
 *   conv3x3: stride 1, pad 1, + BatchNorm + ReLU, pool = 0 | 2 (AvgPool2d(2) of the result: Bottleneck conv2 of a stride-2
 *            block, stem conv3); w f16 [Cout][(cin / 32, kh, kw, 32)].  Cin % 32 == 0, Cout % 8 == 0; pool: H, W even.
 *   stem:    3x3 / stride 2 / pad 1 on the NCHW image (f32, or f16 with x_is_f16; rounded to fp16 first like the
 *            reference's cast), w f32 [kh][kw][3][Cout] with BatchNorm folded, bias f32 [Cout], ReLU, y f16 NHWC.  Cout 32 | 64.
 *   avgpool2: AvgPool2d(2) on f16 NHWC (the downsample branch of a stride-2 block, clip/model.py:36-38). */
int dbmm_conv1x1_bn_act_f16(const void* x, const void* w, const float* scale, const float* bias, const void* residual, void* y,
                            int64_t M, int64_t Cin, int64_t Cout, int act, void* stream);
/* fp16 mode, first block of a stage (clip/model.py:36-38, 42-55): out = act((y2 @ w3^T) * scale3 + (xp @ wd^T) * scale_d + bias) as ONE GEMM
 * over the concatenated K -- the downsample branch's output is never written or re-read.  y2 f16 [M][K] (conv2's output), xp f16 [M][K2]
 * (the pooled block input), w3 f16 [Cout][K], wd f16 [Cout][K2], ratio[n] = scale_d[n] / scale3[n] (scale3 non-zero), bias = both
 * BatchNorm biases added.  The kernel accumulates the second pair first, multiplies the fp32 accumulators by ratio[n] and continues with
 * the first.  On the deep-pipelined GEMM kernel when Cout % 256 == 0, K % 128 == 0, K2 % 128 == 0 and M >= 16384, otherwise on the
 * streaming 1x1 kernel (K % 32 == 0, K2 % 32 == 0, Cout > 64, Cout % 8 == 0: layer 1); DBMM_E_UNSUPPORTED (nothing launched) beyond that. */
int dbmm_conv1x1_dual_bn_act_f16(const void* y2, const void* w3, const float* scale3, const void* xp, const void* wd, const float* ratio,
                                 const float* bias, void* out, int64_t M, int64_t K, int64_t K2, int64_t Cout, int act, void* stream);
/* fp16 mode: conv3 + residual of one bottleneck block chained with conv1 of the NEXT block (clip/model.py:42-55 twice) in one launch:
 *   x_out  = relu((y2 @ w3^T) * scale3 + bias3 + residual)   f16 [M][N]      y1_out = relu((x_out @ w1^T) * scale1 + bias1)   f16 [M][P]
 * The wide tensor x_out is written once and not read back for conv1; x_out is rounded to fp16 before it feeds conv1, so the results equal
 * the two launches bit for bit.  y2 f16 [M][K], w3 f16 [N][K], w1 f16 [P][N].  K, P in {64, 128}, N % 64 == 0 (layers 1 - 2);
 * DBMM_E_UNSUPPORTED (nothing launched) otherwise -- the caller then issues the two convs. */
int dbmm_bottleneck_chain_f16(const void* y2, const void* w3, const float* scale3, const float* bias3, const void* residual, void* x_out,
                              const void* w1, const float* scale1, const float* bias1, void* y1_out, int64_t M, int64_t K, int64_t N,
                              int64_t P, void* stream);
/* the same chain at a stage seam (the block is the LAST of its stage, the next block downsamples): also x_pooled = AvgPool2d(2) of x_out
 * f16 [B*Ho/2*Wo/2][N], the next stage's downsample input (fp32 sum of the fp16 values in (dy, dx) order, times 0.25: equal to
 * dbmm_avgpool2_f16 of x_out bit for bit).  x_out may be NULL: the un-pooled tensor is then not written -- conv1 of the next block is
 * computed here and nothing else reads it.  Ho, Wo even, B*Ho*Wo % 4 == 0. */
int dbmm_bottleneck_chain_pool_f16(const void* y2, const void* w3, const float* scale3, const float* bias3, const void* residual, void* x_out,
                                   void* x_pooled, const void* w1, const float* scale1, const float* bias1, void* y1_out, int64_t B, int64_t Ho,
                                   int64_t Wo, int64_t K, int64_t N, int64_t P, void* stream);
/* the same chain for the FIRST block of a stage, whose shortcut is the downsample conv instead of a residual (dbmm_conv1x1_dual_bn_act_f16's
 * arguments and arithmetic, then conv1 of the next block): x_out = relu((y2 @ w3^T + (xp @ wd^T) * ratio) * scale3 + bias).
 * K = K2 = P = 64 (layer 1's first block); DBMM_E_UNSUPPORTED (nothing launched) otherwise. */
int dbmm_bottleneck_chain_dual_f16(const void* y2, const void* w3, const float* scale3, const void* xp, const void* wd, const float* ratio,
                                   const float* bias, void* x_out, const void* w1, const float* scale1, const float* bias1, void* y1_out,
                                   int64_t M, int64_t K, int64_t K2, int64_t N, int64_t P, void* stream);
int dbmm_conv1x1_bn_act_f16_ws(const void* x, const void* w, const float* scale, const float* bias, const void* residual, void* y,
                               int64_t M, int64_t Cin, int64_t Cout, int act, void* workspace, size_t workspace_bytes,
                               void* stream);          /* with a workspace, as dbmm_gemm_f16_ws */
/* fp16 mode, the LAST block of a stage (clip/model.py:50-54, then 36-38 of the next block): y = relu(conv1x1(x) * scale + bias + residual)
 * f16 [B*Ho*Wo][Cout] AND y_pooled = AvgPool2d(2) of y f16 [B*Ho/2*Wo/2][Cout] (equal to dbmm_avgpool2_f16 of y bit for bit) in one launch.
 * Cin in {128, 256}, Cout % 64 == 0, Ho and Wo even (layers 2 / 3); DBMM_E_UNSUPPORTED (nothing launched) otherwise -- the caller then issues
 * the conv and the pool. */
int dbmm_conv1x1_res_pool_f16(const void* x, const void* w, const float* scale, const float* bias, const void* residual, void* y,
                              void* y_pooled, int64_t B, int64_t Ho, int64_t Wo, int64_t Cin, int64_t Cout, void* stream);
int dbmm_conv3x3_bn_relu_f16(const void* x, const void* w, const float* scale, const float* bias, void* y, int64_t B, int64_t H,
                             int64_t W, int64_t Cin, int64_t Cout, int pool, void* stream);
int dbmm_conv_stem_s2_f16(const void* x_nchw, int x_is_f16, const float* w, const float* bias, void* y_nhwc, int64_t B, int64_t H,
                          int64_t W, int64_t Cout, void* stream);
/* the same conv with BatchNorm as a separate per-channel scale / bias: y = relu(conv(fp16(x), fp16(w)) * scale + bias).  w fp32
 * [kh][kw][cin][cout] holding the model's fp16 conv1 weights (rounded to fp16 here: exact for an fp16-stored model, clip/model.py:146-148
 * under convert_weights); runs on the MFMA gather kernel (option stem_mfma; 0: the FMA kernel). */
int dbmm_conv_stem_s2_bn_f16(const void* x_nchw, int x_is_f16, const float* w, const float* scale, const float* bias, void* y_nhwc,
                             int64_t B, int64_t H, int64_t W, int64_t Cout, void* stream);
int dbmm_avgpool2_f16(const void* x, void* y, int64_t B, int64_t H, int64_t W, int64_t C, void* stream);
/* softmax(q k^T / sqrt(64)) v per (image, head), head_dim 64; qkv f16 [B*L][3E] (q | k | v), out f16 [B*L][E]. */
int dbmm_mha_core_f16(const void* qkv, void* out, int64_t B, int64_t L, int64_t E, int64_t heads, int causal, void* stream);
/* y f16 [rows][ldy] = LayerNorm(x f16 [rows][ldx]) with f32 gamma / beta [E], statistics in fp32.  E % 8 == 0. */
int dbmm_layernorm_f16(const void* x, int64_t ldx, const float* gamma, const float* beta, void* y, int64_t ldy,
                       int64_t rows, int64_t E, float eps, void* stream);
/* NCHW image (f32, or f16 when x_is_f16) -> patch rows f16 [B*(R/P)^2][Kp], K = 3*P*P zero-padded to Kp. */
int dbmm_im2col_patch_f16(const void* x_nchw, int x_is_f16, void* out, int64_t B, int64_t R, int64_t P, int64_t Kp,
                          void* stream);
/* tokens f16 [B][L][W] = (class token | patches f16 [B*(L-1)][W]) + positional embedding (cls, pos f32). */
int dbmm_vit_tokens_f16(const void* patches, const float* cls, const float* pos, void* out, int64_t B, int64_t L,
                        int64_t W, void* stream);
/* out f16 [n][L][W] = table f32 [vocab][W][tokens] + pos f32 [L][W];  out f16 [n][W] = x f16 [n][L][W] at argmax(tokens). */
int dbmm_embed_gather_f16(const int32_t* tokens, const float* table, const float* pos, void* out, int64_t n, int64_t L,
                          int64_t W, int64_t vocab, void* stream);
int dbmm_gather_eot_f16(const int32_t* tokens, const void* x, void* out, int64_t n, int64_t L, int64_t W, void* stream);
int dbmm_cast_f32_f16(const float* x, void* y, int64_t n, void* stream);

/* ---------------------------------------------------------------------------------------
 * Device-side preprocessing (clip/clip.py:79-86: Resize(BICUBIC) -> CenterCrop -> ToTensor ->
 * Normalize) of one decoded RGB uint8 image [H][W][3] resident on the device.  Integer
 * resampling exactly as Pillow's 8-bit resampler (22-bit fixed-point coefficients, horizontal
 * then vertical pass, each rounded to uint8); the caller supplies the coefficient tables of
 * the R crop columns / rows (bounds = (first source index, tap count) pairs; the vertical
 * bounds relative to row0; see preprocess.py::resample_coeffs) as device int32 arrays.
 * mean3 / std3 are HOST pointers to 3 floats.  out_chw = float32 [3][R][R]; out_u8_hwc
 * (optional) = the resized + cropped uint8 image [R][R][3].  workspace >=
 * dbmm_workspace_bytes_preprocess(nrows, R) bytes.
 * ------------------------------------------------------------------------------------ */
size_t dbmm_workspace_bytes_preprocess(int64_t nrows, int64_t R);
int dbmm_resize_crop_normalize_u8(const uint8_t* img_hwc, int64_t H, int64_t W, const int32_t* h_bounds,
                                  const int32_t* h_coeffs, int64_t h_ksize, const int32_t* v_bounds,
                                  const int32_t* v_coeffs, int64_t v_ksize, int64_t row0, int64_t nrows,
                                  int64_t R, const float* mean3, const float* std3, float* out_chw,
                                  uint8_t* out_u8_hwc, void* workspace, size_t workspace_bytes, void* stream);
/* the same for a batch of B images of ONE geometry [B][H][W][3] (e.g. CelebA: every image 218 x 178) in two launches:
 * out_chw = float32 [B][3][R][R], out_u8_hwc (optional) [B][R][R][3], workspace >= B * dbmm_workspace_bytes_preprocess(nrows, R). */
int dbmm_resize_crop_normalize_u8_batch(const uint8_t* img_bhwc, int64_t B, int64_t H, int64_t W, const int32_t* h_bounds,
                                  const int32_t* h_coeffs, int64_t h_ksize, const int32_t* v_bounds,
                                  const int32_t* v_coeffs, int64_t v_ksize, int64_t row0, int64_t nrows,
                                  int64_t R, const float* mean3, const float* std3, float* out_chw,
                                  uint8_t* out_u8_hwc, void* workspace, size_t workspace_bytes, void* stream);

/* profiling aid: the 11 template arguments <BM,BN,WAVES_M,WAVES_N,AMODE,WMODE,BK,MINB,FAST,SK,DMA>
 * of the calling thread's most recent igemm launch (= the kernel name rocprofv3 reports). */
void dbmm_debug_last_igemm(int* out11);

/* `batch` independent GEMMs of identical shape in one launch (grid.y): problem b uses
 * a + b*stride_a, w + b*stride_w, bias + b*stride_bias, c + b*stride_c (element strides,
 * multiples of 4).  Used for the per-head products of the collapsed attention pool. */
int dbmm_gemm_batched(const float* a, int64_t lda, int64_t stride_a, int trans_a, const float* w,
                      int64_t ldw, int64_t stride_w, int trans_w, const float* bias,
                      int64_t stride_bias, float* c, int64_t ldc, int64_t stride_c, int64_t M,
                      int64_t N, int64_t K, int64_t batch, float alpha, int act, void* stream);

/* ---------------------------------------------------------------------------------------
 * ModifiedResNet pieces (clip/model.py:94-154)
 * ------------------------------------------------------------------------------------ */

/* stem conv1: 3x3 stride 2 pad 1 on the NCHW image, BN folded, ReLU, NHWC out
 * (clip/model.py:108-110,140).  w is [3][3][3][Cout] = (kh, kw, cin, cout).  y_absmax (optional,
 * zeroed by the caller) receives max|y| like dbmm_conv_bn_act_x2's. */
int dbmm_conv_stem_s2(const float* x_nchw, const float* w, const float* bias, float* y_nhwc,
                      float* y_absmax, int64_t B, int64_t H, int64_t W, int64_t Cout, void* stream);

/* AvgPool2d(k) with kernel = stride = k, NHWC (clip/model.py:25,37,117). C % 4 == 0. */
int dbmm_avgpool2d(const float* x, float* y, int64_t B, int64_t H, int64_t W, int64_t C,
                   int64_t k, void* stream);

/* AttentionPool2d.forward (clip/model.py:68-91) on x[B,HW,C] (NHWC feature map).
 *   pos [HW+1][C]; wq [C][C] bq [C]; wkv [2C][C] bkv [2C] (k rows then v rows);
 *   wc [Dout][C] bc [Dout]; out [B][Dout].
 * workspace: dbmm_workspace_bytes_attnpool(B, HW, C) bytes. */
size_t dbmm_workspace_bytes_attnpool(int64_t B, int64_t HW, int64_t C);
int dbmm_attnpool(const float* x, const float* pos, const float* wq, const float* bq,
                  const float* wkv, const float* bkv, const float* wc, const float* bc,
                  float* out, int64_t B, int64_t HW, int64_t C, int64_t heads, int64_t Dout,
                  void* workspace, size_t workspace_bytes, void* stream);
/* The same on an fp32 (x_is_f16 = 0) or fp16 (1) feature map: the fp16 mode of the ModifiedResNet towers hands its fp16 map over without a
 * cast pass; tokens, projections and softmax stay in fp32 as above. */
int dbmm_attnpool_x(const void* x, int x_is_f16, const float* pos, const float* wq, const float* bq,
                    const float* wkv, const float* bkv, const float* wc, const float* bc, float* out,
                    int64_t B, int64_t HW, int64_t C, int64_t heads, int64_t Dout, void* workspace,
                    size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------
 * Transformer pieces (clip/model.py:157-240, 343-356)
 * ------------------------------------------------------------------------------------ */

/* LayerNorm over the last dim, fp32 statistics (clip/model.py:157-163).
 * x rows start at x + r*ldx (lets ln_post read only token 0 of every image).  y_absmax
 * (optional, zeroed by the caller) receives max|y| for a following dbmm_gemm_bias_act_x2. */
int dbmm_layernorm(const float* x, int64_t ldx, const float* gamma, const float* beta, float* y,
                   int64_t ldy, int64_t rows, int64_t E, float eps, float* y_absmax, void* stream);

/* softmax(q k^T * hd^-0.5 [+ causal mask]) v for packed qkv[B*L][3E] (batch-first rows),
 * head h uses columns [h*64, h*64+64) of each third.  head_dim must be 64 (all CLIP models).
 * Replaces the attention core inside nn.MultiheadAttention (clip/model.py:185-187). */
int dbmm_mha_core(const float* qkv, float* out, int64_t B, int64_t L, int64_t E, int64_t heads,
                  int causal, void* stream);

/* The same attention core on the 16-bit matrix cores at fp32 accuracy: q, k, v as fp16 (hi, lo) pairs under the exact
 * power-of-two scale derived from qkv_absmax (device scalar >= max|qkv|, the qkv GEMM's c_absmax), p as a pair under
 * 2^13, three partial products per fp32 product, fp32 accumulation (csrc/mha_pair.hip).  Same layouts as
 * dbmm_mha_core. */
int dbmm_mha_core_x2(const float* qkv, const float* qkv_absmax, float* out, int64_t B, int64_t L, int64_t E,
                     int64_t heads, int causal, void* stream);

/* x[n][L][W] = table[tokens[n][l]] + pos[l]  (clip/model.py:344-346).  tokens int32. */
int dbmm_embed_gather(const int32_t* tokens, const float* table, const float* pos, float* out,
                      int64_t n, int64_t L, int64_t W, int64_t vocab, void* stream);

/* patch im2col for the ViT stem conv (kernel = stride = P, clip/model.py:211,224):
 * out[B*g*g][3*P*P] in (cin, kh, kw) order = the flattened conv weight's K order. */
/* (out_absmax: optional device scalar, zeroed by the caller, receives max|out| = max|x| for the fp16-pair patch GEMM) */
int dbmm_im2col_patch(const float* x_nchw, float* out, float* out_absmax, int64_t B, int64_t R, int64_t P, void* stream);

/* tokens[B][g*g+1][W] = concat(class_embedding, patches[B][g*g][W]) + pos (clip/model.py:227-228) */
int dbmm_vit_tokens(const float* patches, const float* cls, const float* pos, float* out,
                    int64_t B, int64_t L, int64_t W, void* stream);

/* out[n][W] = x[n][eot[n]][W] where eot[n] = argmax_l tokens[n][l] (clip/model.py:354) */
int dbmm_gather_eot(const int32_t* tokens, const float* x, float* out, int64_t n, int64_t L,
                    int64_t W, void* stream);

/* ---------------------------------------------------------------------------------------
 * Adapter step (final_main.py:53-174, 426-496; demo/util.py:118-136)
 * ------------------------------------------------------------------------------------ */

/* BatchNorm1d batch statistics of h[B][H] (train mode): mean, invstd = rsqrt(biased var+eps);
 * running stats updated with momentum and the unbiased variance; nbt (int64) += 1.
 * running_* / nbt may be NULL. */
int dbmm_bn1d_stats(const float* h, int64_t B, int64_t H, float eps, float momentum, float* mean,
                    float* invstd, float* running_mean, float* running_var, int64_t* nbt,
                    void* stream);

/* r = relu(gamma * (h - mean) * invstd + beta).  var_mode = 1: the `invstd` argument holds a
 * variance (running_var, eval mode) and rsqrt(var + eps) is applied on the fly. */
int dbmm_bn1d_relu(const float* h, const float* mean, const float* invstd, const float* gamma,
                   const float* beta, float* r, int64_t B, int64_t H, int var_mode, float eps,
                   void* stream);

/* Adapter.forward (final_main.py:173): z = Linear2(relu(bn(Linear1(x)))).
 * train != 0: batch statistics (saved to mean/invstd, running stats updated).
 * Saves h (pre-BN) and r (post-ReLU) for the backward pass. */
int dbmm_adapter_fwd(const float* x, const float* w1, const float* b1, const float* gamma,
                     const float* beta, float* running_mean, float* running_var, int64_t* nbt,
                     const float* w2, const float* b2, float* h, float* mean, float* invstd,
                     float* r, float* z, int64_t B, int64_t D, int64_t H, int train, float eps,
                     float momentum, void* stream);

/* Gradients of the adapter parameters given dz = dLoss/dz[B][D] (no grad w.r.t. x:
 * embeddings.detach(), final_main.py:455).  workspace: dbmm_workspace_bytes_adapter_bwd. */
size_t dbmm_workspace_bytes_adapter_bwd(int64_t B, int64_t D, int64_t H);
int dbmm_adapter_bwd(const float* x, const float* dz, const float* h, const float* mean,
                     const float* invstd, const float* r, const float* gamma, const float* beta,
                     const float* w2, float* dw1, float* db1, float* dgamma, float* dbeta,
                     float* dw2, float* db2, int64_t B, int64_t D, int64_t H, void* workspace,
                     size_t workspace_bytes, void* stream);

/* y[B][D] = x / ||x||_row, no epsilon (CLIP.forward, clip/model.py:362-363).  D % 4 == 0. */
int dbmm_l2norm_rows(const float* x, float* y, int64_t B, int64_t D, void* stream);

/* out[N] = sum over the B rows of x[B][N] in a fixed order (bias gradient of LinearClassifier, final_main.py:43-49).  N % 4 == 0. */
int dbmm_colsum(const float* x, float* out, int64_t B, int64_t N, void* stream);

/* tn[C][D] = (text[D][C] / ||text[:,c]||)^T  (final_main.py:77, column normalisation) */
int dbmm_text_colnorm(const float* text, float* tn, int64_t D, int64_t C, void* stream);

/* Fused row L2-norm + image x text logits + cross-entropy (final_main.py:68,78,302;
 * MultipleAdapter blend :123-127,138; zero-shot tail clip_inference.py:207-216):
 *   f = z/||z||;  feat = f                      (z_old == NULL)
 *                 feat = w*z_old/||z_old|| + (1-w)*f
 *   logits[b][c] = feat . tn[c] / T;   loss_rows[b] = CE(logits[b], labels[b]);
 *   pred[b] = argmax_c logits[b][c].
 * inv_norm[B] (1/||z||) is saved for the backward.  labels/loss_rows/pred/inv_norm may be
 * NULL.  C <= 8.  loss_mean (1 float, may be NULL) = mean of loss_rows. */
int dbmm_l2norm_sim_ce_fwd(const float* z, const float* z_old, float ebd_weight, const float* tn,
                           const int64_t* labels, float temperature, float* logits,
                           float* loss_rows, float* loss_mean, int64_t* pred, float* inv_norm,
                           int64_t B, int64_t D, int64_t C, void* stream);

/* dz[B][D] through the blend weight (1-w) and the row normalisation.
 *   dlogits == NULL: loss = grad_scale * mean_b CE; the softmax is recomputed from `logits`
 *                    and `labels` (fused training path).
 *   dlogits != NULL: dLoss/dlogits [B][C] is given by the caller (autograd of an external
 *                    criterion, e.g. nn.CrossEntropyLoss at final_main.py:456); logits, labels
 *                    and grad_scale are ignored. */
int dbmm_l2norm_sim_ce_bwd(const float* z, const float* inv_norm, float ebd_weight, int blended,
                           const float* tn, const float* logits, const int64_t* labels,
                           const float* dlogits, float temperature, float grad_scale, float* dz,
                           int64_t B, int64_t D, int64_t C, void* stream);

/* torch.optim.SGD step (demo/util.py:118-136; dampening 0, no nesterov) on n tensors in one
 * launch: g' = g + wd*p; buf = first ? g' : mu*buf + g'; p -= lr*buf.   n <= 16. */
int dbmm_sgd_momentum(int64_t n, float* const* params, const float* const* grads,
                      float* const* bufs, const int64_t* sizes, float lr, float momentum,
                      float weight_decay, int first_step, void* stream);

/* One training-step body in one call (final_main.py:455-466, :610-623): adapter forward (train-mode
 * BN), optional frozen old adapter (MultipleAdapter: pass o_* = NULL for CustomCLIP), fused
 * normalise + logits + CE, backward, SGD-momentum on the six trainable tensors (m_* = momentum
 * buffers).  ~20 kernel launches enqueued back to back, no host round trips.  tn = [C][D] from
 * dbmm_text_colnorm.  Outputs: logits [B][C], loss_rows [B], loss_mean [1]. */
size_t dbmm_workspace_bytes_adapter_train_step(int64_t B, int64_t D, int64_t H, int with_old);
int dbmm_adapter_train_step(const float* x, const int64_t* labels, float* w1, float* b1, float* gamma,
                            float* beta, float* running_mean, float* running_var, int64_t* nbt,
                            float* w2, float* b2, float* m_w1, float* m_b1, float* m_gamma,
                            float* m_beta, float* m_w2, float* m_b2, const float* o_w1,
                            const float* o_b1, const float* o_gamma, const float* o_beta,
                            float* o_running_mean, float* o_running_var, int64_t* o_nbt,
                            const float* o_w2, const float* o_b2, float ebd_weight, const float* tn,
                            float temperature, float lr, float momentum, float weight_decay,
                            int first_step, float* logits, float* loss_rows, float* loss_mean,
                            int64_t B, int64_t D, int64_t H, int64_t C, void* workspace,
                            size_t workspace_bytes, void* stream);

/* out[i][:] = table[idx[i]][:] -- batch assembly from a device-resident embedding table
 * (replaces the DataLoader + per-item DataFrame lookups of data/ *_embeddings*.py) */
int dbmm_gather_rows(const float* table, const int64_t* idx, float* out, int64_t n_rows,
                     int64_t n_idx, int64_t D, void* stream);

/* update_dict (final_main.py:383-391) on device: counts[g][0] += 1, counts[g][1] += (argmax
 * logits == y) for every row; counts is int64 [G][2], accumulated (not cleared). */
int dbmm_group_count(const float* logits, const int64_t* y, const int64_t* g, int64_t* counts,
                     int64_t B, int64_t C, int64_t G, void* stream);

/* Per-group mean CE (build-side addition): sums[g] += loss_rows over g. */
int dbmm_group_loss_sum(const float* loss_rows, const int64_t* g, float* sums, int64_t B,
                        int64_t G, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DBMM_H */
