// fp32 implicit-GEMM on the CDNA4 matrix cores (v_mfma_f32_32x32x2_f32, exact f32).
//
//   C[M][N] = act( alpha * (A[M][K] . W[N][K]^T + bias[N]) + R[M][N] )
//
// One kernel family serves every dense contraction of the hot path:
//   AMODE 0  A row-major [M][lda]              (1x1 convs on NHWC, nn.Linear inputs)
//   AMODE 1  A gathered from an NHWC image     (KHxKW conv, zero padding; K = (kh,kw,cin))
//   AMODE 2  A K-major [K][lda]                (weight gradients: reduction over the batch)
//   WMODE 0  W [N][ldw] (nn.Linear / packed conv weight)      WMODE 1  W [K][ldw]
//
// Tiling: 256 threads = 4 wave64; block tile BM x BN x 32; each wave owns TM x TN MFMA tiles
// of 32x32.  Global -> registers -> LDS (double buffered, one barrier per K chunk, next
// chunk's global loads issued before the current chunk's MFMAs).  LDS rows hold 32 floats
// (128 B) with the 16-B chunk index XOR-swizzled by (row>>1)&7, which makes the ds_read_b128
// operand fetches bank-conflict free (16 lanes x 16 B cover all 64 banks).
//
// K is consumed in a permuted order that is identical for A and W: per 8-wide group g the
// lane half h = lane>>5 reads k = 8g+4h .. 8g+4h+3 as one 16-B LDS read and MFMA step j uses
// element j of both halves (k pair {8g+j, 8g+4+j}).  Any fixed K order is a valid fp32 sum.
//
// Roofline: MFMA-bound. 32x32x2 = 4096 FLOP / 64 cycles / SIMD -> 157.3 TFLOP/s chip peak.
#include "common.h"

namespace {

struct IgemmP {
    const float* a;
    const float* w;
    const float* bias;
    const float* res;
    float* c;
    long long lda, ldw, ldr, ldc;
    int M, N, K;
    int H, W, Cin, Ho, Wo, KH, KW, stride, pad;  // AMODE 1 only
    int act;
    float alpha;
    int tiles_n, n_tiles;
};

constexpr int BK = 32;

__device__ __forceinline__ int lds_off(int row, int kchunk) {  // float index of a 16-B chunk
    return row * BK + ((kchunk ^ ((row >> 1) & 7)) << 2);
}

template <int BM, int BN, int WAVES_M, int WAVES_N, int AMODE, int WMODE>
__global__ __launch_bounds__(256) void igemm_f32_kernel(const IgemmP p) {
    constexpr int TM = BM / WAVES_M / 32, TN = BN / WAVES_N / 32;
    constexpr int ALD = BM / 32, WLD = BN / 32;  // float4 loads per thread per chunk
    static_assert(WAVES_M * WAVES_N == 4, "4 waves");
    __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * BK];
    float* As = lds;                 // [2][BM][32]
    float* Ws = lds + 2 * BM * BK;   // [2][BN][32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = xcd_remap(blockIdx.x, p.n_tiles);
    const int m0 = (tile / p.tiles_n) * BM, n0 = (tile % p.tiles_n) * BN;
    const int wm0 = (wave / WAVES_N) * (TM * 32), wn0 = (wave % WAVES_N) * (TN * 32);

    // ---- per-thread global-load bookkeeping -------------------------------------------
    // row-major / conv: thread covers 16-B chunk (tid&7) of rows (tid>>3) + 32*i
    const int lc = tid & 7, lr = tid >> 3;
    long long a_base[ALD];
    int a_hi0[ALD], a_wi0[ALD];
    if constexpr (AMODE == 0) {
#pragma unroll
        for (int i = 0; i < ALD; ++i) {
            const int m = m0 + lr + 32 * i;
            a_base[i] = (m < p.M) ? (long long)m * p.lda : -1;
        }
    } else if constexpr (AMODE == 1) {
#pragma unroll
        for (int i = 0; i < ALD; ++i) {
            const int m = m0 + lr + 32 * i;
            if (m < p.M) {
                const int hw = p.Ho * p.Wo;
                const int n = m / hw, rem = m - n * hw;
                const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
                a_base[i] = (long long)n * p.H * p.W * p.Cin;
                a_hi0[i] = ho * p.stride - p.pad;
                a_wi0[i] = wo * p.stride - p.pad;
            } else {
                a_base[i] = 0; a_hi0[i] = -(1 << 28); a_wi0[i] = 0;
            }
        }
    }
    // K-major operands: thread covers m-quad (tid % (BX/4)) of k rows tid/(BX/4) + step*i
    constexpr int AQ = BM / 4, WQ = BN / 4;

    f32x4 a_reg[ALD], w_reg[WLD];

    auto load_a = [&](int k0) {
        if constexpr (AMODE == 0) {
            const int k = k0 + lc * 4;
#pragma unroll
            for (int i = 0; i < ALD; ++i) {
                a_reg[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (a_base[i] >= 0 && k < p.K) a_reg[i] = *(const f32x4*)(p.a + a_base[i] + k);
            }
        } else if constexpr (AMODE == 1) {
            const int k = k0 + lc * 4;
            const int tap = k / p.Cin, ci = k - tap * p.Cin;
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
            const bool kin = k < p.K;
#pragma unroll
            for (int i = 0; i < ALD; ++i) {
                const int hi = a_hi0[i] + kh, wi = a_wi0[i] + kw;
                a_reg[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (kin && hi >= 0 && hi < p.H && wi >= 0 && wi < p.W)
                    a_reg[i] = *(const f32x4*)(p.a + a_base[i] + ((long long)hi * p.W + wi) * p.Cin + ci);
            }
        } else {
            const int mq = tid % AQ;
#pragma unroll
            for (int i = 0; i < ALD; ++i) {
                const int kk = tid / AQ + (256 / AQ) * i;
                const int k = k0 + kk, m = m0 + mq * 4;
                a_reg[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (k < p.K && m < p.M) a_reg[i] = *(const f32x4*)(p.a + (long long)k * p.lda + m);
            }
        }
    };
    auto load_w = [&](int k0) {
        if constexpr (WMODE == 0) {
            const int k = k0 + lc * 4;
#pragma unroll
            for (int i = 0; i < WLD; ++i) {
                const int n = n0 + lr + 32 * i;
                w_reg[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (n < p.N && k < p.K) w_reg[i] = *(const f32x4*)(p.w + (long long)n * p.ldw + k);
            }
        } else {
            const int nq = tid % WQ;
#pragma unroll
            for (int i = 0; i < WLD; ++i) {
                const int kk = tid / WQ + (256 / WQ) * i;
                const int k = k0 + kk, n = n0 + nq * 4;
                w_reg[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (k < p.K && n < p.N) w_reg[i] = *(const f32x4*)(p.w + (long long)k * p.ldw + n);
            }
        }
    };
    auto store_lds = [&](int buf) {
        float* Ab = As + buf * BM * BK;
        float* Wb = Ws + buf * BN * BK;
        if constexpr (AMODE != 2) {
#pragma unroll
            for (int i = 0; i < ALD; ++i) *(f32x4*)(Ab + lds_off(lr + 32 * i, lc)) = a_reg[i];
        } else {
            const int mq = tid % AQ;
#pragma unroll
            for (int i = 0; i < ALD; ++i) {
                const int kk = tid / AQ + (256 / AQ) * i;
#pragma unroll
                for (int j = 0; j < 4; ++j) Ab[lds_off(mq * 4 + j, kk >> 2) + (kk & 3)] = a_reg[i][j];
            }
        }
        if constexpr (WMODE == 0) {
#pragma unroll
            for (int i = 0; i < WLD; ++i) *(f32x4*)(Wb + lds_off(lr + 32 * i, lc)) = w_reg[i];
        } else {
            const int nq = tid % WQ;
#pragma unroll
            for (int i = 0; i < WLD; ++i) {
                const int kk = tid / WQ + (256 / WQ) * i;
#pragma unroll
                for (int j = 0; j < 4; ++j) Wb[lds_off(nq * 4 + j, kk >> 2) + (kk & 3)] = w_reg[i][j];
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int fr = lane & 31, fh = lane >> 5, fsw = (fr >> 1) & 7;
    const int nk = (p.K + BK - 1) / BK;

    load_a(0); load_w(0);
    store_lds(0);
    __syncthreads();

    for (int kc = 0; kc < nk; ++kc) {
        const int buf = kc & 1;
        if (kc + 1 < nk) { load_a((kc + 1) * BK); load_w((kc + 1) * BK); }
        const float* Ab = As + buf * BM * BK;
        const float* Wb = Ws + buf * BN * BK;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int ch = ((2 * g + fh) ^ fsw) << 2;
            f32x4 af[TM], wf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *(const f32x4*)(Ab + (wm0 + i * 32 + fr) * BK + ch);
#pragma unroll
            for (int j = 0; j < TN; ++j) wf[j] = *(const f32x4*)(Wb + (wn0 + j * 32 + fr) * BK + ch);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][s], wf[j][s], acc[i][j], 0, 0, 0);
        }
        if (kc + 1 < nk) store_lds(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ---------
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn0 + j * 32 + fr;
        if (n >= p.N) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                if (m >= p.M) continue;
                float v = (acc[i][j][r] + bv) * p.alpha;
                if (p.res) v += p.res[(long long)m * p.ldr + n];
                if (p.act == DBMM_ACT_RELU) v = fmaxf(v, 0.f);
                else if (p.act == DBMM_ACT_QUICKGELU) v = v / (1.f + expf(-1.702f * v));
                p.c[(long long)m * p.ldc + n] = v;
            }
        }
    }
}

template <int BM, int BN, int WM, int WN, int AMODE, int WMODE>
int launch_cfg(IgemmP& p, hipStream_t s) {
    const int tiles_m = (p.M + BM - 1) / BM;
    p.tiles_n = (p.N + BN - 1) / BN;
    p.n_tiles = tiles_m * p.tiles_n;
    hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM, WN, AMODE, WMODE>), dim3(p.n_tiles), dim3(256), 0, s, p);
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

template <int AMODE, int WMODE>
int launch_modes(IgemmP& p, hipStream_t s) {
    // tile choice: widest N tile that N fills; drop to 64x64 when the 128-wide grid would
    // leave most of the 256 CUs idle (small-M projections).
    const long long t128 = (long long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    if (p.N <= 32) return launch_cfg<128, 32, 4, 1, AMODE, WMODE>(p, s);
    if (p.N <= 64) return launch_cfg<128, 64, 2, 2, AMODE, WMODE>(p, s);
    if (t128 < 192) return launch_cfg<64, 64, 2, 2, AMODE, WMODE>(p, s);
    return launch_cfg<128, 128, 2, 2, AMODE, WMODE>(p, s);
}

}  // namespace

extern "C" int dbmm_gemm_bias_act(const float* a, int64_t lda, int trans_a, const float* w, int64_t ldw,
                                  int trans_w, const float* bias, const float* residual, int64_t ldr,
                                  float* c, int64_t ldc, int64_t M, int64_t N, int64_t K, float alpha,
                                  int act, void* stream) {
    if (!a || !w || !c) return DBMM_E_ARG;
    if (M <= 0 || N <= 0 || K <= 0 || M > INT32_MAX || N > INT32_MAX || K > INT32_MAX) return DBMM_E_SHAPE;
    if (act < 0 || act > 2) return DBMM_E_ARG;
    if ((lda & 3) || (ldw & 3)) return DBMM_E_ALIGN;
    if (!dbmm_aligned16(a) || !dbmm_aligned16(w)) return DBMM_E_ALIGN;
    if (!trans_a && (K & 3)) return DBMM_E_SHAPE;   // 16-B chunks along K
    if (trans_a && (M & 3)) return DBMM_E_SHAPE;    // 16-B chunks along M
    if (!trans_w && (K & 3)) return DBMM_E_SHAPE;
    if (trans_w && (N & 3)) return DBMM_E_SHAPE;
    IgemmP p{};
    p.a = a; p.w = w; p.bias = bias; p.res = residual; p.c = c;
    p.lda = lda; p.ldw = ldw; p.ldr = ldr; p.ldc = ldc;
    p.M = (int)M; p.N = (int)N; p.K = (int)K; p.act = act; p.alpha = alpha;
    hipStream_t s = (hipStream_t)stream;
    if (!trans_a && !trans_w) return launch_modes<0, 0>(p, s);
    if (!trans_a && trans_w) return launch_modes<0, 1>(p, s);
    if (trans_a && !trans_w) return launch_modes<2, 0>(p, s);
    return launch_modes<2, 1>(p, s);
}

extern "C" int dbmm_conv_bn_act(const float* x, const float* w, const float* bias, const float* residual,
                                float* y, int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                                int64_t KH, int64_t KW, int64_t stride, int64_t pad, int act, void* stream) {
    if (!x || !w || !y) return DBMM_E_ARG;
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0)
        return DBMM_E_SHAPE;
    if (Cin & 3) return DBMM_E_SHAPE;
    if (act < 0 || act > 2) return DBMM_E_ARG;
    if (!dbmm_aligned16(x) || !dbmm_aligned16(w)) return DBMM_E_ALIGN;
    const int64_t Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return DBMM_E_SHAPE;
    const int64_t M = B * Ho * Wo, K = KH * KW * Cin;
    if (M > INT32_MAX || K > INT32_MAX) return DBMM_E_SHAPE;
    IgemmP p{};
    p.a = x; p.w = w; p.bias = bias; p.res = residual; p.c = y;
    p.lda = Cin; p.ldw = K; p.ldr = Cout; p.ldc = Cout;
    p.M = (int)M; p.N = (int)Cout; p.K = (int)K; p.act = act; p.alpha = 1.f;
    p.H = (int)H; p.W = (int)W; p.Cin = (int)Cin; p.Ho = (int)Ho; p.Wo = (int)Wo;
    p.KH = (int)KH; p.KW = (int)KW; p.stride = (int)stride; p.pad = (int)pad;
    hipStream_t s = (hipStream_t)stream;
    if (KH == 1 && KW == 1 && stride == 1 && pad == 0) return launch_modes<0, 0>(p, s);  // plain GEMM
    return launch_modes<1, 0>(p, s);
}

extern "C" int dbmm_conv1x1_bn_act(const float* x, const float* w, const float* bias, const float* residual,
                                   float* y, int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                                   int act, void* stream) {
    return dbmm_conv_bn_act(x, w, bias, residual, y, B, H, W, Cin, Cout, 1, 1, 1, 0, act, stream);
}

extern "C" int dbmm_conv3x3_bn_act(const float* x, const float* w, const float* bias, const float* residual,
                                   float* y, int64_t B, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                                   int act, void* stream) {
    return dbmm_conv_bn_act(x, w, bias, residual, y, B, H, W, Cin, Cout, 3, 3, 1, 1, act, stream);
}
