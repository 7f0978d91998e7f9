// Transformer-side kernels of CLIP's ViT and text towers that are not GEMMs: LayerNorm,
// the attention core, token-embedding gather, patch im2col, class-token assembly and the
// EOT-row gather.  All fp32; the projections (QKV / out / MLP) go through
// dbmm_gemm_bias_act with bias / QuickGELU / residual epilogues.
#include <stdlib.h>

#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------
// LayerNorm: one wave per row, float4 lanes, two-pass statistics in fp32 (biased variance),
// exactly the reference's LayerNorm subclass (fp32 compute) -- HBM-bound: 1 read + 1 write
// per element, re-reads hit L1/L2.
// ---------------------------------------------------------------------------------------
// NV = float4s per lane held in registers (E <= 256 * NV): the row is read from memory once.
// NV = 0: generic fallback that re-reads the row (L1/L2 hits) for any E.
template <int NV>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, long long ldx,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y,
                                                        long long ldy, int rows, int E4, float eps,
                                                        float* __restrict__ y_absmax) {
    const int lane = threadIdx.x & 63;
    const float invE = 1.f / (float)(E4 * 4);
    float amax = 0.f;
    // grid-stride over groups of 4 rows: the grid is capped so that the single-address atomic of
    // the output maximum is issued once per workgroup and a few hundred times per launch (one
    // per wave and row was 25,600 same-address atomics = 3x the kernel's own run time)
    for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += gridDim.x * 4) {
    const f32x4* xr = (const f32x4*)(x + (long long)row * ldx);
    f32x4* yr = (f32x4*)(y + (long long)row * ldy);
    if constexpr (NV > 0) {
        f32x4 v[NV];
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = lane + 64 * j;
            v[j] = i < E4 ? xr[i] : (f32x4){0.f, 0.f, 0.f, 0.f};
            s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        }
        const float mean = wave_sum(s) * invE;
        float q = 0.f;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (lane + 64 * j < E4) {
                const f32x4 d = v[j] - mean;
                q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
            }
        }
        const float rstd = rsqrtf(wave_sum(q) * invE + eps);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int i = lane + 64 * j;
            if (i < E4) {
                const f32x4 o = (v[j] - mean) * rstd * ((const f32x4*)gamma)[i] + ((const f32x4*)beta)[i];
                yr[i] = o;
                amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o[0]), fabsf(o[1]))), fmaxf(fabsf(o[2]), fabsf(o[3])));
            }
        }
    } else {
        float s = 0.f;
        for (int i = lane; i < E4; i += 64) { const f32x4 v = xr[i]; s += (v[0] + v[1]) + (v[2] + v[3]); }
        const float mean = wave_sum(s) * invE;
        float q = 0.f;
        for (int i = lane; i < E4; i += 64) {
            const f32x4 v = xr[i] - mean;
            q += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
        }
        const float rstd = rsqrtf(wave_sum(q) * invE + eps);
        for (int i = lane; i < E4; i += 64) {
            const f32x4 v = (xr[i] - mean) * rstd * ((const f32x4*)gamma)[i] + ((const f32x4*)beta)[i];
            yr[i] = v;
            amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
        }
    }
    }
    if (y_absmax) {          // scale source of the fp16-pair GEMM that consumes y
        __shared__ float wmax[4];
        amax = wave_max(amax);
        if (lane == 0) wmax[threadIdx.x >> 6] = amax;
        __syncthreads();
        if (threadIdx.x == 0) {
            amax = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
            if (amax > *(volatile const float*)y_absmax) atomicMax((unsigned*)y_absmax, __float_as_uint(amax));
        }
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

// ---------------------------------------------------------------------------------------
// Attention core on the fp32-input matrix cores (v_mfma_f32_32x32x2_f32), head_dim 64.
// One wave per (64-query tile, head, image), key blocks of 64, online softmax.
// Both products are computed TRANSPOSED so that a query is an MFMA *column* = a lane:
//   S^T[key][q] = sum_d K[key][d] Q[q][d]         A = K rows (LDS), B = Q (registers, loop-invariant)
//   O^T[d][q]  += sum_key V[key][d] P^T[key][q]   A = V^T (LDS), B = P^T
// In the 32x32 C layout lane l holds column l & 31 and rows (r & 3) + 8 (r >> 2) + 4 (l >> 5):
// the softmax statistics of a query are a reduction over the lane's own registers plus one
// exchange with lane l ^ 32, the rescale of O^T is lane-local, and P^T is already in the B-operand
// layout of the second product (k index = l >> 5 selects which of the step's two keys this lane
// half supplies; the K order of that product is simply chosen to match: step r uses keys
// (r & 3) + 8 (r >> 2) and that + 4).  The d order of the first product is permuted the same way
// for K and Q so that one 16-B LDS read feeds four MFMA k-steps (as in the igemm).
// K rows are stored with the 16-B chunk index XOR (row & 15): the 16 lanes of a ds_read_b128
// group read the same chunk of 16 different rows.
// ---------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

// NW = 1: one wave per workgroup, its own K / V tiles (short sequences: one query tile per head).
// NW = 4: four query tiles of the same (head, image) share each K / V block in LDS; the next block's
// global loads (8 float4 per thread) are issued before the current block's MFMAs and stored after
// the barrier, so the fetch overlaps the compute (long sequences: ViT-L/14, L = 577).
template <int NW>
__global__ __launch_bounds__(64 * NW) void mha_mfma_kernel(const float* __restrict__ qkv, float* __restrict__ out,
                                                           int L, int E, int causal) {
    __shared__ __attribute__((aligned(16))) float Ks[64 * 64];   // K block (first: the Q tile), swizzled chunks
    __shared__ __attribute__((aligned(16))) float Vs[64 * 64];   // V block, plain row-major
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, h = blockIdx.y, b = blockIdx.z;
    const int q0 = (blockIdx.x * NW + wv) * 64;       // may be >= L for the last waves of a row of tiles
    const int fr = lane & 31, fh = lane >> 5;
    const long long ld = 3LL * E;
    const float* base = qkv + (long long)b * L * ld + h * 64;

    // Q tile -> LDS (swizzled) -> B-operand fragments in registers, pre-scaled by head_dim^-0.5
    // (NW = 4: the waves take turns with the K buffer, two at a time via Ks and Vs)
    f32x4 qf[2][8];
#pragma unroll
    for (int turn = 0; turn < (NW + 1) / 2; ++turn) {
        float* Qs = (wv & 1) ? Vs : Ks;
        if ((wv >> 1) == turn) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int idx = i * 64 + lane, r = idx >> 4, c4 = idx & 15;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (q0 + r < L) v = *(const f32x4*)(base + (long long)(q0 + r) * ld + c4 * 4);
                *(f32x4*)(Qs + r * 64 + ((c4 ^ (r & 15)) << 2)) = v * 0.125f;
            }
        }
        __syncthreads();
        if ((wv >> 1) == turn) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int r = 32 * t + fr;
                    qf[t][j] = *(const f32x4*)(Qs + r * 64 + (((2 * j + fh) ^ (r & 15)) << 2));
                }
        }
        __syncthreads();
    }

    f32x16 o[2][2];                       // O^T tiles [d tile][query tile]
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[a][c][r] = 0.f;
    float m_run[2] = {-INFINITY, -INFINITY}, l_run[2] = {0.f, 0.f};   // per query tile; l is this half's partial sum

    // keys beyond the last query of the workgroup's last tile are masked for every wave
    const int kend = causal ? min(L, (int)(blockIdx.x * NW + NW) * 64) : L;
    constexpr int LPT = 16 / NW;          // float4 of K (and of V) per thread per block
    f32x4 kpre[LPT], vpre[LPT];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int idx = i * 64 * NW + (int)threadIdx.x, r = idx >> 4, c4 = idx & 15;
            kpre[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; vpre[i] = kpre[i];
            if (k0 + r < L && k0 < kend) {
                kpre[i] = *(const f32x4*)(base + (long long)(k0 + r) * ld + E + c4 * 4);
                vpre[i] = *(const f32x4*)(base + (long long)(k0 + r) * ld + 2 * E + c4 * 4);
            }
        }
    };
    if (NW > 1) fetch(0);
    for (int k0 = 0; k0 < kend; k0 += 64) {
        __syncthreads();                  // previous block's LDS reads are done
        if (NW == 1) fetch(k0);           // one wave per workgroup: no registers to spare for a prefetch
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int idx = i * 64 * NW + (int)threadIdx.x, r = idx >> 4, c4 = idx & 15;
            *(f32x4*)(Ks + r * 64 + ((c4 ^ (r & 15)) << 2)) = kpre[i];
            *(f32x4*)(Vs + r * 64 + c4 * 4) = vpre[i];
        }
        __syncthreads();
        if (NW > 1) fetch(k0 + 64);       // in flight during this block's MFMAs

        // S^T = K Q^T
        f32x16 sT[2][2];                  // [key tile][query tile]
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) sT[a][c][r] = 0.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int kr = 32 * mt + fr;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const f32x4 kf = *(const f32x4*)(Ks + kr * 64 + (((2 * j + fh) ^ (kr & 15)) << 2));
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    sT[mt][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[0][j][e], sT[mt][0], 0, 0, 0);
                    sT[mt][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], qf[1][j][e], sT[mt][1], 0, 0, 0);
                }
            }
        }

        // masks + online softmax, one query per lane column
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int qg = q0 + 32 * nt + fr;
            float bm = -INFINITY;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kg = k0 + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    const bool ok = kg < L && (!causal || kg <= qg);
                    sT[mt][nt][r] = ok ? sT[mt][nt][r] : -INFINITY;
                    bm = fmaxf(bm, sT[mt][nt][r]);
                }
            bm = fmaxf(bm, __shfl_xor(bm, 32));
            const float mn = fmaxf(m_run[nt], bm);
            // (mn is finite: key 0 is valid for every query in the first block)
            const float sc = expf(m_run[nt] - mn);
            m_run[nt] = mn;
            float ps = 0.f;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float pv = expf(sT[mt][nt][r] - mn);
                    sT[mt][nt][r] = pv;
                    ps += pv;
                }
            l_run[nt] = l_run[nt] * sc + ps;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[dt][nt][r] *= sc;
        }

        // O^T += V^T P^T : step r of key tile mt consumes keys 32 mt + (r & 3) + 8 (r >> 2) (+ 4 for the upper lane half)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int key = 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * fh;
                const float v0 = Vs[key * 64 + fr], v1 = Vs[key * 64 + 32 + fr];
                o[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, sT[mt][0][r], o[0][0], 0, 0, 0);
                o[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v0, sT[mt][1][r], o[0][1], 0, 0, 0);
                o[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, sT[mt][0][r], o[1][0], 0, 0, 0);
                o[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(v1, sT[mt][1][r], o[1][1], 0, 0, 0);
            }
    }

#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int qg = q0 + 32 * nt + fr;
        const float inv = 1.f / (l_run[nt] + __shfl_xor(l_run[nt], 32));
        if (qg < L) {
            float* op = out + ((long long)b * L + qg) * E + h * 64;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 v = {o[dt][nt][4 * g] * inv, o[dt][nt][4 * g + 1] * inv, o[dt][nt][4 * g + 2] * inv,
                                     o[dt][nt][4 * g + 3] * inv};
                    *(f32x4*)(op + 32 * dt + 8 * g + 4 * fh) = v;
                }
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

// walks the IMAGE: a thread takes VEC consecutive pixels of one image row (consecutive threads = consecutive vectors of that row;
// P % VEC == 0 keeps a vector inside one patch row) and writes them at (patch, channel, kh, kw) -- coalesced reads, VEC-float stores
template <int VEC>
__global__ __launch_bounds__(256) void im2col_patch_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                           int R, int P, int g, long long n_vec,
                                                           float* __restrict__ out_absmax) {
    const int K = 3 * P * P, RV = R / VEC;
    float amax = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec;
         i += (long long)gridDim.x * blockDim.x) {
        const int xv = (int)(i % RV);
        const long long row = i / RV;                                     // (b * 3 + c) * R + y
        const int y = (int)(row % R), c = (int)((row / R) % 3);
        const long long b = row / (3LL * R);
        const int x0 = xv * VEC, gx = x0 / P, kw = x0 - gx * P, gy = y / P, kh = y - gy * P;
        typedef float vec_t __attribute__((ext_vector_type(VEC)));        // both sides are VEC-element aligned: R, P and K are multiples of VEC
        const vec_t v = *(const vec_t*)(x + row * R + x0);
        *(vec_t*)(out + ((b * g + gy) * g + gx) * (long long)K + (c * P + kh) * P + kw) = v;
#pragma unroll
        for (int j = 0; j < VEC; ++j) amax = fmaxf(amax, fabsf(v[j]));
    }
    if (out_absmax) {          // scale source of the fp16-pair patch GEMM: one filtered atomic per workgroup
        __shared__ float wmax[4];
        amax = wave_max(amax);
        if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = amax;
        __syncthreads();
        if (threadIdx.x == 0) {
            amax = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
            if (amax > *(volatile const float*)out_absmax) atomicMax((unsigned*)out_absmax, __float_as_uint(amax));
        }
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
    const long long groups = (rows + 3) / 4;
    const dim3 grid((unsigned)(groups < 2048 ? groups : 2048));
    const int E4 = (int)(E / 4);
#define DBMM_LN(NV)                                                                                                  \
    hipLaunchKernelGGL(layernorm_kernel<NV>, grid, dim3(256), 0, (hipStream_t)stream, x, (long long)ldx, gamma, beta, y, \
                       (long long)ldy, (int)rows, E4, eps, y_absmax)
    if (E4 <= 64) DBMM_LN(1);
    else if (E4 <= 128) DBMM_LN(2);
    else if (E4 <= 192) DBMM_LN(3);
    else if (E4 <= 256) DBMM_LN(4);
    else if (E4 <= 512) DBMM_LN(8);
    else DBMM_LN(0);
#undef DBMM_LN
    DBMM_CHECK_LAUNCH();
    return DBMM_OK;
}

extern "C" int dbmm_mha_core(const float* qkv, float* out, int64_t B, int64_t L, int64_t E, int64_t heads,
                             int causal, void* stream) {
    if (!qkv || !out) return DBMM_E_ARG;
    if (B <= 0 || L <= 0 || heads <= 0 || E != heads * 64 || B > 65535 || heads > 65535) return DBMM_E_SHAPE;
    if (!dbmm_aligned16(qkv) || !dbmm_aligned16(out)) return DBMM_E_ALIGN;
    // matrix-core kernel by default; option mha_valu = 1 selects the lane-per-query VALU kernel (ablation)
    const int valu = dbmm_opt(OPT_MHA_VALU);
    const dim3 grid((unsigned)((L + 63) / 64), (unsigned)heads, (unsigned)B);
    const int qt = (int)((L + 63) / 64);
    if (valu)
        hipLaunchKernelGGL(mha_core_kernel, grid, dim3(64), 0, (hipStream_t)stream, qkv, out, (int)L, (int)E, causal);
    else if (qt >= 4)      // long sequences: four query tiles share the K / V blocks
        hipLaunchKernelGGL(mha_mfma_kernel<4>, dim3((unsigned)((qt + 3) / 4), (unsigned)heads, (unsigned)B), dim3(256), 0,
                           (hipStream_t)stream, qkv, out, (int)L, (int)E, causal);
    else
        hipLaunchKernelGGL(mha_mfma_kernel<1>, grid, dim3(64), 0, (hipStream_t)stream, qkv, out, (int)L, (int)E, causal);
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

extern "C" int dbmm_im2col_patch(const float* x_nchw, float* out, float* out_absmax, int64_t B, int64_t R, int64_t P,
                                 void* stream) {
    if (!x_nchw || !out) return DBMM_E_ARG;
    if (B <= 0 || R <= 0 || P <= 0 || R % P) return DBMM_E_SHAPE;
    const int64_t g = R / P;
    const int vec = (P % 4 == 0) ? 4 : ((P % 2 == 0) ? 2 : 1);
    const long long n_vec = (long long)B * 3 * R * (R / vec);
    if (vec == 4)
        hipLaunchKernelGGL(im2col_patch_kernel<4>, dim3(grid_for(n_vec)), dim3(256), 0, (hipStream_t)stream, x_nchw, out, (int)R, (int)P,
                           (int)g, n_vec, out_absmax);
    else if (vec == 2)
        hipLaunchKernelGGL(im2col_patch_kernel<2>, dim3(grid_for(n_vec)), dim3(256), 0, (hipStream_t)stream, x_nchw, out, (int)R, (int)P,
                           (int)g, n_vec, out_absmax);
    else
        hipLaunchKernelGGL(im2col_patch_kernel<1>, dim3(grid_for(n_vec)), dim3(256), 0, (hipStream_t)stream, x_nchw, out, (int)R, (int)P,
                           (int)g, n_vec, out_absmax);
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
