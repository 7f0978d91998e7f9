// Library options (include/dbmm.h: dbmm_set_option / dbmm_get_option).
//
// Every switch the kernels' launchers consult lives in ONE table, read with dbmm_opt(id): no getenv on a launch path.
// A C caller sets an option by name; for developer A/B runs each option is seeded ONCE, when the library is loaded, from
// the environment variable DBMM_<NAME IN UPPER CASE> if that is set.  Defaults are the measured best; no option changes
// results beyond fp32 rounding (DESIGN.md "Switches").
#include <atomic>
#include <ctype.h>
#include <stdlib.h>
#include <string.h>
#include "common.h"

namespace {

struct OptDef { const char* name; int def; };
// order = enum DbmmOpt (common.h)
const OptDef kOpts[DBMM_OPT_COUNT] = {
    {"igemm_epi_direct", 1},   // fp16-pair kernels: epilogue straight from the accumulator layout (0: staged through LDS)
    {"igemm_fast", 1},         // buffer-load operand loader (0: guarded global loads)
    {"igemm_streamk", 1},      // 0 never / 1 where a grid cannot fill the resident slots / 2 always
    {"igemm_x3", 1},           // bf16-triple kernels when bf16 planes are given
    {"igemm_x2", 1},           // fp16-pair kernels when fp16 planes + a_absmax are given
    {"igemm_x2_bk", 32},       // K depth of the fp16-pair chunks (16 | 32)
    {"igemm_bk", 0},           // 32: force the 64-KB fp32-MFMA tiles
    {"igemm_halo", 1},         // stride-1 3x3 convs on the halo kernel (0: per-tap kernel)
    {"igemm_halo_pool", 2},    // pooled 3x3 convs on the window-major halo kernel: 0 never / 1 where Cout % 256 != 0 / 2 all
    {"igemm_bn256", 1},        // 128x256 tiles for wide GEMMs
    {"igemm_bn256_kxk", 1},    // ... and for pooled KxK convs that are not on the halo kernel
    {"gemm_8ph", 1},           // parity GEMMs on gemm_pair_8ph_kernel: 0 never / 1 where it measured ahead / 2 wherever it applies
    {"f16_8ph", 1},            // fp16 GEMMs on the deep-pipelined 256x256 kernel
    {"f16_bn256", 1},          // fp16 GEMMs: 128x256 tiles for wide N
    {"stem_mfma", 1},          // stride-2 stem conv on the MFMA gather kernel (0: FMA kernels)
    {"mha_valu", 0},           // 1: lane-per-query attention kernel instead of the MFMA one
    {"conv_patch", 1},         // (host wrappers) 32-channel stem convs on the persistent patch kernel
    {"mha_x2", 1},             // (host wrappers) parity attention core on fp16-pair products
    {"adapter_step_fused", 1}, // adapter forward / backward on the purpose-built kernels of adapter_step.hip (0: the general GEMM kernel)
    {"conv1x1_stream", 1},     // fp16 mode 1x1 convs: 0 the GEMM kernels / 1 the streaming kernel for HBM-bound shapes (incl. conv3 + residual with K <= 256) /
                               // 2 wherever it applies / 3 round 3's rule (without that exception; same-box fp16 RN50: 52.57 k -> 53.05 k img/s for 1)
    {"conv1x1_8ph", 1},        // parity 1x1 convs on gemm_pair_8ph_kernel: 0 never / 1 where it measured ahead / 2 wherever it applies
    {"chain8", 0},             // 1: conv3 + residual -> next conv1 of the layer-3 geometry (K = P = 256) on the eight-wave chain kernel (measured
                               //    0.754 ms against 0.686 ms for the two launches: bottleneck_chain8.hip; kept, tested, off)
    {"conv1x1_bn256", 0},      // 1: parity 1x1 convs with Cout % 256 == 0 on 128 x 256 tiles (A read once per 256 columns)
    {"tail_split", 1},         // eight-phase kernels whose 256 x 256 tiles leave a short last round on the 256 CUs: 0 one launch / 1 by rule: the round's
                               // tiles cut along K over the idle CUs + a summing launch (parity kernels; fp16 GEMMs with K >= 3072), or its rows on the
                               // 128 x 128 kernel (fp16 GEMMs of at most two rounds) / 2 the K cut wherever it applies / 3 (fp16 GEMMs) the row split only
    {"halo8", 1},              // parity 3x3 convs with Cout % 256 == 0 on conv3x3_halo8_kernel: 0 never / 1 where it measured ahead / 2 wherever it applies
    {"dual_8ph", 1},           // conv3 + downsample dual-source GEMM on gemm_pair_8ph_kernel: 0 never / 1 where it measured ahead / 2 wherever it applies
    {"mha_short", 1},          // attention cores: sequences of at most 64 tokens on two-wave workgroups (0: the four-wave ones, half of them idle)
    {"f16_conv_8ph", 1},       // fp16 mode 3x3 convs with Cout % 256 == 0 on conv3x3_f16_8ph_kernel (0: conv3x3_f16_kernel)
    {"conv1x1_res_stream", 1}, // conv3 + residual with K = 256 and >= 131,072 rows on conv1x1_res_stream_kernel (0: the 128 x 128 tiles)
};

std::atomic<int> g_val[DBMM_OPT_COUNT];

struct Seed {
    Seed() {
        for (int i = 0; i < DBMM_OPT_COUNT; ++i) {
            char env[64] = "DBMM_";
            size_t n = strlen(env);
            for (const char* c = kOpts[i].name; *c && n + 1 < sizeof(env); ++c) env[n++] = (char)toupper((unsigned char)*c);
            env[n] = 0;
            const char* e = getenv(env);
            g_val[i].store(e ? atoi(e) : kOpts[i].def, std::memory_order_relaxed);
        }
    }
} g_seed;

int find(const char* name) {
    if (!name) return -1;
    for (int i = 0; i < DBMM_OPT_COUNT; ++i)
        if (!strcmp(name, kOpts[i].name)) return i;
    return -1;
}

}  // namespace

int dbmm_opt(int id) { return g_val[id].load(std::memory_order_relaxed); }

extern "C" int dbmm_set_option(const char* name, int value) {
    const int i = find(name);
    if (i < 0) return DBMM_E_ARG;
    g_val[i].store(value, std::memory_order_relaxed);
    return DBMM_OK;
}

extern "C" int dbmm_get_option(const char* name, int* value) {
    const int i = find(name);
    if (i < 0 || !value) return DBMM_E_ARG;
    *value = g_val[i].load(std::memory_order_relaxed);
    return DBMM_OK;
}
