"""ctypes binding of libdbmm_hip.so (the C ABI declared in include/dbmm.h).

The product path has no CPU fallback: if the library is missing or a call fails, an
exception is raised.  `build()` compiles the HIP sources in-tree with hipcc for gfx950
(cross-compiles without a GPU).
"""
import ctypes
import glob
import os
import subprocess
from ctypes import c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("DBMM_LIB") or os.path.join(_HERE, "libdbmm_hip.so")   # DBMM_LIB: developer override
INCLUDE = os.path.join(os.path.dirname(_HERE), "include")

_lib = None


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


# which sources define a kernel family (rocprofv3 / ops profile tag prefix -> files under csrc/): profiles/r0N_pmc.json
# stamps every kernel's counters with kernel_source_hash(), bench.py drops counters whose stamp no longer matches
KERNEL_SOURCES = (
    ("igemm_", ("igemm_f32.hip", "igemm_epilogue.inc", "common.h")),
    ("bottleneck_chain_kernel", ("bottleneck_chain.hip", "common.h")),
    ("bottleneck_chain8_kernel", ("bottleneck_chain8.hip", "common.h")),
    ("conv3x3_c32_kernel", ("conv_patch.hip", "common.h")),
    ("gemm_pair_8ph_kernel", ("gemm_pair_8ph.hip", "common.h")),
    ("conv3x3_halo8", ("conv3x3_halo8.hip", "conv3x3_halo8_epilogue.inc", "common.h")),
    ("mha_pair_kernel", ("mha_pair.hip", "common.h")),
    ("gemm_f16", ("f16_ops.hip", "common.h")), ("mha_f16", ("f16_ops.hip", "common.h")), ("layernorm_f16", ("f16_ops.hip", "common.h")),
    ("chain_f16", ("chain_f16.hip", "common.h")),
    ("conv1x1_res_stream", ("conv1x1_res_stream.hip", "common.h")),
    ("conv1x1_res_stream_f16", ("conv1x1_res_stream_f16.hip", "common.h")),
    ("conv3x3_f16", ("conv_f16.hip", "common.h")), ("conv1x1_f16", ("conv_f16.hip", "common.h")), ("stem_s2_f16", ("conv_f16.hip", "common.h")),
    ("avgpool2_f16", ("conv_f16.hip", "common.h")),
    ("stem_s2", ("resnet_ops.hip", "common.h")), ("attnpool", ("resnet_ops.hip", "common.h")),
    ("adapter_step", ("adapter_step.hip", "common.h")),
)


def kernel_source_hash(kernel_name):
    """sha256 (16 hex digits) of the sources that define `kernel_name`'s family; None for an unknown family"""
    import hashlib
    for prefix, files in KERNEL_SOURCES:
        if kernel_name.startswith(prefix):
            h = hashlib.sha256()
            for f in files:
                path = os.path.join(CSRC, f)
                if not os.path.exists(path):
                    return None
                h.update(open(path, "rb").read())
            return h.hexdigest()[:16]
    return None


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> libdbmm_hip.so (in-tree, so it travels with the repo).  Incremental: a source
    is recompiled when it, a shared header or include/dbmm.h is newer than its object; `force` recompiles all.
    Returns the library path; `build.last` lists the sources compiled by the most recent call."""
    srcs = sources()
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) + [os.path.join(INCLUDE, "dbmm.h")]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(os.path.join(_HERE, "build"), exist_ok=True)
    objs, procs = [], []
    newest_hdr = max(os.path.getmtime(h) for h in hdrs)
    for s in srcs:
        o = os.path.join(_HERE, "build", os.path.basename(s) + ".o")
        objs.append(o)
        if not force and os.path.exists(o) and os.path.getmtime(o) >= max(os.path.getmtime(s), newest_hdr):
            continue
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", INCLUDE, "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((s, cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for s, cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            errs = [l for l in out.decode().splitlines() if "error" in l][:8]
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), "\n".join(errs) or out.decode()[-2000:]))
    build.last = [os.path.basename(s) for s, _, _ in procs]
    if procs or not os.path.exists(LIB_PATH) or any(os.path.getmtime(LIB_PATH) < os.path.getmtime(o) for o in objs):
        subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs, check=True)
    return LIB_PATH


build.last = []


_F, _I, _L, _P, _Z = c_float, c_int, c_int64, c_void_p, c_size_t

# name -> argtypes (restype is int unless listed in _RESTYPES)
_SIGS = {
    "dbmm_set_option": [ctypes.c_char_p, _I],
    "dbmm_get_option": [ctypes.c_char_p, ctypes.POINTER(ctypes.c_int)],
    "dbmm_version": [],
    "dbmm_error_string": [_I],
    "dbmm_conv_bn_act": [_P, _P, _P, _P, _P, _L, _L, _L, _L, _L, _L, _L, _L, _L, _I, _P],
    "dbmm_conv1x1_bn_act": [_P, _P, _P, _P, _P, _L, _L, _L, _L, _L, _I, _P],
    "dbmm_conv3x3_bn_act": [_P, _P, _P, _P, _P, _L, _L, _L, _L, _L, _I, _P],
    "dbmm_gemm_bias_act": [_P, _L, _I, _P, _L, _I, _P, _P, _L, _P, _L, _L, _L, _L, _F, _I, _P],
    "dbmm_workspace_bytes_igemm": [],
    "dbmm_debug_last_igemm": [_P],
    "dbmm_conv_bn_act_ws": [_P, _P, _P, _P, _P, _L, _L, _L, _L, _L, _L, _L, _L, _L, _I, _I, _P, _Z, _P],
    "dbmm_gemm_bias_act_ws": [_P, _L, _I, _P, _L, _I, _P, _P, _L, _P, _L, _L, _L, _L, _F, _I, _P, _Z, _P],
    "dbmm_split_planes_bytes": [_L, _L],
    "dbmm_split_weight_planes": [_P, _P, _L, _L, _P],
    "dbmm_conv_bn_act_x3": [_P, _P, _P, _P, _P, _P, _L, _L, _L, _L, _L, _L, _L, _L, _L, _I, _I, _P, _Z, _P],
    "dbmm_split_planes_f16_bytes": [_L, _L],
    "dbmm_split_weight_planes_f16": [_P, _P, _L, _L, _I, _P],
    "dbmm_conv_bn_act_x2": [_P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P, _L, _L, _L, _L, _L, _L, _L, _L, _L, _I, _I, _I, _P, _Z, _P],
    "dbmm_gemm_bias_act_x2": [_P, _L, _P, _P, _P, _I, _I, _L, _P, _P, _P, _L, _P, _L, _P, _L, _L, _L, _F, _I, _P, _Z, _P],
    "dbmm_gemm_dual_bn_act_x2": [_P, _L, _P, _P, _I, _L, _L, _P, _P, _L, _P, _P, _L, _L, _P, _P, _P, _L, _P, _L, _L, _I, _P, _Z, _P],
    "dbmm_bottleneck_chain_x2": [_P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _L, _L, _L, _L, _L, _L, _P],
    "dbmm_bottleneck_chain_dual_x2": [_P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _L, _L, _L, _L, _L, _P],
    "dbmm_gemm_f16": [_P, _L, _P, _L, _P, _P, _L, _P, _L, _L, _L, _L, _I, _P],
    "dbmm_gemm_f16_ws": [_P, _L, _P, _L, _P, _P, _L, _P, _L, _L, _L, _L, _I, _P, _Z, _P],
    "dbmm_mha_core_f16": [_P, _P, _L, _L, _L, _L, _I, _P],
    "dbmm_layernorm_f16": [_P, _L, _P, _P, _P, _L, _L, _L, _F, _P],
    "dbmm_conv1x1_bn_act_f16": [_P, _P, _P, _P, _P, _P, _L, _L, _L, _I, _P],
    "dbmm_conv1x1_bn_act_f16_ws": [_P, _P, _P, _P, _P, _P, _L, _L, _L, _I, _P, _Z, _P],
    "dbmm_conv1x1_dual_bn_act_f16": [_P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _L, _L, _I, _P],
    "dbmm_bottleneck_chain_f16": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _L, _L, _P],
    "dbmm_bottleneck_chain_pool_f16": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _L, _L, _L, _L, _P],
    "dbmm_bottleneck_chain_dual_f16": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _L, _L, _L, _P],
    "dbmm_conv3x3_bn_relu_f16": [_P, _P, _P, _P, _P, _L, _L, _L, _L, _L, _I, _P],
    "dbmm_conv_stem_s2_f16": [_P, _I, _P, _P, _P, _L, _L, _L, _L, _P],
    "dbmm_conv1x1_res_pool_f16": [_P, _P, _P, _P, _P, _P, _P, _L, _L, _L, _L, _L, _P],
    "dbmm_conv_stem_s2_bn_f16": [_P, _I, _P, _P, _P, _P, _L, _L, _L, _L, _P],
    "dbmm_avgpool2_f16": [_P, _P, _L, _L, _L, _L, _P],
    "dbmm_im2col_patch_f16": [_P, _I, _P, _L, _L, _L, _L, _P],
    "dbmm_vit_tokens_f16": [_P, _P, _P, _P, _L, _L, _L, _P],
    "dbmm_embed_gather_f16": [_P, _P, _P, _P, _L, _L, _L, _L, _P],
    "dbmm_gather_eot_f16": [_P, _P, _P, _L, _L, _L, _P],
    "dbmm_cast_f32_f16": [_P, _P, _L, _P],
    "dbmm_conv3x3_c32_bn_relu_x2": [_P, _P, _P, _I, _P, _P, _P, _P, _L, _L, _L, _L, _L, _I, _P],
    "dbmm_bottleneck_block_chain_x2": [_P, _P, _P, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P,
                                       _L, _L, _L, _L, _L, _L, _P],
    "dbmm_gemm_bias_act_x3": [_P, _L, _P, _P, _L, _P, _P, _L, _P, _L, _L, _L, _L, _F, _I, _P, _Z, _P],
    "dbmm_gemm_batched": [_P, _L, _L, _I, _P, _L, _L, _I, _P, _L, _P, _L, _L, _L, _L, _L, _L, _F, _I, _P],
    "dbmm_gemm_pair_8ph": [_P, _L, _P, _P, _I, _L, _P, _P, _P, _L, _P, _L, _P, _L, _L, _L, _F, _I, _P],
    "dbmm_conv_stem_s2": [_P, _P, _P, _P, _P, _L, _L, _L, _L, _P],
    "dbmm_avgpool2d": [_P, _P, _L, _L, _L, _L, _L, _P],
    "dbmm_workspace_bytes_attnpool": [_L, _L, _L],
    "dbmm_attnpool": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _L, _L, _L, _P, _Z, _P],
    "dbmm_attnpool_x": [_P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _L, _L, _L, _L, _L, _P, _Z, _P],
    "dbmm_layernorm": [_P, _L, _P, _P, _P, _L, _L, _L, _F, _P, _P],
    "dbmm_mha_core": [_P, _P, _L, _L, _L, _L, _I, _P],
    "dbmm_mha_core_x2": [_P, _P, _P, _L, _L, _L, _L, _I, _P],
    "dbmm_embed_gather": [_P, _P, _P, _P, _L, _L, _L, _L, _P],
    "dbmm_im2col_patch": [_P, _P, _P, _L, _L, _L, _P],
    "dbmm_vit_tokens": [_P, _P, _P, _P, _L, _L, _L, _P],
    "dbmm_gather_eot": [_P, _P, _P, _L, _L, _L, _P],
    "dbmm_bn1d_stats": [_P, _L, _L, _F, _F, _P, _P, _P, _P, _P, _P],
    "dbmm_bn1d_relu": [_P, _P, _P, _P, _P, _P, _L, _L, _I, _F, _P],
    "dbmm_adapter_fwd": [_P] * 15 + [_L, _L, _L, _I, _F, _F, _P],
    "dbmm_workspace_bytes_adapter_bwd": [_L, _L, _L],
    "dbmm_adapter_bwd": [_P] * 15 + [_L, _L, _L, _P, _Z, _P],
    "dbmm_text_colnorm": [_P, _P, _L, _L, _P],
    "dbmm_l2norm_rows": [_P, _P, _L, _L, _P],
    "dbmm_colsum": [_P, _P, _L, _L, _P],
    "dbmm_l2norm_sim_ce_fwd": [_P, _P, _F, _P, _P, _F, _P, _P, _P, _P, _P, _L, _L, _L, _P],
    "dbmm_l2norm_sim_ce_bwd": [_P, _P, _F, _I, _P, _P, _P, _P, _F, _F, _P, _L, _L, _L, _P],
    "dbmm_sgd_momentum": [_L, _P, _P, _P, _P, _F, _F, _F, _I, _P],
    "dbmm_workspace_bytes_adapter_train_step": [_L, _L, _L, _I],
    "dbmm_adapter_train_step": [_P] * 26 + [_F, _P, _F, _F, _F, _F, _I, _P, _P, _P, _L, _L, _L, _L, _P, _Z, _P],
    "dbmm_workspace_bytes_preprocess": [_L, _L],
    "dbmm_resize_crop_normalize_u8": [_P, _L, _L, _P, _P, _L, _P, _P, _L, _L, _L, _L, _P, _P, _P, _P, _P, _Z, _P],
    "dbmm_resize_crop_normalize_u8_batch": [_P, _L, _L, _L, _P, _P, _L, _P, _P, _L, _L, _L, _L, _P, _P, _P, _P, _P, _Z, _P],
    "dbmm_gather_rows": [_P, _P, _P, _L, _L, _L, _P],
    "dbmm_group_count": [_P, _P, _P, _P, _L, _L, _L, _P],
    "dbmm_group_loss_sum": [_P, _P, _P, _L, _L, _P],
}
_RESTYPES = {
    "dbmm_error_string": ctypes.c_char_p,
    "dbmm_workspace_bytes_attnpool": c_size_t,
    "dbmm_workspace_bytes_preprocess": c_size_t,
    "dbmm_workspace_bytes_igemm": c_size_t,
    "dbmm_split_planes_bytes": c_size_t,
    "dbmm_split_planes_f16_bytes": c_size_t,
    "dbmm_debug_last_igemm": None,
    "dbmm_workspace_bytes_adapter_bwd": c_size_t,
    "dbmm_workspace_bytes_adapter_train_step": c_size_t,
}

EXPORTS = tuple(_SIGS)


def lib():
    """The loaded library; raises (never falls back) when it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback for the dbmm_amd hot path.")
        L = ctypes.CDLL(LIB_PATH)
        for name, argtypes in _SIGS.items():
            fn = getattr(L, name)          # AttributeError if the symbol is not exported
            fn.argtypes = argtypes
            fn.restype = _RESTYPES.get(name, c_int)
        _lib = L
    return _lib


class DbmmError(RuntimeError):
    pass


E_UNSUPPORTED = -5      # DBMM_E_UNSUPPORTED: nothing was launched, the caller composes the unfused calls
E_ALIGN = -2            # DBMM_E_ALIGN


def check(rc, what=""):
    if rc != 0:
        msg = lib().dbmm_error_string(rc).decode()
        raise DbmmError(f"{what or 'dbmm call'} failed with code {rc}: {msg}")


def ptr(t):
    """data pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream():
    import torch
    return torch.cuda.current_stream().cuda_stream


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise DbmmError("dbmm_amd kernels need CUDA/HIP tensors (MI355X); got a CPU tensor. "
                            "There is no CPU fallback on the product path.")
