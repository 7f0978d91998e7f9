#!/usr/bin/env python3
"""images/sec of one "embed + adapter step" (BASELINE.json metric) on N MI355X of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = CLIP-RN50 encode_image on this rank's B_l synthetic 224x224x3 images (resident in HBM) ->
RCCL all-gather of embeddings + labels -> replicated adapter forward + CE + backward + SGD on the
global batch.  N = 1: B = 1024, the configuration BASELINE.json's metric is quoted on ("CLIP-RN50 224px
bs=1024").  N >= 2: the same 1024 images per GPU (weak scaling: per-GPU work fixed; BASELINE configs[2], bs 1024 over two
GPUs, is --batch-per-gpu 512).

Arithmetic ("dtype": "f32"): activations and accumulators are fp32 in HBM / registers; each product runs
on the 16-bit matrix cores as fp16 hi + lo pair (22-bit mantissa, exact power-of-two scale per tensor) x
the checkpoint's fp16-exact weight, fp32 accumulate -- error vs fp64 equal to an fp32-input-MFMA kernel's
(DESIGN.md section 4).  The line also carries `fp32_input_mfma`: the same step with every product on
v_mfma_f32_32x32x2_f32 (plan option conv_split = off), measured in the same process.

Besides the contract line, rank 0 reports
  roofline      the igemm / chain / patch launches of every 4th timed step are bracketed by HIP events on their stream; kernels are
                grouped by instantiation (the names rocprofv3 prints) and ranked by MEASURED time share.  The
                top kernel is reported against its own binding roof: `bound` = whichever of algorithmic
                FLOPs / MFMA peak and algorithmic bytes / 8 TB/s is the larger time; `kernels` holds the same
                for the top five.  `traffic` / `mfma_busy` / `hbm_tbps` come from the committed rocprofv3 PMC
                passes of this command (profiles/r0N_pmc.json) when they match this configuration AND the
                kernel's sources still hash to the value stamped there.
  extra_legs    N = 1 only: the other configurations, each measured the same way in the same process:
                RN50 bs = 512 (the per-GPU-batch-matched baseline of the N >= 2 lines), RN50 in fp16 mode,
                ViT-B/32 parity + fp16, ViT-L/14@336px fp16, the same step fed from decoded uint8 images, the zero-shot tail and
                the adapter step at configs[4]'s global batch (8192 x 768), and the adapter-only train step at bs 256 / 1024.
  cpu_baseline  the oracle (torch-CPU restatement, proved == reference) on the host cores, bounded sample,
                N = 1 only
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def self_launch(argv):
    """`python bench.py --gpus N` with N > 1 and no launcher environment (the driver's command shape): this process becomes a plain
    parent that starts the N ranks as CHILD processes under torch.distributed.run (one rank per GPU, rendezvous on 127.0.0.1),
    lets them write to its stdout / stderr (rank 0 prints the JSON line) and exits with their return code.  It runs before
    anything of this file touches the GPU: no HIP call, no library load, no exec of a process that initialised a device
    (torch.cuda.device_count() only counts, it does not initialise).  Returns None when there is nothing to launch."""
    pre = argparse.ArgumentParser(add_help=False)
    pre.add_argument("--gpus", type=int, default=1)
    n = pre.parse_known_args(argv)[0].gpus
    if n <= 1 or "WORLD_SIZE" in os.environ:
        return None
    import socket
    import subprocess
    backend = os.environ.get("DBMM_DIST_BACKEND", "nccl")
    if backend == "nccl":
        import torch
        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write(f"bench.py: --gpus {n} needs {n} GPUs for RCCL (one rank per GPU), this node shows {have}; "
                             "DBMM_DIST_BACKEND=gloo rehearses the N > 1 flow with ranks sharing a GPU\n")
            return 2
    with socket.socket() as sock:                                    # a free rendezvous port
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")                # dmabuf IPC: what RCCL needs on these hosts
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    sys.stderr.write("bench.py: launching " + " ".join(cmd) + "\n")
    return subprocess.call(cmd, env=env)


def host_cores():
    """CPU threads this process may really use: affinity capped by the cgroup CPU quota (the GPU
    boxes expose 256 logical CPUs but grant 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


if __name__ == "__main__":
    _rc = self_launch(sys.argv[1:])
    if _rc is not None:
        sys.exit(_rc)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import dbmm_amd  # noqa: E402,F401
from dbmm_amd import adapter, dp, ops, optim, synth  # noqa: E402
from dbmm_amd.clip.model import build_model  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3          # MI355X_MICROARCH.md, chip table (v_mfma_f32_32x32x2_f32)
F16_MFMA_PEAK_TFLOPS = 2500.0          # dense fp16 MFMA, same table (same rate as bf16)
# 16-bit partial products a split-precision kernel issues per fp32 product, by plane count
SPLIT_PRODUCTS = {(2, 1): 2, (2, 2): 3, (3, 3): 6}
# algorithmic GFLOP per image, SURVEY.md section 8d (RN50: conv 5.367 + attn-pool 0.426 GMAC)
GFLOP_PER_IMG = {"RN50": 11.59, "ViT-B/32": 8.82, "ViT-L/14@336px": 381.9}
HBM_PEAK_GBS = 8000.0                  # HBM3E spec, same table (measured copy ceiling there: 6290 GB/s)


def kernel_products(tag):
    """16-bit partial products per fp32 product of a kernel name (1 = fp32-input MFMA; 0 = fp16-mode kernel: one fp16
    product per product, priced against the full 2500 TFLOP/s)."""
    if tag.startswith(("gemm_f16_kernel", "gemm_f16_8ph_kernel", "mha_f16_kernel", "conv3x3_f16_kernel", "conv3x3_f16_8ph_kernel", "conv3x3_c32_f16_kernel", "chain_f16_kernel", "conv1x1_f16_kernel", "conv1x1_res_stream_f16_kernel", "stem_s2_f16_kernel", "stem_s2_f16_mfma_kernel",
                       "avgpool2_f16_kernel")):
        return 0
    if tag.startswith("gemm_pair_8ph_kernel"):
        return 2                                                  # fp16 pair x one exact weight plane
    if tag.startswith("mha_pair_kernel"):
        return 3                                                  # activation pair x activation pair
    if tag.startswith("mha_mfma_kernel"):
        return 1
    if tag.startswith(("igemm_halo_kernel<", "conv3x3_halo8_kernel<", "conv3x3_halo8n_kernel<", "bottleneck_chain_kernel<", "bottleneck_chain8_kernel", "conv3x3_c32_kernel<",
                       "conv1x1_res_stream_kernel<")):
        return 2                                                  # fp16 pair x one exact weight plane
    if tag.startswith("igemm_x3_kernel<"):
        a = [v.strip() for v in tag[tag.index("<") + 1:tag.rindex(">")].split(",")]
        return SPLIT_PRODUCTS.get((int(a[7]), int(a[8])), 2)     # <BM, BN, WM, WN, AMODE, MINB, SK, NP, NW, BK, TWO>
    return 1


def pmc_lookup(tag, kernels):
    """the PMC record of a profile tag.  rocprofv3 spells every template argument; the library's tag leaves out the ones it does not
    report (activation / residual of the eight-phase kernels): "conv3x3_halo8_kernel<1>" is "conv3x3_halo8_kernel<1, ACT>",
    "gemm_pair_8ph_kernel" is "<ACT, RES, 0>", "gemm_pair_8ph_kernel<dual>" is "<ACT, 0, 1>" -- the launch-weighted mean of those."""
    if tag in kernels:
        return kernels[tag]
    if tag.startswith(("conv3x3_halo8_kernel<", "conv3x3_halo8n_kernel<")):
        hits = [v for k, v in kernels.items() if k.startswith(tag[:-1] + ",")]
    elif tag == "gemm_pair_8ph_kernel":
        hits = [v for k, v in kernels.items() if k.startswith("gemm_pair_8ph_kernel<") and k.endswith(", 0>")]
    elif tag == "gemm_pair_8ph_kernel<dual>":
        hits = [v for k, v in kernels.items() if k.startswith("gemm_pair_8ph_kernel<") and k.endswith(", 1>")]
    else:
        hits = []
    if not hits:
        return None
    n = sum(h.get("launches", 1) for h in hits)
    out = {"launches": n}
    for key in ("hbm_bytes_per_launch", "avg_launch_ms", "mfma_busy", "waves_per_simd", "lds_bank_conflict"):
        vals = [(h[key], h.get("launches", 1)) for h in hits if h.get(key) is not None]
        if vals:
            out[key] = sum(v * w for v, w in vals) / sum(w for _, w in vals)
    return out


def kernel_rows(prof, steps, step_ms, pmc):
    """one roofline record per kernel instantiation, ranked by measured time: each against the roof that binds it"""
    rows = []
    for tag, (n, fl, ms, by) in prof.items():
        if ms <= 0:
            continue
        nprod = kernel_products(tag)
        mfma_peak = F16_MFMA_PEAK_TFLOPS / max(nprod, 1) if nprod != 1 else FP32_MFMA_PEAK_TFLOPS
        tf, gbs = fl / (ms * 1e-3) / 1e12, by / (ms * 1e-3) / 1e9
        t_mfma, t_hbm = fl / (mfma_peak * 1e12), by / (HBM_PEAK_GBS * 1e9)
        hbm = t_hbm >= t_mfma
        row = {"kernel": tag, "share_of_step": round(ms / steps / step_ms, 4), "launches_per_step": round(n / steps, 2),
               "ms_per_step": round(ms / steps, 4), "avg_launch_ms": round(ms / n, 4),
               "bound": "hbm" if hbm else "mfma",
               "achieved": round(gbs if hbm else tf, 2), "peak": HBM_PEAK_GBS if hbm else round(mfma_peak, 1),
               "unit": "GB/s" if hbm else "TFLOP/s", "frac": round((gbs / HBM_PEAK_GBS) if hbm else (tf / mfma_peak), 4),
               "algorithmic_tflops": round(tf, 2), "algorithmic_gbs": round(gbs, 1),
               "frac_of_mfma_roof": round(tf / mfma_peak, 4), "frac_of_hbm_roof": round(gbs / HBM_PEAK_GBS, 4),
               "flops_per_launch": fl / n, "bytes_per_launch": by / n,
               "partial_products_per_fp32_product": nprod, "traffic": None, "traffic_over_algorithmic": None,
               "mfma_busy": None, "hbm_tbps": None}
        k = pmc_lookup(tag, (pmc or {}).get("kernels", {}))
        if k:                                            # committed rocprofv3 --pmc passes of this very command
            if k.get("hbm_bytes_per_launch"):
                row["traffic"] = round(k["hbm_bytes_per_launch"])
                row["traffic_over_algorithmic"] = round(k["hbm_bytes_per_launch"] / (by / n), 3)
                if k.get("avg_launch_ms"):
                    row["hbm_tbps"] = round(k["hbm_bytes_per_launch"] / (k["avg_launch_ms"] * 1e-3) / 1e12, 3)
            row["mfma_busy"] = k.get("mfma_busy")
            for extra in ("waves_per_simd", "lds_bank_conflict"):
                if extra in k:
                    row[extra] = k[extra]
        rows.append(row)
    rows.sort(key=lambda r: -r["ms_per_step"])
    return rows


def write_text_jsons(D):
    d = tempfile.mkdtemp(prefix="dbmm_bench_")
    paths = []
    for nm, C in (("clip_class", 2), ("clip_spurious", 2), ("clip_group", 4)):
        m = synth.text_matrix(1, D, C, nm)
        p = os.path.join(d, nm + ".json")
        with open(p, "w") as f:
            json.dump({f"{nm}{i}": m[:, i].tolist() for i in range(C)}, f)
        paths.append(p)
    return paths


def cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(sd, D, paths_unused, bs=32, iters=5, warmups=2):
    """oracle encode_image + adapter step on the host cores (kind = "port"): `warmups` untimed iterations at the SAME batch size (the
    first call at a shape builds the oneDNN primitives), then `iters` timed ones; value = the median, min / max beside it, together
    with the thread count torch really uses and the CPU model (the boxes share their hosts: 18 -- 146 images/s were recorded for
    this code on different leases)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import adapter_oracle as AO
    import clip_oracle as CO
    cores = host_cores()
    torch.set_num_threads(cores)
    img = synth.images(0, bs, 224)
    y, c, g = synth.labels(6, bs)
    text = synth.text_matrix(1, D, 2, "clip_class")
    osd = {"adapter." + k: v.clone() for k, v in synth.adapter_state_dict(3, D, 128).items()}
    bufs = {}
    times = []
    for it in range(warmups + iters):
        t0 = time.perf_counter()
        with torch.no_grad():
            emb = CO.rn_encode_image(sd, img)
        AO.train_step(osd, bufs, emb, y, text, 0.1)
        if it >= warmups:
            times.append(time.perf_counter() - t0)
    times.sort()
    t = times[len(times) // 2]
    try:
        load = os.getloadavg()[0]
    except OSError:
        load = None
    return {"value": round(bs / t, 2), "unit": "images/sec", "cores": cores, "kind": "port",
            "min": round(bs / times[-1], 2), "max": round(bs / times[0], 2), "iters": iters, "warmups": warmups,
            "torch_threads": torch.get_num_threads(), "cpu_model": cpu_model(), "host_loadavg_1m": load,
            "sample": f"oracle (torch-CPU fp32) RN50 encode_image + adapter step, bs={bs}, {warmups} warm-ups, median of {iters} "
                      "(min / max = slowest / fastest iteration)"}


def build_step(arch, dev, world, rank, Bl, D_hidden=128, dtype="f32", micro_batches=1):
    sd = synth.clip_state_dict(2, arch)
    model = build_model(sd).to(dev)
    if dtype == "f16":
        from dbmm_amd.clip.model import convert_weights
        convert_weights(model)                          # the reference's GPU path: fp16 weights and activations
    D, R = model.visual.output_dim, model.visual.input_resolution
    paths = write_text_jsons(D)
    ad = adapter.Adapter(D, D_hidden); ad.load_state_dict(synth.adapter_state_dict(3, D, D_hidden))
    clf = adapter.CustomCLIP(ad, *paths, temperature=0.01).to(dev).train()
    from types import SimpleNamespace
    opt = optim.set_optimizer(SimpleNamespace(learning_rate=0.1, momentum=0.9, weight_decay=5e-5), clf)
    return sd, model, D, R, paths, dp.EmbedAdapterStep(model.encode_image, clf, opt, micro_batches=micro_batches)


def synthetic_batch(R, Bl, world, rank, dev):
    """this rank's shard of the global synthetic batch (rows [rank*Bl, (rank+1)*Bl)) + its labels, resident in HBM"""
    base = synth.images(1000 + rank, min(Bl, 64), R)
    reps = (Bl + base.shape[0] - 1) // base.shape[0]
    scale = torch.linspace(0.8, 1.2, reps).repeat_interleave(base.shape[0])[:Bl].view(-1, 1, 1, 1)
    images = (base.repeat(reps, 1, 1, 1)[:Bl] * scale).to(dev).contiguous()
    y, c, g = synth.labels(6, Bl * world)
    lo, hi = dp.shard_rows(Bl * world, world, rank)
    return images, y[lo:hi].to(dev), g[lo:hi].to(dev)


def pmc_table(arch, Bl, dtype):
    """committed rocprofv3 --pmc passes of this configuration (newest profiles/r0N_pmc*.json that matches); every kernel's
    counters are stamped with the hash of the sources that define it and dropped here when those changed since"""
    import glob
    from dbmm_amd import _lib
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r0*_pmc*.json")), reverse=True):
        try:
            pmc = json.load(open(path))
        except (OSError, ValueError):
            continue
        if pmc.get("arch") != arch or pmc.get("batch_per_gpu") != Bl or pmc.get("dtype", "f32") != dtype:
            continue
        ours = {k: v for k, v in pmc.get("kernels", {}).items() if v.get("source_hash")}        # (torch's own kernels carry no stamp)
        kept = {k: v for k, v in ours.items() if v["source_hash"] == _lib.kernel_source_hash(k)}
        return {"kernels": kept, "file": os.path.relpath(path, ROOT), "dropped_stale": len(ours) - len(kept)}
    return None


def roofline_of(prof, steps, step_ms, value, world, arch, pmc):
    rows = kernel_rows(prof, steps, step_ms, pmc)
    top = rows[0] if rows else {"kernel": None, "bound": "hbm", "achieved": 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": 0.0, "traffic": None}
    all_fl = sum(v[1] for v in prof.values()); all_ms = sum(v[2] for v in prof.values()); all_by = sum(v[3] for v in prof.values())
    roof = {k: top.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "share_of_step",
                                    "launches_per_step", "avg_launch_ms", "flops_per_launch", "bytes_per_launch",
                                    "frac_of_mfma_roof", "frac_of_hbm_roof", "traffic_over_algorithmic", "mfma_busy",
                                    "hbm_tbps")}
    roof["how"] = ("kernel = largest measured time share among the timed launches (HIP events on the launch stream inside "
                   f"the timed region, every launch of every {PROFILE_EVERY}th step); achieved = algorithmic bytes (operands read once, outputs written once) or "
                   "2*M*N*K FLOPs / measured time; bound = the larger of bytes/8 TB/s and FLOPs/(2500 TF / partial "
                   "products per fp32 product); traffic / mfma_busy / hbm_tbps from the committed rocprofv3 --pmc passes "
                   + (f"({pmc['file']}, {pmc['dropped_stale']} kernels dropped: sources changed since)" if pmc else "(none match)"))
    roof["kernels"] = rows[:5]
    roof["timed_launches"] = {"ms_per_step": round(all_ms / steps, 3), "share_of_step": round(all_ms / (step_ms * steps), 4),
                              "algorithmic_tflops": round(all_fl / (all_ms * 1e-3) / 1e12, 2) if all_ms else 0.0,
                              "algorithmic_gbs": round(all_by / (all_ms * 1e-3) / 1e9, 1) if all_ms else 0.0}
    if arch in GFLOP_PER_IMG:
        tf = value / world * GFLOP_PER_IMG[arch] * 1e9 / 1e12
        roof["end_to_end_algorithmic_tflops"] = round(tf, 1)
        roof["end_to_end_over_fp32_mfma_peak"] = round(tf / FP32_MFMA_PEAK_TFLOPS, 4)
    return roof


DTYPE_DETAIL = {
    "f16": ("fp16 weights and activations in HBM, one fp16 MFMA per product, fp32 accumulate, fp32 LayerNorm / softmax / "
            "BatchNorm arithmetic (the reference's GPU path); adapter step in fp32 on the .float() embeddings; pinned to the "
            "reference's own fp16 path run on CPU (tests/golden/clip_*_f16.npz) at 3 x its fp16-vs-fp32 distance"),
    "f32": ("fp32 activations/accumulators; products on 16-bit MFMA as fp16 hi+lo pair (22-bit mantissa, exact per-tensor "
            "power-of-two scale) x fp16-exact checkpoint weight; parity suite at 1e-3 logits"),
}


PROFILE_EVERY = 4        # the launches of every 4th timed step are bracketed by HIP events (at least two steps of any run)


def timed_steps(stepper, images, y_l, g_l, steps, warmup, barrier, profiling):
    """-> (seconds for `steps` steps, launch profile, number of steps whose launches were timed)"""
    for _ in range(warmup):
        stepper.step(images, y_l, g_l)
    barrier()
    stride = max(1, min(PROFILE_EVERY, steps // 2))
    sampled = 0
    if profiling:
        ops.profile_begin(conv_only=False)
    t0 = time.perf_counter()
    for i in range(steps):
        if profiling:
            ops.profile_sample(i % stride == 0)
            sampled += i % stride == 0
        loss, logits, emb = stepper.step(images, y_l, g_l)
    barrier()
    dt = time.perf_counter() - t0
    prof = ops.profile_end() if profiling else {}
    if not torch.isfinite(loss).item():
        raise SystemExit("non-finite loss in the timed region")
    return dt, prof, max(sampled, 1)


def extra_leg(arch, dtype, Bl, steps, warmup, dev, note):
    """one more configuration on this GPU (N = 1), same step, same measurement, reported under `extra_legs`"""
    t_build = time.perf_counter()
    sd, model, D, R, paths, stepper = build_step(arch, dev, 1, 0, Bl, dtype=dtype)
    images, y_l, g_l = synthetic_batch(R, Bl, 1, 0, dev)
    if dtype == "f16":
        images = images.half()
    sync = torch.cuda.synchronize
    dt, prof, nprof = timed_steps(stepper, images, y_l, g_l, steps, warmup, sync, True)
    value, step_ms = Bl * steps / dt, dt / steps * 1e3
    roof = roofline_of(prof, nprof, step_ms, value, 1, arch, pmc_table(arch, Bl, dtype))
    keep = ("kernel", "bound", "achieved", "peak", "unit", "frac", "share_of_step", "launches_per_step", "avg_launch_ms",
            "frac_of_mfma_roof", "frac_of_hbm_roof", "end_to_end_algorithmic_tflops")
    leg = {"config": {"workload": f"CLIP-{arch} {R}px encode_image + adapter({D}-128-{D}) CE/SGD step, {Bl} images on one GPU ({note})",
                      "global_batch": Bl, "batch_per_gpu": Bl},
           "dtype": dtype, "value": round(value, 2), "unit": "images/sec", "steps": steps, "warmup": warmup,
           "ms_per_step": round(step_ms, 3), "roofline": {k: roof.get(k) for k in keep},
           "setup_s": round(time.perf_counter() - t_build - dt, 1)}
    del sd, model, stepper, images
    torch.cuda.empty_cache()
    return leg


def from_uint8_leg(dev, Bl=1024, steps=8, warmup=2, H=218, W=178):
    """the headline step fed from DECODED uint8 images resident in HBM (CelebA geometry 218 x 178; clip_inference.py:203-206 feeds the
    encoder from the reference's PIL `preprocess`): device preprocessing (Pillow-exact bicubic resize + centre crop + normalise,
    preprocess.py) -> encode_image -> adapter step.  JPEG decode stays on the host and is not part of this figure."""
    from dbmm_amd import preprocess as PP
    sd, model, D, R, paths, stepper = build_step("RN50", dev, 1, 0, Bl)
    raw = (synth.uniform(77, "u8img", (min(Bl, 64), H, W, 3)) * 255.999).to(torch.uint8)
    raw = raw.repeat((Bl + raw.shape[0] - 1) // raw.shape[0], 1, 1, 1)[:Bl].contiguous().to(dev)
    y, c, g = (t.to(dev) for t in synth.labels(6, Bl))
    buf = torch.empty((Bl, 3, R, R), device=dev, dtype=torch.float32)

    def one():
        PP.preprocess_uniform(raw, R, out=buf)
        return stepper.step(buf, y, g)
    for _ in range(warmup):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss, logits, emb = one()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); PP.preprocess_uniform(raw, R, out=buf); e1.record(); torch.cuda.synchronize()
    if not torch.isfinite(loss).item():
        raise SystemExit("non-finite loss")
    del model, stepper, sd
    torch.cuda.empty_cache()
    return {"config": {"workload": f"decoded uint8 RGB {H}x{W} images in HBM -> device preprocessing (Pillow-exact) -> CLIP-RN50 224px encode_image + "
                                   f"adapter step, {Bl} images on one GPU", "global_batch": Bl, "batch_per_gpu": Bl},
            "dtype": "f32", "value": round(Bl * steps / dt, 2), "unit": "images/sec", "steps": steps, "ms_per_step": round(dt / steps * 1e3, 3),
            "preprocess_ms": round(e0.elapsed_time(e1), 3),
            "roofline": {"bound": "hbm", "kernel": "resize_h_kernel + resize_v_norm_kernel (preprocessing only)",
                         "achieved": round((Bl * (H * W * 3 + 12 * R * R)) / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round((Bl * (H * W * 3 + 12 * R * R)) / (e0.elapsed_time(e1) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}


def from_host_leg(dev, Bl=1024, n_batches=12, H=218, W=178):
    """The stage-1 extraction loop (clip_inference.py:188-271) as the device pipeline of extract.Extractor, fed from HOST memory:
    decoded uint8 218 x 178 images in pinned memory -> side-stream H2D (double buffered) -> device preprocessing -> encode_image ->
    fused zero-shot tail -> minority flags -> one D2H per batch -> rows appended to the binary store.  Reported: images/s of the whole
    loop (file written), and beside it the same batches already resident in HBM through the same compute (what the copies cost)."""
    import shutil
    from dbmm_amd import extract, preprocess as PP
    model = build_model(synth.clip_state_dict(2, "RN50")).to(dev)
    D, R = model.visual.output_dim, model.visual.input_resolution
    Wz = synth.text_matrix(12, D, 2, "zs").to(dev)
    raw = (synth.uniform(77, "u8img", (64, H, W, 3)) * 255.999).to(torch.uint8)
    host = raw.repeat((Bl + 63) // 64, 1, 1, 1)[:Bl].contiguous().pin_memory()       # a decoder's output buffer
    y, c, g = synth.labels(6, Bl)
    names = [f"{i:06d}.jpg" for i in range(Bl)]
    d = tempfile.mkdtemp(prefix="dbmm_extract_")

    def batches(n):
        for k in range(n):
            yield host, (y, g, c, torch.zeros(Bl, dtype=torch.int64)), names
    ex = extract.Extractor(model, Wz, "celeba", max_batch=Bl)
    ex.run(batches(2), os.path.join(d, "warm.emb"), 2 * Bl)
    torch.cuda.synchronize()

    def timed(n):
        t0 = time.perf_counter()
        ex.run(batches(n), os.path.join(d, "clip.emb"), n * Bl)
        torch.cuda.synchronize()
        return time.perf_counter() - t0
    n_short = max(2, n_batches // 3)
    dt_short, dt = timed(n_short), timed(n_batches)
    steady = (dt - dt_short) / (n_batches - n_short)                  # per batch once the pipeline is full (fill, file creation and close cancel)
    # the same device work on a batch that is already resident (no H2D, no D2H, no store)
    dev_raw = host.to(dev)
    for _ in range(2):
        adapter.zeroshot_tail(model.encode_image(PP.preprocess_uniform(dev_raw, R)).float(), Wz, 0.02)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(n_batches):
        adapter.zeroshot_tail(model.encode_image(PP.preprocess_uniform(dev_raw, R)).float(), Wz, 0.02)
    torch.cuda.synchronize()
    dres = time.perf_counter() - t1
    size = os.path.getsize(os.path.join(d, "clip.emb"))
    shutil.rmtree(d, ignore_errors=True)
    del model, ex
    torch.cuda.empty_cache()
    return {"config": {"workload": f"stage-1 extraction loop from host memory: pinned uint8 RGB {H}x{W} batches -> H2D on a side stream -> device "
                                   f"preprocessing -> CLIP-RN50 224px encode_image -> zero-shot tail + minority flags -> one D2H per batch -> binary "
                                   f"store, {n_batches} batches of {Bl}", "global_batch": Bl, "batch_per_gpu": Bl},
            "dtype": "f32", "value": round(n_batches * Bl / dt, 2), "unit": "images/sec", "steps": n_batches, "ms_per_step": round(dt / n_batches * 1e3, 3),
            "device_resident_ms_per_step": round(dres / n_batches * 1e3, 3), "device_resident_images_per_sec": round(n_batches * Bl / dres, 2),
            "steady_state_ms_per_step": round(steady * 1e3, 3), "steady_state_images_per_sec": round(Bl / steady, 2),
            "steady_state_over_device_resident": round(steady / (dres / n_batches), 4),
            "exposed_copy_and_store_ms_per_step": round((steady - dres / n_batches) * 1e3, 3),
            "pipeline_fill_and_file_ms": round((dt - steady * n_batches) * 1e3, 3),
            "h2d_bytes_per_step": Bl * H * W * 3, "d2h_bytes_per_step": Bl * (4 * D + 24), "d2h_copies_per_step": 1,
            "store_bytes": size,
            "roofline": {"bound": "mfma", "kernel": "the RN50 tower (see the headline's roofline); the copies ride a side stream",
                         "achieved": round(n_batches * Bl / dt * GFLOP_PER_IMG["RN50"] / 1e3, 1), "peak": 1250.0, "unit": "TFLOP/s",
                         "frac": round(n_batches * Bl / dt * GFLOP_PER_IMG["RN50"] / 1e3 / 1250.0, 4)}}


def tail_leg(B, D, dev, steps=300):
    """BASELINE configs[4]'s "HBM-bound L2-norm + sim-GEMM roofline check": the zero-shot tail of clip_inference.py:207-216 (row L2-norm,
    x text matrix / T, argmax) on the global batch of that config's embeddings, one fused launch per call"""
    emb = synth.normal(11, f"tail{B}", (B, D), 0.5).to(dev)
    W = synth.text_matrix(12, D, 2, "zs").to(dev)
    tn = W.t().contiguous()
    for _ in range(20):
        ops.l2norm_sim_ce_fwd(emb, tn, 0.02, want_pred=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        logits, _, _, pred, _ = ops.l2norm_sim_ce_fwd(emb, tn, 0.02, want_pred=True)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / steps * 1e3
    by = 4 * B * D + 4 * B * 2 + 8 * B + 4 * B              # embeddings read once; logits, int64 predictions, 1 / norm written
    return {"config": {"workload": f"zero-shot tail (row L2-norm + image x text logits / T + argmax) on [{B}, {D}] fp32 embeddings "
                                   "(BASELINE configs[4]'s global batch and width)", "global_batch": B},
            "dtype": "f32", "value": round(us, 2), "unit": "us/call", "higher_is_better": False, "steps": steps, "launches_per_step": 1,
            "roofline": {"bound": "hbm", "kernel": "l2norm_sim_ce_fwd_kernel<4>", "achieved": round(by / (us * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(by / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes": by,
                         "note": "25 MB per call = 3.2 us at the HBM roof: the call is bounded by the launch floor of the box, not by bytes"}}


def adapter_only_leg(B, D, dev, steps=200, warmup=20):
    """BASELINE configs[0]: the adapter-only train step (final_main.py:455-466) on precomputed embeddings resident in HBM:
    forward, mean CE, backward, SGD as the one-call fused step.  Launch-bound: figure of merit = us/step against
    (launches x ~1.5 us) and against the algorithmic bytes at 8 TB/s."""
    from types import SimpleNamespace
    paths = write_text_jsons(D)
    ad = adapter.Adapter(D, 128); ad.load_state_dict(synth.adapter_state_dict(3, D, 128))
    clf = adapter.CustomCLIP(ad, *paths, temperature=0.01).to(dev).train()
    opt = optim.set_optimizer(SimpleNamespace(learning_rate=0.1, momentum=0.9, weight_decay=5e-5), clf)
    x = synth.normal(5, f"x{B}", (B, D), 0.5).to(dev)
    y, c, g = (t.to(dev) for t in synth.labels(6, B))
    for _ in range(warmup):
        clf.train_step(x, y, opt)
    torch.cuda.synchronize()
    windows = []                                   # three timing windows, the median reported: a launch-bound loop sees every host hiccup
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            loss, logits, rows = clf.train_step(x, y, opt)
        torch.cuda.synchronize()
        windows.append((time.perf_counter() - t0) / steps * 1e6)
    us = sorted(windows)[1]
    if not torch.isfinite(loss).item():
        raise SystemExit("non-finite adapter loss")
    # the launch floor of THIS box: the same dependent launches on four rows (no batch-dependent work left)
    x4, y4 = x[:4].contiguous(), y[:4].contiguous()
    for _ in range(warmup):
        clf.train_step(x4, y4, opt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        clf.train_step(x4, y4, opt)
    torch.cuda.synchronize()
    floor_us = (time.perf_counter() - t0) / steps * 1e6
    P, C = 2 * D * 128 + 3 * 128 + D, 2
    alg_bytes = 8 * B * D + 4 * B * C + 20 * P
    launches = getattr(ops, "adapter_step_launches", lambda *a: None)(B, D, 128, False)
    return {"config": {"workload": f"adapter-only train step (Adapter({D},128) + cosine logits + CE + SGD) on precomputed embeddings, bs={B} "
                                   "(BASELINE configs[0] shape)", "global_batch": B},
            "dtype": "f32", "value": round(us, 2), "unit": "us/step", "higher_is_better": False, "samples_per_sec": round(B / us * 1e6, 1),
            "steps": steps, "windows_us": [round(w, 2) for w in windows], "launches_per_step": launches,
            "roofline": {"bound": "hbm", "kernel": "adapter train step (all launches)", "achieved": round(alg_bytes / (us * 1e-6) / 1e9, 2),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 5),
                         "algorithmic_bytes": alg_bytes,
                         "launch_floor_us_measured": round(floor_us, 2),
                         "launch_floor_us_measured_how": "the same step (same dependent launches) at bs=4, same process",
                         "launch_floor_us_survey_model": None if launches is None else round(launches * 1.5, 1),
                         "launch_floor_us_survey_model_how": "launches x 1.5 us (SURVEY section 8d's kernel-boundary estimate; this chip "
                                                             "measures ~5 us per DEPENDENT launch)",
                         "frac_of_measured_floor": round(floor_us / us, 4)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch-per-gpu", type=int, default=0,
                    help="default: 1024 images per GPU at every N (N = 1 is the metric's bs=1024; weak scaling keeps the per-GPU work fixed); "
                         "512 = BASELINE configs[1] / configs[2]")
    ap.add_argument("--arch", default="RN50")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f16"],
                    help="f32 = parity mode (the headline); f16 = the reference's GPU-path arithmetic "
                         "(BASELINE configs[4]: --arch 'ViT-L/14@336px' --dtype f16)")
    ap.add_argument("--micro-batches", type=int, default=1,
                    help="encode in k row chunks, each chunk's all-gather in flight while the next encodes (BASELINE configs[3]'s overlap). "
                         "Default 1: measured on one MI355X at 512 images per GPU, chunking the encode costs 10 % (k = 2) / 27 % (k = 4) of the "
                         "per-GPU rate (smaller launches), the gather it would hide is 2 MiB per rank = < 0.1 % of the step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32-mfma-leg", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="N = 1: skip the other configurations reported under extra_legs")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    backend = os.environ.get("DBMM_DIST_BACKEND", "nccl")     # "gloo": rehearsal of the N>1 flow on one GPU
    if os.environ.get("DBMM_BENCH_DRYRUN") == "1":
        # launcher / rendezvous check without a GPU (tests/test_bench_launcher.py): the ranks meet on gloo, exchange what the real
        # run exchanges around its timed region (barrier, MAX all-reduce of the time, all_gather_object) and rank 0 prints a line
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.barrier()
        t = torch.tensor([float(rank + 1)], dtype=torch.float64)
        who = [None] * world
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dist.all_gather_object(who, {"rank": rank, "local_rank": local_rank})
        if rank == 0:
            print(json.dumps({"dry_run": True, "n_gpus": world, "max_over_ranks": t.item(), "ranks": who, "steps": args.steps,
                              "warmup": args.warmup}), flush=True)
        if world > 1:
            dist.destroy_process_group()
        return
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # per-GPU work is the same at every N (weak scaling): RN50 runs the metric's 1024 images on every GPU, so that the driver's
    # value(N) / (N * value(1)) compares equal launches; BASELINE configs[2] (bs = 1024 over 2 GPUs) is --batch-per-gpu 512
    default_bl = {"RN50": 1024, "ViT-B/32": 512, "ViT-L/14@336px": 1024 if args.dtype == "f16" else 128}
    Bl = args.batch_per_gpu or default_bl.get(args.arch, 512)
    B = Bl * world
    micro = max(1, args.micro_batches)
    sd, model, D, R, paths, stepper = build_step(args.arch, dev, world, rank, Bl, dtype=args.dtype, micro_batches=micro)
    images, y_l, g_l = synthetic_batch(R, Bl, world, rank, dev)

    def barrier():
        if world > 1:
            if backend == "nccl":
                dist.barrier(device_ids=[dev_index])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    # every igemm launch of the timed region is bracketed by HIP events (an event pair is ~4 us of stream time:
    # ~60 launches per RN50 step = 0.25 ms of a ~30 ms step); DBMM_BENCH_PROFILE=0 times none
    profiling = os.environ.get("DBMM_BENCH_PROFILE", "1") != "0"
    dt, prof, nprof = timed_steps(stepper, images, y_l, g_l, args.steps, args.warmup, barrier, profiling)

    tmax = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
    dist_info = None
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        # proof of what ran: the backend torch.distributed reports, its world size, and every rank's device
        devs = [None] * world
        dist.all_gather_object(devs, {"rank": rank, "device": torch.cuda.current_device(), "name": torch.cuda.get_device_name(dev_index)})
        dist_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks": devs, "micro_batches": micro,
                     "gather": ("all_gather_into_tensor(async_op) per micro-batch on a side stream, overlapped with the next micro-batch's encode"
                                if micro > 1 else "one all_gather_into_tensor of the rank's embeddings + one of its packed (y, g) per step")}
    dt = tmax.item()

    # N > 1: the north_star's own shape beside the weak-scaling line -- the metric's GLOBAL batch (RN50: 1024 = BASELINE configs[2] at
    # N = 2) split over the N ranks (strong scaling), same process group, same barriers, max over ranks
    fixed = None
    if world > 1 and not args.batch_per_gpu and default_bl.get(args.arch) and default_bl[args.arch] % world == 0 and micro == 1:
        Bf = default_bl[args.arch] // world
        img_f, y_f, g_f = synthetic_batch(R, Bf, world, rank, dev)
        if args.dtype == "f16":
            img_f = img_f.half()
        dtf, _, _ = timed_steps(stepper, img_f, y_f, g_f, args.steps, max(1, args.warmup), barrier, False)
        tf = torch.tensor([dtf], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tf, op=dist.ReduceOp.MAX)
        fixed = {"config": {"workload": f"global batch {Bf * world} split over {world} GPUs ({Bf} images/GPU)"
                                        + (" = BASELINE configs[2]" if (args.arch, world) == ("RN50", 2) else ""),
                            "global_batch": Bf * world, "batch_per_gpu": Bf},
                 "value": round(Bf * world * args.steps / tf.item(), 2), "unit": "images/sec", "scaling": "strong",
                 "ms_per_step": round(tf.item() / args.steps * 1e3, 3), "steps": args.steps}
        del img_f

    if rank == 0:
        value = B * args.steps / dt
        step_ms = dt / args.steps * 1e3
        roof = roofline_of(prof, nprof, step_ms, value, world, args.arch, pmc_table(args.arch, Bl, args.dtype))
        roof["timed_steps"] = nprof
        cfgs = {("RN50", 1, 1024): "the metric's configuration, CLIP-RN50 224px bs=1024 on one GPU",
                ("RN50", 1, 512): "BASELINE configs[1]", ("RN50", 2, 512): "BASELINE configs[2]",
                ("RN50", 2, 1024): "the metric's per-GPU batch on 2 GPUs", ("RN50", 4, 1024): "the metric's per-GPU batch on 4 GPUs",
                ("RN50", 8, 1024): "the metric's per-GPU batch on 8 GPUs",
                ("ViT-B/32", 8, 512): "BASELINE configs[3]", ("ViT-L/14@336px", 8, 1024): "BASELINE configs[4]",
                ("ViT-L/14@336px", 1, 1024): "one GPU's share of BASELINE configs[4]"}
        line = {
            "metric": f"images/sec (embed+adapter step), CLIP-{args.arch} {R}px bs={B}", "value": round(value, 2),
            "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(step_ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "dtype_detail": DTYPE_DETAIL[args.dtype],
            "data": "synthetic",
            "config": {"workload": f"CLIP-{args.arch} {R}px encode_image + adapter({D}-128-{D}) CE/SGD step, {Bl} images/GPU, "
                                   f"global batch {B}" + (f" ({cfgs[(args.arch, world, Bl)]})" if (args.arch, world, Bl) in cfgs else ""),
                       "global_batch": B, "batch_per_gpu": Bl, "parallelism": f"dp{world}",
                       "collective": "all_gather(embeddings+labels) per step" if world > 1 else "none"},
            "roofline": roof,
        }
        if dist_info:
            line["dist"] = dist_info
            line["config"]["weak_scaling_note"] = (f"{Bl} images per GPU at every N (the N = 1 line runs the same per-GPU batch): value(N) / (N x value(1)) "
                                                   "compares equal per-GPU work")
        if world > 1 and fixed is not None:
            line["fixed_global_batch"] = fixed
        if world == 1 and not args.no_fp32_mfma_leg and args.arch.startswith("RN") and args.dtype == "f32":
            line["fp32_input_mfma"] = fp32_mfma_leg(args.arch, dev, Bl, images, y_l, g_l)
        if world == 1 and not args.no_extra_legs and args.arch == "RN50" and args.dtype == "f32" and not args.batch_per_gpu:
            del stepper, model, images
            torch.cuda.empty_cache()
            legs = {}
            # (2 warm-up steps and 6 - 12 timed ones per leg: a leg is a witness of its configuration, and enough steps to A/B on)
            legs["rn50_bs512"] = extra_leg("RN50", "f32", 512, 8, 2, dev, "BASELINE configs[1] = one GPU's share of configs[2]")
            legs["rn50_f16_bs1024"] = extra_leg("RN50", "f16", 1024, 8, 2, dev, "the metric's batch in the reference's GPU-path arithmetic")
            legs["vit_b32_f32_bs512"] = extra_leg("ViT-B/32", "f32", 512, 12, 2, dev, "one GPU's share of BASELINE configs[3], parity mode")
            legs["vit_b32_f16_bs512"] = extra_leg("ViT-B/32", "f16", 512, 12, 2, dev, "one GPU's share of BASELINE configs[3], fp16 mode")
            legs["vit_l14_336_f16_bs1024"] = extra_leg("ViT-L/14@336px", "f16", 1024, 3, 1, dev,
                                                       "one GPU's share of BASELINE configs[4] (tests/test_gpu_config_sizes.py runs this size)")
            legs["rn50_from_uint8_bs1024"] = from_uint8_leg(dev)
            legs["rn50_from_host_bs1024"] = from_host_leg(dev)
            legs["tail_bs8192_d768"] = tail_leg(8192, 768, dev)
            legs["adapter_only_bs8192_d768"] = adapter_only_leg(8192, 768, dev, steps=50, warmup=5)
            legs["adapter_only_bs256"] = adapter_only_leg(256, 1024, dev)
            legs["adapter_only_bs1024"] = adapter_only_leg(1024, 1024, dev)
            line["extra_legs"] = legs
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sd, D, paths)
        print(json.dumps(line), flush=True)
    if world > 1:
        barrier()
        dist.destroy_process_group()


def fp32_mfma_leg(arch, dev, Bl, images, y_l, g_l, steps=2):
    """the same step with every product on the fp32-input MFMA (v_mfma_f32_32x32x2_f32): what "f32" would cost
    without the fp16-pair split -- printed beside the headline so the label cannot be misread"""
    from dbmm_amd.clip import model as M
    saved = M.set_plan_option("conv_split", "off")
    try:
        _, _, _, _, _, stepper = build_step(arch, dev, 1, 0, Bl)
        stepper.step(images, y_l, g_l)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            stepper.step(images, y_l, g_l)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        M.set_plan_option("conv_split", saved)
    return {"value": round(Bl * steps / dt, 2), "unit": "images/sec", "steps": steps,
            "note": "plan option conv_split = off: fp32-input MFMA for every product, same process, same batch"}


if __name__ == "__main__":
    main()
