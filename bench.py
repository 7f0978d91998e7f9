#!/usr/bin/env python3
"""images/sec of one "embed + adapter step" (BASELINE.json metric) on N MI355X of one node.

Roofline note: the dense contractions run on the split-precision kernel (each fp32 value = three
bf16 values, six bf16 MFMA partial products per fp32 product, fp32-level accuracy).  Its MFMA
roofline in ALGORITHMIC FLOP/s is the dense bf16 peak / 6 = 416.7 TFLOP/s; `achieved` and
`frac` are quoted against that, and `achieved_over_fp32_mfma_peak` against the 157.3 TFLOP/s an
fp32-input MFMA kernel could reach at most.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = CLIP-RN50 encode_image on this rank's B_l = 512 synthetic 224x224x3 images (resident
in HBM) -> RCCL all-gather of embeddings + labels -> replicated adapter forward + CE +
backward + SGD on the global batch (B = 512 x N: BASELINE.json configs[1] at N = 1,
configs[2] = the bs-1024 CelebA case at N = 2).  Weak scaling.  fp32 end to end (fp32-input
MFMA) -- the precision the 1e-3 logit parity against the reference's CPU path is defined in.

Besides the contract line, rank 0 reports
  roofline      the dominant kernel (3x3 implicit-GEMM conv on the 128x128 tile), HIP events
                around every launch of it inside the timed region; algorithmic FLOPs
                (2*M*N*K per launch) / mean launch time vs the 157.3 TFLOP/s fp32 MFMA peak
  cpu_baseline  the oracle (torch-CPU restatement, proved == reference) on the host cores,
                bounded sample, N = 1 only
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

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
# dominant kernel: 3x3 implicit-GEMM conv, 128x128 tile, one tile per workgroup
# (template arguments <BM, BN, WAVES_M, WAVES_N, AMODE=1 (conv), MINB, SK=0, NP, NW, BK, TWO=0>; NP / NW = 16-bit
#  planes of the activations / weights: 2, 1 = fp16 pair x exact fp16 weight; 2, 2 = fp16 pair x pair; 3, 3 = bf16 triple)
DOMINANT_SPLIT = ("igemm_x3_kernel<128, 128, 2, 2, 1, 3, 0, 2, 1, 32, 0>",  # default path (the three pooled 3x3 convs)
                  "igemm_x3_kernel<128, 128, 2, 2, 1, 2, 0, 2, 2, 32, 0>",  # weights not exact in fp16
                  "igemm_x3_kernel<128, 128, 2, 2, 1, 2, 0, 2, 2, 16, 0>",  # DBMM_IGEMM_X2_BK=16
                  "igemm_x3_kernel<128, 128, 2, 2, 1, 2, 0, 3, 3, 16, 0>")  # DBMM_CONV_SPLIT=bf16
DOMINANT_F32 = "igemm_f32_kernel<128, 128, 2, 2, 1, 0, 16, 4, 1, 0, 1>"    # fp32-MFMA path (DBMM_CONV_SPLIT=off)


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


def cpu_baseline(sd, D, paths_unused, bs=32, iters=3):
    """oracle encode_image + adapter step on the host cores (kind = "port")."""
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
    with torch.no_grad():
        CO.rn_encode_image(sd, img[:4])                       # warm-up
    for _ in range(iters):
        t0 = time.perf_counter()
        with torch.no_grad():
            emb = CO.rn_encode_image(sd, img)
        AO.train_step(osd, bufs, emb, y, text, 0.1)
        times.append(time.perf_counter() - t0)
    times.sort()
    t = times[len(times) // 2]
    return {"value": round(bs / t, 2), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"oracle (torch-CPU fp32) RN50 encode_image + adapter step, bs={bs}, median of {iters}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch-per-gpu", type=int, default=512)
    ap.add_argument("--arch", default="RN50")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    backend = os.environ.get("DBMM_DIST_BACKEND", "nccl")     # "gloo": rehearsal of the N>1 flow on one GPU
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    Bl = args.batch_per_gpu
    B = Bl * world
    sd = synth.clip_state_dict(2, args.arch)
    model = build_model(sd).to(dev)
    D, R = model.visual.output_dim, model.visual.input_resolution
    # this rank's shard of the global synthetic batch (rows [rank*Bl, (rank+1)*Bl))
    base = synth.images(1000 + rank, min(Bl, 64), R)
    reps = (Bl + base.shape[0] - 1) // base.shape[0]
    scale = torch.linspace(0.8, 1.2, reps).repeat_interleave(base.shape[0])[:Bl].view(-1, 1, 1, 1)
    images = (base.repeat(reps, 1, 1, 1)[:Bl] * scale).to(dev).contiguous()
    y, c, g = synth.labels(6, B)
    lo, hi = dp.shard_rows(B, world, rank)
    y_l, g_l = y[lo:hi].to(dev), g[lo:hi].to(dev)

    paths = write_text_jsons(D)
    ad = adapter.Adapter(D, 128); ad.load_state_dict(synth.adapter_state_dict(3, D, 128))
    clf = adapter.CustomCLIP(ad, *paths, temperature=0.01).to(dev).train()
    from types import SimpleNamespace
    opt = optim.set_optimizer(SimpleNamespace(learning_rate=0.1, momentum=0.9, weight_decay=5e-5), clf)
    stepper = dp.EmbedAdapterStep(model.encode_image, clf, opt)

    def barrier():
        if world > 1:
            if backend == "nccl":
                dist.barrier(device_ids=[dev_index])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        stepper.step(images, y_l, g_l)
    barrier()
    # RN towers: time only the KxK conv launches (the dominant kernel lives there); DBMM_BENCH_PROFILE_ALL=1
    # or a transformer tower: every MFMA launch
    conv_only = args.arch.startswith("RN") and os.environ.get("DBMM_BENCH_PROFILE_ALL") != "1"
    ops.profile_begin(conv_only=conv_only)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, logits, emb = stepper.step(images, y_l, g_l)
    barrier()
    dt = time.perf_counter() - t0
    prof = ops.profile_end()

    tmax = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = tmax.item()
    if not torch.isfinite(loss).item():
        raise SystemExit("non-finite loss in the timed region")

    if rank == 0:
        value = B * args.steps / dt
        DOMINANT = next((k for k in DOMINANT_SPLIT + (DOMINANT_F32,) if k in prof), None)
        if DOMINANT is None:     # other towers (ViT): the MFMA kernel instantiation with the most time
            DOMINANT = max(prof, key=lambda k: prof[k][2]) if prof else DOMINANT_F32
        split = DOMINANT.startswith("igemm_x3_kernel<")
        n_prod = SPLIT_PRODUCTS.get(tuple(int(v) for v in DOMINANT.rstrip(">").split(",")[-4:-2]), 2) if split else 1
        peak = F16_MFMA_PEAK_TFLOPS / n_prod if split else FP32_MFMA_PEAK_TFLOPS
        n, fl, ms = prof.get(DOMINANT, (0, 0.0, 0.0))
        ach = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        traffic = None          # HBM-side bytes per launch of the dominant kernel from the committed PMC passes
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            if Bl == 512 and args.arch == "RN50":
                traffic = round(tj["kernels"][DOMINANT]["hbm_bytes_per_launch"])
        except (OSError, KeyError, ValueError):
            pass
        all_fl = sum(v[1] for v in prof.values()); all_ms = sum(v[2] for v in prof.values())
        line = {
            "metric": f"images/sec (embed+adapter step), CLIP-{args.arch} {R}px", "value": round(value, 2),
            "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"CLIP-{args.arch} {R}px encode_image + adapter(1024-128-1024) CE step, "
                                   f"{Bl} images/GPU (BASELINE configs[1]; configs[2] at 2 GPUs)",
                       "global_batch": B, "batch_per_gpu": Bl, "parallelism": f"dp{world}",
                       "collective": "all_gather(embeddings+labels) per step" if world > 1 else "none"},
            "roofline": {"bound": "mfma", "kernel": DOMINANT, "achieved": round(ach, 2),
                         "peak": round(peak, 1), "unit": "TFLOP/s",
                         "frac": round(ach / peak, 4), "traffic": traffic,
                         "peak_basis": (f"dense 16-bit MFMA 2500 TFLOP/s / {n_prod} partial products per fp32 product"
                                        if split else "fp32-input MFMA 157.3 TFLOP/s"),
                         "executed_mfma_tflops": round(ach * n_prod, 1),
                         "achieved_over_fp32_mfma_peak": round(ach / FP32_MFMA_PEAK_TFLOPS, 4),
                         "launches": n, "avg_launch_ms": round(ms / n, 4) if n else None,
                         "flops_per_launch_avg": fl / n if n else None,
                         "timed_scope": "KxK conv launches" if conv_only else "all MFMA launches",
                         "timed_mfma_kernels": {"achieved": round(all_fl / (all_ms * 1e-3) / 1e12, 2) if all_ms else 0.0,
                                              "ms_per_step": round(all_ms / args.steps, 3),
                                              "share_of_step": round(all_ms / (dt * 1e3), 4)},
                         "end_to_end_over_fp32_mfma_peak": round(value / world * GFLOP_PER_IMG.get(args.arch, float("nan")) * 1e9 / (FP32_MFMA_PEAK_TFLOPS * 1e12), 4)},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sd, D, paths)
        print(json.dumps(line), flush=True)
    if world > 1:
        barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
