"""bench.py -- train samples/sec of the GAN-DANet G+D step on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one full G+D update (generator forward/backward with PAM/CAM attention, three discriminator
passes, TV + perceptual(VGG19 random-init) + MSE + BCE losses, SSIM evaluation, two AdamW steps) on
synthetic 8-channel 256x256 tiles -> 1024x1024, batch 32 per GPU (BASELINE.json configs[2]; config 3 of
SURVEY.md 8d), bf16 MFMA operands / fp32 accumulate and storage.  Weak scaling: per-GPU batch fixed,
gradients all-reduced with RCCL.  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (BASELINE config: 32)")
    ap.add_argument("--tile", type=int, default=256, help="generator input tile (BASELINE config: 256)")
    ap.add_argument("--channels", type=int, default=8)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-perceptual", action="store_true", help="exploration only; the reported config has it on")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-steps", type=int, default=4)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                    "the multi-rank path on a one-GPU box, all ranks on cuda:0)")
    return ap.parse_args()


def cpu_baseline(nsteps: int):
    """The CPU oracle (validated against the reference modules, tests/test_oracle_golden.py) timed on this
    host: config 1 of BASELINE.json (B=1, 8-ch 64x64 tile -> 256x256, fp32), since the 256x256-tile workload
    needs ~17 GB per materialised attention matrix on the reference path and cannot run on CPU."""
    from oracle import modules as OM
    from oracle import step as OS
    torch.manual_seed(1234)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    nthreads = max(1, min(avail, 16))      # the GPU box gives one GPU a 16-core share
    torch.set_num_threads(nthreads)
    G, D = OM.FlexibleUpsamplingModule(input_channels=8), OM.Discriminator1()
    x, tgt = torch.randn(1, 8, 64, 64), torch.randn(1, 1, 256, 256)
    with torch.no_grad():
        D(tgt)
    G.apply(OM.weights_init_normal)
    D.apply(OM.weights_init_normal)
    for n, p in G.named_parameters():
        if n.endswith("gamma"):
            p.data.fill_(0.1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        perc = OM.PerceptualLoss(pretrained=False)
    og, od = OS.AdamWState(lr=2e-4), OS.AdamWState(lr=4e-4)
    OS.train_step(G, D, og, od, x, tgt, 0.5, 1e-5, perc)   # warm-up
    t0 = time.perf_counter()
    done = 0
    for _ in range(nsteps):                                 # bounded: stop after ~25 s of CPU work
        OS.train_step(G, D, og, od, x, tgt, 0.5, 1e-5, perc)
        done += 1
        if time.perf_counter() - t0 > 25.0:
            break
    nsteps = done
    dt = time.perf_counter() - t0
    return {"value": round(nsteps * 1 / dt, 4), "unit": "samples/s", "cores": nthreads, "kind": "port",
            "sample": f"{nsteps} G+D steps of BASELINE config 1 (B=1, 8ch 64x64 tile -> 256x256, fp32, "
                      f"perceptual term with random VGG19) by the CPU oracle; {dt / nsteps:.2f} s/step"}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    one_gpu_rehearsal = args.backend != "nccl"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_gpu_rehearsal:
            dist.init_process_group(args.backend)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank if (world > 1 and not one_gpu_rehearsal) else 0)
    torch.cuda.set_device(dev)

    import gan_danet_amd as gd
    from gan_danet_amd import kern
    from gan_danet_amd.parallel import broadcast_module

    gd.set_precision(args.precision)
    B, T, Cin = args.batch, args.tile, args.channels

    # ---- models (random init of the reference architecture; no checkpoints offline) ----
    torch.manual_seed(1234)
    G = gd.FlexibleUpsamplingModule(input_channels=Cin).to(dev)
    D = gd.Discriminator1().to(dev)
    with torch.no_grad():
        D(torch.zeros(1, 1, 4 * T, 4 * T, device=dev))          # materialise LazyLinear fc1
    G.apply(gd.weights_init_normal)
    D.apply(gd.weights_init_normal)
    for n, p in G.named_parameters():
        if n.endswith("gamma"):
            p.data.fill_(0.1)                                     # attention active (reference init is 0)
    perc = None
    if not args.no_perceptual:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            perc = gd.PerceptualLoss(pretrained=False, device=dev)
    broadcast_module(G)
    broadcast_module(D)
    if perc is not None:
        broadcast_module(perc)
    G.train()
    D.train()
    trainer = gd.GanTrainer(G, D, perceptual=perc)

    # ---- synthetic shard, resident in HBM before the timed region ----
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.randn(B, Cin, T, T, device=dev, generator=gen)
    target = torch.randn(B, 1, 4 * T, 4 * T, device=dev, generator=gen)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        trainer.step(x, target, 0.5)
    sync()
    kern.PROFILE.clear()
    kern.PROFILE_ON = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = trainer.step(x, target, 0.5)
    sync()
    dt = time.perf_counter() - t0
    kern.PROFILE_ON = False
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()

    loss_d, loss_g = out.loss_d.item(), out.loss_g.item()
    finite = bool(torch.isfinite(out.loss_d).all() and torch.isfinite(out.loss_g).all())

    # ---- roofline of the dominant kernel from the HIP-event brackets recorded around its launches ----
    roof = None
    traffic = {}
    tpath = os.path.join(ROOT, "profiles", "r01_pam_traffic.json")
    if os.path.exists(tpath) and B == 32 and T == 256:   # PMC bytes measured offline at exactly this launch shape (C=184)
        with open(tpath) as f:
            traffic = json.load(f).get("traffic_bytes_per_launch", {})
    stats = kern.profile_summary()
    if stats:
        name, (n_launch, ms_avg, flops, nbytes) = max(stats.items(), key=lambda kv: kv[1][0] * kv[1][1])
        ach = flops / (ms_avg * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": name, "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS,
                "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic.get(name),
                "launches": n_launch, "avg_ms": round(ms_avg, 3),
                "algorithmic_flop_per_launch": flops,
                "all": {k: {"launches": v[0], "avg_ms": round(v[1], 3),
                            "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 1)} for k, v in stats.items()}}

    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args.cpu_baseline_steps)
        res = {
            "metric": "train samples/sec (G+D step) on 256x256 tiles",
            "value": round(args.steps * B * world / dt, 3),
            "unit": "samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic (randn tiles, random-init weights, random-init VGG19 for the perceptual term)",
            "config": {"workload": f"full G+D train step, {Cin}ch {T}x{T} -> {4 * T}x{4 * T} tiles, batch {B}/GPU "
                                   f"(global {B * world}), perceptual={'on' if perc is not None else 'OFF'}, "
                                   f"SSIM evaluated, AdamW x2",
                       "per_gpu_batch": B, "global_batch": B * world, "tile": T, "parallelism": f"dp{world}",
                       "storage": "fp32 activations/weights, bf16 MFMA operands, fp32 accumulate"},
            "loss_d": loss_d, "loss_g": loss_g, "finite": finite,
            "roofline": roof,
            "cpu_baseline": cpu,
            "max_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
        }
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
