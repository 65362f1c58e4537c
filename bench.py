"""bench.py -- train samples/sec of the GAN-DANet G+D step on MI355X (BASELINE.json metric).

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step = one full G+D update (generator forward/backward with PAM/CAM attention, three discriminator
passes, TV + perceptual(VGG19 random-init) + MSE + BCE losses, SSIM evaluation, two AdamW steps) on
synthetic 8-channel 256x256 tiles -> 1024x1024, batch 32 per GPU (BASELINE.json configs[2]; config 3 of
SURVEY.md 8d), bf16 MFMA operands / fp32 accumulate and storage.  Weak scaling: per-GPU batch fixed,
gradients all-reduced with RCCL.  Prints ONE JSON line on rank 0.

--config selects the other measured configurations of SURVEY.md 8d (default 3 = the metric's own):
    2   DANetAttention(64) FORWARD on a (B, 64, 128, 128) feature map, bf16 (attention bring-up; step = one forward)
    3/4 the full G+D step (4 = the same under torchrun with N ranks)
    5   512x512 tiles, fp16 PAM operands, N = 262 144 tokens, batch 1 per GPU: the FULL G+D step like config 3.
        Discriminator1.fc1 at this tile is 8.6e9 parameters: 34 GB of weights on every rank + 34 GB of gradient; the AdamW
        state (69 GB) is whole at world 1 (~150 GB in all: fits 288 GB) and 1/world per rank under the sharded optimiser
        (parallel.ShardedParam).  --no-disc gives round 2's generator-only variant (forward + backward + AdamW, MSE + TV).
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
    ap.add_argument("--config", type=int, default=3, choices=[2, 3, 4, 5], help="SURVEY.md 8d configuration")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (config 3: 32, config 2: 32, config 5: 1)")
    ap.add_argument("--tile", type=int, default=None, help="generator input tile (config 3: 256, 2: 128, 5: 512)")
    ap.add_argument("--channels", type=int, default=8)
    ap.add_argument("--precision", default=None, choices=["bf16", "fp16", "fp32", "mixed"])
    ap.add_argument("--no-perceptual", action="store_true", help="exploration only; the reported config has it on")
    ap.add_argument("--no-disc", action="store_true", help="config 5 only: generator forward + backward + AdamW, no discriminator")
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


def g_out_rel_err(gd, dev, precision):
    """generator output error of the timed operand mode on the REFERENCE fixture (tests/golden/generator_8ch_16x16.npz,
    generated from /root/reference/models/generator.py by tests/golden/make_golden.py): the north-star quantity"""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from fill import fill_module
    fx = np.load(os.path.join(ROOT, "tests", "golden", "generator_8ch_16x16.npz"))
    Gf = gd.FlexibleUpsamplingModule(input_channels=8)
    fill_module(Gf)
    Gf.to(dev).train()
    with torch.no_grad(), gd.precision(precision):
        y = Gf(torch.from_numpy(fx["x"]).to(dev)).double().cpu()
    ref = torch.from_numpy(fx["y"]).double()
    return {"fixture": "tests/golden/generator_8ch_16x16.npz (output of the reference generator)",
            "rel_l2": float((y - ref).norm() / ref.norm()), "rel_max": float((y - ref).abs().max() / ref.abs().max()),
            "north_star": "1e-3 (held by --precision fp32: 4e-6; --precision mixed: 6.5e-5 at full size; 16-bit operand modes: see profiles/r03_parity_attribution.json)"}


def launch_ranks(args) -> int:
    """``python bench.py --gpus N`` with no launcher on the command line: start N fresh worker processes of this script
    (one rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment) and wait for them.  This parent
    never touches the GPU (no HIP call before or after the spawn, no exec): rank 0 writes the JSON line to our stdout,
    the other ranks' stdout goes to stderr."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    try:
        while procs:
            for p in list(procs):
                code = p.poll()
                if code is None:
                    continue
                procs.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    for q in procs:            # one rank failed: the others would hang in their next collective
                        q.terminate()
            time.sleep(0.2)
    except BaseException:
        for q in procs:
            q.kill()
        raise
    return rc


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    cfg = args.config
    defaults = {2: (32, 128, "bf16"), 3: (32, 256, "bf16"), 4: (32, 256, "bf16"), 5: (1, 512, "fp16")}[cfg]
    args.batch = defaults[0] if args.batch is None else args.batch
    args.tile = defaults[1] if args.tile is None else args.tile
    args.precision = defaults[2] if args.precision is None else args.precision
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
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
    from gan_danet_amd.parallel import GradReducer, broadcast_module, world_size

    gd.set_precision(args.precision)
    B, T, Cin = args.batch, args.tile, args.channels
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    torch.manual_seed(1234)
    perc = None
    extra = {}

    if cfg == 2:
        # ---- BASELINE config 2: DANetAttention(64) forward on (B, 64, T, T) ----
        from gan_danet_amd.generator import DANetAttention
        C2 = 64
        A = DANetAttention(C2).to(dev)
        A.apply(gd.weights_init_normal)
        for n, p in A.named_parameters():
            if n.endswith("gamma"):
                p.data.fill_(0.1)
        broadcast_module(A)
        A.train()
        x = torch.randn(B, C2, T, T, device=dev, generator=gen)

        def step():
            with torch.no_grad():
                return A(x)
        workload = (f"DANetAttention({C2}) forward (PAM N={T * T} d_qk={C2 // 8} + CAM + fuse conv3x3/BN/ReLU) on "
                    f"({B},{C2},{T},{T}), batch {B}/GPU")
        metric = f"attention-forward samples/sec, DANetAttention(64) on {T}x{T}x64 feature maps"
    elif cfg == 5 and args.no_disc:
        # ---- BASELINE config 5 without the discriminator: 512x512 tiles, fp16 PAM operands; G forward + backward + AdamW ----
        G = gd.FlexibleUpsamplingModule(input_channels=Cin).to(dev)
        G.apply(gd.weights_init_normal)
        for n, p in G.named_parameters():
            if n.endswith("gamma"):
                p.data.fill_(0.1)
        broadcast_module(G)
        G.train()
        from gan_danet_amd import ops
        opt = gd.AdamW(G.parameters(), lr=2e-4, betas=(0.5, 0.999), weight_decay=1e-4, grad_scale=1.0 / world_size())
        red = GradReducer(G.parameters())
        x = torch.randn(B, Cin, T, T, device=dev, generator=gen)
        target = torch.randn(B, 1, 4 * T, 4 * T, device=dev, generator=gen)

        def step():
            opt.zero_grad(set_to_none=True)
            hr = G(x)
            loss = ops.weighted_sum([1.0, 1.0], [ops.mse_loss(hr, target), ops.tv_loss(hr, 1e-5)])
            loss.backward()
            red.reduce()
            opt.step()
            return loss.detach()
        workload = (f"generator forward + backward + AdamW (MSE + TV loss), {Cin}ch {T}x{T} -> {4 * T}x{4 * T} tiles, "
                    f"PAM N={T * T} tokens with fp16 MFMA operands, batch {B}/GPU (global {B * world}); discriminator "
                    f"EXCLUDED (Discriminator1.fc1 at this tile = 8.6e9 parameters = 137 GB with gradient + AdamW state)")
        metric = f"generator train samples/sec on {T}x{T} tiles (PAM N={T * T}), discriminator excluded"
    else:
        # ---- BASELINE config 3 / 4: the full G+D step (the metric's own configuration) ----
        G = gd.FlexibleUpsamplingModule(input_channels=Cin).to(dev)
        D = gd.Discriminator1().to(dev)
        with torch.no_grad():
            D(torch.zeros(1, 1, 4 * T, 4 * T, device=dev))          # materialise LazyLinear fc1
        G.apply(gd.weights_init_normal)
        D.apply(gd.weights_init_normal)
        for n, p in G.named_parameters():
            if n.endswith("gamma"):
                p.data.fill_(0.1)                                     # attention active (reference init is 0)
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
        # synthetic shard, resident in HBM before the timed region
        x = torch.randn(B, Cin, T, T, device=dev, generator=gen)
        target = torch.randn(B, 1, 4 * T, 4 * T, device=dev, generator=gen)
        last = {}

        def step():
            last["out"] = trainer.step(x, target, 0.5)
            return last["out"].loss_g
        workload = (f"full G+D train step, {Cin}ch {T}x{T} -> {4 * T}x{4 * T} tiles, batch {B}/GPU (global {B * world}), "
                    f"perceptual={'on' if perc is not None else 'OFF'}, SSIM evaluated, AdamW x2")
        if cfg == 5:
            nfc1 = D.fc1.weight.numel()
            workload += (f"; PAM N={T * T} tokens, fp16 MFMA operands in the fused PAM kernels (bf16 elsewhere); Discriminator1 "
                         f"INCLUDED (fc1 {nfc1 / 1e9:.2f}e9 parameters: weights + gradient replicated, AdamW state "
                         f"{'whole on this rank' if world == 1 else f'sharded 1/{world} per rank'})")
        metric = f"train samples/sec (G+D step) on {T}x{T} tiles"

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    kern.PROFILE.clear()
    kern.PROFILE_ON = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res_t = step()
    sync()
    dt = time.perf_counter() - t0
    kern.PROFILE_ON = False
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()

    finite = bool(torch.isfinite(res_t).all())
    if cfg in (3, 4) or (cfg == 5 and not args.no_disc):
        out = last["out"]
        extra = {"loss_d": out.loss_d.item(), "loss_g": out.loss_g.item()}
        finite = finite and bool(torch.isfinite(out.loss_d).all())
    elif cfg == 5 and args.no_disc:
        extra = {"loss": res_t.item()}

    # ---- roofline of the dominant kernel from the HIP-event brackets recorded around its launches ----
    # `traffic` = HBM bytes per launch from the PMC counters (FETCH_SIZE doubled per MI355X_MICROARCH.md + WRITE_SIZE),
    # measured by tools/run_profiles.sh at the commit named in the file, at exactly config 3's launch shape
    roof = None
    traffic, traffic_commit = {}, None
    tpath = os.path.join(ROOT, "profiles", "r03_pam_traffic.json")
    if os.path.exists(tpath) and cfg in (3, 4) and B == 32 and T == 256:
        with open(tpath) as f:
            tj = json.load(f)
        traffic, traffic_commit = tj.get("traffic_bytes_per_launch", {}), tj.get("commit")
    stats = kern.profile_summary()
    if stats:
        name, (n_launch, ms_avg, flops, nbytes) = max(stats.items(), key=lambda kv: kv[1][0] * kv[1][1])
        ach = flops / (ms_avg * 1e-3) / 1e12
        roof = {"bound": "mfma", "kernel": name, "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS,
                "unit": "TFLOP/s", "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic.get(name),
                "traffic_measured_at_commit": traffic_commit,
                "launches": n_launch, "avg_ms": round(ms_avg, 3),
                "algorithmic_flop_per_launch": flops,
                "all": {k: {"launches": v[0], "avg_ms": round(v[1], 3),
                            "tflops": round(v[2] / (v[1] * 1e-3) / 1e12, 1)} for k, v in stats.items()}}
        if "pam_flash_fwd" in stats and "pam_flash_bwd" in stats:      # the "256x256 PAM GEMM" utilisation (fwd + bwd)
            tf = sum(stats[k][0] * stats[k][1] for k in ("pam_flash_fwd", "pam_flash_bwd")) * 1e-3
            ff = sum(stats[k][0] * stats[k][2] for k in ("pam_flash_fwd", "pam_flash_bwd"))
            roof["pam_fwd_bwd"] = {"tflops": round(ff / tf / 1e12, 1), "frac": round(ff / tf / 1e12 / PEAK_BF16_TFLOPS, 4)}

    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args.cpu_baseline_steps)
        res = {
            "metric": metric,
            "value": round(args.steps * B * world / dt, 3),
            "unit": "samples/s",
            "n_gpus": world,
            "ranks": {"world_size": dist.get_world_size() if world > 1 else 1,
                      "backend": (dist.get_backend() if world > 1 else None),
                      "devices": ("one rank per GPU (LOCAL_RANK)" if not one_gpu_rehearsal or world == 1
                                  else "REHEARSAL: all ranks on cuda:0")},
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.precision,
            "data": "synthetic (randn tiles, random-init weights" + (", random-init VGG19 for the perceptual term)" if perc is not None else ")"),
            "config": {"workload": workload, "survey_config": cfg,
                       "per_gpu_batch": B, "global_batch": B * world, "tile": T, "parallelism": f"dp{world}",
                       "storage": {"fp32": "fp32 everywhere (exact f32 MFMA)",
                                   "mixed": "fp32 activations/weights in the generator, split-bf16 pixel-major activations in the "
                                            "discriminator trunk and VGG; fused PAM on IEEE fp16 operands; every other conv / GEMM on "
                                            "split-bf16 operands (hi*hi + lo*hi + hi*lo, ~2^-16); generator stem, nn.Linear, CAM Gram on "
                                            "the exact f32 MFMA; fp32 accumulate everywhere"}.get(
                           args.precision, "fp32 activations/weights, 16-bit MFMA operands (" + args.precision +
                           "; generator stem conv exact, 1x1-shaped products on split-bf16 operands), fp32 accumulate")},
            **extra, "finite": finite,
            "g_out_rel_err": g_out_rel_err(gd, dev, args.precision),
            "roofline": roof,
            "cpu_baseline": cpu,
            "max_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 1),
        }
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
