"""GPU, 2 processes: the data-parallel G+D step through ``GanTrainer`` on the real HIP modules.

Two FRESH child interpreters (tests/ddp_worker.py; gloo, both ranks on cuda:0 -- the box has one GPU) each run
``GanTrainer.step`` on their half of a 4-sample batch.  Checked:
  (a) the result equals a single-process emulation of two ranks -- two trainers on the two shards whose gradients are
      summed by hand between ``d_backward`` / ``g_backward`` and the optimiser steps (per-shard BatchNorm statistics and
      per-shard TV loss, i.e. DDP semantics);
  (b) the replicas are BIT-IDENTICAL after the steps (same summed gradients, same AdamW arithmetic);
  (c) BN running statistics stay per-replica (they differ between ranks: each saw its own shard).
No scaling is measured here: SCALE_rNN.json is the driver's."""
import os
import socket
import subprocess
import sys

import pytest
import torch

from fill import fill_module, seeded
from gpu_util import DEV, rell2

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _emulate_two_ranks(steps):
    import gan_danet_amd as gd
    gb = 4
    x, tgt = seeded((gb, 8, 16, 16), 21).to(DEV), seeded((gb, 1, 64, 64), 22).to(DEV)
    trs = []
    with gd.precision("fp32"):
        for r in range(2):
            G = gd.FlexibleUpsamplingModule(input_channels=8).to(DEV)
            D = gd.Discriminator1().to(DEV)
            with torch.no_grad():
                D(tgt[:1])
            fill_module(G), fill_module(D)
            G.train(), D.train()
            trs.append(gd.GanTrainer(G, D, perceptual=None, external_world=2))

        def exchange(params_a, params_b):
            for pa, pb in zip(params_a, params_b):
                if pa.grad is None:
                    continue
                s = pa.grad + pb.grad              # the all-reduce SUM (plumbing of the emulation)
                pa.grad.copy_(s), pb.grad.copy_(s)

        for _ in range(steps):
            sts = [t.d_backward(x[2 * r:2 * r + 2], tgt[2 * r:2 * r + 2]) for r, t in enumerate(trs)]
            exchange(list(trs[0].D.parameters()), list(trs[1].D.parameters()))
            for t in trs:
                t.opt_d.step()
            for r, t in enumerate(trs):
                t.g_backward(sts[r], tgt[2 * r:2 * r + 2], 0.5)
            exchange(list(trs[0].G.parameters()), list(trs[1].G.parameters()))
            for t in trs:
                t.opt_g.step()
    torch.cuda.synchronize()
    return trs, [s["loss_g"].item() for s in sts]


def _run_two_ranks(outdir, steps, world=2, **extra_env):
    os.makedirs(outdir, exist_ok=True)
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", DDP_PREC="fp32", **extra_env)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ddp_worker.py"), str(r), str(world), str(port),
                               str(outdir), str(steps)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(world)]
    logs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(out.decode(errors="replace"))
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
    if world == 1:
        return torch.load(os.path.join(outdir, "rank0.pt"), weights_only=True), logs[0]
    return (torch.load(os.path.join(outdir, "rank0.pt"), weights_only=True),
            torch.load(os.path.join(outdir, "rank1.pt"), weights_only=True))


def test_two_rank_ganstep_matches_emulation_and_replicas_stay_identical(tmp_path):
    steps = 2
    # deterministic mode (no atomic split reductions): two separate runs can then be compared bit for bit
    r0, r1 = _run_two_ranks(str(tmp_path / "plain"), steps, DDP_DETERMINISTIC="1")
    # fc1 (and every other tensor >= 1 MiB) through reduce-scatter -> sharded AdamW -> all-gather: the same arithmetic
    # per element, so the result must not change by a single bit
    s0, s1 = _run_two_ranks(str(tmp_path / "sharded"), steps, DDP_DETERMINISTIC="1", DDP_SHARD_BYTES=str(1 << 20))
    for net in ("G", "D"):
        for k in r0[net]:
            assert torch.equal(r0[net][k], s0[net][k]), f"sharded optimiser path changed {net}.{k}"
            assert torch.equal(r1[net][k], s1[net][k]), f"sharded optimiser path changed {net}.{k} (rank 1)"
    # (b) replicas bit-identical in every PARAMETER; (c) BN running statistics are per replica
    bn_keys = [k for k in r0["G"] if k.endswith("running_mean") or k.endswith("running_var")]
    for net in ("G", "D"):
        for k in r0[net]:
            if k in bn_keys or k.endswith("num_batches_tracked"):
                continue
            assert torch.equal(r0[net][k], r1[net][k]), f"replicas differ in {net}.{k}"
    assert any(not torch.equal(r0["G"][k], r1["G"][k]) for k in bn_keys), "BN statistics should be per-replica (DDP semantics)"
    # (a) against the single-process two-shard emulation
    trs, _ = _emulate_two_ranks(steps)
    for net, mod in (("G", trs[0].G), ("D", trs[0].D)):
        for k, v in mod.state_dict().items():
            if k.endswith("num_batches_tracked") or k.endswith("key.bias"):
                continue       # key.bias: analytically ZERO gradient (softmax shift invariance) -> pure round-off,
                               # which AdamW's sign-like first steps turn into +-lr moves in either run
            e = rell2(v, r0[net][k]) if v.dtype.is_floating_point else float(not torch.equal(v.cpu(), r0[net][k]))
            # AdamW's early steps are sign-like: elements whose gradient sits at fp32 round-off level (the order in
            # which atomics / gloo sum differs between the two runs) may move by +-lr either way
            assert e <= 2e-3, f"{net}.{k}: 2-process vs emulation rel err {e:.2e}"
    # rank 1's BN statistics equal the emulation's second trainer
    for k in bn_keys:
        assert rell2(trs[1].G.state_dict()[k], r1["G"][k]) <= 1e-5, k


def test_rccl_backend_world1_sharded_path_equals_plain_step(tmp_path):
    """The "nccl" (= RCCL) branches of parallel.py -- ``reduce_scatter_tensor`` into the gradient shard, the 1/world
    AdamW, the async ``all_gather_into_tensor`` into the weight, the bucketed ``all_reduce``s -- executed ON RCCL
    (ADVICE r02 / VERDICT r02 next #3).  RCCL takes one rank per device and this box has one GPU, so the world is ONE
    rank with GD_FORCE_COLLECTIVES=1 (every collective still goes through librccl).  Checked: the collectives really
    ran on the nccl backend (counted in the worker), and the result equals the plain single-process step bit for bit
    (deterministic mode) -- the sharded route changes no arithmetic."""
    steps = 2
    plain, _ = _run_two_ranks(str(tmp_path / "plain"), steps, world=1, DDP_DETERMINISTIC="1")
    rccl, log = _run_two_ranks(str(tmp_path / "rccl"), steps, world=1, DDP_DETERMINISTIC="1", DDP_BACKEND="nccl",
                               GD_FORCE_COLLECTIVES="1", DDP_SHARD_BYTES=str(1 << 20))
    assert "RCCL collectives executed" in log, log[-2000:]
    for net in ("G", "D"):
        for k in plain[net]:
            assert torch.equal(plain[net][k], rccl[net][k]), f"RCCL sharded path changed {net}.{k}"
    assert plain["loss_g"] == rccl["loss_g"] and plain["loss_d"] == rccl["loss_d"]


def test_sync_bn_two_ranks_equal_one_device_on_the_global_batch(tmp_path):
    """SyncBN flag (optional): two ranks on two samples each, BatchNorm statistics and dy sums reduced over
    the ranks and the TV term scaled to the global batch, take the SAME G+D steps as one device on the four samples --
    parameters, BN running statistics (identical on both ranks now) and the generator output.  Tolerance = the
    emulation test's: AdamW's early steps are sign-like, so elements whose gradient sits at fp32 round-off level may
    move by +-lr either way when the summation order changes."""
    import gan_danet_amd as gd
    steps = 2
    r0, r1 = _run_two_ranks(str(tmp_path / "sync"), steps, DDP_DETERMINISTIC="1", DDP_SYNC_BN="1")
    for net in ("G", "D"):
        for k in r0[net]:
            if not k.endswith("num_batches_tracked"):
                assert torch.equal(r0[net][k], r1[net][k]), f"SyncBN replicas differ in {net}.{k}"
    gb = 4
    x, tgt = seeded((gb, 8, 16, 16), 21).to(DEV), seeded((gb, 1, 64, 64), 22).to(DEV)
    gd.set_deterministic(True)
    try:
        with gd.precision("fp32"):
            G = gd.FlexibleUpsamplingModule(input_channels=8).to(DEV)
            D = gd.Discriminator1().to(DEV)
            with torch.no_grad():
                D(tgt[:1])
            fill_module(G), fill_module(D)
            G.train(), D.train()
            tr = gd.GanTrainer(G, D, perceptual=None)
            for _ in range(steps):
                tr.step(x, tgt, 0.5)
        torch.cuda.synchronize()
    finally:
        gd.set_deterministic(False)
    for net, mod in (("G", G), ("D", D)):
        for k, v in mod.state_dict().items():
            if k.endswith("num_batches_tracked") or k.endswith("key.bias"):
                continue
            e = rell2(v, r0[net][k])
            assert e <= 2e-3, f"{net}.{k}: SyncBN 2 x 2 samples vs one device on 4 samples: rel err {e:.2e}"
    bn_keys = [k for k in r0["G"] if k.endswith("running_mean") or k.endswith("running_var")]
    for k in bn_keys:        # (the second step's statistics see parameters that already differ by the AdamW sign-step noise above)
        assert rell2(G.state_dict()[k], r0["G"][k]) <= 1e-3, f"running statistic {k}"
