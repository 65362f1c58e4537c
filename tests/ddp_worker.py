"""Child process of tests/test_gpu_ddp.py: one rank of a 2-rank data-parallel G+D step on the REAL HIP modules.
Fresh interpreter (nothing GPU-related inherited); both ranks share cuda:0; torch.distributed over gloo (this
one-GPU box has no second device for RCCL; the collective API calls are the same ones the nccl backend serves).
    python tests/ddp_worker.py RANK WORLD PORT OUTDIR STEPS"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
rank, world, port, outdir, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], int(sys.argv[5])
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = str(port)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

backend = os.environ.get("DDP_BACKEND", "gloo")
if backend == "nccl":          # RCCL: one rank per device, so only world = 1 fits this box (GD_FORCE_COLLECTIVES=1)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
else:
    dist.init_process_group("gloo", rank=rank, world_size=world)
import gan_danet_amd as gd  # noqa: E402
from fill import fill_module, seeded  # noqa: E402
from gan_danet_amd.parallel import broadcast_module, shard_batch  # noqa: E402

dev = torch.device("cuda:0")
gd.set_deterministic(os.environ.get("DDP_DETERMINISTIC", "0") == "1")     # bitwise run-to-run comparisons
gb = 4
x, tgt = seeded((gb, 8, 16, 16), 21).to(dev), seeded((gb, 1, 64, 64), 22).to(dev)
torch.manual_seed(100 + rank)                        # replicas start DIFFERENT on purpose: broadcast must fix it
G = gd.FlexibleUpsamplingModule(input_channels=8).to(dev)
D = gd.Discriminator1().to(dev)
prec = os.environ.get("DDP_PREC", "fp32")
with gd.precision(prec):
    with torch.no_grad():
        D(tgt[:1])
    if rank == 0:
        fill_module(G)
        fill_module(D)
    broadcast_module(G)
    broadcast_module(D)
    G.train(), D.train()
    shard_bytes = int(os.environ.get("DDP_SHARD_BYTES", "0"))
    sync_bn = os.environ.get("DDP_SYNC_BN", "0") == "1"
    tr = gd.GanTrainer(G, D, perceptual=None, shard_bytes=shard_bytes, sync_bn=sync_bn,
                       tv_global_batch_semantics=sync_bn)   # built AFTER init_process_group: world = 2
    if shard_bytes:
        assert any(sp.p is D.fc1.weight for sp in tr.sharded), "fc1 should take the reduce-scatter / sharded-AdamW path"
    if backend == "nccl" and shard_bytes:
        # count the RCCL collectives that actually run
        calls = {"rs": 0, "ag": 0, "ar": 0}
        for nm, key in (("reduce_scatter_tensor", "rs"), ("all_gather_into_tensor", "ag"), ("all_reduce", "ar")):
            orig = getattr(dist, nm)

            def wrapped(*a, _o=orig, _k=key, **kw):
                calls[_k] += 1
                return _o(*a, **kw)
            setattr(dist, nm, wrapped)
    sl = shard_batch(gb, world, rank)
    outs = [tr.step(x[sl], tgt[sl], 0.5) for _ in range(steps)]
tr.sync_params()
torch.cuda.synchronize()
if backend == "nccl" and shard_bytes:
    assert dist.get_backend() == "nccl"
    assert calls["rs"] >= steps and calls["ag"] >= steps and calls["ar"] >= 2 * steps, calls
    print("RCCL collectives executed:", calls, flush=True)
if os.environ.get("DDP_SHARD_BYTES"):
    # the optimiser state of a sharded tensor is 1/world of it on each rank; state_dict() returns the full tensors
    st = tr.opt_d.state[D.fc1.weight]
    assert st["exp_avg"].numel() == D.fc1.weight.numel() // world
    full = tr.opt_d.state_dict()
    idx = [i for i, p in enumerate(D.parameters()) if p is D.fc1.weight][0]
    assert full["state"][idx]["exp_avg"].shape == D.fc1.weight.shape
state = {"G": {k: v.detach().cpu() for k, v in G.state_dict().items()},
         "D": {k: v.detach().cpu() for k, v in D.state_dict().items()},
         "loss_d": [o.loss_d.item() for o in outs], "loss_g": [o.loss_g.item() for o in outs]}
torch.save(state, os.path.join(outdir, f"rank{rank}.pt"))
dist.barrier()
dist.destroy_process_group()
