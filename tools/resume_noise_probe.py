"""Run-to-run spread of one resumed G+D step against the uninterrupted run (tests/test_gpu_checkpoint.py): how many generator
tensors differ beyond fp32 round-off, and by how much, when the order of the fp32 atomics changes.  TRIALS=n python tools/resume_noise_probe.py"""
import sys, os, tempfile
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
import gan_danet_amd as gd
from gan_danet_amd import checkpoint as C
from gpu_util import DEV, rell2, seeded
def make(seed=0):
    torch.manual_seed(seed)
    G, D = gd.FlexibleUpsamplingModule(input_channels=8).to(DEV), gd.Discriminator1().to(DEV)
    with torch.no_grad():
        D(torch.zeros(1, 1, 64, 64, device=DEV))
    G.apply(gd.weights_init_normal), D.apply(gd.weights_init_normal)
    return G, D
if os.environ.get("DETERMINISTIC") == "1":
    gd.set_deterministic(True)
x, tgt = seeded((2, 8, 16, 16), 171).to(DEV), seeded((2, 1, 64, 64), 172).to(DEV)
for env in (os.environ.get("TAG", ""),):
    pass
for trial in range(int(os.environ.get("TRIALS", "6"))):
    G, D = make()
    tr = gd.GanTrainer(G, D, None)
    tr.step(x, tgt, 0.25)
    path = tempfile.mktemp()
    C.save_training_state(path, tr, epoch=1, schedulers=[], extra={})
    oa = tr.step(x, tgt, 0.5)
    wa = {k: v.clone() for k, v in G.state_dict().items()}
    G2, D2 = make(123)
    tr2 = gd.GanTrainer(G2, D2, None)
    C.load_training_state(path, tr2, [])
    ob = tr2.step(x, tgt, 0.5)
    ld = abs(ob.loss_d.item() - oa.loss_d.item()) / abs(oa.loss_d.item()); lg = abs(ob.loss_g.item() - oa.loss_g.item()) / abs(oa.loss_g.item())
    errs = sorted(((rell2(v, wa[k].cpu()), k) for k, v in G2.state_dict().items() if v.dtype.is_floating_point), reverse=True)
    nbad = sum(e > 1e-5 for e, _ in errs)
    dmax = max((v.cpu() - wa[k].cpu()).abs().max().item() for k, v in G2.state_dict().items() if v.dtype.is_floating_point)
    w2 = max(rell2(v, wa[k].cpu()) for k, v in G2.state_dict().items() if v.dtype.is_floating_point and v.dim() >= 2)
    print(trial, f"loss_d {ld:.1e} loss_g {lg:.1e}", f"max rel-L2 over weight tensors (dim >= 2): {w2:.1e};", "tensors > 1e-5:", nbad, "of", len(errs), "max |diff|", f"{dmax:.2e}", [(f"{e:.1e}", k) for e, k in errs[:2]], flush=True)
    os.remove(path)
