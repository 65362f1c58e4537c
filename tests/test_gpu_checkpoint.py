"""f4: checkpoint / ensemble layer.  A run resumed from ``save_training_state`` continues where it left off -- to
fp32 round-off, not bit-for-bit: the split-K GEMMs (fc1, weight gradients) combine their partial sums with fp32 atomics
whose order varies from launch to launch; the generator file carries the reference's keys; ensemble members are
independent replicas."""
import os

import pytest
import torch

from fill import fill_module
from gpu_util import DEV, assert_close, rell2, seeded

pytestmark = pytest.mark.gpu


def _make(gd, seed=0):
    torch.manual_seed(seed)
    G, D = gd.FlexibleUpsamplingModule(input_channels=8).to(DEV), gd.Discriminator1().to(DEV)
    with torch.no_grad():
        D(torch.zeros(1, 1, 64, 64, device=DEV))
    G.apply(gd.weights_init_normal), D.apply(gd.weights_init_normal)
    return G, D


def test_resume_matches_uninterrupted_run_and_generator_file_has_reference_keys(tmp_path, golden_dir):
    import gan_danet_amd as gd
    from gan_danet_amd import checkpoint as C
    from torch.optim.lr_scheduler import CosineAnnealingWarmRestarts
    x, tgt = seeded((2, 8, 16, 16), 171).to(DEV), seeded((2, 1, 64, 64), 172).to(DEV)

    def sched(tr):   # GAN_DANet_train.ipynb:L186-187
        return [CosineAnnealingWarmRestarts(tr.opt_d, T_0=10, T_mult=2, eta_min=1e-6),
                CosineAnnealingWarmRestarts(tr.opt_g, T_0=10, T_mult=2, eta_min=1e-6)]

    G, D = _make(gd)
    tr = gd.GanTrainer(G, D, None)
    sc = sched(tr)
    tr.step(x, tgt, 0.25)
    for s_ in sc:
        s_.step()
    path = str(tmp_path / "state.pt")
    C.save_training_state(path, tr, epoch=1, schedulers=sc, extra={"note": "after epoch 1"})
    out_a = tr.step(x, tgt, 0.5)
    wa = {k: v.clone() for k, v in G.state_dict().items()}

    G2, D2 = _make(gd, seed=123)                       # different init: everything must come from the file
    tr2 = gd.GanTrainer(G2, D2, None)
    sc2 = sched(tr2)
    info = C.load_training_state(path, tr2, sc2)
    assert info == {"epoch": 1, "extra": {"note": "after epoch 1"}}
    assert tr2.opt_g.param_groups[0]["lr"] == tr.opt_g.param_groups[0]["lr"]
    out_b = tr2.step(x, tgt, 0.5)
    # the losses of the resumed step: identical in 37 of 40 runs of tools/resume_noise_probe.py, 4.7e-7 / 3.6e-5 apart in the
    # others (the split-K Gram of CAM sums with fp32 atomics; a last-bit difference in the generated image can flip a bf16
    # rounding in the discriminator's pixel-major trunk) -- asserted at 1e-3
    assert_close(out_b.loss_d.view(1), out_a.loss_d.view(1).cpu(), 1e-3, "loss_d after resume")
    assert_close(out_b.loss_g.view(1), out_a.loss_g.view(1).cpu(), 1e-3, "loss_g after resume")
    # Not bit-for-bit here (test_resume_is_bit_exact_in_deterministic_mode is): the order of the fp32 atomics varies between
    # launches, and AdamW divides by sqrt(v).  Gradients that are analytically ZERO -- PAM's key bias (a constant added to
    # every key shifts each softmax row by a constant), every conv bias in front of a BatchNorm -- are pure summation noise,
    # their normalised update a coin flip of size lr.  tools/resume_noise_probe.py over 40 runs: three in four bit-identical
    # to 1e-8, the others in a handful of discrete outcomes with up to 47 of 143 tensors beyond round-off, the weight tensors
    # within 1.7e-4, no element further than 1.8 lr; 40 of 40 identical in deterministic mode (no race behind it).
    # Asserted: every element within one sign-flipped AdamW step (|update| <= 1.06 lr at step 2 with betas (0.5, 0.999)),
    # every weight tensor within 1e-3.
    lr_g = tr.opt_g.param_groups[0]["lr"]
    for k, v in G2.state_dict().items():
        if v.dtype.is_floating_point:
            d = (v.cpu() - wa[k].cpu()).abs().max().item()
            assert d <= 2.2 * lr_g + 1e-7, f"{k}: |resumed - uninterrupted| = {d:.3e} > a sign-flipped AdamW step (lr {lr_g:.1e})"
            if v.dim() >= 2:
                assert_close(v, wa[k].cpu(), 1e-3, k, rell2)
        else:
            assert torch.equal(v, wa[k]), k

    gpath = str(tmp_path / "best_model.pth")
    C.save_generator(G2, gpath)
    sd = torch.load(gpath, map_location="cpu", weights_only=True)
    with open(os.path.join(golden_dir, "generator_state_dict_keys.txt")) as f:
        ref_keys = [ln.split(" ")[0] for ln in f if ln.strip()]
    G46 = gd.FlexibleUpsamplingModule(input_channels=46)
    assert list(G46.state_dict().keys()) == ref_keys and list(sd.keys()) == ref_keys


def test_early_stopping_rule_and_ensemble_helpers(tmp_path):
    import gan_danet_amd as gd
    from gan_danet_amd import checkpoint as C
    G, D = _make(gd)
    es = C.EarlyStopping(patience=2, min_delta=0.1, path=str(tmp_path / "best.pth"))
    assert es.step(1.0, G) is False                    # improvement: saved
    best = {k: v.clone() for k, v in G.state_dict().items()}
    with torch.no_grad():
        G.final.bias.add_(1.0)
    assert es.step(0.95, G) is False                   # within min_delta: first strike
    assert es.step(0.97, G) is True                    # second strike: stop, best weights restored
    assert torch.equal(G.final.bias, best["final.bias"])
    assert [C.member_seed(i) for i in range(3)] == [42, 43, 44]
    assert C.members_of_rank(5, 1, 4) == [1] and C.members_of_rank(5, 0, 4) == [0, 4]
    assert C.member_path("ens", 0).endswith("best_model_member_1.pth")
    # independent replicas: two members from different seeds, one step each with reduce_gradients=False
    x, tgt = seeded((1, 8, 16, 16), 173).to(DEV), seeded((1, 1, 64, 64), 174).to(DEV)
    members = []
    for i in range(2):
        C.set_seed(C.member_seed(i))
        Gi, Di = _make(gd, seed=C.member_seed(i))
        gd.GanTrainer(Gi, Di, None, reduce_gradients=False).step(x, tgt, 0.5)
        members.append(Gi.eval())
    mean, std = C.predict_ensemble(members, x)
    assert tuple(mean.shape) == (1, 1, 64, 64) and torch.isfinite(mean).all() and (std > 0).any()


def test_resume_is_bit_exact_in_deterministic_mode(tmp_path):
    """gd.set_deterministic(True) (no atomic split reductions; PAM dQ through bf16 parts): a run resumed from the
    full-state checkpoint continues BIT FOR BIT like the uninterrupted one (bf16 mode, the fused PAM kernels)"""
    import gan_danet_amd as gd
    from gan_danet_amd import checkpoint as C
    x, tgt = seeded((2, 8, 16, 16), 171).to(DEV), seeded((2, 1, 64, 64), 172).to(DEV)
    gd.set_deterministic(True)
    try:
        with gd.precision("bf16"):
            G, D = _make(gd)
            tr = gd.GanTrainer(G, D, None)
            tr.step(x, tgt, 0.25)
            path = str(tmp_path / "state.pt")
            C.save_training_state(path, tr, epoch=1)
            out_a = tr.step(x, tgt, 0.5)
            G2, D2 = _make(gd, seed=123)
            tr2 = gd.GanTrainer(G2, D2, None)
            C.load_training_state(path, tr2)
            out_b = tr2.step(x, tgt, 0.5)
    finally:
        gd.set_deterministic(False)
    assert torch.equal(out_a.loss_g, out_b.loss_g) and torch.equal(out_a.loss_d, out_b.loss_d)
    for (k, a), b in zip(G.state_dict().items(), G2.state_dict().values()):
        assert torch.equal(a, b), k
    for (k, a), b in zip(D.state_dict().items(), D2.state_dict().values()):
        assert torch.equal(a, b), k
