"""GPU parity, step level: the full G+D update against the 3-step trajectory recorded from the reference
modules + torch.optim.AdamW (tests/golden/trajectory_3steps.npz), and the smoke entry point."""
import pytest
import torch

from fill import fill_module
from gpu_util import DEV, assert_close, load_golden, rell2

pytestmark = pytest.mark.gpu


def test_three_step_trajectory_fp32(golden_dir):
    import gan_danet_amd as gd
    fx = load_golden(golden_dir, "trajectory_3steps")
    G = gd.FlexibleUpsamplingModule(input_channels=8)
    D = gd.Discriminator1()
    fill_module(G)
    G.to(DEV).train()
    D.to(DEV).train()
    x, tgt = fx["x"].to(DEV), fx["target"].to(DEV)
    with gd.precision("fp32"):
        with torch.no_grad():
            D(tgt)
        fill_module(D)
        tr = gd.GanTrainer(G, D, perceptual=None)
        for i in range(3):
            out = tr.step(x, tgt, 0.5)
            ld, lg = out.loss_d.item(), out.loss_g.item()
            assert abs(ld - fx["loss_d"][i].item()) <= 1e-3 * abs(fx["loss_d"][i].item()) + 1e-6, (i, ld)
            # step 1 is a pure forward; later steps inherit the ill-conditioned attention gradients
            tol = 1e-4 if i == 0 else 3e-2
            assert abs(lg - fx["loss_g"][i].item()) <= tol * abs(fx["loss_g"][i].item()), (i, lg)
            assert abs((1 - out.parts["ssim"].item()) - fx["ssim"][i].item()) <= 2e-3
            gn = float(torch.sqrt(sum((p.detach().double() ** 2).sum() for p in G.parameters())))
            assert abs(gn - fx["g_norm"][i].item()) <= 1e-5 * gn
            dn = float(torch.sqrt(sum((p.detach().double() ** 2).sum() for p in D.parameters())))
            assert abs(dn - fx["d_norm"][i].item()) <= 1e-5 * dn
        # AdamW's first steps are sign-like (g/|g|): elements whose gradient is at round-off level move by
        # +-lr in either implementation, so outputs after 3 steps agree to ~5e-2, not to round-off.  Measured
        # 6.2e-2 (profiles/r02_parity_report.json; the CPU oracle itself is held to 2e-2 against the same fixture)
        assert_close(out.hr, fx["hr_last"], 1.0e-1, "hr after 3 steps", rell2)
        assert_close(G.final.weight, fx["final_w"], 5e-4, "final.weight", rell2)  # same sign-like-update effect
    # D weight grads are not formed in the G step, G's are
    assert all(p.grad is not None for p in G.parameters())


def test_step_bf16_with_perceptual_runs_and_is_finite():
    import warnings
    import gan_danet_amd as gd
    torch.manual_seed(0)
    G = gd.FlexibleUpsamplingModule(input_channels=8).to(DEV)
    D = gd.Discriminator1().to(DEV)
    x = torch.randn(2, 8, 16, 16, device=DEV)
    tgt = torch.randn(2, 1, 64, 64, device=DEV)
    with torch.no_grad():
        D(tgt)
    G.apply(gd.weights_init_normal)
    D.apply(gd.weights_init_normal)
    for n, p in G.named_parameters():
        if n.endswith("gamma"):
            p.data.fill_(0.1)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        perc = gd.PerceptualLoss(pretrained=False, device=DEV)
    tr = gd.GanTrainer(G, D, perceptual=perc)
    with gd.precision("bf16"):
        losses = [tr.step(x, tgt, 0.5) for _ in range(3)]
    for o in losses:
        assert torch.isfinite(o.loss_d).all() and torch.isfinite(o.loss_g).all()
    assert losses[-1].loss_g.item() < losses[0].loss_g.item() * 1.5


def test_smoke_entry():
    import __graft_entry__ as ge
    ge.smoke()
