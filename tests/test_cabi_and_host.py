"""CPU: the C-ABI library loads and exports every symbol include/gandanet.h declares (no compute calls),
the ctypes mirror of the descriptor structs matches, the drop-in import paths work, the product modules refuse
CPU tensors (no fallback), and the host-side logic (sharding, structural pin from the reference's graph dump)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "gandanet.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(gd_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    from gan_danet_amd import _lib
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 40
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in gandanet.h but not exported"
    # and the binding knows every declared symbol (a stale binding would silently skip one)
    missing = [n for n in declared if n not in _lib.SIGNATURES]
    assert not missing, f"no ctypes prototype for {missing}"
    assert lib.gd_version() == 100
    assert lib.gd_sizeof_conv_desc() == ctypes.sizeof(_lib.ConvDesc)
    assert lib.gd_sizeof_gemm_nt_desc() == ctypes.sizeof(_lib.GemmNTDesc)


def test_argument_errors_are_reported_not_thrown():
    from gan_danet_amd import _lib
    lib = _lib.load()
    rc = lib.gd_conv2d(None, None)          # rejected on the host before any launch
    assert rc == -1
    assert "null descriptor" in _lib.last_error()


def test_drop_in_import_paths_and_names():
    import model
    import models
    ref_names = ["CBAMBlock", "FlexibleUpsamplingModule", "OriginalRelationshipLearner", "SqueezeExcitation",
                 "Discriminator1", "SRGAND", "PerceptualLoss", "SSIM", "TVLoss", "weights_init_normal"]
    assert sorted(models.__all__) == sorted(ref_names)          # models/__init__.py:12-23 of the reference
    assert sorted(model.__all__) == sorted(ref_names)
    from models.generator import CAMModule, DANetAttention, PAMModule  # noqa: F401  (generator.py importables)
    G = models.FlexibleUpsamplingModule(input_channels=46, attention_type="senet")  # alias, with a warning
    assert G.feature_channels == [160, 176, 184]
    assert sum(p.numel() for p in G.parameters()) == 2_271_993   # SURVEY 2.2 [probed on the reference]
    assert sum(p.numel() for p in models.FlexibleUpsamplingModule(input_channels=8).parameters()) == 2_250_105


def test_no_cpu_fallback():
    import gan_danet_amd as gd
    from gan_danet_amd._lib import GandanetError
    G = gd.FlexibleUpsamplingModule(input_channels=8)
    with pytest.raises(GandanetError, match="no CPU fallback"):
        G(torch.zeros(1, 8, 8, 8))
    with pytest.raises(GandanetError):
        gd.TVLoss(1.0)(torch.zeros(1, 1, 8, 8))
    # and no file of the product package (or the drop-in shims) imports the oracle
    pkg = os.path.join(ROOT, "gan-danet_amd")
    files = [os.path.join(dp, f) for dp, _, fs in os.walk(pkg) for f in fs if f.endswith(".py")]
    files += [os.path.join(ROOT, f) for f in ("gan_danet_amd.py", "model.py", os.path.join("models", "__init__.py"))]
    for path in files:
        src = open(path).read()
        assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{path} imports the oracle"


def test_structural_pin_against_reference_graph_dump(golden_dir):
    """parameter shapes read off the reference's `Generator` autograd-graph dump (input_channels = 46):
    stem (64,46,3,3), PAM q/k (20,160,1,1), value (160,160,1,1), fuse (160,320,3,3), adjust (64,184,1,1),
    final (1,64,3,3)  (Generator:129-130, 246-247, 269-270, 331-332, 815-816, 840-841)."""
    import gan_danet_amd as gd
    G = gd.FlexibleUpsamplingModule(input_channels=46)
    sd = G.state_dict()
    assert tuple(sd["initial.0.weight"].shape) == (64, 46, 3, 3)
    assert tuple(sd["attention_modules.0.position_attention.query.weight"].shape) == (20, 160, 1, 1)
    assert tuple(sd["attention_modules.0.position_attention.key.weight"].shape) == (20, 160, 1, 1)
    assert tuple(sd["attention_modules.0.position_attention.value.weight"].shape) == (160, 160, 1, 1)
    assert tuple(sd["attention_modules.0.fuse.0.weight"].shape) == (160, 320, 3, 3)
    assert tuple(sd["channel_adjust.0.weight"].shape) == (64, 184, 1, 1)
    assert tuple(sd["final.weight"].shape) == (1, 64, 3, 3)
    with open(os.path.join(golden_dir, "generator_state_dict_keys.txt")) as f:
        ref = [ln.strip() for ln in f if ln.strip()]
    assert [f"{k} {tuple(v.shape)}" for k, v in sd.items()] == ref


def test_shard_batch_partitions_exactly():
    from gan_danet_amd.parallel import shard_batch
    for gb, world in ((256, 8), (32, 1), (10, 4), (7, 8)):
        seen = []
        for rk in range(world):
            s = shard_batch(gb, world, rk)
            seen += list(range(gb))[s]
        assert seen == list(range(gb))


def test_oracle_reference_state_dict_loads_into_product_modules():
    """a checkpoint written by the reference architecture (here: the oracle, key-identical to the reference)
    loads into the HIP-backed modules and back"""
    import gan_danet_amd as gd
    from oracle import modules as OM
    Go = OM.FlexibleUpsamplingModule(input_channels=8)
    G = gd.FlexibleUpsamplingModule(input_channels=8)
    missing, unexpected = G.load_state_dict(Go.state_dict(), strict=True)
    assert not missing and not unexpected
    back = OM.FlexibleUpsamplingModule(input_channels=8)
    back.load_state_dict(G.state_dict(), strict=True)


def test_batch_plan_gives_every_rank_the_same_steps():
    """ranks must not run different numbers of steps per epoch (their collectives would pair across
    steps).  n=100, global batch 32, world 8: four steps would leave ranks 4-7 without a shard in the last one."""
    from gan_danet_amd.data import DeviceTileDataset

    class _Fake:
        def __len__(self):
            return 100
    plans = [DeviceTileDataset.batch_plan(_Fake(), 32, r, 8) for r in range(8)]
    assert len({len(p) for p in plans}) == 1 and len(plans[0]) == 3      # 4-sample tail (< 8 ranks) dropped
    sizes = {hi - lo for p in plans for lo, hi in p}
    assert sizes == {4}
    # shards of one step tile the global batch without overlap
    for step in range(3):
        spans = sorted(p[step] for p in plans)
        assert spans[0][0] == 32 * step and all(spans[i][1] == spans[i + 1][0] for i in range(7))
    # world 1 keeps the reference loader's ragged last batch
    assert DeviceTileDataset.batch_plan(_Fake(), 32, 0, 1)[-1] == (96, 100)
    with pytest.raises(ValueError):
        DeviceTileDataset.batch_plan(_Fake(), 30, 0, 8)


def test_operand_modes_per_layer_class():
    """host logic of the precision modes (config.operand_mode): which MFMA operand form each layer class takes, the
    per-class override used by tools/parity_attribution.py, and the environment form of the override"""
    import importlib
    cfg = importlib.import_module("gan_danet_amd.config")        # (the package attribute `config` is the settings object)
    saved = (cfg.config.precision, dict(cfg.config.override))
    try:
        cfg.config.override.clear()
        table = {
            "bf16": {"pam": "16", "dense3x3": "16", "conv1x1": "x3", "cam_apply": "x3", "vgg": "16", "disc": "16", "stem": "exact"},
            "fp16": {"pam": "16", "dense3x3": "16", "conv1x1": "x3", "stem": "exact"},
            "fp32": {"pam": "exact", "dense3x3": "exact", "vgg": "exact", "stem": "exact"},
            "mixed": {"pam": "16", "dense3x3": "x3", "fuse3x3": "x3", "decoder": "x3", "vgg": "x3", "conv1x1": "x3",
                      "cam_apply": "x3", "disc": "x3", "stem": "exact", "other": "x3"},
        }
        for prec, want in table.items():
            with cfg.precision(prec):
                for layer, mode in want.items():
                    assert cfg.operand_mode(layer) == mode, (prec, layer, cfg.operand_mode(layer))
                assert cfg.sixteen_bit("pam") == (prec != "fp32")
        with cfg.precision("bf16"), cfg.layer_override(dense3x3="exact", stem="16"):
            assert cfg.operand_mode("dense3x3") == "exact" and cfg.operand_mode("stem") == "16" and cfg.operand_mode("vgg") == "16"
        assert cfg.operand_mode("dense3x3") == "16"                      # the override is scoped
        for bad in ({"nosuchclass": "exact"}, {"vgg": "fp64"}):
            try:
                with cfg.layer_override(**bad):
                    pass
                raise AssertionError("layer_override accepted " + str(bad))
            except ValueError:
                pass
        try:
            cfg.set_precision("fp8")
            raise AssertionError("set_precision accepted fp8")
        except ValueError:
            pass
    finally:
        cfg.config.precision = saved[0]
        cfg.config.override.clear()
        cfg.config.override.update(saved[1])
