"""gan_danet_amd -- MI355X-native (gfx950) implementation of the GAN-DANet G+D training hot path.

Host side: PyTorch-ROCm for device memory, streams, the autograd tape and ``torch.distributed`` (RCCL).
Arithmetic: hand-written HIP kernels in ``csrc/`` behind the C ABI of ``include/gandanet.h``
(``lib/libgandanet_hip.so``).  There is no CPU path: without the built library, or on CPU tensors, the
modules raise.

The directory is called ``gan-danet_amd`` (repo contract); import it as ``gan_danet_amd`` -- the
repo-root ``gan_danet_amd.py`` registers the package under that name -- or through the reference's own
import paths ``from models import ...`` / ``from model import ...``.
"""
from .config import config, layer_override, precision, set_precision, set_sync_bn
from .kern import set_deterministic
from .discriminator import SRGAND, Discriminator1
from .generator import (CAMModule, CBAMBlock, DANetAttention, DenseBlock, DenseLayer, FlexibleUpsamplingModule,
                        OriginalRelationshipLearner, PAMModule, SqueezeExcitation, TransitionLayer)
from .losses import SSIM, BCEWithLogitsLoss, MSELoss, PerceptualLoss, TVLoss
from .optim import AdamW
from .train import GanTrainer
from .utils import weights_init_normal

__all__ = [
    "CBAMBlock", "FlexibleUpsamplingModule", "OriginalRelationshipLearner", "SqueezeExcitation",
    "Discriminator1", "SRGAND", "PerceptualLoss", "SSIM", "TVLoss", "weights_init_normal",
]
