"""Drop-in for the reference's ``models`` package (models/__init__.py:2-23): the same ten public names,
served by the MI355X-native implementation in ``gan-danet_amd/``."""
import os as _os
import sys as _sys

_root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
if _root not in _sys.path:
    _sys.path.insert(0, _root)
import gan_danet_amd as _impl  # noqa: E402

__all__ = list(_impl.__all__)
globals().update({_n: getattr(_impl, _n) for _n in __all__})
generator = _sys.modules["gan_danet_amd.generator"]
discriminator = _sys.modules["gan_danet_amd.discriminator"]
losses = _sys.modules["gan_danet_amd.losses"]
utils = _sys.modules["gan_danet_amd.utils"]
for _n in ("generator", "discriminator", "losses", "utils"):
    _sys.modules[f"models.{_n}"] = globals()[_n]
