"""Legacy import path of the reference notebooks (``from model import ...``, model.py:2-5)."""
import models as _m

__all__ = _m.__all__
globals().update({_n: getattr(_m, _n) for _n in __all__})
