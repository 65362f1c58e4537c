"""Import shim: registers the package that lives in ``gan-danet_amd/`` (hyphen required by the repo
contract, not importable as written) under the module name ``gan_danet_amd``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gan-danet_amd")
_spec = importlib.util.spec_from_file_location("gan_danet_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gan_danet_amd"] = _mod
_spec.loader.exec_module(_mod)
