"""Import shim: ``import blurred_gan_amd`` loads the package that lives in ``blurred-gan_amd/`` (the
directory name required by the repo layout is not a valid Python identifier)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "blurred-gan_amd")
_spec = importlib.util.spec_from_file_location("blurred_gan_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["blurred_gan_amd"] = _mod
_spec.loader.exec_module(_mod)
