"""Import shim: the package directory is named after the reference repository
(`mitsuba3-differentiable-heightfield-rendering_amd`), which is not a valid Python
identifier, so it is loaded by path and re-exported under `hf_amd`."""
import importlib.util
import os
import sys

_NAME = "mitsuba3_differentiable_heightfield_rendering_amd"
_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                    "mitsuba3-differentiable-heightfield-rendering_amd")

if _NAME not in sys.modules:
    _spec = importlib.util.spec_from_file_location(
        _NAME, os.path.join(_DIR, "__init__.py"), submodule_search_locations=[_DIR])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules[_NAME] = _mod
    _spec.loader.exec_module(_mod)

_pkg = sys.modules[_NAME]
globals().update({k: v for k, v in vars(_pkg).items() if not k.startswith("__")})
PACKAGE_DIR = _DIR
