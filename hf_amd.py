"""Import shim: the package directory is named after the reference repository
(`mitsuba3-differentiable-heightfield-rendering_amd`), which is not a valid Python
identifier.  `import hf_amd` loads that directory as the package `hf_amd`
(this module replaces itself in sys.modules by the real package)."""
import importlib.util
import os
import sys

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                    "mitsuba3-differentiable-heightfield-rendering_amd")
_spec = importlib.util.spec_from_file_location(
    "hf_amd", os.path.join(_DIR, "__init__.py"), submodule_search_locations=[_DIR])
_pkg = importlib.util.module_from_spec(_spec)
sys.modules["hf_amd"] = _pkg
_spec.loader.exec_module(_pkg)
_pkg.PACKAGE_DIR = _DIR
