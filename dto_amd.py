"""Import shim: the package directory is named ``directtrajopt.jl_amd`` (with a dot), which Python's
import statement cannot spell.  ``import dto_amd`` loads it under the module name
``directtrajopt_jl_amd`` and re-exports its public names."""
import importlib.util
import os
import sys

_NAME = "directtrajopt_jl_amd"
if _NAME not in sys.modules:
    _dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "directtrajopt.jl_amd")
    _spec = importlib.util.spec_from_file_location(
        _NAME, os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules[_NAME] = _mod
    _spec.loader.exec_module(_mod)
_pkg = sys.modules[_NAME]
globals().update({k: v for k, v in vars(_pkg).items() if not k.startswith("_")})
