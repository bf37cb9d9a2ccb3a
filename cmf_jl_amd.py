"""Import shim: the package directory is `cmf.jl_amd/` (named after the reference repo),
which is not an importable identifier, so `import cmf_jl_amd` loads it under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cmf.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "cmf_jl_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["cmf_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
