"""Import shim: registers the hyphenated package directory ``llm-groundeddiffusion_amd/`` under the
importable name ``lgd_amd`` (``import lgd_amd`` works whenever the repo root is on sys.path)."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "llm-groundeddiffusion_amd")
_spec = importlib.util.spec_from_file_location(
    "lgd_amd", os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["lgd_amd"] = _mod
_spec.loader.exec_module(_mod)
