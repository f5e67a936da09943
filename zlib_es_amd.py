"""Import alias: ``import zlib_es_amd`` loads the package kept in the directory
``zlib.es_amd/`` (a dotted directory name cannot be imported directly)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "zlib.es_amd")
_spec = importlib.util.spec_from_file_location(
    "zlib_es_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["zlib_es_amd"] = _mod
_spec.loader.exec_module(_mod)
