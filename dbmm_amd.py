"""Import shim: `import dbmm_amd` -> the package in ./debiasing-multi-modal_amd/.

The package directory name is fixed by the build contract and is not a valid Python
identifier, so it is loaded by path and registered under the importable name `dbmm_amd`
(sub-modules resolve through its __path__: `from dbmm_amd import clip`).
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "debiasing-multi-modal_amd")
_spec = importlib.util.spec_from_file_location(
    "dbmm_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["dbmm_amd"] = _mod
_spec.loader.exec_module(_mod)
