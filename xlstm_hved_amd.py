"""Import alias: the package directory is named `xlstm-hved_amd` (not a valid Python identifier), so
`import xlstm_hved_amd` loads that directory as the package `xlstm_hved_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "xlstm-hved_amd")
_spec = importlib.util.spec_from_file_location(
    "xlstm_hved_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["xlstm_hved_amd"] = _mod
_spec.loader.exec_module(_mod)
