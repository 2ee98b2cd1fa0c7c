"""Import alias: the package directory is ``vl-pet_amd/`` (not a valid Python identifier), so
``import vlpet_amd`` loads it under that name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "vl-pet_amd")
_spec = importlib.util.spec_from_file_location(
    "vlpet_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["vlpet_amd"] = _mod
_spec.loader.exec_module(_mod)
