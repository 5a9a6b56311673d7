"""Import alias: the package directory is named ``stable-diffusion_amd`` (not a valid Python
identifier), so ``import stable_diffusion_amd`` resolves to this file, which loads that directory
as the package of the same (underscored) name and hands it back."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "stable-diffusion_amd")
_spec = importlib.util.spec_from_file_location(
    "stable_diffusion_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_pkg = importlib.util.module_from_spec(_spec)
sys.modules["stable_diffusion_amd"] = _pkg
_spec.loader.exec_module(_pkg)
