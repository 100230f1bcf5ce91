"""Import alias: the package directory is `rover-slam_amd/` (not a valid Python identifier),
so `import rover_slam_amd` loads that directory as a package under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "rover-slam_amd")
_spec = importlib.util.spec_from_file_location(
    "rover_slam_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["rover_slam_amd"] = _mod
_spec.loader.exec_module(_mod)
