"""Import shim: the package directory is named `nice-slam-cpp_amd` (not an identifier); `import nice_slam_cpp_amd`
loads it under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nice-slam-cpp_amd")
_spec = importlib.util.spec_from_file_location("nice_slam_cpp_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["nice_slam_cpp_amd"] = _mod
_spec.loader.exec_module(_mod)
