"""nice-slam-cpp_amd: MI355X-native render / map / track hot path of NICE-SLAM behind the reference's
Renderer / NICE / Mapper / Tracker surface.  Import name: nice_slam_cpp_amd (see nice_slam_cpp_amd.py shim)."""
from . import nsk  # noqa: F401
from .nsk import Context, NskError, build  # noqa: F401
