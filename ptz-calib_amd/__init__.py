"""ptz-calib_amd: MI355X-native PTZ-IBA / PTZ-Reloc solver hot path.

The product is csrc/ (HIP kernels + C-ABI, built into libptzcalib_hip.so) and host/ (C++ classes with the
reference's PTZRayOptimizer / KRTOptimizer signatures).  This Python package is plumbing for the tests and
the benchmark: a ctypes binding of the C-ABI (api.py) and the synthetic rig generator (synth.py).
The directory name has a hyphen, so import it through __graft_entry__.load_package().
"""
from . import dataset_io, evalmetrics, hostlib, sharding, synth  # noqa: F401
from .api import *  # noqa: F401,F403
