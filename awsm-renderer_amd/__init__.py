"""awsm-renderer_amd — MI355X-native Geometry Pass + Opaque Pass of dakom/awsm-renderer.

csrc/      hand-written HIP kernels + the C-ABI of include/awsm_hip.h  (libawsm_hip.so)
host/      C++ host layer mirroring the reference's key-based update API (libawsm_host.so)
*.py       thin ctypes bindings, synthetic scene generators, build helpers
"""
import os

PACKAGE_DIR = os.path.dirname(os.path.abspath(__file__))
REPO_ROOT = os.path.dirname(PACKAGE_DIR)
