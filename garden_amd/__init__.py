"""garden_amd — MI355X (gfx950) visibility pass for the Garden engine's ECS frame loop.

The product is `lib/libgarden_vis.so` (C-ABI in `include/garden_vis.h`, HIP kernels in `csrc/`) plus the
C++ ecsm-style shim in `csrc/host/`. This Python package is plumbing for tests and bench.py: ctypes
bindings (`garden_amd.lib`), the reference's component-pool byte layouts as numpy dtypes
(`garden_amd.pools`) and the synthetic scene generator (`garden_amd.scene`).
It never imports anything from `oracle/`.
"""
from .pools import MESH_DTYPE, TRANSFORM_DTYPE, GV_NONE  # noqa: F401
