"""Loads the package directory ``octree-slam_amd/`` (hyphenated, so not importable
by name) and registers it as the module ``octree_slam_amd``."""
import importlib.util
import os
import sys

# pipeline.run_stream uses four HIP streams; with the runtime's default of 4 hardware queues two of them
# would share a queue and run in order.  Only effective before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "octree-slam_amd")


def load():
    if "octree_slam_amd" in sys.modules:
        return sys.modules["octree_slam_amd"]
    spec = importlib.util.spec_from_file_location("octree_slam_amd", os.path.join(PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["octree_slam_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
