"""Builds octree-slam_amd/libsvoslam_hip.so for gfx950 with hipcc (in-tree, no JIT cache).

Every .hip translation unit is compiled to an object in csrc/_obj/ (only when
stale) and linked into one shared library.  hipcc cross-compiles without a GPU.
Floating-point contraction is OFF so device arithmetic follows the source
operation order (see csrc/common.hpp)."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libsvoslam_hip.so")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")

FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    "-ffp-contract=off",                          # IEEE op-by-op arithmetic, explicit fmaf only
    "-fhip-fp32-correctly-rounded-divide-sqrt",   # '/' and sqrtf are correctly rounded (default, stated)
    "-fno-fast-math", "-munsafe-fp-atomics",      # hardware f64 atomic add (sums are integer-valued: exact)
    "-Wall", "-Wno-unused-function", "-Wno-unused-result",
]


# experiments: extra -D flags for every translation unit (rebuild with force=True)
FLAGS += os.environ.get("SVOSLAM_EXTRA_HIPCC_FLAGS", "").split()


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def headers_mtime():
    m = 0.0
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for f in os.listdir(root):
            if f.endswith((".hpp", ".h")):
                m = max(m, os.path.getmtime(os.path.join(root, f)))
    return max(m, os.path.getmtime(__file__))


def compile_one(src, hm, verbose):
    obj = os.path.join(OBJ, src[:-4] + ".o")
    path = os.path.join(CSRC, src)
    if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(path), hm):
        return obj, False
    cmd = [HIPCC] + FLAGS + ["-c", path, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return obj, True


def build(verbose=False, force=False):
    os.makedirs(OBJ, exist_ok=True)
    hm = headers_mtime() if not force else float("inf")
    with ThreadPoolExecutor(max_workers=4) as ex:
        res = list(ex.map(lambda s: compile_one(s, hm, verbose), sources()))
    objs = [o for o, _ in res]
    if any(ch for _, ch in res) or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
