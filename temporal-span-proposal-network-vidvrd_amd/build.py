"""Build the C-ABI shared library (gfx950) in-tree with hipcc.

`python -m` is not usable on a hyphenated package directory, so this module is
driven by `__graft_entry__.build()` or run directly:
    python temporal-span-proposal-network-vidvrd_amd/build.py
The library lands next to this file so that it travels with the repo snapshot.
"""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_NAME = "libtspn_mi355x.so"
LIB_PATH = os.path.join(HERE, LIB_NAME)
ARCH = "gfx950"

FLAGS = [
    f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off",          # keep mul/add unfused where the reference's numpy does (tspn_iou.hip)
    "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
    f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}",
]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h"))
    deps.append(os.path.abspath(__file__))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Compile every HIP source into one shared library; returns its path."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build the TSPN HIP library")
    tmp = LIB_PATH + ".tmp"
    cmd = [hipcc] + FLAGS + sources() + ["-o", tmp]
    if verbose:
        print("[tspn build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
