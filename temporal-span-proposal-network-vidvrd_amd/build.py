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
    f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC",
    "-ffp-contract=off",          # keep mul/add unfused where the reference's numpy does (tspn_iou.hip)
    "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
    f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}",
]
OBJ_DIR = os.path.join(HERE, "build", "obj")   # git-ignored and gpurun-ignored: only the .so travels


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h")) + \
        [os.path.abspath(__file__)]


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in sources() + _headers())


def _obj_of(src):
    return os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")


def _compile_one(hipcc, src, verbose):
    cmd = [hipcc] + FLAGS + ["-c", src, "-o", _obj_of(src)]
    if verbose:
        print("[tspn build]", " ".join(cmd), flush=True)
    res = subprocess.run(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    return src, res.returncode, res.stdout


def build(force=False, verbose=True, jobs=None):
    """Compile every HIP source for gfx950 (one object per source, stale ones only, in parallel) and link
    them into one shared library; returns its path."""
    if not force and not needs_build():
        return LIB_PATH
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build the TSPN HIP library")
    os.makedirs(OBJ_DIR, exist_ok=True)
    newest_header = max(os.path.getmtime(h) for h in _headers())
    stale = [s for s in sources()
             if force or not os.path.exists(_obj_of(s))
             or os.path.getmtime(_obj_of(s)) < max(os.path.getmtime(s), newest_header)]
    from concurrent.futures import ThreadPoolExecutor
    jobs = jobs or min(len(stale) or 1, os.cpu_count() or 1, 8)
    with ThreadPoolExecutor(max_workers=jobs) as pool:
        results = list(pool.map(lambda s: _compile_one(hipcc, s, verbose), stale))
    for src, rc, log in results:
        if log.strip() and (verbose or rc):
            print(log, flush=True)
        if rc:
            raise RuntimeError(f"hipcc failed on {src} (exit {rc})")
    wanted = {_obj_of(s) for s in sources()}
    for o in glob.glob(os.path.join(OBJ_DIR, "*.o")):   # objects of deleted sources must not be linked
        if o not in wanted:
            os.remove(o)
    tmp = LIB_PATH + ".tmp"
    cmd = [hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-fno-gpu-rdc"] + sorted(wanted) + ["-o", tmp]
    if verbose:
        print("[tspn build]", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, cwd=CSRC)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
