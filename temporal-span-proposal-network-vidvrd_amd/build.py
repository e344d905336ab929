"""Build the C-ABI shared library (gfx950) in-tree with hipcc.

`python -m` is not usable on a hyphenated package directory, so this module is
driven by `__graft_entry__.build()` or run directly:
    python temporal-span-proposal-network-vidvrd_amd/build.py
The library lands next to this file so that it travels with the repo snapshot.
"""
import glob
import json
import os
import re
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
    "-Rpass-analysis=kernel-resource-usage",   # per-kernel registers / spills / scratch, parsed below
    f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}",
]
OBJ_DIR = os.path.join(HERE, "build", "obj")   # git-ignored and gpurun-ignored: only the .so travels
RES_PATH = os.path.join(HERE, "kernel_resources.json")   # next to the .so (git-ignored, travels with it)

# Kernels whose inline asm separates a load (`=v` / `=s` outputs of global_load_dwordx4 / s_load_dwordx4) from
# the s_waitcnt that makes the registers valid: a compiler-inserted copy or spill between the two would capture
# stale data.  The build FAILS when one of them spills or uses scratch.
NO_SPILL_KERNELS = ("conv3_wino63_kernel", "heads_pairgrid4_kernel", "heads_pairgrid3_kernel",
                    "conv2d_nhwc_frag_kernel", "conv2d_nhwc_bf16_kernel", "conv2d_nhwc_cin4_kernel",
                    "bottleneck_bf16_kernel", "bottleneck_pipe_bf16_kernel", "conv3_bf16_big_kernel", "heads_pairgrid_bf16_kernel",
                    #  254 - 256 and 233 - 234 VGPRs, hand-counted s_waitcnt vmcnt(N) and the "keep the store data
                    # registers live" hazard fence: a spill would change the VMEM counts and could defeat the fence
                    "tail_io_bf16_kernel", "bottleneck_block_bf16_kernel")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _headers():
    return glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h")) + \
        [os.path.abspath(__file__)]


def needs_build():
    if not os.path.exists(LIB_PATH) or not os.path.exists(RES_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    return any(os.path.getmtime(d) > t for d in sources() + _headers())


def _obj_of(src):
    return os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")


_REMARK = re.compile(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|"
                     r"SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)")
_KEYS = {"TotalSGPRs": "sgprs", "VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize [bytes/lane]": "scratch_bytes",
         "Occupancy [waves/SIMD]": "occupancy", "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill",
         "LDS Size [bytes/block]": "lds_bytes"}


def _demangle(names):
    filt = shutil.which("llvm-cxxfilt") or shutil.which("c++filt")
    if not names or not filt:
        return list(names)
    out = subprocess.run([filt], input="\n".join(names), stdout=subprocess.PIPE, text=True).stdout.split("\n")
    return out[:len(names)]


def _parse_resources(log):
    """{demangled kernel name: {sgprs, vgprs, agprs, scratch_bytes, occupancy, sgpr_spill, vgpr_spill, lds_bytes}}
    from the -Rpass-analysis=kernel-resource-usage remarks; returns (resources, log without the remarks)."""
    res, cur, kept, skip = {}, None, [], 0
    for line in log.split("\n"):
        m = _REMARK.search(line)
        if m:
            if m.group(1) == "Function Name":
                cur = res.setdefault(m.group(2), {})
            elif cur is not None:
                cur[_KEYS[m.group(1)]] = int(m.group(2))
            skip = 2          # the source line and the caret line that follow a remark
            continue
        if "remark:" in line and "kernel-resource-usage" in line:
            skip = 2
            continue
        if skip and (line.lstrip().startswith("|") or re.match(r"\s*\d+ \|", line)):
            skip -= 1
            continue
        skip = 0
        kept.append(line)
    names = list(res)
    return dict(zip(_demangle(names), (res[n] for n in names))), "\n".join(kept)


def _compile_one(hipcc, src, verbose):
    cmd = [hipcc] + FLAGS + ["-c", src, "-o", _obj_of(src)]
    if verbose:
        print("[tspn build]", " ".join(cmd), flush=True)
    res = subprocess.run(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    resources, log = _parse_resources(res.stdout)
    if res.returncode == 0:
        with open(_obj_of(src) + ".res.json", "w") as fh:
            json.dump(resources, fh, indent=1, sort_keys=True)
    return src, res.returncode, log


def kernel_resources():
    """The table written by the last build: {kernel: {vgprs, agprs, sgprs, scratch_bytes, sgpr_spill, ...}}."""
    with open(RES_PATH) as fh:
        return json.load(fh)


def check_no_spill(resources):
    """Names of NO_SPILL_KERNELS instances that spill or use scratch (must be empty)."""
    bad = []
    for name, r in resources.items():
        if any(k in name for k in NO_SPILL_KERNELS) and (r.get("sgpr_spill", 0) or r.get("vgpr_spill", 0)
                                                          or r.get("scratch_bytes", 0)):
            bad.append(f"{name}: {r}")
    return bad


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
             if force or not os.path.exists(_obj_of(s)) or not os.path.exists(_obj_of(s) + ".res.json")
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
            if os.path.exists(o + ".res.json"):
                os.remove(o + ".res.json")
    resources = {}
    for o in sorted(wanted):
        with open(o + ".res.json") as fh:
            resources.update(json.load(fh))
    bad = check_no_spill(resources)
    if bad:
        raise RuntimeError("kernels with split asm load / wait sequences must not spill:\n  " + "\n  ".join(bad))
    with open(RES_PATH, "w") as fh:
        json.dump(resources, fh, indent=1, sort_keys=True)
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
