#!/usr/bin/env python3
"""ISA lint for the 128-bit store-data hazard hipcc (ROCm 7.2, gfx950) does not cover.

A VMEM store of more than 64 bits reads its data registers after it has issued; a vector instruction that writes one
of them in the next wait states corrupts the store.  LLVM's hazard recognizer (GCNHazardRecognizer::createsVALUHazard)
inserts the wait states for FLAT / global stores and for buffer stores WITHOUT a register soffset, but assumes that a
buffer store whose soffset is an SGPR is safe.  On MI355X it is not: `buffer_store_dwordx4 v[32:35], v137, s[4:7], s9
offen` directly followed by `v_add_f32 v32, ...` stored the new v32 in some lanes (found in round 5 on
tspn_block_bf16.hip: ~0.02 % wrong outputs with several waves per SIMD, none with one; profiles/r5/
bottleneck_block_study.md).

This tool compiles every csrc/*.hip to assembly (hipcc -S --cuda-device-only, the build's flags) and reports every
buffer_store_dwordx3 / x4 with a register soffset whose data registers are written by a v_* instruction within the next
`--window` instructions (default 3; LLVM inserts 2 wait states on gfx940 where it sees the hazard; the failing code had
0 and 1).  s_nop N counts as N + 1; matrix instructions are not counted as writers (their results land tens of cycles
later).  Exit status 1 if anything is found.  tests/test_host.py runs it over csrc/ on every CPU test run.

    python tools/lint_store_hazard.py [--window 3] [files.hip ...]
"""
import argparse
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "temporal-span-proposal-network-vidvrd_amd")
CSRC = os.path.join(PKG, "csrc")


def build_flags():
    """The flags of the REAL build (build.py FLAGS: the linted ISA must be the shipped ISA), minus the resource remarks
    and warnings that only add noise here."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_tspn_build_lint", os.path.join(PKG, "build.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    return [f for f in b.FLAGS if not f.startswith(("-Rpass", "-Wall"))] + ["-Wno-everything"]


STORE = re.compile(r"^\s*buffer_store_dwordx([34])\s+v\[(\d+):(\d+)\],\s*(\S+),\s*s\[\d+:\d+\],\s*(\S+)")
VREG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")
LABEL = re.compile(r"^[.\w$]+:")


def dest_regs(line):
    """Vector registers written by a v_* instruction: its first operand (v_cmp* / v_readlane etc. write no VGPR)."""
    m = re.match(r"^\s*(v_\w+)\s+(.*)$", line)
    if not m:
        return set()
    op, rest = m.group(1), m.group(2)
    if op.startswith(("v_cmp", "v_readlane", "v_readfirstlane", "v_nop")):
        return set()
    if op.startswith(("v_mfma", "v_smfma")):
        return set()      # a matrix instruction writes its destination at the END of its 8 - 16 passes (>= 32 cycles later)
    first = rest.split(",")[0].strip()
    m2 = VREG.fullmatch(first)
    if not m2:
        return set()
    if m2.group(3) is not None:
        return {int(m2.group(3))}
    return set(range(int(m2.group(1)), int(m2.group(2)) + 1))


def lint_asm(text, window):
    findings, kernel = [], None
    lines = text.split("\n")
    instrs = []     # (kernel, line number, text)
    for i, ln in enumerate(lines):
        if ln.startswith("_Z") and ln.rstrip().endswith(":") or re.match(r"^_Z\w+:\s", ln):
            kernel = ln.split(":")[0]
        s = ln.split(";")[0].rstrip()
        if not s.strip() or s.lstrip().startswith(".") or LABEL.match(s.strip()):
            if LABEL.match(s.strip()):
                instrs.append((kernel, i, None))      # a label: control flow may join here, stop looking ahead
            continue
        instrs.append((kernel, i, s.strip()))
    for n, (k, i, s) in enumerate(instrs):
        if s is None:
            continue
        m = STORE.match(s)
        if not m:
            continue
        soff = m.group(5)
        if not soff.startswith("s"):                  # `0` / literal soffset: the compiler handles that form
            continue
        data = set(range(int(m.group(2)), int(m.group(3)) + 1))
        slack = 0
        for k2, i2, s2 in instrs[n + 1:n + 1 + 8]:
            if s2 is None or slack >= window:
                break
            hit = dest_regs(s2) & data
            if hit:
                findings.append((k, i + 1, s, i2 + 1, s2, slack))
                break
            mn = re.match(r"s_nop\s+(\d+)", s2)
            slack += (int(mn.group(1)) + 1) if mn else 1
    return findings


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="*")
    ap.add_argument("--window", type=int, default=3)
    ap.add_argument("--jobs", type=int, default=6)
    args = ap.parse_args()
    files = [os.path.abspath(f) for f in args.files] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    bad = 0
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    flags = build_flags()
    jobs = max(1, min(args.jobs, os.cpu_count() or 1))
    with tempfile.TemporaryDirectory() as tmp:
        from concurrent.futures import ThreadPoolExecutor

        def compile_one(f):
            out = os.path.join(tmp, os.path.basename(f) + ".s")
            p = subprocess.run([hipcc] + flags + ["-S", "--cuda-device-only", f, "-o", out], stdout=subprocess.DEVNULL,
                               stderr=subprocess.PIPE, text=True, cwd=CSRC)
            return f, out, p

        with ThreadPoolExecutor(max_workers=jobs) as pool:      # (it used to start every hipcc at once)
            procs = list(pool.map(compile_one, files))
        for f, out, p in procs:
            err = p.stderr
            if p.returncode != 0:
                print(f"{os.path.basename(f)}: hipcc failed\n{err[-400:]}")
                bad = 1
                continue
            text = open(out).read()
            found = lint_asm(text, args.window)
            nstores = len([ln for ln in text.split("\n") if STORE.match(ln.split(";")[0])])
            print(f"{os.path.basename(f)}: {nstores} wide buffer stores, {len(found)} with a vector write of their data within "
                  f"{args.window} wait states")
            for k, l1, s1, l2, s2, slack in found:
                print(f"   {k}\n      {l1}: {s1}\n      {l2}: {s2}   (wait states in between: {slack})")
            bad |= 1 if found else 0
    return bad


if __name__ == "__main__":
    sys.exit(main())
