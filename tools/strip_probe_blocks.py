#!/usr/bin/env python3
"""Resolve `#if defined(M)` / `#if !defined(M)` / `#ifdef M` / `#ifndef M` blocks for probe / ablation macros
as "M is not defined" and write the product source without them.

    python tools/strip_probe_blocks.py FILE.hip [...]       # rewrites in place
    python tools/strip_probe_blocks.py --check FILE.hip ... # exit 1 if any probe block is left

A macro counts as a probe macro when its name matches PROBE_RE.  Other conditionals are left alone.
The lab keeps its scaffolding as reverse patches under tools/probes/ (git diff of this tool's effect).
"""
import re
import sys

PROBE_RE = re.compile(r"TSPN_(\w+_)?(ABL|ABLATE|PROBE)_\w+|TSPN_HPB_SCALAR_ADD|TSPN_BT_BIAS_GLOBAL|TSPN_BP_ONLY_[AB]|"
                      r"TSPN_SCHED_PINNED|TSPN_CONV2D_NOINTERLEAVE|TSPN_CONV2D_BF16_DIRECT_EPILOGUE")
COND = re.compile(r"^\s*#\s*(if|ifdef|ifndef|else|elif|endif)\b(.*)$")


def _probe_value(kind, rest):
    """None if not a probe conditional, else the truth value with the macro undefined."""
    rest = rest.split("//")[0].strip()
    if kind == "ifdef" and PROBE_RE.fullmatch(rest):
        return False
    if kind == "ifndef" and PROBE_RE.fullmatch(rest):
        return True
    if kind == "if":
        m = re.fullmatch(r"(!?)\s*defined\s*\(\s*(\w+)\s*\)", rest)
        if m and PROBE_RE.fullmatch(m.group(2)):
            return bool(m.group(1))
    return None


def strip(text):
    out = []
    # stack entries: (is_probe, emitting_before, branch_taken)
    stack = []
    emitting = True
    for line in text.split("\n"):
        m = COND.match(line)
        if not m:
            if emitting:
                out.append(line)
            continue
        kind, rest = m.group(1), m.group(2)
        if kind in ("if", "ifdef", "ifndef"):
            val = _probe_value(kind, rest)
            if val is None:
                stack.append((False, emitting, None))
                if emitting:
                    out.append(line)
            else:
                stack.append((True, emitting, val))
                emitting = emitting and val
        elif kind in ("else", "elif"):
            is_probe, before, taken = stack[-1]
            if is_probe:
                if kind == "elif":
                    raise SystemExit("probe #elif not supported: " + line)
                emitting = before and not taken
            elif emitting:
                out.append(line)
        else:  # endif
            is_probe, before, _ = stack.pop()
            if is_probe:
                emitting = before
            elif emitting:
                out.append(line)
    if stack:
        raise SystemExit("unbalanced conditionals")
    return "\n".join(out)


def main(argv):
    check = "--check" in argv
    files = [a for a in argv if not a.startswith("--")]
    bad = 0
    for path in files:
        with open(path) as fh:
            src = fh.read()
        new = strip(src)
        if new != src:
            if check:
                print(f"{path}: probe blocks present")
                bad = 1
            else:
                with open(path, "w") as fh:
                    fh.write(new)
                print(f"{path}: {src.count(chr(10)) - new.count(chr(10))} lines removed")
    return bad


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
