#!/bin/bash
# A/B of res4-tail builds in ONE call (boxes differ by a few per cent): tools/ab_tail_io.sh LIB.so [LIB.so ...] -- each library
# (and the shipped one) runs tools/time_tail_io.py 18 36 three times, interleaved.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for rep in 1 2 3; do
  for lib in default "$@"; do
    if [ "$lib" == default ]; then python3 $R/tools/time_tail_io.py 18 36; else TSPN_LIB_PATH=$lib python3 $R/tools/time_tail_io.py 18 36; fi
  done
done
