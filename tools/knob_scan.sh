#!/bin/bash
# Scan of planner knobs on a set of layers: tools/knob_scan.sh "LAYERS" arch batch  (prints fwd/dgrad per setting)
cd "$(dirname "$0")/.."
L="$1"; A=${2:-celeba64}; B=${3:-256}
run() { env "$@" python tools/bench_conv.py --arch $A --batch $B --iters 10 --only "$L" 2>&1 | grep -E "fwd|dgrad" | grep -v TOTAL | awk '{printf "%s_%s %s=%.1f  ", $1, $2, $3, $4*1000}'; echo; }
echo "base:"; run X=1
for pm in 1 2 4; do echo "BG_PMERGE=$pm:"; run BG_PMERGE=$pm; done
for ks in 1 2 3 4; do echo "BG_SPLITK_FORCE=$ks:"; run BG_SPLITK_FORCE=$ks; done
