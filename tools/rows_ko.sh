#!/bin/bash
# Knock-out timing of blur_rows at B x 64x64x3 / 31 taps (profiles/r04_b_ab_notes.md).  Builds the BG_DIAG
# variants of blur.hip (tools/build_variant.sh -> tools/_build/, not tracked) unless they exist, then times each with
# tools/blur_loop.py.  Usage on the GPU box: tools/rows_ko.sh            (build first, here: tools/rows_ko.sh --build-only)
set -e
cd "$(dirname "$0")/.."
for v in NO_MEM NO_H NO_W; do [ -f tools/_build/libbgan_rows_$v.so ] || tools/build_variant.sh rows_$v blur.hip "-DBG_DIAG -DROWS_$v" > /dev/null; done
[ -f tools/_build/libbgan_rows_NO_HW.so ] || tools/build_variant.sh rows_NO_HW blur.hip "-DBG_DIAG -DROWS_NO_H -DROWS_NO_W" > /dev/null
[ "$1" = "--build-only" ] && exit 0
for B in 256 768; do
  python tools/blur_loop.py $B 64 64 3 5.0 200
  for v in NO_MEM NO_H NO_W NO_HW; do BGAN_HIP_LIB=tools/_build/libbgan_rows_$v.so python tools/blur_loop.py $B 64 64 3 5.0 200; done
done
