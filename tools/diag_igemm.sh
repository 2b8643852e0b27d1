#!/bin/bash
# Knock-out budget of the gather-GEMM on the short-K layers (one process per library build, same box):
#   tools/diag_igemm.sh <out-dir-under-gpurun_out> "<layer filter>" [batch]
# needs the variant libraries of tools/build_variant.sh (igemm_NO_EPI, NO_BAR, NO_LST, NO_FET, NO_LDS, MFMA, MFMA_NOEPI, NO_A, NO_B)
cd "$(dirname "$0")/.."
out=gpurun_out/$1; only=$2; batch=${3:-256}
mkdir -p $out
run() {  # name, lib ("" = product)
  if [ -n "$2" ]; then export BGAN_HIP_LIB=$PWD/$2; else unset BGAN_HIP_LIB; fi
  timeout -k 10 120 python tools/bench_conv.py --arch celeba64 --batch $batch --iters 10 --only "$only" > $out/ko_${batch}_$1.log 2>&1 || echo "$1 failed"
}
run product ""
for v in NO_EPI NO_BAR NO_LST NO_FET NO_LDS MFMA MFMA_NOEPI; do
  [ -f tools/_build/libbgan_igemm_$v.so ] && run $v tools/_build/libbgan_igemm_$v.so
done
run product2 ""
unset BGAN_HIP_LIB
python - <<PY
import glob, re, os
rows = {}
names = []
for f in sorted(glob.glob("$out/ko_${batch}_*.log")):
    n = os.path.basename(f)[len("ko_${batch}_"):-4]
    names.append(n)
    for line in open(f):
        m = re.match(r"(\S+ \S+(?: s1)?)\s+(fwd|dgrad)\s+([\d.]+)\s+([\d.]+)\s+([\d.]+)", line)
        if m:
            rows.setdefault((m.group(1), m.group(2)), {})[n] = float(m.group(3)) * 1e3
order = ["product", "product2", "NO_EPI", "NO_BAR", "NO_LST", "NO_FET", "NO_LDS", "MFMA", "MFMA_NOEPI"]
order = [o for o in order if o in names]
print("| layer / op (us) | " + " | ".join(order) + " |")
print("|---|" + "---|" * len(order))
for k, v in rows.items():
    print(f"| {k[0]} {k[1]} | " + " | ".join(f"{v.get(o, float('nan')):.1f}" for o in order) + " |")
PY
