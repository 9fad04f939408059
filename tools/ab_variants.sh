#!/bin/bash
# Same-box comparison of several builds of libnsk.so (compiler-flag experiments): tools/ab_variants.sh <workload> <lib> <lib> ...
# ("tree" = the tree's libnsk.so).  Each variant twice, interleaved; prints step time and per-launch times.
W=$1; shift
mkdir -p gpurun_out/var
for rep in 1 2; do
  for L in "$@"; do
    if [ "$L" = tree ]; then unset NSK_LIB; else export NSK_LIB=$PWD/nice-slam-cpp_amd/csrc/$L; fi
    python bench.py --workload $W --no-extras --no-cpu --steps 300 --warmup 30 > gpurun_out/var/${W}_$L.json 2>&1 || { echo "$L FAILED"; tail -3 gpurun_out/var/${W}_$L.json; continue; }
    python - "gpurun_out/var/${W}_$L.json" "$W $L" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")][-1]; d = json.loads(l)
print(sys.argv[2], round(d["ms_per_step"], 4), {k: round(v["avg_us"], 2) for k, v in d["kernels"].items()})
PY
  done
done
