#!/bin/bash
# Same-box A/B of two builds of libnsk.so (boxes differ by +-5 %, so only runs on one box compare): tools/ab_bench.sh <outdir> <workloads...>
# "prev" = nice-slam-cpp_amd/csrc/libnsk_prev.so (copy the library there before rebuilding), "new" = the tree's libnsk.so; alternates prev/new twice.
out=$1; shift
mkdir -p $out
for w in "$@"; do
  for lib in prev new prev new; do
    if [ $lib = prev ]; then export NSK_LIB=$PWD/nice-slam-cpp_amd/csrc/libnsk_prev.so; else unset NSK_LIB; fi
    python bench.py --workload $w --no-extras --no-cpu --steps 300 --warmup 30 > $out/b_${w}_$lib.json 2>&1 || exit 1
    python - "$out/b_${w}_$lib.json" "$w $lib" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")][-1]; d = json.loads(l)
print(sys.argv[2], round(d["ms_per_step"], 4), "bwd_us", round(d["roofline"]["avg_launch_us"], 2), {k: round(v["avg_us"], 2) for k, v in d["kernels"].items()})
PY
  done
done
