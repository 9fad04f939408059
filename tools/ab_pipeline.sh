#!/bin/bash
# nsk_map_prepare off / on for the bench workloads on one box: tools/ab_pipeline.sh <outdir> <workloads...>
out=$1; shift
mkdir -p $out
for w in "$@"; do for p in 0 1 0 1; do
  python bench.py --workload $w --no-extras --no-cpu --steps 300 --warmup 30 --pipeline $p > $out/p_${w}_$p.json 2>&1 || { tail -5 $out/p_${w}_$p.json; exit 1; }
  python - $out/p_${w}_$p.json "$w pipeline=$p" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")][-1]; d = json.loads(l)
print(sys.argv[2], round(d["ms_per_step"], 4), {k: round(v["avg_us"], 2) for k, v in d["kernels"].items()})
PY
done; done
