#!/bin/bash
# step time of one bench workload against one nsk_set_tuning key, without the extras: tools/sweep_key.sh <workload> <key> v1 v2 ...
W=$1; KEY=$2; shift; shift
mkdir -p gpurun_out/tune
for S in "$@"; do
  python bench.py --workload $W --no-extras --no-cpu --steps 200 --warmup 30 --tune $KEY=$S > gpurun_out/tune/${W}_$KEY$S.json 2>&1 || exit 1
  python - "gpurun_out/tune/${W}_$KEY$S.json" "$W $KEY=$S" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")][-1]; d = json.loads(l)
print(sys.argv[2], round(d["ms_per_step"], 4), {k: round(v["avg_us"], 2) for k, v in d["kernels"].items()})
PY
done
