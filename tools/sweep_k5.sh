#!/bin/bash
# the K5 loop (bench.py --k5-only) against one nsk_set_tuning key: tools/sweep_k5.sh <key> v1 v2 ...
KEY=$1; shift
mkdir -p gpurun_out/tune
for V in "$@"; do
  python bench.py --k5-only --tune $KEY=$V > gpurun_out/tune/k5_$KEY$V.json 2>/dev/null || exit 1
  python - "gpurun_out/tune/k5_$KEY$V.json" "$KEY=$V" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); k = d["kernels_us_per_frame"]
print(sys.argv[2], "kernel ms per frame", round(d["kernel_ms_per_frame_sum"], 3), "bwd", k["decode_bwd_multi"], "fwd", k["decode_fwd_multi"])
PY
done
