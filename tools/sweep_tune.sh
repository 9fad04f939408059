#!/bin/bash
# experiment: step time of the headline workload against one nsk_set_tuning key:  tools/sweep_tune.sh key v1 v2 ...
R=${GRAFT_REPO_ROOT:-$PWD}
KEY=$1; shift
mkdir -p $R/gpurun_out/tune
for S in "$@"; do
  python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --tune $KEY=$S > $R/gpurun_out/tune/$KEY$S.json 2>/dev/null
  python3 - <<PY
import json
d=json.load(open("$R/gpurun_out/tune/$KEY$S.json"))
k=d["kernels"]
print("$KEY $S  ms/step %.4f  fwd %.1f bwd %.1f  K2 %.4f fwd %.1f bwd %.1f  K3fine %.4f  K4shard %.4f bwd %.1f" % (d["ms_per_step"], k["decode_fwd_multi"]["avg_us"], k["decode_bwd_multi"]["avg_us"],
      d["extras"]["K2_color"]["ms_per_step"], d["extras"]["K2_color"]["kernels_avg_us"]["decode_fwd_multi"], d["extras"]["K2_color"]["kernels_avg_us"]["decode_bwd_multi"], d["extras"]["K3_fine_stage"]["ms_per_step"],
      d["extras"]["K4_shard_color"]["ms_per_step"], d["extras"]["K4_shard_color"]["kernels_avg_us"]["decode_bwd_multi"]))
PY
done
