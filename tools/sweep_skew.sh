#!/bin/bash
# experiment: step time of the headline workload against the start offset of the upper four waves (nsk_set_tuning "skew")
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/skew
for S in ${@:-0 2 3 4 5 6 8}; do
  python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --tune skew=$S > $R/gpurun_out/skew/s$S.json 2>/dev/null
  python3 - <<PY
import json
d=json.load(open("$R/gpurun_out/skew/s$S.json"))
k=d["kernels"]
print("skew $S  ms/step %.4f  fwd %.1f bwd %.1f  K2 %.4f fwd %.1f bwd %.1f  K3fine %.4f" % (d["ms_per_step"], k["decode_fwd_multi"]["avg_us"], k["decode_bwd_multi"]["avg_us"],
      d["extras"]["K2_color"]["ms_per_step"], d["extras"]["K2_color"]["kernels_avg_us"]["decode_fwd_multi"], d["extras"]["K2_color"]["kernels_avg_us"]["decode_bwd_multi"], d["extras"]["K3_fine_stage"]["ms_per_step"]))
PY
done
