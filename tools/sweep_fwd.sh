#!/bin/bash
# experiment: forward role costs (fine, colour) of the workgroup split:  tools/sweep_fwd.sh "fine:color" ...
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/tune
for P in "$@"; do
  F=${P%%:*}; C=${P##*:}
  python3 $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --tune fwd_fine_cost=$F --tune fwd_color_cost=$C > $R/gpurun_out/tune/fwd_$F_$C.json 2>/dev/null
  python3 - <<PY
import json
d=json.load(open("$R/gpurun_out/tune/fwd_$F_$C.json"))
k=d["kernels"]; e=d["extras"]
print("fine $F colour $C  ms/step %.4f  fwd %.1f | K2 %.4f fwd %.1f | K3fine %.4f fwd %.1f | K4shard %.4f fwd %.1f" % (d["ms_per_step"], k["decode_fwd_multi"]["avg_us"],
      e["K2_color"]["ms_per_step"], e["K2_color"]["kernels_avg_us"]["decode_fwd_multi"], e["K3_fine_stage"]["ms_per_step"], e["K3_fine_stage"]["kernels_avg_us"]["decode_fwd_multi"],
      e["K4_shard_color"]["ms_per_step"], e["K4_shard_color"]["kernels_avg_us"]["decode_fwd_multi"]))
PY
done
