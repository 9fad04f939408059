out=$1; shift; mkdir -p $out
for w in "$@"; do for cfg in prev0 new0 prev1 new1 prev0 new0 prev1 new1; do
  lib=${cfg%?}; p=${cfg: -1}
  if [ $lib = prev ]; then export NSK_LIB=$PWD/nice-slam-cpp_amd/csrc/libnsk_prev.so; else unset NSK_LIB; fi
  python bench.py --workload $w --no-extras --no-cpu --steps 300 --warmup 30 --pipeline $p > $out/q_${w}_$cfg.json 2>&1 || { tail -5 $out/q_${w}_$cfg.json; exit 1; }
  python - $out/q_${w}_$cfg.json "$w $lib pipeline=$p" <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")][-1]; d = json.loads(l)
print(sys.argv[2], round(d["ms_per_step"], 4), {k: round(v["avg_us"], 2) for k, v in d["kernels"].items()})
PY
done; done
