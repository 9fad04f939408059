"""print the figures of a bench.py JSON line (argv[1]) in a few lines"""
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = d["roofline"]
print("value %.0f rays/s  ms/step %.4f  blocks %s" % (d["value"], d["ms_per_step"], {k: round(v, 4) for k, v in d["block_ms_per_step"].items()}))
print("roofline: %s %s frac %.3f frac_16bit %s issue %s wait %s mfma_busy %s | sibling %s" % (r["kernel"], r["bound"], r["frac"], r.get("frac_16bit"), r.get("issue_frac"), r.get("wait_frac"), r.get("mfma_busy"),
      {k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.get("sibling", {}).items()}))
print("kernels", {k: round(v["avg_us"], 1) for k, v in d["kernels"].items()})
for k, v in d.get("extras", {}).items():
    print("  ", k, "ms", round(v.get("ms_per_step", v.get("ms_per_frame", 0)), 4), "value %.0f" % v["value"], v.get("kernels_avg_us", v.get("kernels_us_per_frame")))
if d.get("cpu_baseline"):
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], "| c port", d.get("cpu_baseline_c_port", {}).get("value"))
