"""print the headline numbers of a bench.py JSON line: python tools/benchsum.py file.json"""
import json
import sys

l = json.loads([x for x in open(sys.argv[1]) if x.startswith("{")][0])
print("ms_per_step", round(l["ms_per_step"], 4), "value", f"{l['value']:.4g}", "n_gpus", l["n_gpus"], "host_issue_ms", round(l["host_issue_ms_per_step"], 4))
for k, v in l["kernels"].items():
    if not k.startswith("(no launch"):
        print("  ", k, round(v["avg_ms"], 4))
r = l["roofline"]
if r:
    print("roofline", r["kernel"], "frac", r["frac"], "stale", r.get("traffic_stale"), "units", r.get("units_processed"))
    for k, v in r["by_kernel"].items():
        print("  ", k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items()})
if l.get("fwd_render"):
    print("fwd", {k: (round(v["ms_per_frame"], 3), round(v["kernel_ms_per_frame"], 3), round(v.get("frac_of_lds_read_peak") or v.get("frac_of_l2_peak_processed") or 0.0, 3)) for k, v in l["fwd_render"].items() if isinstance(v, dict)})
if l.get("highres_render"):
    print("highres mask/no mask", round(l["highres_render"]["ms_per_frame_occupancy_mask"], 3), round(l["highres_render"]["ms_per_frame_no_mask"], 3))
if l.get("strict_dropin"):
    print("dropin ms", round(l["strict_dropin"]["ms_per_step"], 4))
c = l.get("cpu_baseline")
if c:
    print("cpu", f"{c['value']:.3g}", "cores", c["cores"], "all", c.get("all_host_cores"), "fwd", f"{c['forward_only']['value']:.3g}" if c.get("forward_only") else None)
print("errors", l["roofline_model_errors"], "dist", l["distributed"])
