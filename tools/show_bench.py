"""Print the figures of a bench.py JSON line: python tools/show_bench.py <file>"""
import json, sys
d = json.load(open(sys.argv[1]))
def leg(name, v):
    if not v: return
    r = v["roofline"] if "roofline" in v else v
    s = f"  {name:22s} {v['value']:.3e} env-steps/s  {v['ms_per_step'] * 1e3:7.2f} us/step(wall)  launch {r['launch_us']:7.2f} us  frac {r['frac']:.3f}"
    if r.get("frac_traffic") is not None: s += f"  traffic frac {r['frac_traffic']:.3f}"
    if "single_launch" in v:
        sl = v["single_launch"]; s += f"  | single launch {sl['value']:.3e} frac {(sl['roofline'] if 'roofline' in sl else sl)['frac']:.3f}"
    print(s)
print(d["config"]["workload"] if "config" in d else "", "| value", f"{d['value']:.3e}", "ms_per_step", d["ms_per_step"], "frac", d["roofline"]["frac"])
leg("per_tick", d.get("per_tick_stepping")); leg("fused_rollout", d.get("fused_rollout"))
for k, v in (d.get("configs") or {}).items():
    print(k); leg("per_tick", v.get("per_tick_stepping")); leg("fused_rollout", v.get("fused_rollout"))
cl = d.get("closed_loop_grid")
if cl: print("closed loop:", json.dumps(cl)[:600])
cb = d.get("cpu_baseline")
if cb: print("cpu:", json.dumps(cb)[:300])
