import json, sys
d = json.loads([l for l in open(sys.argv[1]).read().splitlines() if l.startswith("{")][-1])
print("pairs/s", d["value"], "ms/step", d["ms_per_step"])
print({k: v["ms_per_launch"] for k, v in d.get("stages", {}).items()})
print(d.get("streaming_kernels_GBps"), d.get("roofline", {}).get("kernel"), d.get("roofline", {}).get("frac"))
if "cpu_baseline" in d: print(d["cpu_baseline"])
