import json,sys
d=json.load(open(sys.argv[1]))
print("seq/s", round(d["value"],1), "ms/step", round(d["ms_per_step"],2))
for k,v in d["kernels"].items(): print(f'{k:18s} n={v["launches"]:5d} avg_ms={v["avg_ms"]:.3f} {v["achieved"]:8.1f} {v["unit"]:8s} frac={v["frac"]:.3f} share={v["share_of_step"]:.3f}')
if d.get("cpu_baseline"): print(d["cpu_baseline"])
if d.get("parity"): print(d["parity"])
