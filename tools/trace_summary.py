"""Aggregate a rocprofv3 kernel-trace CSV per (kernel, grid, VGPR, LDS): python tools/trace_summary.py <trace.csv> <out.csv>"""
import collections, csv, sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0, 1e30, 0.0])


def short(n):
    return n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]


for r in rows:
    key = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]),
           r["VGPR_Count"], r["Accum_VGPR_Count"], r["LDS_Block_Size"])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg[key]
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
with open(sys.argv[2], "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "blocks_x", "grid_y", "grid_z", "vgpr", "agpr", "lds_bytes", "calls", "total_us", "avg_us", "min_us", "max_us"])
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if v[1] >= 500:
            w.writerow(list(k) + [v[0], round(v[1], 1), round(v[1] / v[0], 2), round(v[2], 2), round(v[3], 2)])
