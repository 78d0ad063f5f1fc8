"""Idle time between kernels in bench.py's timed steps, from a rocprofv3 kernel trace (optionally with the memory-copy trace beside it).

Usage: python tools/gap_report.py <kernel_trace.csv> <n_timed_steps> [<memory_copy_trace.csv>] [<out.json>]

A step starts at a merge launch (one per step), so the last n_timed_steps merge starts delimit the timed steps (the last one runs to the
end of the trace's last dispatch before the process tears down).  Per step: the union of the busy intervals on the device, the idle
remainder, and the idle time bucketed by gap length; overall: the largest gaps with the dispatches either side, and idle time grouped
by the (kernel before -> kernel after) pair.  Run the bench with --no-profile so the HIP-event pairs of bench.py's own profiler are not in
the stream."""
import collections, csv, json, sys


def short(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:60]


def main():
    trace, steps = sys.argv[1], int(sys.argv[2])
    rest = sys.argv[3:]
    copies = next((a for a in rest if a.endswith(".csv")), None)
    out = next((a for a in rest if a.endswith(".json")), None)
    ev = []
    for r in csv.DictReader(open(trace)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    if copies:
        for r in csv.DictReader(open(copies)):
            ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "copy:" + r.get("Direction", r.get("Kind", "?"))))
    ev.sort()
    merges = [i for i, e in enumerate(ev) if e[2].startswith("merge_nway")]
    if len(merges) < steps:
        raise SystemExit(f"only {len(merges)} merge launches in the trace")
    first = merges[-steps]
    # the process's tail (result gathers at the epoch end) belongs to the timed region too; cut at the last dispatch
    seg = ev[first:]
    bounds = [ev[i][0] for i in merges[-steps:]] + [seg[-1][1]]
    gaps, pair_idle = [], collections.Counter()
    busy_end, prev = seg[0][0], None
    for s, e, n in seg:
        if s > busy_end and prev is not None:
            gaps.append((s - busy_end, busy_end, prev, n))
            pair_idle[(prev, n)] += s - busy_end
        if e > busy_end:
            busy_end, prev = e, n
    total = bounds[-1] - bounds[0]
    idle = sum(g[0] for g in gaps)
    buckets = collections.OrderedDict((k, [0, 0]) for k in ("<2us", "2-5us", "5-10us", "10-30us", "30-100us", ">=100us"))
    for g, *_ in gaps:
        k = "<2us" if g < 2e3 else "2-5us" if g < 5e3 else "5-10us" if g < 1e4 else "10-30us" if g < 3e4 else "30-100us" if g < 1e5 else ">=100us"
        buckets[k][0] += 1
        buckets[k][1] += g
    per_step = []
    for a, b in zip(bounds[:-1], bounds[1:]):
        gi = sum(g for g, at, *_ in gaps if a <= at < b)
        per_step.append(dict(ms=(b - a) / 1e6, idle_ms=gi / 1e6, dispatches=sum(1 for s, _, _ in seg if a <= s < b)))
    res = dict(timed_steps=steps, window_ms=total / 1e6, ms_per_step=total / 1e6 / steps, idle_ms_per_step=idle / 1e6 / steps,
               idle_frac=idle / total, dispatches_per_step=len(seg) / steps,
               idle_by_gap_length={k: dict(count_per_step=v[0] / steps, ms_per_step=v[1] / 1e6 / steps) for k, v in buckets.items()},
               per_step=per_step,
               largest_gaps=[dict(us=g / 1e3, after=p, before=n) for g, _, p, n in sorted(gaps, reverse=True)[:25]],
               idle_by_pair=[dict(ms_per_step=v / 1e6 / steps, after=p, before=n) for (p, n), v in pair_idle.most_common(25)])
    txt = json.dumps(res, indent=1)
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main()
