"""rocprofv3 kernel-trace durations of bench.py's TIMED epoch only, per kernel family, beside the HIP-event figures bench.py printed in
the same process.

Usage: python tools/timed_epoch_summary.py <kernel_trace.csv> <n_timed_steps> <bench_under_rocprof.json> <out.json>

`--stats` averages every dispatch of the process, and the untimed setup runs the same kernels on other shapes (checkpoint synthesis, a full
catalog encode of short item texts, warm-up steps), so its per-kernel average is not the timed region's.  The timed epoch is ONE catalog
pass + n_timed_steps user passes = the last `per_pass * (n_timed_steps + 1)` dispatches of each per-pass family (the last n_timed_steps of
the per-step ones) -- the same selection tools/pmc_summary.py makes for the traffic counters."""
import collections, csv, json, sys

from pmc_summary import FAMILIES


def main():
    trace, steps, bench_json, out = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        for key, (fam, _) in FAMILIES.items():
            if key in r["Kernel_Name"]:
                per[fam].append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6))
                break
    bench = json.loads(open(bench_json).read().strip().splitlines()[-1])
    def fam_of(k):  # bench.py names the split-precision families by arithmetic; the kernel templates are shared
        if k.endswith(("bf16x3", "bf16x6")):
            return k[:-1]
        return k[:-6] + "_bf16x" if k.endswith("_f16x3") else k

    events = {fam_of(k): v for k, v in bench.get("kernels", {}).items()}
    res = {"_timed_steps": steps, "_bench_ms_per_step_under_rocprof": bench["ms_per_step"]}
    for key, (fam, per_pass) in FAMILIES.items():
        if fam not in per:
            continue
        n = steps if per_pass is None else per_pass * (steps + 1)
        d = [v for _, v in sorted(per[fam])]
        tail = d[-n:]
        ent = dict(dispatches_in_process=len(d), avg_ms_whole_process=sum(d) / len(d), dispatches_timed_epoch=len(tail),
                   avg_ms_timed_epoch=sum(tail) / len(tail))
        ev = events.get(fam)
        if ev:
            ent.update(hip_event_launches=ev["launches"], hip_event_avg_ms=ev["avg_ms"],
                       hip_event_over_trace=ev["avg_ms"] / ent["avg_ms_timed_epoch"])
        res[fam] = ent
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
