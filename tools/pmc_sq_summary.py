"""Matrix-pipe occupancy and the wave-cycle split per kernel family from rocprofv3 SQ passes of bench.py.

Usage: python tools/pmc_sq_summary.py <n_timed_steps> <out.json> <pass_dir> [<pass_dir> ...]

Every <pass_dir> is one `rocprofv3 --pmc <SQ counters...> GRBM_GUI_ACTIVE --kernel-trace` run of the same bench command (8 SQ slots per
pass on gfx950, GRBM independent; MI355X_MICROARCH.md 'rocprofv3 PMC slots').  Only the TIMED epoch's dispatches are used (the same tail
selection as tools/pmc_summary.py).  Units (MI355X_MICROARCH.md, cycle-constants table): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_*
count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs (32 per v_mfma_f32_32x32x16_bf16);
GRBM_GUI_ACTIVE is summed over the 8 XCDs.  Derived per family:

  mfma_busy        = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)      share of SIMD-cycles with the matrix pipe busy
  clock_ghz        = GRBM_GUI_ACTIVE / 8 / kernel-trace duration                         effective shader clock of the dispatch
  mfma_tflops_exec = SQ_INSTS_VALU_MFMA_MOPS_* * 512 / duration                          executed matrix FLOP/s (one MOPS unit = 512 FLOP)
  parked / issue_stall / active = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES   (disjoint, sum ~ 1)
`_csrc_sha16` records the kernel sources profiled (bench.py marks the figure stale when its own differ)."""
import collections, csv, glob, json, sys

from pmc_summary import FAMILIES, sha16

N_SIMD = 256 * 4
N_XCD = 8


def load_pass(d):
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    dur = {}
    if kt:
        for r in csv.DictReader(open(kt[0])):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    per = collections.defaultdict(lambda: collections.defaultdict(dict))  # fam -> dispatch -> counter -> value
    for r in csv.DictReader(open(cc[0])):
        for key, (fam, _) in FAMILIES.items():
            if key in r["Kernel_Name"]:
                ent = per[fam][int(r["Dispatch_Id"])]
                ent[r["Counter_Name"]] = ent.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
                if r["Dispatch_Id"] in dur:
                    ent["_dur"] = dur[r["Dispatch_Id"]]
                break
    return per


def main():
    steps, out, dirs = int(sys.argv[1]), sys.argv[2], sys.argv[3:]
    res = {"_csrc_sha16": sha16(), "_timed_steps": steps, "_passes": len(dirs),
           "_units": "counters are per-launch averages over the timed epoch; *_CYCLES of waves in quad-cycles, MFMA_BUSY in SIMD-cycles, GRBM over 8 XCDs"}
    acc = collections.defaultdict(lambda: collections.defaultdict(list))  # fam -> counter -> [per-pass average]
    launches = {}
    for d in dirs:
        per = load_pass(d)
        for key, (fam, per_pass) in FAMILIES.items():
            if fam not in per:
                continue
            n = steps if per_pass is None else per_pass * (steps + 1)
            tail = [per[fam][k] for k in sorted(per[fam])][-n:]
            launches[fam] = len(tail)
            for c in sorted(set().union(*[set(t) for t in tail])):
                vals = [t[c] for t in tail if c in t]
                acc[fam][c].append(sum(vals) / len(vals))
    fams = {}
    for fam, cs in acc.items():  # GRBM_GUI_ACTIVE and the durations repeat in every pass: mean over the passes
        ent = {c: sum(v) / len(v) for c, v in cs.items() if c != "_dur"}
        ent["launches"] = launches[fam]
        if "_dur" in cs:
            ent["duration_ms"] = sum(cs["_dur"]) / len(cs["_dur"]) * 1e3
        fams[fam] = ent
    for fam, ent in fams.items():
        gui, dur = ent.get("GRBM_GUI_ACTIVE"), ent.get("duration_ms")
        if gui and dur:
            ent["clock_ghz"] = gui / N_XCD / (dur * 1e-3) * 1e-9
        if gui and "SQ_VALU_MFMA_BUSY_CYCLES" in ent:
            ent["mfma_busy"] = ent["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / N_XCD * N_SIMD)
        mops = sum(v for k, v in ent.items() if k.startswith("SQ_INSTS_VALU_MFMA_MOPS_"))
        if mops and dur:
            ent["mfma_tflops_executed"] = mops * 512 / (dur * 1e-3) / 1e12
        wc = ent.get("SQ_WAVE_CYCLES")
        if wc:
            for name, c in (("parked", "SQ_WAIT_ANY"), ("issue_stall", "SQ_WAIT_INST_ANY"), ("active", "SQ_ACTIVE_INST_ANY")):
                if c in ent:
                    ent["wave_" + name] = ent[c] / wc
        if gui and "SQ_BUSY_CYCLES" in ent:
            ent["sq_busy_over_gui"] = ent["SQ_BUSY_CYCLES"] / gui
        res[fam] = ent
        if fam.endswith("bf16x"):
            res[fam + "3"] = res[fam + "6"] = res[fam.replace("bf16x", "f16x3")] = ent
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if not k.endswith(("x3", "x6"))}, indent=1))


if __name__ == "__main__":
    main()
