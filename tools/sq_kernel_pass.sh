#!/bin/bash
# SQ counters of ANY python command, aggregated per kernel name (two rocprofv3 --pmc passes, --kernel-trace only):
#   gpurun --timeout 900 -- 'bash tools/sq_kernel_pass.sh <substring of the kernel names to keep> python3 <script> [args]'
# Prints per kernel: launches, avg us, matrix-pipe busy share, effective clock, wave-cycle split (parked / issue-stalled / issuing), LDS wait share,
# VALU and MFMA instruction counts per launch.  The program after -- is python3 itself (no env / bash hop: MI355X pool rule).
set -eo pipefail
PAT=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sqk1 /tmp/sqk2
timeout -k 10 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace -d /tmp/sqk1 --output-format csv -- "$@" > /tmp/sqk1.out 2> /tmp/sqk1.err || { tail -8 /tmp/sqk1.err; exit 1; }
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace -d /tmp/sqk2 --output-format csv -- "$@" > /tmp/sqk2.out 2> /tmp/sqk2.err || { tail -8 /tmp/sqk2.err; exit 1; }
python3 - "$PAT" <<'PY'
import collections, csv, glob, sys
pat = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for d in ("/tmp/sqk1", "/tmp/sqk2"):
    kt = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    cc = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    dur, name = {}, {}
    for r in csv.DictReader(open(kt)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
        name[r["Dispatch_Id"]] = r["Kernel_Name"]
    seen = set()
    for r in csv.DictReader(open(cc)):
        k = r["Kernel_Name"]
        if pat not in k:
            continue
        import re
        m_ = re.search(r"(\w+<[^>]*>)", k)
        k = m_.group(1) if m_ else k[:60]
        agg[k][r["Counter_Name"] + "@" + d[-1]] += float(r["Counter_Value"])
        if (r["Dispatch_Id"], d) not in seen:
            seen.add((r["Dispatch_Id"], d))
            agg[k]["n@" + d[-1]] += 1
            agg[k]["dur@" + d[-1]] += dur.get(r["Dispatch_Id"], 0.0)
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["dur@1"]):
    n = max(a["n@1"], 1)
    gui = a["GRBM_GUI_ACTIVE@1"] / 8
    wc = max(a["SQ_WAVE_CYCLES@2"], 1)
    print(f"{k}\n   launches {int(n)}  avg {a['dur@1'] / n * 1e6:7.1f} us  clock {gui / max(a['dur@1'], 1e-12) / 1e9:.2f} GHz  matrix pipe busy {a['SQ_VALU_MFMA_BUSY_CYCLES@1'] / max(gui * 1024, 1):.3f}  "
          f"fp32 MFMA {a['SQ_INSTS_VALU_MFMA_MOPS_F32@1'] * 512 / max(a['dur@1'], 1e-12) / 1e12:.1f} TFLOP/s\n"
          f"   waves: parked {a['SQ_WAIT_ANY@2'] / wc:.2f}  issue-stalled {a['SQ_WAIT_INST_ANY@2'] / wc:.2f}  issuing {a['SQ_ACTIVE_INST_ANY@2'] / wc:.2f}  (LDS wait {a['SQ_WAIT_INST_LDS@2'] / wc:.2f}, VALU issue {a['SQ_ACTIVE_INST_VALU@2'] / wc:.2f}, LDS issue {a['SQ_ACTIVE_INST_LDS@2'] / wc:.2f})\n"
          f"   per launch: MFMA insts {a['SQ_INSTS_MFMA@1'] / n:.0f}  VALU insts {a['SQ_INSTS_VALU@1'] / n:.0f}  LDS insts {a['SQ_INSTS_LDS@1'] / n:.0f}")
PY
