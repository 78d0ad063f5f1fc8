#!/bin/bash
# FETCH_SIZE of the encoder GEMM under different tile orders (MR_GEMM_GROUPN = column tiles per group; 0 = library default), with the
# kernel's duration from the bench's own HIP events in a separate plain run.   gpurun --timeout 900 -- 'bash tools/ab_fetch.sh 0 1 2 3'
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for g in "$@"; do
  export MR_GEMM_GROUPN=$g
  rm -rf /tmp/abf_$g
  timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE -d /tmp/abf_$g --output-format csv -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-profile > /dev/null 2> /tmp/abf_$g.err || { tail -5 /tmp/abf_$g.err; exit 1; }
  timeout -k 10 400 python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > /tmp/abf_$g.json 2>> /tmp/abf_$g.err || { tail -5 /tmp/abf_$g.err; exit 1; }
  python3 - $g <<'P'
import csv, glob, json, sys
g = sys.argv[1]
rows = []
for f in glob.glob(f"/tmp/abf_{g}/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "gemm_nt_bf16x6_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
            rows.append((int(r.get("Dispatch_Id", 0)), float(r["Counter_Value"])))
rows.sort()
tail = [v for _, v in rows[-539:]]
d = json.loads(open(f"/tmp/abf_{g}.json").read().strip().splitlines()[-1])
k = d["kernels"]["gemm_nt_bf16x3"]
print(f"GROUPN={g}: FETCH_SIZE raw {sum(tail)/len(tail)*1024/1e6:8.1f} MB/launch (x2 corrected {sum(tail)/len(tail)*2048/1e6:8.1f}) over {len(tail)} launches; "
      f"kernel {k['avg_ms']:.4f} ms; step {d['ms_per_step']:.2f} ms; {d['value']:.0f} seq/s", flush=True)
P
  rm -rf /tmp/abf_$g
done
