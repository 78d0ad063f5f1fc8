#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the kernels matching a name pattern in one Python tool (two separate counter passes, as the guide prescribes).
#   gpurun --timeout 600 -- 'bash tools/pmc_kernel_bytes.sh tools/merge_bwd_bench.py merge_bwd_stage1'
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}
export PYTHONPATH=$R
TOOL=$1; PAT=$2
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pkb_$c
  timeout -k 10 300 rocprofv3 --pmc $c -d /tmp/pkb_$c --output-format csv -- python3 $R/$TOOL > /tmp/pkb_$c.out 2> /tmp/pkb_$c.err || { tail -5 /tmp/pkb_$c.err; exit 1; }
done
python3 - "$PAT" <<'P'
import collections, csv, glob, sys
pat = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(list))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"/tmp/pkb_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"] and r["Counter_Name"] == c:
                name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
                key = (name.split("(")[0].strip() or name[:60], r["Grid_Size"] if "Grid_Size" in r else "")
                res[key][c].append(float(r["Counter_Value"]))
for key, d in sorted(res.items()):
    f, w = d.get("FETCH_SIZE", [0]), d.get("WRITE_SIZE", [0])
    print(f"{key[0][:60]:60s} grid {key[1]:>10s}  launches {len(f):3d}  FETCH raw {sum(f)/len(f)*1024/1e6:9.1f} MB (x2: {sum(f)/len(f)*2048/1e6:9.1f})  WRITE {sum(w)/max(len(w),1)*1024/1e6:8.2f} MB")
P
grep -v amdgpu /tmp/pkb_FETCH_SIZE.out | tail -8
