"""Summarise rocprofv3 --pmc passes of bench.py into per-kernel-family HBM-side traffic per launch.

Usage: python tools/pmc_summary.py <fetch_dir> <write_dir> <n_timed_steps> <out.json> [<l2_dir>]
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced streaming reads by 2x
(MI355X_MICROARCH.md, HBM section) -- both the raw and the corrected read figure are kept.  FETCH_SIZE counts what leaves the
XCD's L2 towards the fabric: Infinity-Cache (MALL) hits are INCLUDED, so for a GEMM whose weight panel (7-9 MB) exceeds the 4 MB L2
the figure holds the panel's re-reads out of the MALL as well as the HBM reads.  <l2_dir> (optional): a pass with
`--pmc TCC_HIT_sum TCC_MISS_sum` -> L2 hit rate per family.
Only the dispatches of the timed epoch are used: bench.py's timed region is ONE catalog pass + n_timed_steps user passes, i.e. the
last `per_pass * (n_timed_steps + 1)` dispatches of the per-pass families and the last `n_timed_steps` of the per-step ones.
`_csrc_sha16` records which kernel sources were profiled; bench.py flags the figure as stale when its own sources differ."""
import collections, csv, glob, hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# kernel-name substring -> (family name used by bench.py's LaunchProfiler, launches per encoder pass (None: one per step))
FAMILIES = {"gemm_nt_bf16x6_kernel": ("gemm_nt_bf16x", 49), "gemm_nt_kernel": ("gemm_nt", 49), "attn_kernel": ("attention", 11),
            "attn_split_kernel": ("attention_bf16x", 11), "attn_split_work_kernel": ("attention_bf16x", 11), "topk_rows_reg_kernel": ("topk_rows", None), "merge_nway_kernel": ("merge_nway", 1), "split_weights_kblock_kernel": ("split_weights", 1),
            "embed_gather_ln_kernel": ("embed_gather_ln", 1), "layernorm_kernel": ("layernorm", 24), "topk_rows_kernel": ("topk_rows", None)}


def load(d, counter):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        for key, (fam, _) in FAMILIES.items():
            if key in r["Kernel_Name"]:
                per[fam].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
                break
    return per


def sha16():
    h = hashlib.sha256()
    for p in sorted(glob.glob(os.path.join(ROOT, "mergerec_amd", "csrc", "*.hip"))):
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def main():
    fetch_dir, write_dir, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    l2_dir = sys.argv[5] if len(sys.argv) > 5 else None
    fe, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    hit = load(l2_dir, "TCC_HIT_sum") if l2_dir else {}
    miss = load(l2_dir, "TCC_MISS_sum") if l2_dir else {}
    res = {"_csrc_sha16": sha16(), "_timed_steps": steps}
    for key, (fam, per_pass) in FAMILIES.items():
        n = steps if per_pass is None else per_pass * (steps + 1)
        tail = lambda d: [v for _, v in sorted(d.get(fam, []))][-n:]
        f, w = tail(fe), tail(wr)
        if not f:
            continue
        ent = dict(launches=len(f), fetch_bytes_raw_per_launch=sum(f) / len(f) * 1024, fetch_bytes_x2_per_launch=sum(f) / len(f) * 2048,
                   write_bytes_per_launch=sum(w) / max(len(w), 1) * 1024)
        h, m = tail(hit), tail(miss)
        if h and m:
            ent["l2_hit_rate"] = sum(h) / max(sum(h) + sum(m), 1.0)
        res[fam] = ent
        if fam.endswith("bf16x"):  # bench.py names the family by its product count
            res[fam + "3"] = res[fam + "6"] = res[fam.replace("bf16x", "f16x3")] = ent  # (f16x3 runs the same kernel templates)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({k: v for k, v in res.items() if not k.endswith(("x3", "x6"))}, indent=1))


if __name__ == "__main__":
    main()
