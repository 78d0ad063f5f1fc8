"""Summarise rocprofv3 --pmc passes of bench.py into per-kernel-family HBM traffic per launch.

Usage: python tools/pmc_summary.py <fetch_dir> <write_dir> <n_timed_steps> <out.json>
FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE under-reports wide coalesced streaming reads by 2x
(MI355X_MICROARCH.md, HBM section) -- both the raw and the corrected read figure are kept.
Only the dispatches of the timed region are used: the last `launches_per_step * n_timed_steps` of each family."""
import collections, csv, glob, json, sys

# kernel-name substring -> (family name used by bench.py's LaunchProfiler, launches per bench step)
FAMILIES = {"gemm_nt_bf16x6_kernel": ("gemm_nt_bf16x", 48), "gemm_nt_kernel": ("gemm_nt", 48), "attn_kernel": ("attention", 12), "attn_split_kernel": ("attention_bf16x", 12),
            "merge_nway_kernel": ("merge_nway", 1), "split_weights_kblock_kernel": ("split_weights", 1),
            "embed_gather_ln_kernel": ("embed_gather_ln", 1), "layernorm_kernel": ("layernorm", 24), "topk_rows_kernel": ("topk_rows", 1)}


def load(d, counter):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        for key, (fam, _) in FAMILIES.items():
            if key in r["Kernel_Name"]:
                per[fam].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
                break
    return per


def main():
    fetch_dir, write_dir, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    fe, wr = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
    res = {}
    for key, (fam, per_step) in FAMILIES.items():
        n = per_step * steps
        f = [v for _, v in sorted(fe.get(fam, []))][-n:]
        w = [v for _, v in sorted(wr.get(fam, []))][-n:]
        if not f:
            continue
        res[fam] = dict(launches=len(f), fetch_bytes_raw_per_launch=sum(f) / len(f) * 1024,
                        fetch_bytes_x2_per_launch=sum(f) / len(f) * 2048, write_bytes_per_launch=sum(w) / max(len(w), 1) * 1024)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
