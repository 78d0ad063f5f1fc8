"""Reduce rocprofv3 --pmc counter_collection.csv files to per-(kernel, grid) averages.
Usage: python tools/pmc_counters.py <dir> [<dir> ...]   (prints a table; kernel names are shortened)"""
import collections, csv, glob, re, sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([\w:]+)(<[^(]*>)?", name)
    return (m.group(1) + (m.group(2) or ""))[:60] if m else name[:60]


def main():
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in sys.argv[1:]:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                acc[(short(r["Kernel_Name"]), r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for (k, g), cs in sorted(acc.items()):
        print(f"{k} grid={g}")
        for c, v in sorted(cs.items()):
            print(f"    {c:32s} n={len(v):4d} avg={sum(v) / len(v):.4g}")


if __name__ == "__main__":
    main()
