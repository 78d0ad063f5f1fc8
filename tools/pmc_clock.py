"""Effective shader clock per kernel family from one rocprofv3 pass with `--pmc GRBM_GUI_ACTIVE --kernel-trace`:
clock = GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration (MI355X_MICROARCH.md, 'DVFS give-back'; reads high on dispatches under ~0.3 ms).
Usage: python tools/pmc_clock.py <dir> <out.json>"""
import collections, csv, glob, json, re, sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name).replace("void ", "")
    return name.split("<")[0].split("(")[0]


def main():
    d, out = sys.argv[1], sys.argv[2]
    cnt = {}
    for r in csv.DictReader(open(glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0])):
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[r["Dispatch_Id"]] = (float(r["Counter_Value"]), r["Kernel_Name"])
    dur = {}
    for r in csv.DictReader(open(glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0])):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    fam = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for k, (c, name) in cnt.items():
        if k in dur and dur[k] > 2e-4:  # only dispatches long enough for the quotient to be meaningful
            f = fam[short(name)]
            f[0] += 1; f[1] += c / 8.0; f[2] += dur[k]
    res = {k: dict(dispatches=v[0], mean_duration_ms=v[2] / v[0] * 1e3, effective_clock_ghz=v[1] / v[2] * 1e-9) for k, v in fam.items()}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
