"""Cut a rocprofv3 --pmc counter_collection.csv of tools/pmc_target.py into its sections (between `spin_kernel` markers) and
sum every counter per section and launch.   usage: python tools/pmc_collect.py <dir with the csv> <manifest.json> <out.json>
Output: [{cls, label, calls_per_microstep, counters: {NAME: value per LAUNCH of the op (all its kernels summed)}}]."""
import csv
import glob
import json
import sys


def main(d, manifest, out):
    man = json.load(open(manifest))
    rows = []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        with open(f, newline="") as fh:
            rows += list(csv.DictReader(fh))
    key = "Dispatch_Id" if rows and "Dispatch_Id" in rows[0] else "Dispatch_ID"
    byd = {}
    for r in rows:
        d_ = byd.setdefault(int(r[key]), dict(name=r["Kernel_Name"], c={}))
        d_["c"][r["Counter_Name"]] = d_["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    sections, cur, nmark = [], None, 0
    for i in sorted(byd):
        if "spin_kernel" in byd[i]["name"]:
            nmark += 1
            if nmark % 2 == 1:
                cur = []                       # opening marker of a pair
            else:
                sections.append(cur)           # closing marker
                cur = None
        elif cur is not None:
            cur.append(byd[i])
    if len(sections) != len(man):
        raise SystemExit(f"{len(sections)} marker sections for {len(man)} manifest entries")
    res = []
    for m, sec in zip(man, sections):
        tot, kern = {}, {}
        for k in sec:
            kern[k["name"][:60]] = kern.get(k["name"][:60], 0) + 1
            for c, v in k["c"].items():
                tot[c] = tot.get(c, 0.0) + v
        res.append(dict(m, kernels={k: n // m["reps"] for k, n in kern.items()}, counters={c: v / m["reps"] for c, v in tot.items()}))
    json.dump(res, open(out, "w"), indent=1)
    print(f"{len(res)} sections -> {out}")


if __name__ == "__main__":
    main(*sys.argv[1:4])
