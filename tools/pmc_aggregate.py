"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel (short name) -> json.
usage: python tools/pmc_aggregate.py <dir with *_counter_collection.csv> <out.json>
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB; the gfx950 correction (FETCH_SIZE tallies 128-B requests
at 64 B -> double it; MI355X_MICROARCH.md "HBM") is applied by the consumer (bench.py), not here."""
import csv, glob, json, re, sys, collections

def short(name):
    name = re.sub(r"^void ", "", name)
    m = re.match(r"([A-Za-z_0-9:]+(<[^()]*>)?)", name)
    s = m.group(1) if m else name[:80]
    return s[:120]

def main(d, out):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                k = (short(row["Kernel_Name"]), row["Counter_Name"])
                agg[k][0] += 1
                agg[k][1] += float(row["Counter_Value"])
    res = {}
    for (k, c), (n, s) in sorted(agg.items()):
        res.setdefault(k, {})[c] = {"launches": n, "sum": s, "mean": s / n}
    json.dump(res, open(out, "w"), indent=1)
    print(f"{len(res)} kernels -> {out}")

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
