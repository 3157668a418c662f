"""Merge the per-shape FETCH_SIZE / WRITE_SIZE passes of tools/r01e_evidence.sh with the per-shape call counts of
tools/class_breakdown.py into profiles/<round>_pmc_gemm_nt.json (the file bench.py reads `roofline.traffic` from).
usage: python tools/pmc_merge.py <gpurun_out dir> <shape_breakdown.txt> <out.json>"""
import glob, json, os, re, sys

NOTE = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes (tools/r01e_evidence.sh) on the dominant GEMM class' own "
        "shapes, 5 launches each; counters are KiB; gfx950 correction applied here: hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 "
        "(MI355X_MICROARCH.md, HBM: FETCH_SIZE tallies 128-B requests at 64 B; Infinity-Cache hits are included, so this is "
        "fabric-side traffic, an upper bound on HBM bytes).  calls/ms per micro-step from tools/class_breakdown.py (AZ_SHAPES=1).")


def gemm_entry(path, counter):
    d = json.load(open(path))
    for k, v in d.items():
        if "gemm_kernel" in k:
            return k, v[counter]["mean"]
    raise SystemExit(f"no gemm kernel in {path}")


def main(odir, breakdown, out):
    calls = {}
    for line in open(breakdown):
        m = re.match(r"\s*gemm_nt (\d+)x(\d+)x(\d+)\s+calls\s+(\d+)\s+([\d.]+) ms", line)
        if m:
            calls[tuple(int(x) for x in m.group(1, 2, 3))] = (int(m.group(4)), float(m.group(5)))
    shapes = []
    for f in sorted(glob.glob(os.path.join(odir, "pmcnt_*_FETCH_SIZE.json"))):
        M, N, K = (int(x) for x in re.search(r"pmcnt_(\d+)_(\d+)_(\d+)_FETCH", f).group(1, 2, 3))
        kern, fetch = gemm_entry(f, "FETCH_SIZE")
        _, write = gemm_entry(f.replace("FETCH_SIZE", "WRITE_SIZE"), "WRITE_SIZE")
        c, ms = calls.get((M, N, K), (0, 0.0))
        shapes.append({"M": M, "N": N, "K": K, "kernel": kern[23:71], "calls_per_microstep": c, "ms_per_microstep": ms,
                       "FETCH_SIZE_KiB": fetch, "WRITE_SIZE_KiB": write, "hbm_bytes_per_launch": 2 * fetch * 1024 + write * 1024,
                       "algorithmic_bytes_per_launch": 2 * (M * K + N * K + M * N)})
    shapes.sort(key=lambda s: -s["ms_per_microstep"])
    n = sum(s["calls_per_microstep"] for s in shapes)
    res = {"note": NOTE, "shapes": shapes, "calls_covered": n, "calls_in_class": sum(c for c, _ in calls.values()),
           "mean_hbm_bytes_per_launch": sum(s["hbm_bytes_per_launch"] * s["calls_per_microstep"] for s in shapes) / max(n, 1),
           "mean_algorithmic_bytes_per_launch": sum(s["algorithmic_bytes_per_launch"] * s["calls_per_microstep"] for s in shapes) / max(n, 1)}
    json.dump(res, open(out, "w"), indent=1)
    print(f"{len(shapes)} shapes, {n} of {res['calls_in_class']} launches -> {out}")


if __name__ == "__main__":
    main(*sys.argv[1:4])
