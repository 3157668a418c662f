"""Merge the four PMC passes of tools/r02_evidence.sh into profiles/r02_pmc.json (every section: counters per launch + derived
HBM-side bytes, MFMA-busy share, LDS share) and one profiles/r02_pmc_<class>.json per class (call-weighted means; bench.py's
fallback when it cannot run the profiler itself).   usage: python tools/pmc_merge_r02.py gpurun_out profiles"""
import json
import os
import sys


def main(odir, pdir):
    merged = {}
    for tag in ("fetch", "write", "sq", "sq2"):
        f = os.path.join(odir, f"r02_pmc_{tag}.json")
        if not os.path.exists(f):
            continue
        for sec in json.load(open(f)):
            key = (sec["cls"], sec["label"])
            m = merged.setdefault(key, {k: v for k, v in sec.items() if k != "counters"})
            m.setdefault("counters", {}).update(sec["counters"])
    out = []
    for (cls, label), m in merged.items():
        c = m["counters"]
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        m["hbm_side_bytes_per_launch"] = 2 * c.get("FETCH_SIZE", 0.0) * 1024 + c.get("WRITE_SIZE", 0.0) * 1024
        m["traffic_over_algorithmic"] = m["hbm_side_bytes_per_launch"] / m["algorithmic_bytes_per_launch"] if m["algorithmic_bytes_per_launch"] else None
        m["mfma_busy_frac"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 256 * 4) if gui > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in c else None
        m["lds_active_frac"] = c["SQ_LDS_IDX_ACTIVE"] / (gui * 256) if gui > 0 and "SQ_LDS_IDX_ACTIVE" in c else None
        m["valu_insts_per_mfma"] = c["SQ_INSTS_VALU"] / c["SQ_INSTS_MFMA"] if c.get("SQ_INSTS_MFMA") else None
        out.append(m)
    note = ("rocprofv3 --pmc, separate passes per counter set (tools/r02_evidence.sh, tools/pmc_target.py: 3 launches per shape between marker "
            "kernels, counters summed over every kernel of the op -- split-K reduce included -- and divided by the launches).  FETCH_SIZE / "
            "WRITE_SIZE are KiB; gfx950 correction: hbm_side_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (FETCH_SIZE tallies 128-B requests at "
            "64 B; Infinity-Cache hits are included: fabric-side bytes, an upper bound on HBM bytes).  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / "
            "(GRBM_GUI_ACTIVE/8 * 256 CUs * 4 SIMDs); lds_active_frac = SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE/8 * 256).")
    json.dump(dict(note=note, sections=out), open(os.path.join(pdir, "r02_pmc.json"), "w"), indent=1)
    classes = {}
    for m in out:
        classes.setdefault(m["cls"], []).append(m)
    for cls, ms in classes.items():
        n = sum(m["calls_per_microstep"] for m in ms)
        w = lambda k: sum(m[k] * m["calls_per_microstep"] for m in ms if m.get(k) is not None) / max(n, 1)
        json.dump(dict(note=note, cls=cls, launches_covered=n, traffic=w("hbm_side_bytes_per_launch"), algorithmic_bytes_per_launch=w("algorithmic_bytes_per_launch"),
                       mfma_busy_frac=w("mfma_busy_frac"), lds_active_frac=w("lds_active_frac")), open(os.path.join(pdir, f"r02_pmc_{cls}.json"), "w"), indent=1)
    for m in sorted(out, key=lambda m: (m["cls"], -m["calls_per_microstep"])):
        print(f'{m["cls"]:11s} {m["label"]:26s} calls {m["calls_per_microstep"]:4d}  traffic/alg {m["traffic_over_algorithmic"] or 0:5.2f}  mfma_busy {100 * (m["mfma_busy_frac"] or 0):5.1f}%  lds {100 * (m["lds_active_frac"] or 0):5.1f}%')


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
