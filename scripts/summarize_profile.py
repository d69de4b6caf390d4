#!/usr/bin/env python3
"""Turn rocprofv3 CSV output (kernel trace + separate --pmc passes) into the small summary
files committed under profiles/.

    python scripts/summarize_profile.py <prof_dir> <tag> <workload> <samples_per_launch> [kernel_substr]

<prof_dir> holds trace/ pmc_fetch/ pmc_write/ as written by the gpurun recipe in DESIGN.md.
HBM traffic per launch follows MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in
KiB; on gfx950 FETCH_SIZE counts 128-byte fabric read requests as 64 bytes for coalesced
streaming reads, so the read side is doubled; WRITE_SIZE is exact for streaming stores (the
synthetic-source kernel in the same trace writes a known 2^30 bytes and checks that here).
"""
import collections
import csv
import glob
import json
import os
import sys


def load(pattern):
    f = sorted(glob.glob(pattern), key=os.path.getmtime)   # the newest run if several were merged
    return list(csv.DictReader(open(f[-1]))) if f else []


def main():
    prof, tag, workload, samples = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    ksub = sys.argv[5] if len(sys.argv) > 5 else "qk::fir"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = os.path.join(root, "profiles")
    os.makedirs(out_dir, exist_ok=True)

    stats = load(os.path.join(prof, "trace", "*", "*_kernel_stats.csv"))
    trace = load(os.path.join(prof, "trace", "*", "*_kernel_trace.csv"))
    durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in trace if ksub in r["Kernel_Name"]]
    meta = next((r for r in trace if ksub in r["Kernel_Name"]), {})
    pmc = {}
    for kind, name in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        rows = load(os.path.join(prof, f"pmc_{kind}", "*", "*_counter_collection.csv"))
        agg = collections.defaultdict(list)
        for r in rows:
            if r["Counter_Name"] == name:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        pmc[name] = {k: sum(v) / len(v) for k, v in agg.items()}
    kname = next((k for k in pmc.get("FETCH_SIZE", {}) if ksub in k), None)
    fetch_kib = pmc.get("FETCH_SIZE", {}).get(kname)
    write_kib = pmc.get("WRITE_SIZE", {}).get(kname)
    synth_w = next((v for k, v in pmc.get("WRITE_SIZE", {}).items() if "synth_iq" in k), None)
    traffic = None
    if fetch_kib is not None and write_kib is not None:
        traffic = 2.0 * fetch_kib * 1024 + write_kib * 1024
    tail = durs[-10:] if len(durs) >= 10 else durs
    summary = {
        "tag": tag,
        "workload": workload,
        "samples_per_launch": samples,
        "kernel": meta.get("Kernel_Name"),
        "launch": {k: meta.get(k) for k in ("Grid_Size_X", "Workgroup_Size_X", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size")},
        "dispatches": len(durs),
        "avg_us_all": round(sum(durs) / len(durs), 2) if durs else None,
        "avg_us_last10": round(sum(tail) / len(tail), 2) if tail else None,
        "min_us": round(min(durs), 2) if durs else None,
        "max_us": round(max(durs), 2) if durs else None,
        "durations_us": [round(d, 1) for d in durs],
        "pmc": {
            "FETCH_SIZE_KiB_per_launch": fetch_kib,
            "WRITE_SIZE_KiB_per_launch": write_kib,
            "read_bytes_corrected_x2": None if fetch_kib is None else 2.0 * fetch_kib * 1024,
            "write_bytes": None if write_kib is None else write_kib * 1024,
            "hbm_bytes_per_launch": traffic,
            "calibration_synth_iq_WRITE_SIZE_KiB": synth_w,
            "calibration_expected_KiB": samples * 8 / 1024,
        },
        "kernel_stats_csv": stats,
    }
    with open(os.path.join(out_dir, f"{tag}_{workload}_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    # copy the raw stats CSVs next to it (small)
    for src, dst in ((os.path.join(prof, "trace", "*", "*_kernel_stats.csv"), f"{tag}_{workload}_kernel_stats.csv"),):
        g = sorted(glob.glob(src), key=os.path.getmtime)
        if g:
            open(os.path.join(out_dir, dst), "w").write(open(g[-1]).read())
    if traffic is not None:
        tj = os.path.join(out_dir, "traffic.json")
        d = json.load(open(tj)) if os.path.exists(tj) else {}
        d[workload] = {"samples": samples, "hbm_bytes_per_launch": traffic,
                       "source": f"profiles/{tag}_{workload}_summary.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)"}
        json.dump(d, open(tj, "w"), indent=1)
    print(json.dumps({k: summary[k] for k in ("kernel", "dispatches", "avg_us_all", "avg_us_last10", "min_us", "pmc")}, indent=1))


if __name__ == "__main__":
    main()
