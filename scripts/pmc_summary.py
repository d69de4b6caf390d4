#!/usr/bin/env python3
"""rocprofv3 --pmc passes of scripts/pmc.sh (one directory per counter set) -> profiles/<tag>_pmc_<kernel>.json:
mean of each counter over the dispatches of the kernel, plus the few derived figures DESIGN.md quotes.

    python scripts/pmc_summary.py <pmc_dir> <tag> <kernel_substr> <units_per_launch> [unit_name]
"""
import collections
import csv
import glob
import json
import os
import sys


def main():
    d, tag, ksub, units = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
    unit = sys.argv[5] if len(sys.argv) > 5 else "segment"
    agg = collections.defaultdict(list)
    names = set()
    for f in glob.glob(os.path.join(d, "p*", "*", "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if ksub in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
                names.add(r["Kernel_Name"].split("(")[0])
    if not agg:
        raise SystemExit(f"no dispatches of {ksub} under {d}")
    c = {k: sum(v) / len(v) for k, v in sorted(agg.items())}
    g = c.get("GRBM_GUI_ACTIVE")
    wc = c.get("SQ_WAVE_CYCLES")
    der = {}
    if g:
        der["kernel_cycles_per_xcd"] = g / 8.0      # GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md, DVFS note)
    if wc:
        for k in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
            if k in c:
                der[k + "_frac_of_wave_cycles"] = round(c[k] / wc, 4)
    if g and "SQ_WAVE_CYCLES" in c:
        der["mean_resident_waves"] = round(c["SQ_WAVE_CYCLES"] * 4.0 / (g / 8.0), 1)      # SQ_WAVE_CYCLES counts quad-cycles
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU"):
        if k in c:
            der[f"{k}_per_{unit}"] = round(c[k] / units, 1)
    if "SQ_LDS_IDX_ACTIVE" in c and g:
        der["lds_busy_frac"] = round(c["SQ_LDS_IDX_ACTIVE"] / 256.0 / (g / 8.0), 4)
        if "SQ_LDS_BANK_CONFLICT" in c and c["SQ_LDS_IDX_ACTIVE"] > 0:
            der["lds_bank_conflict_frac_of_lds_cycles"] = round(c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"], 4)
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        der["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
    out = {"tag": tag, "kernel": sorted(names), "units_per_launch": units, "unit": unit, "dispatches_averaged": {k: len(v) for k, v in agg.items()},
           "counters_mean_per_dispatch": c, "derived": der,
           "how": "scripts/pmc.sh: one rocprofv3 --pmc pass per counter set (never combined with tracing), bench.py --steps 2 --warmup 1"}
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    short = ksub.replace("qk::", "").replace("<", "_").replace(">", "").replace(",", "_").replace(" ", "")
    p = os.path.join(root, "profiles", f"{tag}_pmc_{short}.json")
    json.dump(out, open(p, "w"), indent=1)
    print(p, json.dumps(der))


if __name__ == "__main__":
    main()
