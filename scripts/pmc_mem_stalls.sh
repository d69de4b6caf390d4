#!/bin/bash
# usage (on the GPU box): scripts/pmc_mem_stalls.sh <workload> ... -- L2 / fabric / L1 stall counters of a bench workload's kernel, one rocprofv3 --pmc pass
# per small counter set (never combined with tracing), printed as the mean per dispatch.
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
cd /tmp
for W in "$@"; do
  OUT=$R/gpurun_out/pmc_mem_$W
  rm -rf $OUT; mkdir -p $OUT
  i=0
  for set in "TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_BUSY_sum" \
             "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum TCC_IB_STALL_sum TCC_EA0_WRREQ_sum" \
             "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_BUSY_avr" \
             "TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --kernel-iters 2 --spinup-ms 0 --no-cpu-baseline --no-chain --no-block-call > /dev/null 2> $OUT/p$i.err || echo "pass $i failed"
  done
  python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "qk::" in r["Kernel_Name"] and "synth" not in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0][-44:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(agg): print("$W", k[0], k[1], len(agg[k]), "%.4g" % (sum(agg[k])/len(agg[k])))
PY
done
