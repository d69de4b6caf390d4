#!/bin/bash
# usage (on the GPU box): scripts/pmc.sh <workload> <outdir> -- collects SQ/LDS/GRBM counters in separate passes
W=$1; R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/$2
export TMPDIR=/tmp
mkdir -p $OUT
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS" \
           "GRBM_GUI_ACTIVE GRBM_COUNT" \
           "TCC_HIT_sum TCC_MISS_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 $R/bench.py --workload $W --steps 2 --warmup 1 --kernel-iters 2 --spinup-ms 0 --no-cpu-baseline --no-chain --no-block-call > /dev/null 2> $OUT/p$i.err || echo "pass $i failed"
done
python3 - <<PY
import csv,glob,collections
agg=collections.defaultdict(list)
for f in glob.glob("$OUT/p*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "qk::" in r["Kernel_Name"] and "synth" not in r["Kernel_Name"]:
            agg[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
for k in sorted(agg): print(k[0], k[1], len(agg[k]), sum(agg[k])/len(agg[k]))
PY
