#!/bin/bash
# Run on the GPU box: bench + rocprofv3 kernel trace + PMC passes per workload -> gpurun_out/prof_<w>/
R=${GRAFT_REPO_ROOT:-$PWD}
export TMPDIR=/tmp
cd /tmp
for W in "$@"; do
  P=$R/gpurun_out/prof_$W
  mkdir -p $P
  EXTRA=""; [ "$W" != "fir256" ] && EXTRA="--no-cpu-baseline"
  timeout -k 10 300 python3 $R/bench.py --workload $W $EXTRA > $P/bench.json 2> $P/bench.err || echo "bench $W failed"
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $P/trace -- python3 $R/bench.py --workload $W --steps 20 --warmup 10 --no-cpu-baseline --no-chain --no-block-call > $P/bench_under_trace.json 2> $P/trace.err || echo "trace $W failed"
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/pmc_fetch -- python3 $R/bench.py --workload $W --steps 3 --warmup 1 --kernel-iters 2 --spinup-ms 0 --no-cpu-baseline --no-chain --no-block-call > /dev/null 2> $P/pmc_fetch.err || echo "fetch $W failed"
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/pmc_write -- python3 $R/bench.py --workload $W --steps 3 --warmup 1 --kernel-iters 2 --spinup-ms 0 --no-cpu-baseline --no-chain --no-block-call > /dev/null 2> $P/pmc_write.err || echo "write $W failed"
  cat $P/bench.json
done
