#!/bin/bash
# Throughput of live block graphs as a function of the stream block size (profiles/r04_graph_bench.txt): the default build at the
# reference's 1e6-element blocks and 65536, and graph_check_big (STREAM_BUFFER_SIZE = 2^24) from 2^20 to 2^24.
#   bash scripts/graph_bench.sh > gpurun_out/r04/graph_bench.txt        (on the GPU box)
set -u
H=qdsp_amd/host/build
echo "# graph_check bench <kind> <block> <nblocks> 2400000 48000: SineSource -> ... -> sinks on the block-graph mirror, one thread per block, device-resident links"
for kind in vfo chain split4 split16 hostfir hostvfo; do
  for bs in 65536 1000000; do $H/graph_check bench $kind $bs 300 2400000 48000 2>/dev/null; done
  for bs in 1048576 2097152 4194304 8388608 16777216; do
    nb=$((300 * 1048576 / bs)); [ $nb -lt 20 ] && nb=20
    $H/graph_check_big bench $kind $bs $nb 2400000 48000 2>/dev/null | sed 's/^bench/bench[2^24 build]/'
  done
done
