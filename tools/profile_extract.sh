#!/bin/bash
# rocprofv3 kernel trace of the strain-extract leg of one config (graph kernels: k_vertex_scan, k_edge_flow, k_chain_*,
# k_links_*), next to the host-side split of the flow/scan operation.  Run on the GPU box:
#   bash tools/profile_extract.sh 4        -> gpurun_out/extract_prof_c4/
cfg=${1:-2}
out=gpurun_out/extract_prof_c${cfg}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
VS_STAGE_OP_TIMING=1 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$out -o extract -- python3 $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 2 --warmup 1 \
    --cpu-seconds 0 --ingest-pairs 0 --extract > $GRAFT_REPO_ROOT/$out/bench.json 2> $GRAFT_REPO_ROOT/$out/bench.err
cd $GRAFT_REPO_ROOT
grep "\[vs\]" $out/bench.err
f=$(find $out -name "*kernel_stats.csv" | head -1)
head -25 "$f"
cp "$f" $out/kernel_stats.csv 2>/dev/null
