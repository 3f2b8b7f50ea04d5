#!/bin/bash
# GPU-side: LDS cell table with the cells of a 64-byte stretch in neighbouring slots (ACC_SEG) against the plain hash
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
export VS_EXPERIMENT=1
python -m pytest tests/test_pe_gpu.py tests/test_configs_gpu.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r3_seg_tests.log; tail -3 gpurun_out/r3_seg_tests.log
for c in 4 2 3; do CFG=$c SWEEP_ENVS="VS_ACC_FILL=12 VS_ACC_FILL=25" bash tools/acc_sweep.sh "-DACC_SEG=0" 2>&1 | tee gpurun_out/r3_seg_c$c.log; done
python -m pytest tests/test_graph_gpu.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r3_seg_graph_tests.log; tail -3 gpurun_out/r3_seg_graph_tests.log
