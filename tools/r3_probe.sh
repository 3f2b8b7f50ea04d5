#!/bin/bash
# GPU-side: table probe with one / two slots per round trip, table fill 1/8 .. 1/64
R="$GRAFT_REPO_ROOT"; cd "$R"; mkdir -p gpurun_out
CFG=2 SWEEP_ENVS="VS_TABLE_SHIFT=4 VS_TABLE_SHIFT=5 VS_TABLE_SHIFT=6" bash tools/acc_sweep.sh "-DVS_PROBE_PAIR=0" 2>&1 | tee gpurun_out/r3_probe_c2.log
CFG=4 SWEEP_ENVS="VS_TABLE_SHIFT=5" bash tools/acc_sweep.sh "-DVS_PROBE_PAIR=0" 2>&1 | tee gpurun_out/r3_probe_c4.log
CFG=3 SWEEP_ENVS="VS_TABLE_SHIFT=5" bash tools/acc_sweep.sh "-DVS_PROBE_PAIR=0" 2>&1 | tee gpurun_out/r3_probe_c3.log
