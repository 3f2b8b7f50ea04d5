#!/bin/bash
# GPU-side: tests that use the native FASTQ ingest, then the bench ingest leg at a few thread counts
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pe_gpu.py tests/test_host_cpu.py -x -q -k "golden or ingest or cli or ragged or variable" 2>&1 | tail -4
P='import sys,json; d=json.loads(sys.stdin.read()); print(json.dumps(d.get("fastq_ingest")))'
for t in 64 128 32; do
  echo "== VS_HOST_THREADS=$t"; VS_HOST_THREADS=$t timeout 600 python bench.py --config 2 --steps 2 --warmup 1 --cpu-seconds 0 --pairs 4000000 2>gpurun_out/ingest.err | python -c "$P" || tail -5 gpurun_out/ingest.err
done
