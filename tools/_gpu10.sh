cd $GRAFT_REPO_ROOT
F='^RCCL\|^HIP ver\|^ROCm\|^Host\|^Librccl\|amdgpu.ids'
timeout 900 python -m pytest tests/test_pe_gpu.py -m gpu -x -q -k "long_stride or occupied" 2>&1 | grep -v "$F" | tail -3
LINE='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print(r["kernel"], "tiles %.3f counters %.3f sort %.3f overflow %.3f step %.3f pairs/s %.4g" % (r["kernel_ms_avg"], r["accumulate_ms_avg"], r["locus_sort_ms_avg"], r["slow_kernel_ms_avg"], d["ms_per_step"], d["value"]))'
{
for L in 250 256 270 300 317 318; do
  echo -n "configs[3] graph, 4 M pairs of 2 x $L: "
  timeout 600 python bench.py --config 3 --read-len $L --pairs 4000000 --steps 5 --warmup 1 --cpu-seconds 0 --ingest-pairs 0 --no-extract 2>/dev/null | python -c "$LINE"
done
echo -n "configs[3] graph, 4 M pairs of 2 x 270, VS_NO_FAST=1 (the generic kernel these lengths took before round 5): "
VS_EXPERIMENT=1 VS_NO_FAST=1 timeout 600 python bench.py --config 3 --read-len 270 --pairs 4000000 --steps 5 --warmup 1 --cpu-seconds 0 --ingest-pairs 0 --no-extract 2>/dev/null | python -c "$LINE"
} 2>&1 | tee gpurun_out/r5_long_reads_k127.log
timeout 900 python bench.py --config 5 --steps 5 --warmup 1 --cpu-seconds 0 --ingest-pairs 0 > gpurun_out/r5_bench_config5_with_extract.json 2> gpurun_out/r5_bench_config5.err; tail -c 300 gpurun_out/r5_bench_config5.err
timeout 900 python bench.py > gpurun_out/r5_bench_default_args.json 2> gpurun_out/r5_bench_default_args.err; tail -c 300 gpurun_out/r5_bench_default_args.err
timeout 1500 python tools/e2e_cli.py 10000000 15 13200 2>&1 | grep -v "$F" | tail -25 | tee gpurun_out/r5_e2e_cli_10m.log
