cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_pe_gpu.py tests/test_configs_gpu.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r5_pe_tests_1.log; tail -5 gpurun_out/r5_pe_tests_1.log
bash tools/campaign.sh sweep "2 3 4" "X=0" "VS_NO_STD=1" "VS_PHASE0=1"
for c in 2 3 4; do for e in X=0 VS_PHASE0=1; do echo "postings config $c [$e]"; env VS_EXPERIMENT=1 VS_DEBUG_POSTINGS=1 $e timeout 600 python bench.py --config $c --steps 1 --warmup 0 --cpu-seconds 0 --ingest-pairs 0 --no-extract 2>&1 | grep "postings expanded" | tail -1; done; done 2>&1 | tee gpurun_out/r5_postings_phase.log
