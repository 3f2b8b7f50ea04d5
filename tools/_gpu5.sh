cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_pe_gpu.py -m gpu -x -q -k "occupied or two_ranks or sharded" 2>&1 | tail -4
timeout 2400 python tools/scaling_model.py --config 2 --out gpurun_out/scaling_model_config2.json 2>&1 | tail -34
