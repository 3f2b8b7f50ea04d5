cd $GRAFT_REPO_ROOT
timeout 2400 python tools/scaling_model.py --config 2 --out gpurun_out/scaling_model_config2.json 2>&1 | tail -45
