export VS_EXPERIMENT=timing  # the switches below exist only in experiment mode (VsTuning)
cd "$GRAFT_REPO_ROOT"
P='import sys,json; d=json.loads(sys.stdin.read()); r=d["roofline"]; print("acc %.3f step %.3f" % (r["accumulate_ms_avg"], d["ms_per_step"]))'
for e in "VS_ACC_CHECK=0" "VS_ACC_CHECK=0 VS_ACC_FILL=50" "VS_ACC_CHECK=4 VS_ACC_FILL=50" "VS_ACC_CHECK=8 VS_ACC_FILL=50" "VS_ACC_CHECK=4 VS_ACC_FILL=80" "VS_ACC_CHECK=16 VS_ACC_FILL=60" "VS_ACC_CHECK=0 VS_ACC_MERGE=1" "VS_ACC_CHECK=0 VS_ACC_GRID_PER_CU=8"; do
  echo "== $e"; env $e VS_DEBUG_ACC=1 timeout 600 python bench.py --config 4 --steps 2 --warmup 1 --cpu-seconds 0 --no-extract 2>/tmp/err.txt | python -c "$P"; grep "k_pe_accumulate:" /tmp/err.txt | tail -1
done
