#!/bin/bash
# GPU-side measurement campaigns, one script with subcommands (run on the GPU box through gpurun; outputs under
# gpurun_out/, copy what is to be judged into profiles/r<N>/):
#
#   tools/campaign.sh sweep  "<configs>" "<env 1>" "<env 2>" ...   per-kernel timing line of bench.py per config and environment
#                                                                  (tuning switches need VS_EXPERIMENT=1, which this sets), e.g.
#                                                                  tools/campaign.sh sweep "2 4" "X=0" "VS_GRID_PER_CU=64" "VS_NO_XCD_MAP=1"
#   tools/campaign.sh profiles "<configs>" [TAG]                   rocprofv3 kernel stats + HBM / L2 counters per config
#                                                                  (tools/profile.sh), SQ / TCP counters for configs 2 and 3
#                                                                  (tools/pmc.sh), summaries kept as gpurun_out/prof_<TAG>_c<i>/
#   tools/campaign.sh bench  "<configs>" [TAG] [bench args]        untraced bench lines -> gpurun_out/<TAG>_bench_config<i>.json
#   tools/campaign.sh extract "<configs>" [TAG]                    bench lines with the strain-extract leg (--extract)
#   tools/campaign.sh fuzz   [T_PE T_STD T_GRAPH]                  the randomized campaigns at length (tests/fuzz_pe.py, fuzz_graph.py)
#   tools/campaign.sh suite  [TAG]                                 pytest -m gpu, tail of the log -> gpurun_out/<TAG>_gpu_tests.log
#   tools/campaign.sh final  [TAG]                                 suite + fuzz + profiles of all five configs + bench / extract lines
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; cd "$R"; mkdir -p gpurun_out
cmd="${1:-final}"; shift
LINE='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]; print("tiles %.3f counters %.3f sort %.3f overflow %.3f step %.3f  extract %s" % (r["kernel_ms_avg"], r["accumulate_ms_avg"], r["locus_sort_ms_avg"], r["slow_kernel_ms_avg"], d["ms_per_step"], d.get("strain_extract_s")))'
case "$cmd" in
  sweep)
    configs="$1"; shift
    export VS_EXPERIMENT=1
    for c in $configs; do
      for e in "$@"; do
        echo -n "config $c [$e]: "
        env $e timeout 900 python bench.py --config $c --steps 10 --warmup 2 --cpu-seconds 0 --ingest-pairs 0 --no-extract 2>/dev/null | python -c "$LINE"
      done
    done 2>&1 | tee -a gpurun_out/sweep.log ;;
  profiles)
    configs="${1:-2 1 3 4 0}"; tag="${2:-r5}"
    for c in $configs; do
      bash tools/profile.sh ${tag}_c$c $c --no-extract > gpurun_out/prof_${tag}_c$c.log 2>&1; tail -2 gpurun_out/prof_${tag}_c$c.log
    done
    for c in 2 3; do case " $configs " in *" $c "*) CFG=$c bash tools/pmc.sh ${tag}_c$c > gpurun_out/pmc_${tag}_c$c.log 2>&1; tail -4 gpurun_out/pmc_${tag}_c$c.log;; esac; done
    cd gpurun_out
    for c in $configs; do  # keep the summaries, drop the raw traces (gpurun_out is capped at 64 MiB)
      d=prof_${tag}_c$c; mkdir -p keep_$d
      cp $d/pmc_summary.json $d/trace_bench.json keep_$d/ 2>/dev/null
      find $d/trace -name "*kernel_stats.csv" -exec cp {} keep_$d/kernel_stats.csv \;
      rm -rf $d; mv keep_$d $d
    done
    for c in 2 3; do [ -d pmc_${tag}_c$c ] && { mkdir -p keep_pmc; cp pmc_${tag}_c$c/summary.json keep_pmc/ 2>/dev/null; rm -rf pmc_${tag}_c$c; mv keep_pmc pmc_${tag}_c$c; }; done
    cd "$R" ;;
  bench)
    configs="${1:-2 1 0 3 4}"; tag="${2:-r5}"; shift; shift
    for c in $configs; do
      steps=20; [ $c -ge 3 ] && steps=10
      timeout 1500 python bench.py --config $c --steps $steps --warmup 2 "$@" > gpurun_out/${tag}_bench_config${c}.json 2> gpurun_out/${tag}_bench_config${c}.err
      tail -c 300 gpurun_out/${tag}_bench_config${c}.err; python -c "$LINE" < gpurun_out/${tag}_bench_config${c}.json
    done ;;
  extract)
    configs="${1:-2 3 4}"; tag="${2:-r5}"
    for c in $configs; do
      timeout 1500 python bench.py --config $c --steps 3 --warmup 1 --extract --cpu-seconds 0 --ingest-pairs 0 > gpurun_out/${tag}_bench_config${c}_with_extract.json 2> gpurun_out/${tag}_bench_config${c}_with_extract.err
      python -c "$LINE" < gpurun_out/${tag}_bench_config${c}_with_extract.json
    done ;;
  fuzz)
    python tests/fuzz_pe.py ${1:-600} 41 2>&1 | tail -3 | tee gpurun_out/fuzz_pe_boundaries.log
    FUZZ_STD=1 python tests/fuzz_pe.py ${2:-200} 42 2>&1 | tail -3 | tee gpurun_out/fuzz_pe_std.log
    python tests/fuzz_graph.py ${3:-400} 44 2>&1 | tail -3 | tee gpurun_out/fuzz_graph.log ;;
  suite)
    tag="${1:-r5}"
    python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP ver\|^ROCm\|^Host\|^Librccl" | tail -30 > gpurun_out/${tag}_gpu_tests.log; tail -3 gpurun_out/${tag}_gpu_tests.log ;;
  final)
    tag="${1:-r5}"
    bash "$0" suite $tag; bash "$0" fuzz 420 200 300; bash "$0" profiles "2 1 3 4 0" $tag; bash "$0" bench "2 1 0 3 4" $tag; bash "$0" extract "2 3 4" $tag ;;
  *) echo "unknown subcommand $cmd"; exit 2 ;;
esac
