cd $GRAFT_REPO_ROOT
for c in 2 3 4; do
timeout 900 python bench.py --config $c --steps 2 --warmup 1 --extract --cpu-seconds 0 --ingest-pairs 0 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); st=d['strain_extract']['stages']
print($c, d['strain_extract_s'], {k:v for k,v in st['sections'].items() if k.startswith('rb.') or k.startswith('reinit')}, st['reinit_calls']-st['reinit_reused'], 'writer busy', st['file_writer_busy_s'])"
done
