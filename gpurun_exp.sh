timeout 600 python -m pytest tests -x -q -m gpu -k "tracked or stage or variant or edge" 2>&1 | grep -E "^>|^E|passed|failed" | head
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_x -o x -- python bench.py --steps 30 --warmup 10 --cpu-frames 0 > gpurun_out/x.log 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_x/x_kernel_stats.csv')))
for r in rows:
    if any(k in r['Name'] for k in ('clean_flags','fuse_stream')):
        print("%-28s calls %5s avg %9.1f us min %8.1f max %8.1f" % (r['Name'][:28], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
grep "^{" gpurun_out/x.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fps %.1f fuse_ms %.4f frac %.3f'%(d['value'], d['roofline']['avg_kernel_ms'], d['roofline']['frac']))"
