cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_x -o x -- python bench.py --steps 30 --warmup 10 --cpu-frames 0 > /dev/null 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_x/x_kernel_stats.csv')))
for r in rows:
    if any(k in r['Name'] for k in ('clean_flags','fuse_stream')):
        print("%-28s calls %5s avg %9.1f us min %8.1f max %8.1f" % (r['Name'][:28], r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
timeout 120 python bench.py --steps 60 --warmup 20 --cpu-frames 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fps %.1f fuse_ms %.4f frac %.3f'%(d['value'], d['roofline']['avg_kernel_ms'], d['roofline']['frac'])); print(d['config']['last_frame_region_ms'])"
timeout 400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
