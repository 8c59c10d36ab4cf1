cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_x -o x -- python bench.py --steps 20 --warmup 5 --cpu-frames 0 > gpurun_out/x.log 2>&1
tail -1 gpurun_out/x.log
python - <<'PY'
import csv, collections
tr=list(csv.DictReader(open('gpurun_out/prof_x/x_kernel_trace.csv')))
acc=collections.defaultdict(list)
for r in tr:
    n=r['Kernel_Name'].split('(')[0]
    acc[(n, r['Grid_Size_X'])].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
tot=sum(sum(v) for v in acc.values())
for k,v in sorted(acc.items(), key=lambda kv:-sum(kv[1]))[:40]:
    print(k, "n=%d avg=%.1f us min=%.1f share=%.1f%%"%(len(v), sum(v)/len(v), min(v), 100*sum(v)/tot))
PY
timeout 200 python bench.py 2>&1 | tail -1
