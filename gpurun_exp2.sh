cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for o in "--surfels 4200000" "--surfels 8400000" "--surfels 4200000 --virtual-shards 4"; do
echo "== $o"; timeout 600 python bench.py --cpu-frames 0 --steps 40 --warmup 10 $o 2>/dev/null | tail -1 | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config']['parallelism'], d['config']['surfels_end'], d['config']['final_translation_error_mm'], d['roofline']['frac'], d['roofline']['avg_kernel_ms'], d['config']['last_frame_region_ms'])"
done
