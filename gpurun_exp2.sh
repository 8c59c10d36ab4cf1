cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -x -q -m gpu --timeout 400 2>&1 | grep -E "passed|failed|^E " | head -5
for i in 1 2; do timeout 200 python bench.py --cpu-frames 0 2>&1 | grep '"metric"' | cut -c1-170; done
