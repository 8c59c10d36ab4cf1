cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests -x -q -m gpu --timeout 400 2>&1 | grep -E "passed|failed|^E " | head -5
timeout 200 python bench.py --cpu-frames 0 2>&1 | grep '"metric"' | python -c "
import sys, json
d=json.loads(sys.stdin.readline()); print(d['value'], d['config']['pcie_inclusive_fps'], d['roofline']['frac'])"
