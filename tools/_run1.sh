cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2500 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|FAILED" | tail -3
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -c "'ok': True"
timeout 600 python bench.py 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['roofline']['rocprof']['frac'], {k:v.get('value') for k,v in d.get('extra',{}).items()} if isinstance(d.get('extra'),dict) else '')"
