for u in 0 1; do
  MP2G_GATES_UNFUSED=$u python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --no-cpu-baseline --no-verify 2>/dev/null | tail -1 > /tmp/g.json
  python3 -c "import json; d=json.load(open('/tmp/g.json')); print('unfused=$u:', round(d['value'],1), {k: v['quotient'] for k, v in d['stage_ms'].items()})"
done
