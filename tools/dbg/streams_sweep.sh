for st in 2 4 6 8; do
  python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --streams $st --no-cpu-baseline --no-verify 2>/dev/null | tail -1 > /tmp/st.json
  python3 -c "import json; d=json.load(open('/tmp/st.json')); print('streams $st:', round(d['value'],1))"
done
