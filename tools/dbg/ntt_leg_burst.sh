for k in 20 200 1000; do
  python3 $GRAFT_REPO_ROOT/bench.py --workload ntt --steps $k --warmup 5 2>/dev/null | tail -1 > /tmp/nk.json
  python3 -c "import json; d=json.load(open('/tmp/nk.json')); print('$k', round(d['roofline']['launch_ms']*1e3,1), round(d['roofline']['frac'],4), d['clocks'].get('sclk clock speed:'))"
done
