# latency of small batches: leaf proofs (base 2^13 + wrap 2^12) one stream, B = 1, 2, 4, 8
for b in 1 2 4 8; do
  python3 $GRAFT_REPO_ROOT/bench.py --workload leaves --batch $b --streams 1 --steps 20 --warmup 2 --no-cpu-baseline --no-verify 2>/dev/null | tail -1 > /tmp/lp.json
  python3 -c "import json; d=json.load(open('/tmp/lp.json')); print('B=$b:', round(d['ms_per_step'],2), 'ms per step =', round(d['value'],1), 'leaf proofs/s;', {k: round(sum(v.values()),2) for k, v in d['stage_ms'].items()}, d['stage_ms'])"
done
