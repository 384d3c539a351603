# bench.py --workload recursion over --trees / --batch ("T:B" pairs as arguments)
for tb in "$@"; do t=${tb%%:*}; b=${tb##*:}
  python3 $GRAFT_REPO_ROOT/bench.py --workload recursion --batch $b --trees $t --steps 2 --warmup 1 2>/dev/null | tail -1 > /tmp/rt.json
  python3 -c "import json; d=json.load(open('/tmp/rt.json')); print('trees $t batch $b:', round(d['value'],1), 'leaf proofs/s', round(d['framework_proofs_per_s'],1), 'framework proofs/s', round(d['ms_per_step']), 'ms/step')"
done
