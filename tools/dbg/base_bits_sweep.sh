# leaf proofs/s for every base size the recursion framework can hand to a wrap (k in 12..15)
for cfg in "12 128" "13 128" "14 64" "15 32"; do set -- $cfg
  echo "== base_bits=$1 batch=$2"
  timeout 300 python bench.py --base-bits $1 --batch $2 --steps 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], 'leaf proofs/s', d['ms_per_step'], 'ms/step')"
done
