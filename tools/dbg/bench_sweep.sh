for cfg in "128 2" "128 4" "256 4"; do set -- $cfg
  echo "== batch=$1 streams=$2"; timeout 200 python bench.py --batch $1 --streams $2 --steps 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
