for B in 2 16 64; do
  echo "== B=$B"; timeout 120 python -X faulthandler bench.py --steps 1 --warmup 1 --batch $B --no-cpu-baseline 2>&1 | tail -3 | cut -c1-400
done
