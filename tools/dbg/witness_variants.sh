#!/bin/bash
# A/B of the witness kernel's footprint (lanes per proof block, VGPR cap) on the table workload: builds variant libraries on the box
R=$GRAFT_REPO_ROOT
cd $R/mapreduce-plonky2_amd/csrc
for v in "512 512" "512 1024" "256 256" "256 1024" "1024 1024"; do
  set -- $v
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -I../../include -DWIT_LANES_N=$1 -DWIT_BOUNDS=$2 -c witness_dev.hip -o witness_dev.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libmp2gpu.so *.o
  cd $R
  python tools/dbg/witness_dev_timing.py 2>/dev/null | grep "B=32"
  python bench.py --steps 2 --warmup 1 --no-leaves-leg --no-verify 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('lanes $1 bounds $2:', round(d['value'],1), 'proofs/s')"
  cd $R/mapreduce-plonky2_amd/csrc
done
