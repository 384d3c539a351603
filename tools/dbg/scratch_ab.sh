#!/bin/bash
# round 5: the provers' shared scratch (csrc/ctx.h). Table build, one block of 5120 rows, side legs off: the same 4 x 48 proofs in flight
# with buffers per prover (MP2G_SHARE_SCRATCH=0) and shared, then what the freed memory buys: wider batches, and the full batch in the
# reference-equivalent regime (--pad-base-bits 13 / 14, where round 4 had to halve the proofs in flight per degree)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05; mkdir -p $O
QUIET="--no-leaves-leg --no-verify --config2-leaves 0 --degree-sweep= --no-cpu-baseline"
one() { python3 $R/bench.py --steps 5 --warmup 2 --rows 1024 $QUIET "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c=d['config']; print('   ', round(d['value'],1), 'proofs/s  used', round(c['device_memory_used_bytes']/1e9,1), 'GB  planned', round(c['device_memory_planned_bytes']/1e9,1), 'GB')"; }
{
echo "4 x 48, MP2G_SHARE_SCRATCH=0"; MP2G_SHARE_SCRATCH=0 one --workers 4 --table-batch 48
for cfg in "4 48" "4 64" "4 96" "4 128" "3 128" "6 64"; do set -- $cfg; echo "workers $1 batch $2 (shared scratch)"; one --workers $1 --table-batch $2; done
echo "k = 13, 4 x 16, MP2G_SHARE_SCRATCH=0"; MP2G_SHARE_SCRATCH=0 python3 $R/bench.py --pad-base-bits 13 --steps 1 --warmup 1 --rows 1024 --table-batch 16 $QUIET 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('   ', round(d['value'],1), 'proofs/s', round(d['config']['device_memory_used_bytes']/1e9,1), 'GB')"
for b in 48 96; do echo "k = 13, 4 x $b (shared scratch)"; python3 $R/bench.py --pad-base-bits 13 --steps 1 --warmup 1 --rows 1024 --table-batch $b $QUIET 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('   ', round(d['value'],1), 'proofs/s', round(d['config']['device_memory_used_bytes']/1e9,1), 'GB')"; done
} 2>&1 | tee $O/scratch_ab.txt
