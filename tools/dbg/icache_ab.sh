#!/bin/bash
# Does the size of the sponge kernels' text matter inside a proving step? The leaf kernel holds two copies of the Poseidon2
# permutation (full chunks, last chunk) of 46 KB each; the instruction cache two CUs share is 64 KB, and inside a step the
# kernels of the other streams compete for it. Variants of merkle.hip (leaf sponge + tree levels), each measured alone
# (tools/dbg/commit_only.py: 135 x 2^17 commit) and inside the table build (4 workers, one block of 5120 rows):
#   base            the product as built
#   onecopy         -DLEAF_ONE_COPY                     one copy of the permutation in the leaf kernel (49 KB)
#   onecopy_u2      -DLEAF_ONE_COPY -DP2_UNROLL_INT=2   + internal rounds unrolled by 2 instead of 11 (27 KB)
#   u2              -DP2_UNROLL_INT=2                   two copies of the short form (53 KB)
#   onecopy_u2_all  the same flags on fri.hip too (the proof-of-work search, the FRI layers' leaf sponge)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05; mkdir -p $O
QUIET="--no-leaves-leg --no-verify --config2-leaves 0 --degree-sweep= --no-cpu-baseline"
run() {  # name
  echo "== $1 (MP2G_LIB=${MP2G_LIB:-product})"
  python3 $R/tools/dbg/commit_only.py | sort -t: -k2 -n | sed -n 6p
  for rep in 1 2; do
    python3 $R/bench.py --steps 5 --warmup 2 --rows 1024 $QUIET 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('   table 4 x 32:', round(d['value'],1), 'proofs/s')"
  done
}
{
unset MP2G_LIB; run base
export MP2G_LIB=$($R/tools/dbg/build_variant.sh onecopy "-DLEAF_ONE_COPY" merkle.hip | tail -1); run onecopy
export MP2G_LIB=$($R/tools/dbg/build_variant.sh onecopy_u2 "-DLEAF_ONE_COPY -DP2_UNROLL_INT=2" merkle.hip | tail -1); run onecopy_u2
export MP2G_LIB=$($R/tools/dbg/build_variant.sh u2 "-DP2_UNROLL_INT=2" merkle.hip | tail -1); run u2
export MP2G_LIB=$($R/tools/dbg/build_variant.sh onecopy_u2_all "-DLEAF_ONE_COPY -DP2_UNROLL_INT=2" merkle.hip fri.hip | tail -1); run onecopy_u2_all   # + the PoW search and the FRI-layer leaves
unset MP2G_LIB; run base_again
} 2>&1 | tee $O/icache_ab.txt
