#!/bin/bash
R=$GRAFT_REPO_ROOT
for v in ubench_b ubench_bdgl_split_tail_nop ubench_oldtail ubench_b ubench_bdgl_split_tail_nop ubench_oldtail; do
  echo "== $v"; $R/tools/ubench/$v 2>&1 | grep -E "mismatch|mulw\(32chain\)|poseidon2"
done
