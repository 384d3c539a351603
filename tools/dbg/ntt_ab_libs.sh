for rep in 1 2; do for v in old new; do echo "== $v"; MP2G_LIB=$GRAFT_REPO_ROOT/build_dbg/libmp2gpu_$v.so python3 $GRAFT_REPO_ROOT/tools/dbg/ntt_batched.py; done; done
