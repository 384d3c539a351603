#!/bin/bash
# per-class timing of the witness kernel's levels: a -DWIT_PROF variant of the library built beside the product
# (build_dbg/witprof/libmp2gpu.so, selected with MP2G_LIB; the product library is not touched), then tools/dbg/witness_prof.py
R=$GRAFT_REPO_ROOT
bash $R/tools/dbg/build_variant.sh witprof "-DWIT_PROF" witness_dev.hip > /dev/null
cd $R && MP2G_LIB=$R/build_dbg/witprof/libmp2gpu.so python tools/dbg/witness_prof.py
