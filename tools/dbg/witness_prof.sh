#!/bin/bash
# per-class timing of the witness kernel's levels: a library with -DWIT_PROF built on the box, then tools/dbg/witness_prof.py
R=$GRAFT_REPO_ROOT
cd $R/mapreduce-plonky2_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -I../../include -DWIT_PROF -c witness_dev.hip -o witness_dev.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../libmp2gpu.so *.o
cd $R && python tools/dbg/witness_prof.py
