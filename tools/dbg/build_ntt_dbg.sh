# debug builds of libmp2gpu.so for tools/dbg/ntt_phases.sh: NTT_DBG=1 (no butterflies), NTT_DBG=2 (no global traffic)
cd "$(dirname "$0")/../../mapreduce-plonky2_amd/csrc" && mkdir -p ../../build_dbg && for d in 1 2; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -DNTT_DBG=$d -c ntt.hip -o ../../build_dbg/ntt_dbg$d.o &&
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o ../../build_dbg/libmp2gpu_dbg$d.so ../../build_dbg/ntt_dbg$d.o $(ls *.o | grep -v '^ntt.o')
done
