# memory phases vs butterflies of the 2^22 NTT: the product library, a build without the butterflies (NTT_DBG=1) and one
# without the global traffic (NTT_DBG=2); rocprofv3 kernel trace gives the per-kernel durations.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "" dbg1 dbg2; do
  lib=$R/mapreduce-plonky2_amd/libmp2gpu.so; [ -n "$v" ] && lib=$R/build_dbg/libmp2gpu_$v.so
  export MP2G_LIB=$lib
  rm -rf /tmp/ph_$v; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ph_$v -- python3 $R/tools/dbg/traffic_run.py > /dev/null 2>&1
  echo "== ${v:-product}"; grep "ntt_" /tmp/ph_$v/*/*_kernel_stats.csv | cut -d, -f1-5
done
