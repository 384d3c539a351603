# 2^22 NTT kernel durations for alternative builds of the library (arguments: paths relative to the repo root)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  rm -rf /tmp/tr; ( export MP2G_LIB=$R/$lib; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -- python3 $R/tools/dbg/traffic_run.py > /dev/null 2>&1 )
  echo "== $lib"; grep -E "ntt_(rows|cols)" /tmp/tr/*/*_kernel_stats.csv | grep -v nat | sed "s/(mp2g::NttArgs[^\"]*\"//" | cut -d, -f1-4
done
