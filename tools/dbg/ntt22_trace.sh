# per-kernel durations of the 2^22 NTT under env settings given as arguments ("A=1 B=2" per argument; "" = default)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for envs in "$@"; do
  rm -rf /tmp/tr; ( export $envs MP2G_DUMMY=1; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -- python3 $R/tools/dbg/traffic_run.py > /dev/null 2>&1 )
  echo "== ${envs:-default}"; grep -E "ntt_(rows|cols)" /tmp/tr/*/*_kernel_stats.csv | grep -v nat | sed "s/(mp2g::NttArgs[^\"]*\"//" | cut -d, -f1-4
done
