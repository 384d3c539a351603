# HBM traffic of the 2^22 NTT launches from the TCC counters, calibrated on a kernel of the same
# access width (8 B/lane coalesced) with a known byte count: scale_powers_kernel reads and writes
# batch*n*8 bytes exactly once (MI355X_MICROARCH.md "HBM": FETCH_SIZE is uncalibrated for widths
# other than 16 B/lane, so calibrate in your own pattern; separate --pmc passes).
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/traffic_$c -- python3 $GRAFT_REPO_ROOT/tools/dbg/traffic_run.py > /dev/null 2>&1
done
ls $GRAFT_REPO_ROOT/gpurun_out/traffic_*/*/
