for v in 0_4 0_3 1_4 1_3 1_2; do
  echo "== NTT_PIPE_MINW=$v"
  MP2G_LIB=$PWD/mapreduce-plonky2_amd/libmp2gpu_ntt_$v.so timeout 120 python tools/dbg/ntt_only.py 2>&1 | grep -v amdgpu
done
