#!/bin/bash
# the multiset digest kernels (ecgfp5.hip) under declared launch bounds (which reach the out-of-line bodies), register caps on the kernels, without the dedicated GF(p^5) squaring, and with the plain SWU (a square
# root to learn which candidate is square, a second one inside Point::decode): variant libraries beside the
# product (MP2G_LIB), 2^20 rows x 5 columns (4 value columns + the secondary index: what bench.py times as table_digest_2p20_rows_ms
# has 5 columns; tools/dbg/ec_bench.py takes 4), correctness of every variant by the Ecgfp5 tests
R=$GRAFT_REPO_ROOT
cd $R
bash tools/dbg/build_variant.sh ec_nosqr "-DEC_NO_SQR" ecgfp5.hip > /dev/null
bash tools/dbg/build_variant.sh ec_swu_plain "-DEC_SWU_PLAIN" ecgfp5.hip > /dev/null
bash tools/dbg/build_variant.sh ec_bitserial "-DEC_MUL_BITSERIAL" ecgfp5.hip > /dev/null
bash tools/dbg/build_variant.sh ec_r03 "-DEC_MUL_BITSERIAL -DEC_SWU_PLAIN -DEC_NO_SQR" ecgfp5.hip > /dev/null
bash tools/dbg/build_variant.sh ec_dbl_plain "-DEC_DBL_PLAIN" ecgfp5.hip > /dev/null
bash tools/dbg/build_variant.sh ec_lb768 "-DEC_LB=768" ecgfp5.hip > /dev/null
bash tools/dbg/build_variant.sh ec_lb1024 "-DEC_LB=1024" ecgfp5.hip > /dev/null
bash tools/dbg/build_variant.sh ec_w3 "-DEC_WAVES_ATTR=__attribute__((amdgpu_waves_per_eu(3,3)))" ecgfp5.hip > /dev/null
bash tools/dbg/build_variant.sh ec_w4 "-DEC_WAVES_ATTR=__attribute__((amdgpu_waves_per_eu(4,4)))" ecgfp5.hip > /dev/null
for v in ${EC_VARIANTS:-product ec_dbl_plain ec_lb768 ec_lb1024 ec_swu_plain ec_bitserial ec_r03 ec_nosqr ec_w3 ec_w4}; do
  if [ $v = product ]; then unset MP2G_LIB; else export MP2G_LIB=$R/build_dbg/$v/libmp2gpu.so; fi
  echo "== $v"
  python tools/dbg/kernel_resources.py ${MP2G_LIB:-} 2>/dev/null | grep -E "row_digest_kernel<0>|map_to_curve_kernel<0>|scalar_mul_kernel" | awk '{print "   ", $1, "vgpr", $2, "scratch", $(NF-1)}'
  python tools/dbg/ec_bench.py 20 2>&1 | tail -2 | head -1
  timeout 600 python -m pytest tests/test_gpu_ecgfp5.py -x -q -m gpu 2>&1 | tail -1
done
