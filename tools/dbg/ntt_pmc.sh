cd /tmp && export TMPDIR=/tmp

timeout 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc1 -- python3 $GRAFT_REPO_ROOT/tools/dbg/ntt_only.py > /dev/null 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc2 -- python3 $GRAFT_REPO_ROOT/tools/dbg/ntt_only.py > /dev/null 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/pmc1/*/ $GRAFT_REPO_ROOT/gpurun_out/pmc2/*/
