ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/pmc_more; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
BENCH="python3 $ROOT/bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-profile --render-views 0"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d $OUT/p1 -- $BENCH > $OUT/p1.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_INSTS_VALU SQ_INSTS_SALU --output-format csv -d $OUT/p2 -- $BENCH > $OUT/p2.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/p3 -- $BENCH > $OUT/p3.log 2>&1
ls $OUT/*
