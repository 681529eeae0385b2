cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_ragged
rm -rf $O; mkdir -p $O
B="python $R/tools/bench_paths.py --only ragged --eager 4"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p1 -- $B > $O/p1.out 2> $O/p1.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/p2 -- $B > $O/p2.out 2> $O/p2.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p3 -- $B > $O/p3.out 2> $O/p3.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/p4 -- $B > $O/p4.out 2> $O/p4.err
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/p5 -- $B > $O/p5.out 2> $O/p5.err
find $O -name "*.csv" | head -20; tail -2 $O/p*.err | tail -20
# keep only csv small: drop big agent info
find $O -name "*agent_info*" -delete
du -sh $O
