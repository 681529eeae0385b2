cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_bm2
rm -rf $O; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -oE "\b(TA|TCP|TCC|SQ)_[A-Z0-9_a-z]+" | sort -u > $O/counters.txt
export EAGER=1 PYTHONPATH=$R LIBXSMM_HIP_BITMASK_FORM=1 WL='bp.bitmask_gemm(api, 8192, 16, 8192, 0.5)'
cd $R
B="python $R/tools/time_one.py"
rocprofv3 --pmc TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum --kernel-trace --output-format csv -d $O/p1 -- $B > $O/p1.out 2> $O/p1.err
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_UTCL1_TRANSLATION_MISS_sum --kernel-trace --output-format csv -d $O/p2 -- $B > $O/p2.out 2> $O/p2.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $O/p3 -- $B > $O/p3.out 2> $O/p3.err
find $O -name "*agent_info*" -delete
python3 - <<PY
import pandas as pd, glob
for p in ['p1','p2','p3']:
    fs=glob.glob('$O/'+p+'/*/*_counter_collection.csv')
    if not fs: print(p,'no data'); print(open('$O/'+p+'.err').read()[-800:]); continue
    d=pd.read_csv(fs[0])
    d=d[d.Kernel_Name.str.contains('gemm_bitmask16')]
    g=d.groupby(['Counter_Name']).Counter_Value.mean()
    print(g.to_string())
PY
grep -c . $O/counters.txt
