# PMC passes over one workload expression of tools/time_one.py: WL='bp.brgemm_mx4i8(api, 64, 2 ** 17)' bash tools/pmc_wl.sh [outdir]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-pmc_wl}
rm -rf $O; mkdir -p $O
export EAGER=1 PYTHONPATH=$R
B="python $R/tools/time_one.py"
cd $R
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/p1 -- $B > $O/p1.out 2> $O/p1.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $O/p2 -- $B > $O/p2.out 2> $O/p2.err
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES --kernel-trace --output-format csv -d $O/p3 -- $B > $O/p3.out 2> $O/p3.err
rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum --kernel-trace --output-format csv -d $O/p4 -- $B > $O/p4.out 2> $O/p4.err
find $O -name "*agent_info*" -delete
python3 - <<PY
import pandas as pd, glob
for p in ['p1','p2','p3','p4']:
    fs=glob.glob('$O/'+p+'/*/*_counter_collection.csv')
    if not fs: print(p,'no data'); print(open('$O/'+p+'.err').read()[-600:]); continue
    d=pd.read_csv(fs[0])
    d=d[~d.Kernel_Name.str.contains('at::|elementwise|Memset|memcpy|distribution|fill', regex=True)]
    d['k']=d.Kernel_Name.str.slice(0,50)
    g=d.groupby(['k','Counter_Name']).Counter_Value.mean().unstack()
    print(g.T.to_string())
PY
