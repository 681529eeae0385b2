# PMC passes over one bench.py entry: bash tools/pmc_one.sh reuse:bf16_m64_blocked [outdir]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ONLY=${1:-reuse:bf16_m64_blocked}
O=$R/gpurun_out/${2:-pmc_one}
rm -rf $O; mkdir -p $O
B="python $R/bench.py --no-cpu-baseline --eager --min-seconds 0.002 --only $ONLY"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/p1 -- $B > $O/p1.out 2> $O/p1.err
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/p2 -- $B > $O/p2.out 2> $O/p2.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/p3 -- $B > $O/p3.out 2> $O/p3.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/p4 -- $B > $O/p4.out 2> $O/p4.err
find $O -name "*agent_info*" -delete
python3 - <<PY
import pandas as pd, glob
for p in ['p1','p2','p3','p4']:
    fs=glob.glob('$O/'+p+'/*/*_counter_collection.csv')
    if not fs: print(p,'no data'); continue
    d=pd.read_csv(fs[0])
    d=d[~d.Kernel_Name.str.contains('at::|elementwise|Memset|memcpy', regex=True)]
    d['k']=d.Kernel_Name.str.slice(0,60)
    g=d.groupby(['k','Counter_Name']).Counter_Value.mean().unstack()
    print(g.to_string())
PY
