# PMC passes over one group of tools/bench_paths.py: bash tools/pmc_paths.sh bcsc [outdir]   (per launch, summed over the 8 XCDs; SQ_* cycle counters are in 4-cycle units)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ONLY=${1:-bcsc}
O=$R/gpurun_out/${2:-pmc_paths}
rm -rf $O; mkdir -p $O
B="python $R/tools/bench_paths.py --only $ONLY --eager 3"   # --eager: a few plain launches per workload (the timed replay loop under counter collection takes minutes)
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/p1 -- $B > $O/p1.out 2> $O/p1.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/p2 -- $B > $O/p2.out 2> $O/p2.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $O/p3 -- $B > $O/p3.out 2> $O/p3.err
rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace --output-format csv -d $O/p4 -- $B > $O/p4.out 2> $O/p4.err
find $O -name "*agent_info*" -delete
python3 - <<PY
import pandas as pd, glob
for p in ['p1','p2','p3','p4']:
    fs=glob.glob('$O/'+p+'/*/*_counter_collection.csv')
    if not fs: print(p,'no data'); continue
    d=pd.read_csv(fs[0])
    d=d[~d.Kernel_Name.str.contains('at::|elementwise|Memset|memcpy|invert', regex=True)]
    d['k']=d.Kernel_Name.str.replace('void xamd::','').str.slice(0,48)+' g'+d.Grid_Size.astype(str)
    g=d.groupby(['k','Counter_Name']).Counter_Value.mean().unstack()
    print(g.to_string())
PY
