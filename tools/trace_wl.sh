# kernel durations of one (or several ";;"-separated) workload expression(s) of tools/time_one.py under rocprofv3 --kernel-trace:
#   WL='bp.brgemm(api, 32, "f32", 1, br=4096)' bash tools/trace_wl.sh [outdir]      -> gpurun_out/<outdir>/kernels.txt (mean / min us per kernel name)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-trace_wl}
rm -rf $O; mkdir -p $O
export PYTHONPATH=$R
cd $R
rocprofv3 --kernel-trace --output-format csv -d $O/t -- python $R/tools/time_one.py > $O/run.out 2> $O/run.err
find $O -name "*agent_info*" -delete
python3 - <<PY | tee $O/kernels.txt
import pandas as pd, glob
fs = glob.glob('$O/t/*/*_kernel_trace.csv')
if not fs:
    print('no trace'); print(open('$O/run.err').read()[-800:]); raise SystemExit
d = pd.read_csv(fs[0])
d = d[~d.Kernel_Name.str.contains('at::|elementwise|Memset|memcpy|distribution|fill', regex=True)]
d['us'] = (d.End_Timestamp - d.Start_Timestamp) / 1e3
d['k'] = d.Kernel_Name.str.slice(0, 70)
g = d.groupby('k').us.agg(['count', 'mean', 'median', 'min'])
print(g.round(2).to_string())
PY
cat $O/run.out | tail -5
