cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export PYTHONPATH=$R
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_paths2 -- python $R/tools/bench_paths.py --only lowbit,bitmask,bcsc --eager 20 > $R/gpurun_out/prof_paths2.jsonl 2>&1
cd $R
python - <<'PY'
import glob, pandas as pd
f = glob.glob("gpurun_out/prof_paths2/*/*kernel_stats.csv")[0]
d = pd.read_csv(f)
d = d[d.Name.str.contains("xamd")]
d.to_csv("gpurun_out/r03_paths2_kernel_stats.csv", index=False)
print(d[["Name","Calls","AverageNs"]].to_string()[:3000])
PY
