R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
UNET_FWD_REPACK=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_fwd_repack -o f -- python3 $R/profiles/forward_only.py 128 > $R/gpurun_out/prof_fwd_repack.log 2>&1
tail -1 $R/gpurun_out/prof_fwd_repack.log
python3 - $R <<'PY'
import csv,sys,collections
R=sys.argv[1]
rows=list(csv.DictReader(open(R+'/gpurun_out/prof_fwd_repack/f_kernel_stats.csv')))
for r in rows[:14]: print(r['Name'][:90], r['Calls'], r['AverageNs'], r['Percentage'])
PY
cd $R
UNET_FWD_REPACK=1 python profiles/forward_only.py 128 2>/dev/null | tail -1
UNET_FWD_REPACK=1 python profiles/forward_only.py 128 2>/dev/null | tail -1
