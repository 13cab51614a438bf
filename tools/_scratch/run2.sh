set -e
mkdir -p /root/repo/gpurun_out
cd /tmp && export TMPDIR=/tmp
EVI_PROFILE_PRECISION=bf16-mixed timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bf16 -o t -- python3 /root/repo/tools/scorer_forward_profile.py train > /tmp/prof.log 2>&1 || { tail -5 /tmp/prof.log; exit 1; }
python3 - <<'PY'
import csv,glob,shutil
f=glob.glob('/tmp/prof_bf16/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
shutil.copy(f,'/root/repo/gpurun_out/r02_train_step_bf16_mixed_kernel_stats.csv')
for r in rows[:16]: print(r['Name'][:60].ljust(60), r['Calls'], r['AverageNs'], r['Percentage'])
PY
