set -e
cd /root/repo; mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_backward_gpu.py tests/test_train_gpu.py tests/test_retriever_gpu.py -x -q 2>&1 | tail -5
for p in 32-true bf16-mixed; do EVI_PROFILE_PRECISION=$p timeout -k 10 120 python tools/scorer_forward_profile.py train; done
cd /tmp && export TMPDIR=/tmp
EVI_PROFILE_PRECISION=bf16-mixed timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_bf16 -o t -- python3 /root/repo/tools/scorer_forward_profile.py train > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('/tmp/prof_bf16/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
import shutil; shutil.copy(f,'/root/repo/gpurun_out/r02_train_step_bf16_mixed_kernel_stats.csv')
for r in rows[:14]: print(r['Name'][:60].ljust(60), r['Calls'], r['AverageNs'], r['Percentage'])
PY
