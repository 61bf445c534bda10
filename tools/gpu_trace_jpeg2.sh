export TMPDIR=/tmp
cd /root/repo
rm -rf gpurun_out/profj; timeout 400 rocprofv3 --kernel-trace --stats -d gpurun_out/profj -o r -- python tools/gpu_jpeg_bench.py 256 80 > gpurun_out/profj.log 2>&1
python tools/rocpd_stats.py $(find gpurun_out/profj -name '*.db' | head -1) > gpurun_out/jpeg_kernel_stats.csv
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/jpeg_kernel_stats.csv')):
    print(r['Name'][:60].ljust(60), r['Calls'], 'avg', r['AverageNs'], 'max', r['MaxNs'], r['Percentage'])
PY
rm -rf gpurun_out/profj
