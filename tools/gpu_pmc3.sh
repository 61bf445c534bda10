#!/bin/bash
# PMC traffic passes (FETCH_SIZE, WRITE_SIZE; own runs, kernel-trace only): first the calibration kernels with known byte
# counts (tools/pmc_calib.hip), then the pipeline on a 64-frame batch.  Result: gpurun_out/pmc3/pmc_traffic.json
export TMPDIR=/tmp
cd /root/repo
OUT=gpurun_out/pmc3
rm -rf $OUT; mkdir -p $OUT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $OUT/pmc_calib tools/pmc_calib.hip > $OUT/build.log 2>&1 || { cat $OUT/build.log; exit 1; }
B=${PMC_BATCH:-64}
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/calib_$c -o p -- $OUT/pmc_calib > $OUT/calib_$c.log 2>&1
  FID_SUB_FRAMES=$B timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o p -- python bench.py --batch $B --unique 16 --steps 1 --warmup 1 --no-cpu-baseline > $OUT/$c.log 2>&1
done
python tools/pmc_summary.py $OUT $B fiducials_amd/lib/libfid_amd.so > $OUT/pmc_traffic.json
python - <<'PY'
import json
d=json.load(open('gpurun_out/pmc3/pmc_traffic.json'))
print(json.dumps(d['calibration']['factor_true_over_counter']))
print(json.dumps(d['per_frame_bytes']), d['pipeline_bytes_per_frame'])
for k,v in d['kernels'].items(): print(k[:40].ljust(40), v)
PY
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*.db' -delete; rm -f $OUT/pmc_calib
