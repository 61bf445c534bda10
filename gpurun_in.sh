cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r6long
for seed in 101 202 303; do timeout 900 python tools/gpu_stag_ab_libs.py tmp_ab/libfid_r5.so fiducials_amd/lib/libfid_amd.so 500 $seed 2>&1 | tail -1 | tee -a gpurun_out/r6long/stag_ab.log; done
timeout 600 python tools/gpu_stag_spec_stress.py 600 9 2>&1 | tail -1 | tee gpurun_out/r6long/stag_spec.log
timeout 900 python tools/gpu_stress.py 600 2>&1 | tail -2 | tee gpurun_out/r6long/aruco_stress.log
