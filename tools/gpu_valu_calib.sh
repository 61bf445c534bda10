#!/bin/bash
# VALU / SALU / LDS issue-rate calibration (tools/valu_calib.hip) -> gpurun_out/valu_calib.json (copied to profiles/ by hand)
export TMPDIR=/tmp
cd /root/repo
mkdir -p gpurun_out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_calib tools/valu_calib.hip > gpurun_out/valu_calib_build.log 2>&1 || { cat gpurun_out/valu_calib_build.log; exit 1; }
timeout 120 /tmp/valu_calib ${1:-2048} $2 > gpurun_out/valu_calib.json || { echo "valu_calib failed"; exit 2; }
python - <<'PY'
import json
d = json.load(open('gpurun_out/valu_calib.json'))
print(d['device'], d['cus'], 'CUs', d['clock_mhz_reported'], 'MHz')
print('kind'.ljust(52), *[f'{w} w/SIMD'.rjust(30) for w in (1, 2, 4, 8)])
for k, v in d['kinds'].items():
    row = []
    for w in ('1', '2', '4', '8'):
        r = v['by_waves_per_simd'][w]
        mn, p90, mx = r['cyc_per_instr_wave_min_p90_max']
        row.append(f"{r['cyc_per_instr_wave']:.1f}[{mn:.1f}-{mx:.1f}] {r['chip_ginstr_s']:5.0f}G/s {r['ticks_per_us']:.0f}".rjust(30))
    print(k[:52].ljust(52), *row)
print('columns: ticks per instruction and wave, median [min - max]; chip wave-instructions/s; s_memtime ticks per us of wall time')
PY
