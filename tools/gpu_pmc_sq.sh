#!/bin/bash
# SQ counter passes (own runs, kernel-trace only, batch calls only: --no-extras) on one 64-frame sub-batch: wave-instructions
# per kernel and frame, VALU-busy, waits -> gpurun_out/pmcsq/sq_summary.json (copied to profiles/ by hand) and a table on stdout
export TMPDIR=/tmp
cd /root/repo
OUT=gpurun_out/pmcsq
B=${SQ_BATCH:-64}
rm -rf $OUT; mkdir -p $OUT
timeout 300 python -c "import bench; bench.make_frames(bench.shard_seeds(0, 1, 16))" > /dev/null 2>&1
FID_SUB_FRAMES=$B timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU --output-format csv -d $OUT/a -o p -- python bench.py --in-flight 1 --batch $B --unique 16 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/a.log 2>&1
FID_SUB_FRAMES=$B timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/b -o p -- python bench.py --in-flight 1 --batch $B --unique 16 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/b.log 2>&1
# pass c (round 5): lane work under the wave-instructions -- SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU) of the SAME pass
FID_SUB_FRAMES=$B timeout 300 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/c -o p -- python bench.py --in-flight 1 --batch $B --unique 16 --steps 2 --warmup 1 --no-cpu-baseline --no-extras > $OUT/c.log 2>&1
python tools/sq_summary.py $B $(find $OUT/a -name '*counter_collection.csv' | head -1) $(find $OUT/b -name '*counter_collection.csv' | head -1) $(find $OUT/c -name '*counter_collection.csv' | head -1)
mkdir -p gpurun_out/sq; cp $OUT/sq_summary.json gpurun_out/sq/sq_${FID_TRACE:-cycles}.json  # (kept: the next run clears $OUT)
tail -1 $OUT/a.log | cut -c1-200
find $OUT -name '*kernel_trace.csv' -delete; find $OUT -name '*.db' -delete
