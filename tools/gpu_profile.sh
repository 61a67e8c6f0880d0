#!/bin/bash
# the profiles a round commits (profiles/README.md): kernel trace + PMC groups of the driver's bench command, bench lines, the frame
# loop's kernels, the band sweep.  usage: tools/gpu_profile.sh [tag] -> gpurun_out/prof_<tag>/
# (PMC passes: at most 4 TCC counters and 8 SQ counters per run; no trace domain combined with --pmc beyond --kernel-trace)
set -u
export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_${1:-r6}
rm -rf $OUT; mkdir -p $OUT
ARGS="--gpus 1 --steps 20 --warmup 5 --no-cpu --no-traffic"
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1
run_pmc () { name=$1; shift; timeout 300 rocprofv3 --pmc $* --kernel-trace --output-format csv -d $OUT/pmc_$name -- python3 $ROOT/bench.py $ARGS > $OUT/pmc_$name.log 2>&1 || echo "pmc $name failed"; }
run_pmc sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU
run_pmc sq2 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
run_pmc tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum
run_pmc ea TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_WRREQ_DRAM_sum TCC_EA0_RDREQ_128B_sum
run_pmc grbm GRBM_GUI_ACTIVE GRBM_COUNT
cd $ROOT
python3 tools/summarize_pmc.py $OUT > $OUT/summary.txt 2>&1
timeout 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_driver.json 2>$OUT/bench_driver.err
timeout 900 python3 bench.py > $OUT/bench_default.json 2>$OUT/bench_default.err
head -30 $OUT/summary.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/draw_both_trace -- python3 $ROOT/tools/deposit_bench.py 40 --both > $OUT/draw_both_trace.log 2>&1
cd $ROOT
grep '^{' $OUT/draw_both_trace.log > $OUT/draw_lines.txt
python3 tools/sharded_draw_probe.py 30 2>/dev/null | grep draw_both > $OUT/sharded_draw.txt
python3 tools/frame_wall_probe.py 200 2>/dev/null | grep wall_ms > $OUT/frame_wall.txt
python3 tools/band_sweep.py --out $OUT/band_sweep.txt > /dev/null 2>&1
TH_N=8192 TH_BENCH_TRACE=1 python3 tools/deposit_bench.py 100 --both > $OUT/c4_frame_loop.txt 2>&1
