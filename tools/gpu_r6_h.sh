# round 6: why is the emit 4 % slower than round 5's with the same instructions?  skip_unseen on / off on both libraries
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r6h
mkdir -p $OUT
cd /tmp
for skip in 1 0; do
TH_SKIP_UNSEEN=$skip TH_LIB=$GRAFT_REPO_ROOT/tools/bin/r5/libtendrils_hip.so timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_r5_skip$skip -- python3 $GRAFT_REPO_ROOT/tools/deposit_bench.py 60 --both > $OUT/trace_r5_skip$skip.log 2>&1
TH_SKIP_UNSEEN=$skip TH_EXP_BLEND=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_head_skip$skip -- python3 $GRAFT_REPO_ROOT/tools/deposit_bench.py 60 --both > $OUT/trace_head_skip$skip.log 2>&1
done
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --config c5 --steps 20 --warmup 5 --no-cpu --no-traffic > $OUT/bench_c5.json 2> $OUT/bench_c5.err
timeout 900 python bench.py --config c4 --steps 20 --warmup 5 --no-cpu --no-traffic > $OUT/bench_c4.json 2> $OUT/bench_c4.err
ls $OUT
