set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06b; mkdir -p $O; cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_c3_driver_protocol.json 2> $O/bench_c3.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --lean > $O/bench_c3_under_rocprof.json 2>/dev/null
cd $R
python tools/gemm_by_grid.py $O/ks > $O/kernels_by_grid.txt
python tools/trace_gaps.py $O/ks > $O/trace_gaps.txt
find $O/ks -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/ks
