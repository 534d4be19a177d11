set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --config c2 --steps 10 --warmup 2 > $O/bench_c2.json 2>/dev/null
python bench.py --config c1 --steps 20 --warmup 2 --no-cpu-baseline > $O/bench_c1.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2>/dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_hbm_traffic.json | head -5
rm -rf $O/pmc_fetch $O/pmc_write
ls -la $O
