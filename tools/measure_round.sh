# Measurement set committed under profiles/ once per round (run on the GPU box through gpurun): bash tools/measure_round.sh <tag>
set -x
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_c3_driver_protocol.json 2> $O/bench_c3.err
python bench.py --config c2 --steps 20 --warmup 5 > $O/bench_c2.json 2>/dev/null
python bench.py --config c1 --steps 20 --warmup 2 --no-cpu-baseline > $O/bench_c1.json 2>/dev/null
python bench.py --workload online --steps 6 --warmup 2 > $O/bench_c4_online.json 2>/dev/null
python bench.py --workload online --steps 6 --warmup 2 --tau-method TNC --no-cpu-baseline --lean > $O/bench_c4_online_reference_tau_TNC.json 2>/dev/null
PGPFA_PLAN_TRACE=1 python tools/jump_probe.py 6 > $O/jump_probe.txt 2>&1
PGPFA_PLAN_TRACE=1 python tools/jump_probe.py 6 extrapolate_guard=0 start_guard=0 workspace_grow_budget_ms=0 > $O/jump_probe_round5_behaviour.txt 2>&1
python bench.py --workload floor --steps 3 --warmup 1 --trials 128 > $O/bench_c3_floor_dense_engine.json 2>/dev/null
python bench.py --workload dual --config c5 --trials 256 > $O/bench_c5_dual_estep_mixed.json 2>/dev/null
python bench.py --workload dual --config c5 --trials 256 --precision f64 > $O/bench_c5_dual_estep_f64.json 2>/dev/null
python bench.py --workload dual --config c5 --trials 256 --dual-iters 6 --warmup 1 > $O/bench_c5_dual_unit_mixed.json 2>/dev/null
python bench.py --workload dual --config c5 --trials 256 --dual-iters 6 --warmup 1 --precision f64 > $O/bench_c5_dual_unit_f64.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ks -o ks -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --lean > $O/bench_c3_under_rocprof.json 2>/dev/null
cd $R
python tools/gemm_by_grid.py $O/ks > $O/kernels_by_grid.txt
python tools/trace_gaps.py $O/ks > $O/trace_gaps.txt
find $O/ks -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/ks
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --lean > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --lean > /dev/null 2>&1
cd $R
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write $O/pmc_hbm_traffic.json | head -8
rm -rf $O/pmc_fetch $O/pmc_write
python tools/gemm_shapes.py 12 > $O/gemm_shapes.txt 2>/dev/null
python tools/split_probe.py 512 > $O/split_probe.txt 2>/dev/null
python tools/pcg_probe.py 1024 4 init 210 > $O/pcg_probe.txt 2>/dev/null
python tools/pcg_probe.py 1024 4 init 01 pcg_rx32 > $O/pcg_probe_rx32.txt 2>/dev/null
PGPFA_MSTEP_OVERLAP=0 python bench.py --steps 20 --warmup 5 --lean --no-cpu-baseline > $O/bench_c3_mstep_one_after_the_other.json 2>/dev/null
python tools/fixed_point_probe.py 16 c2 1 0 > $O/fixed_point_probe_c2.txt 2>/dev/null
python tools/cold_start_probe.py > $O/cold_start.txt 2>/dev/null
# a solve's kernels and gaps (round 5): kernel trace of four E-steps in the default form
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/nt -o nt -- python3 $R/tools/pcg_probe.py 1024 4 init 2 > /dev/null 2>&1
cd $R
python tools/newton_trace.py $O/nt > $O/newton_trace.txt
rm -rf $O/nt
python tools/leak_probe.py 64 > $O/leak_probe.txt 2>&1
if [ -x tools/probes/pmc_width_probe ]; then
  cd /tmp
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/cal_f -o f -- $R/tools/probes/pmc_width_probe > $O/pmc_width_probe.txt 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/cal_w -o w -- $R/tools/probes/pmc_width_probe >> $O/pmc_width_probe.txt 2>&1
  cd $R
  python tools/pmc_calibrate.py $O/cal_f $O/cal_w 2147483648 $O/pmc_calibration.json > $O/pmc_calibration.txt
  rm -rf $O/cal_f $O/cal_w
fi
# (stand-alone probes: built here by make -C tools/probes, they travel with the snapshot)
if [ -x tools/probes/thin_probe ]; then for l in 1024 400 100; do timeout 60 tools/probes/thin_probe $l 48; done > $O/thin_probe.txt 2>&1; fi
if [ -x tools/probes/stride_probe ]; then timeout 60 tools/probes/stride_probe > $O/stride_probe.txt 2>&1; fi
python tools/em_trace.py 60 > $O/em_trace_60_iterations.txt 2>/dev/null
ls -la $O
