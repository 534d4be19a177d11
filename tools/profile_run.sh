# usage (on the GPU box, via gpurun): bash tools/profile_run.sh <tag> [bench args...]
# rocprofv3 kernel trace + stats of a lean bench run; leaves stats csv, the per-(kernel, grid) summary and the bench line
# under gpurun_out/<tag>/ (the raw trace is deleted: tens of MB).
TAG=$1; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o kt -- python3 $R/bench.py --no-cpu-baseline --lean "$@" > $O/bench.json 2> $O/bench.err
cd $R
python tools/gemm_by_grid.py $O/kt > $O/by_grid.txt
python tools/trace_gaps.py $O/kt > $O/gaps.txt
find $O/kt -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \;
rm -rf $O/kt
cat $O/by_grid.txt
tail -2 $O/bench.err
