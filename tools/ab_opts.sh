R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for o in "$@"; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_$o -o r -- python3 $R/bench.py --steps 6 --warmup 2 --lean --no-cpu-baseline --opts $o > /tmp/b_$o.txt 2>&1
  f=$(find /tmp/p_$o -name "*kernel_stats.csv" | head -1)
  echo "== $o $(grep -o '"ms_per_step": [0-9.]*' /tmp/b_$o.txt)"
  if [ -n "$f" ]; then python3 $R/tools/kstat.py "$f" ${KS:-mix}; fi
done
