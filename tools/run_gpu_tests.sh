# usage (on the GPU box): bash tools/run_gpu_tests.sh <log> [pytest args...]
# pytest -m gpu with glibc's fatal messages on stderr and a post-mortem backtrace if the interpreter dies on a signal.
LOG=$1; shift
export LIBC_FATAL_STDERR_=1 PGPFA_BACKTRACE=1
ulimit -c unlimited
cd $GRAFT_REPO_ROOT
rm -f core core.* /tmp/core*
python -X faulthandler -m pytest tests -m gpu -x -q "$@" > $LOG 2>&1
RC=$?
tail -4 $LOG | cut -c1-300
if [ $RC -ge 128 ] || grep -q "Fatal Python error" $LOG; then
  cat /proc/sys/kernel/core_pattern
  CORE=$(ls -t core core.* /tmp/core* 2>/dev/null | head -1)
  echo "interpreter died (rc $RC), core: $CORE"
  if [ -n "$CORE" ]; then /opt/rocm/bin/rocgdb -batch -ex bt -ex "info threads" python "$CORE" > $LOG.bt 2>&1; grep -n "^#" $LOG.bt | head -40; fi
fi
exit $RC
