#!/bin/bash
# One GPU-box session: GPU tests, the default bench line, a rocprofv3 kernel trace of the same command and its
# summaries.  usage: tools/gpu_round.sh <tag> [pytest-args...]   (outputs under gpurun_out/<tag>_*)
# (the traced run takes --no-side: the side measurements start a child process, whose trace database the summary might pick)
tag=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd $R
if [ "${SKIP_TESTS:-0}" != "1" ]; then
  timeout 1500 python -m pytest tests -m gpu -q "$@" > $O/${tag}_tests.log 2>&1
  echo "tests rc=$?" >> $O/${tag}_tests.log
  tail -5 $O/${tag}_tests.log
fi
timeout 600 python bench.py ${BENCH_ARGS:-} > $O/${tag}_bench.json 2> $O/${tag}_bench.err
echo "bench rc=$?"; tail -c 600 $O/${tag}_bench.json
if [ "${SKIP_PROF:-0}" != "1" ]; then
  export TMPDIR=/tmp
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/${tag}_prof -- python $R/bench.py --steps 40 --no-cpu-baseline --no-side ${BENCH_ARGS:-} > $O/${tag}_prof_bench.json 2> $O/${tag}_prof.err)
  python tools/prof_summary.py $O/${tag}_prof $O/${tag}_kernel_stats.txt
  python tools/prof_timeline.py $O/${tag}_prof 2 > $O/${tag}_timeline.txt 2>&1
  rm -rf $O/${tag}_prof
  head -30 $O/${tag}_kernel_stats.txt
fi
