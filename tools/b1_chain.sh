#!/bin/bash
# B = 1 launch chain: frames/s and the per-kernel trace of one stereo frame per launch set.  bash tools/b1_chain.sh <outdir> [env assignments...]
set -e
out=$1; shift
mkdir -p $out
for kv in "$@"; do export "$kv"; done
python tools/bench_b1.py 1 2000 > $out/b1.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$out/trace -o b1 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/bench_b1.py 1 60 > $GRAFT_REPO_ROOT/$out/trace.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_gaps.py $out/trace > $out/chain.txt 2>&1 || true
cat $out/b1.txt; tail -32 $out/chain.txt
