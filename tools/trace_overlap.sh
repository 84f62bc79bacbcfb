#!/bin/bash
# On the GPU box: kernel trace of one bench command, then the concurrency census of its timed frames (last 40 % of the run).
# usage: tools/trace_overlap.sh <tag> [bench args...]   -> gpurun_out/<tag>_overlap.txt
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace -d /tmp/tr_$TAG -o t -- python3 $ROOT/bench.py "$@" --no-cpu-baseline > /tmp/tr_$TAG.log 2>&1 || { tail -5 /tmp/tr_$TAG.log; exit 1; }
DB=$(find /tmp/tr_$TAG -name "*.db" | head -1)
python3 $ROOT/tools/kframe.py $DB ${FRAME:-3} all > $ROOT/gpurun_out/${TAG}_frame.txt; head -24 $ROOT/gpurun_out/${TAG}_frame.txt; exit 0
cat $ROOT/gpurun_out/${TAG}_overlap.txt
