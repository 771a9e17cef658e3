#!/bin/bash
# LeakGAN rollout on the GPU box: its parity tests, the bench line with the call spread over four streams and with
# everything on one (TG_LEAK_ONE_STREAM=1), and a kernel trace of a short run.  Usage: gpurun -- bash tools/gpu_leak_check.sh <tag>
set -o pipefail
tag=${1:-leak}
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -u -m pytest tests -m gpu -x -q -k "leak" --timeout 120 --timeout-method thread > $out/test_leak.log 2>&1
rc=$?
tail -3 $out/test_leak.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 120 python bench.py --workload leak > $out/leak_streams.json 2> $out/leak_streams.err || exit 1
TG_LEAK_ONE_STREAM=1 timeout -k 10 120 python bench.py --workload leak > $out/leak_one.json 2> $out/leak_one.err || exit 1
TG_LEAK_SPLIT_CELL=1 timeout -k 10 120 python bench.py --workload leak > $out/leak_splitcell.json 2> $out/leak_splitcell.err || exit 1
cut -c1-330 $out/leak_streams.json $out/leak_one.json $out/leak_splitcell.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$tag -o leak -- python3 $GRAFT_REPO_ROOT/bench.py --workload leak --steps 3 --warmup 2 > $out/prof.log 2>&1 || exit 1
cp /tmp/prof_$tag/leak_kernel_trace.csv $out/leak_trace.csv
# the same trace with everything on one stream: every kernel's duration without a neighbour
export TG_LEAK_ONE_STREAM=1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof1_$tag -o leak -- python3 $GRAFT_REPO_ROOT/bench.py --workload leak --steps 3 --warmup 2 > $out/prof1.log 2>&1 || exit 1
cp /tmp/prof1_$tag/leak_kernel_trace.csv $out/leak_trace_one.csv
