# Round check on one MI355X: the whole GPU test suite, every bench workload, rocprofv3 kernel stats and counter passes of
# the default workload, the self-launched 2-rank rehearsal.  Outputs under gpurun_out/<tag>/ (copy what is to be kept
# into profiles/).
cd $GRAFT_REPO_ROOT
T=${1:-round}
O=gpurun_out/$T
mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread > $O/pytest.log 2>&1 && \
timeout -k 10 500 python bench.py --steps 20 --warmup 3 --scaled-steps 5 --scaled-warmup 1 > $O/bench_default.json 2> $O/bench_default.err && \
timeout -k 10 120 python bench.py --workload leak --steps 10 --warmup 2 > $O/bench_leak.json 2> $O/bench_leak.err && \
timeout -k 10 120 python bench.py --workload c1 --steps 20 --warmup 3 > $O/bench_c1.json 2> $O/bench_c1.err && \
timeout -k 10 120 python bench.py --workload c5 --steps 10 --warmup 2 > $O/bench_c5.json 2> $O/bench_c5.err && \
timeout -k 10 120 python bench.py --workload pg --steps 50 --warmup 5 > $O/bench_pg.json 2> $O/bench_pg.err && \
timeout -k 10 200 python bench.py --workload pgw --steps 10 --warmup 2 > $O/bench_pgw.json 2> $O/bench_pgw.err && \
timeout -k 10 200 python bench.py --workload pgw --steps 10 --warmup 2 --gen-precision fp32 --no-cpu-baseline > $O/bench_pgw_fp32.json 2> $O/bench_pgw_fp32.err && \
timeout -k 10 120 python bench.py --workload c4 --steps 10 --warmup 2 > $O/bench_c4.json 2> $O/bench_c4.err && \
timeout -k 10 300 python bench.py --workload c3s --steps 3 --warmup 1 > $O/bench_c3s.json 2> $O/bench_c3s.err && \
TG_BENCH_REHEARSE=1 timeout -k 10 200 python3 bench.py --gpus 2 --steps 3 --warmup 1 --scaled-steps 1 --scaled-warmup 0 > $O/bench_rehearse_2ranks.json 2> $O/bench_rehearse_2ranks.err && \
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c2 -o c2 -- python3 bench.py --workload c2 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c2_under_rocprof.json 2> $O/rocprof_c2.err
echo exit=$?
tail -n 3 $O/pytest.log
