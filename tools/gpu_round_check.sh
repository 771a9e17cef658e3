mkdir -p gpurun_out/r01f
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/r01f/pytest.log 2>&1 && \
timeout -k 10 200 python bench.py --steps 20 --warmup 3 > gpurun_out/r01f/bench_c2.json 2> gpurun_out/r01f/bench_c2.err && \
timeout -k 10 120 python bench.py --workload c1 --steps 20 --warmup 3 > gpurun_out/r01f/bench_c1.json 2>gpurun_out/r01f/bench_c1.err && \
timeout -k 10 120 python bench.py --workload c5 --steps 10 --warmup 2 > gpurun_out/r01f/bench_c5.json 2>gpurun_out/r01f/bench_c5.err && \
timeout -k 10 120 python bench.py --workload pg --steps 50 --warmup 5 > gpurun_out/r01f/bench_pg.json 2>gpurun_out/r01f/bench_pg.err && \
timeout -k 10 120 python bench.py --workload c4 --steps 10 --warmup 2 > gpurun_out/r01f/bench_c4.json 2>gpurun_out/r01f/bench_c4.err && \
export TMPDIR=/tmp && \
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r01f/prof_c2 -o c2 -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r01f/bench_c2_under_rocprof.json 2> gpurun_out/r01f/rocprof_c2.err && \
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/r01f/prof_pg -o pg -- python3 bench.py --workload pg --steps 20 --warmup 2 --no-cpu-baseline > gpurun_out/r01f/bench_pg_under_rocprof.json 2> gpurun_out/r01f/rocprof_pg.err
echo exit=$?
tail -3 gpurun_out/r01f/pytest.log
