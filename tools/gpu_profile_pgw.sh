# rocprofv3 kernel stats of the policy-gradient step at configs[2]'s generator shape (bench.py --workload pgw)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${1:-prof_pgw}
mkdir -p $O
timeout -k 10 200 python3 bench.py --workload pgw --steps 10 --warmup 2 > $O/bench_pgw.json 2> $O/bench_pgw.err && \
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --workload pgw --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_pgw_under_rocprof.json 2> $O/stats.err
echo rc=$?
