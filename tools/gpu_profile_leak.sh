# rocprofv3 kernel stats of the LeakGAN rollout side workload (bench.py --workload leak)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${1:-prof_leak}
mkdir -p $O
timeout -k 10 120 python3 bench.py --workload leak --steps 10 --warmup 2 > $O/bench_leak.json 2> $O/bench_leak.err && \
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- python3 bench.py --workload leak --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_leak_under_rocprof.json 2> $O/stats.err
echo rc=$?
