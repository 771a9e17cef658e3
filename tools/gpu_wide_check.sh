# generator tests + the scaled-config slice bench (tile shape / staging switches exist in `make DIAG=1` builds only)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-wide}
mkdir -p $O
K="wide or rollout or test_sample or get_reward or teacher_forced or nll"
timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread -k "$K" > $O/pytest.log 2>&1 && \
timeout -k 10 400 python bench.py --workload c3s --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_c3s.json 2> $O/bench_c3s.err
echo rc=$?
tail -n 3 $O/pytest.log
tail -n 5 $O/bench_c3s.err
