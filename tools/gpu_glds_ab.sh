# A/B of the GEMM staging variants on the scaled-config slice (diagnostic switches of tg_genw.hip)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-glds}
mkdir -p $O
K="wide or rollout_tokens or test_sample or nll"
TG_GENW_GLDS=1 TG_GENW_TILE=256 timeout -k 10 300 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread -k "$K" > $O/pytest_glds.log 2>&1 && \
TG_GENW_GLDS=1 timeout -k 10 300 python bench.py --workload c3s --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_glds.json 2> $O/bench.err && \
timeout -k 10 300 python bench.py --workload c3s --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_reg.json 2>> $O/bench.err && \
TG_GENW_PERSIST=0 timeout -k 10 300 python bench.py --workload c3s --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_reg_nopersist.json 2>> $O/bench.err && \
TG_GENW_GLDS=1 TG_GENW_PERSIST=0 timeout -k 10 300 python bench.py --workload c3s --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_glds_nopersist.json 2>> $O/bench.err
echo rc=$?
tail -n 3 $O/pytest_glds.log
