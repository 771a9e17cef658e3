# rocprofv3 passes over the scaled-config slice (bench.py --workload c3s): kernel stats, then counters in runs of their own
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${1:-prof_c3s}
mkdir -p $O
B="python3 bench.py --workload c3s --steps 1 --warmup 1 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- $B > $O/bench_under_rocprof.json 2> $O/stats.err && \
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_clock -o p -- $B > /dev/null 2> $O/pmc_clock.err && \
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- $B > /dev/null 2> $O/pmc_fetch.err && \
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- $B > /dev/null 2> $O/pmc_write.err
echo rc=$?
ls -R $O | head -40
# full-size configs[2] on this one GPU (rollout_num 64, ~40 s per call)
timeout -k 10 400 python3 bench.py --workload c3 --steps 1 --warmup 0 --no-cpu-baseline > $O/bench_c3_full.json 2> $O/bench_c3_full.err
echo rc_full=$?
