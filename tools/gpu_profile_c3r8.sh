# rocprofv3 passes over the `scaled` sub-record's workload (bench.py --workload c3r8: configs[2] shape at rollout_num 8):
# kernel stats, then counters in runs of their own
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${1:-prof_c3r8}
mkdir -p $O
B="python3 bench.py --workload c3r8 --steps 1 --warmup 1 --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o s -- $B > $O/bench_under_rocprof.json 2> $O/stats.err && \
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_clock -o p -- $B > /dev/null 2> $O/pmc_clock.err && \
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- $B > /dev/null 2> $O/pmc_fetch.err && \
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- $B > /dev/null 2> $O/pmc_write.err
echo rc=$?
