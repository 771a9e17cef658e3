# rocprofv3 counter passes over the default workload (bench.py, configs[1]), each in a run of its own
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/${1:-prof_c2}
mkdir -p $O
B="python3 bench.py --workload c2 --steps 5 --warmup 2 --no-cpu-baseline"
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_clock -o p -- $B > /dev/null 2> $O/pmc_clock.err && \
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o p -- $B > /dev/null 2> $O/pmc_fetch.err && \
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o p -- $B > /dev/null 2> $O/pmc_write.err
echo rc=$?
