"""The multi-GPU path on hardware, within what a one-GPU box allows: each case runs in processes
of its own (tests/_gpu_dist_worker.py) under a time limit, so a stuck rendezvous or collective fails this test and
nothing else.  Collected after the parity tests (file name)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
WORKER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_gpu_dist_worker.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _finish(procs, limit):
    """Wait for the workers (they end themselves after 75 s with a traceback of where they are stuck); on a timeout
    kill exactly the PIDs started here and report what they printed."""
    outs, timed_out = [], False
    for p in procs:
        try:
            outs.append(p.communicate(timeout=limit)[0].decode(errors="replace"))
        except subprocess.TimeoutExpired:
            timed_out = True
            p.kill()
            outs.append(p.communicate()[0].decode(errors="replace"))
    for p, out in zip(procs, outs):
        assert not timed_out and p.returncode == 0 and "WORKER OK" in out, "\n----\n".join(o[-3000:] for o in outs)


def test_world_size_1_nccl_group_and_allgather_rows():
    """ROLLOUT.get_reward through an RCCL process group (one rank) equals the ungrouped call bit for bit;
    tg_allgather_rows runs on a real ncclComm_t."""
    p = subprocess.Popen([sys.executable, "-u", WORKER, "nccl1", str(_free_port())], stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT)
    _finish([p], 100)


def test_two_ranks_sharded_get_reward_on_the_gpu():
    """Two ranks (gloo rendezvous, both on this GPU) run the product's sharded get_reward: weights re-synchronised
    from rank 0, shards computed by the HIP kernels, raw rows all-gathered, same rewards on both ranks as the
    single-rank call."""
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, "-u", WORKER, "gloo2", str(port), str(r)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT) for r in range(2)]
    _finish(procs, 100)
