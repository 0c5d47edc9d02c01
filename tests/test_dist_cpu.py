"""CPU suite, part 5: the N>1 path over gloo, world sizes 2, 3, 4 and 8 (see tests/dist_worker.py)."""
import os
import socket
import subprocess
import sys


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


import pytest


@pytest.mark.parametrize("world", [2, 3, 4, 8])
def test_unit_sharded_exchange(tmp_path, world):
    """G = 2 (9/9 units), G = 3 (one source domain per rank), G = 4 (uneven: 5/5/4/4 units), G = 8 (the node of the scaling run: 3/3/2/2/2/2/2/2) -- SURVEY 8e."""
    port = _free_port()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(port), str(tmp_path)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o[-3000:])
        assert os.path.exists(os.path.join(str(tmp_path), "ok_%d" % r))


def test_forced_one_rank_plan_takes_the_sharded_path():
    """bench.py --force_dist: a ONE-rank RowPlan with force=True keeps the sharded code path -- domain-major local order, padded
    gather buffer, index_select back into collate order -- so that the first RCCL run of that path needs no second GPU."""
    import numpy as np
    import torch
    from aadg_amd.distributed import RowPlan
    D, B, M = 3, 2, 6
    N = D * B * M
    plain, forced = RowPlan(D, B, M), RowPlan(D, B, M, 0, 1, 'unit', force=True)
    assert not plain.sharded and forced.sharded and forced.world == 1 and forced.counts == [N] and forced.loss_weight == 1.0
    assert np.array_equal(np.sort(forced.rows), np.arange(N)) and not np.array_equal(forced.rows, np.arange(N))   # domain-major order
    d = forced.rows // M % D
    assert np.array_equal(d, np.sort(d))                                   # all rows of domain 0, then 1, then 2
    local = torch.arange(N, dtype=torch.float32)[torch.from_numpy(forced.rows)].view(N, 1)      # "embeddings" of the local rows
    back = forced.gather(local)                                            # no process group: all_gather degenerates to a copy
    assert torch.equal(back.view(-1), torch.arange(N, dtype=torch.float32))
