"""CPU suite, part 5: the N>1 path over gloo, world sizes 2, 3 and 4 (see tests/dist_worker.py)."""
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


@pytest.mark.parametrize("world", [2, 3, 4])
def test_unit_sharded_exchange(tmp_path, world):
    """G = 2 (9/9 units), G = 3 (one source domain per rank), G = 4 (uneven: 5/5/4/4 units) -- SURVEY 8e."""
    port = _free_port()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(port), str(tmp_path)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o[-3000:])
        assert os.path.exists(os.path.join(str(tmp_path), "ok_%d" % r))
