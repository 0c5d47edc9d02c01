"""CPU suite, part 5: the N>1 path over gloo, world_size 2 (see tests/dist_worker.py)."""
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


def test_row_sharded_exchange_world2(tmp_path):
    port = _free_port()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(port), str(tmp_path)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, "rank %d failed:\n%s" % (r, o[-3000:])
        assert os.path.exists(os.path.join(str(tmp_path), "ok_%d" % r))
