"""GPU: the N > 1 path of bench.py on the one GPU of the test box -- `python bench.py --gpus 2 --all_ranks_on_gpu0` starts its
two ranks itself (torch.distributed.run, gloo because RCCL refuses two ranks on one device), shards the (domain, policy) units,
all-gathers the embeddings, all-reduces the BatchNorm sums between the HIP kernels and the gradients through the package's own
bucketed reducer (aadg_amd/reducer.py; DistributedDataParallel until round 5).

With synchronised statistics, a float32 backbone and dropout off, the sharded job computes the SAME function as the single
rank, for both placement laws and for an uneven 3-rank split: the first step's rewards (forward pass) agree to rounding, and
so does every parameter gradient (test_ddp_sync_bn_gradients_match_single_process).  Later steps only agree loosely: Adam's
first updates are lr * sign(g) and many gradient elements sit at the float32 noise level, so two mathematically equal runs with
different summation orders drift by a few per cent (measured: single-rank reruns reproduce to 1e-6 at step 2, 1e-3 at step 3)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(tmp_path, gpus, tag, extra=(), batch=2, backbone="mobilenet_v2", dtype="fp32", size=64):
    dump = os.path.join(str(tmp_path), "rewards_%s.json" % tag)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(gpus), "--steps", "3", "--warmup", "0", "--size", str(size),
           "--batch", str(batch), "--backbone", backbone, "--backbone_dtype", dtype, "--no_cpu_baseline", "--no_dropout",
           "--dump_rewards", dump] + list(extra)
    if gpus > 1:
        cmd.append("--all_ranks_on_gpu0")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=1500)
    assert p.returncode == 0, p.stderr.decode()[-4000:]
    line = [l for l in p.stdout.decode().splitlines() if l.startswith("{")][-1]
    return json.loads(line), json.load(open(dump))


def test_bench_gpus_flag_launches_the_ranks_and_matches_single_rank(tmp_path):
    one, r1 = _bench(tmp_path, 1, "g1")
    assert one["n_gpus"] == 1
    # (4, "unit"): 18 (domain, policy) units over 4 ranks = 5 / 5 / 4 / 4 -- the uneven split of BASELINE configs[3], with the HIP kernels
    for gpus, law in ((2, "row"), (2, "unit"), (3, "unit"), (4, "unit")):
        out, rg = _bench(tmp_path, gpus, "g%d%s" % (gpus, law), ["--placement", law])
        assert out["n_gpus"] == gpus and out["steps"] == 3
        assert "BatchNorm statistics all-reduce" in out["config"]["parallelism"]
        if gpus == 4:
            assert "rows per rank 10/10/8/8" in out["config"]["parallelism"], out["config"]["parallelism"]
        # raw rewards = sums of Sinkhorn divergences (the normalised ones divide by their small spread)
        a, b = r1["raw"][0], rg["raw"][0]
        assert max(abs(x - y) for x, y in zip(a, b)) < 1e-4, (gpus, law, a, b)           # step 1: the forward pass is the same function
        for a, b in zip(r1["raw"][1:], rg["raw"][1:]):
            assert max(abs(x - y) for x, y in zip(a, b)) < 0.25 * max(abs(x) for x in a), (gpus, law, a, b)


def test_bench_eight_ranks_like_the_scaling_run(tmp_path):
    """The world size of the driver's scaling run (8 ranks: 18 units = 3/3/2/2/2/2/2/2, ragged padded gather) with the HIP kernels, all
    ranks on the one GPU of the test box: first-step rewards equal the single rank's."""
    one, r1 = _bench(tmp_path, 1, "g1b3", batch=3)
    out, rg = _bench(tmp_path, 8, "g8unit", ["--placement", "unit"], batch=3)
    assert out["n_gpus"] == 8 and out["steps"] == 3 and out["config"]["distributed"]["world_size"] == 8
    assert "rows per rank 9/9/6/6/6/6/6/6" in out["config"]["parallelism"], out["config"]["parallelism"]
    a, b = r1["raw"][0], rg["raw"][0]
    assert max(abs(x - y) for x, y in zip(a, b)) < 1e-4, (a, b)


@pytest.mark.parametrize("name,dtype,world", [("mobilenet_v2", "fp32", 3), ("resnet50", "fp32", 2), ("resnet50", "bf16", 3),
                                              ("resnet50", "f32x3", 3)])
def test_ddp_sync_bn_gradients_match_single_process(name, dtype, world):
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "gpu_ddp_worker.py"), name, dtype]
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, env=env, timeout=1500)
    assert p.returncode == 0, p.stdout.decode()[-4000:]


def test_force_dist_one_rank_over_rccl(tmp_path):
    """Round 4: `bench.py --gpus 1 --force_dist --dist_backend nccl` runs the ONE rank through the whole
    distributed path over RCCL -- init_process_group("nccl", device_id=...), the row plan in sharded (domain-major) order, the padded
    all_gather_into_tensor of the embeddings and the float64 BatchNorm statistics all-reduce on the small-collectives process group,
    the policy broadcasts, DDP-wrapped model and discriminator -- and must compute the same rewards as the plain run."""
    one, r1 = _bench(tmp_path, 1, "plain")
    assert one["config"]["distributed"]["initialized"] is False and one["config"]["world_size"] == 1
    out, rg = _bench(tmp_path, 1, "rccl", ["--force_dist", "--dist_backend", "nccl"])
    d = out["config"]["distributed"]
    assert d["initialized"] and d["backend"] == "nccl" and d["world_size"] == 1 and d["rccl_version"], d
    assert d["forced_one_rank_run_of_the_distributed_path"] and "own process group" in d["small_collectives_group"], d
    assert d["collectives_per_step"]["batchnorm_statistics_all_reduce"] > 50, d           # MobileNetV2: 52 layers, both directions
    t = d["small_collectives_gpu_ms_per_step"]
    assert t["all_gather"]["calls_per_step"] == 1 and t["policy_broadcast"]["calls_per_step"] == 2, t
    assert t["batchnorm_statistics_all_reduce"]["calls_per_step"] == d["collectives_per_step"]["batchnorm_statistics_all_reduce"], t
    assert "BatchNorm statistics all-reduce" in out["config"]["parallelism"]
    a, b = r1["raw"][0], rg["raw"][0]
    assert max(abs(x - y) for x, y in zip(a, b)) < 1e-4, (a, b)
    for a, b in zip(r1["raw"][1:], rg["raw"][1:]):
        assert max(abs(x - y) for x, y in zip(a, b)) < 0.25 * max(abs(x) for x in a), (a, b)


@pytest.mark.parametrize("dtype", ["f32x3", "bf16"])
def test_force_dist_resnet50_sync_bn_over_rccl(tmp_path, dtype):
    """Round 5: the ResNet-50 paths of the synchronised BatchNorm -- `sync_batch_norm_act_group` (the five ASPP
    layers share one all-reduce per direction) and `sync_batch_norm_shortcut_pair` (bn3 + the projection shortcut's BatchNorm) -- had run
    over gloo only.  One rank through the whole distributed path over RCCL with the headline's backbone, in the headline's arithmetic
    (f32x3) and under bfloat16 autocast: the first step's rewards equal the plain single-process run's (float32-grade: 1e-3; bfloat16:
    loosely), and the collectives are the expected ones (DeepLabV3+/ResNet-50: 62 BatchNorm layers, the five of ASPP grouped)."""
    one, r1 = _bench(tmp_path, 1, "r50plain_" + dtype, backbone="resnet50", dtype=dtype, size=128)
    out, rg = _bench(tmp_path, 1, "r50rccl_" + dtype, ["--force_dist", "--dist_backend", "nccl"], backbone="resnet50", dtype=dtype, size=128)
    d = out["config"]["distributed"]
    assert d["initialized"] and d["backend"] == "nccl" and d["world_size"] == 1 and d["forced_one_rank_run_of_the_distributed_path"], d
    assert "own process group" in d["small_collectives_group"] and "own process group" in d["side_stream_collectives_group"], d
    n_bn = d["collectives_per_step"]["batchnorm_statistics_all_reduce"]
    assert 100 <= n_bn <= 124, n_bn              # 62 BatchNorm layers x 2 directions, minus the grouped (ASPP: 5 -> 1) / paired ones
    t = d["small_collectives_gpu_ms_per_step"]
    assert t["all_gather"]["calls_per_step"] == 1 and t["policy_broadcast"]["calls_per_step"] == 2, t
    a, b = r1["raw"][0], rg["raw"][0]
    tol = 1e-3 if dtype == "f32x3" else 0.1 * max(abs(x) for x in a)     # (other BatchNorm kernels on the sharded path: float64 totals)
    assert max(abs(x - y) for x, y in zip(a, b)) < tol, (a, b)
