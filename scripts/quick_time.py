"""Ad-hoc timing of the two hot kernels at BASELINE size (not the contract bench; see bench.py)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from aadg_amd import _lib
from helpers import random_units, synth_pool

def main():
    H = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    rs = np.random.RandomState(1023)
    D, B, M = 3, 8, 6
    P, N = D * B, D * B * M
    t0 = time.time(); imgs, msks = synth_pool(rs, P, H, H); print("synth %.1fs" % (time.time() - t0))
    units = random_units(rs, N, P, H, H, H, (1.0, 1.5))
    d_img, d_msk = torch.from_numpy(imgs).cuda(), torch.from_numpy(msks).cuda()
    oi = torch.empty((N, 3, H, H), device="cuda"); ol = torch.empty((N, 2, H, H), device="cuda")
    for _ in range(3): _lib.aug_u8_forward(d_img, d_msk, units, H, 0, oi, ol)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 20
    pe = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    pe[0].record(); pe[1].record(); torch.cuda.synchronize()
    _lib.PROFILE_EVENTS = pe
    _lib.aug_u8_forward(d_img, d_msk, units, H, 0, oi, ol); torch.cuda.synchronize()
    print("dominant kernel(s): %.3f ms" % pe[0].elapsed_time(pe[1]))
    _lib.PROFILE_EVENTS = None
    e0.record()
    for _ in range(K): _lib.aug_u8_forward(d_img, d_msk, units, H, 0, oi, ol)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / K
    alg = N * (3 * H * H + H * H + 5 * H * H * 4)
    print("aug %dx%d N=%d: %.3f ms/batch  -> %.1f img/s, algorithmic %.1f MB -> %.0f GB/s" % (H, H, N, ms, N / ms * 1e3, alg / 1e6, alg / ms / 1e6))
    fe = torch.randn(N, 128, device="cuda"); fe = torch.nn.functional.leaky_relu(fe, 0.2)
    r = torch.zeros(M, device="cuda")
    for _ in range(3): _lib.sinkhorn_rewards(fe, D, B, M, rewards=r)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(100): _lib.sinkhorn_rewards(fe, D, B, M, rewards=r)
    e1.record(); torch.cuda.synchronize()
    print("sinkhorn rewards (18 problems 8x8x128): %.1f us/call" % (e0.elapsed_time(e1) / 100 * 1e3))

main()
