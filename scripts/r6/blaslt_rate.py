"""What the vendor GEMM (hipBLASLt through torch.bmm / matmul, bfloat16) reaches on the 1x1 layers' shapes: out[n] [M, HW] = W [M, K] @ x[n] [K, HW]."""
import torch
N = 144
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return sorted(a.elapsed_time(b) for a, b in ev)[reps // 2]
for (M, K, HW) in ((2048, 512, 1024), (512, 2048, 1024), (1024, 256, 1024), (256, 2048, 1024), (2048, 1024, 1024), (2048, 1536, 1024), (256, 64, 16384)):
    w = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    x = torch.randn(N, K, HW, device="cuda", dtype=torch.bfloat16)
    out = torch.empty(N, M, HW, device="cuda", dtype=torch.bfloat16)
    ms = timed(lambda: torch.matmul(w, x, out=out))
    fl = 2.0 * N * M * K * HW
    # the same contraction as ONE GEMM over all images: [M, K] @ [K, N HW] needs x as [K, N, HW] (a transposed copy) -- library's best case
    x2 = x.permute(1, 0, 2).reshape(K, N * HW).contiguous()
    out2 = torch.empty(M, N * HW, device="cuda", dtype=torch.bfloat16)
    ms2 = timed(lambda: torch.mm(w, x2, out=out2))
    print("%4d->%4d HW %5d: batched %.3f ms %.0f TF (%.2f of 2500) | one GEMM %.3f ms %.0f TF (%.2f)" % (K, M, HW, ms, fl / ms / 1e9, fl / ms / 1e9 / 2500, ms2, fl / ms2 / 1e9, fl / ms2 / 1e9 / 2500), flush=True)
