"""developer tool: time pcad_gemm_nt on the in_proj / out_proj shapes and check it against torch.
   python tools/gemm_time.py [M] [--no-vendor]      (M token-rows, default 65536; the benchmark's launches are 524288)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plantcaduceus_amd import ops
M0 = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 65536
vendor = "--no-vendor" not in sys.argv
for (M, N, K) in [(M0, 4096, 1024), (M0, 1024, 2048)]:
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / K ** 0.5).bfloat16()
    out = ops.linear(x, w)
    ref = (x[:4096].float() @ w.float().t())
    err = ((out[:4096].float() - ref).abs().max() / ref.abs().max()).item()
    ref2 = (x[-512:].float() @ w.float().t())
    err2 = ((out[-512:].float() - ref2).abs().max() / ref2.abs().max()).item()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(20): ops.linear(x, w)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print(f"M={M} N={N} K={K}: {ms:.4f} ms  {2.0*M*N*K/ms/1e9:.0f} TF  relerr {err:.2e} {err2:.2e}")
    if not vendor:
        continue
    # yardstick only (not used by the product path): the vendor library on the same shape
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for _ in range(3): torch.matmul(x, w.t(), out=y)
    torch.cuda.synchronize(); a.record()
    for _ in range(20): torch.matmul(x, w.t(), out=y)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print(f"   vendor library (torch.matmul): {ms:.4f} ms  {2.0*M*N*K/ms/1e9:.0f} TF")
