"""developer tool (GPU box): does the row stride of the bf16 operands matter to the 4-wave GEMM?  Times pcad_gemm_nt (bf16 in, fp32
out: the instantiation the split-bf16 GEMMs of the fp32 model run) on in_proj / out_proj shaped problems whose operands sit in
buffers with chosen row strides (power-of-two strides of 4 / 8 KiB vs the 6 / 12 KiB of round 5's concatenated operands vs padded
ones), and pcad_gemm_nt_split (the wrap-around form on tight [hi | lo] operands, conversion passes included).
    python tools/gemm_stride_probe.py [M]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from plantcaduceus_amd.engine import load_library, _check, _stream_ptr
lib = load_library()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
dev = "cuda:0"


def run(tag, N, K, lda, ldw, reps=10):
    a = (torch.randn(M, lda, device=dev) * 0.5).bfloat16()
    w = (torch.randn(N, ldw, device=dev) / K ** 0.5).bfloat16()
    c = torch.empty(M, N, device=dev, dtype=torch.float32)
    def go():
        _check(lib.pcad_gemm_nt(a.data_ptr(), lda, w.data_ptr(), ldw, c.data_ptr(), N, M, N, K, 1, 0, _stream_ptr()), "pcad_gemm_nt")
    for _ in range(2): go()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): go()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"{tag:58s} N={N} K={K} lda={lda} ldw={ldw}: {ms:8.3f} ms  {2.0 * M * N * K / ms / 1e9:7.0f} TF", flush=True)
    del a, w, c


print(f"M = {M} token-rows, bf16 operands -> fp32 result (gemm256q_kernel<bf16, float, 0>)")
for N, Ko in ((4096, 1024), (1024, 2048)):
    run("concatenated operands of round 5 (K = 3 Ko, tight)", N, 3 * Ko, 3 * Ko, 3 * Ko)
    run("K = 2 Ko, tight rows (power-of-two stride)", N, 2 * Ko, 2 * Ko, 2 * Ko)
    run("K = 2 Ko, A and W rows padded by 128 bytes", N, 2 * Ko, 2 * Ko + 64, 2 * Ko + 64)
    run("K = 2 Ko, only A rows padded", N, 2 * Ko, 2 * Ko + 64, 2 * Ko)
    run("K = 2 Ko, only W rows padded", N, 2 * Ko, 2 * Ko, 2 * Ko + 64)
    run("K = 2 Ko, padded by 256 bytes", N, 2 * Ko, 2 * Ko + 128, 2 * Ko + 128)
    # the wrap-around form itself (operands converted to tight [hi | lo] inside the call: + one pass over A and W)
    x = torch.randn(M, Ko, device=dev); wt = torch.randn(N, Ko, device=dev) / Ko ** 0.5
    out = torch.empty(M, N, device=dev)
    nb = lib.pcad_gemm_nt_split_scratch_bytes(M, N, Ko)
    scr = torch.empty(nb + 256, dtype=torch.uint8, device=dev)
    def gs():
        _check(lib.pcad_gemm_nt_split(x.data_ptr(), Ko, wt.data_ptr(), Ko, out.data_ptr(), N, M, N, Ko, (scr.data_ptr() + 255) // 256 * 256, nb, _stream_ptr()), "split")
    for _ in range(2): gs()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(10): gs()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print(f"{'pcad_gemm_nt_split (wrap-around cursor, incl. conversions)':58s} N={N} Ko={Ko}: {ms:8.3f} ms  {2.0 * M * N * 3 * Ko / ms / 1e9:7.0f} TF (3 Ko)", flush=True)
    del x, wt, out, scr
