"""Round 6 / config 5: the e4m3 attention-score kernel (crd_attn_scores_fp8) beside the bf16 one (crd_attn_scores) on the four encoder
stages, graph-replayed, plus the per-head quantisation launches an e4m3 path would add in front of it.  Usage: python tools/bench_attn_fp8.py [B]"""
import sys

import torch

from camradepth_amd import lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = lib.load()


def timed(fn, reps=20):
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps):
                fn()
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print(f"B = {B}; us per launch, graph-replayed x20")
for stage, (N, heads, d) in enumerate([(6656, 1, 64), (1664, 2, 64), (416, 4, 40), (104, 8, 32)], 1):
    M, C = 104, heads * d
    q = (torch.randn(B, N, C) * 0.8).to(torch.bfloat16).cuda()
    k = (torch.randn(B, M, C) * 0.8).to(torch.bfloat16).cuda()
    q8 = torch.zeros(B, N, heads, 64, dtype=torch.uint8, device="cuda")
    k8 = torch.zeros(B, M, heads, 64, dtype=torch.uint8, device="cuda")
    S, idx = torch.zeros(B, N, device="cuda"), torch.zeros(B, N, heads, dtype=torch.int16, device="cuda")
    sc = d ** -0.5

    def quant():
        for h in range(heads):
            lib.check(L.crd_quant_fp8(q.data_ptr(), B * N, C, h * d, d, q8.data_ptr(), heads * 64, h * 64, 0.01, lib.stream()))
            lib.check(L.crd_quant_fp8(k.data_ptr(), B * M, C, h * d, d, k8.data_ptr(), heads * 64, h * 64, 0.01, lib.stream()))
    quant()
    t16 = timed(lambda: lib.check(L.crd_attn_scores(q.data_ptr(), k.data_ptr(), B, N, M, heads, d, sc, S.data_ptr(), idx.data_ptr(), lib.stream())))
    t8 = timed(lambda: lib.check(L.crd_attn_scores_fp8(q8.data_ptr(), k8.data_ptr(), B, N, M, heads, 1e-4, sc, S.data_ptr(), idx.data_ptr(), lib.stream())))
    tq = timed(quant)
    print(f"stage {stage}: N {N:5d} heads {heads} d {d}:  bf16 {t16:6.2f}   e4m3 {t8:6.2f}   ({100 * (t16 - t8) / t16:+.1f} %)   "
          f"+ quantising q and k as {2 * heads} separate launches {tq:6.2f}")
