"""Developer microbench: attention score kernels at the four encoder stage shapes, graph-replayed over rotating operands."""
import sys
sys.path.insert(0, ".")
import torch
from camradepth_amd import lib
L = lib.load()
P_ = lambda t: None if t is None else t.data_ptr()
B = 8
STAGES = [(6656, 104, 1, 64), (1664, 104, 2, 64), (416, 104, 5, 32), (104, 104, 8, 32)]
NSET, REPS = 6, 48


def timeit(fn):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            fn()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / REPS


for N, M, heads, d in STAGES:
    C = heads * d
    sets = []
    for i in range(NSET):
        q = torch.randn(B, N, C, device="cuda").to(torch.bfloat16)
        k = torch.randn(B, M, C, device="cuda").to(torch.bfloat16)
        S = torch.zeros(B, N, device="cuda"); idx = torch.zeros(B, N, heads, dtype=torch.int16, device="cuda")
        dS = torch.randn(B, N, device="cuda"); dq = torch.zeros_like(q)
        sets.append((q, k, S, idx, dS, dq))
        L.crd_attn_scores(P_(q), P_(k), B, N, M, heads, d, d ** -0.5, P_(S), P_(idx), lib.stream())
    nparts = L.crd_attn_scores_bwd_partials(B, N, M, heads, d)
    parts = torch.zeros(max(nparts, 1) * B * M * C, device="cuda")
    dk = torch.zeros(B, M, C, dtype=torch.int64, device="cuda")   # crd_sum_t

    def fwd():
        for r in range(REPS):
            q, k, S, idx, dS, dq = sets[r % NSET]
            L.crd_attn_scores(P_(q), P_(k), B, N, M, heads, d, d ** -0.5, P_(S), P_(idx), lib.stream())

    def bwd_parts():
        for r in range(REPS):
            q, k, S, idx, dS, dq = sets[r % NSET]
            L.crd_attn_scores_bwd(P_(q), P_(k), P_(dS), P_(idx), B, N, M, heads, d, d ** -0.5, P_(dq), None, P_(parts), lib.stream())

    def bwd_atomic():
        for r in range(REPS):
            q, k, S, idx, dS, dq = sets[r % NSET]
            L.crd_attn_scores_bwd(P_(q), P_(k), P_(dS), P_(idx), B, N, M, heads, d, d ** -0.5, P_(dq), P_(dk), None, lib.stream())

    print(f"N{N:5d} C{C:4d} heads {heads}: scores {timeit(fwd):6.2f} us   bwd(partials, {nparts} wg/sample) {timeit(bwd_parts):6.2f} us   "
          f"bwd(atomics) {timeit(bwd_atomic):6.2f} us")
