"""Per-dispatch cost of tiny kernels inside a captured HIP graph: same kernel repeated vs distinct kernels alternating."""
import torch
from camradepth_amd import lib as L

lib = L.load()
dev = "cuda"
a = torch.zeros(4096, device=dev); b = torch.zeros(4096, device=dev)
h = torch.zeros(4096, device=dev, dtype=torch.bfloat16)
N = 1200


def seq_same():
    st = L.stream()
    for i in range(N):
        lib.crd_scale_f32(a.data_ptr(), b.data_ptr(), 4096, 1.0, st)


def seq_mix():
    st = L.stream()
    for i in range(N // 3):
        lib.crd_scale_f32(a.data_ptr(), b.data_ptr(), 4096, 1.0, st)
        lib.crd_f32_to_bf16_rows(b.data_ptr(), 64, h.data_ptr(), 64, 0, 64, 64, None, 64, None, 0, 0, st)
        lib.crd_sigmoid_bwd(h.data_ptr(), h.data_ptr(), 4096, st)


def seq_torch():
    for i in range(N // 2):
        torch.mul(a, 1.0, out=b); torch.add(b, 1.0, out=a)


for name, fn in [("same kernel", seq_same), ("3 distinct kernels", seq_mix), ("torch mul/add", seq_torch)]:
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            fn()
        for _ in range(3): g.replay()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        print(f"{name:20s}: {e0.elapsed_time(e1) * 1e3 / N:.2f} us / kernel in graph")
