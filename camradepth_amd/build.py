"""Builds libcamradepth_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcamradepth_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result", "-Wno-inline-asm"]
# Per-file flags.  -fno-slp-vectorize: the SLP vectoriser pairs the per-channel fp32 loops of the elementwise / stencil kernels into
# v_pk_fma_f32 and keeps every unpacked operand alive to do so (k_bicubic_bwd_tile: 452 instead of 106 registers, k_bicubic 194 -> 124,
# k_dwconv_wgrad 354 -> 255: two workgroups per CU instead of one); norm.hip measured 0.17 ms per step SLOWER without it, the MFMA
# kernels are left as they were (conv3x3.hip gets scratch without it).
NO_SLP = set((os.environ.get("CRD_NOSLP_FILES") or "decoder_ops,encoder_ops").split(","))
EXTRA = (os.environ.get("CRD_EXTRA_FLAGS") or "").split()      # developer switch: e.g. -DCRD_DWW_WGS=1 for an A/B build


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def build(force=False, verbose=False):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    newest_hdr = max(os.path.getmtime(h) for h in hdrs)
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s)[:-4] + ".o")
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), newest_hdr):
            jobs.append((s, o))
    hipcc = _hipcc()

    def cc(job):
        s, o = job
        extra = ["-fno-slp-vectorize"] if os.path.basename(s)[:-4] in NO_SLP else []
        cmd = [hipcc] + FLAGS + extra + EXTRA + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{r.stderr[-4000:]}")
        return r.stderr

    with ThreadPoolExecutor(max_workers=4) as ex:
        for err in ex.map(cc, jobs):
            if verbose and err.strip():
                print(err)
    objs = [os.path.join(objdir, os.path.basename(s)[:-4] + ".o") for s in srcs]
    if jobs or not os.path.exists(LIB):
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs,
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
