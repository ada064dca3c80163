"""Execution plan of the CamRaDepth hot path on one MI355X.

A Plan is built once per (batch, height, width, train/eval): every activation, gradient and
scratch buffer is allocated up front (288 GB of HBM make recomputation pointless), every kernel
call of the forward and backward pass is recorded as (C-ABI function, ctypes arguments), and
running a pass is a flat loop over those records on the current HIP stream -- which is also what
makes the whole step capturable into a HIP graph (no allocation, no host sync inside).

PyTorch is used for memory (torch.empty / zero_) and streams only; all arithmetic is in
libcamradepth_hip.so.  Forward structure follows the reference: CamRaDepth.forward
(src/models/CamRaDepth.py:99-176), SimplifiedTransformer.forward_features
(src/models/simplified_attention.py:265-306), Block/Attention_MaxPool/Mlp (:34-43,90-109,141-145),
Decoder/ShortResBlock/ConvLayer/Depth_Activation (src/utils/utils.py:127-135,223-228,249-257,285-289).
The backward pass is hand-derived.
"""
import bisect
import ctypes as C
import os

import torch

from . import lib as L
from . import trace
from .config import MID_CHANNELS, UNSUP_CLASSES, ModelConfig
from .params import short_res_block_plan

# Tuning / A-B knobs whose verdict is recorded in DESIGN.md are CONSTANTS here; only with CRD_DEV_SWITCHES=1 (developer runs: the
# sweep and ablation scripts under tools/) are they read from the environment again.
_DEV = os.environ.get("CRD_DEV_SWITCHES") == "1"


def _dev_flag(name):
    return _DEV and os.environ.get(name) is not None


def _dev_int(name, default):
    return int(os.environ.get(name, default)) if _DEV else int(default)


# every small weight gradient as its own launch instead of the grouped launch (A/B): dev switch CRD_NO_GROUP_WGRAD
GROUP_WGRAD = not _dev_flag("CRD_NO_GROUP_WGRAD")
# Side branches (q projection, rank-one vector path, depthwise weight gradient on separate graph branches) are recorded
# only on request: measured 27.7-28.9 ms/step against 26.7 on one stream -- every fork / join costs more in the captured
# graph than the short kernels it takes off the chain.
SIDE_STREAMS = _dev_flag("CRD_SIDE_STREAMS")
# GroupNorm-backward reduce of Mlp.norm2 inside the fc2 data-gradient epilogue (crd_conv_desc.red_*): opt-in.  It removes
# a pass over (H2, d(H3)) per block but measured 26.6 vs 26.4 ms/step: the GELU' + fold + atomics in the GEMM epilogue
# cost more than the streaming reduce kernel they replace.  (The same fusion into the depthwise data gradient, for
# Mlp.norm1 without activation, does pay and is always on.)
FUSE_GN_RED = _dev_flag("CRD_FUSE_GN_RED")
# ... except on grids of <= this many pixels per sample, where the launch it saves outweighs the slower epilogue (25.2 -> 25.0 ms)
FUSE_GN_RED_MAXPIX = _dev_int("CRD_FUSE_GN_RED_MAXPIX", 416)
# ... and, as an experiment (dev switch CRD_FUSE_GN_RED_WIDE + a library built with CRD_PW_WIDE_RED), wherever the wide pointwise
# kernel (k_gn_pw_wide, round 4) runs fc2's data gradient: there the reduce's sums never leave the registers until the workgroup ends
FUSE_GN_RED_WIDE = _dev_flag("CRD_FUSE_GN_RED_WIDE")       # measured: 18.01 ms per step with it, 17.91 without (the GELU' in the epilogue)
# GroupNorm statistics of the residual stream produced by the kernels that write it (attn_out_residual -> norm2, fc2's
# epilogue -> the next block's norm1) instead of crd_gn_stats launches; CRD_NO_FUSE_STATS restores the launches
FUSE_STATS = not _dev_flag("CRD_NO_FUSE_STATS")
# The decoder's weight gradients (2.9 ms of MFMA work nothing in the backward pass waits for) can leave the chain: the
# single-GPU graph step replays them as a graph of their own on a second stream while the encoder's latency-bound backward
# runs (trainer.py; CRD_NO_LATE_WGRAD keeps program order).  W3_LATE_WGS: workgroups their streaming kernels may use in
# that mode, so that the encoder's kernels still find free CUs (256: 22.1 ms, 160: 20.7, 128: 20.9, 64: 22.3).
# GroupNorm(+GELU)-apply folded into the A-operand load of the consuming pointwise / patch GEMM (crd_gn_conv) instead of a
# crd_gn_apply launch + pass per GroupNorm of an encoder block.  CRD_GN_CONV=0 turns it off (read when a plan is built)
def gn_conv_default():
    """0 = off, 1 = q / k / fc1 / fc2, 2 = q / k / fc1 only (fc2's GroupNorm carries the exact GELU: ~25 VALU operations per
    element, which a GEMM workgroup does once per column tile next to its MFMAs)."""
    return _dev_int("CRD_GN_CONV", 2)


# Mlp of a Block as ONE launch per (sample, 64-channel hidden slab) + a reduce launch (csrc/mlp_fused.hip) wherever the
# library covers the shape (Mlp.norm2 group == 64 hidden channels and the pixel grid fits in LDS: encoder stages 3 and 4
# at 256 x 416).  CRD_MLP_FUSED=0 keeps the four launches (A/B).
MLP_FUSED = _dev_int("CRD_MLP_FUSED", 1) != 0
# ... up to this many pixels per sample.  One workgroup per (sample, slab) is hidden/64 x B workgroups: at 8 x 13 pixels
# (stage 4: 128 workgroups) the fused launch takes 18 us against ~45 us for the four it replaces; at 16 x 26 (stage 3: 80
# workgroups on 256 CUs) it is bound by the VALU work of the stencil / GELU phases on those 80 CUs -- 35 us + the 11 us
# reduce against 41 us unfused (tools/prof_mlp.py) -- so stage 3 keeps the four launches.
MLP_FUSED_MAXPIX = _dev_int("CRD_MLP_FUSED_MAXPIX", 128)
FC2_FOLD_MINROWS = _dev_int("CRD_FC2_FOLD_MINROWS", 16384)   # (B = 16 inference 10.51 -> 10.16 ms; B = 1, 8: unchanged)
GN_CONV_MAXROWS = _dev_int("CRD_GN_CONV_MAXROWS", 1 << 30)   # pixels x batch up to which a Block's GEMMs are fused
LATE_WGRAD = not _dev_flag("CRD_NO_LATE_WGRAD")
W3_LATE_WGS = _dev_int("CRD_W3_LATE_WGS", 160)
LATE = 3            # Op.stream id of those ops
SPLIT_N = not _dev_flag("CRD_NO_SPLIT_N")      # developer switch for the ragged-tile split of 3x3 data gradients
W3_PARTIALS = not _dev_flag("CRD_NO_W3_PARTIALS")   # developer switch: streaming 3x3 wgrad with atomics instead
DW_REPLICAS = 16      # accumulator copies of a depthwise weight gradient (spreads contended fp32 atomics)
HEAD_ROWS = 64        # same for Depth_Activation.conv_2 (2048 workgroups x 289 sums)

BF16, F32 = torch.bfloat16, torch.float32
SUM = torch.int64    # crd_sum_t: the 64-bit fixed-point accumulators every multi-workgroup "+=" goes through (include/camradepth_hip.h)


def rup(x, m=8):
    return (x + m - 1) // m * m


class PM:
    """Pixel-major view: element (b, p, c) at t[b, p, coff + c]; t has shape [B, H*W, ld]."""
    __slots__ = ("t", "ld", "coff", "C", "H", "W", "f32")

    def __init__(self, t, C_, H, W, coff=0):
        self.t, self.ld, self.coff, self.C, self.H, self.W = t, t.shape[-1], coff, C_, H, W
        self.f32 = 1 if t.dtype == F32 else 0

    def sl(self, c0, c1):
        return PM(self.t, c1 - c0, self.H, self.W, self.coff + c0)

    @property
    def P(self):
        return self.H * self.W

    @property
    def ptr(self):
        return self.t.data_ptr()


class Op:
    """One recorded kernel call.  stream: 0 = the main stream; 1, 2 = side branches (ops that do not depend on the main
    ops recorded after the point where the branch was opened); fn None = join marker: the main stream waits for the branch."""
    __slots__ = ("fn", "args", "name", "region", "acc_slot", "meta", "stream", "io", "cond")

    def __init__(self, fn, args, name, region=None, acc_slot=None, meta=None, stream=0, io=None, cond=None):
        self.fn, self.args, self.name, self.region, self.acc_slot, self.meta = fn, args, name, region, acc_slot, meta
        self.stream = stream
        self.cond = cond        # None, or (plan attribute, value): the op runs only while getattr(plan, attribute) == value (Plan.live)
        self.io = io            # algorithmic HBM bytes of the launch (int, or a callable evaluated after Plan._finalise): see nbytes()


def nbytes(*ts):
    """ALGORITHMIC bytes of the tensors a launch must read or write once (bench.py's floor budget, tools/floor_table.py): a PM
    counts its own C channels of every pixel, not the row stride of the buffer it is a slice of; halo re-reads, padding channels
    and cache effects are deliberately not in here -- that is what the measured traffic is compared against."""
    n = 0
    for t in ts:
        if t is None:
            continue
        if isinstance(t, PM):
            n += t.t.shape[0] * t.P * t.C * (4 if t.f32 else 2)
        elif isinstance(t, _Lazy):
            n += t.numel * 8
        elif isinstance(t, torch.Tensor):
            n += t.numel() * t.element_size()
        else:
            n += int(t)
    return n


def igemm_tile(cout, ohw=1 << 30, batch=1):
    """Tile configuration crd_conv_igemm dispatches to (csrc/igemm.hip), as the kernel's template arguments."""
    if cout > 32 and -(-ohw // 128) * -(-cout // 128) * batch < 192:
        return "k_igemm<2,2,1,1>"
    if cout <= 32:
        return "k_igemm<4,1,1,1>"
    if cout <= 64:
        return "k_igemm<2,2,2,1>"
    if cout <= 96:
        return "k_igemm<4,1,1,3>"
    if 128 < cout <= 160:
        return "k_igemm<4,1,1,5>"
    return "k_igemm<2,2,2,2>"


# Round 6: the APPLY phase of a GroupNorm backward inside the pointwise data-gradient GEMM that consumes it (crd_gn_bwd_conv,
# csrc/xfgemm.hip) instead of a crd_gn_bwd_apply launch: Mlp.norm1 in front of fc1's data gradient (GNB_FC1) and attn.norm in front of
# the sr patch scatter (GNB_SR).  Bit s of a mask = encoder stage s + 1 (developer switches CRD_GNB_FC1 / CRD_GNB_SR for the A/B).
# fc1 at stage 4 stays on the two launches: 832 rows x K = 1024 are 52 workgroups walking 32 serial K slabs in the fused kernel, 21.7 us
# against 4.4 + 10.5 (split-K crd_conv_igemm); masks 15 / 7 / 3 / 1 in one call: 17.39 / 17.33-17.37 / 17.42-17.46 / 17.51 ms per step.
# Round 6: the data gradients of a decoder stage's three ConvLayers (ShortResBlock, utils.py:127-135) as WRITE-ONCE launches over the
# K-concatenated gradient buffer [d(raw2) | d(raw1) | d(raw0)]: columns [o1, o1+64) of the concat-buffer gradient from layer 2 alone
# (K = 128 x 9), [o0, o1) from layers 2 | 1 (K = 192 x 9), [0, o0) from all three (K = 288 x 9) -- no read-modify-write of the 304-channel
# gradient, no accumulating epilogue.  Developer switch CRD_KCAT=0: the per-layer accumulating launches of rounds 1-5 (the A/B).
KCAT = _dev_int("CRD_KCAT", 1) != 0
GNB_FC1 = _dev_int("CRD_GNB_FC1", 7)
GNB_SR = _dev_int("CRD_GNB_SR", 15)
# attn.q and the attn.sr patch convolution of a Block as ONE launch (crd_gn_conv2: both read Block.norm1(x); sr normalises its own rows
# instead of waiting for q's stored copy).  Bit s = encoder stage s + 1.  Stage 3 only: measured in one call (profiles/r06_ab_q_sr_grouped.txt)
# 17.50 / 17.51 ms without, 17.46 / 17.47 with stage 3, 17.56 / 17.53 with stages 2 + 3 -- the sr convolutions of stages 1-2 (K = 4096 /
# 2048: 64 / 32 serial slabs on this path) are better off on crd_conv_igemm's 4-way split-K; the pair it saves at stage 3 was 6.6 + 8.2 us.
QSR_GROUP = _dev_int("CRD_QSR_GROUP", 4)
# (Round 6, measured and removed: the fold of crd_attn_bwd's dK partials inside k's data gradient -- a GEMM of 16-48 workgroups whose A
# loader sums 4-52 fp32 partials took 16.9 / 17.9 / 15.1 us at stages 3 / 2 / 1 against 3.2 + 8.0 for crd_sum_partials_bf16 + the GEMM:
# 17.36 ms per step with it, 17.19 without, gpurun_out/r6/ab_ps2.txt.)
FUSE_NORM2_APPLY = not _dev_flag("CRD_NO_FUSE_NORM2_APPLY")    # developer switch (A/B): Block.norm2's backward apply inside crd_attn_out_bwd
FUSE_BLOCK_RED = not _dev_flag("CRD_NO_FUSE_BLOCK_RED")    # developer switch (A/B): Block.norm1 / norm2 reduces in GEMM epilogues


def fused_reduce_tile_ok(cout, ohw, B):
    """crd_conv_igemm's rule for red_x (csrc/igemm.hip): the fused GroupNorm-backward reduce lives in the vector epilogue of the
    32 / 64 / 128-column tiles (small grids always use 64-column tiles), not the 96- and 160-column ones."""
    small = cout > 32 and -(-ohw // 128) * -(-cout // 128) * B < 192
    return FUSE_BLOCK_RED and cout % 16 == 0 and (small or cout <= 64 or 96 < cout <= 128 or cout > 160)


def persistent_conv3(spec, B):
    """Does crd_conv_igemm send this 3x3 launch to the persistent one-wave-per-SIMD kernel (csrc/conv3x3p.hip)?  Plain bf16
    store / accumulate (+ GroupNorm sums) on grids of >= 192 tiles of 16 x 32 pixels."""
    if os.environ.get("CRD_CONV3P", "1") == "0":
        return False
    y = spec["y"]
    plain = (not y.f32 and spec["bias"] is None and not spec["act"] and spec["res"] is None and spec["out_mode"] == 0
             and spec.get("red") is None and spec.get("chan") is None)
    halo = spec["k"] == 3 and spec["stride"] == 1 and spec["OW"] >= 32 and spec["OH"] >= 8
    tiles = -(-spec["OW"] // 32) * -(-spec["OH"] // 16) * B
    return bool(plain and halo and tiles >= 192 and spec["cout"] >= 64 and spec["cout"] % 8 == 0)


def halo_tile(cout, OH=1 << 20, OW=1 << 20, B=1):
    """Tile configuration of the halo-tile 3x3 kernel (csrc/conv3x3.hip: crd_conv3x3_halo), for the bench labels."""
    tiles = -(-OW // 32) * -(-OH // 8) * B
    if cout > 32 and tiles * -(-cout // 128) < 512:      # under-filled grid: 64- or 32-column tiles
        return "k_conv3x3<4,1,2,2>" if tiles * -(-cout // 64) >= 512 else "k_conv3x3<4,1,2,1>"
    if cout <= 32:
        return "k_conv3x3<4,1,2,1>"
    if cout <= 64:
        return "k_conv3x3<4,1,2,2>"
    if cout <= 96:
        return "k_conv3x3<4,1,2,3>"
    if 128 < cout <= 160 or 256 < cout <= 320:      # two launches: 128-wide tiles + the remaining columns
        return "k_conv3x3<4,1,2,4>+<4,1,2,1>" if cout - (256 if cout > 256 else 128) <= 32 else "k_conv3x3<4,1,2,4>+<4,1,2,2>"
    return "k_conv3x3<4,1,2,4>"


def wgrad_tile(cout):
    if cout <= 32:
        return "k_wgrad<1,4,2,2>"
    if cout <= 64:
        return "k_wgrad<1,4,4,2>"
    if cout <= 96:
        return "k_wgrad<2,2,3,4>"
    return "k_wgrad<2,2,4,4>"


class ConvW:
    """A dense convolution's parameters and packed forms."""

    def __init__(self, name, cout, cin_ref, k, cmap, bias, need_dgrad, scatter=False, dgrad_rows=None):
        self.name, self.cout, self.cin_ref, self.k, self.taps = name, cout, cin_ref, k, k * k
        self.cmap = cmap                      # list[int] internal channel -> reference channel (or -1), len = cin_pad
        self.cin_pad = len(cmap) if cmap is not None else rup(cin_ref)
        self.cout_pad = rup(cout)
        self.bias, self.need_dgrad, self.scatter = bias, need_dgrad, scatter
        self.identity = (cmap is None and self.cin_pad == cin_ref and self.taps == 1)
        self.w_fwd = self.w_dgrad = self.w_scatter = None   # bf16 tensors
        self.dw = None                                        # fp32 [cout][taps][cin_pad] (scratch or direct grad view)
        self.wg_budget = 0
        self.dw_parts, self.dw_S, self.stream3_geom = None, 0, None   # per-split copies of dw for the streaming 3x3 wgrad
        self.cmap_dev = None


class Plan:
    def __init__(self, model, B, H, W, training):
        cfg = model.cfg
        assert H % 32 == 0 and W % 32 == 0, "H and W must be multiples of 32 (SURVEY.md section 8 note on 900x1600)"
        self.model, self.cfg, self.B, self.H, self.W, self.training = model, cfg, B, H, W, training
        self.dev = model.flat.device
        self.lib = L.load()
        self.fwd, self.bwd_groups, self.bwd_tags = [], [], []
        self._tag = "enc0"
        self.zero_fwd, self.zero_bwd = [], []       # (numel) requests into the two zero arenas
        self._zf_views, self._zb_views = [], []
        self.convs = []
        self.keep = []                              # keep ctypes structs / tensors alive
        self.dw_entries, self.dw_grads = [], []
        self.row_grads = []                         # (param, C, rows buffer, R, tag, offset, row stride): gradient = sum of the rows
        self.kcat_entries = []                      # (ConvW, K-concatenated weight matrix, ld, column offset, first row, rows): extra pack-table entries
        self.shapes = {}
        self.fwd_marks = []
        self._defer = None
        self._cur_stream = 0
        self._side_streams = None
        self.split_late = False            # trainer: ops of stream LATE are skipped by backward() and run by run_late()
        self.attn_parts = None
        self.mlp_parts = None              # fc2 partial tiles of the fused Mlp: produced and consumed back to back
        self.stats_scratch = None
        self.buffers = []
        # parameters with requires_grad=False (the reference's optimizer skips `grad is None`, diffGradNorm.py:54-55):
        # their weight-gradient launches are not recorded and whatever fused kernels still produce for them lands in a
        # scratch buffer instead of the flat gradient
        self.gn_conv_on = gn_conv_default()
        self.enc_taps = {}
        self.block_ops = {}
        # need_grad False (plan built under torch.no_grad()): tensors only a backward pass reads are not written
        self.need_grad = bool(getattr(model, "_need_grad", True))
        self.frozen = frozenset(n for n in model._names if not model._param(n).requires_grad)
        self._dump = {}
        # fp8 (e4m3) forward of the big decoder ConvLayers (BASELINE.json config 5): inference plans only, with the
        # per-stage activation scales of model.calibrate_fp8(); {decoder stage name: scale of its concat buffer}
        f8 = getattr(model, "fp8_scales", None)
        # (training plans only when the model asks for it -- fp8_train: fp8 FORWARD convolutions, bf16 backward on the bf16
        # activations, which the producers then store next to the e4m3 copy)
        self.fp8_keep_bf16 = bool(training or self.need_grad)
        self.fp8 = dict(f8) if (f8 and (not self.fp8_keep_bf16 or getattr(model, "fp8_train", False))) else None
        self.fp8_convs = []
        self.fp8_kcat = []                 # (first layer ConvW, K-concatenated bf16 data-gradient matrix, rows, K, its e4m3 copy, per-row scales)
        # fp8 DATA GRADIENTS of the same ConvLayers (round 5; model.calibrate_fp8(x, train=True, grads=True)): per layer one device
        # float (the e4m3 scale of its dy) and CRD_FP8_AMAX_SLOTS amax slots.  fp8_jit True: just-in-time scaling -- the layer's scale
        # is set from THIS step's amax and dy re-quantised before its data gradient runs (one extra pass; eager plans and the
        # calibration iteration of TrainStep); False: delayed scaling -- the scales of the previous step's amax, updated once at the
        # head of the backward pass (TrainStep's graphs).
        self.fp8_grad = bool(self.fp8 is not None and training and getattr(model, "fp8_grad", False))
        # margin 2 (round 6): with delayed scaling a step whose gradients GROW is quantised with the previous step's amax -- one
        # binade of headroom before e4m3 saturates (clamps at +-448 x scale); e4m3's relative precision does not depend on the scale
        self.fp8_grad_layers, self.fp8_jit, self.fp8_margin = [], True, 2.0
        if self.fp8_grad:
            self.g8_scales = torch.ones(16, dtype=F32, device=self.dev)
            self.g8_amax = torch.zeros((16, 64), dtype=torch.int32, device=self.dev)
            self.buffers += [self.g8_scales, self.g8_amax]
        self._build()

    # ------------------------------------------------------------------ allocation helpers
    def new(self, shape, dtype=BF16):
        t = torch.zeros(shape, dtype=dtype, device=self.dev)
        self.buffers.append(t)                     # ops hold raw pointers only: the plan owns every buffer
        if os.environ.get("CRD_DEBUG_NAN"):
            import traceback
            fr = [f"{f.name}:{f.lineno}" for f in traceback.extract_stack()[-5:-1]]
            self.keep.append(("buf", len(self.keep), tuple(shape), str(dtype), fr, t))
        return t

    def act(self, C_, H, W, dtype=BF16, ld=None):
        return PM(self.new((self.B, H * W, ld or C_), dtype), C_, H, W)

    def zf(self, *shape):
        """crd_sum_t accumulators (GroupNorm / channel sums) that must be zero at the start of every forward."""
        v = _Lazy(shape)
        self._zf_views.append(v)
        return v

    def zb(self, *shape):
        """crd_sum_t accumulators (gradient sums) that must be zero at the start of every backward."""
        v = _Lazy(shape)
        self._zb_views.append(v)
        return v

    def _materialise(self, views):
        n = sum(v.numel for v in views)
        arena = torch.zeros(max(n, 1), dtype=SUM, device=self.dev)
        off = 0
        for v in views:
            v.t = arena[off:off + v.numel].view(v.shape)
            off += v.numel
        return arena

    # ------------------------------------------------------------------ op emitters
    def _emit(self, lst, fn_name, args, region=None, acc_slot=None, io=None):
        op = Op(getattr(self.lib, fn_name), list(args), fn_name, region, acc_slot, stream=self._cur_stream, io=io)
        lst.append(op)
        return op

    def live(self, op):
        """Is this recorded op part of the pass as the plan is configured now?  (join markers and conditional ops: the just-in-time
        / delayed-scaling variants of the fp8 gradient path are both recorded; plan.fp8_jit selects)"""
        return op.fn is not None and (op.cond is None or getattr(self, op.cond[0]) == op.cond[1])

    @staticmethod
    def op_bytes(op):
        """Algorithmic HBM bytes of a recorded launch (0 where none were recorded)."""
        io = op.io
        return int(io() if callable(io) else (io or 0))

    class _Side:
        def __init__(self, plan, s):
            self.plan, self.s = plan, s

        def __enter__(self):
            self.prev, self.plan._cur_stream = self.plan._cur_stream, self.s

        def __exit__(self, *a):
            self.plan._cur_stream = self.prev

    def side(self, s):
        """`with self.side(1): ...` records the enclosed ops on side branch s (1 or 2): they may only read what was
        produced before the first op of the branch, and nothing recorded on the main stream before the matching join()
        may touch what they write.  The step is a chain of ~1500 mostly latency-bound kernels; branches take short
        independent pieces (a weight gradient, the q projection, the rank-one attention vector path) off that chain."""
        return Plan._Side(self, s if SIDE_STREAMS else 0)

    def join(self, lst, s):
        if SIDE_STREAMS:
            lst.append(Op(None, [], "join", stream=s))

    def _push(self, grp):
        """Register the backward ops of one forward unit; units are replayed in reverse order."""
        self.bwd_groups.append(grp)
        self.bwd_tags.append(self._tag)

    def conv_desc(self, x, w_t, cout, k, stride, pad, OH, OW, y, cin=None, gather=0, out_mode=0, patch_k=0, patch_c=0,
                  bias=None, bias_bstride=0, act=0, res=None, res_scale=None, stats=None, accumulate=0, red=None, chan=None):
        """Specification of one crd_conv_igemm call; turned into a ctypes ConvDesc in _finalise."""
        return dict(x=x, w=w_t, cout=cout, k=k, stride=stride, pad=pad, OH=OH, OW=OW, y=y, cin=cin if cin is not None else x.C,
                    gather=gather, out_mode=out_mode, patch_k=patch_k, patch_c=patch_c, bias=bias, bias_bstride=bias_bstride,
                    act=act, res=res, res_scale=res_scale, stats=stats, accumulate=accumulate, red=red, chan=chan)

    def conv(self, lst, spec, region=None):
        w, x = spec["w"], spec["x"]
        if isinstance(w, tuple) and w[0] == "kcat":
            flops = spec["flops_override"]
        elif isinstance(w, tuple):     # data gradient of convolution w[1]: algorithmic work of that layer's forward
            cw = w[1]
            flops = 2.0 * self.B * x.H * x.W * cw.cout * cw.taps * min(spec["cout"] // (cw.taps if w[0] == "scatter" else 1),
                                                                         cw.cin_ref)
        else:
            flops = 2.0 * self.B * spec["OH"] * spec["OW"] * spec["cout"] * w.cin_ref * w.taps
        halo = spec["k"] == 3 and spec["stride"] == 1 and spec["out_mode"] == 0 and spec["OW"] >= 32 and spec["OH"] >= 8
        # The halo kernel's widest tile is 128 output channels.  A data gradient with N = 144 / 296 / 304 would run its last
        # tile 13-37 % full; the ragged tail goes to a second launch with a narrower tile instead (same operands, weight
        # rows and output slice offset by the split point).
        cout = spec["cout"]
        covered = cout <= 160 or 256 < cout <= 320          # the library's 160-column tile handles these without a ragged tail
        if (halo and SPLIT_N and cout > 128 and 0 < cout % 128 <= 64 and not covered and spec["bias"] is None
                and spec["stats"] is None and spec["res"] is None and "w_row0" not in spec):
            c_main = cout // 128 * 128
            for c0, c1 in ((0, c_main), (c_main, cout)):
                sub = dict(spec)
                sub.update(cout=c1 - c0, y=spec["y"].sl(c0, c1), w_row0=c0, flops_override=flops * (c1 - c0) / cout)
                self.conv(lst, sub, None if region is None else region[:-2] + (region[-2] + c0, region[-2] + c1))
            return None
        flops = spec.get("flops_override", flops)
        kname = halo_tile(spec["cout"], spec["OH"], spec["OW"], self.B) if halo else igemm_tile(spec["cout"], spec["OH"] * spec["OW"], self.B)
        if self._narrow_takes(spec, False):
            kname = "k_pw_narrow"
        if halo and persistent_conv3(spec, self.B):
            c = spec["cout"]
            rest = c % 128
            kname = "k_conv3x3p<2>" if c <= 64 else "k_conv3x3p<3>" if c <= 96 else "k_conv3x3p<4>" + ("" if rest == 0 or rest > 64 else
                                                                                                 "+k_conv3x3<4,1,2,1>" if rest <= 32 else "+k_conv3x3<4,1,2,2>")
        meta = {"kernel": kname, "flops": flops,
                "shape": f"{('dgrad-kcat' if w[0] == 'kcat' else 'dgrad') if isinstance(w, tuple) else 'fwd'} Cin{spec['cin']} Cout{spec['cout']} k{spec['k']} s{spec['stride']} "
                         f"out{spec['OH']}x{spec['OW']}"}
        wbytes = spec["cout"] * spec["k"] * spec["k"] * spec["cin"] * 2

        def io(spec=spec, wbytes=wbytes):        # (accumulate flags of gradient regions are only known after _finalise)
            xs, ys = spec["x"], spec["y"]
            n = xs.t.shape[0] * xs.P * spec["cin"] * (4 if xs.f32 else 2) + wbytes
            n += nbytes(ys) * (2 if spec["accumulate"] else 1) + nbytes(spec["res"])
            if spec.get("red") is not None:
                n += nbytes(spec["red"][0])
            return n
        op = Op(self.lib.crd_conv_igemm, [spec], "crd_conv_igemm", region, ("spec", spec) if region else None, meta,
                stream=self._cur_stream, io=io)
        lst.append(op)
        return op

    def _narrow_takes(self, spec, with_gn):
        """Label only (bench.py / tools): does the library send this 1x1 launch to the narrow streaming kernel (csrc/pw_narrow.hip:
        crd_pw_narrow_applicable)?  K = 1024 only behind a GroupNorm (the generic tiles are faster on rows that are already activated)."""
        y = spec["y"]
        if not (spec["k"] == 1 and spec["stride"] == 1 and spec["out_mode"] == 0 and not spec["act"] and not spec["accumulate"]
                and (with_gn or spec["cin"] == 512)
                and int(self.lib.crd_pw_narrow_supported(spec["cin"], spec["cout"], spec["OH"] * spec["OW"])) > 0):
            return False
        if y.f32:
            return spec.get("red") is None
        red = spec.get("red")
        return spec["res"] is None and spec["stats"] is None and spec.get("chan") is None and (red is None or red[5] == 0)

    def gn_conv(self, lst, spec, stats, gmul, gname, act, xn):
        """spec as for conv() with x the RAW (un-normalised) tensor: the GroupNorm `gname` (+ GELU when act) is applied while
        the GEMM loads its A operand (crd_gn_conv); xn: PM that also receives the normalised bf16 tensor, or None."""
        w, x = spec["w"], spec["x"]
        assert x.coff == 0 and (xn is None or xn.coff == 0)
        flops = 2.0 * self.B * spec["OH"] * spec["OW"] * spec["cout"] * w.cin_ref * w.taps
        small = spec["cout"] <= 64 or -(-spec["OH"] * spec["OW"] // 64) * -(-spec["cout"] // 128) * self.B < 256
        kname = "k_gngemm_reg" + ("<2,2,1,1>" if small else "<2,2,1,2>")
        if act and not x.f32 and self._narrow_takes(spec, True):
            kname = "k_pw_narrow<gn+gelu>"
        meta = {"kernel": kname, "flops": flops,
                "shape": f"gn+fwd Cin{spec['cin']} Cout{spec['cout']} k{spec['k']} s{spec['stride']} out{spec['OH']}x{spec['OW']}"}
        gn = dict(gn_in=True, x_f32=x.f32, gmul=gmul, stats=stats, gamma=self.p(gname + ".weight"), beta=self.p(gname + ".bias"),
                  act=act, xn=xn)
        io = nbytes(x, spec["y"], xn, spec["res"]) + spec["cout"] * spec["cin"] * 2
        op = Op(self.lib.crd_gn_conv, [spec, gn], "crd_gn_conv", None, None, meta, stream=self._cur_stream, io=io)
        lst.append(op)
        return op

    def gn_bwd_conv(self, lst, spec, gx, stats, gmul, gname, act, r, dx, mask=None):
        """spec as for conv() with x = dy of the GroupNorm `gname` (its raw input gx, forward sums `stats`, reduce sums r): the apply phase of
        that GroupNorm's backward runs while the pointwise data-gradient GEMM loads its A operand (crd_gn_bwd_conv); dx: PM that also
        receives the applied gradient (the weight gradients read it), or None."""
        w, x = spec["w"], spec["x"]
        cw = w[1]
        assert isinstance(w, tuple) and x.coff == 0 and gx.coff == 0 and (dx is None or dx.coff == 0) and spec["k"] == 1
        flops = 2.0 * self.B * x.H * x.W * cw.cout * cw.taps * min(spec["cout"] // (cw.taps if w[0] == "scatter" else 1), cw.cin_ref)
        small = spec["cout"] <= 64 or -(-spec["OH"] * spec["OW"] // 64) * -(-spec["cout"] // 128) * self.B < 256 or spec["out_mode"] == 1
        kname = "k_gnbwd_gemm" + ("<1>" if small else "<2>")
        meta = {"kernel": kname, "flops": flops,
                "shape": f"gnbwd+{'scatter' if w[0] == 'scatter' else 'dgrad'} Cin{spec['cin']} Cout{spec['cout']} out{spec['OH']}x{spec['OW']}"}
        frozen = self.is_frozen(gname + ".weight", gname + ".bias")
        gnb = dict(gnb_in=True, gx=gx, gmul=gmul, act=act, stats=stats, gamma=self.p(gname + ".weight"), beta=self.p(gname + ".bias"),
                   mask=mask, r=r, dx=dx, dgamma=None if frozen else self.g(gname + ".weight"), dbeta=None if frozen else self.g(gname + ".bias"))

        def io(spec=spec, x=x, gx=gx, dx=dx):
            n = nbytes(x, gx, dx) + spec["cout"] * spec["cin"] * 2
            n += nbytes(spec["y"]) * (2 if spec["accumulate"] else 1)
            if spec.get("red") is not None:
                n += nbytes(spec["red"][0])
            return n
        op = Op(self.lib.crd_gn_bwd_conv, [spec, gnb], "crd_gn_bwd_conv", None, None, meta, stream=self._cur_stream, io=io)
        lst.append(op)
        return op

    def wgrad(self, lst, x, dy, cw, k, stride, pad, OH, OW, dbias=None, cin=None):
        """dbias: NAME of the bias parameter.  Weight and bias gradients are accumulated as crd_sum_t (order-independent
        integer atomics) in backward scratch; the segment's crd_wgrad_unpack converts and adds them into the flat gradient."""
        if cw.frozen:                 # no trainable parameter behind this launch
            return
        if dbias is not None:
            rows = self.zb(cw.cout)
            self.row_grads.append((dbias, cw.cout, rows, 1, cw.tag, 0, cw.cout))
            dbias = rows
        spec = dict(wg=True, x=x, dy=dy, cw=cw, k=k, stride=stride, pad=pad, OH=OH, OW=OW, dbias=dbias,
                    cin=cin if cin is not None else x.C)
        stream3 = k == 3 and stride == 1 and OW >= 32 and OH >= 8
        if stream3:
            cw.stream3_geom = (x.H, x.W, spec["cin"])
        kname = ("k_wgrad3x3<2,4,1>" if cw.cout <= 32 else "k_wgrad3x3<2,4,2>" if cw.cout <= 64 else "k_wgrad3x3<2,4,3>" if cw.cout <= 96 else "k_wgrad3x3<2,4,4>") if stream3 \
            else wgrad_tile(cw.cout)
        meta = {"kernel": kname, "flops": 2.0 * self.B * OH * OW * cw.cout * cw.cin_ref * cw.taps, "param": cw.name,
                "shape": f"wgrad Cin{spec['cin']} Cout{cw.cout} k{k} s{stride} out{OH}x{OW}"}
        meta["bytes"] = x.t.shape[0] * x.P * spec["cin"] * (4 if x.f32 else 2) + nbytes(dy) + cw.cout * cw.taps * spec["cin"] * 4
        if self._defer is not None and not stream3:
            self._defer.append((spec, meta))      # runs in the segment's grouped launch (see flush_deferred)
            return
        stream = LATE if self._cur_stream == 0 else self._cur_stream     # nothing in the backward pass waits for a weight gradient
        lst.append(Op(self.lib.crd_conv_wgrad, [spec], "crd_conv_wgrad", meta=meta, stream=stream, io=meta["bytes"]))

    def flush_deferred(self, lst):
        """Emit ONE grouped weight-gradient launch for every wgrad deferred since `self._defer = []` (the small GEMMs
        of a stage's blocks).  Their dy operands are per-block buffers, so they are all still valid here."""
        items, self._defer = self._defer, None
        if not items:
            return
        meta = {"kernel": "k_wgrad_grouped", "flops": sum(m["flops"] for _, m in items), "shape": f"{len(items)} wgrads",
                "params": [m["param"] for _, m in items]}
        lst.append(Op(self.lib.crd_conv_wgrad_grouped, [{"wg_group": [sp for sp, _ in items]}], "crd_conv_wgrad_grouped", meta=meta,
                      stream=LATE, io=sum(m["bytes"] for _, m in items)))

    def _make_group(self, specs):
        descs = (L.WgradDesc * len(specs))()
        for i, sp in enumerate(specs):
            self._make_desc(sp, into=descs[i])
        info = L.WgradGroupInfo()
        L.check(self.lib.crd_wgrad_group_build(descs, len(specs), None, 0, C.byref(info)), "crd_wgrad_group_build")
        host = (C.c_uint8 * info.bytes)()
        L.check(self.lib.crd_wgrad_group_build(descs, len(specs), host, info.bytes, C.byref(info)), "crd_wgrad_group_build")
        table = torch.frombuffer(bytearray(bytes(host)), dtype=torch.uint8).to(self.dev)
        self.buffers.append(table)
        self.keep.append(info)
        return [table.data_ptr(), C.byref(info)]

    def _make_desc(self, sp, into=None):
        def P(v):
            if v is None:
                return None
            if isinstance(v, PM):
                return v.t.data_ptr()
            return v.data_ptr()
        if sp.get("gn_in"):
            n = L.GnInput()
            n.x_f32, n.gmul, n.act = sp["x_f32"], sp["gmul"], sp["act"]
            n.stats, n.gamma, n.beta = P(sp["stats"]), P(sp["gamma"]), P(sp["beta"])
            n.xn, n.xn_ld = (P(sp["xn"]), sp["xn"].ld) if sp["xn"] is not None else (None, 0)
            self.keep.append(n)
            return C.byref(n)
        if sp.get("gnb_in"):
            n = L.GnBwdInput()
            gx = sp["gx"]
            n.gx, n.gx_f32, n.gx_ld, n.gmul, n.act = P(gx), gx.f32, gx.ld, sp["gmul"], sp["act"]
            n.stats, n.gamma, n.beta, n.mask, n.r = P(sp["stats"]), P(sp["gamma"]), P(sp["beta"]), P(sp["mask"]), P(sp["r"])
            n.dx, n.dx_ld = (P(sp["dx"]), sp["dx"].ld) if sp["dx"] is not None else (None, 0)
            n.dgamma, n.dbeta = P(sp["dgamma"]), P(sp["dbeta"])
            self.keep.append(n)
            return C.byref(n)
        if sp.get("mlp"):
            d = L.MlpDesc()
            for k_, v in sp["ptrs"].items():
                setattr(d, k_, P(v))
            d.B, d.H, d.W, d.C, d.hidden = sp["dims"]
            self.keep.append(d)
            return C.byref(d)
        if sp.get("fp8d"):
            y = sp["y"]
            d = L.ConvDesc()
            d.x, d.x_ld, d.x_coff, d.B, d.IH, d.IW, d.Cin = sp["x8"].data_ptr(), sp["x8_ld"], 0, self.B, sp["H"], sp["W"], sp["cin"]
            d.w, d.Cout, d.KH, d.KW, d.stride, d.pad, d.OH, d.OW = sp["w8"].data_ptr(), sp["cout"], 3, 3, 1, 1, sp["H"], sp["W"]
            d.gather_mode, d.y, d.y_ld, d.y_coff, d.accumulate = 1, P(y), y.ld, y.coff, sp["accumulate"]
            self.keep.append(d)
            return C.byref(d)
        if sp.get("fp8"):
            cw, y = sp["cw"], sp["y"]
            d = L.ConvDesc()
            d.x, d.x_ld, d.x_coff, d.B, d.IH, d.IW, d.Cin = sp["x8"].data_ptr(), sp["x8_ld"], 0, self.B, sp["H"], sp["W"], sp["cin"]
            d.w, d.Cout, d.KH, d.KW, d.stride, d.pad, d.OH, d.OW = cw.w8.data_ptr(), cw.cout, 3, 3, 1, 1, sp["H"], sp["W"]
            d.y, d.y_ld, d.y_coff = P(y), y.ld, y.coff
            need = self.B * -(-sp["W"] // 32) * -(-sp["H"] // 16) * 4 * (cw.cout // 16) * 2
            if self.stats_scratch is None or self.stats_scratch.numel() < need:
                self.stats_scratch = self.new((need,), F32)
            d.stats, d.stats_partial, d.stats_partial_capacity = P(sp["stats"]), self.stats_scratch.data_ptr(), self.stats_scratch.numel()
            self.keep.append(d)
            return C.byref(d)
        if sp.get("wg"):
            x, dy, cw = sp["x"], sp["dy"], sp["cw"]
            d = into if into is not None else L.WgradDesc()
            d.x, d.x_ld, d.x_coff, d.B, d.IH, d.IW, d.Cin = P(x), x.ld, x.coff, self.B, x.H, x.W, sp["cin"]
            d.dy, d.dy_ld, d.dy_coff, d.OH, d.OW, d.Cout = P(dy), dy.ld, dy.coff, sp["OH"], sp["OW"], cw.cout
            d.KH, d.KW, d.stride, d.pad = sp["k"], sp["k"], sp["stride"], sp["pad"]
            d.dw, d.dbias = P(cw.dw), P(sp["dbias"])
            if cw.dw_parts is not None:
                d.dw_partials, d.dw_partial_capacity, d.wg_budget = cw.dw_parts.data_ptr(), cw.dw_S, cw.wg_budget
        else:
            x, y, w = sp["x"], sp["y"], sp["w"]
            if isinstance(w, tuple) and w[0] == "kcat":
                w = w[2]
            elif isinstance(w, tuple):
                w = w[1].w_dgrad if w[0] == "dgrad" else w[1].w_scatter
                assert w is not None, "packed data-gradient weights were not requested for this convolution"
            elif isinstance(w, ConvW):
                w = w.w_fwd
            d = L.ConvDesc()
            d.x, d.x_ld, d.x_coff, d.B, d.IH, d.IW, d.Cin = P(x), x.ld, x.coff, self.B, x.H, x.W, sp["cin"]
            d.w, d.Cout, d.KH, d.KW, d.stride, d.pad = P(w), sp["cout"], sp["k"], sp["k"], sp["stride"], sp["pad"]
            if sp.get("w_row0"):                          # output-channel sub-range: skip the packed weight rows before it
                d.w += sp["w_row0"] * sp["k"] * sp["k"] * sp["cin"] * 2
            d.OH, d.OW, d.gather_mode = sp["OH"], sp["OW"], sp["gather"]
            d.y, d.y_ld, d.y_coff, d.y_f32 = P(y), y.ld, y.coff, y.f32
            d.out_mode, d.patch_k, d.patch_c = sp["out_mode"], sp["patch_k"], sp["patch_c"]
            d.bias, d.bias_bstride, d.act = P(sp["bias"]), sp["bias_bstride"], sp["act"]
            res = sp["res"]
            d.res, d.res_ld, d.res_scale = P(res), (res.ld if res is not None else 0), P(sp["res_scale"])
            d.accumulate, d.stats = sp["accumulate"], P(sp["stats"])
            if sp.get("stats") is not None and persistent_conv3(sp, self.B):
                # GroupNorm sums of the persistent 3x3 kernel: per-(tile, wave) partial rows + a finalize launch
                need = self.B * -(-sp["OW"] // 32) * -(-sp["OH"] // 16) * 8 * (sp["cout"] // 16) * 2
                if self.stats_scratch is None or self.stats_scratch.numel() < need:
                    self.stats_scratch = self.new((need,), F32)      # produced and consumed inside one crd_conv_igemm call
                d.stats_partial, d.stats_partial_capacity = self.stats_scratch.data_ptr(), self.stats_scratch.numel()
            d.chan_sums = P(sp.get("chan"))
            if sp.get("red") is not None:        # fused reduce phase of the GroupNorm backward this output feeds
                rx, rstats, rgamma, rbeta, rgmul, ract, rr = sp["red"]
                assert rx.coff == 0, "the fused reduce reads the GroupNorm input from channel 0"
                d.red_x, d.red_x_ld, d.red_gmul, d.red_act, d.red_x_f32 = P(rx), rx.ld, rgmul, ract, rx.f32
                d.red_stats, d.red_gamma, d.red_beta, d.red_r = P(rstats), P(rgamma), P(rbeta), P(rr)
            # stats_partial stays NULL: workgroup-level sums go in with one fp32 atomic each.  The library's deterministic
            # partial-store + finalize path measured the same step time (34.6 vs 34.9 ms) and costs 261 more dispatches.
        self.keep.append(d)
        return C.byref(d)

    # parameter access: fp32 views into the flat parameter / gradient buffers
    def p(self, name):
        return self.model.param_view(name)

    def g(self, name):
        if name in self.frozen:
            if name not in self._dump:
                self._dump[name] = self.new(tuple(self.model._param(name).shape), F32)
            return self._dump[name]
        return self.model.grad_view(name)

    def is_frozen(self, *names):
        """True when every existing parameter among `names` is frozen."""
        return all(n in self.frozen for n in names if self.model.has_param(n))

    def new_conv(self, name, cmap=None, need_dgrad=True, scatter=False):
        w = self.p(name + ".weight")
        cout, cin_ref = w.shape[0], w.shape[1]
        k = w.shape[2] if w.dim() == 4 else 1
        bias = self.p(name + ".bias") if self.model.has_param(name + ".bias") else None
        cw = ConvW(name, cout, cin_ref, k, cmap, bias, need_dgrad, scatter)
        cw.frozen = self.is_frozen(name + ".weight", name + ".bias")
        cw.tag = self._tag
        self.convs.append(cw)
        return cw

    # ------------------------------------------------------------------ building blocks
    def gn_fwd(self, x, stats, gmul, gname, act, mask, y):
        op = self._emit(self.fwd, "crd_gn_apply", [x.t, x.f32, x.ld, x.coff, self.B, x.P, x.C, stats, gmul,
                                                   self.p(gname + ".weight"), self.p(gname + ".bias"), act, mask, y.t, y.f32, y.ld,
                                                   y.coff], io=nbytes(x, y))
        op.meta = None
        self.shapes[id(op)] = f"P{x.P} C{x.C} xf32={x.f32}"

    def gn_stats_apply(self, x, stats, chan, gmul, gname, act, mask, y):
        """GroupNorm whose statistics no producer epilogue supplies (norm1 / norm2 on the residual stream)."""
        self._emit(self.fwd, "crd_gn_stats", [x.t, x.f32, x.ld, x.coff, self.B, x.P, x.C, stats, chan], io=nbytes(x))
        self.gn_fwd(x, stats, gmul, gname, act, mask, y)

    def gn_bwd(self, grp, x, stats, gmul, gname, act, mask, dy, dx, region=None, dx_acc=0, r=None, dx2=None, scale2=None):
        """r given: the reduce phase was fused into the kernel that produced dy (only the apply phase is emitted).
        dx2: PM that also receives bf16(scale2[b] * dx)."""
        common = [x.t, x.f32, x.ld, x.coff, dy.t, dy.f32, dy.ld, dy.coff, self.B, x.P, x.C, stats, gmul,
                  self.p(gname + ".weight"), self.p(gname + ".bias"), act, mask]
        if r is None:
            r = self.zb(self.B * x.C * 2 + self.B * (x.C // (16 * gmul)) * 2)
            self._emit(grp, "crd_gn_bwd_reduce", common + [r, None, 0], io=nbytes(x, dy))
        args = common + [r, self.g(gname + ".weight"), self.g(gname + ".bias"), dx.t, dx.f32, dx.ld, dx.coff, dx_acc]
        acc_idx = len(args) - 1
        assert dx2 is None or dx2.coff == 0
        args += [dx2.t if dx2 is not None else None, dx2.ld if dx2 is not None else 0, scale2]
        op = self._emit(grp, "crd_gn_bwd_apply", args, region, acc_idx if region else None)
        op.io = lambda op=op, i=acc_idx, x=x, dy=dy, dx=dx, dx2=dx2: nbytes(x, dy, dx, dx2) + (nbytes(dx) if op.args[i] else 0)

    def conv_layer(self, name, x, k, out, mask=None, dout=None, dx=None, dx_region=None, x8=None, out8=None, kcat=None):
        """ConvLayer (utils.py:210-228): conv(no bias) -> GN(Cout/16) -> GELU [-> Dropout2d mask].
        x: input PM (Cin = x.C incl. padding, cmap given by caller through self._cmap), out: PM slice
        to write; dout: PM slice holding d(out); dx: PM slice receiving d(x)."""
        # kcat (round 6, decoder stages): dict(draw=slice of the stage's K-concatenated d(raw) buffer this layer's GroupNorm backward writes,
        # packs=[(matrix, ld, column offset, first row, rows)]: where this layer's data-gradient weights go, launch=(x = the buffer's K
        # prefix, matrix, n0, n1, dcb): the write-once data gradient of concat-gradient columns [n0, n1) issued behind this layer)
        cw = self.new_conv(name + ".model.0", cmap=self._cmap, need_dgrad=dx is not None and kcat is None)
        H, W = x.H, x.W
        raw = self.act(cw.cout, H, W)
        stats = self.zf(self.B, cw.cout // 16, 2)
        if x8 is not None:
            # fp8 route: e4m3 activations x8 = (tensor, ld, scale) covering the same channels as x, weights quantised per
            # output channel after every weight pack (forward()), fp32 accumulation, bf16 raw output
            assert k == 3 and x.coff == 0
            cin16 = rup(cw.cin_pad, 16)
            cw.w8 = self.new((cw.cout, 9, cin16), torch.uint8)
            cw.w8_scales = self.new((cw.cout,), F32)
            self.fp8_convs.append((cw, cin16))
            spec = dict(fp8=True, x8=x8[0], x8_ld=x8[1], cin=cin16, H=H, W=W, cw=cw, y=raw, stats=stats, cout=cw.cout)
            meta = {"kernel": "k_conv3x3_fp8<%d>" % (2 if cw.cout <= 64 else 3 if cw.cout <= 96 else 4),
                    "flops": 2.0 * self.B * H * W * cw.cout * cw.cin_ref * 9, "shape": f"fwd fp8 Cin{cin16} Cout{cw.cout} k3 s1 out{H}x{W}"}
            self.fwd.append(Op(self.lib.crd_conv3x3_fp8, [spec, cw.w8_scales, float(x8[2])], "crd_conv3x3_fp8", meta=meta,
                               io=self.B * H * W * cin16 + nbytes(raw) + cw.cout * 9 * cin16))
        else:
            self.conv(self.fwd, self.conv_desc(x, cw, cw.cout, k, 1, k // 2, H, W, raw, stats=stats))
        if out8 is not None:       # the output feeds fp8 convolutions: its e4m3 copy (inference: all that is stored)
            yb = [out.t, out.ld, out.coff] if out is not None else [None, 0, 0]
            op = self._emit(self.fwd, "crd_gn_apply_fp8", [raw.t, 0, raw.ld, raw.coff, self.B, raw.P, raw.C, stats, 1,
                                                           self.p(name + ".model.1.weight"), self.p(name + ".model.1.bias"), 1, mask,
                                                           out8[0], out8[1], out8[2], float(out8[3])] + yb)
            op.meta = None
            op.io = nbytes(raw, out) + raw.t.shape[0] * raw.P * raw.C
            self.shapes[id(op)] = f"P{raw.P} C{raw.C} -> fp8"
        else:
            self.gn_fwd(raw, stats, 1, name + ".model.1", 1, mask, out)
        if dout is None:
            return
        grp = []
        draw = kcat["draw"] if kcat is not None else self.act(cw.cout, H, W)
        f8g = kcat.get("f8") if kcat is not None else None
        if f8g is not None:
            # Config 5 (round 6): ALL the stage's data gradients in e4m3 on the K-concatenated buffer.  This layer's GroupNorm backward writes
            # its bf16 slice of d(raw) (the weight gradient reads it) AND the e4m3 copy of that slice, quantised with the STAGE's scale (one
            # device float for the three slices: a launch multiplies its accumulators by ONE activation scale); the write-once launch behind
            # it reads the e4m3 K prefix.  Round 5 ran the e4m3 kernel on the first-writer layer only: the accumulating 64 / 96-channel
            # launches were bound by the read-modify-write of the concat gradient, which the write-once form removes.
            # Scaling: delayed (TrainStep's graphs) -- the previous step's max over the three slices, updated once at the head of the
            # backward pass; just-in-time (eager plans, the calibration iteration) -- the running max of the slices produced so far, and the
            # whole K prefix re-quantised with it before each launch.
            draw8, ld8, sc_ptr, am_ptr = f8g["draw8"], f8g["ld8"], f8g["sc_ptr"], f8g["am_ptr"]
            xcat, Wt, n0, n1, dcb, real = kcat["launch"]
            K = xcat.C
            gname = name + ".model.1"
            common = [raw.t, raw.f32, raw.ld, raw.coff, dout.t, dout.f32, dout.ld, dout.coff, self.B, raw.P, raw.C, stats, 1,
                      self.p(gname + ".weight"), self.p(gname + ".bias"), 1, mask]
            r = self.zb(self.B * raw.C * 2 + self.B * (raw.C // 16) * 2)
            self._emit(grp, "crd_gn_bwd_reduce", common + [r, None, 0], io=nbytes(raw, dout))
            self._emit(grp, "crd_gn_bwd_apply_fp8", common + [r, self.g(gname + ".weight"), self.g(gname + ".bias"), draw.t, draw.ld, draw.coff,
                                                              draw8, ld8, draw.coff, sc_ptr, am_ptr], io=nbytes(raw, dout, draw) + self.B * H * W * cw.cout)
            self._emit(grp, "crd_fp8_scale_update", [am_ptr, sc_ptr, 1, self.fp8_margin, f8g["keep"]], io=256).cond = ("fp8_jit", True)
            self._emit(grp, "crd_quant_fp8_dev", [xcat.t, self.B * H * W, xcat.ld, 0, K, draw8, ld8, 0, sc_ptr],
                       io=nbytes(xcat) + self.B * H * W * K).cond = ("fp8_jit", True)
            self.wgrad(grp, x, draw, cw, k, 1, k // 2, H, W)
            for (Wt_, ld, coff, row0, rows) in kcat["packs"]:
                self.kcat_entries.append((cw, Wt_, ld, coff, row0, rows))
            W8 = self.new((n1 - n0, 9, K), torch.uint8)
            W8s = self.new((n1 - n0,), F32)
            self.fp8_kcat.append((cw, Wt, n1 - n0, K, W8, W8s))
            spec = dict(fp8d=True, x8=draw8, x8_ld=ld8, cin=K, H=H, W=W, w8=W8, y=dcb.sl(n0, n1), cout=n1 - n0, accumulate=0)
            meta = {"kernel": "k_conv3x3_fp8<dgrad>", "flops": 2.0 * self.B * H * W * 9 * real,
                    "shape": f"dgrad-kcat fp8 Cin{K} Cout{n1 - n0} k3 s1 out{H}x{W}"}
            grp.append(Op(self.lib.crd_conv3x3_fp8_dgrad, [spec, W8s, sc_ptr], "crd_conv3x3_fp8_dgrad", ("dcb", id(dcb.t), n0, n1),
                          ("spec", spec), meta, io=self.B * H * W * K + (n1 - n0) * 9 * K + nbytes(dcb.sl(n0, n1))))
            self._push(grp)
            return
        self.gn_bwd(grp, raw, stats, 1, name + ".model.1", 1, mask, dout, draw)
        self.wgrad(grp, x, draw, cw, k, 1, k // 2, H, W)
        if kcat is not None:
            for (Wt, ld, coff, row0, rows) in kcat["packs"]:
                self.kcat_entries.append((cw, Wt, ld, coff, row0, rows))
            xcat, Wt, n0, n1, dcb, real = kcat["launch"]
            spec = self.conv_desc(xcat, ("kcat", cw, Wt), n1 - n0, k, 1, k // 2, H, W, dcb.sl(n0, n1), gather=1)
            spec["flops_override"] = 2.0 * self.B * H * W * 9 * real
            self.conv(grp, spec, region=("dcb", id(dcb.t), n0, n1))
        elif dx is not None:
            self.conv(grp, self.conv_desc(draw, ("dgrad", cw), dx.C, k, 1, k // 2, H, W, dx, gather=1), region=dx_region)
        self._push(grp)

    # ------------------------------------------------------------------ the model
    def _build(self):
        cfg, B, H, W = self.cfg, self.B, self.H, self.W
        m = self.model
        tr = self.training
        nblk = sum(cfg.depths)
        seg = cfg.supervised_seg or cfg.unsupervised_seg
        n_drop = 5 + (2 if seg else 0)
        # train-mode masks (device, regenerated per forward unless injected)
        self.dp_masks = self.new((nblk, B), F32) if tr else None
        self.d2_masks = self.new((n_drop, B, MID_CHANNELS), F32) if tr else None
        if tr:
            self.dp_keep = torch.tensor([1.0 - r for r in cfg.drop_path_rates], dtype=F32, device=self.dev)
            self.d2_keep = torch.full((n_drop * B,), 0.8, dtype=F32, device=self.dev)
            self.rng_counter = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self._drop_i = 0

        def dmask():
            i = self._drop_i
            self._drop_i += 1
            return self.d2_masks[i] if tr else None

        Cin = cfg.input_channels
        # ---- input: NCHW fp32 -> pixel-major bf16 (8 channels) ----
        self.x_in = torch.zeros((B, Cin, H, W), dtype=F32, device=self.dev)   # static input (graph-safe)
        X8 = self.act(8, H, W)
        self._emit(self.fwd, "crd_nchw_to_pm", [self.x_in, B, Cin, H, W, X8.t, X8.ld, 0, 8], io=nbytes(self.x_in, X8))

        # ---- encoder ----
        enc_out_b = []          # bf16 copies of the four stage outputs
        d_enc_out = []          # fp32 gradients of the four stage outputs
        src, bi = X8, 0
        for s in range(4):
            self._tag = f"enc{s}"
            self.fwd_marks.append((f"enc{s}", len(self.fwd)))
            Cs, heads, ratio, sr = cfg.dims[s], cfg.heads[s], cfg.ff_expansion[s], cfg.reduction_ratio[s]
            k, stride = (7, 4) if s == 0 else (3, 2)
            Hs, Ws = src.H // stride, src.W // stride
            N = Hs * Ws
            pe = f"dest_encoder.patch_embed{s + 1}"
            cmap = list(range(Cin)) + [-1] * (8 - Cin) if s == 0 else None
            cw = self.new_conv(pe + ".proj", cmap=cmap, need_dgrad=s > 0)
            raw = self.act(Cs, Hs, Ws)
            st = self.zf(B, Cs // 16, 2)
            self.conv(self.fwd, self.conv_desc(src, cw, Cs, k, stride, k // 2, Hs, Ws, raw, bias=cw.bias, stats=st))
            X = self.act(Cs, Hs, Ws, F32)
            self.gn_fwd(raw, st, 1, pe + ".norm", 0, None, X)
            DX = self.act(Cs, Hs, Ws, F32)       # running residual-stream gradient of this stage
            grp = []
            draw = self.act(Cs, Hs, Ws)
            self.gn_bwd(grp, raw, st, 1, pe + ".norm", 0, None, DX, draw)
            # (round 5: the patch embed's own weight gradient rides in the stage's grouped launch as well -- alone it was 25-39 us
            # of split-K atomics on the late stream; its operands, the previous stage's output and `draw`, outlive the stage)
            self._defer = [] if GROUP_WGRAD else None
            self.wgrad(grp, src, draw, cw, k, stride, k // 2, Hs, Ws, dbias=pe + ".proj.bias")
            if s > 0:
                self.conv(grp, self.conv_desc(draw, ("dgrad", cw), src.C, k, stride, k // 2, src.H, src.W, d_enc_out[s - 1],
                                              gather=1), region=("dxs", s - 1, 0, src.C))
            self._push(grp)
            # per-stage scratch shared by all blocks of the stage
            hid = Cs * ratio
            sc = {"DH": self.act(Cs, Hs, Ws), "DHID": self.act(hid, Hs, Ws), "DHID2": self.act(hid, Hs, Ws),
                  "DXN": self.act(Cs, Hs, Ws), "DQ": self.act(Cs, Hs, Ws), "hid": hid}
            pre = dh = None
            for i in range(cfg.depths[s]):
                X, pre, dh = self.block(f"dest_encoder.block{s + 1}.{i}", X, DX, Cs, heads, ratio, sr, Hs, Ws, bi, sc,
                                        pre=pre, want_next=i + 1 < cfg.depths[s], dh_prev=dh)
                bi += 1
            self.flush_deferred(grp)       # grp (patch embed) is the LAST backward unit of this stage
            Xb = self.act(Cs, Hs, Ws)
            self._emit(self.fwd, "crd_f32_to_bf16_rows", [X.t, Cs, Xb.t, Cs, 0, B * N, Cs, None, 1, None, 0, 0], io=nbytes(X, Xb))
            enc_out_b.append(Xb)
            d_enc_out.append(DX)
            src = Xb

        # ---- decoder ----
        self._tag = "dec"
        self.fwd_marks.append(("dec", len(self.fwd)))
        d = cfg.dims
        hs = [(H // 32, W // 32), (H // 16, W // 16), (H // 8, W // 8), (H // 4, W // 4), (H // 2, W // 2), (H, W)]
        # from_encoder_1 -> own buffer E1 (bicubic source of stage 0); from_encoder_2..4 write into skip slices
        cup = [d[3], MID_CHANNELS, MID_CHANNELS, MID_CHANNELS + 1, MID_CHANNELS + 1]
        cskip = [d[2], d[1], d[0], 0, Cin]
        CB, dCB, lay = [], [], []
        for j in range(5):
            up_p, sk_p = rup(cup[j]), rup(cskip[j])
            ld = up_p + sk_p + 96 + 64
            Hj, Wj = hs[j + 1]
            CB.append(self.act(ld, Hj, Wj))
            dCB.append(self.act(ld, Hj, Wj))
            lay.append((up_p, sk_p))
        self._cmap = None
        # the decoder's SMALL weight gradients (the four 1x1 skip adapters, the three 3x3 layers of the 16 x 26 level: 16-49 us each as
        # individual split-K launches, 90 TFLOP/s) go into one grouped launch at the end of the decoder's backward segment; the
        # streaming 3x3 kernel keeps the large ones.  Operands: forward activations and per-layer `draw` buffers (never reused).
        self._defer = [] if GROUP_WGRAD else None
        dec_first_group = len(self.bwd_groups)
        self.stage_buffers = {}                           # decoder stage name -> its concat buffer
        E1, dE1 = self.act(d[3], *hs[0]), self.act(d[3], *hs[0])
        # gradient regions of encoder outputs: from_encoder dgrad stores (fp32), patch-embed dgrad accumulates
        self.conv_layer("from_encoder_1", enc_out_b[3], 1, E1, dout=dE1, dx=d_enc_out[3], dx_region=("dxs", 3, 0, d[3]))
        for j, e in ((0, 2), (1, 1), (2, 0)):
            up_p, sk_p = lay[j]
            self.conv_layer(f"from_encoder_{j + 2}", enc_out_b[e], 1, CB[j].sl(up_p, up_p + d[e]),
                            dout=dCB[j].sl(up_p, up_p + d[e]), dx=d_enc_out[e], dx_region=("dxs", e, 0, d[e]))

        def cat_map(j, nseg_out, head_extra=None):
            """internal->reference channel map of stage j's concat buffer prefix covering up|skip|out0..out{n-1}."""
            up_p, sk_p = lay[j]
            mp = list(range(cup[j])) + [-1] * (up_p - cup[j])
            mp += [cup[j] + c for c in range(cskip[j])] + [-1] * (sk_p - cskip[j])
            base = cup[j] + cskip[j]
            if nseg_out >= 1:
                mp += [base + c for c in range(96)]
            if nseg_out >= 2:
                mp += [base + 96 + c for c in range(64)]
            return mp

        def kcat_plan(j, dcb, Hj, Wj, o0, o1):
            """The write-once data gradients of stage j's three ConvLayers: the K-concatenated d(raw) buffer [layer 2 (128) | layer 1 (64) |
            layer 0 (96)] and the three weight matrices WA = concat channels [o1, o1 + 64) x K 128, WB [o0, o1) x K 192, WC [0, o0) x K 288;
            per layer (index = layer): where its GroupNorm backward writes, where its weights are packed, the launch issued behind it."""
            DRAW = self.act(288, Hj, Wj)
            WA, WB, WC = self.new((64, 9, 128)), self.new((96, 9, 192)), self.new((o0, 9, 288))
            real0 = cup[j] + cskip[j]                       # reference channels among the o0 padded ones
            kc = [None, None, None]
            # (last entry of `launch`: algorithmic MACs per pixel and tap = (channels of the layers feeding it) x (reference columns it produces))
            kc[2] = dict(draw=DRAW.sl(0, 128), packs=[(WA, 128, 0, o1, 64), (WB, 192, 0, o0, 96), (WC, 288, 0, 0, o0)],
                         launch=(DRAW.sl(0, 128), WA, o1, o1 + 64, dcb, 128 * 64))
            kc[1] = dict(draw=DRAW.sl(128, 192), packs=[(WB, 192, 128, o0, 96), (WC, 288, 128, 0, o0)],
                         launch=(DRAW.sl(0, 192), WB, o0, o1, dcb, 192 * 96))
            kc[0] = dict(draw=DRAW.sl(192, 288), packs=[(WC, 288, 192, 0, o0)],
                         launch=(DRAW.sl(0, 288), WC, 0, o0, dcb, 288 * real0))
            return kc

        def stage(j, name, cb, dcb, up_src, d_up_src, up_region, out, dout, mask):
            """Decoder stage (utils.py:249-257 + ShortResBlock :127-135) on concat buffer cb."""
            up_p, sk_p = lay[j]
            Hj, Wj = cb.H, cb.W
            o0, o1 = up_p + sk_p, up_p + sk_p + 96
            self.stage_buffers[name] = cb              # (calibrate_fp8 takes the amax of these)
            f8 = self.fp8 is not None and name in self.fp8 and -(-Wj // 32) * -(-Hj // 16) * B >= 192
            if f8:
                # the three ConvLayers read an e4m3 copy of the concat buffer (its own row stride, a multiple of 16 bytes);
                # upsample and the first two GroupNorm+GELU write that copy directly, the skip channels are quantised
                sc8 = float(self.fp8[name])
                ld8 = rup(cb.ld, 16)
                cb8 = self.new((B, Hj * Wj, ld8), torch.uint8)
                keep = self.fp8_keep_bf16           # a backward pass follows: the bf16 tensors are stored as well
                self._emit(self.fwd, "crd_bicubic2x_fp8", [up_src.t, up_src.ld, up_src.coff, B, up_src.H, up_src.W, up_p, cb8, ld8, 0, sc8]
                           + ([cb.t, cb.ld, 0] if keep else [None, 0, 0]),
                           io=nbytes(up_src.sl(0, up_p)) * (1 + (4 if keep else 0)) + B * Hj * Wj * up_p)
                if sk_p:
                    self._emit(self.fwd, "crd_quant_fp8", [cb.t, B * Hj * Wj, cb.ld, up_p, sk_p, cb8, ld8, up_p, sc8],
                               io=B * Hj * Wj * sk_p * 3)
                if keep:
                    grp = []
                    args = [dcb.t, dcb.ld, 0, B, up_src.H, up_src.W, up_p, d_up_src.t, d_up_src.ld, d_up_src.coff, 0]
                    op = self._emit(grp, "crd_bicubic2x_bwd", args, up_region, len(args) - 1)
                    op.io = lambda op=op, n=nbytes(up_src.sl(0, up_p)): n * (5 + (1 if op.args[-1] else 0))
                    self._push(grp)
                bw = lambda c0, c1: dict(dout=dcb.sl(c0, c1), dx=dcb.sl(0, c0), dx_region=("dcb", id(dcb), 0, c0)) if keep else {}
                kc = [None, None, None]
                if keep and KCAT:
                    kc = kcat_plan(j, dcb, Hj, Wj, o0, o1)
                    if self.fp8_grad:
                        # e4m3 data gradients of the whole stage: one e4m3 copy of the K-concatenated d(raw) buffer, one scale for it
                        si = len(self.fp8_grad_layers)
                        assert si < self.g8_scales.numel()
                        self.fp8_grad_layers.append(name)
                        DRAW8 = self.new((B, Hj * Wj, 288), torch.uint8)
                        for li_, keep_ in ((2, 1), (1, 1), (0, 0)):      # backward order: layer 2 first, layer 0 closes the stage's running max
                            kc[li_]["f8"] = dict(draw8=DRAW8, ld8=288, sc_ptr=self.g8_scales.data_ptr() + 4 * si,
                                                 am_ptr=self.g8_amax.data_ptr() + 4 * 64 * si, keep=keep_)
                self._cmap = cat_map(j, 0)
                self.conv_layer(f"{name}.conv.layers.0", cb.sl(0, o0), 3, cb.sl(o0, o0 + 96) if keep else None, x8=(cb8, ld8, sc8),
                                out8=(cb8, ld8, o0, sc8), kcat=kc[0], **bw(o0, o0 + 96))
                self._cmap = cat_map(j, 1)
                self.conv_layer(f"{name}.conv.layers.1", cb.sl(0, o1), 3, cb.sl(o1, o1 + 64) if keep else None, x8=(cb8, ld8, sc8),
                                out8=(cb8, ld8, o1, sc8), kcat=kc[1], **bw(o1, o1 + 64))
                self._cmap = cat_map(j, 2)
                last = dict(dout=dout, dx=dcb.sl(0, o1 + 64), dx_region=("dcb", id(dcb), 0, o1 + 64)) if keep else {}
                self.conv_layer(f"{name}.conv.layers.2", cb.sl(0, o1 + 64), 3, out, mask=mask, x8=(cb8, ld8, sc8), kcat=kc[2], **last)
                self._cmap = None
                return
            self._emit(self.fwd, "crd_bicubic2x", [up_src.t, up_src.ld, up_src.coff, B, up_src.H, up_src.W, up_p, cb.t, cb.ld, 0],
                       io=5 * nbytes(up_src.sl(0, up_p)))
            grp = []
            args = [dcb.t, dcb.ld, 0, B, up_src.H, up_src.W, up_p, d_up_src.t, d_up_src.ld, d_up_src.coff, 0]
            op = self._emit(grp, "crd_bicubic2x_bwd", args, up_region, len(args) - 1)
            op.io = lambda op=op, n=nbytes(up_src.sl(0, up_p)): n * (5 + (1 if op.args[-1] else 0))
            self._push(grp)
            kc = kcat_plan(j, dcb, Hj, Wj, o0, o1) if KCAT else [None, None, None]
            self._cmap = cat_map(j, 0)
            self.conv_layer(f"{name}.conv.layers.0", cb.sl(0, o0), 3, cb.sl(o0, o0 + 96), dout=dcb.sl(o0, o0 + 96),
                            dx=dcb.sl(0, o0), dx_region=("dcb", id(dcb), 0, o0), kcat=kc[0])
            self._cmap = cat_map(j, 1)
            self.conv_layer(f"{name}.conv.layers.1", cb.sl(0, o1), 3, cb.sl(o1, o1 + 64), dout=dcb.sl(o1, o1 + 64),
                            dx=dcb.sl(0, o1), dx_region=("dcb", id(dcb), 0, o1), kcat=kc[1])
            self._cmap = cat_map(j, 2)
            self.conv_layer(f"{name}.conv.layers.2", cb.sl(0, o1 + 64), 3, out, mask=mask, dout=dout,
                            dx=dcb.sl(0, o1 + 64), dx_region=("dcb", id(dcb), 0, o1 + 64), kcat=kc[2])
            self._cmap = None

        n_extra = int(cfg.supervised_seg) + int(cfg.unsupervised_seg)
        # source buffers S[j] = output of stage j (128 ch) | depth (1) | pad | [seg maps]; S[0..1] plain 128
        S, dS = [], []
        for j in range(5):
            Hj, Wj = hs[j + 1]
            ld = 128 if j < 2 else (136 + (8 if (n_extra and j >= 3) else 0))
            S.append(self.act(ld, Hj, Wj))
            dS.append(self.act(ld, Hj, Wj))
        self.out_depth = {}

        def head(j, name, src, dsrc, cin_ref, cmap):
            """Depth_Activation (utils.py:285-289) on src[0:len(cmap)]; writes fp32 depth [B,P,1] and the bf16 copy
            into src channel 128 (the reference's torch.cat([stage, depth]) -- CamRaDepth.py:120,146)."""
            Hj, Wj = src.H, src.W
            self._cmap = cmap
            c1 = self.new_conv(name + ".conv_1", cmap=cmap)
            self._cmap = None
            A = self.act(32, Hj, Wj)
            xin = src.sl(0, len(cmap))
            self.conv(self.fwd, self.conv_desc(xin, c1, 32, 3, 1, 1, Hj, Wj, A, bias=c1.bias, act=1))
            depth = PM(self.new((B, Hj * Wj, 1), F32), 1, Hj, Wj)
            w2, b2 = self.p(name + ".conv_2.weight"), self.p(name + ".conv_2.bias")
            cp = (src.t, src.ld, 128) if j < 5 else (None, 0, 0)
            self._emit(self.fwd, "crd_head_conv2_fwd", [A.t, w2, b2, B, Hj, Wj, depth.t, cp[0], cp[1], cp[2]], io=nbytes(A, depth) + B * Hj * Wj * 2)
            self.out_depth[j] = depth
            # backward: dy = loss gradient (fp32 [B,P,1]) [+ d(src[128]) from the next stage]
            gd = PM(self.new((B, Hj * Wj, 1), F32), 1, Hj, Wj)
            self.out_depth[("grad", j)] = gd
            grp = []
            add = (dsrc.t, dsrc.ld, 128) if j < 5 else (None, 0, 0)
            dA = self.act(32, Hj, Wj)
            rows = self.zb(HEAD_ROWS, 289)            # copies of [dw (288) | dbias]; the unpack kernel sums them
            self.row_grads.append((name + ".conv_2.weight", 288, rows, HEAD_ROWS, self._tag, 0, 289))
            self.row_grads.append((name + ".conv_2.bias", 1, rows, HEAD_ROWS, self._tag, 288, 289))
            self._emit(grp, "crd_head_conv2_bwd_data", [gd.t, add[0], add[1], add[2], A.t, w2, B, Hj, Wj, dA.t], io=nbytes(gd, A, dA) + B * Hj * Wj * 2)
            if not self.is_frozen(name + ".conv_2.weight", name + ".conv_2.bias"):
                self._emit(grp, "crd_head_conv2_wgrad", [gd.t, add[0], add[1], add[2], A.t, B, Hj, Wj, rows, HEAD_ROWS],
                           io=nbytes(gd, A) + B * Hj * Wj * 2).stream = LATE
            self.wgrad(grp, xin, dA, c1, 3, 1, 1, Hj, Wj, dbias=name + ".conv_1.bias")
            self.conv(grp, self.conv_desc(dA, ("dgrad", c1), 128, 3, 1, 1, Hj, Wj, dsrc.sl(0, 128), gather=1),
                      region=("ds", id(dsrc), 0, 128))
            self._push(grp)

        stage(0, "depth_upsample.0", CB[0], dCB[0], E1, dE1, ("de1", 0, 0, d[3]), S[0].sl(0, 128), dS[0].sl(0, 128), dmask())
        stage(1, "depth_upsample.1", CB[1], dCB[1], S[0], dS[0], ("ds", id(dS[0]), 0, 128), S[1].sl(0, 128), dS[1].sl(0, 128), dmask())
        stage(2, "depth_upsample.2", CB[2], dCB[2], S[1], dS[1], ("ds", id(dS[1]), 0, 128), S[2].sl(0, 128), dS[2].sl(0, 128), dmask())
        head(3, "depth_activation_3", S[2], dS[2], 128, list(range(128)))
        stage(3, "depth_upsample.3", CB[3], dCB[3], S[2], dS[2], ("ds", id(dS[2]), 0, 136), S[3].sl(0, 128), dS[3].sl(0, 128), dmask())
        self.seg_logits = None
        self.seg_logits_grad = None
        self.unsup_map = None

        def seg_head(name, feat, classes, dests, with_grad):
            """3x3 conv to class logits (+bias) then Seg_Block argmax/num_classes (CamRaDepth.py:128-134,155-161)."""
            cw = self.new_conv(name, need_dgrad=with_grad)
            Hj, Wj = feat.H, feat.W
            logits = PM(self.new((B, Hj * Wj, rup(classes)), F32), classes, Hj, Wj)
            self.conv(self.fwd, self.conv_desc(feat.sl(0, 128), cw, classes, 3, 1, 1, Hj, Wj, logits, bias=cw.bias))
            for (buf, ch) in dests:
                if ch is None:    # fp32 [B,1,H,W] module output
                    self._emit(self.fwd, "crd_seg_argmax", [logits.t, logits.ld, B, Hj * Wj, classes, classes, buf, 1, 1, 0], io=nbytes(logits) + B * Hj * Wj * 4)
                else:
                    self._emit(self.fwd, "crd_seg_argmax", [logits.t, logits.ld, B, Hj * Wj, classes, classes, buf.t, 0, buf.ld, ch], io=nbytes(logits) + B * Hj * Wj * 2)
            return cw, logits

        if seg:
            CBs0, dCBs0 = self.act(CB[3].ld, *hs[4]), self.act(CB[3].ld, *hs[4])
            SF0, dSF0 = self.act(136, *hs[4]), self.act(136, *hs[4])
            has_seg_grad = cfg.supervised_seg        # only the supervised branch has a loss (runner.py:197)
            stage_seg = stage if has_seg_grad else self._stage_fwd_only(stage)
            stage_seg(3, "seg_upsample.0", CBs0, dCBs0, S[2], dS[2], ("ds", id(dS[2]), 0, 136), SF0.sl(0, 128), dSF0.sl(0, 128), dmask())
            ch = 136
            if cfg.supervised_seg:
                seg_head("seg_conv_stage_4", SF0, cfg.num_classes, [(S[3], ch), (SF0, 128)], False)
                ch += 1
            if cfg.unsupervised_seg:
                dests = [(S[3], ch)] + ([] if cfg.supervised_seg else [(SF0, 128)])
                seg_head("unsup_stage_4", SF0, UNSUP_CLASSES, dests, False)
        hmap = list(range(128)) + ([-1] * 8 + [128 + c for c in range(n_extra)] + [-1] * (8 - n_extra) if n_extra else [])
        head(4, "depth_activation_4", S[3], dS[3], 128 + n_extra, hmap)
        stage(4, "depth_upsample.4", CB[4], dCB[4], S[3], dS[3], ("ds", id(dS[3]), 0, 136), S[4].sl(0, 128), dS[4].sl(0, 128), dmask())
        # the raw 7-channel input is the skip of the last stage (CamRaDepth.py:149,152)
        up_p, sk_p = lay[4]
        self.fwd.insert(1, Op(self.lib.crd_slice_copy, [X8.t, 8, 0, CB[4].t, CB[4].ld, up_p, B * H * W, 8, 0], "crd_slice_copy", io=2 * nbytes(X8)))
        self.fwd_marks = [(n_, i_ + (1 if i_ > 1 else 0)) for n_, i_ in self.fwd_marks]      # (the phase marks behind the inserted op)
        if seg:
            CBs1, dCBs1 = self.act(CB[4].ld, *hs[5]), self.act(CB[4].ld, *hs[5])
            self.fwd.insert(2, Op(self.lib.crd_slice_copy, [X8.t, 8, 0, CBs1.t, CBs1.ld, up_p, B * H * W, 8, 0], "crd_slice_copy", io=2 * nbytes(X8)))
            self.fwd_marks = [(n_, i_ + (1 if i_ > 2 else 0)) for n_, i_ in self.fwd_marks]
            SF1, dSF1 = self.act(128, *hs[5]), self.act(128, *hs[5])
            stage_seg(4, "seg_upsample.1", CBs1, dCBs1, SF0, dSF0, ("ds", id(dSF0), 0, 136), SF1, dSF1, dmask())
            ch = 136
            if cfg.supervised_seg:
                cw, logits = seg_head("seg_conv_final", SF1, cfg.num_classes, [(S[4], ch)], True)
                ch += 1
                self.seg_logits = logits
                self.seg_out = torch.zeros((B, cfg.num_classes, H, W), dtype=F32, device=self.dev)
                self._emit(self.fwd, "crd_pm_to_nchw", [logits.t, 1, logits.ld, 0, B, cfg.num_classes, H, W, self.seg_out], io=nbytes(logits, self.seg_out))
                self.seg_grad_in = torch.zeros((B, cfg.num_classes, H, W), dtype=F32, device=self.dev)
                DL = self.act(rup(cfg.num_classes), H, W)
                grp = []
                self._emit(grp, "crd_nchw_to_pm", [self.seg_grad_in, B, cfg.num_classes, H, W, DL.t, DL.ld, 0, DL.ld], io=nbytes(self.seg_grad_in, DL))
                self.wgrad(grp, SF1, DL, cw, 3, 1, 1, H, W, dbias="seg_conv_final.bias")
                self.conv(grp, self.conv_desc(DL, ("dgrad", cw), 128, 3, 1, 1, H, W, dSF1, gather=1, cin=DL.ld),
                          region=("ds", id(dSF1), 0, 128))
                self._push(grp)
            if cfg.unsupervised_seg:
                self.unsup_map = torch.zeros((B, 1, H, W), dtype=F32, device=self.dev)
                seg_head("unsup_final", SF1, UNSUP_CLASSES, [(S[4], ch), (self.unsup_map, None)], False)
        head(5, "depth_activation_5", S[4], dS[4], 128 + n_extra, hmap)
        if self._defer is not None:
            self.flush_deferred(self.bwd_groups[dec_first_group])       # the first decoder group recorded = the last one executed
        self._finalise()

    def _stage_fwd_only(self, stage):
        """Run a decoder stage builder but drop its backward groups (branches without any loss)."""
        def run(*a, **k):
            n = len(self.bwd_groups)
            nd = len(self._defer) if self._defer is not None else 0
            stage(*a, **k)
            del self.bwd_groups[n:]
            del self.bwd_tags[n:]
            if self._defer is not None:
                del self._defer[nd:]               # (their deferred weight gradients as well: no dy is ever produced for them)
        return run

    # ------------------------------------------------------------------ encoder block
    def block(self, name, X, DX, Cs, heads, ratio, sr, Hs, Ws, bi, sc, pre=None, want_next=False, dh_prev=None):
        """Block.forward (simplified_attention.py:141-145) with the rank-one form of the max-pool attention
        (SURVEY.md Q2 / Appendix B3).  X: fp32 residual stream in; returns the fp32 stream out.  DX is the stage's
        running fp32 gradient buffer (the same buffer flows through every block of the stage).
        pre: (g16 sums, channel sums) of X when the previous block's fc2 epilogue produced them; want_next: have this
        block's fc2 epilogue produce them for the next block.  dh_prev: (DH, dp) of the previous block: this block's last
        backward kernel also writes bf16(dp * d(X)) there.  Returns (stream out, sums of the stream out or None,
        (DH, dp) of this block when a later block is to fill it)."""
        B, N, hid, dh = self.B, Hs * Ws, Cs * ratio, Cs // heads
        scale = dh ** -0.5
        a, ml = name + ".attn", name + ".mlp1"
        stage_i = int(name.split("block")[1].split(".")[0]) - 1
        dp = self.dp_masks[bi] if self.training else None
        M = (Hs // sr) * (Ws // sr)
        F_ = self.fwd
        n_fwd0 = len(F_)
        # ---- attention branch ----
        XN = self.act(Cs, Hs, Ws)
        # (CRD_GN_CONV_MAXROWS restricts the fusion to the small, latency-bound stages; measured at B = 8 / 16, training and
        # inference: none 19.73 / 30.68 / 6.73 / 10.62 ms, <= 4096 rows 19.60 / 30.65 / 6.67 / 10.64, all 19.51 / 30.48 / 6.65 / 10.63)
        fused = bool(self.gn_conv_on) and B * Hs * Ws <= GN_CONV_MAXROWS
        if pre is not None and FUSE_STATS:
            st1, ch1 = pre
        else:
            st1, ch1 = self.zf(B, Cs // 16, 2), self.zf(B, Cs, 2)
            self._emit(F_, "crd_gn_stats", [X.t, X.f32, X.ld, X.coff, self.B, X.P, X.C, st1, ch1], io=nbytes(X))
        if not fused:
            self.gn_fwd(X, st1, 1, name + ".norm1", 0, None, XN)
        cq, ck, cp = self.new_conv(a + ".q"), self.new_conv(a + ".k"), self.new_conv(a + ".proj")
        Q = self.act(Cs, Hs, Ws)
        xbar = PM(self.new((B, 1, Cs)), Cs, 1, 1)
        U = PM(self.new((B, 1, Cs), F32), Cs, 1, 1)
        assert cp.cin_pad == Cs and cp.cout_pad == Cs
        K = self.act(Cs, Hs // sr, Ws // sr)
        tr = self.need_grad
        q_spec = self.conv_desc(XN, cq, Cs, 1, 1, 0, Hs, Ws, Q, bias=cq.bias)
        if fused:           # Block.norm1 applied while q loads X; XN is written on the way (the key path and the weight
            q_spec["x"] = X  # gradients read it)
            self.gn_conv(F_, q_spec, st1, 1, name + ".norm1", 0, XN)
        if sr > 1:
            csr = self.new_conv(a + ".sr", scatter=True)
            KR = self.act(Cs, Hs // sr, Ws // sr)
            stk = self.zf(B, Cs // 16, 2)
            KRN = self.act(Cs, Hs // sr, Ws // sr)
            if not fused:
                with self.side(1):       # q projection: independent of the key path below
                    self.conv(F_, q_spec)
            sr_spec = self.conv_desc(XN, csr, Cs, sr, sr, 0, Hs // sr, Ws // sr, KR, bias=csr.bias, stats=stk)
            small64 = lambda ohw: Cs <= 64 or -(-ohw // 64) * -(-Cs // 128) * B < 256        # crd_gn_conv2's rule: both problems on the 64 x 64 tiles
            if (fused and (QSR_GROUP >> stage_i) & 1 and F_ and F_[-1].name == "crd_gn_conv" and F_[-1].meta["kernel"].startswith("k_gngemm_reg")
                    and small64(Hs * Ws) and small64((Hs // sr) * (Ws // sr))):
                # round 6: q (recorded just above) and sr in one launch
                q_op = F_.pop()
                sr_spec["x"] = X
                flops_sr = 2.0 * self.B * sr_spec["OH"] * sr_spec["OW"] * Cs * csr.cin_ref * csr.taps
                gn_sr = dict(gn_in=True, x_f32=X.f32, gmul=1, stats=st1, gamma=self.p(name + ".norm1.weight"), beta=self.p(name + ".norm1.bias"),
                             act=0, xn=None)
                meta = {"kernel": "k_gngemm_reg2<2,2,1,1>", "flops": q_op.meta["flops"] + flops_sr,
                        "shape": f"gn+fwd q Cin{Cs} Cout{Cs} {Hs}x{Ws} | sr k{sr} Cin{Cs} Cout{Cs} {Hs // sr}x{Ws // sr}"}
                io = self.op_bytes(q_op) + nbytes(X, KR) + Cs * csr.taps * Cs * 2
                F_.append(Op(self.lib.crd_gn_conv2, q_op.args + [sr_spec, gn_sr], "crd_gn_conv2", None, None, meta, stream=self._cur_stream, io=io))
            else:
                self.conv(F_, sr_spec)
            k_spec = self.conv_desc(KRN, ck, Cs, 1, 1, 0, Hs // sr, Ws // sr, K, bias=ck.bias)
            if fused:       # attn.norm applied while k loads KR
                k_spec["x"] = KR
                self.gn_conv(F_, k_spec, stk, 1, a + ".norm", 0, KRN if tr else None)
            else:
                self.gn_fwd(KR, stk, 1, a + ".norm", 0, None, KRN)
                self.conv(F_, k_spec)
        else:
            if not fused:
                with self.side(1):
                    self.conv(F_, q_spec)
            self.conv(F_, self.conv_desc(XN, ck, Cs, 1, 1, 0, Hs, Ws, K, bias=ck.bias))
        self.join(F_, 1)
        Ssum = self.new((B, N), F32)
        idx = self.new((B, N, heads), torch.int16)
        self.keep.append(("idx", name, idx, M))
        # scores + the rank-one value path (xbar = mean_n GN(x), U = proj(xbar): needs norm1's sums only) in one launch
        self._emit(F_, "crd_attn_fwd", [Q.t, K.t, B, N, M, heads, dh, scale, Ssum, idx, ch1, st1, self.p(name + ".norm1.weight"),
                                        self.p(name + ".norm1.bias"), _WPtr(cp, "w_fwd"), xbar.t, U.t],
                   io=nbytes(Q, K, Ssum, idx))
        X1 = self.act(Cs, Hs, Ws, F32)
        # ---- MLP branch ----
        st2 = self.zf(B, Cs // 16, 2)
        XN2 = self.act(Cs, Hs, Ws)
        if FUSE_STATS:       # norm2's statistics come out of the kernel that writes X1
            self._emit(F_, "crd_attn_out_residual_stats", [X.t, U.t, Ssum, cp.bias, dp, B, N, Cs, X1.t, st2], io=nbytes(X, Ssum, X1))
        else:
            self._emit(F_, "crd_attn_out_residual", [X.t, U.t, Ssum, cp.bias, dp, B, N, Cs, X1.t], io=nbytes(X, Ssum, X1))
            self._emit(F_, "crd_gn_stats", [X1.t, X1.f32, X1.ld, X1.coff, self.B, X1.P, X1.C, st2, None], io=nbytes(X1))
        c1, c2 = self.new_conv(ml + ".fc1"), self.new_conv(ml + ".fc2")
        H1, H2, H3 = (self.act(hid, Hs, Ws) for _ in range(3))
        sth1, sth2 = self.zf(B, hid // 16, 2), self.zf(B, hid // 16, 2)
        fc1_spec = self.conv_desc(XN2, c1, hid, 1, 1, 0, Hs, Ws, H1, bias=c1.bias, stats=sth1)
        n1 = [sth1, 1, self.p(ml + ".norm1.weight"), self.p(ml + ".norm1.bias")]
        w9 = self.new((9, hid), F32)
        # fp32 [9][hid], values rounded to bf16 like every convolution weight under autocast (developer switch CRD_DW_F32: un-rounded,
        # the round-3 behaviour -- the A/B of tools/ab_dw_rounding.sh)
        self.dw_entries.append((ml + ".dwconv.dwconv", hid, w9, 1 if _dev_flag("CRD_DW_F32") else 2))
        X2 = self.act(Cs, Hs, Ws, F32)
        nxt = (self.zf(B, Cs // 16, 2), self.zf(B, Cs, 2)) if (want_next and FUSE_STATS) else None
        # Mlp.norm1 is applied by the depthwise kernels while they stage H1 (the normalised tensor is never stored)
        dw_args = [H1.t, B, Hs, Ws, hid, w9, self.p(ml + ".dwconv.dwconv.bias"), 0, H2.t, sth2] + n1 + [None, None, None, None]
        fc2_spec = self.conv_desc(H3, c2, Cs, 1, 1, 0, Hs, Ws, X2, bias=c2.bias, res=X1, res_scale=dp,
                                  stats=nxt[0] if nxt else None, chan=nxt[1] if nxt else None)
        slabs = int(self.lib.crd_mlp_fused_supported(Hs, Ws, Cs, hid)) if (MLP_FUSED and N <= MLP_FUSED_MAXPIX) else 0
        if slabs > 0:
            # the whole Mlp per (sample, 64-channel hidden slab) in one launch, its fc2 partial tiles folded (with bias, DropPath
            # scale, residual and the next block's norm1 sums) by a second one
            need = slabs * B * N * Cs
            if self.mlp_parts is None or self.mlp_parts.numel() < need:
                self.mlp_parts = self.new((need,), F32)
            ptrs = dict(x1=X1.t, x1_stats=st2, norm_gamma=self.p(name + ".norm2.weight"), norm_beta=self.p(name + ".norm2.bias"),
                        w_fc1=_WPtr(c1, "w_fwd"), b_fc1=c1.bias, norm1_gamma=n1[2], norm1_beta=n1[3], w9=w9,
                        b_dw=self.p(ml + ".dwconv.dwconv.bias"), norm2_gamma=self.p(ml + ".norm2.weight"),
                        norm2_beta=self.p(ml + ".norm2.bias"), w_fc2=_WPtr(c2, "w_fwd"), xn=XN2.t if tr else None,
                        h1=H1.t if tr else None, h2=H2.t if tr else None, h3=H3.t if tr else None, h1_stats=sth1, h2_stats=sth2,
                        fc2_partials=_BufPtr(self, "mlp_parts"))
            op = self._emit(F_, "crd_mlp_fwd", [dict(mlp=True, ptrs=ptrs, dims=(B, Hs, Ws, Cs, hid))])
            op.meta = {"kernel": "k_mlp_fwd", "flops": 2.0 * B * N * hid * Cs * 2, "shape": f"fused Mlp C{Cs} hid{hid} {Hs}x{Ws}"}
            op.io = nbytes(X1) + (nbytes(XN2, H1, H2, H3) if tr else 0) + slabs * B * N * Cs * 4 + 2 * hid * Cs * 2
            self._emit(F_, "crd_mlp_reduce", [_BufPtr(self, "mlp_parts"), slabs, X1.t, c2.bias, dp, B, N, Cs, X2.t,
                                             nxt[0] if nxt else None, nxt[1] if nxt else None],
                       io=slabs * B * N * Cs * 4 + nbytes(X1, X2))
        else:
            if fused:            # Block.norm2 applied while fc1 loads X1
                fc1_spec["x"] = X1
                self.gn_conv(F_, fc1_spec, st2, 1, name + ".norm2", 0, XN2 if tr else None)
            else:
                self.gn_fwd(X1, st2, 1, name + ".norm2", 0, None, XN2)
                self.conv(F_, fc1_spec)
            self._emit(F_, "crd_dwconv3x3", dw_args, io=nbytes(H1, H2))
            # Mlp.norm2 + GELU applied while fc2 loads H2 (H3 is kept for fc2's weight gradient only).  Inference plans
            # (nothing saved) take it from FC2_FOLD_MINROWS pixels x batch on: on small grids fc2 is a long-K GEMM on few
            # workgroups, which k_igemm's intra-workgroup split-K handles better than the register-path kernel
            # Round 5: at stages 1-2 the narrow streaming kernel (csrc/pw_narrow.hip) takes the folded form in training plans too --
            # it activates the rows in registers on their way into LDS and writes H3 from there: crd_gn_apply's pass disappears
            narrow = fused and not c2.frozen and int(self.lib.crd_pw_narrow_supported(hid, Cs, N)) > 0
            if fused and (narrow or self.gn_conv_on == 1 or (not tr and B * Hs * Ws >= FC2_FOLD_MINROWS)):
                fc2_spec["x"] = H2
                self.gn_conv(F_, fc2_spec, sth2, ratio, ml + ".norm2", 1, H3 if tr else None)
            else:
                self.gn_fwd(H2, sth2, ratio, ml + ".norm2", 1, None, H3)
                self.conv(F_, fc2_spec)

        # (test hooks: the block's own forward ops with its input / output -- tests/test_gpu_blocks.py runs ONE block on the oracle's
        # input, "teacher forcing" -- and its forward tensors by name)
        self.block_ops[name] = dict(ops=list(F_[n_fwd0:]), x=X, x2=X2, st1=st1, ch1=ch1, own_stats=pre is None)
        self.enc_taps[name] = dict(st1=st1, ch1=ch1, xn=XN, q=Q, k=K, ssum=Ssum, idx=idx, xbar=xbar, u=U, x1=X1, st2=st2, xn2=XN2, h1=H1,
                                   sth1=sth1, h2=H2, sth2=sth2, h3=H3, x2=X2, **(dict(kr=KR, stk=stk, krn=KRN) if sr > 1 else {}))
        # ---- backward (executed after the later blocks'; DX holds d(X2) on entry, d(X) on exit) ----
        g = []
        gen = ("blk", bi)
        DH, DHID, DHID2, DXN, DQ = sc["DH"], sc["DHID"], sc["DHID2"], sc["DXN"], sc["DQ"]
        if self._defer is not None:      # operands of deferred / late weight gradients must outlive the block
            DH, DHID2, DQ = self.act(Cs, Hs, Ws), self.act(hid, Hs, Ws), self.act(Cs, Hs, Ws)
            DHID = self.act(hid, Hs, Ws)
        dh_out = None
        if want_next and FUSE_STATS and self._defer is not None:
            dh_out = (DH, dp)        # d(X2) arrives in bf16 from the next block's norm1 backward
        else:
            self._emit(g, "crd_f32_to_bf16_rows", [DX.t, Cs, DH.t, Cs, 0, B * N, Cs, dp, N, None, 0, 0], io=nbytes(DX, DH))
        self.wgrad(g, H3, DH, c2, 1, 1, 0, Hs, Ws, dbias=ml + ".fc2.bias")
        # fc2's data gradient also runs the reduce phase of Mlp.norm2's backward on its own output (FUSE_GN_RED)
        # (on every grid where the wide pointwise kernel takes the launch: its sums stay in registers across a workgroup's tiles)
        wide = FUSE_GN_RED_WIDE and Cs in (64, 128, 160) and hid >= 256 and hid % 128 == 0 and ratio >= 2
        r2 = self.zb(B * hid * 2 + B * (hid // (16 * ratio)) * 2) if hid > 160 and (FUSE_GN_RED or wide or N <= FUSE_GN_RED_MAXPIX) else None
        red = None if r2 is None else (H2, sth2, self.p(ml + ".norm2.weight"), self.p(ml + ".norm2.bias"), ratio, 1, r2)
        self.conv(g, self.conv_desc(DH, ("dgrad", c2), hid, 1, 1, 0, Hs, Ws, DHID, gather=1, red=red))
        self.gn_bwd(g, H2, sth2, ratio, ml + ".norm2", 1, None, DHID, DHID, r=r2)           # in place: d(H2)
        dw10 = self.zb(DW_REPLICAS, 10, hid)      # [copy][9 taps + bias][channel]; the unpack kernel sums the copies
        self.dw_grads.append((ml + ".dwconv.dwconv", hid, dw10, self._tag))
        late = self._defer is not None   # off the chain: nothing reads dw10 before the segment's unpack, DHID is this block's own
        if not self.is_frozen(ml + ".dwconv.dwconv.weight", ml + ".dwconv.dwconv.bias"):
            op = self._emit(g, "crd_dwconv3x3_wgrad", [H1.t, DHID.t, B, Hs, Ws, hid, dw10, DW_REPLICAS] + n1, io=nbytes(H1, DHID))
            op.stream = LATE if late else op.stream
        # d(H1N), with the reduce phase of Mlp.norm1's backward fused in (it needs exactly this output and H1)
        r1 = self.zb(B * hid * 2 + B * (hid // 16) * 2)
        self._emit(g, "crd_dwconv3x3", [DHID.t, B, Hs, Ws, hid, w9, None, 1, DHID2.t, None, None, 1, None, None,
                                        H1.t, sth1, self.p(ml + ".norm1.weight"), r1], io=nbytes(DHID, DHID2, H1))
        # fc1's data gradient also runs the reduce phase of Block.norm2's backward on its own output (a launch less per block;
        # the GroupNorm's input is the fp32 residual stream: red_x_f32)
        rb2 = self.zb(B * Cs * 2 + B * (Cs // 16) * 2) if (FUSE_STATS and fused_reduce_tile_ok(Cs, N, B)) else None
        redb2 = None if rb2 is None else (X1, st2, self.p(name + ".norm2.weight"), self.p(name + ".norm2.bias"), 1, 0, rb2)
        fc1_dgrad = self.conv_desc(DHID2, ("dgrad", c1), Cs, 1, 1, 0, Hs, Ws, DXN, gather=1, red=redb2)
        if (GNB_FC1 >> stage_i) & 1 and self._defer is not None:
            # round 6: Mlp.norm1's backward apply runs in the operand load of fc1's data gradient; d(H1) is stored once, for fc1's weight gradient
            DH1 = self.act(hid, Hs, Ws)
            self.gn_bwd_conv(g, fc1_dgrad, H1, sth1, 1, ml + ".norm1", 0, r1, DH1)
            self.wgrad(g, XN2, DH1, c1, 1, 1, 0, Hs, Ws, dbias=ml + ".fc1.bias")
        else:
            self.gn_bwd(g, H1, sth1, 1, ml + ".norm1", 0, None, DHID2, DHID2, r=r1)             # in place: d(H1)
            self.wgrad(g, XN2, DHID2, c1, 1, 1, 0, Hs, Ws, dbias=ml + ".fc1.bias")
            self.conv(g, fc1_dgrad)
        # attention branch
        T, dSv = self.zb(B, Cs), self.new((B, N), F32)
        dbp_rows = self.zb(B, Cs)
        self.row_grads.append((a + ".proj.bias", Cs, dbp_rows, B, self._tag, 0, Cs))
        if rb2 is not None and FUSE_NORM2_APPLY and Cs <= 512:
            # the apply phase of Block.norm2's backward (DX += ...: DX = d(X1)) runs inside the launch that reads DX next
            self._emit(g, "crd_attn_out_bwd_gn", [DX.t, U.t, Ssum, dp, B, N, Cs, T, dbp_rows, dSv, X1.t, DXN.t, st2,
                                                   self.p(name + ".norm2.weight"), rb2, self.g(name + ".norm2.weight"),
                                                   self.g(name + ".norm2.bias")], io=nbytes(DX, DX, X1, DXN, Ssum, dSv))
        else:
            self.gn_bwd(g, X1, st2, 1, name + ".norm2", 0, None, DXN, DX, dx_acc=1, r=rb2)       # DX = d(X1)
            self._emit(g, "crd_attn_out_bwd", [DX.t, U.t, Ssum, dp, B, N, Cs, T, dbp_rows, dSv], io=nbytes(DX, Ssum, dSv))
        Tb = PM(self.new((B, 1, Cs)), Cs, 1, 1)
        Es = PM(self.new((B, 1, Cs), F32), Cs, 1, 1)
        # rank-one vector path (Tb = bf16(T), Es = d(xbar)/N: the bias of the q data gradient) rides in the launch of the
        # score backward: both consume attn_out_bwd's outputs.
        vec = [T, _WPtr(cp, "w_dgrad"), cp.cout_pad, 1.0 / N, Tb.t, Es.t]
        # dK: per-workgroup partial accumulators (plain stores) folded by the bf16 conversion below; fp32-atomic
        # accumulation into one buffer only when [M][C] does not fit in LDS
        nparts = self.lib.crd_attn_scores_bwd_partials(B, N, M, heads, dh)
        if nparts > 0:
            if self.attn_parts is None or self.attn_parts.numel() < nparts * B * M * Cs:
                self.attn_parts = self.new((nparts * B * M * Cs,), F32)     # shared scratch: produced and consumed back to back
            dK = None
            self._emit(g, "crd_attn_bwd", [Q.t, K.t, dSv, idx, B, N, M, heads, dh, scale, DQ.t, None, self.attn_parts] + vec,
                       io=nbytes(Q, K, dSv, idx, DQ) + nparts * B * M * Cs * 4)
        else:
            dK = self.zb(B, M, Cs)
            self._emit(g, "crd_attn_bwd", [Q.t, K.t, dSv, idx, B, N, M, heads, dh, scale, DQ.t, dK, None] + vec,
                       io=nbytes(Q, K, dSv, idx, DQ, dK))
        self.wgrad(g, xbar, Tb, cp, 1, 1, 0, 1, 1)
        self.wgrad(g, XN, DQ, cq, 1, 1, 0, Hs, Ws, dbias=a + ".q.bias")
        # d(norm1(X)) = q's data gradient (+ the rank-one vector path's Es) + the key path's.  With the fused reduce the key path
        # writes DXN first (its patches cover every pixel once) and q's data gradient goes LAST: it accumulates and, being the
        # final writer in the plain layout, also runs the reduce phase of Block.norm1's backward (fp32 input X): a launch less.
        rb1 = self.zb(B * Cs * 2 + B * (Cs // 16) * 2) if (FUSE_STATS and fused_reduce_tile_ok(Cs, N, B)) else None
        q_dgrad = self.conv_desc(DQ, ("dgrad", cq), Cs, 1, 1, 0, Hs, Ws, DXN, gather=1, bias=Es.t, bias_bstride=Cs)
        if rb1 is None:
            self.conv(g, q_dgrad)
        else:
            q_dgrad.update(accumulate=1, red=(X, st1, self.p(name + ".norm1.weight"), self.p(name + ".norm1.bias"), 1, 0, rb1))
        key_acc = 1 if rb1 is None else 0
        DKb = self.act(Cs, Hs // sr, Ws // sr)
        if dK is None:
            self._emit(g, "crd_sum_partials_bf16", [self.attn_parts, nparts, B * M * Cs, DKb.t, B * M * Cs], io=nparts * B * M * Cs * 4 + nbytes(DKb))
        else:
            self._emit(g, "crd_gsum_to_bf16", [dK, DKb.t, B * M * Cs], io=nbytes(dK, DKb))
        if sr > 1:
            DKR = self.act(Cs, Hs // sr, Ws // sr)
            # k's data gradient also runs the reduce phase of attn.norm's backward on its own output (a launch less)
            rk = self.zb(B * Cs * 2 + B * (Cs // 16) * 2) if FUSE_STATS else None
            redk = None if rk is None else (KR, stk, self.p(a + ".norm.weight"), self.p(a + ".norm.bias"), 1, 0, rk)
            self.conv(g, self.conv_desc(DKb, ("dgrad", ck), Cs, 1, 1, 0, Hs // sr, Ws // sr, DKR, gather=1, red=redk))
            self.wgrad(g, KRN, DKb, ck, 1, 1, 0, Hs // sr, Ws // sr, dbias=a + ".k.bias")      # (behind the launch that stores DKb)
            sr_scatter = self.conv_desc(DKR, ("scatter", csr), sr * sr * Cs, 1, 1, 0, Hs // sr, Ws // sr, DXN, out_mode=1,
                                        patch_k=sr, patch_c=Cs, accumulate=key_acc)
            if rk is not None and (GNB_SR >> stage_i) & 1 and self._defer is not None:
                # round 6: attn.norm's backward apply runs in the operand load of the sr patch scatter; d(KR) stored for sr's weight gradient
                DKR2 = self.act(Cs, Hs // sr, Ws // sr)
                self.gn_bwd_conv(g, sr_scatter, KR, stk, 1, a + ".norm", 0, rk, DKR2)
                self.wgrad(g, XN, DKR2, csr, sr, sr, 0, Hs // sr, Ws // sr, dbias=a + ".sr.bias")
            else:
                self.gn_bwd(g, KR, stk, 1, a + ".norm", 0, None, DKR, DKR, r=rk)
                self.wgrad(g, XN, DKR, csr, sr, sr, 0, Hs // sr, Ws // sr, dbias=a + ".sr.bias")
                self.conv(g, sr_scatter)
        else:
            self.conv(g, self.conv_desc(DKb, ("dgrad", ck), Cs, 1, 1, 0, Hs, Ws, DXN, gather=1, accumulate=key_acc))
            self.wgrad(g, XN, DKb, ck, 1, 1, 0, Hs, Ws, dbias=a + ".k.bias")
        if rb1 is not None:
            self.conv(g, q_dgrad)
        self.gn_bwd(g, X, st1, 1, name + ".norm1", 0, None, DXN, DX, dx_acc=1, r=rb1,         # DX = d(X)
                    dx2=dh_prev[0] if dh_prev else None, scale2=dh_prev[1] if dh_prev else None)
        self._push(g)
        return X2, nxt, dh_out

    # ------------------------------------------------------------------ finalisation
    def _finalise(self):
        dev = self.dev
        # packed weight arena (bf16) + pack table
        n_bf16 = 0
        for cw in self.convs:
            n_bf16 += cw.cout * cw.taps * cw.cin_pad
            if cw.need_dgrad and not cw.scatter:
                n_bf16 += cw.cin_pad * cw.taps * cw.cout_pad
            if cw.scatter:
                n_bf16 += cw.taps * cw.cin_pad * cw.cout_pad
        self.w_arena = torch.zeros(n_bf16 + 8, dtype=BF16, device=dev)
        off = 0

        def take(n):
            nonlocal off
            t = self.w_arena[off:off + n]
            off += rup(n)
            return t
        entries, pack_elems = [], []
        unpack, max_unpack = [], 1
        for cw in self.convs:
            cw.w_fwd = take(cw.cout * cw.taps * cw.cin_pad)
            if cw.need_dgrad and not cw.scatter:
                cw.w_dgrad = take(cw.cin_pad * cw.taps * cw.cout_pad)
            if cw.scatter:
                cw.w_scatter = take(cw.taps * cw.cin_pad * cw.cout_pad)
            if cw.cmap is not None:
                cw.cmap_dev = torch.tensor(cw.cmap, dtype=torch.int32, device=dev)
            e = L.PackEntry()
            e.src = self.p(cw.name + ".weight").data_ptr()
            e.dst_fwd = cw.w_fwd.data_ptr()
            e.dst_dgrad = cw.w_dgrad.data_ptr() if cw.w_dgrad is not None else None
            e.dst_scatter = cw.w_scatter.data_ptr() if cw.w_scatter is not None else None
            e.cmap = cw.cmap_dev.data_ptr() if cw.cmap_dev is not None else None
            e.Cout, e.Cin_ref, e.taps, e.Cin_pad, e.Cout_pad, e.dst_f32 = cw.cout, cw.cin_ref, cw.taps, cw.cin_pad, cw.cout_pad, 0
            entries.append(e)
            pack_elems.append(max(cw.cout * cw.taps * cw.cin_pad, cw.cin_pad * cw.taps * cw.cout_pad))
            # weight-gradient destination: a crd_sum_t scratch block (order-independent integer atomics) + unpack.
            # The streaming 3x3 kernel splits the pixels S ways; each split stores its block into its own copy (no
            # atomics, nothing to zero) and the segment's unpack kernel sums the copies.
            if cw.frozen:                  # no weight-gradient launch was recorded: nothing to accumulate or un-pack
                continue
            if cw.stream3_geom is not None and W3_PARTIALS:
                probe = L.WgradDesc()
                probe.B, probe.IH, probe.IW, probe.OH, probe.OW = self.B, cw.stream3_geom[0], cw.stream3_geom[1], cw.stream3_geom[0], cw.stream3_geom[1]
                probe.Cin, probe.Cout, probe.KH, probe.KW, probe.stride, probe.pad = cw.stream3_geom[2], cw.cout, 3, 3, 1, 1
                wcap = getattr(self.model, "w3_total_wgs", None)      # set by TrainStep in late-wgrad mode
                cw.wg_budget = int(wcap or 0)
                probe.wg_budget = cw.wg_budget
                cw.dw_S = int(self.lib.crd_conv_wgrad_splits(C.byref(probe)))
            if cw.dw_S > 0:
                cw.dw_parts = self.new((cw.dw_S, cw.cout, cw.taps, cw.cin_pad), F32)
                cw.dw = cw.dw_parts
                unpack.append(cw)
            else:
                cw.dw = self.zb(cw.cout, cw.taps, cw.cin_pad)
                unpack.append(cw)
        for (cw, Wt, ld, coff, row0, rows) in self.kcat_entries:    # this layer's columns / row range of a K-concatenated data-gradient matrix
            e = L.PackEntry()
            e.src, e.dst_dgrad = self.p(cw.name + ".weight").data_ptr(), Wt.data_ptr()
            e.cmap = cw.cmap_dev.data_ptr() if cw.cmap_dev is not None else None
            e.Cout, e.Cin_ref, e.taps, e.Cin_pad, e.Cout_pad, e.dst_f32 = cw.cout, cw.cin_ref, cw.taps, cw.cin_pad, cw.cout_pad, 0
            e.dgrad_ld, e.dgrad_coff, e.dgrad_row0, e.dgrad_rows = ld, coff, row0, rows
            entries.append(e)
            pack_elems.append(cw.cin_pad * cw.taps * cw.cout_pad)
        for (name, hid, w9, fmt) in self.dw_entries:          # fmt: crd_pack_entry.dst_f32 (2: fp32 holding bf16-rounded values, 0: bf16)
            e = L.PackEntry()
            e.src, e.dst_fwd = self.p(name + ".weight").data_ptr(), w9.data_ptr()
            e.Cout, e.Cin_ref, e.taps, e.Cin_pad, e.Cout_pad, e.dst_f32 = 1, hid, 9, hid, 8, fmt
            entries.append(e)
            pack_elems.append(9 * hid)
        # Sorted by the parameter's position in the model's flat buffer, so that the entries of a gradient bucket (a contiguous
        # range of that buffer, trainer.GradSync) are a contiguous range of the table: pack(lo, hi) re-packs one bucket right
        # behind its optimizer slice instead of everything at the head of the next step's forward (169 us on the critical path).
        base = self.model.flat.data_ptr()
        order = sorted(range(len(entries)), key=lambda i: entries[i].src)
        entries = [entries[i] for i in order]
        self.pack_offs = [(e.src - base) // 4 for e in entries]
        self.pack_elems = [pack_elems[i] for i in order]
        self.pack_table = _struct_table(entries, dev)
        self.pack_stride = C.sizeof(L.PackEntry)
        self.n_pack = len(entries)
        self.packed_version = None          # model._param_version the packed weights correspond to (None: never packed)
        # zero arenas
        self.zf_arena = self._materialise(self._zf_views)
        self.zb_arena = self._materialise(self._zb_views)
        order = ["dec", "enc3", "enc2", "enc1", "enc0"]
        items = [(order.index(cw.tag), cw, None) for cw in unpack]
        for t in self.dw_grads:                        # depthwise: one entry for the 9 taps, one for the bias row
            items += [(order.index(t[3]), None, t + ("weight",)), (order.index(t[3]), None, t + ("bias",))]
        for name, Cn, rows, R, tag, off, stride in self.row_grads:
            items.append((order.index(tag), None, (name, Cn, rows, tag, ("rows", R, off, stride))))
        items.sort(key=lambda it: it[0])
        uentries, self.unpack_ranges = [], {}
        for seg_i, cw, dwt in items:
            u = L.UnpackEntry()
            if cw is not None:
                u.src = cw.dw_parts.data_ptr() if cw.dw_parts is not None else cw.dw.t.data_ptr()
                u.dst = self.g(cw.name + ".weight").data_ptr()
                u.cmap = cw.cmap_dev.data_ptr() if cw.cmap_dev is not None else None
                u.Cout, u.Cin_ref, u.taps, u.Cin_pad = cw.cout, cw.cin_ref, cw.taps, cw.cin_pad
                if cw.dw_parts is not None:          # fp32 partial copies (plain stores), added in index order
                    u.replicas, u.replica_stride = cw.dw_S, cw.cout * cw.taps * cw.cin_pad
                else:
                    u.src_sum = 1
                nel = cw.cout * cw.taps * cw.cin_pad
            else:
                name, hid, dw10, _, which = dwt
                if isinstance(which, tuple):           # per-sample rows of a vector gradient
                    u.src, u.dst, u.cmap = dw10.t.data_ptr() + 8 * which[2], self.g(name).data_ptr(), None
                    u.Cout, u.Cin_ref, u.taps, u.Cin_pad = 1, hid, 1, hid
                    u.replicas, u.replica_stride, u.src_sum = which[1], which[3], 1
                    nel = hid
                    lo, hi, mx = self.unpack_ranges.get(order[seg_i], (len(uentries), len(uentries), 1))
                    self.unpack_ranges[order[seg_i]] = (lo, len(uentries) + 1, max(mx, nel))
                    uentries.append(u)
                    max_unpack = max(max_unpack, nel)
                    continue
                taps = 9 if which == "weight" else 1
                u.src = dw10.t.data_ptr() + (0 if which == "weight" else 9 * hid * 8)
                u.dst, u.cmap = self.g(name + "." + which).data_ptr(), None
                u.Cout, u.Cin_ref, u.taps, u.Cin_pad = 1, hid, taps, hid
                u.replicas, u.replica_stride, u.src_sum = DW_REPLICAS, 10 * hid, 1
                nel = taps * hid
            lo, hi, mx = self.unpack_ranges.get(order[seg_i], (len(uentries), len(uentries), 1))
            self.unpack_ranges[order[seg_i]] = (lo, len(uentries) + 1, max(mx, nel))
            uentries.append(u)
            max_unpack = max(max_unpack, nel)
        self.unpack_table = _struct_table(uentries, dev)
        self.unpack_stride = C.sizeof(L.UnpackEntry)
        self.n_unpack, self.max_unpack = len(uentries), max_unpack
        # backward op order + accumulate flags
        written = {}
        self.bwd, self.bwd_segments = [], []       # segments: (tag, first op, one-past-last op) in execution order
        if self.fp8_grad_layers:
            # delayed scaling: the scales every layer of this pass quantises with = the amax its dy had in the PREVIOUS pass
            self.bwd_groups[-1].insert(0, Op(self.lib.crd_fp8_scale_update, [self.g8_amax.data_ptr(), self.g8_scales.data_ptr(),
                                                                             len(self.fp8_grad_layers), self.fp8_margin, 0],
                                             "crd_fp8_scale_update", io=256 * len(self.fp8_grad_layers), cond=("fp8_jit", False)))
        for grp, tag in zip(reversed(self.bwd_groups), reversed(self.bwd_tags)):
            if self.bwd_segments and self.bwd_segments[-1][0] == tag:
                self.bwd_segments[-1][2] = len(self.bwd) + len(grp)
            else:
                self.bwd_segments.append([tag, len(self.bwd), len(self.bwd) + len(grp)])
            for op in grp:
                if op.region is not None:
                    key, c0, c1 = op.region[:-2], op.region[-2], op.region[-1]
                    prev = written.setdefault(key, [])
                    overlap = [r for r in prev if r[0] < c1 and c0 < r[1]]
                    if overlap:
                        lo, hi = min(r[0] for r in overlap), max(r[1] for r in overlap)
                        assert lo <= c0 and hi >= c1, f"partial gradient overlap at {op.name} {op.region}: {overlap}"
                        acc = 1
                    else:
                        prev.append((c0, c1))
                        acc = 0
                    if op.acc_slot is not None:
                        if isinstance(op.acc_slot, tuple):
                            op.acc_slot[1]["accumulate"] = acc
                        else:
                            op.args[op.acc_slot] = acc
                self.bwd.append(op)
        # resolve lazy arguments to raw pointers / ctypes
        for op in self.fwd + self.bwd:
            if op.args and isinstance(op.args[0], dict) and "wg_group" in op.args[0]:
                op.args = self._make_group(op.args[0]["wg_group"])
            else:
                op.args = [self._resolve(a) for a in op.args]
        self.bwd_groups = None

    def _resolve(self, a):
        if isinstance(a, dict):
            return self._make_desc(a)
        if isinstance(a, (_Lazy, _WPtr, _BufPtr, torch.Tensor)):
            return a.data_ptr()
        return a

    # ------------------------------------------------------------------ execution
    def run_ops(self, ops):
        st = L.stream()
        lib = self.lib
        if os.environ.get("CRD_DEBUG_SYNC"):      # developer aid: name the faulting kernel
            for i, op in enumerate(ops):
                if not self.live(op):
                    continue
                print(f"[crd] op {i} {op.name}", flush=True)
                rc = op.fn(*op.args, st)
                torch.cuda.synchronize()
                if rc != 0:
                    raise L.CrdError(f"{op.name} failed ({rc}): {lib.crd_last_error().decode()}")
                if os.environ.get("CRD_DEBUG_NAN") and self.model.flat_grad is not None:
                    bad = ~torch.isfinite(self.model.flat_grad)
                    if bool(bad.any()) or not bool(torch.isfinite(self.zb_arena).all()):
                        where = int(bad.nonzero()[0]) if bool(bad.any()) else -1
                        pname = "zb_arena"
                        for n_, o_ in zip(self.model._names, self.model._offsets):
                            if o_ <= where:
                                pname = n_
                        for k in self.keep:
                            if isinstance(k, tuple) and k[0] == "buf" and k[5].is_floating_point() \
                                    and not bool(torch.isfinite(k[5].float()).all()):
                                print("[crd] non-finite buffer", k[1:5], flush=True)
                        raise L.CrdError(f"non-finite gradient after op {i} {op.name} (first bad param {pname})")
                if _dev_flag("CRD_DEBUG_IDX"):
                    for k in self.keep:
                        if isinstance(k, tuple) and k[0] == "idx" and (int(k[2].min()) < 0 or int(k[2].max()) >= k[3]):
                            raise L.CrdError(f"argmax table of {k[1]} corrupted after op {i} {op.name}")
            return
        main = torch.cuda.current_stream()
        open_ = {}                                 # side branch -> its stream handle, while the branch is open
        for op in ops:
            if op.cond is not None and getattr(self, op.cond[0]) != op.cond[1]:
                continue
            if op.fn is None:                      # join marker
                if op.stream in open_:
                    ev = torch.cuda.Event()
                    ev.record(self._side_streams[op.stream - 1])
                    main.wait_event(ev)
                    del open_[op.stream]
                continue
            sh = st
            if op.stream == LATE:
                if self.split_late:
                    continue                       # runs in run_late()
            elif op.stream:
                if op.stream not in open_:         # open the branch here: it sees everything enqueued on main so far
                    if self._side_streams is None:
                        self._side_streams = [torch.cuda.Stream(), torch.cuda.Stream()]
                    ev = torch.cuda.Event()
                    ev.record(main)
                    self._side_streams[op.stream - 1].wait_event(ev)
                    open_[op.stream] = C.c_void_p(self._side_streams[op.stream - 1].cuda_stream)
                sh = open_[op.stream]
            rc = op.fn(*op.args, sh)
            if rc != 0:
                raise L.CrdError(f"{op.name} failed ({rc}): {lib.crd_last_error().decode()}")
        for sid in list(open_):                    # never leave a branch open past the end of the list
            ev = torch.cuda.Event()
            ev.record(self._side_streams[sid - 1])
            main.wait_event(ev)

    def pack(self, lo=None, hi=None):
        """fp32 parameters -> the bf16 (and e4m3) operand layouts of the kernels, on the current stream: all of them, or those
        whose parameter lies in elements [lo, hi) of the model's flat buffer (one optimizer bucket)."""
        st = L.stream()
        i0 = 0 if lo is None else bisect.bisect_left(self.pack_offs, lo)
        i1 = self.n_pack if hi is None else bisect.bisect_left(self.pack_offs, hi)
        if i1 > i0:
            L.check(self.lib.crd_weight_pack(self.pack_table.data_ptr() + i0 * self.pack_stride, i1 - i0,
                                             max(self.pack_elems[i0:i1]), st), "crd_weight_pack")
        base = self.model.flat.data_ptr()
        for cw, cin16 in self.fp8_convs:
            off = (self.p(cw.name + ".weight").data_ptr() - base) // 4
            if (lo is None or off >= lo) and (hi is None or off < hi):
                L.check(self.lib.crd_weight_quant_fp8(cw.w_fwd.data_ptr(), cw.cout, 9, cw.cin_pad, cin16, cw.w8.data_ptr(),
                                                      cw.w8_scales.data_ptr(), st), "crd_weight_quant_fp8")
        for cw, Wt, rows, K, W8, W8s in self.fp8_kcat:       # the K-concatenated data-gradient matrices: one e4m3 scale per concat channel (row)
            off = (self.p(cw.name + ".weight").data_ptr() - base) // 4
            if (lo is None or off >= lo) and (hi is None or off < hi):
                L.check(self.lib.crd_weight_quant_fp8(Wt.data_ptr(), rows, 9, K, K, W8.data_ptr(), W8s.data_ptr(), st), "crd_weight_quant_fp8")
        if lo is None and hi is None:
            self.packed_version = getattr(self.model, "_param_version", 0)

    def ensure_packed(self):
        """Re-packs everything if the parameters changed since the last full pack (model.mark_params_changed(): optimizer steps,
        load_state_dict): callers that keep the pack out of their captured forward (TrainStep, InferenceGraph) call this first."""
        if self.packed_version != getattr(self.model, "_param_version", 0):
            self.pack()

    def forward(self, masks=None, pack=True):
        """x must already be in self.x_in.  masks: optional injected {'drop_path': [...], 'dropout2d': [...]}.
        pack=False: the packed weights are current (ensure_packed() / pack(lo, hi) behind the optimizer)."""
        st = L.stream()
        self.zf_arena.zero_()
        if pack:
            self.pack()
        if self.training and not getattr(self, "training_masks_fixed", False):
            if masks is not None:
                self.dp_masks.copy_(torch.stack([t.to(self.dev, F32) for t in masks["drop_path"]]))
                self.d2_masks.copy_(torch.stack([t.to(self.dev, F32) for t in masks["dropout2d"]]))
            else:
                nblk, n_drop = self.dp_masks.shape[0], self.d2_masks.shape[0]
                # one RNG stream per data-parallel rank: the reference's single process draws independent masks for
                # every sample of the gathered batch (CamRaDepth.py:96, simplified_attention.py:123); rank 0 keeps `seed`
                seed = (self.model.seed + getattr(self.model, "rng_rank", 0) * 0x632BE59BD9B4E019) & 0xFFFFFFFFFFFFFFFF
                L.check(self.lib.crd_dropout_masks(self.dp_masks.data_ptr(), self.dp_keep.data_ptr(), nblk, self.B,
                                                   seed, self.rng_counter.data_ptr(), st), "crd_dropout_masks")
                L.check(self.lib.crd_dropout_masks(self.d2_masks.data_ptr(), self.d2_keep.data_ptr(), n_drop * self.B,
                                                   MID_CHANNELS, (seed + 1) & 0xFFFFFFFFFFFFFFFF, self.rng_counter.data_ptr(), st),
                        "crd_dropout_masks")
        with trace.range("forward"):
            self.run_ops(self.fwd)

    def backward(self, tags=None):
        """Loss gradients must already be in out_depth[('grad', j)] (and seg_grad_in).  Adds into the flat gradient.
        tags=None runs the whole pass; otherwise only the segments with those tags (in execution order: 'dec',
        'enc3', 'enc2', 'enc1', 'enc0'), which lets the caller start a gradient all-reduce between segments."""
        if tags is None or "dec" in tags:
            self.zb_arena.zero_()
        for tag, a, b in self.bwd_segments:
            if tags is not None and tag not in tags:
                continue
            with trace.range("backward:" + tag):
                self.run_ops(self.bwd[a:b])
            if tag in self.unpack_ranges and not self.split_late:
                lo, hi, mx = self.unpack_ranges[tag]
                L.check(self.lib.crd_wgrad_unpack(self.unpack_table.data_ptr() + lo * self.unpack_stride, hi - lo, mx, 1,
                                                  L.stream()), "crd_wgrad_unpack")

    def run_late(self, tags):
        """split_late: the weight-gradient ops (stream LATE) of the segments in `tags` and the un-packing of those segments'
        gradients, on the current stream.  Their inputs are forward activations and per-layer / per-block gradient buffers
        that nothing overwrites before the next step."""
        st = L.stream()
        for tag, a, b in self.bwd_segments:
            if tag not in tags:
                continue
            for op in self.bwd[a:b]:
                if op.fn is not None and op.stream == LATE:
                    rc = op.fn(*op.args, st)
                    if rc != 0:
                        raise L.CrdError(f"{op.name} failed ({rc}): {self.lib.crd_last_error().decode()}")
            if tag in self.unpack_ranges:
                lo, hi, mx = self.unpack_ranges[tag]
                L.check(self.lib.crd_wgrad_unpack(self.unpack_table.data_ptr() + lo * self.unpack_stride, hi - lo, mx, 1, st),
                        "crd_wgrad_unpack")


class _WPtr:
    """Packed bf16 weights of a conv (allocated in Plan._finalise) as a raw-pointer op argument."""
    __slots__ = ("cw", "kind")

    def __init__(self, cw, kind):
        self.cw, self.kind = cw, kind

    def data_ptr(self):
        return getattr(self.cw, self.kind).data_ptr()


class _BufPtr:
    """A plan-owned scratch buffer that may still be re-allocated (grown) while the plan is built, as a raw-pointer op argument."""
    __slots__ = ("plan", "attr")

    def __init__(self, plan, attr):
        self.plan, self.attr = plan, attr

    def data_ptr(self):
        return getattr(self.plan, self.attr).data_ptr()


class _Lazy:
    """Placeholder for a slice of a zero-arena, materialised in Plan._finalise."""
    __slots__ = ("shape", "numel", "t")

    def __init__(self, shape):
        self.shape = tuple(shape)
        n = 1
        for s in shape:
            n *= s
        self.numel, self.t = n, None

    def data_ptr(self):
        return self.t.data_ptr()


def _struct_table(entries, dev):
    if not entries:
        return torch.zeros(8, dtype=torch.uint8, device=dev)
    raw = b"".join(bytes(e) for e in entries)
    return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)


