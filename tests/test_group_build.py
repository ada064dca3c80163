"""crd_wgrad_group_build is host code: the work-item table of a grouped weight-gradient launch, checked without a GPU.
Round 6: the items are dealt to the 8 XCDs (workgroup index mod 8) so that the tiles of one (problem, K split) -- which re-read the same
x and dy rows -- run on ONE XCD and share its L2, as long as that leaves the XCDs evenly loaded."""
import ctypes as C

import numpy as np

from camradepth_amd import lib

BM = {0: 32, 1: 64, 2: 96, 3: 128}


def cfg(cout):
    return 0 if cout <= 32 else 1 if cout <= 64 else 2 if cout <= 96 else 3


def build(shapes):
    lb = lib.load()
    descs = (lib.WgradDesc * len(shapes))()
    keep = (C.c_uint8 * 64)()                      # any non-null address: the builder never dereferences the tensors
    p = C.addressof(keep)
    for d, (B, Cin, H, W, Cout, k, s, pad) in zip(descs, shapes):
        OH, OW = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
        d.x, d.x_ld, d.x_coff, d.B, d.IH, d.IW, d.Cin = p, Cin, 0, B, H, W, Cin
        d.dy, d.dy_ld, d.dy_coff, d.OH, d.OW, d.Cout = p, Cout, 0, OH, OW, Cout
        d.KH, d.KW, d.stride, d.pad, d.dw, d.dbias = k, k, s, pad, p, p
    info = lib.WgradGroupInfo()
    lib.check(lb.crd_wgrad_group_build(descs, len(shapes), None, 0, C.byref(info)), "size query")
    host = (C.c_uint8 * info.bytes)()
    info2 = lib.WgradGroupInfo()
    lib.check(lb.crd_wgrad_group_build(descs, len(shapes), host, info.bytes, C.byref(info2)), "build")
    assert list(info2.n_items) == list(info.n_items) and info2.bytes == info.bytes and list(info2.item_offset) == list(info.item_offset)
    head = info.bytes - 16 * sum(info.n_items)
    assert head >= len(shapes) and head % 16 == 0
    items = np.frombuffer(bytes(host), dtype=np.int32, offset=head).reshape(-1, 4)
    return [items[info.item_offset[c]:info.item_offset[c] + info.n_items[c]].tolist() for c in range(4)]


def check_cover(shapes, parts):
    """every tile of every split of every problem exactly once, in its configuration's list; padding only inside, and little of it"""
    seen = set()
    for c, part in enumerate(parts):
        real = [it for it in part if it[0] >= 0]
        assert not part or part[-1][0] >= 0, "trailing padding"
        assert len(real) >= 0.8 * len(part) - 8, "more than a fifth of the launch would be no-op workgroups"
        for prob, t, m, split in real:
            B, Cin, H, W, Cout, k, s, pad = shapes[prob]
            assert cfg(Cout) == c and 0 <= t < -(-(k * k * Cin) // 128) and 0 <= m < -(-Cout // BM[c]) and split >= 0
            assert (prob, t, m, split) not in seen
            seen.add((prob, t, m, split))
    for prob, (B, Cin, H, W, Cout, k, s, pad) in enumerate(shapes):
        tn, tm = -(-(k * k * Cin) // 128), -(-Cout // BM[cfg(Cout)])
        splits = {sp for (p_, _, _, sp) in seen if p_ == prob}
        assert splits == set(range(len(splits))) and len(splits) >= 1
        for sp in splits:
            assert {(t, m) for (p_, t, m, s_) in seen if p_ == prob and s_ == sp} == {(t, m) for t in range(tn) for m in range(tm)}


def test_group_table_covers_every_tile_once():
    # pointwise layers, a strided sr convolution, 3x3 on small grids, narrow / ragged outputs: every tile configuration, some with a handful of units
    shapes = [(8, 320, 16, 26, 1280, 1, 1, 0), (8, 1280, 16, 26, 320, 1, 1, 0), (8, 320, 16, 26, 320, 2, 2, 0), (8, 128, 10, 11, 24, 3, 1, 1),
              (2, 136, 19, 23, 96, 3, 1, 1), (8, 64, 64, 104, 64, 1, 1, 0), (8, 64, 64, 104, 512, 1, 1, 0), (8, 160, 1, 1, 160, 1, 1, 0)]
    check_cover(shapes, build(shapes))
    check_cover(shapes[:1], build(shapes[:1]))      # one problem alone must still spread over the XCDs


def test_tiles_that_share_operands_share_an_xcd():
    """The 16 Blocks of encoder stage 3 (B = 8, 16 x 26 pixels): each (problem, split) unit on one XCD, almost no padding."""
    blk = [(8, 320, 16, 26, 1280, 1, 1, 0), (8, 1280, 16, 26, 320, 1, 1, 0), (8, 320, 16, 26, 320, 1, 1, 0), (8, 320, 8, 13, 320, 1, 1, 0),
           (8, 320, 16, 26, 320, 1, 1, 0), (8, 320, 16, 26, 320, 2, 2, 0)]
    shapes = blk * 16
    parts = build(shapes)
    check_cover(shapes, parts)
    part = parts[3]
    assert len(part) > 1000 and sum(it[0] < 0 for it in part) <= 0.02 * len(part)
    xcd_of = {}
    for idx, (prob, t, m, split) in enumerate(part):
        if prob >= 0:
            assert xcd_of.setdefault((prob, split), idx % 8) == idx % 8, "tiles of one (problem, split) on different XCDs"
    loads = np.bincount([i % 8 for i, it in enumerate(part) if it[0] >= 0], minlength=8)
    assert loads.max() - loads.min() <= 0.05 * loads.mean() + 40, loads
