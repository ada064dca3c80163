"""CPU-side checks: the C-ABI library builds, loads and exports every symbol include/camradepth_hip.h declares with the
signature the ctypes binding uses; host-side structures; no compute calls (no GPU here)."""
import ctypes
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    from camradepth_amd import lib
    return lib


def _header_functions():
    h = open(os.path.join(REPO, "include", "camradepth_hip.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    out = {}
    for m in re.finditer(r"\b(?:int|const char\*)\s+(crd_\w+)\s*\((.*?)\)\s*;", h, flags=re.S):
        name, args = m.group(1), [a.strip() for a in m.group(2).split(",")]
        sig = ""
        for a in args:
            if a in ("void", ""):
                continue
            if "*" in a or "crd_stream_t" in a:
                sig += "p"
            elif "uint64_t" in a:
                sig += "L"
            elif "int64_t" in a:
                sig += "l"
            elif "float" in a:
                sig += "f"
            else:
                sig += "i"
        out[name] = sig
    return out


def test_library_exports_every_declared_symbol(built):
    lib = built
    L = lib.load()
    fns = _header_functions()
    assert len(fns) >= 30
    raw = ctypes.CDLL(lib.LIB_PATH)
    for name in fns:
        assert hasattr(raw, name), f"{name} declared in the header but not exported"
    assert L.crd_version() == lib.ABI_VERSION and L.crd_arch() == b"gfx950"


def test_binding_signatures_match_header(built):
    fns = _header_functions()
    for name, sig in built._SIGS.items():
        assert fns[name] == sig, (name, fns[name], sig)
    assert set(built._SIGS) == {n for n in fns if n not in ("crd_last_error", "crd_version", "crd_arch")}


def test_struct_layouts_match_header(built, tmp_path):
    """sizeof/offsetof of the descriptor structs as gcc lays them out vs the ctypes mirror."""
    import subprocess
    fields = {"crd_conv_desc": (built.ConvDesc, ["x", "w", "OH", "y", "bias", "res", "res_scale", "stats", "red_x", "red_act", "red_stats", "red_r"]),
              "crd_gn_input": (built.GnInput, ["x_f32", "gmul", "stats", "gamma", "beta", "act", "xn_ld", "xn"]),
              "crd_gn_bwd_input": (built.GnBwdInput, ["gx", "gx_f32", "gx_ld", "gmul", "act", "stats", "mask", "r", "dx", "dx_ld", "dgamma", "dbeta"]),
              "crd_wgrad_desc": (built.WgradDesc, ["x", "dy", "Cout", "dw", "dbias", "dw_partials", "dw_partial_capacity", "wg_budget"]),
              "crd_pack_entry": (built.PackEntry, ["src", "cmap", "Cout", "dst_f32", "dgrad_ld", "dgrad_rows"]),
              "crd_unpack_entry": (built.UnpackEntry, ["src", "cmap", "Cin_pad", "replicas", "replica_stride"]),
              "crd_wgrad_group_info": (built.WgradGroupInfo, ["n_problems", "n_items", "item_offset", "bytes"])}
    src = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{REPO}/include/camradepth_hip.h"', "int main(void){"]
    for cname, (_, fl) in fields.items():
        src.append(f'printf("{cname} %zu", sizeof({cname}));')
        for f in fl:
            src.append(f'printf(" %zu", offsetof({cname}, {f}));')
        src.append('printf("\\n");')
    src.append("return 0;}")
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-o", str(exe), str(c)])
    lines = subprocess.check_output([str(exe)], text=True).strip().splitlines()
    for line in lines:
        parts = line.split()
        ct, fl = fields[parts[0]]
        assert ctypes.sizeof(ct) == int(parts[1]), parts[0]
        for f, off in zip(fl, parts[2:]):
            assert getattr(ct, f).offset == int(off), (parts[0], f)


def test_invalid_arguments_are_reported_without_a_gpu(built):
    L = built.load()
    d = built.ConvDesc()
    rc = L.crd_conv_igemm(ctypes.byref(d), None)
    assert rc == -1 and b"null" in L.crd_last_error()
    with pytest.raises(built.CrdError):
        built.check(rc, "crd_conv_igemm")


def test_module_parameter_inventory_and_flat_views():
    import torch
    from camradepth_amd.model import CamRaDepth
    from tests.util import param_order
    for variant, (sup, unsup) in {"base": (False, False), "sup_unsup_seg": (True, True)}.items():
        m = CamRaDepth(input_channels=7, supervised_seg=sup, unsupervised_seg=unsup)
        ref = param_order(variant)
        assert [[n, list(p.shape)] for n, p in m.named_parameters()] == ref["params"]
        assert sum(p.numel() for p in m.parameters()) == ref["num_params"]
        assert m._flat_ok()
        p = m.dest_encoder.block2[3].attn.q.weight if hasattr(m.dest_encoder.block2, "__getitem__") else None
        w = m._param("dest_encoder.block2.3.attn.q.weight")
        with torch.no_grad():
            w.fill_(3.0)
        assert float(m.param_view("dest_encoder.block2.3.attn.q.weight").min()) == 3.0
        # the reference's initialisation statistics
        m2 = CamRaDepth(input_channels=7)
        assert abs(float(m2._param("dest_encoder.block1.0.attn.q.weight").std()) - 0.02) < 2e-3
        assert float(m2._param("dest_encoder.block1.0.norm1.weight").min()) == 1.0
    with pytest.raises(AssertionError):
        CamRaDepth(heads=(1, 2, 4), input_channels=7)


def test_status_query_without_a_gpu(built):
    L = built.load()
    assert L.crd_nonfinite_status(0, None) in (0, -3)          # no device here: either "nothing flagged" or a reported HIP error, never a crash
