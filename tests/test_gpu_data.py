"""GPU tests of the steps either side of the hot path: device-side image / label resizes of the dataloader
(N1), the segmentation IoU of Trainer.test (N2), and the graph-replayed inference forward (N4)."""
import numpy as np
import pytest
import torch

from camradepth_amd import synth
from camradepth_amd.config import ModelConfig

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("src,dst", [((450, 800), (416, 800)), ((900, 1600), (416, 800)), ((37, 53), (64, 96)), ((450, 800), (256, 416)),
                                     ((20, 31), (7, 5))])
def test_image_resize_nearest_is_bit_exact(src, dst):
    """crd_resize_nearest_u8 against the restated OpenCV INTER_NEAREST index rule (dataloader.py:227)."""
    from camradepth_amd.batch import resize_image_nearest
    from oracle import data as od
    rs = np.random.RandomState(3)
    img = rs.randint(0, 256, size=(2,) + src + (3,)).astype(np.uint8)
    got = resize_image_nearest(torch.from_numpy(img).cuda(), dst).cpu().numpy()
    for b in range(2):
        assert np.array_equal(got[b], od.resize_image_nearest(img[b], dst))


@pytest.mark.parametrize("src", [(450, 800), (416, 800), (123, 77), (900, 1600)])
def test_seg_targets_match_scipy_zoom_bit_exact(src):
    """crd_resize_labels_nearest against scipy.ndimage.zoom(order=0, grid_mode=True) -- what scikit-image 0.19.3's
    resize(order=0, anti_aliasing=False) runs (dataloader.py:262-267) -- and the oracle restatement."""
    import scipy.ndimage as ndi
    from camradepth_amd.batch import seg_targets
    from oracle import data as od
    rs = np.random.RandomState(4)
    m = rs.randint(0, 22, size=(2,) + src).astype(np.uint8)
    m[m == 21] = 255
    out = seg_targets(torch.from_numpy(m).cuda())
    assert out["final_seg"].dtype == torch.int64 and out["final_seg"].shape == (2, 416, 800)
    for key, size in (("final_seg", (416, 800)), ("intermediate_seg", (208, 400))):
        got = out[key].cpu().numpy()
        for b in range(2):
            s = m[b][:416].astype(np.float64)
            ref = ndi.zoom(s, [size[0] / s.shape[0], size[1] / s.shape[1]], order=0, mode="reflect", grid_mode=True)
            assert np.array_equal(got[b], ref.astype(np.int64))
            assert np.array_equal(got[b], od.resize_labels(m[b], size))


def test_seg_iou_matches_oracle():
    """crd_seg_confusion + SegIoU against the restated torchmetrics 0.10.2 JaccardIndex (runner.py:432-438), including the
    frame whose labels contain 255 (NaN in the reference) and the nanmean over frames (:508)."""
    from camradepth_amd.metrics import SegIoU
    from oracle import losses as ol
    rs = np.random.RandomState(5)
    B, C, H, W = 4, 21, 48, 80
    logits = torch.from_numpy(rs.standard_normal(size=(B, C, H, W)).astype(np.float32))
    labels = torch.from_numpy(rs.randint(0, 21, size=(B, H, W)).astype(np.int64))
    labels[1, :5, :7] = 255
    labels[2][labels[2] > 14] = 3                 # classes absent from the target (present in the predictions)
    logits[3, :, :, :] += 3.0 * torch.nn.functional.one_hot(labels[3], C).permute(2, 0, 1)     # a mostly right frame
    iou = SegIoU(C)
    iou.update(logits[:2].cuda(), labels[:2].cuda())
    iou.update(logits[2:].cuda(), labels[2:].cuda())
    got = iou.per_frame()
    ref = [ol.seg_iou(logits[f:f + 1], labels[f:f + 1], C) for f in range(B)]
    assert np.isnan(got[1]) and np.isnan(ref[1])
    for f in (0, 2, 3):
        assert got[f] == pytest.approx(ref[f], rel=1e-12)
    assert iou.result() == pytest.approx(np.nanmean(ref), rel=1e-12)
    assert got[3] > 0.5 > got[0]
    # exact confusion counts
    mat = torch.cat(iou.mats).cpu()
    assert int(mat[0].sum()) == H * W and int(mat[1].sum()) == H * W - 35 and int(torch.cat(iou.oor)[1]) == 35


@pytest.mark.parametrize("variant,B,H,W", [("base", 1, 416, 800), ("supervised_seg", 2, 64, 96)])
def test_inference_graph_equals_eager_eval_forward(variant, B, H, W):
    """The graph-replayed eval forward (the reference's `runtime` path, runner.py:417-420, native 416x800 frame) returns
    what the eager module forward returns, and a second frame through the same graph is not stale."""
    from camradepth_amd.inference import InferenceGraph
    from camradepth_amd.model import CamRaDepth
    cfg = ModelConfig.variant(variant)
    m = CamRaDepth(input_channels=7, supervised_seg=cfg.supervised_seg).cuda().eval()
    ig = InferenceGraph(m, B, H, W)
    for seed in (1234, 99):
        x = synth.make_batch(B, H, W, seed=seed)["image"].cuda()
        with torch.no_grad():
            ref = m(x)
        out = ig.run(x)
        torch.cuda.synchronize()

        # every multi-workgroup sum is order-independent (crd_sum_t): the graph replay gives the eager forward's bits
        assert torch.equal(out["depth"]["final_depth"], ref["depth"]["final_depth"])
        assert torch.equal(out["depth"]["intermediate_depths"][3], ref["depth"]["intermediate_depths"][3])
        assert torch.equal(out["depth"]["intermediate_depths"][2], ref["depth"]["intermediate_depths"][2])
        if cfg.supervised_seg:
            assert torch.equal(out["seg"]["final_seg"], ref["seg"]["final_seg"])
        else:
            assert out["seg"]["final_seg"] is None
    assert not m.training
