"""SURVEY §8 f4 -- eval-time image pre-processing: detectron2 ResizeShortestEdge (Pillow bilinear on uint8) + FixedSizeCrop (pad 128) +
(x - mean) / std  (reference: psalm/model/datasets_mapper/coco_panoptic_mapper.py:60-91,134-163).
  * the oracle's restatement of Pillow's resampler is pinned bit-for-bit to the installed Pillow itself;
  * the product's coefficient tables == the oracle's; the HIP kernels (emulator here, MI355X with -m gpu) == the oracle, BIT-exact
    (integer work; the final fp32 normalisation is one subtraction and one correctly-rounded division);
  * ImagePreprocessor.preprocess(dataset_dict, ...) honours the mapper's contract."""
import numpy as np
import pytest
import torch

from ops_backend import ops  # noqa: F401
from oracle import pil_resample as PR
from psalm_amd.builder import ImagePreprocessor, resize_shortest_edge_shape
from psalm_amd.preprocess import pil_bilinear_tables

MEAN, STD = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]
SHAPES = [(40, 30, 64), (48, 64, 64), (97, 233, 96), (150, 100, 64), (64, 64, 64), (33, 64, 64), (20, 300, 128)]


@pytest.mark.parametrize("h,w,nh,nw", [(40, 30, 64, 48), (480, 640, 768, 1024), (700, 500, 512, 366), (97, 233, 41, 99), (50, 50, 50, 70), (33, 77, 12, 77)])
def test_oracle_restatement_is_pillow_bit_for_bit(h, w, nh, nw):
    from PIL import Image
    img = np.random.default_rng(h * w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    want = np.asarray(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
    assert np.array_equal(PR.resize_bilinear_u8(img, nh, nw), want)


@pytest.mark.parametrize("a,b", [(30, 48), (640, 1024), (1500, 1024), (233, 99), (77, 12), (4000, 1024), (5, 1024), (1024, 1024)])
def test_product_tables_equal_oracle_tables(a, b):
    b0, k0, ks0 = PR.precompute_coeffs(a, b)
    b1, k1, ks1 = pil_bilinear_tables(a, b)
    assert ks0 == ks1 and np.array_equal(b0, b1) and np.array_equal(k0, k1)


def test_resize_shape_rule():
    for h, w, s in [(480, 640, 1024), (1500, 1000, 1024), (1024, 1024, 1024), (427, 640, 1024), (333, 500, 384)]:
        assert resize_shortest_edge_shape(h, w, s, s) == PR.resize_shortest_edge_shape(h, w, s, s)
        assert max(resize_shortest_edge_shape(h, w, s, s)) == s


@pytest.mark.parametrize("h,w,S", SHAPES)
def test_image_preprocess_kernel_bit_exact(ops, h, w, S):
    img = np.random.default_rng(h + 7 * w).integers(0, 256, (h, w, 3), dtype=np.uint8)
    want, pm, (nh, nw) = PR.preprocess(img, S, MEAN, STD)
    got, gpm = ops.image_preprocess(torch.from_numpy(img).to(ops.device), nh, nw, S, torch.tensor(MEAN), torch.tensor(STD))
    assert np.array_equal(gpm.cpu().numpy(), pm)
    assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))       # bit-identical float32


def test_preprocess_contract_and_host_path_match_oracle(tmp_path):
    from PIL import Image
    img = np.random.default_rng(5).integers(0, 256, (60, 90, 3), dtype=np.uint8)
    f = tmp_path / "img.png"
    Image.fromarray(img).save(f)
    p = ImagePreprocessor(64, MEAN, STD)
    d = p.preprocess({"file_name": str(f), "height": 60, "width": 90, "image_id": 7}, mask_format="bitmask", region_mask_type="point")
    want, pm, (nh, nw) = PR.preprocess(img, 64, MEAN, STD)
    assert d["image_id"] == 7 and (d["height"], d["width"]) == (60, 90)
    assert np.array_equal(d["image"].numpy().view(np.uint32), want.view(np.uint32)) and np.array_equal(d["padding_mask"].numpy(), pm)
    assert d["transforms"]["resize"] == (60, 90, nh, nw)
    with pytest.raises(ValueError):
        p.preprocess({"file_name": str(f), "height": 61, "width": 90})
    d2 = p(img)                                                # plain callable form on an in-memory image
    assert torch.equal(d2["image"], d["image"])
