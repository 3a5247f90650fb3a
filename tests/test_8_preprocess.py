"""SURVEY §8 f4 -- eval-time image pre-processing: detectron2 ResizeShortestEdge (Pillow bilinear on uint8) + FixedSizeCrop (pad 128) +
(x - mean) / std  (reference: psalm/model/datasets_mapper/coco_panoptic_mapper.py:60-91,134-163).
  * the oracle's restatement of Pillow's resampler is pinned bit-for-bit to the installed Pillow itself;
  * the product's coefficient tables == the oracle's; the HIP kernels (emulator here, MI355X with -m gpu) == the oracle, BIT-exact
    (integer work; the final fp32 normalisation is one subtraction and one correctly-rounded division);
  * ImagePreprocessor.preprocess(dataset_dict, ...) honours the mapper's contract."""
import os

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


# ---------------------------------------------------------------------------------------------- region prompts of the interactive task
def test_enhance_with_circles_matches_the_reference_functions():
    """psalm_amd.preprocess.enhance_with_circles (row-wise disc dilation) against outputs of the reference's own draw_circle / enhance_with_circles
    (coco_instance_mapper.py:17-32, executed by tests/golden/make_region_prompt_golden.py): points (radius 10), a scribble (radius 5), prompts on
    the image border."""
    import numpy as np
    from psalm_amd.preprocess import enhance_with_circles
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "region_prompts.npz"))
    n = 0
    while f"in_{n}" in z:
        got = enhance_with_circles(z[f"in_{n}"], int(z[f"radius_{n}"]))
        assert got.dtype == np.uint8 and np.array_equal(got, z[f"out_{n}"]), n
        n += 1
    assert n == 4


def test_rle_to_mask_both_count_forms_and_the_runs_order():
    """pycocotools' decode restated (psalm_amd.preprocess.rle_to_mask): uncompressed count lists and the compressed string of maskApi.c, column-major
    runs starting with zeros -- against the oracle's codec (oracle/evalout_ref.py) and a hand-written case."""
    import numpy as np
    from oracle import evalout_ref as E
    from psalm_amd.preprocess import rle_to_mask
    assert np.array_equal(rle_to_mask({"size": [2, 3], "counts": [1, 2, 3]}), np.array([[0, 1, 0], [1, 0, 0]], np.uint8))    # columns: [0,1] [1,0] [0,0]
    g = np.random.default_rng(3)
    for h, w, p in ((23, 31, 0.3), (64, 48, 0.02), (5, 7, 0.9)):
        m = (g.random((h, w)) < p).astype(np.uint8)
        c = E.rle_encode(m)
        assert np.array_equal(rle_to_mask({"size": [h, w], "counts": c}), m)
        assert np.array_equal(rle_to_mask({"size": [h, w], "counts": E.rle_to_string(c)}), m)
        assert np.array_equal(rle_to_mask({"size": [h, w], "counts": E.rle_to_string(c).decode()}), m)
    with pytest.raises(ValueError):
        rle_to_mask({"size": [4, 4], "counts": [3, 3]})


def test_preprocess_builds_region_masks_from_prompt_annotations():
    """ImagePreprocessor.preprocess(dataset_dict, region_mask_type=...) for the interactive task (coco_instance_mapper.py:233-252): one region mask
    per object -- prompt RLE decoded, points widened to radius-10 discs, NEAREST-resized and zero-padded to the (S, S) canvas exactly as the image's
    own geometry -- plus the kept objects' ground-truth masks; crowd objects and objects without a prompt of the requested kind are left out; prompt
    annotations none of which is usable raise instead of returning `instances` without region masks."""
    import numpy as np
    from PIL import Image
    from oracle import evalout_ref as E
    from psalm_amd.builder import ImagePreprocessor
    from psalm_amd.preprocess import enhance_with_circles
    h, w, S = 60, 90, 128
    g = np.random.default_rng(5)
    img = g.integers(0, 255, (h, w, 3), dtype=np.uint8)

    def rle(m):
        return {"size": [h, w], "counts": E.rle_to_string(E.rle_encode(m))}
    empty = np.zeros((h, w), np.uint8)
    pts = [empty.copy() for _ in range(3)]
    pts[0][20, 30] = 1
    pts[1][59, 0] = 1
    box = empty.copy()
    box[10:30, 40:70] = 1
    gts = [(g.random((h, w)) < 0.2).astype(np.uint8) for _ in range(4)]
    annos = [{"iscrowd": 0, "segmentation": rle(gts[0]), "point_visual_prompt_mask": rle(pts[0]), "box_visual_prompt_mask": rle(box), "mask_visual_prompt_mask": None,
              "scribble_visual_prompt_mask": rle(empty)},
             {"iscrowd": 1, "segmentation": rle(gts[1]), "point_visual_prompt_mask": rle(pts[0]), "box_visual_prompt_mask": None, "mask_visual_prompt_mask": None,
              "scribble_visual_prompt_mask": None},
             {"iscrowd": 0, "segmentation": rle(gts[2]), "point_visual_prompt_mask": rle(pts[1]), "box_visual_prompt_mask": None, "mask_visual_prompt_mask": None,
              "scribble_visual_prompt_mask": None},
             {"iscrowd": 0, "segmentation": rle(gts[3]), "point_visual_prompt_mask": rle(empty), "box_visual_prompt_mask": rle(box), "mask_visual_prompt_mask": None,
              "scribble_visual_prompt_mask": None}]
    proc = ImagePreprocessor(S, [123.675, 116.28, 103.53], [58.395, 57.12, 57.375])
    d = proc.preprocess({"image": img, "height": h, "width": w, "annotations": annos}, region_mask_type=["point_visual_prompt_mask"])
    nh, nw = d["transforms"]["resize"][2:]
    assert (nh, nw) == (85, 128)
    rm = d["instances"].region_masks.tensor
    assert d["region_annotation_indices"] == [0, 1] and rm.shape == (2, S, S) and rm.dtype == torch.bool       # non-crowd objects 0 and 2; object 3 has no point
    for k, src in enumerate((pts[0], pts[1])):
        want = np.zeros((S, S), np.uint8)
        want[:nh, :nw] = np.asarray(Image.fromarray(enhance_with_circles(src, 10)).resize((nw, nh), Image.NEAREST))
        assert np.array_equal(rm[k].numpy().astype(np.uint8), want) and want.sum() > 100
    gt = d["instances"].gt_masks
    assert gt.shape == (2, S, S) and gt.dtype == torch.float32 and float(gt[:, nh:].abs().max()) == 0
    assert np.array_equal(gt[1, :nh, :nw].numpy().astype(np.uint8), np.asarray(Image.fromarray(gts[2]).resize((nw, nh), Image.NEAREST)))
    # box prompts: no widening; the kind is drawn per object from those that are present
    d2 = proc.preprocess({"image": img, "annotations": annos}, region_mask_type=["box_visual_prompt_mask"])
    assert d2["region_annotation_indices"] == [0, 2] and int(d2["instances"].region_masks.tensor[0].sum()) > 0
    assert "instances" not in proc.preprocess({"image": img}, region_mask_type=["point_visual_prompt_mask"])    # no annotations: as the reference, nothing to do
    with pytest.raises(ValueError):
        proc.preprocess({"image": img, "annotations": annos}, region_mask_type=["scribble_visual_prompt_mask"])  # nothing non-empty of that kind
