"""TEST INFRASTRUCTURE: golden vectors for the region-prompt preparation of the interactive task, produced by the REFERENCE's own functions.

`draw_circle` / `enhance_with_circles` are taken from /root/reference/psalm/model/datasets_mapper/coco_instance_mapper.py by extracting those two
function definitions from the module's syntax tree (the module itself imports detectron2 / cv2 / pycocotools, absent here) and executing them
unchanged against numpy.  Inputs: seeded sparse binary masks (points and a scribble).  -> tests/golden/region_prompts.npz (inputs + expected outputs).

    python tests/golden/make_region_prompt_golden.py"""
import ast
import os

import numpy as np

REF = "/root/reference/psalm/model/datasets_mapper/coco_instance_mapper.py"
HERE = os.path.dirname(os.path.abspath(__file__))


def reference_functions():
    tree = ast.parse(open(REF).read())
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("draw_circle", "enhance_with_circles")]
    assert len(keep) == 2
    ns = {"np": np}
    exec(compile(ast.Module(body=keep, type_ignores=[]), REF, "exec"), ns)
    return ns["enhance_with_circles"]


def main():
    enhance = reference_functions()
    rng = np.random.default_rng(20251001)
    out = {}
    for i, (h, w, n, radius) in enumerate([(120, 160, 3, 10), (97, 131, 1, 10), (64, 64, 6, 5), (50, 200, 2, 10)]):
        m = np.zeros((h, w), np.uint8)
        ys, xs = rng.integers(0, h, n), rng.integers(0, w, n)
        m[ys, xs] = 1
        if radius == 5:                                              # a scribble: a short polyline of set pixels
            for t in range(40):
                m[min(h - 1, 10 + t // 2), min(w - 1, 5 + t)] = 1
        m[0, 0] = 1                                                   # a prompt on the image corner (discs clipped by the border)
        out[f"in_{i}"], out[f"radius_{i}"], out[f"out_{i}"] = m, np.array(radius), enhance(m, radius).astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, "region_prompts.npz"), **out)
    print({k: (v.shape, int(v.sum())) for k, v in out.items() if k.startswith("out_")})


if __name__ == "__main__":
    main()
