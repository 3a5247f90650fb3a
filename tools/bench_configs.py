"""BASELINE.json configs as side-lines (bench.py stays on the headline config): throughput of the HIP path and parity
against the CPU oracle on the same seeded inputs, full-size architecture.

    python tools/bench_configs.py [--json out.json] [--skip-oracle] [--fp32]
  config 2  COCO-panoptic 1024x1024 batch=1            (bf16; --fp32 adds the exact-fp32 mode)
  config 3  RefCOCO referring 640x640 batch=4 (ragged sentences)
  config 5  interactive (region prompts) 1024x1024 batch=2 -- bf16 LLM and fp8 (OCP e4m3) LLM projections
"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
from psalm_amd.config import PsalmConfig  # noqa: E402
from psalm_amd.model import PSALM  # noqa: E402
from psalm_amd.synthetic import make_inputs, make_state_dict  # noqa: E402


def iou_stats(got, want):
    gm, wm = got["mask_pred"].cpu() > 0, want["mask_pred"] > 0
    inter = (gm & wm).flatten(1).sum(1).float()
    union = (gm | wm).flatten(1).sum(1).float()
    iou = torch.where(union > 0, inter / union.clamp(min=1), torch.ones_like(union))
    return {"mask_iou_mean": round(float(iou.mean()), 5), "mask_pixel_agreement": round(float((gm == wm).float().mean()), 6)}


def run(name, task, size, batch, precision, oracle=True, steps=8):
    cfg = PsalmConfig(seg_task=task)
    sd = make_state_dict(cfg, seed=0)
    model = PSALM(cfg, sd, precision=precision, use_graphs=True)
    inputs = make_inputs(cfg, task, size=size, batch=batch, seed=3)
    inputs["images"] = inputs["images"].cuda()
    torch.manual_seed(7)
    for _ in range(3):
        torch.manual_seed(7)
        out = model.eval_seg(**inputs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        torch.manual_seed(7)
        out = model.eval_seg(**inputs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    res = {"config": name, "task": task, "size": size, "batch": batch, "precision": precision, "ms_per_batch": round(dt * 1e3, 2),
           "images_per_s": round(batch / dt, 2)}
    if oracle:
        from oracle import psalm_oracle as O
        torch.set_num_threads(64)
        cin = make_inputs(cfg, task, size=size, batch=batch, seed=3)
        torch.manual_seed(7)
        t1 = time.perf_counter()
        want = O.eval_seg(sd, cfg, **cin)
        res["oracle_s"] = round(time.perf_counter() - t1, 1)
        res["parity"] = [iou_stats(out[b], want[b]) for b in range(batch)]
        if task == "panoptic":
            res["parity"][0]["semantic_argmax_agreement"] = round(float((out[0]["sem_seg"].argmax(0).cpu() == want[0]["sem_seg"].argmax(0)).float().mean()), 6)
            res["parity"][0]["panoptic_id_agreement"] = round(float((out[0]["panoptic_seg"][0].cpu() == want[0]["panoptic_seg"][0]).float().mean()), 6)
    print(json.dumps(res), flush=True)
    del model
    torch.cuda.empty_cache()
    return res


def main():
    oracle = "--skip-oracle" not in sys.argv
    out = [run("2: panoptic 1024 b1", "panoptic", 1024, 1, "bf16", oracle),
           run("3: referring 640 b4", "referring", 640, 4, "bf16", oracle),
           run("5*: region 1024 b2 (bf16 LLM)", "region", 1024, 2, "bf16", oracle),
           run("5: region 1024 b2 (fp8 e4m3 LLM projections)", "region", 1024, 2, "fp8", oracle),
           run("2': panoptic 1024 b1 (fp8 e4m3 LLM projections)", "panoptic", 1024, 1, "fp8", oracle)]
    if "--fp32" in sys.argv:
        out.append(run("2: panoptic 1024 b1 (exact fp32 mode)", "panoptic", 1024, 1, "fp32", oracle, steps=3))
    if "--json" in sys.argv:
        with open(sys.argv[sys.argv.index("--json") + 1], "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
