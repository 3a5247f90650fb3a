# r05c: (1) post-processing kernels of r05 (panoptic arg-max vec4, top-k digit selection, mask-score reduction) + single-product GEMM + gRefCOCO
# fusion on the hardware; (2) quick bench line with breakdown; (3) BASELINE configs 5 and 2 with the reduced-precision LLM side mode
# (llm_products = 1), bf16 for contrast, and both again on the contractive weight set.
set -x
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_1_ops.py tests/test_2_gemm.py tests/test_8_evalout.py tests/test_0_abi.py -m gpu -q -x -p no:cacheprovider -k "topk or panoptic or semantic or mask or single_product or grefcoco or abi or evalout" > gpurun_out/r05c_pytest_ops.log 2>&1; tail -3 gpurun_out/r05c_pytest_ops.log
timeout 300 python bench.py --no-side-modes --no-cpu-baseline --no-varied --breakdown gpurun_out/r05c_bench_breakdown.json > gpurun_out/r05c_bench_quick.json 2> gpurun_out/r05c_bench_quick.err; tail -1 gpurun_out/r05c_bench_quick.json | cut -c1-260
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05c_bench_breakdown.json"))
for k in ("psalm_panoptic", "psalm_topk_select", "psalm_semantic_from_masks_x3", "psalm_binarize_gather", "psalm_resize_planes", "psalm_causal_attention_f32_split"):
    print(k, d.get(k))
PY
timeout 900 python tools/bench_configs.py --only 5 --llm1 --seeds 2 --json gpurun_out/r05c_config5_llm1.json > gpurun_out/r05c_config5.log 2>&1
python - <<'PY'
import json
for r in json.load(open("gpurun_out/r05c_config5_llm1.json")):
    p = r.get("parity", {})
    print(r["config"], r["precision"], "llm_products", r["llm_products"], "img/s", r["images_per_s"], {k: p.get(k) for k in ("mask_iou_mean_min", "mask_iou_pooled_min", "mask_iou_mean_area_ge_64_min", "mask_pixel_agreement_min", "mask_logit_rel_err_max", "flipped_pixels_max")})
PY
timeout 1200 python tools/bench_configs.py --only 2 --llm1 --contractive 0.4 --seeds 2 --json gpurun_out/r05c_config2_llm1_contractive.json > gpurun_out/r05c_config2.log 2>&1
python - <<'PY'
import json
for r in json.load(open("gpurun_out/r05c_config2_llm1_contractive.json")):
    p = r.get("parity", {})
    print(r["config"], r["precision"], "llm_products", r["llm_products"], "scale", r["residual_branch_scale"], "img/s", r["images_per_s"], {k: p.get(k) for k in ("mask_iou_mean_min", "mask_iou_pooled_min", "semantic_argmax_agreement_min", "panoptic_id_agreement_min", "mask_logit_rel_err_max", "flipped_pixels_max")})
PY
tail -3 gpurun_out/r05c_config2.log | cut -c1-300
