# The round-6 gpurun calls that were issued inline (the others: tools/gpu_r06{a,b,d,e,g,h}.sh, gpu_r06_profile.sh, gpu_r06_final.sh), as run:
# r06c  K-loop ablation of the generic slice loop (stamped build: python tools/experiments/gemm_timeline.py --build)
python tools/experiments/gemm_timeline.py --ablate-mid > gpurun_out/r06c_mid_ablation.jsonl
# r06f  loader-wave blocks + split-K (experiment forms 4412 / 4413)
python -m pytest tests/test_2_gemm.py -m gpu -q -x -k mid; python tools/bench_gemm_x3.py --midsplit gpurun_out/r06f_midsplit_sweep.json
# r06   the 78-input parity run under gate version 4 with per-image knife-edge entries
python tools/parity_wide.py --modes f16x3 --sets panoptic:1024:1:0-15,referring:640:4:3-15,region:1024:2:3-7 --out gpurun_out/r06_parity_wide.jsonl > gpurun_out/r06_parity_wide_summary.json
# r06   referring seed 10: bucketing A/B and the stage bisection (product stage outputs into the oracle's stages and back)
python tools/experiments/r06_seed10_bucket_ab.py 10 > gpurun_out/r06_seed10_bucket_ab.jsonl
python tools/experiments/r06_seed10_stage_bisect.py 10 2 > gpurun_out/r06_seed10_stage_bisect.jsonl
# r06i  the knife-edge GPU tests
python -m pytest tests/test_9_e2e_gpu.py -m gpu -q -x -k "knife_edge or config3"
# r06j  256 x 256 loader-wave form against the phased loop (instantiation removed afterwards)
python tools/experiments/r06_phi_lw256.py
# r06   tile choice of the mask GEMM M100 N65536 K256; top-k timing
python tools/experiments/r06_mask_gemm_tiles.py; python tools/experiments/r06_topk_time.py
