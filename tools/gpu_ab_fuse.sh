# One GPU call: (1) the new split-output kernels against their tests on the hardware, (2) interleaved A/B of the f16x3 mode with / without
# the fused operand hand-over (same process, same box), with parity vs the exact-fp32 mode, (3) the per-stage operand-width experiment.
mkdir -p gpurun_out/ab
timeout 400 python -m pytest tests/test_2_gemm.py tests/test_1_ops.py -m gpu -q -x -p no:cacheprovider \
    -k "split_output or ln_split or window_attention or causal_attention or gemm_x3" > gpurun_out/ab/unit.log 2>&1
echo "unit rc=$? $(tail -1 gpurun_out/ab/unit.log)"
timeout 700 python tools/exp_modes.py 1024 "f16x3+overlap,f16x3+overlap+nofuse,f16x3+overlap,f16x3+overlap+nofuse" > gpurun_out/ab/modes.log 2> gpurun_out/ab/modes.err
echo "modes rc=$?"; grep -o '^f16x3[^ ]* {"ms_per_image_graph": [0-9.]*, "images_per_s": [0-9.]*' gpurun_out/ab/modes.log
grep -o '"iou_mean": [0-9.]*, "iou_min": [0-9.]*, "iou_pooled": [0-9.]*' gpurun_out/ab/modes.log | head -4
timeout 300 python tools/exp_stage_bits.py 1024 > gpurun_out/ab/stage_bits.log 2> gpurun_out/ab/stage_bits.err
echo "stage_bits rc=$?"; grep -v '^{' gpurun_out/ab/stage_bits.log | cut -c1-200
