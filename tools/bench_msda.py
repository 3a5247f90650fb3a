"""MSDeformAttn fused gather micro-benchmark at the 1024^2 pixel-decoder shape (S = 21504, 8 heads x 32, 3 levels x 4 points), fp32 and
bf16 value / output.      python tools/bench_msda.py"""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from psalm_amd.hip_ops import get_ops

ops = get_ops()
shapes = [(32, 32), (64, 64), (128, 128)]
starts = [0, 1024, 5120]
S, M, D = 21504, 8, 32
g = torch.Generator().manual_seed(0)
for pol, name in ((1, "quad_shared_taps"), (0, "taps_per_lane")):
  ops.msda_policy(pol)
  out = {"kernel": name}
  for dt in (torch.float32, torch.bfloat16):
      value = torch.randn(1, S, M * D, generator=g).to(dt).cuda()
      ow = (torch.randn(1, S, M * 3 * 4 * 3, generator=g) * 2.0).cuda()
      for _ in range(5):
          o = ops.msda_fused(value, shapes, starts, ow, M, out_dtype=dt)
      torch.cuda.synchronize()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record()
      for _ in range(50):
          o = ops.msda_fused(value, shapes, starts, ow, M, out_dtype=dt)
      e1.record()
      torch.cuda.synchronize()
      us = e0.elapsed_time(e1) / 50 * 1e3
      esz = 4 if dt == torch.float32 else 2
      nbytes = S * (M * D * esz + M * 36 * 4 + M * D * esz)
      out[str(dt)] = {"us": round(us, 2), "algorithmic_MB": round(nbytes / 1e6, 1), "GBps": round(nbytes / us / 1e3, 1), "frac_of_8TBps": round(nbytes / us / 1e3 / 8000, 4),
                      "checksum": float(o.float().sum())}
  print(json.dumps(out), flush=True)

ops.msda_policy(1)
