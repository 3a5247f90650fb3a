import sys, torch
sys.path.insert(0, '.')
from psalm_amd.hip_ops import get_ops
ops = get_ops()
g = torch.Generator().manual_seed(1)
for (Q, C) in [(100, 133), (100, 1), (100, 847)]:
    vals = torch.rand(Q, C + 1, generator=g).softmax(-1).cuda()
    ms = torch.rand(Q, generator=g).cuda()
    thing = (torch.rand(C, generator=g) < 0.6).to(torch.int32).cuda()
    for _ in range(3): ops.topk_select(vals, C, 100, thing, ms)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.topk_select(vals, C, 100, thing, ms)
    e1.record(); torch.cuda.synchronize()
    print(Q, C, round(e0.elapsed_time(e1) / 20 * 1e3, 1), "us per call (incl. 4 memsets)")
