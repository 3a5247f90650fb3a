import torch, sys
sys.path.insert(0,'.')
from psalm_amd import hip_ops as H
ops = H.get_ops()
torch.manual_seed(0)
for (M,N,K,res,bias,act) in [(300,130,128,True,True,1),(300,130,128,False,False,0),(128,128,128,False,False,0),(256,256,128,True,False,0)]:
    a = torch.randn(M,K).bfloat16().cuda(); w=torch.randn(N,K).bfloat16().cuda()
    b = torch.randn(N).cuda() if bias else None
    r = torch.randn(M,N).cuda() if res else None
    got = ops.gemm(a,w,b,r,act,0,out_dtype=torch.float32).cpu().double()
    want = a.cpu().double()@w.cpu().double().t()
    if bias: want += b.cpu().double()
    if act: want = torch.relu(want)
    if res: want += r.cpu().double()
    err=(got-want).abs()
    print(M,N,K,res,bias,act,'maxerr',err.max().item(), 'bad frac', (err>1e-3).double().mean().item())
    bad=(err>1e-3)
    if bad.any():
        rows=bad.any(1).nonzero().view(-1); cols=bad.any(0).nonzero().view(-1)
        print(' bad rows', rows[:40].tolist(), '... n', len(rows)); print(' bad cols', cols[:40].tolist(), 'n', len(cols))
        i,j = bad.nonzero()[0].tolist(); print(' first', i,j, got[i,j].item(), want[i,j].item())
