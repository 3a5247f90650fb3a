"""r06: the in-launch stream-K hand-off (PSALM_TUNE_GEMM_STREAMK) on the Phi [k|v|q|fc1] launch -- M 899, N 14336, K 2048: 224 tiles of
256 x 256 on 256 compute units -- timed against the kernel without it, per number of K slices handed to the helper blocks, in the two
forms the image runs (paired split-f16 output behind gelu_new; plain fp32 output).  Also checks the outputs against each other.
    python tools/experiments/r06_streamk.py [out.json]
Needs a library built with tools/experiments/r06_streamk.patch applied (the hand-off was measured slower and is not in the product:
profiles/r06m_streamk_handoff.json)."""
import json, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import psalm_amd.hip_ops as H
from psalm_amd.hip_ops import get_ops


def split_bound_par(w, bias, g1, g0):
    return torch.tensor([2.0 ** 14 * float(w.abs().sum(1).max()), float(bias.abs().max()), g1, g0], dtype=torch.float32)


def timed(fn, n=40, warm=8):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(n))
    return {"median_us": ts[n // 2], "min_us": ts[0], "p90_us": ts[int(n * 0.9)]}


def main():
    ops = get_ops()
    d = ops.device
    out = {}
    M, H_, I_ = 899, 2048, 8192
    N, K = 3 * H_ + I_, H_
    g = torch.Generator().manual_seed(1)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) * 0.02
    bias = torch.randn(N, generator=g) * 0.1
    col_start, col_off = 3 * H_, H_
    Ns = N - col_start
    Kp_out = H_ + I_
    par = split_bound_par(w[col_start:], bias[col_start:], 2.0 ** 14 * 3.0, 0.5).to(d)
    perm = torch.cat([torch.arange(col_start), col_start + H.Ops.so_pair_perm(Ns)])
    asp, wsp_p, wsp = ops.split_f16(a.to(d)), ops.split_f16(w[perm].to(d)), ops.split_f16(w.to(d))
    bq, bd = bias[perm].to(d), bias.to(d)
    so = torch.zeros(M, 2 * Kp_out, dtype=torch.float16, device=d)
    inv = torch.zeros(M, device=d)
    big = torch.zeros(M, col_start, device=d)

    def paired():
        ops.gemm_x3_split(asp, wsp_p, bq, H.ACT_GELU_NEW, so, inv, par, split_col_off=col_off, split_col_start=col_start, act_col_start=col_start,
                          out=big, global_rows=True, paired=True)

    def plain():
        return ops.gemm_x3(asp, wsp, bd, None, H.ACT_NONE)

    ref = {}
    for name, fn in (("paired_split_output", paired), ("fp32_output", plain)):
        rows = {}
        for s in (0, 1, 4, 5, 6, 7, 8, 9, 10):
            ops.set_tuning(H.Ops.TUNE_GEMM_STREAMK, s)
            r = fn()
            torch.cuda.synchronize()
            kern = ops.gemm_last_kernel()
            if name == "paired_split_output":
                cur = (so.clone(), inv.clone(), big.clone())
            else:
                cur = (r.clone(),)
            t = timed(fn)
            t["kernel"] = kern
            t["status"] = ops.gemm_workspace_status()
            if s == 0:
                ref[name] = cur
            else:
                if name == "paired_split_output":
                    v0 = ref[name][0][:, col_off:col_off + Ns].double() + ref[name][0][:, Kp_out + col_off:Kp_out + col_off + Ns].double()
                    v1 = cur[0][:, col_off:col_off + Ns].double() + cur[0][:, Kp_out + col_off:Kp_out + col_off + Ns].double()
                    t["max_operand_diff_scaled_units"] = float((v0 - v1).abs().max())       # scaled values < 2^14
                    t["inv_equal"] = bool(torch.equal(ref[name][1], cur[1]))
                    t["fp32_cols_max_rel_diff"] = float((ref[name][2] - cur[2]).abs().max() / ref[name][2].abs().max())
                else:
                    t["max_rel_diff"] = float((ref[name][0] - cur[0]).abs().max() / ref[name][0].abs().max())
            rows["off" if s == 0 else ("auto" if s == 1 else f"slices_{s}")] = t
            print(name, s, json.dumps(t), flush=True)
        out[name] = rows
    ops.set_tuning(H.Ops.TUNE_GEMM_STREAMK, H.Ops.GEMM_STREAMK_DEFAULT)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
