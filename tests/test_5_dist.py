"""N>1 path on CPU: two gloo ranks, weights broadcast from rank 0, images sharded round-robin, metric all-reduce.
(The GPU path is identical with backend "nccl" = RCCL; bench.py --gpus N uses these same functions.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _FakeModel:
    def __init__(self, seed):
        g = torch.Generator().manual_seed(seed)
        self.w = {"a.w": torch.randn(37, 16, generator=g).to(torch.bfloat16), "a.b": torch.randn(16, generator=g),
                  "b.w": torch.randn(5, 8, generator=g).to(torch.bfloat16), "tab": torch.randn(3, 3, generator=g)}


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from psalm_amd.dist import broadcast_weights, reduce_metrics, shard_indices
    m = _FakeModel(seed=rank)                      # ranks start with DIFFERENT weights
    nbytes, _ = broadcast_weights(m, src=0, bucket_bytes=256, direct_bytes=1000)   # tiny buckets: the multi-bucket path; a.w (1184 B) goes in place
    ref = _FakeModel(seed=0)
    same = all(torch.equal(m.w[k], ref.w[k]) for k in ref.w)
    mine = shard_indices(7, rank, world)
    v = reduce_metrics(torch.tensor([float(len(mine)), float(sum(mine)), 1.0]))
    q.put((rank, same, nbytes, mine, v.tolist()))
    dist.destroy_process_group()


@pytest.mark.slow
def test_two_rank_broadcast_shard_reduce():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=120) for _ in ps)
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    expect_bytes = sum(t.numel() * t.element_size() for t in _FakeModel(0).w.values())
    assert [r[1] for r in res] == [True, True]
    assert all(r[2] == expect_bytes for r in res)
    assert res[0][3] == [0, 2, 4, 6] and res[1][3] == [1, 3, 5]          # round-robin, disjoint, complete
    assert res[0][4] == res[1][4] == [7.0, 21.0, 2.0]


def _worker_real_model(rank, world, port, q):
    """Real tiny PSALM on each rank (kernels in the host emulator), weights built from DIFFERENT seeds, then overwritten by rank 0's
    through broadcast_weights -- in both weight layouts (plain fp32 tensors and the f16x3 mode's split-f16 pairs) -- and each rank runs
    eval_seg on ITS shard of the images; the per-rank IoU meters are combined by the all-reduce."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests"), os.path.join(root, "tests", "emu")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ops_backend import make_ops
    from psalm_amd import evalout as E
    from psalm_amd.config import PsalmConfig
    from psalm_amd.dist import broadcast_weights, shard_indices
    from psalm_amd.model import PSALM
    from psalm_amd.synthetic import make_inputs, make_state_dict
    ops = make_ops("emu")
    cfg = PsalmConfig.tiny("referring")
    out = {"rank": rank}
    from psalm_amd.dist import weights_checksum
    for precision in ("fp32", "f16x3"):
        model = PSALM(cfg, make_state_dict(cfg, seed=100 + rank), ops=ops, precision=precision)        # different weights per rank
        before = weights_checksum(model)
        nbytes, _ = broadcast_weights(model, src=0, bucket_bytes=1 << 16)
        out[precision] = (before, weights_checksum(model), int(nbytes))        # (the parent holds rank 0's checksum from its own build)
    # images sharded round-robin: 2 images -> rank 0 gets {0}, rank 1 gets {1}; same pixels whoever computes them
    meters = E.IoUMeters()
    mine = shard_indices(2, rank, world)
    digest = []
    for i in mine:
        inputs = make_inputs(cfg, "referring", size=96, batch=1, seed=40 + i)
        r = model.eval_seg(**inputs)[0]
        inst = r["instances"]
        top = int(inst.scores.argmax())
        gt = (torch.rand(1, 96, 96, generator=torch.Generator().manual_seed(i)) < 0.3).to(torch.uint8)
        inter, union, _ = E.iou_counts(inst.pred_masks, gt, [(top, 0)], ops=ops)
        meters.update(inter, union)
        digest.append((i, float(r["mask_pred"].double().sum())))
    meters.all_reduce()
    out["mine"], out["digest"], out["meters"] = mine, digest, meters.results()
    q.put(out)
    dist.destroy_process_group()


@pytest.mark.slow
def test_two_rank_real_model_broadcast_and_sharded_eval():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
    from ops_backend import make_ops
    make_ops("emu")                                   # build the host-emulation library ONCE here: the two workers only load it
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_real_model, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    # while the ranks run: what a single process holds / computes with rank 0's weights (seed 100)
    from psalm_amd.config import PsalmConfig
    from psalm_amd.dist import weights_checksum
    from psalm_amd.model import PSALM
    from psalm_amd.synthetic import make_inputs, make_state_dict
    cfg = PsalmConfig.tiny("referring")
    ops = make_ops("emu")
    csum = {pr: weights_checksum(PSALM(cfg, make_state_dict(cfg, seed=100), ops=ops, precision=pr)) for pr in ("fp32",)}
    m = PSALM(cfg, make_state_dict(cfg, seed=100), ops=ops, precision="f16x3")
    csum["f16x3"] = weights_checksum(m)
    want = {i: float(m.eval_seg(**make_inputs(cfg, "referring", size=96, batch=1, seed=40 + i))[0]["mask_pred"].double().sum()) for i in range(2)}
    res = sorted((q.get(timeout=600) for _ in ps), key=lambda d: d["rank"])
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    for pr in ("fp32", "f16x3"):                       # both weight layouts: plain fp32 tensors and split-f16 pairs + scales
        assert res[0][pr][0] == csum[pr] and res[1][pr][0] != csum[pr]          # rank 1 started from other weights ...
        assert res[0][pr][1] == csum[pr] and res[1][pr][1] == csum[pr]          # ... and holds rank 0's after the broadcast
        assert res[0][pr][2] == res[1][pr][2] > 0
    assert res[0]["mine"] == [0] and res[1]["mine"] == [1]
    assert res[0]["meters"] == res[1]["meters"] and res[0]["meters"]["n"] == 2          # all-reduced: every rank holds the global meters
    got = dict(res[0]["digest"] + res[1]["digest"])
    assert got == want                                 # the shards are what a single process computes for the same images


def _worker_four(rank, world, port, q):
    """bench.py's start-up protocol on 4 ranks: rank 0 owns the checkpoint, the others build their arena from shape-only placeholders,
    one broadcast, a checksum MIN / MAX all-reduce proves the arenas identical; then 5 images (not a multiple of 4) round-robin."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "tests"), os.path.join(root, "tests", "emu")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ops_backend import make_ops
    from psalm_amd.config import PsalmConfig
    from psalm_amd.dist import broadcast_weights, check_weights_identical, shard_indices, weights_checksum
    from psalm_amd.model import PSALM
    from psalm_amd.synthetic import make_inputs, make_state_dict
    ops = make_ops("emu")
    cfg = PsalmConfig.tiny("panoptic")
    model = PSALM(cfg, make_state_dict(cfg, seed=7, shapes_only=rank != 0), ops=ops, precision="f16x3")
    before = weights_checksum(model)
    same_before, _ = check_weights_identical(model)
    nbytes, _ = broadcast_weights(model, src=0, bucket_bytes=1 << 18)
    same_after, csum = check_weights_identical(model)
    mine = shard_indices(5, rank, world)
    digest = []
    for i in mine:
        inputs = make_inputs(cfg, "panoptic", size=64, batch=1, seed=60 + i, num_classes=9)
        r = model.eval_seg(**inputs)[0]
        digest.append((i, float(r["mask_pred"].double().sum()), int(r["panoptic_seg"][0].to(torch.int64).sum())))
    q.put({"rank": rank, "before": before, "same_before": same_before, "same_after": same_after, "csum": csum, "nbytes": nbytes, "mine": mine,
           "digest": digest})
    dist.destroy_process_group()


@pytest.mark.slow
def test_four_ranks_placeholder_arenas_broadcast_checksum_and_ragged_shards():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
    from ops_backend import make_ops
    ops = make_ops("emu")
    world, port = 4, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker_four, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    # the single-process answers are computed here WHILE the ranks run (the emulator is slow; the suite's wall time is what is saved)
    from psalm_amd.config import PsalmConfig
    from psalm_amd.model import PSALM
    from psalm_amd.synthetic import make_inputs, make_state_dict
    cfg = PsalmConfig.tiny("panoptic")
    m = PSALM(cfg, make_state_dict(cfg, seed=7), ops=ops, precision="f16x3")
    want = {}
    for i in range(5):
        r = m.eval_seg(**make_inputs(cfg, "panoptic", size=64, batch=1, seed=60 + i, num_classes=9))[0]
        want[i] = (float(r["mask_pred"].double().sum()), int(r["panoptic_seg"][0].to(torch.int64).sum()))
    res = sorted((q.get(timeout=900) for _ in ps), key=lambda d: d["rank"])
    for p in ps:
        p.join(120)
        assert p.exitcode == 0
    assert not any(r["same_before"] for r in res)                         # placeholders != rank 0's weights: the check can fail
    assert res[1]["before"] == res[2]["before"] == res[3]["before"] != res[0]["before"]
    assert all(r["same_after"] for r in res) and len({r["csum"] for r in res}) == 1 and res[0]["csum"] == res[0]["before"]
    assert [r["mine"] for r in res] == [[0, 4], [1], [2], [3]]
    got = {i: (a, b) for r in res for i, a, b in r["digest"]}
    assert got == want


@pytest.mark.slow
def test_bench_py_self_launch_two_ranks_gloo_emu():
    """VERDICT r04 weak #11 / "Next" #8: bench.py's OWN N > 1 path before the hardware runs it -- `python bench.py --gpus 2` re-executes itself
    under torch.distributed.run (127.0.0.1 rendezvous), rank 1 builds its arena from shape-only placeholders, host cores are sliced per rank,
    the weights arrive through broadcast_weights, the MIN / MAX checksum all-reduce proves the arenas identical, every rank times its own
    steps, the times are all_gather'ed, and rank 0 prints ONE JSON line LAST.  `--emu`: host-emulated kernels, the tiny architecture, gloo
    instead of RCCL -- the same script lines otherwise."""
    import json
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
    from ops_backend import make_ops
    make_ops("emu")                                   # build the host-emulation library once, here
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--emu", "--eager", "--precision", "fp32", "--steps", "2",
                        "--warmup", "1"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.strip()]
    line = json.loads(lines[-1])                       # the JSON line is the LAST line of the job's stdout
    assert sum(ln.lstrip().startswith("{") for ln in lines) == 1
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["warmup"] == 1 and line["scaling"] == "weak" and line["emu"] is True
    wb = line["weight_broadcast"]
    assert wb["world_size"] == 2 and wb["backend"] == "gloo" and wb["weights_identical"] is True and wb["bytes"] > 0
    pr = line["per_rank_images_per_s"]
    assert 0 < pr["min"] <= pr["max"]
    # whole-job value = images of ALL ranks / the slowest rank's time
    assert abs(line["value"] - 2 * line["steps"] / (line["ms_per_step"] * line["steps"] / 1e3)) < 1e-2 * line["value"] + 1e-3
    assert line["value"] <= 2 * pr["min"] * (1 + 1e-2) + 1e-3
    # --gpus N under a launcher of another size is refused, not silently mis-reported
    env2 = dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--emu", "--eager"], capture_output=True, text=True, timeout=120,
                        env=env2, cwd=root)
    assert r2.returncode != 0 and "WORLD_SIZE=3" in (r2.stderr + r2.stdout)
