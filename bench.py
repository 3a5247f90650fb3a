"""Headline benchmark: images/sec of PSALM panoptic inference at 1024x1024 on N MI355X (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W          (N>1: launched by torch.distributed.run, one rank per GPU)

A "step" is one full `eval_seg` on one synthetic 1024x1024 COCO-panoptic-shaped image per GPU: Swin-B -> projector ->
24-layer Phi-1.5 over image + 134 class-name groups + 100 seg queries -> MSDeformAttn pixel decoder -> 9-layer masked
decoder -> semantic / instance / panoptic post-processing at full resolution.  Random-init weights of the reference
architecture (seeded).  Images are independent, so N GPUs = N replicas of the weights (one RCCL broadcast at start-up)
each processing its own image: weak scaling, no data-path collective.  `--gpus N` with N > 1 and no torchrun environment
re-launches itself as N ranks under torch.distributed.run (127.0.0.1 rendezvous).

Default precision = "f16x3": the mode that MEETS the north star's parity bar (mask IoU within 1e-3 of the fp32 CPU reference,
identical labels) -- every GEMM in split-f16 arithmetic on the f16 matrix cores (fp32-class operands, fp32 accumulate), exact-fp32
norms / softmax / attention.  The bf16 mode is ~2.2x faster but does not meet that bar on this network, so it is
reported as a side line (`other_modes`), never as `value`.

Prints ONE JSON line (rank 0).  `roofline` describes the dominant kernel -- the un-split bf16 MFMA GEMM instantiation with the
largest share of the step -- from HIP events (on the launch stream) around every C-ABI launch in extra, instrumented eager steps
after the timed region; `traffic` = its HBM bytes per launch from the committed rocprofv3 PMC passes of this workload
(profiles/r01_pmc_hbm_traffic.json).  `cpu_baseline` is the oracle (CPU restatement of the reference, fp32) timed on this host
for one image of the same workload; `parity_vs_cpu_oracle` compares the timed run's last result with it.
"""
import argparse
import json
import os
import sys
import time

# RCCL's inter-process buffers travel as dmabuf handles on this stack; the legacy IPC mode fails with `hipIpcGetMemHandle: invalid
# argument`.  Set before the HIP / HSA runtime initialises (first GPU call) -- for ranks launched by the driver's own torchrun line too.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBPS = 8000.0         # HBM3E, same guide
# HBM bytes per launch per kernel from the rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of THIS workload at THIS round's kernels
# (tools/gpu_r06_profile.sh -> tools/make_traffic_json.py), and the SQ-counter fractions + clock of the same workload (one more --pmc pass,
# tools/make_sq_json.py).  Both are looked up under the EXACT template instantiation the library reports for the timed launch
# (psalm_gemm_last_kernel); a kernel the committed passes do not contain gets `traffic: null` / no `sq_counters` -- never another
# instantiation's numbers (r04 looked the PH8 = 4 kernel up under the PH8 = 3 name).
TRAFFIC_JSON = os.path.join(ROOT, "profiles", "r06_pmc_hbm_traffic.json")
SQ_JSON = os.path.join(ROOT, "profiles", "r06_sq_summary.json")
# the oracle's host-thread count: the fastest of the 8 / 16 / 32 / 64 sweep on the GPU box's host (tools/cpu_baseline_sweep.py ->
# profiles/r04_cpu_baseline_threads.json), and the reference's OWN eval_seg timed in the authoring container (profiles/r04_reference_cpu.json)
CPU_THREADS_JSON = os.path.join(ROOT, "profiles", "r04_cpu_baseline_threads.json")
REFERENCE_CPU_JSON = os.path.join(ROOT, "profiles", "r04_reference_cpu.json")


def varied_streams(model, cfg, size, n_panoptic, fixed_images_per_s):
    """Two input streams shaped like the reference's own evaluation loops, through the SAME model object (hipGraph replay):
      panoptic : `n_panoptic` inputs, each its own image / prompt tensors, whose un-padded box and original (height, width) follow COCO-like
                 sizes through the reference's eval transform (T.ResizeShortestEdge(size, size) + T.FixedSizeCrop, coco_panoptic_mapper.py:81-89;
                 crop / resize of the results at llava_phi.py:1418-1429) -- a few pixels of jitter make (almost) every geometry distinct;
      referring: batches of 4 at 640^2 (BASELINE.json configs[2]) whose sentences have 5-25 tokens (train_datasets.py:644-695), each batch
                 with its own crop boxes.
    Per stream: images/s over the whole stream after a 4-call warm-up, the graph-cache statistics of the timed part (signature misses =
    calls that ran eagerly or captured), and the same model's rate on ONE input of the stream repeated (the fixed-shape figure it is
    compared with)."""
    import random
    from psalm_amd.synthetic import COCO_LIKE_SIZES, make_inputs, resized_box
    rnd = random.Random(5)

    def geom(sz):
        h, w = COCO_LIKE_SIZES[rnd.randrange(len(COCO_LIKE_SIZES))]
        h, w = h - rnd.randrange(0, 24), w - rnd.randrange(0, 24)
        oh, ow = resized_box(h, w, sz)
        return (oh, ow, h, w)

    def run(stream, per_call):
        for inp in stream[:4]:
            model.eval_seg(**inp)
        torch.cuda.synchronize()
        st0 = dict(model.graph_stats)
        t0 = time.perf_counter()
        for inp in stream:
            model.eval_seg(**inp)
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        st = {k: model.graph_stats[k] - st0[k] for k in st0}
        one = stream[len(stream) // 2]
        for _ in range(3):
            model.eval_seg(**one)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(len(stream)):
            model.eval_seg(**one)
        torch.cuda.synchronize()
        fixed = per_call * len(stream) / (time.perf_counter() - t0)
        v = per_call * len(stream) / dt_
        return {"images_per_s": round(v, 3), "same_stream_fixed_shape_images_per_s": round(fixed, 3), "ratio_to_fixed_shape": round(v / fixed, 4),
                "calls": st["calls"], "signature_misses": st["eager"] + st["captures"], "replays": st["replays"],
                "signature_miss_rate": round((st["eager"] + st["captures"]) / max(st["calls"], 1), 4)}

    out = {"note": "side metric, not `value`: per-call inputs differ in crop box, original size and (referring) sentence length; every call makes "
                   "its own prompt / padding-mask tensors as a data loader would"}
    graphs0 = len(model._graphs)
    geos = [geom(size) for _ in range(n_panoptic)]
    stream = []
    for i, g_ in enumerate(geos):
        inp = make_inputs(cfg, "panoptic", size=size, batch=1, seed=100 + i, geometry=[g_])
        inp["images"] = inp["images"].cuda()
        stream.append(inp)
    out["panoptic"] = dict(run(stream, 1), inputs=len(stream), distinct_geometries=len(set(geos)), size=size,
                           ratio_to_value=None)
    out["panoptic"]["ratio_to_value"] = round(out["panoptic"]["images_per_s"] / fixed_images_per_s, 4)
    out["panoptic"]["graphs_added"] = len(model._graphs) - graphs0
    del stream
    torch.cuda.empty_cache()
    task0 = model.seg_task
    try:
        model.seg_task = "referring"                       # same weights; the task decides the prompt splice and the inference tail only
        graphs0 = len(model._graphs)
        stream, lens_seen = [], []
        for i in range(max(4, n_panoptic // 4)):
            lens = [rnd.randint(5, 25) for _ in range(4)]
            lens_seen += lens
            inp = make_inputs(cfg, "referring", size=640, batch=4, seed=300 + i, refer_lens=lens, geometry=[geom(640) for _ in range(4)])
            inp["images"] = inp["images"].cuda()
            stream.append(inp)
        out["referring"] = dict(run(stream, 4), batches=len(stream), batch=4, size=640, sentence_tokens=[min(lens_seen), max(lens_seen)],
                                graphs_added=len(model._graphs) - graphs0)
    finally:
        model.seg_task = task0
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "bf16", "fp32"])
    ap.add_argument("--no-side-modes", action="store_true", help="skip the bf16 side-line measurement (rank 0, N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--parity-seeds", type=int, default=5, help="inputs the parity leg compares with the CPU oracle (rank 0, N=1; ~12 s of CPU each)")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from Python instead of replaying the captured hipGraph")
    ap.add_argument("--breakdown", default=None, help="write the per-kernel time breakdown JSON here")
    ap.add_argument("--gemm-policy", default="", help="comma-separated psalm_gemm_set_tile_policy codes applied before the first call (kernel A/B runs)")
    ap.add_argument("--tuning", default="", help="comma-separated key=value pairs for psalm_set_tuning (PSALM_TUNE_* of include/psalm_hip.h; kernel A/B runs)")
    ap.add_argument("--set", default="", help="comma-separated attr=int pairs set on the PSALM object before the first call (A/B switches such as kv_side=0)")
    ap.add_argument("--force-dist", action="store_true",
                    help="N = 1: still run init_process_group('nccl'), the weight broadcast, the checksum all-reduce and the barriers (RCCL dry run on one GPU)")
    ap.add_argument("--no-overlap", action="store_true", help="single stream (for kernel traces / PMC passes: per-kernel durations undisturbed)")
    ap.add_argument("--no-varied", action="store_true",
                    help="skip the varied-input-stream leg (rank 0, N=1): 64 panoptic inputs with COCO-like crop boxes / original sizes and a referring "
                         "stream with 5-25-token sentences -> other_modes.varied (images/s, signature misses)")
    ap.add_argument("--varied-n", type=int, default=64, help="panoptic inputs of the varied stream (the referring stream has a quarter as many batches of 4)")
    ap.add_argument("--graph-tail", action="store_true", help="A/B: put the per-image crop / resize / inference tail back into the captured graph (r04 behaviour)")
    ap.add_argument("--emu", action="store_true",
                    help="TEST MODE (tests/test_5_dist.py): host-emulated kernels (tests/emu), the tiny architecture, backend gloo -- exercises this "
                         "script's launcher / affinity / placeholder-rank / broadcast / checksum / all_gather / one-JSON-line path for N > 1 on a "
                         "machine without GPUs.  Its numbers mean nothing; the line says \"emu\": true")
    args = ap.parse_args()
    emu = args.emu

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # stand-alone multi-GPU launch: one rank per GPU under torch.distributed.run (the driver's own launch line sets WORLD_SIZE
        # and skips this); HSA_ENABLE_IPC_MODE_LEGACY=0 is required for RCCL's dmabuf IPC on this stack
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))

    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world} (launch with --nproc-per-node {args.gpus}, or without torchrun)")
    if world > 1 and hasattr(os, "sched_setaffinity"):
        # keep each rank's host threads on its own slice of the cores (contiguous slices follow the NUMA layout of the 2-socket hosts)
        try:
            cores = sorted(os.sched_getaffinity(0))
            per = max(1, len(cores) // world)
            os.sched_setaffinity(0, set(cores[local_rank * per:(local_rank + 1) * per]) or set(cores))
        except OSError:
            pass
    if not emu:
        torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:                          # --force-dist without a launcher: a one-rank rendezvous on the loopback
            import socket
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(so.getsockname()[1]), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        torch.set_num_threads(max(1, (os.cpu_count() or 8) // world))      # N ranks build their (seeded) weights concurrently on the host
        t_pg = time.perf_counter()
        if emu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        t_pg = time.perf_counter() - t_pg
        assert dist.get_world_size() == args.gpus

    from psalm_amd.config import PsalmConfig
    from psalm_amd.dist import broadcast_weights, check_weights_identical
    from psalm_amd.model import PSALM
    from psalm_amd.synthetic import make_inputs, make_state_dict

    cfg = PsalmConfig.tiny("panoptic") if emu else PsalmConfig(seg_task="panoptic")
    dev = "cpu" if emu else "cuda"
    ops = None
    if emu:
        sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
        import build_emu
        from psalm_amd.hip_ops import Ops
        ops = Ops(build_emu.build(verbose=False))
        args.size = 96
        args.no_side_modes = args.no_cpu_baseline = args.no_varied = True
    # rank 0 owns the (seeded) checkpoint; the other ranks build their weight arena from placeholders of the same shapes and receive
    # rank 0's prepared weights through the broadcast -- which is thereby real, not a copy of identical data onto itself
    sd = make_state_dict(cfg, seed=0, shapes_only=rank != 0)
    model = PSALM(cfg, sd, ops=ops, precision=args.precision, use_graphs=not args.eager)
    model.graph_tail = bool(args.graph_tail)
    for code in [int(c) for c in args.gemm_policy.split(",") if c]:
        model.ops.gemm_tile_policy(code)
    for kv in [c for c in args.tuning.split(",") if c]:
        model.ops.set_tuning(int(kv.split("=")[0]), int(kv.split("=")[1]))
    for kv in [c for c in getattr(args, "set").split(",") if c]:
        assert hasattr(model, kv.split("=")[0]), kv
        setattr(model, kv.split("=")[0], type(getattr(model, kv.split("=")[0]))(int(kv.split("=")[1])))
    # Results are consumed (here: dropped) before the next step, as the reference's eval loop does (evaluator.process right after
    # eval_seg): hand out the graph's own output buffers instead of a private ~1 GB copy per image (see PSALM.graph_outputs).
    model.graph_outputs = "alias"
    if args.no_overlap:
        model.overlap_streams = False
    bcast = None
    if use_dist:
        nbytes, secs = broadcast_weights(model, src=0)          # RCCL over xGMI, one-off
        same, csum = check_weights_identical(model, force=args.force_dist)   # MIN / MAX all-reduce of a checksum over every weight byte
        bcast = {"bytes": int(nbytes), "seconds": round(secs, 4), "GB_per_s": round(nbytes / max(secs, 1e-9) / 1e9, 1),
                 "weights_identical": bool(same), "checksum": csum, "backend": dist.get_backend(), "world_size": world,
                 "init_process_group_seconds": round(t_pg, 3)}
        if not same:
            raise SystemExit(f"bench.py: rank {rank}: weights differ across ranks after the broadcast")
        # RCCL writes a version banner to the C stdout at its first collective (r04a: "RCCL version : 2.26.6 ..." landed BEHIND the JSON line,
        # flushed at exit): flush it now, on every rank, so that the JSON line is the last thing this job prints
        import ctypes
        ctypes.CDLL(None).fflush(None)
        if rank != 0:
            del sd                                              # placeholders are not needed any more
    inputs = make_inputs(cfg, "panoptic", size=args.size, batch=1, seed=rank, **({"num_classes": 9} if emu else {}))
    inputs["images"] = inputs["images"].to(dev)                 # inputs resident in HBM before the timed region

    def sync():
        if not emu:
            torch.cuda.synchronize()

    def barrier():
        sync()
        if use_dist:
            dist.barrier()
        sync()

    if not args.eager:
        for _ in range(2):                                      # 1st call eager, 2nd captures the hipGraph (one-off set-up)
            model.eval_seg(**inputs)
    for _ in range(args.warmup):
        model.eval_seg(**inputs)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = model.eval_seg(**inputs)
    barrier()
    dt = time.perf_counter() - t0
    model_graph_stats = dict(model.graph_stats) if not args.eager else None
    # The parity leg compares the TIMED loop's own last result with the oracle.  With graph_outputs = "alias" that result's `mask_pred` is
    # the captured graph's own buffer, which every later replay through this model overwrites (the varied-stream leg feeds OTHER inputs:
    # r05a compared seed 0's oracle with the last varied image) -- so it is copied here, right behind the timed region.
    timed_out = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        o_ = out[0]
        timed_out = [{"mask_pred": o_["mask_pred"].clone(), "sem_seg": o_["sem_seg"].clone(),
                      "panoptic_seg": (o_["panoptic_seg"][0].clone(), list(o_["panoptic_seg"][1]))}]
    per_rank = None
    if use_dist:
        mine = torch.tensor([dt], device=dev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)                            # each rank's own wall time for its K steps
        per_rank = [args.steps / float(t_.item()) for t_ in every]
        dt = max(float(t_.item()) for t_ in every)              # the job is done when its slowest rank is

    # ---- is the host on the critical path?  GPU time of one step = K calls issued back to back WITHOUT the one host read-back per image
    # (`_finalize` replaced by a pass-through: graph replay + the crop / resize / inference tail, nothing the host waits for), bracketed by
    # events on the launch stream; `ms_per_step` - `gpu_ms_per_step` is what prompt handling, the blob upload and the result read-back add
    gpu_ms = None
    if not args.eager and rank == 0 and not emu:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        model._finalize = lambda r_, info_: r_
        try:
            model.eval_seg(**inputs)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(args.steps):
                model.eval_seg(**inputs)
            e1.record()
            torch.cuda.synchronize()
            gpu_ms = e0.elapsed_time(e1) / args.steps
        finally:
            del model._finalize

    # ---- side metric, NOT `value`: two images in flight on one GPU.  `value` above is the reference's own usage (one synchronous eval_seg
    # at a time); a serving loop can drive a second `PSALM.replica()` (shared weights, own buffers / graphs) from a second host thread on a
    # second stream, and the hardware fills one image's partial waves and latency-bound launches with the other's (r03a: +11 %).
    inflight = None
    if rank == 0 and world == 1 and not args.eager and not args.no_side_modes and args.precision == "f16x3" and not emu:
        try:                                                    # an auxiliary leg must never cost the run its JSON line
            import threading
            rep = model.replica()
            rep.graph_outputs = "alias"
            st = [torch.cuda.Stream(), torch.cuda.Stream()]
            inputs_b = make_inputs(cfg, "panoptic", size=args.size, batch=1, seed=rank + 1)
            inputs_b["images"] = inputs_b["images"].cuda()
            with torch.cuda.stream(st[1]):
                for _ in range(3):                                     # eager, capture, one replay
                    out_b = rep.eval_seg(**inputs_b)
            torch.cuda.synchronize()
            ref_b = model.eval_seg(**inputs_b)                          # same image through the first instance: bit-identical results expected
            torch.cuda.synchronize()
            same = bool(torch.equal(ref_b[0]["mask_pred"], out_b[0]["mask_pred"]) and torch.equal(ref_b[0]["sem_seg"], out_b[0]["sem_seg"]) and
                        torch.equal(ref_b[0]["panoptic_seg"][0], out_b[0]["panoptic_seg"][0]))

            def worker(m, inp, stream, k, bar):
                with torch.cuda.stream(stream):
                    bar.wait(timeout=120)
                    for _ in range(k):
                        m.eval_seg(**inp)
                    stream.synchronize()
            rates = []
            for _ in range(3):
                bar = threading.Barrier(3)
                th = [threading.Thread(target=worker, args=(m_, i_, s_, args.steps, bar)) for m_, i_, s_ in ((model, inputs, st[0]), (rep, inputs_b, st[1]))]
                for t_ in th:
                    t_.start()
                torch.cuda.synchronize()
                bar.wait(timeout=120)
                t1 = time.perf_counter()
                for t_ in th:
                    t_.join()
                torch.cuda.synchronize()
                rates.append(2 * args.steps / (time.perf_counter() - t1))
            inflight = {"images_in_flight": 2, "images_per_s": round(sorted(rates)[1], 3), "runs": [round(r_, 2) for r_ in rates],
                        "replica_results_identical": same,
                        "note": "two PSALM instances (shared weights, own graphs) driven by two host threads on two HIP streams; not `value`"}
            del rep, out_b, ref_b
        except Exception as ex:  # noqa: BLE001
            inflight = {"error": f"{type(ex).__name__}: {ex}"[:300]}
        torch.cuda.empty_cache()

    # ---- side metric, NOT `value`: does the throughput hold on a stream the reference's evaluation loops would feed?  (VERDICT r04 missing #2)
    varied = None
    if rank == 0 and world == 1 and not args.eager and not args.no_varied and not emu and args.precision == "f16x3":
        try:
            varied = varied_streams(model, cfg, args.size, args.varied_n, fixed_images_per_s=args.steps / dt)
        except Exception as ex:  # noqa: BLE001  (auxiliary leg)
            varied = {"error": f"{type(ex).__name__}: {ex}"[:300]}
        torch.cuda.empty_cache()

    # ---- instrumented steps (not part of `value`): HIP events (torch's current stream = the launch stream) around every
    # C-ABI launch, attributed to kernel instantiations through psalm_gemm_describe (the library's own selection function)
    roof = None
    if rank == 0 and not emu:
        recs = []
        model.use_graphs = False                                  # per-launch events need the eager launch path
        model.overlap_streams = False                             # ... and one stream: the timed region runs the pixel decoder concurrently with
        #                                                           the LLM (f16x3), which would stretch every per-launch duration measured here
        model.eval_seg(**inputs)
        model.ops.lib.records = recs
        nprof = 3
        for _ in range(nprof):
            model.eval_seg(**inputs)
        torch.cuda.synchronize()
        model.ops.lib.records = None
        # an event pair around NOTHING still measures ~5 us of stream overhead: measured here and used ONLY to rank the kernel
        # instantiations (reported times stay raw: for the long dominant kernel they agree with the rocprofv3 kernel trace)
        cal = []
        for _ in range(200):
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
            c1.record()
            cal.append((c0, c1))
        torch.cuda.synchronize()
        ev_over = sorted(c0.elapsed_time(c1) for c0, c1 in cal)[len(cal) // 2]
        # Per-launch event times are summed as  median over the launches of the same (entry point / kernel, shape)  x  their number: the
        # instrumented steps launch eagerly from Python, and ONE stalled launch (r03p: a single ~30 ms gap inside one event pair, nothing
        # in the rocprofv3 trace of the same command) otherwise moves a whole kernel's line.
        def small_ints(a_):
            return tuple(x for x in a_ if isinstance(x, int) and not isinstance(x, bool) and abs(x) < (1 << 24))

        def robust(ts):
            return sorted(ts)[len(ts) // 2] * len(ts)
        by_call, by_shape = {}, {}
        # ALGORITHMIC rows of the token-major GEMMs: the sequence runs at its bucketed length (PSALM.len_bucket: 899 -> 928 rows, the padding
        # positions are computed like the shorter prompts of a ragged batch), the flops that count are those of the real tokens
        lm = getattr(model, "_last_meta", None) or {}
        rows_padded = int(lm.get("B", 0)) * int(lm.get("L", 0))
        rows_real = int(sum(lm.get("lens", []) or [0]))

        def real_rows(geo_):
            return (rows_real, geo_[1], geo_[2]) if geo_ is not None and rows_padded and geo_[0] == rows_padded and rows_real < rows_padded else geo_
        for name, a, e0, e1, kname in recs:
            ms = e0.elapsed_time(e1)              # raw event time: agrees with the rocprofv3 kernel-trace durations for long kernels
            by_call.setdefault((name, small_ints(a)), []).append(ms)
            # (M, N, algorithmic K) of a matrix-core GEMM from its launch arguments; the KERNEL is named by the library itself
            # (psalm_gemm_last_kernel: the exact template instantiation, as a rocprofv3 kernel trace spells it)
            geo = None
            if name in ("psalm_gemm", "psalm_gemm_ln") and a[4] == 1:            # w_dtype == bf16 -> MFMA bf16 arithmetic
                geo = (a[12], a[13], a[14])
            elif name == "psalm_conv2d_nhwc":                                     # implicit GEMM: M = B*Ho*Wo, N = Cout, K = k*k*Cin
                B_, H_, W_, Cin, Cout, ks, st, pd_ = a[1], a[2], a[3], a[4], a[6], a[7], a[8], a[9]
                Ho, Wo = (H_ + 2 * pd_ - ks) // st + 1, (W_ + 2 * pd_ - ks) // st + 1
                geo = (B_ * Ho * Wo, Cout, ks * ks * Cin)
            elif name in ("psalm_gemm_x3", "psalm_gemm_x3_ln_split"):            # split-f16 GEMM: ALGORITHMIC flops 2 M N Kp of the fp32 product it stands for
                geo = (a[12], a[13], a[6])               # (A2, lda, a_scale, W2, ldw, w_scale, Kp, bias, residual, ldr, C, ldc, M, N, ...)
            elif name == "psalm_gemm_x3_split":
                geo = (a[10], a[11], a[6])               # (A2, lda, a_scale, W2, ldw, w_scale, Kp, bias, C, ldc, M, N, ...)
            if geo is not None and kname and "mfma" not in kname and ("glds" in kname or "skinny_kernel<float, true>" in kname or "gemm_bf16" in kname):
                by_shape.setdefault((kname, real_rows(geo)), []).append(ms)
        agg, shapes, kern = {}, {}, {}
        for (name, _sig), ts in by_call.items():
            d = agg.setdefault(name, [0, 0.0])
            d[0] += len(ts)
            d[1] += robust(ts)
        for (kname, (M, N, K)), ts in by_shape.items():
            kd = kern.setdefault(kname, [0, 0.0, 0.0])
            kd[0] += len(ts)
            kd[1] += robust(ts)
            kd[2] += 2.0 * M * N * K * len(ts)
            shapes[f"M{M} N{N} K{K} -> {kname}"] = [len(ts), robust(ts), 2.0 * M * N * K]
        breakdown = {k: {"launches_per_step": v[0] / nprof, "ms_per_step": v[1] / nprof} for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}
        breakdown["_gemm_shapes"] = {k: {"launches_per_step": v[0] / nprof, "ms_per_step": round(v[1] / nprof, 4),
                                         "TFLOPs": round(v[2] * v[0] / (v[1] * 1e-3) / 1e12, 1) if v[1] > 0 else None}
                                     for k, v in sorted(shapes.items(), key=lambda kv: -kv[1][1])}
        breakdown["_gemm_kernels"] = {k: {"launches_per_step": v[0] / nprof, "ms_per_step": round(v[1] / nprof, 4),
                                          "TFLOPs": round(v[2] / (v[1] * 1e-3) / 1e12, 1)} for k, v in sorted(kern.items(), key=lambda kv: -kv[1][1])}
        # HBM-bound kernels of the path (SURVEY §8d asks for both rooflines): algorithmic bytes per launch from the launch arguments
        #   semantic_from_masks: read (Q, HW) f32 logits once + write (C, HW) f32                      
        #   msda_fused: per (b, q): value row in (D*M) + offsets/logits (M*L*P*3 f32) + out row         (compulsory bytes)
        hbm = {}
        for name, a, e0, e1, _k in recs:
            if name == "psalm_semantic_from_masks":
                nbytes, kn = (a[5] + a[6]) * a[7] * 4, "semantic_from_masks_kernel"
            elif name == "psalm_semantic_from_masks_x3":                          # (mask, probsT, out, mask_score, workspace, Q, C, HW, Kpad, stream)
                nbytes, kn = (a[5] + a[6]) * a[7] * 4, "semantic_from_masks_x3_pair_kernel"
            elif name == "psalm_msda_fused":
                esz_v, esz_o = (2 if a[1] == 1 else 4), (2 if a[6] == 1 else 4)
                B_, S_, M_, D_, L_, P_ = a[7], a[8], a[9], a[10], a[11], a[12]
                nbytes, kn = B_ * S_ * (M_ * D_ * esz_v + M_ * L_ * P_ * 3 * 4 + M_ * D_ * esz_o), "msda_fused8_kernel"
            elif name == "psalm_resize_planes":                                   # (x, x_dtype, out, out_dtype, N, h, w, hc, wc, H, W, stream): read the crop, write the planes
                nbytes = a[4] * (a[7] * a[8] * (2 if a[1] == 1 else 4) + a[9] * a[10] * (2 if a[3] == 1 else 4))
                if nbytes < (64 << 20):
                    continue                                                      # only the full-resolution mask upsampling is a roofline-sized launch
                kn = "resize_planes_rows_kernel<8>"
            elif name == "psalm_panoptic":                                        # (... Q, HW ...): the call's dominant kernel reads the (Q, HW) f32 logits once
                nbytes, kn = a[10] * a[11] * 4 + a[11] * 8, "panoptic_argmax_kernel (+ the call's small kernels)"
            else:
                continue
            h = hbm.setdefault(kn, [0, [], 0])
            h[0] += 1
            h[1].append(e0.elapsed_time(e1))
            h[2] += nbytes
        for h in hbm.values():
            h[1] = robust(h[1])                                                   # (median per kernel x launches, as above)
        hbm_roof = None
        if hbm:
            tj = {}
            tpath = TRAFFIC_JSON
            if os.path.exists(tpath):
                with open(tpath) as f:
                    tj = json.load(f).get("kernels", {})
            hbm_roof = [{"kernel": k, "bound": "hbm", "achieved": round(v[2] / (v[1] * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                         "frac": round(v[2] / (v[1] * 1e-3) / 1e9 / PEAK_HBM_GBPS, 4), "traffic": next((v_.get("hbm_bytes_per_launch") for kk_, v_ in tj.items() if kk_.startswith(k.replace("_kernel", ""))), None),
                         "algorithmic_bytes_per_launch": v[2] // v[0], "launches_per_step": v[0] / nprof,
                         "avg_launch_us": round(v[1] / v[0] * 1e3, 2)} for k, v in sorted(hbm.items(), key=lambda kv: -kv[1][1])]
        for h_ in hbm_roof or []:
            if h_["kernel"].startswith("panoptic_argmax"):
                # (r04: found while reading the kernel for the SQ-counter table -- its `parked` share is LDS atomics and latency, not HBM)
                h_["note"] = ("UPPER BOUND: `achieved` counts all Q mask planes; the kernel (like LP:325-386) reads only the planes of the queries it keeps "
                              "(not void, score > 0.8: 50 of 100 on the seed-0 input, by the oracle), i.e. the true byte rate is Q / kept = ~2x lower; the "
                              "launch is bound by its per-pixel LDS atomics and load latency, not by HBM")
        if kern:
            # dominant kernel = the single-kernel (un-split) GEMM instantiation with the largest share of the step
            cands = {k: v for k, v in kern.items() if " + " not in k} or kern
            # ranking only: take the per-launch event overhead out, otherwise the instantiation with the most launches wins
            kname, (n, ms, fl) = max(cands.items(), key=lambda kv: kv[1][1] - kv[1][0] * ev_over)
            ach = fl / (ms * 1e-3) / 1e12
            all_ms = sum(v[1] for v in kern.values())
            all_fl = sum(v[2] for v in kern.values())
            traffic = None
            tpath = TRAFFIC_JSON     # rocprofv3 --pmc passes of this same workload
            if os.path.exists(tpath):                                            # (tools/gpu_final.sh + tools/make_traffic_json.py)
                with open(tpath) as f:
                    tj = json.load(f).get("kernels", {})
                traffic = tj.get(kname.split(" + ")[0], {}).get("hbm_bytes_per_launch")
            targs = [t.strip() for t in kname.split("<", 1)[1].split(">", 1)[0].split(",")] if "glds_kernel<" in kname else []
            x3_form = int(targs[9]) if len(targs) >= 11 else 0                   # template argument X3: 1 / 2 = split-f16 K-panel / slice form (3 products)
            is_x3 = x3_form != 0
            prods = 3                                                            # f16 products issued per algorithmic product
            roof = {"bound": "mfma", "kernel": kname, "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                    "note": ("split-f16 kernel: `achieved` counts the ALGORITHMIC 2*M*N*K of the fp32-class product; the kernel issues "
                             "3 f16 MFMA products (hi.hi + lo.hi + hi.lo) per algorithmic product, see `mfma_issue`") if is_x3 else None,
                    "mfma_issue": {"f16_product_equivalents": prods, "TFLOPs": round(prods * ach, 1),
                                   "frac_of_f16_peak": round(prods * ach / PEAK_BF16_TFLOPS, 4)} if is_x3 else None,
                    "launches_per_step": n / nprof, "avg_launch_us": round(ms / n * 1e3, 2),
                    "algorithmic_gflop_per_launch": round(fl / n / 1e9, 2),
                    "algorithmic_rows": ({"real_tokens": rows_real, "launched_rows": rows_padded,
                                          "note": "token-major GEMMs are launched on the bucketed sequence length; `achieved` counts the real tokens' flops"}
                                         if rows_padded and rows_real < rows_padded else None),
                    "share_of_step_ms": round(ms / nprof, 3), "event_pair_overhead_us": round(ev_over * 1e3, 2),
                    "all_mfma_gemms": {"ms_per_step": round(all_ms / nprof, 3), "TFLOPs": round(all_fl / (all_ms * 1e-3) / 1e12, 1),
                                       "gflop_per_step": round(all_fl / nprof / 1e9, 1), "launches_per_step": sum(v[0] for v in kern.values()) / nprof},
                    "hbm_bound_kernels": hbm_roof}
            if os.path.exists(SQ_JSON):                                          # committed SQ-counter pass of this workload (tools/make_sq_json.py)
                with open(SQ_JSON) as f:
                    sq = json.load(f)
                e = sq.get("kernels", {}).get(kname.split(" + ")[0])
                if e:
                    roof["sq_counters"] = dict(e, source=sq.get("source"), kernel_in_the_pass=kname.split(" + ")[0],
                                               note="fractions of the wavefront cycles (parked at s_waitcnt / s_barrier, issue-stalled, issuing) and of the chip's "
                                                    "matrix-pipe CYCLES; `frac` / `mfma_issue` are against the 2.4 GHz peak, this launch ran at clock_GHz")
        if args.breakdown:
            os.makedirs(os.path.dirname(os.path.abspath(args.breakdown)), exist_ok=True)
            with open(args.breakdown, "w") as f:
                json.dump(breakdown, f, indent=1)

    # ---- CPU baseline (rank 0, N=1 only): the oracle, one image of the same workload
    cpu = None
    parity = None
    side = None
    side_fp32 = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import psalm_oracle as O
        cores = min(os.cpu_count() or 1, 16)                     # default when no sweep file is present
        sweep = None
        if os.path.exists(CPU_THREADS_JSON):
            with open(CPU_THREADS_JSON) as f:
                sweep = json.load(f)
            cores = min(int(sweep.get("best_threads", cores)), os.cpu_count() or 1)
        torch.set_num_threads(cores)
        cin = make_inputs(cfg, "panoptic", size=args.size, batch=1, seed=rank)
        O.eval_seg(sd, cfg, **make_inputs(cfg, "panoptic", size=256, batch=1, seed=rank))   # warm-up (thread pool, allocator) on a small image
        tcs = []
        for _ in range(5):                                       # (r06: five samples and their spread in the line -- VERDICT r05 weak #9: 3 samples scattered 7.1 - 8.6 s)
            t1 = time.perf_counter()
            want = O.eval_seg(sd, cfg, **cin)
            tcs.append(time.perf_counter() - t1)
        tc = sorted(tcs)[2]
        ref_note = None
        if os.path.exists(REFERENCE_CPU_JSON):
            with open(REFERENCE_CPU_JSON) as f:
                rj = json.load(f)
            ref_note = (f"the reference's OWN eval_seg (unmodified source through tests/golden/ref_shim.py), same weights / input, timed in the authoring container "
                        f"(no /root/reference on the GPU box): {rj.get('seconds_per_image_median')} s per image on {rj.get('threads')} threads of {rj.get('cpu')} "
                        f"= {rj.get('images_per_s')} images/s (profiles/r04_reference_cpu.json)")
        cpu = {"value": round(1.0 / tc, 4), "unit": "images/s", "cores": cores, "kind": "port",
               "sample": f"1 image, {args.size}x{args.size} panoptic, full model, fp32; warm-up on a 256x256 image, then median of 5 timed runs "
                         f"({', '.join(f'{t:.1f}' for t in tcs)} s)",
               "seconds_per_image": {"min": round(min(tcs), 2), "median": round(tc, 2), "max": round(max(tcs), 2), "samples": len(tcs)},
               "value_range": [round(1.0 / max(tcs), 4), round(1.0 / min(tcs), 4)],
               "threads_chosen_by": (f"sweep on this host class (profiles/r04_cpu_baseline_threads.json: {sweep.get('seconds_by_threads')})" if sweep else "default"),
               "reference_itself": ref_note}
        # ---- parity gate, version 4 (oracle/parity_gate.py: definitions, the flip-margin property, the knife-edge list)
        from oracle import parity_gate as PG

        def judged(g_, w_, seed_):
            p_ = PG.parity_of(g_, w_)
            PG.judge(p_, g_, w_, PG.knife_edge_entry("panoptic", args.size, seed_, 0))
            return dict(p_, inputs_seed=seed_)
        parity = judged(timed_out[0], want[0], rank)
        # ... and over more inputs (same weights, other seeded images / prompts): one image is a noisy gate -- 0.3 % positive pixels, ~10
        # empty reference masks, masks of a few pixels whose IoU moves in steps of 1/area.  min / max over the seeds.
        per_seed = [dict(parity)]
        wants = {rank: want[0]}
        if args.parity_seeds > 1 and not args.eager:
            for s_ in range(1, args.parity_seeds):
                pin = make_inputs(cfg, "panoptic", size=args.size, batch=1, seed=rank + s_)
                w_s = O.eval_seg(sd, cfg, **pin)[0]
                wants[rank + s_] = w_s
                pin["images"] = pin["images"].cuda()
                g_s = model.eval_seg(**pin)[0]
                torch.cuda.synchronize()
                per_seed.append(judged(g_s, w_s, rank + s_))
            parity["seeds"] = {"n": len(per_seed), "inputs_seeds": [p_["inputs_seed"] for p_ in per_seed],
                               "mask_iou_mean_min": min(p_["mask_iou_mean"] for p_ in per_seed),
                               "mask_iou_pooled_min": min(p_["mask_iou_pooled"] for p_ in per_seed),
                               "mask_iou_mean_area_ge_64_min": min((p_["mask_iou_mean_area_ge_64"] for p_ in per_seed if p_["mask_iou_mean_area_ge_64"] is not None), default=None),
                               "mask_logit_rel_err_max": max(p_["mask_logit_rel_err"] for p_ in per_seed),
                               "flip_margin_rel_max": max(p_["flip_margin_rel_max"] for p_ in per_seed),
                               "semantic_argmax_agreement_min": min(p_["semantic_argmax_agreement"] for p_ in per_seed),
                               "panoptic_id_agreement_min": min(p_["panoptic_id_agreement"] for p_ in per_seed),
                               "flipped_mask_pixels_max": max(p_["flipped_mask_pixels"] for p_ in per_seed), "per_seed": per_seed}
        parity["meets_bar_pooled"] = all(p_["meets_bar_pooled"] for p_ in per_seed)
        parity["meets_bar_plain_mean"] = all(p_["meets_bar_plain_mean"] for p_ in per_seed)
        parity["flips_within_margin"] = all(p_["flips_within_margin"] for p_ in per_seed)
        parity["meets_north_star_bar"] = all(p_["passes_gate"] for p_ in per_seed)
        small = [p_["inputs_seed"] for p_ in per_seed if p_["flips_within_margin"] and not p_["meets_bar_plain_mean"]]
        if small:
            parity["note"] = (f"inputs {small}: every differing pixel is within the flip margin of the oracle's threshold, the plain mean over the 100 queries is "
                              "below 0.999 because a reference mask of a few pixels quantises its IoU in steps of 1 / area")
        parity["gate"] = PG.GATE
        side_fp32 = None
        if not args.no_side_modes and args.precision == "f16x3":
            # side line, reported as information (NOT part of any pass / fail decision since gate version 4): the exact-fp32 GPU mode (fp32
            # MFMA GEMMs -- the reference's arithmetic width in another summation order) on the same inputs: its throughput, and per seed how
            # far an input moves under re-ordering alone (the floor any re-implementation sits on).
            try:
                m32 = PSALM(cfg, sd, precision="fp32", use_graphs=not args.eager)
                m32.graph_outputs = "alias"
                for _ in range(2 + args.warmup):
                    o32 = m32.eval_seg(**inputs)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    o32 = m32.eval_seg(**inputs)
                torch.cuda.synchronize()
                t32 = time.perf_counter() - t1
                ctrl = []
                for sd_, w_s in wants.items():
                    pin = make_inputs(cfg, "panoptic", size=args.size, batch=1, seed=sd_)
                    pin["images"] = pin["images"].cuda()
                    ctrl.append(judged(m32.eval_seg(**pin)[0], w_s, sd_))
                    torch.cuda.synchronize()
                side_fp32 = {"value": round(args.steps / t32, 3), "unit": "images/s", "ms_per_step": round(t32 / args.steps * 1e3, 3),
                             "parity_vs_cpu_oracle": {"n": len(ctrl), "meets_bar_pooled": all(c_["meets_bar_pooled"] for c_ in ctrl),
                                                      "meets_bar_plain_mean": all(c_["meets_bar_plain_mean"] for c_ in ctrl),
                                                      "flips_within_margin": all(c_["flips_within_margin"] for c_ in ctrl), "per_seed": ctrl},
                             "note": "exact-fp32 MFMA GEMMs + fp32 attention: the reference's own arithmetic width in another summation order; side line, "
                                     "information only"}
                del m32, o32
                torch.cuda.empty_cache()
            except Exception as ex:  # noqa: BLE001  (auxiliary leg)
                side_fp32 = {"error": f"{type(ex).__name__}: {ex}"[:300]}
        if not args.no_side_modes and args.precision != "bf16":
            try:
                # side line: the bf16 fast mode on the same image (NOT `value`: it does not meet the parity bar on this network)
                del model, out
                torch.cuda.empty_cache()
                mb = PSALM(cfg, sd, precision="bf16", use_graphs=not args.eager)
                mb.graph_outputs = "alias"
                for _ in range(2 + args.warmup):
                    ob = mb.eval_seg(**inputs)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(args.steps):
                    ob = mb.eval_seg(**inputs)
                torch.cuda.synchronize()
                tb = time.perf_counter() - t1
                pb = PG.parity_of(ob[0], want[0])
                pb["n"] = 1
                side = {"bf16": {"value": round(args.steps / tb, 3), "unit": "images/s", "ms_per_step": round(tb / args.steps * 1e3, 3),
                                 "parity_vs_cpu_oracle": pb}}
            except Exception as ex:  # noqa: BLE001  (auxiliary leg)
                side = {"bf16": {"error": f"{type(ex).__name__}: {ex}"[:300]}}

    if rank == 0:
        L = None
        line = {
            "metric": "images/sec at 1024x1024 COCO-panoptic inference", "value": round(world * args.steps / dt, 3), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "gpu_ms_per_step": round(gpu_ms, 3) if gpu_ms is not None else None,
            "host_ms_per_step": round(dt / args.steps * 1e3 - gpu_ms, 3) if gpu_ms is not None else None,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": (f"BASELINE.json configs[1]: COCO-panoptic {args.size}x{args.size} batch=1 per GPU, PSALM (Swin-B + Phi-1.5 24L + Mask2Former head), "
                                    "134 class prompts, 100 queries, full semantic+instance+panoptic post-processing") if not emu else
                                   f"TEST MODE: the tiny architecture at {args.size}x{args.size} on host-emulated kernels (launcher / collective path only)",
                       "arithmetic": ("GEMMs in split-f16 (22-bit operands as hi + lo f16 pairs, three f16 MFMA products, fp32 accumulate)"
                                      "; fp32 norms / softmax / attention") if args.precision == "f16x3" else args.precision,
                       "parallelism": f"image-sharded x{world} (replicated weights, RCCL broadcast at init)",
                       "launch": "eager" if args.eager else "hipGraph replay (one graph per input signature)"},
            "roofline": roof, "cpu_baseline": cpu, "parity_vs_cpu_oracle": parity,
            "other_modes": ({**(side or {}), **({"fp32": side_fp32} if side_fp32 else {}), **({"varied": varied} if varied else {})} or None), "two_in_flight": inflight,
            "graph": dict(model_graph_stats, tail_in_graph=bool(args.graph_tail)) if model_graph_stats else None,
            **({"emu": True, "backend": "gloo" if use_dist else None} if emu else {}),
            "weight_broadcast": bcast,
            "per_rank_images_per_s": ({"min": round(min(per_rank), 3), "max": round(max(per_rank), 3)} if per_rank else None),
        }
    if use_dist:
        dist.barrier()                                           # rank 0's instrumented steps are done before anyone leaves
        dist.destroy_process_group()
    if rank == 0:
        import ctypes
        ctypes.CDLL(None).fflush(None)                           # anything the native libraries still hold in the C stdout buffer goes first
        sys.stdout.flush()
        print(json.dumps(line), flush=True)                      # ... and the ONE JSON line is the last line of the job


if __name__ == "__main__":
    main()
