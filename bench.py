#!/usr/bin/env python3
"""bench.py -- Mpix/s of dense pyramidal LK flow on MI355X (BASELINE.json metric), one JSON line on rank 0.

A step = one frame pair: load the new frame's level 0 (already resident in HBM), build its pyramid, run every
pyramid level coarse->fine against the previous frame's pyramid, swap.  That is main.cu:246-272 of the reference.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 4k|1080p|8k|vga] [--mode lk_float|compat_cpu]

N > 1 (launched by torch.distributed.run, one rank per GPU): the SAME frame pair is row-sharded over the ranks with
a halo exchange per pyramid level over RCCL (cuda_optical_flow_2_amd/parallel.py) -- strong scaling.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {  # BASELINE.json configs: (width, height, levels, window)
    "vga": (640, 480, 3, 5),
    "1080p": (1920, 1080, 4, 7),
    "4k": (3840, 2160, 5, 9),
    "8k": (7680, 4320, 6, 15),
}
BASELINE_ITERS = {"vga": 3, "1080p": 5, "4k": 5, "8k": 10}  # the "iters" of BASELINE.json's configs
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
LK_BYTES_PER_PX = 10   # fused level kernel: 2 u8 read + one (u,v) float pair written


def cpu_baseline(workload, w, h, levels, window):
    """The reference's own CPU path (oracle/_ref, kind 'reference') or the oracle port, 1 thread, bounded sample."""
    import numpy as np

    import oracle as orc
    from cuda_optical_flow_2_amd import synth

    # bounded sample: full pipeline on a pair whose size keeps the run at ~10-20 s on one core
    sw, sh = (w, h) if w * h <= 3840 * 2160 else (3840, 2160)
    p, n = synth.smooth_pair(sw, sh)
    p3, n3 = synth.to_3ch(p), synth.to_3ch(n)
    use_ref = orc.have_reference() and window == 9
    t0 = time.perf_counter()
    reps = 0
    while True:
        if use_ref:
            orc.Reference().flow_pair(p3, n3, levels)          # cpu::gauss_pyramid x2 + cpu::calc_optical_flow per level
        else:
            orc.Oracle().flow_pair(p3, n3, levels, window, "compat_cpu")
        reps += 1
        dt = time.perf_counter() - t0
        if dt > 10.0 or reps >= 8:
            break
    return {
        "value": round(sw * sh * reps / dt / 1e6, 3), "unit": "Mpix/s", "cores": 1,
        "kind": "reference" if use_ref else "port",
        "sample": f"{reps} pair(s) {sw}x{sh}, {levels} levels, window {window}x{window}, both pyramids + all levels, "
                  f"{'OptFlowCPU.cpp compiled as oracle/_ref' if use_ref else 'oracle/ofx_oracle.c (compat_cpu)'}, {dt:.1f} s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=104)
    ap.add_argument("--workload", default="4k", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="lk_float", choices=["lk_float", "compat_cpu"])
    ap.add_argument("--path", default="stream", choices=["stream", "staged", "plain"],
                    help="single-GPU execution path: stream pipeline (default), two-stream staged pairs, or the plain sequence")
    ap.add_argument("--iters", type=int, default=1,
                    help="refinement iterations per level (extension; 1 = the reference's algorithm). iters > 1 runs the plain path")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the additional legs reported under extra (profiling runs: only the timed configuration's launches)")
    ap.add_argument("--copy-frames", action="store_true",
                    help="stream path: the session keeps its own copy of level 0 of every frame instead of reading the caller's ring "
                         "of frames in place (ofx_params.borrow_frames = 0); the default run reports this variant under extra")
    ap.add_argument("--batch", type=int, default=0, choices=[0, 1, 2, 4, 8],
                    help="stream path: frames per launch (ofx_params.stream_batch); a step is still one frame.  0 = "
                         "engine.suggest_stream_batch: by the working set of the pipeline (4K: 4 on one GPU, 8 per rank of a sharded pair)")
    ap.add_argument("--shard-halo", default="recompute", choices=["recompute", "exchange"],
                    help="N > 1: halo rows of every level rebuilt from a wider level-0 halo (default) or exchanged with the neighbouring "
                         "ranks per level (RCCL send/recv; pair-at-a-time, implies --shard-corner broadcast)")
    ap.add_argument("--shard-corner", default="local", choices=["local", "broadcast"],
                    help="N > 1: where a rank gets the shift vectors from (local = its own top-left patch, no collective; "
                         "broadcast = rank 0's corner kernel + one RCCL broadcast per pair)")
    args = ap.parse_args()
    # A stream tick carries `batch` frames and a step is one frame: the timed K steps must be whole ticks, or frames would be
    # counted that were only queued.  Use the largest batch that divides K.
    args.borrow = not args.copy_frames
    if args.batch == 0:
        from cuda_optical_flow_2_amd.engine import suggest_stream_batch
        from cuda_optical_flow_2_amd.parallel import ShardPlan
        bw, bh, bl, bwin = WORKLOADS[args.workload]
        n_ranks = max(args.gpus, int(os.environ.get("WORLD_SIZE", "1")))
        args.batch = suggest_stream_batch(bw, bh, bl, ShardPlan(bw, bh, bl, bwin, 0, n_ranks) if n_ranks > 1 else None, args.borrow)
    while args.batch > 1 and (args.steps % args.batch or
                              args.batch * WORKLOADS[args.workload][2] > 40):  # OFX_MAX_LK_ITEMS: (pair, level) items per launch
        args.batch //= 2
    # (the warm-up is rounded UP to whole ticks -- a few more untimed steps -- so that only K constrains the frames per launch)
    warmup_steps = (args.warmup + args.batch - 1) // args.batch * args.batch

    import numpy as np
    import torch
    import torch.distributed as dist

    from cuda_optical_flow_2_amd import engine, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run"
    torch.cuda.set_device(local_rank)
    # Everything is enqueued on one explicit (non-null) HIP stream: the legacy null stream serialises against every other
    # stream of the process and costs several microseconds more per launch.
    work_stream = torch.cuda.Stream()
    torch.cuda.set_stream(work_stream)
    force_dist = os.environ.get("OFX_BENCH_FORCE_DIST") == "1"  # rehearsal: run the N > 1 driver (RCCL init, broadcast) on one rank
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # RCCL prints a version banner on stdout when its communicator is created; stdout carries only the JSON line, so
        # the banner is sent to stderr (fd-level: it comes from the C library)
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)

    w, h, levels, window = WORKLOADS[args.workload]
    # a short ring of resident frames: a smooth texture translating by (2,1) px per frame
    nframes = 4
    # experiments: OFX_BENCH_MOTION="mx,my" scales the per-frame translation (2,1) px; "0,0" = identical frames
    mx, my = (float(t) for t in os.environ.get("OFX_BENCH_MOTION", "1,1").split(","))
    frames = [synth.smooth_pair(w, h, 2.0 * i * mx, 1.0 * i * my)[1] for i in range(nframes)]
    d_frames = [torch.from_numpy(f).cuda() for f in frames]
    # The stream paths take their frames from a ring of DISTINCT device buffers, as a capture / decoder surface pool would
    # hand them over: long enough for ofx_params.borrow_frames (a buffer stays untouched for 3 * batch further submits), and
    # large enough that a frame is not still sitting in the 256 MB Infinity Cache when it comes round again (four buffers
    # would be: with borrowed frames that alone made the LK stage ~10 % faster).  Contents repeat every four buffers.
    ring_n = int(os.environ.get("OFX_BENCH_RING", "0")) or (3 * max(args.batch, 4) + 4 + 3) // 4 * 4  # (experiments: other ring sizes)
    d_ring = [d_frames[i % nframes] if i < nframes else d_frames[i % nframes].clone() for i in range(ring_n)]
    # the frames of one tick go down in one call (ofx_session_stream_submit_frames): step i of a tick only counts, the tick's
    # last step submits -- the per-frame FFI crossing is what limits small frames and the ranks of a sharded pair.
    # (tick j takes ring buffers j*B .. j*B + B - 1 modulo the ring; the sequence of ticks repeats after lcm(ring, B) frames)
    import math
    d_groups = [engine.FrameGroup([d_ring[(j * args.batch + k) % ring_n] for k in range(args.batch)])
                for j in range(math.lcm(ring_n, args.batch) // args.batch)]

    def stream_step_fn(submit_frames):
        def step(i):
            if i % args.batch == args.batch - 1:
                submit_frames(d_groups[(i // args.batch) % len(d_groups)])
        return step

    if world == 1 and not force_dist:
        if args.iters > 1:
            args.path = "plain"
        sess = engine.Session(w, h, levels, window, args.mode, device=local_rank, iters=args.iters,
                              stream_batch=args.batch if args.path == "stream" else 1, borrow_frames=args.borrow and args.path == "stream")
        sess.push_frame_host(frames[0])

        if args.path == "stream":
            # one launch per frame: pyramid(frame j) | corner(pair j-1) | fused LK(pair j-2, global shift in its loads) side by side in
            # one grid (ofx_session_stream_submit); every step completes exactly one pair once the pipeline is full
            sess.stream_begin()
            step = stream_step_fn(sess.stream_submit_frames)
            for i in range(3 * args.batch):
                step(i)
        elif args.path == "staged":
            # pair at a time, staging (frame load, pyramid, corner, shifts) on the session's aux stream under the previous
            # pair's LK launch
            def step(i):
                sess.submit_device(d_frames[(i + 1) % nframes])
        else:
            def step(i):
                sess.set_frame_device(d_frames[(i + 1) % nframes])
                sess.build_pyramid()
                sess.run_flow()
                sess.swap()

        driver = None
    else:
        from cuda_optical_flow_2_amd import parallel

        # One pair row-sharded over the ranks (strong scaling).  Default: every rank runs the one-launch-per-frame stream
        # pipeline on its row block with the corner flows computed from its own top-left patch -- no collective on the data
        # path; --shard-corner broadcast keeps rank 0's corner kernel + one RCCL broadcast per pair (staged halves).
        if args.shard_halo == "exchange":
            args.shard_corner = "broadcast"
        driver = parallel.ShardedFlow(w, h, levels, window, args.mode, rank, world, device=local_rank, corner=args.shard_corner,
                                      stream_batch=args.batch, halo_mode=args.shard_halo,
                                      borrow_frames=args.borrow and args.shard_corner == "local" and args.shard_halo != "exchange")
        sess = driver.session
        if args.shard_corner == "local":
            driver.stream_begin()
            step = stream_step_fn(driver.stream_submit_frames)
            for i in range(3 * args.batch):
                step(i)
        else:
            driver.push_frame(d_frames[0])

            def step(i):
                driver.step(d_frames[(i + 1) % nframes])

    def fence():
        torch.cuda.synchronize()
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    # Untimed clock ramp: the first ~10 ms of a kernel stream run 15-20 % slower than the sustained rate on MI355X (a
    # 200-step run right after start-up measured 175-180k Mpix/s, the same steps after 0.1 s of load 210k+), and the default
    # timed region is only tens of ms long.  So the device first works for OFX_BENCH_RAMP_S seconds on the very steps that
    # are measured afterwards; then come the W warm-up steps and the K timed steps of the contract.
    ramp_s = float(os.environ.get("OFX_BENCH_RAMP_S", "0.3"))
    t_ramp = time.perf_counter() + ramp_s
    i_ramp = 0
    while time.perf_counter() < t_ramp:
        for _ in range(64):
            step(i_ramp)
            i_ramp += 1
        torch.cuda.synchronize()
    for i in range(warmup_steps):
        step(i)
    fence()
    # pass 1 -- the throughput: EXACTLY args.steps steps, no instrumentation inside the timed region
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(warmup_steps + i)
    fence()
    dt = time.perf_counter() - t0
    # pass 2 -- the dominant kernel's duration: the same steps again with a pair of HIP events recorded around every
    # launch of that kernel on the stream it runs on (two extra packets per launch, so this pass is not the one timed above)
    sess.timing(args.steps * max(1, args.iters))
    for i in range(args.steps):
        step(warmup_steps + args.steps + i)
    fence()
    k_avg_us, k_min_us, k_n = sess.timing_read()
    sess.timing(0)

    # N > 1 only, reported under `extra`: the other way to use N GPUs on a frame stream -- every rank runs the unsharded
    # pipeline on its own pairs (no sharding, nothing shared): N times the pairs per second at unchanged latency per pair.
    # Same step count, same fences, max over ranks.
    dt_indep = None
    if driver is not None:
        # (frames per launch as at N = 1: eight only pay when a launch carries a fraction of a pair)
        s4 = engine.Session(w, h, levels, window, args.mode, device=local_rank, borrow_frames=args.borrow,
                            stream_batch=engine.suggest_stream_batch(w, h, levels, None, args.borrow))
        s4.stream_begin()
        t_ramp = time.perf_counter() + 0.1
        while time.perf_counter() < t_ramp:
            for i in range(64):
                s4.stream_submit(d_ring[i % ring_n])
            torch.cuda.synchronize()
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            s4.stream_submit(d_ring[i % ring_n])
        fence()
        dt_indep = time.perf_counter() - t0
        s4.close()
        t4 = torch.tensor([dt_indep], dtype=torch.float64, device="cuda")
        dist.all_reduce(t4, op=dist.ReduceOp.MAX)
        dt_indep = float(t4.item())

    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    if world > 1 or force_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())

    if rank == 0:
        ms = dt / args.steps * 1e3
        # the timed launch is the fused LK kernel over ALL pyramid levels (one launch, ofx_lk_levels): algorithmic
        # bytes = 10 B x the pixels of every level this rank owns
        own_px = sum((w >> k) * ((h >> k) if driver is None else (driver.plan.own[k][1] - driver.plan.own[k][0]))
                     for k in range(levels))
        lk_bytes = LK_BYTES_PER_PX * own_px
        if args.iters > 1:
            # every LK launch is timed: the first writes the flow (10 B/px), the others also read it back (18 B/px)
            lk_bytes = (10 + (args.iters - 1) * 18) * own_px // args.iters
        sharded_stream = driver is not None and args.shard_corner == "local"
        if (driver is None and args.path == "stream") or sharded_stream:
            # the stream launch also builds the next frame's pyramid: + 5 B per destination pixel of levels 1..
            # (a rank of a sharded run builds the rows it owns)
            lk_bytes += 5 * sum((w >> k) * ((h >> k) if driver is None else (driver.plan.own[k][1] - driver.plan.own[k][0]))
                                for k in range(1, levels))
        stream_like = (driver is None and args.path == "stream") or sharded_stream
        pairs_per_launch = args.batch if stream_like else 1   # a stream tick carries args.batch frames / pairs
        lk_bytes *= pairs_per_launch
        achieved = lk_bytes / (k_avg_us * 1e-6) / 1e9 if k_n else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                # per workload: {"stream_kernel": bytes, "lk_level_kernel": bytes} (tools/pmc_parse.py on separate --pmc passes)
                t = json.load(open(tpath)).get(args.workload)
                kname = "stream_kernel" if driver is None and args.path == "stream" else "lk_level_kernel"
                traffic = t.get(kname) if isinstance(t, dict) and driver is None else None  # measured for whole frames only
            except Exception:
                traffic = None
        out = {
            "metric": "Mpix/s dense LK flow",
            "value": round(w * h / (ms * 1e-3) / 1e6, 1),
            "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 5),
            "frames_per_s": round(1e3 / ms, 1),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "i32/f64",
            "data": "synthetic",
            "config": {
                "untimed_clock_ramp_s": ramp_s, "warmup_steps_run": warmup_steps,
                "workload": f"{w}x{h} pair, {levels}-level pyramid, {window}x{window} window, iters={args.iters} "
                            f"({'the only value the reference defines' if args.iters <= 1 else 'extension: bilinear-warp refinement, DESIGN.md lk_iter'}), "
                            f"mode {args.mode}: new frame's pyramid + every LK level, inputs resident in HBM",
                "frames": ("four resident device buffers, level 0 copied into the session per pair"
                           if driver is None and args.path != "stream" else f"a ring of {ring_n} distinct device buffers, " +
                           (f"read in place (ofx_params.borrow_frames: a buffer stays unmodified for {3 * args.batch} further submits)"
                            if args.borrow and ((driver is None and args.path == "stream") or (driver is not None and args.shard_corner == "local" and args.shard_halo != "exchange"))
                            else "level 0 copied into the session")),
                "sharding": "none" if driver is None else (
                    f"row blocks over {world} rank(s), halos recomputed from a wider level-0 halo; " +
                    ("every rank runs the one-launch stream pipeline on its block and forms the shift vectors from its own top-left "
                     "patch of the frame: no collective on the data path (DESIGN.md section 5)" if sharded_stream else
                     "rank 0's corner kernel + one RCCL broadcast of the shift vectors per pair (DESIGN.md section 5)"))
                    .replace("halos recomputed from a wider level-0 halo", "halo rows of every level exchanged with the neighbouring ranks"
                             if args.shard_halo == "exchange" else "halos recomputed from a wider level-0 halo"),
            },
            "roofline": {
                "bound": "hbm", "kernel": (f"stream_kernel (one launch per {pairs_per_launch} frame(s): pyramid(s) of the newest frame(s) | corner flows of the "
                            f"{pairs_per_launch} pair(s) before | fused LK of all levels of the {pairs_per_launch} pair(s) before those; bytes per pair = "
                            "10 B/px LK + 5 B/px pyramid)"
                           if (driver is None and args.path == "stream") or sharded_stream else
                           "lk_level_kernel (all pyramid levels in one launch: fused derivatives + window sums + 2x2 solve)"),
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "algorithmic_bytes_per_launch": lk_bytes, "pairs_per_launch": pairs_per_launch, "avg_launch_us": round(k_avg_us, 2), "min_launch_us": round(k_min_us, 2),
                "launches_timed": k_n, "timed_in": "second pass over the same steps with hipEventRecord around each launch on its stream",
                "traffic": traffic,
            },
        }
        if world == 1 and not force_dist and not args.no_extras and args.iters <= 1 and args.mode == "lk_float" and args.workload in BASELINE_ITERS:
            # BASELINE.json's configs carry "N iters"; the reference has no iterations, so they run as the
            # lk_iter extension here, next to the reference-defined line above (same process, same frames, plain path)
            it = BASELINE_ITERS[args.workload]
            s2 = engine.Session(w, h, levels, window, args.mode, device=local_rank, iters=it)
            s2.set_frame_device(d_frames[0]); s2.build_pyramid(); s2.swap()
            def step2(i):
                s2.set_frame_device(d_frames[(i + 1) % nframes]); s2.build_pyramid(); s2.run_flow(); s2.swap()
            n2 = max(10, min(args.steps, 50))
            for i in range(5):
                step2(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n2):
                step2(5 + i)
            torch.cuda.synchronize()
            ms2 = (time.perf_counter() - t0) / n2 * 1e3
            out["extra"] = {"baseline_config_with_iters": {
                "workload": f"{w}x{h}, {levels} levels, {window}x{window}, iters={it} (extension lk_iter: bilinear-warp refinement)",
                "value": round(w * h / (ms2 * 1e-3) / 1e6, 1), "unit": "Mpix/s", "ms_per_step": round(ms2, 5),
                "frames_per_s": round(1e3 / ms2, 1), "steps": n2}}
            s2.close()
            if args.path == "stream" and args.borrow:
                # the same stream path with the session's own copy of level 0 of every frame (ofx_params.borrow_frames = 0: the
                # caller may reuse a frame buffer as soon as the launch that took it has run), at the frames per launch that suit it
                b3 = engine.suggest_stream_batch(w, h, levels, None, False)
                s3 = engine.Session(w, h, levels, window, args.mode, device=local_rank, stream_batch=b3, borrow_frames=False)
                s3.stream_begin()
                t_ramp = time.perf_counter() + 0.1
                while time.perf_counter() < t_ramp:
                    for i in range(64):
                        s3.stream_submit(d_ring[i % ring_n])
                    torch.cuda.synchronize()
                n3 = max(8 * b3, args.steps // b3 * b3)
                t0 = time.perf_counter()
                for i in range(n3):
                    s3.stream_submit(d_ring[i % ring_n])
                torch.cuda.synchronize()
                ms3 = (time.perf_counter() - t0) / n3 * 1e3
                out["extra"]["stream_with_copied_frames"] = {
                    "workload": f"as value, but the session copies level 0 of every frame ({b3} frames per launch)",
                    "value": round(w * h / (ms3 * 1e-3) / 1e6, 1), "unit": "Mpix/s", "ms_per_step": round(ms3, 5), "steps": n3}
                s3.close()
            if args.path == "stream" and args.workload == "4k":
                # the metric names 1080p pairs next to 4K ones (BASELINE.json): the same pipeline on the 1080p configuration
                w5, h5, l5, win5 = WORKLOADS["1080p"]
                b5 = engine.suggest_stream_batch(w5, h5, l5)
                f5 = [torch.from_numpy(synth.smooth_pair(w5, h5, 2.0 * i * mx, 1.0 * i * my)[1]).cuda() for i in range(nframes)]
                r5 = (3 * b5 + 4 + 3) // 4 * 4
                f5 = [f5[i % nframes] if i < nframes else f5[i % nframes].clone() for i in range(r5)]
                s5 = engine.Session(w5, h5, l5, win5, args.mode, device=local_rank, stream_batch=b5)
                s5.stream_begin()
                t_ramp = time.perf_counter() + 0.1
                while time.perf_counter() < t_ramp:
                    for i in range(64):
                        s5.stream_submit(f5[i % r5])
                    torch.cuda.synchronize()
                n5 = max(8 * b5, args.steps // b5 * b5)
                t0 = time.perf_counter()
                for i in range(n5):
                    s5.stream_submit(f5[i % r5])
                torch.cuda.synchronize()
                ms5 = (time.perf_counter() - t0) / n5 * 1e3
                out["extra"]["workload_1080p"] = {
                    "workload": f"{w5}x{h5} pair, {l5}-level pyramid, {win5}x{win5} window, iters=1, stream path, {b5} frames per launch",
                    "value": round(w5 * h5 / (ms5 * 1e-3) / 1e6, 1), "unit": "Mpix/s", "ms_per_step": round(ms5, 5),
                    "frames_per_s": round(1e3 / ms5, 1), "steps": n5}
                s5.close()
        if dt_indep is not None:
            ms4 = dt_indep / args.steps * 1e3
            out.setdefault("extra", {})["independent_pairs_per_rank"] = {
                "workload": f"every one of the {world} rank(s) runs the unsharded stream pipeline on its own frame pairs (no sharding, no "
                            "communication): aggregate pairs/s, weak scaling, latency per pair as on one GPU",
                "value": round(world * w * h / (ms4 * 1e-3) / 1e6, 1), "unit": "Mpix/s", "ms_per_step_per_rank": round(ms4, 5),
                "steps_per_rank": args.steps}
        if driver is None and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, w, h, levels, window)
        print(json.dumps(out), flush=True)
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
