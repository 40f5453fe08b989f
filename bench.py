#!/usr/bin/env python3
"""bench.py -- Mpix/s of dense pyramidal LK flow on MI355X (BASELINE.json metric), one JSON line on rank 0.

A step = one frame pair: take the new frame's level 0 (already resident in HBM), build its pyramid, run every
pyramid level coarse->fine against the previous frame's pyramid, swap.  That is main.cu:246-272 of the reference.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 4k|1080p|8k|vga] [--mode lk_float|compat_cpu]

N > 1 (launched by torch.distributed.run, one rank per GPU): the SAME frame pair is row-sharded over the ranks
(cuda_optical_flow_2_amd/parallel.py) -- strong scaling.

The line carries, besides the contract's fields: `roofline` (dominant kernel, HIP events on its stream), `self_check`
(flows of the timed session compared with an independent plain session after the timed region; the run fails when they
differ), `extra` (further legs measured in the same process: the literal BASELINE configuration with its iterations,
cache-cold inputs, the bug-for-bug compat_cpu mode, the host-pointer gpu:: API with PCIe, the frame front end, ...) and
`cpu_baseline` (the reference's own CPU code on this host: one thread and all cores).
"""
import argparse
import json
import math
import os
import subprocess
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
OFX_TWO_STAGE_MIN_PIXELS = 30e6   # level-0 pixels per launch from which ofx_params.stream_two_stage pays (4K: five frames; measured)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
# Algorithmic bytes (each array once at its stored size)
LK_BYTES_PER_PX = 10       # fused level kernel: 2 u8 read + one (u,v) float pair written
LK_ACC_BYTES_PER_PX = 18   # refinement launch: the same + the 8-byte flow read back
WARP_BYTES_PER_PX = 10     # bilinear warp: 1 u8 + 8 flow read, 1 u8 written
PYR_BYTES_PER_DST_PX = 5   # downsample: 4 u8 read + 1 written per destination pixel


def level_px(w, h, levels, rows=None):
    """pixels of every level (rows: per-level (y0, y1) of a shard's own rows)"""
    return [(w >> k) * ((h >> k) if rows is None else (rows[k][1] - rows[k][0])) for k in range(levels)]


def pair_bytes(w, h, levels, rows=None, pyramid=True):
    px = level_px(w, h, levels, rows)
    return LK_BYTES_PER_PX * sum(px) + (PYR_BYTES_PER_DST_PX * sum(px[1:]) if pyramid else 0)


def host_cpu():
    """model name, online cores of this process (lscpu / /proc/cpuinfo)"""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        cores = sorted(os.sched_getaffinity(0))
    except AttributeError:
        cores = list(range(os.cpu_count() or 1))
    # a container may be limited to fewer CPUs than its affinity mask shows (cgroup quota): that is the share to fill
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = cores[: max(1, int(int(quota) / int(period)))]
    except (OSError, ValueError):
        pass
    return model, cores[:64]


def cpu_baseline(workload, w, h, levels, window, all_cores=True):
    """The reference's own CPU path (oracle/_ref, kind 'reference') or the oracle port: one thread (the reference is
    single-threaded), then throughput mode -- P independent pairs on P pinned processes."""
    import oracle as orc
    from cuda_optical_flow_2_amd import synth

    # bounded sample: full pipeline on a pair whose size keeps the run at ~10-20 s on one core
    sw, sh = (w, h) if w * h <= 3840 * 2160 else (3840, 2160)
    p, n = synth.smooth_pair(sw, sh)
    p3, n3 = synth.to_3ch(p), synth.to_3ch(n)
    have_ref = orc.have_reference()
    use_ref = have_ref and window == 9   # the reference's CPU window is hard-coded (OptFlowCPU.cpp:344)
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
    model, cores = host_cpu()
    what = "OptFlowCPU.cpp compiled as oracle/_ref" if use_ref else "oracle/ofx_oracle.c (compat_cpu)"
    out = {
        "value": round(sw * sh * reps / dt / 1e6, 3), "unit": "Mpix/s", "cores": 1,
        "kind": "reference" if use_ref else "port",
        # a clean checkout has no oracle/_ref (it is built from /root/reference, which does not travel): say which it was
        "ref_so": "present" if have_ref else "absent",
        "cpu_model": model, "host_cores": len(cores),
        "sample": f"{reps} pair(s) {sw}x{sh}, {levels} levels, window {window}x{window}, both pyramids + all levels, {what}, {dt:.1f} s",
    }
    if all_cores and len(cores) > 1:
        procs = []
        worker = os.path.join(ROOT, "oracle", "cpu_worker.py")
        secs = 8.0
        for c in cores:
            procs.append(subprocess.Popen([sys.executable, worker, str(c), str(sw), str(sh), str(levels), str(window), str(secs),
                                           "1" if use_ref else "0"], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True))
        for pr in procs:
            pr.stdout.readline()         # "ready": inputs generated, library loaded
        t0 = time.perf_counter()
        for pr in procs:
            pr.stdin.write("go\n")
            pr.stdin.flush()
        done = [pr.stdout.readline().split() for pr in procs]
        wall = time.perf_counter() - t0
        for pr in procs:
            pr.wait()
        pairs = sum(int(d[0]) for d in done if len(d) == 2)
        out["all_cores"] = {
            "value": round(sw * sh * pairs / wall / 1e6, 3), "unit": "Mpix/s", "cores": len(cores),
            "sample": f"{len(cores)} pinned processes, each its own {sw}x{sh} pair(s) for >= {secs:.0f} s: {pairs} pairs in {wall:.1f} s",
        }
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=104)
    ap.add_argument("--workload", default="4k", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="lk_float", choices=["lk_float", "compat_cpu", "lk_float_fast"],
                    help="lk_float: gpu::calc_opt_flow's arithmetic with the solve replayed bit for bit (default); lk_float_fast: the same with "
                         "the solve in its <= 1 ulp formulation (OFX_MODE_LK_FLOAT_FAST); compat_cpu: cpu::calc_optical_flow bug for bug")
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
    ap.add_argument("--three-stage", action="store_true",
                    help="stream path: the three-tick pipeline (pyramid | corner a tick later | LK two ticks later) instead of "
                         "ofx_params.stream_two_stage, which an unsharded stream with borrowed frames uses by default")
    ap.add_argument("--batch", type=int, default=0, choices=list(range(0, 17)),
                    help="stream path: frames per launch (ofx_params.stream_batch) = frames per step.  0 = "
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
    # two ticks instead of three between a frame and its flow (the corner blocks build their own patch pyramids): a third less
    # of everything the pipeline keeps in flight, which is what lets eight 4K frames per launch stay in the Infinity Cache
    args.two_stage = (args.borrow and not args.three_stage and args.path == "stream" and args.gpus == 1 and args.iters <= 1
                      and os.environ.get("OFX_BENCH_FORCE_DIST") != "1")
    if args.batch == 0:
        from cuda_optical_flow_2_amd.engine import suggest_stream_batch
        from cuda_optical_flow_2_amd.parallel import ShardPlan
        bw, bh, bl, bwin = WORKLOADS[args.workload]
        n_ranks = max(args.gpus, int(os.environ.get("WORLD_SIZE", "1")))
        args.batch = suggest_stream_batch(bw, bh, bl, ShardPlan(bw, bh, bl, bwin, 0, n_ranks) if n_ranks > 1 else None, args.borrow, args.two_stage)
    while args.batch > 1 and args.batch * WORKLOADS[args.workload][2] > 80:  # OFX_MAX_LK_ITEMS: (pair, level) items per launch
        args.batch -= 1
    # (in a short launch the two-stage pipeline's corner blocks -- they build their own patch pyramids -- are what
    # the launch waits for: three stages are the better plan there)
    if args.two_stage and (args.batch < 5 or args.batch * WORKLOADS[args.workload][0] * WORKLOADS[args.workload][1] < OFX_TWO_STAGE_MIN_PIXELS):
        args.two_stage = False   # (8K, two frames per launch: the patch of six levels and a 15x15 window is 544 pixels wide -- 129k vs 217k Mpix/s)
    # A STEP of the stream path is one tick = one launch = args.batch frames (one pass of the hot path over one batch of
    # input); of the pair-at-a-time paths one pair.  W and K count steps; every per-frame figure of the line says so.
    warmup_steps = args.warmup

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
    distributed = world > 1 or force_dist
    rccl_world = None
    if distributed:
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
            # proof that RCCL saw every rank: a one-element all-reduce of ones over the communicator the run uses
            ones = torch.ones(1, dtype=torch.int32, device="cuda")
            dist.all_reduce(ones)
            torch.cuda.synchronize()
            rccl_world = int(ones.item())
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

    def make_ring(src, n):
        """n DISTINCT device buffers whose contents repeat every len(src) buffers"""
        return [src[i % len(src)] if i < len(src) else src[i % len(src)].clone() for i in range(n)]

    def ring_size(batch, depth=None):
        depth = depth or (2 if args.two_stage else 3)   # ticks a borrowed frame stays in use
        return (depth * max(batch, 4) + 4 + 3) // 4 * 4

    # The stream paths take their frames from a ring of DISTINCT device buffers, as a capture / decoder surface pool would
    # hand them over: long enough for ofx_params.borrow_frames (frame f's buffer is read until the launch of submit f + 3 * batch), and
    # large enough that a frame is not still sitting in the 256 MB Infinity Cache when it comes round again (four buffers
    # would be: with borrowed frames that alone made the LK stage ~10 % faster).  Contents repeat every four buffers.
    ring_n = int(os.environ.get("OFX_BENCH_RING", "0")) or ring_size(args.batch)  # (experiments: other ring sizes)
    d_ring = make_ring(d_frames, ring_n)

    class StreamFeed:
        """Hands a session the ring's frames a tick at a time (ofx_session_stream_submit_frames: one FFI crossing per tick --
        the per-frame crossing is what limits small frames and the ranks of a sharded pair).  step(i) counts frames and
        submits on a tick's last one; tick j takes ring buffers j*B .. j*B + B - 1 modulo the ring."""

        def __init__(self, submit_frames, ring, batch):
            self.submit, self.batch, self.frames_in = submit_frames, batch, 0
            n = len(ring)
            self.groups = [engine.FrameGroup([ring[(j * batch + k) % n] for k in range(batch)]) for j in range(math.lcm(n, batch) // batch)]

        def step(self, _i=None):   # one frame
            self.frames_in += 1
            if self.frames_in % self.batch == 0:
                self.submit(self.groups[(self.frames_in // self.batch - 1) % len(self.groups)])

        def tick(self, _i=None):   # a whole tick: self.batch frames, one launch
            assert self.frames_in % self.batch == 0
            self.frames_in += self.batch
            self.submit(self.groups[(self.frames_in // self.batch - 1) % len(self.groups)])

    feed = None
    if not distributed:
        sess = engine.Session(w, h, levels, window, args.mode, device=local_rank, iters=args.iters,
                              stream_batch=args.batch if args.path == "stream" else 1,
                              borrow_frames=args.borrow and (args.path == "stream" or (args.path == "plain" and w % 64 == 0)),
                              two_stage=args.two_stage)
        sess.push_frame_host(frames[0])

        if args.path == "stream":
            # one launch per tick: pyramid(newest frames) | corner(the pairs before) | fused LK(the pairs before those, global shift
            # in its loads) side by side in one grid (ofx_session_stream_submit); every step completes exactly one pair once the
            # pipeline is full
            sess.stream_begin()
            feed = StreamFeed(sess.stream_submit_frames, d_ring, args.batch)
            step = feed.tick
            for i in range(3):   # (fill the pipeline)
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
                                      stream_batch=args.batch, halo_mode=args.shard_halo, iters=args.iters,
                                      borrow_frames=args.borrow and args.shard_corner == "local" and args.shard_halo != "exchange")
        sess = driver.session
        if args.shard_corner == "local":
            driver.stream_begin()
            feed = StreamFeed(driver.stream_submit_frames, d_ring, args.batch)
            step = feed.tick
            for i in range(3):   # (fill the pipeline)
                step(i)
        else:
            driver.push_frame(d_frames[0])

            def step(i):
                driver.step(d_frames[(i + 1) % nframes])

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if not distributed:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # Untimed clock ramp: the first ~10 ms of a kernel stream run 15-20 % slower than the sustained rate on MI355X (a
    # 200-step run right after start-up measured 175-180k Mpix/s, the same steps after 0.1 s of load 210k+), and the default
    # timed region is only tens of ms long.  So the device first works for OFX_BENCH_RAMP_S seconds on the very steps that
    # are measured afterwards; then come the W warm-up steps and the K timed steps of the contract.
    ramp_s = float(os.environ.get("OFX_BENCH_RAMP_S", "0.3"))
    t_ramp = time.perf_counter() + ramp_s
    i_ramp = 0
    fps = args.batch if feed is not None else 1   # frames (pairs) per step
    while time.perf_counter() < t_ramp:
        for _ in range(max(8, 64 // fps)):
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
    # launch on the stream it runs on (two extra packets per launch, so this pass is not the one timed above)
    # (at least ~100 launches, so that a short driver run -- K = 20 -- still gives a stable average)
    roof_steps = max(args.steps, -(-400 // args.batch) if feed is not None else 400)
    sess.timing(roof_steps * (2 * max(1, args.iters) + 3))
    for i in range(roof_steps):
        step(warmup_steps + args.steps + i)
    fence()
    kinds = {k: sess.timing_read_kind(k) for k in engine.Session.TIME_KINDS}
    k_avg_us, k_min_us, k_n = sess.timing_read()
    sess.timing(0)
    dt = max_over_ranks(dt)

    stream_like = feed is not None
    own_rows = None if driver is None else driver.plan.own

    # ---- self-check: the session that was just timed against an independent plain session ----------------------------
    def same_bits(a, b):
        return bool(((a == b) | (torch.isnan(a) & torch.isnan(b))).all().item())

    def self_check():
        """After the timed region: a short stream through the SAME session (same plan: frames per launch, borrowed ring,
        shard rows), then every level of its newest pairs against a plain pair-at-a-time session (the sequence the parity
        tests tie to the oracle).  Returns a description or raises."""
        plain = engine.Session(w, h, levels, window, args.mode, device=local_rank, iters=args.iters)

        def plain_pair(a, b):
            plain.set_frame_device(a); plain.build_pyramid(); plain.swap()
            plain.set_frame_device(b); plain.build_pyramid(); plain.run_flow()
            return [plain.flow(k)[0] for k in range(levels)]

        checked = []
        if stream_like:
            drain = sess.stream_drain if driver is None else driver.stream_drain
            while drain() != -2:
                pass
            (sess.stream_begin if driver is None else driver.stream_begin)()
            nf = 4 * args.batch
            submit = sess.stream_submit if driver is None else driver.stream_submit
            for i in range(nf):
                submit(d_ring[i % ring_n])
            while drain() != -2:
                pass
            for p in sorted({nf - 1, nf - args.batch}):
                ref = plain_pair(d_ring[(p - 1) % ring_n], d_ring[p % ring_n])
                for k in range(levels):
                    got = sess.flow_of(p, k)[0]
                    want = ref[k] if own_rows is None else ref[k][own_rows[k][0]:own_rows[k][1]]
                    if not same_bits(got, want):
                        raise SystemExit(f"bench.py self-check FAILED: stream pair {p} level {k} differs from the plain sequence")
                checked.append(p)
            what = f"stream session (batch {args.batch}) pairs {checked}, all {levels} levels == plain sequence, bit for bit"
        else:
            # pair-at-a-time paths: the flow of the last timed pair against the reference's literal level-by-level sequence
            last = warmup_steps + args.steps + roof_steps - 1
            a, b = d_frames[last % nframes], d_frames[(last + 1) % nframes]
            if driver is None:
                got = [sess.flow(k)[0] for k in range(levels)]
            else:
                got = [driver.backend.flow(k) for k in range(levels)]
            if args.iters > 1:
                ref = plain_pair(a, b)
            else:
                plain.set_frame_device(a); plain.build_pyramid(); plain.swap()
                plain.set_frame_device(b); plain.build_pyramid(); plain.run_flow_sequential()
                ref = [plain.flow(k)[0] for k in range(levels)]
            for k in range(levels):
                want = ref[k] if own_rows is None else ref[k][own_rows[k][0]:own_rows[k][1]]
                if not same_bits(got[k], want):
                    raise SystemExit(f"bench.py self-check FAILED: last pair, level {k} differs from the sequential plain path")
            what = f"last timed pair, all {levels} levels == level-by-level plain sequence, bit for bit"
        torch.cuda.synchronize()
        plain.close()
        if driver is not None:
            st = driver.corner_status()
            if st != 0:
                raise SystemExit(f"bench.py self-check FAILED: rank {rank} status word {st:#x} (a shift left the patch / the shard's halo)")
            what += "; shard status word 0"
        elif args.two_stage:
            st = sess.corner_status()
            if st != 0:
                raise SystemExit(f"bench.py self-check FAILED: status word {st:#x} (a corner shift left the patch: ofx_params.stream_two_stage)")
            what += "; status word 0 (every corner shift stayed inside its patch)"
        return what

    # (timing experiments with ablated kernels, OFX_BUILD_DEFS=-DOFX_X_*: their results are wrong by construction)
    check_msg = "SKIPPED (OFX_BENCH_SKIP_CHECK)" if os.environ.get("OFX_BENCH_SKIP_CHECK") == "1" else self_check()
    if distributed:
        dist.barrier()

    # ---- generic stream leg for the extras: wall-clock throughput + event-timed launches ---------------------------------
    def stream_leg(wl, mode, batch, borrow, ring, steps, events=True, two_stage=None):
        w2, h2, l2, win2 = wl
        two_stage = (args.two_stage and borrow) if two_stage is None else two_stage   # like the main run wherever the frames are borrowed
        assert len(ring) >= (2 if two_stage else 3) * batch + 1 or not borrow, "the ring is too short for borrowed frames"
        s2 = engine.Session(w2, h2, l2, win2, mode, device=local_rank, stream_batch=batch, borrow_frames=borrow, two_stage=two_stage)
        s2.stream_begin()
        fd = StreamFeed(s2.stream_submit_frames, ring, batch)
        t_end = time.perf_counter() + 0.15
        while time.perf_counter() < t_end:
            for _ in range(16 * batch):
                fd.step()
            torch.cuda.synchronize()
        n = max(400 // batch * batch, steps // batch * batch)   # (a short driver run must not shrink the extras to a handful of launches)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fd.step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / n * 1e3
        res = {"value": round(w2 * h2 / (ms * 1e-3) / 1e6, 1), "unit": "Mpix/s", "ms_per_step": round(ms, 5),
               "frames_per_s": round(1e3 / ms, 1), "steps": n, "frames_per_launch": batch}
        if events:
            s2.timing(n // batch)
            for _ in range(n):
                fd.step()
            torch.cuda.synchronize()
            avg, mn, cnt = s2.timing_read()
            s2.timing(0)
            nbytes = batch * pair_bytes(w2, h2, l2)
            res["roofline"] = {"bound": "hbm", "kernel": "stream_kernel", "achieved": round(nbytes / (avg * 1e-6) / 1e9, 1), "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": round(nbytes / (avg * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                               "algorithmic_bytes_per_launch": nbytes, "avg_launch_us": round(avg, 2), "launches_timed": cnt}
        s2.close()
        return res

    # N > 1 only, reported under `extra`: the other way to use N GPUs on a frame stream -- every rank runs the unsharded
    # pipeline on its own pairs (no sharding, nothing shared): N times the pairs per second at unchanged latency per pair.
    # Same step count, same fences, max over ranks.
    dt_indep = dt_exch = None
    if driver is not None:
        # (frames per launch as at N = 1: eight only pay when a launch carries a fraction of a pair)
        b4 = engine.suggest_stream_batch(w, h, levels, None, args.borrow)
        s4 = engine.Session(w, h, levels, window, args.mode, device=local_rank, borrow_frames=args.borrow, stream_batch=b4)
        s4.stream_begin()
        r4 = make_ring(d_frames, ring_size(b4))
        fd4 = StreamFeed(s4.stream_submit_frames, r4, b4)
        t_ramp = time.perf_counter() + 0.1
        while time.perf_counter() < t_ramp:
            for i in range(16 * b4):
                fd4.step()
            torch.cuda.synchronize()
        n4 = max(args.steps * fps // b4 * b4, b4)   # as many frames as the timed region held
        fence()
        t0 = time.perf_counter()
        for i in range(n4):
            fd4.step()
        fence()
        dt_indep = max_over_ranks(time.perf_counter() - t0) / n4   # seconds per frame and rank
        s4.close()
        del r4
        # north_star's literal formulation, so that a scaling run shows RCCL carrying the halos: every rank holds only its own
        # rows (+ halo) of the pair, exchanges the halo rows of every pyramid level with its neighbours (batched send/recv) and
        # receives the shift vectors by broadcast.  Pair-at-a-time, latency-bound: a short leg.
        if not args.no_extras and args.shard_halo != "exchange":
            try:
                drv2 = parallel.ShardedFlow(w, h, levels, window, args.mode, rank, world, device=local_rank, corner="broadcast",
                                            halo_mode="exchange")
                drv2.push_frame(d_frames[0])
                n5 = max(8, min(args.steps, 64))
                for i in range(4):
                    drv2.step(d_frames[(i + 1) % nframes])
                fence()
                t0 = time.perf_counter()
                for i in range(n5):
                    drv2.step(d_frames[(i + 1) % nframes])
                fence()
                dt_exch = (max_over_ranks(time.perf_counter() - t0), n5, drv2.corner_status())
                drv2.session.close()
            except ValueError as e:   # a rank owns fewer rows than the halo at some level
                dt_exch = (None, 0, str(e))

    if rank == 0:
        ms = dt / args.steps * 1e3
        # the timed launch is the fused LK kernel over ALL pyramid levels (one launch, ofx_lk_levels): algorithmic
        # bytes = 10 B x the pixels of every level this rank owns
        own_px = sum(level_px(w, h, levels, own_rows))
        lk_bytes = LK_BYTES_PER_PX * own_px
        if args.iters > 1:
            # every LK launch is timed: the first writes the flow (10 B/px), the others also read it back (18 B/px)
            lk_bytes = (LK_BYTES_PER_PX + (args.iters - 1) * LK_ACC_BYTES_PER_PX) * own_px // args.iters
        if stream_like:
            # the stream launch also builds the next frame's pyramid: + 5 B per destination pixel of levels 1..
            # (a rank of a sharded run builds the rows it owns)
            lk_bytes = pair_bytes(w, h, levels, own_rows)
        pairs_per_launch = args.batch if stream_like else 1   # a stream tick carries args.batch frames / pairs
        lk_bytes *= pairs_per_launch
        achieved = lk_bytes / (k_avg_us * 1e-6) / 1e9 if k_n else 0.0
        iters_pair = None
        if args.iters > 1:
            # refinement iterations: the roofline is that of the whole pair -- every launch of the second pass event-timed
            # and tagged (ofx_session_timing_read_kind); bytes per SURVEY 8d: 10 + (iters - 1) * (10 + 18) B/px + 5 B/px pyramid
            pair_alg = ((LK_BYTES_PER_PX + (args.iters - 1) * (WARP_BYTES_PER_PX + LK_ACC_BYTES_PER_PX)) * own_px +
                        PYR_BYTES_PER_DST_PX * sum(level_px(w, h, levels, own_rows)[1:]))
            us_pair = sum(v[0] * v[2] for v in kinds.values() if v[2]) / (roof_steps * fps)
            iters_pair = {"algorithmic_bytes_per_pair": pair_alg, "kernel_us_per_pair": round(us_pair, 2),
                          "launches": {k: {"avg_us": round(v[0], 2), "count": v[2]} for k, v in kinds.items() if v[2]}}
            lk_bytes, k_avg_us, k_min_us, k_n = pair_alg, us_pair, us_pair, roof_steps
            achieved = pair_alg / (us_pair * 1e-6) / 1e9 if us_pair else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if os.path.exists(tpath):
            try:
                # per workload: {"stream_kernel": bytes, "lk_level_kernel": bytes} (tools/pmc_parse.py on separate --pmc passes)
                t = json.load(open(tpath)).get(args.workload)
                kname = "stream_kernel" if driver is None and args.path == "stream" else "lk_level_kernel"
                if args.mode != "lk_float" or args.iters > 1:
                    kname += f"_{args.mode}_iters{args.iters}"
                traffic = t.get(kname) if isinstance(t, dict) and driver is None else None  # measured for whole frames only
                # (tools/pmc_run.py measures the default plan: two stages with the suggested frames per launch)
                if args.path == "stream" and args.iters <= 1 and not (args.two_stage and args.batch == engine.suggest_stream_batch(w, h, levels, None, True, True)):
                    traffic = None
            except Exception:
                traffic = None
        borrowed = args.borrow and ((driver is None and args.path == "stream") or
                                    (driver is not None and args.shard_corner == "local" and args.shard_halo != "exchange"))
        out = {
            "metric": "Mpix/s dense LK flow",
            "value": round(fps * w * h / (ms * 1e-3) / 1e6, 1),
            "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 5),
            "frames_per_step": fps,
            "frames_per_s": round(fps * 1e3 / ms, 1),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "i32/f64",
            "data": "synthetic",
            "self_check": "skipped" if check_msg.startswith("SKIPPED") else "ok",
            "config": {
                "untimed_clock_ramp_s": ramp_s, "warmup_steps_run": warmup_steps,
                "step": (f"one tick of the stream pipeline = one launch = {fps} frames (pairs): K = {args.steps} steps are {args.steps * fps} pairs"
                         if stream_like else "one frame pair"),
                "workload": f"{w}x{h} pair, {levels}-level pyramid, {window}x{window} window, iters={args.iters} "
                            f"({'the only value the reference defines' if args.iters <= 1 else 'extension: bilinear-warp refinement, DESIGN.md lk_iter'}), "
                            f"mode {args.mode}: new frame's pyramid + every LK level, inputs resident in HBM",
                "frames": (("four resident device buffers, " + ("read in place (ofx_params.borrow_frames)" if driver is None and args.path == "plain"
                                                                   and args.borrow and w % 64 == 0 else "level 0 copied into the session per pair"))
                           if not stream_like else f"a ring of {ring_n} distinct device buffers, " +
                           (f"read in place (ofx_params.borrow_frames: a buffer stays unmodified for {(2 if args.two_stage else 3) * args.batch} further submits)"
                            if borrowed else "level 0 copied into the session")),
                "sharding": "none" if driver is None else (
                    f"row blocks over {world} rank(s), halos recomputed from a wider level-0 halo; " +
                    ("every rank runs the one-launch stream pipeline on its block and forms the shift vectors from its own top-left "
                     "patch of the frame: no collective on the data path (DESIGN.md section 5)" if stream_like else
                     "rank 0's corner kernel + one RCCL broadcast of the shift vectors per pair (DESIGN.md section 5)"))
                    .replace("halos recomputed from a wider level-0 halo", "halo rows of every level exchanged with the neighbouring ranks"
                             if args.shard_halo == "exchange" else "halos recomputed from a wider level-0 halo"),
                "self_check": check_msg,
            },
            "roofline": {
                "bound": "hbm", "kernel": ((f"stream_kernel (one launch per {pairs_per_launch} frame(s): pyramid(s) of the newest frame(s) | corner flows of the "
                            f"{pairs_per_launch} pair(s) those frames complete, on patch pyramids the corner blocks build | fused LK of all levels of the "
                            f"{pairs_per_launch} pair(s) before; bytes per pair = 10 B/px LK + 5 B/px pyramid)" if args.two_stage and driver is None else
                            f"stream_kernel (one launch per {pairs_per_launch} frame(s): pyramid(s) of the newest frame(s) | corner flows of the "
                            f"{pairs_per_launch} pair(s) before | fused LK of all levels of the {pairs_per_launch} pair(s) before those; bytes per pair = "
                            "10 B/px LK + 5 B/px pyramid)")
                           if stream_like else
                           "lk_level_kernel (all pyramid levels in one launch: fused derivatives + window sums + 2x2 solve)"),
                "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "algorithmic_bytes_per_launch": lk_bytes, "pairs_per_launch": pairs_per_launch, "avg_launch_us": round(k_avg_us, 2), "min_launch_us": round(k_min_us, 2),
                "launches_timed": k_n, "timed_in": "second pass over the same steps with hipEventRecord around each launch on its stream",
                "traffic": traffic,
            },
        }
        if iters_pair is not None:
            out["roofline"].update(iters_pair)
            out["roofline"]["kernel"] = (f"all launches of a pair with {args.iters} iterations (stream tick / LK, shift, {args.iters - 1} x warp, "
                                         f"{args.iters - 1} x accumulating LK); avg_launch_us = kernel time per pair")
        if rccl_world is not None:
            out["rccl_world"] = rccl_world   # sum of ones over the communicator: the ranks RCCL actually connected
        if not stream_like:
            # pair-at-a-time paths: every launch of a pair, event-timed in the same second pass
            out["roofline"]["launches_per_pair_us"] = {k: round(v[0], 2) for k, v in kinds.items() if v[2]}
        extra = {}
        single = driver is None
        if single and not args.no_extras and args.iters <= 1 and args.mode == "lk_float" and args.workload in BASELINE_ITERS:
            # BASELINE.json's configs carry "N iters"; the reference has no iterations, so they run as the
            # lk_iter extension here, next to the reference-defined line above (same process, same frames, plain path), with
            # every launch of a pair event-timed: bytes per SURVEY 8d = 10 + (iters - 1) * (10 + 18) B/px + 5 B/px pyramid
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
            s2.timing(n2 * (2 * it + 3))
            for i in range(n2):
                step2(5 + n2 + i)
            torch.cuda.synchronize()
            kk = {k: s2.timing_read_kind(k) for k in engine.Session.TIME_KINDS}
            s2.timing(0)
            px_all = sum(level_px(w, h, levels))
            px_shift = sum(level_px(w, h, levels)[:-1])
            per_kind_bytes = {"lk": LK_BYTES_PER_PX * px_all, "lk_acc": LK_ACC_BYTES_PER_PX * px_all, "warp": WARP_BYTES_PER_PX * px_all,
                              "shift": 2 * px_shift, "pyramid": PYR_BYTES_PER_DST_PX * sum(level_px(w, h, levels)[1:])}
            launches = {}
            for k, (avg, mn, cnt) in kk.items():
                if cnt:
                    launches[k] = {"avg_us": round(avg, 2), "per_pair": cnt // n2}
                    if k in per_kind_bytes:
                        launches[k]["algorithmic_bytes"] = per_kind_bytes[k]
                        launches[k]["frac"] = round(per_kind_bytes[k] / (avg * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
            pair_alg = (LK_BYTES_PER_PX + (it - 1) * (WARP_BYTES_PER_PX + LK_ACC_BYTES_PER_PX)) * px_all + per_kind_bytes["pyramid"]
            kernel_us = sum(v[0] * (v[2] // n2) for v in kk.values() if v[2])
            extra["baseline_config_with_iters"] = {
                "workload": f"{w}x{h}, {levels} levels, {window}x{window}, iters={it} (extension lk_iter: bilinear-warp refinement)",
                "value": round(w * h / (ms2 * 1e-3) / 1e6, 1), "unit": "Mpix/s", "ms_per_step": round(ms2, 5),
                "frames_per_s": round(1e3 / ms2, 1), "steps": n2,
                "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "algorithmic_bytes_per_pair": pair_alg,
                             "kernel_us_per_pair": round(kernel_us, 2), "achieved": round(pair_alg / (kernel_us * 1e-6) / 1e9, 1),
                             "frac": round(pair_alg / (kernel_us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4), "launches": launches,
                             "timed_in": "second pass, hipEventRecord around every launch of the pair"}}
            s2.close()
            # the same configuration through the stream pipeline: the tick's LK stage is iteration 1 of its B pairs, every
            # further iteration one warp + one accumulating LK launch over all levels of all B pairs (taller strips, 2 + 2 *
            # (iters - 1) launches per B pairs instead of 3 + 2 * iters per pair); frames read in place like the headline's
            b9 = 4 if 4 * levels <= 80 else 2
            s9 = engine.Session(w, h, levels, window, args.mode, device=local_rank, iters=it, stream_batch=b9, borrow_frames=args.borrow)
            s9.stream_begin()
            fd9 = StreamFeed(s9.stream_submit_frames, d_ring, b9)
            for _ in range(4 * b9):
                fd9.step()
            torch.cuda.synchronize()
            n9 = max(4 * b9, min(args.steps, 48) // b9 * b9)
            t0 = time.perf_counter()
            for _ in range(n9):
                fd9.step()
            torch.cuda.synchronize()
            ms9 = (time.perf_counter() - t0) / n9 * 1e3
            s9.timing((n9 // b9) * (2 * it + 3))
            for _ in range(n9):
                fd9.step()
            torch.cuda.synchronize()
            k9 = {k: s9.timing_read_kind(k) for k in engine.Session.TIME_KINDS}
            s9.timing(0)
            s9.close()
            us9 = sum(v[0] * v[2] for v in k9.values() if v[2]) / n9            # kernel time per pair
            pair_alg9 = pair_alg                                                # same algorithmic bytes per pair
            streamed = {
                "workload": f"as above through the stream pipeline, {b9} pairs per launch, frames " + ("read in place" if args.borrow else "copied"),
                "value": round(w * h / (ms9 * 1e-3) / 1e6, 1), "unit": "Mpix/s", "ms_per_step": round(ms9, 5), "steps": n9,
                "roofline": {"bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "algorithmic_bytes_per_pair": pair_alg9,
                             "kernel_us_per_pair": round(us9, 2), "achieved": round(pair_alg9 / (us9 * 1e-6) / 1e9, 1),
                             "frac": round(pair_alg9 / (us9 * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                             "launches": {k: {"avg_us": round(v[0], 2), "per_tick": v[2] // (n9 // b9)} for k, v in k9.items() if v[2]}}}
            extra["baseline_config_with_iters"]["streamed"] = streamed
            if args.path == "stream" and args.borrow:
                # the same stream path with the session's own copy of level 0 of every frame (ofx_params.borrow_frames = 0: the
                # caller may reuse a frame buffer as soon as the launch that took it has run), at the frames per launch that suit it
                b3 = engine.suggest_stream_batch(w, h, levels, None, False)
                r3 = stream_leg((w, h, levels, window), args.mode, b3, False, d_ring[:16] if len(d_ring) > 16 else d_ring, args.steps)
                r3["workload"] = f"as value, but the session copies level 0 of every frame ({b3} frames per launch)"
                extra["stream_with_copied_frames"] = r3
                # cache-cold inputs: the ring is long enough that neither the frames nor the session's image sets survive in the
                # 256 MB Infinity Cache between their uses (ring of 32 4K buffers = 265 MB on its own), i.e. every image row the
                # level kernel reads comes from HBM -- what a pipeline fed from a large surface pool sees
                cold_n = max(32, 2 * ring_n)
                if w * h * cold_n < 40e9:
                    cold_ring = make_ring(d_frames, cold_n)
                    r6 = stream_leg((w, h, levels, window), args.mode, args.batch, True, cold_ring, args.steps)
                    r6["workload"] = (f"as value, frames read in place from a ring of {cold_n} distinct buffers ({w * h * cold_n / 1e6:.0f} MB: nothing "
                                      "survives in the Infinity Cache between uses)")
                    extra["cold_inputs"] = r6
                    del cold_ring
            if args.path == "stream":
                # the same pipeline with the solve in its <= 1 ulp(float) formulation (OFX_MODE_LK_FLOAT_FAST: SURVEY 8c's stated
                # tolerance for the solve, identical NaN / Inf positions; window sums, shift and pyramid stay bit-exact)
                r8 = stream_leg((w, h, levels, window), "lk_float_fast", args.batch, args.borrow, d_ring, args.steps)
                r8["workload"] = "as value, mode lk_float_fast (solve within 1 float ulp of the replayed reference solve instead of bit-identical)"
                extra["fast_solve"] = r8
                # the mode that IS pinned against the reference's own execution (cpu::calc_optical_flow bug for bug)
                bc = engine.suggest_stream_batch(w, h, levels, None, args.borrow, args.two_stage)
                r7 = stream_leg((w, h, levels, window), "compat_cpu", bc, args.borrow, d_ring if bc == args.batch else make_ring(d_frames, ring_size(bc)), args.steps)
                r7["workload"] = f"as value, mode compat_cpu (OptFlowCPU.cpp:312-399 bug for bug; stream path, {bc} frames per launch)"
                extra["compat_cpu"] = r7
            if args.path == "stream" and args.workload == "4k":
                # the metric names 1080p pairs next to 4K ones (BASELINE.json): the same pipeline on the 1080p configuration
                wl5 = WORKLOADS["1080p"]
                b5 = engine.suggest_stream_batch(*wl5[:3], None, True, args.two_stage)  # (the leg reads its frames in place)
                f5 = make_ring([torch.from_numpy(synth.smooth_pair(wl5[0], wl5[1], 2.0 * i * mx, 1.0 * i * my)[1]).cuda() for i in range(nframes)],
                               ring_size(b5))
                r5 = stream_leg(wl5, args.mode, b5, True, f5, args.steps)
                r5["workload"] = f"{wl5[0]}x{wl5[1]} pair, {wl5[2]}-level pyramid, {wl5[3]}x{wl5[3]} window, iters=1, stream path, {b5} frames per launch"
                extra["workload_1080p"] = r5
                del f5
            # API-compat timing: host pointers through the reference's own call surface -- gpu::gauss_pyramid for both
            # frames + gpu::calc_opt_flow per level (OptFlowGpu.cu:1909; window 19 is hard-coded there) -- PCIe included
            from cuda_optical_flow_2_amd.compat import GpuCompat
            gc = GpuCompat()
            api = {}
            for nm in ("1080p", "4k"):
                wa, ha, la, _ = WORKLOADS[nm]
                pa, na = synth.smooth_pair(wa, ha)
                p3, n3 = synth.to_3ch(pa), synth.to_3ch(na)
                gc.flow_pair(p3, n3, la)
                reps = 3
                t0 = time.perf_counter()
                for _ in range(reps):
                    gc.flow_pair(p3, n3, la)
                dta = (time.perf_counter() - t0) / reps
                api[nm] = {"value": round(wa * ha / dta / 1e6, 1), "unit": "Mpix/s", "ms_per_pair": round(dta * 1e3, 2),
                           "workload": f"{wa}x{ha}, {la} levels, window 19 (the reference's GPU constant), 3-channel host images in, "
                                       "host flow pyramid out, both pyramids rebuilt per pair as gpu::gauss_pyramid's signature demands"}
            extra["api_compat"] = api
            # the front end of main.cu's frame (main.cu:232-240, in front of the pyramid): grayscale + the 9x9 bilateral
            # pre-filter (sigma 2 / 10), device-resident, 3-channel images as the reference passes them.  Bytes per pixel
            # (SURVEY 8d): grayscale 3 read + 3 written, bilateral 3 (src) + 3 (gray) read + 3 written.  The bilateral filter
            # is bit-exact fp64 arithmetic in the reference's tap order (81 taps x ~16 double operations per pixel): it is
            # compute-bound by that definition, the HBM fraction says how far.
            from cuda_optical_flow_2_amd import lib as _l
            L_ = _l.load()
            fe = {}
            for nm in ("1080p", "4k"):
                wa, ha = WORKLOADS[nm][:2]
                img = torch.randint(0, 256, (ha, wa, 3), dtype=torch.uint8, device="cuda")
                gray, filt = torch.empty_like(img), torch.empty_like(img)
                st_ = torch.cuda.current_stream().cuda_stream

                def timed(fn, reps):
                    fn(); torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(reps):
                        fn()
                    e1.record(); torch.cuda.synchronize()
                    return e0.elapsed_time(e1) / reps * 1e3
                t_g = timed(lambda: _l.check(L_.ofx_grayscale_avg_3ch(img.data_ptr(), gray.data_ptr(), wa, ha, st_), "grayscale"), 20)
                t_b = timed(lambda: _l.check(L_.ofx_bilateral_3ch(gray.data_ptr(), gray.data_ptr(), filt.data_ptr(), wa, ha, 9, 9, 2.0, 10.0, st_), "bilateral"), 5)
                fe[nm] = {"grayscale_us": round(t_g, 1), "grayscale_frac": round(6 * wa * ha / (t_g * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                          "bilateral_9x9_us": round(t_b, 1), "bilateral_frac": round(9 * wa * ha / (t_b * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                          "value": round(wa * ha / ((t_g + t_b) * 1e-6) / 1e6, 1), "unit": "Mpix/s"}
                del img, gray, filt
            extra["frontend"] = fe
        if dt_indep is not None:
            ms4 = dt_indep * 1e3
            extra["independent_pairs_per_rank"] = {
                "workload": f"every one of the {world} rank(s) runs the unsharded stream pipeline on its own frame pairs (no sharding, no "
                            "communication): aggregate pairs/s, weak scaling, latency per pair as on one GPU",
                "value": round(world * w * h / (ms4 * 1e-3) / 1e6, 1), "unit": "Mpix/s", "ms_per_step_per_rank": round(ms4, 5),
                "frames_per_rank": n4}
        if dt_exch is not None:
            if dt_exch[0] is None:
                extra["halo_exchange"] = {"skipped": dt_exch[2]}
            else:
                ms5 = dt_exch[0] / dt_exch[1] * 1e3
                extra["halo_exchange"] = {
                    "workload": f"north_star's literal formulation over {world} rank(s): own rows only, halo rows of every pyramid level exchanged "
                                "with the neighbouring ranks (batched RCCL send/recv per level), shift vectors by RCCL broadcast; pair at a time",
                    "value": round(w * h / (ms5 * 1e-3) / 1e6, 1), "unit": "Mpix/s", "ms_per_step": round(ms5, 5), "steps": dt_exch[1],
                    "status_word": dt_exch[2]}
        if extra:
            out["extra"] = extra
        if driver is None and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, w, h, levels, window)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
