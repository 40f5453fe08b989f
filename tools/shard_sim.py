"""What one rank of an N-rank row-sharded run costs, measured on ONE GPU: a sharded local-corner session for rank r of N
running the stream pipeline alone (ranks share nothing on the data path, so this is the per-rank time of the real run).
    python tools/shard_sim.py [workload] [N ...]      (OFX_SIM_BATCH=1|2|4|8: frames per launch, default engine.suggest_stream_batch, as bench.py)"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cuda_optical_flow_2_amd import engine, synth
from cuda_optical_flow_2_amd.parallel import ShardPlan

WORK = {"4k": (3840, 2160, 5, 9), "1080p": (1920, 1080, 4, 7), "8k": (7680, 4320, 6, 15)}
name = sys.argv[1] if len(sys.argv) > 1 else "4k"
worlds = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8]
w, h, L, win = WORK[name]
frames = [torch.from_numpy(synth.smooth_pair(w, h, 2.0 * i, 1.0 * i)[1]).cuda() for i in range(4)]
BORROW = os.environ.get("OFX_SIM_BORROW", "1") == "1"
TWO = os.environ.get("OFX_SIM_TWO_STAGE", "0") == "1"  # ofx_params.stream_two_stage  # as bench.py: the frames are read in place from a ring of distinct buffers
st = torch.cuda.Stream()
torch.cuda.set_stream(st)
for N in worlds:
    res, host = [], []
    batch = int(os.environ.get("OFX_SIM_BATCH", "0")) or engine.suggest_stream_batch(w, h, L, ShardPlan(w, h, L, win, 0, N) if N > 1 else None, BORROW, TWO)
    ring = ((2 if TWO else 3) * max(batch, 4) + 4 + 3) // 4 * 4
    while len(frames) < ring:
        frames.append(frames[len(frames) % 4].clone())
    # a tick's frames go down in one call, as in bench.py (tick j takes ring buffers j*B .. j*B + B - 1 modulo the ring)
    groups = [engine.FrameGroup([frames[(j * batch + k) % ring] for k in range(batch)]) for j in range(math.lcm(ring, batch) // batch)]
    for r in sorted({0, N // 2, N - 1}):
        s = engine.Session(w, h, L, win, "lk_float", shard=ShardPlan(w, h, L, win, r, N), local_corner=True,
                           stream_batch=batch, borrow_frames=BORROW, two_stage=TWO)
        s.stream_begin()
        t_ramp = time.perf_counter() + 0.2  # untimed: the first ~10 ms after start-up run 15-20 % slow (see bench.py)
        while time.perf_counter() < t_ramp:
            for i in range(64 // batch):
                s.stream_submit_frames(groups[i % len(groups)])
            torch.cuda.synchronize()
        steps = 2000
        t0 = time.perf_counter()
        for i in range(steps // batch):
            s.stream_submit_frames(groups[i % len(groups)])
        t_host = time.perf_counter() - t0  # enqueue only: the host's share (it runs ahead of the GPU unless it is the limit)
        torch.cuda.synchronize()
        res.append((r, (time.perf_counter() - t0) / steps * 1e6))
        host.append(t_host / steps * 1e6)
        s.close()
    worst = max(t for _, t in res)
    print(f"{name} N={N}: " + "  ".join(f"rank {r}: {t:.1f} us" for r, t in res) + f"  -> {w * h / worst:.0f} Mpix/s if all ranks run like the slowest"
          f"  (host enqueue {max(host):.1f} us per frame; {batch} frames per launch, {'borrowed' if BORROW else 'copied'} frames, ring {ring}{', two stages' if TWO else ''})")
