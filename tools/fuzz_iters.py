"""Randomised parity of the refinement iterations (lk_iter): the launches that write the next iteration's warped image themselves
(csrc/lk_body_warp.h; the default) against one warp launch per iteration (OFX_ITER_FUSED=0), pair at a time and through the
stream pipeline -- random size, levels, window, iterations, solve, frames per tick, borrowed frames, deep row fetch, row sharding
(logical ranks on one device; a result that differs must come with a non-zero status word), and frames with flat blocks and
noise (non-finite and huge flows).  tests/test_gpu_parity.py pins the two-launch form against the oracle.
    python tools/fuzz_iters.py [n_configs] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cuda_optical_flow_2_amd import engine as eng, synth
from cuda_optical_flow_2_amd.parallel import ShardPlan


def same(a, b):
    return a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32))


def run(n_cfg: int, seed: int) -> int:
    rng = np.random.default_rng(seed)
    bad = 0
    for it in range(n_cfg):
        L = int(rng.integers(1, 6))
        step = 1 << (L - 1)
        w = step * int(rng.integers(3, max(4, 1500 // step)))
        h = step * int(rng.integers(3, max(4, 900 // step)))
        win = int(rng.choice([3, 5, 7, 9, 11, 15, 19, 23]))
        iters = int(rng.integers(2, 7))
        mode = "lk_float" if rng.random() < 0.75 else "lk_float_fast"
        B = int(rng.choice([1, 2, 4, 8]))
        borrow = bool(rng.random() < 0.5)
        dma = int(rng.choice([0, 1]))
        R = int(rng.choice([1, 1, 2, 3])) if L >= 2 else 1
        if R > 1 and (h >> (L - 1)) < 2 * R:
            R = 1
        nf = int(rng.integers(3, 4 + 2 * B))
        kind = rng.choice(["smooth", "noise", "mixed"])
        pitch = (w + 63) // 64 * 64   # (borrowed frames of an iterating session need the session's pitch)
        frames = []
        for i in range(nf):
            a = synth.smooth_pair(w, h, 1.3 * i, -0.8 * i, seed=it + 3)[1] if kind != "noise" else synth.random_pair(w, h, seed=it * 50 + i)[0]
            if kind == "mixed":
                a = a.copy()
                a[h // 2:] = synth.random_pair(w, h, seed=it * 50 + i)[0][h // 2:]
                a[h // 4: h // 4 + max(8, h // 6), w // 5: w // 5 + max(8, w // 4)] = 90   # a flat block: 0 / 0 in its windows
            buf = torch.zeros((h, pitch), dtype=torch.uint8, device="cuda")
            buf[:, :w] = torch.from_numpy(a).cuda()
            frames.append(buf[:, :w])
        desc = f"{w}x{h} L{L} w{win} iters{iters} {mode} B{B} borrow={borrow} dma={dma} R{R} nf={nf} {kind}"
        if os.environ.get("OFX_FUZZ_ONLY") and it != int(os.environ["OFX_FUZZ_ONLY"]):
            continue
        os.environ["OFX_ITER_DMA"] = str(dma)
        try:
            def plain_all(fused):
                os.environ["OFX_ITER_FUSED"] = "1" if fused else "0"
                s = eng.Session(w, h, L, win, mode, iters=iters)
                s.set_frame_device(frames[0]); s.build_pyramid(); s.swap()
                out = {}
                for i in range(1, nf):
                    s.set_frame_device(frames[i]); s.build_pyramid(); s.run_flow()
                    torch.cuda.synchronize()
                    out[i] = [s.flow_host(k) for k in range(L)]
                    s.swap()
                s.close()
                return out
            want = plain_all(False)
            got_plain = plain_all(True)
            os.environ["OFX_ITER_FUSED"] = "1"
            got, seen = {}, 0
            status = 0
            if L >= 2:
                if R == 1:
                    ranks = [eng.Session(w, h, L, win, mode, iters=iters, stream_batch=B, borrow_frames=borrow)]
                else:
                    ranks = [eng.Session(w, h, L, win, mode, shard=ShardPlan(w, h, L, win, r, R, iters=iters, warp_margin=24), local_corner=True,
                                         iters=iters, stream_batch=B, borrow_frames=borrow, strict=False) for r in range(R)]
                for s in ranks:
                    s.stream_begin()

                def snap(done):
                    nonlocal seen
                    if done >= 1:
                        for p in range(max(seen + 1, done - B + 1), done + 1):
                            got[p] = [torch.cat([s.flow_of(p, k)[0] for s in ranks], dim=0).cpu().numpy() for k in range(L)]
                        seen = done
                for f in frames:
                    snap([s.stream_submit(f) for s in ranks][0])
                while True:
                    d = [s.stream_drain() for s in ranks][0]
                    if d == -2:
                        break
                    snap(d)
                torch.cuda.synchronize()
                for s in ranks:
                    status |= s.corner_status() if R > 1 else 0
                    s.close()
            nbad = 0
            for p in want:
                for k in range(L):
                    if not same(got_plain[p][k], want[p][k]):
                        nbad += 1
                    if L >= 2 and (p not in got or not same(got[p][k], want[p][k])) and status == 0:
                        nbad += 1   # (a sharded run whose status word is set has said that its halo rows did not suffice)
            nonfinite = sum(int((~np.isfinite(want[p][0])).sum()) for p in want)
            print(f"[{it}] {'ok ' if nbad == 0 else 'BAD'} {desc}  non-finite flow values at level 0: {nonfinite}" + (f"  status {status:#x}" if status else "") + (f"  mismatching (pair, level) results: {nbad}" if nbad else ""), flush=True)
            bad += 1 if nbad else 0
        except Exception as e:  # a configuration the library rejects is reported, not counted
            print(f"[{it}] skipped {desc}: {type(e).__name__}: {str(e)[:160]}", flush=True)
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    nb = run(n, seed)
    print(f"{n} configurations, {nb} failing")
    sys.exit(1 if nb else 0)
