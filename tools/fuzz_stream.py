"""Randomised parity: the stream pipeline (random size, levels, window, mode, frames per tick, borrowed frames, row
sharding with local corner flows, padded frame buffers) against the plain pair-at-a-time sequence of the same library,
which tests/test_gpu_parity.py pins against the oracle.   python tools/fuzz_stream.py [n_configs] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cuda_optical_flow_2_amd import engine as eng, synth
from cuda_optical_flow_2_amd.parallel import ShardPlan

def run(n_cfg: int, seed: int, verbose: bool = True) -> int:
  """Returns the number of failing configurations."""
  rng = np.random.default_rng(seed)
  bad = 0
  for it in range(n_cfg):
      L = int(rng.integers(2, 7))
      step = 1 << (L - 1)
      w = step * int(rng.integers(2, max(3, 1400 // step)))
      h = step * int(rng.integers(2, max(3, 900 // step)))
      mode = "lk_float" if rng.random() < 0.7 else "compat_cpu"
      win = int(rng.choice([3, 5, 7, 9, 11, 15, 19, 23]))
      B = int(rng.choice([1, 2, 4, 8, 16]))
      if B * L > 80:
          B = 2
      borrow = bool(rng.random() < 0.5)
      two_stage = borrow and bool(rng.random() < 0.5)   # ofx_params.stream_two_stage (needs borrowed frames)
      R = int(rng.choice([1, 1, 2, 3, 4]))
      if (h >> (L - 1)) < R:
          R = 1
      nf = int(rng.integers(3, 12 if B < 8 else (30 if B < 16 else 56)))
      pitch = (w + 3) // 4 * 4 + 4 * int(rng.integers(0, 3))
      def padded(a):
          buf = torch.full((h, pitch), 0x77, dtype=torch.uint8, device="cuda")
          buf[:, :w] = torch.from_numpy(a).cuda()
          return buf[:, :w]
      gen = synth.smooth_pair if rng.random() < 0.8 else None
      if os.environ.get("OFX_FUZZ_ONLY") and it != int(os.environ["OFX_FUZZ_ONLY"]):  # replay one configuration of a seed
          continue
      frames = []
      for i in range(nf):
          a = synth.smooth_pair(w, h, 1.1 * i, -0.7 * i, seed=it + 5)[1] if gen else synth.random_pair(w, h, seed=it * 100 + i)[0]
          frames.append(padded(a))
      desc = f"{w}x{h} L{L} w{win} {mode} B{B} borrow={borrow} two_stage={two_stage} R{R} nf={nf} pitch={pitch}"
      try:
          plain = eng.Session(w, h, L, win, mode)
          plain.set_frame_device(frames[0]); plain.build_pyramid(); plain.swap()
          want = {}
          for i in range(1, nf):
              plain.set_frame_device(frames[i]); plain.build_pyramid(); plain.run_flow()
              torch.cuda.synchronize()
              want[i] = [plain.flow_host(k) for k in range(L)]
              plain.swap()
          plain.close()
          if R == 1:
              ranks = [eng.Session(w, h, L, win, mode, stream_batch=B, borrow_frames=borrow, two_stage=two_stage)]
          else:
              ranks = [eng.Session(w, h, L, win, mode, shard=ShardPlan(w, h, L, win, r, R), local_corner=True, stream_batch=B, strict=False,
                                   borrow_frames=borrow, two_stage=two_stage) for r in range(R)]
          got, seen = {}, 0
          for s in ranks:
              s.stream_begin()
          def snap(done):
              nonlocal seen
              if done >= 1:
                  for p in range(max(seen + 1, done - B + 1), done + 1):
                      got[p] = [torch.cat([s.flow_of(p, k)[0] for s in ranks], dim=0).cpu().numpy() for k in range(L)]
                  seen = done
          for i in range(nf):
              snap([s.stream_submit(frames[i]) for s in ranks][0])
          while True:
              d = [s.stream_drain() for s in ranks][0]
              if d == -2:
                  break
              snap(d)
          torch.cuda.synchronize()
          status = [s.corner_status() for s in ranks] if R > 1 else [0]
          ok = sorted(got) == list(range(1, nf))
          nbad = 0
          for p in got:
              for k in range(L):
                  a, b = got[p][k], want[p][k]
                  same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
                  nbad += int((~same).sum())
                  if not same.all() and os.environ.get("OFX_FUZZ_ONLY"):
                      rows = np.where(~same.all(axis=(1, 2)))[0]
                      cols = np.where(~same.all(axis=(0, 2)))[0]
                      print(f"   pair {p} L{k}: {int((~same).sum())} values differ, rows {rows.min()}..{rows.max()} ({len(rows)}), cols {cols.min()}..{cols.max()} ({len(cols)})")
          for s in ranks:
              s.close()
          if any(st >> 8 for st in status):
              # a vertical shift larger than the shard's halo slack: outside the sharding contract, and reported as such
              print("skip", desc, "-> shift beyond the shard margin, status", [hex(st) for st in status], "mismatches", nbad)
          elif not ok or nbad or any(status):
              bad += 1
              print("FAIL", desc, "pairs", sorted(got), "mismatches", nbad, "status", status)
          else:
              print("ok  ", desc) if verbose else None
      except Exception as e:  # configuration rejected by the library (e.g. shard too small for the plan)
          print("skip", desc, "->", str(e)[:100])
  return bad


if __name__ == "__main__":
    failures = run(int(sys.argv[1]) if len(sys.argv) > 1 else 50, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("failures:", failures)
    sys.exit(1 if failures else 0)
