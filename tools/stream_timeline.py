"""Timeline of ONE stream-kernel launch, from per-wave start/end timestamps the kernel records (ofx_debug_stream_trace):
when do the LK waves start and end, when do the pyramid blocks run.   python tools/stream_timeline.py [batch]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cuda_optical_flow_2_amd import engine, lib, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
w, h, L, win = 3840, 2160, 5, 9
frames = [torch.from_numpy(synth.smooth_pair(w, h, 2.0 * i, 1.0 * i)[1]).cuda() for i in range(4)]
RING = int(os.environ.get("OFX_TL_RING", str(((2 if B >= 5 and os.environ.get("OFX_TL_TWO_STAGE", "1") == "1" else 3) * max(B, 4) + 4 + 3) // 4 * 4)))  # distinct source buffers, as bench.py
frames = [frames[i % 4] if i < 4 else frames[i % 4].clone() for i in range(RING)]
BORROW = os.environ.get("OFX_TL_BORROW", "1") == "1"  # as bench.py
TWO = os.environ.get("OFX_TL_TWO_STAGE", "1" if B >= 5 else "0") == "1"  # ofx_params.stream_two_stage, as bench.py from five frames per launch
st = torch.cuda.Stream(); torch.cuda.set_stream(st)
s = engine.Session(w, h, L, win, "lk_float", stream_batch=B, borrow_frames=BORROW, two_stage=TWO and BORROW)
s.stream_begin()
for i in range(10 * B):
    s.stream_submit(frames[i % RING])
torch.cuda.synchronize()
cap = 16384
buf = torch.zeros(8 * cap, dtype=torch.int64, device="cuda")
Lib = lib.load()
Lib.ofx_debug_stream_trace(buf.data_ptr(), cap, None)
for i in range(B):
    s.stream_submit(frames[(10 * B + i) % RING])
torch.cuda.synchronize()
MAXB = 16  # OFX_STREAM_MAX_BATCH: blocks [0, MAXB) are the corner blocks, 2 * MAXB pyramid stages
first = (C.c_int * (2 * MAXB + 1))()
Lib.ofx_debug_stream_trace(None, 0, first)
first = list(first)
raw = buf.cpu().numpy().reshape(-1, 2)
nb = first[-1]
raw = raw[: 4 * nb]
xcc = (raw[:, 0] >> 48) & 0xf
hw = (raw[:, 1] >> 48) & 0xffff
t = (raw & 0x0000ffffffffffff).astype(np.float64)
ok = t[:, 1] > 0
t0 = t[ok, 0].min()
us = (t - t0) / 100.0  # 100 MHz
def rng(a, b):
    m = ok[4 * a: 4 * b]
    x = us[4 * a: 4 * b][m]
    return x
lk = rng(MAXB, first[0])
print(f"blocks: corner 0..{MAXB - 1}, LK {MAXB}..{first[0]}, pyramid stages {first}")
print(f"LK waves {len(lk)}: start min/median/max {lk[:,0].min():.1f}/{np.median(lk[:,0]):.1f}/{lk[:,0].max():.1f} us, "
      f"end min/median/max {lk[:,1].min():.1f}/{np.median(lk[:,1]):.1f}/{lk[:,1].max():.1f} us, duration median {np.median(lk[:,1]-lk[:,0]):.1f}")
for i in range(2 * MAXB):
    if first[i + 1] > first[i]:
        p = rng(first[i], first[i + 1])
        print(f"pyramid stage {i}: {first[i+1]-first[i]} blocks, start min/median/max {p[:,0].min():.1f}/{np.median(p[:,0]):.1f}/{p[:,0].max():.1f}, "
              f"end max {p[:,1].max():.1f}, block duration median {np.median(p[:,1]-p[:,0]):.2f} p90 {np.percentile(p[:,1]-p[:,0],90):.2f} us")
c = rng(0, MAXB)
print(f"corner waves: {[(round(a,1), round(b,1)) for a, b in c if b > a]}")
print(f"kernel span {us[ok,1].max():.1f} us")
hist, edges = np.histogram(lk[:, 1], bins=12)
print("LK end-time histogram:", [(round(e, 0), int(n)) for e, n in zip(edges[:-1], hist)])

# where did the LK waves run?  HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
sl = slice(4 * MAXB, 4 * first[0])
m = ok[sl]
simd = (hw[sl] >> 4) & 3; cu = (hw[sl] >> 8) & 0xf; sh = (hw[sl] >> 12) & 1; se = (hw[sl] >> 13) & 7
key = (xcc[sl] * 8 + se) * 2 + sh
key = (key * 16 + cu) * 4 + simd
import collections
cnt = collections.Counter(key[m].tolist())
print("distinct SIMDs hosting LK waves:", len(cnt), " LK waves per SIMD histogram:", sorted(collections.Counter(cnt.values()).items()))
ends = us[sl][m][:, 1]
by = collections.defaultdict(list)
for k, e in zip(key[m].tolist(), ends.tolist()):
    by[k].append(e)
for nw in sorted(set(cnt.values())):
    last = [max(v) for k, v in by.items() if len(v) == nw]
    print(f"  SIMDs with {nw} LK waves: {len(last)}, last LK end median {np.median(last):.1f} max {max(last):.1f} us")

# ---- what makes a SIMD late?  (the launch ends with its slowest SIMD) --------------------------------------------------------
def simd_key(sl_):
    simd_ = (hw[sl_] >> 4) & 3; cu_ = (hw[sl_] >> 8) & 0xf; sh_ = (hw[sl_] >> 12) & 1; se_ = (hw[sl_] >> 13) & 7
    return (((xcc[sl_] * 8 + se_) * 2 + sh_) * 16 + cu_) * 4 + simd_
last_by = {k: max(v) for k, v in by.items()}
lasts = np.array(list(last_by.values()))
print("per-SIMD last LK end percentiles 5/25/50/75/95/100:", [round(float(np.percentile(lasts, q)), 1) for q in (5, 25, 50, 75, 95, 100)])
psl = slice(4 * first[0], 4 * nb)
pk = simd_key(psl)[ok[psl]]
pyr_cnt = collections.Counter(pk.tolist())
for n in sorted(set(pyr_cnt.values()) | {0}):
    sel = [last_by[k] for k in last_by if pyr_cnt.get(k, 0) == n]
    if sel:
        print(f"  SIMDs hosting {n} pyramid wave(s): {len(sel)}, last LK end median {np.median(sel):.1f} p95 {np.percentile(sel, 95):.1f} max {max(sel):.1f}")
kx = np.array(list(last_by.keys())) // (4 * 16 * 2 * 8)
for x in range(8):
    sel = lasts[kx == x]
    if len(sel):
        print(f"  XCD {x}: {len(sel)} SIMDs, last LK end median {np.median(sel):.1f} max {sel.max():.1f}")
dur = (us[sl][:, 1] - us[sl][:, 0])
nw_ = len(dur)
print("LK wave duration by wave index (16 buckets, median):", [round(float(np.median(dur[i * nw_ // 16:(i + 1) * nw_ // 16][m[i * nw_ // 16:(i + 1) * nw_ // 16]])), 1) for i in range(16)])
print("LK wave end by wave index (16 buckets, median / max):", [(round(float(np.median(us[sl][i * nw_ // 16:(i + 1) * nw_ // 16, 1])), 1), round(float(us[sl][i * nw_ // 16:(i + 1) * nw_ // 16, 1].max()), 1)) for i in range(16)])
# the late SIMDs: the wave indices they host
late = sorted(last_by, key=last_by.get)[-8:]
widx = np.arange(nw_)
allk = key
for k in late:
    ws = widx[(allk == k) & m]
    print(f"  late SIMD {k}: last end {last_by[k]:.1f}, pyramid waves {pyr_cnt.get(k, 0)}, LK waves {[(int(i), round(float(dur[i]), 1), round(float(us[sl][i, 0]), 1)) for i in ws]}")
# is block b on XCD b mod 8?  (what an XCD-aware plan would assume)
blk = np.arange(len(raw)) // 4
okm = ok
agree = float(np.mean((xcc[okm] == (blk[okm] % 8))))
print(f"waves whose XCC_ID == block index mod 8: {100 * agree:.2f} %")
