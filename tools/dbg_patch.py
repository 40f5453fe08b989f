import sys
sys.path.insert(0, ".")
import numpy as np, torch
from cuda_optical_flow_2_amd import engine as eng
w, h, L, win = 640, 480, 3, 3
yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
base_img = xx / 2 + yy / 3 + xx * yy / 64
for d in (5, 10, 20):
    a = np.clip(np.floor(base_img), 0, 255).astype(np.uint8)
    b = np.clip(np.floor(base_img + d), 0, 255).astype(np.uint8)
    frames = [torch.from_numpy(x).cuda() for x in (a, b, a, b)]
    s = eng.Session(w, h, L, win, "lk_float", local_corner=True, patch_size=12)
    s.stream_begin()
    for f in frames:
        s.stream_submit(f)
    torch.cuda.synchronize()
    base = s.uv(0).data_ptr()
    both = eng.DeviceView(base, (2 * 12 * 2,), "<f4").tensor().cpu().tolist()
    print(d, "uv slots", [round(x, 2) for x in both[:4]], [round(x, 2) for x in both[24:28]])
    while s.stream_drain() != -2:
        pass
    torch.cuda.synchronize()
    print("status", s.corner_status())
    s.close()
