"""bench.py --gpus N outside torch.distributed.run: the driver starts the N > 1 bench exactly as it starts
--gpus 1.  bench.py must then start torch.distributed.run itself, as a CHILD process, before anything touches the GPU, and pass
the child's exit code on.  No GPU here: the ranks report how far they got (OFX_BENCH_RANK_PROBE) or fail at torch.cuda."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=300):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=timeout, cwd=ROOT)


def test_gpus_2_starts_two_ranks_as_a_child_process():
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"OFX_BENCH_RANK_PROBE": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    import re
    probes = [json.loads(m) for m in re.findall(r'\{"probe".*?\}', r.stderr)]   # (two ranks write to one pipe: lines may run together)
    assert sorted(p["rank"] for p in probes) == [0, 1], r.stderr[-2000:]
    assert all(p["world"] == 2 and p["gpus"] == 2 for p in probes)
    # the N > 1 headline: ranks are handed their own rows only, the config as written
    assert all(p["shard_halo"] == "stream_exchange" and p["iters"] == 5 for p in probes)
    assert "starting -m torch.distributed.run" in r.stderr


def test_gpus_2_without_a_gpu_fails_at_the_device_not_at_the_launch():
    import torch

    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is present: the ranks would run")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-extras", "--no-cpu-baseline"], {})
    assert r.returncode != 0
    assert "launch with torch.distributed.run" not in r.stderr and "WORLD_SIZE=1" not in r.stderr, r.stderr[-3000:]
    low = r.stderr.lower()
    assert "hip" in low or "cuda" in low or "gpu" in low, r.stderr[-3000:]


def test_default_arguments_name_the_config_as_written():
    sys.path.insert(0, ROOT)
    import bench

    old = sys.argv
    try:
        sys.argv = ["bench.py"]
        a = bench.parse_args()
        assert a.gpus == 1 and a.iters == 5 and a.workload == "4k"
        sys.argv = ["bench.py", "--workload", "8k"]
        assert bench.parse_args().iters == 10
        sys.argv = ["bench.py", "--iters", "1"]
        assert bench.parse_args().iters == 1
    finally:
        sys.argv = old
    # the as-launched accounting: 4K, five iterations -- (10 + 2) + 3 * 20 + 18 B/px, the shift's 2 B/px below the top level, 5 B/px pyramid
    px = bench.level_px(3840, 2160, 5)
    lb = bench.launch_bytes(3840, 2160, 5, 5)
    assert lb["stream"] == 12 * sum(px) + 5 * sum(px[1:]) and lb["lk_acc_warp"] == 20 * sum(px) and lb["lk_acc"] == 18 * sum(px)
    assert lb["shift"] == 2 * sum(px[:-1])
    kinds = {"stream": (400.0, 390.0, 10), "shift": (60.0, 55.0, 10), "lk_acc_warp": (450.0, 440.0, 30), "lk_acc": (330.0, 320.0, 10)}
    r = bench.pair_roofline(kinds, 80, 3840, 2160, 5, 5, pairs_per_launch=8)
    want = (lb["stream"] + lb["shift"] + 3 * lb["lk_acc_warp"] + lb["lk_acc"])
    assert r["algorithmic_bytes_per_pair"] == want
    assert r["frac"] < r["frac_r02_accounting"]   # the old accounting counted bytes of launches that no longer run
    d = bench.dominant_block(kinds, 8, 3840, 2160, 5, 5)
    assert d["kind"] == "lk_acc_warp" and d["algorithmic_bytes_per_launch"] == 8 * 20 * sum(px)


def test_frames_hint_follows_the_ring_against_the_infinity_cache(monkeypatch):
    """bench.py tells its stream sessions where their frames come from (ofx_params.deep_fetch): +1 for a ring of never-rewritten
    buffers longer than the 256 MiB Infinity Cache (each buffer comes back from HBM), -1 for one that fits; an override for A/Bs."""
    sys.path.insert(0, ROOT)
    import bench

    monkeypatch.delenv("OFX_BENCH_DEEP_FETCH", raising=False)
    px4k = 3840 * 2160
    assert bench.frames_hint(bench.cold_ring_size(8, True, px4k), px4k) == 1
    assert bench.frames_hint(bench.ring_size(8, True), px4k) == -1          # 20 x 8.3 MB = 166 MB
    assert bench.frames_hint(16, 7680 * 4320) == 1
    monkeypatch.setenv("OFX_BENCH_DEEP_FETCH", "0")
    assert bench.frames_hint(1000, px4k) == 0


def test_power_reading_is_parsed_from_rocm_smi_text():
    """bench.py's `power` field (the stream launches run at the board's power limit) comes from rocm-smi's text"""
    sys.path.insert(0, ROOT)
    import bench

    text = (
        "WARNING: AMD GPU device(s) is/are in a low-power state. Check power control/runtime_status\\n"
        "GPU[0]\\t\\t: fclk clock level: 0: (1250Mhz)\\n"
        "GPU[0]\\t\\t: mclk clock level: 0: (2000Mhz)\\n"
        "GPU[0]\\t\\t: sclk clock level: 1: (2184Mhz)\\n"
        "GPU[0]\\t\\t: socclk clock level: S: (72Mhz)\\n"
        "GPU[0]\\t\\t: Current Socket Graphics Package Power (W): 1374.0\\n"
        "GPU[0]\\t\\t: Max Graphics Package Power (W): 1400.0\\n")
    assert bench.parse_rocm_smi(text) == (1374.0, 1400.0, 2184)
    assert bench.parse_rocm_smi("no such tool") == (None, None, None)
