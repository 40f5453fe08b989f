import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc

    orc.build()
    return orc.Oracle()


@pytest.fixture(scope="session")
def ref():
    import oracle as orc

    if not orc.have_reference():
        pytest.skip("oracle/_ref/libref_cpu.so not built (needs /root/reference: make -C oracle ref)")
    return orc.Reference()


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))

    return load


def assert_same(a, b, what=""):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, f"{what}: shape {a.shape} vs {b.shape}"
    if a.dtype.kind == "f":
        bad = ~((a == b) | (np.isnan(a) & np.isnan(b)))
    else:
        bad = a != b
    n = int(bad.sum())
    assert n == 0, f"{what}: {n}/{a.size} elements differ, first at {np.argwhere(bad)[:3].tolist()}"


def ulp_diff_f32(a, b):
    """Distance in float32 ulps between finite values (monotone integer mapping of the bit patterns)."""
    ai = np.asarray(a, np.float32).view(np.int32).astype(np.int64)
    bi = np.asarray(b, np.float32).view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, -(ai & 0x7FFFFFFF), ai)
    bi = np.where(bi < 0, -(bi & 0x7FFFFFFF), bi)
    return np.abs(ai - bi)


def assert_flow_close(got, want, max_ulp=1, what=""):
    """Solve tolerance of SURVEY 8c: identical NaN/Inf mask, finite values within `max_ulp` float32 ulps."""
    got, want = np.asarray(got, np.float32), np.asarray(want, np.float32)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    assert np.array_equal(np.isnan(got), np.isnan(want)), f"{what}: NaN masks differ"
    inf = np.isinf(want)
    assert np.array_equal(np.isinf(got), inf) and np.array_equal(got[inf], want[inf]), f"{what}: Inf masks differ"
    fin = np.isfinite(want)
    d = ulp_diff_f32(got[fin], want[fin])
    assert d.size == 0 or d.max() <= max_ulp, f"{what}: max ulp distance {d.max()} > {max_ulp} at {int(d.argmax())}"
