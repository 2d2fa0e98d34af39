"""Forests at the edges of the certified screens (src/writer.rs:1201-1207 needs only the SIGN of a margin; the screens
decide it from binary16 / int8 copies when a proven bound allows):
  * rows at 7e-24, where f32 squares underflow to 0 and the measured norms behind the bound collapse (round-2 advisor) —
    whole forests against the oracle, on the GPU, for the metrics whose margin carries a bias / an extra dimension;
  * Cosine rows that contain inf / NaN with the int8 stage forced on (round-3 advisor: such a row must never be decided by
    a screen) — content digest against the f32-only build."""
import numpy as np
import pytest

import test_gpu_parity as P
from oracle import oracle as O
from test_gpu_parity import check_forest_valid, make_data

pytestmark = pytest.mark.gpu

from arroy_amd import _lib  # noqa: E402
from arroy_amd import distances as D  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _imports():
    import arroy_amd
    assert arroy_amd.device_count() >= 1, "no GPU visible: these tests must run on an MI355X"
    P.D, P.O = D, O


@pytest.mark.parametrize("scale", [7e-24, 3e-31])
@pytest.mark.parametrize("cls,dims", [(D.Euclidean, 128), (D.Manhattan, 256), (D.DotProduct, 96), (D.Cosine, 64)],
                         ids=["euclidean", "manhattan", "dot", "cosine"])
def test_tiny_magnitude_forests_equal_the_oracle(cls, dims, scale):
    n, trees = 12_000, 6
    ds, oracle, vecs, ids = make_data(cls, n, dims, seed=int(dims * 7), scale=scale)
    seeds = list(range(300, 300 + trees))
    ref = [oracle.build_tree(0, s).canonical() for s in seeds]
    for tun in (dict(), dict(AH_SCREEN8=1, AH_SCREEN_VERIFY=1), dict(AH_DENSE=1, AH_SCREEN_VERIFY=1),
                dict(AH_MARGIN_MODE=_lib.MARGIN_NODE_MAJOR, AH_SCREEN8=1, AH_SCREEN_VERIFY=1), dict(AH_SCREEN=0)):
        with _lib.tuning(**tun):
            f = ds.build_forest(seeds)
        check_forest_valid(f, n)
        assert f.stats["screen_violations"] == 0, (tun, f.stats)
        for t in range(trees):
            assert f.canonical(t) == ref[t], f"tree {t} at scale {scale} differs from the oracle under {tun}"
        f.close()
    ds.close()


@pytest.mark.parametrize("poison", ["inf", "nan", "both"])
def test_cosine_rows_with_non_finite_values_are_never_decided_by_the_int8_stage(poison):
    from arroy_amd import Dataset
    n, dims, trees = 20_000, 128, 8
    rng = np.random.default_rng(11)
    vecs = rng.standard_normal((n, dims)).astype(np.float32)
    bad = rng.choice(n, 400, replace=False)
    for i, r in enumerate(bad):
        c = rng.choice(dims, 1 + i % 3, replace=False)
        if poison == "inf" or (poison == "both" and i % 2 == 0):
            vecs[r, c] = np.float32(np.inf) * np.float32(1 if i % 4 < 2 else -1)
        else:
            vecs[r, c] = np.float32(np.nan)
    ds = Dataset(D.Cosine, dims, n)
    ds.upload_vectors(np.arange(n, dtype=np.uint32), vecs)
    ds.finalize()
    seeds = list(range(70, 70 + trees))
    with _lib.tuning(AH_SCREEN=0):
        want = ds.build_forest(seeds)
    wt, wper = want.digest()
    for tun in (dict(AH_SCREEN8=1), dict(AH_SCREEN8=1, AH_MARGIN_MODE=_lib.MARGIN_NODE_MAJOR), dict(AH_SCREEN8=0), dict(AH_DENSE=1),
                dict(AH_SCREEN8=1, AH_SCREEN8_LO=0, AH_MARGIN_MODE=_lib.MARGIN_NODE_MAJOR)):
        with _lib.tuning(**tun):
            got = ds.build_forest(seeds)
        gt, gper = got.digest()
        assert (gper == wper).all() and gt == wt, (poison, tun)
        got.close()
    want.close()
    ds.close()
