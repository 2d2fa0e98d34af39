"""The certified screens of the forest build, restated in numpy and checked against the ORACLE's f32 margins (no GPU).

The build only uses the sign of a margin (`D::side`, src/distance/mod.rs:103-110).  The device decides it from a coarse
copy of the row and of the normal — int8 first (arroy_amd/csrc/forest.hip: k_shadow_rows8 / k_forest_shadow_normals8 /
stage 0 of k_forest_screen_node), binary16 second (screen_device.h) — whenever |screen value| exceeds a bound E on its
distance to the reference's f32 margin:

    |s - r| <= |n - n~||x~| + |n||x - x~| + gamma_s |n~||x~| + gamma_r |n||x|        (Cauchy-Schwarz, measured norms)

This test rebuilds both copies and both bounds with float32 arithmetic as the kernels do, takes the reference margin r from
the oracle (the restated AVX order), and asserts on several data distributions that a decided pair never has the wrong
sign — the property the GPU tests check under AH_SCREEN_VERIFY=1, here without a GPU — and that the stages decide the share
of the pairs DESIGN.md quotes for the benchmark data."""
import numpy as np
import pytest

from oracle import oracle as O

F = np.float32
UP = lambda pitch: F(1.0) + F(pitch + 64) * F(1.2e-7)  # noqa: E731  (the kernels' round-up factor for measured norms)


def norm_up(v, pitch):
    return np.sqrt((v.astype(F) ** 2).sum(axis=-1, dtype=F)).astype(F) * UP(pitch)


def gamma_r(dims):
    return F(4.0 * (dims // 32 + 6.0 + 62.0) * 5.9604645e-8)


def stage_int8(X, nv, dims):
    """(screen value, bound) of every row against `nv`, as stage 0 of k_forest_screen_node computes them."""
    pitch8 = (dims + 127) // 128 * 128
    maxabs = F(np.abs(X).max())
    scale, inv = F(maxabs / F(127.0)), F(F(127.0) / maxabs)
    q = np.clip(np.rint(X * inv), -127, 127).astype(np.int32)
    y = q.astype(F) * scale
    a8 = norm_up(y, pitch8).max()
    b8 = (norm_up(X - y, pitch8) + F(127.0) * scale * F(6.0e-8) * np.sqrt(F(pitch8))).max()
    xmax = norm_up(X, pitch8).max()
    mn = F(np.abs(nv).max())
    sn, invn = F(mn / F(127.0)), F(F(127.0) / mn)
    qn = np.clip(np.rint(nv * invn), -127, 127).astype(np.int32)
    yn = qn.astype(F) * sn
    an, cn = norm_up(yn, pitch8), norm_up(nv, pitch8)
    bn = norm_up(nv - yn, pitch8) + F(127.0) * sn * F(6.0e-8) * np.sqrt(F(pitch8))
    s8 = (q @ qn).astype(F) * F(scale * sn)  # exact integer dot, two scale products
    e = bn * a8 + cn * b8 + F(1.0e-6) * (an * a8) + gamma_r(dims) * (cn * xmax)
    return s8, F(e * F(1.002) + F(1e-30))


def stage_binary16(X, nv, dims):
    hpitch = (dims + 63) // 64 * 64
    def shadow(v):
        h = v.astype(np.float16)
        h[np.abs(h.astype(F)) < F(6.103515625e-05)] = 0  # binary16 subnormals become 0
        return h.astype(F)
    Xh, nh = shadow(X), shadow(nv)
    ax, bx, cx = norm_up(Xh, hpitch), norm_up(X - Xh, hpitch), norm_up(X, hpitch)
    an, bn, cn = norm_up(nh, hpitch), norm_up(nv - nh, hpitch), norm_up(nv, hpitch)
    gamma_s = F(4.0 * (2.0 * (hpitch // 16) + 8.0) * 5.9604645e-8)
    s = (Xh.astype(np.float64) @ nh.astype(np.float64)).astype(F)  # any accumulation order: covered by gamma_s
    e = bn * ax + cn * bx + gamma_s * (an * ax) + gamma_r(dims) * (cn * cx)
    return s, (e * F(1.002) + F(1e-30)).astype(F)


def cases():
    rng = np.random.default_rng(11)
    yield "uniform-768 (the benchmark rows)", rng.uniform(-1, 1, (4000, 768)).astype(F)
    yield "gaussian-768", rng.standard_normal((4000, 768)).astype(F)
    yield "gaussian-96 x 1e-3", (rng.standard_normal((4000, 96)) * 1e-3).astype(F)
    yield "uniform-200 shifted", (rng.uniform(-1, 1, (4000, 200)) + 5.0).astype(F)
    yield "mixed norms-256", (rng.standard_normal((4000, 256)) * rng.uniform(0.01, 3.0, (4000, 1))).astype(F)


@pytest.mark.parametrize("name,X", list(cases()), ids=[c[0] for c in cases()])
def test_a_decided_pair_never_has_the_wrong_sign(name, X):
    dims = X.shape[1]
    data = O.Data(O.COSINE, X)
    rng = np.random.default_rng(dims)
    decided8 = decided16 = total = 0
    for _ in range(6):
        nv, nh = data.create_split(rng.choice(len(X), 12, replace=False).astype(np.uint32))  # a real two-means normal
        nv = np.asarray(nv, dtype=F)[:dims]
        _sides, _n_left, r = data.split_sides(nv, nh)  # reference f32 margins (cosine: the dot in the AVX order)
        for stage in (stage_int8, stage_binary16):
            s, e = stage(X, nv, dims)
            dec = np.abs(s) > e
            assert np.all(np.signbit(s[dec]) == np.signbit(r[dec])), f"{name}: {stage.__name__} decided a pair wrongly"
            assert np.all(r[dec] != 0)
            if stage is stage_int8:
                decided8 += int(dec.sum())
            else:
                decided16 += int(dec.sum())
        total += len(X)
    if name.startswith("uniform-768"):
        assert decided8 > 0.70 * total, decided8 / total    # 76 % decided by the int8 stage
        assert decided16 > 0.985 * total, decided16 / total  # §2.2: ~1 % fall back to f32


def _rtz32(x64):
    """float64 -> float32 rounded toward zero (a truncating adder)."""
    y = np.float32(x64)
    if abs(float(y)) > abs(float(x64)):
        y = np.nextafter(y, np.float32(0.0))
    return y


def test_gamma_of_the_dense_product_covers_any_order_and_a_truncating_adder():
    """The dense MFMA screen (dense_device.h) sums the hpitch exact binary16 products of a pair in one f32 chain whose order
    and rounding mode the matrix unit does not document.  Its gamma_s = 2 (hpitch + hpitch / 16 + 16) 2^-23 must cover
    |sum_f32 - sum_exact| / sum |x~_i n~_i| for sequential, reversed, shuffled and 16-wide-blocked orders, with round-to-
    nearest and with truncation after every addition."""
    rng = np.random.default_rng(3)
    hpitch = 768
    gamma = 2.0 * (hpitch + hpitch / 16 + 16.0) * 1.1920929e-7
    worst = 0.0
    for _ in range(12):
        x = rng.uniform(-1, 1, hpitch).astype(np.float16).astype(np.float64)
        n = (rng.standard_normal(hpitch) / np.sqrt(hpitch)).astype(np.float16).astype(np.float64)
        p = x * n  # exact in float64 (and in f32: 11-bit x 11-bit significands)
        exact, scale = p.sum(), np.abs(p).sum()
        orders = [np.arange(hpitch), np.arange(hpitch)[::-1], rng.permutation(hpitch)]
        for order in orders:
            acc_rn, acc_tz = np.float32(0.0), np.float32(0.0)
            for v in p[order]:
                acc_rn = np.float32(np.float64(acc_rn) + v)        # round to nearest after every addition
                acc_tz = _rtz32(np.float64(acc_tz) + v)             # truncate after every addition
            blocked = np.float32(0.0)
            for b in range(0, hpitch, 16):                           # a 16-wide step summed exactly, then one rounding
                blocked = _rtz32(np.float64(blocked) + p[order][b:b + 16].sum())
            for got in (acc_rn, acc_tz, blocked):
                worst = max(worst, abs(float(got) - exact) / scale)
    assert worst < gamma / 2, (worst, gamma)  # at least a factor 2 of head-room on these inputs
