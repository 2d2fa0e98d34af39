"""The SCHEDULE paths of the forest build that only big builds take, forced on small shapes and compared with the oracle.

Round-2 verdict: the XCD-pinned row-major groups (`groups >= 16 && trees per group <= 4`), the non-temporal row stream
(> 5 MB of normals per group) and the passes cut into several launches (> 2^32 - 1 work-items) were only ever selected by
the 100-tree 10M x 768 build, whose forest was compared with nothing but itself.  Here:

* every one of those paths is forced on a cosine-768 shape small enough for the oracle to build all 64 trees
  (`ah_tuning_set`: the thresholds are tunables), `forest.canonical(t) == oracle.build_tree(...)` for every tree, and the
  counters of ah_build_stats (ABI v4) prove that the path ran;
* every kernel-selecting tunable is flipped in-process on a mid-size shape and the forest digest must not move;
* the block -> work-item maps of the row-major, dense and exact-pairs launches are run ON THE DEVICE — the same device
  functions the kernels call, the same host-side launch plans the build uses (ah_debug_launch_coverage) — and every work
  item must be served exactly once (this replaces a Python restatement of the index formulas);
* BASELINE configs[2] itself — 10M x 768 cosine, ALL 100 trees, default build: the counters say the three paths ran, the
  sides the ORACLE computes (src/writer.rs:1201-1207) for the recorded normals of sampled split nodes of four trees from
  different tree groups — the tail group (trees 96-99) included — at every depth equal the recorded children, and the
  digest equals that of the f32-only build (AH_MARGIN_EXACT_ONLY)."""
import numpy as np
import pytest

import test_gpu_parity as P
from oracle import oracle as O
from test_gpu_parity import check_forest_valid, subtree_items

pytestmark = pytest.mark.gpu

from arroy_amd import _lib  # noqa: E402
from arroy_amd import distances as D  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _imports():
    import arroy_amd
    assert arroy_amd.device_count() >= 1, "no GPU visible: these tests must run on an MI355X"
    P.D, P.O = D, O
    yield
    _lib.check(_lib.lib().ah_tuning_reset())


_cache = {}


def cosine768(n, trees, seed=77):
    """Dataset + oracle forest of an n x 768 cosine shape (built once per module: the oracle is the slow side)."""
    key = (n, trees, seed)
    if key not in _cache:
        from concurrent.futures import ThreadPoolExecutor

        from arroy_amd import Dataset
        vecs = O.synth(seed, 2, n, 768)  # ~N(0,1) rows: sign-balanced margins
        ds = Dataset(D.Cosine, 768, n)
        ds.upload_vectors(np.arange(n, dtype=np.uint32), vecs)
        ds.finalize()
        oracle = O.Data(O.COSINE, vecs)
        seeds = [int(x) for x in np.random.default_rng(seed).integers(0, 2**63, trees)]
        with ThreadPoolExecutor(8) as pool:
            ref = list(pool.map(lambda s: oracle.build_tree(0, s).canonical(), seeds))
        _cache[key] = (ds, seeds, ref)
    return _cache[key]


# ---- forced schedule variants against the oracle --------------------------------------------------------------------

FORCED = [
    # name, margin mode (trees per group), tunables, counters that must be > 0
    ("xcd_pinned_tc2", 2, dict(AH_ROWS_CHUNK_ROWS=4096), ["rows_xcd_launches"]),
    ("xcd_pinned_tc4", 4, dict(AH_ROWS_CHUNK_ROWS=4096), ["rows_xcd_launches"]),
    ("xcd_pinned_nt_rows_tc2", 2, dict(AH_ROWS_CHUNK_ROWS=4096, AH_ROWS_NT_BYTES=20_000), ["rows_xcd_launches", "rows_nt_launches"]),
    ("xcd_pinned_nt_rows_tc4", 4, dict(AH_ROWS_CHUNK_ROWS=4096, AH_ROWS_NT=1), ["rows_xcd_launches", "rows_nt_launches"]),
    ("split_launches_tc4", 4, dict(AH_ROWS_CHUNK_ROWS=2048, AH_LAUNCH_MAX_ITEMS=3_000_000, AH_ROWS_NT_BYTES=20_000),
     ["rows_xcd_launches", "rows_nt_launches", "rows_split_launches"]),
    ("split_launches_unpinned_tc2", 2, dict(AH_ROWS_CHUNK_ROWS=2048, AH_LAUNCH_MAX_ITEMS=2_500_000, AH_ROWS_XCD=0),
     ["rows_split_launches"]),
    ("spread_over_xcds_tc4", 4, dict(AH_ROWS_CHUNK_ROWS=4096, AH_ROWS_XCD=0), []),
    ("odd_chunks_tc2", 2, dict(AH_ROWS_CHUNK_ROWS=3 * 32 * 29, AH_ROWS_XCD_MIN_GROUPS=3), ["rows_xcd_launches"]),
]


@pytest.mark.parametrize("verify", [0, 1], ids=["plain", "verify_every_pair"])
@pytest.mark.parametrize("case", FORCED, ids=[c[0] for c in FORCED])
def test_forced_schedule_paths_equal_the_oracle(case, verify):
    """cosine-768, 64 trees over 48 000 rows (32 groups of 2, 16 groups of 4: enough groups for one XCD per group)."""
    _name, mode, knobs, counters = case
    n, trees = 48_000, 64
    ds, seeds, ref = cosine768(n, trees)
    with _lib.tuning(AH_SCREEN_VERIFY=verify, **knobs):
        forest = ds.build_forest(seeds, margin_mode=mode)
    st = forest.stats
    check_forest_valid(forest, n)
    for t in range(trees):
        assert forest.canonical(t) == ref[t], f"tree {t} differs from the oracle under {knobs}"
    assert st["margin_mode_launches"][_lib.MODE_LAUNCH_INDEX[mode]] > 0 and st["screened_launches"] > 0
    for c in counters:
        assert st[c] > 0, (c, st)
    if knobs.get("AH_ROWS_XCD") == 0:
        assert st["rows_xcd_launches"] == 0 and st["rows_nt_launches"] == 0
    assert st["screen_violations"] == 0
    forest.close()


def test_tail_group_and_uneven_tree_counts_with_pinned_groups():
    """67 trees in groups of 4 = 16 full groups (one XCD each) + a tail of 3 in a launch of its own; 35 in groups of 2."""
    n = 48_000
    ds, seeds, ref = cosine768(n, 64)
    extra = [int(x) for x in np.random.default_rng(9).integers(0, 2**63, 3)]
    vecs = O.synth(77, 2, n, 768)
    oracle = O.Data(O.COSINE, vecs)
    ref_extra = [oracle.build_tree(0, s).canonical() for s in extra]
    with _lib.tuning(AH_ROWS_CHUNK_ROWS=4096, AH_ROWS_NT_BYTES=20_000):
        f = ds.build_forest(seeds + extra, margin_mode=4)
        assert f.stats["rows_xcd_launches"] > 0 and f.stats["rows_nt_launches"] > 0
        for t in range(64):
            assert f.canonical(t) == ref[t], t
        for i in range(3):
            assert f.canonical(64 + i) == ref_extra[i], 64 + i
        f.close()
        g = ds.build_forest(seeds[:35], margin_mode=2)
        for t in range(35):
            assert g.canonical(t) == ref[t], t
        g.close()


# ---- every kernel-selecting switch, in-process, digest against the default's -------------------------------------------

SWITCHES = [dict(AH_ROWS_XCD=0), dict(AH_ROWS_NT=1), dict(AH_ROWS_NT=0), dict(AH_DENSE=1), dict(AH_DENSE=0), dict(AH_SCREEN8=0),
            dict(AH_SCREEN8=1), dict(AH_SCREEN=0), dict(AH_ROWMAJOR=0), dict(AH_ROWMAJOR=1), dict(AH_ROWMAJOR_LDS=0),
            dict(AH_ROWMAJOR_ADVANCE=0), dict(AH_ROWMAJOR_MAX_TC=4), dict(AH_ROWS_PER_BLOCK=64), dict(AH_ROWS_CHUNK_MB=1),
            dict(AH_FOREST_NODE_BLOCKS=300), dict(AH_FOREST_TILE_BLOCKS=100, AH_FOREST_SPLIT_BLOCKS=50, AH_FOREST_ROW_BLOCKS=64),
            dict(AH_LAUNCH_MAX_ITEMS=1_000_000, AH_ROWS_CHUNK_ROWS=1024), dict(AH_MARGIN_MODE=8), dict(AH_READBACK_DIRECT=1),
            dict(AH_NODE_PREFETCH=0), dict(AH_SCREEN8_LO=0), dict(AH_MASK_BITS=1)]


def test_every_tunable_leaves_the_forest_digest_alone():
    """200 000 x 256 Euclidean rows, 40 trees, AUTO: the default build's digest, then the same build under every
    kernel-selecting switch.  (The digest itself is checked against canonical() equality on the way.)"""
    from arroy_amd import Dataset
    n, dims, trees = 200_000, 256, 40
    ds = Dataset(D.Euclidean, dims, n)
    ds.fill_synthetic(5, 2, n)
    ds.finalize()
    seeds = [int(x) for x in np.random.default_rng(1).integers(0, 2**63, trees)]
    base = ds.build_forest(seeds)
    total, per = base.digest()
    exact = ds.build_forest(seeds, margin_mode=_lib.MARGIN_EXACT_ONLY)
    assert exact.digest()[0] == total and (exact.digest()[1] == per).all()
    assert exact.canonical(3) == base.canonical(3) and exact.canonical(39) == base.canonical(39)
    # the digest sees content, not layout: two batches of 20 trees lay the normals out differently
    batched = ds.build_forest(seeds, max_trees_in_flight=20)
    assert batched.digest()[0] == total
    # ... and it does see content: another seed for one tree changes that tree's digest only
    other = ds.build_forest(seeds[:7] + [12345] + seeds[8:])
    t2, p2 = other.digest()
    assert t2 != total and p2[7] != per[7] and (np.delete(p2, 7) == np.delete(per, 7)).all()
    for f in (exact, batched, other):
        f.close()
    for knobs in SWITCHES:
        with _lib.tuning(**knobs):
            f = ds.build_forest(seeds)
        assert f.digest()[0] == total, f"the forest changed under {knobs}"
        assert f.stats["screen_violations"] == 0
        f.close()
    base.close()
    ds.close()


def test_screen_unavailable_is_reported_not_silent():
    """AH_SCREEN=0 is a request, not a failure: the flag stays 0 and no screened launch runs."""
    ds, seeds, _ref = cosine768(48_000, 64)
    with _lib.tuning(AH_SCREEN=0):
        f = ds.build_forest(seeds[:4])
    assert f.stats["screened_launches"] == 0 and f.stats["screen_unavailable"] == 0
    f.close()


# ---- launch maps: the real device functions and host plans, exactly-once coverage ---------------------------------------

def test_row_major_launch_maps_cover_every_group_and_row_once():
    shapes = [(100_000, 768, 2, 32, {}), (100_000, 768, 4, 16, {}), (100_000, 768, 4, 25, {}), (33_333, 768, 2, 50, {}),
              (100_000, 768, 8, 12, {}), (100_000, 768, 16, 6, {}), (70_001, 128, 4, 17, dict(AH_ROWS_CHUNK_ROWS=4096)),
              (100_000, 768, 4, 25, dict(AH_ROWS_XCD=0)), (64_000, 768, 2, 40, dict(AH_LAUNCH_MAX_ITEMS=2_000_000, AH_ROWS_CHUNK_ROWS=2048)),
              (64_000, 768, 4, 19, dict(AH_LAUNCH_MAX_ITEMS=700_000, AH_ROWS_CHUNK_ROWS=1024, AH_ROWS_XCD=0)),
              (50_000, 768, 2, 33, dict(AH_ROWS_PER_BLOCK=64, AH_ROWS_CHUNK_ROWS=6400)), (31, 64, 2, 16, {})]
    for n, dims, tc, groups, knobs in shapes:
        with _lib.tuning(**knobs):
            counts = _lib.launch_coverage(0, n, dims, tc, groups)
        assert counts.shape == (groups, n)
        assert (counts == 1).all(), (n, dims, tc, groups, knobs, int(counts.min()), int(counts.max()))


def test_row_major_launch_map_of_the_headline_build():
    """The exact launch geometry of levels 9-10 of the 10M x 768 x 100-tree build: 25 groups of 4 trees on one XCD each."""
    counts = _lib.launch_coverage(0, 10_000_000, 768, 4, 25)
    assert counts.min() == 1 and counts.max() == 1


def test_dense_and_exact_pairs_launch_maps_cover_every_tile_once():
    for n, cols in [(10_000_000, 3200), (10_000_000, 100), (1_000_000, 1600), (5000, 129), (255, 1), (70_001, 12_800)]:
        counts = _lib.launch_coverage(1, n, 768, cols)
        assert (counts == 1).all(), (n, cols, int(counts.min()), int(counts.max()))
    for n, trees in [(10_000_000, 100), (1_000_000, 50), (1023, 1), (1025, 5), (70_001, 13)]:
        counts = _lib.launch_coverage(2, n, 768, trees)
        assert (counts == 1).all(), (n, trees, int(counts.min()), int(counts.max()))


# ---- BASELINE configs[2] in full -------------------------------------------------------------------------------------------

def test_baseline_config_3_all_100_trees_take_the_checked_paths():
    """10M x 768 cosine, n_trees = 100, default build (the headline `build_10m` number of bench.py)."""
    from arroy_amd import Dataset, shard
    n, dims, trees = 10_000_000, 768, 100
    ds = Dataset(D.Cosine, dims, n)
    ds.fill_synthetic(42, 1, n)
    ds.finalize()
    seeds = shard.tree_seeds(42, range(trees))
    f = ds.build_forest(seeds)
    st = f.stats
    # the schedule variant only a build with this many trees selects — groups of <= 4 trees pinned one XCD each — really
    # ran, next to the three kernel families.  (Round 2 also streamed the rows of level 10 non-temporally; with the
    # two-digit int8 stage that level is node-major now.  The non-temporal and multi-launch paths are forced on small
    # shapes above and at this size below.)
    assert st["rows_xcd_launches"] > 0, st
    assert st["dense_launches"] > 0 and st["margin_mode_launches"][0] > 0 and sum(st["margin_mode_launches"][1:5]) > 0, st
    assert st["screened_launches"] > 0 and st["screen_violations"] == 0 and st["screen_unavailable"] == 0
    assert st["screen8_pairs"] > 0 and st["screen8_decided"] > 0.7 * st["screen8_pairs"], st  # the int8 stage carries the deep levels
    total, per = f.digest()
    vecs = O.synth(42, 1, n, dims)
    oracle = O.Data(O.COSINE, vecs)
    nodes = f.nodes
    rng = np.random.default_rng(3)
    checked = {}
    picks_trees = (1, 38, 71, 98)  # groups 0 / 9 / 17 of the 4-tree passes, and the tail group (trees 96-99)
    for t in picks_trees:
        of_tree = np.flatnonzero((nodes["kind"] == 2) & (nodes["tree"] == t) & (nodes["has_normal"] == 1))
        by_depth = {}
        for i in of_tree[rng.permutation(of_tree.size)]:
            by_depth.setdefault(int(nodes[i]["depth"]), []).append(int(i))
        picks = []
        for depth, cand in sorted(by_depth.items()):
            if depth >= 2:  # (the roots and level 1 stream 30 GB per node through the oracle: the 20-tree test covers them)
                picks += cand[:2 if depth >= 6 else 1]
        for i in picks:
            nd = nodes[i]
            left, right = subtree_items(f, nd["left"]), subtree_items(f, nd["right"])
            rows = np.sort(np.concatenate([left, right]))  # ids are 0..n-1: row == id
            hdr, vec = f.normal_of(i)
            sides, n_left, _ = oracle.split_sides(vec, hdr, rows)
            assert n_left == left.size, (t, i, int(nd["depth"]))
            assert np.array_equal(rows[sides == 0], left) and np.array_equal(rows[sides == 1], right), (t, i)
            checked[int(nd["depth"])] = checked.get(int(nd["depth"]), 0) + 1
    assert len(checked) >= 11 and all(checked.get(d, 0) >= 4 for d in range(2, 12)), checked
    f.close()
    # the f32-only build of the same seeds: identical content
    g = ds.build_forest(seeds, margin_mode=_lib.MARGIN_EXACT_ONLY)
    assert g.stats["screened_launches"] == 0
    gt, gper = g.digest()
    assert (gper == per).all() and gt == total
    g.close()
    # the round-2 schedule of this build (levels 9-10 as 25 XCD-pinned groups of 4 trees, level 10 with non-temporal rows)
    # and a pass cut into several launches, at full size: same content
    with _lib.tuning(AH_SCREEN8=0, AH_LAUNCH_MAX_ITEMS=1 << 30):
        h = ds.build_forest(seeds)
    hs = h.stats
    assert hs["rows_xcd_launches"] > 0 and hs["rows_nt_launches"] > 0 and hs["rows_split_launches"] > 0, hs
    assert hs["screen8_pairs"] == 0 and h.digest()[0] == total
    h.close()
    ds.close()
