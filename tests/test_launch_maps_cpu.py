"""The block -> work-item maps of the forest kernels that are more than `blockIdx.x`, restated in Python and checked
for exactly-once coverage over many shapes (no GPU needed).  The formulas are copied from the kernels they describe:

* `k_forest_dense_screen` (arroy_amd/csrc/dense_device.h): block -> (row tile, column tile), the column tiles of a group
  of row tiles back to back on one XCD (block b runs on XCD b & 7);
* `k_forest_screen_rows` (arroy_amd/csrc/forest.hip, RowsSchedule): block -> (chunk, tree group, tile), either every
  (chunk, group) spread over all XCDs or one XCD per (chunk, group) with the groups rotated over the chunks, launches cut
  into chunk ranges (a dispatch carries < 2^32 work-items);
* `k_forest_exact_pairs`: grid-stride unit -> (row block, tree), all tree groups of a row block on one XCD.

A map that skipped or repeated a work item would leave stale side bytes behind (wrong forests) or waste a pass."""
import itertools
import random


def dense_map(n_row_tiles, n_col_tiles, group):
    r8 = (n_row_tiles + 7) // 8
    grid = 8 * ((r8 + group - 1) // group) * group * n_col_tiles
    seen = {}
    for b in range(grid):
        xcd, slot = b & 7, b >> 3
        per_group = group * n_col_tiles
        grp, within = slot // per_group, slot % per_group
        ct, rt = within // group, (grp * group + within % group) * 8 + xcd
        if rt >= n_row_tiles:
            continue
        seen[(rt, ct)] = seen.get((rt, ct), 0) + 1
    return seen


def rows_map(n_chunks, groups, tiles, xcd, threads=256):
    slots = (groups + 7) // 8 if xcd else 0
    per_chunk = 8 * slots * tiles if slots else groups * tiles
    assert per_chunk * threads < 0xFFFFFFFF
    chunks_per_launch = min(n_chunks, (0xFFFFFFFF // threads) // per_chunk)
    seen = {}
    c0 = 0
    while c0 < n_chunks:
        grid = min(chunks_per_launch, n_chunks - c0) * per_chunk
        assert grid * threads <= 0xFFFFFFFF
        for b in range(grid):
            if slots:
                x, slot = b & 7, b >> 3
                chunk = c0 + slot // (slots * tiles)
                rem = slot % (slots * tiles)
                tile = rem % tiles
                group = ((x + 8 - (chunk & 7)) & 7) + 8 * (rem // tiles)
                if group >= groups:
                    continue
            else:
                chunk = c0 + b // (groups * tiles)
                in_chunk = b % (groups * tiles)
                group, tile = in_chunk // tiles, in_chunk % tiles
            key = (chunk, group, tile)
            seen[key] = seen.get(key, 0) + 1
        c0 += chunks_per_launch
    return seen


def exact_map(n_rows, n_trees, grid):
    blocks_per_tree = (n_rows + 1023) >> 10
    tree_groups = (n_trees + 3) >> 2
    n_units = ((blocks_per_tree + 7) >> 3) * 8 * tree_groups
    seen = {}
    for b in range(grid):
        for unit in range(b, n_units, grid):
            assert unit & 7 == b & 7  # a unit keeps the XCD of the block that takes it (grid is a multiple of 8)
            slot = unit >> 3
            rb = (slot // tree_groups) * 8 + (unit & 7)
            for wave in range(4):
                t = (slot % tree_groups) * 4 + wave
                if rb >= blocks_per_tree or t >= n_trees:
                    continue
                seen[(rb, t)] = seen.get((rb, t), 0) + 1
    return seen


def test_dense_tiles_are_covered_exactly_once():
    for n_row_tiles, n_col_tiles in itertools.product([1, 7, 8, 9, 47, 64, 153], [1, 2, 5, 25]):
        seen = dense_map(n_row_tiles, n_col_tiles, 8 if n_col_tiles > 1 else 1)
        assert len(seen) == n_row_tiles * n_col_tiles and set(seen.values()) == {1}, (n_row_tiles, n_col_tiles)


def test_row_major_items_are_covered_exactly_once_in_both_orders():
    rng = random.Random(5)
    for _ in range(60):
        n_chunks, groups, tiles = rng.randint(1, 11), rng.choice([1, 2, 6, 8, 12, 16, 25, 50]), rng.choice([1, 3, 8, 16])
        for xcd in (False, True):
            seen = rows_map(n_chunks, groups, tiles, xcd)
            assert len(seen) == n_chunks * groups * tiles and set(seen.values()) == {1}, (n_chunks, groups, tiles, xcd)


def test_row_major_launches_stay_below_the_dispatch_limit():
    # 10M rows in 48 MB chunks of binary16 rows (306 chunks of 1024 tiles), 100 trees in groups of 2: 15.7 M workgroups
    # spread over all XCDs, 17.5 M with one XCD per group — more than 2^32 / 256: the level goes out in two launches
    tiles, n_chunks = 1024, 306
    for groups, slots in ((50, 0), (50, 7)):
        per_chunk = 8 * slots * tiles if slots else groups * tiles
        chunks_per_launch = min(n_chunks, (0xFFFFFFFF // 256) // per_chunk)
        assert 1 <= chunks_per_launch and chunks_per_launch * per_chunk * 256 <= 0xFFFFFFFF
        assert (chunks_per_launch < n_chunks) == (n_chunks * per_chunk * 256 > 0xFFFFFFFF)


def test_exact_pass_units_cover_every_row_block_of_every_tree_once():
    for n_rows, n_trees in [(1, 1), (1000, 3), (1024, 4), (24_000, 16), (100_003, 9), (300_000, 13)]:
        units = ((((n_rows + 1023) >> 10) + 7) >> 3) * 8 * ((n_trees + 3) >> 2)
        for grid in {8, min(units, 1 << 16), max(8, (units // 3) // 8 * 8)}:
            seen = exact_map(n_rows, n_trees, grid)
            assert len(seen) == ((n_rows + 1023) >> 10) * n_trees and set(seen.values()) == {1}, (n_rows, n_trees, grid)
