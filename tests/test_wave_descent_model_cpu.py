"""CPU model of k_descend_wave (arroy_amd/csrc/search.hip) against the sequential best-first loop of
`Reader::nns_by_leaf` (src/reader.rs:338-374), on random forests whose margins are drawn so that equal keys are frequent.

The kernel's claim: the candidates of a query are the leaves in decreasing key order until `search_k` ids are held, so the 8
octets of a wave may search the trees t = octet (mod 8) independently, record the leaves they pop, call a leaf settled once
its key is strictly above the top of every queue, and take the prefix of the settled leaves sorted by (key, octet, the
octet's own pop order) -- unless the leaf that reaches `search_k` shares its key with a settled leaf of another octet,
in which case the query is left to the sequential queue.  This file restates both sides in Python (the GPU tests check
the kernel itself against the oracle) and checks: whenever the model answers, it answers the sequential loop's set."""
import heapq

import numpy as np
import pytest


class Forest:
    """Random binary trees: node = ("split", left, right, margin) | ("leaf", n_ids).  Node ids are handed out like the
    library does (children before parents inside a tree, trees one after the other)."""

    def __init__(self, rng, n_trees, depth, tie_level):
        self.nodes, self.roots = [], []
        for _ in range(n_trees):
            self.roots.append(self._grow(rng, depth, tie_level))

    def _grow(self, rng, depth, tie_level):
        if depth == 0 or rng.random() < 0.15:
            self.nodes.append(("leaf", int(rng.integers(1, 40))))
            return len(self.nodes) - 1
        left = self._grow(rng, depth - 1, tie_level)
        right = self._grow(rng, depth - 1, tie_level)
        # margins: floats, small integers (binary-quantized metrics), exact zeros (`normal: None`), both signs
        kind = rng.random()
        if kind < tie_level:
            margin = float(rng.integers(-3, 4))
        elif kind < tie_level + 0.05:
            margin = 0.0
        else:
            margin = float(np.float32(rng.standard_normal() * 2))
        self.nodes.append(("split", left, right, margin))
        return len(self.nodes) - 1


def children(forest, node, key):
    """`D::pq_distance` (src/distance/mod.rs:63-68): the keys of the two children of a popped split node."""
    _, left, right, margin = forest.nodes[node]
    return (min(-margin, key), left), (min(margin, key), right)


def sequential(forest, search_k):
    """BinaryHeap<(OrderedFloat<f32>, NodeId)>: pop the greatest (key, node id) until search_k ids are held."""
    heap = [(-float("inf"), -r) for r in forest.roots]  # max-heap on (key, node) through negation
    heapq.heapify(heap)
    taken, held = [], 0
    while heap and held < search_k:
        nk, nn = heapq.heappop(heap)
        key, node = -nk, -nn
        if forest.nodes[node][0] == "leaf":
            taken.append(node)
            held += forest.nodes[node][1]
        else:
            for ck, child in children(forest, node, key):
                heapq.heappush(heap, (-ck, -child))
    return taken


def wave_model(forest, search_k, heap_cap=256, leaf_cap=64):
    """The kernel's rules; returns the list of taken leaves, or None where the kernel leaves the query to k_descend."""
    heaps = [[] for _ in range(8)]
    leaves = [[] for _ in range(8)]  # (key, node, ids) in the octet's pop order
    for t, r in enumerate(forest.roots):
        heapq.heappush(heaps[t % 8], (-float("inf"), -r))
    while True:
        for o in range(8):  # one pop per octet and step
            if not heaps[o]:
                continue
            nk, nn = heapq.heappop(heaps[o])
            key, node = -nk, -nn
            if forest.nodes[node][0] == "leaf":
                if forest.nodes[node][1] == 0:
                    continue  # a leaf that adds no id (everything filtered out) is not recorded
                if len(leaves[o]) == leaf_cap:
                    return None
                leaves[o].append((key, node, forest.nodes[node][1]))
            else:
                if len(heaps[o]) + 2 > heap_cap:
                    return None
                for ck, child in children(forest, node, key):
                    heapq.heappush(heaps[o], (-ck, -child))
        tops = [-h[0][0] for h in heaps if h]
        any_queued = bool(tops)
        top = max(tops) if tops else None
        settled = [(key, o, i, node, ids) for o in range(8) for i, (key, node, ids) in enumerate(leaves[o])
                   if not any_queued or key > top]
        if not any_queued or sum(x[4] for x in settled) >= search_k:
            break
    # key descending; equal keys: octet descending (arbitrary but fixed), then the octet's own pop order
    settled.sort(key=lambda x: (-x[0], -x[1], x[2]))
    taken, held = [], 0
    for e, (key, o, _i, node, ids) in enumerate(settled):
        if held >= search_k:
            break
        taken.append(node)
        held += ids
        if held >= search_k and any(k2 == key and o2 != o for (k2, o2, *_rest) in settled):
            return None  # the cut runs through equal keys of several octets
    return taken


@pytest.mark.parametrize("tie_level", [0.0, 0.3, 0.9])
def test_wave_model_takes_the_sequential_candidates(tie_level):
    rng = np.random.default_rng(int(tie_level * 10) + 1)
    answered = left_to_sequential = 0
    for _ in range(1500):
        n_trees = int(rng.choice([1, 2, 3, 8, 9, 17, 20]))
        forest = Forest(rng, n_trees, int(rng.integers(1, 7)), tie_level)
        total = sum(n[1] for n in forest.nodes if n[0] == "leaf")
        search_k = int(rng.choice([1, 5, 30, 90, 300, total, 10 * total]))
        want = sequential(forest, search_k)
        got = wave_model(forest, search_k)
        if got is None:
            left_to_sequential += 1
            continue
        answered += 1
        assert sorted(got) == sorted(want), (n_trees, search_k)
        assert sum(forest.nodes[n][1] for n in got) == sum(forest.nodes[n][1] for n in want)
    # given up: a tree that is a single leaf (key +inf, like every root) next to another one, exact-zero margins in
    # several trees, integer margins -- the more of those, the more often; never a wrong answer
    assert answered > left_to_sequential or tie_level > 0.5, (answered, left_to_sequential)


def test_equal_keys_inside_one_octet_follow_its_pop_order():
    """Trees 0 and 8 of a 9-tree forest share octet 0.  With integer margins their leaves often carry one key; the order
    between them is by node id and by which parent was popped first, and the octet's own queue reproduces it: with the other
    seven trees out of the way (single leaves, taken first with key +inf or not at all) no query is given up for it."""
    rng = np.random.default_rng(5)
    checked = 0
    for _ in range(300):
        forest = Forest(rng, 2, 5, 0.9)
        a, b = forest.roots
        if forest.nodes[a][0] == "leaf" or forest.nodes[b][0] == "leaf":
            continue
        forest.roots = [a] + [len(forest.nodes) + i for i in range(7)] + [b]
        forest.nodes += [("leaf", 0)] * 7  # seven empty trees: they hold no ids and tie with nothing that matters
        search_k = int(rng.choice([5, 30, 90]))
        got = wave_model(forest, search_k)
        want = sequential(forest, search_k)
        assert got is not None
        assert sorted(got) == sorted(n for n in want if forest.nodes[n][1])
        checked += 1
    assert checked > 100


def test_the_cut_inside_equal_keys_of_two_octets_is_given_up():
    """The case of tests/golden/search_equal_keys_bq.npz in miniature: leaves of 83 and 227 ids under one key 5.0 in
    trees of different octets, search_k = 90.  The sequential queue pops the bigger node id first."""
    f = Forest(np.random.default_rng(0), 0, 0, 0.0)
    f.nodes = [("leaf", 83), ("leaf", 7), ("split", 1, 0, 5.0),      # tree 0: right child (node 0) gets key 5
               ("leaf", 227), ("leaf", 9), ("split", 4, 3, 5.0)]     # tree 1: right child (node 3) gets key 5
    f.roots = [2, 5]
    assert sequential(f, 90) == [3]            # node 3 > node 0: the 227-id leaf alone
    assert wave_model(f, 90) is None           # trees 0 and 1 are different octets: left to the sequential queue
    assert sorted(wave_model(f, 400)) == sorted(sequential(f, 400))  # everything is taken: no cut inside the group


def multi_model(forest, search_k, n_blocks, rng, leaf_cap=32):
    """k_descend_multi (round 6): the trees dealt over `n_blocks` blocks of eight queues (t = block + n_blocks * (octet + 8 i)),
    every block popping at its own pace and learning of the other blocks' leaves LATE (a random delay per leaf).  A block
    stops when its queues are empty, or when the leaves it knows of — its own and the reported ones, settled or not — hold
    search_k ids at some key x and everything it still has queued lies strictly below x.  The last block orders ALL popped
    leaves (key, list, pop order) and takes the prefix; a cut through equal keys of two lists is left to the sequential
    queue.  Returns the taken leaves or None."""
    heaps = [[[] for _ in range(8)] for _ in range(n_blocks)]
    lists = [[[] for _ in range(8)] for _ in range(n_blocks)]   # (key, node, ids) in the queue's pop order
    known = [[] for _ in range(n_blocks)]                       # (key, ids) a block knows of
    in_flight = []                                              # (arrival step, destination block, key, ids)
    for t, r in enumerate(forest.roots):
        g, slot = t % n_blocks, t // n_blocks
        heapq.heappush(heaps[g][slot % 8], (-float("inf"), -r))
    running = [True] * n_blocks
    step = 0
    while any(running):
        step += 1
        for g in range(n_blocks):
            if not running[g] or rng.random() < 0.3:  # a block that is slow this step
                continue
            for o in range(8):
                if not heaps[g][o]:
                    continue
                nk, nn = heapq.heappop(heaps[g][o])
                key, node = -nk, -nn
                if forest.nodes[node][0] == "leaf":
                    ids = forest.nodes[node][1]
                    if ids == 0:
                        continue
                    if len(lists[g][o]) == leaf_cap:
                        return None
                    lists[g][o].append((key, node, ids))
                    known[g].append((key, ids))
                    for other in range(n_blocks):
                        if other != g:
                            in_flight.append((step + int(rng.integers(0, 6)), other, key, ids))
                else:
                    for ck, child in children(forest, node, key):
                        heapq.heappush(heaps[g][o], (-ck, -child))
        late = [x for x in in_flight if x[0] > step]
        for item in in_flight:
            if item[0] <= step:
                known[item[1]].append((item[2], item[3]))
        in_flight = late
        for g in range(n_blocks):
            if not running[g]:
                continue
            tops = [-h[0][0] for h in heaps[g] if h]
            if not tops:
                running[g] = False
                continue
            # x = the largest key at which the known leaves hold search_k ids
            x, held = None, 0
            for key, ids in sorted(known[g], reverse=True):
                held += ids
                if held >= search_k:
                    x = key  # (every leaf with this key or a greater one is counted by the time the sum is reached, or later: >=)
                    break
            if x is not None and max(tops) < x:
                running[g] = False
    merged = [(key, g * 8 + o, i, node, ids) for g in range(n_blocks) for o in range(8)
              for i, (key, node, ids) in enumerate(lists[g][o])]
    merged.sort(key=lambda e: (-e[0], -e[1], e[2]))
    taken, held = [], 0
    for key, lst, _i, node, ids in merged:
        if held >= search_k:
            break
        taken.append(node)
        held += ids
        if held >= search_k and any(k2 == key and l2 != lst for (k2, l2, *_r) in merged):
            return None
    return taken


@pytest.mark.parametrize("tie_level", [0.0, 0.3, 0.9])
def test_multi_block_model_takes_the_sequential_candidates(tie_level):
    rng = np.random.default_rng(int(tie_level * 10) + 77)
    answered = given_up = 0
    for _ in range(600):
        n_trees = int(rng.choice([9, 12, 17, 20, 33, 100]))
        n_blocks = min(16, (n_trees + 7) // 8)
        forest = Forest(rng, n_trees, int(rng.integers(1, 6)), tie_level)
        total = sum(n[1] for n in forest.nodes if n[0] == "leaf")
        search_k = int(rng.choice([1, 5, 30, 90, 300, total, 10 * total]))
        want = sequential(forest, search_k)
        got = multi_model(forest, search_k, n_blocks, rng)
        if got is None:
            given_up += 1
            continue
        answered += 1
        assert sorted(got) == sorted(n for n in want if forest.nodes[n][1]), (n_trees, n_blocks, search_k)
    assert answered > given_up or tie_level > 0.5, (answered, given_up)
