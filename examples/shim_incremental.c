/* shim_incremental.c — the INCREMENTAL half of the arroy-side shim (integration/arroy-hip: `route_into_current_trees`,
 * `build_large_descendants`), played through the C ABI in the patch's call order on the reference's own database file:
 *
 *   0. map tests/golden/large_v0_6.mdb (100 Euclidean items x 30 dims), collect the record pointers (lmdb_walk.h);
 *   1. "the index before the update": items 0..89 staged, a forest of TREES trees built (split_after = 8) — its nodes play the
 *      tree nodes `Writer::build` finds in LMDB (`ImmutableTrees`), node id = index in that forest;
 *   2. `Writer::build` after `add_item` x 10 (ids 90..99), with the `hip` feature:
 *      a. insert_items_in_current_trees (src/writer.rs:846-889) -> hip::route_into_current_trees: ONLY the new items are
 *         staged, the split planes of the existing trees are mirrored WITHOUT their item lists (routing never reads them),
 *         one ah_route_items call sends every new item down every tree (`D::side`, src/writer.rs:1424-1431); the Descendants
 *         nodes that received items become `stored | new` (:1411-1414);
 *      b. insert_descendants_in_file_and_spawn_tasks (:744-844) -> hip::build_large_descendants: the touched nodes that no
 *         longer fit (`fit_in_descendant`, :474-477) are collected, ONLY their members are staged, ONE ah_build_subtrees call
 *         builds a sub-tree per node (`make_tree_in_file`, :1167-1261), and the nodes are written children first — the root of
 *         a sub-tree under the id of the descendant it replaces (`next_id: Some(descendant_id)`, :693-702), every other node
 *         under the next free id (`ConcurrentNodeIds::next`);
 *   3. the updated forest is checked here (every tree holds each of the 100 items exactly once, no node id is reused) and
 *      written to argv[2] for tests/test_gpu_staging.py, which compares it node for node with the CPU oracle replaying the
 *      same steps (ao_route_items + ao_build_tree_on).
 *
 * Seeds: tree t of step 1 = 42 + t; the routing coin of `normal: None` nodes is keyed `seed + root` like the reference's
 * `R::seed_from_u64(seed.wrapping_add(root))` (:1133) — no such node exists on this data; the sub-tree that replaces a
 * descendant of tree t whose smallest item is m = 1000 + 1000 t + m (any `rng.gen()` would do: keyed by content so that the
 * oracle's replay, which numbers nodes differently, draws the same).
 *
 *   gcc -std=c99 -Iinclude examples/shim_incremental.c -Larroy_amd -larroy_hip -Wl,-rpath,$PWD/arroy_amd -o shim_incremental
 * Exit codes: 0 ok, 2 no GPU, other = failure. */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "arroy_hip.h"
#include "lmdb_walk.h"

#define CHECK(call)                                                         \
    do {                                                                    \
        int rc_ = (call);                                                   \
        if (rc_ != AH_OK) {                                                 \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ah_last_error()); \
            return 1;                                                       \
        }                                                                   \
    } while (0)
#define REQUIRE(cond, msg)                                      \
    do {                                                        \
        if (!(cond)) {                                          \
            fprintf(stderr, "shim_incremental: %s\n", msg);     \
            return 1;                                           \
        }                                                       \
    } while (0)

enum { TREES = 4, SPLIT_AFTER = 8, N_OLD = 90, N_ALL = 100, N_NEW = N_ALL - N_OLD };

/* stage `n` of the walked records (positions `pick[0..n)` in the id-sorted record table) as a finalized dataset */
static int stage(const item_rec *items, const uint32_t *pick, size_t n, uint32_t dims, ah_dataset **out) {
    uint32_t *ids = (uint32_t *)malloc(n * sizeof(uint32_t));
    const uint8_t **ptrs = (const uint8_t **)malloc(n * sizeof(uint8_t *));
    if (!ids || !ptrs) return 1;
    for (size_t i = 0; i < n; i++) {
        ids[i] = items[pick[i]].id;
        ptrs[i] = items[pick[i]].ptr;
    }
    CHECK(ah_dataset_create(AH_EUCLIDEAN, dims, n, 0, out));
    CHECK(ah_dataset_upload_records(*out, ids, ptrs, items[0].len, n));
    CHECK(ah_dataset_finalize(*out));
    free(ids);
    free(ptrs);
    return 0;
}

static int cmp_u32(const void *a, const void *b) {
    const uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return x < y ? -1 : x > y;
}

/* the tree nodes "in the database": id -> node; item lists and normals in growing blobs */
typedef struct {
    ah_node *nodes;
    uint32_t n_nodes, cap_nodes;
    uint8_t *normals;
    size_t normals_len, stride;
    uint32_t *desc;
    size_t desc_len, desc_cap;
} store;

static int store_desc(store *s, uint32_t id, uint32_t tree, const uint32_t *ids, uint32_t count) {
    if (s->desc_len + count > s->desc_cap) return 1;
    memcpy(s->desc + s->desc_len, ids, (size_t)count * 4);
    memset(&s->nodes[id], 0, sizeof(ah_node));
    s->nodes[id].kind = AH_NODE_DESCENDANTS;
    s->nodes[id].tree = tree;
    s->nodes[id].offset = s->desc_len;
    s->nodes[id].count = count;
    s->desc_len += count;
    return 0;
}

int main(int argc, char **argv) {
    const char *path = argc > 1 ? argv[1] : "tests/golden/large_v0_6.mdb";
    const char *dump = argc > 2 ? argv[2] : NULL;
    size_t n_items = 0, psize = 0;
    const char *why = "";
    item_rec *items = lmdb_items(path, &n_items, &psize, &why);
    REQUIRE(items != NULL, why);
    REQUIRE(n_items == N_ALL, "large.mdb holds 100 items (src/tests/upgrade.rs:110)");
    const size_t rec_len = items[0].len;
    const uint32_t dims = (uint32_t)((rec_len - 1 - 4) / 4);
    REQUIRE(dims == 30, "Euclidean records of 30 dimensions expected");
    for (size_t i = 0; i < n_items; i++) REQUIRE(items[i].id == i && items[i].len == rec_len, "ids 0..99, constant-length records");

    int devices = 0;
    CHECK(ah_device_count(&devices));
    if (devices < 1) {
        fprintf(stderr, "no GPU visible: stopping after the LMDB walk\n");
        return 2;
    }

    /* ---- 1. the index before the update: items 0..89, TREES trees ---------------------------------------------------- */
    uint32_t pick[N_ALL];
    for (uint32_t i = 0; i < N_ALL; i++) pick[i] = i;
    ah_dataset *ds_old = NULL;
    if (stage(items, pick, N_OLD, dims, &ds_old)) return 1;
    uint64_t seeds[TREES];
    for (int t = 0; t < TREES; t++) seeds[t] = 42u + (uint64_t)t;
    ah_build_options opt;
    memset(&opt, 0, sizeof opt);
    opt.n_trees = TREES;
    opt.split_after = SPLIT_AFTER;
    opt.tree_seeds = seeds;
    ah_forest *forest = NULL;
    CHECK(ah_build_forest(ds_old, &opt, &forest));
    ah_forest_view fv;
    CHECK(ah_forest_view_get(forest, &fv));
    ah_build_stats bst;
    CHECK(ah_forest_stats(forest, &bst));
    REQUIRE(bst.dummy_normals == 0, "no `normal: None` node expected on this data (the routing coin would need the oracle's node numbering)");

    store db;
    memset(&db, 0, sizeof db);
    db.cap_nodes = (uint32_t)fv.n_nodes + 4096;
    db.nodes = (ah_node *)calloc(db.cap_nodes, sizeof(ah_node));
    db.stride = (size_t)fv.normal_stride;
    db.normals = (uint8_t *)calloc(db.cap_nodes, db.stride);
    db.desc_cap = 4 * (size_t)TREES * N_ALL + 4096;
    db.desc = (uint32_t *)malloc(db.desc_cap * sizeof(uint32_t));
    REQUIRE(db.nodes && db.normals && db.desc, "out of memory");
    memcpy(db.nodes, fv.nodes, (size_t)fv.n_nodes * sizeof(ah_node));
    memcpy(db.normals, fv.normals, (size_t)fv.normals_len);
    memcpy(db.desc, fv.descendants, (size_t)fv.descendants_len * 4);
    db.n_nodes = (uint32_t)fv.n_nodes;
    db.normals_len = (size_t)fv.normals_len;
    db.desc_len = (size_t)fv.descendants_len;
    uint32_t roots[TREES];
    memcpy(roots, fv.roots, sizeof roots);
    const uint64_t vec_off = fv.normal_vector_offset, hdr_off = fv.normal_header_offset;
    CHECK(ah_forest_destroy(forest));
    CHECK(ah_dataset_destroy(ds_old)); /* `Writer::build` of the update starts from LMDB alone */
    const uint32_t n_before = db.n_nodes;

    /* ---- 2a. route the new items through the existing trees ---------------------------------------------------------- */
    ah_dataset *ds_new = NULL;
    if (stage(items, pick + N_OLD, N_NEW, dims, &ds_new)) return 1; /* only `to_insert` is staged */
    ah_node *planes = (ah_node *)malloc((size_t)db.n_nodes * sizeof(ah_node));
    REQUIRE(planes, "out of memory");
    memcpy(planes, db.nodes, (size_t)db.n_nodes * sizeof(ah_node));
    for (uint32_t i = 0; i < db.n_nodes; i++)
        if (planes[i].kind == AH_NODE_DESCENDANTS) planes[i].offset = planes[i].count = 0; /* the routing never reads item lists */
    ah_forest_view image;
    memset(&image, 0, sizeof image);
    image.n_trees = TREES;
    image.n_nodes = db.n_nodes;
    image.roots = roots;
    image.nodes = planes;
    image.normals = db.normals;
    image.normals_len = db.normals_len;
    image.normal_stride = db.stride;
    image.normal_vector_offset = vec_off;
    image.normal_header_offset = hdr_off;
    ah_index *index = NULL;
    CHECK(ah_index_create_from_view(ds_new, &image, &index));
    const uint64_t seed = 0x5EEDull; /* `rng.next_u64()`, once for all trees (src/writer.rs:1128) */
    uint64_t route_seeds[TREES];
    for (int t = 0; t < TREES; t++) route_seeds[t] = seed + roots[t];
    uint32_t new_ids[N_NEW], landed[TREES * N_NEW];
    for (uint32_t i = 0; i < N_NEW; i++) new_ids[i] = N_OLD + i;
    CHECK(ah_route_items(index, new_ids, N_NEW, route_seeds, landed));
    CHECK(ah_index_destroy(index));
    CHECK(ah_dataset_destroy(ds_new));
    free(planes);

    /* descendants_to_update: node -> stored | new.  `grown[k]` = a touched node, its merged list in `merged + moff[k]` */
    uint32_t grown[TREES * N_NEW], gcount[TREES * N_NEW], n_grown = 0;
    size_t moff[TREES * N_NEW];
    uint32_t *merged = (uint32_t *)malloc((size_t)TREES * N_NEW * (SPLIT_AFTER + N_NEW + 1) * sizeof(uint32_t));
    REQUIRE(merged, "out of memory");
    size_t mlen = 0;
    for (uint32_t t = 0; t < TREES; t++)
        for (uint32_t i = 0; i < N_NEW; i++) {
            const uint32_t node = landed[t * N_NEW + i];
            REQUIRE(node < n_before && db.nodes[node].kind == AH_NODE_DESCENDANTS && db.nodes[node].tree == t,
                    "ah_route_items must end at a Descendants node of the tree it walked");
            uint32_t k = 0;
            while (k < n_grown && grown[k] != node) k++;
            if (k == n_grown) { /* first item landing here: start from the stored list */
                grown[n_grown] = node;
                moff[n_grown] = mlen;
                gcount[n_grown] = db.nodes[node].count;
                memcpy(merged + mlen, db.desc + db.nodes[node].offset, (size_t)db.nodes[node].count * 4);
                mlen += SPLIT_AFTER + N_NEW + 1; /* room for every new item */
                n_grown++;
            }
            merged[moff[k] + gcount[k]++] = new_ids[i];
        }
    for (uint32_t k = 0; k < n_grown; k++) qsort(merged + moff[k], gcount[k], 4, cmp_u32);

    /* ---- 2b. small ones are rewritten in place, large ones become sub-trees in ONE call -------------------------------- */
    uint32_t large[TREES * N_NEW], n_large = 0, members[N_ALL], n_members = 0;
    uint8_t is_member[N_ALL];
    memset(is_member, 0, sizeof is_member);
    for (uint32_t k = 0; k < n_grown; k++) {
        if (gcount[k] <= SPLIT_AFTER) {
            if (store_desc(&db, grown[k], db.nodes[grown[k]].tree, merged + moff[k], gcount[k])) return 1;
            continue;
        }
        large[n_large++] = k;
        for (uint32_t j = 0; j < gcount[k]; j++) is_member[merged[moff[k] + j]] = 1;
    }
    for (uint32_t i = 0; i < N_ALL; i++)
        if (is_member[i]) members[n_members++] = i;
    printf("%u new items routed through %d trees: %u descendants touched, %u of them outgrew split_after = %d (%u distinct members)\n",
           (unsigned)N_NEW, TREES, n_grown, n_large, SPLIT_AFTER, n_members);
    REQUIRE(n_large > 0, "the scenario must re-split at least one descendant");
    uint32_t next_free = db.n_nodes;
    if (n_large) {
        ah_dataset *ds_sub = NULL;
        if (stage(items, members, n_members, dims, &ds_sub)) return 1; /* only the members of the large descendants */
        uint64_t sub_seeds[TREES * N_NEW], offsets[TREES * N_NEW + 1];
        uint32_t *sub_ids = (uint32_t *)malloc((size_t)n_large * (SPLIT_AFTER + N_NEW + 1) * sizeof(uint32_t));
        REQUIRE(sub_ids, "out of memory");
        offsets[0] = 0;
        for (uint32_t a = 0; a < n_large; a++) {
            const uint32_t k = large[a];
            memcpy(sub_ids + offsets[a], merged + moff[k], (size_t)gcount[k] * 4);
            offsets[a + 1] = offsets[a] + gcount[k];
            sub_seeds[a] = 1000u + 1000u * (uint64_t)db.nodes[grown[k]].tree + merged[moff[k]];
        }
        ah_build_options sopt;
        memset(&sopt, 0, sizeof sopt);
        sopt.n_trees = n_large;
        sopt.split_after = SPLIT_AFTER;
        sopt.tree_seeds = sub_seeds;
        ah_forest *sub = NULL;
        CHECK(ah_build_subtrees(ds_sub, &sopt, sub_ids, offsets, &sub));
        ah_forest_view sv;
        CHECK(ah_forest_view_get(sub, &sv));
        REQUIRE(sv.n_trees == n_large && sv.normal_stride == db.stride, "one sub-tree per large descendant, same record layout");
        REQUIRE(db.n_nodes + sv.n_nodes <= db.cap_nodes, "node table too small");
        /* sub-forest-local index -> node id: the root takes the descendant's id, the others the next free ones, in the
         * order the nodes arrive (post-order: children before parents, as `TmpNodes::put` receives them) */
        uint32_t *global = (uint32_t *)malloc((size_t)sv.n_nodes * sizeof(uint32_t));
        REQUIRE(global, "out of memory");
        for (uint64_t i = 0; i < sv.n_nodes; i++) global[i] = 0xFFFFFFFFu;
        for (uint32_t a = 0; a < n_large; a++) global[sv.roots[a]] = grown[large[a]];
        for (uint64_t i = 0; i < sv.n_nodes; i++) {
            const ah_node nd = sv.nodes[i];
            if (global[i] == 0xFFFFFFFFu) global[i] = next_free++;
            const uint32_t id = global[i], tree = db.nodes[grown[large[nd.tree]]].tree;
            if (nd.kind == AH_NODE_DESCENDANTS) {
                if (store_desc(&db, id, tree, sv.descendants + nd.offset, nd.count)) return 1;
            } else {
                REQUIRE(nd.left < i && nd.right < i, "children arrive before their parent");
                memset(&db.nodes[id], 0, sizeof(ah_node));
                db.nodes[id].kind = AH_NODE_SPLIT;
                db.nodes[id].tree = tree;
                db.nodes[id].has_normal = nd.has_normal;
                db.nodes[id].left = global[nd.left];
                db.nodes[id].right = global[nd.right];
                if (nd.has_normal) {
                    db.nodes[id].offset = db.normals_len;
                    memcpy(db.normals + db.normals_len, sv.normals + nd.offset, db.stride);
                    db.normals_len += db.stride;
                }
            }
        }
        db.n_nodes = next_free;
        free(global);
        free(sub_ids);
        CHECK(ah_forest_destroy(sub));
        CHECK(ah_dataset_destroy(ds_sub));
    }

    /* ---- 3. the updated forest: every tree holds each of the 100 items exactly once ------------------------------------- */
    uint32_t *stack = (uint32_t *)malloc((size_t)db.n_nodes * sizeof(uint32_t));
    uint8_t *seen_node = (uint8_t *)calloc(db.n_nodes, 1);
    REQUIRE(stack && seen_node, "out of memory");
    for (uint32_t t = 0; t < TREES; t++) {
        uint8_t seen[N_ALL];
        memset(seen, 0, sizeof seen);
        uint32_t total = 0;
        size_t sp = 0;
        stack[sp++] = roots[t];
        while (sp) {
            const uint32_t i = stack[--sp];
            REQUIRE(i < db.n_nodes && !seen_node[i] && db.nodes[i].tree == t, "a node is reachable once, from its own tree");
            seen_node[i] = 1;
            if (db.nodes[i].kind == AH_NODE_SPLIT) {
                stack[sp++] = db.nodes[i].left;
                stack[sp++] = db.nodes[i].right;
            } else {
                REQUIRE(db.nodes[i].count <= SPLIT_AFTER || i < n_before, "a rewritten descendant fits split_after");
                for (uint32_t j = 0; j < db.nodes[i].count; j++) {
                    const uint32_t id = db.desc[db.nodes[i].offset + j];
                    REQUIRE(id < N_ALL && !seen[id], "an item appears once per tree");
                    REQUIRE(j == 0 || id > db.desc[db.nodes[i].offset + j - 1], "descendants ascend");
                    seen[id] = 1;
                    total++;
                }
            }
        }
        REQUIRE(total == N_ALL, "every tree holds all 100 items after the update");
    }
    printf("updated forest: %u nodes (%u before the update), %zu item ids, every tree partitions the 100 items\n", db.n_nodes,
           n_before, db.desc_len);

    if (dump) { /* for the oracle's replay: [n_trees n_nodes normals_len desc_len stride vec_off hdr_off] roots nodes normals desc */
        FILE *f = fopen(dump, "wb");
        REQUIRE(f != NULL, "cannot write the dump");
        const uint64_t head[7] = {TREES, db.n_nodes, db.normals_len, db.desc_len, db.stride, vec_off, hdr_off};
        fwrite(head, sizeof head, 1, f);
        fwrite(roots, sizeof roots, 1, f);
        fwrite(db.nodes, sizeof(ah_node), db.n_nodes, f);
        fwrite(db.normals, 1, db.normals_len, f);
        fwrite(db.desc, 4, db.desc_len, f);
        REQUIRE(fclose(f) == 0, "write failed");
    }
    puts("ok");
    return 0;
}
