/* shim_roundtrip.c — the arroy-side shim of INTEGRATION.md played end to end in plain C, on the reference's own
 * database file (a stand-in for the Rust shim until a box with cargo + liblmdb exists):
 *
 *   1. map an LMDB data file (default tests/golden/large_v0_6.mdb = src/tests/assets/v0_6/large.mdb of the reference),
 *      walk its B-tree and collect, for every item key, the POINTER to the stored value inside the page — what
 *      `ImmutableLeafs::new` collects (src/parallel.rs:271-293); values sit at odd offsets, some on overflow pages;
 *   2. stage the records from those pointers (ah_dataset_upload_records), build a forest (ah_build_forest);
 *   3. drain it through the node sink (ah_forest_visit) the way `TmpNodes::put` receives nodes (src/parallel.rs:130-147):
 *      every node is ENCODED as `NodeCodec` v0.7 stores it (src/node.rs:224-241) —
 *          split plane   [2u8][left u32 BE][right u32 BE][header][vector]      (normal omitted for `normal: None`)
 *          descendants   [1u8][RoaringBitmap, portable serialisation]
 *      into one byte arena + a table of bounds (the tmp file of src/parallel.rs:100-147), children ids being the
 *      global ids handed out in arrival order (children arrive before their parent);
 *   4. DECODE the arena again (src/node.rs:252-273), rebuild the arrays `Reader::open` would hold and mirror them on
 *      the device (ah_index_create_from_view), search, and check the reference's golden result for this database
 *      (src/tests/upgrade.rs:116-128: nns(3) of the zero vector = ids 92, 24, 78).
 *
 *   gcc -std=c99 -Iinclude examples/shim_roundtrip.c -Larroy_amd -larroy_hip -Wl,-rpath,$PWD/arroy_amd -o shim_roundtrip
 * Exit codes: 0 ok, 2 no GPU (everything up to the staging still runs and is checked), other = failure. */
#define _POSIX_C_SOURCE 200809L
#include <fcntl.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "arroy_hip.h"

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
            fprintf(stderr, "shim_roundtrip: %s\n", msg);       \
            return 1;                                           \
        }                                                       \
    } while (0)

#include "lmdb_walk.h"

static void put_be32(uint8_t *p, uint32_t v) { p[0] = (uint8_t)(v >> 24); p[1] = (uint8_t)(v >> 16); p[2] = (uint8_t)(v >> 8); p[3] = (uint8_t)v; }

/* ---- 3. the node sink: NodeCodec v0.7 encoding into a byte arena ------------------------------------------------------ */
typedef struct {
    uint8_t *bytes; size_t len, cap;          /* the tmp file */
    size_t *bounds; uint32_t n_nodes, cap_n;   /* node i occupies bytes[bounds[i], bounds[i + 1]) */
    uint32_t *roots; uint32_t n_roots;         /* last node of every tree = its root (post-order) */
    uint32_t last_tree; int have_tree;
    size_t hs, vs, vec_off, hdr_off;           /* D::Header / vector bytes and where they sit in the library's record */
} tmp_nodes;

static uint8_t *arena_take(tmp_nodes *t, size_t n) {
    if (t->len + n > t->cap) {
        t->cap = 2 * (t->len + n) + 4096;
        t->bytes = (uint8_t *)realloc(t->bytes, t->cap);
        if (!t->bytes) return NULL;
    }
    uint8_t *p = t->bytes + t->len;
    t->len += n;
    return p;
}
/* RoaringBitmap::serialize_into for ascending u32 ids: the portable format without run containers — cookie 12346,
 * container count, (key, cardinality - 1) pairs, byte offsets, then per container a sorted u16 array (<= 4096 values) or a
 * 8 KiB bitmap. */
static int roaring_serialize(tmp_nodes *t, const uint32_t *ids, size_t n) {
    size_t n_cont = 0;
    for (size_t i = 0; i < n; i++)
        if (i == 0 || (ids[i] >> 16) != (ids[i - 1] >> 16)) n_cont++;
    uint8_t *hdr = arena_take(t, 8 + 8 * n_cont);
    if (!hdr) return 1;
    const size_t hdr_at = (size_t)(hdr - t->bytes);
    const uint32_t cookie = 12346, nc = (uint32_t)n_cont;
    memcpy(hdr, &cookie, 4);
    memcpy(hdr + 4, &nc, 4);
    size_t c = 0, i = 0;
    uint32_t offset = (uint32_t)(8 + 8 * n_cont);
    while (i < n) {
        size_t j = i;
        while (j < n && (ids[j] >> 16) == (ids[i] >> 16)) j++;
        const size_t card = j - i;
        const uint16_t key = (uint16_t)(ids[i] >> 16), cm1 = (uint16_t)(card - 1);
        const size_t body = card <= 4096 ? 2 * card : 8192;
        uint8_t *out = arena_take(t, body); /* may move the arena: re-derive the header pointer */
        if (!out) return 1;
        hdr = t->bytes + hdr_at;
        memcpy(hdr + 8 + 4 * c, &key, 2);
        memcpy(hdr + 8 + 4 * c + 2, &cm1, 2);
        memcpy(hdr + 8 + 4 * n_cont + 4 * c, &offset, 4);
        if (card <= 4096) {
            for (size_t k = 0; k < card; k++) {
                const uint16_t low = (uint16_t)(ids[i + k] & 0xFFFFu);
                memcpy(out + 2 * k, &low, 2);
            }
        } else {
            memset(out, 0, 8192);
            for (size_t k = 0; k < card; k++) out[(ids[i + k] & 0xFFFFu) >> 3] |= (uint8_t)(1u << (ids[i + k] & 7u));
        }
        offset += (uint32_t)body;
        c++;
        i = j;
    }
    return 0;
}
static int sink(void *user, uint32_t tree, uint32_t node, uint8_t kind, uint32_t left, uint32_t right, const void *payload,
                size_t payload_len) {
    tmp_nodes *t = (tmp_nodes *)user;
    if (t->n_nodes + 1 >= t->cap_n) {
        t->cap_n = t->cap_n ? 2 * t->cap_n : 256;
        t->bounds = (size_t *)realloc(t->bounds, (t->cap_n + 1) * sizeof(size_t));
        if (!t->bounds) return 1;
    }
    if (node != t->n_nodes) return 2; /* nodes arrive in id order, children first: the global id is the arrival index */
    if (t->have_tree && tree != t->last_tree) t->roots[t->n_roots++] = node - 1; /* the previous tree ended with its root */
    t->last_tree = tree;
    t->have_tree = 1;
    t->bounds[node] = t->len;
    if (kind == AH_NODE_SPLIT) {
        if (left >= node || right >= node) return 3; /* children before parents (src/writer.rs:1235-1258) */
        const size_t n = 1 + 4 + 4 + (payload ? t->hs + t->vs : 0);
        uint8_t *p = arena_take(t, n);
        if (!p) return 1;
        p[0] = 2; /* SPLIT_PLANE_NORMAL_TAG */
        put_be32(p + 1, left);
        put_be32(p + 5, right);
        if (payload) { /* the library's record is [vector .. header ..]; LMDB wants [header][vector] */
            if (payload_len < t->hdr_off + t->hs || payload_len < t->vec_off + t->vs) return 4;
            memcpy(p + 9, (const uint8_t *)payload + t->hdr_off, t->hs);
            memcpy(p + 9 + t->hs, (const uint8_t *)payload + t->vec_off, t->vs);
        }
    } else {
        uint8_t *p = arena_take(t, 1);
        if (!p) return 1;
        p[0] = 1; /* DESCENDANTS_TAG */
        if (roaring_serialize(t, (const uint32_t *)payload, payload_len / 4)) return 1;
    }
    t->n_nodes = node + 1;
    t->bounds[t->n_nodes] = t->len;
    return 0;
}

/* ---- 4. decoding (src/node.rs:252-273) -------------------------------------------------------------------------------- */
static size_t roaring_deserialize(const uint8_t *p, size_t len, uint32_t *out, size_t cap) {
    if (len < 8 || rd32(p) != 12346) return (size_t)-1;
    const uint32_t nc = rd32(p + 4);
    size_t n = 0;
    for (uint32_t c = 0; c < nc; c++) {
        const uint32_t key = rd16(p + 8 + 4 * c), card = (uint32_t)rd16(p + 8 + 4 * c + 2) + 1;
        const uint8_t *body = p + rd32(p + 8 + 4 * nc + 4 * c);
        if (n + card > cap) return (size_t)-1;
        if (card <= 4096) {
            for (uint32_t k = 0; k < card; k++) out[n++] = key << 16 | rd16(body + 2 * k);
        } else {
            for (uint32_t v = 0; v < 65536; v++)
                if (body[v >> 3] >> (v & 7) & 1) out[n++] = key << 16 | v;
        }
    }
    return n;
}

int main(int argc, char **argv) {
    const char *path = argc > 1 ? argv[1] : "tests/golden/large_v0_6.mdb";
    walker w;
    memset(&w, 0, sizeof w);
    size_t psize = 0;
    const char *why = "";
    w.items = lmdb_items(path, &w.n, &psize, &why);
    REQUIRE(w.items != NULL, why);
    REQUIRE(w.n == 100, "large.mdb holds 100 items (src/tests/upgrade.rs:110)");
    const size_t rec_len = w.items[0].len;
    const uint32_t dims = (uint32_t)((rec_len - 1 - 4) / 4);
    REQUIRE(dims == 30 && rec_len == 1 + 4 + 4 * (size_t)dims, "Euclidean records of 30 dimensions expected");
    size_t misaligned = 0;
    for (size_t i = 0; i < w.n; i++) {
        REQUIRE(w.items[i].len == rec_len && w.items[i].ptr[0] == 0, "constant-length LEAF records expected");
        REQUIRE(i == 0 || w.items[i].id > w.items[i - 1].id, "ids must ascend");
        misaligned += ((uintptr_t)(w.items[i].ptr + 5) & 3u) != 0;
    }
    printf("%s: %zu items x %u dims, page size %zu, %zu vectors misaligned inside their pages\n", path, w.n, dims, psize, misaligned);

    int devices = 0;
    CHECK(ah_device_count(&devices));
    if (devices < 1) {
        fprintf(stderr, "no GPU visible: stopping after the LMDB walk\n");
        return 2;
    }

    /* 2. stage from the page pointers, build */
    uint32_t *ids = (uint32_t *)malloc(w.n * sizeof(uint32_t));
    const uint8_t **ptrs = (const uint8_t **)malloc(w.n * sizeof(uint8_t *));
    REQUIRE(ids && ptrs, "out of memory");
    for (size_t i = 0; i < w.n; i++) {
        ids[i] = w.items[i].id;
        ptrs[i] = w.items[i].ptr;
    }
    ah_dataset *ds = NULL;
    CHECK(ah_dataset_create(AH_EUCLIDEAN, dims, w.n, 0, &ds));
    CHECK(ah_dataset_upload_records(ds, ids, ptrs, rec_len, w.n));
    CHECK(ah_dataset_finalize(ds));
    enum { TREES = 10 };
    uint64_t seeds[TREES];
    for (int t = 0; t < TREES; t++) seeds[t] = 42u + (uint64_t)t;
    ah_build_options opt;
    memset(&opt, 0, sizeof opt);
    opt.n_trees = TREES;
    opt.split_after = 12; /* a few levels per tree on 100 items */
    opt.tree_seeds = seeds;
    ah_forest *forest = NULL;
    CHECK(ah_build_forest(ds, &opt, &forest));
    ah_forest_view fv;
    CHECK(ah_forest_view_get(forest, &fv));

    /* 3. drain through the sink: every node encoded as NodeCodec stores it */
    tmp_nodes tmp;
    memset(&tmp, 0, sizeof tmp);
    tmp.hs = ah_header_size(AH_EUCLIDEAN);
    tmp.vs = ah_vector_size(AH_EUCLIDEAN, dims);
    tmp.vec_off = (size_t)fv.normal_vector_offset;
    tmp.hdr_off = (size_t)fv.normal_header_offset;
    tmp.roots = (uint32_t *)malloc(TREES * sizeof(uint32_t));
    REQUIRE(tmp.roots, "out of memory");
    CHECK(ah_forest_visit(forest, sink, &tmp));
    REQUIRE(tmp.n_nodes == fv.n_nodes && tmp.n_nodes > 0, "the sink must see every node");
    tmp.roots[tmp.n_roots++] = tmp.n_nodes - 1;
    REQUIRE(tmp.n_roots == TREES, "one root per tree");
    for (int t = 0; t < TREES; t++) REQUIRE(tmp.roots[t] == fv.roots[t], "roots are the last node of every tree");
    CHECK(ah_forest_destroy(forest)); /* from here on only the encoded bytes exist */

    /* 4. decode them again into the arrays `Reader::open` would hold */
    ah_node *nodes = (ah_node *)calloc(tmp.n_nodes, sizeof(ah_node));
    const size_t stride = tmp.hs + tmp.vs; /* [header][vector], as stored */
    uint8_t *normals = (uint8_t *)calloc((size_t)tmp.n_nodes + 1, stride);
    uint32_t *desc = (uint32_t *)malloc((size_t)TREES * w.n * sizeof(uint32_t));
    REQUIRE(nodes && normals && desc, "out of memory");
    size_t n_desc = 0, n_split = 0, n_normals = 0;
    for (uint32_t i = 0; i < tmp.n_nodes; i++) {
        const uint8_t *p = tmp.bytes + tmp.bounds[i];
        const size_t len = tmp.bounds[i + 1] - tmp.bounds[i];
        REQUIRE(len >= 1, "empty node");
        if (p[0] == 2) {
            REQUIRE(len == 9 || len == 9 + stride, "split node: [tag][left][right] (+ [header][vector])");
            nodes[i].kind = AH_NODE_SPLIT;
            nodes[i].left = be32(p + 1);
            nodes[i].right = be32(p + 5);
            nodes[i].has_normal = len > 9;
            if (len > 9) {
                nodes[i].offset = (uint64_t)n_normals * stride;
                memcpy(normals + nodes[i].offset, p + 9, stride);
                n_normals++;
            }
            n_split++;
        } else {
            REQUIRE(p[0] == 1, "unknown node tag");
            const size_t got = roaring_deserialize(p + 1, len - 1, desc + n_desc, (size_t)TREES * w.n - n_desc);
            REQUIRE(got != (size_t)-1 && got > 0, "descendants do not decode");
            for (size_t k = 1; k < got; k++) REQUIRE(desc[n_desc + k] > desc[n_desc + k - 1], "descendants must ascend");
            nodes[i].kind = AH_NODE_DESCENDANTS;
            nodes[i].offset = n_desc;
            nodes[i].count = (uint32_t)got;
            n_desc += got;
        }
    }
    /* tree index of every node (the view wants it): walk down from the roots */
    {
        uint32_t *stack = (uint32_t *)malloc(tmp.n_nodes * sizeof(uint32_t));
        REQUIRE(stack, "out of memory");
        for (uint32_t t = 0; t < TREES; t++) {
            size_t sp = 0;
            stack[sp++] = tmp.roots[t];
            while (sp) {
                const uint32_t i = stack[--sp];
                nodes[i].tree = t;
                if (nodes[i].kind == AH_NODE_SPLIT) {
                    stack[sp++] = nodes[i].left;
                    stack[sp++] = nodes[i].right;
                }
            }
        }
        free(stack);
    }
    REQUIRE(n_desc == (size_t)TREES * w.n, "every tree must hold every item exactly once");
    ah_forest_view view;
    memset(&view, 0, sizeof view);
    view.n_trees = TREES;
    view.n_nodes = tmp.n_nodes;
    view.roots = tmp.roots;
    view.nodes = nodes;
    view.normals = normals;
    view.normals_len = (uint64_t)n_normals * stride;
    view.normal_stride = stride;
    view.normal_header_offset = 0;
    view.normal_vector_offset = tmp.hs;
    view.descendants = desc;
    view.descendants_len = n_desc;
    ah_index *index = NULL;
    CHECK(ah_index_create_from_view(ds, &view, &index));

    /* the golden search of src/tests/upgrade.rs:116-128: nns(3), search_k = 100, the zero vector */
    float query[30];
    memset(query, 0, sizeof query);
    uint32_t got_ids[3], counts[1];
    float got_d[3];
    CHECK(ah_search_batch(index, query, NULL, 1, 3, 100, 0, NULL, 0, 0, got_ids, got_d, counts));
    static const uint32_t want_ids[3] = {92, 24, 78};
    static const char *want_d[3] = {"2.4881108", "2.5068686", "2.5809734"};
    REQUIRE(counts[0] == 3, "three neighbours expected");
    for (int k = 0; k < 3; k++) {
        printf("  id(%u): distance(%.7f)\n", got_ids[k], (double)got_d[k]);
        REQUIRE(got_ids[k] == want_ids[k] && got_d[k] == strtof(want_d[k], NULL), "differs from the reference's snapshot");
    }
    printf("%u nodes (%zu split planes, %zu descendants) encoded into %zu bytes, decoded, mirrored on the device and searched\n",
           tmp.n_nodes, n_split, tmp.n_nodes - n_split, tmp.len);
    CHECK(ah_index_destroy(index));
    CHECK(ah_dataset_destroy(ds));
    puts("ok");
    return 0;
}
