/* Plain-C use of libarroy_hip.so (the same calls a Rust `extern "C"` block would make):
 *   gcc -std=c99 -Iinclude examples/c_abi_demo.c -Larroy_amd -larroy_hip -Wl,-rpath,$PWD/arroy_amd -o c_abi_demo
 * Builds a small cosine index on the GPU, searches it, walks the forest through the node sink, and builds the same
 * forest again with the sink fed DURING the build (what a writer that appends to TmpNodes wants). */
#include <stdio.h>
#include <stdlib.h>

#include "arroy_hip.h"
#include "arroy_hip_policy.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != AH_OK) {                                                      \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, ah_last_error());      \
            return 1;                                                            \
        }                                                                        \
    } while (0)

static int count_nodes(void *user, uint32_t tree, uint32_t node, uint8_t kind, uint32_t left, uint32_t right,
                       const void *payload, size_t payload_len) {
    (void)tree; (void)node; (void)left; (void)right; (void)payload; (void)payload_len;
    ((size_t *)user)[kind == AH_NODE_SPLIT ? 0 : 1]++;
    return 0;
}

/* the sink of ah_build_forest_stream: one batch of finished nodes at a time, from one library thread */
typedef struct stream_tally {
    size_t splits, leaves, items, payload_bytes;
    uint32_t next_id;
    int in_order;
} stream_tally;

static int tally_batch(void *user, const ah_node_batch *batch) {
    stream_tally *t = (stream_tally *)user;
    for (uint64_t i = 0; i < batch->n_nodes; i++) {
        const ah_stream_node *nd = &batch->nodes[i];
        if (nd->kind == AH_NODE_SPLIT) {
            t->splits++;
            if (nd->right != nd->left + 1) t->in_order = 0; /* children are id-consecutive */
        } else {
            t->leaves++;
            t->items += nd->count;
        }
    }
    t->payload_bytes += (size_t)batch->payload_len; /* valid only during this call: encode out of it here */
    return 0;
}

int main(void) {
    enum { N = 20000, DIMS = 128, TREES = 4, K = 5 };
    int devices = 0;
    CHECK(ah_device_count(&devices));
    if (devices < 1) {
        fprintf(stderr, "no GPU visible\n");
        return 2;
    }
    /* items 0..N-1 with the library's own synthetic generator (any float data uploaded with
     * ah_dataset_upload_vectors / _upload_records works the same) */
    ah_dataset *ds = NULL;
    CHECK(ah_dataset_create(AH_COSINE, DIMS, N, 0, &ds));
    CHECK(ah_dataset_fill_synthetic(ds, 42, AH_SYNTH_UNIFORM_PM1, N));
    CHECK(ah_dataset_finalize(ds));

    uint64_t seeds[TREES] = {1, 2, 3, 4};
    ah_build_options opt = {0};
    opt.n_trees = TREES;
    opt.tree_seeds = seeds;
    ah_forest *forest = NULL;
    CHECK(ah_build_forest(ds, &opt, &forest));
    size_t kinds[2] = {0, 0};
    CHECK(ah_forest_visit(forest, count_nodes, kinds));
    printf("forest: %d trees, %zu split nodes, %zu descendants nodes\n", TREES, kinds[0], kinds[1]);

    ah_index *index = NULL;
    CHECK(ah_index_create(ds, forest, &index));
    uint32_t query_item = 123, ids[K], counts[1];
    float dists[K];
    CHECK(ah_search_batch(index, NULL, &query_item, 1, K, 0, 0, NULL, 0, 0, ids, dists, counts));
    for (uint32_t i = 0; i < counts[0]; i++) printf("  #%u: item %u at distance %g\n", i, ids[i], dists[i]);
    if (counts[0] == 0 || ids[0] != query_item) {
        fprintf(stderr, "the item itself should come first\n");
        return 3;
    }
    ah_search_stats st;
    CHECK(ah_index_search_stats(index, &st, 0));
    printf("search: %llu queries in %llu calls\n", (unsigned long long)st.queries, (unsigned long long)st.calls);

    /* the same forest, never materialised on the host: nodes arrive level by level while the build runs */
    stream_tally tally = {0, 0, 0, 0, 0, 1};
    uint32_t roots[TREES];
    CHECK(ah_build_forest_stream(ds, &opt, tally_batch, &tally, roots, NULL));
    printf("streamed: %zu split nodes, %zu descendants nodes holding %zu items, %zu payload bytes\n", tally.splits,
           tally.leaves, tally.items, tally.payload_bytes);
    if (tally.splits != kinds[0] || tally.leaves != kinds[1] || tally.items != (size_t)N * TREES || !tally.in_order) {
        fprintf(stderr, "the streamed build should deliver the forest ah_build_forest returns\n");
        return 4;
    }
    CHECK(ah_index_destroy(index));
    CHECK(ah_forest_destroy(forest));
    CHECK(ah_dataset_destroy(ds));
    puts("ok");
    return 0;
}
