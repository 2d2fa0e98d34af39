/* Plain-C use of libarroy_hip.so (the same calls a Rust `extern "C"` block would make):
 *   gcc -std=c99 -Iinclude examples/c_abi_demo.c -Larroy_amd -larroy_hip -Wl,-rpath,$PWD/arroy_amd -o c_abi_demo
 * Builds a small cosine index on the GPU, searches it and walks the forest through the node sink. */
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
    CHECK(ah_index_destroy(index));
    CHECK(ah_forest_destroy(forest));
    CHECK(ah_dataset_destroy(ds));
    puts("ok");
    return 0;
}
