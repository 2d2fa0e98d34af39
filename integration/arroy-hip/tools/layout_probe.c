/* Prints the size and the field offsets of every struct of include/arroy_hip.h that crosses the FFI by value or by
 * pointer-to-struct, one "Struct field offset" / "Struct = size" line each.  integration/arroy-hip/tools/gen_layout.py turns the
 * output into the `const _: () = assert!(...)` block of src/hip.rs; tests/test_integration_patch.py compiles and runs this
 * program and checks that the block in hip.rs is exactly what it produces (gcc x86-64 is the C ABI rustc's #[repr(C)] follows). */
#include <stddef.h>
#include <stdio.h>

#include "arroy_hip.h"

#define SIZE(T, R) printf("%s = %zu %zu\n", R, sizeof(T), _Alignof(T))
#define OFF(T, R, f) printf("%s %s %zu\n", R, #f, offsetof(T, f))

int main(void) {
    SIZE(ah_build_options, "AhBuildOptions");
    OFF(ah_build_options, "AhBuildOptions", n_trees);
    OFF(ah_build_options, "AhBuildOptions", split_after);
    OFF(ah_build_options, "AhBuildOptions", tree_seeds);
    OFF(ah_build_options, "AhBuildOptions", cancel);
    OFF(ah_build_options, "AhBuildOptions", progress);
    OFF(ah_build_options, "AhBuildOptions", progress_user);
    OFF(ah_build_options, "AhBuildOptions", max_trees_in_flight);
    OFF(ah_build_options, "AhBuildOptions", margin_mode);
    OFF(ah_build_options, "AhBuildOptions", max_host_threads);
    OFF(ah_build_options, "AhBuildOptions", reserved0);
    SIZE(ah_error_detail, "AhErrorDetail");
    OFF(ah_error_detail, "AhErrorDetail", status);
    OFF(ah_error_detail, "AhErrorDetail", item);
    OFF(ah_error_detail, "AhErrorDetail", expected);
    OFF(ah_error_detail, "AhErrorDetail", received);
    SIZE(ah_stream_node, "AhStreamNode");
    OFF(ah_stream_node, "AhStreamNode", id);
    OFF(ah_stream_node, "AhStreamNode", tree);
    OFF(ah_stream_node, "AhStreamNode", kind);
    OFF(ah_stream_node, "AhStreamNode", has_normal);
    OFF(ah_stream_node, "AhStreamNode", reserved);
    OFF(ah_stream_node, "AhStreamNode", left);
    OFF(ah_stream_node, "AhStreamNode", right);
    OFF(ah_stream_node, "AhStreamNode", count);
    OFF(ah_stream_node, "AhStreamNode", depth);
    OFF(ah_stream_node, "AhStreamNode", payload_offset);
    SIZE(ah_node_batch, "AhNodeBatch");
    OFF(ah_node_batch, "AhNodeBatch", kind);
    OFF(ah_node_batch, "AhNodeBatch", level);
    OFF(ah_node_batch, "AhNodeBatch", n_nodes);
    OFF(ah_node_batch, "AhNodeBatch", nodes);
    OFF(ah_node_batch, "AhNodeBatch", payload);
    OFF(ah_node_batch, "AhNodeBatch", payload_len);
    OFF(ah_node_batch, "AhNodeBatch", normal_stride);
    OFF(ah_node_batch, "AhNodeBatch", normal_vector_offset);
    OFF(ah_node_batch, "AhNodeBatch", normal_header_offset);
    SIZE(ah_node, "AhNode");
    OFF(ah_node, "AhNode", kind);
    OFF(ah_node, "AhNode", has_normal);
    OFF(ah_node, "AhNode", reserved);
    OFF(ah_node, "AhNode", tree);
    OFF(ah_node, "AhNode", left);
    OFF(ah_node, "AhNode", right);
    OFF(ah_node, "AhNode", offset);
    OFF(ah_node, "AhNode", count);
    OFF(ah_node, "AhNode", depth);
    SIZE(ah_forest_view, "AhForestView");
    OFF(ah_forest_view, "AhForestView", n_trees);
    OFF(ah_forest_view, "AhForestView", n_nodes);
    OFF(ah_forest_view, "AhForestView", roots);
    OFF(ah_forest_view, "AhForestView", nodes);
    OFF(ah_forest_view, "AhForestView", normals);
    OFF(ah_forest_view, "AhForestView", normals_len);
    OFF(ah_forest_view, "AhForestView", normal_stride);
    OFF(ah_forest_view, "AhForestView", normal_vector_offset);
    OFF(ah_forest_view, "AhForestView", normal_header_offset);
    OFF(ah_forest_view, "AhForestView", descendants);
    OFF(ah_forest_view, "AhForestView", descendants_len);
    return 0;
}
