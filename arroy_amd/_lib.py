"""ctypes binding of libarroy_hip.so (the C ABI declared in include/arroy_hip.h).

There is deliberately NO fallback: if the shared object is missing or a call fails, an exception is
raised.  The product never imports anything from ``oracle/``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
# AH_LIB_PATH: load another build of the library (A/B measurements of kernel variants); default: the in-tree build
LIB_PATH = os.environ.get("AH_LIB_PATH") or os.path.join(HERE, "libarroy_hip.so")
CSRC = os.path.join(HERE, "csrc")

# ah_status (include/arroy_hip.h) -> arroy::Error (src/error.rs:6-85)
AH_OK = 0
STATUS_NAMES = {
    1: "InvalidVecDimension", 2: "BuildCancelled", 3: "Panic(device)", 4: "Panic(out of memory)",
    5: "Panic(invalid argument)", 6: "MissingKey", 7: "Panic(not finalized)", 8: "Panic(need preprocess)",
}
AH_SPLIT_SAMPLES = 12


class ArroyHipError(RuntimeError):
    def __init__(self, status: int, message: str, detail=None):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status
        self.message = message
        # the typed fields of arroy::Error (ah_last_error_detail): InvalidVecDimension {expected, received},
        # MissingKey {item}
        self.item = detail.item if detail is not None else 0
        self.expected = detail.expected if detail is not None else 0
        self.received = detail.received if detail is not None else 0


class InvalidVecDimension(ArroyHipError):
    """arroy::Error::InvalidVecDimension (src/error.rs:17-23)."""


class BuildCancelled(ArroyHipError):
    """arroy::Error::BuildCancelled (src/error.rs:54-55)."""


class MissingKey(ArroyHipError):
    """arroy::Error::MissingKey (src/error.rs:38-46)."""


class AhNode(C.Structure):
    _fields_ = [("kind", C.c_uint8), ("has_normal", C.c_uint8), ("reserved", C.c_uint16), ("tree", C.c_uint32),
                ("left", C.c_uint32), ("right", C.c_uint32), ("offset", C.c_uint64), ("count", C.c_uint32),
                ("depth", C.c_uint32)]


class AhForestView(C.Structure):
    _fields_ = [("n_trees", C.c_uint32), ("n_nodes", C.c_uint64), ("roots", C.POINTER(C.c_uint32)),
                ("nodes", C.POINTER(AhNode)), ("normals", C.POINTER(C.c_uint8)), ("normals_len", C.c_uint64),
                ("normal_stride", C.c_uint64), ("normal_vector_offset", C.c_uint64), ("normal_header_offset", C.c_uint64),
                ("descendants", C.POINTER(C.c_uint32)),
                ("descendants_len", C.c_uint64)]


PROGRESS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64)


class AhBuildOptions(C.Structure):
    _fields_ = [("n_trees", C.c_uint32), ("split_after", C.c_uint32), ("tree_seeds", C.POINTER(C.c_uint64)),
                ("cancel", C.POINTER(C.c_int)), ("progress", PROGRESS_FN), ("progress_user", C.c_void_p),
                ("max_trees_in_flight", C.c_uint32), ("margin_mode", C.c_uint32), ("max_host_threads", C.c_uint32),
                ("reserved0", C.c_uint32)]


class AhErrorDetail(C.Structure):
    _fields_ = [("status", C.c_int), ("item", C.c_uint32), ("expected", C.c_uint64), ("received", C.c_uint64)]


# ah_margin_mode (include/arroy_hip.h)
MARGIN_AUTO, MARGIN_NODE_MAJOR = 0, 1
MARGIN_ROWS = {2: 2, 4: 4, 8: 8, 16: 16}
MARGIN_ROWS_LDS = {8: 0x108, 16: 0x110}
MARGIN_DENSE_MFMA = 0x200
MARGIN_EXACT_ONLY = 0x1000
# index of every kernel family in ah_build_stats.margin_mode_launches
MODE_LAUNCH_INDEX = {1: 0, 2: 1, 4: 2, 8: 3, 16: 4, 0x108: 5, 0x110: 6}


class AhBuildStats(C.Structure):
    _fields_ = [("seconds_total", C.c_double), ("seconds_device", C.c_double), ("seconds_margin", C.c_double),
                ("margin_evaluations", C.c_uint64), ("margin_launches", C.c_uint64), ("margin_row_passes", C.c_uint64), ("split_nodes", C.c_uint64),
                ("descendant_nodes", C.c_uint64), ("dummy_normals", C.c_uint64), ("retries", C.c_uint64),
                ("levels", C.c_uint32), ("margin_mode_launches", C.c_uint64 * 8), ("screened_launches", C.c_uint64),
                ("screen_fallbacks", C.c_uint64), ("screen_violations", C.c_uint64),
                ("dense_launches", C.c_uint64), ("dense_columns", C.c_uint64),
                ("rows_xcd_launches", C.c_uint64), ("rows_nt_launches", C.c_uint64), ("rows_split_launches", C.c_uint64),
                ("screen8_pairs", C.c_uint64), ("screen8_decided", C.c_uint64), ("screen8b_decided", C.c_uint64),
                ("screen_unavailable", C.c_uint32),
                ("tail_groups", C.c_uint32), ("seconds_setup", C.c_double), ("seconds_after_device", C.c_double),
                ("host_blob_recycled", C.c_uint64), ("seconds_reserve", C.c_double), ("seconds_reserve_wait", C.c_double)]


class AhRerankStats(C.Structure):
    _fields_ = [("calls", C.c_uint64), ("queries", C.c_uint64), ("candidates", C.c_uint64), ("seconds_wall", C.c_double),
                ("seconds_prep", C.c_double), ("seconds_ids", C.c_double), ("seconds_enqueue", C.c_double),
                ("seconds_sync_wait", C.c_double), ("seconds_device_span", C.c_double), ("queries_screened", C.c_uint64),
                ("survivors", C.c_uint64), ("chunks_int8", C.c_uint64), ("chunks_int8_retried", C.c_uint64)]


class AhSearchStats(C.Structure):
    _fields_ = [(f, C.c_uint64) for f in (
        "calls", "chunks", "queries", "descent_wave_small", "descent_wave_big", "descent_octet_lds", "descent_octet_global",
        "dedup_flag_bitmap", "dedup_flag_hash", "dedup_sorted_bitmap", "dedup_sort_lds", "dedup_sort_global",
        "rerank_tiles", "rerank_sorted", "tile_visits", "tile_units_16", "tile_units_8", "tile_units_4",
        "fallback_chunks", "fallback_non_finite", "fallback_select", "fallback_queue", "fallback_visits", "fallback_launch",
        "filtered_queries", "leaf_kept_passes", "rerank_screened", "screen_survivors", "descent_block", "rerank_screened8",
        "screen8_retried_chunks", "descent_multi")]


class AhStreamNode(C.Structure):
    _fields_ = [("id", C.c_uint32), ("tree", C.c_uint32), ("kind", C.c_uint8), ("has_normal", C.c_uint8), ("reserved", C.c_uint16),
                ("left", C.c_uint32), ("right", C.c_uint32), ("count", C.c_uint32), ("depth", C.c_uint32),
                ("payload_offset", C.c_uint64)]


class AhNodeBatch(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("level", C.c_uint32), ("n_nodes", C.c_uint64), ("nodes", C.POINTER(AhStreamNode)),
                ("payload", C.POINTER(C.c_uint8)), ("payload_len", C.c_uint64), ("normal_stride", C.c_uint64),
                ("normal_vector_offset", C.c_uint64), ("normal_header_offset", C.c_uint64)]


NODE_BATCH_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(AhNodeBatch))


# name -> (restype, argtypes): exactly the declarations of include/arroy_hip.h
_VP, _U32P, _F32P, _U64P = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
SIGNATURES = {
    "ah_header_size": (C.c_size_t, [C.c_int]),
    "ah_vector_size": (C.c_size_t, [C.c_int, C.c_uint32]),
    "ah_abi_version": (C.c_int, []),
    "ah_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "ah_last_error": (C.c_char_p, []),
    "ah_last_error_detail": (C.c_int, [C.POINTER(AhErrorDetail)]),
    "ah_dataset_upload_flush": (C.c_int, [_VP]),
    "ah_dataset_set_preprocessed": (C.c_int, [_VP, C.c_int]),
    "ah_dataset_replicate": (C.c_int, [_VP, C.c_int, C.POINTER(C.c_void_p)]),
    "ah_dataset_create": (C.c_int, [C.c_int, C.c_uint32, C.c_uint64, C.c_int, C.POINTER(C.c_void_p)]),
    "ah_dataset_upload_records": (C.c_int, [_VP, _U32P, _VP, C.c_size_t, C.c_size_t]),
    "ah_dataset_upload_vectors": (C.c_int, [_VP, _U32P, _F32P, C.c_size_t]),
    "ah_dataset_fill_synthetic": (C.c_int, [_VP, C.c_uint64, C.c_int, C.c_uint64]),
    "ah_dataset_finalize": (C.c_int, [_VP]),
    "ah_dataset_reserve_build": (C.c_int, [_VP, C.c_uint32, C.c_uint32]),
    "ah_dataset_rerank_stats": (C.c_int, [_VP, C.POINTER(AhRerankStats), C.c_int]),
    "ah_dataset_len": (C.c_int, [_VP, C.POINTER(C.c_uint64)]),
    "ah_dataset_item_vector": (C.c_int, [_VP, C.c_uint32, _F32P]),
    "ah_dataset_read_headers": (C.c_int, [_VP, C.c_uint64, C.c_uint64, _VP]),
    "ah_dataset_destroy": (C.c_int, [_VP]),
    "ah_preprocess_dot": (C.c_int, [_VP, C.POINTER(C.c_float)]),
    "ah_distances_by_vector": (C.c_int, [_VP, _F32P, _U32P, C.c_size_t, _F32P]),
    "ah_distances_by_item": (C.c_int, [_VP, C.c_uint32, _U32P, C.c_size_t, _F32P]),
    "ah_rerank_by_vector": (C.c_int, [_VP, _F32P, _U32P, C.c_size_t, C.c_size_t, _U32P, _F32P,
                                      C.POINTER(C.c_size_t)]),
    "ah_rerank_by_item": (C.c_int, [_VP, C.c_uint32, _U32P, C.c_size_t, C.c_size_t, _U32P, _F32P,
                                    C.POINTER(C.c_size_t)]),
    "ah_rerank_batch": (C.c_int, [_VP, _F32P, C.c_size_t, _U32P, _U64P, C.c_size_t, _U32P, _F32P, _U32P]),
    "ah_split_sides": (C.c_int, [_VP, _VP, _VP, _U32P, C.c_size_t, _VP, C.POINTER(C.c_uint64), _F32P]),
    "ah_margins": (C.c_int, [_VP, _VP, _VP, _U32P, C.c_size_t, _F32P]),
    "ah_create_split": (C.c_int, [_VP, _U32P, _VP, _VP]),
    "ah_build_forest": (C.c_int, [_VP, C.POINTER(AhBuildOptions), C.POINTER(C.c_void_p)]),
    "ah_build_forest_stream": (C.c_int, [_VP, C.POINTER(AhBuildOptions), NODE_BATCH_FN, _VP, _U32P, C.POINTER(AhBuildStats)]),
    "ah_build_subtrees": (C.c_int, [_VP, C.POINTER(AhBuildOptions), _U32P, _U64P, C.POINTER(C.c_void_p)]),
    "ah_forest_view_get": (C.c_int, [_VP, C.POINTER(AhForestView)]),
    "ah_forest_stats": (C.c_int, [_VP, C.POINTER(AhBuildStats)]),
    "ah_forest_digest": (C.c_int, [_VP, _U64P, C.POINTER(C.c_uint64)]),
    "ah_host_cache_trim": (C.c_int, [C.POINTER(C.c_uint64)]),
    "ah_device_cache_trim": (C.c_int, [C.c_int, C.POINTER(C.c_uint64)]),
    "ah_synth_rows_host": (C.c_int, [C.c_uint64, C.c_int, C.c_uint64, C.c_uint64, C.c_uint32, _F32P]),
    "ah_tuning_set": (C.c_int, [C.c_char_p, C.c_int64]),
    "ah_tuning_get": (C.c_int, [C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "ah_tuning_reset": (C.c_int, []),
    "ah_debug_launch_coverage": (C.c_int, [C.c_int, C.c_int, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, _U32P, C.c_uint64]),
    "ah_forest_digest_keyed": (C.c_int, [_VP, _U64P, _U64P]),
    "ah_device_cache_stats": (C.c_int, [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "ah_debug_dense_tiles": (C.c_int, [C.c_uint64, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "ah_forest_visit": (C.c_int, [_VP, _VP, _VP]),
    "ah_forest_destroy": (C.c_int, [_VP]),
    "ah_index_create": (C.c_int, [_VP, _VP, C.POINTER(C.c_void_p)]),
    "ah_index_create_from_view": (C.c_int, [_VP, C.POINTER(AhForestView), C.POINTER(C.c_void_p)]),
    "ah_index_destroy": (C.c_int, [_VP]),
    "ah_search_batch": (C.c_int, [_VP, _F32P, _U32P, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, _U32P, C.c_size_t,
                                  C.c_int, _U32P, _F32P, _U32P]),
    "ah_index_search_stats": (C.c_int, [_VP, C.POINTER(AhSearchStats), C.c_int]),
    "ah_route_items": (C.c_int, [_VP, _U32P, C.c_size_t, _U64P, _U32P]),
    "ah_bench_scan": (C.c_int, [_VP, C.c_uint32, C.c_uint64, C.c_uint32, _F32P, C.POINTER(C.c_double)]),
    "ah_bench_memcpy": (C.c_int, [C.c_int, C.c_uint64, C.c_uint32, C.POINTER(C.c_double)]),
    "ah_bench_read": (C.c_int, [C.c_int, C.c_uint64, C.c_uint32, C.POINTER(C.c_double)]),
    "ah_device_name": (C.c_int, [C.c_int, C.c_char_p, C.c_size_t]),
}

_lib = None


def build(force: bool = False) -> str:
    """Compile libarroy_hip.so for gfx950 with the in-tree Makefile (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean", "-s"])
    subprocess.check_call(["make", "-C", CSRC, "-j4", "-s"])
    return LIB_PATH


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(arroy_amd has no CPU fallback)")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(status: int) -> None:
    if status == AH_OK:
        return
    msg = lib().ah_last_error().decode("utf-8", "replace")
    det = AhErrorDetail()
    lib().ah_last_error_detail(C.byref(det))
    cls = {1: InvalidVecDimension, 2: BuildCancelled, 6: MissingKey}.get(status, ArroyHipError)
    raise cls(status, msg, det if det.status == status else None)


def device_count() -> int:
    n = C.c_int(0)
    check(lib().ah_device_count(C.byref(n)))
    return n.value


def device_name(device: int = 0) -> str:
    buf = C.create_string_buffer(256)
    check(lib().ah_device_name(device, buf, 256))
    return buf.value.decode()


def tuning_set(name: str, value: int) -> None:
    """ah_tuning_set: a measurement / test aid that steers the schedule of the kernels, never a result."""
    check(lib().ah_tuning_set(name.encode(), int(value)))


def tuning_get(name: str) -> tuple[int, int]:
    """(current value, built-in default) of a tunable."""
    v, d = C.c_int64(0), C.c_int64(0)
    check(lib().ah_tuning_get(name.encode(), C.byref(v), C.byref(d)))
    return int(v.value), int(d.value)


class tuning:
    """`with tuning(AH_ROWS_XCD=0, AH_DENSE=1): ...` — set tunables for a block, put the previous values back after."""

    def __init__(self, **values):
        self.values = values
        self.saved = {}

    def __enter__(self):
        for k, v in self.values.items():
            self.saved[k] = tuning_get(k)[0]
            tuning_set(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.saved.items():
            tuning_set(k, v)
        return False


def host_cache_trim() -> int:
    """ah_host_cache_trim: give the recycled host blobs of destroyed forests back to the system; returns the bytes released."""
    out = C.c_uint64(0)
    check(lib().ah_host_cache_trim(C.byref(out)))
    return int(out.value)


def device_cache_trim(device: int = -1) -> int:
    """ah_device_cache_trim: give the idle HBM blocks of the caching allocator back to the driver; returns the bytes released."""
    out = C.c_uint64(0)
    check(lib().ah_device_cache_trim(int(device), C.byref(out)))
    return int(out.value)


def synth_rows_host(seed: int, distribution: int, n: int, dims: int, first_item: int = 0, out=None):
    """ah_synth_rows_host (benchmark harness): the policy generator's rows in host memory."""
    import numpy as np
    if out is None:
        out = np.empty((n, dims), dtype=np.float32)
    assert out.dtype == np.float32 and out.flags.c_contiguous and out.size == n * dims
    check(lib().ah_synth_rows_host(seed, distribution, first_item, n, dims, out.ctypes.data_as(C.c_void_p)))
    return out


def launch_coverage(kind: int, n_rows: int, dims: int, a: int, b: int = 0, device: int = 0):
    """ah_debug_launch_coverage: how often the block -> work-item map of a build launch serves every work item."""
    import numpy as np
    if kind == 0:
        shape = (b, n_rows)
    elif kind == 1:
        tr, tc = C.c_uint32(0), C.c_uint32(0)
        check(lib().ah_debug_dense_tiles(n_rows, a, C.byref(tr), C.byref(tc)))
        shape = ((n_rows + tr.value - 1) // tr.value, (a + tc.value - 1) // tc.value)
    else:
        shape = (a, (n_rows + 1023) // 1024)
    out = np.zeros(shape, dtype=np.uint32)
    check(lib().ah_debug_launch_coverage(device, kind, n_rows, dims, a, b, out.ctypes.data_as(C.c_void_p), out.size))
    return out


def bench_read(device: int, nbytes: int, iterations: int) -> float:
    """Read-only stream over `nbytes` x `iterations`; returns total milliseconds (HIP events)."""
    ms = C.c_double(0)
    check(lib().ah_bench_read(device, nbytes, iterations, C.byref(ms)))
    return ms.value


def bench_memcpy(device: int, nbytes: int, iterations: int) -> float:
    """Device-to-device copy of `nbytes` x `iterations`; returns total milliseconds (HIP events)."""
    ms = C.c_double(0)
    check(lib().ah_bench_memcpy(device, nbytes, iterations, C.byref(ms)))
    return ms.value


def device_cache_stats(device: int = -1):
    """(live, idle) bytes of HBM the library holds on `device` (ah_device_cache_stats; < 0: every device)."""
    live, idle = C.c_uint64(0), C.c_uint64(0)
    check(lib().ah_device_cache_stats(device, C.byref(live), C.byref(idle)))
    return live.value, idle.value
