//! MI355X back end of arroy's hot loops (cargo feature `hip`).
//!
//! Thin `extern "C"` bindings of `libarroy_hip.so` (`include/arroy_hip.h`, ABI v7) plus the places where arroy hands a
//! whole *loop* to the GPU instead of running it per item:
//!
//! * [`stage_leafs`]  — `ImmutableLeafs::new` (`src/parallel.rs`): the stored item records, straight from their LMDB
//!   pages, into HBM (`ah_dataset_upload_records`);
//! * [`build_new_trees`] — the `rayon::scope` over the root descendants + `make_tree_in_file` (`src/writer.rs`): whole
//!   trees built on the device and handed back node by node WHILE they are built (`ah_build_forest_stream`), encoded
//!   with `NodeCodec` and appended to a `TmpNodes` exactly like the CPU path does;
//! * [`Rerank`] — the distance loop + `median_based_top_k` of `Reader::nns_by_leaf` (`src/reader.rs`);
//! * [`HipSearch`] — the WHOLE `nns_by_leaf` (best-first descent, candidate collection, sort + dedup, re-rank, top-k) for a
//!   batch of queries in one call (`ah_search_batch`), over a mirror of the tree nodes made by `Reader::stage_on_gpu`
//!   ([`ForestImage`] -> `ah_index_create_from_view`);
//! * [`route_into_current_trees`] — `Writer::insert_items_in_current_trees` (`insert_items_in_descendants_from_frozen_reader`
//!   for every tree): a big insertion is sent down all the existing trees in one call (`ah_route_items`) over a mirror of the
//!   tree nodes made from `ImmutableTrees` ([`image_of_trees`]);
//! * [`build_large_descendants`] — the `incremental_index_large_descendant` tasks of one
//!   `insert_descendants_in_file_and_spawn_tasks` call: every Descendants node that outgrew `split_after` becomes a sub-tree in
//!   ONE device call (`ah_build_subtrees`), its nodes appended to the caller's `TmpNodes`;
//! * [`preprocess_dot_records`] — both passes of `DotProduct::preprocess` (the max norm, every item's `extra_dim` / `norm`) on
//!   the device, from the pages the reference's first pass walks; the second pass only writes the headers back.
//!
//! LMDB, roaring, `NodeCodec`, `TmpNodes`, node-id allocation, the RNG and the public API stay as they are.
//! Link with `RUSTFLAGS="-L <dir of libarroy_hip.so>"`; the library needs `libamdhip64` at run time.

use std::borrow::Cow;
use std::ffi::CStr;
use std::marker::PhantomData;
use std::mem::{align_of, offset_of, size_of, MaybeUninit};
use std::os::raw::{c_char, c_int, c_void};
use std::sync::atomic::{AtomicI32, AtomicU64, Ordering};

use bytemuck::pod_read_unaligned;
use nohash::IntMap;
use rand::{Rng, RngCore};
use roaring::RoaringBitmap;

use crate::distance::Distance;
use crate::node::{Descendants, Leaf, Node, SplitPlaneNormal};
use crate::parallel::{ConcurrentNodeIds, ImmutableLeafs, ImmutableTrees, TmpNodes};
use crate::unaligned_vector::UnalignedVector;
use crate::writer::BuildOption;
use crate::{Error, ItemId, Key, Result};

pub const AH_ABI_VERSION: c_int = 7;
const AH_NODE_DESCENDANTS: u8 = 1;
const AH_NODE_SPLIT: u8 = 2;

#[repr(C)]
pub struct AhDataset {
    _p: [u8; 0],
}

#[repr(C)]
pub struct AhIndex {
    _p: [u8; 0],
}

#[repr(C)]
pub struct AhForest {
    _p: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy)]
pub struct AhNode {
    pub kind: u8,       // AH_NODE_DESCENDANTS | AH_NODE_SPLIT
    pub has_normal: u8, // 0 = `normal: None`
    pub reserved: u16,
    pub tree: u32,
    pub left: u32,      // forest-local node indices (SPLIT)
    pub right: u32,
    pub offset: u64,    // SPLIT: byte offset into `normals`; DESCENDANTS: first id index
    pub count: u32,
    pub depth: u32,
}

#[repr(C)]
pub struct AhForestView {
    pub n_trees: u32,
    pub n_nodes: u64,
    pub roots: *const u32,
    pub nodes: *const AhNode,
    pub normals: *const u8,
    pub normals_len: u64,
    pub normal_stride: u64,
    pub normal_vector_offset: u64,
    pub normal_header_offset: u64,
    pub descendants: *const u32,
    pub descendants_len: u64,
}

#[repr(C)]
pub struct AhBuildOptions {
    pub n_trees: u32,
    pub split_after: u32,       // 0 = dimensions (src/writer.rs: fit_in_descendant)
    pub tree_seeds: *const u64, // one per tree: `rng.gen()` per root task
    pub cancel: *const c_int,   // polled while a level runs -> Error::BuildCancelled
    pub progress: Option<extern "C" fn(*mut c_void, u32, u64, u64)>,
    pub progress_user: *mut c_void,
    pub max_trees_in_flight: u32,
    pub margin_mode: u32,       // 0 = AH_MARGIN_AUTO
    pub max_host_threads: u32,  // 0 = 8
    pub reserved0: u32,
}

#[repr(C)]
pub struct AhErrorDetail {
    pub status: c_int,
    pub item: u32,
    pub expected: u64,
    pub received: u64,
}

#[repr(C)]
pub struct AhStreamNode {
    pub id: u32,
    pub tree: u32,
    pub kind: u8,
    pub has_normal: u8,
    pub reserved: u16,
    pub left: u32,
    pub right: u32,
    pub count: u32,
    pub depth: u32,
    pub payload_offset: u64,
}

#[repr(C)]
pub struct AhNodeBatch {
    pub kind: u32,
    pub level: u32,
    pub n_nodes: u64,
    pub nodes: *const AhStreamNode,
    pub payload: *const u8,
    pub payload_len: u64,
    pub normal_stride: u64,
    pub normal_vector_offset: u64,
    pub normal_header_offset: u64,
}

#[link(name = "arroy_hip")]
extern "C" {
    fn ah_abi_version() -> c_int;
    fn ah_last_error() -> *const c_char;
    fn ah_last_error_detail(out: *mut AhErrorDetail) -> c_int;
    fn ah_dataset_create(metric: c_int, dims: u32, capacity: u64, device: c_int, out: *mut *mut AhDataset) -> c_int;
    fn ah_dataset_upload_records(
        ds: *mut AhDataset,
        ids: *const u32,
        records: *const *const u8,
        record_len: usize,
        n: usize,
    ) -> c_int;
    fn ah_dataset_set_preprocessed(ds: *mut AhDataset, preprocessed: c_int) -> c_int;
    fn ah_dataset_finalize(ds: *mut AhDataset) -> c_int;
    fn ah_dataset_destroy(ds: *mut AhDataset) -> c_int;
    fn ah_build_forest_stream(
        ds: *mut AhDataset,
        options: *const AhBuildOptions,
        sink: extern "C" fn(*mut c_void, *const AhNodeBatch) -> c_int,
        user: *mut c_void,
        out_roots: *mut u32,
        out_stats: *mut c_void,
    ) -> c_int;
    fn ah_rerank_by_vector(
        ds: *mut AhDataset,
        query: *const f32,
        sorted_ids: *const u32,
        n: usize,
        k: usize,
        out_ids: *mut u32,
        out_dists: *mut f32,
        out_n: *mut usize,
    ) -> c_int;
    fn ah_index_create_from_view(ds: *mut AhDataset, view: *const AhForestView, out: *mut *mut AhIndex) -> c_int;
    fn ah_index_destroy(index: *mut AhIndex) -> c_int;
    fn ah_search_batch(
        index: *mut AhIndex,
        queries: *const f32,
        query_items: *const u32,
        nq: usize,
        count: usize,
        search_k: usize,
        oversampling: usize,
        filter_sorted: *const u32,
        n_filter: usize,
        have_filter: c_int,
        out_ids: *mut u32,
        out_dists: *mut f32,
        out_counts: *mut u32,
    ) -> c_int;
    fn ah_route_items(index: *mut AhIndex, item_ids: *const u32, n: usize, tree_seeds: *const u64, out_leaf: *mut u32) -> c_int;
    fn ah_build_subtrees(
        ds: *mut AhDataset,
        options: *const AhBuildOptions,
        item_ids: *const u32,
        offsets: *const u64,
        out: *mut *mut AhForest,
    ) -> c_int;
    fn ah_forest_view_get(forest: *const AhForest, out: *mut AhForestView) -> c_int;
    fn ah_forest_destroy(forest: *mut AhForest) -> c_int;
    fn ah_preprocess_dot(ds: *mut AhDataset, out_max_norm: *mut f32) -> c_int;
    fn ah_dataset_read_headers(ds: *mut AhDataset, first_row: u64, n: u64, out_headers: *mut c_void) -> c_int;
    fn ah_tuning_set(name: *const c_char, value: i64) -> c_int;
}

// ---- layout self-checks: generated by integration/arroy-hip/tools/gen_layout.py from include/arroy_hip.h ----
const _: () = assert!(size_of::<AhBuildOptions>() == 56 && align_of::<AhBuildOptions>() == 8);
const _: () = assert!(offset_of!(AhBuildOptions, n_trees) == 0);
const _: () = assert!(offset_of!(AhBuildOptions, split_after) == 4);
const _: () = assert!(offset_of!(AhBuildOptions, tree_seeds) == 8);
const _: () = assert!(offset_of!(AhBuildOptions, cancel) == 16);
const _: () = assert!(offset_of!(AhBuildOptions, progress) == 24);
const _: () = assert!(offset_of!(AhBuildOptions, progress_user) == 32);
const _: () = assert!(offset_of!(AhBuildOptions, max_trees_in_flight) == 40);
const _: () = assert!(offset_of!(AhBuildOptions, margin_mode) == 44);
const _: () = assert!(offset_of!(AhBuildOptions, max_host_threads) == 48);
const _: () = assert!(offset_of!(AhBuildOptions, reserved0) == 52);
const _: () = assert!(size_of::<AhErrorDetail>() == 24 && align_of::<AhErrorDetail>() == 8);
const _: () = assert!(offset_of!(AhErrorDetail, status) == 0);
const _: () = assert!(offset_of!(AhErrorDetail, item) == 4);
const _: () = assert!(offset_of!(AhErrorDetail, expected) == 8);
const _: () = assert!(offset_of!(AhErrorDetail, received) == 16);
const _: () = assert!(size_of::<AhStreamNode>() == 40 && align_of::<AhStreamNode>() == 8);
const _: () = assert!(offset_of!(AhStreamNode, id) == 0);
const _: () = assert!(offset_of!(AhStreamNode, tree) == 4);
const _: () = assert!(offset_of!(AhStreamNode, kind) == 8);
const _: () = assert!(offset_of!(AhStreamNode, has_normal) == 9);
const _: () = assert!(offset_of!(AhStreamNode, reserved) == 10);
const _: () = assert!(offset_of!(AhStreamNode, left) == 12);
const _: () = assert!(offset_of!(AhStreamNode, right) == 16);
const _: () = assert!(offset_of!(AhStreamNode, count) == 20);
const _: () = assert!(offset_of!(AhStreamNode, depth) == 24);
const _: () = assert!(offset_of!(AhStreamNode, payload_offset) == 32);
const _: () = assert!(size_of::<AhNodeBatch>() == 64 && align_of::<AhNodeBatch>() == 8);
const _: () = assert!(offset_of!(AhNodeBatch, kind) == 0);
const _: () = assert!(offset_of!(AhNodeBatch, level) == 4);
const _: () = assert!(offset_of!(AhNodeBatch, n_nodes) == 8);
const _: () = assert!(offset_of!(AhNodeBatch, nodes) == 16);
const _: () = assert!(offset_of!(AhNodeBatch, payload) == 24);
const _: () = assert!(offset_of!(AhNodeBatch, payload_len) == 32);
const _: () = assert!(offset_of!(AhNodeBatch, normal_stride) == 40);
const _: () = assert!(offset_of!(AhNodeBatch, normal_vector_offset) == 48);
const _: () = assert!(offset_of!(AhNodeBatch, normal_header_offset) == 56);
const _: () = assert!(size_of::<AhNode>() == 32 && align_of::<AhNode>() == 8);
const _: () = assert!(offset_of!(AhNode, kind) == 0);
const _: () = assert!(offset_of!(AhNode, has_normal) == 1);
const _: () = assert!(offset_of!(AhNode, reserved) == 2);
const _: () = assert!(offset_of!(AhNode, tree) == 4);
const _: () = assert!(offset_of!(AhNode, left) == 8);
const _: () = assert!(offset_of!(AhNode, right) == 12);
const _: () = assert!(offset_of!(AhNode, offset) == 16);
const _: () = assert!(offset_of!(AhNode, count) == 24);
const _: () = assert!(offset_of!(AhNode, depth) == 28);
const _: () = assert!(size_of::<AhForestView>() == 88 && align_of::<AhForestView>() == 8);
const _: () = assert!(offset_of!(AhForestView, n_trees) == 0);
const _: () = assert!(offset_of!(AhForestView, n_nodes) == 8);
const _: () = assert!(offset_of!(AhForestView, roots) == 16);
const _: () = assert!(offset_of!(AhForestView, nodes) == 24);
const _: () = assert!(offset_of!(AhForestView, normals) == 32);
const _: () = assert!(offset_of!(AhForestView, normals_len) == 40);
const _: () = assert!(offset_of!(AhForestView, normal_stride) == 48);
const _: () = assert!(offset_of!(AhForestView, normal_vector_offset) == 56);
const _: () = assert!(offset_of!(AhForestView, normal_header_offset) == 64);
const _: () = assert!(offset_of!(AhForestView, descendants) == 72);
const _: () = assert!(offset_of!(AhForestView, descendants_len) == 80);
// ---- end of the generated block ----

/// `ah_status` -> `arroy::Error` (src/error.rs).  Nothing unwinds across the ABI; the typed variants are rebuilt from
/// `ah_last_error_detail` (thread-local, like `ah_last_error`).
fn check(code: c_int, index: u16) -> Result<()> {
    let detail = || {
        let mut d = MaybeUninit::<AhErrorDetail>::zeroed();
        unsafe {
            ah_last_error_detail(d.as_mut_ptr());
            d.assume_init()
        }
    };
    match code {
        0 => Ok(()),
        1 => {
            let d = detail();
            Err(Error::InvalidVecDimension { expected: d.expected as usize, received: d.received as usize })
        }
        2 => Err(Error::BuildCancelled),
        6 => Err(Error::MissingKey { index, mode: "Item", item: detail().item }),
        // 3 device, 4 out of memory, 5 contract violation, 7 not finalized, 8 DotProduct not preprocessed
        _ => Err(Error::Panic(unsafe { CStr::from_ptr(ah_last_error()) }.to_string_lossy().into_owned())),
    }
}

/// `Distance` -> `ah_metric` by the distance's own name (`Metadata::distance`, src/metadata.rs): no extra trait bound, so
/// the calls fit inside the `impl<D: Distance>` blocks of `Writer` / `Reader` as they are.
fn metric_of<D: Distance>() -> Result<c_int> {
    Ok(match D::name() {
        "euclidean" => 0,
        "manhattan" => 1,
        "cosine" => 2,
        "dot-product" => 3,
        "binary quantized euclidean" => 4,
        "binary quantized manhattan" => 5,
        "binary quantized cosine" => 6,
        other => return Err(Error::Panic(format!("libarroy_hip.so does not implement the distance `{other}`"))),
    })
}

/// Bytes of one stored vector of `D` at `dimensions` (`UnalignedVector` codecs: 4 per dimension, or one bit per dimension in
/// whole 64-bit words).
pub fn vector_len<D: Distance>(dimensions: usize) -> usize {
    match D::name() {
        "binary quantized euclidean" | "binary quantized manhattan" | "binary quantized cosine" => dimensions.div_ceil(64) * 8,
        _ => dimensions * 4,
    }
}

/// The HBM-resident image of `ImmutableLeafs` (an `ah_dataset`).  Immutable once staged; `Sync` like the reference's
/// structure (the library gives every calling thread its own stream and scratch).
pub struct HipLeafs<D> {
    ds: *mut AhDataset,
    index: u16,
    /// bytes of one stored vector (record length - tag - header): 4 x dims, or 8 x ceil(dims / 64) for the 1-bit codecs
    vector_len: usize,
    _marker: PhantomData<D>,
}
unsafe impl<D> Send for HipLeafs<D> {}
unsafe impl<D> Sync for HipLeafs<D> {}

impl<D> Drop for HipLeafs<D> {
    fn drop(&mut self) {
        unsafe { ah_dataset_destroy(self.ds) };
    }
}

/// `ImmutableLeafs::new` for the device: the records `[0u8][header][vector]` are copied out of their LMDB pages (odd
/// offsets, overflow pages: whatever `bytes.as_ptr()` was) into pinned staging buffers and sent to HBM; no pointer is
/// kept after a call returns.  `preprocessed`: DotProduct headers already hold `extra_dim` / `norm` (a `Reader`, or a
/// `Writer` after `pre_process_items`).
pub fn stage_leafs<D: Distance>(
    leafs: &ImmutableLeafs<D>,
    items: &RoaringBitmap,
    dimensions: usize,
    index: u16,
    device: i32,
    preprocessed: bool,
) -> Result<HipLeafs<D>> {
    assert_eq!(unsafe { ah_abi_version() }, AH_ABI_VERSION, "libarroy_hip.so of another ABI version");
    // A `Writer` stages one dataset per build and drops it afterwards: by default the library would give its cached device
    // memory and host blobs back every time the last dataset goes (a fresh 30 GB `hipMalloc` is ~1 s of page scrubbing on the
    // next build).  arroy keeps them for the life of the process unless the operator says otherwise in the environment.
    static KEEP_CACHES: std::sync::Once = std::sync::Once::new();
    KEEP_CACHES.call_once(|| {
        if std::env::var_os("AH_CACHE_KEEP_IDLE").is_none() {
            let _ = unsafe { ah_tuning_set(b"AH_CACHE_KEEP_IDLE\0".as_ptr() as *const c_char, 1) };
        }
    });
    let mut ds = std::ptr::null_mut();
    check(unsafe { ah_dataset_create(metric_of::<D>()?, dimensions as u32, items.len(), device, &mut ds) }, index)?;
    let (ids, ptrs, record_len) = leafs.raw_records(items);
    let vector_len = record_len.saturating_sub(1 + size_of::<D::Header>());
    let staged = HipLeafs { ds, index, vector_len, _marker: PhantomData };
    for (ids, ptrs) in ids.chunks(1 << 16).zip(ptrs.chunks(1 << 16)) {
        // ascending ids: RoaringBitmap order; asynchronous: returns once the records are copied out of the pages
        check(unsafe { ah_dataset_upload_records(ds, ids.as_ptr(), ptrs.as_ptr(), record_len, ids.len()) }, index)?;
    }
    if preprocessed {
        check(unsafe { ah_dataset_set_preprocessed(ds, 1) }, index)?;
    }
    check(unsafe { ah_dataset_finalize(ds) }, index)?;
    Ok(staged)
}

/// What the sink needs while the build runs.
struct SinkState<'a, D: Distance> {
    tmp_nodes: &'a mut TmpNodes<D>,
    node_ids: &'a ConcurrentNodeIds,
    /// stream id (dense from 0, a parent before its children) -> the `ItemId` of the tree node in the database
    global: Vec<ItemId>,
    vector_len: usize,
    error: Option<Error>,
}

const UNSET: ItemId = ItemId::MAX;

impl<D: Distance> SinkState<'_, D> {
    fn global_id(&mut self, stream_id: u32) -> Result<ItemId> {
        let i = stream_id as usize;
        if i >= self.global.len() {
            self.global.resize(i + 1, UNSET);
        }
        if self.global[i] == UNSET {
            self.global[i] = self.node_ids.next()?;
        }
        Ok(self.global[i])
    }

    /// One batch = the split planes of (a piece of) a level, or a run of Descendants nodes; `payload` is the pinned DMA
    /// buffer itself and only valid during the call: everything is encoded out of it right here.
    fn take(&mut self, batch: &AhNodeBatch) -> Result<()> {
        let nodes = unsafe { std::slice::from_raw_parts(batch.nodes, batch.n_nodes as usize) };
        let payload = unsafe { std::slice::from_raw_parts(batch.payload, batch.payload_len as usize) };
        for nd in nodes {
            let id = self.global_id(nd.id)?;
            if nd.kind == AH_NODE_DESCENDANTS {
                let bytes = &payload[nd.payload_offset as usize..][..nd.count as usize * 4];
                // ascending inside a node (the build keeps items in id order, like the reference's bitmaps)
                let ids = bytes.chunks_exact(4).map(|b| u32::from_ne_bytes([b[0], b[1], b[2], b[3]]));
                let bitmap = RoaringBitmap::from_sorted_iter(ids).map_err(|e| Error::Panic(e.to_string()))?;
                self.tmp_nodes.put(id, &Node::Descendants(Descendants { descendants: Cow::Owned(bitmap) }))?;
            } else {
                let normal = if nd.has_normal != 0 {
                    let rec = &payload[nd.payload_offset as usize..][..batch.normal_stride as usize];
                    let header: D::Header =
                        pod_read_unaligned(&rec[batch.normal_header_offset as usize..][..size_of::<D::Header>()]);
                    let vector = UnalignedVector::<D::VectorCodec>::from_bytes(
                        &rec[batch.normal_vector_offset as usize..][..self.vector_len],
                    )
                    .map_err(|e| Error::Panic(format!("{e:?}")))?;
                    Some(Leaf { header, vector })
                } else {
                    None // the random split fallback (`normal: None`)
                };
                let (left, right) = (self.global_id(nd.left)?, self.global_id(nd.right)?);
                self.tmp_nodes.put(id, &Node::SplitPlaneNormal(SplitPlaneNormal { normal, left, right }))?;
            }
        }
        Ok(())
    }
}

extern "C" fn sink_trampoline<D: Distance>(user: *mut c_void, batch: *const AhNodeBatch) -> c_int {
    // No panic may cross the C frames: catch it like the reference catches its workers' (src/writer.rs).
    let state = unsafe { &mut *(user as *mut SinkState<D>) };
    let outcome = std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| state.take(unsafe { &*batch })));
    match outcome {
        Ok(Ok(())) => 0,
        Ok(Err(e)) => {
            state.error = Some(e);
            1
        }
        Err(_) => {
            state.error = Some(Error::Panic("panic in the node sink".to_string()));
            2
        }
    }
}

/// `n_trees` new trees over ALL staged items — the job of the `rayon::scope` + `make_tree_in_file` for the roots that
/// `Writer::build` creates when trees are missing.  Every node goes through `NodeCodec` into `tmp_nodes`, with ids from
/// the shared `ConcurrentNodeIds`, while the device is still building the levels below it; returns the roots.
pub fn build_new_trees<D: Distance, R: Rng>(
    leafs: &HipLeafs<D>,
    rng: &mut R,
    options: &BuildOption,
    n_trees: usize,
    node_ids: &ConcurrentNodeIds,
    tmp_nodes: &mut TmpNodes<D>,
    progress: &AtomicU64,
) -> Result<Vec<ItemId>> {
    let seeds: Vec<u64> = (0..n_trees).map(|_| rng.gen()).collect(); // as `StdRng::from_seed(rng.gen())` per task
    let cancel = AtomicI32::new(0);
    // `SubStep.current` of the reference counts descendants handled: the build bumps it by its `n_trees` roots, level by level
    let progress_state = ProgressState { counter: progress, n_trees: n_trees as u64, reported: AtomicU64::new(0) };
    let opt = AhBuildOptions {
        n_trees: n_trees as u32,
        split_after: options.split_after.unwrap_or(0) as u32,
        tree_seeds: seeds.as_ptr(),
        cancel: cancel.as_ptr() as *const c_int,
        progress: Some(progress_trampoline),
        progress_user: &progress_state as *const ProgressState as *mut c_void,
        max_trees_in_flight: 0,
        margin_mode: 0,
        max_host_threads: 0,
        reserved0: 0,
    };
    let mut state =
        SinkState::<D> { tmp_nodes, node_ids, global: Vec::new(), vector_len: leafs.vector_len, error: None };
    let mut roots = vec![0u32; n_trees];
    let done = AtomicI32::new(0);
    let code = std::thread::scope(|s| {
        // `options.cancel` is a closure: a watcher evaluates it while the device works and raises the flag the
        // library polls between launches
        let watcher = s.spawn(|| {
            while done.load(Ordering::Relaxed) == 0 {
                if (options.cancel)() {
                    cancel.store(1, Ordering::Relaxed);
                    break;
                }
                std::thread::sleep(std::time::Duration::from_micros(500));
            }
        });
        let code = unsafe {
            ah_build_forest_stream(
                leafs.ds,
                &opt,
                sink_trampoline::<D>,
                &mut state as *mut SinkState<D> as *mut c_void,
                roots.as_mut_ptr(),
                std::ptr::null_mut(),
            )
        };
        done.store(1, Ordering::Relaxed);
        let _ = watcher.join();
        code
    });
    if let Some(e) = state.error.take() {
        return Err(e);
    }
    check(code, leafs.index)?;
    progress_state.finish();
    roots.into_iter().map(|r| state.global_id(r)).collect()
}

/// `ah_progress_fn` -> the `SubStep.current` counter of `WriterProgress` (src/writer.rs): the library reports every finished
/// level; a forest of `n_items / split_after` leaves per tree has about 16 of them, so the `n_trees` units of the sub-step
/// are credited a sixteenth per level and the remainder when the build returns.
struct ProgressState<'a> {
    counter: &'a AtomicU64,
    n_trees: u64,
    reported: AtomicU64,
}

impl ProgressState<'_> {
    fn credit(&self, target: u64) {
        let target = target.min(self.n_trees);
        let before = self.reported.fetch_max(target, Ordering::Relaxed);
        if target > before {
            self.counter.fetch_add(target - before, Ordering::Relaxed);
        }
    }

    fn finish(&self) {
        self.credit(self.n_trees);
    }
}

extern "C" fn progress_trampoline(user: *mut c_void, level: u32, _nodes_done: u64, _items_routed: u64) {
    let state = unsafe { &*(user as *const ProgressState) };
    state.credit(state.n_trees * (level as u64 + 1) / 16);
}

/// The distance loop + `median_based_top_k` + `normalized_distance` of `Reader::nns_by_leaf`: `nns` sorted and
/// de-duplicated, `(item, distance)` pairs ordered by `(OrderedFloat(distance), item)` — the reference's bits.
/// Object-safe so that `Reader` can hold the staged items without a new type parameter.
pub trait Rerank {
    fn rerank(&self, query: &[f32], nns: &[ItemId], count: usize) -> Result<Vec<(ItemId, f32)>>;
}

impl<D: Distance> Rerank for HipLeafs<D> {
    fn rerank(&self, query: &[f32], nns: &[ItemId], count: usize) -> Result<Vec<(ItemId, f32)>> {
        let k = count.min(nns.len());
        let (mut ids, mut dists, mut n) = (vec![0u32; k], vec![0f32; k], 0usize);
        check(
            unsafe {
                ah_rerank_by_vector(self.ds, query.as_ptr(), nns.as_ptr(), nns.len(), k, ids.as_mut_ptr(), dists.as_mut_ptr(), &mut n)
            },
            self.index,
        )?;
        Ok(ids.into_iter().zip(dists).take(n).collect())
    }
}

/// The tree nodes of an index in the shape `ah_index_create_from_view` takes (`ah_forest_view`): filled by
/// `Reader::stage_on_gpu` from the nodes it decodes out of LMDB.  Node identity — the tie-break of the reference's
/// `BinaryHeap<(OrderedFloat<f32>, NodeId)>` — is the forest-local index, so the caller pushes the nodes in ascending
/// `NodeId` order (`(mode, item)`, src/node_id.rs) and every tie falls like the reference's.
pub struct ForestImage {
    nodes: Vec<AhNode>,
    roots: Vec<u32>,
    normals: Vec<u8>,
    descendants: Vec<u32>,
    vector_len: usize,
    header_len: usize,
    stride: usize,
}

impl ForestImage {
    /// `vector_len`: bytes of one stored vector; the record of a normal is `[vector][header]`, padded to 4 bytes.
    pub fn new<D: Distance>(vector_len: usize) -> ForestImage {
        let header_len = size_of::<D::Header>();
        let stride = (vector_len + header_len + 3) & !3;
        ForestImage { nodes: Vec::new(), roots: Vec::new(), normals: Vec::new(), descendants: Vec::new(), vector_len, header_len, stride }
    }

    pub fn push_root(&mut self, local: u32) {
        self.roots.push(local);
    }

    /// A `SplitPlaneNormal`; `left` / `right` are forest-local indices (ranks of the children's `NodeId`s).
    pub fn push_split<D: Distance>(&mut self, left: u32, right: u32, normal: Option<&Leaf<D>>) {
        let offset = self.normals.len() as u64;
        if let Some(normal) = normal {
            let bytes = normal.vector.as_bytes();
            debug_assert_eq!(bytes.len(), self.vector_len);
            self.normals.extend_from_slice(bytes);
            self.normals.extend_from_slice(bytemuck::bytes_of(&normal.header));
            self.normals.resize(offset as usize + self.stride, 0);
        }
        self.nodes.push(AhNode {
            kind: AH_NODE_SPLIT,
            has_normal: normal.is_some() as u8,
            reserved: 0,
            tree: 0,
            left,
            right,
            offset,
            count: 0,
            depth: 0,
        });
    }

    /// A `Descendants` node, or an item a split points to directly (`NodeId::item`: a one-id list).
    pub fn push_descendants(&mut self, ids: impl Iterator<Item = ItemId>) {
        let offset = self.descendants.len() as u64;
        self.descendants.extend(ids);
        let count = (self.descendants.len() as u64 - offset) as u32;
        self.nodes.push(AhNode {
            kind: AH_NODE_DESCENDANTS,
            has_normal: 0,
            reserved: 0,
            tree: 0,
            left: 0,
            right: 0,
            offset,
            count,
            depth: 0,
        });
    }

    fn view(&self) -> AhForestView {
        AhForestView {
            n_trees: self.roots.len() as u32,
            n_nodes: self.nodes.len() as u64,
            roots: self.roots.as_ptr(),
            nodes: self.nodes.as_ptr(),
            normals: self.normals.as_ptr(),
            normals_len: self.normals.len() as u64,
            normal_stride: self.stride as u64,
            normal_vector_offset: 0,
            normal_header_offset: self.vector_len as u64,
            descendants: self.descendants.as_ptr(),
            descendants_len: self.descendants.len() as u64,
        }
    }
}

/// Items AND tree nodes in HBM: what `Reader::stage_on_gpu` keeps.  `rerank` serves `nns_by_leaf` as before (the descent stays in
/// Rust); `search_batch` is the whole `nns_by_leaf` of many queries in one call.
pub struct HipSearch<D> {
    leafs: HipLeafs<D>,
    index: *mut AhIndex,
    dimensions: usize,
}
unsafe impl<D> Send for HipSearch<D> {}
unsafe impl<D> Sync for HipSearch<D> {}

impl<D> Drop for HipSearch<D> {
    fn drop(&mut self) {
        unsafe { ah_index_destroy(self.index) }; // before `leafs`: the dataset must outlive the index
    }
}

impl<D: Distance> HipSearch<D> {
    /// Copies `image` to the device of `leafs` (validated there: a forest, ranges inside the blobs); nothing of it is kept.
    pub fn new(leafs: HipLeafs<D>, image: &ForestImage, dimensions: usize) -> Result<HipSearch<D>> {
        let mut index = std::ptr::null_mut();
        check(unsafe { ah_index_create_from_view(leafs.ds, &image.view(), &mut index) }, leafs.index)?;
        Ok(HipSearch { leafs, index, dimensions })
    }

    /// `QueryBuilder::by_vector` for `queries.len() / dimensions` queries at once: `(item, distance)` lists in the reference's
    /// order with the reference's bits.  `search_k` 0 = `count * n_trees`, `oversampling` 0 = `D::DEFAULT_OVERSAMPLING`
    /// (src/reader.rs `nns_by_leaf`); `candidates` = `QueryBuilder::candidates`.
    pub fn search_batch(
        &self,
        queries: &[f32],
        count: usize,
        search_k: usize,
        oversampling: usize,
        candidates: Option<&RoaringBitmap>,
    ) -> Result<Vec<Vec<(ItemId, f32)>>> {
        if self.dimensions == 0 || queries.len() % self.dimensions != 0 {
            return Err(Error::InvalidVecDimension { expected: self.dimensions, received: queries.len() });
        }
        self.search(queries.as_ptr(), std::ptr::null(), queries.len() / self.dimensions, count, search_k, oversampling, candidates)
    }

    /// `QueryBuilder::by_item` for many stored items at once: the query leaves are the items' own records in HBM, header
    /// included (src/reader.rs `by_item` -> `item_leaf` -> `nns_by_leaf`).  An id that is not stored is `Error::MissingKey`.
    pub fn search_batch_items(
        &self,
        items: &[ItemId],
        count: usize,
        search_k: usize,
        oversampling: usize,
        candidates: Option<&RoaringBitmap>,
    ) -> Result<Vec<Vec<(ItemId, f32)>>> {
        self.search(std::ptr::null(), items.as_ptr(), items.len(), count, search_k, oversampling, candidates)
    }

    #[allow(clippy::too_many_arguments)]
    fn search(
        &self,
        queries: *const f32,
        query_items: *const u32,
        nq: usize,
        count: usize,
        search_k: usize,
        oversampling: usize,
        candidates: Option<&RoaringBitmap>,
    ) -> Result<Vec<Vec<(ItemId, f32)>>> {
        let filter: Option<Vec<u32>> = candidates.map(|c| c.iter().collect()); // ascending: RoaringBitmap order
        let (mut ids, mut dists, mut counts) = (vec![0u32; nq * count], vec![0f32; nq * count], vec![0u32; nq]);
        check(
            unsafe {
                ah_search_batch(
                    self.index,
                    queries,
                    query_items,
                    nq,
                    count,
                    search_k,
                    oversampling,
                    filter.as_ref().map_or(std::ptr::null(), |f| f.as_ptr()),
                    filter.as_ref().map_or(0, |f| f.len()),
                    filter.is_some() as c_int,
                    ids.as_mut_ptr(),
                    dists.as_mut_ptr(),
                    counts.as_mut_ptr(),
                )
            },
            self.leafs.index,
        )?;
        Ok((0..nq)
            .map(|q| {
                let n = counts[q] as usize;
                ids[q * count..][..n].iter().copied().zip(dists[q * count..][..n].iter().copied()).collect()
            })
            .collect())
    }

    /// `insert_items_in_descendants_from_frozen_reader` (src/writer.rs) for every tree at once: the Descendants node (a
    /// forest-local index of the image this index was made from) each of `items` lands in, `[tree][item]`.  `tree_seeds`
    /// key the coin thrown at `normal: None` nodes (include/arroy_hip_policy.h).
    pub fn route_items(&self, items: &[ItemId], tree_seeds: &[u64]) -> Result<Vec<u32>> {
        let mut out = vec![0u32; items.len() * tree_seeds.len()];
        check(
            unsafe { ah_route_items(self.index, items.as_ptr(), items.len(), tree_seeds.as_ptr(), out.as_mut_ptr()) },
            self.leafs.index,
        )?;
        Ok(out)
    }
}

impl<D: Distance> Rerank for HipSearch<D> {
    fn rerank(&self, query: &[f32], nns: &[ItemId], count: usize) -> Result<Vec<(ItemId, f32)>> {
        self.leafs.rerank(query, nns, count)
    }
}

/// `incremental_index_large_descendant` (src/writer.rs): `make_tree_in_file` over every Descendants node that outgrew
/// `split_after`, all of them in one call — sub-tree `t` covers the ascending ids `subsets[t]`.  The nodes come back through
/// `visit` in post-order per sub-tree (children before parents, as `TmpNodes::put` receives them), with sub-forest-local
/// indices: `visit(node index, node, view)`; `view.roots[t]` is the root of sub-tree `t`.
pub fn build_subtrees<D: Distance, R: Rng>(
    leafs: &HipLeafs<D>,
    rng: &mut R,
    options: &BuildOption,
    subsets: &[Vec<ItemId>],
    mut visit: impl FnMut(u32, &AhNode, &AhForestView) -> Result<()>,
) -> Result<()> {
    let seeds: Vec<u64> = (0..subsets.len()).map(|_| rng.gen()).collect();
    let mut offsets = Vec::with_capacity(subsets.len() + 1);
    let mut ids = Vec::new();
    offsets.push(0u64);
    for s in subsets {
        ids.extend_from_slice(s);
        offsets.push(ids.len() as u64);
    }
    let opt = AhBuildOptions {
        n_trees: subsets.len() as u32,
        split_after: options.split_after.unwrap_or(0) as u32,
        tree_seeds: seeds.as_ptr(),
        cancel: std::ptr::null(),
        progress: None,
        progress_user: std::ptr::null_mut(),
        max_trees_in_flight: 0,
        margin_mode: 0,
        max_host_threads: 0,
        reserved0: 0,
    };
    let mut forest = std::ptr::null_mut();
    check(unsafe { ah_build_subtrees(leafs.ds, &opt, ids.as_ptr(), offsets.as_ptr(), &mut forest) }, leafs.index)?;
    let mut view = MaybeUninit::<AhForestView>::zeroed();
    let outcome = check(unsafe { ah_forest_view_get(forest, view.as_mut_ptr()) }, leafs.index).and_then(|()| {
        let view = unsafe { view.assume_init() };
        let nodes = unsafe { std::slice::from_raw_parts(view.nodes, view.n_nodes as usize) };
        nodes.iter().enumerate().try_for_each(|(i, nd)| visit(i as u32, nd, &view))
    });
    unsafe { ah_forest_destroy(forest) };
    outcome
}

/// `DotProduct::preprocess` (src/distance/dot_product.rs) on the staged items: the max norm, then `extra_dim` / `norm` of
/// every item, on the device; returns the headers in item order (8 bytes each) for the caller to write back into LMDB.
fn preprocess_dot<D: Distance>(leafs: &HipLeafs<D>, n_items: usize) -> Result<Vec<[f32; 2]>> {
    let mut max_norm = 0f32;
    check(unsafe { ah_preprocess_dot(leafs.ds, &mut max_norm) }, leafs.index)?;
    let mut headers = vec![[0f32; 2]; n_items];
    check(
        unsafe { ah_dataset_read_headers(leafs.ds, 0, n_items as u64, headers.as_mut_ptr() as *mut c_void) },
        leafs.index,
    )?;
    Ok(headers)
}

/// Smallest insertion `Writer::insert_items_in_current_trees` routes on the GPU: below it the reference's own walk — a few
/// margins per item and tree, no staging — is cheaper than mirroring the split planes of every tree in HBM.
pub const ROUTE_MIN_ITEMS: u64 = 4096;

/// The tree nodes reachable from `roots` as a [`ForestImage`], from the frozen view `Writer::build` holds
/// (`ImmutableTrees`, src/parallel.rs): local index = rank of the node's id, so ties between nodes fall as in the reference.
/// Also returns the id of every local node.
pub fn image_of_trees<D: Distance>(
    trees: &ImmutableTrees<D>,
    roots: &[ItemId],
    index: u16,
    dimensions: usize,
) -> Result<(ForestImage, Vec<ItemId>)> {
    let get = |id: ItemId| -> Result<Node<D>> { trees.get(id)?.ok_or_else(|| Error::missing_key(Key::tree(index, id))) };
    let mut reachable: Vec<ItemId> = roots.to_vec();
    let mut next = 0;
    while next < reachable.len() {
        if let Node::SplitPlaneNormal(SplitPlaneNormal { left, right, .. }) = get(reachable[next])? {
            reachable.push(left);
            reachable.push(right);
        }
        next += 1;
    }
    reachable.sort_unstable();
    reachable.dedup();
    let rank = |id: ItemId| {
        reachable.binary_search(&id).map(|i| i as u32).map_err(|_| Error::missing_key(Key::tree(index, id)))
    };
    let mut image = ForestImage::new::<D>(vector_len::<D>(dimensions));
    for id in &reachable {
        match get(*id)? {
            Node::Leaf(_) => unreachable!("a tree node is a Descendants or a SplitPlaneNormal"),
            Node::Descendants(Descendants { descendants }) => image.push_descendants(descendants.iter()),
            Node::SplitPlaneNormal(SplitPlaneNormal { normal, left, right }) => {
                image.push_split::<D>(rank(left)?, rank(right)?, normal.as_ref())
            }
        }
    }
    for root in roots {
        image.push_root(rank(*root)?);
    }
    Ok((image, reachable))
}

/// `Writer::insert_items_in_current_trees` (src/writer.rs) on the GPU: every item of `to_insert` goes down every tree of
/// `roots` (`insert_items_in_descendants_from_frozen_reader`: `D::side` at every split, a coin where `normal` is `None`) in
/// one `ah_route_items` call; returns, like the reference, the Descendants nodes that received items, each as
/// `its stored items | the new ones`.  Only the NEW items are staged (their rows are the only ones the margins read) plus
/// the split planes of the trees.  The coin of a `normal: None` node is keyed by `(seed + root, node, item)` — the
/// reference seeds one sequential `R::seed_from_u64(seed + root)` per tree with the same `seed`.
#[allow(clippy::too_many_arguments)]
pub fn route_into_current_trees<D: Distance, R: Rng>(
    rng: &mut R,
    options: &BuildOption,
    to_insert: &RoaringBitmap,
    roots: &[ItemId],
    leafs: &ImmutableLeafs<D>,
    trees: &ImmutableTrees<D>,
    dimensions: usize,
    index: u16,
    progress: &AtomicU64,
) -> Result<IntMap<ItemId, RoaringBitmap>> {
    options.cancelled()?;
    let staged = stage_leafs(leafs, to_insert, dimensions, index, 0, true)?;
    let (image, node_of_local) = image_of_trees(trees, roots, index, dimensions)?;
    let search = HipSearch::new(staged, &image, dimensions)?;
    let seed = rng.next_u64(); // `repeat_n(rng.next_u64(), roots.len())`
    let seeds: Vec<u64> = roots.iter().map(|root| seed.wrapping_add(*root as u64)).collect();
    let items: Vec<ItemId> = to_insert.iter().collect();
    let landed = search.route_items(&items, &seeds)?; // [tree][item] -> local index of a Descendants node
    let mut descendants = IntMap::<ItemId, RoaringBitmap>::default();
    for tree in 0..roots.len() {
        options.cancelled()?;
        for (i, item) in items.iter().enumerate() {
            let node = node_of_local[landed[tree * items.len() + i] as usize];
            descendants.entry(node).or_default().insert(*item); // ascending: `items` is the bitmap's order
        }
        progress.fetch_add(items.len() as u64, Ordering::Relaxed);
    }
    for (node, bitmap) in descendants.iter_mut() {
        // `descendants.into_owned() | to_insert`
        match trees.get(*node)?.ok_or_else(|| Error::missing_key(Key::tree(index, *node)))? {
            Node::Descendants(Descendants { descendants: stored }) => *bitmap |= stored.as_ref(),
            _ => unreachable!("ah_route_items ends at Descendants nodes"),
        }
    }
    Ok(descendants)
}

/// The `incremental_index_large_descendant` tasks of one `insert_descendants_in_file_and_spawn_tasks` call
/// (src/writer.rs): `large` = the Descendants nodes that no longer fit (`(their id, their items)`).  Only their members are
/// staged; `ah_build_subtrees` builds every sub-tree down to nodes that fit (`make_tree_in_file` + the recursion of the spawned
/// tasks) in one call, and the nodes are appended to `tmp_nodes` children first, as `make_tree_in_file` writes them: the root
/// of a sub-tree keeps the id of the descendant it replaces (`next_id: Some(descendant_id)`), every other node takes the next
/// free id.  The whole subset is resident in HBM, so the reference's `fit_in_memory` rounds have nothing to do here.
#[allow(clippy::too_many_arguments)]
pub fn build_large_descendants<D: Distance, R: Rng>(
    rng: &mut R,
    options: &BuildOption,
    leafs: &ImmutableLeafs<D>,
    node_ids: &ConcurrentNodeIds,
    dimensions: usize,
    index: u16,
    large: Vec<(ItemId, RoaringBitmap)>,
    tmp_nodes: &mut TmpNodes<D>,
    items_progress: &AtomicU64,
) -> Result<()> {
    options.cancelled()?;
    let mut members = RoaringBitmap::new();
    for (_, items) in &large {
        members |= items;
    }
    let staged = stage_leafs(leafs, &members, dimensions, index, 0, true)?;
    let subsets: Vec<Vec<ItemId>> = large.iter().map(|(_, items)| items.iter().collect()).collect();
    let vector_len = staged.vector_len;
    let mut global: Vec<ItemId> = Vec::new(); // sub-forest-local index -> tree node id
    let mut roots_known = false;
    build_subtrees(&staged, rng, options, &subsets, |i, nd, view| {
        if !roots_known {
            global = vec![UNSET; view.n_nodes as usize];
            let roots = unsafe { std::slice::from_raw_parts(view.roots, view.n_trees as usize) };
            for (t, root) in roots.iter().enumerate() {
                global[*root as usize] = large[t].0;
            }
            roots_known = true;
        }
        if global[i as usize] == UNSET {
            global[i as usize] = node_ids.next()?;
        }
        let id = global[i as usize];
        if nd.kind == AH_NODE_DESCENDANTS {
            let ids = unsafe { std::slice::from_raw_parts(view.descendants.add(nd.offset as usize), nd.count as usize) };
            let bitmap =
                RoaringBitmap::from_sorted_iter(ids.iter().copied()).map_err(|e| Error::Panic(e.to_string()))?;
            items_progress.fetch_add(nd.count as u64, Ordering::Relaxed);
            tmp_nodes.put(id, &Node::Descendants(Descendants { descendants: Cow::Owned(bitmap) }))?;
        } else {
            let normal = if nd.has_normal != 0 {
                let rec = unsafe {
                    std::slice::from_raw_parts(view.normals.add(nd.offset as usize), view.normal_stride as usize)
                };
                let header: D::Header =
                    pod_read_unaligned(&rec[view.normal_header_offset as usize..][..size_of::<D::Header>()]);
                let vector =
                    UnalignedVector::<D::VectorCodec>::from_bytes(&rec[view.normal_vector_offset as usize..][..vector_len])
                        .map_err(|e| Error::Panic(format!("{e:?}")))?;
                Some(Leaf { header, vector })
            } else {
                None
            };
            // post-order: both children were visited (and named) before their parent
            let (left, right) = (global[nd.left as usize], global[nd.right as usize]);
            tmp_nodes.put(id, &Node::SplitPlaneNormal(SplitPlaneNormal { normal, left, right }))?;
        }
        Ok(())
    })
}

/// Both passes of `DotProduct::preprocess` (src/distance/dot_product.rs) on the device.  `iter` is the reference's own
/// first-pass iterator over the stored items (ascending keys): the records are staged straight from the pages it walks,
/// `ah_preprocess_dot` computes the max norm and every item's `{extra_dim, norm}` with the reference's arithmetic, and the
/// headers come back in iteration order for the caller's second pass to write with `put_current`.  `Ok(None)`: a vector codec
/// that does not borrow from the page — the caller keeps the CPU passes.
pub fn preprocess_dot_records<'a, D: Distance>(
    iter: impl Iterator<Item = heed::Result<(Key, Node<'a, D>)>>,
) -> Result<Option<Vec<[f32; 2]>>> {
    let (mut ids, mut ptrs, mut vector_len, mut index) = (Vec::new(), Vec::<*const u8>::new(), 0usize, 0u16);
    for result in iter {
        let (key, node) = result?;
        let leaf = match node.leaf() {
            Some(leaf) => leaf,
            None => break,
        };
        match leaf.vector {
            Cow::Borrowed(vector) => {
                let bytes = vector.as_bytes();
                vector_len = bytes.len();
                index = key.index;
                ids.push(key.node.item);
                // the stored record `[tag][header][vector]` this vector was decoded out of (src/node.rs)
                ptrs.push(unsafe { bytes.as_ptr().sub(1 + size_of::<D::Header>()) });
            }
            Cow::Owned(_) => return Ok(None),
        }
    }
    if ids.is_empty() {
        return Ok(Some(Vec::new()));
    }
    assert_eq!(unsafe { ah_abi_version() }, AH_ABI_VERSION, "libarroy_hip.so of another ABI version");
    let mut ds = std::ptr::null_mut();
    check(unsafe { ah_dataset_create(metric_of::<D>()?, (vector_len / 4) as u32, ids.len() as u64, 0, &mut ds) }, index)?;
    let staged = HipLeafs::<D> { ds, index, vector_len, _marker: PhantomData };
    let record_len = 1 + size_of::<D::Header>() + vector_len;
    for (ids, ptrs) in ids.chunks(1 << 16).zip(ptrs.chunks(1 << 16)) {
        check(unsafe { ah_dataset_upload_records(ds, ids.as_ptr(), ptrs.as_ptr(), record_len, ids.len()) }, index)?;
    }
    check(unsafe { ah_dataset_finalize(ds) }, index)?;
    preprocess_dot(&staged, ids.len()).map(Some)
}
