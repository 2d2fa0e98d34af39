//! MI355X back end of arroy's hot loops (cargo feature `hip`).
//!
//! Thin `extern "C"` bindings of `libarroy_hip.so` (`include/arroy_hip.h`, ABI v5) plus the three places where arroy
//! hands a whole *loop* to the GPU instead of running it per item:
//!
//! * [`stage_leafs`]  — `ImmutableLeafs::new` (`src/parallel.rs`): the stored item records, straight from their LMDB
//!   pages, into HBM (`ah_dataset_upload_records`);
//! * [`build_new_trees`] — the `rayon::scope` over the root descendants + `make_tree_in_file` (`src/writer.rs`): whole
//!   trees built on the device and handed back node by node WHILE they are built (`ah_build_forest_stream`), encoded
//!   with `NodeCodec` and appended to a `TmpNodes` exactly like the CPU path does;
//! * [`Rerank`] — the distance loop + `median_based_top_k` of `Reader::nns_by_leaf` (`src/reader.rs`).
//!
//! LMDB, roaring, `NodeCodec`, `TmpNodes`, node-id allocation, the RNG and the public API stay as they are.
//! Link with `RUSTFLAGS="-L <dir of libarroy_hip.so>"`; the library needs `libamdhip64` at run time.

use std::borrow::Cow;
use std::ffi::CStr;
use std::marker::PhantomData;
use std::mem::{size_of, MaybeUninit};
use std::os::raw::{c_char, c_int, c_void};
use std::sync::atomic::{AtomicI32, Ordering};

use bytemuck::pod_read_unaligned;
use rand::Rng;
use roaring::RoaringBitmap;

use crate::distance::Distance;
use crate::node::{Descendants, Leaf, Node, SplitPlaneNormal};
use crate::parallel::{ConcurrentNodeIds, ImmutableLeafs, TmpNodes};
use crate::unaligned_vector::UnalignedVector;
use crate::writer::BuildOption;
use crate::{Error, ItemId, Result};

pub const AH_ABI_VERSION: c_int = 6;
const AH_NODE_DESCENDANTS: u8 = 1;

#[repr(C)]
pub struct AhDataset {
    _p: [u8; 0],
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
}

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
) -> Result<Vec<ItemId>> {
    let seeds: Vec<u64> = (0..n_trees).map(|_| rng.gen()).collect(); // as `StdRng::from_seed(rng.gen())` per task
    let cancel = AtomicI32::new(0);
    let opt = AhBuildOptions {
        n_trees: n_trees as u32,
        split_after: options.split_after.unwrap_or(0) as u32,
        tree_seeds: seeds.as_ptr(),
        cancel: cancel.as_ptr() as *const c_int,
        progress: None,
        progress_user: std::ptr::null_mut(),
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
    roots.into_iter().map(|r| state.global_id(r)).collect()
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
