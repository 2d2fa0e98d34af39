#!/usr/bin/env python3
"""Regenerates arroy-hip.patch: the call-site changes of the `hip` feature as a unified diff against arroy v0.7.0
(/root/reference), made by editing a scratch copy of the five touched files and diffing — so the committed patch always
applies (`git apply --check`, tests/test_integration_patch.py) and never drifts from the edits listed here.

    python integration/arroy-hip/make_patch.py [/path/to/arroy] > integration/arroy-hip/arroy-hip.patch
"""
import difflib
import os
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"

EDITS = {
    "Cargo.toml": [(
        """# Enabling this feature provide a method on the reader that assert its own validity.
assert-reader-validity = []
""",
        """# Enabling this feature provide a method on the reader that assert its own validity.
assert-reader-validity = []

# The hot loops (item staging, the forest build, the candidate re-rank) on an AMD MI355X through libarroy_hip.so
# (`extern "C"`, src/hip.rs).  Link with RUSTFLAGS="-L <dir of libarroy_hip.so>".
hip = []
""")],
    "src/lib.rs": [(
        """mod error;
mod item_iter;
""",
        """mod error;
#[cfg(feature = "hip")]
mod hip;
mod item_iter;
""")],
    "src/parallel.rs": [(
        """    /// Returns the leafs identified by the given ID.
    pub fn get(&self, item_id: ItemId) -> heed::Result<Option<Leaf<'t, D>>> {
        let len = match self.constant_length {
""",
        """    /// The stored records `[tag][header][vector]` of `items`, ascending: their ids, the pointers into the LMDB pages
    /// and their (constant) length — what `hip::stage_leafs` copies into HBM.
    #[cfg(feature = "hip")]
    pub(crate) fn raw_records(&self, items: &RoaringBitmap) -> (Vec<ItemId>, Vec<*const u8>, usize) {
        let mut ids = Vec::with_capacity(items.len() as usize);
        let mut ptrs = Vec::with_capacity(items.len() as usize);
        for item_id in items {
            if let Some(ptr) = self.leafs.get(&item_id) {
                ids.push(item_id);
                ptrs.push(*ptr);
            }
        }
        (ids, ptrs, self.constant_length.unwrap_or(0))
    }

    /// Returns the leafs identified by the given ID.
    pub fn get(&self, item_id: ItemId) -> heed::Result<Option<Leaf<'t, D>>> {
        let len = match self.constant_length {
""")],
    "src/writer.rs": [(
        """        for _ in 0..nb_missing_trees {
            progress.fetch_add(1, Ordering::Relaxed);
            let new_id = concurrent_node_ids.next()?;
            roots.push(new_id);
            descendants.insert(new_id, item_indices.clone());
        }
""",
        """        // With the `hip` feature the missing trees are built on the GPU: the items are staged once from their LMDB
        // pages, every tree node comes back through the streaming sink WHILE the device builds the levels below it and is
        // appended to this thread's `TmpNodes` (drained into LMDB below, with the files of the rayon tasks).
        #[cfg(feature = "hip")]
        let nb_missing_trees = if nb_missing_trees > 0 {
            let staged = crate::hip::stage_leafs(&leafs, &item_indices, self.dimensions, self.index, 0, true)?;
            let tmp_node = files_tls.get_or_try(|| match self.tmpdir.as_ref() {
                Some(path) => TmpNodes::new_in(path).map(RefCell::new),
                None => TmpNodes::new().map(RefCell::new),
            })?;
            let new_roots = crate::hip::build_new_trees(
                &staged,
                rng,
                options,
                nb_missing_trees as usize,
                &concurrent_node_ids,
                &mut tmp_node.borrow_mut(),
                &progress, // `SubStep.current`: credited level by level while the device builds
            )?;
            roots.extend(new_roots);
            0
        } else {
            nb_missing_trees
        };

        for _ in 0..nb_missing_trees {
            progress.fetch_add(1, Ordering::Relaxed);
            let new_id = concurrent_node_ids.next()?;
            roots.push(new_id);
            descendants.insert(new_id, item_indices.clone());
        }
"""), (
        # insert_items_in_current_trees: a big insertion goes down all the existing trees on the GPU
        """        if roots.is_empty() {
            return Ok(IntMap::default());
        }

        let mut descendants = IntMap::<ItemId, RoaringBitmap>::default();
""",
        """        if roots.is_empty() {
            return Ok(IntMap::default());
        }

        // With the `hip` feature a big insertion is routed on the GPU: the new items and the split planes of the existing
        // trees are staged, every item goes down every tree in one call (`D::side` at each split, a keyed coin where the
        // normal is `None`), and the Descendants nodes that received items come back as `stored items | new items`.  A
        // small one keeps the walk below: a few margins per item and tree cost less than mirroring the trees in HBM.
        #[cfg(feature = "hip")]
        if to_insert.len() >= crate::hip::ROUTE_MIN_ITEMS {
            return crate::hip::route_into_current_trees(
                rng,
                options,
                &to_insert,
                roots,
                frozen_reader.leafs,
                frozen_reader.trees,
                self.dimensions,
                self.index,
                &progress,
            );
        }

        let mut descendants = IntMap::<ItemId, RoaringBitmap>::default();
"""), (
        # insert_descendants_in_file_and_spawn_tasks: the large descendants of one call become ONE device call
        """        let mut tmp_node = tmp_node.borrow_mut();

        for (item_id, item_indices) in descendants.into_iter() {
            options.cancelled()?;
            if error_snd.is_full() {
                return Ok(());
            }
            if let Some(nb_descendants_progress) = nb_descendants_progress {
""",
        """        let mut tmp_node = tmp_node.borrow_mut();
        // With the `hip` feature the descendants that outgrew `split_after` are not handed to rayon tasks one by one
        // (`incremental_index_large_descendant`): they are collected and built as sub-trees in ONE device call after the loop.
        #[cfg(feature = "hip")]
        let mut large: Vec<(ItemId, RoaringBitmap)> = Vec::new();

        for (item_id, item_indices) in descendants.into_iter() {
            options.cancelled()?;
            if error_snd.is_full() {
                return Ok(());
            }
            if let Some(nb_descendants_progress) = nb_descendants_progress {
"""), (
        """            } else {
                let tmp_nodes = tmp_nodes.clone();
                let rng = StdRng::from_seed(rng.gen());
                let error_snd = error_snd.clone();
                let nb_items_progress = nb_items_progress.clone();
                scope.spawn(move |s| {
""",
        """            } else {
                #[cfg(feature = "hip")]
                {
                    large.push((item_id, item_indices));
                    continue;
                }
                #[cfg(not(feature = "hip"))]
                {
                let tmp_nodes = tmp_nodes.clone();
                let rng = StdRng::from_seed(rng.gen());
                let error_snd = error_snd.clone();
                let nb_items_progress = nb_items_progress.clone();
                scope.spawn(move |s| {
"""), (
        """                            let _ = error_snd.try_send(Error::Panic(msg.to_string()));
                        }
                    }
                });
            }
        }

        if nb_descendants_progress.is_some() {
""",
        """                            let _ = error_snd.try_send(Error::Panic(msg.to_string()));
                        }
                    }
                });
                }
            }
        }

        #[cfg(feature = "hip")]
        if !large.is_empty() {
            // `make_tree_in_file` over every large descendant, and over whatever is still too large below it, on the device:
            // only the members of these descendants are staged; the nodes land in this thread's `TmpNodes`, children first,
            // the root of every sub-tree under the id of the descendant it replaces.
            crate::hip::build_large_descendants(
                &mut rng,
                options,
                frozen_reader.leafs,
                frozen_reader.concurrent_node_ids,
                self.dimensions,
                self.index,
                large,
                &mut tmp_node,
                &nb_items_progress,
            )?;
        }

        if nb_descendants_progress.is_some() {
""")],
    "src/reader.rs": [(
        """        let mut nns_distances = Vec::with_capacity(nns.len());
        for nn in nns {
""",
        """        // With the `hip` feature and a staged copy of the items (`Reader::stage_on_gpu`) the distance loop, the top-k
        // and `normalized_distance` are one call: `(item, distance)` pairs in the same order, with the same bits.
        #[cfg(feature = "hip")]
        if let Some(staged) = self.hip.as_ref() {
            let query: Vec<f32> = query_leaf.vector.iter().collect();
            return crate::hip::Rerank::rerank(staged.as_ref(), &query, &nns, opt.count);
        }

        let mut nns_distances = Vec::with_capacity(nns.len());
        for nn in nns {
"""), (
        """    version: Version,
    _marker: marker::PhantomData<D>,
}

impl<'t, D: Distance> Reader<'t, D> {
""",
        """    version: Version,
    /// The items and the tree nodes in HBM (`hip` feature): set by `Reader::stage_on_gpu`.
    #[cfg(feature = "hip")]
    hip: Option<Box<crate::hip::HipSearch<D>>>,
    _marker: marker::PhantomData<D>,
}

impl<'t, D: Distance> Reader<'t, D> {
"""), (
        """            items: metadata.items,
            version,
            _marker: marker::PhantomData,
        })
    }
""",
        """            items: metadata.items,
            version,
            #[cfg(feature = "hip")]
            hip: None,
            _marker: marker::PhantomData,
        })
    }

    /// Copies the items AND the tree nodes into the memory of GPU `device` (`hip` feature): from then on `nns_by_vector` /
    /// `nns_by_item` re-rank their candidates there, and `nns_by_vectors_on_gpu` answers whole batches of queries on the
    /// device.  The pointers into the LMDB pages are only used during this call.
    #[cfg(feature = "hip")]
    pub fn stage_on_gpu(&mut self, rtxn: &'t RoTxn, device: i32) -> Result<()>
    where
        D: 't,
    {
        let options = crate::writer::BuildOption::default();
        let leafs =
            crate::parallel::ImmutableLeafs::new(rtxn, &options, self.database, &self.items, self.index)?;
        let staged =
            crate::hip::stage_leafs(&leafs, &self.items, self.dimensions, self.index, device, true)?;
        // Every node reachable from the roots, in ascending `NodeId` order: the forest-local index of a node is its rank, so the
        // ties of the reference's `BinaryHeap<(OrderedFloat<f32>, NodeId)>` fall the same way on the device.
        let mut reachable: Vec<NodeId> = self.roots.iter().map(NodeId::tree).collect();
        let mut next = 0;
        while next < reachable.len() {
            let key = Key::new(self.index, reachable[next]);
            if let GenericReadNode::SplitPlaneNormal(GenericReadSplitPlaneNormal { left, right, .. }) =
                self.database_get(rtxn, &key)?.ok_or(Error::missing_key(key))?
            {
                reachable.push(left);
                reachable.push(right);
            }
            next += 1;
        }
        reachable.sort_unstable();
        // A tree node is reached once (the trees are a forest).  An ITEM that splits point to directly (`Leaf` children, written
        // by older arroy versions, src/reader.rs) may hang under several parents: it stays once per parent — a one-id leaf each,
        // equal `NodeId`s are interchangeable in the reference's heap — because `ah_index_create_from_view` takes forests only.
        reachable.dedup_by(|a, b| a == b && a.mode != crate::NodeMode::Item);
        let mut handed_out = std::collections::BTreeMap::<NodeId, u32>::new();
        let mut rank = |id: &NodeId| -> Result<u32> {
            let first = reachable.partition_point(|x| x < id);
            if first == reachable.len() || reachable[first] != *id {
                return Err(Error::missing_key(Key::new(self.index, *id)));
            }
            if id.mode != crate::NodeMode::Item {
                return Ok(first as u32);
            }
            let nth = handed_out.entry(*id).or_insert(0);
            *nth += 1;
            Ok(first as u32 + *nth - 1)
        };
        let vector_len = crate::hip::vector_len::<D>(self.dimensions);
        let mut image = crate::hip::ForestImage::new::<D>(vector_len);
        for id in &reachable {
            let key = Key::new(self.index, *id);
            match self.database_get(rtxn, &key)?.ok_or(Error::missing_key(key))? {
                GenericReadNode::Leaf(_) => image.push_descendants(std::iter::once(id.item)),
                GenericReadNode::Descendants(Descendants { descendants }) => image.push_descendants(descendants.iter()),
                GenericReadNode::SplitPlaneNormal(GenericReadSplitPlaneNormal { normal, left, right }) => {
                    image.push_split::<D>(rank(&left)?, rank(&right)?, normal.as_ref())
                }
            }
        }
        for root in self.roots.iter() {
            image.push_root(rank(&NodeId::tree(root))?);
        }
        self.hip = Some(Box::new(crate::hip::HipSearch::new(staged, &image, self.dimensions)?));
        Ok(())
    }

    /// `nns_by_vector` for many queries at once, entirely on the GPU (`hip` feature, after `stage_on_gpu`): best-first descent,
    /// candidate collection, sort + dedup, distances, top-`count` and `normalized_distance` of every query in ONE call —
    /// `vectors` holds the queries back to back (`dimensions` floats each).  Same lists, same order, same bits as calling
    /// `nns(count).search_k(..).oversampling(..).candidates(..).by_vector(..)` once per query.
    #[cfg(feature = "hip")]
    pub fn nns_by_vectors_on_gpu(
        &self,
        vectors: &[f32],
        count: usize,
        search_k: Option<NonZeroUsize>,
        oversampling: Option<NonZeroUsize>,
        candidates: Option<&RoaringBitmap>,
    ) -> Result<Vec<Vec<(ItemId, f32)>>> {
        let staged = self.hip.as_ref().ok_or_else(|| Error::Panic("Reader::stage_on_gpu has not been called".to_string()))?;
        staged.search_batch(vectors, count, search_k.map_or(0, NonZeroUsize::get), oversampling.map_or(0, NonZeroUsize::get), candidates)
    }

    /// `nns_by_item` for many items at once, entirely on the GPU (`hip` feature, after `stage_on_gpu`): the query leaves are the
    /// stored items themselves (header included), as `QueryBuilder::by_item` reads them.  An id that is not stored is
    /// `Error::MissingKey` (the reference answers `None` for that one item).
    #[cfg(feature = "hip")]
    pub fn nns_by_items_on_gpu(
        &self,
        items: &[ItemId],
        count: usize,
        search_k: Option<NonZeroUsize>,
        oversampling: Option<NonZeroUsize>,
        candidates: Option<&RoaringBitmap>,
    ) -> Result<Vec<Vec<(ItemId, f32)>>> {
        let staged = self.hip.as_ref().ok_or_else(|| Error::Panic("Reader::stage_on_gpu has not been called".to_string()))?;
        staged.search_batch_items(items, count, search_k.map_or(0, NonZeroUsize::get), oversampling.map_or(0, NonZeroUsize::get), candidates)
    }
""")],
    "src/distance/dot_product.rs": [(
        """        // Step one: compute the norm of each vector and find the maximum norm
        let mut max_norm = 0.0;
""",
        """        // With the `hip` feature both passes' arithmetic runs on the GPU: the records are staged from the pages the first
        // iterator walks, the device computes the max norm and every item's `{extra_dim, norm}` (the same f32 operations in the
        // same order), and the second pass below only writes the headers back.
        #[cfg(feature = "hip")]
        {
            let headers = crate::hip::preprocess_dot_records::<Self>(new_iter(wtxn)?)
                .map_err(|e| heed::Error::Encoding(Box::new(e)))?;
            if let Some(headers) = headers {
                let mut cursor = new_iter(wtxn)?;
                let mut nth = 0;
                while let Some((item_id, node)) = cursor.next().transpose()? {
                    let leaf = match node.leaf() {
                        Some(leaf) => leaf,
                        None => break,
                    };
                    let mut leaf = leaf.into_owned();
                    leaf.header.extra_dim = headers[nth][0];
                    leaf.header.norm = headers[nth][1];
                    nth += 1;
                    // safety: We do not keep a reference to the current value, we own it.
                    unsafe { cursor.put_current(&item_id, &Node::Leaf(leaf))? };
                }
                return Ok(());
            }
        }

        // Step one: compute the norm of each vector and find the maximum norm
        let mut max_norm = 0.0;
""")],
}


def main():
    out = []
    for rel, edits in EDITS.items():
        old = open(os.path.join(REF, rel)).read()
        new = old
        for before, after in edits:
            if new.count(before) != 1:
                sys.exit(f"{rel}: the anchor of an edit occurs {new.count(before)} times in the reference (expected once):\n{before}")
            new = new.replace(before, after)
        out += difflib.unified_diff(old.splitlines(keepends=True), new.splitlines(keepends=True), f"a/{rel}", f"b/{rel}")
    sys.stdout.write("".join(out))


if __name__ == "__main__":
    main()
