#!/usr/bin/env python3
"""Regenerate tests/golden/*.json from the reference's own test assets and test sources.

Run in the build container only (needs /root/reference):  python tests/golden/make_golden.py

Sources (data and expected values only — no reference source code is copied):
  * src/tests/assets/v0_6/large.mdb, smol.mdb — LMDB files parsed WITHOUT liblmdb following the
    page layout described in SURVEY.md Appendix B (16-byte page header, u16 node offsets,
    F_BIGDATA overflow pages).  Item records are `[0u8][bias f32][dims x f32]` (src/node.rs:224-228).
  * expected nearest neighbours transcribed from src/tests/upgrade.rs:58-67 and :116-128.
  * literal vectors of the SIMD unit tests src/spaces/simple_avx.rs:120-133 and
    src/spaces/simple_sse.rs:120-131.
The vectors are stored as hex-encoded little-endian f32 so the JSON round-trips bit-exactly.
  * large_v0_6.mdb — a byte-for-byte copy of the reference's test ASSET src/tests/assets/v0_6/large.mdb (data, not
    source; 180 224 bytes): tests/test_gpu_staging.py maps it and stages the item records from the pointers a reader
    would get from LMDB (16-byte page headers, records at odd offsets, 768-byte+ values on overflow pages), which
    cannot be done on the GPU box from /root/reference (absent there).
"""
import json
import os
import struct
import sys

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def parse_mdb(path):
    data = open(path, "rb").read()
    # meta page 0: page header (16) + magic u32, version u32, address u64, mapsize u64, dbs[2] (48 B each)
    def meta(pg_off):
        off = pg_off + 16
        magic, version = struct.unpack_from("<II", data, off)
        assert magic == 0xBEEFC0DE, hex(magic)
        off += 8 + 8 + 8
        dbs = []
        for _ in range(2):
            pad, flags, depth, branch, leaf, overflow, entries, root = struct.unpack_from("<IHHQQQQQ", data, off)
            dbs.append(dict(pad=pad, depth=depth, entries=entries, root=root))
            off += 48
        last_pgno, txnid = struct.unpack_from("<QQ", data, off)
        return dbs, txnid
    dbs0, _ = meta(0)
    psize = dbs0[0]["pad"]  # page size lives in the free DB's md_pad
    metas = [meta(0), meta(psize)]
    dbs, _ = max(metas, key=lambda m: m[1])
    main = dbs[1]
    entries = []

    def walk(pgno):
        base = pgno * psize
        _pg, _pad, flags, lower, upper = struct.unpack_from("<QHHHH", data, base)
        n = (lower - 16) // 2
        offs = struct.unpack_from("<%dH" % n, data, base + 16)
        for o in offs:
            lo, hi, nflags, ksize = struct.unpack_from("<HHHH", data, base + o)
            key = data[base + o + 8: base + o + 8 + ksize]
            if flags & 0x01:  # branch
                walk(lo | (hi << 16) | (nflags << 32))
            else:
                dsize = lo | (hi << 16)
                if nflags & 0x01:  # F_BIGDATA
                    (ov,) = struct.unpack_from("<Q", data, base + o + 8 + ksize)
                    val = data[ov * psize + 16: ov * psize + 16 + dsize]
                else:
                    val = data[base + o + 8 + ksize: base + o + 8 + ksize + dsize]
                entries.append((key, val))

    walk(main["root"])
    assert len(entries) == main["entries"], (len(entries), main["entries"])
    return entries


def copy_assets():
    import shutil
    shutil.copyfile(os.path.join(REF, "src/tests/assets/v0_6/large.mdb"), os.path.join(HERE, "large_v0_6.mdb"))


def items_of(entries, dims):
    """Keys are [index u16 BE][mode u8][item u32 BE][pad u8] (src/key.rs:56-71); mode 3 = Item."""
    out = []
    for key, val in entries:
        index, mode, item = struct.unpack(">HBI", key[:7])
        if mode != 3:
            continue
        assert val[0] == 0 and len(val) == 1 + 4 + 4 * dims, (len(val), dims)
        out.append((item, val[1:5].hex(), val[5:].hex()))
    out.sort()
    return out


def hexf(xs):
    return struct.pack("<%df" % len(xs), *xs).hex()


def inline_snapshots(src):
    import re
    out = {}
    fns = list(re.finditer(r"^fn (\w+)\(\) \{", src, re.M))
    for i, m in enumerate(fns):
        body = src[m.end(): fns[i + 1].start() if i + 1 < len(fns) else len(src)]
        line0 = src[: m.start()].count("\n") + 1
        dumps = []
        for snap in re.findall(r'insta::assert_snapshot!\(handle, @r#"(.*?)"#\);', body, re.S):
            if "Dumping index" not in snap or snap.count("Dumping index") != 1:
                continue
            trees, items, roots, item_ids = {}, {}, None, None
            for line in snap.splitlines():
                line = line.strip()
                r = re.match(r"Root: Metadata \{ dimensions: (\d+), items: RoaringBitmap<(.*?)>, roots: \[(.*?)\], "
                             r'distance: "(.*?)" \}', line)
                if r:
                    if r.group(2).startswith("["):
                        item_ids = [int(x) for x in r.group(2).strip("[]").split(",") if x.strip()]
                    else:  # "100 values between 0 and 99"
                        item_ids = r.group(2)
                    roots = [int(x) for x in r.group(3).split(",") if x.strip()]
                    continue
                d = re.match(r"Tree (\d+): Descendants\(Descendants \{ descendants: \[(.*?)\] \}\)", line)
                if d:
                    trees[d.group(1)] = {"kind": "D", "descendants": [int(x) for x in d.group(2).split(",") if x.strip()]}
                    continue
                sp = re.match(r"Tree (\d+): SplitPlaneNormal\(SplitPlaneNormal<(\w+)> \{ left: (\d+), right: (\d+), "
                              r"normal: (.*) \}\)$", line)
                if sp:
                    node = {"kind": "S", "left": int(sp.group(3)), "right": int(sp.group(4))}
                    nm = re.match(r'Leaf \{ header: \w+ \{ (\w+): "(.*?)" \}, vector: \[(.*?)\] \}', sp.group(5))
                    if nm:
                        node["bias"] = nm.group(2)
                        node["vector"] = [c.strip() for c in nm.group(3).split(",") if "other" not in c]
                    else:
                        assert sp.group(5).strip() == "None", sp.group(5)
                        node["bias"] = node["vector"] = None
                    trees[sp.group(1)] = node
                    continue
                it = re.match(r"Item (\d+): Leaf\(Leaf \{ header: .*?vector: \[(.*?)\] \}\)", line)
                if it:
                    items[it.group(1)] = [c.strip() for c in it.group(2).split(",") if "other" not in c]
            if roots is None:
                continue
            dumps.append({"roots": roots, "item_ids": item_ids, "trees": trees, "items": items})
        if dumps:
            out[m.group(1)] = {"source": f"src/tests/writer.rs:{line0}", "dumps": dumps}
    return out


def main():
    if not os.path.isdir(REF):
        sys.exit("reference not mounted; fixtures are committed, nothing to do")
    copy_assets()
    large = items_of(parse_mdb(f"{REF}/src/tests/assets/v0_6/large.mdb"), 30)
    smol = items_of(parse_mdb(f"{REF}/src/tests/assets/v0_6/smol.mdb"), 2)
    assert len(large) == 100 and len(smol) == 6
    golden = {
        "_source": "generated by tests/golden/make_golden.py from /root/reference test assets",
        "large_v0_6": {
            "source": "src/tests/assets/v0_6/large.mdb; expectations src/tests/upgrade.rs:116-128",
            "metric": "euclidean", "dims": 30,
            "ids": [i for i, _, _ in large],
            "headers_hex": [h for _, h, _ in large],
            "vectors_hex": [v for _, _, v in large],
            "query": [0.0] * 30, "count": 3, "search_k": 100,
            "expected": [[92, "2.4881108"], [24, "2.5068686"], [78, "2.5809734"]],
            "item0_prefix": ["0.59189945", "0.9953131", "0.7271174", "0.7734485"],
        },
        "smol_v0_6": {
            "source": "src/tests/assets/v0_6/smol.mdb; expectations src/tests/upgrade.rs:58-67",
            "metric": "euclidean", "dims": 2,
            "ids": [i for i, _, _ in smol],
            "headers_hex": [h for _, h, _ in smol],
            "vectors_hex": [v for _, _, v in smol],
            "query": [1.0, 0.0], "count": 3, "search_k": 100,
            "expected": [[1, "0"], [0, "1"], [2, "1"]],
        },
        # src/spaces/simple_avx.rs:120-133 — 70 dims, "simd == scalar" for euclid and dot
        "avx_unit": {
            "source": "src/spaces/simple_avx.rs:120-144",
            "v1_hex": hexf([float(x) for x in list(range(10, 26)) * 4 + list(range(26, 32))]),
            "v2_hex": hexf([float(x) for x in list(range(40, 56)) + list(range(10, 26)) * 3 + list(range(56, 62))]),
        },
    }
    # src/spaces/simple_sse.rs:120-131 — read the literals from the test to avoid transcription slips
    import re
    src = open(f"{REF}/src/spaces/simple_sse.rs").read()
    lits = re.findall(r"vec!\[(.*?)\]", src, re.S)
    v1 = [float(x) for x in re.findall(r"[-+]?\d+\.?\d*", lits[0])]
    v2 = [float(x) for x in re.findall(r"[-+]?\d+\.?\d*", lits[1])]
    golden["sse_unit"] = {"source": "src/spaces/simple_sse.rs:120-142", "v1_hex": hexf(v1), "v2_hex": hexf(v2)}
    # cross-check the avx literals the same way
    src = open(f"{REF}/src/spaces/simple_avx.rs").read()
    lits = re.findall(r"vec!\[(.*?)\]", src, re.S)
    a1 = [float(x) for x in re.findall(r"[-+]?\d+\.?\d*", lits[0])]
    a2 = [float(x) for x in re.findall(r"[-+]?\d+\.?\d*", lits[1])]
    assert hexf(a1) == golden["avx_unit"]["v1_hex"] and hexf(a2) == golden["avx_unit"]["v2_hex"]
    # the 10-tree insta snapshot of write_and_update_lot_of_random_points (first build): tree nodes and items as
    # printed (4 decimals, first 10 components) -- src/tests/writer.rs:296-308
    snap = open(f"{REF}/src/tests/snapshots/arroy__tests__writer__write_and_update_lot_of_random_points.snap").read()
    trees, items = {}, {}
    for line in snap.splitlines():
        m = re.match(r"Tree (\d+): Descendants\(Descendants \{ descendants: \[(.*?)\] \}\)", line)
        if m:
            trees[m.group(1)] = {"kind": "D", "descendants": [int(x) for x in m.group(2).split(",") if x.strip()]}
            continue
        m = re.match(r"Tree (\d+): SplitPlaneNormal\(.*left: (\d+), right: (\d+), normal: Leaf \{ header: "
                     r"NodeHeaderEuclidean \{ bias: \"(.*?)\" \}, vector: \[(.*?)\] \}", line)
        if m:
            comps = [c.strip() for c in m.group(5).split(",") if "other" not in c]
            trees[m.group(1)] = {"kind": "S", "left": int(m.group(2)), "right": int(m.group(3)), "bias": m.group(4),
                                 "vector10": comps}
            continue
        m = re.match(r"Item (\d+): Leaf\(Leaf \{ header: .*?vector: \[(.*?)\] \}\)", line)
        if m:
            items[m.group(1)] = [c.strip() for c in m.group(2).split(",") if "other" not in c]
    roots = [int(x) for x in re.search(r"roots: \[(.*?)\]", snap).group(1).split(",")]
    assert len(items) == 100 and len(roots) == 10
    golden["random_points_10_trees"] = {
        "source": "src/tests/snapshots/arroy__tests__writer__write_and_update_lot_of_random_points.snap "
                  "(src/tests/writer.rs:296-308)",
        "dims": 30, "n_items": 100, "n_trees": 10, "roots": roots, "trees": trees, "items10": items}
    # second dump of the same test (after 50 of the 100 items were overwritten and the index rebuilt incrementally)
    snap2 = open(f"{REF}/src/tests/snapshots/arroy__tests__writer__write_and_update_lot_of_random_points-2.snap").read()
    trees2 = {}
    for line in snap2.splitlines():
        m = re.match(r"Tree (\d+): Descendants\(Descendants \{ descendants: \[(.*?)\] \}\)", line)
        if m:
            trees2[m.group(1)] = {"kind": "D", "descendants": [int(x) for x in m.group(2).split(",") if x.strip()]}
            continue
        m = re.match(r"Tree (\d+): SplitPlaneNormal\(.*left: (\d+), right: (\d+), normal: Leaf \{ header: "
                     r"NodeHeaderEuclidean \{ bias: \"(.*?)\" \}, vector: \[(.*?)\] \}", line)
        if m:
            comps = [c.strip() for c in m.group(5).split(",") if "other" not in c]
            trees2[m.group(1)] = {"kind": "S", "left": int(m.group(2)), "right": int(m.group(3)), "bias": m.group(4),
                                  "vector10": comps}
    roots2 = [int(x) for x in re.search(r"roots: \[(.*?)\]", snap2).group(1).split(",")]
    golden["random_points_10_trees_updated"] = {
        "source": "src/tests/snapshots/arroy__tests__writer__write_and_update_lot_of_random_points-2.snap "
                  "(src/tests/writer.rs:310-320)", "roots": roots2, "trees": trees2}
    # write_and_update_lot_of_random_points_with_little_memory (src/tests/writer.rs:1378-1403): Cosine, 3 dimensions,
    # available_memory(0): both file snapshots, every tree node
    golden["little_memory"] = {"source": "src/tests/writer.rs:1378-1403 + its two .snap files", "dumps": []}
    for suffix in ("", "-2"):
        text = open(f"{REF}/src/tests/snapshots/arroy__tests__writer__write_and_update_lot_of_random_points_with_little_"
                    f"memory{suffix}.snap").read()
        body = "\n".join(line for line in text.splitlines() if not line.startswith(("---", "source:", "expression:")))
        parsed = inline_snapshots('fn little_memory() {\ninsta::assert_snapshot!(handle, @r#"' + body + '"#);\n')
        golden["little_memory"]["dumps"].append(parsed["little_memory"]["dumps"][0])
    # inline insta snapshots of the incremental writer tests (src/tests/writer.rs): for every test function the
    # sequence of database dumps it asserts, parsed into {roots, items, trees}
    golden["writer_inline_snapshots"] = inline_snapshots(open(f"{REF}/src/tests/writer.rs").read())
    with open(os.path.join(HERE, "reference_golden.json"), "w") as f:
        json.dump(golden, f, indent=1)
    print("wrote", os.path.join(HERE, "reference_golden.json"), len(v1), len(a1))


if __name__ == "__main__":
    main()
