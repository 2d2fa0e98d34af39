"""integration/arroy-hip: the Rust side of the drop-in as FILES (review item 9).  It cannot be compiled in this image (no
cargo / rustc), but it can be kept applicable and in sync with the C header: the patch must `git apply --check` against
the reference, be exactly what its generator produces, and `src/hip.rs` must bind only symbols the header declares, with
struct fields in the header's order."""
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT

REF = "/root/reference"
HERE = os.path.join(ROOT, "integration", "arroy-hip")


def header():
    return open(os.path.join(ROOT, "include", "arroy_hip.h")).read()


def rust():
    return open(os.path.join(HERE, "src", "hip.rs")).read()


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="the reference checkout is not on this machine")
def test_patch_applies_to_the_reference_and_matches_its_generator():
    patch = os.path.join(HERE, "arroy-hip.patch")
    out = subprocess.run([sys.executable, os.path.join(HERE, "make_patch.py"), REF], capture_output=True, text=True, check=True)
    assert out.stdout == open(patch).read(), "arroy-hip.patch is stale: python integration/arroy-hip/make_patch.py > arroy-hip.patch"
    chk = subprocess.run(["git", "apply", "--check", "--verbose", patch], cwd=REF, capture_output=True, text=True)
    assert chk.returncode == 0, chk.stderr
    touched = set(re.findall(r"^\+\+\+ b/(\S+)", open(patch).read(), re.M))
    assert touched == {"Cargo.toml", "src/lib.rs", "src/parallel.rs", "src/writer.rs", "src/reader.rs", "src/distance/dot_product.rs"}


def test_every_public_function_of_hip_rs_has_a_call_site():
    """Round-5 review: `route_items`, `build_subtrees` and `preprocess_dot` were bindings nothing called — an arroy built with
    `--features hip` still routed every update through rayon.  Every `pub fn` of src/hip.rs must be called from the patch
    (`hip::name(`, `.name(`, `::name(`) or from another function of hip.rs that is."""
    src, patch = rust(), open(os.path.join(HERE, "arroy-hip.patch")).read()
    added = "\n".join(l[1:] for l in patch.splitlines() if l.startswith("+") and not l.startswith("+++"))
    names = re.findall(r"^\s*pub fn (\w+)", src, re.M)
    assert len(names) >= 14, names
    called_from_patch = {n for n in names if re.search(r"(?:hip::|::|\.)%s(?:::<[^>]*>)?\(" % n, added)}
    # transitive closure inside hip.rs: the body of a reachable function names the others it calls
    bodies = {n: src.split("fn %s" % n, 1)[1].split("\n}\n", 1)[0] for n in names}
    reach, grew = set(called_from_patch), True
    while grew:
        grew = False
        for n in list(reach):
            for m in names:
                if m not in reach and re.search(r"\b%s(?:::<[^>]*>)?\(" % m, bodies[n]):
                    reach.add(m)
                    grew = True
    missing = sorted(set(names) - reach)
    assert not missing, f"pub fn of hip.rs without a call site in arroy-hip.patch: {missing}"
    for must in ("route_into_current_trees", "build_large_descendants", "preprocess_dot_records", "search_batch_items", "build_new_trees"):
        assert must in called_from_patch, must


def test_rust_bindings_name_only_declared_symbols():
    declared = set(re.findall(r"^AH_API [^;(]*?\b(ah_[a-z_0-9]+)\s*\(", header(), re.M))
    block = rust().split('extern "C" {', 1)[1].split("\n}\n", 1)[0]
    bound = set(re.findall(r"fn (ah_[a-z_0-9]+)\(", block))
    assert bound and bound <= declared, bound - declared
    assert "AH_ABI_VERSION: c_int = " + re.search(r"#define AH_ABI_VERSION (\d+)", header()).group(1) in rust()


def test_layout_self_checks_are_what_the_header_compiles_to():
    """`const _: () = assert!(size_of::<T>() == ..)` / `offset_of!(T, f) == ..` for every #[repr(C)] struct: the block in
    src/hip.rs must be exactly what tools/gen_layout.py prints from the C compiler's view of include/arroy_hip.h (sizes,
    alignments, offsets: a field whose TYPE drifts no longer compiles on the Rust side)."""
    sys.path.insert(0, os.path.join(HERE, "tools"))
    import gen_layout
    block = gen_layout.block()
    assert block.count("const _: () = assert!(") >= 55
    assert block in rust(), "src/hip.rs: layout block is stale: python integration/arroy-hip/tools/gen_layout.py"
    # every #[repr(C)] struct with fields has its checks
    for name in re.findall(r"#\[repr\(C\)\]\n(?:#\[derive[^\n]*\n)?pub struct (\w+) \{\n    pub (?!_p)", rust()):
        assert f"size_of::<{name}>()" in block, name


@pytest.mark.parametrize("c_name,rust_name", [("ah_build_options", "AhBuildOptions"), ("ah_stream_node", "AhStreamNode"),
                                              ("ah_node_batch", "AhNodeBatch"), ("ah_error_detail", "AhErrorDetail"),
                                              ("ah_node", "AhNode"), ("ah_forest_view", "AhForestView")])
def test_repr_c_structs_follow_the_header(c_name, rust_name):
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (c_name, c_name), header(), re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    rs = re.search(r"pub struct %s \{(.*?)\n\}" % rust_name, rust(), re.S).group(1)
    rs_fields = re.findall(r"pub (\w+):", rs)
    # a C declaration `uint32_t left, right;` lists two fields: compare flattened name sequences
    flat = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.split(",")
        first = re.findall(r"(\w+)\s*$", names[0].replace("*", " "))[0]
        flat.append(first)
        flat += [re.sub(r"[\*\s]", "", x) for x in names[1:]]
    assert rs_fields == flat, (rs_fields, flat)
