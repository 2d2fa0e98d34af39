"""CPU-side checks of the drop-in boundary: the shared object loads and exports exactly the symbols that
include/arroy_hip.h declares.  No compute calls (there is no GPU in the build container)."""
import ctypes
import os
import re

import numpy as np

from conftest import ROOT


# first row of ah_synth_value(seed 42, AH_SYNTH_NORMAL, dims 8), as bits: pins the generator across compilers / devices
PINNED_NORMAL_BITS = [[3191097216, 1057535632, 3198033120, 3212783840, 3214833192, 3208274208, 1068122128, 1041940608]]
# ... and of AH_SYNTH_CLUSTERED / AH_SYNTH_LOW_RANK (the structured distributions of round 6)
PINNED_CLUSTERED_BITS = [[3187003504, 3205884967, 3212976908, 3216004463, 1074059183, 3218876601, 1073384969, 3200919612]]
PINNED_LOW_RANK_BITS = [[3195146240, 1058129408, 1071143936, 3199393792, 1067211520, 1075205248, 1076122752, 1070983168]]


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "arroy_hip.h")).read()
    return sorted(set(re.findall(r"^AH_API [^;(]*?\b(ah_[a-z_0-9]+)\s*\(", src, re.M)))


def test_library_builds_and_exports_every_declared_symbol():
    from arroy_amd import _lib
    _lib.build()
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 28
    for name in names:
        assert hasattr(L, name), f"{name} declared in include/arroy_hip.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), "ctypes signature table out of sync with the header"


def test_abi_scalars_without_a_gpu():
    from arroy_amd import _lib
    L = _lib.lib()
    assert L.ah_abi_version() == 7
    assert [L.ah_header_size(m) for m in range(7)] == [4, 4, 4, 8, 4, 4, 4]
    assert L.ah_vector_size(2, 768) == 3072
    assert L.ah_vector_size(6, 768) == 96
    assert L.ah_vector_size(6, 65) == 16  # rounds up to whole 64-bit words (binary_quantized.rs:67-69)
    assert L.ah_header_size(99) == 0
    assert _lib.device_count() >= 0
    assert L.ah_last_error() is not None


def test_tunables_without_a_gpu():
    """ah_tuning_set / _get / _reset: the run-time form of the AH_* environment switches (schedule only, never a result)."""
    import pytest

    from arroy_amd import _lib
    assert _lib.tuning_get("AH_ROWS_XCD") == (1, 1) and _lib.tuning_get("AH_SCREEN8")[1] == -1
    with _lib.tuning(AH_ROWS_XCD=0, AH_LAUNCH_MAX_ITEMS=1 << 20):
        assert _lib.tuning_get("AH_ROWS_XCD")[0] == 0 and _lib.tuning_get("AH_LAUNCH_MAX_ITEMS") == (1 << 20, 0xFFFFFFFF)
    assert _lib.tuning_get("AH_ROWS_XCD")[0] == 1
    _lib.tuning_set("AH_DENSE", 1)
    _lib.check(_lib.lib().ah_tuning_reset())
    assert _lib.tuning_get("AH_DENSE")[0] == -1
    with pytest.raises(_lib.ArroyHipError):
        _lib.tuning_set("AH_NO_SUCH_SWITCH", 1)
    # every tunable the design document lists exists under that name, and the other way round
    doc = open(os.path.join(ROOT, "DESIGN.md")).read()
    hdr = open(os.path.join(ROOT, "arroy_amd", "csrc", "common.h")).read()
    names = re.findall(r'X\(\w+, "(AH_[A-Z0-9_]+)"', hdr)
    assert len(names) >= 30
    for name in names:
        assert _lib.tuning_get(name) is not None
        assert name in doc, f"{name} is a tunable but DESIGN.md does not mention it"


def test_errors_are_codes_not_exceptions_across_the_abi():
    import pytest

    from arroy_amd import _lib
    h = ctypes.c_void_p()
    # unknown metric -> AH_ERR_INVALID_ARGUMENT with a message; nothing thrown through C
    st = _lib.lib().ah_dataset_create(42, 8, 10, 0, ctypes.byref(h))
    assert st == 5 and b"metric" in _lib.lib().ah_last_error()
    st = _lib.lib().ah_dataset_create(0, 0, 10, 0, ctypes.byref(h))
    assert st == 1  # InvalidVecDimension
    with pytest.raises(_lib.ArroyHipError):
        _lib.check(st)
    assert _lib.lib().ah_dataset_destroy(None) == 0


def test_policy_header_matches_between_host_compilers():
    """The RNG / synthetic-data policy is plain C shared by host and device code; pin a few values so an
    accidental edit is caught on CPU (the GPU tests compare device output with these functions)."""
    from oracle import oracle as O
    x = O.synth(42, 0, 4, 8)
    assert x.shape == (4, 8) and x.min() >= 0.0 and x.max() < 1.0
    y = O.synth(42, 1, 4, 8)
    assert y.min() >= -1.0 and y.max() < 1.0
    assert (y == x * 2 - 1).all()
    assert not (O.synth(43, 0, 4, 8) == x).all()
    # ~N(0,1) as a sum of twelve uniforms (integers: no libm), and its variant with outlier dimensions (dim % 97 == 13)
    z = O.synth(42, 2, 4000, 100)
    assert abs(float(z.mean())) < 0.01 and abs(float(z.std()) - 1.0) < 0.01 and float(abs(z).max()) < 6.0
    assert (z * 1048576.0 == np.round(z * 1048576.0)).all()  # multiples of 2^-20
    w = O.synth(42, 3, 4000, 100)
    assert (w[:, 13] == z[:, 13] * np.float32(20.0)).all() and (np.delete(w, 13, axis=1) == np.delete(z, 13, axis=1)).all()
    assert O.synth(42, 2, 1, 8).view(np.uint32).tolist() == PINNED_NORMAL_BITS


def test_structured_synthetic_distributions():
    """AH_SYNTH_CLUSTERED / AH_SYNTH_LOW_RANK: the row-wise fills (oracle and library, tables computed once) equal the
    per-component definition `ah_synth_value`; the data has the structure the names promise."""
    from arroy_amd import _lib
    from oracle import oracle as O
    L = O.lib()
    assert O.synth(42, O.SYNTH_CLUSTERED, 1, 8).view(np.uint32).tolist() == PINNED_CLUSTERED_BITS
    assert O.synth(42, O.SYNTH_LOW_RANK, 1, 8).view(np.uint32).tolist() == PINNED_LOW_RANK_BITS
    rng = np.random.default_rng(0)
    n, dims = 30_000, 48
    for dist in (O.SYNTH_CLUSTERED, O.SYNTH_LOW_RANK):
        a = O.synth(9, dist, n, dims, first_item=1000)
        assert _lib.synth_rows_host(9, dist, n, dims, first_item=1000).tobytes() == a.tobytes()
        for _ in range(500):
            i, d = int(rng.integers(0, n)), int(rng.integers(0, dims))
            assert np.float32(L.ao_synth_value(9, 1000 + i, d, dims, dist)).tobytes() == a[i, d].tobytes(), (dist, i, d)
        assert abs(float(a.std()) - 1.0) < 0.1
    c = O.synth(9, O.SYNTH_CLUSTERED, n, dims)
    _u, inverse, counts = np.unique(c, axis=0, return_inverse=True, return_counts=True)
    in_groups = int((counts[inverse] > 1).sum())  # rows that have an exact twin: copies of the centres of the big clusters
    assert counts.max() >= 3 and 0.004 * n < in_groups < n / 61 * 1.3
    r = O.synth(9, O.SYNTH_LOW_RANK, 2000, 96)
    sv = np.linalg.svd(r - r.mean(0), compute_uv=False)
    assert sv[31] > 4 * sv[32]  # 32 factors carry the rows, the rest is the 7 % noise


def test_public_headers_are_plain_c99(tmp_path):
    """The boundary is a C ABI: both public headers must compile as strict C99 with gcc (no C++/HIP needed), and a C
    program using them must link against the shared object."""
    import subprocess
    src = tmp_path / "use.c"
    src.write_text(
        '#include "arroy_hip.h"\n#include "arroy_hip_policy.h"\n#include <stdio.h>\n'
        "int main(void) {\n"
        "  ah_build_options o; ah_forest_view v; ah_build_stats s; ah_error_detail d; (void)o; (void)v; (void)s; (void)d;\n"
        "  uint64_t a, b; ah_choose_two(ah_node_key_root(42), 0, 10, &a, &b);\n"
        "  if (a == b || a >= 10 || b >= 10) return 2;\n"
        "  if (ah_abi_version() != AH_ABI_VERSION) return 3;\n"
        "  if (ah_header_size(AH_DOT_PRODUCT) != 8 || ah_vector_size(AH_BQ_COSINE, 768) != 96) return 4;\n"
        '  printf("%s%d\\n", ah_last_error(), (int)sizeof(ah_node));\n  return 0;\n}\n')
    exe = tmp_path / "use"
    lib_dir = os.path.join(ROOT, "arroy_amd")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           str(src), "-o", str(exe), "-L", lib_dir, "-larroy_hip", f"-Wl,-rpath,{lib_dir}",
                           "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stderr)
    assert out.stdout.strip() == "32"


def test_c_example_compiles_and_links(tmp_path):
    """examples/c_abi_demo.c (build + search + node sink from plain C) must keep compiling as strict C99 and linking;
    without a GPU it stops at `ah_device_count` with exit code 2."""
    import subprocess
    exe = tmp_path / "c_abi_demo"
    lib_dir = os.path.join(ROOT, "arroy_amd")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", "c_abi_demo.c"), "-o", str(exe), "-L", lib_dir, "-larroy_hip",
                           f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert out.returncode in (0, 2), (out.returncode, out.stderr)


def build_c_example(name, tmp_path):
    import subprocess
    exe = tmp_path / name
    lib_dir = os.path.join(ROOT, "arroy_amd")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "examples", name + ".c"), "-o", str(exe), "-L", lib_dir, "-larroy_hip",
                           f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_shim_roundtrip_example_walks_the_reference_database(tmp_path):
    """examples/shim_roundtrip.c: the arroy-side shim in plain C on the reference's own LMDB file.  Without a GPU it still
    maps the file, walks the B-tree and finds the 100 misaligned item records (exit code 2 = "no GPU", after the walk);
    tests/test_gpu_staging.py runs it to the end on the GPU box."""
    import subprocess
    exe = build_c_example("shim_roundtrip", tmp_path)
    out = subprocess.run([str(exe), os.path.join(ROOT, "tests", "golden", "large_v0_6.mdb")], capture_output=True, text=True,
                         timeout=120)
    assert out.returncode in (0, 2), (out.returncode, out.stdout, out.stderr)
    assert "100 items x 30 dims, page size 16384" in out.stdout
