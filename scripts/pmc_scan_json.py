"""profiles/rNN_pmc_scan.json from the FETCH_SIZE / WRITE_SIZE summaries of the scan kernel (scripts/pmc_summary.py output).

gfx950 correction per MI355X_MICROARCH.md (HBM section): FETCH_SIZE (KiB) reports half of a 16 B/lane coalesced read
stream, WRITE_SIZE (KiB) is not halved.  The file is stamped with the hash of the kernel sources it was measured on;
bench.py quotes `roofline.traffic` only while that hash matches."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def mean_of(path, counter):
    for line in open(path):
        f = line.strip().split(",")
        # kernel names contain a comma ("k_distances_f32<2, false>"): the last four fields are counter, n, mean, sum
        if len(f) >= 5 and "k_distances_f32<2, false>" in line and f[-4] == counter:
            return float(f[-2]), int(f[-3])
    raise SystemExit(f"{counter} of k_distances_f32<2, false> not found in {path}")


fetch, n_f = mean_of(sys.argv[1], "FETCH_SIZE")
write, n_w = mean_of(sys.argv[2], "WRITE_SIZE")
print(json.dumps({
    "kernel": "ah::k_distances_f32<2, false>", "dispatches_fetch": n_f, "dispatches_write": n_w,
    "fetch_size_kib_per_launch": fetch, "write_size_kib_per_launch": write,
    "hbm_bytes_per_launch": fetch * 1024 * 2 + write * 1024,
    "correction": "reads = FETCH_SIZE x 1024 x 2 (gfx950: the counter tallies 128-byte requests at 64 bytes), writes = WRITE_SIZE x 1024",
    "source_sha16": bench.scan_source_hash(), "sources": bench.SCAN_KERNEL_SOURCES,
}, indent=1))
