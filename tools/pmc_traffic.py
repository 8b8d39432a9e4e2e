#!/usr/bin/env python3
"""Summarises the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_traffic.sh into profiles/hbm_traffic.json, the table
bench.py reads `roofline.traffic` from.  Per MI355X_MICROARCH.md §HBM: rocprofv3 reports both in KiB; on gfx950
FETCH_SIZE counts half of the bytes of wide (16 B per lane) streaming reads, so it is doubled; WRITE_SIZE is exact for
16-byte streaming stores.  Each record carries the hash of the kernel's source file: bench.py reports the figure only
while that file is unchanged."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402  (WORKLOADS, KERNEL_SOURCE)

KERNEL_MATCH = {"kz_tower_resident_f16": "kz_tower_resident<", "kz_tower_resident_f32": "kz_tower_resident_f32",
                "kz_tower_resident_split": "kz_tower_resident_split", "kz_tower_resident_f16g": "kz_tower_resident_split",
                "kz_board_conv_f16": "kz_board_conv_f16", "kz_board_conv_split16": "kz_board_conv_split16", "kz_conv_igemm_f16": "kz_conv_igemm", "kz_conv_igemm_f32": "kz_conv_igemm",
                "kz_att_tower_f16": "OpsF16", "kz_att_tower_f32": "OpsF32", "kz_att_tower_f32_valu": "kz_att_tower_f32"}


def mean_counter(folder, counter, match):
    files = glob.glob(os.path.join(folder, "**", "*counter_collection.csv"), recursive=True)
    vals = collections.defaultdict(list)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and match in r["Kernel_Name"]:
                vals[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    if not vals:
        raise SystemExit(f"no {counter} rows for a kernel matching '{match}' under {folder}")
    name, v = max(vals.items(), key=lambda kv: len(kv[1]))
    v = v[len(v) // 2:]  # steady state: the second half of the launches
    return name, sum(v) / len(v), len(v)


def main():
    wl, dt, batch, fetch_dir, write_dir = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
    batch = batch or bench.WORKLOADS[wl]["batch"]
    line = None
    for ln in open(fetch_dir + ".log"):
        if ln.startswith("{"):
            line = json.loads(ln)
    if line is None:
        raise SystemExit("no bench line in " + fetch_dir + ".log")
    kernel = line["roofline"]["kernel"]
    match = KERNEL_MATCH[kernel]
    kname, fetch_kib, n = mean_counter(fetch_dir, "FETCH_SIZE", match)
    _, write_kib, _ = mean_counter(write_dir, "WRITE_SIZE", match)
    rec = {"kernel": kernel, "kernel_symbol": kname, "workload": wl, "dtype": dt, "batch": batch,
           "source_file": bench.KERNEL_SOURCE[kernel],
           "source_headers": bench.KERNEL_DEVICE_HEADERS.get(bench.KERNEL_SOURCE[kernel], []),
           "source_sha256_16": bench.kernel_source_hash(kernel),
           "fetch_size_kib_per_launch": round(fetch_kib, 2), "write_size_kib_per_launch": round(write_kib, 2),
           "launches_averaged": n,
           "traffic_bytes_per_launch": round((2 * fetch_kib + write_kib) * 1024, 1),
           "correction": "FETCH_SIZE x2 (gfx950 wide-read correction), WRITE_SIZE as is",
           "profile": f"profiles/hbm_traffic.json <- tools/pmc_traffic.sh {wl} {dt} (separate --pmc passes, 1 engine)"}
    path = os.path.join(REPO, "profiles", "hbm_traffic.json")
    table = json.load(open(path)) if os.path.exists(path) else {"records": []}
    table["records"] = [r for r in table["records"]
                        if not (r["kernel"] == kernel and r["workload"] == wl and r["batch"] == batch)] + [rec]
    json.dump(table, open(path, "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
