"""The ONE JSON line bench.py prints, and the full record beside it.

Round 5's line had grown to 27 KB (12 sub-records, 10 seam configurations with per-thread arrays, prose) and the
driver no longer recovered it from the tail of stdout it keeps.  The line is now a fixed, small set of keys — the
headline, its roofline, the CPU baseline, one tuple per other BASELINE config — and never more than `MAX_LINE` bytes;
everything else bench.py measures goes to `bench_full.json` (path in the line's `full_record`).

`compact(full)` is pure (dict in, dict out) so tests/test_bench_line.py can hold it to the size and key contract
without a GPU.
"""
import json
import os

MAX_LINE = 4096

# the contract keys of the measurement section, in the order they are printed
HEAD_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
             "vs_baseline", "dtype", "data", "regions", "value_min", "value_max", "value_device_resident",
             "value_host_boundary_raw", "value_parity_default", "shared_gpu")
CONFIG_KEYS = ("workload", "boundary", "tower_path", "engines_per_gpu", "parallelism", "flop_per_eval",
               "device_resident_evals_s", "parity_default_dtype")
ROOFLINE_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "launch_frac", "avg_launch_ms", "launches",
                 "concurrent_launches", "traffic", "algorithmic_bytes_per_launch", "traffic_ratio", "error")
PER_RANK_KEYS = ("rank", "device", "bus_id", "numa_node", "numa_bound", "host_cpus", "evals_s")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample", "error")


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[:n - 1] + "…"


NULLABLE = ("vs_baseline", "traffic")  # contract keys that are printed also when null


def _pick(d, keys):
    return {k: d[k] for k in keys if k in d and (d[k] is not None or k in NULLABLE)}


def compact(full: dict) -> dict:
    """The printed line's dict from bench.py's full record.  Fixed key set; strings capped; lists are tuples of numbers."""
    out = _pick(full, HEAD_KEYS)
    if "config" in full:
        out["config"] = {k: (_short(v, 120) if isinstance(v, str) else v) for k, v in _pick(full["config"], CONFIG_KEYS).items()}
    if "roofline" in full:
        out["roofline"] = _pick(full["roofline"], ROOFLINE_KEYS)
    if "cpu_baseline" in full:
        cb = _pick(full["cpu_baseline"], CPU_KEYS)
        for k in ("sample", "error"):
            if k in cb:
                cb[k] = _short(cb[k], 160)
        a0 = full["cpu_baseline"].get("a0") or {}
        if "value" in a0:
            cb["a0_ataxx_4x64_1thread"] = a0["value"]
        out["cpu_baseline"] = cb
    # the other single-GPU BASELINE configs: [workload, dtype, evals/s, chip frac of the MFMA peak] (or [.., "error"])
    if "others" in full:
        rows = []
        for o in full["others"]:
            if o.get("weights") and not o.get("headline_weights"):
                continue  # (weight-statistics variants of the headline live in the full record)
            if "value" in o:
                rows.append([o.get("workload"), o.get("dtype"), o["value"], (o.get("roofline") or {}).get("frac")])
            else:
                rows.append([o.get("workload"), o.get("dtype"), None, _short(o.get("error", "error"), 60)])
        out["others"] = rows
        out["others_columns"] = ["workload", "dtype", "evals/s", "frac"]
    # HBM-bound side kernels of the per-layer paths: [workload, kernel, GB/s, frac of 8 TB/s]; any fraction > 1 is an error
    hb = []
    for o in [full] + list(full.get("others", [])):
        for k in o.get("hbm_bound_kernels", []) or []:
            if o.get("dtype") == "f16" or o is full:
                hb.append([o.get("workload", (full.get("config") or {}).get("workload", "")).split(" ")[0], k["kernel"],
                           k.get("achieved_GBps"), k.get("frac_of_hbm_peak")] + (["error"] if k.get("error") else []))
    if hb:
        out["hbm_bound"] = hb[:8]
    # the whole seam (generators -> channel -> executor -> PCIe -> replies): headline config only
    for key in ("seam", "seam_parity", "seam_one_process"):
        s = full.get(key)
        if isinstance(s, dict):
            out[key] = ({"value": s["value"], "dtype": s.get("dtype"), "executor_work_util_max": s.get("executor_work_util_max")}
                        if "value" in s else {"error": _short(s.get("error", s.get("skipped", "?")), 100)})
    # every rank's own rate (a slow rank must be visible next to the aggregate) and where it ran
    if "per_rank" in full:
        out["per_rank"] = [_pick(r, PER_RANK_KEYS) for r in full["per_rank"]][:16]
    if "devices_seen" in full:
        out["devices_seen"] = full["devices_seen"][:16]
    if "full_record" in full:
        out["full_record"] = full["full_record"]
    return out


def render(full: dict) -> str:
    """The line itself.  Should a future key push it past MAX_LINE, optional sections are dropped — never the
    measurement (metric/value/roofline/cpu_baseline)."""
    c = compact(full)
    line = json.dumps(c, separators=(",", ":"))
    for drop in ("hbm_bound", "seam_one_process", "seam_parity", "seam", "others_columns",
                 "others", "per_rank", "devices_seen"):
        if len(line.encode()) <= MAX_LINE:
            break
        c.pop(drop, None)
        c["truncated"] = True
        line = json.dumps(c, separators=(",", ":"))
    assert len(line.encode()) <= MAX_LINE, "bench line over the size contract"
    return line


def emit(full: dict, path: str = None) -> str:
    """Writes the full record to `path` (best effort: a read-only tree must not cost the line) and returns the line."""
    if path:
        try:
            tmp = path + ".tmp"
            with open(tmp, "w") as f:
                json.dump(full, f, indent=1)
            os.replace(tmp, path)
            full = dict(full, full_record=os.path.basename(path))
        except OSError as ex:
            full = dict(full, full_record=f"not written: {ex.strerror}")
    return render(full)
