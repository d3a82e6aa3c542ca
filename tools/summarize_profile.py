#!/usr/bin/env python3
"""tools/summarize_profile.py <tag> -- condense gpurun_out/prof_<tag>/ (tools/profile.sh) into profiles/.

Writes  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary, verbatim
        profiles/<tag>_pmc_summary.json   per-kernel means of FETCH_SIZE / WRITE_SIZE (separate passes) and
                                          the calibration on a pure 16-B/lane stream of known size
        profiles/traffic.json             HBM bytes per launch of the dominant kernel, read by bench.py
Correction (MI355X_MICROARCH.md, "HBM"): FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies the
128-B requests of a wide coalesced read at 64 B, i.e. reports exactly half the bytes; WRITE_SIZE is exact.  The
calibration passes re-measure both factors on a pure 16-B/lane stream of known size and the summary applies what THEY
gave.  Access classes (round 4, profiles/r04_calibration.json, tools/calib_counters.py): a 4-byte gather that misses every
cache is ONE request for the whole 128-byte line, tallied 64 bytes like a stream's request (touching both halves of the line
costs no second request) -- so the stream factor is right for the gather kernels (k_panel, k_colsort) as well; what their
counters show above the HBM rate is traffic the 256 MiB Infinity Cache serves, which FETCH_SIZE counts too.
(bench.py measures the same two counters itself since round 4 -- child passes cut at marker dispatches, the calibration
kernels of csrc/kernels_calib.hip in the same passes; this script is the stand-alone route for a single workload.)"""
import csv
import glob
import json
import shutil
import sys
from collections import defaultdict
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def newest(pattern):
    """gpurun_out/ accumulates across calls: take the most recent file that matches."""
    files = sorted(glob.glob(str(pattern)), key=lambda f: Path(f).stat().st_mtime)
    if not files:
        raise SystemExit(f"no file matches {pattern}")
    return files[-1]


def means(path):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def main():
    tag = sys.argv[1]
    src = ROOT / "gpurun_out" / f"prof_{tag}"
    dst = ROOT / "profiles"
    dst.mkdir(exist_ok=True)
    stats = newest(src / "trace" / "*" / "*_kernel_stats.csv")
    shutil.copy(stats, dst / f"{tag}_kernel_stats.csv")
    fetch = means(newest(src / "fetch" / "*" / "*_counter_collection.csv"))
    write = means(newest(src / "write" / "*" / "*_counter_collection.csv"))
    cal_f = means(newest(src / "cal_fetch" / "*" / "*_counter_collection.csv"))
    cal_w = means(newest(src / "cal_write" / "*" / "*_counter_collection.csv"))
    bench = json.loads((src / "bench_trace.json").read_text().strip().splitlines()[-1])
    plan = bench.get("plan", "")
    variant = bench["config"]["variant"]
    if variant == "auto" and plan.startswith("auto -> "):      # what SPMV_AUTO resolved to
        variant = plan.split("auto -> ")[1].split(":")[0]

    # calibration: k_stream<16,false> reads exactly 2^31 B (2^28 col_idx + 2^28 vals) and writes 2^26 B
    ck = next(k for k in cal_f if "k_stream<16, false>" in k)
    known_read, known_write = float(1 << 31), float(1 << 26)
    read_factor = known_read / (cal_f[ck][0] * 1024)
    write_factor = known_write / (cal_w[ck][0] * 1024)

    kernels = {}
    for k in fetch:
        if not k.startswith(("void spmv::", "spmv::")):
            continue
        f_kib, n = fetch[k]
        w_kib = write.get(k, (0.0, 0))[0]
        kernels[k] = {"launches": n, "FETCH_SIZE_KiB": round(f_kib, 1), "WRITE_SIZE_KiB": round(w_kib, 1),
                      "hbm_bytes_per_launch": int(round((read_factor * f_kib + write_factor * w_kib) * 1024))}
    # the plan autotunes over several instantiations of k_adaptive: the timed one has the most launches
    binned = variant == "panel" and "binned" in plan
    hot = (("k_bin_products", "k_bs_products") if binned else ("k_panel(", "k_colsort") if variant == "panel" else
           ("k_wave_bundle",) if variant in ("wave_pipe", "scalar", "wave") else
           ("k_adaptive", "k_tiled16", "k_tiled_mixed", "k_sorted"))
    dom = max((k for k in kernels if any(h in k for h in hot)), key=lambda k: kernels[k]["launches"])
    if variant in ("wave_pipe", "wave"):      # one SpMV = bundles + pieces + combine: the counters of all three
        for k in kernels:
            if k != dom and ("k_wave_pieces" in k or "k_wave_combine" in k):
                for key in ("FETCH_SIZE_KiB", "WRITE_SIZE_KiB", "hbm_bytes_per_launch"):
                    kernels[dom][key] = kernels[dom][key] + kernels[k][key]
        kernels[dom]["note"] = "counters of k_wave_bundle + k_wave_pieces + k_wave_combine: one SpMV"
    if binned:                      # one SpMV = the product launch + the sum launch
        for k in kernels:
            if k != dom and ("k_bin_sums" in k or "k_bs_sums" in k):
                for key in ("FETCH_SIZE_KiB", "WRITE_SIZE_KiB", "hbm_bytes_per_launch"):
                    kernels[dom][key] = kernels[dom][key] + kernels[k][key]
        kernels[dom]["note"] = "counters of the product launch + the sum launch: one SpMV"
    # the panel sweep covers the matrix in several launches of the same kernel ("launches=N" in the plan string):
    # scale the per-launch counters to one SpMV so they compare with the algorithmic bytes of one SpMV
    per_spmv = 1
    if variant == "panel" and not binned and "launches=" in bench.get("plan", ""):
        per_spmv = int(bench["plan"].split("launches=")[1].split()[0])
    if per_spmv > 1:
        for key in ("FETCH_SIZE_KiB", "WRITE_SIZE_KiB", "hbm_bytes_per_launch"):
            kernels[dom][key] = kernels[dom][key] * per_spmv
        kernels[dom]["note"] = f"counters scaled by {per_spmv}: one SpMV = {per_spmv} launches of this kernel"
    summary = {"tag": tag, "command": "python3 bench.py --steps 50 --warmup 5 --no-extras --no-cpu-baseline"
                                      + (" " + " ".join(sys.argv[2:]) if len(sys.argv) > 2 else ""),
               "workload": bench["config"]["workload"], "variant": variant, "plan": plan,
               "calibration": {"kernel": ck, "known_read_bytes": int(known_read), "FETCH_SIZE_KiB": cal_f[ck][0],
                               "read_bytes_per_FETCH_KiB": round(read_factor * 1024, 2),
                               "read_correction_factor": round(read_factor, 4),
                               "known_write_bytes": int(known_write), "WRITE_SIZE_KiB": cal_w[ck][0],
                               "write_correction_factor": round(write_factor, 4)},
               "kernels": kernels, "dominant_kernel": dom,
               "fabric_read_requests_per_nonzero(FETCH_KiB*1024/64/nnz)":
                   round(kernels[dom]["FETCH_SIZE_KiB"] * 1024 / 64 / bench["config"]["nnz_per_gpu"], 4),
               "algorithmic_bytes_per_launch": bench["config"]["algorithmic_bytes_per_gpu"],
               "traffic_over_algorithmic": round(kernels[dom]["hbm_bytes_per_launch"] /
                                                 bench["config"]["algorithmic_bytes_per_gpu"], 4),
               "bench_line_under_profiler": bench}
    (dst / f"{tag}_pmc_summary.json").write_text(json.dumps(summary, indent=1))
    tfile = dst / "traffic.json"
    tj = json.loads(tfile.read_text()) if tfile.exists() else {}
    wname = bench["config"]["workload"].split(":")[0]
    band = 0
    if "band " in bench["config"]["workload"]:
        band = int(bench["config"]["workload"].split("band ")[1].split(")")[0])
    tj[f"{variant}:{wname}:band{band}"] = {
        "hbm_bytes_per_launch": kernels[dom]["hbm_bytes_per_launch"], "kernel": dom, "profile": f"{tag}_pmc_summary.json",
        "plan": plan}      # bench.py replays the figure only while its own plan string equals this one
    tfile.write_text(json.dumps(tj, indent=1))
    print(json.dumps({k: summary[k] for k in ("calibration", "dominant_kernel", "traffic_over_algorithmic")}, indent=1))
    print(open(dst / f"{tag}_kernel_stats.csv").read())


if __name__ == "__main__":
    main()
