#!/usr/bin/env python3
"""tools/calib_counters.py -- the calibration kernels of include/spmv_hip.h (spmv_calib_*) alone, for a counter pass:

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/calib/fetch -- python3 tools/calib_counters.py
    python3 tools/calib_counters.py --parse gpurun_out/calib/fetch      -> bytes tallied per known byte / per line

Dispatches (each behind a marker whose grid carries the id): 1 = 16-byte stream of 1 GiB; 2/3/4 = one 4-byte gather
per distinct 128-byte line of a 2 GiB table (all 16Mi lines once) touching 1 / 2 / 4 places of the line."""
import csv
import glob
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

STREAM_BYTES = 1 << 30
TABLE_LINES = 1 << 24          # 2 GiB


def run():
    import torch
    import __graft_entry__ as ge
    capi = ge.load_package().capi
    dev = torch.device("cuda", 0)
    table = torch.zeros(TABLE_LINES * 32, dtype=torch.float32, device=dev)
    sink = torch.zeros(16, dtype=torch.float32, device=dev)
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = {}
    for ident, name, fn in ((1, "stream16", lambda: capi.calib_stream(table, STREAM_BYTES, sink)),
                            (2, "gather_touch1", lambda: capi.calib_gather(table, TABLE_LINES, TABLE_LINES, 1, sink)),
                            (3, "gather_touch2", lambda: capi.calib_gather(table, TABLE_LINES, TABLE_LINES, 2, sink)),
                            (4, "gather_touch4", lambda: capi.calib_gather(table, TABLE_LINES, TABLE_LINES, 4, sink))):
        capi.calib_marker(ident)
        fn()
        torch.cuda.synchronize()
        ev0.record()
        for _ in range(3):
            fn()
        ev1.record()
        torch.cuda.synchronize()
        out[name] = {"ms": round(ev0.elapsed_time(ev1) / 3, 4)}
    out["stream16"]["GBps"] = round(STREAM_BYTES / out["stream16"]["ms"] / 1e6, 1)
    for k in ("gather_touch1", "gather_touch2", "gather_touch4"):
        out[k]["G_lines_per_s"] = round(TABLE_LINES / out[k]["ms"] / 1e6, 2)
    print(json.dumps(out))


def parse(d):
    f = sorted(glob.glob(str(Path(d) / "**" / "*_counter_collection.csv"), recursive=True), key=lambda p: Path(p).stat().st_mtime)[-1]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Dispatch_Id"]))
    seg, cur = {}, None
    for r in rows:
        if "k_marker" in r["Kernel_Name"]:
            cur = int(r["Grid_Size"]) // int(r["Workgroup_Size"])
            continue
        if cur is not None and "k_calib" in r["Kernel_Name"]:
            seg.setdefault(cur, []).append(float(r["Counter_Value"]))
    name = rows[0]["Counter_Name"]
    res = {"counter": name, "file": f}
    known = {1: ("stream16", STREAM_BYTES, "bytes"), 2: ("gather_touch1", TABLE_LINES, "lines"),
             3: ("gather_touch2", TABLE_LINES, "lines"), 4: ("gather_touch4", TABLE_LINES, "lines")}
    for ident, vals in seg.items():
        nm, n, unit = known[ident]
        mean_kib = sum(vals) / len(vals)
        res[nm] = {"dispatches": len(vals), f"{name}_KiB": round(mean_kib, 1),
                   f"tallied_bytes_per_{'byte' if unit == 'bytes' else 'line'}": round(mean_kib * 1024 / n, 4)}
    print(json.dumps(res))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--parse":
        parse(sys.argv[2])
    else:
        run()
