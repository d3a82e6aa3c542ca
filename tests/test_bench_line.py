"""bench.py as the driver calls it: the pieces that need no GPU here, the whole line on the GPU box.

The line's roofline is the north-star's quantity (rocprofv3 FETCH_SIZE / WRITE_SIZE bytes over the kernel time), measured
by child passes inside the run; `python3 bench.py --gpus N` starts its own ranks (the reference drives everything from one
process, /root/reference/test/main.cpp:3-7)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def _csv(path, rows):
    head = ('"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id","Kernel_Name",'
            '"Workgroup_Size","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name",'
            '"Counter_Value","Start_Timestamp","End_Timestamp"\n')
    with open(path, "w") as f:
        f.write(head)
        for i, (grid, wg, name, counter, value) in enumerate(rows, 1):
            f.write(f'{i},{i},"Agent 2",1,1,1,{grid},1,"{name}",{wg},0,0,8,0,16,"{counter}",{value},{i * 10},{i * 10 + 5}\n')


def test_counter_file_is_cut_at_the_markers(tmp_path):
    """A per-dispatch counter file is attributed to workloads by the marker dispatches (grid / 64 = id), in dispatch
    order, whatever order the rows are written in; rows of another counter are ignored."""
    import bench
    d = tmp_path / "fetch" / "host"
    d.mkdir(parents=True)
    rows = [(64, 64, "spmv::k_marker()", "FETCH_SIZE", 0.0),
            (1 << 20, 256, "spmv::k_calib_stream(a)", "FETCH_SIZE", 512.0),
            (1 << 20, 256, "spmv::k_calib_stream(a)", "FETCH_SIZE", 512.0),
            (300 * 64, 64, "spmv::k_marker()", "FETCH_SIZE", 0.0),
            (4096, 256, "void spmv::k_tiled_mixed<512, false>(x)", "FETCH_SIZE", 1000.0),
            (4096, 256, "spmv::k_carry_fixup(x)", "FETCH_SIZE", 10.0),
            (4096, 256, "void spmv::k_tiled_mixed<512, false>(x)", "WRITE_SIZE", 77.0),
            (4096, 256, "void spmv::k_tiled_mixed<512, false>(x)", "FETCH_SIZE", 1000.0),
            (4096, 256, "spmv::k_carry_fixup(x)", "FETCH_SIZE", 10.0)]
    _csv(d / "1_counter_collection.csv", rows)
    seg = bench.cut_counter_file(tmp_path / "fetch", "FETCH_SIZE")
    assert seg[1]["kib"] == 1024.0 and seg[1]["dispatches"] == 2
    assert seg[300]["kib"] == 2020.0 and seg[300]["dispatches"] == 4
    assert max(seg[300]["kernels"], key=seg[300]["kernels"].get).startswith("void spmv::k_tiled_mixed")


def test_frac_fields_carry_their_flags():
    """No field named *frac* may pass the copy ceiling without the flag that says why."""
    import bench
    alg = 2_348_810_244
    f = bench.rate_fields("auto_", int(0.78 * alg), alg, 0.3192)           # the round-3 headline: 0.92 algorithmic, 0.72 moved
    assert 0.70 < f["auto_frac_hbm"] < 0.74 and "exceeds_copy_ceiling" not in f
    assert f["auto_effective_frac"] > 0.9 and "effective_exceeds_copy_ceiling" in f
    g = bench.rate_fields("auto_", None, alg, 1.0)
    assert "auto_frac_hbm" not in g and "effective_exceeds_copy_ceiling" not in g
    h = bench.rate_fields("auto_", int(7.9 * alg), alg, 2.7)               # counted above what HBM can deliver: flagged
    assert h["auto_frac_hbm"] > 0.79 and "Infinity Cache" in h["exceeds_copy_ceiling"]


def test_replay_only_under_the_same_plan():
    import bench
    tj = json.loads((ROOT / "profiles" / "traffic.json").read_text())
    key, ent = next(iter(tj.items()))
    b, src = bench.replayed_traffic(key, ent["plan"])
    assert b == ent["hbm_bytes_per_launch"] and "replayed" in src
    b, src = bench.replayed_traffic(key, ent["plan"] + " changed=1")
    assert b is None and "another plan" in src


def _line(cmd, timeout):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, str(ROOT / "bench.py")] + cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0]), p.stderr


@pytest.mark.gpu
def test_headline_traffic_is_measured_in_the_run(built, gpu):
    """N = 1, a small matrix of the headline's laws: the two rocprofv3 child passes run before the first GPU call, the
    line's roofline is counters / time, the calibration kernels give the guide's factors (2.0 for 16-byte streams; a
    missing gather is one 128-byte request tallied like a stream's; stores exact)."""
    import shutil
    if shutil.which("rocprofv3") is None and not Path("/opt/rocm/bin/rocprofv3").exists():
        pytest.skip("no rocprofv3 on this box")
    out, err = _line(["--config", "c2", "--band", "8192", "--no-extras", "--no-cpu-baseline", "--steps", "5", "--warmup", "2"], 900)
    r = out["roofline"]
    assert "measured in this run" in r["traffic_source"], (r["traffic_source"], err[-1500:])
    assert r["traffic"] > 0 and abs(r["frac"] - r["traffic"] / (r["kernel_ms"] * 1e-3) / 8e12) < 2e-3
    assert abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-3
    cal = r["calibration"]
    assert 1.9 < cal["read_factor_16B_stream"] < 2.1
    assert 0.9 < cal["write_factor_4B_per_lane"] < 1.1
    assert cal["one_read_factor_valid"] is True, cal
    # c2 band 8192 through the tiled plan reads 16-bit columns: it moves fewer bytes than the CSR-algorithmic count
    assert 0.5 < r["traffic_over_algorithmic"] < 1.2
    assert out["value"] == pytest.approx(r["traffic"] * out["steps"] / (out["ms_per_step"] * out["steps"] * 1e-3) / 1e9, rel=2e-3)
    assert r["effective_frac"] == pytest.approx(out["config"]["algorithmic_bytes_per_gpu"] / (r["kernel_ms"] * 1e-3) / 8e12, abs=2e-3)


@pytest.mark.gpu
def test_bench_starts_its_own_ranks_gloo(built, gpu):
    """`python3 bench.py --gpus 2` with no launcher: two ranks appear, share the one device (the line says so), y is
    exchanged through gloo; the extra exchange modes of --exchange all run under the watchdog."""
    out, err = _line(["--gpus", "2", "--backend", "gloo", "--no-extras", "--no-cpu-baseline", "--steps", "2", "--warmup", "1",
                      "--rows-per-gpu", str(1 << 20), "--traffic", "off"], 900)
    assert out["n_gpus"] == 2 and "RANKS SHARE DEVICES" in out["config"]["devices"]
    assert out["config"]["launch"] == "bench.py started its own ranks"
    assert out["exchange_only_ms"] is not None
    modes = out["exchange_modes"]
    assert "watchdog" not in modes
    peer = modes["native peer_store (one process, bin/spmv_dist_selftest)"]
    assert "error" not in peer, peer
    assert modes["torch p2p"]["y_bit_identical_to_headline"] is True


@pytest.mark.gpu
def test_bench_starts_its_own_ranks_native(built, gpu):
    """--backend native with fewer devices than ranks: the C++ pipeline with all ranks in one process (peer stores),
    every rank's y bit-identical to the single handle."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("more than one device: the native backend runs one process per GPU over RCCL here")
    out, err = _line(["--gpus", "2", "--backend", "native", "--no-extras", "--no-cpu-baseline", "--steps", "2"], 900)
    assert out["n_gpus"] == 2 and "RANKS SHARE DEVICES" in out["config"]["devices"]
    assert out["rows_differing_from_single_handle"] == 0
