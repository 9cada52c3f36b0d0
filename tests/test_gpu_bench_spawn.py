"""bench.py's own multi-rank path (the driver launches `bench.py --gpus N`): a bare `python bench.py --gpus 2` starts its two ranks as a
child process tree before touching the GPU, the ranks rendezvous (gloo here: one shared GPU, no RCCL between two ranks on one device),
shard the cfg4 windows by range, gather them, and rank 0 prints ONE JSON line with n_gpus = 2 and the cfg4_strong sub-record.  GPU only."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def drivers_parser(rec, max_scalars=20, max_chars=1000):
    """A local copy of what the driver's parser did to BENCH_r05.json's line: of `config`, `roofline`, `cpu_baseline` it keeps the leading SCALAR entries
    (nested records dropped, strings cut at ~136 characters, ~21 entries / ~1100 characters per record); unknown top-level keys go to `extra_keys`.  The
    record must survive a STRICTER cut (20 entries, 1000 characters) unchanged."""
    out = {}
    for name in ("config", "roofline", "cpu_baseline"):
        kept = {}
        for k, v in (rec.get(name) or {}).items():
            if isinstance(v, (dict, list)):
                continue
            v = v[:136] if isinstance(v, str) else v
            if len(kept) >= max_scalars or len(json.dumps({**kept, k[:40]: v})) > max_chars:
                break
            kept[k[:40]] = v
        out[name] = kept
    return out


def _fits_the_drivers_parser(rec):
    cut = drivers_parser(rec)
    for name in ("config", "roofline", "cpu_baseline"):
        assert cut[name] == (rec.get(name) or {}), (name, sorted(set(rec.get(name) or {}) - set(cut[name])))


def _run(extra, env_extra=None):
    env = dict(os.environ, **(env_extra or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu", "--steps", "1", "--warmup", "0",
           "--no-cpu-baseline", "--no-general-path", "--no-alt-storage"] + extra
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_spawns_its_ranks_and_reports_both_scaling_records():
    rec = _run(["--log2n", "16", "--iters", "50", "--nwin", "8"])
    assert rec["n_gpus"] == 2 and rec["steps"] == 1 and rec["scaling"] == "weak" and rec["unit"] == "signals/s"
    assert rec["collective_ranks"] == 2 and rec["backend"] == "gloo" and rec["rccl_ranks"] == 0
    assert "cfg4_strong" not in rec                        # (the sub-record rides on the judged size only: --log2n 20)
    rec4 = _run(["--workload", "cfg4", "--iters", "50", "--nwin", "8"])
    assert rec4["n_gpus"] == 2 and rec4["scaling"] == "strong" and rec4["detail"]["config"]["windows_per_gpu"] == [4, 4]
    assert rec4["psd_argmax"] == 33 and rec4["unit"] == "windows/s"


@pytest.mark.gpu
def test_bench_sub_records_cannot_cost_the_main_line():
    """(a) a rank whose local part of cfg4_strong raises -> the ranks stay in lock-step, the main line prints
    with cfg4_strong = {"error": ...}; (b) a rank that hangs in a sub-record -> the watchdog ends the run, rank 0 prints the main
    line it already holds; (c) without faults, rank 0 alone drives the devices of all ranks ([0, 0] on this box) through the
    C-ABI's several-device drivers: the single_process_{cfg3,cfg4,cfg5} records."""
    base = ["--log2n", "16", "--iters", "70", "--nwin", "32", "--rehearse-sub-records"]
    rec = _run(base)
    assert rec["n_gpus"] == 2 and rec["value"] > 0
    assert rec["cfg4_strong"]["windows_per_gpu"] == [16, 16] and rec["cfg4_strong"]["psd_argmax"] == 33
    sp = rec["single_process"]
    assert sp["n_devices"] == 2 and sp["devices"] == [0, 0]
    assert sp["single_process_cfg3"]["signals"] == 4 and sp["single_process_cfg3"]["iters_min_max"] == [70, 70]
    assert sp["single_process_cfg4"]["psd_argmax"] == 33 and sp["single_process_cfg4"]["rccl_gather_ranks"] == 0     # (shards share the device: host gather)
    assert sp["single_process_cfg5"]["channels"] == 4 and sp["single_process_cfg5"]["value"] > 0
    # round 5: BASELINE's other configurations and the row-sharded Gram ride on the same line, and everything a reader
    # of the driver's record needs is repeated as SCALARS of config / roofline (its parser drops nested records and unknown top-level keys)
    assert rec["cfg2"]["value"] > 0 and rec["cfg2"]["n_gpus"] == 2 and rec["cfg2"]["roofline"]["kernel"].startswith("admm_small_iter_kernel")
    assert rec["cfg5"]["value"] > 0 and rec["cfg5"]["roofline"]["kernel"].startswith("symv_tile_mfma_ws_kernel")
    rs = rec["cfg3_row_sharded"]
    assert rs["value"] > 0 and rs["iters"] == 70 and rs["scaling"] == "strong" and rs["allreduce_ms_rank0"] > 0
    cfg = rec["config"]
    for k in ("admm_iters_per_sec", "phase_factor_ms", "phase_admm_ms", "cfg2_signals_per_s", "cfg2_iteration_us", "cfg4_windows_per_s", "cfg4_roofline_frac",
              "cfg5_signals_per_s", "cfg5_roofline_frac", "cfg5_iteration_ms", "rowsharded_signals_per_s", "one_process_cfg4_windows_per_s"):
        assert isinstance(cfg[k], (int, float)) and cfg[k] > 0, k
    assert cfg["sub_record_errors"] == 0 and rec["collective_ranks"] == 2 and rec["roofline"]["factor_frac_of_f64_mfma"] > 0
    _fits_the_drivers_parser(rec)
    rec = _run(base + ["--no-single-process"], {"LPVS_BENCH_INJECT": "fail:1"})
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and "error" in rec["cfg4_strong"]
    rec = _run(base + ["--no-single-process", "--sub-timeout", "25"], {"LPVS_BENCH_INJECT": "hang:1"})
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and "watchdog" in rec["sub_records_note"]


@pytest.mark.gpu
def test_single_process_drivers_through_the_library_rccl_binding(monkeypatch):
    """One device, LPVS_MULTI_FORCE_RCCL: the gather of lpvs_windows_estimate_multi_f64 goes through the dlopen'ed RCCL (a one-rank
    communicator) and the record says so; the same coefficients as the plain single-device engine."""
    sys.path.insert(0, ROOT)
    import bench
    import lpvspectral_jl_amd as L
    monkeypatch.setenv("LPVS_MULTI_FORCE_RCCL", "1")
    r = bench.single_process_records(L, [0], iters4=70, nwin4=32, log2n4=12, which=("cfg4",))
    rec = r["single_process_cfg4"]
    assert "error" not in rec, rec
    assert rec["rccl_gather_ranks"] == 1 and rec["psd_argmax"] == 33 and rec["iters_min_max"] == [70, 70]
    monkeypatch.delenv("LPVS_MULTI_FORCE_RCCL")
    r3 = bench.single_process_records(L, [0], iters3=60, log2n3=15, nf3=64, which=("cfg3",))["single_process_cfg3"]
    assert "error" not in r3 and r3["signals"] == 2 and r3["iters_min_max"] == [60, 60], r3


def _run1(extra):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "0", "--no-cpu-baseline"] + extra, env=env,
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_single_rank_lines_of_every_workload():
    """One rank, reduced sizes / iteration counts: every workload prints ONE line with the contract's fields, a roofline record and the
    sub-records the full-size runs carry (two solves in flight through Python threads and through lpvs_lpv_signals_multi_f64)."""
    need = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline"}
    r3 = _run1(["--log2n", "16", "--iters", "60", "--no-general-path", "--no-alt-storage"])
    assert need <= set(r3) and r3["n_gpus"] == 1 and r3["roofline"]["bound"] == "hbm"
    # (at N = 2^16 the inverse is not diagonally dominant enough for the mixed storage: the uniform 6-byte kernel runs)
    assert r3["roofline"]["kernel"].split(" ")[0] in ("admm_iter_mixed_kernel", "symv_tile_split_kernel", "symv_tile_mixed_kernel")
    _fits_the_drivers_parser(r3)
    two = r3["detail"]["config"]["two_solves_in_flight"]
    assert two["value"] > 0 and two["through_lpvs_lpv_signals_multi_f64"]["same_coefficients_as_the_timed_steps"] is True
    r2 = _run1(["--workload", "cfg2", "--iters", "200"])
    assert need <= set(r2) and r2["unit"] == "signals/s" and r2["roofline"]["launch_us"] > 0
    _fits_the_drivers_parser(r2)
    r5 = _run1(["--workload", "cfg5", "--log2n", "17", "--iters", "30", "--channels", "3", "--steps", "1"])
    assert need <= set(r5) and r5["roofline"]["kernel"] == "symv_tile_mfma_ws_kernel" and "4x4x4" in r5["detail"]["roofline"]["kernel"]
    _fits_the_drivers_parser(r5)
    assert len(r5["nnz_per_channel"]) == 3 and r5["factorisation"]["ms"] > 0


def test_bench_refuses_to_spawn_under_a_profiler_preload():
    """(CPU test: the refusal happens before anything touches a GPU.)  The marker variable only has to LOOK like a profiler's."""
    env = dict(os.environ, ROCPROFILER_LPVS_TEST_MARKER="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode != 0 and "profiler preload" in out.stderr


def test_driver_record_fits_the_parser_and_carries_every_configuration():
    """(CPU test.)  bench.driver_record on a line shaped like the default run's: `config` / `roofline` / `cpu_baseline` come out as at most 20 short scalars each
    -- surviving the driver's cut unchanged -- and hold every BASELINE configuration's value and roofline fraction, the like-for-like figures (8-byte M, host
    arrays, two in flight) and both MFMA fractions; the nested records stay on the line under `detail` / their own keys; a failed sub-record is counted and named."""
    sys.path.insert(0, ROOT)
    import bench
    long = "x" * 400
    sub = lambda v, frac, us: {"value": v, "ms_per_step": 10.0, "steps": 3, "roofline": {"frac": frac, "launch_us": us, "kernel": "k (prose " + long + ")", "achieved": 5e3},
                               "iteration_ms_all_channels": 0.7, "cpu_baseline": {"value": 1e-3, "cores": 2}}
    o = {"metric": "m", "value": 14.5, "unit": "signals/s", "admm_iters_per_sec": 38547.37477307728, "rccl_ranks": 0, "collective_ranks": 1,
         "phase_ms": {"basis_ms": 0.3, "gram_ms": 2.1, "reduce_rhs_ms": 0.6, "factor_ms": 13.06, "admm_ms": 51.9, "xcorr_ms": 0.7},
         "factorisation": {"frac": 0.535, "ms": 13.06, "achieved": 42.1}, "gram_general_path": {"frac": 0.82, "achieved": 64.6, "achieved_algorithmic": 102.0, "launch_ms": 689.7, "kernel": "gram_kernel<KRS>"},
         "cfg2": sub(44.98, 0.27, 3.93), "cfg4_strong": sub(4243.0, 0.83, 104.5), "cfg5": sub(3.72, 0.63, 602.8), "single_process": {"single_process_cfg4": {"value": 2320.0, "unit": "windows/s"}},
         "config": {"workload": "cfg3 " + long, "gram": long, "matvec_storage": long, "xupdate_corrections_per_solve": 5, "nibble_refreshes_per_solve": 103,
                    "whole_step_with_8_byte_storage": {"signals_per_s_per_gpu": 8.71}, "from_host_arrays": {"signals_per_s_per_gpu": 14.4}, "two_solves_in_flight": {"value": 17.3}},
         "roofline": {"bound": "hbm", "kernel": "admm_iter_mixed_kernel (" + long + ")", "achieved": 5478.6, "peak": 8000.0, "unit": "GB/s", "frac": 0.6848248398298236,
                      "frac_of_measured_stream_6290_GBps": 0.871, "traffic": 148826630.7, "traffic_source": long, "algorithmic_bytes_per_launch": 140218368.0, "launch_us": 25.59,
                      "launch_us_all_of_admm": 25.94, "launches_per_step": 2000, "share_of_step": 0.742, "same_matvec_with_8_byte_storage": {"frac_of_hbm_peak": 0.81, "launch_us": 41.6},
                      "frac_if_priced_with_36_bit_bytes": 0.7645, "matvec_only_launch_us": 23.9, "note": long},
         "cpu_baseline": {"value": 1.5e-6, "unit": "signals/s", "cores": 2, "kind": "port", "sample": long, "admm_iters_per_sec": 0.0031, "linear_in_N_check": {"a": 1}, "cpus_visible": 2,
                          "threads_used": 2, "achieved_gemv_stream_GBps": 34.9, "bytes_streamed_per_cg_iteration_at_full_size": 1.37e11}}
    bench.driver_record(o)
    _fits_the_drivers_parser(o)
    cfg, roof = o["config"], o["roofline"]
    assert len(cfg) <= 20 and len(roof) <= 20 and len(o["cpu_baseline"]) <= 10
    assert (cfg["cfg4_windows_per_s"], cfg["cfg4_roofline_frac"], cfg["cfg5_signals_per_s"], cfg["cfg5_roofline_frac"], cfg["cfg2_signals_per_s"]) == (4243.0, 0.83, 3.72, 0.63, 44.98)
    assert (cfg["signals_per_s_8_byte_M"], cfg["signals_per_s_host_arrays"], cfg["two_in_flight_signals_per_s"], cfg["admm_iters_per_sec"]) == (8.71, 14.4, 17.3, 38547.4)
    assert cfg["sub_record_errors"] == 0 and "first_error" not in cfg and cfg["workload"].startswith("cfg3 ")
    assert (roof["frac"], roof["kernel"], roof["factor_frac_of_f64_mfma"], roof["gram_mfma_frac_of_f64_peak"], roof["matvec_8_byte_frac"]) == (0.684825, "admm_iter_mixed_kernel", 0.535, 0.82, 0.81)
    assert roof["launch_us"] == 25.59 and roof["launch_us_all_of_admm"] == 25.94 and roof["bytes_per_launch"] == 140218368
    assert o["detail"]["roofline"]["note"] == long and o["detail"]["config"]["two_solves_in_flight"] == {"value": 17.3} and o["cfg5"]["value"] == 3.72
    o2 = {"config": {"workload": "cfg3"}, "roofline": {"frac": 0.5}, "cfg5": {"error": "OutOfMemory: " + long}, "phase_ms": {}}
    bench.driver_record(o2)
    assert o2["config"]["sub_record_errors"] == 1 and o2["config"]["first_error"].startswith("cfg5: OutOfMemory") and len(o2["config"]["first_error"]) <= 120
    _fits_the_drivers_parser(o2)
