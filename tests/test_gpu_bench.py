"""-m gpu: the bench.py contract the driver depends on — one JSON line with the agreed keys — on a 2-layer model so that it
runs in seconds (the full-depth numbers live in profiles/)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_contract_json_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--layers", "2",
                          "--no-cpu-baseline", "--no-two-in-flight"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in r, key
    assert r["n_gpus"] == 1 and r["steps"] == 1 and r["warmup"] == 1 and r["higher_is_better"] is True
    assert r["unit"] == "videos/s" and r["value"] > 0 and r["scaling"] == "weak"
    assert "workload" in r["config"] and r["config"]["DEBUG_num_layers_override"] == 2
    roof = r["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in roof, key
    assert roof["bound"] == "mfma" and 0 < roof["frac"] < 1 and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9
    assert 0 < r["roofline_attention"]["frac"] < 1
    assert "hipGraph" in r["launch_mode"]


def test_bench_two_in_flight_extra_runs_single_threaded():
    """The opt-in serving-style extra (two videos interleaved step by step from ONE host thread, sampler.rcm_sample_iter): it
    must finish and report a positive rate beside — never as — the headline value (round 2's two-thread form hung at 14B sizes)."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--layers", "2",
                          "--no-cpu-baseline", "--no-box-calibration", "--two-in-flight"], cwd=ROOT, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert isinstance(r["two_videos_in_flight_videos_per_s"], float) and r["two_videos_in_flight_videos_per_s"] > 0
    assert r["value"] > 0 and "box" not in r


def test_bench_prompt_to_pixels_extra():
    """The opt-in user-visible extra: umT5 -> sampling -> whole-clip VAE decode, reported beside the headline."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--layers", "2",
                          "--no-cpu-baseline", "--no-box-calibration", "--prompt-to-pixels"], cwd=ROOT, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    p = r["prompt_to_pixels"]
    assert "error" not in p, p
    assert p["finite"] and p["video_shape"] == [1, 3, 81, 480, 832] and 0 < p["seconds_per_video"] < 60
    assert p["umt5_ms"] > 0 and p["sampling_ms"] > 0 and p["vae_decode_ms"] > 0 and r["value"] > 0


def test_bench_self_spawns_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no launcher environment (the driver's N = 1 command form with another N) re-executes
    itself as 2 ranks under torch.distributed.run and still prints ONE JSON line (rank 0's).  On the one-GPU box the ranks
    share the device and talk gloo through host memory (TD_BENCH_BACKEND=gloo: the development rig)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["TD_BENCH_BACKEND"] = "gloo"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                          "--layers", "2", "--no-cpu-baseline", "--no-box-calibration", "--no-replica-leg"], cwd=ROOT,
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["value"] > 0 and r["scaling"] == "strong"
    assert "sp2" in r["config"]["parallelism"]
    # round 5: what the first real multi-GPU record needs in order to check the scaling model
    assert r["rccl"]["world_size"] == 2 and r["rccl"]["backend"] == "gloo" and "segments" in r["rccl"]["graph_mode"], r["rccl"]
    pr = r["per_rank_dit_step_ms"]
    assert len(pr["all"]) == 2 and 0 < pr["min"] <= pr["max"]
    ew = r["exposed_wait"]
    assert ew["waits_probed_per_forward"] == 0 and "exposed_wait_ms_per_dit_step" in ew     # gloo gathers synchronously through the host


def test_bench_emulated_rank_reports_compute_and_wire_terms():
    """--emulate-rank r/N: one rank's work of an N-way sequence split on the one GPU (no communication), replayed as ONE
    hipGraph; the line is marked as an emulation and carries the measured compute term and the modelled wire term."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--emulate-rank", "7/8", "--steps", "1", "--warmup", "1",
                          "--layers", "2", "--no-cpu-baseline", "--no-box-calibration"], cwd=ROOT, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    e = r["emulated_rank"]
    assert r["metric"].startswith("EMULATED") and r["vs_baseline"] is None
    assert e["rank"] == 7 and e["of"] == 8 and e["tokens_per_rank_padded"] == 4096 and e["tokens_of_rank"] == 32760 - 7 * 4096
    assert e["measured_compute_ms_per_dit_step"] > 0 and e["pack_bytes_per_layer"] > 4096 * 1536 * 3
    assert "one hipGraph" in r["launch_mode"], r["launch_mode"]


def test_bench_one_rank_rccl_rig_captures_the_collectives_at_full_token_count():
    """--rccl-one-rank: the sequence-parallel forward over a REAL (one-rank) RCCL communicator at the full 32 760 tokens (two
    layers): every all-gather issued, captured inside ONE hipGraph — a capture that stays open long enough for the process
    group's watchdog to poll in it (graph.quiesce_collective_watchdog) — and replayed for every DiT step of the videos."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--rccl-one-rank", "--steps", "2", "--warmup", "1",
                          "--layers", "2", "--no-cpu-baseline", "--no-box-calibration"], cwd=ROOT, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    o = r["rccl_one_rank"]
    assert r["metric"].startswith("ONE-RANK RCCL") and r["vs_baseline"] is None and o["backend"] == "nccl"
    assert o["whole_graph_error"] is None and "one hipGraph" in o["graph_mode"], o
    assert o["dit_step_ms"] > 0 and "one hipGraph" in r["launch_mode"]


def test_bench_collects_its_traffic_counters_in_the_run():
    """--collect-traffic: `roofline.traffic` from two rocprofv3 --pmc child passes of THIS run instead of the committed summary."""
    import shutil
    if shutil.which("rocprofv3") is None:
        pytest.skip("rocprofv3 not on PATH")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--layers", "2",
                          "--no-cpu-baseline", "--no-box-calibration", "--collect-traffic"], cwd=ROOT, capture_output=True,
                         text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    roof, ra = r["roofline"], r["roofline_attention"]
    assert "collected in this run" in roof.get("traffic_source", ""), (roof.get("traffic_collect_error"), out.stderr[-1500:])
    assert roof["traffic_stale"] is False and roof["traffic"] > roof["algorithmic_bytes"] * 0.5
    assert ra["traffic"] > 0 and 0 < ra["hbm_side_frac_informational"] < 1
