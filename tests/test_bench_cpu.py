"""-m "not gpu": the host side of bench.py that needs no GPU — the BASELINE.json configurations behind `--config`, the contract
keys of the line's static parts, the no-progress watchdog."""
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_every_baseline_configuration_has_a_config_flag():
    """BASELINE.json names five configurations; `bench.py --config C1..C5` selects each by name (C1 is the reference's own CPU
    case: the workload the `cpu_baseline` leg times; `headline` = the configuration `metric` is quoted on)."""
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert len(base["configs"]) == 5 and set(bench.CONFIGS) == {"C1", "C2", "C3", "C4", "C5", "headline"}
    want = {   # (BASELINE.json wording, what the preset must select)
        "C1": (r"plain eager SDPA \+ fp16 linear", dict(attention_type="original", quant_linear=False), "Wan2.1-1.3B", "480p"),
        "C2": (r"SageAttention INT8-QK.*dense attn only", dict(attention_type="sage", quant_linear=False), "Wan2.1-1.3B", "480p"),
        "C3": (r"SageAttention \+ SLA block-sparse", dict(attention_type="sagesla", quant_linear=False), "Wan2.1-1.3B", "480p"),
        "C4": (r"14B 720p.*SageSLA \+ W8A8", dict(attention_type="sagesla", quant_linear=True), "Wan2.1-14B", "720p"),
        "C5": (r"Wan2.2-I2V-A14B 720p.*full SageSLA \+ W8A8", dict(attention_type="sagesla", quant_linear=True), "Wan2.2-A14B", "720p"),
    }
    for i, (name, (words, flags, model, res)) in enumerate(sorted(want.items())):
        assert re.search(words, base["configs"][i]), (name, base["configs"][i])
        preset = bench.CONFIGS[name]
        wl = bench.WORKLOADS[preset["workload"]]
        assert {k: wl[k] for k in flags} == flags and preset["model"] == model and preset["res"] == res, (name, preset, wl)
    assert bench.CONFIGS["C5"].get("two_experts") is True          # both A14B experts resident, the switch inside the timed region
    head = bench.CONFIGS["headline"]
    assert bench.WORKLOADS[head["workload"]]["quant_linear"] and head["model"] == "Wan2.1-1.3B" and head["res"] == "480p"
    assert "Wan2.1-1.3B 480p" in base["metric"]


def test_peaks_are_the_dense_figures_of_the_guide():
    assert bench.HBM_PEAK == 8.0e12 and bench.I8_PEAK == 5.0e15 and bench.F16_PEAK == 2.5e15


def test_no_progress_watchdog_fires_only_without_progress(monkeypatch):
    """a limit on the time WITHOUT a sign of progress (``phase()`` and every timed video re-arm it), not on the run."""
    fired = []
    wd = bench.NoProgressWatchdog()
    monkeypatch.setattr(wd, "_fire", lambda: fired.append(time.perf_counter()))     # (the real one dumps stacks and exits 3)
    wd.start(0.5, rank=0)
    try:
        for _ in range(6):            # 1.2 s of steady progress: longer than the window, never idle for it
            time.sleep(0.2)
            wd.kick()
        assert not fired
        t0 = time.perf_counter()
        time.sleep(1.0)               # silence
        assert len(fired) == 1 and 0.3 < fired[0] - t0 < 0.9
        wd.cancel()
        wd.kick()                     # cancelled: the final JSON line is printed with the watchdog off
        time.sleep(0.7)
        assert len(fired) == 1
    finally:
        wd.cancel()
