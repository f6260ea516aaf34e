#!/usr/bin/env python
"""Headline benchmark: end-to-end videos/sec of the 4-step rCM Wan-DiT denoising loop
(BASELINE.json metric) on synthetic Wan2.1-T2V-1.3B 480p shapes, N MI355X GPUs of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload turbo|c2|c3|original]

A "step" is one whole video: the 4 DiT forwards + sampler updates on latents already resident in
HBM (text encoding / VAE are outside the metric, reference README.md:207).  For N > 1 either launch with
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`` or just run ``python bench.py --gpus N``: without
a launcher's environment (WORLD_SIZE unset) the script re-executes itself under ``torch.distributed.run`` on 127.0.0.1 and
passes rank 0's JSON line through.  ``--emulate-rank r/N`` (one GPU): ONE rank's work of an N-way sequence split with locally
fabricated gathered buffers and no communication — the measured compute term of the scaling table in DESIGN.md §6.

Multi-GPU (one process per GPU, RCCL over xGMI).  BASELINE.json's configs are ONE sample and its north star shards
the DiT forward by SEQUENCE, so for N > 1 the timed region is one video sharded over all N ranks
(turbodiffusion_amd.seqpar: every rank owns a 128-token-aligned slice of the tokens; per self-attention layer one
packed RCCL all-gather of the quantised K / V^T / pooled-K / linear-branch partials, pipelined over head groups; one
all-gather of the head output per step): ``value`` = videos/s of that single video, ``"scaling": "strong"``.  Per layer
every rank must receive 3 B per token-channel of the other ranks' K/V (C1: 151 MB/layer in total = 1.0 ms at N = 2 over
one link, 0.25 ms at N = 8 over seven) beside 3.3 ms / N of compute, so this is the LATENCY mode; the THROUGHPUT mode —
N independent videos, no data-path collective — is measured after the timed region and reported beside it in
``replicas`` (never as ``value``).  ``--sp S`` (S < N) makes the timed region run N/S sequence-parallel groups of S
GPUs; ``--sp 1`` N independent videos.  A collective that fails or never returns fails the run (no masking).  With
``--two-in-flight`` (N = 1) the line also carries ``two_videos_in_flight_videos_per_s``: the same GPU with two independent
videos in flight on two streams, enqueued step by step from one host thread (a serving-style extra; never the headline
``value``; opt-in, see its help).  Rank 0 prints ONE JSON line.

Workloads (BASELINE.json configs):
  turbo    TurboWan2.1-T2V-1.3B-480P as published: SageSLA top-k 0.1 + W8A8 linears + fused norms
           (the configuration the reference's 1.9 s / 0.526 video/s figure is quoted on)  [default]
  c2       configs[1]: dense SageAttention INT8-QK (no sparsity), bf16 linears on td_gemm_bf16 (`c2w8a8`: with W8A8)
  c3       configs[2]: SageSLA top-k 0.1, bf16 linears on td_gemm_bf16
  original dense bf16 attention, bf16 linears (the arithmetic of configs[0] on the GPU)
``--config C2|C3|C4|C5`` selects workload + model + resolution (+ both experts for C5) of a BASELINE.json configuration in one flag.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

_T0 = time.perf_counter()


class NoProgressWatchdog:
    """A multi-rank run that STOPS MAKING PROGRESS (a collective whose peer died, a graph replay that never returns) must end
    with a message and a non-zero status, not sit until someone's outer timeout.  A limit on the time since the last sign of
    progress — ``phase()`` and every timed video re-arm it — not on the whole run: a healthy long run (A14B at 720p on 8
    ranks with two experts and the replica leg) is never cut, and the final JSON line is printed with the watchdog cancelled."""

    def __init__(self):
        self.timer, self.limit, self.rank = None, 0.0, 0

    def start(self, limit_s, rank):
        self.limit, self.rank = float(limit_s), rank
        self.kick()

    def _fire(self):
        print(f"[bench] rank {self.rank}: no progress for {self.limit:.0f} s (TD_BENCH_HARD_TIMEOUT_S) — a collective or a graph "
              f"replay is stuck; Python stacks follow", file=sys.stderr, flush=True)
        import faulthandler
        faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
        os._exit(3)

    def kick(self):
        if self.limit <= 0:
            return
        import threading
        if self.timer is not None:
            self.timer.cancel()
        self.timer = threading.Timer(self.limit, self._fire)
        self.timer.daemon = True
        self.timer.start()

    def cancel(self):
        if self.timer is not None:
            self.timer.cancel()
        self.timer, self.limit = None, 0.0


WATCHDOG = NoProgressWatchdog()


def phase(msg):
    """Progress on stderr (rank 0), stamped with the seconds since start: a slow box and a hang must be
    distinguishable from the log of a timed-out run (the 720p configurations of round 2, profiles/r02_head_note.txt).
    Every rank's no-progress watchdog is re-armed here."""
    WATCHDOG.kick()
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.perf_counter() - _T0:8.1f} s] {msg}", file=sys.stderr, flush=True)

# BASELINE.json's configurations as ONE flag each (`--config C2` etc.; C1 = the reference's eager path on the
# CPU = this line's `cpu_baseline`; on the GPU its arithmetic is `--workload original`).  Explicit flags still override.
CONFIGS = {
    "C1": dict(workload="original", model="Wan2.1-1.3B", res="480p"),
    "C2": dict(workload="c2", model="Wan2.1-1.3B", res="480p"),
    "C3": dict(workload="c3", model="Wan2.1-1.3B", res="480p"),
    "C4": dict(workload="turbo", model="Wan2.1-14B", res="720p"),
    "C5": dict(workload="turbo", model="Wan2.2-A14B", res="720p", two_experts=True),
    "headline": dict(workload="turbo", model="Wan2.1-1.3B", res="480p"),
}
BASELINE_VIDEOS_PER_S = 1.0 / 1.9  # README.md:32,298 — TurboWan2.1-T2V-1.3B-480P, 1x RTX 5090
# published TurboDiffusion latencies (s per video, 1x RTX 5090) for the other model/resolution pairs
PUBLISHED_S = {("Wan2.1-1.3B", "480p"): 1.9, ("Wan2.1-14B", "480p"): 9.9, ("Wan2.1-14B", "720p"): 24.0,
               ("Wan2.2-A14B", "720p"): 38.0}
HBM_PEAK = 8.0e12                  # MI355X_MICROARCH.md: HBM3E 8 TB/s
I8_PEAK = 5.0e15                   # dense INT8 MFMA (= dense FP8 rate), MI355X_MICROARCH.md
F16_PEAK = 2.5e15                  # dense FP16/BF16 MFMA
RES = {"480p": (832, 480), "720p": (1280, 720)}
# the HBM-bound family of a block (round 6, `roofline_hbm`): C-ABI entry point -> (what, algorithmic bytes per launch in units of
# L x dim bytes [+ per-row bytes]) — DESIGN.md 3: every tensor the operator must read or write once, nothing for re-reads
HBM_FAMILY = {
    "td_layernorm_quant_stats": ("LayerNorm apply (+ AdaLN modulate) -> INT8 codes + block scales (a5-a7 -> a16); row statistics from the GEMM epilogue", 3.0, 8.0),
    "td_qk_norm_rope": ("RMSNorm(dim) + RoPE + head-major relayout of q or k (a3 / a8)", 4.0, 512.0),
    "td_qk_norm_rope_pair": ("the same for q and k in one launch", 8.0, 512.0),
    "td_sage_quant_pool": ("Sage per-block INT8 quantiser (+ smooth-K) + block mean pooling of q or k (a11 / a13)", 3.0, 0.0),
    "td_sla_linear_kv": ("linear branch, pass over K and V^T: sum phi(k)^T v, sum phi(k) (+ the smooth-K mean) (a14)", 4.0, 0.0),
    "td_sla_linear_out_t": ("linear branch, pass over Q: phi(q) (kv) / (phi(q) . ksum) -> proj_l -> o_l (a14)", 4.0, 0.0),
    "td_sla_linear_out": ("linear branch, pass over Q, added to the attention output in place", 6.0, 0.0),
    "td_row_stats_finalize": ("row statistics from the GEMM epilogue's per-64-column pieces", 0.125, 8.0),
}
WORKLOADS = {
    "turbo": dict(attention_type="sagesla", quant_linear=True,
                  desc="TurboWan2.1-T2V-1.3B-480P 4-step: SageSLA top-k 0.1 + W8A8 + fused norms"),
    "c2": dict(attention_type="sage", quant_linear=False,
               desc="Wan2.1-T2V-1.3B 480p 4-step: dense SageAttention INT8-QK (no sparsity), bf16 linears (SURVEY §8d: C2)"),
    "c2w8a8": dict(attention_type="sage", quant_linear=True,
                   desc="Wan2.1-T2V-1.3B 480p 4-step: dense SageAttention INT8-QK + W8A8 + fused norms (rounds 2-4 ran this as 'c2')"),
    "c3": dict(attention_type="sagesla", quant_linear=False,
               desc="Wan2.1-T2V-1.3B 480p 4-step: SageSLA top-k 0.1, bf16 linears"),
    "original": dict(attention_type="original", quant_linear=False,
                     desc="Wan2.1-T2V-1.3B 480p 4-step: dense bf16 attention, bf16 linears"),
}


def build_model(name, wl, dev, topk, num_layers=None):
    from turbodiffusion_amd.wan import MODEL_CONFIGS, WanModel
    from turbodiffusion_amd import kernels as K

    cfg = dict(MODEL_CONFIGS[name])
    if num_layers:
        cfg["num_layers"] = num_layers
    with torch.device(dev):
        net = WanModel(attention_type=wl["attention_type"], sla_topk=topk, quant_linear=wl["quant_linear"], **cfg)
    g = torch.Generator(device=dev).manual_seed(0)
    dim = cfg["dim"]
    with torch.no_grad():
        for name_, mod in net.named_modules():
            if hasattr(mod, "int8_weight"):  # Int8Linear: seeded bf16 weight -> HIP block quantiser
                o, i = mod.int8_weight.shape
                w = (torch.randn(o, i, device=dev, generator=g) * (1.0 / math.sqrt(i))).bfloat16()
                mod.int8_weight, mod.scale = K.quant_i8_block128(w)
                mod.bias = (torch.randn(o, device=dev, generator=g) * 0.02).bfloat16()
            elif isinstance(mod, torch.nn.Linear):
                std = 0.02 if ("proj_l" in name_ or "embedding" in name_ or "projection" in name_ or "head" in name_) \
                    else 1.0 / math.sqrt(mod.in_features)
                mod.weight.copy_((torch.randn(mod.weight.shape, device=dev, generator=g) * std).to(mod.weight.dtype))
                mod.bias.copy_((torch.randn(mod.bias.shape, device=dev, generator=g) * 0.02).to(mod.bias.dtype))
            elif hasattr(mod, "weight") and isinstance(getattr(mod, "weight"), torch.Tensor) and mod.weight is not None \
                    and mod.weight.dim() == 1:
                mod.weight = (1 + 0.1 * torch.randn(mod.weight.shape, device=dev, generator=g)).to(mod.weight.dtype)
                if getattr(mod, "bias", None) is not None:
                    mod.bias = (0.02 * torch.randn(mod.bias.shape, device=dev, generator=g)).to(mod.bias.dtype)
        for blk in net.blocks:
            blk.modulation.copy_(torch.randn(1, 6, dim, device=dev, generator=g) / math.sqrt(dim))
        net.head.modulation.copy_(torch.randn(1, 2, dim, device=dev, generator=g) / math.sqrt(dim))
    return net.eval(), cfg


def pmc_traffic(prefixes):
    """HBM-side bytes per launch (fetch + write) of the kernels whose names start with one of ``prefixes``, from the
    committed rocprofv3 PMC summary of this same command (profiles/rNN_pmc_hbm_traffic.json, latest round; produced by
    tools/gpu/pmc_hbm_traffic.sh + tools/pmc_traffic.py: FETCH_SIZE and WRITE_SIZE in separate passes, gfx950 x2 fetch correction).
    bench.py itself cannot collect counters while it times; None if the summary is absent."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")))   # the latest round's summary
    if not cands:
        return None
    path = cands[-1]
    ks = json.load(open(path))["kernels"]
    n = b = 0.0
    for name, v in ks.items():
        if any(name.startswith(p) for p in prefixes):
            n += v["launches"]
            b += v["launches"] * (v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"])
    return (b / n) if n else None


_PMC_SOURCES = ("gemm_w8a8_fi.hip", "gemm_w8a8.hip", "gemm_w8a8_m32.hip", "attn.hip", "td_common.h")


def kernel_source_digest():
    """sha1 over the sources of the kernels the PMC summary describes — what tools/pmc_traffic.py stamps into the summary
    and what ``pmc_meta`` compares (the GPU box has no .git, so a commit hash could not be checked there)."""
    import hashlib
    h = hashlib.sha1()
    for name in _PMC_SOURCES:
        with open(os.path.join(ROOT, "turbodiffusion_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_meta():
    """{"traffic_file", "traffic_commit", "traffic_stale"}: which committed counter summary `traffic` came from and
    whether the GEMM / attention kernel sources are still the ones it was collected on (None: the summary predates the
    stamp — treat as stale)."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_hbm_traffic.json")))
    if not cands:
        return {}
    j = json.load(open(cands[-1]))
    dig = j.get("kernel_source_digest")
    return {"traffic_file": os.path.relpath(cands[-1], ROOT), "traffic_commit": j.get("commit"),
            "traffic_stale": (dig != kernel_source_digest()) if dig else None}


def collect_traffic(argv_model):
    """--collect-traffic: the HBM-side bytes per launch of the GEMM / attention kernels, collected IN THIS RUN (after the timed
    region): two child processes under ``rocprofv3 --pmc`` (FETCH_SIZE and WRITE_SIZE need separate passes: TCC counter slots,
    MI355X_MICROARCH.md) run ONE eager DiT forward of the same model (full-size launches: token split off), and the
    counter_collection.csv files are reduced exactly as tools/pmc_traffic.py reduces them.  ~40 s per pass.  Opt-in: the
    default line reads the committed summary and says whether it is stale (``traffic_stale``)."""
    import shutil
    import subprocess
    import tempfile
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_traffic as PT
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    out = {}
    with tempfile.TemporaryDirectory(dir="/tmp") as td:
        for name, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
            d = os.path.join(td, name)
            cmd = ["rocprofv3", "--pmc", ctr, "--kernel-trace", "-d", d, "-o", "b", "--output-format", "csv", "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--num-steps", "1", "--no-graph", "--no-cpu-baseline",
                   "--no-box-calibration"] + argv_model
            env = dict(os.environ, TD_BENCH_MODEL_FLAGS="split_tokens=0,split_qkv=0", TMPDIR="/tmp")
            r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=900)
            csvs = [os.path.join(dp, f) for dp, _, fs in os.walk(d) for f in fs if f.endswith("counter_collection.csv")]
            if r.returncode != 0 or not csvs:
                return None, f"rocprofv3 pass {name} failed (exit {r.returncode}): {r.stderr[-300:]}"
            out[name] = PT.load(csvs[0], ctr)
    res = {}
    for k in set(out["fetch"]) | set(out["write"]):
        f, w = out["fetch"].get(k, []), out["write"].get(k, [])
        res[k] = {"launches": max(len(f), len(w)), "fetch_bytes_per_launch": 2.0 * 1024.0 * sum(f) / max(1, len(f)),
                  "write_bytes_per_launch": 1024.0 * sum(w) / max(1, len(w))}
    return res, None


def traffic_of(kernels, prefixes):
    n = b = 0.0
    for name, v in kernels.items():
        if any(name.startswith(p) for p in prefixes):
            n += v["launches"]
            b += v["launches"] * (v["fetch_bytes_per_launch"] + v["write_bytes_per_launch"])
    return (b / n) if n else None


def cpu_operator_times(cfg, cores):
    """each operator's CPU restatement (oracle/ops_ref.py, oracle/sla_ref.py — the reference's arithmetic,
    the port the parity tests hold the HIP kernels to) timed on the host cores on a bounded slice (4096 of the L rows; per-row
    costs scale linearly, the attention entries carry their own L): ms per call, the cores used."""
    from oracle import ops_ref as O, sla_ref as S
    dim, H = cfg["dim"], cfg["num_heads"]
    Mc = 4096
    use = min(cores, 32)                 # the oracle's small per-block ops do not scale past a few dozen threads
    torch.set_num_threads(use)

    def ms(fn, reps=2):
        fn()
        t0 = time.time()
        for _ in range(reps):
            fn()
        return (time.time() - t0) / reps * 1e3

    g = torch.Generator().manual_seed(1)
    xc = torch.randn(Mc, dim, generator=g).bfloat16()
    wc = (torch.randn(dim, dim, generator=g) / math.sqrt(dim)).bfloat16()
    xq_, xs_ = O.quant_block128(xc)
    wq_, ws_ = O.quant_block128(wc)
    z = torch.zeros(1, 1, dim)
    H = min(H, 4)                        # attention entries: 4 heads (per-head cost is independent of the head count)
    qc, kc, vc = [torch.randn(1, Mc, H, 128, generator=g).bfloat16() for _ in range(3)]
    wp_, bp_ = torch.randn(128, 128, generator=g) * 0.05, torch.zeros(128)
    ops = {
        f"quant_block128 [{Mc}, {dim}]": ms(lambda: O.quant_block128(xc)),
        f"gemm_w8a8 [{Mc} x {dim} x {dim}]": ms(lambda: O.gemm_w8a8(xq_, xs_, wq_, ws_)),
        f"layernorm_fast + modulate [{Mc}, {dim}]": ms(lambda: O.modulate(O.layernorm_fast(xc, None, None, 1e-6), z, z)),
        f"rmsnorm_fast [{Mc}, {dim}]": ms(lambda: O.rmsnorm_fast(xc, torch.ones(dim), 1e-6)),
        f"sagesla_forward (block map + sparse INT8 attention + linear branch) L = {Mc}, H = {H}, top-k 0.1":
            ms(lambda: S.sagesla_forward(qc, kc, vc, wp_, bp_, 0.1), 1),
        f"scaled_dot_product_attention (C1's dense bf16) L = {Mc}, H = {H}":
            ms(lambda: torch.nn.functional.scaled_dot_product_attention(qc.transpose(1, 2), kc.transpose(1, 2), vc.transpose(1, 2)), 1),
    }
    return {"rows": Mc, "cores": use, "ms_per_call": {k: round(v, 2) for k, v in ops.items()},
            "what": "the oracle's CPU restatement of each accelerated operator (oracle/ops_ref.py, oracle/sla_ref.py) on a bounded slice"}


def cpu_baseline(cfg, lat_shape, topk, reps=3, blocks=2, full_video=False):
    """The reference's ORIGINAL eager path (config C1: SDPA, plain Linear, eager norms) timed on the host cores on a bounded
    sample of the same workload, by BASELINE.md §3's protocol: (ii) a reduced-depth model — ``blocks`` = 2 of the 30 blocks,
    so that the block -> block path is in the sample — of ONE DiT step at the full token count, ``reps`` = 3 repetitions, raw
    times and the extrapolation (x num_layers / blocks x 4 steps) both in the record; the embeddings + head once (a forward
    with 0 blocks), not extrapolated; (iii) the accelerated operators' CPU restatements in ``operators``; (i) one whole
    4-step video timed once only with ``--cpu-full-video`` (tens of minutes on any host: beyond a default bench run).

    kind "reference": the reference's own ``WanModel`` (rcm/networks/wan2pt1.py:598-721), imported unmodified through
    oracle/ref_harness.py — only where the reference tree exists (``TD_REFERENCE_ROOT`` / ``/root/reference``: the build
    container, never the GPU box).  kind "port": the oracle's restatement of that path (oracle/wan_ref.py), bit-pinned to the
    real reference by tests/test_oracle_cpu.py.  The record says which one ran."""
    from oracle import wan_ref as W
    from oracle import ref_harness as RH

    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    cb = dict(cfg, num_layers=blocks)
    sd = W.make_state_dict(cb, seed=0, with_proj_l=False)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(lat_shape, generator=g)
    ctx = torch.randn(1, 512, cfg.get("text_dim", 4096), generator=g).bfloat16()
    t = torch.tensor([[987.654]]).bfloat16()

    kind, nets = "port", None
    if RH.available() and os.environ.get("TD_CPU_BASELINE", "") != "port":
        try:
            nets = {}
            for nl in (0, blocks):
                cn = dict(cfg, num_layers=nl)
                sdn = {k: v for k, v in sd.items() if nl or not k.startswith("blocks.")}
                nets[nl] = RH.reference_wan_from_sd(cn, sdn, act_dtype=torch.bfloat16)
            kind = "reference"
        except Exception as e:   # the reference tree is there but does not import here: say so, time the port
            phase(f"cpu baseline: the reference WanModel could not be built ({e!r}); timing the oracle port")
            nets = None

    def run(nl):
        t0 = time.time()
        with torch.no_grad():
            if nets is not None:
                nets[nl](x.bfloat16(), t, ctx)
            else:
                W.wan_forward(sd, cb, x, t, ctx, mode="eager", act_dtype=torch.bfloat16, return_tokens=True, num_layers=nl)
        return time.time() - t0

    run(0)             # untimed: one-time costs (the reference's rope tables, first-call dispatch) are not the workload
    t_emb = run(0)
    raws = []
    for r in range(reps):
        raws.append(run(blocks))
        phase(f"cpu baseline ({kind}): {blocks}-block forward, repetition {r + 1} of {reps}: {raws[-1]:.1f} s")
    raw = sorted(raws)[len(raws) // 2]
    blk = (raw - t_emb) / blocks
    video_s = 4 * (t_emb + cfg["num_layers"] * blk)
    what = ("the reference's own WanModel (rcm/networks/wan2pt1.py, imported unmodified; bf16 weights, SDPA + nn.Linear)"
            if kind == "reference" else "oracle eager bf16 DiT (SDPA + nn.Linear; the port pinned bit for bit to the reference)")
    rec = {"value": 1.0 / video_s, "unit": "videos/s", "cores": cores, "kind": kind,
           "embeddings_s": t_emb, "block_s": blk, "blocks_in_sample": blocks, "forward_s_repetitions": raws,
           "extrapolation": f"x{cfg['num_layers'] / blocks:g} (blocks) x4 (steps)",
           "sample": f"{what}: embeddings + head {t_emb:.1f} s (measured once, not extrapolated) + {blocks} of {cfg['num_layers']} "
                     f"blocks of 1 of 4 steps at full L, {reps} repetitions (raw forwards {', '.join('%.1f' % b for b in raws)} s; "
                     f"median {raw:.1f} s = {blk:.1f} s per block), blocks x{cfg['num_layers'] / blocks:g}, steps x4 "
                     f"(BASELINE.md §3 (ii)); per-operator CPU timings in `operators` ((iii))"}
    try:
        rec["operators"] = cpu_operator_times(cfg, cores)
    except Exception as e:   # reported beside the baseline, never required for it
        rec["operators"] = {"error": repr(e)}
    if full_video:
        # (i) one whole 4-step video, once, through the reference's own sampler update on the full-depth model
        sdf = W.make_state_dict(cfg, seed=0, with_proj_l=False)
        gn = torch.Generator().manual_seed(1)
        noises = [torch.randn(lat_shape, generator=gn) for _ in range(4)]
        netf = RH.reference_wan_from_sd(cfg, sdf, act_dtype=torch.bfloat16) if kind == "reference" else None
        t0 = time.time()
        with torch.no_grad():
            W.rcm_sample((lambda xb, tb: netf(xb, tb, ctx)) if netf is not None else
                         (lambda xb, tb: W.wan_forward(sdf, cfg, xb, tb, ctx, mode="eager", act_dtype=torch.bfloat16)), x, noises)
        rec["full_video_s_measured_once"] = time.time() - t0
        phase(f"cpu baseline ({kind}): one whole 4-step video: {rec['full_video_s_measured_once']:.0f} s")
    torch.set_num_threads(cores)
    return rec


def box_record(net, cfg, L_tok, dev):
    """In-run calibration of the box (K.box_calibration): INT8 matrix-pipe rate, HBM read bandwidth, shader clock under a
    production ffn.2-shaped W8A8 GEMM of this model — so that two driver runs on two boxes can be compared."""
    from turbodiffusion_amd import kernels as K
    gemm_fn = None
    lin = net.blocks[0].ffn[2]
    if hasattr(lin, "int8_weight"):
        a = torch.randn(L_tok, cfg["ffn_dim"], device=dev).bfloat16()
        aq, as_ = K.quant_i8_block128(a)
        del a

        def gemm_fn():
            K.gemm_w8a8(aq, as_, lin.int8_weight, lin.scale, torch.bfloat16, bias=lin.bias)
    rec = K.box_calibration(gemm_fn, dev)
    rec["note"] = ("measured in this run before the timed region: v_mfma_i32_32x32x32_i8 on all SIMDs (constant operands), "
                   "4 GiB non-temporal read, shader clock = s_memtime / s_memrealtime x 100 MHz from a one-wave probe "
                   "running beside the model's ffn.2 GEMM")
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="timed videos (each = 4 DiT steps)")
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="", choices=[""] + sorted(CONFIGS), help="a BASELINE.json configuration (SURVEY.md §8d) as one "
                    "flag: sets --workload / --model / --res (/ --two-experts); explicit flags override")
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--model", default=None)
    ap.add_argument("--res", default=None, choices=sorted(RES))
    ap.add_argument("--num-steps", type=int, default=4, help="sampler steps per video")
    ap.add_argument("--topk", type=float, default=0.1)
    ap.add_argument("--two-experts", action="store_true", help="Wan2.2-A14B as the reference runs it (wan2.2_i2v_infer.py:"
                    "186-197): high- and low-noise experts, switched at t < 0.9 — BOTH resident in HBM (2 x 14 GB int8), "
                    "the switch inside the timed region; sigma_max = 200")
    ap.add_argument("--sage-pv", default="fp16", choices=["fp16", "fp8"], help="SageAttention P.V arithmetic: fp16 (the "
                    "reference's sm80 branch; north-star default) or fp8 (its sm89+ branch, SLA/core.py:217-239: e4m3 P and V, "
                    "fp8 MFMA)")
    ap.add_argument("--gemm-fast", type=int, default=0, choices=[0, 2, 4, 8], help="W8A8 GEMM one-VALU dequant, re-centred "
                    "every G K blocks (bounded difference to the exact arithmetic, csrc/gemm_w8a8_fi.hip); 0 = the library's "
                    "default (round 6: G = 8)")
    ap.add_argument("--gemm-exact", action="store_true", help="W8A8 GEMM with the reference's exact dequant (ops/gemm/utils.hpp:116-121: "
                    "two VALU per element and K block) instead of the library's default one-VALU form — the A/B arm of "
                    "profiles/r06_fast_dequant_ab.txt")
    ap.add_argument("--sigma-max", type=float, default=0.0, help="0: 80 for T2V, 200 for I2V (the scripts' defaults)")
    ap.add_argument("--layers", type=int, default=0, help="debug: override num_layers (INVALID as a bench number)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-full-video", action="store_true", help="cpu_baseline: also time ONE whole 4-step video of the eager path on "
                    "the host cores, once (BASELINE.md §3 (i); tens of minutes)")
    ap.add_argument("--no-box-calibration", action="store_true", help="skip the ~0.1 s in-run calibration of the box "
                    "(INT8 MFMA rate, HBM read bandwidth, shader clock under the ffn.2 GEMM) reported in `box`")
    ap.add_argument("--no-graph", action="store_true", help="enqueue every kernel eagerly instead of replaying "
                    "one captured hipGraph per DiT forward (sequence-parallel runs are always eager)")
    ap.add_argument("--sp", type=int, default=0, help="sequence-parallel group size of the TIMED region (GPUs sharing "
                    "one video); N/sp groups run independent videos.  Default 0 = N: ONE video sharded over all GPUs")
    ap.add_argument("--emulate-rank", default="", metavar="r/N", help="ONE GPU: run rank r's work of an N-way sequence-parallel "
                    "split (its token shard, its launches, gathered buffers of the real size filled with copies of its own shard, "
                    "no communication) and report the measured per-rank time beside a modelled wire term (`emulated_rank`); "
                    "a measurement tool for DESIGN.md §6 — the line is marked and carries no vs_baseline")
    ap.add_argument("--rccl-one-rank", action="store_true", help="ONE GPU: run the sequence-parallel code path over a REAL 1-rank "
                    "RCCL communicator (torch.distributed 'nccl', world size 1): every collective of the multi-GPU forward is "
                    "issued, captured inside the forward's hipGraph and replayed for real — all a one-GPU box can host of the "
                    "N-GPU run.  The line is marked (`rccl_one_rank`) and carries no vs_baseline")
    ap.add_argument("--collect-traffic", action="store_true", help="N = 1: collect `roofline.traffic` / `roofline_attention.traffic` in "
                    "THIS run (two rocprofv3 --pmc child passes over one eager DiT forward after the timed region, ~40 s each) "
                    "instead of reading the committed counter summary")
    ap.add_argument("--no-replica-leg", action="store_true", help="N > 1: skip the throughput-mode measurement (N "
                    "independent videos) that follows the timed region")
    ap.add_argument("--prompt-to-pixels", action="store_true", help="N = 1: after the timed region also time the user-visible "
                    "path prompt ids -> umT5-XXL text embedding -> the 4 steps -> whole-clip VAE decode (random-init weights of the "
                    "real architectures, all three models resident); reported in `prompt_to_pixels`, never as `value`")
    ap.add_argument("--two-in-flight", action="store_true", help="N = 1: after the timed region also measure two independent "
                    "videos in flight on two streams, interleaved step by step from one host thread (reported beside the "
                    "headline, never as `value`).  Opt-in since round 3: the round-2 form — two host THREADS replaying hipGraphs "
                    "concurrently — hung intermittently at the 14B sizes (the 600-s timeouts of the 720p runs, "
                    "profiles/r03_720p_timeout_root_cause.txt), and a hang here would cost the whole bench line")
    ap.add_argument("--no-two-in-flight", action="store_true", help="(accepted for old command lines; the leg is opt-in now)")
    ap.add_argument("--gemm-variant", type=int, default=int(os.environ.get("TD_GEMM_VARIANT", "0")),
                    help="W8A8 GEMM kernel selection knob (0 = automatic); all variants are bit-identical")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE",
                    help="td_set_tuning knob for A/B runs (integers, include/turbodiffusion_amd.h: TD_TUNE_*)")
    args = ap.parse_args()
    preset = CONFIGS[args.config or "headline"]
    args.workload = args.workload or preset["workload"]
    args.model = args.model or preset["model"]
    args.res = args.res or preset["res"]
    args.two_experts = args.two_experts or preset.get("two_experts", False)
    wd = float(os.environ.get("TD_BENCH_WATCHDOG_S", "0"))
    if wd > 0:   # every thread's Python stack to stderr if the run is still going after wd seconds (and every wd after)
        import faulthandler
        faulthandler.dump_traceback_later(wd, repeat=True, file=sys.stderr, exit=False)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher around us: become one.  `python bench.py --gpus N` re-executes itself as N ranks under
        # torch.distributed.run (rendezvous on 127.0.0.1, a free port); the children's stdout / stderr are ours, so rank 0's
        # ONE JSON line is this process's output; the exit code is the launcher's.
        import socket
        import subprocess
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        phase(f"no launcher environment: re-executing as {args.gpus} ranks under torch.distributed.run (port {port})")
        sys.exit(subprocess.call(cmd, env=env))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if world > 1:
        WATCHDOG.start(float(os.environ.get("TD_BENCH_HARD_TIMEOUT_S", "900")), rank)   # (re-armed by phase() and per timed video)
    emu = None
    if args.emulate_rank:
        er, en = (int(v) for v in args.emulate_rank.split("/"))
        assert world == 1 and 0 <= er < en and en > 1, "--emulate-rank r/N runs on ONE GPU, 0 <= r < N, N > 1"
        emu = (er, en)
    # TD_BENCH_BACKEND=gloo: development rig only — several ranks on the GPUs that exist (one, on the test box), gloo
    # collectives through host memory; exercises every line of the multi-rank path without an 8-GPU node
    backend = os.environ.get("TD_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    import torch.distributed as dist
    sp = args.sp if args.sp > 0 else world
    assert world % sp == 0, f"--sp {sp} must divide the number of GPUs ({world})"
    dp = world // sp
    sp_group, my_group = None, rank // sp
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
        if sp > 1:
            if dp == 1:
                sp_group = dist.group.WORLD
            else:  # every rank creates every group (collective call), keeps its own
                for gi in range(dp):
                    grp = dist.new_group(ranks=list(range(gi * sp, (gi + 1) * sp)))
                    if gi == my_group:
                        sp_group = grp

    from turbodiffusion_amd import kernels as K
    from turbodiffusion_amd.sampler import rcm_sample

    wl = WORKLOADS[args.workload]
    phase(f"library loaded; building {args.model} ({args.workload}) weights on {dev}")
    net, cfg = build_model(args.model, wl, dev, args.topk, args.layers or None)
    net_low = None
    if args.two_experts:
        assert cfg["model_type"] == "i2v", "--two-experts is the Wan2.2-A14B configuration"
        net_low, _ = build_model(args.model, wl, dev, args.topk, args.layers or None)   # second weight set, resident
    sigma_max = args.sigma_max or (200.0 if cfg["model_type"] == "i2v" else 80.0)
    if sp > 1:
        from turbodiffusion_amd import seqpar
        seqpar.enable(net, sp_group)
        if net_low is not None:
            seqpar.enable(net_low, sp_group)
    if emu is not None:
        from turbodiffusion_amd import seqpar
        for m_ in filter(None, (net, net_low)):
            seqpar.enable(m_, seqpar.EmulatedGroup(*emu))
    if args.rccl_one_rank:
        assert world == 1 and emu is None and sp == 1, "--rccl-one-rank runs on ONE GPU, alone"
        from turbodiffusion_amd import seqpar
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as s_:
                s_.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("TD_SP_WHOLE_GRAPH", "1")   # this rig is where the collectives-inside-the-graph form is exercised
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        for m_ in filter(None, (net, net_low)):
            seqpar.enable(m_, dist.group.WORLD)

    w, h = RES[args.res]
    lat_shape = (1, 16, 21, h // 8, w // 8)  # 81 frames -> 21 latent frames, VAE 8x spatial
    L_tok = 21 * (h // 16) * (w // 16)
    g = torch.Generator(device=dev).manual_seed(my_group)  # same stream on the ranks of a group, another video per group
    # (the seed only changes the synthetic latents; every rank does identical work)
    init_noise = torch.randn(lat_shape, dtype=torch.float32, device=dev, generator=g)
    text = torch.randn(1, 512, cfg.get("text_dim", 4096), device=dev, generator=g).bfloat16()
    # a NEW prompt per video (the real workload): every timed video gets its own text-embedding tensor, so the per-prompt
    # work — text MLP, the all-blocks cross-attention K|V GEMM, K's RMSNorm, the V^T tiles — runs INSIDE the timed region
    # once per video (the tensors stand in for umT5 outputs and are made before it; TD_BENCH_SAME_TEXT=1 restores round 2's
    # single text for A/B)
    n_text = 1 if os.environ.get("TD_BENCH_SAME_TEXT") == "1" else max(args.warmup, 1) + args.steps + 1
    texts = [text] + [torch.randn(1, 512, cfg.get("text_dim", 4096), device=dev, generator=g).bfloat16()
                      for _ in range(n_text - 1)]
    text_i = [0]
    y = None
    if cfg["model_type"] == "i2v":
        y = torch.cat([torch.zeros(1, 4, *lat_shape[2:], device=dev),
                       torch.randn(1, 16, *lat_shape[2:], device=dev, generator=g)], 1)
        y[:, :4, 0] = 1.0

    if args.gemm_variant:
        K.set_tuning(K.TUNE_GEMM_VARIANT, args.gemm_variant)
    if args.gemm_exact:
        assert not args.gemm_fast, "--gemm-exact and --gemm-fast exclude each other"
        K.set_tuning(K.TUNE_GEMM_FAST, 1)
    if args.gemm_fast:
        # round 6: every epilogue of the default (16x16x64) kernel carries the one-VALU dequant; --gemm-variant 5 = round 2's form
        # (the 32x32x32-MFMA kernel with its early-barrier schedule)
        K.set_tuning(K.TUNE_GEMM_FAST, args.gemm_fast)
        if args.gemm_variant == 5:
            K.set_tuning(4, 3)
    for m_ in filter(None, (net, net_low)):
        m_.sage_pv = args.sage_pv
    for kv in args.tune:
        key, val = kv.split("=")
        K.set_tuning(int(key), int(val))
    for kv in filter(None, os.environ.get("TD_BENCH_MODEL_FLAGS", "").split(",")):   # A/B runs: fuse_* attributes of WanModel
        key, val = kv.split("=")
        assert hasattr(net, key), key
        for m_ in filter(None, (net, net_low)):
            setattr(m_, key, bool(int(val)) if val.isdigit() else val)
    # sp > 1: the forward is a chain of graph segments with the RCCL all-gathers re-issued eagerly between them (graph.py)
    use_graph = not args.no_graph
    run_net, run_low = net, net_low
    if use_graph:
        from turbodiffusion_amd.graph import GraphedModel
        run_net = GraphedModel(net)
        run_low = None if net_low is None else GraphedModel(net_low)

    def one_video(model=None):
        eager = model is not None
        txt = texts[text_i[0] % len(texts)]
        text_i[0] += 1
        return rcm_sample(model or run_net, init_noise, txt, num_steps=args.num_steps, generator=g, y=y,
                          sigma_max=sigma_max, net_low=(net_low if eager else run_low), boundary=0.9)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    box = None
    if rank == 0 and not args.no_box_calibration:
        try:
            sp_eff = emu[1] if emu is not None else sp
            box = box_record(net, cfg, L_tok if sp_eff == 1 else -(-L_tok // 128 // sp_eff) * 128, dev)
            phase(f"box calibration: {box['i8_pops']:.2f} POP/s int8 MFMA, {box['hbm_read_tbps']:.2f} TB/s HBM read, "
                  f"shader clock {box.get('sclk_mhz_gemm', float('nan')):.0f} MHz under the ffn.2 GEMM ({box['sclk_mhz_idle']:.0f} alone)")
        except Exception as e:   # a reported extra
            box = {"error": repr(e)}
    phase(f"model built ({torch.cuda.memory_allocated() / 2**30:.1f} GiB allocated); warm-up"
          + (" + hipGraph capture" if use_graph else ""))
    for wi in range(max(args.warmup, 1 if use_graph else 0)):  # (graph capture happens in the first call)
        out = one_video()
        torch.cuda.synchronize()
        phase(f"warm-up video {wi + 1} done ({torch.cuda.memory_reserved() / 2**30:.1f} GiB reserved)")
    timer = K.KernelTimer({"td_gemm_w8a8", "td_attn_i8", "td_gemm_bf16"})
    if not use_graph:
        K.set_timer(timer)  # HIP events around the dominant kernels, on the launch stream, in the timed region
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = one_video()
        WATCHDOG.kick()
    sync()
    elapsed = time.perf_counter() - t0
    K.set_timer(None)
    phase(f"timed region done: {args.steps} videos in {elapsed:.2f} s")
    eager_elapsed = None
    hbm_timer = None
    if use_graph:
        # a replayed graph has no per-launch Python hook: the per-kernel HIP events are taken on one more
        # video of the SAME workload enqueued eagerly right after the timed region (same kernels, same stream)
        # ... with the token-half split of the block tails OFF for that video: with it two GEMMs share the chip most of the
        # time and a launch's event-to-event duration would be that of two kernels; the roofline wants one kernel's own time
        # (round 5: the same for the two-launch q|k|v projection, whose Q GEMM runs beside the K-side glue chain on purpose:
        # for this video the projection is ONE launch with nothing beside it)
        saved_split = [(m_, m_.split_tokens, m_.split_qkv) for m_ in filter(None, (net, net_low))]
        for m_, _v, _q in saved_split:
            m_.split_tokens = False
            m_.split_qkv = False
        K.set_timer(timer)
        sync()
        t1 = time.perf_counter()
        one_video(net)
        sync()
        eager_elapsed = time.perf_counter() - t1
        K.set_timer(None)
        for m_, v_, q_ in saved_split:
            m_.split_tokens = v_
            m_.split_qkv = q_
        phase(f"eager video with per-kernel events done ({eager_elapsed:.2f} s)")
        # round 6: the HBM-bound family (quantiser / norms / modulate / RoPE / pooling / linear branch) against ITS
        # roofline — one more eager video with every schedule switch that runs two kernels side by side off (two streams too),
        # HIP events around the C-ABI entry points of the family (kernels.call): each launch's own duration
        if world == 1 and wl["quant_linear"] and emu is None:
            saved_sched = [(m_, m_.split_tokens, m_.split_qkv, m_.two_streams) for m_ in filter(None, (net, net_low))]
            for m_, *_r in saved_sched:
                m_.split_tokens = m_.split_qkv = m_.two_streams = False
            hbm_timer = K.KernelTimer((), entries=HBM_FAMILY)
            K.set_timer(hbm_timer)
            sync()
            one_video(net)
            sync()
            K.set_timer(None)
            for m_, v_, q_, t_ in saved_sched:
                m_.split_tokens, m_.split_qkv, m_.two_streams = v_, q_, t_
            phase("eager single-stream video with events around the HBM-bound entry points done")
    # ---- serving-style extra (N = 1): two independent videos in flight (two graph replays on two streams); the GPU
    #      fills one video's bubbles (GEMM prologues / store phases, barrier waits) with the other's kernels
    two_in_flight = None
    if world == 1 and use_graph and args.two_in_flight and not args.no_two_in_flight:
        try:
            from turbodiffusion_amd.graph import GraphedModel
            from turbodiffusion_amd.sampler import rcm_sample_iter
            ctxs = []
            for sd in (11, 12):
                g2 = torch.Generator(device=dev).manual_seed(sd)
                n2 = torch.randn(lat_shape, dtype=torch.float32, device=dev, generator=g2)
                ctxs.append((g2, n2, GraphedModel(net), None if net_low is None else GraphedModel(net_low), torch.cuda.Stream()))

            def run2(n):
                """n rounds of two videos, interleaved step by step from THIS thread: video i's step is enqueued on its
                own stream (graph replay + sampler update are asynchronous), then the other video's."""
                for r in range(n):
                    its = []
                    for i, (g2, n2, gm, gml, st) in enumerate(ctxs):
                        with torch.cuda.stream(st):
                            its.append(rcm_sample_iter(gm, n2, texts[(2 * r + i) % len(texts)], num_steps=args.num_steps,
                                                       generator=g2, y=y, sigma_max=sigma_max, net_low=gml, boundary=0.9))
                    for _ in range(args.num_steps):
                        for i, it_ in enumerate(its):
                            with torch.cuda.stream(ctxs[i][4]):
                                next(it_)

            run2(1)     # capture + warm-up
            sync()
            phase("two-videos-in-flight: both contexts captured")
            t2 = time.perf_counter()
            run2(args.steps)
            sync()
            two_in_flight = 2 * args.steps / (time.perf_counter() - t2)
            del ctxs
            phase("two-videos-in-flight leg done")
        except Exception as e:  # an extra, never fatal
            two_in_flight = repr(e)

    # ---- user-visible extra (N = 1): prompt ids -> text embedding -> sampling -> pixels (scope row f4), never the headline
    p2p = None
    if world == 1 and args.prompt_to_pixels:
        try:
            from turbodiffusion_amd import text_encoder as TE, vae_decode as VD
            enc = TE.Umt5Encoder(TE.synthetic_state_dict(device=dev), device=dev)       # XXL: 24 layers, dim 4096 (10.7 GiB)
            vae = VD.WanVaeDecoder(VD.synthetic_state_dict(), device=dev)               # dim 96, HIP convolution kernel
            gi = torch.Generator(device=dev).manual_seed(5)
            ids = torch.randint(1, 256384, (1, 512), device=dev, generator=gi)
            msk = torch.zeros(1, 512, dtype=torch.long, device=dev)
            msk[0, :64] = 1                                                             # a 64-token prompt

            i2v_enc = img = None
            if cfg["model_type"] == "i2v":      # the conditioning channels come from the VAE encoder (wan2.2_i2v_infer.py:139-152)
                from turbodiffusion_amd import vae_encode as VE
                from turbodiffusion_amd.pipeline import i2v_condition
                i2v_enc = VE.WanVaeEncoder(VE.synthetic_state_dict(), device=dev)
                img = torch.rand(1, 3, h, w, device=dev, generator=gi) * 2 - 1

            def prompt_to_pixels():
                emb = enc(ids, msk)
                if emb.shape[-1] != cfg.get("text_dim", 4096):
                    raise RuntimeError("text width mismatch")
                yy = y if i2v_enc is None else i2v_condition(i2v_enc, img, 81)
                z = rcm_sample(run_net, init_noise, emb, num_steps=args.num_steps, generator=g, y=yy, sigma_max=sigma_max,
                               net_low=run_low, boundary=0.9)
                return vae.decode(z)

            vid = prompt_to_pixels()
            sync()
            ts = {}
            t0 = time.perf_counter()
            emb = enc(ids, msk)
            sync()
            ts["umt5_ms"] = (time.perf_counter() - t0) * 1e3
            if i2v_enc is not None:
                t0 = time.perf_counter()
                i2v_condition(i2v_enc, img, 81)
                sync()
                ts["vae_encode_ms"] = (time.perf_counter() - t0) * 1e3
            t0 = time.perf_counter()
            z = rcm_sample(run_net, init_noise, emb, num_steps=args.num_steps, generator=g, y=y, sigma_max=sigma_max,
                           net_low=run_low, boundary=0.9)
            sync()
            ts["sampling_ms"] = (time.perf_counter() - t0) * 1e3
            t0 = time.perf_counter()
            vid = vae.decode(z)
            sync()
            ts["vae_decode_ms"] = (time.perf_counter() - t0) * 1e3
            t0 = time.perf_counter()
            for _ in range(3):
                vid = prompt_to_pixels()
            sync()
            p2p = {"seconds_per_video": (time.perf_counter() - t0) / 3, **{k: round(v, 2) for k, v in ts.items()},
                   "video_shape": list(vid.shape), "finite": bool(torch.isfinite(vid).all()),
                   "what": "64-token prompt ids -> umT5-XXL (random init) -> [I2V: image -> VAE encode -> conditioning] -> 4-step "
                           "sampling (hipGraph) -> whole-clip VAE decode (random init), all resident; tokenisation and file "
                           "writing excluded"}
            del enc, vae, vid
            torch.cuda.empty_cache()
            phase("prompt-to-pixels leg done")
        except Exception as e:  # an extra, never fatal
            p2p = {"error": repr(e)}

    multi = None
    if world > 1 or args.rccl_one_rank:     # (round 6: the one-rank RCCL rig emits the same record — the probe path proven before any 8-GPU run)
        if args.rccl_one_rank:
            backend = "nccl"
        cdev = dev if backend == "nccl" else "cpu"
        mine = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)                        # per-rank wall time of the timed region (same barriers on both sides)
        per_rank = [t_.item() / args.steps / args.num_steps * 1e3 for t_ in every]
        tt = torch.tensor([elapsed], device=cdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = tt.item()
        # ---- what the first real multi-GPU record must be able to validate (the 120 GB/s wire model, "only the
        # first head group's transfer exposed"): one more DiT forward enqueued EAGERLY with an event pair around every
        # stream-side wait for an asynchronous gather (seqpar.WAIT_PROBE) — the time the compute stream sat waiting for the
        # wire, per collective, on every rank
        waits = None
        if sp > 1 or args.rccl_one_rank:
            from turbodiffusion_amd import seqpar as _sq
            _sq.WAIT_PROBE = []
            sync()
            with torch.no_grad():
                net(init_noise.to(net.dtype), torch.full((1, 1), 900.0, device=dev, dtype=net.dtype), texts[0], y_B_C_T_H_W=y)
            sync()
            probe, _sq.WAIT_PROBE = _sq.WAIT_PROBE, None
            ms = [a_.elapsed_time(b_) for a_, b_, _ in probe]
            nl = cfg["num_layers"]
            loc = torch.tensor([sum(ms), max(ms) if ms else 0.0, float(len(ms))], device=cdev, dtype=torch.float64)
            allw = [torch.zeros_like(loc) for _ in range(world)]
            dist.all_gather(allw, loc)
            waits = {"waits_probed_per_forward": int(loc[2].item()),
                     "exposed_wait_ms_per_dit_step": {"min_over_ranks": min(w_[0].item() for w_ in allw), "max_over_ranks": max(w_[0].item() for w_ in allw)},
                     "exposed_wait_us_per_layer_rank0": (sum(ms) / nl * 1e3) if nl else None,
                     "longest_single_wait_ms": max(w_[1].item() for w_ in allw),
                     "bytes_gathered_per_forward_rank0": int(sum(b_ for _, _, b_ in probe)),
                     "what": "HIP events around every work.wait() of an asynchronous all-gather in ONE eagerly enqueued DiT forward after the "
                             "timed region (compute-stream time spent waiting for the wire; 0 probes: the backend gathers synchronously — the gloo rig)"}
        nccl_ver = None
        try:
            nccl_ver = ".".join(str(v_) for v_ in torch.cuda.nccl.version())
        except Exception:
            pass
        multi = {"rccl": {"world_size": dist.get_world_size(), "backend": dist.get_backend(), "nccl_version": nccl_ver,
                          "sequence_parallel_group_size": sp, "graph_mode": None},
                 "per_rank_dit_step_ms": {"min": min(per_rank), "max": max(per_rank), "all": [round(v_, 3) for v_ in per_rank]},
                 "exposed_wait": waits}
    assert torch.isfinite(out).all(), "non-finite latents"

    # ---- throughput mode beside it (N > 1, sequence-parallel timed region): N independent videos, one per rank, the
    #      single-GPU path (hipGraph replay), no data-path collective; reported in `replicas`, never as `value`
    replicas = None
    if world > 1 and sp > 1 and not args.no_replica_leg:
        from turbodiffusion_amd import seqpar
        from turbodiffusion_amd.graph import GraphedModel
        seqpar.disable(net)
        if net_low is not None:
            seqpar.disable(net_low)
        g_own = torch.Generator(device=dev).manual_seed(1000 + rank)
        noise_own = torch.randn(lat_shape, dtype=torch.float32, device=dev, generator=g_own)
        gm = GraphedModel(net)
        gm_low = None if net_low is None else GraphedModel(net_low)

        def own_video():
            return rcm_sample(gm, noise_own, text, num_steps=args.num_steps, generator=g_own, y=y, sigma_max=sigma_max,
                              net_low=gm_low, boundary=0.9)
        own_video()                       # capture + warm-up
        sync()
        t3 = time.perf_counter()
        for _ in range(args.steps):
            o3 = own_video()
        sync()
        dt3 = time.perf_counter() - t3
        tt3 = torch.tensor([dt3], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt3, op=dist.ReduceOp.MAX)
        assert torch.isfinite(o3).all(), "non-finite latents (replica leg)"
        replicas = {"videos_per_s": world * args.steps / tt3.item(), "ms_per_video_per_gpu": tt3.item() / args.steps * 1e3,
                    "mode": f"{world} independent videos, one per GPU, hipGraph replay, no data-path collective"}

    live_traffic, live_err = None, None
    if rank == 0 and world == 1 and args.collect_traffic and emu is None:
        argv_model = ["--workload", args.workload, "--model", args.model, "--res", args.res, "--topk", str(args.topk), "--sage-pv", args.sage_pv]
        if args.two_experts:
            argv_model.append("--two-experts")
        if args.layers:
            argv_model += ["--layers", str(args.layers)]
        phase("collecting HBM traffic counters (two rocprofv3 --pmc passes over one eager forward)")
        try:
            live_traffic, live_err = collect_traffic(argv_model)
        except Exception as e:   # an extra: never fatal
            live_err = repr(e)
        phase("traffic counters " + ("collected" if live_traffic else f"NOT collected: {live_err}"))
    if rank == 0:
        per_video = elapsed / args.steps   # per sequence-parallel group
        value = dp / per_video             # whole job: dp groups generate dp videos per step
        summ = timer.summary()
        roof = None
        if "td_gemm_w8a8" in summ and wl["quant_linear"]:
            gs = summ["td_gemm_w8a8"]
            flops = sum(2.0 * m * n * k for (m, n, k) in gs["metas"]) / gs["launches"]
            ach = flops / (gs["avg_ms"] * 1e-3)
            roof = {"kernel": "gemm_w8a8_kernel (W8A8 block-scaled INT8 GEMM)", "bound": "mfma",
                    "achieved": ach / 1e12, "peak": I8_PEAK / 1e12, "unit": "TFLOP/s", "frac": ach / I8_PEAK,
                    "traffic": pmc_traffic(("gemm_w8a8_",)), "traffic_unit": "B/launch (HBM-side fetch+write, PMC)",
                    "algorithmic_bytes": sum(m * k + n * k + 2.0 * m * n for (m, n, k) in gs["metas"]) / gs["launches"],
                    "traffic_source": "committed rocprofv3 PMC summary of this command (profiles/, latest round) — not "
                                      "collected in this run; traffic_stale says whether the kernel sources changed since",
                    **pmc_meta(),
                    "avg_launch_ms": gs["avg_ms"], "launches": gs["launches"],
                    # GEMM time of one video (events, eager video) over the TIMED video (graph replay); the GEMMs run
                    # back to back on the main stream, so this is the fraction of the step they occupy
                    "share_of_step": gs["total_ms"] * 1e-3 / per_video}
            if box and box.get("i8_pops"):
                roof["frac_of_box"] = ach / (box["i8_pops"] * 1e15)
        roof16 = None
        if "td_gemm_bf16" in summ and not wl["quant_linear"]:
            # C2 / C3 / original: the blocks' Linears are plain 16-bit GEMMs on td_gemm_bf16 (only launches with m >= 1024 count:
            # the text MLP's and the split-K pieces are another regime)
            gs = summ["td_gemm_bf16"]
            recs = [(m_, ms_) for m_, ms_ in zip(gs["metas"], [a_.elapsed_time(b_) for a_, b_, _ in timer.records["td_gemm_bf16"]]) if m_[0] >= 1024]
            if recs:
                fl16 = sum(2.0 * m * n * k for (m, n, k), _ in recs)
                t16 = sum(ms_ for _, ms_ in recs) * 1e-3
                roof16 = {"kernel": "gemm_bf16_kernel (16-bit GEMM, fp32 accumulate)", "bound": "mfma", "achieved": fl16 / t16 / 1e12,
                          "peak": F16_PEAK / 1e12, "unit": "TFLOP/s", "frac": fl16 / t16 / F16_PEAK, "traffic": None,
                          "algorithmic_bytes": sum(2.0 * (m * k + n * k + m * n) for (m, n, k), _ in recs) / len(recs),
                          "avg_launch_ms": t16 * 1e3 / len(recs), "launches": len(recs), "share_of_step": t16 / per_video}
        roof_attn = None
        if "td_attn_i8" in summ:
            # The sparse attention kernel re-streams K/V per Q block from the L2 / Infinity Cache (PMC: ~0.8 GB of
            # HBM-side traffic per launch against 4.0 GB streamed), so HBM is not what bounds it; its roofline is the
            # matrix pipe: INT8 QK^T at 5 POP/s + FP16 PV at 2.5 PFLOP/s.  frac = that matrix time / measured time.
            a = summ["td_attn_i8"]
            by = mt = fl = 0.0
            for (H, L_, Lk, nsel) in a["metas"]:
                qb, kb = (L_ + 127) // 128, (Lk + 63) // 64
                ns = nsel if nsel else kb
                by += H * qb * (128 * 128 * 1 + ns * 64 * 128 * 3 + 128 * 128 * 2)
                half = 2.0 * H * qb * 128 * ns * 64 * 128            # FLOPs of QK^T = FLOPs of PV
                fl += 2 * half
                mt += half / I8_PEAK + half / (I8_PEAK if args.sage_pv == "fp8" else F16_PEAK)   # fp8 MFMA = the int8 rate
            by, mt, fl = by / a["launches"], mt / a["launches"], fl / a["launches"]
            t_l = a["avg_ms"] * 1e-3
            roof_attn = {"kernel": f"attn_kernel<int8 QK, {args.sage_pv} PV>", "bound": "mfma",
                         "achieved": fl / t_l / 1e12, "peak": fl / mt / 1e12, "unit": f"TFLOP/s (int8 QK^T + {args.sage_pv} PV, harmonic)",
                         "frac": mt / t_l, "traffic": pmc_traffic(("attn_kernel<true",)),
                         "traffic_unit": "B/launch (HBM-side fetch+write, PMC)",
                         # K/V tiles are re-read per Q block from L2 / the Infinity Cache: that stream is CACHE traffic, not
                         # HBM traffic — the HBM-side rate is `hbm_side_frac_informational` below
                         "cache_streamed_bytes": by, "cache_streamed_GBps": by / t_l / 1e9,
                         "avg_launch_ms": a["avg_ms"], "launches": a["launches"],
                         "share_of_step": a["total_ms"] * 1e-3 / per_video}
            if roof_attn["traffic"]:
                # HBM side of the attention kernel: counter bytes per launch / launch time / 8 TB/s.  The north star's ">= 60 %
                # of the HBM roofline" is NOT met and cannot be by a kernel whose K/V re-reads hit in cache: it is
                # issue-bound on the matrix / VALU port, so its roofline is `frac` (matrix pipe)
                roof_attn["hbm_side_GBps"] = roof_attn["traffic"] / t_l / 1e9
                roof_attn["hbm_side_frac_informational"] = roof_attn["traffic"] / t_l / HBM_PEAK
                roof_attn["held_to"] = ("the matrix pipe (`frac`; DESIGN.md §5: target >= 0.45) — the HBM-side figure is information: "
                                        "the kernel's K / V re-reads are L2 / Infinity-Cache hits and it is issue-bound")
                roof_attn.update(pmc_meta())
        if live_traffic is not None:     # counters of THIS run replace the committed summary's
            for r_, pre in ((roof, ("gemm_w8a8_",)), (roof_attn, ("attn_kernel<true",))):
                tv = traffic_of(live_traffic, pre) if r_ is not None else None
                if tv is not None:
                    r_["traffic"] = tv
                    r_["traffic_source"] = "collected in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over one eager DiT forward after the timed region"
                    for k_ in ("traffic_file", "traffic_commit"):
                        r_.pop(k_, None)
                    r_["traffic_stale"] = False
                    if r_ is roof_attn:
                        r_["hbm_side_GBps"] = tv / (r_["avg_launch_ms"] * 1e-3) / 1e9
                        r_["hbm_side_frac_informational"] = tv / (r_["avg_launch_ms"] * 1e-3) / HBM_PEAK
        elif live_err is not None and roof is not None:
            roof["traffic_collect_error"] = live_err
        if roof is None:    # no W8A8 GEMM in this workload: `roofline` = whichever of the 16-bit GEMM / the attention kernel holds more of the step
            cands = [r_ for r_ in (roof16, roof_attn) if r_ is not None]
            roof = max(cands, key=lambda r_: r_["share_of_step"]) if cands else None
        roof_hbm = None
        if hbm_timer is not None and hbm_timer.records:
            # per entry point: the launches over the video's own token count (the text side's 512-row launches are another regime)
            dim_ = cfg["dim"]
            kern, tot_b, tot_t = [], 0.0, 0.0
            for name, recs in hbm_timer.records.items():
                what, per_ld, per_row = HBM_FAMILY[name]
                ms_ = [a_.elapsed_time(b_) for a_, b_, _ in recs]
                big = [t_ for t_ in ms_ if t_ > 0.5 * max(ms_)] if name != "td_row_stats_finalize" else ms_
                byts = per_ld * L_tok * dim_ + per_row * L_tok
                t_avg = sum(big) / len(big) * 1e-3
                kern.append({"entry": name, "what": what, "launches": len(big), "avg_launch_us": t_avg * 1e6, "algorithmic_bytes": byts,
                             "achieved_GBps": byts / t_avg / 1e9, "frac": byts / t_avg / HBM_PEAK})
                tot_b += byts * len(big)
                tot_t += t_avg * len(big)
            kern.sort(key=lambda r_: -r_["avg_launch_us"] * r_["launches"])
            roof_hbm = {"bound": "hbm", "peak": HBM_PEAK / 1e9, "unit": "GB/s", "achieved": tot_b / tot_t / 1e9, "frac": tot_b / tot_t / HBM_PEAK,
                        "share_of_step": tot_t / per_video, "launches_per_video": sum(k_["launches"] for k_ in kern),
                        "how": "HIP events around each C-ABI entry point on its launch stream, one eager single-stream video (no second "
                               "stream, no token split) after the timed region; algorithmic bytes = every operand read or written once",
                        "kernels": kern}
        res = {
            "metric": f"end-to-end videos/sec (4-step rCM denoising loop, {args.model} {args.res})",
            "value": value, "unit": "videos/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": per_video * 1e3, "dit_step_ms": per_video * 1e3 / args.num_steps,
            "higher_is_better": True, "scaling": "strong" if (dp == 1 and world > 1) else "weak",
            "vs_baseline": (value * PUBLISHED_S[(args.model, args.res)]) if (
                args.workload == "turbo" and (args.model, args.res) in PUBLISHED_S and not args.layers
                and args.num_steps == 4) else None,
            "dtype": (f"int8 (W8A8 linears, {'exact dequant' if args.gemm_exact else 'one-VALU dequant G=%d' % (args.gemm_fast or 8)}, QK^T) + "
                      f"{args.sage_pv} PV + bf16 activations") if wl["quant_linear"] else f"bf16 (+int8 QK^T, {args.sage_pv} PV)",
            "data": "synthetic (seeded N(0,1) latents/text embedding, random-init weights of the named architecture)",
            "config": {"workload": wl["desc"].replace("Wan2.1-T2V-1.3B 480p", f"{args.model} {args.res}").replace(
                           "TurboWan2.1-T2V-1.3B-480P", f"Turbo{args.model}-{args.res.upper()}"), "model": args.model, "resolution": args.res, "tokens": L_tok,
                       "sampler_steps": args.num_steps, "sla_topk": args.topk, "sigma_max": sigma_max,
                       "experts": ("high-noise + low-noise, both resident in HBM, switch at t < 0.9 inside the timed region "
                                   "(steps: " + "/".join(__import__("turbodiffusion_amd.sampler", fromlist=["x"]).expert_schedule(
                                       args.num_steps, sigma_max, 0.9)) + ")") if net_low is not None else 1,
                       "global_batch": dp,
                       "parallelism": (f"EMULATION of rank {emu[0]} of {emu[1]} (sequence-parallel) on one GPU" if emu is not None
                                       else "single GPU") if world == 1 else (
                           f"dp{dp} x sp{sp}: {dp} independent videos, each sharded by sequence over {sp} GPUs "
                           f"(RCCL all-gather of the quantised K/V per layer)" if sp > 1 else f"dp{dp}: {dp} independent videos")},
            "roofline": roof, "roofline_attention": roof_attn, **({"roofline_gemm16": roof16} if roof16 is not None else {}),
            **({"roofline_hbm": roof_hbm} if roof_hbm is not None else {}),
            "launch_mode": (("hipGraph replay, one graph per DiT forward" if sp == 1 else
                             "hipGraph replay in segments, the all-gathers issued eagerly between them") +
                            "; kernel events from one eager video (full-size launches: token-half split off, q|k|v projection as one launch) after the timed region"
                            if use_graph else "eager enqueue; kernel events inside the timed region"),
        }
        if use_graph and getattr(run_net, "sp_graph_mode", None):
            res["launch_mode"] = "hipGraph replay: " + run_net.sp_graph_mode + "; kernel events from one eager video after the timed region"
            if getattr(run_net, "sp_whole_graph_error", None):
                res["sp_whole_graph_error"] = run_net.sp_whole_graph_error
        if args.rccl_one_rank:
            res["metric"] = "ONE-RANK RCCL rig: " + res["metric"]
            res["vs_baseline"] = None
            res["config"]["parallelism"] = "the sequence-parallel path over a 1-rank RCCL communicator on one GPU"
            res["rccl_one_rank"] = {
                "backend": dist.get_backend(), "graph_mode": getattr(run_net, "sp_graph_mode", None) if use_graph else "eager",
                "whole_graph_error": getattr(run_net, "sp_whole_graph_error", None) if use_graph else None,
                "dit_step_ms": per_video * 1e3 / args.num_steps,
                "what": "every collective of the N-GPU forward (smooth-K mean, the K-side all-gather per head group and layer, the "
                        "head output's gather) issued on a real RCCL communicator of ONE rank, captured inside the forward's "
                        "hipGraph and replayed; the difference to the single-GPU line is the sharded path's own overhead "
                        "(pack kernels, gathered-layout reads, RCCL's copy kernels)"}
        if emu is not None:
            from turbodiffusion_amd.seqpar import PackLayout
            spo = net.seq_parallel.sp
            at = wl["attention_type"]
            lay = PackLayout(cfg["num_heads"], spo.per, 128, spo.groups_for(cfg["num_heads"], spo.per), at in ("sage", "sagesla"), at in ("original", "sage"),
                             torch.bfloat16)
            pack = lay.total                            # bytes a rank sends to EVERY peer per self-attention layer
            link = 120e9                                # effective B/s of one xGMI link (153 GB/s peak; full mesh: one link per peer)
            nl = cfg["num_layers"]
            wire_layer = pack / link                    # every peer's pack arrives over its own link, all in parallel
            # what the emulation ADDS to a real rank's kernels: the device copies that fill the peers' slots of the gathered
            # buffers (in a real run those bytes arrive over xGMI, written by the fabric, not by this GPU's CUs) — timed
            # alone here, the same shapes back to back, so that the table can state the compute term with and without them
            srcs_ = [torch.empty(n_, dtype=torch.uint8, device=dev) for _, n_ in lay.pieces]
            dsts_ = [torch.empty((emu[1], n_), dtype=torch.uint8, device=dev) for _, n_ in lay.pieces]

            def copies():
                for s_, d_ in zip(srcs_, dsts_):
                    d_.copy_(s_.view(1, -1).expand(emu[1], -1))
            for _ in range(3):
                copies()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            ev0.record()
            for _ in range(nl):
                copies()
            ev1.record()
            torch.cuda.synchronize()
            copy_ms = ev0.elapsed_time(ev1)
            res["metric"] = "EMULATED rank: " + res["metric"]
            res["vs_baseline"] = None
            res["emulated_rank"] = {
                "rank": emu[0], "of": emu[1], "tokens_of_rank": spo.stop - spo.start, "tokens_per_rank_padded": spo.per,
                "measured_compute_ms_per_dit_step": per_video * 1e3 / args.num_steps,
                "of_which_emulation_gather_copies_ms": copy_ms,
                "branches_in_parallel": bool(spo.branches_in_parallel(cfg["num_heads"], spo.per, lay.G)),
                "pack_bytes_per_layer": pack, "head_groups": lay.G,
                "modelled_wire_ms_per_dit_step": {"link_GBps": link / 1e9, "fully_exposed": nl * wire_layer * 1e3,
                                                  "first_head_group_exposed": nl * wire_layer * lay.pieces[0][1] / lay.total * 1e3,
                                                  # round 6: bytes are not what decides a small shard — every collective also costs a launch +
                                                  # rendezvous latency that no byte model sees.  Exposed per layer: the early exchange(s) (the K
                                                  # quantiser waits for the sums) + the first piece; the later pieces fly under attention.
                                                  "collectives_per_layer": spo.collectives_per_layer(lay.G, at not in ("original", "sage")),
                                                  "exposed_collectives_per_layer": spo.exposed_collectives_per_layer(at not in ("original", "sage")),
                                                  "with_latency_us_per_collective": {
                                                      str(lat_us): nl * (wire_layer * lay.pieces[0][1] / lay.total + spo.exposed_collectives_per_layer(at not in ("original", "sage")) * lat_us * 1e-6) * 1e3
                                                      for lat_us in (10, 20, 40)}},
                "what": "one rank's kernels of an N-way sequence split on one GPU: its token shard, gathered buffers of the real "
                        "size filled by device copies of its own pack (the HBM writes of the incoming xGMI traffic, NOT overlapped), "
                        "no communication; the step of a real N-GPU run = this compute term + the exposed part of the wire term"}
        if use_graph and getattr(run_net, "sp_capture_error", None):
            res["launch_mode"] = "eager enqueue (segmented hipGraph capture failed: " + run_net.sp_capture_error + ")"
        if multi is not None:
            multi["rccl"]["graph_mode"] = (getattr(run_net, "sp_graph_mode", None) or "one hipGraph per DiT forward") if use_graph else "eager enqueue"
            res.update(multi)
        if box is not None:
            res["box"] = box
        res["config"]["prompts"] = ("a new text embedding per video: the per-prompt work (text MLP, all-blocks cross-attention "
                                    "K|V, K RMSNorm, V^T tiles) is inside the timed region") if len(texts) > 1 else "one text for all videos"
        if eager_elapsed is not None:
            res["eager_videos_per_s"] = 1.0 / eager_elapsed
        if replicas is not None:
            res["replicas"] = replicas
        if two_in_flight is not None:
            res["two_videos_in_flight_videos_per_s"] = two_in_flight
        if p2p is not None:
            res["prompt_to_pixels"] = p2p
        if args.layers:
            res["config"]["DEBUG_num_layers_override"] = args.layers
        if world == 1 and not args.no_cpu_baseline:
            try:
                res["cpu_baseline"] = cpu_baseline(cfg, lat_shape, args.topk, full_video=args.cpu_full_video)
            except Exception as e:  # the baseline is reported, never required for the GPU number
                res["cpu_baseline"] = {"error": repr(e)}
        WATCHDOG.cancel()
        print(json.dumps(res), flush=True)
    WATCHDOG.cancel()
    if world > 1 or args.rccl_one_rank:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
