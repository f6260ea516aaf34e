"""Does an eager RCCL collective that finished just before a hipGraph capture kill the process when the capture pulls the
communicator stream into capture mode?  (turbodiffusion_amd/graph.py: quiesce_collective_watchdog.)  One rank, one GPU:

    python tools/rccl_capture_race.py            # both legs in child processes, prints the outcome of each

leg "no drain":  eager all_gather -> synchronize -> capture (with an all_gather inside) held open for 0.6 s
leg "drain":     the same with quiesce_collective_watchdog() before the capture."""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def child(drain: bool):
    import torch
    import torch.distributed as dist
    from turbodiffusion_amd.graph import quiesce_collective_watchdog
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29500 + os.getpid() % 2000))
    dist.init_process_group("nccl", rank=0, world_size=1)
    torch.cuda.set_device(0)
    src = torch.arange(1 << 20, device="cuda", dtype=torch.float32)
    out = torch.empty(1 << 20, device="cuda", dtype=torch.float32)
    for _ in range(3):                                   # eager Works on the watchdog's list
        w = dist.all_gather_into_tensor(out, src, async_op=True)
        w.wait()
    torch.cuda.synchronize()
    if drain:
        quiesce_collective_watchdog()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.graph(g, capture_error_mode="thread_local"):
        w = dist.all_gather_into_tensor(out, src, async_op=True)     # the communicator stream joins the capture here
        y = src * 2.0
        w.wait()
        z = out + y
        time.sleep(0.6)                                              # several watchdog periods inside the capture
    g.replay()
    torch.cuda.synchronize()
    ok = bool(torch.equal(z, src * 3.0))
    time.sleep(0.3)
    dist.destroy_process_group()
    print(f"child finished: replay correct = {ok}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1] == "drain")
    else:
        for leg in ("nodrain", "drain", "nodrain", "drain"):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), leg], capture_output=True, text=True, timeout=180)
            why = [ln for ln in (r.stderr + r.stdout).splitlines() if "capturing stream" in ln or "child finished" in ln]
            print(f"leg {leg:8s}: exit code {r.returncode}; {why[0][:200] if why else (r.stderr.strip().splitlines() or ['?'])[-1][:200]}", flush=True)
