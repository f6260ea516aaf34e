"""Host cost of enqueueing one eager DiT forward (ctypes launches + torch allocations): a forward on a tiny latent, where
the GPU work is negligible, so wall time ~ enqueue time.  Matters for the sequence-parallel latency mode (eager)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
net, cfg = bench.build_model("Wan2.1-1.3B", bench.WORKLOADS["turbo"], dev, 0.1)
x = torch.randn(1, 16, 2, 64, 64, device=dev).bfloat16()     # L = 2*32*32 = 2048 tokens
t = torch.tensor([[900.0]], device=dev).bfloat16()
ctx = torch.randn(1, 512, 4096, device=dev).bfloat16()
for _ in range(3):
    net(x, t, ctx)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); net(x, t, ctx); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append((round((t1 - t0) * 1e3, 2), round((t2 - t0) * 1e3, 2)))
print(json.dumps({"tokens": 2048, "enqueue_ms, total_ms": ts}))
