"""Throughput with TWO independent videos in flight on one GPU (two hipGraph replays on two streams) vs one."""
import json, os, sys, threading, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from turbodiffusion_amd.sampler import rcm_sample
from turbodiffusion_amd.graph import GraphedModel
dev = torch.device("cuda", 0)
net, cfg = bench.build_model("Wan2.1-1.3B", bench.WORKLOADS["turbo"], dev, 0.1)
NV = 4


def make(seed):
    g = torch.Generator(device=dev).manual_seed(seed)
    noise = torch.randn((1, 16, 21, 60, 104), dtype=torch.float32, device=dev, generator=g)
    text = torch.randn(1, 512, 4096, device=dev, generator=g).bfloat16()
    return g, noise, text, GraphedModel(net), torch.cuda.Stream()


ctx = [make(0), make(1)]
for (g, noise, text, gm, st) in ctx:          # capture + warm-up, one after the other
    with torch.cuda.stream(st):
        rcm_sample(gm, noise, text, num_steps=4, generator=g)
torch.cuda.synchronize()


def run(i, n):
    g, noise, text, gm, st = ctx[i]
    with torch.cuda.stream(st):
        for _ in range(n):
            rcm_sample(gm, noise, text, num_steps=4, generator=g)


torch.cuda.synchronize(); t0 = time.perf_counter()
run(0, NV)
torch.cuda.synchronize(); one = (time.perf_counter() - t0) / NV
torch.cuda.synchronize(); t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(i, NV)) for i in range(2)]
[t.start() for t in th]; [t.join() for t in th]
torch.cuda.synchronize(); two = (time.perf_counter() - t0) / (2 * NV)
print(json.dumps({"one_in_flight_videos_per_s": round(1 / one, 4), "two_in_flight_videos_per_s": round(1 / two, 4),
                  "gain": round(one / two, 4)}))
