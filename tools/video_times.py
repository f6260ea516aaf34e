"""Per-video wall times of the default bench workload (is there a clock / power ramp after start-up?)."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from turbodiffusion_amd.sampler import rcm_sample
from turbodiffusion_amd.graph import GraphedModel
dev = torch.device("cuda", 0)
net, cfg = bench.build_model("Wan2.1-1.3B", bench.WORKLOADS["turbo"], dev, 0.1)
g = torch.Generator(device=dev).manual_seed(0)
noise = torch.randn((1, 16, 21, 60, 104), dtype=torch.float32, device=dev, generator=g)
text = torch.randn(1, 512, 4096, device=dev, generator=g).bfloat16()
gnet = GraphedModel(net)
ts = []
for i in range(14):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rcm_sample(gnet, noise, text, num_steps=4, generator=g)
    torch.cuda.synchronize(); ts.append(round((time.perf_counter() - t0) * 1e3, 1))
print(json.dumps({"ms_per_video": ts}))
