"""Per-launch floor of dependent tiny kernels: eager stream vs hipGraph replay (torch.cuda.CUDAGraph)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from glass_amd import ops, _lib

dev = torch.device("cuda", 0)
st = ops.rng_state(dev)
lib = _lib.load()
K = 200

def launches():
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(K):
        lib.glass_rng_advance(st.data_ptr(), s)

def timed(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps

print(f"eager: {timed(launches)/K*1e6:.2f} us per launch (host-bound if > graph)")
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    launches()
torch.cuda.synchronize()
with torch.cuda.graph(g):
    launches()
print(f"graph: {timed(g.replay)/K*1e6:.2f} us per kernel node (1-thread kernels, serial dependency)")
x = torch.randn(17080, 64, device=dev); o = torch.ones(64, device=dev); z = torch.zeros(64, device=dev)
def gns():
    for _ in range(20):
        ops.graphnorm(x, o, z, o)
with torch.cuda.stream(side):
    gns()
torch.cuda.synchronize()
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    gns()
print(f"graph: {timed(g2.replay)/60*1e6:.2f} us per GraphNorm kernel (3 per call, N=17080 C=64)")
