"""One-pair Farneback calc() captured by the CALLER into a HIP graph (torch.cuda.CUDAGraph) and replayed: what the launch chain costs
without the host in it.  usage: python tools/fb_graph.py [W H [n]]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from opencv_contrib_amd import cuda, synth
W = int(sys.argv[1]) if len(sys.argv) > 1 else 640
H = int(sys.argv[2]) if len(sys.argv) > 2 else 480
n = int(sys.argv[3]) if len(sys.argv) > 3 else 300
dev = torch.device("cuda:0")
A0, A1, _ = synth.flow_pair(H, W, seed=5, dtype="u8")
a, b = torch.from_numpy(A0).to(dev), torch.from_numpy(A1).to(dev)
alg = cuda.FarnebackOpticalFlow.create()
out = alg.calc(a, b)
ref = out.clone()
for _ in range(5):
    alg.calc(a, b, out)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(n):
    alg.calc(a, b, out)
torch.cuda.synchronize()
plain = n / (time.perf_counter() - t)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    alg.calc(a, b, out)          # warm-up on the capture stream
torch.cuda.current_stream().wait_stream(s)
out.zero_()
with torch.cuda.graph(g, stream=s):
    alg.calc(a, b, out)
g.replay()
torch.cuda.synchronize()
same = bool(torch.equal(out, ref))
t = time.perf_counter()
for _ in range(n):
    g.replay()
torch.cuda.synchronize()
graph = n / (time.perf_counter() - t)
print(f"farneback {W}x{H}: {plain:.1f} calc/s enqueued by the host, {graph:.1f} calc/s as a captured graph; same flow: {same}", flush=True)
