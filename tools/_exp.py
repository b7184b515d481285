import time, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from opencv_contrib_amd import cuda, synth
dev = torch.device("cuda:0")
prs = [synth.flow_pair(1080, 1920, seed=1234 + i) for i in range(4)]
I0 = torch.stack([torch.from_numpy(prs[i % 4][0]) for i in range(16)]).to(dev)
I1 = torch.stack([torch.from_numpy(prs[i % 4][1]) for i in range(16)]).to(dev)
flows = torch.empty((16, 1080, 1920, 2), dtype=torch.float32, device=dev)
def t(alg, steps, warm):
    for _ in range(warm): alg.calc_batch(I0, I1, flows)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): alg.calc_batch(I0, I1, flows)
    torch.cuda.synchronize(); return 16 * steps / (time.perf_counter() - t0)
d = cuda.OpticalFlowDual_TVL1.create()
print("defaults fresh", round(t(d, 2, 1), 1))
a30 = cuda.OpticalFlowDual_TVL1.create(iterations=30, epsilon=0.0)
print("n30", round(t(a30, 4, 1), 1))
print("defaults right after n30 (same object)", round(t(d, 2, 0), 1))
d2 = cuda.OpticalFlowDual_TVL1.create()
print("defaults new object after n30", round(t(d2, 2, 1), 1))
del a30
d3 = cuda.OpticalFlowDual_TVL1.create()
print("defaults new object after del", round(t(d3, 2, 1), 1))
a10 = cuda.OpticalFlowDual_TVL1.create(iterations=10, epsilon=0.0)
print("n10", round(t(a10, 5, 2), 1))
print("defaults again", round(t(d, 2, 0), 1), round(t(d, 4, 0), 1))
