"""Dev probe: does a hipGraph of the step's four launches (sampler, chain, dW, step tail) beat four direct launches?
TIMING ONLY -- the per-step scalars (Philox offsets, AdamW step count) are frozen in the captured graph, so this is not a
usable training loop; it answers whether a device-side step-state buffer (which a graph-replayable step would need) is worth building."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from isdf_amd.engine import Engine, NetConfig, LossConfig, SampleConfig, pinned_stream
from isdf_amd import synthetic

eng = Engine(NetConfig(transform=synthetic.bounds_transform()), "cuda")
torch.manual_seed(0); eng.params.normal_(0, 0.06); eng.pack()
cam = dict(synthetic.REPLICA_CAM) if hasattr(synthetic, "REPLICA_CAM") else dict(synthetic.SCANNET_CAM)
d, n, T = synthetic.keyframes(5, cam, seed=1)
dev = lambda a: torch.as_tensor(a).cuda()
d, n, T = dev(d), dev(n), dev(T)
sc = SampleConfig(n_rays=200, **cam); lc = LossConfig()
idx = torch.arange(5, dtype=torch.int32, device="cuda")
optim = dict(lr=0.0013, weight_decay=0.012, betas=(0.9, 0.999), eps=1e-8)

def one(i):
    s = eng.sample(d, T, n, idx, idx, sc, seed=1, offset=i, reuse=True)
    eng.train_step(s, lc, sc, noise_std=0.05, noise_seed=1, noise_offset=i, optim=optim)

K = 400
side = torch.cuda.Stream()
with torch.cuda.stream(side), pinned_stream(torch.device("cuda")) as st:
    for i in range(300): one(i)
    st.synchronize()
    t = time.perf_counter()
    for i in range(K): one(i)
    st.synchronize()
    direct = (time.perf_counter() - t) / K
    # synchronised, direct
    t = time.perf_counter()
    for i in range(K):
        one(i); st.synchronize()
    direct_sync = (time.perf_counter() - t) / K
    g = torch.cuda.CUDAGraph()
    g.capture_begin()
    one(0)
    g.capture_end()
    for i in range(100): g.replay()
    st.synchronize()
    t = time.perf_counter()
    for i in range(K): g.replay()
    st.synchronize()
    graph = (time.perf_counter() - t) / K
    t = time.perf_counter()
    for i in range(K):
        g.replay(); st.synchronize()
    graph_sync = (time.perf_counter() - t) / K
print("direct launches: %.1f us/step pipelined, %.1f us synchronised;  graph replay: %.1f us/step pipelined, %.1f us synchronised"
      % (direct * 1e6, direct_sync * 1e6, graph * 1e6, graph_sync * 1e6))
