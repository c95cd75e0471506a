"""What-if (GPU box): the two utterance groups of a batch as two INDEPENDENT hipGraphs on two streams, phase-shifted by a
fraction of a forward, against the production form (one graph, two parallel branches, joined per batch).  With a shift the
conv stem of one group (store / VALU bound) overlaps the encoder layers of the other (matrix-core bound)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from interspeech_ser_amd import config as C
from interspeech_ser_amd.engine import build_encoder
from interspeech_ser_amd.weights import synthetic_state_dict
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
geo = C.WAVLM_LARGE
sd = synthetic_state_dict(geo, 0)
enc = build_encoder(geo, sd, "cuda:0", mode)
waves = bench.synth_batch(16, 160000, 1234)
groups = [(enc.upload(waves[:8], slot=0), [160000] * 8), (enc.upload(waves[8:], slot=1), [160000] * 8)]
torch.cuda.synchronize()
N = 160
# production form
g2, hs = enc.capture_concurrent(groups)
for _ in range(10): g2.replay()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N): g2.replay()
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"one graph, two joined branches : {16 * N / dt:8.1f} utt/s  {1e3 * dt / N:6.3f} ms/batch", flush=True)
ref = [h.states.clone() for h in hs]
# two independent graphs
graphs = []
for slot, (w, l) in enumerate(groups):
    enc._plan(l, slot)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        enc.forward(w, l, slot=slot)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        out = enc.forward(w, l, slot=slot)
    graphs.append((g, s, out))
for frac in (0.0, 0.25, 0.5):
    for g, s, _ in graphs:
        with torch.cuda.stream(s):
            for _ in range(5): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(graphs[1][1]):
        torch.cuda._sleep(int(frac * 8.0e-3 * 2.1e9))           # ~frac of a concurrent forward (8 ms) in GPU cycles
    for _ in range(N):
        for g, s, _ in graphs:
            with torch.cuda.stream(s):
                g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    same = all(torch.equal(r, o.states) for r, (_, _, o) in zip(ref, graphs))
    print(f"two graphs, shift {frac:4.2f} forward : {16 * N / dt:8.1f} utt/s  {1e3 * dt / N:6.3f} ms/batch  (states equal: {same})", flush=True)
