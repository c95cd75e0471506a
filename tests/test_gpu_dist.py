"""-m gpu: the collectives of interspeech_ser_amd/dist.py on the RCCL ("nccl") backend.  One GPU cannot host two RCCL ranks, so
the process group has ONE rank (dist.init(force=True)) -- every call, the device-resident fp32 bucket, its alignment, an encoder
built from its views and the release of the bucket afterwards are the real thing; the N-rank control flow runs under gloo
(tests/test_dist_gloo.py).  No scaling curve has been measured from inside a round.
Runs in a child process: a process group is global state."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys
sys.path.insert(0, os.environ["SER_ROOT"])
import numpy as np, torch
from interspeech_ser_amd import config as C, dist as D
from interspeech_ser_amd.engine import SpeechEncoder
from interspeech_ser_amd.weights import synthetic_state_dict
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
D.init("nccl", dev, force=True)
assert torch.distributed.is_initialized() and torch.distributed.get_backend() == "nccl"
assert D.broadcast_int(41) == 41 and D.max_over_ranks(2.5) == 2.5 and D.sum_over_ranks(2.5) == 2.5
geo = C.TINY_WAVLM
sd = synthetic_state_dict(geo, 11)
base = torch.cuda.memory_allocated()
out, dt, nbytes = D.broadcast_state_dict(dict(sd))
padded = sum((v.numel() + 3) // 4 * 4 for v in sd.values()) * 4
assert nbytes == padded, (nbytes, padded)
for k, v in sd.items():
    assert out[k].device.type == "cuda" and out[k].data_ptr() % 16 == 0, k          # views of the bucket, 16-byte aligned
    assert torch.equal(out[k].cpu(), v.float()), k
held = torch.cuda.memory_allocated() - base
assert held >= nbytes
waves = [(0.1 * np.random.default_rng(i).standard_normal(n)).astype(np.float32) for i, n in enumerate((16000, 9000))]
lens = [len(w) for w in waves]
enc_b = SpeechEncoder(geo, out, "cuda:0", mode="fp32x")                           # from the device-resident bucket
before = torch.cuda.memory_allocated()
del out
torch.cuda.synchronize()
freed = before - torch.cuda.memory_allocated()
assert freed >= nbytes, (freed, nbytes)                                            # the encoder kept no view of the bucket
enc_c = SpeechEncoder(geo, sd, "cuda:0", mode="fp32x")                            # from the CPU state dict
a = enc_b.forward(enc_b.upload(waves), lens).states.clone()
b = enc_c.forward(enc_c.upload(waves), lens).states
torch.cuda.synchronize()
err = float((a - b).abs().max() / max(1.0, float(b.abs().max())))
assert err < 2e-5, err            # the fp64 load-time folds ran on the device instead of the host: last-bit differences only
D.shutdown()
print("RCCL_SINGLE_RANK_OK", nbytes, freed)
"""


def test_rccl_single_rank_collectives_and_bucket_release():
    env = dict(os.environ, SER_ROOT=ROOT, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29513")
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_SINGLE_RANK_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
