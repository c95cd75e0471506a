"""CPU: the N>1 path (weight broadcast C1, integer broadcast, counters, sharding) with
world_size 2 over gloo."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from interspeech_ser_amd import config as C
    from interspeech_ser_amd import dist as D
    from interspeech_ser_amd.weights import state_dict_digest, synthetic_state_dict
    assert D.init("gloo") == (rank, world, rank)
    sd = synthetic_state_dict(C.TINY_WAVLM, 5) if rank == 0 else None
    sd, dt, nbytes = D.broadcast_state_dict(sd)
    assert nbytes == sum((v.numel() + 3) // 4 * 4 for v in sd.values()) * 4 and dt >= 0     # entries padded to 16-byte boundaries
    digest = state_dict_digest(sd)
    import bench                                     # the record bench.py prints at N > 1: read from the live process group
    rec = bench.collective_record(world, dt, nbytes)
    assert rec["backend"] == "gloo" and rec["ranks"] == world and rec["bytes"] == nbytes and rec["GB_per_s"] is not None, rec
    assert bench.collective_record(1, 0.0, 0)["backend"] is None
    n = D.broadcast_int(17 if rank == 0 else -1)
    files = [f"f{i}" for i in range(9)]
    mine = D.shard_files(files, [100 - i for i in range(9)], rank, world)
    total = D.sum_over_ranks(len(mine))
    slow = D.max_over_ranks(1.0 + rank)
    torch.save({"digest": digest, "n": n, "mine": mine, "total": total, "slow": slow},
               os.path.join(out_dir, f"r{rank}.pt"))
    D.shutdown()


def test_world_size_2_gloo(tmp_path):
    from interspeech_ser_amd import config as C
    from interspeech_ser_amd.weights import state_dict_digest, synthetic_state_dict
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r = [torch.load(os.path.join(tmp_path, f"r{i}.pt")) for i in range(2)]
    want = state_dict_digest(synthetic_state_dict(C.TINY_WAVLM, 5))
    assert r[0]["digest"] == r[1]["digest"] == want            # rank 1 got rank 0's weights bit-for-bit
    assert r[0]["n"] == r[1]["n"] == 17
    assert sorted(r[0]["mine"] + r[1]["mine"]) == [f"f{i}" for i in range(9)]
    assert not set(r[0]["mine"]) & set(r[1]["mine"])
    assert r[0]["total"] == r[1]["total"] == 9.0 and r[0]["slow"] == r[1]["slow"] == 2.0


# ---------------------------------------------------------------------------------------------------------------------
# The drivers' multi-rank behaviour with the model call stubbed out: everything around it (deterministic sharding, the
# compat layer index taken once on rank 0 and broadcast, disjoint .pt writes, --skip_existing applied per shard) runs
# over gloo on the CPU, so that RCCL itself is the only thing the 8-GPU run adds.
LENGTHS = [4000, 5000, 6000, 7000, 8000, 9000, 10000]


class _StubExtractor:
    """Stands in for driver._Extractor: a 'feature' that encodes the layer index it was asked for and the rank that
    produced it.  Rank 1 is slow to come up (weights arrive later than on rank 0), which is what exposed the
    --skip_existing race the advisor found."""
    pipelined = False

    def __init__(self, args, whisper, device):
        import time
        from interspeech_ser_amd import config as C
        self.geo = C.TINY_WAVLM                      # 2 layers -> 3 hidden states
        self.weight_source = "stub"
        self.rank = int(os.environ["RANK"])
        self.log = os.path.join(os.environ["SER_TEST_LOG_DIR"], f"extracted_rank{self.rank}.txt")
        if self.rank == 1:
            time.sleep(1.0)

    def extract(self, waves, layer_index):
        with open(self.log, "a") as f:
            for w in waves:
                f.write(f"{len(w)}\n")
        return [torch.full((self.geo.frames_for(len(w)), 4), 100.0 * self.rank + float(-1 if layer_index is None else layer_index))
                for w in waves]


def _write_wavs(wav_dir):
    import wave
    import numpy as np
    os.makedirs(wav_dir, exist_ok=True)
    for i, n in enumerate(LENGTHS):
        with wave.open(os.path.join(wav_dir, f"utt_{i}.wav"), "wb") as wf:
            wf.setnchannels(1)
            wf.setsampwidth(2)
            wf.setframerate(16000)
            wf.writeframes((np.arange(n) % 100).astype("<i2").tobytes())


def _driver_worker(rank, world, port, root, extra):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), SER_TEST_LOG_DIR=root)
    from interspeech_ser_amd import dist as D
    from interspeech_ser_amd import driver
    D.init("gloo")
    rc = driver._run(["--ssl_type", "wavlm-large", "--wav_dir", os.path.join(root, "wav"), "--save_path",
                      os.path.join(root, "pt"), "--batch_size", "2", "--num_workers", "2"] + list(extra),
                     whisper=False, extractor_factory=_StubExtractor)
    assert rc == 0


def _extracted(root, rank):
    fn = os.path.join(root, f"extracted_rank{rank}.txt")
    return [int(x) for x in open(fn).read().split()] if os.path.isfile(fn) else []


def test_driver_two_ranks_compat_index_and_disjoint_writes(tmp_path):
    root = str(tmp_path)
    _write_wavs(os.path.join(root, "wav"))
    os.makedirs(os.path.join(root, "pt"))
    open(os.path.join(root, "pt", "leftover.txt"), "w").write("x")       # N = 1 file at start-up -> hidden_states[1]
    mp.spawn(_driver_worker, args=(2, _free_port(), root, ["--compat_layer_quirk"]), nprocs=2, join=True)
    got = [_extracted(root, r) for r in range(2)]
    assert sorted(got[0] + got[1]) == LENGTHS and not set(got[0]) & set(got[1])          # disjoint, exhaustive
    assert abs(sum(got[0]) - sum(got[1])) <= max(LENGTHS)                                  # equal audio within one file
    from interspeech_ser_amd import config as C
    for i, n in enumerate(LENGTHS):
        t = torch.load(os.path.join(root, "pt", f"utt_{i}.pt"))
        assert t.shape == (C.TINY_WAVLM.frames_for(n), 4) and t.dtype == torch.float32
        # the layer index is the file count rank 0 saw BEFORE anybody wrote, on both ranks
        assert float(t[0, 0]) % 100.0 == 1.0, (i, float(t[0, 0]))
        assert (float(t[0, 0]) >= 100.0) == (n in got[1])


def test_driver_two_ranks_skip_existing_shards_before_filtering(tmp_path):
    """Two outputs exist already; rank 1 starts a second after rank 0 has begun writing.  Every missing utterance
    is extracted exactly once and the existing files are left alone."""
    root = str(tmp_path)
    _write_wavs(os.path.join(root, "wav"))
    os.makedirs(os.path.join(root, "pt"))
    sentinel = torch.zeros(1, 1)
    for i in (1, 4):
        torch.save(sentinel, os.path.join(root, "pt", f"utt_{i}.pt"))
    mp.spawn(_driver_worker, args=(2, _free_port(), root, ["--skip_existing", "--use_n_layer", "--n_layer", "1"]), nprocs=2, join=True)
    got = [_extracted(root, r) for r in range(2)]
    want = [n for i, n in enumerate(LENGTHS) if i not in (1, 4)]
    assert sorted(got[0] + got[1]) == want, got
    assert not set(got[0]) & set(got[1])
    for i in (1, 4):
        assert torch.equal(torch.load(os.path.join(root, "pt", f"utt_{i}.pt")), sentinel)
    for i in (0, 2, 3, 5, 6):
        assert torch.load(os.path.join(root, "pt", f"utt_{i}.pt")).shape[1] == 4
