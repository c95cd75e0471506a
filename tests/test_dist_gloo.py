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
    assert nbytes == sum(v.numel() for v in sd.values()) * 4 and dt >= 0
    digest = state_dict_digest(sd)
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
