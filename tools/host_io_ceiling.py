"""Host-side ceiling of the 8-GPU run, measured WITHOUT eight GPUs (round-4 verdict item 4; SURVEY 8e: "expected scaling is limited by
host-side decode / write and PCIe, not xGMI").  N processes over gloo run the product's own driver (driver._run: file discovery, sharding,
native WAV decode into host memory, the two-slot pipeline's control flow, native .pt writer threads, atomic renames) on tmpfs, with a STUB
extractor in the place of the GPU: a batch "computes" for the batch time measured on the MI355X (7.7 ms per 16 x 10 s in bf16, 16.2 ms in
f16m, 18.9 ms in f16x; two batches in flight like the real pipeline) and returns a [T, D] fp32 block per utterance from a per-slot buffer.
What comes out is files/s per rank and in aggregate against the GPU-side rate the stub allows, per --num_workers: the number at which the
host side stops being able to feed N GPUs.

    python tools/host_io_ceiling.py [--ranks 8] [--files 1024] [--workers 4,8] [--batch-ms 7.7,16.2,18.9]

Reference: none (README.md:41-43: "run the script again on another GPU"); the per-file work is preprocess_speech.py:47 (decode) and :69-71 (write)."""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class StubExtractor:
    """driver._Extractor's pipelined interface (submit / collect / hold, SLOTS) with a clock in the place of the encoder."""
    pipelined = True
    SLOTS = 2
    RUNNING = 2

    def __init__(self, args, whisper, device):
        from interspeech_ser_amd import config as C
        self.geo = C.geometry_for("microsoft/wavlm-large")
        self.weight_source = "stub (tools/host_io_ceiling.py)"
        self.batch_s = float(os.environ["SER_STUB_BATCH_MS"]) * 1e-3
        self.free_at = [0.0, 0.0]                       # when each of the two "GPU lanes" (batches in flight) is free again
        self.buf = {}
        self.held = {}

    def _out(self, slot, rows):
        for f in self.held.pop(slot, []):
            f.result()                                  # writers still reading the slot's buffer
        b = self.buf.get(slot)
        if b is None or b.shape[0] < rows:
            b = self.buf[slot] = torch.randn((max(rows, 16 * 499), self.geo.hidden), dtype=torch.float32)
        return b[:rows]

    def submit(self, waves, layer_index, slot):
        lengths = [len(w) for w in waves]
        frames = [self.geo.frames_for(n) for n in lengths]
        # the batch occupies the earlier-free lane for 2 x batch_s (two batches share the GPU: each takes twice the per-batch time)
        now = time.perf_counter()
        lane = 0 if self.free_at[0] <= self.free_at[1] else 1
        start = max(now, self.free_at[lane])
        self.free_at[lane] = start + 2.0 * self.batch_s
        offs = np.concatenate([[0], np.cumsum(frames)])
        return dict(slot=slot, ready=self.free_at[lane], host=self._out(slot, int(offs[-1])), frame_offs=[int(x) for x in offs], lengths=lengths)

    def collect(self, ticket):
        dt = ticket["ready"] - time.perf_counter()
        if dt > 0:
            time.sleep(dt)
        host, fo = ticket["host"], ticket["frame_offs"]
        return [host[fo[b]: fo[b + 1]] for b in range(len(ticket["lengths"]))]

    def hold(self, slot, futures):
        self.held[slot] = list(futures)

    def extract(self, waves, layer_index):
        return self.collect(self.submit(waves, layer_index, 0))


def worker(rank, world, port, root, workers, batch_ms, out_q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                      SER_STUB_BATCH_MS=str(batch_ms), SER_PINNED_DECODE="0", CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="")
    torch.set_num_threads(1)
    import contextlib
    import io
    from interspeech_ser_amd import dist as D
    from interspeech_ser_amd import driver
    D.init("gloo")
    sink = io.StringIO()
    t0 = time.perf_counter()
    c0 = time.process_time()
    with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(io.StringIO()):
        driver._run(["--ssl_type", "microsoft/wavlm-large", "--wav_dir", os.path.join(root, "wav"), "--save_path", os.path.join(root, "pt"),
                     "--batch_size", "16", "--num_workers", str(workers), "--use_n_layer", "--n_layer", "-1"],
                    whisper=False, extractor_factory=StubExtractor)
    wall, cpu = time.perf_counter() - t0, time.process_time() - c0
    line = [x for x in sink.getvalue().splitlines() if x.startswith("SER_RUN ")]
    out_q.put((rank, wall, cpu, json.loads(line[0][8:]) if line else None))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--files", type=int, default=1024, help="files PER RANK")
    ap.add_argument("--workers", type=str, default="4,8")
    ap.add_argument("--batch-ms", type=str, default="7.7,16.2,18.9")
    ap.add_argument("--seconds", type=float, default=10.0)
    args = ap.parse_args()
    import wave as wavmod
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    root = tempfile.mkdtemp(prefix="ser_hostio_", dir=base)
    n = int(args.seconds * 16000)
    total = args.files * args.ranks
    try:
        free = shutil.disk_usage(root).free
        per_file = 2.0 * n + 4.0 * 1024 * 499 + 8192
        if total * per_file > 0.6 * free:
            total = int(0.6 * free / per_file) // (16 * args.ranks) * 16 * args.ranks
        os.makedirs(os.path.join(root, "wav"))
        rng = np.random.default_rng(7)
        clips = [(np.clip(0.1 * rng.standard_normal(n), -1, 1) * 32767).astype("<i2").tobytes() for _ in range(32)]
        for i in range(total):
            with wavmod.open(os.path.join(root, "wav", f"syn_{i:06d}.wav"), "wb") as wf:
                wf.setnchannels(1); wf.setsampwidth(2); wf.setframerate(16000); wf.writeframes(clips[i % 32])
        print(f"# {args.ranks} ranks over gloo, {total} PCM16 wav files of {args.seconds:.0f} s on {'tmpfs' if base else 'the local disk'} "
              f"({total // args.ranks} per rank), outputs [499, 1024] fp32 .pt (2.04 MB each) on the same file system; host: {os.cpu_count()} CPUs")
        print("# stub GPU: a batch of 16 takes `batch ms` (two in flight), i.e. the GPU side allows 16 / batch_ms per rank")
        import socket
        for batch_ms in [float(x) for x in args.batch_ms.split(",")]:
            for workers in [int(x) for x in args.workers.split(",")]:
                shutil.rmtree(os.path.join(root, "pt"), ignore_errors=True)
                s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
                ctx = mp.get_context("spawn")
                q = ctx.Queue()
                t0 = time.perf_counter()
                procs = [ctx.Process(target=worker, args=(r, args.ranks, port, root, workers, batch_ms, q)) for r in range(args.ranks)]
                for p in procs:
                    p.start()
                res = [q.get() for _ in procs]
                for p in procs:
                    p.join()
                wall_all = time.perf_counter() - t0
                res.sort()
                run = next(r[3] for r in res if r[3])
                written = len([f for f in os.listdir(os.path.join(root, "pt")) if f.endswith(".pt")])
                gpu_rate = 16.0 / (batch_ms * 1e-3)
                per_rank = [total / args.ranks / r[1] for r in res]
                cpu_util = sum(r[2] for r in res) / max(r[1] for r in res)
                print(f"batch {batch_ms:5.1f} ms  --num_workers {workers:2d}: {run['utt_per_s']:8.1f} files/s aggregate (driver's own clock; "
                      f"GPU side would allow {gpu_rate * args.ranks:8.1f} = {args.ranks} x {gpu_rate:6.1f}) = {100 * run['utt_per_s'] / (gpu_rate * args.ranks):5.1f} %; "
                      f"per rank {min(per_rank):6.1f} .. {max(per_rank):6.1f}; {written} files written; host CPU busy {cpu_util:5.1f} cores "
                      f"(process time / wall); incl. start-up {wall_all:5.1f} s", flush=True)
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
