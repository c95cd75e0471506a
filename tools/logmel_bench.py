"""Time ser_logmel_whisper alone (GPU box): us per call, algorithmic HBM bytes (waveform in, log-mel out) per second against
the 8 TB/s peak, fp64 matrix-core rate of the DFT.  SER_HIP_LIB selects an alternative build for same-device A/B."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd import _lib as L
from interspeech_ser_amd.frontend import whisper_mel_filters
DEV = "cuda:0"
st = torch.cuda.current_stream().cuda_stream
mel = torch.from_numpy(whisper_mel_filters(128)).to(DEV)
for B in (8, 16):
    wav = (0.1 * torch.randn(B * 480000)).to(DEV)
    offs = torch.arange(B + 1, dtype=torch.int64, device=DEV) * 480000
    out = torch.empty(B, 128, 3000, device=DEV)
    work = torch.empty(L.lib.ser_workspace_bytes(L.WS_LOGMEL, B, 0, 0, 0, 1), dtype=torch.uint8, device=DEV)
    L.check(L.lib.ser_logmel_init(work.data_ptr(), B, st))
    ts = []
    for r in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            L.check(L.lib.ser_logmel_whisper(wav.data_ptr(), offs.data_ptr(), B, mel.data_ptr(), 128, out.data_ptr(), work.data_ptr(), st))
        e1.record()
        torch.cuda.synchronize()
        if r:
            ts.append(e0.elapsed_time(e1) / 5)
    ms = float(np.median(ts))
    algo = B * (480000 * 4 + 128 * 3000 * 4)                      # waveform read once, log-mel written once
    dft = B * 3000 * 400 * 402 * 2.0
    print(f"B={B}: {ms*1e3:8.1f} us/call  {algo/ms/1e9:7.3f} TB/s algorithmic (peak 8)  DFT {dft/ms/1e9:6.1f} TFLOP/s fp64 (matrix peak 78.6)  lib={os.path.basename(L.LIB_PATH)}", flush=True)
