"""End-to-end driver rate on the GPU box: N synthetic 16 kHz wav files (3-10 s, ragged) on tmpfs ->
preprocess_speech driver -> .pt files on tmpfs.  Includes decode, H2D, forward, D2H, torch.save."""
import os, sys, time, wave, shutil, tempfile
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from interspeech_ser_amd import driver

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mode = sys.argv[2] if len(sys.argv) > 2 else "bf16"
root = tempfile.mkdtemp(dir="/dev/shm")
wav_dir, out = os.path.join(root, "wav"), os.path.join(root, "pt")
os.makedirs(wav_dir)
rng = np.random.default_rng(4321)
secs = 0.0
for i in range(n):
    L = int(rng.uniform(3.0, 10.0) * 16000) if (len(sys.argv) > 3 and sys.argv[3] == "ragged") else 160000
    pcm = (np.clip(0.1 * rng.standard_normal(L), -1, 1) * 32767).astype("<i2")
    with wave.open(os.path.join(wav_dir, f"syn_{i:05d}.wav"), "wb") as wf:
        wf.setnchannels(1); wf.setsampwidth(2); wf.setframerate(16000); wf.writeframes(pcm.tobytes())
    secs += L / 16000
t0 = time.perf_counter()
driver.run_speech(["--ssl_type", "microsoft/wavlm-large", "--wav_dir", wav_dir, "--save_path", out, "--synthetic_weights",
                   "--use_n_layer", "--n_layer", "-1",        # the last state: every layer runs (the reference's default rule would stop after state 0)
                   "--mode", mode, "--batch_size", os.environ.get("BS", "16"), "--num_workers", os.environ.get("NW", "4"), "--timing"])
dt = time.perf_counter() - t0
print(f"E2E {n} files ({secs:.0f} s audio) incl. weight init: {dt:.1f} s; files written: {len(os.listdir(out))}")
shutil.rmtree(root)
