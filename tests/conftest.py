import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

TESTS = os.path.join(ROOT, "tests")
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)          # tests/depth_envelope.py (helper shared by test_gpu_depth.py and the report generator)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Plain ``pytest`` on a box without a HIP device skips the gpu tests instead of failing inside them."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no HIP device visible (gpu tests run on the MI355X box: pytest -m gpu)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def built_library():
    """libserhip.so, built in-tree by hipcc (cross-compiles without a GPU)."""
    path = os.path.join(ROOT, "interspeech_ser_amd", "lib", "libserhip.so")
    if not os.path.isfile(path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "interspeech_ser_amd", "csrc"), "-j4"])
    return path


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def write_tiny_roberta_tokenizer(path):
    """A RobertaTokenizer that loads OFFLINE: byte-level vocab.json (4 specials + the 256 byte symbols + a few merges + <mask>,
    266 entries: fits the tiny RoBERTa fixture geometry's 300-row embedding) and merges.txt, in the layout
    ``RobertaTokenizer.from_pretrained(dir, local_files_only=True)`` reads -- the call the text drivers make
    (driver.hf_tokenize_fn; reference: preprocessing/preprocess_roberta.py:48-54).  The hub's real vocabulary cannot be fetched here."""
    import json
    os.makedirs(path, exist_ok=True)
    bs = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAC + 1)) + list(range(0xAE, 0xFF + 1))
    cs = bs[:]
    n = 0
    for b in range(256):
        if b not in bs:
            bs.append(b)
            cs.append(256 + n)
            n += 1
    vocab = {"<s>": 0, "<pad>": 1, "</s>": 2, "<unk>": 3}
    for c in cs:
        vocab[chr(c)] = len(vocab)
    merges = ["\u0120 t", "h e", "\u0120t he", "a n", "\u0120 a"]
    for m in merges:
        vocab["".join(m.split())] = len(vocab)
    vocab["<mask>"] = len(vocab)
    with open(os.path.join(path, "vocab.json"), "w", encoding="utf-8") as f:
        json.dump(vocab, f, ensure_ascii=False)
    with open(os.path.join(path, "merges.txt"), "w", encoding="utf-8") as f:
        f.write("#version: 0.2\n" + "\n".join(merges) + "\n")
    return path
