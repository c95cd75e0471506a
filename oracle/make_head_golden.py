"""Generate tests/golden/fusion_head_pins.npz -- run ONCE in the build container, output committed.

TEST INFRASTRUCTURE ONLY.  Pins oracle/fusion_head.py (the restated consumer of the feature files) to the
reference's own ``MultiModalEmotionClassifier`` and ``collate_fn``.  The reference script
(bin/train_cat_bimodal_lazy_1head.py) cannot be imported: it reads its config, builds data loaders and trains at module
level and imports packages this image does not have (benchmark.utils pulls librosa / parselmouth).  So the two
definitions are cut out of the source with ``ast`` -- nothing else of the file runs -- and executed here; what is
stored is DATA: the state-dict keys and shapes, and the logits of the reference class on seeded weights and a fixed
synthetic batch at the dimensions of BASELINE configs[4] (HuBERT-xlarge 1280 + RoBERTa-large 1024).

    python oracle/make_head_golden.py

The GPU box never runs this file and has no /root/reference.
"""
import ast
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import fusion_head as H           # noqa: E402

REF = "/root/reference/bin/train_cat_bimodal_lazy_1head.py"
FEAT1, FEAT2, SEED_W, SEED_X = 1280, 1024, 31, 32


def reference_definitions():
    tree = ast.parse(open(REF).read())
    wanted = {"MultiModalEmotionClassifier", "collate_fn"}
    body = [n for n in tree.body if isinstance(n, (ast.ClassDef, ast.FunctionDef)) and n.name in wanted]
    assert {n.name for n in body} == wanted
    ns = {}
    exec("import torch\nimport torch.nn as nn\nimport torch.nn.functional as F\nfrom torch.nn.utils.rnn import pad_sequence\n", ns)
    exec(compile(ast.Module(body=body, type_ignores=[]), REF, "exec"), ns)
    return ns["MultiModalEmotionClassifier"], ns["collate_fn"]


def main():
    RefHead, ref_collate = reference_definitions()
    ref = RefHead(features1_dim=FEAT1, features2_dim=FEAT2, fusion_hidden_dim=512, num_emotions=8, dropout=0.5).eval()
    shapes = {k: tuple(v.shape) for k, v in ref.state_dict().items()}
    weights = H.seeded_head_weights(shapes, SEED_W)
    ref.load_state_dict(weights, strict=True)
    batch = H.synthetic_batch(FEAT1, FEAT2, SEED_X)
    # the reference's collate on the same items gives the same padded batch
    items = [{"feat1": batch["feat1"][i, :t], "feat2": batch["feat2"][i], "label": batch["label"][i]} for i, t in enumerate((149, 499, 37))]
    rb = ref_collate(items)
    assert all(torch.equal(rb[k], batch[k]) for k in ("feat1", "feat2", "label"))
    with torch.no_grad():
        logits = ref(batch["feat1"], batch["feat2"])
    ours = H.MultiModalEmotionClassifier(FEAT1, FEAT2, 512, 8, 0.5).eval()
    assert list(ours.state_dict().keys()) == list(shapes.keys()), "state-dict keys differ from the reference class"
    ours.load_state_dict(weights, strict=True)
    with torch.no_grad():
        mine = ours(batch["feat1"], batch["feat2"])
    err = float((mine - logits).abs().max())
    print(f"restated head vs reference class: {len(shapes)} state-dict keys identical, logits max abs diff {err:.2e}")
    assert err < 1e-5
    keys = list(shapes.keys())
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "fusion_head_pins.npz"),
                        keys=np.array(keys), shapes=np.array([",".join(map(str, shapes[k])) for k in keys]),
                        logits=logits.numpy().astype(np.float32), feat1_dim=np.array(FEAT1), feat2_dim=np.array(FEAT2),
                        seed_weights=np.array(SEED_W), seed_batch=np.array(SEED_X),
                        weight_digest=np.array(float(sum(float(v.double().sum()) for v in weights.values()))))


if __name__ == "__main__":
    main()
