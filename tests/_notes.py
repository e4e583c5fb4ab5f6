"""Measured parity numbers of a GPU test run, one JSON object per line in gpurun_out/parity_notes.jsonl (merged back from the
GPU box by gpurun; the round's digest is committed as profiles/r0N_parity_notes.jsonl).  Bars in the tests are set from these."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def note(name, obj):
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "parity_notes.jsonl"), "a") as fh:
        fh.write(json.dumps({"test": name, **obj}) + "\n")


def rel_l2(a, b):
    import torch
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))
