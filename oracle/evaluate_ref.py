"""CPU oracle: the evaluation caller (TEST INFRASTRUCTURE ONLY).

Own-words restatement of /root/reference/code/tester.py:140-201 (similarity, duplicate removal, nearest-neighbour index),
utils/datautils.py:211-222 (normalize01) and main_train_masked.py:60-87 (data-mean histogram), with the reference's
pair-at-a-time loops kept (small cases only).
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def normalize01(data):
    b = data.shape[0]
    hi = torch.amax(data, dim=(1, 2, 3)).reshape(b, 1, 1, 1)
    lo = torch.amin(data, dim=(1, 2, 3)).reshape(b, 1, 1, 1)
    return torch.nan_to_num((data - lo) / (hi - lo), nan=0.0)


def compute_similarity(source, target):
    """[M][B]: cosine similarity of every target image with every source image (tester.py:140-145)."""
    s, t = source.flatten(1), target.flatten(1)
    return F.cosine_similarity(s[None, :, :], t[:, None, :], dim=2)


def pair_similarity(a, b):
    return F.cosine_similarity(a.flatten(), b.flatten(), dim=0)       # tester.py:135-138


def remove_duplicates_in_batches(batch, th=0.9):
    kept = [batch[0]]
    for img in batch[1:]:
        if not any(pair_similarity(img, k) >= th for k in kept):
            kept.append(img)
    return torch.stack(kept)


def remove_duplicates_across_batches(unique_in_batch, previous, th=0.9):
    out = [img for img in unique_in_batch if not any(pair_similarity(img, p) > th for p in previous)]
    return torch.stack(out) if out else torch.empty(0, *unique_in_batch.shape[1:])


def get_nearest_neighbor_idx(source, data, chunk=7):
    score = torch.Tensor()
    for i in range(0, data.shape[0], chunk):                          # the DataLoader batches of tester.py:189-197
        score = torch.cat((score, compute_similarity(source, normalize01(data[i:i + chunk]))), dim=0)
    return score.max(dim=0)[1]


def data_mean_histogram(data, sample_num, mean_area):
    means = data.mean(dim=[2, 3]) if mean_area == "channel-wise" else data.mean(dim=[1, 2, 3]).unsqueeze(-1)
    hist, edges = torch.histogramdd(means, bins=sample_num, density=True)
    shape = hist.shape
    hist = torch.ravel(hist)
    hist = hist / torch.sum(hist)
    return [shape, edges, torch.cumsum(hist, dim=0)]
