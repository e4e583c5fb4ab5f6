"""Evaluation caller: what runs right after the sampler -- same names and semantics as the reference.

Mirrors reference tester.py (`Tester.train` :57-134: sample with the EMA weights until `data_subset_num` unique images
exist; `remove_duplicates_in_batches` :148-160, `remove_duplicates_across_batches` :163-183, `get_nearest_neighbor_idx`
:186-201, `_compute_similarity` :140-145), sampler.py `get_nearest_neighbor` (:487-518), utils/datautils.py `normalize01`
(:211-222) and the data-mean histogram that feeds `sample_latent_shape='data'` (main_train_masked.py:60-87).

The reference compares images one pair at a time in Python loops (B x M cosine similarities, each its own kernel launch
and host sync).  Here every comparison of a batch is ONE fp32 contraction on the GPU -- unit-length rows (mdm_unit_rows)
times unit-length rows (mdm_gemm, exact-fp32 MFMA) -- and the greedy keep/drop decisions, which depend on the order of the
images, walk the resulting matrix on the host in the reference's order.  Image grids / plots are out of scope.
"""
from __future__ import annotations

import torch

from . import _lib, ops
from ._lib import F32, call, ptr, stream

COSINE_TH = 0.9          # tester.py:54


def _dev():
    if not torch.cuda.is_available():
        raise RuntimeError("mdm.evaluate needs a GPU and libmdm_hip.so; there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def normalize01(data):
    """Per-image min-max normalisation, NaN -> 0 (utils/datautils.py:211-222)."""
    x = data.to(_dev(), torch.float32).contiguous()
    n = x.shape[0]
    y = torch.empty_like(x)
    call("mdm_normalize01", ptr(x), ptr(y), n, x.numel() // n, stream())
    return y


def cosine_similarity_matrix(source, target):
    """-> S[M][B], S[m][b] = cos(target[m], source[b]) over the flattened images (`_compute_similarity`, tester.py:140-145).
    B and M are padded to multiples of 4 / kept as they are by zero rows (dropped again on return)."""
    dev = _dev()
    src = source.to(dev, torch.float32).reshape(source.shape[0], -1)
    tgt = target.to(dev, torch.float32).reshape(target.shape[0], -1)
    B, D = src.shape
    M = tgt.shape[0]
    assert tgt.shape[1] == D
    Dp, Bp = (D + 3) // 4 * 4, (B + 3) // 4 * 4
    a = torch.zeros(M, Dp, device=dev)
    b = torch.zeros(Bp, Dp, device=dev)
    a[:, :D] = tgt
    b[:B, :D] = src
    au, bu = torch.empty_like(a), torch.empty_like(b)
    call("mdm_unit_rows", ptr(a), ptr(au), M, Dp, 1e-8, stream())
    call("mdm_unit_rows", ptr(b), ptr(bu), Bp, Dp, 1e-8, stream())
    S = torch.empty(M, Bp, device=dev)
    ops.matmul(F32, 0, M, Bp, Dp, au, Dp, bu, Dp, S, Bp)
    return S[:, :B]


def col_argmax(S):
    """(values, indices) of the column maxima of S[M][B]; ties -> the first row, like `score.max(dim=0)`."""
    S = S.contiguous()
    M, B = S.shape
    val = torch.empty(B, device=S.device)
    idx = torch.empty(B, dtype=torch.int64, device=S.device)
    call("mdm_col_argmax", ptr(S), M, B, ptr(val), ptr(idx), stream())
    return val, idx


def _dataset_tensor(dataset):
    if torch.is_tensor(dataset):
        return dataset
    return torch.stack([dataset[i][0] for i in range(len(dataset))])


def get_nearest_neighbor_idx(source, dataset):
    """Index of the data image with the largest cosine similarity to each source image; data images are
    normalize01-ed first (tester.py:186-201)."""
    data = normalize01(_dataset_tensor(dataset))
    return col_argmax(cosine_similarity_matrix(source, data))[1]


def get_nearest_neighbor(source, dataset):
    """The nearest data image (un-normalised, as stored) for every source image (sampler.py:487-518, augment=False;
    the reference resizes both sides to 32x32 first -- a no-op at data_size 32, the benchmark shape)."""
    data = _dataset_tensor(dataset)
    idx = get_nearest_neighbor_idx(source, data)
    return data.to(source.device)[idx.to(source.device)]


def remove_duplicates_in_batches(current_batch, th=COSINE_TH):
    """Greedy, in order: an image is kept unless some ALREADY KEPT image of the batch has cosine similarity >= th
    (tester.py:148-160)."""
    S = cosine_similarity_matrix(current_batch, current_batch).cpu()
    kept = [0]
    for i in range(1, current_batch.shape[0]):
        if not bool((S[kept, i] >= th).any()):
            kept.append(i)
    return current_batch[kept]


def remove_duplicates_across_batches(unique_in_batch, previous_images, th=COSINE_TH):
    """Keep the images whose similarity to EVERY previously collected image is <= th (tester.py:163-183)."""
    if unique_in_batch.shape[0] == 0 or previous_images.shape[0] == 0:
        return unique_in_batch
    S = cosine_similarity_matrix(unique_in_batch, previous_images)        # [M prev][B]
    keep = ~(S > th).any(dim=0)
    return unique_in_batch[keep.to(unique_in_batch.device)]


def data_mean_histogram(dataset, args):
    """[hist_shape, hist_bin_edges, hist_mean_cum_sum] of the per-image (image-wise) or per-channel (channel-wise) means
    of the data set, `sample_num` bins per axis (main_train_masked.py:60-87); what `Sampler(dataset, args, Scheduler,
    dataset_hist)` takes for `sample_latent_shape='data'`.  Host arithmetic like upstream (once per run)."""
    if args.sample_latent_shape.lower() != "data":
        return [None, None, None]
    data = _dataset_tensor(dataset).to("cpu", torch.float32)
    if args.mean_area == "channel-wise":
        means = data.mean(dim=[2, 3])
    elif args.mean_area == "image-wise":
        means = data.mean(dim=[1, 2, 3]).unsqueeze(-1)
    else:
        raise UnboundLocalError("mean_area")
    hist, edges = torch.histogramdd(means, bins=args.sample_num, density=True)
    shape = hist.shape
    hist = torch.ravel(hist)
    hist = hist / torch.sum(hist)
    return [shape, edges, torch.cumsum(hist, dim=0)]


class Tester:
    """tester.py `Tester`: sample with the EMA weights until `args.data_subset_num` mutually distinct images exist."""

    def __init__(self, args, dataloader, dataset, model, ema_model, optimizer, lr_scheduler, accelerator):
        from .sampler import Sampler
        from .scheduler import Scheduler
        self.args, self.dataloader, self.dataset = args, dataloader, dataset
        self.model, self.ema_model, self.accelerator = model, ema_model, accelerator
        self.Scheduler = Scheduler(args, device=model.device)
        self.Sampler = Sampler(self.dataset, self.args, self.Scheduler, getattr(args, "dataset_hist", [None] * 3))
        self.cosine_similarity_th = COSINE_TH
        self.timesteps_used_epoch = None

    def train(self, epoch_start, epoch_length, resume_step, global_step, dirs, visualizer, max_rounds=1000):
        a = self.args
        a.updated_ddpm_num_steps = self.Scheduler.update_ddpm_num_steps(a.ddpm_num_steps)
        self.timesteps_used_epoch = self.Scheduler.get_timesteps_epoch(1, 10)          # tester.py:61
        total = torch.empty(0, a.out_channel, a.data_size, a.data_size)
        self.num_total_unique_images = []
        self.nearest_idx = []
        for _ in range(max_rounds):
            if len(total) >= a.data_subset_num:
                break
            self.ema_model.store(None)
            self.ema_model.copy_to(None)
            # (the plan is fetched behind copy_to: a bf16 model's sampling plan copies the weights it finds; see Trainer)
            net = self.model.sampling_plan(self.Sampler.local_sample_num(), getattr(a, "sample_precision", "f32_split")).eval()
            generated, _ = self.Sampler.sample(net, self.timesteps_used_epoch)
            self.ema_model.restore(None)
            uniq = remove_duplicates_in_batches(generated, self.cosine_similarity_th)
            uniq = remove_duplicates_across_batches(uniq, total, self.cosine_similarity_th)
            total = torch.cat((total, uniq.cpu()), dim=0)
            self.num_total_unique_images.append(total.shape[0])
            if uniq.shape[0] and self.dataset is not None:
                self.nearest_idx.append(get_nearest_neighbor_idx(uniq, self.dataset).cpu())
        self.model.train()
        self.total_unique_images = total
        return total
