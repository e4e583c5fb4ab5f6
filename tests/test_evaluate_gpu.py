"""GPU parity of the evaluation caller (mdm/evaluate.py over csrc/eval.hip + the fp32 MFMA contraction) against outputs
of the reference's own `Tester` methods and `normalize01` (tests/golden/evaluate.npz), and a run of `Tester.train`."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from golden.make_golden import TINY, base_args  # noqa: E402


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_similarity_dedup_and_nearest_neighbour_vs_reference(golden):
    from mdm import evaluate as E
    g = golden("evaluate")
    data, batch, prev = T(g["ev_data"]), T(g["ev_batch"]), T(g["ev_prev"])
    const = torch.full((1, 3, 8, 8), 0.3)
    n01 = E.normalize01(torch.cat([data[:3], const])).cpu().numpy()
    assert np.allclose(n01, g["ev_norm01"], rtol=0, atol=1e-6) and np.array_equal(n01[3], np.zeros_like(n01[3]))
    S = E.cosine_similarity_matrix(batch, E.normalize01(data))
    assert tuple(S.shape) == (20, 12)
    assert np.allclose(S.cpu().numpy(), g["ev_sim"], rtol=0, atol=2e-6)
    uniq = E.remove_duplicates_in_batches(batch)
    assert np.array_equal(uniq.numpy(), g["ev_unique_in_batch"])
    assert np.array_equal(E.remove_duplicates_across_batches(uniq, prev).numpy(), g["ev_unique_across"])
    assert np.array_equal(E.get_nearest_neighbor_idx(batch, data).cpu().numpy(), g["ev_nn_idx"])
    nn = E.get_nearest_neighbor(batch, data)
    assert torch.equal(nn, data[T(g["ev_nn_idx"])])


def test_similarity_matrix_at_benchmark_size():
    """100 generated 3x32x32 images against 1000 data images: one contraction; against torch on the host."""
    from mdm import evaluate as E
    g = torch.Generator().manual_seed(3)
    src, tgt = torch.rand(100, 3, 32, 32, generator=g), torch.rand(1000, 3, 32, 32, generator=g)
    S = E.cosine_similarity_matrix(src, tgt).cpu()
    want = torch.nn.functional.cosine_similarity(src.flatten(1)[None], tgt.flatten(1)[:, None], dim=2)
    assert float((S - want).abs().max()) < 6e-6          # fp32 sums over 3072 elements in another order
    val, idx = E.col_argmax(S.cuda())
    wv, wi = want.max(dim=0)
    assert torch.equal(idx.cpu(), wi) or float((val.cpu() - wv).abs().max()) < 1e-6


def test_tester_collects_unique_samples():
    """tester.py `Tester.train`: EMA weights in, sample, drop duplicates, stop at data_subset_num."""
    import mdm
    from mdm.evaluate import Tester
    from oracle.unet_ref import random_params
    a = base_args(data_size=16, ddpm_schedule="linear", ddpm_num_steps=8, shift_type="noise_with_perturbation", sample_num=4,
                  sampling_mask_dependency="independent", momentum_adaptive="base_momentum", sample_latent_shape="uniform",
                  sample_history=False, rng_mode="device", seed=8, data_subset_num=6)
    model = mdm.UNet(TINY, N=4, H=16, W=16, dtype=mdm.BF16, params=random_params(TINY))
    ema = mdm.EMA(model)
    data = torch.rand(10, 3, 16, 16) * 2 - 1
    tst = Tester(a, None, data, model, ema, None, None, mdm.Accelerator())
    P0 = model.store.P.clone()
    total = tst.train(0, 1, 0, 0, None, None, max_rounds=6)
    assert total.shape[0] >= 6 and total.shape[1:] == (3, 16, 16) and bool(torch.isfinite(total).all())
    assert tst.num_total_unique_images == sorted(tst.num_total_unique_images)
    assert torch.equal(model.store.P, P0)                                  # training weights restored
    S = mdm.evaluate.cosine_similarity_matrix(total, total).cpu()
    S.fill_diagonal_(0)
    assert float(S.max()) <= 0.9 + 1e-5                                    # mutually distinct
