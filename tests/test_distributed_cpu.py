"""CPU, gloo, world_size 2: the data-parallel path (feature all-gather with reduce-scatter
backward, rank-offset labels, per-rank row blocks, gradient averaging) reproduces the
single-process loss and gradient on the concatenated batch.

The towers are HIP-only, so the features here are a differentiable stand-in (a linear map of
per-sample inputs) and the logits/criterion math is the CPU oracle's; what is under test is
the product's distributed module."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, W, port, B, K, out, paired):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=W)
    torch.set_num_threads(1)
    from oracle import clip_oracle as O
    from clip_event_amd import distributed as D
    g = torch.Generator().manual_seed(7)
    E = 16
    Wi = torch.randn(8, E, generator=g, requires_grad=True)      # shared "tower" parameters
    Wt = torch.randn(8, E, generator=g, requires_grad=True)
    ls = torch.tensor(2.0, requires_grad=True)
    xi_all = torch.randn(W * B, 8, generator=g)
    xt_all = torch.randn(W * B * K, 8, generator=g)
    xi, xt = xi_all[rank * B:(rank + 1) * B], xt_all[rank * B * K:(rank + 1) * B * K]
    fi, ft = xi @ Wi, xt @ Wt
    fi_all, ft_all = D.gather_feature_pair(fi, ft) if paired else (D.gather_features(fi), D.gather_features(ft))
    lpi, _ = O.logits_from_features(fi, ft_all, ls, True)          # local image rows x all texts
    _, lpt = O.logits_from_features(fi_all, ft, ls, True)          # local text rows x all images
    yi, yt, ip = D.global_labels(B, 1, K - 1, True, rank_=rank)
    ld = O.criterion_contrastive(lpi, lpt, yi, yt, ip, "ce")
    (ld["loss_i"] + ld["loss_t"]).backward()
    grads = [Wi.grad, Wt.grad, ls.grad.reshape(1)]
    for gr in grads:                                               # DDP-style mean of parameter gradients
        dist.all_reduce(gr)
        gr /= W
    red = D.reduce_dict({k: v.detach() for k, v in ld.items()})
    if rank == 0:
        torch.save({"Wi": Wi.grad, "Wt": Wt.grad, "ls": ls.grad, "loss_i": red["loss_i"], "loss_t": red["loss_t"]}, out)
    dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("paired", [True, False])
def test_global_batch_gradient_equals_single_process(tmp_path, paired):
    from oracle import clip_oracle as O
    W, B, K = 2, 3, 2
    out = str(tmp_path / "r0.pt")
    mp.spawn(_worker, args=(W, _free_port(), B, K, out, paired), nprocs=W, join=True)
    got = torch.load(out, weights_only=True)
    g = torch.Generator().manual_seed(7)
    E = 16
    Wi = torch.randn(8, E, generator=g, requires_grad=True)
    Wt = torch.randn(8, E, generator=g, requires_grad=True)
    ls = torch.tensor(2.0, requires_grad=True)
    xi_all = torch.randn(W * B, 8, generator=g)
    xt_all = torch.randn(W * B * K, 8, generator=g)
    lpi, lpt = O.logits_from_features(xi_all @ Wi, xt_all @ Wt, ls, True)
    yi, yt, ip = O.build_labels(W * B, 1, K - 1, True)
    ld = O.criterion_contrastive(lpi, lpt, yi, yt, ip, "ce")
    (ld["loss_i"] + ld["loss_t"]).backward()
    assert torch.allclose(got["loss_i"], ld["loss_i"].detach(), atol=1e-6)
    assert torch.allclose(got["loss_t"], ld["loss_t"].detach(), atol=1e-6)
    assert torch.allclose(got["Wi"], Wi.grad, atol=1e-5, rtol=1e-4)
    assert torch.allclose(got["Wt"], Wt.grad, atol=1e-5, rtol=1e-4)
    assert torch.allclose(got["ls"], ls.grad, atol=1e-5, rtol=1e-4)
