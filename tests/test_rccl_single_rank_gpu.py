"""RCCL API surface on the real backend with ONE rank (all this one-GPU box allows; the multi-rank semantics are
proven on gloo in test_distributed_cpu.py and, with the HIP model on two ranks, in test_ddp_gpu.py): every collective the data-parallel path issues
-- all_gather_into_tensor, reduce_scatter_tensor, all_reduce(AVG, async) on slices of the flat gradient buffer from a
side stream, all_reduce of the loss dict -- exists on "nccl" (= RCCL), accepts our tensors and leaves them unchanged
at world size 1."""
import os
import socket

import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_rccl_collectives_single_rank():
    from clip_event_amd import distributed as D
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        assert dist.get_backend() == "nccl"
        # feature exchange: forward all-gather, backward reduce-scatter (the W = 1 short cut is bypassed on purpose)
        x = torch.randn(6, 16, device=DEV, requires_grad=True)
        y = D._AllGatherFn.apply(x)
        assert torch.equal(y, x)
        (y * 2).sum().backward()
        assert torch.equal(x.grad, torch.full_like(x, 2.0))
        # gradient pieces: async AVG all-reduce of a slice of a flat buffer, issued from a side stream
        class _M:
            pass
        m = _M()
        m._flat_grad = torch.arange(4096, device=DEV, dtype=torch.float32)
        m._ranges = {"head": (0, 64), "visual": (64, 2048), "text": (2048, 4096)}
        m._layer_end = {"visual": {1: 1024}, "text": {1: 3072}}
        ref = m._flat_grad.clone()
        gs = D.GradSync.__new__(D.GradSync)
        gs.model, gs.pieces, gs.pending, gs.done, gs.progress = m, 2, [], set(), {}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            gs._reduce_range(64, 1024, async_op=True)
            gs._reduce_range(1024, 2048, async_op=True)
        gs._reduce_range(2048, 4096, async_op=True)
        gs._reduce_range(0, 64, async_op=False)
        for h in gs.pending:
            h.wait()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        assert torch.equal(m._flat_grad, ref)
        # logging reduction
        vals = torch.stack([torch.tensor(1.5, device=DEV), torch.tensor(2.5, device=DEV)])
        dist.all_reduce(vals)
        assert vals.tolist() == [1.5, 2.5]
        dist.barrier()
    finally:
        dist.destroy_process_group()
    assert not D.is_dist()
