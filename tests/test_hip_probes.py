"""GPU: the lane maps the kernels are written against (MFMA operand / accumulator layout,
ds_read_b64_tr_b16 gather).  If one of these fails the printed raw data says what the
hardware really does."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def test_mfma16_lane_map():
    from clip_event_amd import ops
    rng = np.random.default_rng(1)
    A = torch.from_numpy(rng.integers(-4, 5, size=(16, 32)).astype(np.float32))   # A[row][k]
    B = torch.from_numpy(rng.integers(-4, 5, size=(32, 16)).astype(np.float32))   # B[k][col]
    lanes = torch.arange(64)
    k = (8 * (lanes // 16)).unsqueeze(1) + torch.arange(8).unsqueeze(0)            # [64,8]
    a_fr = A[(lanes % 16).unsqueeze(1), k].to(torch.bfloat16).to(_dev())
    b_fr = B[k, (lanes % 16).unsqueeze(1)].to(torch.bfloat16).to(_dev())
    out = ops.probe_mfma(16, a_fr.contiguous(), b_fr.contiguous()).cpu()        # [64,4]
    D = A @ B
    exp = torch.stack([D[4 * (lanes // 16) + r, lanes % 16] for r in range(4)], dim=1)
    if not torch.equal(out, exp):
        print("mfma16 raw out:\n", out, "\nexpected D:\n", D)
    assert torch.equal(out, exp)


def test_mfma32_lane_map():
    from clip_event_amd import ops
    rng = np.random.default_rng(2)
    A = torch.from_numpy(rng.integers(-4, 5, size=(32, 16)).astype(np.float32))
    B = torch.from_numpy(rng.integers(-4, 5, size=(16, 32)).astype(np.float32))
    lanes = torch.arange(64)
    k = (8 * (lanes // 32)).unsqueeze(1) + torch.arange(8).unsqueeze(0)
    a_fr = A[(lanes % 32).unsqueeze(1), k].to(torch.bfloat16).to(_dev())
    b_fr = B[k, (lanes % 32).unsqueeze(1)].to(torch.bfloat16).to(_dev())
    out = ops.probe_mfma(32, a_fr.contiguous(), b_fr.contiguous()).cpu()        # [64,16]
    D = A @ B
    regs = torch.arange(16)
    rows = (regs % 4).unsqueeze(0) + 8 * (regs // 4).unsqueeze(0) + 4 * (lanes // 32).unsqueeze(1)   # [64,16]
    exp = D[rows, (lanes % 32).unsqueeze(1)]
    if not torch.equal(out, exp):
        print("mfma32 raw out:\n", out, "\nexpected D:\n", D)
    assert torch.equal(out, exp)


def test_tr16_gather():
    """ds_read_b64_tr_b16: within a 16-lane group, lane 4q+p supplies the address of block row q,
    columns 4p..4p+3; lane i receives column i of the 4 rows (row q in element q)."""
    from clip_event_amd import ops
    rows, cols = 16, 64                       # image[r][c] = 100*r + c, row stride 128 B
    img = (100 * torch.arange(rows).unsqueeze(1) + torch.arange(cols).unsqueeze(0)).to(torch.int16)
    lanes = torch.arange(64)
    g, li = lanes // 16, lanes % 16
    # group g reads block rows 4g..4g+3, columns 16..31
    r = 4 * g + li // 4
    c = 16 + 4 * (li % 4)
    off = (r * cols * 2 + c * 2).to(torch.int32)
    out = ops.probe_tr16(img.to(_dev()).contiguous(), off.to(_dev())).cpu()      # [64,4]
    exp = torch.stack([100 * (4 * g + q) + 16 + li for q in range(4)], dim=1).to(torch.int16)
    if not torch.equal(out, exp):
        print("tr16 raw out (lane: 4 values):")
        for l in range(64):
            print(l, out[l].tolist(), "expected", exp[l].tolist())
    assert torch.equal(out, exp)
