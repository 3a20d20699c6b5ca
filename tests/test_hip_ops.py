"""GPU parity of the op-level HIP kernels (through the C ABI) against plain PyTorch fp32 on
the same bf16-rounded inputs.  Tolerances are stated per test."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _randn(rng, *shape, scale=1.0):
    return torch.from_numpy(rng.standard_normal(shape, dtype=np.float32) * scale)


def _report(name, got, exp):
    got, exp = got.double().cpu(), exp.double().cpu()
    err = (got - exp).abs().max().item()
    rel = ((got - exp).norm() / (exp.norm() + 1e-30)).item()
    print(f"[{name}] max_abs_err={err:.3e} rel_l2={rel:.3e} ref_rms={exp.pow(2).mean().sqrt().item():.3e}")
    return err, rel


GEMM_SHAPES = [(128, 128, 64), (256, 384, 128), (400, 768, 512), (77, 512, 2048), (20, 64, 128), (1000, 2304, 768),
               (130, 132, 72)]


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_nt_bf16_and_f32(M, N, K):
    from clip_event_amd import ops, _lib as L
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    a = _randn(rng, M, K).to(torch.bfloat16)
    b = _randn(rng, N, K, scale=K ** -0.5).to(torch.bfloat16)
    ref = a.float() @ b.float().t()
    out32 = ops.gemm_nt(a.to(DEV), b.to(DEV), L.EPI_F32)
    torch.cuda.synchronize()
    err, rel = _report(f"nt f32 {M}x{N}x{K}", out32, ref)
    assert rel < 1e-5          # fp32 accumulation of exact bf16 products: only summation order differs
    out16 = ops.gemm_nt(a.to(DEV), b.to(DEV), L.EPI_BF16)
    torch.cuda.synchronize()
    assert torch.equal(out16.cpu(), out32.cpu().to(torch.bfloat16))


@pytest.mark.parametrize("M,N,K", [(400, 512, 256), (1100, 768, 512), (2000, 520, 256), (12800, 768, 768), (3000, 2048, 512),
                                   (1300, 512, 192), (9000, 768, 256), (5000, 2304, 128), (20000, 3072, 64), (2000, 512, 64)])
def test_gemm_nt_epilogues(M, N, K):
    """Every fused epilogue at shapes that reach each kernel family (128^2 register-staged; the single-round loader-wave
    kernel at its three tile heights: 96 rows (1100, 2000, 1300), 128 (9000), 160 (12800, 3000), ragged M / N edges, K of
    1-3 tiles = shorter than its ring (2000 x 512 x 64: a single K tile); the multi-round 256-column LDS-DMA kernels with the epilogue-operand prefetch)."""
    from clip_event_amd import ops, _lib as L
    rng = np.random.default_rng(5 + M)
    a = _randn(rng, M, K).to(torch.bfloat16)
    b = _randn(rng, N, K, scale=K ** -0.5).to(torch.bfloat16)
    bias = _randn(rng, N)
    resid = _randn(rng, M, N)
    aux = _randn(rng, M, N).to(torch.bfloat16)
    acc = a.float() @ b.float().t()
    A, B = a.to(DEV), b.to(DEV)
    o = ops.gemm_nt(A, B, L.EPI_BIAS_BF16, bias=bias.to(DEV)).float().cpu()
    assert _report("bias_bf16", o, (acc + bias).to(torch.bfloat16).float())[1] < 3e-3
    o = ops.gemm_nt(A, B, L.EPI_BIAS_F32, bias=bias.to(DEV)).cpu()
    assert _report("bias_f32", o, acc + bias)[1] < 1e-5
    o = ops.gemm_nt(A, B, L.EPI_BIAS_RESID_F32, bias=bias.to(DEV), resid=resid.to(DEV)).cpu()
    assert _report("bias_resid", o, acc + bias + resid)[1] < 1e-5
    dact, g = ops.gemm_nt(A, B, L.EPI_BIAS_GELU, bias=bias.to(DEV))
    h = acc + bias
    sg = torch.sigmoid(1.702 * h)
    assert _report("gelu derivative", dact.float().cpu(), sg * (1 + 1.702 * h * (1 - sg)))[1] < 3e-3
    assert _report("gelu act", g.float().cpu(), h * sg)[1] < 3e-3
    colsum = torch.zeros(N, device=DEV)
    o = ops.gemm_nt(A, B, L.EPI_GELUGRAD_BF16, aux=aux.to(DEV), out2=colsum).float().cpu()
    want = acc * aux.float()                 # aux = the derivative the forward epilogue saved
    assert _report("gelugrad", o, want)[1] < 3e-3
    assert _report("gelugrad colsum", colsum.cpu(), want.sum(0))[1] < 3e-3


@pytest.mark.parametrize("variant", [5, 8, 32, 104, 160, 161, 162, 163, 164])
def test_gemm_nt_forced_tile_variants(variant):
    """Every NT tile family forced through ce_gemm_nt_tune (the per-shape policy reaches only some of them at test sizes):
    8-wave 256-column tiles (5, 8), 160x256x32 pairs (32), 160x128 pairs (104), three-stage ring (160), loader waves
    single-round (161) and persistent at tile heights chosen / 96 / 128 rows (162-164); ragged M and N edges, K = 3 tiles,
    the epilogues with extra operands."""
    from clip_event_amd import ops, _lib as L
    M, N, K = 2900, 1032, 192
    rng = np.random.default_rng(variant)
    a = _randn(rng, M, K).to(torch.bfloat16)
    b = _randn(rng, N, K, scale=K ** -0.5).to(torch.bfloat16)
    bias, resid, aux = _randn(rng, N), _randn(rng, M, N), _randn(rng, M, N).to(torch.bfloat16)
    acc = a.float() @ b.float().t()
    A, B = a.to(DEV), b.to(DEV)
    lib = L.lib()
    lib.ce_gemm_nt_tune(variant)
    try:
        o = ops.gemm_nt(A, B, L.EPI_F32).cpu()
        assert _report(f"v{variant} f32", o, acc)[1] < 1e-5
        o = ops.gemm_nt(A, B, L.EPI_BIAS_RESID_F32, bias=bias.to(DEV), resid=resid.to(DEV)).cpu()
        assert _report(f"v{variant} bias_resid", o, acc + bias + resid)[1] < 1e-5
        dact, g = ops.gemm_nt(A, B, L.EPI_BIAS_GELU, bias=bias.to(DEV))
        h = acc + bias
        sg = torch.sigmoid(1.702 * h)
        assert _report(f"v{variant} gelu derivative", dact.float().cpu(), sg * (1 + 1.702 * h * (1 - sg)))[1] < 3e-3
        assert _report(f"v{variant} gelu act", g.float().cpu(), h * sg)[1] < 3e-3
        colsum = torch.zeros(N, device=DEV)
        o = ops.gemm_nt(A, B, L.EPI_GELUGRAD_BF16, aux=aux.to(DEV), out2=colsum).float().cpu()
        want = acc * aux.float()
        assert _report(f"v{variant} gelugrad", o, want)[1] < 3e-3
        assert _report(f"v{variant} colsum", colsum.cpu(), want.sum(0))[1] < 3e-3
    finally:
        lib.ce_gemm_nt_tune(0)


@pytest.mark.parametrize("M,N,K", [(12800, 3072, 768), (10837, 1536, 512), (5000, 2304, 128), (4100, 2048, 256)])
def test_gemm_nt_persistent_xcd_owned_walk(M, N, K):
    """Multi-round launches of the persistent loader-wave kernel (more than 256 tiles: N = 3d / 4d at the towers' row
    counts) with the XCD-owned tile walk (gemm_common.hpp persist_walk): every output element is written exactly once --
    a tile skipped or visited twice shows as a wrong block -- for full chunks (12800 rows = 80 panels = 8 chunks of 10),
    ragged ones (10837 rows: 68 panels, chunks of 9 with a last chunk of 5; XCD ranges straddle chunk borders) and tile
    counts that are not multiples of 8."""
    from clip_event_amd import ops, _lib as L
    rng = np.random.default_rng(M + N)
    a = _randn(rng, M, K).to(torch.bfloat16)
    b = _randn(rng, N, K, scale=K ** -0.5).to(torch.bfloat16)
    bias = _randn(rng, N)
    aux = _randn(rng, M, N).to(torch.bfloat16)
    A, B = a.to(DEV), b.to(DEV)
    acc = (A.float() @ B.float().t()).cpu()        # fp32 reference product of the same bf16 operands
    lib = L.lib()
    for walk in (1000, 1001, 1004):                # XCD-owned chunks of tiles_m / 8, launch-wide (the default), chunks of 3
        lib.ce_gemm_nt_tune(walk)
        try:
            o = ops.gemm_nt(A, B, L.EPI_BIAS_F32, bias=bias.to(DEV)).cpu()
            assert _report(f"walk {walk} bias_f32", o, acc + bias)[1] < 2e-5
            assert torch.isfinite(o).all()
            colsum = torch.zeros(N, device=DEV)
            o = ops.gemm_nt(A, B, L.EPI_GELUGRAD_BF16, aux=aux.to(DEV), out2=colsum).float().cpu()
            want = acc * aux.float()
            assert _report(f"walk {walk} gelugrad", o, want)[1] < 3e-3
            assert _report(f"walk {walk} colsum", colsum.cpu(), want.sum(0))[1] < 3e-3
        finally:
            lib.ce_gemm_nt_tune(1001)


@pytest.mark.parametrize("M,N,K", [(12800, 3072, 768), (12800, 768, 768), (11137, 2048, 512), (10807, 512, 2048),
                                   (9000, 1536, 256), (3000, 512, 256), (33000, 768, 384), (2 * 128 * 86 + 5, 768, 640)])
@pytest.mark.parametrize("cus", [224, 97])
def test_gemm_nt_under_a_cu_budget(M, N, K, cus):
    """``ce_gemm_set_cu_budget``: the NT launch policies sized for fewer CUs than the chip has (what is left while RCCL's
    channel kernels hold some) -- one-round launches re-tiled or moved to the persistent kernel, persistent grids
    of ``cus`` workgroups with longer tile lists, ragged last panels -- against the fp32 product of the same bf16 operands.
    Every output element written exactly once, guard rows untouched; the budget is restored afterwards."""
    from clip_event_amd import ops, _lib as L
    rng = np.random.default_rng(M + N + K)
    a = _randn(rng, M, K).to(torch.bfloat16)
    b = _randn(rng, N, K, scale=K ** -0.5).to(torch.bfloat16)
    bias = _randn(rng, N)
    A, B = a.to(DEV), b.to(DEV)
    acc = (A.float() @ B.float().t()).cpu()
    lib = L.lib()
    assert lib.ce_gemm_set_cu_budget(cus) == 0
    try:
        guard = torch.full((M + 64, N), 7.0, device=DEV, dtype=torch.bfloat16)      # rows past M must stay untouched
        out = guard[:M]
        ops.gemm_nt(A, B, L.EPI_BF16, out=out)
        torch.cuda.synchronize()
        assert torch.equal(out.cpu(), acc.to(torch.bfloat16)) or _report("budget bf16", out.float().cpu(), acc.to(torch.bfloat16).float())[1] < 1e-3
        assert bool((guard[M:] == 7.0).all())
        o = ops.gemm_nt(A, B, L.EPI_BIAS_BF16, bias=bias.to(DEV)).float().cpu()
        assert _report("budget bias_bf16", o, (acc + bias).to(torch.bfloat16).float())[1] < 3e-3
        dact, g = ops.gemm_nt(A, B, L.EPI_BIAS_GELU, bias=bias.to(DEV))
        h = acc + bias
        sg = torch.sigmoid(1.702 * h)
        assert _report("budget gelu derivative", dact.float().cpu(), sg * (1 + 1.702 * h * (1 - sg)))[1] < 3e-3
        assert _report("budget gelu act", g.float().cpu(), h * sg)[1] < 3e-3
        assert torch.isfinite(g.float()).all() and torch.isfinite(dact.float()).all()
    finally:
        assert lib.ce_gemm_set_cu_budget(0) == 0
    assert lib.ce_gemm_set_cu_budget(7) != 0          # out of range: refused, budget unchanged


@pytest.mark.parametrize("M,N,K", [(12800, 3072, 768), (12800, 2304, 768), (11137, 2048, 512), (10807, 1536, 512),
                                   (9000, 1536, 256), (33000, 768, 384), (2 * 128 * 86 + 5, 768, 640), (5000, 2048, 192)])
@pytest.mark.parametrize("cus", [0, 61])
def test_gemm_nt_dynamic_tile_list(M, N, K, cus):
    """``ce_gemm_set_dynamic_tiles``: the persistent NT kernel with its tiles handed out by a device counter (first tile
    static, the rest fetched one tile ahead by one wave) -- every output element written exactly once, guard rows untouched,
    for the towers' multi-round shapes, three to twelve K iterations, ragged last panels, and a 61-workgroup grid where
    every workgroup walks many tiles; twice in a row on one stream (consecutive launches take consecutive counters of the
    stream's ring)."""
    from clip_event_amd import ops, _lib as L
    rng = np.random.default_rng(M + N + K)
    a = _randn(rng, M, K).to(torch.bfloat16)
    b = _randn(rng, N, K, scale=K ** -0.5).to(torch.bfloat16)
    bias = _randn(rng, N)
    A, B = a.to(DEV), b.to(DEV)
    acc = (A.float() @ B.float().t()).cpu()
    lib = L.lib()
    assert lib.ce_gemm_set_dynamic_tiles(1) == 0 and lib.ce_gemm_set_cu_budget(cus) == 0
    lib.ce_gemm_nt_tune(162)                    # the persistent kernel whatever the size policy says
    try:
        for rep in range(2):
            guard = torch.full((M + 64, N), 7.0, device=DEV, dtype=torch.bfloat16)
            out = guard[:M]
            ops.gemm_nt(A, B, L.EPI_BF16, out=out)
            torch.cuda.synchronize()
            assert torch.equal(out.cpu(), acc.to(torch.bfloat16)) or _report("dynamic bf16", out.float().cpu(), acc.to(torch.bfloat16).float())[1] < 1e-3
            assert bool((guard[M:] == 7.0).all())
        dact, g = ops.gemm_nt(A, B, L.EPI_BIAS_GELU, bias=bias.to(DEV))
        h = acc + bias
        sg = torch.sigmoid(1.702 * h)
        assert _report("dynamic gelu derivative", dact.float().cpu(), sg * (1 + 1.702 * h * (1 - sg)))[1] < 3e-3
        assert _report("dynamic gelu act", g.float().cpu(), h * sg)[1] < 3e-3
    finally:
        lib.ce_gemm_nt_tune(0)
        assert lib.ce_gemm_set_dynamic_tiles(-1) == 0 and lib.ce_gemm_set_cu_budget(0) == 0


TN_SHAPES = [(64, 128, 128), (256, 256, 384), (1000, 768, 512), (77 * 8, 512, 2048), (50, 64, 72), (12800, 768, 768),
             (4096, 512, 2048), (2120, 256, 256), (11137, 2048, 512)]      # the last three: the 256x256-tile kernel, ragged M


@pytest.mark.parametrize("M,Nn,Kk", TN_SHAPES)
def test_gemm_tn(M, Nn, Kk):
    from clip_event_amd import ops
    rng = np.random.default_rng(M + Nn + Kk)
    p = _randn(rng, M, Nn).to(torch.bfloat16)
    q = _randn(rng, M, Kk, scale=M ** -0.5).to(torch.bfloat16)
    ref = p.float().t() @ q.float()
    base = _randn(rng, Nn, Kk)
    for splits in (0, 1):
        out = base.clone().to(DEV)
        ops.gemm_tn(p.to(DEV), q.to(DEV), out, splits=splits)
        torch.cuda.synchronize()
        err, rel = _report(f"tn {M}x{Nn}x{Kk} splits={splits}", out.cpu() - base, ref)
        assert rel < 2e-5


@pytest.mark.parametrize("M,D", [(7, 128), (400, 768), (616, 512), (33, 1024), (5, 64), (37, 260), (21, 1000)])     # 260, 1000: a ragged last 256-column chunk
def test_layernorm_fwd_bwd(M, D):
    from clip_event_amd import ops
    rng = np.random.default_rng(M * 13 + D)
    x = (_randn(rng, M, D) * 2 + 0.5).requires_grad_(True)
    w = (1 + 0.1 * _randn(rng, D)).requires_grad_(True)
    b = (0.1 * _randn(rng, D)).requires_grad_(True)
    y_ref = torch.nn.functional.layer_norm(x, (D,), w, b, 1e-5)
    y, mean, rstd = ops.layernorm_fwd(x.detach().to(DEV), w.detach().to(DEV), b.detach().to(DEV))
    assert _report("ln fwd bf16", y.float().cpu(), y_ref.detach().to(torch.bfloat16).float())[1] < 3e-3
    y32, _, _ = ops.layernorm_fwd(x.detach().to(DEV), w.detach().to(DEV), b.detach().to(DEV), out_f32=True)
    assert _report("ln fwd f32", y32.cpu(), y_ref.detach())[1] < 2e-6
    dy = _randn(rng, M, D).to(torch.bfloat16)
    dx_in = _randn(rng, M, D)
    y_ref.backward(dy.float())
    dw = torch.zeros(D, device=DEV)
    db = torch.zeros(D, device=DEV)
    dxb = torch.empty(M, D, device=DEV, dtype=torch.bfloat16)
    dx = ops.layernorm_bwd(dy.to(DEV), x.detach().to(DEV), mean, rstd, w.detach().to(DEV), dw, db,
                           dx_in=dx_in.to(DEV), dxb=dxb)
    torch.cuda.synchronize()
    assert _report("ln dx", dx.cpu(), x.grad + dx_in)[1] < 1e-5
    assert torch.equal(dxb.cpu(), dx.cpu().to(torch.bfloat16))
    assert _report("ln dw", dw.cpu(), w.grad)[1] < 1e-5
    assert _report("ln db", db.cpu(), b.grad)[1] < 1e-5


def test_layernorm_row_gather_scatter():
    from clip_event_amd import ops
    rng = np.random.default_rng(3)
    B, Ltok, D = 6, 5, 128
    x = _randn(rng, B * Ltok, D)
    w = 1 + 0.1 * _randn(rng, D)
    b = 0.1 * _randn(rng, D)
    rows = (torch.arange(B) * Ltok + torch.tensor([0, 1, 4, 2, 3, 0])).to(torch.int32)
    y, mean, rstd = ops.layernorm_fwd(x.to(DEV), w.to(DEV), b.to(DEV), rows=rows.to(DEV), out_f32=True)
    ref = torch.nn.functional.layer_norm(x[rows.long()], (D,), w, b, 1e-5)
    assert _report("ln gather", y.cpu(), ref)[1] < 2e-6
    dy = _randn(rng, B, D)
    xs = x[rows.long()].clone().requires_grad_(True)
    torch.nn.functional.layer_norm(xs, (D,), w, b, 1e-5).backward(dy)
    dw = torch.zeros(D, device=DEV)
    db = torch.zeros(D, device=DEV)
    dx = ops.layernorm_bwd(dy.to(DEV), x.to(DEV), mean, rstd, w.to(DEV), dw, db, rows=rows.to(DEV))
    full = torch.zeros(B * Ltok, D)
    full[rows.long()] = xs.grad
    assert _report("ln scatter dx", dx.cpu(), full)[1] < 1e-5


def _attn_ref(qkv, B, L, H, causal):
    D = H * 64
    q, k, v = qkv.float().view(B, L, 3, H, 64).permute(2, 0, 3, 1, 4)      # [B,H,L,64] each
    s = (q @ k.transpose(-1, -2)) * 0.125
    if causal:
        s = s + torch.triu(torch.full((L, L), float("-inf")), diagonal=1)
    a = torch.softmax(s, dim=-1)
    o = (a @ v).permute(0, 2, 1, 3).reshape(B * L, D)
    return o, torch.logsumexp(s, dim=-1)


@pytest.mark.parametrize("B,L,H,causal", [(3, 50, 12, False), (2, 77, 8, True), (4, 5, 2, False), (3, 20, 2, True),
                                          (2, 128, 2, True), (1, 16, 1, False), (2, 33, 3, True)])
def test_attention_fwd_bwd(B, L, H, causal):
    from clip_event_amd import ops
    rng = np.random.default_rng(B * 100 + L)
    D = H * 64
    qkv = _randn(rng, B * L, 3 * D).to(torch.bfloat16)
    qkv_r = qkv.float().requires_grad_(True)
    o_ref, lse_ref = _attn_ref(qkv_r, B, L, H, causal)
    o, lse = ops.attention_fwd(qkv.to(DEV), B, L, H, causal)
    torch.cuda.synchronize()
    # bf16 P and bf16 output rounding: ~2^-8 relative
    assert _report(f"attn fwd o L={L}", o.float().cpu(), o_ref.detach())[1] < 1e-2
    assert _report("attn lse", lse.cpu().view(B, H, L), lse_ref.detach())[0] < 1e-3
    dout = _randn(rng, B * L, D).to(torch.bfloat16)
    o_ref.backward(dout.float())
    bg = torch.zeros(3 * D, device=DEV)
    dqkv = ops.attention_bwd(qkv.to(DEV), o, dout.to(DEV), lse, B, L, H, causal, bias_grad=bg)
    torch.cuda.synchronize()
    g = qkv_r.grad
    assert _report(f"attn in_proj bias grad L={L}", bg.cpu(), g.sum(0))[1] < 2e-2
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        assert _report(f"attn {name} L={L}", dqkv[:, sl].float().cpu(), g[:, sl])[1] < 2e-2


@pytest.mark.parametrize("B,L,H,causal", [(2, 197, 12, False), (1, 257, 4, False), (2, 577, 2, False), (3, 130, 2, True),
                                          (1, 200, 1, True), (2, 129, 3, False), (1, 640, 1, True)])
def test_attention_long_sequences(B, L, H, causal):
    """L > 128 (ViT-B/16: 197, ViT-L/14: 257 / 577): flash-style tiled kernels, same tolerances as the short ones."""
    from clip_event_amd import ops
    rng = np.random.default_rng(B * 1000 + L)
    D = H * 64
    qkv = _randn(rng, B * L, 3 * D).to(torch.bfloat16)
    qkv_r = qkv.float().requires_grad_(True)
    o_ref, lse_ref = _attn_ref(qkv_r, B, L, H, causal)
    o, lse = ops.attention_fwd(qkv.to(DEV), B, L, H, causal)
    torch.cuda.synchronize()
    assert _report(f"long attn fwd o L={L}", o.float().cpu(), o_ref.detach())[1] < 1e-2
    assert _report("long attn lse", lse.cpu().view(B, H, L), lse_ref.detach())[0] < 1e-3
    dout = _randn(rng, B * L, D).to(torch.bfloat16)
    o_ref.backward(dout.float())
    bg = torch.zeros(3 * D, device=DEV)
    dqkv = ops.attention_bwd(qkv.to(DEV), o, dout.to(DEV), lse, B, L, H, causal, bias_grad=bg)
    torch.cuda.synchronize()
    g = qkv_r.grad
    assert bool(torch.isfinite(dqkv.float()).all())
    assert _report(f"long attn in_proj bias grad L={L}", bg.cpu(), g.sum(0))[1] < 2e-2
    for name, sl in (("dq", slice(0, D)), ("dk", slice(D, 2 * D)), ("dv", slice(2 * D, 3 * D))):
        assert _report(f"long attn {name} L={L}", dqkv[:, sl].float().cpu(), g[:, sl])[1] < 2e-2


@pytest.mark.parametrize("lens,Lmax,H,causal", [([77, 10, 33, 1, 64, 77, 17], 77, 8, True), ([5, 50, 32, 31], 50, 3, False),
                                                ([128, 3, 96, 97], 128, 2, True), ([16], 20, 1, True)])
def test_attention_packed_variable_length(lens, Lmax, H, causal):
    """Packed batch (cu_seqlens): each sample equals the dense kernel's / the fp32 reference's result on its own rows."""
    from clip_event_amd import ops
    rng = np.random.default_rng(sum(lens) + Lmax)
    D, B = H * 64, len(lens)
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    R = int(cu[-1])
    qkv = _randn(rng, R, 3 * D).to(torch.bfloat16)
    dout = _randn(rng, R, D).to(torch.bfloat16)
    cu_d = torch.from_numpy(cu).to(DEV)
    o, lse = ops.attention_fwd(qkv.to(DEV), B, Lmax, H, causal, cu_seqlens=cu_d)
    bg = torch.zeros(3 * D, device=DEV)
    dqkv = ops.attention_bwd(qkv.to(DEV), o, dout.to(DEV), lse, B, Lmax, H, causal, bias_grad=bg, cu_seqlens=cu_d)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(o.float()).all()) and bool(torch.isfinite(dqkv.float()).all())
    lse = lse.cpu().view(B, H, Lmax)
    bias_ref = torch.zeros(3 * D)
    for b, ln in enumerate(lens):
        sl = slice(int(cu[b]), int(cu[b + 1]))
        x = qkv[sl].float().requires_grad_(True)
        o_ref, lse_ref = _attn_ref(x, 1, ln, H, causal)
        o_ref.backward(dout[sl].float())
        assert _report(f"packed attn o len={ln}", o[sl].float().cpu(), o_ref.detach())[1] < 1e-2
        assert _report("packed attn lse", lse[b, :, :ln], lse_ref.detach()[0])[0] < 1e-3
        assert _report(f"packed attn dqkv len={ln}", dqkv[sl].float().cpu(), x.grad)[1] < 2e-2
        bias_ref += x.grad.sum(0)
    assert _report("packed attn in_proj bias grad", bg.cpu(), bias_ref)[1] < 2e-2


@pytest.mark.parametrize("M,Nn,Kk", [(1000, 768, 512), (616, 512, 2048), (50, 64, 72), (12800, 768, 768), (77, 1536, 512)])
def test_gemm_tn_fused_bias_grad(M, Nn, Kk):
    from ctypes import c_int, c_long
    from clip_event_amd._lib import check, lib, ptr, stream
    rng = np.random.default_rng(M + 3 * Nn + Kk)
    p = _randn(rng, M, Nn).to(torch.bfloat16)
    q = _randn(rng, M, Kk, scale=M ** -0.5).to(torch.bfloat16)
    P, Q = p.to(DEV), q.to(DEV)
    out = torch.zeros(Nn, Kk, device=DEV)
    bg = torch.zeros(Nn, device=DEV)
    check(lib().ce_gemm_tn_bias(ptr(P), c_long(Nn), ptr(Q), c_long(Kk), c_int(M), c_int(Nn), c_int(Kk), ptr(out), c_long(Kk),
                                ptr(bg), c_int(0), stream()), "ce_gemm_tn_bias")
    torch.cuda.synchronize()
    assert _report(f"tn+bias {M}x{Nn}x{Kk} dW", out.cpu(), p.float().t() @ q.float())[1] < 2e-5
    assert _report("tn+bias db", bg.cpu(), p.float().sum(0))[1] < 2e-5


@pytest.mark.parametrize("M,N,K", [(2048, 512, 256), (6400, 3072, 128), (400, 512, 256)])
def test_gemm_nt_gelugrad_fused_colsum(M, N, K):
    """GELUGRAD epilogue with out2 = float[N]: += column sums of the output (the c_fc bias gradient)."""
    from ctypes import c_int, c_long
    from clip_event_amd._lib import check, lib, ptr, stream, EPI_GELUGRAD_BF16
    rng = np.random.default_rng(M + N + K)
    a = _randn(rng, M, K).to(torch.bfloat16)
    b = _randn(rng, N, K, scale=K ** -0.5).to(torch.bfloat16)
    aux = _randn(rng, M, N).to(torch.bfloat16)
    A, B, AUX = a.to(DEV), b.to(DEV), aux.to(DEV)
    out = torch.empty(M, N, device=DEV, dtype=torch.bfloat16)
    cs = torch.zeros(N, device=DEV)
    check(lib().ce_gemm_nt(ptr(A), c_long(K), ptr(B), c_long(K), c_int(M), c_int(N), c_int(K), c_int(EPI_GELUGRAD_BF16), None,
                           None, c_long(0), ptr(out), c_long(N), ptr(cs), c_long(N), ptr(AUX), c_long(N), stream()), "gemm")
    torch.cuda.synchronize()
    ref = (a.float() @ b.float().t()) * aux.float()          # aux = the QuickGELU' values the forward epilogue saved
    assert _report("gelugrad out", out.float().cpu(), ref)[1] < 3e-3
    assert _report("gelugrad colsum", cs.cpu(), ref.sum(0))[1] < 2e-3


def test_multi_transpose_table():
    """One launch transposes a table of bf16 matrices (W^T operand copies): 16-byte path and the ragged 2-byte path."""
    from ctypes import c_int
    from clip_event_amd._lib import TransposeJob, check, lib, ptr, stream
    rng = np.random.default_rng(42)
    shapes = [(768, 2304), (512, 512), (72, 64), (100, 36), (50, 7), (3072, 768), (8, 8)]
    srcs = [torch.from_numpy(rng.standard_normal(sh).astype(np.float32)).to(torch.bfloat16).to(DEV) for sh in shapes]
    dsts = [torch.full((sh[1], sh[0]), float("nan"), dtype=torch.bfloat16, device=DEV) for sh in shapes]
    jobs = (TransposeJob * len(shapes))()
    tiles = 0
    for i, (a, b) in enumerate(zip(srcs, dsts)):
        jobs[i].src, jobs[i].dst = a.data_ptr(), b.data_ptr()
        jobs[i].rows, jobs[i].cols, jobs[i].tile_start = a.shape[0], a.shape[1], tiles
        tiles += ((a.shape[0] + 63) // 64) * ((a.shape[1] + 63) // 64)
    tab = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(DEV)
    check(lib().ce_multi_transpose_bf16(ptr(tab), c_int(len(shapes)), c_int(tiles), stream()), "ce_multi_transpose_bf16")
    torch.cuda.synchronize()
    for a, b in zip(srcs, dsts):
        assert torch.equal(b.view(torch.int16).cpu(), a.t().contiguous().view(torch.int16).cpu()), tuple(a.shape)


# ---------------------------------------------------------------------------------------------------------------
# fp8 (OCP e4m3) operand path, BASELINE config 5

def _q8_ref(x_bf16: torch.Tensor):
    """torch restatement of ce_quant_rows_fp8 (== oracle.clip_oracle._q8): bytes, scales, dequantisable values."""
    xb = x_bf16.float()
    amax = xb.abs().amax(dim=-1, keepdim=True)
    m, k = torch.frexp(amax)
    e = 9 - k - (m > 0.875).to(k.dtype)
    live = amax >= 2.0 ** -100
    one = torch.ones_like(amax)
    inv = torch.where(live, torch.ldexp(one, e), one)
    q = (xb * inv).to(torch.float8_e4m3fn)
    return q.view(torch.uint8), torch.where(live, torch.ldexp(one, -e), one).flatten(), q.float()


@pytest.mark.parametrize("M,K", [(7, 128), (300, 768), (64, 1024), (33, 4096), (5, 72)])
def test_quant_rows_fp8_is_bit_exact(M, K):
    """Integer / byte work: the e4m3 bytes and the fp32 scales equal torch's float8_e4m3fn conversion exactly
    (OCP e4m3fn on gfx950, round to nearest even), including an all-zero row and a row with one huge outlier."""
    from clip_event_amd import ops
    rng = np.random.default_rng(M + K)
    x = _randn(rng, M, K) * torch.exp(_randn(rng, M, 1) * 3)
    x[0] = 0.0
    x[-1, 3] = 6.0e4
    xb = x.to(torch.bfloat16)
    q_ref, s_ref, _ = _q8_ref(xb)
    q, s = ops.quant_rows_fp8(xb.to(DEV))
    torch.cuda.synchronize()
    assert torch.equal(s.cpu(), s_ref)
    bad = (q.cpu() != q_ref).nonzero()
    detail = [(int(r), int(c), float(xb[r, c]), float(s_ref[r]), int(q.cpu()[r, c]), int(q_ref[r, c])) for r, c in bad[:8]]
    assert len(bad) == 0, f"{len(bad)} of {M * K} e4m3 bytes differ: (row, col, x, scale, hip byte, torch byte) {detail}"
    amax_q = (q_ref.view(torch.float8_e4m3fn).float().abs().amax(dim=-1))
    assert bool(((amax_q >= 224) | (xb.float().abs().amax(dim=-1) == 0)).all()) and float(amax_q.max()) <= 448


F8_SHAPES = [(128, 128, 128), (400, 768, 512), (77, 512, 2048), (1000, 2304, 768), (130, 136, 256), (2000, 1024, 4096)]


@pytest.mark.parametrize("M,N,K", F8_SHAPES)
def test_gemm_nt_fp8(M, N, K):
    """The fp8 MFMA product on exactly representable operands against fp32 PyTorch on the SAME dequantised values:
    products of two e4m3 numbers are exact in fp32, so only the summation order differs (1e-5 relative before the
    bf16 store; the bf16 output is then compared at bf16 rounding)."""
    from clip_event_amd import ops, _lib as L
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    a = _randn(rng, M, K).to(torch.bfloat16)
    b = _randn(rng, N, K, scale=K ** -0.5).to(torch.bfloat16)
    qa, sa, fa = _q8_ref(a)
    qb, sb, fb = _q8_ref(b)
    ref = (fa @ fb.t()) * sa[:, None] * sb[None, :]
    bias = _randn(rng, N)
    resid = _randn(rng, M, N)
    A8, SA, B8, SB = qa.to(DEV), sa.to(DEV), qb.to(DEV), sb.to(DEV)
    o = ops.gemm_nt_fp8(A8, SA, B8, SB, L.EPI_BIAS_RESID_F32, bias=bias.to(DEV), resid=resid.to(DEV)).cpu()
    torch.cuda.synchronize()
    err, rel = _report(f"fp8 nt {M}x{N}x{K} bias+resid f32", o, ref + bias + resid)
    assert rel < 1e-5
    o16 = ops.gemm_nt_fp8(A8, SA, B8, SB, L.EPI_BF16).float().cpu()
    assert _report("fp8 nt bf16", o16, ref.to(torch.bfloat16).float())[1] < 3e-3
    # and against the UNquantised product: what e4m3 costs (reported, loosely bounded)
    full = a.float() @ b.float().t()
    qerr = ((ref - full).norm() / full.norm()).item()
    print(f"   e4m3 quantisation error of the product vs bf16 operands: rel_l2={qerr:.3e}")
    assert qerr < 6e-2


def test_gemm_nt_fp8_epilogues():
    from clip_event_amd import ops, _lib as L
    M, N, K = 400, 512, 256
    rng = np.random.default_rng(5)
    a = _randn(rng, M, K).to(torch.bfloat16)
    b = _randn(rng, N, K, scale=K ** -0.5).to(torch.bfloat16)
    qa, sa, fa = _q8_ref(a)
    qb, sb, fb = _q8_ref(b)
    acc = (fa @ fb.t()) * sa[:, None] * sb[None, :]
    bias = _randn(rng, N)
    aux = _randn(rng, M, N).to(torch.bfloat16)
    A8, SA, B8, SB = qa.to(DEV), sa.to(DEV), qb.to(DEV), sb.to(DEV)
    o = ops.gemm_nt_fp8(A8, SA, B8, SB, L.EPI_BIAS_BF16, bias=bias.to(DEV)).float().cpu()
    assert _report("fp8 bias_bf16", o, (acc + bias).to(torch.bfloat16).float())[1] < 3e-3
    dact, g = ops.gemm_nt_fp8(A8, SA, B8, SB, L.EPI_BIAS_GELU, bias=bias.to(DEV))
    h = acc + bias
    sg = torch.sigmoid(1.702 * h)
    assert _report("fp8 gelu derivative", dact.float().cpu(), sg * (1 + 1.702 * h * (1 - sg)))[1] < 3e-3
    assert _report("fp8 gelu act", g.float().cpu(), h * sg)[1] < 3e-3
    colsum = torch.zeros(N, device=DEV)
    o = ops.gemm_nt_fp8(A8, SA, B8, SB, L.EPI_GELUGRAD_BF16, aux=aux.to(DEV), colsum=colsum).float().cpu()
    want = acc * aux.float()
    assert _report("fp8 gelugrad", o, want)[1] < 3e-3
    assert _report("fp8 gelugrad colsum", colsum.cpu(), want.sum(0))[1] < 2e-3


# ---------------------------------------------------------------------------------------------------------------
# fused contrastive head (no logits matrix)

@pytest.mark.parametrize("nq,nk,E,use_sel", [(70, 333, 128, True), (256, 256, 512, False), (33, 1000, 512, True),
                                             (512, 2560, 512, False), (40, 96, 768, True), (5, 7, 128, False)])
def test_fused_infonce_against_pytorch(nq, nk, E, use_sel):
    """ce_infonce_fwd/bwd against fp32 PyTorch autograd of  mean CE(s q^ k^T, labels)  on the selected query rows:
    loss and log-sum-exp 1e-5, gradients w.r.t. the RAW features and logit_scale 2e-5 relative (fp32 MFMA = exact
    fp32 FMA chains: summation order only).  Ragged sizes, query gather, labels anywhere in the key range."""
    from clip_event_amd.functional import InfoNCEFn
    rng = np.random.default_rng(nq * 31 + nk + E)
    rows_q = nq + 9 if use_sel else nq
    q = _randn(rng, rows_q, E).requires_grad_(True)
    k = _randn(rng, nk, E).requires_grad_(True)
    ls = torch.tensor(float(np.log(1 / 0.07)), requires_grad=True)
    sel = torch.from_numpy(rng.permutation(rows_q)[:nq].astype(np.int64)) if use_sel else None
    labels = torch.from_numpy(rng.integers(0, nk, size=rows_q).astype(np.int64))
    qs = q if sel is None else q.index_select(0, sel)
    ys = labels if sel is None else labels.index_select(0, sel)
    logits = ls.exp() * torch.nn.functional.normalize(qs, dim=-1) @ torch.nn.functional.normalize(k, dim=-1).t()
    ref = torch.nn.functional.cross_entropy(logits, ys)
    (ref * 1.3).backward()
    qd = q.detach().to(DEV).requires_grad_(True)
    kd = k.detach().to(DEV).requires_grad_(True)
    lsd = ls.detach().to(DEV).requires_grad_(True)
    loss = InfoNCEFn.apply(qd, kd, lsd, labels.to(DEV), None if sel is None else sel.to(DEV))
    (loss * 1.3).backward()
    torch.cuda.synchronize()
    print(f"[infonce {nq}x{nk}x{E}] loss {float(loss):.6f} (ref {float(ref):.6f})")
    assert abs(float(loss) - float(ref)) < 1e-5 * max(1.0, abs(float(ref)))
    assert _report("infonce dq", qd.grad.cpu(), q.grad)[1] < 2e-5
    assert _report("infonce dk", kd.grad.cpu(), k.grad)[1] < 2e-5
    assert abs(float(lsd.grad) - float(ls.grad)) < 2e-5 * max(1.0, abs(float(ls.grad)))


@pytest.mark.parametrize("n,T", [(256, 77), (5, 200), (1, 1), (1000, 64)])
def test_eot_rows_is_first_argmax(n, T):
    """ce_eot_rows (the EOT position that model_clip.py:415 takes with text.argmax(dim=-1)): first maximum of every row,
    rows with repeated maxima included; bit-exact against torch.argmax."""
    import ctypes
    from clip_event_amd import _lib as L
    rng = np.random.default_rng(n + T)
    ids = torch.from_numpy(rng.integers(0, 7, size=(n, T)).astype(np.int64))          # few distinct values: many ties
    ids[::3, rng.integers(0, T)] = 49407
    d = ids.to(DEV)
    rows = torch.empty(n, dtype=torch.int32, device=DEV)
    L.check(L.lib().ce_eot_rows(L.ptr(d), L.ptr(rows), ctypes.c_long(n), ctypes.c_int(T), L.stream()), "ce_eot_rows")
    want = torch.arange(n) * T + ids.argmax(dim=-1)
    assert torch.equal(rows.cpu().long(), want)


def test_quant_rows_fp8_multi_equals_single_launches():
    """ce_quant_rows_fp8_multi (one launch for a table of matrices: the per-step weight requantisation) writes the same
    bytes and scales as one ce_quant_rows_fp8 launch per matrix; row counts not multiples of 4, K from 64 to 4096."""
    import ctypes
    from clip_event_amd import ops, _lib as L
    rng = np.random.default_rng(7)
    shapes = [(768, 768), (13, 64), (3072, 1024), (1024, 4096), (5, 512)]
    mats = [(_randn(rng, m, k) * (10.0 ** rng.uniform(-3, 2))).to(torch.bfloat16).to(DEV) for m, k in shapes]
    want = [ops.quant_rows_fp8(x) for x in mats]
    outs = [(torch.zeros(m, k, dtype=torch.uint8, device=DEV), torch.zeros(m, dtype=torch.float32, device=DEV)) for m, k in shapes]
    jobs, groups = [], 0
    for x, (q, sc), (m, k) in zip(mats, outs, shapes):
        jobs.append(L.QuantJob(x.data_ptr(), q.data_ptr(), sc.data_ptr(), k, k, m, k, groups, 0))
        groups += (m + 3) // 4
    arr = (L.QuantJob * len(jobs))(*jobs)
    dev_jobs = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(DEV)
    L.check(L.lib().ce_quant_rows_fp8_multi(L.ptr(dev_jobs), ctypes.c_int(len(jobs)), ctypes.c_int(groups), L.stream()),
            "ce_quant_rows_fp8_multi")
    torch.cuda.synchronize()
    for (q, sc), (wq, wsc) in zip(outs, want):
        assert torch.equal(q, wq) and torch.equal(sc, wsc)



@pytest.mark.parametrize("M,D", [(5000, 768), (3000, 512), (40, 1024), (7, 64)])
def test_layernorm_bwd_partial_sums_and_fold(M, D):
    """ce_layernorm_bwd_partials + ce_layernorm_fold (per-workgroup partial sums of d gamma / d beta / the column sums of dx, added
    by one fold launch) against the atomic form ce_layernorm_bwd_t on the same inputs: same dx / bf16 copy bit for bit, parameter
    gradients equal up to summation order, accumulation (+=) into non-zero destinations, two launches folded by one call (each into
    its own destinations: the jobs of one fold call must not share a destination)."""
    import ctypes
    from clip_event_amd import _lib as L
    from clip_event_amd._lib import check, lib, ptr, stream
    rng = np.random.default_rng(M + D)
    x = _randn(rng, M, D).to(DEV)
    dy = _randn(rng, M, D).to(torch.bfloat16).to(DEV)
    din = _randn(rng, M, D).to(DEV)
    gamma = (1 + 0.1 * _randn(rng, D)).to(DEV)
    mean = x.mean(1).contiguous()
    rstd = (x.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
    cl = lib()

    def fresh():
        return (torch.empty(M, D, device=DEV), torch.empty(M, D, device=DEV, dtype=torch.bfloat16), torch.full((D,), 0.5, device=DEV),
                torch.full((D,), -0.25, device=DEV), torch.full((D,), 2.0, device=DEV))

    def run_atomic():
        dx, dxb, dw, db, dxs = fresh()
        check(cl.ce_layernorm_bwd_t(ptr(dy), L.T_BF16, ctypes.c_long(D), ptr(x), L.T_F32, ctypes.c_long(D), None, ptr(mean), ptr(rstd),
                                    ptr(gamma), ptr(din), L.T_F32, ptr(dx), L.T_F32, ctypes.c_long(D), ptr(dxb), ctypes.c_long(D),
                                    ptr(dw), ptr(db), ptr(dxs), None, M, D, stream()), "ln bwd")
        return dx, dxb, dw, db, dxs

    class Job(ctypes.Structure):
        _fields_ = [("partials", ctypes.c_void_p), ("dw", ctypes.c_void_p), ("db", ctypes.c_void_p), ("dxsum", ctypes.c_void_p),
                    ("blocks", ctypes.c_int), ("D", ctypes.c_int)]

    def run_fold():       # two launches (the second without the dx column sums), folded by ONE call into their own destinations
        blocks = cl.ce_layernorm_bwd_blocks(M, D)
        outs, jobs, keep = [], [], []
        for want in (1, 0):
            dx, dxb, dw, db, dxs = fresh()
            pbuf = torch.full((blocks, 3, D), float("nan"), device=DEV)      # every element the fold reads must have been written
            check(cl.ce_layernorm_bwd_partials(ptr(dy), L.T_BF16, ctypes.c_long(D), ptr(x), L.T_F32, ctypes.c_long(D), None, ptr(mean), ptr(rstd),
                                               ptr(gamma), ptr(din), L.T_F32, ptr(dx), L.T_F32, ctypes.c_long(D), ptr(dxb), ctypes.c_long(D),
                                               want, None, M, D, None, ctypes.c_long(0), None, ptr(pbuf), stream()), "ln bwd partials")
            jobs.append(Job(pbuf.data_ptr(), dw.data_ptr(), db.data_ptr(), dxs.data_ptr() if want else None, blocks, D))
            outs.append((dx, dxb, dw, db, dxs))
            keep.append(pbuf)
        check(cl.ce_layernorm_fold((Job * 2)(*jobs), 2, stream()), "ln fold")
        torch.cuda.synchronize()
        return outs

    a = run_atomic()
    f1, f2 = run_fold()
    torch.cuda.synchronize()
    for f in (f1, f2):
        assert torch.equal(a[0], f[0]) and torch.equal(a[1], f[1])
        for name, u, v in zip(("dgamma", "dbeta"), a[2:4], f[2:4]):
            assert torch.isfinite(v).all()
            assert _report(f"ln fold {name}", v.cpu(), u.cpu())[1] < 2e-5
    assert _report("ln fold dx column sums", f1[4].cpu(), a[4].cpu())[1] < 2e-5
    assert bool((f2[4] == 2.0).all())                  # not asked for: untouched


@pytest.mark.parametrize("M,D", [(400, 768), (616, 512), (33, 1024), (9, 2048), (5, 64), (37, 260), (21, 1000)])
def test_layernorm_fp16_stream_operands(M, D):
    """LayerNorm with the residual stream and the gradient stream in IEEE fp16 (ce_layernorm_*_t, model.stream16): the
    kernel must equal the fp32 computation on the SAME fp16-rounded inputs -- forward 2e-6 (fp32 output) / bf16 rounding,
    backward 1e-5 before the output rounding -- and the fp16 gradient stream must hold gradient * gscale (a power of two,
    so the scaling itself is exact; checked with an ulp-level bound on the stored halves)."""
    from clip_event_amd import ops
    rng = np.random.default_rng(M * 17 + D)
    GS = 65536.0
    gs = torch.tensor([GS], device=DEV)                  # the scale lives in device memory (ce_grad_scale writes it)
    x16 = (_randn(rng, M, D) * 2 + 0.5).to(torch.float16)
    w = 1 + 0.1 * _randn(rng, D)
    b = 0.1 * _randn(rng, D)
    x = x16.float().requires_grad_(True)
    wr, br = w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    y_ref = torch.nn.functional.layer_norm(x, (D,), wr, br, 1e-5)
    y, mean, rstd = ops.layernorm_fwd_t(x16.to(DEV), w.to(DEV), b.to(DEV), torch.bfloat16)
    assert _report("ln16 fwd bf16", y.float().cpu(), y_ref.detach().to(torch.bfloat16).float())[1] < 3e-3
    y32, _, _ = ops.layernorm_fwd_t(x16.to(DEV), w.to(DEV), b.to(DEV), torch.float32)
    assert _report("ln16 fwd f32", y32.cpu(), y_ref.detach())[1] < 2e-6
    yh, _, _ = ops.layernorm_fwd_t(x16.to(DEV), w.to(DEV), b.to(DEV), torch.float16)        # ln_pre writing an fp16 stream
    assert torch.equal(yh.cpu(), y32.cpu().to(torch.float16))
    # backward: dy bf16 (a GEMM output), gradient stream in (scaled fp16) and out (scaled fp16) + bf16 copy
    dy = (_randn(rng, M, D) * 1e-4).to(torch.bfloat16)
    din16 = (_randn(rng, M, D) * 1e-4 * GS).to(torch.float16)                                # holds gradient * GS
    y_ref.backward(dy.float())
    want = x.grad + din16.float() / GS
    dw, db, dxs = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    dxb = torch.empty(M, D, device=DEV, dtype=torch.bfloat16)
    dx16 = torch.empty(M, D, device=DEV, dtype=torch.float16)
    ops.layernorm_bwd_t(dy.to(DEV), x16.to(DEV), mean, rstd, w.to(DEV), dw, db, dx16, gscale=gs, dx_in=din16.to(DEV),
                        dxb=dxb, dxsum=dxs)
    dx32 = torch.empty(M, D, device=DEV, dtype=torch.float32)                                # same call, fp32 out: pre-rounding values
    dw2, db2 = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    ops.layernorm_bwd_t(dy.to(DEV), x16.to(DEV), mean, rstd, w.to(DEV), dw2, db2, dx32, gscale=gs, dx_in=din16.to(DEV))
    torch.cuda.synchronize()
    assert _report("ln16 dx (fp32 out)", dx32.cpu(), want)[1] < 1e-5
    assert torch.equal(dx16.cpu(), (dx32.cpu() * GS).to(torch.float16))                      # the stored stream = RNE(gradient * GS)
    assert torch.equal(dxb.cpu(), dx32.cpu().to(torch.bfloat16))                             # the GEMM operand copy is in true units
    assert _report("ln16 dw", dw.cpu(), wr.grad)[1] < 1e-5
    assert _report("ln16 db", db.cpu(), br.grad)[1] < 1e-5
    assert _report("ln16 dxsum", dxs.cpu(), want.sum(0))[1] < 1e-4
    # a gradient stream as dy (ln_pre's backward reads the tower's dx): fp16-scaled in, fp32 out
    dys = (_randn(rng, M, D) * 1e-4 * GS).to(torch.float16)
    x2 = x16.float().requires_grad_(True)
    torch.nn.functional.layer_norm(x2, (D,), w, b, 1e-5).backward(dys.float() / GS)
    dw3, db3 = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    out = torch.empty(M, D, device=DEV, dtype=torch.float32)
    ops.layernorm_bwd_t(dys.to(DEV), x16.to(DEV), mean, rstd, w.to(DEV), dw3, db3, out, gscale=gs)
    assert _report("ln16 dy=stream", out.cpu(), x2.grad)[1] < 1e-5


def test_fp16_stream_saturates_instead_of_overflowing():
    from clip_event_amd import ops
    x = torch.tensor([[7.0e4, -7.0e4, 65504.0, 1.0] * 4], device=DEV)
    h = ops.cast_t(x, torch.float16)
    assert torch.equal(h.cpu(), torch.tensor([[65504.0, -65504.0, 65504.0, 1.0] * 4], dtype=torch.float16))
    back = ops.cast_t(h, torch.float32, mul=0.5)
    assert torch.equal(back.cpu(), torch.tensor([[32752.0, -32752.0, 32752.0, 0.5] * 4]))
    assert torch.equal(ops.cast_t(torch.tensor([[1.0, 2.5, -3.0, 4.0]], device=DEV), torch.bfloat16).cpu(),
                       torch.tensor([[1.0, 2.5, -3.0, 4.0]], dtype=torch.bfloat16))


@pytest.mark.parametrize("n,amax", [(768 * 256, 3.1e-4), (1003, 900.0), (5, 1.0), (64, 0.0), (512 * 577, 2.0 ** -30)])
def test_grad_scale_is_the_power_of_two_below_target(n, amax):
    """ce_grad_scale: s = 2^k with s * max|x| in (target / 2, target]; exact integers in the exponent, so bit-exact; an
    all-zero gradient gives 1.  ce_cast_scaled applies / removes it exactly (power of two)."""
    from clip_event_amd import ops
    rng = np.random.default_rng(n)
    x = _randn(rng, n) * 1e-3
    x = x * (0.5 * amax / max(float(x.abs().max()), 1e-30)) if amax > 0 else torch.zeros(n)
    if amax > 0:
        x[rng.integers(n)] = -amax
    s = float(ops.grad_scale(x.to(DEV), 1024.0).cpu())
    if amax == 0:
        assert s == 1.0
        return
    assert s == 2.0 ** np.floor(np.log2(1024.0 / amax)) and 512.0 < s * amax <= 1024.0
    if n % 4 == 0:
        sc = torch.tensor([s], device=DEV)
        h = ops.cast_scaled(x.to(DEV), torch.float16, sc)
        assert torch.equal(h.cpu(), (x * s).to(torch.float16))
        back = ops.cast_scaled(h, torch.float32, sc, divide=True)
        assert torch.equal(back.cpu(), h.cpu().float() / s)


@pytest.mark.parametrize("M,N,K", [(400, 512, 256), (1100, 768, 512), (12800, 768, 768), (3000, 2048, 512), (5000, 2304, 128),
                                   (256, 768, 3072), (2000, 520, 256), (2000, 512, 64)])
def test_gemm_nt_fp16_residual_epilogue(M, N, K):
    """CE_EPI_BIAS_RESID_F16 (the residual add on an fp16 stream) in every NT kernel family the shapes reach (128^2, skinny,
    loader-wave single-round at its three tile heights, persistent): out = RNE_fp16(resid + acc + bias) of the fp32 sum,
    i.e. equal to the fp32-residual epilogue's result rounded to fp16, up to the summation order inside acc."""
    from clip_event_amd import ops, _lib as L
    rng = np.random.default_rng(11 + M + N)
    a = _randn(rng, M, K).to(torch.bfloat16)
    b = _randn(rng, N, K, scale=K ** -0.5).to(torch.bfloat16)
    bias = _randn(rng, N)
    resid = (_randn(rng, M, N) * 3).to(torch.float16)
    A, B = a.to(DEV), b.to(DEV)
    want32 = (A.float() @ B.float().t()).cpu() + bias + resid.float()
    o = ops.gemm_nt(A, B, L.EPI_BIAS_RESID_F16, bias=bias.to(DEV), resid=resid.to(DEV))
    assert o.dtype == torch.float16
    assert _report("resid_f16", o.float().cpu(), want32)[1] < 4e-4           # fp16 rounding: 2^-11 relative per element
    ref32 = ops.gemm_nt(A, B, L.EPI_BIAS_RESID_F32, bias=bias.to(DEV), resid=resid.float().to(DEV)).cpu()
    assert torch.equal(o.cpu(), ref32.to(torch.float16))                    # same accumulator, same sum, one rounding


def test_token_embed_fp16_stream():
    from clip_event_amd import _lib as L
    from clip_event_amd._lib import check, lib, ptr, stream
    from ctypes import c_int, c_long
    rng = np.random.default_rng(4)
    n, T, D, V = 6, 20, 128, 300
    ids = torch.from_numpy(rng.integers(0, V, (n, T))).to(DEV)
    table = _randn(rng, V, D).to(DEV)
    pos = _randn(rng, T, D).to(DEV)
    x32 = torch.empty(n * T, D, device=DEV)
    x16 = torch.empty(n * T, D, device=DEV, dtype=torch.float16)
    check(lib().ce_token_embed_t(ptr(ids), None, ptr(table), ptr(pos), ptr(x32), c_int(L.T_F32), c_long(n * T), c_int(T), c_int(D),
                                 c_int(V), stream()), "ce_token_embed_t")
    check(lib().ce_token_embed_t(ptr(ids), None, ptr(table), ptr(pos), ptr(x16), c_int(L.T_F16), c_long(n * T), c_int(T), c_int(D),
                                 c_int(V), stream()), "ce_token_embed_t")
    ref = table[ids.flatten()] + pos.repeat(n, 1)
    assert torch.equal(x32, ref) and torch.equal(x16, ref.to(torch.float16))


def test_scatter_rows_zero_fill():
    """ce_scatter_rows_zero = memset + row scatter in one pass (pruned last block's backward): rows not named are zero,
    named rows carry the source rows, for fp32 / fp16 / bf16 row widths and selections at both ends."""
    from clip_event_amd._lib import check, lib, ptr, stream
    from ctypes import c_int, c_long
    rng = np.random.default_rng(9)
    for M, n, D, dt in ((50 * 7, 7, 128, torch.float32), (1000, 13, 768, torch.float16), (77, 77, 64, torch.bfloat16), (9, 1, 8, torch.float32)):
        rows = torch.from_numpy(np.sort(rng.choice(M, n, replace=False)).astype(np.int32))
        if n > 1:
            rows[0], rows[-1] = 0, M - 1
        src = _randn(rng, n, D).to(dt)
        dst = torch.full((M, D), 7.0, dtype=dt, device=DEV)
        es = src.element_size()
        src_d, rows_d = src.to(DEV), rows.to(DEV)            # keep the device copies alive across the asynchronous launch
        check(lib().ce_scatter_rows_zero(ptr(src_d), c_long(D * es), ptr(dst), c_long(D * es), ptr(rows_d), c_int(n), c_int(M),
                                         c_int(D * es), stream()), "ce_scatter_rows_zero")
        want = torch.zeros(M, D, dtype=dt)
        want[rows.long()] = src
        assert torch.equal(dst.cpu(), want)


def _mx_quant_ref(x: torch.Tensor):
    """Torch restatement of ce_quant_mx_fp8: per 32-column block the power of two that maps the block's amax into (224, 448]."""
    xb = x.float()
    M, K = xb.shape
    blk = xb.view(M, K // 32, 32)
    amax = blk.abs().amax(dim=-1, keepdim=True)
    m, k = torch.frexp(amax)
    e = 9 - k - (m > 0.875).to(k.dtype)
    live = amax >= 2.0 ** -100
    one = torch.ones_like(amax)
    inv = torch.where(live, torch.ldexp(one, e), one)
    q = (blk * inv).to(torch.float8_e4m3fn).view(M, K)
    s8 = torch.where(live, 127 - e, torch.full_like(e, 127)).to(torch.uint8).view(M, K // 32)
    deq = (q.float().view(M, K // 32, 32) * torch.where(live, torch.ldexp(one, -e), one)).view(M, K)
    return q.view(torch.uint8), s8, deq


@pytest.mark.parametrize("M,K", [(7, 128), (300, 768), (64, 1024), (33, 4096), (5, 32), (9, 8192)])
def test_quant_mx_fp8_is_bit_exact(M, K):
    """MX block quantisation (one E8M0 scale per 32 values): e4m3 bytes and scale bytes equal the torch restatement bit for
    bit (power-of-two scaling is exact), incl. all-zero blocks (scale 2^0) and blocks 10 orders of magnitude apart in one row."""
    from clip_event_amd import ops
    rng = np.random.default_rng(M + K)
    x = _randn(rng, M, K) * torch.from_numpy(10.0 ** rng.uniform(-6, 4, (M, K // 32))).float().repeat_interleave(32, dim=1)
    x[0, :32] = 0.0
    x = x.to(torch.bfloat16)
    q, s8 = ops.quant_mx_fp8(x.to(DEV))
    q_ref, s_ref, _ = _mx_quant_ref(x)
    assert torch.equal(s8.cpu(), s_ref)
    assert torch.equal(q.cpu(), q_ref)
    assert int(s8[0, 0]) == 127 and int(q[0, :32].max()) == 0


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 520, 256), (1100, 768, 1024), (12800, 768, 768 + 256), (2000, 3072, 1024)])
def test_gemm_nt_mx8_block_scales(M, N, K):
    """e4m3 GEMM with E8M0 block scales applied by v_mfma_scale_f32_16x16x128_f8f6f4 itself: equals the fp32 product of the
    DEQUANTISED operands (1e-5: fp32 accumulation of exact products), for operands whose blocks differ by orders of magnitude --
    a wrong lane / block / byte assignment of a scale is an error of order 1.  Lane map: tools/diag/probe_mfma_scale.py."""
    from clip_event_amd import ops, _lib as L
    rng = np.random.default_rng(M + N + K)
    a = (_randn(rng, M, K) * torch.from_numpy(2.0 ** rng.integers(-6, 7, (M, K // 32))).float().repeat_interleave(32, dim=1)).to(torch.bfloat16)
    b = (_randn(rng, N, K, scale=K ** -0.5) * torch.from_numpy(2.0 ** rng.integers(-4, 5, (N, K // 32))).float().repeat_interleave(32, dim=1)).to(torch.bfloat16)
    a8, sa8 = ops.quant_mx_fp8(a.to(DEV))
    b8, sb8 = ops.quant_mx_fp8(b.to(DEV))
    _, _, a_deq = _mx_quant_ref(a)
    _, _, b_deq = _mx_quant_ref(b)
    ref = (a_deq.to(DEV).double() @ b_deq.to(DEV).double().t()).float().cpu()      # exact products, fp64 sums
    bias = _randn(rng, N)
    o = ops.gemm_nt_mx8(a8, sa8, b8, sb8, L.EPI_BIAS_RESID_F32, bias=bias.to(DEV), resid=torch.zeros(M, N, device=DEV)).cpu()
    assert _report("mx8 bias_resid_f32", o, ref + bias)[1] < 3e-5       # fp32 accumulation over blocks 2^12 apart
    o16 = ops.gemm_nt_mx8(a8, sa8, b8, sb8, L.EPI_BF16).float().cpu()
    assert _report("mx8 bf16", o16, ref.to(torch.bfloat16).float())[1] < 3e-3


@pytest.mark.parametrize("M,D", [(400, 768), (616, 512), (33, 1024), (7, 128)])
def test_layernorm_emits_the_fp8_operand_copy(M, D):
    """fp8 operand path: LayerNorm forward / backward write the e4m3 copy (+ per-row scale) of their bf16 output from the
    registers that hold the row (ce_layernorm_fwd_q8 / _bwd_q8) -- bit for bit what ce_quant_rows_fp8 makes of that output, so
    the consuming GEMM's quantisation pass can be dropped without changing a result."""
    from clip_event_amd import ops, _lib as L
    from clip_event_amd._lib import check, lib, ptr, stream
    from ctypes import c_float, c_int, c_long
    rng = np.random.default_rng(M + D)
    x = (_randn(rng, M, D) * 2 + 0.5).to(torch.float16).to(DEV)
    w = (1 + 0.1 * _randn(rng, D)).to(DEV)
    b = (0.1 * _randn(rng, D)).to(DEV)
    y = torch.empty(M, D, device=DEV, dtype=torch.bfloat16)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    q8 = torch.empty(M, D, device=DEV, dtype=torch.uint8)
    qs = torch.empty(M, device=DEV)
    check(lib().ce_layernorm_fwd_q8(ptr(x), c_int(L.T_F16), c_long(D), None, ptr(w), ptr(b), ptr(y), c_int(L.T_BF16), c_long(D), ptr(mean),
                                    ptr(rstd), c_int(M), c_int(D), c_float(1e-5), ptr(q8), c_long(D), ptr(qs), stream()), "ln_fwd_q8")
    q_ref, s_ref = ops.quant_rows_fp8(y)
    assert torch.equal(q8, q_ref) and torch.equal(qs, s_ref)
    dy = (_randn(rng, M, D) * 1e-3).to(torch.bfloat16).to(DEV)
    gs = torch.tensor([1024.0], device=DEV)
    din = (_randn(rng, M, D) * 1e-3 * 1024).to(torch.float16).to(DEV)
    dx = torch.empty(M, D, device=DEV, dtype=torch.float16)
    dxb = torch.empty(M, D, device=DEV, dtype=torch.bfloat16)
    dw, db = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    check(lib().ce_layernorm_bwd_q8(ptr(dy), c_int(L.T_BF16), c_long(D), ptr(x), c_int(L.T_F16), c_long(D), None, ptr(mean), ptr(rstd), ptr(w),
                                    ptr(din), c_int(L.T_F16), ptr(dx), c_int(L.T_F16), c_long(D), ptr(dxb), c_long(D), ptr(dw), ptr(db), None,
                                    ptr(gs), c_int(M), c_int(D), ptr(q8), c_long(D), ptr(qs), stream()), "ln_bwd_q8")
    q_ref, s_ref = ops.quant_rows_fp8(dxb)
    assert torch.equal(q8, q_ref) and torch.equal(qs, s_ref)


@pytest.mark.parametrize("nI,K,E,partial_sel", [(256, 1, 512, False), (8, 3, 512, False), (37, 3, 64, True), (200, 5, 768, True), (1, 1, 128, False)])
def test_small_head_against_pytorch(nI, K, E, partial_sel):
    """The three-launch head (csrc/head_small.hip: normalise, logits over the batch, 'ce' criterion with index_pos, and the whole
    backward down to the raw features) against fp32 PyTorch autograd of model_clip.py:496-521 + :633-662: both losses, both
    feature gradients and the logit_scale gradient, for config 2's shape, hard negatives (index_pos selects every K-th text
    row), ragged sizes, a selection that skips rows, a single image, and unequal upstream weights of the two losses."""
    from clip_event_amd.functional import SmallHeadFn, small_head_ok
    rng = np.random.default_rng(nI * 7 + K + E)
    nT = nI * K
    fi0 = _randn(rng, nI, E)
    ft0 = _randn(rng, nT, E)
    ls0 = torch.tensor(float(np.log(1 / 0.07)))
    sel = torch.arange(0, nT, K)
    if partial_sel:
        sel = sel[torch.from_numpy(rng.permutation(nI)[: max(1, nI // 2)].copy()).sort().values]
    yi = torch.from_numpy(rng.integers(0, nT, size=nI))
    yt = torch.from_numpy(rng.integers(0, nI, size=nT))              # one label per text row; the selected rows' are used
    w_i, w_t = 1.0, 0.37

    def ref():
        fi, ft, ls = fi0.clone().requires_grad_(True), ft0.clone().requires_grad_(True), ls0.clone().requires_grad_(True)
        In, Tn = fi / fi.norm(dim=-1, keepdim=True), ft / ft.norm(dim=-1, keepdim=True)
        lpi = ls.exp() * In @ Tn.t()
        lpt = ls.exp() * Tn @ In.t()
        li = torch.nn.functional.cross_entropy(lpi, yi)
        lt = torch.nn.functional.cross_entropy(lpt.index_select(0, sel), yt.index_select(0, sel))
        (w_i * li + w_t * lt).backward()
        return li.item(), lt.item(), fi.grad, ft.grad, ls.grad

    fi, ft, ls = fi0.to(DEV).requires_grad_(True), ft0.to(DEV).requires_grad_(True), ls0.to(DEV).requires_grad_(True)
    assert small_head_ok(fi, ft, sel)
    li, lt = SmallHeadFn.apply(fi, ft, ls, yi.to(DEV), yt.to(DEV), sel.to(DEV))
    (w_i * li + w_t * lt).backward()
    torch.cuda.synchronize()
    rli, rlt, rfi, rft, rls = ref()
    print(f"loss_i {float(li):.6f} / {rli:.6f}  loss_t {float(lt):.6f} / {rlt:.6f}")
    assert abs(float(li) - rli) < 2e-5 * max(1.0, abs(rli)) and abs(float(lt) - rlt) < 2e-5 * max(1.0, abs(rlt))
    assert _report("small head dfi", fi.grad.cpu(), rfi)[1] < 2e-5
    assert _report("small head dft", ft.grad.cpu(), rft)[1] < 2e-5
    assert abs(float(ls.grad) - float(rls)) < 2e-5 * max(1.0, abs(float(rls)))


@pytest.mark.parametrize("M,N,K,plan", [(12800, 3072, 768, (64, 4)), (11137, 2048, 512, None), (10807, 1536, 512, None),
                                        (18200, 3072, 768, None), (12800, 2304, 768, (0, 0)), (7700, 2048, 256, None)])
@pytest.mark.parametrize("dynamic", [0, 1])
def test_gemm_nt_two_tile_heights(M, N, K, plan, dynamic):
    """The persistent NT kernel with TWO tile heights in one launch (160-row panels first, the rest in panels of 32 TS rows,
    split by `launch_nt` so that every workgroup's list costs the same): bit-identical to the one-height persistent kernel
    (the same MFMA sequence per output element) for all four epilogues that have the form, guard rows untouched, with the
    static and the dynamic tile list.  The plan the launcher took is read back (`ce_gemm_nt_last_plan`): 12800 x 3072 = 64
    panels of 160 + 20 of 128 (3 rounds + 1 instead of 3.75 rounds of 160); 12800 x 2304 has no better split."""
    from ctypes import byref, c_int
    from clip_event_amd import ops, _lib as L
    rng = np.random.default_rng(M + N + K)
    a = _randn(rng, M, K).to(torch.bfloat16)
    b = _randn(rng, N, K, scale=K ** -0.5).to(torch.bfloat16)
    bias = _randn(rng, N)
    aux = _randn(rng, M, N).to(torch.bfloat16)
    A, B, AUX, BIAS = a.to(DEV), b.to(DEV), aux.to(DEV), bias.to(DEV)
    lib = L.lib()
    assert lib.ce_gemm_set_dynamic_tiles(dynamic) == 0

    def run_all():
        out = {}
        guard = torch.full((M + 64, N), 7.0, device=DEV, dtype=torch.bfloat16)
        ops.gemm_nt(A, B, L.EPI_BF16, out=guard[:M])
        out["bf16"] = guard[:M].clone()
        tall, ts = c_int(-1), c_int(-1)
        lib.ce_gemm_nt_last_plan(byref(tall), byref(ts))
        out["bias_bf16"] = ops.gemm_nt(A, B, L.EPI_BIAS_BF16, bias=BIAS)
        out["dact"], out["gelu"] = ops.gemm_nt(A, B, L.EPI_BIAS_GELU, bias=BIAS)
        cs = torch.zeros(N, device=DEV)
        out["gelugrad"] = ops.gemm_nt(A, B, L.EPI_GELUGRAD_BF16, aux=AUX, out2=cs)
        out["colsum"] = cs
        torch.cuda.synchronize()
        assert bool((guard[M:] == 7.0).all()), "rows past M were written"
        return out, (tall.value, ts.value)

    try:
        got, took = run_all()
        print(f"two heights {M}x{N}x{K}: {took[0]} panels of 160 rows + panels of {32 * took[1]} rows")
        if plan is not None:
            assert took == plan, took
        lib.ce_gemm_nt_tune(165)                  # one height: the persistent kernel with 160-row tiles
        want, took1 = run_all()
        assert took1 == (0, 0)
    finally:
        lib.ce_gemm_nt_tune(0)
        assert lib.ce_gemm_set_dynamic_tiles(-1) == 0
    for k in ("bf16", "bias_bf16", "dact", "gelu", "gelugrad"):
        assert torch.equal(got[k], want[k]), k
    assert _report("two heights colsum", got["colsum"].cpu(), want["colsum"].cpu())[1] < 1e-5      # (float atomics: order)
    acc = (A.float() @ B.float().t()).cpu()
    assert _report("two heights bf16", got["bf16"].float().cpu(), acc.to(torch.bfloat16).float())[1] < 1e-3


@pytest.mark.parametrize("M,N,K", [(18464, 1024, 1024), (18464, 4096, 1024), (9000, 2048, 512)])
def test_gemm_nt_fp8_two_tile_heights(M, N, K):
    """The e4m3 persistent kernel with two tile heights (128-row panels + a shorter tail height; `launch_nt_fp8_lw`): bit-identical
    to the one-height launch (CE_NT_MIXED is read once per process, so the one-height result comes from the same kernel family forced
    through `ce_gemm_nt_fp8_tune`), and the plan the launcher took is a real split at ViT-L/14@336's shapes."""
    from ctypes import byref, c_int
    from clip_event_amd import ops, _lib as L
    rng = np.random.default_rng(M + N + K)
    a = _randn(rng, M, K).to(torch.bfloat16)
    b = _randn(rng, N, K, scale=K ** -0.5).to(torch.bfloat16)
    qa, sa, fa = _q8_ref(a)
    qb, sb, fb = _q8_ref(b)
    bias = _randn(rng, N)
    resid = _randn(rng, M, N)
    A8, SA, B8, SB = qa.to(DEV), sa.to(DEV), qb.to(DEV), sb.to(DEV)
    lib = L.lib()
    got = ops.gemm_nt_fp8(A8, SA, B8, SB, L.EPI_BIAS_RESID_F32, bias=bias.to(DEV), resid=resid.to(DEV))
    tall, ts = c_int(-1), c_int(-1)
    lib.ce_gemm_nt_last_plan(byref(tall), byref(ts))
    got16 = ops.gemm_nt_fp8(A8, SA, B8, SB, L.EPI_BF16)
    torch.cuda.synchronize()
    print(f"fp8 two heights {M}x{N}x{K}: {tall.value} panels of 128 rows + panels of {32 * ts.value} rows")
    if M == 18464:
        assert tall.value > 0 and 1 <= ts.value <= 3, (tall.value, ts.value)
    ref = (fa @ fb.t()) * sa[:, None] * sb[None, :]
    assert _report("fp8 two heights bias+resid f32", got.cpu(), ref + bias + resid)[1] < 1e-5
    assert _report("fp8 two heights bf16", got16.float().cpu(), ref.to(torch.bfloat16).float())[1] < 3e-3
