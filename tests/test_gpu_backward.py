"""GPU parity of the training-step backward kernels against fp64 torch autograd of the same op."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rnd(*shape, seed=0, scale=1.0, shift=0.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float64) * scale + shift


def dev32(t):
    return t.to(torch.float32).to(DEV).contiguous()


def rel_err(got, want):
    return ((got.detach().cpu().double() - want).abs().max() / want.abs().max().clamp_min(1e-30)).item()


def gln64(x, gamma, beta):
    m = x.mean(dim=(1, 2), keepdim=True)
    v = ((x - m) ** 2).mean(dim=(1, 2), keepdim=True)
    return gamma.view(1, -1, 1) * (x - m) / torch.sqrt(v + 1e-8) + beta.view(1, -1, 1)


def sums64(x):
    from sudo_rm_rf_amd import _lib
    s = torch.zeros(x.shape[0], _lib.STAT_BUCKETS, 2, dtype=torch.float64)
    s[:, 0, 0] = x.sum(dim=(1, 2))
    s[:, 0, 1] = (x * x).sum(dim=(1, 2))
    return s.to(DEV)


@pytest.mark.parametrize("Bt,Cin,Cout,L", [(3, 256, 512, 3200), (2, 512, 256, 832), (2, 64, 42, 200), (5, 48, 160, 132),
                                           (1, 16, 32, 36), (32, 256, 512, 3200), (64, 16, 32, 400), (24, 32, 16, 3200),
                                           (8, 16, 48, 200), (5, 48, 16, 76), (4, 64, 64, 132), (3, 8, 8, 20),
                                           # round 6: the wide-tile kernel's second form (128 x 256), its single-tile case, a full
                                           # 128-multiple shape that stays on the 128 x 128 kernel's unmasked form, one chunk of one k-tile
                                           (3, 256, 128, 96), (2, 128, 256, 1664), (2, 128, 384, 64), (1, 256, 256, 32)])
@pytest.mark.parametrize("pro", [0, 1, 2, 3])
def test_pw_wgrad(Bt, Cin, Cout, L, pro):
    from sudo_rm_rf_amd import ops
    if Bt == 32 and pro not in (0, 2):
        pytest.skip("full-size case: proj / res_conv prologues only")
    x = rnd(Bt, Cin, L, seed=1, scale=1.3, shift=0.2)
    g = rnd(Bt, Cout, L, seed=2, scale=0.7)
    gamma, beta = rnd(Cin, seed=3, scale=0.3, shift=1.0), rnd(Cin, seed=4, scale=0.3)
    slope = torch.tensor([0.17], dtype=torch.float64)
    fx, kw = x, {}
    if pro in (1, 2):
        fx = gln64(x, gamma, beta)
        kw.update(in_sums=sums64(x), in_gamma=dev32(gamma), in_beta=dev32(beta))
    if pro in (2, 3):
        fx = torch.where(fx >= 0, fx, slope * fx)
        kw.update(in_prelu=dev32(slope))
    want_w = torch.einsum("bml,bnl->mn", g, fx)
    want_b = g.sum(dim=(0, 2))
    dw, db = ops.pw_wgrad(dev32(g), dev32(x), **kw)
    assert rel_err(dw, want_w) <= 2e-5, rel_err(dw, want_w)
    assert rel_err(db, want_b) <= 2e-5
    dw2, db2 = ops.pw_wgrad(dev32(g), dev32(x), dw=dw.clone(), dbias=db.clone(), **kw)     # accumulate
    assert rel_err(dw2, 2 * want_w) <= 2e-5 and rel_err(db2, 2 * want_b) <= 2e-5


@pytest.mark.parametrize("flags", [1 << 18, (1 << 18) | (1 << 19)])
@pytest.mark.parametrize("Bt,Cin,Cout,L,pro", [(3, 256, 512, 3200, 2), (2, 512, 256, 832, 0), (2, 128, 256, 1664, 1), (1, 256, 256, 32, 3)])
def test_pw_wgrad_kernel_forms_agree(Bt, Cin, Cout, L, pro, flags):
    """Round 6: full shapes run the wide-tile kernel (256 x 128 / 128 x 256); debug flag 1 << 18 keeps them on the 128 x 128 kernel's
    unmasked form, 1 << 19 on its masked form (what ragged shapes run).  All three against fp64, and against each other."""
    from sudo_rm_rf_amd import ops
    x = rnd(Bt, Cin, L, seed=1, scale=1.3, shift=0.2)
    g = rnd(Bt, Cout, L, seed=2, scale=0.7)
    gamma, beta = rnd(Cin, seed=3, scale=0.3, shift=1.0), rnd(Cin, seed=4, scale=0.3)
    slope = torch.tensor([0.17], dtype=torch.float64)
    fx, kw = x, {}
    if pro in (1, 2):
        fx = gln64(x, gamma, beta)
        kw.update(in_sums=sums64(x), in_gamma=dev32(gamma), in_beta=dev32(beta))
    if pro in (2, 3):
        fx = torch.where(fx >= 0, fx, slope * fx)
        kw.update(in_prelu=dev32(slope))
    want_w, want_b = torch.einsum("bml,bnl->mn", g, fx), g.sum(dim=(0, 2))
    dw0, db0 = ops.pw_wgrad(dev32(g), dev32(x), **kw)
    ops.set_debug_flags(flags)
    try:
        dw1, db1 = ops.pw_wgrad(dev32(g), dev32(x), **kw)
    finally:
        ops.set_debug_flags(0)
    for dw, db in ((dw0, db0), (dw1, db1)):
        assert rel_err(dw, want_w) <= 2e-5 and rel_err(db, want_b) <= 2e-5
    assert rel_err(dw1, dw0.double().cpu()) <= 1e-5


@pytest.mark.parametrize("Bt,C,L", [(3, 64, 3200), (2, 20, 203), (4, 5, 1), (32, 512, 400)])
@pytest.mark.parametrize("act", [False, True])
def test_gln_bwd(Bt, C, L, act):
    from sudo_rm_rf_amd import ops
    x = rnd(Bt, C, L, seed=10, scale=1.7, shift=-0.4).requires_grad_(True)
    gamma = rnd(C, seed=11, scale=0.3, shift=1.0).requires_grad_(True)
    beta = rnd(C, seed=12, scale=0.3).requires_grad_(True)
    slope = torch.tensor([0.23], dtype=torch.float64, requires_grad=True)
    gout, gout2 = rnd(Bt, C, L, seed=13), rnd(Bt, C, L, seed=14, scale=0.5)
    y = gln64(x, gamma, beta)
    if act:
        y = torch.where(y >= 0, y, slope * y)
    y.backward(gout + gout2)
    gx, dg, db, ds = ops.gln_bwd(dev32(gout), dev32(x.detach()), sums64(x.detach()), dev32(gamma.detach()),
                                 dev32(beta.detach()), prelu=dev32(slope.detach()) if act else None,
                                 gout2=dev32(gout2))
    assert rel_err(gx, x.grad) <= 3e-5, rel_err(gx, x.grad)
    assert rel_err(dg, gamma.grad) <= 3e-5 and rel_err(db, beta.grad) <= 3e-5
    if act:
        assert rel_err(ds, slope.grad) <= 3e-5
    # accumulation into existing buffers
    gx2, dg2, db2, _ = ops.gln_bwd(dev32(gout), dev32(x.detach()), sums64(x.detach()), dev32(gamma.detach()),
                                   dev32(beta.detach()), prelu=dev32(slope.detach()) if act else None,
                                   gout2=dev32(gout2), gx=gx.clone(), dgamma=dg.clone(), dbeta=db.clone(),
                                   dslope=ds.clone() if act else None)
    assert rel_err(gx2, 2 * x.grad) <= 3e-5 and rel_err(dg2, 2 * gamma.grad) <= 3e-5 and rel_err(db2, 2 * beta.grad) <= 3e-5


@pytest.mark.parametrize("Bt,C,L,D", [(2, 16, 3200, 5), (3, 5, 64, 3), (1, 7, 32, 6), (2, 4, 10, 1), (2, 8, 640, 6), (3, 4, 48, 4),
                                      (1, 3, 12, 2)])
def test_merge_bwd(Bt, C, L, D):
    """The one-pass kernel (every level's pair sums from one read of g_merged) against torch autograd AND bitwise against the
    chain of pair-sum launches (kernel mode 1): same additions, same order."""
    from sudo_rm_rf_amd import ops
    levels = [rnd(Bt, C, L >> k, seed=20 + k).requires_grad_(True) for k in range(D)]
    out = levels[-1]
    for k in range(D - 2, -1, -1):                                       # improved_sudormrf.py:214-216
        out = levels[k] + F.interpolate(out, scale_factor=2, mode="nearest")
    gm = rnd(Bt, C, L, seed=30)
    out.backward(gm)
    got = ops.merge_bwd(dev32(gm), D)
    for k in range(D):
        assert rel_err(got[k], levels[k].grad) <= 1e-6
    try:
        ops.set_kernel_mode(1)
        chain = ops.merge_bwd(dev32(gm), D)
    finally:
        ops.set_kernel_mode(0)
    for k in range(D):
        assert torch.equal(got[k], chain[k])


@pytest.mark.parametrize("Bt,C,Lin,stride", [(2, 64, 3200, 1), (2, 64, 3200, 2), (3, 20, 200, 2), (1, 3, 7, 1),
                                             (2, 3, 5, 2), (1, 2, 1, 1), (4, 512, 400, 2)])
@pytest.mark.parametrize("pro", [0, 1, 2])
def test_dwconv5_bwd(Bt, C, Lin, stride, pro):
    from sudo_rm_rf_amd import ops
    x = rnd(Bt, C, Lin, seed=40, scale=1.4, shift=0.3)
    w = rnd(C, 1, 5, seed=41, scale=0.4).requires_grad_(True)
    b = rnd(C, seed=42, scale=0.2).requires_grad_(True)
    gamma, beta = rnd(C, seed=43, scale=0.3, shift=1.0), rnd(C, seed=44, scale=0.3)
    slope = torch.tensor([0.21], dtype=torch.float64)
    u, kw = x, {}
    if pro >= 1:
        u = gln64(x, gamma, beta)
        kw.update(in_sums=sums64(x), in_gamma=dev32(gamma), in_beta=dev32(beta))
    if pro == 2:
        u = torch.where(u >= 0, u, slope * u)
        kw.update(in_prelu=dev32(slope))
    u = u.detach().requires_grad_(True)
    d = F.conv1d(u, w, b, stride=stride, padding=2, groups=C)
    gd = rnd(*d.shape, seed=45)
    d.backward(gd)
    gin, dw, db = ops.dwconv5_bwd(dev32(gd), dev32(x), dev32(w.detach()), stride, **kw)
    assert rel_err(gin, u.grad) <= 2e-6
    assert rel_err(dw, w.grad) <= 2e-5 and rel_err(db, b.grad) <= 2e-5
    _, dw2, db2 = ops.dwconv5_bwd(dev32(gd), dev32(x), dev32(w.detach()), stride, dw=dw.clone(), dbias=db.clone(),
                                  want_gin=False, **kw)
    assert rel_err(dw2, 2 * w.grad) <= 2e-5 and rel_err(db2, 2 * b.grad) <= 2e-5


@pytest.mark.parametrize("Bt,S,N,L", [(2, 2, 24, 300), (1, 3, 5, 17), (3, 2, 512, 256)])
def test_mask_apply_and_bwd(Bt, S, N, L):
    from sudo_rm_rf_amd import ops
    m = rnd(Bt, S * N, L, seed=50).requires_grad_(True)
    e = rnd(Bt, N, L, seed=51).requires_grad_(True)
    v = (torch.relu(m).view(Bt, S, N, L) * e.unsqueeze(1)).reshape(Bt, S * N, L)   # improved_sudormrf.py:296-298
    gv = rnd(Bt, S * N, L, seed=52)
    v.backward(gv)
    assert rel_err(ops.mask_apply(dev32(m.detach()), dev32(e.detach())), v.detach()) <= 1e-6
    gm, ge = ops.mask_bwd(dev32(gv), dev32(m.detach()), dev32(e.detach()))
    assert rel_err(gm, m.grad) <= 1e-6 and rel_err(ge, e.grad) <= 2e-6
    _, ge2 = ops.mask_bwd(dev32(gv), dev32(m.detach()), dev32(e.detach()), genc=ge.clone())
    assert rel_err(ge2, 2 * e.grad) <= 2e-6


def test_prelu_bwd():
    from sudo_rm_rf_amd import ops
    x = rnd(3, 40, 1001, seed=60).requires_grad_(True)
    a = torch.tensor([0.31], dtype=torch.float64, requires_grad=True)
    g = rnd(3, 40, 1001, seed=61)
    F.prelu(x, a).backward(g)
    gx, da = ops.prelu_bwd(dev32(g), dev32(x.detach()), dev32(a.detach()))
    assert rel_err(gx, x.grad) <= 1e-6 and rel_err(da, a.grad) <= 1e-5


@pytest.mark.parametrize("Bt,A,T,K,N", [(2, 1, 3200, 21, 48), (1, 2, 330, 11, 16), (3, 1, 50, 21, 8)])
def test_encoder_weight_grad_via_frames(Bt, A, T, K, N):
    """dW_enc = wgrad GEMM over gathered input frames == autograd of conv1d(stride K//2, padding K//2)."""
    from sudo_rm_rf_amd import ops
    h = K // 2
    Tp = ((T + h * 4 - 1) // (h * 4)) * (h * 4)
    x = torch.zeros(Bt, A, Tp, dtype=torch.float64)
    x[..., :T] = rnd(Bt, A, T, seed=70)
    w = rnd(N, A, K, seed=71, scale=0.3).requires_grad_(True)
    s = F.conv1d(x, w, None, stride=h, padding=h)
    L = Tp // h
    s = s[..., :L]
    gs = rnd(Bt, N, L, seed=72)
    s.backward(gs)
    frames = ops.frames_gather(dev32(x[..., :T]), K, h, h, L)
    dw, _ = ops.pw_wgrad(dev32(gs), frames, want_bias=False)
    assert rel_err(dw.view(N, A, K), w.grad) <= 2e-5


@pytest.mark.parametrize("Bt,Ci,Co,K,L", [(2, 64, 2, 21, 100), (1, 96, 2, 21, 64), (2, 32, 4, 11, 40)])
def test_decoder_backward_via_frames(Bt, Ci, Co, K, L):
    """g_v = W_d (padded to 64 columns) x gathered output-gradient frames, dW_d = wgrad(v, frames)."""
    from sudo_rm_rf_amd import ops
    h = K // 2
    T = h * L
    v = rnd(Bt, Ci, L, seed=80).requires_grad_(True)
    w = rnd(Ci, Co, K, seed=81, scale=Ci ** -0.5).requires_grad_(True)
    out = F.conv_transpose1d(v, w, None, stride=h, padding=h, output_padding=h - 1)[..., :T]
    go = rnd(Bt, Co, T, seed=82)
    out.backward(go)
    rows = ((Co * K + 63) // 64) * 64
    frames = ops.frames_gather(dev32(go), K, h, h, L, rows_out=rows)
    wp = torch.zeros(Ci, rows, dtype=torch.float32, device=DEV)
    wp[:, :Co * K] = dev32(w.detach()).view(Ci, Co * K)
    gv = ops.pw_conv(frames, wp, torch.zeros(Ci, dtype=torch.float32, device=DEV))
    assert rel_err(gv, v.grad) <= 2e-5
    dw, _ = ops.pw_wgrad(dev32(v.detach()), frames, want_bias=False)
    assert rel_err(dw[:, :Co * K].reshape(Ci, Co, K), w.grad) <= 2e-5


@pytest.mark.parametrize("Bt,G,n,L", [(2, 16, 16, 300), (1, 4, 8, 76), (2, 8, 4, 132), (2, 2, 2, 64), (3, 16, 8, 40)])
def test_tac_bwd(Bt, G, n, L):
    """TAC MLP backward (groupcomm_sudormrf_v2.py:356-377) against fp64 autograd."""
    from sudo_rm_rf_amd import ops
    H = 3 * n
    x = rnd(Bt, G, n, L, seed=90).requires_grad_(True)
    P = [rnd(H, n, seed=91, scale=n ** -0.5), rnd(H, seed=92, scale=0.2), torch.tensor([0.2], dtype=torch.float64),
         rnd(H, H, seed=93, scale=H ** -0.5), rnd(H, seed=94, scale=0.2), torch.tensor([0.3], dtype=torch.float64),
         rnd(n, 2 * H, seed=95, scale=(2 * H) ** -0.5), rnd(n, seed=96, scale=0.2),
         torch.tensor([0.15], dtype=torch.float64)]
    P = [p.requires_grad_(True) for p in P]
    pr = lambda t, a: torch.where(t >= 0, t, a * t)
    rows = x.permute(0, 3, 1, 2).reshape(-1, n)
    z = pr(rows @ P[0].T + P[1], P[2]).view(Bt, L, G, H)
    q = pr(z.mean(2).view(Bt * L, H) @ P[3].T + P[4], P[5])
    cat = torch.cat([z.view(Bt * L, G, H), q.unsqueeze(1).expand(Bt * L, G, H)], 2).reshape(-1, 2 * H)
    o = pr(cat @ P[6].T + P[7], P[8]).view(Bt, L, G, n).permute(0, 2, 3, 1).contiguous()
    go = rnd(Bt, G, n, L, seed=97)
    o.backward(go)
    gx, grads = ops.tac_bwd(dev32(x.detach()), dev32(go), [dev32(p.detach()) for p in P])
    assert rel_err(gx, x.grad) <= 3e-5, rel_err(gx, x.grad)
    for i, (g, p) in enumerate(zip(grads, P)):
        assert rel_err(g, p.grad) <= 5e-5, (i, rel_err(g, p.grad))
