"""Per-kernel Python wrappers over the C ABI (unit parity tests and sub-module forwards).

Every function takes CUDA float32 tensors, launches on torch's current stream and returns new
tensors.  GlobLN statistics travel as float64 ``sums`` tensors of shape [groups, 2]
({sum, sum of squares}); producers accumulate into them, so pass zero-initialised tensors.
"""
import ctypes as C
import weakref

import torch

from . import _lib


def _chk(*ts):
    dev = None
    for t in ts:
        if t is None:
            continue
        if t.device.type != "cuda":
            raise _lib.SrfError("sudo_rm_rf_amd ops need CUDA (ROCm) tensors; got %s" % t.device)
        if not t.is_contiguous():
            raise _lib.SrfError("tensor must be contiguous")
        dev = t.device
    return dev


def new_sums(groups, device):
    return torch.zeros((groups, _lib.STAT_BUCKETS, 2), dtype=torch.float64, device=device)


def _norm(sums, gamma, beta, prelu):
    if sums is None and gamma is None and beta is None and prelu is None:
        return None
    return C.byref(_lib.make_norm(sums, gamma, beta, prelu))


def encoder(wav, weight, L, sums=None):
    """wav [Bt,A,T], weight [N,A,K] -> [Bt,N,L]  (reference: improved_sudormrf.py:247-251,286)."""
    dev = _chk(wav, weight, sums)
    Bt, A, T = wav.shape
    N, A2, K = weight.shape
    assert A == A2
    out = torch.empty((Bt, N, L), dtype=torch.float32, device=dev)
    _lib.check(_lib.load().srf_encoder(_lib.ptr(wav), _lib.ptr(weight), _lib.ptr(out), _lib.ptr(sums),
                                       Bt, A, T, N, K, L, _lib.current_stream(dev)), "srf_encoder")
    return out


def gln_stats(x, groups):
    dev = _chk(x)
    sums = new_sums(groups, dev)
    per_group = x.numel() // groups
    _lib.check(_lib.load().srf_gln_stats(_lib.ptr(x), _lib.ptr(sums), groups, per_group,
                                         _lib.current_stream(dev)), "srf_gln_stats")
    return sums


def gln_apply(x, sums, gamma, beta, prelu=None, residual=None):
    """x [groups, channels, length]; y = GlobLN(x) (+PReLU), or residual + GlobLN(x)."""
    dev = _chk(x, sums, gamma, beta, prelu, residual)
    groups, channels = x.shape[0], x.shape[1]
    length = x.numel() // (groups * channels)
    y = torch.empty_like(x)
    n = _lib.make_norm(sums, gamma, beta, prelu)
    lib = _lib.load()
    if residual is None:
        rc = lib.srf_gln_apply(_lib.ptr(x), _lib.ptr(y), C.byref(n), groups, channels, length,
                               _lib.current_stream(dev))
    else:
        rc = lib.srf_gln_apply_add(_lib.ptr(residual), _lib.ptr(x), _lib.ptr(y), C.byref(n), groups,
                                   channels, length, _lib.current_stream(dev))
    _lib.check(rc, "srf_gln_apply")
    return y


def glob_ln(x, gamma, beta):
    """GlobLN.forward (improved_sudormrf.py:30-47) on [batch, channels, *]."""
    x = x.contiguous()
    return gln_apply(x, gln_stats(x, x.shape[0]), gamma, beta)


def pack_pw_weight(weight):
    """Split a 1x1 weight into bf16 hi/lo tiles for the split-precision GEMM (None if the shape does not
    qualify).  srf_forward does this itself, once per forward, for all its 1x1 convolutions."""
    dev = _chk(weight)
    Cout = weight.shape[0]
    Cin = weight.numel() // Cout
    lib = _lib.load()
    nbytes = lib.srf_packed_pw_weight_bytes(Cout, Cin)
    if not nbytes:
        return None
    packed = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _lib.check(lib.srf_pack_pw_weights((C.c_void_p * 1)(weight.data_ptr()), (C.c_void_p * 1)(packed.data_ptr()),
                                       (C.c_int * 1)(Cout), (C.c_int * 1)(Cin), 1, _lib.current_stream(dev)),
               "srf_pack_pw_weights")
    return packed


def pack3_pw_weight(weight):
    """Three-part (h | m | l) bf16 image of a 1x1 weight for the six-MFMA GEMM (None if the shape does not qualify)."""
    dev = _chk(weight)
    Cout = weight.shape[0]
    Cin = weight.numel() // Cout
    lib = _lib.load()
    nbytes = lib.srf_packed3_pw_weight_bytes(Cout, Cin)
    if not nbytes:
        return None
    packed = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _lib.check(lib.srf_pack3_pw_weights((C.c_void_p * 1)(weight.data_ptr()), (C.c_void_p * 1)(packed.data_ptr()),
                                        (C.c_int * 1)(Cout), (C.c_int * 1)(Cin), 1, _lib.current_stream(dev)),
               "srf_pack3_pw_weights")
    # the library keeps a (device, address) -> format record of every packed3 image; it goes when the buffer does (the caching
    # allocator will hand the address to something else)
    weakref.finalize(packed, lib.srf_pack3_forget, packed.data_ptr())
    return packed


def pw_conv3(x, weight, bias, packed3, in_sums=None, in_gamma=None, in_beta=None, in_prelu=None, residual=None, out_sums=None):
    """1x1 conv on the three-part split GEMM (srf_pw_conv_packed3): the training forward's GEMM."""
    dev = _chk(x, weight, bias, in_sums, in_gamma, in_beta, in_prelu, residual, out_sums)
    Bt, Cin, L = x.shape
    Cout = weight.shape[0]
    y = torch.empty((Bt, Cout, L), dtype=torch.float32, device=dev)
    rc = _lib.load().srf_pw_conv_packed3(_lib.ptr(x), _lib.ptr(weight), _lib.ptr(packed3), _lib.ptr(bias), _lib.ptr(y), Bt, Cin,
                                         Cout, L, _norm(in_sums, in_gamma, in_beta, in_prelu), _lib.ptr(residual),
                                         _lib.ptr(out_sums), _lib.current_stream(dev))
    _lib.check(rc, "srf_pw_conv_packed3")
    return y


def pw_conv(x, weight, bias, in_sums=None, in_gamma=None, in_beta=None, in_prelu=None, residual=None,
            out_sums=None, mask_mul=None, packed=None):
    """1x1 conv with fused prologue / epilogue.  x [Bt,Cin,L], weight [Cout,Cin(,1)] -> [Bt,Cout,L]."""
    dev = _chk(x, weight, bias, in_sums, in_gamma, in_beta, in_prelu, residual, out_sums, mask_mul)
    Bt, Cin, L = x.shape
    Cout = weight.shape[0]
    assert weight.numel() == Cout * Cin
    y = torch.empty((Bt, Cout, L), dtype=torch.float32, device=dev)
    lib = _lib.load()
    rc = lib.srf_pw_conv_packed(
        _lib.ptr(x), _lib.ptr(weight), _lib.ptr(packed), _lib.ptr(bias), _lib.ptr(y), Bt, Cin, Cout, L,
        _norm(in_sums, in_gamma, in_beta, in_prelu), _lib.ptr(residual), _lib.ptr(out_sums),
        1 if mask_mul is not None else 0, _lib.ptr(mask_mul),
        mask_mul.shape[1] if mask_mul is not None else 0, _lib.current_stream(dev))
    _lib.check(rc, "srf_pw_conv")
    return y


def pw_conv_pair_supported(Bt, Cin1, Cmid, Cout2, L):
    """Whether srf_pw_conv_pair serves this shape on this device under the current kernel mode / debug flags."""
    return bool(_lib.load().srf_pw_conv_pair_supported(Bt, Cin1, Cmid, Cout2, L))


def pw_conv_pair(x, packed1, bias1, in_sums, in_gamma, in_beta, in_prelu, residual, packed2, bias2, Cmid, Cout2, out_sums2=None):
    """Two 1x1 convs in one launch (srf_pw_conv_pair): y = W1 f(x) + b1 (+ residual), y2 = W2 y + b2 (+ statistics of y2).
    packed1 / packed2 = pack_pw_weight(W1 / W2).  Returns (y, y2)."""
    dev = _chk(x, packed1, bias1, in_sums, in_gamma, in_beta, in_prelu, residual, packed2, bias2, out_sums2)
    Bt, Cin1, L = x.shape
    y = torch.empty((Bt, Cmid, L), dtype=torch.float32, device=dev)
    y2 = torch.empty((Bt, Cout2, L), dtype=torch.float32, device=dev)
    rc = _lib.load().srf_pw_conv_pair(_lib.ptr(x), _lib.ptr(packed1), _lib.ptr(bias1), _lib.ptr(y),
                                      _norm(in_sums, in_gamma, in_beta, in_prelu), _lib.ptr(residual), _lib.ptr(packed2),
                                      _lib.ptr(bias2), _lib.ptr(y2), _lib.ptr(out_sums2), Bt, Cin1, Cmid, Cout2, L,
                                      _lib.current_stream(dev))
    _lib.check(rc, "srf_pw_conv_pair")
    return y, y2


def pw_conv_pair3_supported(Bt, Cin1, Cmid, Cout2, L):
    return bool(_lib.load().srf_pw_conv_pair_packed3_supported(Bt, Cin1, Cmid, Cout2, L))


def pw_conv_pair3(x, packed3_1, bias1, in_sums, in_gamma, in_beta, in_prelu, residual, packed3_2, bias2, Cmid, Cout2, out_sums2=None):
    """The training forward's fused pair (srf_pw_conv_pair_packed3): pw_conv_pair on the two-fp16-part images of pack3_pw_weight."""
    dev = _chk(x, packed3_1, bias1, in_sums, in_gamma, in_beta, in_prelu, residual, packed3_2, bias2, out_sums2)
    Bt, Cin1, L = x.shape
    y = torch.empty((Bt, Cmid, L), dtype=torch.float32, device=dev)
    y2 = torch.empty((Bt, Cout2, L), dtype=torch.float32, device=dev)
    rc = _lib.load().srf_pw_conv_pair_packed3(_lib.ptr(x), _lib.ptr(packed3_1), _lib.ptr(bias1), _lib.ptr(y),
                                              _norm(in_sums, in_gamma, in_beta, in_prelu), _lib.ptr(residual), _lib.ptr(packed3_2),
                                              _lib.ptr(bias2), _lib.ptr(y2), _lib.ptr(out_sums2), Bt, Cin1, Cmid, Cout2, L,
                                              _lib.current_stream(dev))
    _lib.check(rc, "srf_pw_conv_pair_packed3")
    return y, y2


def dwconv5(x, weight, bias, stride, in_sums=None, in_gamma=None, in_beta=None, in_prelu=None,
            out_sums=None):
    """depthwise k=5 conv.  x [Bt,C,Lin], weight [C,1,5] -> [Bt,C,Lout]."""
    dev = _chk(x, weight, bias, in_sums, in_gamma, in_beta, in_prelu, out_sums)
    Bt, Cc, Lin = x.shape
    Lout = (Lin - 1) // stride + 1
    y = torch.empty((Bt, Cc, Lout), dtype=torch.float32, device=dev)
    rc = _lib.load().srf_dwconv5(_lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(y), Bt, Cc, Lin,
                                 stride, _norm(in_sums, in_gamma, in_beta, in_prelu), _lib.ptr(out_sums),
                                 _lib.current_stream(dev))
    _lib.check(rc, "srf_dwconv5")
    return y


def conv1d(x, weight, bias, stride=1, padding=0, dilation=1, groups=1, out_sums=None):
    """General Conv1d (srf_conv1d): x [Bt,Cin,Lin], weight [Cout,Cin/groups,K] -> [Bt,Cout,Lout]; not on the model's path."""
    dev = _chk(x, weight, bias, out_sums)
    Bt, Cin, Lin = x.shape
    Cout, cpg, K = weight.shape
    if cpg * groups != Cin:
        raise RuntimeError("weight %s does not match %d input channels in %d groups" % (tuple(weight.shape), Cin, groups))
    Lout = (Lin + 2 * padding - dilation * (K - 1) - 1) // stride + 1
    if Lout <= 0:
        raise RuntimeError("kernel size %d (dilation %d) exceeds the padded input length %d" % (K, dilation, Lin + 2 * padding))
    y = torch.empty((Bt, Cout, Lout), dtype=torch.float32, device=dev)
    rc = _lib.load().srf_conv1d(_lib.ptr(x), _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(y), Bt, Cin, Cout, Lin, K, stride,
                                padding, dilation, groups, _lib.ptr(out_sums), _lib.current_stream(dev))
    _lib.check(rc, "srf_conv1d")
    return y


def merge(levels, sums, gammas, betas, out_sums=None):
    """levels[k] [Bt,C,L>>k] (pre-norm) -> merged [Bt,C,L] (improved_sudormrf.py:214-216)."""
    dev = _chk(*levels, *sums, *gammas, *betas, out_sums)
    D = len(levels)
    Bt, Cc, L = levels[0].shape
    y = torch.empty((Bt, Cc, L), dtype=torch.float32, device=dev)
    lv = (C.c_void_p * D)(*[t.data_ptr() for t in levels])
    norms = (_lib.srf_norm * D)(*[_lib.make_norm(s, g, b, None) for s, g, b in zip(sums, gammas, betas)])
    rc = _lib.load().srf_merge(lv, norms, D, _lib.ptr(y), Bt, Cc, L, _lib.ptr(out_sums),
                               _lib.current_stream(dev))
    _lib.check(rc, "srf_merge")
    return y


def pyramid(y1, in_sums, in_gamma, in_beta, in_prelu, weights, biases, gammas, betas, out_sums=None):
    """Fused depthwise pyramid + merge: y1 [groups,C,L] -> merged [groups,C,L] (srf_pyramid)."""
    dev = _chk(y1, in_sums, in_gamma, in_beta, in_prelu, *weights, *biases, *gammas, *betas, out_sums)
    groups, Cc, L = y1.shape
    D = len(weights)
    lib = _lib.load()
    if not lib.srf_pyramid_supported(Cc, L, D):
        raise _lib.SrfError("srf_pyramid: unsupported shape C=%d L=%d D=%d" % (Cc, L, D))
    merged = torch.empty_like(y1)
    scratch = torch.empty(lib.srf_pyramid_scratch_bytes(groups, Cc, L, D), dtype=torch.uint8, device=dev)
    arr = lambda ts: (C.c_void_p * D)(*[t.data_ptr() for t in ts])
    n = _lib.make_norm(in_sums, in_gamma, in_beta, in_prelu)
    rc = lib.srf_pyramid(_lib.ptr(y1), _lib.ptr(merged), C.byref(n), arr(weights), arr(biases), arr(gammas),
                         arr(betas), groups, Cc, L, D, _lib.ptr(scratch), _lib.ptr(out_sums),
                         _lib.current_stream(dev))
    _lib.check(rc, "srf_pyramid")
    return merged


def decoder(v, weight, T):
    """v [Bt,Ci,L], weight [Ci,Co,K] -> [Bt,Co,T]  (improved_sudormrf.py:272-279,300,316-318)."""
    dev = _chk(v, weight)
    Bt, Ci, L = v.shape
    Ci2, Co, K = weight.shape
    assert Ci == Ci2
    lib = _lib.load()
    scratch = torch.empty(lib.srf_decoder_scratch_floats(Bt, Ci, Co, K, L), dtype=torch.float32, device=dev)
    out = torch.empty((Bt, Co, T), dtype=torch.float32, device=dev)
    rc = lib.srf_decoder(_lib.ptr(v), _lib.ptr(weight), _lib.ptr(out), Bt, Ci, Co, K, L, T,
                         _lib.ptr(scratch), _lib.current_stream(dev))
    _lib.check(rc, "srf_decoder")
    return out


def tac(x4, params, out_sums=None):
    """x4 [Bt,G,n,L]; params = the 9 TAC tensors in state_dict order -> pre-norm q [Bt,G,n,L]."""
    dev = _chk(x4, *params, out_sums)
    Bt, G, n, L = x4.shape
    H = params[0].shape[0]
    q = torch.empty_like(x4)
    pp = (C.c_void_p * 9)(*[p.data_ptr() for p in params])
    rc = _lib.load().srf_tac(_lib.ptr(x4), _lib.ptr(q), pp, Bt, G, n, H, L, _lib.ptr(out_sums),
                             _lib.current_stream(dev))
    _lib.check(rc, "srf_tac")
    return q


def mixture_consistency(pr_batch, input_mixture, mix_weights_type="uniform"):
    dev = _chk(pr_batch, input_mixture)
    Bt, S, T = pr_batch.shape
    out = torch.empty_like(pr_batch)
    if mix_weights_type == "magsq":
        work = torch.empty(Bt * S, dtype=torch.float32, device=pr_batch.device)
        rc = _lib.load().srf_mixture_consistency_magsq(_lib.ptr(pr_batch), _lib.ptr(input_mixture), _lib.ptr(out),
                                                       Bt, S, T, _lib.ptr(work), _lib.current_stream(dev))
        _lib.check(rc, "srf_mixture_consistency_magsq")
        return out
    rc = _lib.load().srf_mixture_consistency(_lib.ptr(pr_batch), _lib.ptr(input_mixture), _lib.ptr(out),
                                             Bt, S, T, _lib.current_stream(dev))
    _lib.check(rc, "srf_mixture_consistency")
    return out


def pw_wgrad(g, x, in_sums=None, in_gamma=None, in_beta=None, in_prelu=None, dw=None, dbias=None, want_bias=True):
    """Weight / bias gradient of a pointwise conv.  g [Bt,Cout,L], x [Bt,Cin,L] -> (dw [Cout,Cin], dbias [Cout]).
    Passing dw / dbias accumulates into them."""
    dev = _chk(g, x, in_sums, in_gamma, in_beta, in_prelu, dw, dbias)
    Bt, Cout, L = g.shape
    Cin = x.shape[1]
    assert x.shape == (Bt, Cin, L)
    lib = _lib.load()
    acc = dw is not None
    if dw is None:
        dw = torch.empty((Cout, Cin), dtype=torch.float32, device=dev)
    if dbias is None and want_bias:
        assert not acc
        dbias = torch.empty((Cout,), dtype=torch.float32, device=dev)
    scratch = torch.empty(lib.srf_pw_wgrad_scratch_bytes(Bt, Cout, Cin, L), dtype=torch.uint8, device=dev)
    rc = lib.srf_pw_wgrad(_lib.ptr(g), _lib.ptr(x), _norm(in_sums, in_gamma, in_beta, in_prelu), Bt, Cin, Cout, L,
                          _lib.ptr(dw), _lib.ptr(dbias), 1 if acc else 0, _lib.ptr(scratch),
                          _lib.current_stream(dev))
    _lib.check(rc, "srf_pw_wgrad")
    return dw, dbias


def gln_bwd(gout, x, sums, gamma, beta, prelu=None, gout2=None, gx=None, dgamma=None, dbeta=None, dslope=None):
    """GlobLN (+PReLU) backward.  Returns (gx, dgamma, dbeta, dslope); passing gx / d* accumulates into them."""
    dev = _chk(gout, x, sums, gamma, beta, prelu, gout2, gx, dgamma, dbeta, dslope)
    groups, Cc, L = x.shape
    lib = _lib.load()
    acc = gx is not None
    if gx is None:
        gx = torch.empty_like(x)
    dgamma = torch.zeros(Cc, dtype=torch.float32, device=dev) if dgamma is None else dgamma
    dbeta = torch.zeros(Cc, dtype=torch.float32, device=dev) if dbeta is None else dbeta
    if prelu is not None and dslope is None:
        dslope = torch.zeros(1, dtype=torch.float32, device=dev)
    scratch = torch.empty(lib.srf_gln_bwd_scratch_bytes(groups, Cc), dtype=torch.uint8, device=dev)
    n = _norm(sums, gamma, beta, prelu)
    rc = lib.srf_gln_bwd(_lib.ptr(gout), _lib.ptr(gout2), _lib.ptr(x), n, groups, Cc, L, _lib.ptr(gx),
                         1 if acc else 0, _lib.ptr(dgamma), _lib.ptr(dbeta), _lib.ptr(dslope), _lib.ptr(scratch),
                         _lib.current_stream(dev))
    _lib.check(rc, "srf_gln_bwd")
    return gx, dgamma, dbeta, dslope


def merge_bwd(g_merged, D):
    """-> [g_merged, g_n_1, ..., g_n_{D-1}] (level k: [groups, C, L >> k])."""
    dev = _chk(g_merged)
    groups, Cc, L = g_merged.shape
    outs = [g_merged] + [torch.empty((groups, Cc, L >> k), dtype=torch.float32, device=dev) for k in range(1, D)]
    arr = (C.c_void_p * D)(*[t.data_ptr() for t in outs])
    rc = _lib.load().srf_merge_bwd(_lib.ptr(g_merged), arr, D, groups * Cc, L, _lib.current_stream(dev))
    _lib.check(rc, "srf_merge_bwd")
    return outs


def dwconv5_bwd(gd, xin, weight, stride, in_sums=None, in_gamma=None, in_beta=None, in_prelu=None, dw=None,
                dbias=None, want_gin=True):
    """Depthwise conv backward -> (gin, dw [C,1,5], dbias [C]); dw / dbias passed in are accumulated into."""
    dev = _chk(gd, xin, weight, in_sums, in_gamma, in_beta, in_prelu, dw, dbias)
    groups, Cc, Lin = xin.shape
    lib = _lib.load()
    gin = torch.empty_like(xin) if want_gin else None
    dw = torch.zeros((Cc, 1, 5), dtype=torch.float32, device=dev) if dw is None else dw
    dbias = torch.zeros((Cc,), dtype=torch.float32, device=dev) if dbias is None else dbias
    scratch = torch.empty(lib.srf_dwconv5_bwd_scratch_bytes(groups, Cc), dtype=torch.uint8, device=dev)
    rc = lib.srf_dwconv5_bwd(_lib.ptr(gd), _lib.ptr(xin), _norm(in_sums, in_gamma, in_beta, in_prelu), _lib.ptr(weight),
                             groups, Cc, Lin, stride, _lib.ptr(gin), _lib.ptr(dw), _lib.ptr(dbias), _lib.ptr(scratch),
                             _lib.current_stream(dev))
    _lib.check(rc, "srf_dwconv5_bwd")
    return gin, dw, dbias


def mask_apply(m, enc):
    dev = _chk(m, enc)
    Bt, SN, L = m.shape
    N = enc.shape[1]
    v = torch.empty_like(m)
    _lib.check(_lib.load().srf_mask_apply(_lib.ptr(m), _lib.ptr(enc), _lib.ptr(v), Bt, SN // N, N, L,
                                          _lib.current_stream(dev)), "srf_mask_apply")
    return v


def mask_bwd(gv, m, enc, genc=None):
    dev = _chk(gv, m, enc, genc)
    Bt, SN, L = m.shape
    N = enc.shape[1]
    acc = genc is not None
    genc = torch.empty_like(enc) if genc is None else genc
    gm = torch.empty_like(gv)
    _lib.check(_lib.load().srf_mask_bwd(_lib.ptr(gv), _lib.ptr(m), _lib.ptr(enc), _lib.ptr(gm), _lib.ptr(genc),
                                        1 if acc else 0, Bt, SN // N, N, L, _lib.current_stream(dev)), "srf_mask_bwd")
    return gm, genc


def prelu_bwd(gout, x, slope, dslope=None):
    dev = _chk(gout, x, slope, dslope)
    gx = torch.empty_like(gout)
    dslope = torch.zeros(1, dtype=torch.float32, device=dev) if dslope is None else dslope
    _lib.check(_lib.load().srf_prelu_bwd(_lib.ptr(gout), _lib.ptr(x), _lib.ptr(slope), _lib.ptr(gx), _lib.ptr(dslope),
                                         gout.numel(), _lib.current_stream(dev)), "srf_prelu_bwd")
    return gx, dslope


def frames_gather(src, K, hop, pad, L, rows_out=None):
    dev = _chk(src)
    Bt, R, T = src.shape
    rows_out = R * K if rows_out is None else rows_out
    out = torch.empty((Bt, rows_out, L), dtype=torch.float32, device=dev)
    _lib.check(_lib.load().srf_frames_gather(_lib.ptr(src), _lib.ptr(out), Bt, R, T, K, hop, pad, L, rows_out,
                                             _lib.current_stream(dev)), "srf_frames_gather")
    return out


def tac_bwd(x, go, params, grads=None):
    """TAC MLP backward.  x, go [Bt,G,n,L]; params: the 9 TAC tensors -> (gx, grads) (grads accumulated into)."""
    dev = _chk(x, go, *params)
    Bt, G, n, L = x.shape
    lib = _lib.load()
    if grads is None:
        grads = [torch.zeros_like(p) for p in params]
    gx = torch.empty_like(x)
    scratch = torch.empty(lib.srf_tac_bwd_scratch_bytes(Bt, G, n, L), dtype=torch.uint8, device=dev)
    parr = (C.c_void_p * 9)(*[p.data_ptr() for p in params])
    garr = (C.c_void_p * 9)(*[g.data_ptr() for g in grads])
    rc = lib.srf_tac_bwd(_lib.ptr(x), _lib.ptr(go), parr, garr, Bt, G, n, 3 * n, L, _lib.ptr(gx), _lib.ptr(scratch),
                         _lib.current_stream(dev))
    _lib.check(rc, "srf_tac_bwd")
    return gx, grads


def wav_normalize(wav):
    """Per-row (x - mean) / (std + 1e-9), std unbiased (README.md:100-103).  wav [rows,T] or [Bt,1,T] ->
    (normalised wav of the same shape, stats [rows,2] = {mean, std})."""
    dev = _chk(wav)
    T = wav.shape[-1]
    rows = wav.numel() // T
    out = torch.empty_like(wav)
    stats = torch.empty((rows, 2), dtype=torch.float32, device=dev)
    rc = _lib.load().srf_wav_normalize(_lib.ptr(wav), _lib.ptr(out), _lib.ptr(stats), rows, T,
                                       _lib.current_stream(dev))
    _lib.check(rc, "srf_wav_normalize")
    return out, stats


def wav_denormalize(est, stats, mix_norm=None):
    """est * std + mean per example (README.md:108-109); with mix_norm [Bt,1,T] additionally
    mixture_consistency.apply(., mix_norm) (README.md:113-114).  est [Bt,S,T], stats [Bt,2]."""
    dev = _chk(est, stats, mix_norm)
    Bt, S, T = est.shape
    assert stats.shape == (Bt, 2)
    out = torch.empty_like(est)
    rc = _lib.load().srf_wav_denormalize(_lib.ptr(est), _lib.ptr(stats), _lib.ptr(mix_norm), _lib.ptr(out),
                                         Bt, S, T, _lib.current_stream(dev))
    _lib.check(rc, "srf_wav_denormalize")
    return out


def set_debug_flags(flags):
    _lib.load().srf_set_debug_flags(int(flags))


def set_kernel_mode(mode):
    """0 = fast paths, split-bf16x3 MFMA GEMMs (default); 1 = generic kernels only; 2 = fast paths with
    exact-fp32 MFMA GEMMs."""
    _lib.load().srf_set_kernel_mode(int(mode))


class kernel_trace:
    """Context manager over the in-library profiler (srf_profile_begin / _end / _get): ``with ops.kernel_trace(dev) as tr``
    records one (kernel family name, milliseconds) pair per launch this library makes on torch's current stream of `dev`
    inside the block; afterwards ``tr.launches`` holds them in launch order and ``tr.names`` the set of family names.
    Used by bench.py's per-kernel pass and by the tests that must prove WHICH kernel a shape was dispatched to."""

    def __init__(self, device):
        self.device = torch.device(device)
        self.launches, self.names = [], set()

    def __enter__(self):
        self._stream = _lib.current_stream(self.device)
        _lib.check(_lib.load().srf_profile_begin(self._stream), "srf_profile_begin")
        return self

    def __exit__(self, *exc):
        lib = _lib.load()
        cnt = C.c_int(0)
        _lib.check(lib.srf_profile_end(self._stream, C.byref(cnt)), "srf_profile_end")
        name, ms = C.c_char_p(), C.c_float()
        for i in range(cnt.value):
            lib.srf_profile_get(i, C.byref(name), C.byref(ms))
            if not name.value.startswith(b"("):          # "(gap)": host-side idle before a forward, not a kernel
                self.launches.append((name.value.decode(), ms.value))
        self.names = {n for n, _ in self.launches}
        return False
