"""Training graph of the ResNet encoder on the library's kernels: forward with batch-statistics BatchNorm and the full backward pass.

What `ResNetEncoder.forward` runs in train() mode under autograd (train/train_humaniflow.py:248-347 back-propagates the loss through
models/resnet.py).  One `torch.autograd.Function` spans the whole encoder, so no per-layer autograd bookkeeping happens between launches:

  forward, per conv + BN (+ ReLU) (+ residual) unit
      y    = conv(x, W)                    hmf_conv_winograd (stride-1 3x3: G W G^T of every layer rebuilt on the device by ONE launch at the top of the
                                           step, hmf_conv_winograd_pack_device_batch) | hmf_conv_patch (stride-2 3x3) | hmf_conv1x1_gemm /
                                           hmf_conv2d_nhwc (1x1) | hmf_stem_conv (7x7/2, filters re-packed on the device)
      stats                                hmf_bn_stats + hmf_bn_finalize (mean, rstd, scale, shift; running statistics updated as nn.BatchNorm2d does)
      out  = relu?(scale y + shift (+ idt)) hmf_bn_apply
  backward, per unit (g = dL/d out)
      gz = g * (out > 0); sums            hmf_bn_bwd_reduce + hmf_bn_bwd_finalize (d gamma, d beta, coefficients)
      gy = a gz + b y + c                  hmf_bn_bwd_apply (also emits gz = the identity branch's gradient)
      dW                                   hmf_conv_wgrad_winograd (stride-1 3x3: Winograd F(3x3, 2x2), written in the parameter's own layout) |
                                           hmf_conv_wgrad (MFMA GEMM over the pixel axis; the stem reads the NCHW input directly)
      dx                                   stride 1: the forward kernels with the flipped / transposed filter; stride-2 3x3: hmf_conv_dgrad_s2
                                           (four parity-class GEMMs in one launch, + the block's 1x1 downsample input gradient);
                                           the identity branch's gradient rides in that launch's residual epilogue
  max-pool: hmf_maxpool3x3s2_nhwc / hmf_maxpool3x3s2_nhwc_backward; global average pool: torch mean / broadcast (B x C values).
Filter re-layouts (flip, transpose, channels_last) are small torch copies per step (the filters total 45-94 MB).
"""
import os

import torch

from . import _lib

CL = torch.channels_last
WINOGRAD_WGRAD = True          # stride-1 3x3 weight gradients on hmf_conv_wgrad_winograd where the shape is built (False: hmf_conv_wgrad everywhere)
WGRAD_SIDE_STREAM = False      # weight gradients on a second stream beside the BN-backward / input-gradient chain (they only need gy and the saved
                               # input).  Measured on MI355X, B = 72, HIP-graph replay: 14.34 ms off, 15.12 ms on -- the co-running MFMA kernels
                               # (k_wgw 72 KB + k_conv_winograd 49-55 KB of LDS) evict each other from the CUs; kept for other shapes / parts
STRIDED_DGRAD = True           # stride-2 3x3 input gradients as parity-class GEMMs (hmf_conv_dgrad_s2); False: zero-inserted map + stride-1 convolution


def _cl_empty(B, C, H, W, device):
    return torch.empty((B, C, H, W), device=device, dtype=torch.float32, memory_format=CL)


def _conv_raw(x, w_cl, stride, residual=None):
    """Bias-free convolution on the implicit-GEMM kernel: x (B,Cin,H,W) channels_last, w_cl (Cout,Cin,k,k) channels_last -> (B,Cout,OH,OW)."""
    lib = _lib.load()
    B, Cin, H, W = x.shape
    Cout, _, ks, _ = w_cl.shape
    pad = ks // 2
    OH, OW = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    y = _cl_empty(B, Cout, OH, OW, x.device)
    if ks == 1 and stride == 1 and Cin % 32 == 0 and Cout % 64 == 0:
        # 1x1 layers and their input-gradient convolutions (the transposed filter): the LDS-DMA pipelined GEMM (csrc/conv_gemm.hip)
        _lib.check(lib.hmf_conv1x1_gemm(x.data_ptr(), None, w_cl.data_ptr(), None, None if residual is None else residual.data_ptr(), y.data_ptr(),
                                        B, H, W, Cin, 0, 0, 0, 1, Cout, 0, 0, _lib.current_stream()), 'hmf_conv1x1_gemm')
        return y
    _lib.check(lib.hmf_conv2d_nhwc(x.data_ptr(), w_cl.data_ptr(), None, None if residual is None else residual.data_ptr(), y.data_ptr(), B, H, W,
                                   Cin, Cout, ks, stride, 0, 0, _lib.current_stream()), 'hmf_conv2d_nhwc')
    return y


_PATCH_MAPS = {}


def _patch_packed(w4, cfg, native=None):
    """Packed B operand of hmf_conv_patch for a (Cout, Cin, 3, 3) filter, built on the device: the host packer only permutes, so packing an
    index-valued array once per (shape, cfg, device) gives a gather map (the filters change every optimiser step).
    native = (parameter, dgrad): w4 is the parameter itself (dgrad False) or its flipped, in/out-transposed view (dgrad True); the map is then
    composed with that view, so ONE gather reads the parameter's own storage (no permute / flip copy first)."""
    lib = _lib.load()
    Cout, Cin = w4.shape[:2]
    mode = 0 if native is None else (2 if native[1] else 1)
    key = (Cin, Cout, cfg, str(w4.device), mode)
    m = _PATCH_MAPS.get(key)
    if m is None:
        n = Cout * 9 * Cin
        idx = torch.arange(n, dtype=torch.float32)
        packed = torch.empty(int(lib.hmf_conv_patch_packed_floats(Cin, Cout, 3)), dtype=torch.float32)
        _lib.check(lib.hmf_conv_patch_pack_host(idx.data_ptr(), packed.data_ptr(), Cin, Cout, 3, cfg), 'hmf_conv_patch_pack_host')
        m = packed.long()                                              # position in (Cout, kh, kw, Cin) order
        if mode:
            src = torch.arange(n).reshape(native[0].shape)             # flat position in the parameter's (out, in, 3, 3) storage
            if mode == 2:
                src = src.flip(2, 3).transpose(0, 1)
            m = src.permute(0, 2, 3, 1).reshape(-1)[m]
        m = _PATCH_MAPS[key] = m.to(w4.device)
    if mode:
        return native[0].detach().reshape(-1)[m]
    return w4.permute(0, 2, 3, 1).reshape(-1)[m]                      # (Cout, kh, kw, Cin) order, gathered into MFMA lane order


def _conv3x3_patch(x, w4, stride, residual=None, native=None):
    """Bias-free 3x3 convolution on the patch-resident kernel (2x the implicit-GEMM rate on these shapes): w4 (Cout, Cin, 3, 3), any memory format."""
    lib = _lib.load()
    B, Cin, H, W = x.shape
    Cout = w4.shape[0]
    cfg = 8210 if (stride == 2 or Cin != 64) else 16410
    packed = _patch_packed(w4, cfg, native)
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    y = _cl_empty(B, Cout, OH, OW, x.device)
    _lib.check(lib.hmf_conv_patch(x.data_ptr(), packed.data_ptr(), None, None if residual is None else residual.data_ptr(), y.data_ptr(), B, H, W,
                                  Cin, Cout, 3, stride, 0, cfg, _lib.current_stream()), 'hmf_conv_patch')
    return y


_WINO_PACKS = {}               # (id(parameter), dgrad) -> (parameter version, data_ptr, packed filter): filled by _pack_all_winograd at the top of a step


def _pack_all_winograd(enc, device):
    """Transformed filters G g G^T of every stride-1 3x3 convolution of `enc`, for the layer itself and for its input-gradient convolution, in ONE
    launch (hmf_conv_winograd_pack_device_batch) into one buffer that lives with the module.  Entries carry the parameter's version and address:
    a filter that changed since (optimiser step, re-bound .data) is re-packed by its consumer."""
    lib = _lib.load()
    convs = enc.__dict__.get('_wino_convs')
    if convs is None:
        convs = enc.__dict__['_wino_convs'] = [m for m in enc.modules() if isinstance(m, torch.nn.Conv2d) and m.kernel_size == (3, 3) and m.stride == (1, 1)]
    jobs = []                                                        # (parameter, dgrad, Cin, Cout) in the kernel's sense
    for m in convs:
        w = m.weight
        if not (w.is_contiguous() and w.dtype == torch.float32):
            continue
        co, ci = w.shape[:2]
        if ci % 64 == 0 and co % 32 == 0:
            jobs.append((w, 0, ci, co))
        if co % 64 == 0 and ci % 32 == 0:
            jobs.append((w, 1, co, ci))
    if not jobs:
        return
    sizes = [int(lib.hmf_conv_winograd_packed_floats(ci, co)) for _, _, ci, co in jobs]
    key = '_wino_buf:%s' % device                  # a STRING key: nn.Module.__dir__ sorts the __dict__ keys
    buf = enc.__dict__.get(key)
    if buf is None or buf.numel() != sum(sizes):
        buf = enc.__dict__[key] = torch.empty(sum(sizes), dtype=torch.float32, device=device)
    import ctypes as C
    n = len(jobs)
    views, o = [], 0
    for sz in sizes:
        views.append(buf[o:o + sz])
        o += sz
    wp = (C.c_void_p * n)(*[j[0].detach().data_ptr() for j in jobs])
    pp = (C.c_void_p * n)(*[v.data_ptr() for v in views])
    ci = (C.c_int * n)(*[j[2] for j in jobs])
    co = (C.c_int * n)(*[j[3] for j in jobs])
    dg = (C.c_int * n)(*[j[1] for j in jobs])
    _lib.check(lib.hmf_conv_winograd_pack_device_batch(wp, pp, ci, co, dg, n, _lib.current_stream()), 'hmf_conv_winograd_pack_device_batch')
    keys = [(id(w), bool(d)) for w, d, _, _ in jobs]
    for k, (w, d, _, _), v in zip(keys, jobs, views):
        _WINO_PACKS[k] = (w._version, w.data_ptr(), v)
    if '_wino_finalizer' not in enc.__dict__:                        # the entries (and the buffer they view) go with the module
        import weakref
        enc.__dict__['_wino_finalizer'] = weakref.finalize(enc, _drop_wino_packs, keys)


def _drop_wino_packs(keys):
    for k in keys:
        _WINO_PACKS.pop(k, None)


def _conv3x3_winograd(x, native, residual=None):
    """Bias-free stride-1 3x3 convolution on the Winograd F(2x2,3x3) kernel (1.6-1.8x the patch kernel at B = 72): the transformed filters
    G g G^T are rebuilt on the device from the parameter's own storage (one launch for the whole encoder at the top of the step,
    _pack_all_winograd; here only if that copy is missing or stale), for the layer itself or, with native = (parameter, True), for its
    input-gradient convolution."""
    lib = _lib.load()
    param, dgrad = native
    B, Cin, H, W = x.shape
    Cout = param.shape[1] if dgrad else param.shape[0]
    st = _lib.current_stream()
    hit = _WINO_PACKS.get((id(param), bool(dgrad)))
    if hit is not None and hit[0] == param._version and hit[1] == param.data_ptr() and hit[2].device == x.device:
        packed = hit[2]
    else:
        packed = torch.empty(int(lib.hmf_conv_winograd_packed_floats(Cin, Cout)), dtype=torch.float32, device=x.device)
        _lib.check(lib.hmf_conv_winograd_pack_device(param.detach().data_ptr(), packed.data_ptr(), Cin, Cout, int(dgrad), st), 'hmf_conv_winograd_pack_device')
    y = _cl_empty(B, Cout, H, W, x.device)
    _lib.check(lib.hmf_conv_winograd(x.data_ptr(), packed.data_ptr(), None, None if residual is None else residual.data_ptr(), y.data_ptr(), B, H, W,
                                     Cin, Cout, 0, st), 'hmf_conv_winograd')
    return y


def _conv_any(x, w4, stride, residual=None, native=None):
    """Bias-free convolution, filter given in the parameter's (Cout, Cin, k, k) shape (any strides).  native: see _patch_packed."""
    if (w4.shape[2] == 3 and stride == 1 and native is not None and native[0].is_contiguous() and x.shape[1] % 64 == 0 and w4.shape[0] % 32 == 0
            and (x.shape[3] >= 16 or x.shape[3] == 8)):
        return _conv3x3_winograd(x, native, residual)
    if w4.shape[2] == 3 and x.shape[1] % 64 == 0 and w4.shape[0] % 64 == 0:
        return _conv3x3_patch(x, w4, stride, residual, native if (native is not None and native[0].is_contiguous()) else None)
    return _conv_raw(x, w4.contiguous(memory_format=CL), stride, residual)


class _ZeroPool:
    """fp64 accumulators of the BatchNorm reductions of one pass: one zero-filled buffer, handed out in (2, C) slices (one fill launch per pass
    instead of one per BatchNorm)."""

    def __init__(self, enc, device):
        n = sum(2 * m.num_features for m in enc.modules() if isinstance(m, torch.nn.BatchNorm2d))
        self.buf, self.o = torch.zeros(n, dtype=torch.float64, device=device), 0

    def take(self, C):
        v = self.buf[self.o:self.o + 2 * C].view(2, C)
        self.o += 2 * C
        return v


class _Saved:
    __slots__ = ('conv', 'bn', 'x', 'y', 'out', 'mean', 'rstd', 'relu', 'stride', 'stem', 'pooled', 'argmax')


# The stem's bn1 -> relu -> maxpool as one pass over the convolution's output (hmf_bn_relu_maxpool3x3s2_nhwc) with a two-pass backward: the activated
# 302 MB map of the training batch is never written or re-read.  HMF_STEM_TAIL=0: the three separate passes (A/B runs, tests).
STEM_TAIL_FUSED = os.environ.get('HMF_STEM_TAIL', '1') != '0'


def _bn_forward(y, bn, residual, relu, pool=None, tracked=None, fuse_maxpool=False):
    """Training-mode BatchNorm on the NHWC rows of y, running statistics updated in place.  Returns (out, mean, rstd).
    pool: _ZeroPool for the fp64 sums; tracked: list collecting num_batches_tracked tensors for ONE batched increment by the caller.
    fuse_maxpool: returns ((pooled, argmax), mean, rstd) instead -- relu(bn(y)) max-pooled 3x3 / 2 / pad 1 without materialising it."""
    lib = _lib.load()
    B, C, H, W = y.shape
    M = B * H * W
    dev = y.device
    acc = pool.take(C) if pool is not None else torch.zeros(2, C, dtype=torch.float64, device=dev)
    st = _lib.current_stream()
    _lib.check(lib.hmf_bn_stats(y.data_ptr(), M, C, acc.data_ptr(), st), 'hmf_bn_stats')
    mean, rstd, scale, shift = (torch.empty(C, dtype=torch.float32, device=dev) for _ in range(4))
    if bn.momentum is None:
        raise NotImplementedError('humaniflow_amd: BatchNorm2d(momentum=None) (cumulative moving average) is not built for the training graph; the '
                                  'reference uses the default momentum 0.1 (models/resnet.py)')
    if not bn.training:
        raise NotImplementedError('humaniflow_amd: a BatchNorm2d in eval() mode inside a train() encoder (frozen statistics) is not built for the training graph')
    mom = bn.momentum
    track = bn.track_running_stats and bn.running_mean is not None
    _lib.check(lib.hmf_bn_finalize(acc.data_ptr(), M, C, bn.weight.data_ptr(), bn.bias.data_ptr(), float(bn.eps), float(mom),
                                   bn.running_mean.data_ptr() if track else None, bn.running_var.data_ptr() if track else None,
                                   mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr(), st), 'hmf_bn_finalize')
    if track and bn.num_batches_tracked is not None:
        if tracked is not None:
            tracked.append(bn.num_batches_tracked)
        else:
            bn.num_batches_tracked += 1
    if fuse_maxpool:
        assert relu and residual is None
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        pooled = _cl_empty(B, C, OH, OW, dev)
        argmax = torch.empty(B, OH, OW, C, dtype=torch.uint8, device=dev)
        _lib.check(lib.hmf_bn_relu_maxpool3x3s2_nhwc(y.data_ptr(), scale.data_ptr(), shift.data_ptr(), pooled.data_ptr(), argmax.data_ptr(), B, H, W, C, st),
                   'hmf_bn_relu_maxpool3x3s2_nhwc')
        return (pooled, argmax), mean, rstd
    out = _cl_empty(B, C, H, W, dev)
    _lib.check(lib.hmf_bn_apply(y.data_ptr(), scale.data_ptr(), shift.data_ptr(), None if residual is None else residual.data_ptr(), out.data_ptr(),
                                M, C, int(relu), st), 'hmf_bn_apply')
    return out, mean, rstd


def _bn_pool_backward(g, s, pool=None):
    """Backward of the fused stem tail: g = dL/d pooled map -> (gy = dL/d conv output, d gamma, d beta)."""
    lib = _lib.load()
    B, C, H, W = s.y.shape
    dev = g.device
    st = _lib.current_stream()
    acc = pool.take(C) if pool is not None else torch.zeros(2, C, dtype=torch.float64, device=dev)
    _lib.check(lib.hmf_bn_relu_maxpool3x3s2_nhwc_bwd_reduce(g.data_ptr(), s.argmax.data_ptr(), s.pooled.data_ptr(), s.y.data_ptr(), s.mean.data_ptr(),
                                                            s.rstd.data_ptr(), B, H, W, C, acc.data_ptr(), st), 'hmf_bn_relu_maxpool3x3s2_nhwc_bwd_reduce')
    gg, gb, ca, cb, cc = (torch.empty(C, dtype=torch.float32, device=dev) for _ in range(5))
    _lib.check(lib.hmf_bn_bwd_finalize(acc.data_ptr(), B * H * W, C, s.bn.weight.data_ptr(), s.mean.data_ptr(), s.rstd.data_ptr(), gg.data_ptr(), gb.data_ptr(),
                                       ca.data_ptr(), cb.data_ptr(), cc.data_ptr(), st), 'hmf_bn_bwd_finalize')
    gy = _cl_empty(B, C, H, W, dev)
    _lib.check(lib.hmf_bn_relu_maxpool3x3s2_nhwc_bwd_apply(g.data_ptr(), s.argmax.data_ptr(), s.pooled.data_ptr(), s.y.data_ptr(), ca.data_ptr(), cb.data_ptr(),
                                                           cc.data_ptr(), gy.data_ptr(), B, H, W, C, st), 'hmf_bn_relu_maxpool3x3s2_nhwc_bwd_apply')
    return gy, gg, gb


def _bn_backward(g, s, want_gz, pool=None):
    """g = dL/d out of unit s -> (gy = dL/d conv output, gz or None, d gamma, d beta)."""
    lib = _lib.load()
    B, C, H, W = s.y.shape
    M = B * H * W
    dev = g.device
    st = _lib.current_stream()
    acc = pool.take(C) if pool is not None else torch.zeros(2, C, dtype=torch.float64, device=dev)
    out_ptr = s.out.data_ptr() if s.relu else None
    _lib.check(lib.hmf_bn_bwd_reduce(g.data_ptr(), out_ptr, s.y.data_ptr(), s.mean.data_ptr(), s.rstd.data_ptr(), M, C, int(s.relu), acc.data_ptr(), st),
               'hmf_bn_bwd_reduce')
    gg, gb, ca, cb, cc = (torch.empty(C, dtype=torch.float32, device=dev) for _ in range(5))
    _lib.check(lib.hmf_bn_bwd_finalize(acc.data_ptr(), M, C, s.bn.weight.data_ptr(), s.mean.data_ptr(), s.rstd.data_ptr(), gg.data_ptr(), gb.data_ptr(),
                                       ca.data_ptr(), cb.data_ptr(), cc.data_ptr(), st), 'hmf_bn_bwd_finalize')
    gy = _cl_empty(B, C, H, W, dev)
    gz = _cl_empty(B, C, H, W, dev) if want_gz else None
    _lib.check(lib.hmf_bn_bwd_apply(g.data_ptr(), out_ptr, s.y.data_ptr(), ca.data_ptr(), cb.data_ptr(), cc.data_ptr(), gy.data_ptr(),
                                    None if gz is None else gz.data_ptr(), M, C, int(s.relu), st), 'hmf_bn_bwd_apply')
    return gy, gz, gg, gb


def _wgrad(gy, s):
    """dL/dW of unit s in the parameter's own (Cout, Cin, k, k) shape."""
    lib = _lib.load()
    Cout, Cin, ks, _ = s.conv.weight.shape
    if s.stem:
        B, _, H, W = s.x.shape                                        # NCHW proxy input
        dw = torch.empty(Cout, Cin, ks, ks, dtype=torch.float32, device=gy.device)
        _lib.check(lib.hmf_conv_wgrad(gy.data_ptr(), s.x.data_ptr(), dw.data_ptr(), B, H, W, Cin, Cout, ks, s.stride, 1, _lib.current_stream()), 'hmf_conv_wgrad')
        return dw
    B, _, H, W = s.x.shape
    if ks == 3 and s.stride == 1 and WINOGRAD_WGRAD and lib.hmf_conv_wgrad_winograd_supported(H, W, Cin, Cout):
        # stride-1 3x3 layers: Winograd F(3x3, 2x2) weight gradient, written in the parameter's (Cout, Cin, 3, 3) layout (no permute pass)
        dw = torch.empty(Cout, Cin, 3, 3, dtype=torch.float32, device=gy.device)
        ws = torch.empty(int(lib.hmf_conv_wgrad_winograd_workspace_floats(Cin, Cout)), dtype=torch.float32, device=gy.device)
        _lib.check(lib.hmf_conv_wgrad_winograd(gy.data_ptr(), s.x.data_ptr(), dw.data_ptr(), ws.data_ptr(), B, H, W, Cin, Cout, _lib.current_stream()),
                   'hmf_conv_wgrad_winograd')
        return dw
    dw = torch.empty(Cout, ks, ks, Cin, dtype=torch.float32, device=gy.device)
    _lib.check(lib.hmf_conv_wgrad(gy.data_ptr(), s.x.data_ptr(), dw.data_ptr(), B, H, W, Cin, Cout, ks, s.stride, 0, _lib.current_stream()), 'hmf_conv_wgrad')
    return dw.permute(0, 3, 1, 2).contiguous()


def _dgrad(gy, s, residual=None, extra=None):
    """dL/dx of unit s (+ residual): a convolution of gy with the flipped, in/out-transposed filter.  extra = (gy2, unit2): + dL/dx of the block's
    stride-2 1x1 downsample unit in the same launch (stride-2 3x3 units only)."""
    w = s.conv.weight.detach()
    ks = w.shape[2]
    B, Cin, H, W = s.x.shape
    Cout = w.shape[0]
    if s.stride == 2 and ks == 3 and STRIDED_DGRAD and Cout % 32 == 0 and Cin % 64 == 0:
        # four parity-class GEMMs in one launch (csrc/conv_dgrad_s2.hip): no zero-inserted map, a quarter of its MFMA work
        lib = _lib.load()
        wt = w.permute(1, 2, 3, 0).contiguous()                      # (Cin, 3, 3, Cout)
        gy2 = wt2 = None
        Co2 = 0
        if extra is not None:
            gy2, s2 = extra
            Co2 = s2.conv.weight.shape[0]
            wt2 = s2.conv.weight.detach().reshape(Co2, Cin).t().contiguous()
        gx = _cl_empty(B, Cin, H, W, gy.device)
        _lib.check(lib.hmf_conv_dgrad_s2(gy.data_ptr(), wt.data_ptr(), None if gy2 is None else gy2.data_ptr(), None if wt2 is None else wt2.data_ptr(),
                                         None if residual is None else residual.data_ptr(), gx.data_ptr(), B, H, W, Cout, Co2, Cin, _lib.current_stream()),
                   'hmf_conv_dgrad_s2')
        return gx
    assert extra is None
    nat = (s.conv.weight, True)
    if (s.stride == 1 and ks == 3 and s.conv.weight.is_contiguous() and gy.shape[1] % 64 == 0 and Cin % 32 == 0
            and (gy.shape[3] >= 16 or gy.shape[3] == 8)):
        return _conv3x3_winograd(gy, nat, residual)                    # the Winograd pack reads the parameter itself: no flipped copy (torch.flip is a launch)
    if s.stride == 1 and ks == 3 and s.conv.weight.is_contiguous() and gy.shape[1] % 64 == 0 and Cin % 64 == 0:
        return _conv3x3_patch(gy, w.transpose(0, 1), 1, residual, nat)  # the gather map is composed with the flip: only the view's shape is read
    wt = w.flip(2, 3).transpose(0, 1) if ks > 1 else w.transpose(0, 1)          # (Cin, Cout, k, k) view: flipped, in/out swapped
    if s.stride == 1:
        return _conv_any(gy, wt, 1, residual, native=nat)
    if ks == 1:                                                       # strided 1x1 (downsample branch): only the sampled pixels receive gradient
        gx = torch.zeros((B, Cin, H, W), device=gy.device, dtype=torch.float32).contiguous(memory_format=CL) if residual is None else residual.clone()
        c = _conv_any(gy, wt, 1)
        gx[:, :, ::s.stride, ::s.stride] += c
        return gx
    gd = torch.zeros((B, gy.shape[1], H, W), device=gy.device, dtype=torch.float32).contiguous(memory_format=CL)    # zero-inserted gradient map
    gd[:, :, ::s.stride, ::s.stride] = gy
    return _conv_any(gd, wt, 1, residual, native=nat)


def _folds_downsample(main0, ds):
    """The block's first unit is a stride-2 3x3 convolution of the block input and its downsample branch a stride-2 1x1 of the same input
    (BasicBlock of layer2-4.0): both input gradients come out of one hmf_conv_dgrad_s2 launch."""
    w, w2 = main0.conv.weight, ds.conv.weight
    return (STRIDED_DGRAD and main0.stride == 2 and ds.stride == 2 and w.shape[2] == 3 and w2.shape[2] == 1 and main0.x is ds.x
            and w.shape[0] % 32 == 0 and w2.shape[0] % 32 == 0 and w.shape[1] % 64 == 0)


_SIDE_STREAMS = {}


def _side_stream(device):
    key = (device.type, device.index)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


class _WgradIssuer:
    """Issues the weight-gradient launches of one backward pass on a side stream: dW of a unit depends on gy and the saved input only, while the
    critical path of the pass is BN backward -> input gradient -> the next unit's BN backward.  Off by default (WGRAD_SIDE_STREAM: slower on
    MI355X at the training batch).  Fork / join by stream
    waits, which a HIP-graph capture of the step records as graph edges; `record_stream` keeps the allocator from recycling a tensor's
    block before the other stream is done with it."""

    def __init__(self, device):
        self.on = WGRAD_SIDE_STREAM
        self.main = torch.cuda.current_stream(device)
        self.side = _side_stream(device) if self.on else None

    def __call__(self, gy, s):
        if not self.on:
            return _wgrad(gy, s)
        self.side.wait_stream(self.main)                              # gy was produced on the main stream
        with torch.cuda.stream(self.side):
            dw = _wgrad(gy, s)
        gy.record_stream(self.side)
        dw.record_stream(self.main)
        return dw

    def join(self):
        if self.on:
            self.main.wait_stream(self.side)


class EncoderTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, enc, x, *params):
        lib = _lib.load()
        dev = x.device
        saved = []
        pool, tracked = _ZeroPool(enc, dev), []

        def unit(conv, bn, xin, relu, residual=None, stem=False):
            s = _Saved()
            s.conv, s.bn, s.x, s.relu, s.stride, s.stem = conv, bn, xin, relu, conv.stride[0], stem
            if stem:
                B, C, H, W = xin.shape
                OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
                s.y = _cl_empty(B, 64, OH, OW, dev)
                if enc.stem_polyphase:                                # csrc/stem_wino.hip: transformed filters rebuilt from the live parameter by one small launch
                    packed = torch.empty(int(lib.hmf_stem_wino_packed_floats()), dtype=torch.float32, device=dev)
                    _lib.check(lib.hmf_stem_wino_pack_device(conv.weight.detach().data_ptr(), packed.data_ptr(), _lib.current_stream()), 'hmf_stem_wino_pack_device')
                    _lib.check(lib.hmf_stem_wino(xin.data_ptr(), packed.data_ptr(), None, s.y.data_ptr(), B, C, H, W, 0, _lib.current_stream()), 'hmf_stem_wino')
                else:
                    packed = enc._stem_packed_from_device(conv.weight.detach())
                    _lib.check(lib.hmf_stem_conv(xin.data_ptr(), packed.data_ptr(), None, s.y.data_ptr(), B, C, H, W, 0, _lib.current_stream()), 'hmf_stem_conv')
            else:
                s.y = _conv_any(xin, conv.weight.detach(), s.stride, native=(conv.weight, False))
            s.pooled = s.argmax = None
            if stem and STEM_TAIL_FUSED and s.y.shape[1] % 64 == 0:
                (s.pooled, s.argmax), s.mean, s.rstd = _bn_forward(s.y, bn, None, True, pool, tracked, fuse_maxpool=True)
                s.out = None
                saved.append(s)
                return s.pooled
            s.out, s.mean, s.rstd = _bn_forward(s.y, bn, residual, relu, pool, tracked)
            saved.append(s)
            return s.out

        with torch.cuda.device(dev), torch.no_grad():
            _pack_all_winograd(enc, dev)
            xin = x.detach().float().contiguous()
            a0 = unit(enc.conv1, enc.bn1, xin, True, stem=True)
            if saved[0].pooled is not None:                           # bn1 + relu + maxpool in one pass: `a0` is already the pooled map
                p0, a0 = a0, None
            else:
                B, C, H, W = a0.shape
                p0 = _cl_empty(B, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1, dev)
                _lib.check(lib.hmf_maxpool3x3s2_nhwc(a0.data_ptr(), p0.data_ptr(), B, H, W, C, _lib.current_stream()), 'hmf_maxpool3x3s2_nhwc')
            cur = p0
            plan = []
            for li in range(1, 5):
                for blk in getattr(enc, 'layer%d' % li):
                    first = len(saved)
                    idt = cur
                    if blk.downsample is not None:
                        idt = unit(blk.downsample[0], blk.downsample[1], cur, False)
                    h = cur
                    st = blk.stages()
                    for conv, bn in st[:-1]:
                        h = unit(conv, bn, h, True)
                    conv, bn = st[-1]
                    cur = unit(conv, bn, h, True, residual=idt)
                    plan.append((first, len(saved), blk.downsample is not None))
            feats = cur.mean(dim=(2, 3))
            if tracked:
                import ctypes as C
                _lib.check(lib.hmf_i64_increment_batch((C.c_void_p * len(tracked))(*[t.data_ptr() for t in tracked]), len(tracked), _lib.current_stream()),
                           'hmf_i64_increment_batch')
        ctx.enc, ctx.saved, ctx.plan, ctx.a0, ctx.final_shape = enc, saved, plan, a0, tuple(cur.shape)
        ctx.param_ids = [id(p) for p in params]
        # backward reads the LIVE conv / BatchNorm weights (input-gradient filters, BN coefficients): remember their versions so that an in-place
        # update between forward and backward (two forwards then one step, an EMA swap) raises as torch autograd would, instead of silently
        # differentiating a different function
        ctx.param_versions = [p._version for p in params]
        ctx.params_ref = params
        return feats

    @staticmethod
    def backward(ctx, g_feats):
        lib = _lib.load()
        if ctx.saved is None:
            raise RuntimeError('humaniflow_amd: the encoder training graph was already back-propagated; its saved activations are freed after the '
                               'first backward (retain_graph is not supported for EncoderTrainFn)')
        for p, v in zip(ctx.params_ref, ctx.param_versions):
            if p._version != v:
                raise RuntimeError('humaniflow_amd: an encoder parameter was modified in place between the forward and the backward of the training '
                                   'graph (version %d -> %d); the backward reads the live weights' % (v, p._version))
        enc, saved, plan = ctx.enc, ctx.saved, ctx.plan
        grads = {}
        dev = g_feats.device
        with torch.cuda.device(dev), torch.no_grad():
            pool = _ZeroPool(enc, dev)
            wgrad = _WgradIssuer(dev)
            B, C, H, W = ctx.final_shape
            g = (g_feats.detach().float() / (H * W))[:, :, None, None].expand(B, C, H, W).contiguous(memory_format=CL)
            for first, last, has_ds in reversed(plan):
                units = saved[first:last]
                ds = units[0] if has_ds else None
                main = units[1:] if has_ds else units
                # last unit of the block: out = relu(bn(conv(h)) + idt)
                s = main[-1]
                gy, gz, gg, gb = _bn_backward(g, s, want_gz=True, pool=pool)
                grads[id(s.bn.weight)], grads[id(s.bn.bias)] = gg, gb
                grads[id(s.conv.weight)] = wgrad(gy, s)
                g_idt = gz                                            # gradient of the identity branch (masked by the block's ReLU)
                only = len(main) == 1
                g_h = _dgrad(gy, s, residual=g_idt if (only and ds is None) else None)
                fold = ds is not None and len(main) >= 2 and _folds_downsample(main[0], ds)
                gy_ds = None
                if fold:                                              # downsample branch first: its input gradient rides in the main path's launch
                    gy_ds, _, gg, gb = _bn_backward(g_idt, ds, want_gz=False, pool=pool)
                    grads[id(ds.bn.weight)], grads[id(ds.bn.bias)] = gg, gb
                    grads[id(ds.conv.weight)] = wgrad(gy_ds, ds)
                for k in range(len(main) - 2, -1, -1):
                    s = main[k]
                    gy, _, gg, gb = _bn_backward(g_h, s, want_gz=False, pool=pool)
                    grads[id(s.bn.weight)], grads[id(s.bn.bias)] = gg, gb
                    grads[id(s.conv.weight)] = wgrad(gy, s)
                    if k == 0 and fold:
                        g_h = _dgrad(gy, s, extra=(gy_ds, ds))
                    else:
                        g_h = _dgrad(gy, s, residual=g_idt if (k == 0 and ds is None) else None)     # block input: main path + identity in one launch
                if ds is not None and not fold:
                    gy, _, gg, gb = _bn_backward(g_idt, ds, want_gz=False, pool=pool)
                    grads[id(ds.bn.weight)], grads[id(ds.bn.bias)] = gg, gb
                    grads[id(ds.conv.weight)] = wgrad(gy, ds)
                    g_h = _dgrad(gy, ds, residual=g_h)
                g = g_h
            # max-pool and stem
            s = saved[0]
            if s.pooled is not None:
                gy, gg, gb = _bn_pool_backward(g, s, pool=pool)
            else:
                a0 = ctx.a0
                B, C, H, W = a0.shape
                g_a0 = _cl_empty(B, C, H, W, dev)
                _lib.check(lib.hmf_maxpool3x3s2_nhwc_backward(a0.data_ptr(), g.data_ptr(), g_a0.data_ptr(), B, H, W, C, _lib.current_stream()),
                           'hmf_maxpool3x3s2_nhwc_backward')
                gy, _, gg, gb = _bn_backward(g_a0, s, want_gz=False, pool=pool)
            grads[id(s.bn.weight)], grads[id(s.bn.bias)] = gg, gb
            grads[id(s.conv.weight)] = wgrad(gy, s)
            wgrad.join()
        ctx.saved = None
        ctx.params_ref = None
        return (None, None) + tuple(grads.get(i) for i in ctx.param_ids)
