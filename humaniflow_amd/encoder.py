"""ResNet-18/50 image encoder (no classifier head) with the reference's state-dict names.

Role of models/resnet.py in the reference (ResNet.forward :202-217, resnet18/50 :229,251):
(B, C_in, 256, 256) proxy representation -> (B, 512|2048) pooled features.  Parameter and
buffer names follow torchvision (`conv1`, `bn1`, `layerL.B.convK/bnK`, `layerL.B.downsample.{0,1}`)
so a reference checkpoint loads with ``strict=True``.

On the GPU the eval-mode forward runs an *inference plan*: eval-mode BatchNorm folded into the
preceding convolution's weight plus a per-channel bias, activations kept NHWC, and every convolution
executed by a hand-written fp32-MFMA kernel of the library with bias + residual + ReLU in its epilogue
(one launch and one HBM write per convolution): hmf_stem_conv (7x7/2, reads the NCHW input directly) +
hmf_maxpool3x3s2_nhwc, hmf_conv_winograd (stride-1 3x3), hmf_conv_patch (stride-2 3x3), hmf_conv2d_nhwc /
hmf_conv1x1_dual_nhwc (1x1; a downsample branch rides in its block's conv3 as a K-concatenated GEMM).
The plan is built lazily from the live parameters and rebuilt when they change (parameter version
counters), so the checkpoint layout is untouched.  There is no vendor-library inference path
and no CPU path: `forward` on a CPU tensor raises.  Training (train() under autograd) uses batch-statistics
BatchNorm, for which the folded plan does not apply: that mode runs encoder_train.py -- forward with hmf_bn_stats /
hmf_bn_apply around the same convolution kernels and a hand-written backward (hmf_bn_bwd_*, hmf_conv_wgrad, dgrad
through hmf_conv2d_nhwc with transformed filters, hmf_maxpool3x3s2_nhwc_backward).
"""
import torch
from torch import nn
import torch.nn.functional as F


_SPECS = {18: (False, [2, 2, 2, 2]), 34: (False, [3, 4, 6, 3]), 50: (True, [3, 4, 6, 3])}


def _conv(cin, cout, k, stride=1):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=k // 2, bias=False)


class _Residual(nn.Module):
    """BasicBlock (two 3x3) or Bottleneck (1x1 -> 3x3(stride) -> 1x1 x4)."""

    def __init__(self, cin, width, stride, bottleneck):
        super().__init__()
        self.bottleneck = bottleneck
        cout = width * (4 if bottleneck else 1)
        if bottleneck:
            self.conv1, self.bn1 = _conv(cin, width, 1), nn.BatchNorm2d(width)
            self.conv2, self.bn2 = _conv(width, width, 3, stride), nn.BatchNorm2d(width)
            self.conv3, self.bn3 = _conv(width, cout, 1), nn.BatchNorm2d(cout)
        else:
            self.conv1, self.bn1 = _conv(cin, width, 3, stride), nn.BatchNorm2d(width)
            self.conv2, self.bn2 = _conv(width, cout, 3), nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(_conv(cin, cout, 1, stride), nn.BatchNorm2d(cout))

    def stages(self):
        s = [(self.conv1, self.bn1), (self.conv2, self.bn2)]
        if self.bottleneck:
            s.append((self.conv3, self.bn3))
        return s

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        st = self.stages()
        for conv, bn in st[:-1]:
            x = F.relu(bn(conv(x)))
        conv, bn = st[-1]
        return F.relu(bn(conv(x)) + idt)


class ResNetEncoder(nn.Module):
    def __init__(self, num_layers, in_channels):
        super().__init__()
        bottleneck, depths = _SPECS[num_layers]
        self.out_dim = 512 * (4 if bottleneck else 1)
        self.conv1 = nn.Conv2d(in_channels, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        cin = 64
        for li, (width, n) in enumerate(zip((64, 128, 256, 512), depths)):
            blocks = []
            for bi in range(n):
                blocks.append(_Residual(cin, width, 2 if (li > 0 and bi == 0) else 1, bottleneck))
                cin = width * (4 if bottleneck else 1)
            setattr(self, 'layer%d' % (li + 1), nn.Sequential(*blocks))
        for m in self.modules():                                  # models/resnet.py:162-167
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        self._plan = None
        self._plan_key = None
        self.__dict__['_plan_tensors'] = None
        self.__dict__['_stem_ws'] = {}

    # ---- plain module graph: batch-statistics BatchNorm under autograd (training), and the float64 reference of the tests ----
    def forward_module(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x))), kernel_size=3, stride=2, padding=1)
        for li in range(1, 5):
            x = getattr(self, 'layer%d' % li)(x)
        return torch.flatten(F.adaptive_avg_pool2d(x, 1), 1)

    # ---- folded inference plan ----------------------------------------------------------
    # eval-mode BatchNorm folded into the conv weight + a per-channel bias that the convolution kernel adds in its epilogue; a
    # downsample branch's bias is added to the block's last bias so that branch needs no pass of its own.
    fuse_stem_pool = True    # stem convolution + max-pool as one launch (hmf_stem_conv_pool / hmf_stem_wino_pool)
    patch_s2_cfg = 8210      # tiling of the stride-2 3x3 layers (hmf_conv_patch): 8210 = 64-channel slabs, 8211 = 32-channel slabs (half the LDS)
    stem_polyphase = True    # the stem as a polyphase fast convolution (csrc/stem_wino.hip: 100 instead of 196 products per 2 x 2 outputs and channel,
                             # F(2,3)-class arithmetic); False: the direct form (csrc/stem_conv.hip)
    fuse_pairs = True        # a bottleneck's last 1x1 + the next bottleneck's first 1x1 as one launch where built (hmf_conv_pair: layer1 / layer2)
    fuse_pairs_wide = False  # also the seams with a 128-channel input (layer1.0's K-concatenated downsample branch, layer2): slower than two launches
                             # of hmf_conv1x1_gemm when run alone (5-12 us per seam); see _build_plan
    fuse_downsample = True   # bottleneck conv3 + downsample branch as one K-concatenated GEMM (hmf_conv1x1_gemm with a second K range)
    gemm_1x1 = True          # stride-1 1x1 layers on the LDS-DMA pipelined GEMM (hmf_conv1x1_gemm); False: the implicit-GEMM kernel (hmf_conv2d_nhwc)
    gemm_terms = 0           # OPT-IN: 3 / 6 = the 1x1 GEMMs on the bf16 matrix pipe with split fp32 operands (hmf_conv1x1_gemm_split: 3 = hi/lo
                             # products, ~2^-16 relative per product; 6 = three-piece products, fp32-level); 0 = fp32 MFMA (the default everywhere)

    def _fold(self, conv, bn):
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        w = (conv.weight * scale[:, None, None, None]).float().contiguous(memory_format=torch.channels_last)
        b = (bn.bias - bn.running_mean * scale).float().contiguous()
        return w, b, conv.stride, conv.padding

    # The plan is a cache of the live parameters and BatchNorm statistics.  It is dropped by train()/eval(), .to()/.cuda() and
    # load_state_dict, and -- for in-place updates such as optimizer.step(), p.data.copy_() or in-place EMA swaps -- re-validated on
    # every forward against the tensors' version counters (_params_key).  Re-binding `p.data = other` does not bump a version
    # counter: call invalidate_plan() after that.
    def invalidate_plan(self):
        self._plan = None
        self.__dict__['_plan_tensors'] = None

    def _params_key(self):
        ts = self.__dict__.get('_plan_tensors')
        if ts is None:                                        # tensor list cached with the plan (~25 us per call for ResNet-50)
            ts = self.__dict__['_plan_tensors'] = list(self.parameters()) + list(self.buffers())
        return tuple([t._version for t in ts])

    def train(self, mode=True):
        self.invalidate_plan()
        return super().train(mode)

    def _apply(self, fn, *a, **k):
        self.invalidate_plan()
        return super()._apply(fn, *a, **k)

    def _load_from_state_dict(self, *a, **k):
        self.invalidate_plan()
        return super()._load_from_state_dict(*a, **k)

    def __getstate__(self):                                   # the packed device copies are a cache: never pickled / deep-copied
        st = self.__dict__.copy()
        st['_plan'] = None
        st['_plan_key'] = None
        st['_plan_tensors'] = None
        st.pop('_stem_map', None)
        st['_stem_ws'] = {}
        for k in [k for k in st if k.startswith('_wino')]:     # training-graph filter packs (encoder_train.py)
            st.pop(k)
        return st

    @torch.no_grad()
    def _build_plan(self):
        plan = {'stem': self._fold(self.conv1, self.bn1), 'blocks': []}
        if self.conv1.weight.is_cuda:
            if self.conv1.in_channels != 18:
                raise NotImplementedError('humaniflow_amd: the stem kernel is built for the 18-channel proxy representation '
                                          '(NUM_IN_CHANNELS = 18), got %d' % self.conv1.in_channels)
            from . import _lib
            lib = _lib.load()
            w = plan['stem'][0].detach().cpu().contiguous()                      # (64,18,7,7) row-major, BN scale folded in
            packed = torch.empty(int(lib.hmf_stem_packed_floats()), dtype=torch.float32)
            _lib.check(lib.hmf_stem_pack_host(w.data_ptr(), packed.data_ptr()), 'hmf_stem_pack_host')
            plan['stem_packed'] = packed.to(self.conv1.weight.device)
            packed = torch.empty(int(lib.hmf_stem_wino_packed_floats()), dtype=torch.float32)
            _lib.check(lib.hmf_stem_wino_pack_host(w.data_ptr(), packed.data_ptr()), 'hmf_stem_wino_pack_host')
            plan['stem_packed_polyphase'] = packed.to(self.conv1.weight.device)
        for li in range(1, 5):
            for blk in getattr(self, 'layer%d' % li):
                stages = [self._fold(c, b) for c, b in blk.stages()]
                ds = None
                if blk.downsample is not None:
                    ds = self._fold(blk.downsample[0], blk.downsample[1])
                    w, b, s, p = stages[-1]
                    stages[-1] = (w, (b + ds[1]).contiguous(), s, p)
                dual = None
                if ds is not None and self.fuse_downsample and stages[-1][0].shape[2] == 1:
                    # last 1x1 conv + downsample branch as one GEMM over the concatenated K: W = [W_conv3 | W_downsample]
                    w3, wd = stages[-1][0], ds[0]
                    wcat = torch.cat([w3.reshape(w3.shape[0], -1), wd.reshape(wd.shape[0], -1)], dim=1).contiguous()
                    dual = (wcat, stages[-1][1], ds[2][0])
                plan['blocks'].append((stages, ds, dual))
        plan['packed'], plan['wino'], plan['pair'], plan['split'] = {}, {}, {}, {}
        if self.gemm_terms not in (0, 3, 6):
            raise ValueError('humaniflow_amd: gemm_terms = %r (0 = fp32 MFMA, 3 or 6 = split bf16 operands)' % (self.gemm_terms,))
        if self.conv1.weight.is_cuda and self.gemm_terms and self.gemm_1x1:
            # split every GEMM-shaped 1x1 filter once (the seam kernel is fp32 only: no seams in this mode)
            for stages, ds, dual in plan['blocks']:
                for w in [st[0] for st in (stages[:-1] if dual is not None else stages)] + ([dual[0]] if dual is not None else []):
                    w2 = w.reshape(w.shape[0], -1) if (w.dim() == 2 or w.shape[2] == 1) else None
                    if w2 is not None and w2.shape[0] % 64 == 0 and w2.shape[1] % 32 == 0:
                        plan['split'][id(w)] = self._pack_split(w2)
        if self.conv1.weight.is_cuda and self.fuse_pairs and not plan['split']:
            # block i's last 1x1 (+ downsample branch) and block i+1's first 1x1 as one launch where the kernel is built for the shapes
            # (layer1 and layer2 of ResNet-50: the 256 / 512-channel tensor between them is what bounds both layers)
            blocks = plan['blocks']
            for i in range(len(blocks) - 1):
                stages, ds, dual = blocks[i]
                w3, b3, s3, _ = stages[-1]
                w1, b1, s1, _ = blocks[i + 1][0][0]
                if len(stages) != 3 or w3.shape[2] != 1 or w1.shape[2] != 1 or s3[0] != 1 or s1[0] != 1 or (ds is not None and dual is None):
                    continue
                wk = dual[0] if dual is not None else w3.reshape(w3.shape[0], -1)
                C, Kt, N2 = wk.shape[0], wk.shape[1], w1.shape[0]
                if Kt not in (64, 128) or C % 64 or N2 not in (64, 128, 256) or C > 512 or (dual is not None and dual[2] != 1):
                    continue
                if self.gemm_1x1 and Kt != 64 and not self.fuse_pairs_wide:
                    # measured at B = 32 against two launches of hmf_conv1x1_gemm (tools/gemm1x1_probe.py, tools/encoder_layers.py): the seam
                    # launch wins only where the wide tensor between the two layers bounds them through HBM (layer1: conv3 64 -> 256 + identity
                    # at 64x64, 73.8 + 39.3 us as two launches vs 103 us fused); with a 128-channel input (layer1.0's K-concatenated downsample
                    # branch, all of layer2) two launches are 5 - 12 us faster than the seam kernel (82 vs 87 us, 113 vs 125 us, 122 vs 134 us)
                    continue
                plan['pair'][i] = self._pack_pair(wk, w1.reshape(N2, C)) + (N2,)
        if self.conv1.weight.is_cuda:
            for stages, ds, dual in plan['blocks']:
                for w, b, s, p in stages:
                    cfg = self._patch_cfg(w)
                    if cfg and s[0] == 2:
                        cfg = self.patch_s2_cfg
                    if cfg and s[0] == 1:
                        # stride-1 3x3 layers run as Winograd F(2x2,3x3) on maps >= 16 pixels wide (8x16-pixel tiles) and on 8-pixel maps (two
                        # images per tile); the patch kernel stays packed too for other widths (chosen per call from the activation width)
                        plan['wino'][id(w)] = self._pack_winograd(w)
                    if cfg:
                        pk = self._pack_patch(w, cfg)
                        plan['packed'][id(w)] = (cfg, pk)              # the packing depends only on the slab size (64 for every 3x3 cfg)
        return plan

    @staticmethod
    def _patch_cfg(w):
        """Kernel choice per layer, measured on MI355X (tools/conv_probe.py): 3x3 layers run patch-resident (8x16-pixel x 32-channel
        tiles for the 64-channel layers, 8x8 x 64 otherwise); 1x1 layers stay on the implicit-GEMM kernel (equal or faster there)."""
        cout, cin, ks, _ = w.shape
        if ks != 3 or cin % 64 or cout % 64:
            return 0
        return 16410 if cin == 64 else 8210

    def _pack_patch(self, w, cfg):
        from . import _lib
        lib = _lib.load()
        cout, cin, ks, _ = w.shape
        wh = w.detach().permute(0, 2, 3, 1).contiguous().cpu()               # (Cout, ks, ks, Cin)
        packed = torch.empty(int(lib.hmf_conv_patch_packed_floats(cin, cout, ks)), dtype=torch.float32)
        _lib.check(lib.hmf_conv_patch_pack_host(wh.data_ptr(), packed.data_ptr(), cin, cout, ks, cfg), 'hmf_conv_patch_pack_host')
        return packed.to(w.device)

    def _pack_split(self, w2):
        from . import _lib
        lib = _lib.load()
        N, K = w2.shape
        w2 = w2.detach().float().contiguous()
        packed = torch.empty(int(lib.hmf_conv1x1_split_packed_bytes(N, K, self.gemm_terms)), dtype=torch.uint8, device=w2.device)
        with torch.cuda.device(w2.device):
            _lib.check(lib.hmf_conv1x1_split_pack(w2.data_ptr(), packed.data_ptr(), N, K, self.gemm_terms, _lib.current_stream()), 'hmf_conv1x1_split_pack')
        return packed

    def _pack_pair(self, w3, w1):
        from . import _lib
        lib = _lib.load()
        dev = w3.device
        w3, w1 = w3.detach().cpu().contiguous(), w1.detach().cpu().contiguous()
        p3, p1 = torch.empty_like(w3), torch.empty_like(w1)
        _lib.check(lib.hmf_conv_pair_pack_host(w3.data_ptr(), w1.data_ptr(), p3.data_ptr(), p1.data_ptr(), w3.shape[1], w3.shape[0], w1.shape[0]),
                   'hmf_conv_pair_pack_host')
        return p3.to(dev), p1.to(dev)

    def _conv_pair(self, h, x2, residual, pk, bias3, bias1):
        """(y, z): y = relu([h | x2] . W3 + bias3 (+ residual)), z = relu(y . W1 + bias1) in one launch (csrc/conv_pair.hip)."""
        from . import _lib
        lib = _lib.load()
        B, K1, OH, OW = h.shape
        p3, p1, N2 = pk
        C = p1.numel() // N2
        y = torch.empty((B, C, OH, OW), device=h.device, dtype=torch.float32, memory_format=torch.channels_last)
        z = torch.empty((B, N2, OH, OW), device=h.device, dtype=torch.float32, memory_format=torch.channels_last)
        K2, H2, W2 = (x2.shape[1], x2.shape[2], x2.shape[3]) if x2 is not None else (0, 0, 0)
        _lib.check(lib.hmf_conv_pair(h.data_ptr(), None if x2 is None else x2.data_ptr(), p3.data_ptr(), bias3.data_ptr(),
                                     None if residual is None else residual.data_ptr(), y.data_ptr(), p1.data_ptr(), bias1.data_ptr(), z.data_ptr(),
                                     B, OH, OW, K1, H2, W2, K2, 1, C, N2, 0, _lib.current_stream()), 'hmf_conv_pair')
        return y, z

    def _conv_dual(self, h, x, wcat, bias, stride):
        """relu(conv1x1(h) + conv1x1(x, stride) + bias) in one launch (bias = both folded-BN shifts)."""
        from . import _lib
        lib = _lib.load()
        B, Cin, OH, OW = h.shape
        _, Cin2, H2, W2 = x.shape
        Cout = wcat.shape[0]
        y = torch.empty((B, Cout, OH, OW), device=h.device, dtype=torch.float32, memory_format=torch.channels_last)
        sp = self._plan['split'].get(id(wcat)) if self._plan is not None else None
        if sp is not None:
            _lib.check(lib.hmf_conv1x1_gemm_split(h.data_ptr(), x.data_ptr(), sp.data_ptr(), bias.data_ptr(), None, y.data_ptr(), B, OH, OW, Cin,
                                                  H2, W2, Cin2, stride, Cout, 1, self.gemm_terms, 0, _lib.current_stream()), 'hmf_conv1x1_gemm_split')
            return y
        if self.gemm_1x1:
            _lib.check(lib.hmf_conv1x1_gemm(h.data_ptr(), x.data_ptr(), wcat.data_ptr(), bias.data_ptr(), None, y.data_ptr(), B, OH, OW, Cin,
                                            H2, W2, Cin2, stride, Cout, 1, 0, _lib.current_stream()), 'hmf_conv1x1_gemm')
            return y
        _lib.check(lib.hmf_conv1x1_dual_nhwc(h.data_ptr(), x.data_ptr(), wcat.data_ptr(), bias.data_ptr(), y.data_ptr(), B, OH, OW, Cin,
                                             H2, W2, Cin2, stride, Cout, 1, _lib.current_stream()), 'hmf_conv1x1_dual_nhwc')
        return y

    def _pack_winograd(self, w):
        from . import _lib
        lib = _lib.load()
        cout, cin = w.shape[:2]
        wh = w.detach().permute(0, 2, 3, 1).contiguous().cpu()
        packed = torch.empty(int(lib.hmf_conv_winograd_packed_floats(cin, cout)), dtype=torch.float32)
        _lib.check(lib.hmf_conv_winograd_pack_host(wh.data_ptr(), packed.data_ptr(), cin, cout), 'hmf_conv_winograd_pack_host')
        return packed.to(w.device)

    def _conv_hip(self, x, w, bias, stride, residual, relu, pool=False):
        """y = relu?(conv(x, w) + bias + residual) in one launch; x / residual / y are channels_last (B,C,H,W) tensors.
        pool=True (the encoder's last convolution): returns the (B, Cout) global average pool of y (models/resnet.py:214-215) -- from the GEMM's
        own epilogue where built (1x1 layer on an 8x8 map: hmf_conv1x1_gemm_avgpool, the map never reaches HBM), else conv + hmf_avgpool_nhwc."""
        from . import _lib
        lib = _lib.load()
        B, Cin, H, W = x.shape
        Cout, _, ks, _ = w.shape
        pad = ks // 2
        OH, OW = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
        if pool:
            feats = torch.empty((B, Cout), device=x.device, dtype=torch.float32)
            split = self._plan is not None and self._plan['split'].get(id(w)) is not None
            if ks == 1 and stride == 1 and self.gemm_1x1 and not split and OH * OW == 64 and Cin % 32 == 0 and Cout % 64 == 0:
                _lib.check(lib.hmf_conv1x1_gemm_avgpool(x.data_ptr(), w.data_ptr(), None if bias is None else bias.data_ptr(),
                                                        None if residual is None else residual.data_ptr(), None, feats.data_ptr(), B, OH, OW, Cin, Cout,
                                                        int(relu), 0, _lib.current_stream()), 'hmf_conv1x1_gemm_avgpool')
                return feats
            y = self._conv_hip(x, w, bias, stride, residual, relu)
            _lib.check(lib.hmf_avgpool_nhwc(y.data_ptr(), B, OH * OW, Cout, feats.data_ptr(), _lib.current_stream()), 'hmf_avgpool_nhwc')
            return feats
        y = torch.empty((B, Cout, OH, OW), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
        wk = self._plan['wino'].get(id(w)) if self._plan is not None else None
        if wk is not None and stride == 1 and (W >= 16 or W == 8):
            nbytes = int(lib.hmf_conv_winograd_workspace_bytes(B, H, W, Cin, Cout))          # > 0: a small batch, the slab loop is split over work-groups
            ws = self._small_launch_ws(x.device, nbytes) if nbytes else None
            _lib.check(lib.hmf_conv_winograd_ws(x.data_ptr(), wk.data_ptr(), None if bias is None else bias.data_ptr(),
                                                None if residual is None else residual.data_ptr(), y.data_ptr(), B, H, W, Cin, Cout,
                                                int(relu), None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel(), _lib.current_stream()),
                       'hmf_conv_winograd')
            return y
        pk = self._plan['packed'].get(id(w)) if self._plan is not None else None
        if pk is not None:
            cfg = pk[0]                                                     # stride 2: the 8x8 tile (8210 / 8211), chosen at plan time
            nbytes = int(lib.hmf_conv_patch_workspace_bytes(B, H, W, Cin, Cout, ks, stride, cfg))      # > 0: a small batch, the slab loop is split
            ws = self._small_launch_ws(x.device, nbytes) if nbytes else None
            _lib.check(lib.hmf_conv_patch_ws(x.data_ptr(), pk[1].data_ptr(), None if bias is None else bias.data_ptr(),
                                             None if residual is None else residual.data_ptr(), y.data_ptr(), B, H, W, Cin, Cout, ks, stride,
                                             int(relu), cfg, None if ws is None else ws.data_ptr(), 0 if ws is None else ws.numel(), _lib.current_stream()),
                       'hmf_conv_patch')
            return y
        sp = self._plan['split'].get(id(w)) if (self._plan is not None and ks == 1 and stride == 1) else None
        if sp is not None:
            _lib.check(lib.hmf_conv1x1_gemm_split(x.data_ptr(), None, sp.data_ptr(), None if bias is None else bias.data_ptr(),
                                                  None if residual is None else residual.data_ptr(), y.data_ptr(), B, H, W, Cin, 0, 0, 0, 1, Cout,
                                                  int(relu), self.gemm_terms, 0, _lib.current_stream()), 'hmf_conv1x1_gemm_split')
            return y
        if ks == 1 and stride == 1 and self.gemm_1x1 and Cin % 32 == 0 and Cout % 64 == 0:
            _lib.check(lib.hmf_conv1x1_gemm(x.data_ptr(), None, w.data_ptr(), None if bias is None else bias.data_ptr(),
                                            None if residual is None else residual.data_ptr(), y.data_ptr(), B, H, W, Cin, 0, 0, 0, 1, Cout,
                                            int(relu), 0, _lib.current_stream()), 'hmf_conv1x1_gemm')
            return y
        _lib.check(lib.hmf_conv2d_nhwc(x.data_ptr(), w.data_ptr(), None if bias is None else bias.data_ptr(),
                                       None if residual is None else residual.data_ptr(), y.data_ptr(), B, H, W, Cin, Cout, ks, stride,
                                       int(relu), 0, _lib.current_stream()), 'hmf_conv2d_nhwc')
        return y

    def _small_launch_ws(self, device, nbytes):
        """Partial-sum workspace of the split small-batch convolutions, one per stream (launches of a stream run in order and consume it at once)."""
        key = ('splitk', device.index, torch.cuda.current_stream(device).cuda_stream)
        ws = self._stem_ws.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = self._stem_ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return ws

    def _stem_hip(self, x, w_packed, bias):
        """NCHW proxy input -> conv7x7/2 + bias + ReLU + max-pool 3x3/2 (NHWC): one launch (the pool is applied to the tiles before they leave
        the CU; `fuse_stem_pool = False` keeps the two-launch form), no layout-conversion pass."""
        from . import _lib
        lib = _lib.load()
        B, C, H, W = x.shape
        OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        PH, PW = (OH - 1) // 2 + 1, (OW - 1) // 2 + 1
        st = _lib.current_stream()
        z = torch.empty((B, 64, PH, PW), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
        conv = lib.hmf_stem_wino if self.stem_polyphase else lib.hmf_stem_conv
        if self.fuse_stem_pool and self.stem_polyphase:
            nbytes = int(lib.hmf_stem_wino_pool_workspace_bytes(B, H, W))
            key = (x.device.index, torch.cuda.current_stream(x.device).cuda_stream)
            ws = self._stem_ws.get(key)
            if ws is None or ws.numel() < nbytes:
                ws = self._stem_ws[key] = torch.empty(nbytes, dtype=torch.uint8, device=x.device)     # per stream: overlapped batches must not share it
            _lib.check(lib.hmf_stem_wino_pool(x.data_ptr(), w_packed.data_ptr(), bias.data_ptr(), z.data_ptr(), B, C, H, W, ws.data_ptr(), ws.numel(), st),
                       'hmf_stem_wino_pool')
            return z
        if self.fuse_stem_pool:
            _lib.check(lib.hmf_stem_conv_pool(x.data_ptr(), w_packed.data_ptr(), bias.data_ptr(), z.data_ptr(), B, C, H, W, st), 'hmf_stem_conv_pool')
            return z
        y = torch.empty((B, 64, OH, OW), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
        _lib.check(conv(x.data_ptr(), w_packed.data_ptr(), bias.data_ptr(), y.data_ptr(), B, C, H, W, 1, st), 'hmf_stem_conv')
        _lib.check(lib.hmf_maxpool3x3s2_nhwc(y.data_ptr(), z.data_ptr(), B, OH, OW, 64, st), 'hmf_maxpool3x3s2_nhwc')
        return z

    @torch.no_grad()
    def forward_folded(self, x):
        """Inference plan on the HIP kernels: (B, 18, H, W) NCHW proxy input on a HIP device -> (B, out_dim)."""
        from . import _lib
        _lib.require_cuda(x, 'input')
        key = self._params_key() + (self.gemm_terms,)
        if self._plan is None or self._plan_key != key:
            self._plan, self._plan_key = self._build_plan(), key
        p = self._plan
        with torch.cuda.device(x.device):
            x = self._stem_hip(x.float().contiguous(), p['stem_packed_polyphase' if self.stem_polyphase else 'stem_packed'], p['stem'][1])
            x, _ = self._run_blocks(p, x, None, 0, len(p['blocks']))
            if x.dim() == 4:                                 # a plan whose last block ended in a fused launch: pool the map
                B, C, H, W = x.shape
                feats = torch.empty((B, C), device=x.device, dtype=torch.float32)
                _lib.check(_lib.load().hmf_avgpool_nhwc(x.data_ptr(), B, H * W, C, feats.data_ptr(), _lib.current_stream()), 'hmf_avgpool_nhwc')
                x = feats
            return x

    def _run_blocks(self, p, x, pre, i0, i1):
        """Residual blocks i0 .. i1-1 of the plan on x; `pre` = the first block's first 1x1 stage when a seam launch already produced it.
        Returns (x, pre) in the same sense for block i1."""
        nblk = len(p['blocks'])
        for i in range(i0, i1):
            stages, ds, dual = p['blocks'][i]
            h = x if pre is None else pre
            for w, b, s, pad in stages[(0 if pre is None else 1):-1]:
                h = self._conv_hip(h, w, b, s[0], None, True)
            pre = None
            pk = p['pair'].get(i)
            if pk is not None and (h.shape[0] * h.shape[2] * h.shape[3]) % 64 == 0:
                nb1 = p['blocks'][i + 1][0][0][1]
                x, pre = self._conv_pair(h, x if dual is not None else None, None if dual is not None else x, pk, stages[-1][1], nb1)
            elif dual is not None:
                x = self._conv_dual(h, x, dual[0], dual[1], dual[2])
            else:
                idt = x if ds is None else self._conv_hip(x, ds[0], None, ds[2][0], None, False)
                w, b, s, pad = stages[-1]
                x = self._conv_hip(h, w, b, s[0], idt, True, pool=(i == nblk - 1))       # the last convolution also pools (:214-215)
        return x, pre

    def _stem_packed_from_device(self, w):
        """The stem kernel's packed filter operand from the live (64, 18, 7, 7) parameter without leaving the device: the host packer is a
        pure permutation + zero padding, so packing an index-valued array ONCE yields a gather map (training re-packs every step)."""
        from . import _lib
        m = self.__dict__.get('_stem_map')
        if m is None or m[0].device != w.device:
            lib = _lib.load()
            n = w.numel()
            idx = torch.arange(1, n + 1, dtype=torch.float32).reshape(w.shape).contiguous()
            packed = torch.empty(int(lib.hmf_stem_packed_floats()), dtype=torch.float32)
            _lib.check(lib.hmf_stem_pack_host(idx.data_ptr(), packed.data_ptr()), 'hmf_stem_pack_host')
            pm = packed.long()
            m = self.__dict__['_stem_map'] = ((pm.clamp(min=1) - 1).to(w.device), (pm > 0).float().to(w.device))
        return w.reshape(-1).float()[m[0]] * m[1]

    def forward(self, x):
        from . import _lib
        _lib.require_cuda(x, 'input')
        wants_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        if not self.training and not wants_grad:
            return self.forward_folded(x)
        if self.training and wants_grad and not x.requires_grad and self.conv1.in_channels == 18:
            # training step: batch-statistics BatchNorm forward + full backward on the library's kernels (encoder_train.py)
            from .encoder_train import EncoderTrainFn
            return EncoderTrainFn.apply(self, x, *list(self.parameters()))
        # remaining combinations (train() without gradients, eval() with gradients = frozen statistics under autograd, gradients w.r.t. the
        # input) have no hand-written path and there is no vendor-library path in the product: fail loudly.  (`forward_module` is the plain
        # nn.Module graph; tests call it explicitly as the float64 reference.)
        raise NotImplementedError(
            'humaniflow_amd: ResNetEncoder.forward is built for inference (eval() under torch.no_grad()) and for the training step (train() with '
            'gradients w.r.t. the parameters, 18-channel input that does not require grad); got training=%s, grad=%s, input.requires_grad=%s, '
            'in_channels=%d.  Wrap inference in torch.no_grad(); for eval() under autograd (frozen BatchNorm fine-tuning, gradients w.r.t. the '
            'input) call `forward_module` on the encoder explicitly (the plain nn.Module graph on ATen: not the library kernels, never chosen silently)'
            % (self.training, wants_grad, x.requires_grad, self.conv1.in_channels))


def resnet18(in_channels, pretrained=False):
    return ResNetEncoder(18, in_channels)


def resnet50(in_channels, pretrained=False):
    return ResNetEncoder(50, in_channels)
