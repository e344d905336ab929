"""RoI feature head on the GPU: res4 feature maps + tracklet boxes -> per-(tracklet, frame) RoI features,
the `tracklet_feats [N,T,D]` the pair builder consumes (first slice).

The reference obtains these features from detectron2's R101-C4 model (lib/detectron/trainer.py:23-33
configures it; 35 classes) — it owns no code for the step.  This module mirrors that model's ROI head
(detectron2 v0.6 `Res5ROIHeads`: ROIAlign 14x14 aligned / adaptive sampling on the stride-16 map -> res5 =
3 BottleneckBlocks with stride_in_1x1 and FrozenBatchNorm -> mean over 7x7) with the SAME parameter names
(`res5.{b}.{conv1,conv2,conv3,shortcut}.weight`, `...norm.{weight,bias,running_mean,running_var}`), so the
`roi_heads.res5.*` entries of a detectron2 checkpoint load with `load_state_dict` after stripping the
`roi_heads.` prefix.  Every tensor operation runs in the hand-written HIP library (ops.roi_align_nhwc,
ops.conv2d_nhwc: fp32 MFMA implicit GEMM, channels-last); there is no CPU path.

`ResNetC4` (below) is the C4 backbone that produces the res4 maps from frames; both modules run fp32 or bf16
(bf16 maps select the bf16 MFMA kernels: dedicated stem kernel, one fused launch per bottleneck tail) and alternate
their chunks (frames / RoIs) between two HIP streams.
"""
import torch
import torch.nn as nn

from . import ops
from .model import _CachedWeightsMixin, _DeviceCache, _compute_device, _f32

BN_EPS = 1e-5   # detectron2 FrozenBatchNorm2d default


class FrozenBatchNorm2d(nn.Module):
    """Parameter holder of detectron2's FrozenBatchNorm2d (layers/batch_norm.py): four buffers."""

    def __init__(self, num_features):
        super().__init__()
        self.register_buffer("weight", torch.ones(num_features))
        self.register_buffer("bias", torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))


class ConvFrozenBN(nn.Module):
    """detectron2 `Conv2d(bias=False, norm=FrozenBN)`: `weight` [Cout,Cin,k,k] + `norm.*`.  The batch norm is
    folded into the packed weight and a bias once per parameter version:
        w' = w * scale[co],  b' = bias - running_mean * scale,  scale = weight * rsqrt(running_var + eps)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, pad_cin_to=0):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels, kernel_size, kernel_size))
        nn.init.kaiming_normal_(self.weight, mode="fan_out", nonlinearity="relu")
        self.norm = FrozenBatchNorm2d(out_channels)
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.pad_cin_to = pad_cin_to      # the kernels take Cin % 16 == 0: the RGB stem runs on zero-padded channels
        self._cache = _DeviceCache()

    def folded(self, dev):
        def build(ts):
            w, g, b, m, v = ts
            scale = g * torch.rsqrt(v + BN_EPS)
            wf = (w * scale.reshape(-1, 1, 1, 1)).contiguous()
            if self.pad_cin_to > wf.shape[1]:
                wf = torch.nn.functional.pad(wf, (0, 0, 0, 0, 0, self.pad_cin_to - wf.shape[1])).contiguous()
            # fragment-major weights select the registers-direct kernel (Cout % 32 == 0, Cin % 16 == 0)
            pack = ops.pack_conv2d_frag if (wf.shape[0] % 32 == 0 and wf.shape[1] % 16 == 0) else ops.pack_conv2d
            return pack(wf), (b - m * scale).contiguous()
        n = self.norm
        return self._cache.get("folded", (self.weight, n.weight, n.bias, n.running_mean, n.running_var), dev, build)

    def folded_bf16(self, dev):
        """bf16 operand set: the folded fp32 weight rounded once to bf16 (fragment-major), fp32 bias."""
        def build(ts):
            w, g, b, m, v = ts
            scale = g * torch.rsqrt(v + BN_EPS)
            return (ops.pack_conv2d_frag_bf16((w * scale.reshape(-1, 1, 1, 1)).contiguous()),
                    (b - m * scale).contiguous())
        n = self.norm
        return self._cache.get("folded_bf16", (self.weight, n.weight, n.bias, n.running_mean, n.running_var),
                               dev, build)

    def forward(self, x, residual=None, relu=False, stride=None):
        """x channels-last [NB,H,W,Cin] on the HIP device -> act(bn(conv(x)) + residual); bf16 x selects
        the bf16-operand kernel (bf16 out).  `stride` overrides the module's (a caller that hands over an
        already subsampled map runs a strided 1x1 conv with stride 1)."""
        k = (self.kernel_size, self.kernel_size)
        stride = self.stride if stride is None else stride
        if x.dtype == torch.bfloat16:
            frag, bias = self.folded_bf16(x.device)
            return ops.conv2d_nhwc_bf16(x, frag, k, stride, self.padding, bias=bias, residual=residual, relu=relu)
        packed, bias = self.folded(x.device)
        return ops.conv2d_nhwc(x, packed, k, stride, self.padding, bias=bias, residual=residual, relu=relu)


class BottleneckBlock(nn.Module):
    """detectron2 BottleneckBlock (modeling/backbone/resnet.py), stride_in_1x1=True: 1x1 (stride) -> 3x3 -> 1x1,
    projection shortcut when the channel count changes, ReLU after the residual add (fused into conv3)."""

    def __init__(self, in_channels, out_channels, bottleneck_channels, stride=1):
        super().__init__()
        self.shortcut = ConvFrozenBN(in_channels, out_channels, 1, stride) if in_channels != out_channels else None
        self.conv1 = ConvFrozenBN(in_channels, bottleneck_channels, 1, stride)
        self.conv2 = ConvFrozenBN(bottleneck_channels, bottleneck_channels, 3, 1, 1)
        self.conv3 = ConvFrozenBN(bottleneck_channels, out_channels, 1)
        self.stride = stride
        # bf16 maps: run conv2 + conv3 + residual + ReLU as ONE launch (csrc/tspn_bottleneck_bf16.hip; h2 stays in
        # LDS, results bit-identical to the two separate launches).  Needs 64 / 128 / 256 bottleneck channels and
        # the 4x expansion of the standard block; switched off by ResNetC4 / Res5RoIHead `fuse_bottlenecks = False`.
        self.fuse_tail = True
        # ... and let that launch compute the NEXT block's conv1 too where it can (CM = 256: the res4 chain), so that
        # the 1024-channel map is read once per block; switched by ResNetC4 `fuse_next_conv1`
        self.fuse_next = True
        # ... and whole identity blocks of the memory-bound stages (64 / 128 bottleneck channels, stride 1, no projection)
        # as ONE launch that reads the 4 CM-channel map once (csrc/tspn_block_bf16.hip, round 5); same bits
        self.fuse_block = True
        # ... and the FIRST block of res2 (projection shortcut, stride 1) as one launch that computes the shortcut
        # on the tile's own input pixels instead of writing the 4 CM-channel shortcut map and reading it back
        self.fuse_block_proj = True
        # 256 bottleneck channels (res4): the fused tail with the work split by role -- four MFMA waves + four io waves per
        # workgroup (tspn_bottleneck_tail_io_bf16, round 5; bit-identical, 16 - 20 % faster per launch at the res4 shape); switched by ResNetC4 `tail_io_waves`
        self.tail_io_waves = True

    PROJ_SHAPES = ((64, 64, 1),)        # (input channels, bottleneck channels, stride) tspn_bottleneck_block_proj_bf16 is enabled for

    def _can_fuse_block(self, x, h1):
        c1, c2, c3 = self.conv1, self.conv2, self.conv3
        cm = c2.weight.shape[0]
        return (self.fuse_tail and self.fuse_block and h1 is None and x.dtype == torch.bfloat16 and cm in (64, 128)
                and self.shortcut is None and self.stride == 1 and c1.kernel_size == 1 and c1.padding == 0
                and tuple(c1.weight.shape[:2]) == (cm, 4 * cm) and c2.weight.shape[1] == cm and c2.kernel_size == 3
                and c2.stride == 1 and c2.padding == 1 and tuple(c3.weight.shape[:2]) == (4 * cm, cm) and c3.kernel_size == 1)

    def _can_fuse_block_proj(self, x, h1, presampled):
        """First block of a stage (projection shortcut): the shapes tspn_bottleneck_block_proj_bf16 is built for."""
        c1, c2, c3, sc = self.conv1, self.conv2, self.conv3, self.shortcut
        if sc is None or not (self.fuse_tail and self.fuse_block and self.fuse_block_proj) or h1 is not None or presampled:
            return False
        cm, cin = c2.weight.shape[0], c1.weight.shape[1]
        return (x.dtype == torch.bfloat16 and (cin, cm, self.stride) in self.PROJ_SHAPES and c1.kernel_size == 1
                and c1.padding == 0 and c1.stride == self.stride and sc.kernel_size == 1 and sc.padding == 0
                and sc.stride == self.stride and c1.weight.shape[0] == cm and tuple(sc.weight.shape[:2]) == (4 * cm, cin)
                and c2.weight.shape[1] == cm and c2.kernel_size == 3 and c2.stride == 1 and c2.padding == 1
                and tuple(c3.weight.shape[:2]) == (4 * cm, cm) and c3.kernel_size == 1)

    RES_SHAPES = ((256, 128, 2),)       # ... tspn_bottleneck_block_res_bf16 (shortcut launched separately) is enabled for

    def _can_fuse_block_res(self, x, h1, presampled):
        c1, c2, c3, sc = self.conv1, self.conv2, self.conv3, self.shortcut
        if sc is None or not (self.fuse_tail and self.fuse_block and self.fuse_block_proj) or h1 is not None or presampled:
            return False
        cm, cin = c2.weight.shape[0], c1.weight.shape[1]
        return (x.dtype == torch.bfloat16 and (cin, cm, self.stride) in self.RES_SHAPES and c1.kernel_size == 1
                and c1.padding == 0 and c1.stride == self.stride and sc.kernel_size == 1 and sc.padding == 0
                and sc.stride == self.stride and c1.weight.shape[0] == cm and tuple(sc.weight.shape[:2]) == (4 * cm, cin)
                and c2.weight.shape[1] == cm and c2.kernel_size == 3 and c2.stride == 1 and c2.padding == 1
                and tuple(c3.weight.shape[:2]) == (4 * cm, cm) and c3.kernel_size == 1)

    def _can_fuse(self, x):
        c2, c3 = self.conv2, self.conv3
        cm = c2.weight.shape[0]
        return (self.fuse_tail and x.dtype == torch.bfloat16 and cm in (64, 128, 256) and c2.weight.shape[1] == cm
                and c3.weight.shape[0] == 4 * cm and c2.kernel_size == 3 and c2.stride == 1 and c2.padding == 1)

    def _can_take_h1(self, x):
        """This block's conv1 can be computed by the PREVIOUS block's tail launch (ops.bottleneck_tail_bf16 with
        next_frag1): identity shortcut, stride 1, 256 bottleneck channels, 1x1 conv1 on 4 x 256 channels."""
        c1 = self.conv1
        return (self.fuse_next and self.shortcut is None and self.stride == 1 and c1.kernel_size == 1 and c1.padding == 0
                and c1.weight.shape[0] == 256 and c1.weight.shape[1] == 1024 and x.dtype == torch.bfloat16)

    def forward(self, x, presampled=False, out=None, h1=None, next_block=None):
        """`presampled`: x already holds only the pixels the strided 1x1 convs (conv1, shortcut) read -- every
        `stride`-th row and column -- so they run with stride 1 (Res5RoIHead lets ROIAlign produce just those bins).
        `out`: where to write the block's result (a contiguous tensor of its shape; the fused bf16 tail writes into it
        directly, the other paths copy).
        `h1`: this block's conv1 output, already computed by the previous block's tail launch.
        `next_block`: the block that follows; when it qualifies (`_can_take_h1`) this block's fused tail computes its
        conv1 as well and the call returns (y, h1 of the next block) instead of y."""
        st = 1 if presampled else None
        if self._can_fuse_block(x, h1):
            f1, b1 = self.conv1.folded_bf16(x.device)
            f2, b2 = self.conv2.folded_bf16(x.device)
            f3, b3 = self.conv3.folded_bf16(x.device)
            y = ops.bottleneck_block_bf16(x.contiguous(), f1, b1, f2, b2, f3, b3, out=out)
            return (y, None) if next_block is not None else y
        if self._can_fuse_block_proj(x, h1, presampled):
            f1, b1 = self.conv1.folded_bf16(x.device)
            f2, b2 = self.conv2.folded_bf16(x.device)
            f3, b3 = self.conv3.folded_bf16(x.device)
            fs, bs = self.shortcut.folded_bf16(x.device)
            y = ops.bottleneck_block_proj_bf16(x.contiguous(), self.stride, f1, b1, f2, b2, f3, b3, fs, bs, out=out)
            return (y, None) if next_block is not None else y
        if self._can_fuse_block_res(x, h1, presampled):
            # res3.0: the projection shortcut as its own launch, conv1 + 3x3 + expand + residual as one
            f1, b1 = self.conv1.folded_bf16(x.device)
            f2, b2 = self.conv2.folded_bf16(x.device)
            f3, b3 = self.conv3.folded_bf16(x.device)
            y = ops.bottleneck_block_res_bf16(x.contiguous(), self.stride, f1, b1, f2, b2, f3, b3, self.shortcut(x).contiguous(), out=out)
            return (y, None) if next_block is not None else y
        h = h1 if h1 is not None else self.conv1(x, relu=True, stride=st)
        if self.shortcut is not None:
            sc = self.shortcut(x, stride=st)
        elif self.stride == 1:
            sc = x
        else:
            raise ValueError("identity shortcut needs stride 1")
        hand_over = next_block is not None and next_block._can_take_h1(x) and self.conv2.weight.shape[0] == 256
        if self._can_fuse(x):
            f2, b2 = self.conv2.folded_bf16(x.device)
            f3, b3 = self.conv3.folded_bf16(x.device)
            if hand_over:
                f1n, b1n = next_block.conv1.folded_bf16(x.device)
                return ops.bottleneck_tail_bf16(h, f2, b2, f3, b3, sc.contiguous(), out=out, next_frag1=f1n, next_bias1=b1n)
            io = bool(self.tail_io_waves) and self.conv2.weight.shape[0] == 256
            y = ops.bottleneck_tail_bf16(h, f2, b2, f3, b3, sc.contiguous(), out=out, io_waves=io)
            return (y, None) if next_block is not None else y
        y = self.conv3(self.conv2(h, relu=True), residual=sc, relu=True)
        if out is not None:
            out.copy_(y)
            y = out
        return (y, None) if next_block is not None else y


def _warm_conv_weights(module, dev, bf16):
    """Fold + pack the weights of every ConvFrozenBN under `module` on the CURRENT (the caller's) stream.  The
    multi-stream forwards call this before their side streams are made to wait for the caller's: `_DeviceCache` has no
    stream ordering of its own, so a cache filled lazily by the first chunk on side stream 0 could be read half-built
    by side stream 1.  A cache hit costs a few microseconds per conv, so this runs on every forward and
    also catches weights that changed since the last one."""
    for m in module.modules():
        if isinstance(m, BasicStem):
            c = m.conv1
            if bf16:
                m._folded_bf16(dev)
            elif m.in_channels <= 4 and c.weight.shape[0] % 32 == 0:
                m._folded_cin4(dev)
            else:
                c.folded(dev)
        elif isinstance(m, BottleneckBlock):
            for c in (m.conv1, m.conv2, m.conv3, m.shortcut):
                if c is not None:
                    c.folded_bf16(dev) if bf16 else c.folded(dev)


class Res5RoIHead(_CachedWeightsMixin, nn.Module):
    """ROIAlign + res5 + spatial mean (detectron2 Res5ROIHeads._shared_roi_transform, then .mean([2,3])).

    forward(feature_maps, tracklet_boxes):
        feature_maps  [T,Hf,Wf,C] channels-last res4 maps, one per frame (use `from_nchw` for NCHW maps)
        tracklet_boxes [N,T,4] (left, top, right, bottom) in image pixels
        -> tracklet_feats [N,T,out_channels] on the device of `tracklet_boxes`
    bf16 feature maps select the bf16-operand kernels (bf16 MFMA, fp32 accumulation; every layer's output
    rounded to bf16 once) and return bf16 features — what the bf16 scorer path consumes (BASELINE cfg3 / cfg5).
    Needs channel counts that are multiples of 64.
    """

    def __init__(self, in_channels=1024, bottleneck_channels=512, out_channels=2048, num_blocks=3,
                 pooler_resolution=14, spatial_scale=1.0 / 16, sampling_ratio=0, first_stride=2, roi_chunk=2400):
        super().__init__()
        blocks, cin = [], in_channels
        for b in range(num_blocks):
            blocks.append(BottleneckBlock(cin, out_channels, bottleneck_channels, first_stride if b == 0 else 1))
            cin = out_channels
        self.res5 = nn.Sequential(*blocks)
        self.fuse_bottlenecks = True
        self.subsample_roi_align = True     # ROIAlign only the bins res5's strided first block reads (same results)
        self.streams = 1                    # > 1: RoI chunks alternate between HIP streams (measured at cfg5: 105 -> 108 ms, so off)
        self._side_streams = {}
        self.in_channels, self.out_channels = in_channels, out_channels
        self.pooler_resolution, self.spatial_scale, self.sampling_ratio = pooler_resolution, spatial_scale, sampling_ratio
        self.roi_chunk = int(roi_chunk)

    @staticmethod
    def from_nchw(feature_maps):
        """[T,C,Hf,Wf] (what a detectron2 backbone returns) -> channels-last [T,Hf,Wf,C]."""
        return feature_maps.permute(0, 2, 3, 1).contiguous()

    def forward(self, feature_maps, tracklet_boxes):
        with torch.no_grad():
            for blk in self.res5:
                blk.fuse_tail = bool(self.fuse_bottlenecks)
            dev = _compute_device(feature_maps, tracklet_boxes, self.res5[0].conv1.weight)
            bf16 = isinstance(feature_maps, torch.Tensor) and feature_maps.dtype == torch.bfloat16
            # RoIAlign interpolates in fp32 and reads a bf16 map as it is (bf16 values are exact in fp32)
            fm = feature_maps.to(dev).contiguous() if bf16 else _f32(feature_maps, dev)
            if fm.dim() != 4 or fm.shape[3] != self.in_channels:
                raise ValueError(f"feature_maps must be channels-last [T,Hf,Wf,{self.in_channels}], got {tuple(fm.shape)}")
            boxes = _f32(tracklet_boxes, dev)
            if boxes.dim() != 3 or boxes.shape[2] != 4 or boxes.shape[1] != fm.shape[0]:
                raise ValueError(f"tracklet_boxes must be [N,T={fm.shape[0]},4], got {tuple(boxes.shape)}")
            n, t, _ = boxes.shape
            idx = torch.arange(t, dtype=torch.float32, device=dev).repeat(n)
            rois = torch.cat([idx[:, None], boxes.reshape(n * t, 4)], dim=1).contiguous()
            feats = torch.empty((n * t, self.out_channels), dtype=torch.float32, device=dev)
            # RoI chunks alternate between `streams` HIP streams (as the backbone's frame chunks do): the memory-heavy
            # parts of one chunk run under the MFMA-heavy parts of the other
            nchunks = -(-(n * t) // self.roi_chunk)
            ns = min(int(self.streams), nchunks) if fm.is_cuda else 1
            main = torch.cuda.current_stream(dev)
            side = None
            if ns > 1:
                side = self._side_streams.setdefault((dev.index, ns), [torch.cuda.Stream(device=dev) for _ in range(ns)])
                _warm_conv_weights(self, dev, bf16)      # packed weights exist before any side stream may read them
                for st in side:
                    st.wait_stream(main)
            # res5's first block reads its input only through 1x1 convs of stride s (stride_in_1x1: conv1 and the
            # projection shortcut): ROIAlign produces just the bins (s i, s j) -- a quarter of the work and of the
            # 14 x 14 x 1024 output for s = 2 -- and the two convs run with stride 1; same values, bit for bit
            b0 = self.res5[0]
            bs = b0.stride if (self.subsample_roi_align and b0.shortcut is not None and b0.conv1.kernel_size == 1
                               and b0.shortcut.kernel_size == 1 and b0.conv1.padding == 0) else 1

            def one_chunk(lo):
                x = ops.roi_align_nhwc(fm, rois[lo:lo + self.roi_chunk].contiguous(), self.pooler_resolution,
                                       self.spatial_scale, self.sampling_ratio, aligned=True, out_bf16=bf16, bin_stride=bs)
                x = b0(x, presampled=bs > 1)
                for blk in list(self.res5)[1:]:
                    x = blk(x)
                r, h, w, c = x.shape
                if bf16:
                    feats[lo:lo + r] = ops.temporal_mean_bf16(x.view(r, h * w, c))   # bf16-rounded means
                else:
                    feats[lo:lo + r] = ops.temporal_mean(x.view(r, h * w, c), layout_tc=True)

            for k, lo in enumerate(range(0, n * t, self.roi_chunk)):
                if side is None:
                    one_chunk(lo)
                else:
                    with torch.cuda.stream(side[k % ns]):
                        one_chunk(lo)
            if side is not None:
                for st in side:
                    main.wait_stream(st)
            src = tracklet_boxes.device if isinstance(tracklet_boxes, torch.Tensor) else torch.device("cpu")
            feats = feats.view(n, t, self.out_channels)
            return (ops.cast_bf16(feats) if bf16 else feats).to(src)


class BasicStem(nn.Module):
    """detectron2 BasicStem (modeling/backbone/resnet.py): 7x7 stride-2 conv + FrozenBN + ReLU, then
    max_pool2d(3, 2, 1).  fp32 backbone: the RGB input runs on zero-padded channels through the fp32 kernels.
    bf16 backbone (`out_bf16`): bf16-operand stem kernel + bf16 pool (csrc/tspn_stem_bf16.hip)."""

    def __init__(self, in_channels=3, out_channels=64):
        super().__init__()
        self.conv1 = ConvFrozenBN(in_channels, out_channels, 7, 2, 3, pad_cin_to=16)
        self.in_channels = in_channels
        self.fuse_pool = True               # bf16 path: conv + max pool in one kernel (ops.stem_pool_bf16)

    def _folded_cin4(self, dev):
        c = self.conv1

        def build(ts):
            w, g, b, m, v = ts
            scale = g * torch.rsqrt(v + BN_EPS)
            return ops.pack_conv2d_frag_cin4((w * scale.reshape(-1, 1, 1, 1)).contiguous()), (b - m * scale).contiguous()
        n = c.norm
        return c._cache.get("folded_cin4", (c.weight, n.weight, n.bias, n.running_mean, n.running_var), dev, build)

    def _folded_bf16(self, dev):
        c = self.conv1

        def build(ts):
            w, g, b, m, v = ts
            scale = g * torch.rsqrt(v + BN_EPS)
            return ops.pack_stem_bf16((w * scale.reshape(-1, 1, 1, 1)).contiguous()), (b - m * scale).contiguous()
        n = c.norm
        return c._cache.get("folded_stem_bf16", (c.weight, n.weight, n.bias, n.running_mean, n.running_var), dev, build)

    def forward(self, x, out_bf16=False):
        c = self.conv1
        if out_bf16:
            # bf16 backbone: the stem is a bf16-operand conv like every other layer (image and folded weights rounded
            # to bf16, fp32 accumulation, one rounding of relu(acc + bias)); 7x7/2 as a 4x4/1 conv on the 2x2
            # space-to-depth image (csrc/tspn_stem_bf16.hip) with the 3x3/2 max pool inside the conv kernel
            # (`fuse_pool`; off: the conv map is written and pooled by a second kernel - same bits)
            if self.in_channels != 3 or c.kernel_size != 7 or c.stride != 2 or c.padding != 3 or c.weight.shape[0] not in (32, 64):
                raise ValueError("the bf16 backbone needs detectron2's BasicStem: 7x7 / stride 2 / padding 3 on 3 channels, "
                                 f"32 or 64 output channels (got {tuple(c.weight.shape)}, stride {c.stride}, padding {c.padding})")
            frag, bias = self._folded_bf16(x.device)
            if self.fuse_pool:
                return ops.stem_pool_bf16(x.contiguous(), frag, bias)
            return ops.max_pool_nhwc_bf16(ops.stem_conv_bf16(x.contiguous(), frag, bias), 3, 2, 1)
        if self.in_channels <= 4 and c.weight.shape[0] % 32 == 0:
            # stem form: RGB + zero channel, one K chunk = four taps (13 chunks for 7x7 instead of 49)
            frag, bias = self._folded_cin4(x.device)
            x = torch.nn.functional.pad(x, (0, 4 - self.in_channels)).contiguous()    # [NB,H,W,4]
            y = ops.conv2d_nhwc_cin4(x, frag, (c.kernel_size, c.kernel_size), c.stride, c.padding, bias=bias, relu=True)
        else:
            x = torch.nn.functional.pad(x, (0, 16 - self.in_channels)).contiguous()   # [NB,H,W,16]
            y = c(x, relu=True)
        return ops.max_pool_nhwc(y, 3, 2, 1, out_bf16=out_bf16)


def _run_blocks(blocks, x, last_out=None):
    """A stage's bottleneck blocks in a row.  Each block is told which block follows: where both qualify (bf16 map, 256
    bottleneck channels: the res4 chain) its fused tail launch also computes the follower's conv1 on the tile it has
    just produced, and the follower starts from that h1.  `last_out`: where the last block writes its result."""
    blocks = list(blocks)
    h1 = None
    for i, blk in enumerate(blocks):
        if i + 1 < len(blocks):
            x, h1 = blk(x, h1=h1, next_block=blocks[i + 1])
        else:
            x = blk(x, h1=h1, out=last_out)
    return x


class ResNetC4(_CachedWeightsMixin, nn.Module):
    """The C4 backbone of detectron2's R-50/101-C4 models (build_resnet_backbone with OUT_FEATURES = res4):
    stem -> res2 -> res3 -> res4, bottleneck blocks with stride_in_1x1 and FrozenBN, parameter names as in a
    detectron2 checkpoint after stripping `backbone.` (`stem.conv1.weight`, `res4.22.conv3.norm.running_var`, ...).

    forward(images): channels-last frames [T,H,W,3] float (already mean-subtracted, BGR for the MSRA weights the
    reference's config uses) -> res4 maps [T,H/16,W/16,1024] channels-last on the HIP device, fp32 or — with
    `bf16=True` — bf16 (every convolution, the stem included, on bf16 operands with fp32 accumulation and one
    rounding per layer): the input of Res5RoIHead."""

    BLOCKS = {50: (3, 4, 6), 101: (3, 4, 23)}

    def __init__(self, depth=101, stem_out=64, res2_out=256, blocks=None, frame_chunk=90):
        super().__init__()
        blocks = tuple(blocks) if blocks is not None else self.BLOCKS[depth]
        self.stem = BasicStem(3, stem_out)
        cin, cout = stem_out, res2_out
        for i, nb in enumerate(blocks):
            stage = []
            for b in range(nb):
                stage.append(BottleneckBlock(cin, cout, cout // 4, 2 if (b == 0 and i > 0) else 1))
                cin = cout
            setattr(self, f"res{i + 2}", nn.Sequential(*stage))
            cout *= 2
        self.out_channels = cin
        # frames per launch: 90 (round 6; 36 in round 5, 18 before, 9 in round 4).  Measured on whole cfg5 videos, backbone ms
        # per 900 frames in one box (profiles/r6/frame_chunk_sweep.txt): 9: 304.7, 18: 302.1, 36: 298.5, 45: 295.0 | another
        # box: 45: 288 - 290, 63: 286.5, 90: 285.8, 180: 284.5.  Ten launches of 90 frames: 9.9 rounds of res4-tail tiles per
        # launch (2 531 tiles on 256 CUs, one workgroup each) instead of 3.96, 2.5 x fewer launches; the maps of a chunk
        # (2.7 GB at res2) are nothing beside 288 GB of HBM.  Fitting a chunk's res4 map into the 256-MB Infinity Cache
        # (27 frames) buys nothing: 299.3 against 298.5 ms.
        self.frame_chunk = int(frame_chunk)
        self.fuse_bottlenecks = True     # bf16 maps: conv2 + conv3 + residual of every block in one launch
        # ... which at res4 can also compute conv1 of the block that follows (round 4, tspn_bottleneck_tail_next_bf16): the
        # 1024-channel map is then read once per block (-0.16 GB of fabric traffic per 720p frame) and the pair of launches
        # takes 5 % less time on its own (109 against 115 us per 8 frames) -- but that launch needs 136 KB of LDS, one
        # workgroup per CU, so the chunks of the other HIP stream no longer share CUs with it: 27.5 against 26.0 ms per 64
        # frames with two streams (equal, 31.9 / 32.0, with one).  Off by default for that reason.
        self.fuse_next_conv1 = False
        # identity blocks of res2 / res3 (64 / 128 bottleneck channels) as ONE launch each (round 5, tspn_bottleneck_block_bf16:
        # the 4 CM-channel map read once, h1 / h2 on the CU; same bits as conv1 + fused tail)
        self.fuse_blocks = True
        # ... and the first block of res2 (projection shortcut) as one launch too (tspn_bottleneck_block_proj_bf16)
        self.fuse_first_blocks = True
        # res4's fused tails with the work split by role (round 5, tspn_bottleneck_tail_io_bf16: four MFMA waves whose only
        # memory traffic is the weight stream + four io waves that own the h1 DMA, the residual rows and the stores)
        self.tail_io_waves = True
        # frame chunks alternate between this many HIP streams: a launch's workgroups run in lockstep (all in their MFMA
        # phase, then all in their memory phase), two chunks in flight put the memory phase of one under the MFMA phase
        # of the other (tools/probe_tail_stagger.py: -10 % on the res4 tails; backbone -5 %)
        self.streams = 2
        self._side_streams = {}

    def forward(self, images, bf16=False):
        with torch.no_grad():
            dev = _compute_device(images, self.stem.conv1.weight)
            if images.dim() != 4 or images.shape[3] != 3:
                raise ValueError(f"images must be channels-last [T,H,W,3], got {tuple(images.shape)}")
            for m in self.modules():
                if isinstance(m, BottleneckBlock):
                    m.fuse_tail = bool(self.fuse_bottlenecks)
                    m.fuse_next = bool(self.fuse_bottlenecks and self.fuse_next_conv1)
                    m.fuse_block = bool(self.fuse_bottlenecks and self.fuse_blocks)
                    m.fuse_block_proj = bool(self.fuse_first_blocks)
                    m.tail_io_waves = bool(self.tail_io_waves)
            out = []
            nchunks = -(-images.shape[0] // self.frame_chunk)
            ns = min(int(self.streams), nchunks) if images.is_cuda else 1
            if ns <= 1:
                for lo in range(0, images.shape[0], self.frame_chunk):
                    x = self.stem(_f32(images[lo:lo + self.frame_chunk], dev), out_bf16=bf16)
                    out.append(_run_blocks(self.res4, _run_blocks(self.res3, _run_blocks(self.res2, x))))
                return torch.cat(out)
            main = torch.cuda.current_stream(dev)
            side = self._side_streams.setdefault((dev.index, ns), [torch.cuda.Stream(device=dev) for _ in range(ns)])
            imgs = _f32(images, dev)
            _warm_conv_weights(self, dev, bf16)          # packed weights exist before any side stream may read them
            for st in side:
                st.wait_stream(main)
            res4 = list(self.res4)
            last = res4[-1]
            direct = all(b.stride == 1 for b in res4[1:])   # the last block's output has the shape of its input: it writes
            res = None                         # its frames of the result directly (no torch.cat of the chunks)
            for k, lo in enumerate(range(0, images.shape[0], self.frame_chunk)):
                with torch.cuda.stream(side[k % ns]):
                    x = self.stem(imgs[lo:lo + self.frame_chunk], out_bf16=bf16)
                    x = _run_blocks(self.res3, _run_blocks(self.res2, x))
                    if not direct:
                        out.append(_run_blocks(res4, x))
                        continue
                    if res is None:
                        s4 = res4[0].stride
                        hw = ((x.shape[1] - 1) // s4 + 1, (x.shape[2] - 1) // s4 + 1)
                        with torch.cuda.stream(main):      # the caller's stream owns the result
                            res = torch.empty((images.shape[0],) + hw + (last.conv3.weight.shape[0],),
                                              dtype=x.dtype, device=dev)
                    _run_blocks(res4, x, last_out=res[lo:lo + x.shape[0]])
            for st in side:
                main.wait_stream(st)
            return res if direct else torch.cat(out)
