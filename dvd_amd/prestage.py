"""Pre-stage conditioning nets on the HIP conv-net executor (dvd_convnet_* in include/dvd_hip.h) - SURVEY 8(f) rank 1.

Mirror of what the reference runs once per document before the diffusion loop (train_settings/dvd/evaluation.py:162-216):
  GeoTr_Seg_Inf  (geotr_core.py:997-1019)  document mask -> mask_x (`mask_cat`)
  Seg            (geotr_core.py:984-995)   masked image + the six U2NETP decoder maps -> `mask_y512`
  UNet           (unet_model.py:4-37)      text-line features -> `line_msk`
with the reference's class names, state_dict keys and call signatures (val_TDiff.py:57-75 builds and loads them), so
`seg.pth`, `seg_model.pth` and `line_model2.pth` load unchanged.  The architecture is written here ONCE as an op-list
builder; the library executes the list (gather + exact-f32 MFMA GEMM per conv).  No ATen compute, no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import logging
from collections import OrderedDict

import numpy as np
import torch
from torch import nn

from . import lib, synth
from .lib import ptr, stream_ptr

_log = logging.getLogger(__name__)

CONV, POOL, RESIZE, ADD, SIGMOID = range(5)
BN_EPS = 1e-5


# ---------------------------------------------------------------------------------------------------------
# op-list builder
# ---------------------------------------------------------------------------------------------------------
class Program:
    def __init__(self, in_c: int):
        self.ops, self.convs = [], []      # convs: (weight source descriptor, cin, cout, ks, kpad)
        self.ch = {0: in_c}
        self.w_floats = 0

    def _new(self, c):
        s = len(self.ch)
        self.ch[s] = c
        return s

    def conv(self, a, cout, src, ks=3, dil=1, act=2, b=-1):
        cin = self.ch[a] + (self.ch[b] if b >= 0 else 0)
        kp = (ks * ks * cin + 15) // 16 * 16
        dst = self._new(cout)
        self.ops.append(dict(op=CONV, a=a, b=b, dst=dst, ks=ks, dil=dil, cout=cout, act=act, w_off=self.w_floats))
        self.convs.append((src, cin, cout, ks, kp))
        self.w_floats += cout * kp + (cout + 3) // 4 * 4        # bias padded to 16 bytes
        return dst

    def pool(self, a, ceil_mode):
        dst = self._new(self.ch[a])
        self.ops.append(dict(op=POOL, a=a, dst=dst, flag=int(ceil_mode)))
        return dst

    def resize_like(self, a, like, align):
        dst = self._new(self.ch[a])
        self.ops.append(dict(op=RESIZE, a=a, b=like, dst=dst, flag=int(align)))
        return dst

    def add(self, a, b):
        dst = self._new(self.ch[a])
        self.ops.append(dict(op=ADD, a=a, b=b, dst=dst))
        return dst

    def sigmoid(self, a):
        dst = self._new(self.ch[a])
        self.ops.append(dict(op=SIGMOID, a=a, dst=dst))
        return dst

    def c_ops(self):
        arr = (lib.CnOp * len(self.ops))()
        for i, o in enumerate(self.ops):
            arr[i].op, arr[i].a, arr[i].b, arr[i].dst = o["op"], o["a"], o.get("b", -1), o["dst"]
            arr[i].ks, arr[i].dil, arr[i].cout, arr[i].act = o.get("ks", 0), o.get("dil", 0), o.get("cout", 0), o.get("act", 0)
            arr[i].w_off, arr[i].h, arr[i].w, arr[i].flag = o.get("w_off", 0), o.get("h", 0), o.get("w", 0), o.get("flag", 0)
        return arr

    # ---- weights: eval-mode BatchNorm folded, [cout, kpad] with K order (ky*ks+kx)*cin + c, then the bias ----
    def pack(self, sd) -> torch.Tensor:
        t = lambda k: (sd[k] if torch.is_tensor(sd[k]) else torch.from_numpy(np.asarray(sd[k]))).double()  # noqa: E731
        out = torch.zeros(self.w_floats, dtype=torch.float32)
        off = 0
        for (kind, *keys), cin, cout, ks, kp in self.convs:
            if kind == "eye":           # _concat2: identity over the concatenated channels
                blk = torch.zeros(cout, kp)
                blk[:, :cin] = torch.eye(cout, cin)
                out[off:off + cout * kp] = blk.reshape(-1)
                off += cout * kp + (cout + 3) // 4 * 4
                continue
            if kind == "rebn":          # REBNCONV: conv_s1 + bn_s1 (geotr_core.py:28-34)
                p = keys[0]
                w, b = t(p + "conv_s1.weight"), t(p + "conv_s1.bias")
                bn = (t(p + "bn_s1.weight"), t(p + "bn_s1.bias"), t(p + "bn_s1.running_mean"), t(p + "bn_s1.running_var"))
            elif kind == "dc":          # DoubleConv entry idx: conv idx + BatchNorm idx+1 (unet_parts.py:16-23)
                p, idx = keys
                w, b = t(p + f"double_conv.{idx}.weight"), t(p + f"double_conv.{idx}.bias")
                q = p + f"double_conv.{idx + 1}."
                bn = (t(q + "weight"), t(q + "bias"), t(q + "running_mean"), t(q + "running_var"))
            else:                       # plain conv
                w, b, bn = t(keys[0]), t(keys[1]), None
            if tuple(w.shape) != (cout, cin, ks, ks):
                raise lib.DvdError(f"conv weight {keys}: shape {tuple(w.shape)}, the net expects {(cout, cin, ks, ks)}")
            if bn is not None:
                g, beta, mean, var = bn
                s = g / torch.sqrt(var + BN_EPS)
                w = w * s[:, None, None, None]
                b = (b - mean) * s + beta
            flat = w.permute(0, 2, 3, 1).reshape(cout, ks * ks * cin)
            blk = torch.zeros(cout, kp, dtype=torch.float64)
            blk[:, :flat.shape[1]] = flat
            out[off:off + cout * kp] = blk.reshape(-1).float()
            out[off + cout * kp:off + cout * kp + cout] = b.float()
            off += cout * kp + (cout + 3) // 4 * 4
        return out


def _rebn(P, p, a, cout, dil=1, b=-1):
    return P.conv(a, cout, ("rebn", p), 3, dil, 2, b)


def _rsu(P, p, x, kind):
    """RSU-7/6/5/4 (kind = depth; geotr_core.py:48-296) or RSU-4F (kind = '4f'; :300-332) over one source."""
    return _rsu_tail(P, p, _rebn(P, p + "rebnconvin.", x, 64), kind)


def build_u2netp(prefix: str = ""):
    """U2NETP.forward (geotr_core.py:779-845) -> (program, output slots of sigmoid(d0), hx6, hx5d, hx4d, hx3d, hx2d, hx1d).
    The channel concatenations torch.cat((up, skip), 1) are folded into the consuming conv's gather."""
    P, q = Program(3), prefix
    hx1 = _rsu(P, q + "stage1.", 0, 7)
    hx2 = _rsu(P, q + "stage2.", P.pool(hx1, True), 6)
    hx3 = _rsu(P, q + "stage3.", P.pool(hx2, True), 5)
    hx4 = _rsu(P, q + "stage4.", P.pool(hx3, True), 4)
    hx5 = _rsu(P, q + "stage5.", P.pool(hx4, True), "4f")
    hx6 = _rsu(P, q + "stage6.", P.pool(hx5, True), "4f")

    def dec(name, lower, skip, kind):
        """stageNd(torch.cat((upsample_like(lower, skip), skip), 1)): the RSU's first conv reads the two sources."""
        up = P.resize_like(lower, skip, False)
        p = q + name + "."
        hxin = _rebn(P, p + "rebnconvin.", up, 64, 1, skip)
        return _rsu_tail(P, p, hxin, kind)
    hx5d = dec("stage5d", hx6, hx5, "4f")
    hx4d = dec("stage4d", hx5d, hx4, 4)
    hx3d = dec("stage3d", hx4d, hx3, 5)
    hx2d = dec("stage2d", hx3d, hx2, 6)
    hx1d = dec("stage1d", hx2d, hx1, 7)
    side = lambda k, t: P.conv(t, 1, ("plain", q + f"side{k}.weight", q + f"side{k}.bias"), 3, 1, 0)  # noqa: E731
    d1 = side(1, hx1d)
    ds = [d1] + [P.resize_like(side(k, t), d1, False) for k, t in ((2, hx2d), (3, hx3d), (4, hx4d), (5, hx5d), (6, hx6))]
    # outconv over cat(d1..d6): a 1x1 conv of six 1-channel maps = chain the concatenation two sources at a time
    cat = ds[0]
    for nxt in ds[1:]:
        cat = _concat2(P, cat, nxt)
    d0 = P.conv(cat, 1, ("plain", q + "outconv.weight", q + "outconv.bias"), 1, 1, 0)
    return P, [P.sigmoid(d0), hx6, hx5d, hx4d, hx3d, hx2d, hx1d]


def _concat2(P, a, b):
    """torch.cat((a, b), 1) as an identity 1x1 conv over the two sources (exact: one non-zero product per output)."""
    ca, cb = P.ch[a], P.ch[b]
    return P.conv(a, ca + cb, ("eye", ca + cb), 1, 1, 0, b)


def _rsu_tail(P, p, hxin, kind, mid=16, cout=64):
    """Everything of an RSU after rebnconvin (which the caller emitted, possibly over two concatenated sources)."""
    if kind == "4f":
        h1 = _rebn(P, p + "rebnconv1.", hxin, mid, 1)
        h2 = _rebn(P, p + "rebnconv2.", h1, mid, 2)
        h3 = _rebn(P, p + "rebnconv3.", h2, mid, 4)
        h4 = _rebn(P, p + "rebnconv4.", h3, mid, 8)
        h3d = _rebn(P, p + "rebnconv3d.", h4, mid, 4, h3)
        h2d = _rebn(P, p + "rebnconv2d.", h3d, mid, 2, h2)
        h1d = _rebn(P, p + "rebnconv1d.", h2d, cout, 1, h1)
        return P.add(h1d, hxin)
    depth = kind
    enc, hx = [], hxin
    for k in range(1, depth):
        hk = _rebn(P, p + f"rebnconv{k}.", hx, mid)
        enc.append(hk)
        hx = P.pool(hk, True) if k <= depth - 2 else hk
    d = _rebn(P, p + f"rebnconv{depth}.", hx, mid, 2)
    for k in range(depth - 1, 0, -1):
        d = _rebn(P, p + f"rebnconv{k}d.", d, mid if k > 1 else cout, 1, enc[k - 1])
        if k > 1:
            d = P.resize_like(d, enc[k - 2], False)
    return P.add(d, hxin)


def build_unet():
    """UNet.forward, bilinear=True (unet_model.py:25-37; unet_parts.py:28-68) -> (program, [x (64 ch), logits])."""
    P = Program(3)

    def dconv(p, a, mid, cout, b=-1):
        h = P.conv(a, mid, ("dc", p, 0), 3, 1, 2, b)
        return P.conv(h, cout, ("dc", p, 3), 3, 1, 2)
    x1 = dconv("inc.", 0, 64, 64)
    skips, h = [x1], x1
    for k, c in enumerate((128, 256, 512, 512), 1):
        h = dconv(f"down{k}.maxpool_conv.1.", P.pool(h, False), c, c)
        skips.append(h)
    for k, (cin, cout) in enumerate(((1024, 256), (512, 128), (256, 64), (128, 64)), 1):
        skip = skips[4 - k]
        up = P.resize_like(h, skip, True)        # nn.Upsample(scale_factor=2, align_corners=True); sizes are even, pad = 0
        h = dconv(f"up{k}.conv.", skip, cin // 2, cout, up)      # cat([skip, up], 1)
    logits = P.conv(h, 1, ("plain", "outc.conv.weight", "outc.conv.bias"), 1, 1, 0)
    return P, [h, logits]


# ---------------------------------------------------------------------------------------------------------
# executor handle
# ---------------------------------------------------------------------------------------------------------
class ConvNet:
    """One net at one input size on one device: op list -> library handle, workspace, packed weights."""

    def __init__(self, program: Program, outputs, in_hw, device="cuda", batch: int = 1):
        self.program, self.outputs, self.device = program, list(outputs), torch.device(device)
        self.in_c, (self.in_h, self.in_w), self.batch = program.ch[0], in_hw, int(batch)
        ops = program.c_ops()
        h = C.c_void_p()
        lib.call("dvd_convnet_create_batched", ops, len(program.ops), len(program.ch), self.in_c, self.in_h, self.in_w,
                 self.batch, C.byref(h))
        self._h = h
        if lib.raw().dvd_convnet_weight_floats(h) != program.w_floats:
            raise lib.DvdError("conv-net weight layout mismatch between the host builder and the library")
        nbytes = lib.raw().dvd_convnet_workspace_bytes(h)
        self.workspace = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
        self._ws = self.workspace.data_ptr() + (-self.workspace.data_ptr()) % 256
        self._ws_bytes = nbytes
        self.shapes = []
        hh, ww, cc = C.c_int(), C.c_int(), C.c_int()
        for s in self.outputs:
            lib.call("dvd_convnet_slot_shape", h, s, C.byref(hh), C.byref(ww), C.byref(cc))
            self.shapes.append((cc.value, hh.value, ww.value))
        self.weights = None

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib.raw().dvd_convnet_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def weight_bytes(self) -> int:
        return self.program.w_floats * 4

    def bind_weights(self, blob_f32: torch.Tensor):
        assert blob_f32.dtype == torch.float32 and blob_f32.numel() == self.program.w_floats
        assert blob_f32.device == self.workspace.device and blob_f32.data_ptr() % 16 == 0
        self.weights = blob_f32

    def load_state_dict(self, sd):
        self.bind_weights(self.program.pack(sd).to(self.device))

    def run(self, x: torch.Tensor):
        """x [N, C, H, W] f32 on the device (N = the batch this executor was built for) -> list (one per requested output)
        of [N, c, h, w] tensors.  ONE pass of the op list for the whole batch."""
        if self.weights is None:
            raise lib.DvdError("conv net has no weights bound")
        from .engine import _is_dev
        if x.dim() != 4 or tuple(x.shape[1:]) != (self.in_c, self.in_h, self.in_w) or x.dtype != torch.float32 \
                or not _is_dev(x) or not x.is_contiguous():
            raise lib.DvdError(f"conv net expects a contiguous f32 device tensor [N,{self.in_c},{self.in_h},{self.in_w}], "
                               f"got {tuple(x.shape)}")
        n = x.shape[0]
        if n != self.batch:
            raise lib.DvdError(f"conv net executor built for a batch of {self.batch}, got {n} images")
        outs = [torch.empty((n, *shp), dtype=torch.float32, device=self.device) for shp in self.shapes]
        slots = (C.c_int * len(self.outputs))(*self.outputs)
        ptrs = (C.c_void_p * len(outs))(*[o.data_ptr() for o in outs])
        lib.call("dvd_convnet_run", self._h, ptr(x), ptr(self.weights), C.c_void_p(self._ws), self._ws_bytes,
                 len(outs), slots, ptrs, stream_ptr())
        return outs


def resize_bilinear(x: torch.Tensor, size, align_corners: bool) -> torch.Tensor:
    """F.interpolate(x, size, mode='bilinear', align_corners=...) for a contiguous f32 device tensor [N,C,H,W]."""
    hout, wout = (size, size) if isinstance(size, int) else size
    n, c, hin, win = x.shape
    out = torch.empty((n, c, hout, wout), dtype=torch.float32, device=x.device)
    lib.call("dvd_resize_bilinear_nchw", ptr(x), ptr(out), n * c, hin, win, hout, wout, int(align_corners), stream_ptr())
    return out


def threshold_mask_mul(d0: torch.Tensor, x: torch.Tensor, thr: float = 0.5):
    """(d0 > thr).float() * x per sample (geotr_core.py:989-990): d0 [N,1,H,W], x [N,C,H,W] -> (mskx, mask)."""
    n, c, h, w = x.shape
    out, mask = torch.empty_like(x), torch.empty_like(d0)
    lib.call("dvd_threshold_mask_mul_batch", ptr(d0), ptr(x), ptr(out), ptr(mask), n, c, h * w, C.c_float(thr), stream_ptr())
    return out, mask


# ---------------------------------------------------------------------------------------------------------
# nn.Module mirrors with the reference's names
# ---------------------------------------------------------------------------------------------------------
class _RefNet(nn.Module):
    """Holds the reference-named tensors of one conv net (so checkpoints load unchanged) and runs it on the executor."""
    kind = None

    def __init__(self):
        super().__init__()
        self._names = OrderedDict()
        spec = synth.u2netp_spec() if self.kind == "u2netp" else synth.unet_spec()
        for key, (shape, knd) in spec.items():
            flat = key.replace(".", "__")
            self._names[key] = flat
            if knd in ("bn_m", "bn_v", "bn_n"):
                init = torch.ones(shape) if knd == "bn_v" else torch.zeros(shape, dtype=torch.long if knd == "bn_n" else torch.float32)
                self.register_buffer(flat, init)
            else:
                self.register_parameter(flat, nn.Parameter(torch.zeros(shape), requires_grad=False))
        self._nets, self._version, self._blob, self._blob_version = {}, 0, None, -1
        self._program, self._outputs = build_u2netp() if self.kind == "u2netp" else build_unet()

    def state_dict(self, *args, prefix="", **kwargs):
        sd = super().state_dict(*args, **kwargs)
        return type(sd)((prefix + key, sd[flat]) for key, flat in self._names.items() if flat in sd)

    def load_state_dict(self, state_dict, strict=True, **kw):
        mapped = {self._names[k]: v for k, v in state_dict.items() if k in self._names}
        unexpected = [k for k in state_dict if k not in self._names]
        res = super().load_state_dict(mapped, strict=False, **kw)
        missing = [k for k, f in self._names.items() if f in res.missing_keys]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing[:5]}..., unexpected {unexpected[:5]}...")
        self._version += 1
        return res

    @property
    def device(self):
        return next(self.parameters()).device

    # ---- flat weight blob (rank 0 packs, one broadcast hands it to every rank: dist_util.materialize_blobs) ----
    def blob_bytes(self) -> int:
        return self._program.w_floats * 4

    def pack_into(self, view_u8: torch.Tensor):
        packed = self._program.pack({k: v.detach().cpu() for k, v in self.state_dict().items()})
        view_u8.copy_(packed.view(torch.uint8))

    def bind_blob(self, view_u8: torch.Tensor):
        self._blob, self._blob_version = view_u8.view(torch.float32), self._version

    MAX_EXECUTORS_PER_SHAPE = 2

    def _net(self, h, w, batch=1) -> ConvNet:
        """The executor for (batch, h, w): one per batch size - its activation slots hold all the images of a batch, so the
        net's ~500 ops are enqueued once per BATCH of documents, not once per document."""
        from .cross_model import _require_gpu
        dev = self.device
        _require_gpu(dev)
        key = (batch, h, w, dev.index)
        net = self._nets.get(key)
        if net is None:
            # At most TWO executors (and workspaces) per image size and device, least recently used out: the ragged last
            # batch of a run or a service that alternates single documents with batches keeps both of its executors
            # (one per batch size evicted the other on every call - a silent rebuild + weight re-bind each time), while a
            # sweep over many batch sizes still cannot pile up full sets of activation slots.
            same = [k for k in self._nets if k[1:] == key[1:]]
            for old in same[:max(0, len(same) - (self.MAX_EXECUTORS_PER_SHAPE - 1))]:
                del self._nets[old]
                _log.debug("prestage: evicted the executor for batch %d at %dx%d", old[0], h, w)
            net = ConvNet(self._program, self._outputs, (h, w), device=dev, batch=batch)
            net._bound = -1
        else:
            del self._nets[key]                      # re-inserted below: dict order = recency
        self._nets[key] = net
        if net._bound != self._version:
            if self._blob is None or self._blob_version != self._version:
                from . import dist_util
                if dist_util.world_size() > 1:
                    raise RuntimeError("weights changed in a multi-rank run: call dist_util.materialize_blobs([...]) on "
                                       "every rank first (the executor never communicates)")
                dist_util.materialize_blobs([self])
            net.bind_weights(self._blob)
            net._bound = self._version
        return net

    def _run(self, x):
        x = x.to(self.device, torch.float32).contiguous()
        return self._net(x.shape[2], x.shape[3], x.shape[0]).run(x)


class U2NETP(_RefNet):
    """geotr_core.py:745-845: forward(x) -> (sigmoid(d0), hx6, hx5d, hx4d, hx3d, hx2d, hx1d)."""
    kind = "u2netp"

    def __init__(self, in_ch=3, out_ch=1):
        if (in_ch, out_ch) != (3, 1):
            raise NotImplementedError("the sampling path builds U2NETP(3, 1)")
        super().__init__()

    def forward(self, x):
        return tuple(self._run(x))


class UNet(_RefNet):
    """unet_model.py:4-37: forward(x) -> (x [N,64,H,W], logits [N,1,H,W])."""
    kind = "unet"

    def __init__(self, n_channels=3, n_classes=1, bilinear=True):
        if (n_channels, n_classes, bilinear) != (3, 1, True):
            raise NotImplementedError("the sampling path builds UNet(n_channels=3, n_classes=1)")
        super().__init__()

    def forward(self, x):
        h, logits = self._run(x)
        return h, logits


class _MskHolder(nn.Module):
    """A module whose only trained child is `self.msk = U2NETP(3, 1)` (state_dict keys 'msk.<...>')."""

    def __init__(self):
        super().__init__()
        self.msk = U2NETP(3, 1)

    def state_dict(self, *args, **kwargs):
        return self.msk.state_dict(prefix="msk.")

    # the flat-blob protocol of dist_util.materialize_blobs is the child's
    @property
    def device(self):
        return self.msk.device

    def blob_bytes(self):
        return self.msk.blob_bytes()

    def pack_into(self, view):
        self.msk.pack_into(view)

    def bind_blob(self, view):
        self.msk.bind_blob(view)

    def load_state_dict(self, state_dict, strict=True, **kw):
        own = {k[4:]: v for k, v in state_dict.items() if k.startswith("msk.")}
        extra = [k for k in state_dict if not k.startswith("msk.")]
        if strict and extra:
            raise RuntimeError(f"load_state_dict: unexpected {extra[:5]}...")
        return self.msk.load_state_dict(own, strict=strict, **kw)


class Seg(_MskHolder):
    """geotr_core.py:984-995: forward(x) -> (mskx, d0 resized to 512 (align_corners=True), hx6, hx5d, hx4d, hx3d, hx2d, hx1d)."""

    def forward(self, x):
        d0, hx6, hx5d, hx4d, hx3d, hx2d, hx1d = self.msk(x)
        mskx, _ = threshold_mask_mul(d0, x.to(d0.device, torch.float32).contiguous(), 0.5)
        return mskx, resize_bilinear(d0, 512, True), hx6, hx5d, hx4d, hx3d, hx2d, hx1d


class GeoTr_Seg_Inf(_MskHolder):
    """geotr_core.py:997-1019: forward(x) -> (bm, msk resized to 512).  The GeoTr branch that produces `bm` is DEAD on
    the live configuration - val_TDiff.py:57-58 only loads weights into `.msk`, and evaluation.py:176-181 reads `bm`
    only under env.use_init_flow - so it is not evaluated and `bm` is None."""

    def forward(self, x):
        msk = self.msk(x)[0]
        return None, resize_bilinear(msk, 512, True)


def _strip_and_load(model, sd, n):
    own = model.state_dict()
    picked = {k[n:]: v for k, v in sd.items() if k[n:] in own}
    own.update(picked)
    model.load_state_dict(own, strict=True)
    return model


def reload_segmodel(model, path=""):
    """geotr_core.py:1090-1112: load a checkpoint whose keys carry a 6-character prefix ('model.')."""
    if not bool(path):
        return model
    return _strip_and_load(model, torch.load(path, map_location="cpu"), 6)


def reload_model(model, path=""):
    """geotr_core.py:1075-1088: the same with a 7-character prefix ('module.')."""
    if not bool(path):
        return model
    return _strip_and_load(model, torch.load(path, map_location="cpu"), 7)


# ---------------------------------------------------------------------------------------------------------
# the glue of evaluation.py:162-216
# ---------------------------------------------------------------------------------------------------------
_SIDE_STREAMS = {}


def _side_stream(device):
    """One side HIP stream per device, created once: creating a stream costs several hundred microseconds of host time - it was
    the largest single gap in the GPU timeline of a document at the reference's operating point (profiles/r5_native_single_trace.txt)."""
    key = torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device()
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return st


def conditioning(dewarp_model, seg_model, line_model, source512: torch.Tensor, grid: int):
    """source512 [N,3,512,512] in 0..1 on the device -> dict(mask_cat [N,1,512,512], mask_y512 [N,384,G,G],
    line_msk [N,64,G,G]) exactly as evaluation.py:162-216 builds them (use_gt_mask False, use_line_mask True)."""
    source_288 = resize_bilinear(source512.to(torch.float32).contiguous(), 288, True)                    # :162
    # The document-mask pass (GeoTr_Seg_Inf.msk) and the Seg pass are independent U2NETP evaluations of the same input,
    # each a chain of several hundred small, latency-bound kernels that fill a few of the 256 CUs: the first runs on a
    # side HIP stream beside the Seg -> line-UNet chain and is joined at the end.  The LONGER chain (Seg, then the line
    # UNet) is enqueued first: the host needs ~0.6 ms to enqueue a net's launches, and whichever chain is enqueued second
    # starts that much later - the short one can afford it (round 5; it waits for the resized image only, through an event).
    side = _side_stream(source_288.device) if source_288.is_cuda else None
    if side is not None:
        cur = torch.cuda.current_stream(source_288.device)
        ready = torch.cuda.Event()
        ready.record(cur)
    mskx, d0, hx6, hx5d, hx4d, hx3d, hx2d, hx1d = seg_model(source_288)                                   # :198
    seg_map_all = torch.cat([resize_bilinear(t, grid, False) for t in (hx6, hx5d, hx4d, hx3d, hx2d, hx1d)], dim=1)
    textline_map, _ = line_model(mskx)                                                                   # :209
    line_msk = resize_bilinear(textline_map, grid, False)
    if side is not None:
        with torch.cuda.stream(side):
            side.wait_event(ready)
            _, mask_x = dewarp_model(source_288)                                                         # :176
        cur.wait_stream(side)
        mask_x.record_stream(cur)
        source_288.record_stream(side)
    else:
        _, mask_x = dewarp_model(source_288)
    return {"mask_cat": mask_x, "mask_y512": seg_map_all, "line_msk": line_msk,
            "mskx": mskx, "d0": d0}
