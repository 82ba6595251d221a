"""`network` / `siamese_network` -- the reference's model API (model/model_dupl.py:9-213) on the HIP engine.

Same constructor arguments, forward() modes, attributes read by callers (.encoder.patch_size,
.encoder.embed_dim, .decoder, .classifier, .aux_classifier), get_param_groups() grouping and
state_dict keys (157 per student) as the reference, so reference checkpoints load and the
reference's training loop drives it unchanged.  What differs is underneath:

* all parameters of all students live in ONE flat fp32 buffer (engine.FlatStorage); the nn.Parameters
  are views into it and their .grad are views into one flat gradient buffer;
* forward/backward are explicit HIP kernel schedules (engine.network_forward / network_backward);
  autograd sees one node per student forward, whose backward accumulates parameter gradients
  directly into the flat gradient buffer (they are not returned through autograd);
* `cam_with_grad` (model_dupl.py:100-104): the CAM logits of the detached classifier are a batched HIP GEMM on the
  x4 output of the student's autograd node (`_CamGradFn`), so their gradient reaches the encoder through `dx4`.
* need_sp: the reference first runs both students on the full 2b batch and discards the result
  (model_dupl.py:191-192); that dead forward is skipped here -- outputs are identical.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import os
import weakref

import torch
import torch.nn as nn

from .. import engine, ops
from ..engine import EncoderConfig, FlatStorage, StudentParams
from .backbone import encoder_config, load_pretrained_encoder
from .decoder.conv_head import LargeFOV


_STREAM_PAIRS = {}     # device -> the two student streams (see siamese_network.enable_dual_stream)

class _Holder(nn.Module):
    """Bare container used to reproduce the reference's state_dict key hierarchy."""

    def forward(self, *a, **k):
        raise RuntimeError("parameter container: compute is scheduled by dupl_amd.engine")


def _attach(root: nn.Module, dotted: str, param: nn.Parameter):
    parts = dotted.split(".")
    m = root
    for p in parts[:-1]:
        if not hasattr(m, p):
            setattr(m, p, _Holder())
        m = getattr(m, p)
    m.register_parameter(parts[-1], param)


class _NetworkFn(torch.autograd.Function):
    """One autograd node per student forward.  Parameters are NOT autograd inputs: backward writes their
    gradients into the flat gradient buffer and returns nothing for them."""

    @staticmethod
    def forward(ctx, anchor, x, net, enc_cache=None):
        outs, sv = engine.network_forward(net._P, x, save=True, enc_cache=enc_cache)
        ctx.net, ctx.sv = net, sv
        ctx.set_materialize_grads(False)
        net._live_graphs += 1     # forwards of this student whose backward has not run yet (phase C has two)
        return outs

    @staticmethod
    def backward(ctx, dcls, dseg, dx4, dcls_aux):
        net = ctx.net
        on_ready = None
        if net._grad_ready_hooks and net._live_graphs <= 1:
            # last pending forward of this student: its gradient ranges become final block by block (phase C has
            # two forwards per student; the first one to be back-propagated only accumulates)
            def on_ready(event):
                for hook in net._grad_ready_hooks:
                    hook(net, event)
        engine.network_backward(net._P, ctx.sv, dcls, dseg, dx4, dcls_aux, on_ready=on_ready)
        ctx.sv = None
        net._live_graphs = max(0, net._live_graphs - 1)
        for hook in net._post_backward_hooks:
            hook(net)
        return None, None, None, None


class _EncoderFn(torch.autograd.Function):
    """forward_features (vit.py:308-326) as one autograd node: (x[:, 0], x[:, 1:], embeds[aux_layer][:, 1:]).  Parameter gradients go
    to the flat gradient buffer (engine.encoder_backward), as for _NetworkFn."""

    @staticmethod
    def forward(ctx, anchor, x, net):
        tf, aux, enc = engine.encoder_forward(net._P, x, save=True)
        B = x.shape[0]
        D = tf.shape[1]
        N = tf.shape[0] // B
        ctx.net, ctx.enc, ctx.dims = net, enc, (B, N, D)
        ctx.aux_is_final = aux.data_ptr() == tf.data_ptr()
        ctx.set_materialize_grads(False)
        tf3, aux3 = tf.view(B, N, D), aux.view(B, N, D)
        return tf3[:, 0].clone(), tf3[:, 1:].clone(), aux3[:, 1:].clone()

    @staticmethod
    def backward(ctx, dcls, dtok, daux):
        net = ctx.net
        B, N, D = ctx.dims
        dev = net._store.data.device
        dtf = ops.zeros((B * N, D), dev)
        d3 = dtf.view(B, N, D)
        if dcls is not None:
            d3[:, 0] += dcls
        if dtok is not None:
            d3[:, 1:] += dtok
        dta = None
        if daux is not None:
            if ctx.aux_is_final:
                d3[:, 1:] += daux
            else:
                dta = ops.zeros((B * N, D), dev)
                dta.view(B, N, D)[:, 1:] += daux
        engine.encoder_backward(net._P, ctx.enc, dtf, dta)
        ctx.enc = None
        for hook in net._post_backward_hooks:
            hook(net)
        return None, None, None


class _Encoder(_Holder):
    """`network.encoder`: the parameter hierarchy of the reference's VisionTransformer (state_dict keys) plus its one compute entry
    point on the DuPL path, forward_features (vit.py:308-326), scheduled on the HIP engine."""

    def forward_features(self, x):
        """(x[:, 0] (B, D), x[:, 1:] (B, n, D), embeds[aux_layer][:, 1:] (B, n, D)) -- final-LayerNorm cls token and patch tokens, and the
        patch tokens of the un-normalised output of block `aux_layer` (the final-LayerNorm ones when it is the last block)."""
        net = self._net()
        if not x.is_cuda:
            raise RuntimeError("dupl_amd runs on an MI355X only (no CPU path); move the model and inputs to cuda")
        x = x.contiguous().float()
        if torch.is_grad_enabled() and net.classifier.weight.requires_grad:
            return _EncoderFn.apply(net._anchor_for(x.device), x, net)
        tf, aux, _ = engine.encoder_forward(net._P, x, save=False)
        B = x.shape[0]
        D = tf.shape[1]
        t3, a3 = tf.view(B, -1, D), aux.view(B, -1, D)
        return t3[:, 0], t3[:, 1:], a3[:, 1:]

    def forward(self, x):
        """VisionTransformer.forward is never called on the DuPL path (vit.py:328-335 applies the unused 1000-way head); the call the
        path makes is forward_features."""
        return self.forward_features(x)


class _CamGradFn(torch.autograd.Function):
    """F.conv2d(x4, classifier.weight.detach()) of model_dupl.py:101 as a batched MFMA GEMM on NCHW operands:
    cam[b] (C x hw) = W (C x D) . x4[b] (D x hw); backward dx4[b] = W^T . dcam[b] (the weight is detached: no dW)."""

    @staticmethod
    def forward(ctx, x4, W):
        B, D, h, w = x4.shape
        C = W.shape[0]
        x4 = x4.contiguous()
        cam = torch.empty((B, C, h, w), device=x4.device, dtype=torch.float32)
        ops.gemm_raw(W.data_ptr(), x4.data_ptr(), cam.data_ptr(), C, h * w, D, D, h * w, h * w,
                     flags=ops._lib.GEMM_B_NCONTIG, batch=B, sB=(D * h * w, 0), sC=(C * h * w, 0))
        ctx.save_for_backward(W)
        ctx.dims = (B, D, h, w, C)
        return cam

    @staticmethod
    def backward(ctx, dcam):
        (W,) = ctx.saved_tensors
        B, D, h, w, C = ctx.dims
        dcam = dcam.contiguous()
        dx4 = torch.empty((B, D, h, w), device=dcam.device, dtype=torch.float32)
        ops.gemm_raw(W.data_ptr(), dcam.data_ptr(), dx4.data_ptr(), D, h * w, C, D, h * w, h * w,
                     flags=ops._lib.GEMM_A_MCONTIG | ops._lib.GEMM_B_NCONTIG, batch=B, sB=(C * h * w, 0), sC=(D * h * w, 0))
        return dx4, None


def _rebuild_model(kind: str, ctor, state, device, dual: bool):
    """copy.deepcopy / pickle of a dupl_amd model: the Parameters are views of ONE flat buffer (and the encoder holds a
    weak reference to its network), so a member-wise copy would detach them from the storage the engine computes on.  A copy is a NEW
    model of the same configuration with the state loaded into its own flat buffer, on the same device, with the same stream mode."""
    backbone, num_classes, aux_layer = ctor
    cls = siamese_network if kind == "siamese" else network
    m = cls(backbone, num_classes=num_classes, pretrained=False, aux_layer=aux_layer)
    m.load_state_dict(state, strict=True)
    if device is not None and torch.device(device).type != "cpu":
        m.to(device)
    if dual and kind == "siamese":
        m.enable_dual_stream(True)
    return m


class network(nn.Module):
    def __init__(self, backbone, num_classes=None, pretrained=None, aux_layer=None, add_mlp=False,
                 _store: Optional[FlatStorage] = None, _student: int = 0):
        super().__init__()
        if add_mlp:
            raise NotImplementedError("add_mlp is never enabled on the DuPL training path (model_dupl.py:112,115)")
        self.num_classes = num_classes
        self.add_mlp = add_mlp
        cfg = encoder_config(backbone, aux_layer)
        self._cfg = cfg
        self._ctor = (backbone, num_classes, aux_layer)
        self._owns_store = _store is None
        self._store = FlatStorage(cfg, num_classes, 1) if _store is None else _store
        self._student = _student
        self._post_backward_hooks: List[Callable] = []
        self._grad_ready_hooks: List[Callable] = []     # hook(net, event) during the last pending backward (ddp.py)
        self._live_graphs = 0
        self._anchor = None
        self._build_modules()
        if self._owns_store:
            _reference_init(self._store, 0)
            self._rebind()
        self.in_channels = [cfg.embed_dim] * 4
        if pretrained and not isinstance(pretrained, bool):
            load_pretrained_encoder(self.encoder, pretrained, cfg.patch)
        elif pretrained:
            raise RuntimeError("pretrained=True needs network access (torch.hub / timm URLs in the reference); pass a "
                               "local state_dict path or pretrained=False")
        self.register_load_state_dict_post_hook(lambda m, k: m._invalidate())

    # ---- construction -------------------------------------------------------------------------
    def _build_modules(self):
        st, s = self._store, self._student
        enc = _Encoder()
        object.__setattr__(enc, "_net", weakref.ref(self))     # not a submodule: the encoder is a child of the network
        enc.patch_size = self._cfg.patch
        enc.embed_dim = enc.num_features = self._cfg.embed_dim
        enc.aux_layer = self._cfg.aux_layer
        self.encoder = enc
        params: Dict[str, nn.Parameter] = {}
        for key in engine.student_param_shapes(self._cfg, self.num_classes):
            p = nn.Parameter(st.view(s, key), requires_grad=(key != "encoder.pos_embed"))
            params[key] = p
        # encoder.blocks must be an indexable ModuleList-like holder with integer names
        blocks = nn.ModuleList([_Holder() for _ in range(self._cfg.depth)])
        enc.blocks = blocks
        for key, p in params.items():
            if key.startswith("encoder.blocks."):
                _, _, idx, rest = key.split(".", 3)
                _attach(blocks[int(idx)], rest, p)
            elif key.startswith("encoder."):
                _attach(enc, key[len("encoder."):], p)
        self.decoder = LargeFOV(params["decoder.conv6.weight"], params["decoder.conv7.weight"],
                                params["decoder.conv8.weight"], dilation=self._cfg.decoder_dilation)
        self.classifier = _Holder()
        self.classifier.register_parameter("weight", params["classifier.weight"])
        self.aux_classifier = _Holder()
        self.aux_classifier.register_parameter("weight", params["aux_classifier.weight"])
        self._params_by_key = params
        if st.version_anchor is None:
            st.version_anchor = next(iter(params.values()))     # (FlatStorage._param_key: the Parameters' shared version counter)
        self._P = StudentParams(st, s)

    def _rebind(self):
        """Point every Parameter (and its .grad) at the current flat buffers."""
        st, s = self._store, self._student
        for key, p in self._params_by_key.items():
            p.data = st.view(s, key)
            p.grad = st.view(s, key, grad=True) if p.requires_grad else None
        self._P = StudentParams(st, s)
        self._anchor = None

    def _invalidate(self):
        self._P._pos_cache.clear()
        self._P._pos_version = None

    def _apply(self, fn, recurse=True):
        if not self._owns_store:
            # a student of a siamese_network is a view into the pair's flat buffer: moving / casting one student alone
            # would silently do nothing -- refuse instead
            raise RuntimeError("this student's parameters live in its siamese_network's flat storage: call "
                               ".to() / .cuda() / .float() on the siamese_network, not on branch1 / branch2")
        self._store.apply(fn)
        assert self._store.data.dtype == torch.float32, "dupl_amd is an fp32-storage engine"
        self._rebind()
        return self

    def _copy_args(self):
        if not self._owns_store:
            raise RuntimeError("a student of a siamese_network shares the pair's flat storage: copy / pickle the siamese_network")
        sd = {k: v.detach().cpu().clone() for k, v in self.state_dict().items()}
        return ("single", self._ctor, sd, str(self._store.data.device), False)

    def __deepcopy__(self, memo):
        return _rebuild_model(*self._copy_args())

    def __reduce_ex__(self, protocol):
        return _rebuild_model, self._copy_args()

    # ---- reference API ------------------------------------------------------------------------
    def get_param_groups(self):
        """[backbone, backbone_norm ("norm" in name), cls heads, seg head] -- the grouping the reference intends
        (model_dupl.py:43-62; as shipped it dereferences self.mlp_layer and raises when add_mlp=False)."""
        groups = [[], [], [], []]
        for name, p in self.encoder.named_parameters():
            groups[1 if "norm" in name else 0].append(p)
        groups[2] += [self.classifier.weight, self.aux_classifier.weight]
        groups[3] += list(self.decoder.parameters())
        return groups

    def to_2D(self, x, h, w):
        n, hw, c = x.shape
        return x.transpose(1, 2).reshape(n, c, h, w)

    def _anchor_for(self, device):
        if self._anchor is None or self._anchor.device != device:
            self._anchor = torch.zeros(1, device=device, requires_grad=True)
        return self._anchor

    def forward_shared(self, x, enc_cache):
        """Training forward that reuses the encoder pass ms-CAM already ran on this x (engine.cam_logits_shared)."""
        return _NetworkFn.apply(self._anchor_for(x.device), x, self, enc_cache)

    def forward(self, x, cam_only=False, val=False, cam_with_grad=False):
        if not x.is_cuda:
            raise RuntimeError("dupl_amd.network runs on an MI355X only (no CPU path); move the model and inputs to cuda")
        x = x.contiguous().float()
        if cam_only:
            cam_aux_t, cam_t = engine.cam_logits(self._P, x)
            B = x.shape[0]
            h, w = x.shape[2] // self._cfg.patch, x.shape[3] // self._cfg.patch
            C = self.num_classes - 1
            cam = ops.tokens_to_nchw(cam_t, B, h * w, C, h, w, skip_cls=True)
            cam_aux = ops.tokens_to_nchw(cam_aux_t, B, h * w, C, h, w, skip_cls=True)
            return cam_aux, cam
        need_grad = torch.is_grad_enabled() and any(p.requires_grad for p in (self.classifier.weight,))
        if need_grad:
            outs = _NetworkFn.apply(self._anchor_for(x.device), x, self)
        else:
            outs, _ = engine.network_forward(self._P, x, save=False)
        if val or not cam_with_grad:
            return outs
        # model_dupl.py:100-104: CAM of the DETACHED classifier on x4, min-shifted and max-normalised; note the
        # reference's operator precedence: (cam / max) + 1e-5
        cls_x4, seg, x4, cls_aux = outs
        C = self.num_classes - 1
        cam_grad = _CamGradFn.apply(x4, self.classifier.weight.detach().view(C, -1))
        cam_grad = cam_grad - cam_grad.amin(dim=(2, 3), keepdim=True)
        cam_grad = cam_grad / cam_grad.amax(dim=(2, 3), keepdim=True) + 1e-5
        return cls_x4, seg, x4, cls_aux, cam_grad


def _trunc_normal_(t: torch.Tensor, std: float):
    torch.nn.init.trunc_normal_(t, std=std)


def _reference_init(store: FlatStorage, student: int):
    """Reference initialisation (vit.py:265-276; decoder/classifier convs keep PyTorch's default
    kaiming_uniform(a=sqrt(5)) since LargeFOV._init_weights is never called, conv_head.py:24-30)."""
    import math
    for key, shp in store.shapes.items():
        t = store.view(student, key)
        if key.startswith("decoder.") or "classifier" in key:
            fan_in = shp[1] * shp[2] * shp[3]
            bound = 1.0 / math.sqrt(fan_in)   # kaiming_uniform_(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in))
            t.uniform_(-bound, bound)
        elif key == "encoder.patch_embed.proj.weight":
            fan_in = shp[1] * shp[2] * shp[3]
            bound = 1.0 / math.sqrt(fan_in)
            t.uniform_(-bound, bound)
        elif key == "encoder.patch_embed.proj.bias":
            bound = 1.0 / math.sqrt(3 * store.cfg.patch ** 2)
            t.uniform_(-bound, bound)
        elif "norm" in key:
            t.fill_(1.0 if key.endswith("weight") else 0.0)
        elif key.endswith("bias"):
            t.zero_()
        else:  # Linear weights, pos_embed, cls_token
            _trunc_normal_(t, 0.02)


def _cu_masked_streams(dev, spec: str):
    """Experiment knob DUPL_CU_MASK="lo:hi,lo:hi": the two student streams as HIP streams restricted to the CU bit ranges
    [lo, hi) of the 256-bit CU mask (hipExtStreamCreateWithCUMask; the driver deals mask bits round-robin over the 8 XCDs,
    so "0:128,128:256" gives each student half of every XCD).  Default (unset): two ordinary streams on all CUs."""
    import ctypes
    path = next((ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln), "libamdhip64.so")
    hip = ctypes.CDLL(path)
    out = []
    for part in spec.split(","):
        lo, hi = (int(v) for v in part.split(":"))
        bits = sum(1 << b for b in range(lo, hi))
        words = [(bits >> (32 * i)) & 0xFFFFFFFF for i in range(8)]
        arr = (ctypes.c_uint32 * 8)(*words)
        h = ctypes.c_void_p()
        with torch.cuda.device(dev):
            rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), 8, arr)
        if rc != 0:
            raise RuntimeError(f"hipExtStreamCreateWithCUMask failed ({rc})")
        out.append(torch.cuda.ExternalStream(h.value, device=dev))
    assert len(out) == 2
    return out


def _cu_mask_persist_blocks(spec: str) -> int:
    """Persistent-grid size for CU-masked student streams (the width of the first mask)."""
    lo, hi = (int(v) for v in spec.split(",")[0].split(":"))
    return int(os.environ.get("DUPL_PERSIST_BLOCKS", (hi - lo) // 8 * 8))


class siamese_network(nn.Module):
    def __init__(self, backbone, num_classes=None, pretrained=None, aux_layer=None):
        super().__init__()
        cfg = encoder_config(backbone, aux_layer)
        self._ctor = (backbone, num_classes, aux_layer)
        self._store = FlatStorage(cfg, num_classes, 2)
        self.branch1 = network(backbone, num_classes=num_classes, pretrained=None, aux_layer=aux_layer,
                               _store=self._store, _student=0)
        self.branch2 = network(backbone, num_classes=num_classes, pretrained=None, aux_layer=aux_layer,
                               _store=self._store, _student=1)
        for s in (0, 1):
            _reference_init(self._store, s)
        self._rebind()
        if pretrained and not isinstance(pretrained, bool):
            load_pretrained_encoder(self.branch1.encoder, pretrained, cfg.patch)
            load_pretrained_encoder(self.branch2.encoder, pretrained, cfg.patch)
        elif pretrained:
            raise RuntimeError("pretrained=True needs network access; pass a local state_dict path or pretrained=False")

    def _rebind(self):
        self.branch1._rebind()
        self.branch2._rebind()

    def _apply(self, fn, recurse=True):
        self._store.apply(fn)
        assert self._store.data.dtype == torch.float32, "dupl_amd is an fp32-storage engine"
        self._rebind()
        return self

    def __deepcopy__(self, memo):
        return _rebuild_model(*self._copy_args())

    def __reduce_ex__(self, protocol):
        return _rebuild_model, self._copy_args()

    def _copy_args(self):
        sd = {k: v.detach().cpu().clone() for k, v in self.state_dict().items()}
        return ("siamese", self._ctor, sd, str(self._store.data.device), bool(getattr(self, "_dual", False) and self._store.streams))

    @property
    def flat_storage(self) -> FlatStorage:
        return self._store

    # ---- two-student concurrency --------------------------------------------------------------
    def enable_dual_stream(self, on: bool = True):
        """Run the two (independent) students on two HIP streams so that their kernels share the 256 CUs:
        a single student's N=768 GEMMs / attention grids do not fill the chip.  Autograd replays each student's
        backward on the stream its forward ran on; the optimiser and the gradient exchange wait on both."""
        self._dual = bool(on)
        if on and self._store.data.is_cuda and not self._store.streams:
            # one pair per device for the whole process, reused by every model and toggle: HIP maps streams onto a few hardware queues
            # round-robin, and a second pair can land on queues that serialise against each other (measured: toggling
            # off and on with fresh streams ran at single-stream speed)
            dev = self._store.data.device
            if dev not in _STREAM_PAIRS:
                spec = os.environ.get("DUPL_CU_MASK", "")
                # (round 6, measured and dropped: HIP stream priorities -- student 1 ahead of student 2, so that its light CAM / label
                # section would run under the other's forward: 51.80 / 51.86 ms per step vs 51.71 / 51.71 on the same box; both
                # streams at high priority 51.58 = noise.  profiles/r06_stream_priority.txt)
                _STREAM_PAIRS[dev] = (_cu_masked_streams(dev, spec) if spec else
                                      [torch.cuda.Stream(device=dev) for _ in range(2)])
            self._store.streams = list(_STREAM_PAIRS[dev])
        tn = self._store.gemm16_tuning       # THIS model's launch tuning (engine.FlatStorage.gemm16_tuning): nothing process-wide
        if not on:
            self._store.streams = []
        # the CU-mask experiment (DUPL_CU_MASK) sizes the persistent grids for its masks; a single-stream model does not inherit that
        spec = os.environ.get("DUPL_CU_MASK", "")
        tn["persist_blocks"] = (_cu_mask_persist_blocks(spec) if (on and spec and self._store.streams)
                                else int(os.environ.get("DUPL_PERSIST_BLOCKS", 0)))
        # the split GEMM picks its tile for the number of launches that share the chip (dupl_gemm16_desc.concurrency, a per-call field)
        if self._store.data.is_cuda:
            tn["concurrency"] = 2 if (on and self._store.streams) else 1
        return self

    def ms_cam_and_forward(self, inputs, scales, inputs_aug=None):
        """Fused step front-end: for each student the multi-scale CAMs (cam_helper.multi_scale_cam2_siamese) AND the
        training forward (self(inputs)) -- the scale-1.0 un-flipped encoder pass is shared between the two (the
        reference runs it twice with identical weights and input; outputs are identical).
        With `inputs_aug` (phase C, forward(cat([inputs, inputs_aug]), need_sp=True), model_dupl.py:193-205) the
        0.75x strong-aug segmentation forward of each student runs on that student's stream too and the result
        carries "branch1_aug" / "branch2_aug".
        Returns ((cam_1, cam_aux_1), (cam_2, cam_aux_2), {"branch1": ..., "branch2": ...})."""
        from ..utils import cam_helper
        inputs = inputs.contiguous().float()
        x_aug = None
        if inputs_aug is not None:
            H, W = inputs_aug.shape[2:]
            x_aug = ops.resize_bilinear(inputs_aug.contiguous().float(), int(H * 0.75), int(W * 0.75))

        def one(net):
            if torch.is_grad_enabled():
                share = {}
                cams = cam_helper._ms_cam(net._P, inputs, scales, share=share)
                outs = net.forward_shared(share["x"], share["enc"])
            else:
                # no-grad callers (in-loop validation): nothing to share -- a saving pass would keep every block's
                # activations alive only to drop them, and net(inputs) runs its own (cheap, non-saving) forward anyway
                cams = cam_helper._ms_cam(net._P, inputs, scales)
                outs = net(inputs)
            seg_aug = net(x_aug)[1] if x_aug is not None else None
            return cams, outs, seg_aug

        (c1, o1, a1), (c2, o2, a2) = self.per_student(lambda: one(self.branch1), lambda: one(self.branch2))
        res = {"branch1": o1, "branch2": o2}
        if x_aug is not None:
            res["branch1_aug"], res["branch2_aug"] = a1, a2
        return c1, c2, res

    def per_student(self, fn1, fn2):
        """Evaluate fn1() for student 1 and fn2() for student 2, concurrently when dual-stream is enabled."""
        if not (getattr(self, "_dual", False) and self._store.streams):
            return fn1(), fn2()
        main = torch.cuda.current_stream()
        s1, s2 = self._store.streams
        s1.wait_stream(main)
        s2.wait_stream(main)
        # (round 6, measured and dropped: starting student 2 late -- a spin of 0.3 / 1.4 / 4.9 ms in front of its forward, so that
        # one student's light kernels would meet the other's GEMMs -- only adds the delay: 51.0 / 52.3 / 55.8 vs 50.9 ms per step,
        # profiles/r06_stream_offset.txt.  The students are not in lock-step to begin with: the host issues student 1's whole
        # forward (~4 ms of host time) before student 2's first launch.  Removing THAT skew does not pay either: the two launch
        # sequences issued from two host threads 51.96 / 52.07 vs 51.90 / 51.88 ms (2 img/GPU: 29.06 / 29.36 vs 28.99 / 28.97);
        # and a third / fourth stream -- each student's no-grad pass of the other scales next to its saved scale-1.0 pass --
        # 53.65 / 53.61 ms (2 img/GPU: 29.01 / 28.97).  profiles/r06_issue_order.txt: the step is bound by what its MFMA-heavy
        # kernels cost in total, not by how they are interleaved.)
        with torch.cuda.stream(s1):
            r1 = fn1()
        with torch.cuda.stream(s2):
            r2 = fn2()
        main.wait_stream(s1)
        main.wait_stream(s2)

        def rec(r):
            if torch.is_tensor(r):
                r.record_stream(main)
            elif isinstance(r, (tuple, list)):
                for t in r:
                    rec(t)

        rec(r1)
        rec(r2)
        return r1, r2

    def get_param_groups(self):
        """model_dupl.py:119-154: [backbone, backbone_norm, cls heads, decoders] over both students."""
        groups = [[], [], [], []]
        for br in (self.branch1, self.branch2):
            for name, p in br.encoder.named_parameters():
                groups[1 if "norm" in name else 0].append(p)
        for br in (self.branch1, self.branch2):
            groups[2] += [br.classifier.weight, br.aux_classifier.weight]
        for br in (self.branch1, self.branch2):
            groups[3] += list(br.decoder.parameters())
        return groups

    def forward(self, x, val=False, cam_only=False, cam_with_grad=False, branch=None, need_sp=False):
        """model_dupl.py:156-213 dispatch."""
        res = {}
        if val:
            if branch is None:
                res["branch1"] = self.branch1(x)
                res["branch2"] = self.branch2(x)
                return res
            return self.branch1(x) if branch == 1 else self.branch2(x)
        if cam_only:
            if branch is None:
                cam_aux_1, cam_1 = self.branch1(x, cam_only=cam_only)
                cam_aux_2, cam_2 = self.branch2(x, cam_only=cam_only)
                return cam_aux_1, cam_1, cam_aux_2, cam_2
            return self.branch1(x, cam_only=cam_only) if branch == 1 else self.branch2(x, cam_only=cam_only)
        if cam_with_grad:
            if branch is None:
                res["branch1"], res["branch2"] = self.per_student(lambda: self.branch1(x, cam_with_grad=True),
                                                                  lambda: self.branch2(x, cam_with_grad=True))
                return res
            return self.branch1(x, cam_with_grad=True) if branch == 1 else self.branch2(x, cam_with_grad=True)
        if branch is None:
            if need_sp:
                x, x_aug = x.chunk(2)
                b, _, H, W = x_aug.shape
                x_aug = ops.resize_bilinear(x_aug.contiguous(), int(H * 0.75), int(W * 0.75))
                # the reference first runs both students on the whole 2b batch and discards the result
                # (model_dupl.py:190-191): that dead forward is not computed here
                (res["branch1"], res["branch1_aug"]), (res["branch2"], res["branch2_aug"]) = self.per_student(
                    lambda: (self.branch1(x), self.branch1(x_aug)[1]), lambda: (self.branch2(x), self.branch2(x_aug)[1]))
                return res
            res["branch1"], res["branch2"] = self.per_student(lambda: self.branch1(x), lambda: self.branch2(x))
            return res
        return self.branch1(x) if branch == 1 else self.branch2(x)
