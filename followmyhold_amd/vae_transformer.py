"""The ShapeVAE transformer of `latent2sdf` on the MI355X matrix cores (`foho_vae_fwd` / `foho_vae_bwd`, csrc/foho_vae.inc).

Reference: third_party_patches/hy3dgen/shapegen/pipelines.py:295 -- `pred = vae(pred)`: hy3dgen's ShapeVAE.forward = post_kl -> Transformer
(sixteen pre-norm blocks over the 3072 latent tokens, width 1024, 16 heads of 64, qk_norm, MLP 4096), run AND back-propagated in every one
of the 550 inner iterations per image (PL:1391-1393, 1507-1509, 1600).  In torch that is 25 launches per layer and direction and half of
an inner iteration; here one library call per direction: 4 GEMMs + the attention kernels per layer, LayerNorm / GELU / residual / qk_norm /
row statistics folded into the GEMM epilogues, weight gradients never formed (the weights are constants of the guidance).

    tr = HipVaeTransformer.from_module(vae)         # weights packed once: fp16 matrices with the LayerNorm gains folded in, their transposes
    tokens = tr(vae.post_kl(latents))               # (B, L, width) -> (B, L, width); differentiable w.r.t. its input

`install(vae)` attaches it as `vae.hip_transformer`; `pipeline.latent2sdf` then routes `vae(pred)` through it.  Layouts understood: the
stand-in (`standins.StandInShapeVAE.transformer`: _Block with ln_q / ln_kv / q / kv) and hy3dgen's (`vae.transformer.resblocks[i]`:
ln_1, attn.c_qkv with q | k | v interleaved per head, attn.attention.{q_norm, k_norm}, attn.c_proj, ln_2, mlp.c_fc, mlp.c_proj).
There is no CPU path: the constructor raises `FohoError` for shapes the kernels do not take, and the caller keeps the torch module.
"""
import ctypes

import torch

from . import _lib as L


class FohoVaeLayer(ctypes.Structure):
    """include/foho_hip.h: foho_vae_layer"""
    _fields_ = [("w_qkv", L.vp), ("fold_qkv", L.vp), ("w_qkv_t", L.vp), ("w_proj", L.vp), ("b_proj", L.vp), ("w_proj_t", L.vp),
                ("w_fc1", L.vp), ("fold_fc1", L.vp), ("w_fc1_t", L.vp), ("w_fc2", L.vp), ("b_fc2", L.vp), ("w_fc2_t", L.vp),
                ("eps1", L.c_f), ("eps2", L.c_f), ("qk_norm", L.c_i), ("reserved", L.c_i)]


class FohoVaeDesc(ctypes.Structure):
    """include/foho_hip.h: foho_vae_desc"""
    _fields_ = [("width", L.c_i), ("heads", L.c_i), ("hidden", L.c_i), ("n_layers", L.c_i), ("n_tokens", L.c_i), ("batch", L.c_i),
                ("layers", ctypes.POINTER(FohoVaeLayer)), ("zeros", L.vp), ("flags", L.c_i), ("reserved", L.c_i)]


def _f32(t, dev):
    return t.detach().to(dev, torch.float32)


def _bias(lin, dev):
    return _f32(lin.bias, dev) if lin.bias is not None else torch.zeros(lin.weight.shape[0], dtype=torch.float32, device=dev)


def _norm64(nrm, name):
    """qk_norm of one side: None, or (gain, bias, eps) of a LayerNorm over the 64 head dimensions; anything else is refused."""
    if nrm is None or isinstance(nrm, torch.nn.Identity):
        return None
    if isinstance(nrm, torch.nn.LayerNorm) and tuple(nrm.normalized_shape) == (64,):
        g = nrm.weight.detach().float().reshape(-1).cpu() if nrm.weight is not None else torch.ones(64)
        b = nrm.bias.detach().float().reshape(-1).cpu() if nrm.bias is not None else torch.zeros(64)
        return g, b, float(nrm.eps)
    raise L.FohoError(f"HipVaeTransformer: {name} of type {type(nrm).__name__} is not supported (LayerNorm over the 64 head dimensions is)")


def _blocks(vae):
    """[(per-layer dict)] for either layout: ln1 = [(LayerNorm, Linear, rows it feeds)], heads, q_norm / k_norm, proj, ln2, fc1, fc2."""
    tr = getattr(vae, "transformer", None)
    if tr is None:
        raise L.FohoError("HipVaeTransformer: the module has no `transformer`")
    if hasattr(tr, "resblocks"):                                   # hy3dgen: Transformer.resblocks[i] = ResidualAttentionBlock
        out = []
        for blk in tr.resblocks:
            att = blk.attn
            heads = int(att.attention.heads)
            W = att.c_qkv.weight.shape[1]
            d = W // heads
            # c_qkv's output is viewed (tokens, heads, 3 d) and split into q | k | v per head: rows [h][q | k | v][d] -> [q | k | v][h][d]
            perm = torch.arange(3 * W).view(heads, 3, d).permute(1, 0, 2).reshape(-1)
            for name in ("drop_path",):
                dp = getattr(blk, name, None)
                if dp is not None and not isinstance(dp, torch.nn.Identity) and getattr(dp, "drop_prob", 0.0):
                    raise L.FohoError("HipVaeTransformer: a block with stochastic depth (drop_path > 0) is not supported")
            out.append(dict(qkv=[(blk.ln_1, att.c_qkv, perm)], heads=heads, q_norm=_norm64(getattr(att.attention, "q_norm", None), "q_norm"),
                            k_norm=_norm64(getattr(att.attention, "k_norm", None), "k_norm"), proj=att.c_proj, ln2=blk.ln_2, fc1=blk.mlp.c_fc,
                            fc2=blk.mlp.c_proj))
        return out
    out = []
    for blk in tr:                                                 # standins._Block: q behind ln_q, k | v (heads side by side) behind ln_kv
        out.append(dict(qkv=[(blk.ln_q, blk.q, None), (blk.ln_kv, blk.kv, None)], heads=int(blk.heads), q_norm=None, k_norm=None, proj=blk.proj,
                        ln2=blk.ln_2, fc1=blk.fc1, fc2=blk.fc2))
    return out


def _fold(ln, lin, dev, perm=None):
    """LayerNorm -> Linear as one GEMM on the un-normalised rows: (W gamma) rounded to fp16, b' = b + W beta, s = row sums of the rounded
    folded weights (a row's constant offset then cancels exactly whatever the rounding was)."""
    Wt, g, b = _f32(lin.weight, dev), _f32(ln.weight, dev), _f32(ln.bias, dev)
    bias = _bias(lin, dev) + Wt @ b
    wf = (Wt * g[None, :]).half()
    if perm is not None:
        wf, bias = wf[perm.to(dev)], bias[perm.to(dev)]
    return wf.contiguous(), bias, wf.float().sum(dim=1)


class HipVaeTransformer:
    def __init__(self, blocks, device="cuda"):
        self.lib = L.lib()
        self.lib.foho_vae_abi_size.restype = ctypes.c_int64
        mine = ctypes.sizeof(FohoVaeLayer) * 1000 + ctypes.sizeof(FohoVaeDesc)
        if int(self.lib.foho_vae_abi_size()) != mine:
            raise L.FohoError(f"HipVaeTransformer: libfoho_hip.so's foho_vae structs ({int(self.lib.foho_vae_abi_size())}) differ from this binding's ({mine}): rebuild")
        self.device = dev = torch.device(device)
        if dev.type == "cuda" and dev.index is None and torch.cuda.is_available():
            self.device = dev = torch.device("cuda", torch.cuda.current_device())
        if not blocks:
            raise L.FohoError("HipVaeTransformer: no layers")
        W = blocks[0]["proj"].weight.shape[0]
        F = blocks[0]["fc1"].weight.shape[0]
        heads = blocks[0]["heads"]
        self.width, self.hidden, self.heads = W, F, heads
        self.t = []                       # the packed tensors, kept alive
        self.layers = (FohoVaeLayer * len(blocks))()
        for i, p in enumerate(blocks):
            if p["heads"] != heads or p["proj"].weight.shape != (W, W) or p["fc1"].weight.shape != (F, W) or p["fc2"].weight.shape != (W, F):
                raise L.FohoError("HipVaeTransformer: layers of different shapes")
            eps1 = {float(ln.eps) for ln, _, _ in p["qkv"]}
            if len(eps1) != 1:
                raise L.FohoError("HipVaeTransformer: the LayerNorms in front of q and of k | v have different eps")
            ws, bs, ss = zip(*[_fold(ln, lin, dev, perm) for ln, lin, perm in p["qkv"]])
            w_qkv, b_qkv, s_qkv = torch.cat(ws), torch.cat(bs), torch.cat(ss)
            if w_qkv.shape != (3 * W, W):
                raise L.FohoError(f"HipVaeTransformer: q | k | v projection of shape {tuple(w_qkv.shape)}, expected {(3 * W, W)}")
            qn, kn = p["q_norm"], p["k_norm"]
            if (qn is None) != (kn is None):
                raise L.FohoError("HipVaeTransformer: qk_norm on one side only is not supported")
            fold = [b_qkv, s_qkv]
            if qn is not None:
                for g, b, eps in (qn, kn):
                    fold.append(torch.cat([g, b, torch.tensor([eps, 0.0, 0.0, 0.0])]).to(dev))
            w_fc1, b_fc1, s_fc1 = _fold(p["ln2"], p["fc1"], dev)
            t = dict(w_qkv=w_qkv, fold_qkv=torch.cat(fold).contiguous(), w_qkv_t=w_qkv.t().contiguous(),
                     w_proj=p["proj"].weight.detach().to(dev, torch.float16).contiguous(), b_proj=_bias(p["proj"], dev).contiguous(),
                     w_fc1=w_fc1, fold_fc1=torch.cat([b_fc1, s_fc1]).contiguous(), w_fc1_t=w_fc1.t().contiguous(),
                     w_fc2=p["fc2"].weight.detach().to(dev, torch.float16).contiguous(), b_fc2=_bias(p["fc2"], dev).contiguous())
            t["w_proj_t"], t["w_fc2_t"] = t["w_proj"].t().contiguous(), t["w_fc2"].t().contiguous()
            self.t.append(t)
            y = self.layers[i]
            for name, _ in FohoVaeLayer._fields_:
                if name in t:
                    setattr(y, name, t[name].data_ptr())
            y.eps1, y.eps2, y.qk_norm = eps1.pop(), float(p["ln2"].eps), int(qn is not None)
        self.zeros = torch.zeros(max(F, 3 * W), dtype=torch.float32, device=dev)
        self._ws = {}                     # (batch, tokens) -> workspace tensor (stream-ordered reuse)
        for fn in (self.lib.foho_vae_workspace_bytes, self.lib.foho_vae_saved_bytes):
            fn.restype = ctypes.c_size_t
            fn.argtypes = [ctypes.POINTER(FohoVaeDesc)]
        self.lib.foho_geo_last_error.restype = ctypes.c_char_p
        self.flags = 0                    # foho_vae_desc.flags (A/B switches; 0 = the product path)
        self.calls = 0                    # forwards served (pipeline diagnostics: was the HIP route taken?)
        d = self._desc(1, 128)
        if int(self.lib.foho_vae_workspace_bytes(ctypes.byref(d))) == 0:
            raise L.FohoError(f"HipVaeTransformer: {self.lib.foho_geo_last_error().decode()}")

    @classmethod
    def from_module(cls, vae, device="cuda"):
        return cls(_blocks(vae), device=device)

    def _desc(self, batch, tokens):
        d = FohoVaeDesc()
        d.width, d.heads, d.hidden, d.n_layers, d.n_tokens, d.batch = self.width, self.heads, self.hidden, len(self.layers), tokens, batch
        d.layers = ctypes.cast(self.layers, ctypes.POINTER(FohoVaeLayer))
        d.zeros = self.zeros.data_ptr()
        d.flags = self.flags
        return d

    def accepts(self, x):
        """Can this input go through the kernels?  (B, L, width) on the device, L a multiple of 128; fp16, or fp32 (a pipeline held in
        float32: the tokens are rounded to fp16 on the way in -- the kernels' storage type, and the reference's own VAE dtype, PL:522 -- and
        handed back as fp32, like HipGeoDecoder does with its latents)."""
        return (torch.is_tensor(x) and x.is_cuda and x.dtype in (torch.float16, torch.float32) and x.dim() == 3 and x.shape[2] == self.width
                and x.shape[1] >= 128 and x.shape[1] % 128 == 0 and (self.device.index is None or x.device.index == self.device.index))

    def _workspace(self, d):
        key = (d.batch, d.n_tokens)
        ws = self._ws.get(key)
        if ws is None:
            n = int(self.lib.foho_vae_workspace_bytes(ctypes.byref(d)))
            if n == 0:
                raise L.FohoError(f"HipVaeTransformer: {self.lib.foho_geo_last_error().decode()}")
            if len(self._ws) >= 2:
                self._ws.clear()
            ws = self._ws[key] = torch.empty(n, dtype=torch.uint8, device=self.device)
        return ws

    def _check(self, rc, what):
        if rc != 0:
            raise L.FohoError(f"{what} failed ({rc}): {self.lib.foho_geo_last_error().decode()}")

    def forward_raw(self, x, keep):
        """-> (out (B, L, width) fp16, saved or None)"""
        if not self.accepts(x) or x.dtype != torch.float16:
            raise L.FohoError("HipVaeTransformer: (B, L, width) fp16 tokens on the transformer's device, L a multiple of 128")
        x = x.contiguous()
        d = self._desc(x.shape[0], x.shape[1])
        ws = self._workspace(d)
        saved = None
        if keep:
            saved = torch.empty(int(self.lib.foho_vae_saved_bytes(ctypes.byref(d))), dtype=torch.uint8, device=self.device)
        out = torch.empty_like(x)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self.lib.foho_vae_fwd(ctypes.byref(d), L.vp(x.data_ptr()), L.vp(out.data_ptr()), L.vp(ws.data_ptr()), ctypes.c_size_t(ws.numel()),
                                          L.vp(saved.data_ptr()) if saved is not None else None, ctypes.c_size_t(saved.numel() if saved is not None else 0),
                                          L.vp(stream)), "foho_vae_fwd")
        self.calls += 1
        return out, saved

    def backward_raw(self, grad_out, saved, shape):
        g = grad_out.to(torch.float16).contiguous()
        d = self._desc(shape[0], shape[1])
        ws = self._workspace(d)
        gx = torch.empty(shape, dtype=torch.float16, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self.lib.foho_vae_bwd(ctypes.byref(d), L.vp(g.data_ptr()), L.vp(gx.data_ptr()), L.vp(ws.data_ptr()), ctypes.c_size_t(ws.numel()),
                                          L.vp(saved.data_ptr()), ctypes.c_size_t(saved.numel()), L.vp(stream)), "foho_vae_bwd")
        return gx

    def __call__(self, x):
        x16 = x if x.dtype == torch.float16 else x.to(torch.float16)
        if torch.is_grad_enabled() and x.requires_grad:
            return _VaeFn.apply(x16, self).to(x.dtype)
        return self.forward_raw(x16, keep=False)[0].to(x.dtype)


class _VaeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, tr):
        out, saved = tr.forward_raw(x, keep=True)
        ctx.tr, ctx.shape = tr, tuple(x.shape)
        ctx.save_for_backward(saved)
        return out

    @staticmethod
    def backward(ctx, g):
        (saved,) = ctx.saved_tensors
        return ctx.tr.backward_raw(g, saved, ctx.shape), None


def install(vae, device="cuda"):
    """Attach a HipVaeTransformer built from `vae.transformer` as `vae.hip_transformer` (`pipeline.latent2sdf` then runs `vae(pred)` as
    post_kl -> these kernels).  Raises `FohoError` when the layout or the shapes are outside what the kernels take."""
    if not hasattr(vae, "post_kl"):
        raise L.FohoError("HipVaeTransformer: the VAE has no post_kl")
    vae.hip_transformer = HipVaeTransformer.from_module(vae, device=device)
    return vae.hip_transformer
