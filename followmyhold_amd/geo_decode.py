"""The ShapeVAE geometry decoder of `latent2sdf` on the MI355X matrix cores (`foho_geo_decode_fwd`, csrc/foho_geo.hip).

Reference: third_party_patches/hy3dgen/shapegen/pipelines.py:298-308 -- 35 chunks of 8000 grid points through
`vae.geo_decoder(queries, latents)` per decode of a 65^3 grid (hy3dgen's CrossAttentionDecoder: Fourier embedding,
query projection, one cross-attention block over the latent tokens, MLP, LayerNorm, one logit per point).  Here one call
decodes all points: `pipeline.latent2sdf` uses it for the reference's no-gradient decodes (the per-step clean-sample
decode PL:1614-1662, 385^3 points on the last step) and, through `foho_geo_decode_bwd`, for the decodes autograd has to
cross on its way to the latent (PL:1391-1393, 1507-1509): the gradient reaches the latent tokens (the decoder's weights
are constants of the guidance, as in the reference, where only the latent / pose parameters are optimised).

    dec = HipGeoDecoder.from_module(vae.geo_decoder)         # weights packed once (fp16 matrices, fp32 vectors)
    logits = dec(queries (1, N, 3), latents (1, L, width))    # (1, N, 1), dtype of the latents -- the module's signature;
                                                              # differentiable w.r.t. the latents when they require grad

There is no CPU path: the constructor raises when the module's shape is outside what the kernels take (head dimension
64, width % 128 == 0 and <= 1024, n_latents % 64 == 0, hidden % 128 == 0).
"""
import ctypes
import weakref

import torch

from . import _lib as L


class FohoGeoWeights(ctypes.Structure):
    _fields_ = [("width", L.c_i), ("heads", L.c_i), ("n_latents", L.c_i), ("hidden", L.c_i), ("n_freqs", L.c_i), ("flags", L.c_i),
                ("freqs", L.vp), ("w_qproj", L.vp), ("b_qproj", L.vp), ("ln_q_g", L.vp), ("ln_q_b", L.vp), ("ln_kv_g", L.vp),
                ("ln_kv_b", L.vp), ("w_q", L.vp), ("b_q", L.vp), ("w_kv", L.vp), ("b_kv", L.vp), ("w_proj", L.vp), ("b_proj", L.vp),
                ("ln_2_g", L.vp), ("ln_2_b", L.vp), ("w_fc1", L.vp), ("b_fc1", L.vp), ("w_fc2", L.vp), ("b_fc2", L.vp),
                ("ln_post_g", L.vp), ("ln_post_b", L.vp), ("w_out", L.vp), ("b_out", L.c_f), ("ln_eps", L.c_f),
                ("prior_radius", L.c_f), ("prior_sharpness", L.c_f), ("out_gain", L.c_f),
                ("w_fc2_t", L.vp), ("w_fc1_t", L.vp), ("w_proj_t", L.vp), ("zeros", L.vp), ("q_norm", L.vp), ("k_norm", L.vp),
                ("ln_q_eps", L.c_f), ("ln_kv_eps", L.c_f), ("ln_2_eps", L.c_f), ("reserved2", L.c_f)]


def _bias(lin, n, device):
    return (lin.bias.detach() if lin.bias is not None else torch.zeros(n)).to(device=device, dtype=torch.float32).contiguous()


def _parts(m):
    """The pieces of a geometry decoder module by role: the stand-in (`standins._GeoDecoder`) or hy3dgen's
    CrossAttentionDecoder (query_proj, cross_attn_decoder.{ln_1, ln_2, ln_3, attn.{c_q, c_kv, c_proj}, mlp.{c_fc, c_proj}},
    ln_post, output_proj; c_kv's output interleaves K and V per head)."""
    if hasattr(m, "block"):                                   # standins._GeoDecoder: kv rows are [K of all heads | V of all heads]
        b = m.block
        return dict(freqs=m.freqs, query_proj=m.query_proj, ln_q=b.ln_q, ln_kv=b.ln_kv, ln_2=b.ln_2, q=b.q, kv=b.kv, proj=b.proj, fc1=b.fc1,
                    fc2=b.fc2, ln_post=m.ln_post, out=m.out, heads=b.heads, kv_interleaved=False,
                    prior=(float(m.radius), float(m.sharpness), float(m.gain)))
    # what this decoder does not implement is refused, never adopted silently
    if getattr(m, "latents_proj", None) is not None and not isinstance(m.latents_proj, torch.nn.Identity):
        raise L.FohoError("HipGeoDecoder: a decoder with latents_proj (downsample_ratio != 1) is not supported")
    if getattr(m, "ln_post", None) is None or not isinstance(m.ln_post, torch.nn.LayerNorm):
        raise L.FohoError("HipGeoDecoder: a decoder without ln_post (enable_ln_post=False) is not supported")
    if m.output_proj.out_features != 1:
        raise L.FohoError(f"HipGeoDecoder: output_proj of {m.output_proj.out_features} channels (one occupancy logit is supported)")
    blk = m.cross_attn_decoder
    att = blk.attn
    qk = {}
    for name in ("q_norm", "k_norm"):      # qk_norm: LayerNorm over the head dimension (attention_blocks.py); anything else is refused
        nrm = getattr(att.attention, name, None)
        if nrm is None or isinstance(nrm, torch.nn.Identity):
            qk[name] = None
        elif isinstance(nrm, torch.nn.LayerNorm) and tuple(nrm.normalized_shape) == (64,):
            qk[name] = nrm
        else:
            raise L.FohoError(f"HipGeoDecoder: {name} of type {type(nrm).__name__} is not supported (LayerNorm over the 64 head dimensions is)")
    return dict(q_norm=qk["q_norm"], k_norm=qk["k_norm"], freqs=m.fourier_embedder.frequencies, query_proj=m.query_proj, ln_q=blk.ln_1, ln_kv=blk.ln_2, ln_2=blk.ln_3, q=att.c_q,
                kv=att.c_kv, proj=att.c_proj, fc1=blk.mlp.c_fc, fc2=blk.mlp.c_proj, ln_post=m.ln_post, out=m.output_proj,
                heads=att.attention.heads, kv_interleaved=True, prior=(0.0, 0.0, 1.0))


class HipGeoDecoder:
    CHUNK = 49152        # rows per block of the chain (14.5 KB of activations per row at width 1024: 0.7 GB of workspace, 1.7 GB for
                         # the backward's).  Measured per 65^3 decode, forward / forward + backward, ms: 16384 rows (the block that
                         # stays in the 256 MB Infinity Cache) 11.3 / 26.8, 24576 11.9 / 27.8, 32768 11.0 / 26.1, 49152 10.9 / 25.7,
                         # 65536 10.8 / 25.8 -- fewer, fuller launches beat cache residency

    def __init__(self, parts, device="cuda", chunk_rows=None):
        self.lib = L.lib()
        self.lib.foho_geo_abi_size.restype = ctypes.c_int64
        if int(self.lib.foho_geo_abi_size()) != ctypes.sizeof(FohoGeoWeights):
            raise L.FohoError(f"HipGeoDecoder: libfoho_hip.so was built with a foho_geo_weights of {int(self.lib.foho_geo_abi_size())} bytes, "
                              f"this binding's is {ctypes.sizeof(FohoGeoWeights)}: rebuild (make -C followmyhold_amd/csrc)")
        self.device = torch.device(device)
        dev, h = self.device, torch.float16
        p = parts
        width = p["q"].weight.shape[0]
        heads = int(p["heads"])
        hidden = p["fc1"].weight.shape[0]
        n_freqs = int(p["freqs"].numel())
        emb = 3 * (2 * n_freqs + 1)
        if p["query_proj"].weight.shape[1] != emb or emb > 64:
            raise L.FohoError(f"HipGeoDecoder: query projection of {p['query_proj'].weight.shape[1]} inputs, embedding of {emb} (at most 64)")
        t = {}
        wq = torch.zeros(width, 64, dtype=h, device=dev)
        wq[:, :emb] = p["query_proj"].weight.detach().to(dev, h)
        t["w_qproj"], t["b_qproj"] = wq, _bias(p["query_proj"], width, dev)
        for name, ln in (("ln_q", p["ln_q"]), ("ln_kv", p["ln_kv"]), ("ln_2", p["ln_2"]), ("ln_post", p["ln_post"])):
            t[name + "_g"] = ln.weight.detach().to(dev, torch.float32).contiguous()
            t[name + "_b"] = ln.bias.detach().to(dev, torch.float32).contiguous()
        wkv, bkv = p["kv"].weight.detach().to(dev, torch.float32), _bias(p["kv"], 2 * width, dev)
        if p["kv_interleaved"]:                          # rows [h][k | v][d] -> [k | v][h][d]
            wkv = wkv.view(heads, 2, width // heads, width).permute(1, 0, 2, 3).reshape(2 * width, width)
            bkv = bkv.view(heads, 2, width // heads).permute(1, 0, 2).reshape(2 * width)
        t["w_kv"], t["b_kv"] = wkv.to(h).contiguous(), bkv.contiguous()
        for name, lin in (("q", p["q"]), ("proj", p["proj"]), ("fc1", p["fc1"]), ("fc2", p["fc2"])):
            t["w_" + name] = lin.weight.detach().to(dev, h).contiguous()
            t["b_" + name] = _bias(lin, lin.weight.shape[0], dev)
        for name in ("fc2", "fc1", "proj"):              # the backward's GEMMs multiply by the transposes
            t[f"w_{name}_t"] = t["w_" + name].t().contiguous()
        t["zeros"] = torch.zeros(max(hidden, 2 * width), dtype=torch.float32, device=dev)
        for name in ("q_norm", "k_norm"):                 # 129 floats: gain, bias, eps
            nrm = p.get(name)
            if nrm is not None:
                g = nrm.weight.detach().float() if nrm.weight is not None else torch.ones(64)
                b = nrm.bias.detach().float() if nrm.bias is not None else torch.zeros(64)
                t[name] = torch.cat([g.reshape(-1).cpu(), b.reshape(-1).cpu(), torch.tensor([float(nrm.eps)])]).to(dev).contiguous()
                t[name + "_eps"] = float(nrm.eps)
        t["w_out"] = p["out"].weight.detach().reshape(-1).to(dev, torch.float32).contiguous()
        t["freqs"] = p["freqs"].detach().to(dev, torch.float32).contiguous()
        self.t = t
        w = FohoGeoWeights()
        w.width, w.heads, w.hidden, w.n_freqs, w.n_latents = width, heads, hidden, n_freqs, 64
        for name, _ in FohoGeoWeights._fields_:
            if name in t:
                setattr(w, name, t[name].data_ptr())
        w.b_out = float(p["out"].bias.detach().reshape(-1)[0]) if p["out"].bias is not None else 0.0
        # one eps per LayerNorm: hy3dgen builds the block's ln_1 / ln_2 / ln_3 with 1e-6 and ln_post with torch's default
        w.ln_eps, w.ln_q_eps, w.ln_kv_eps, w.ln_2_eps = float(p["ln_post"].eps), float(p["ln_q"].eps), float(p["ln_kv"].eps), float(p["ln_2"].eps)
        w.prior_radius, w.prior_sharpness, w.out_gain = p["prior"]
        self.w = w
        self.chunk = int(chunk_rows or self.CHUNK)
        self.workspace = None
        self.bwd_workspace = None
        # How a decode under autograd gets its backward:
        #   "rows"      (default) plain forward, nothing kept; the backward compacts the rows whose logit gradient is not zero on the
        #               device, recomputes the chain for them and back-propagates them only (foho_geo_decode_bwd_rows) -- exact, and
        #               the guidance loop's gradient (out of FlexiCubes) is non-zero on 5-10 % of the grid
        #   "keep"      the forward keeps the activations of ALL rows (18 KB per query), the backward is dense: the fastest route for a
        #               dense gradient
        #   "recompute" dense backward, the forward recomputed per row block (no memory)
        self.backward_mode = "rows"
        self.row_cap = None               # upper bound on the active rows a caller can vouch for (None: all rows -- cannot overflow)
        self.last_row_stats = None        # device int32[2] of the last "rows" backward: active rows, rows dropped for lack of capacity
        self.rows_dropped_total = None    # device int32[1]: rows dropped by ALL "rows" backwards since the last take_rows_dropped()
        self.query_cache_limit = 4 << 30  # bytes: grids whose cached query side (4 KB per point at width 1024) fits are cached
        self._qcache = None               # (weakref to the query tensor, its version, shape, cache buffer)
        self._prepared = None
        self._ws_epoch = 0                # bumped by everything that rewrites the workspace's latent side (prepare, set_kv, a re-allocation)
        for fn in (self.lib.foho_geo_workspace_bytes, self.lib.foho_geo_bwd_workspace_bytes):
            fn.restype = ctypes.c_size_t
            fn.argtypes = [ctypes.POINTER(FohoGeoWeights), ctypes.c_int32]
        self.lib.foho_geo_last_error.restype = ctypes.c_char_p
        if self.lib.foho_geo_workspace_bytes(ctypes.byref(w), self.chunk) == 0:
            raise L.FohoError(f"HipGeoDecoder: {self.lib.foho_geo_last_error().decode()}")

    @classmethod
    def from_module(cls, module, device="cuda", chunk_rows=None):
        return cls(_parts(module), device=device, chunk_rows=chunk_rows)

    def _check(self, status, what):
        if status != 0:
            raise L.FohoError(f"{what} failed ({status}): {self.lib.foho_geo_last_error().decode()}")

    def prepare(self, latents):
        """LayerNorm + K/V projection of the latent tokens (L, width), once per set of tokens."""
        lat = latents.reshape(-1, latents.shape[-1]).to(self.device, torch.float16).contiguous()
        if lat.shape[1] != self.w.width:
            raise L.FohoError(f"HipGeoDecoder: latent tokens of width {lat.shape[1]}, decoder of width {self.w.width}")
        self._size_for(lat.shape[0])
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self.lib.foho_geo_prepare(ctypes.byref(self.w), L.vp(lat.data_ptr()), ctypes.c_int32(self.chunk), L.vp(self.workspace.data_ptr()),
                                              ctypes.c_size_t(self.workspace.numel()), L.vp(stream)), "foho_geo_prepare")
        self._lat = lat        # kept alive until the stream has consumed it
        self._ws_epoch += 1

    def _size_for(self, n_latents):
        if n_latents != self.w.n_latents or self.workspace is None:
            self.w.n_latents = n_latents
            n = int(self.lib.foho_geo_workspace_bytes(ctypes.byref(self.w), self.chunk))
            if n == 0:
                raise L.FohoError(f"HipGeoDecoder: {self.lib.foho_geo_last_error().decode()}")
            self.workspace = torch.empty(n, dtype=torch.uint8, device=self.device)
            self.bwd_workspace = None
            self._ws_epoch += 1

    def kv_of(self, latents):
        """c_kv(ln(latents)) with torch ops, rows [K of all heads | V of all heads], fp16: the part of the decoder autograd
        differentiates itself (3072 tokens; the 274 625 query rows are foho_geo_decode_bwd's)."""
        t = self.t
        lat = latents.reshape(-1, latents.shape[-1]).to(self.device)
        x = torch.nn.functional.layer_norm(lat.float(), (self.w.width,), t["ln_kv_g"], t["ln_kv_b"], self.w.ln_kv_eps)
        kv = x.half() @ t["w_kv"].t() + t["b_kv"].half()
        if "k_norm" in t:                                 # qk_norm on the key side, differentiable like the rest of this function
            W, kn = self.w.width, t["k_norm"]
            k = torch.nn.functional.layer_norm(kv[:, :W].float().reshape(-1, self.w.heads, 64), (64,), kn[:64], kn[64:128], t["k_norm_eps"])
            kv = torch.cat([k.reshape(-1, W).half(), kv[:, W:]], dim=1)
        return kv.contiguous()

    def set_kv(self, kv):
        """Install K / V (L, 2 width) fp16 computed by the caller in place of prepare()."""
        kv = kv.detach().to(self.device, torch.float16).contiguous()
        if kv.dim() != 2 or kv.shape[1] != 2 * self.w.width:
            raise L.FohoError(f"HipGeoDecoder: kv of shape {tuple(kv.shape)}, decoder of width {self.w.width}")
        self._size_for(kv.shape[0])
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self.lib.foho_geo_set_kv(ctypes.byref(self.w), L.vp(kv.data_ptr()), ctypes.c_int32(self.chunk), L.vp(self.workspace.data_ptr()),
                                             ctypes.c_size_t(self.workspace.numel()), L.vp(stream)), "foho_geo_set_kv")
        self._prepared = None
        self._ws_epoch += 1

    @property
    def ln_fuse(self):
        """True (the product path): ln_2 runs inside fc1 and ln_post + output_proj inside fc2's epilogue; False: the forward chain with
        its LayerNorm kernels (foho_geo_weights.flags & FOHO_GEO_NO_LNFUSE) -- A/B measurements, the parity test of the folded form."""
        return not (self.w.flags & 1)

    @ln_fuse.setter
    def ln_fuse(self, on):
        self.w.flags = (self.w.flags & ~1) | (0 if on else 1)

    @property
    def keep_activations(self):
        return self.backward_mode == "keep"

    @keep_activations.setter
    def keep_activations(self, keep):       # the switch of rounds 3-4: True = "keep", False = "recompute"
        self.backward_mode = "keep" if keep else "recompute"

    def _bwd_ws(self):
        if self.bwd_workspace is None:
            n = int(self.lib.foho_geo_bwd_workspace_bytes(ctypes.byref(self.w), self.chunk))
            self.bwd_workspace = torch.empty(n, dtype=torch.uint8, device=self.device)
        return self.bwd_workspace

    def decode_keep(self, queries):
        """decode() for a forward whose backward will follow: -> (logits (N,) float32, saved), `saved` = the activations
        decode_bwd needs (18 KB per query at width 1024: 5 GB for a 65^3 grid), so that it does not recompute them."""
        q = queries.reshape(-1, 3).to(self.device, torch.float32).contiguous()
        self.lib.foho_geo_saved_bytes.restype = ctypes.c_size_t
        n = int(self.lib.foho_geo_saved_bytes(ctypes.byref(self.w), ctypes.c_int32(self.chunk), ctypes.c_int64(q.shape[0])))
        saved = torch.empty(n, dtype=torch.uint8, device=self.device)
        out = torch.empty(q.shape[0], dtype=torch.float32, device=self.device)
        bws = self._bwd_ws()
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self.lib.foho_geo_decode_fwd_keep(ctypes.byref(self.w), L.vp(q.data_ptr()), ctypes.c_int64(q.shape[0]), L.vp(out.data_ptr()),
                                                      ctypes.c_int32(self.chunk), L.vp(self.workspace.data_ptr()), ctypes.c_size_t(self.workspace.numel()),
                                                      L.vp(bws.data_ptr()), ctypes.c_size_t(bws.numel()), L.vp(saved.data_ptr()),
                                                      ctypes.c_size_t(saved.numel()), L.vp(stream)), "foho_geo_decode_fwd_keep")
        return out, saved

    def decode_bwd(self, queries, grad_logits, saved=None):
        """d sum(grad_logits . logits) / d kv -> (L, 2 width) float32, for the K / V of the last set_kv() / prepare(); with
        `saved` (decode_keep) from the kept activations, without from a recomputation of the forward per row block."""
        q = queries.reshape(-1, 3).to(self.device, torch.float32).contiguous()
        g = grad_logits.reshape(-1).to(self.device, torch.float32).contiguous()
        if g.shape[0] != q.shape[0]:
            raise L.FohoError(f"HipGeoDecoder: {q.shape[0]} queries, {g.shape[0]} logit gradients")
        bws = self._bwd_ws()
        out = torch.empty(self.w.n_latents, 2 * self.w.width, dtype=torch.float32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self.lib.foho_geo_decode_bwd(ctypes.byref(self.w), L.vp(q.data_ptr()), ctypes.c_int64(q.shape[0]), L.vp(g.data_ptr()),
                                                 L.vp(out.data_ptr()), ctypes.c_int32(self.chunk), L.vp(self.workspace.data_ptr()),
                                                 ctypes.c_size_t(self.workspace.numel()), L.vp(bws.data_ptr()), ctypes.c_size_t(bws.numel()),
                                                 L.vp(saved.data_ptr()) if saved is not None else None,
                                                 ctypes.c_size_t(saved.numel() if saved is not None else 0), L.vp(stream)), "foho_geo_decode_bwd")
        return out

    def decode_bwd_rows(self, queries, grad_logits, row_cap=None):
        """decode_bwd over the rows whose logit gradient is not zero (foho_geo_decode_bwd_rows): no kept activations, no host
        synchronisation; `last_row_stats` (device int32[2]) holds the number of active rows and of rows dropped because they
        exceeded `row_cap` (None: all rows, nothing can be dropped)."""
        q = queries.reshape(-1, 3).to(self.device, torch.float32).contiguous()
        g = grad_logits.reshape(-1).to(self.device, torch.float32).contiguous()
        if g.shape[0] != q.shape[0]:
            raise L.FohoError(f"HipGeoDecoder: {q.shape[0]} queries, {g.shape[0]} logit gradients")
        n = q.shape[0]
        cap = int(row_cap or self.row_cap or n)
        cap = n if cap <= 0 or cap > n else cap
        self.lib.foho_geo_rows_workspace_bytes.restype = ctypes.c_size_t
        need = int(self.lib.foho_geo_rows_workspace_bytes(ctypes.c_int64(n), ctypes.c_int64(cap), ctypes.c_int32(self.chunk)))
        if getattr(self, "_rows_ws", None) is None or self._rows_ws.numel() < need:
            self._rows_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        bws = self._bwd_ws()
        out = torch.empty(self.w.n_latents, 2 * self.w.width, dtype=torch.float32, device=self.device)
        stats = torch.zeros(2, dtype=torch.int32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self.lib.foho_geo_decode_bwd_rows(ctypes.byref(self.w), L.vp(q.data_ptr()), ctypes.c_int64(n), L.vp(g.data_ptr()), L.vp(out.data_ptr()),
                                                      ctypes.c_int64(cap), ctypes.c_int32(self.chunk), L.vp(self.workspace.data_ptr()),
                                                      ctypes.c_size_t(self.workspace.numel()), L.vp(bws.data_ptr()), ctypes.c_size_t(bws.numel()),
                                                      L.vp(self._rows_ws.data_ptr()), ctypes.c_size_t(self._rows_ws.numel()), L.vp(stats.data_ptr()),
                                                      L.vp(stream)), "foho_geo_decode_bwd_rows")
        self.last_row_stats = stats
        if self.rows_dropped_total is None:
            self.rows_dropped_total = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.rows_dropped_total += stats[1:2]      # (on the device: no synchronisation; every backward of a phase counts, not only the last)
        return out

    def take_rows_dropped(self):
        """Rows dropped for lack of `row_cap` by all "rows" backwards since the last call (one read-back), and reset."""
        if self.rows_dropped_total is None:
            return 0
        n = int(self.rows_dropped_total.item())
        self.rows_dropped_total.zero_()
        return n

    def prepare_queries(self, queries):
        """Cache the latent-independent half of the chain (embedding -> query_proj -> ln_1 -> c_q) for this query tensor: decodes of the
        same tensor object (same version) skip it (foho_geo_decode_fwd_cached; logits bitwise equal).  The guidance loop decodes the
        same 65^3 grid 550 times per image (PL:1125-1143)."""
        q = queries.reshape(-1, 3)
        if q.device.type != self.device.type or q.dtype != torch.float32 or not queries.is_contiguous():
            raise L.FohoError("HipGeoDecoder.prepare_queries: a contiguous float32 tensor on the decoder's device (it is cached by identity)")
        if self.workspace is None:
            self._size_for(self.w.n_latents)
        self.lib.foho_geo_query_cache_bytes.restype = ctypes.c_size_t
        need = int(self.lib.foho_geo_query_cache_bytes(ctypes.byref(self.w), ctypes.c_int64(q.shape[0])))
        buf = torch.empty(need, dtype=torch.uint8, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        self._check(self.lib.foho_geo_prepare_queries(ctypes.byref(self.w), L.vp(q.data_ptr()), ctypes.c_int64(q.shape[0]), ctypes.c_int32(self.chunk),
                                                      L.vp(self.workspace.data_ptr()), ctypes.c_size_t(self.workspace.numel()), L.vp(buf.data_ptr()),
                                                      ctypes.c_size_t(need), L.vp(stream)), "foho_geo_prepare_queries")
        # keyed by storage address + version + size, with the tensor kept alive (so the address cannot be handed to another one)
        self._qcache = (queries, queries._version, buf)
        return buf

    def _cached_queries(self, queries):
        c = self._qcache
        if (c is not None and torch.is_tensor(queries) and queries.dtype == torch.float32 and queries.data_ptr() == c[0].data_ptr()
                and queries._version == c[1] == c[0]._version and queries.numel() == c[0].numel() and queries.is_contiguous()):
            return c[2]
        return None

    def grid_queries(self, xyz):
        """The query tensor latent2sdf hands to the decoder for the grid positions `xyz` (N, 3): on the device, rounded to fp16 like
        PL:303, float32, shape (1, N, 3) -- built once per `xyz` tensor (same object, same version) and, when the cached query side fits
        `query_cache_limit`, with the latent-independent half of the chain prepared (prepare_queries)."""
        g = getattr(self, "_grid", None)
        if g is not None and g[0] is xyz and g[1] == xyz._version:
            return g[2]
        q = xyz.to(self.device).half().float().reshape(1, -1, 3).contiguous()
        self.lib.foho_geo_query_cache_bytes.restype = ctypes.c_size_t
        if int(self.lib.foho_geo_query_cache_bytes(ctypes.byref(self.w), ctypes.c_int64(q.shape[1]))) <= self.query_cache_limit:
            self.prepare_queries(q)
        else:
            self._qcache = None
        self._grid = (xyz, xyz._version, q)
        return q

    def decode(self, queries):
        """queries (N, 3) -> logits (N,) float32, against the tokens of the last prepare()."""
        cache = self._cached_queries(queries)
        q = queries.reshape(-1, 3).to(self.device, torch.float32).contiguous()
        out = torch.empty(q.shape[0], dtype=torch.float32, device=self.device)
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if cache is not None:
            self._check(self.lib.foho_geo_decode_fwd_cached(ctypes.byref(self.w), L.vp(q.data_ptr()), ctypes.c_int64(q.shape[0]), L.vp(cache.data_ptr()),
                                                            ctypes.c_size_t(cache.numel()), L.vp(out.data_ptr()), ctypes.c_int32(self.chunk),
                                                            L.vp(self.workspace.data_ptr()), ctypes.c_size_t(self.workspace.numel()), L.vp(stream)),
                        "foho_geo_decode_fwd_cached")
            return out
        self._check(self.lib.foho_geo_decode_fwd(ctypes.byref(self.w), L.vp(q.data_ptr()), ctypes.c_int64(q.shape[0]), L.vp(out.data_ptr()),
                                                 ctypes.c_int32(self.chunk), L.vp(self.workspace.data_ptr()), ctypes.c_size_t(self.workspace.numel()),
                                                 L.vp(stream)), "foho_geo_decode_fwd")
        return out

    def __call__(self, queries, latents):
        """The module's signature: queries (1, N, 3), latents (1, L, width) -> (1, N, 1) in the latents' dtype."""
        if latents.dim() == 3 and latents.shape[0] != 1:
            raise L.FohoError("HipGeoDecoder: one set of latent tokens per call")
        if torch.is_grad_enabled() and latents.requires_grad:
            return _GeoDecodeFn.apply(self.kv_of(latents), queries, self).to(latents.dtype).reshape(1, -1, 1)
        # K / V are reused only for the very same tensor OBJECT at the same version: an address is no identity (the allocator hands
        # a freed latent's block to the next one, version 0 again)
        hit = self._prepared is not None and self._prepared[0]() is latents and self._prepared[1] == latents._version
        if not hit:
            self.prepare(latents)
            self._prepared = (weakref.ref(latents), latents._version)
        return self.decode(queries).to(latents.dtype).reshape(1, -1, 1)


class _GeoDecodeFn(torch.autograd.Function):
    """logits(kv), by `dec.backward_mode`: "rows" -- forward = foho_geo_set_kv + foho_geo_decode_fwd[_cached], backward =
    foho_geo_decode_bwd_rows over the rows with a non-zero gradient; "keep" -- forward = foho_geo_decode_fwd_keep (the activations
    the backward needs stay in HBM, 5 GB per 65^3 grid), backward = foho_geo_decode_bwd; "recompute" -- plain forward, dense
    backward with the forward recomputed per row block."""

    @staticmethod
    def forward(ctx, kv, queries, dec):
        dec.set_kv(kv)
        ctx.dec = dec
        ctx.epoch = dec._ws_epoch
        ctx.mode = dec.backward_mode
        if ctx.mode == "keep":
            out, saved = dec.decode_keep(queries)
            ctx.save_for_backward(kv.detach(), queries, saved)
            return out
        if ctx.mode not in ("rows", "recompute"):
            raise L.FohoError(f"HipGeoDecoder.backward_mode {ctx.mode!r}: 'rows', 'keep' or 'recompute'")
        ctx.save_for_backward(kv.detach(), queries)
        return dec.decode(queries)

    @staticmethod
    def backward(ctx, grad):
        kv, queries, *saved = ctx.saved_tensors
        if ctx.dec._ws_epoch != ctx.epoch:    # the workspace has served another set of tokens since (else K / V^T / the folded weights are
            ctx.dec.set_kv(kv)                # still this forward's: four launches less in the guidance loop's backward)
        if ctx.mode == "rows":
            return ctx.dec.decode_bwd_rows(queries, grad).to(kv.dtype), None, None
        return ctx.dec.decode_bwd(queries, grad, saved[0] if saved else None).to(kv.dtype), None, None


def install(vae, device="cuda", chunk_rows=None):
    """Attach a HipGeoDecoder built from `vae.geo_decoder` as `vae.hip_geo`: `pipeline.latent2sdf` then decodes with it,
    with or without gradients to the latent.  Raises when the decoder's shape is outside what the kernels take."""
    vae.hip_geo = HipGeoDecoder.from_module(vae.geo_decoder, device=device, chunk_rows=chunk_rows)
    return vae.hip_geo
