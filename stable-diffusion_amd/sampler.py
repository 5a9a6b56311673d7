"""DDIM + classifier-free-guidance sampling loop on the HIP path.

Follows LatentDiffusionInpainting.sample (reference LatentDiffusionInpainting.py:137-230): same timestep
selection, same schedule arithmetic and dtypes (:150-159), same per-step math (:198-212, fused into one
kernel, csrc/cfg_ddim.hip), same decode + uint8 conversion (:223-230).  Differences in HOW: everything stays
resident on the GPU (no GPUManager moves - 288 GB HBM), the UNet input (2b,9,h,w) is assembled once and only
its 4 latent channels are rewritten each step by the update kernel, the per-step noise is injected by the
caller (the reference draws it from the unseeded global RNG, :211), and the UNet step can be replayed from a
HIP graph (shapes are static across steps)."""
from functools import partial

import os
import numpy as np
import torch

from . import engine as E

SCALE_FACTOR = 0.18215


def ddim_schedule(num_ddim_timesteps, eta=1.0, num_train=1000, linear_start=1e-4, linear_end=2e-2):
    """Timesteps + coefficient tables with the reference's dtypes (fp32 alphas_cumprod, float64 prev/sigma)."""
    betas = torch.linspace(linear_start ** 0.5, linear_end ** 0.5, num_train, dtype=torch.float64) ** 2
    acp = partial(torch.tensor, dtype=torch.float32)(np.cumprod(1. - betas.numpy(), axis=0))
    step = num_train // num_ddim_timesteps
    ts = np.asarray(list(range(0, num_train, step))) + 1
    if ts[-1] >= num_train:
        raise IndexError("num_ddim_timesteps=%d selects timestep %d >= %d (the reference fails the same way, "
                         "LatentDiffusionInpainting.py:156)" % (num_ddim_timesteps, ts[-1], num_train))
    a_t = acp[ts]
    a_prev = np.asarray([acp[0]] + acp[ts[:-1]].tolist())
    sig = eta * np.sqrt((1 - a_prev) / (1 - a_t) * (1 - a_t / a_prev))
    return ts, a_t, a_prev, sig, np.sqrt(1. - a_t)


def _f32(v):
    """The value torch.full((b,1,1,1), v) would hold (fp32 rounding of a tensor / numpy / python scalar)."""
    return float(torch.full((1,), v).item()) if not torch.is_tensor(v) else float(v.float().item())


class Sampler(object):
    def __init__(self, unet, vae=None, dual_stream=None, cfg_group=None):
        """cfg_group: a two-rank torch.distributed group (shard.new_cfg_pair_group).  Its rank 0 evaluates only the
        unconditional and its rank 1 only the conditional half of every step; the halves are exchanged once per step and
        both ranks apply the same update (single-image latency mode).  Every rank of the pair must call
        sample() with the same arguments."""
        self.unet, self.vae = unet, vae
        self.cfg_group = cfg_group
        if cfg_group is not None:
            import torch.distributed as dist
            self.cfg_role = dist.get_rank(cfg_group)
        self._graph = None
        self._gkey = None
        # dual_stream: evaluate the unconditional and the conditional half of the CFG batch as two independent UNet
        # passes on two HIP streams (per-image results do not depend on the batch, so the bits are the same)
        # Default on: +2.8 % images/s at 768x768 b=1 (the latency-bound 24x24 / 12x12 levels of one half fill the gaps
        # of the other); SDMI_DUAL_STREAM=0 runs the CFG batch as one pass.
        self.dual_stream = (os.environ.get("SDMI_DUAL_STREAM", "1") == "1") if dual_stream is None else bool(dual_stream)
        self._side = None
        # the context-free head of the UNet is the same for the unconditional and the conditional half (sample() feeds
        # both the same latent / timestep / image conditioning): compute it once per step.  SDMI_SHARE_PREFIX=0: A/B.
        self.share_prefix = os.environ.get("SDMI_SHARE_PREFIX", "1") == "1"

    @staticmethod
    def weights_signature(module):
        """(storage, version) of every parameter: changes on load_state_dict / in-place updates / .to(device).  Captured
        graphs and cached tables are keyed on it, so they are rebuilt instead of replaying stale packed weights."""
        return tuple((p.data_ptr(), p._version) for p in module.parameters()) if module is not None else ()

    def _time_tables(self, steps, bc, dev):
        key = (tuple(int(s) for s in steps), bc, str(dev), self._usig)
        if getattr(self, "_tt_key", None) != key:
            t_buf = torch.empty((bc,), dtype=torch.long, device=dev)
            rows = []
            for s in steps:
                t_buf.fill_(int(s))
                rows.append(self.unet.precompute_time_embedding(t_buf).table)
            self._tt, self._tt_key = torch.stack(rows).contiguous(), key
        return self._tt

    def _emb_static(self, table, dev):
        from Unet import TimeEmb
        cur = getattr(self, "_emb_cur", None)
        if cur is None or cur.table.shape != table.shape[1:] or cur.table.device != table.device:
            self._emb_cur = cur = TimeEmb(torch.empty_like(table[0]))
        return cur

    def make_context(self, uncond, cond):
        """Step-invariant text-context K/V of the CFG batch [uncond | cond] (one object per half when dual_stream).
        The objects are owned by the sampler and refilled IN PLACE for every prompt of the same batch size: the captured
        UNet step reads their buffers, so they must not move between images."""
        key = (tuple(uncond.shape), str(uncond.device), self.dual_stream, self.cfg_group is not None)
        if self.cfg_group is not None:
            halves = ((uncond, cond)[self.cfg_role].contiguous(),)      # this rank's half only
        else:
            halves = (uncond.contiguous(), cond.contiguous()) if self.dual_stream else (torch.cat([uncond, cond]).contiguous(),)
        if getattr(self, "_cc_key", None) != key:
            self._cc = tuple(self.unet.precompute_context(h) for h in halves)
            self._cc_key = key
        else:
            for c, h in zip(self._cc, halves):
                c.update(h)
        return self._cc if (self.dual_stream and self.cfg_group is None) else self._cc[0]

    def _static_xc(self, shape, dev):
        """The UNet input buffer [x | mask | masked latent] x [uncond, cond]: static, the captured step reads it."""
        xc = getattr(self, "_xc", None)
        if xc is None or tuple(xc.shape) != tuple(shape) or xc.device != dev:
            self._xc = xc = torch.empty(shape, dtype=torch.float32, device=dev)
        return xc

    @staticmethod
    def _half(t_buf, b, which):
        if hasattr(t_buf, "table"):
            from Unet import TimeEmb
            return TimeEmb(t_buf.table[which * b:(which + 1) * b])
        return t_buf[which * b:(which + 1) * b]

    def _eps_buffer(self, xc):
        """(2b, 4, h, w) fp32 noise-prediction buffer [uncond | cond]: each CFG half's last conv writes its half."""
        shape = (xc.shape[0], 4, xc.shape[2], xc.shape[3])
        eps = getattr(self, "_eps", None)
        if eps is None or tuple(eps.shape) != shape or eps.device != xc.device:
            self._eps = eps = torch.empty(shape, dtype=torch.float32, device=xc.device)
        return eps

    def _unet_eval(self, xc, t_buf, cc, same_halves=False):
        if self.cfg_group is not None:
            # this rank's half only; the partner's arrives by exchange_cfg_halves() after the (captured) pass
            b, r = xc.shape[0] // 2, self.cfg_role
            eps = self._eps_buffer(xc)
            self.unet.forward_into(xc[r * b:(r + 1) * b], self._half(t_buf, b, r), cc, eps[r * b:(r + 1) * b])
            return eps
        if not self.dual_stream:
            return self.unet(xc, timesteps=t_buf, context=cc)
        b = xc.shape[0] // 2
        eps = self._eps_buffer(xc)
        cur = torch.cuda.current_stream()
        if self._side is None:
            # first use: one pass on the current stream fills the weight pre-pack cache (shared by both streams)
            self.unet(xc[:b], timesteps=self._half(t_buf, b, 0), context=cc[0])
            torch.cuda.synchronize()
            self._side = torch.cuda.Stream()
        side = self._side
        if self.share_prefix and same_halves:
            # both halves see the same latent, timestep and image conditioning: everything before the first
            # cross-attention (conv_in, the first ResBlock, the first self-attention) is computed once
            pre = self.unet.cfg_prefix(xc[:b], self._half(t_buf, b, 0))
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for t in (pre.a1.buf, pre.t1.buf, pre.cat[0].buf):
                    t.record_stream(side)
                self.unet.forward_from_prefix(pre.branch(self.unet), cc[1], eps_out=eps[b:])
            self.unet.forward_from_prefix(pre, cc[0], eps_out=eps[:b])
            cur.wait_stream(side)
            return eps
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            self.unet.forward_into(xc[b:], self._half(t_buf, b, 1), cc[1], eps[b:])
        self.unet.forward_into(xc[:b], self._half(t_buf, b, 0), cc[0], eps[:b])
        cur.wait_stream(side)
        return eps

    # one UNet evaluation on the static buffers, eager or replayed
    def _unet_step(self, xc, t_buf, cc, use_graph, same_halves=False):
        """same_halves: the caller guarantees xc[:b] == xc[b:] and equal timesteps (sample() builds them that way), which
        lets the two CFG halves share the context-free head of the UNet."""
        if not use_graph:
            return self._unet_eval(xc, t_buf, cc, same_halves)
        # the graph is valid while the buffers it read are alive and in place: key on the objects that own them (the
        # sampler keeps xc, the TimeEmb row and the ContextKV alive; a caller's own buffers must outlive the graph)
        key = (xc.data_ptr(), t_buf.table.data_ptr() if hasattr(t_buf, "table") else t_buf.data_ptr(), id(cc), tuple(xc.shape), same_halves,
               getattr(self, "_usig", None) or self.weights_signature(self.unet))
        if self._gkey != key:
            self._unet_eval(xc, t_buf, cc, same_halves)         # warm-up: weight pre-pack, allocator pools
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            # thread_local: only this thread's calls are checked during capture; another thread's (the RCCL watchdog of a
            # multi-GPU run polling its events) must not invalidate it
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                out = self._unet_eval(xc, t_buf, cc, same_halves)
            self._graph, self._gout, self._gkey, self._gkeep = g, out, key, (xc, t_buf, cc)
        self._graph.replay()
        return self._gout

    @torch.no_grad()
    def sample(self, num_ddim_timesteps, cond, uncond, c_concat, x_T, noises, scale, eta=1.0, temperature=1.0,
               decode=True, use_graph=False, trace=None):
        """cond/uncond (b,77,1024), c_concat (b,5,h,w), x_T (b,4,h,w), noises: sequence of (b,4,h,w) in loop order;
        all fp32 cuda tensors.  Returns (latent, uint8 images (b,H,W,3) or None)."""
        ts, a_t, a_prev, sig, s1m = ddim_schedule(num_ddim_timesteps, eta)
        total = ts.shape[0]
        self._usig = self.weights_signature(self.unet)      # once per image: keys the captured step and the cached tables
        b, _, h, w = x_T.shape
        dev = x_T.device
        xc = self._static_xc((2 * b, 4 + c_concat.shape[1], h, w), dev)
        xc[:b, :4] = x_T
        xc[b:, :4] = x_T
        xc[:b, 4:] = c_concat
        xc[b:, 4:] = c_concat
        # text-context K/V projections do not depend on the step: project once per prompt
        cc = self.make_context(uncond, cond)
        # the timestep-embedding MLPs depend on t only: one table per schedule step, computed once per schedule and kept
        # across images; a step copies its row into the static buffer the (graph-captured) UNet reads
        emb_table = self._time_tables(np.flip(ts), 2 * b, dev)
        emb_cur = self._emb_static(emb_table, dev)
        x = x_T.clone()
        for i, step in enumerate(np.flip(ts)):
            index = total - i - 1
            emb_cur.table.copy_(emb_table[i])
            eps = self._unet_step(xc, emb_cur, cc, use_graph, same_halves=True)   # xc halves are built identical above
            if self.cfg_group is not None:
                from .shard import exchange_cfg_halves
                exchange_cfg_halves(eps, self.cfg_role, self.cfg_group)
            E.cfg_ddim_step(eps, x, noises[i], scale, _f32(s1m[index]), _f32(a_t[index]), _f32(a_prev[index]),
                            _f32(sig[index]), temperature, x_out=x, xc_out=xc)
            if trace is not None:
                trace.append(x.clone())
        if not decode:
            return x, None
        return x, self.decode_to_uint8(x, use_graph=use_graph)

    @torch.no_grad()
    def prepare_c_concat(self, image, mask, noise=None):
        """Image conditioning of the inpainting UNet, on the device (reference make_batch_sd + the c_cat loop,
        LatentDiffusionInpainting.py:77-79 and :279-293): image (b,3,H,W) fp32 in [-1,1], mask (b,1,H,W) in {0,1} ->
        c_concat (b,5,H/8,W/8) = [nearest-resized mask | 0.18215 * encode(image * (mask < 0.5)).sample(noise)].
        noise (b,4,H/8,W/8) is the posterior's N(0,1) draw (drawn here when None)."""
        from .inpaint import image_conditioning
        return image_conditioning(self.vae, {"mask": mask.float(), "masked_image": image * (mask < 0.5)}, noise)

    @torch.no_grad()
    def decode_to_uint8(self, latent, use_graph=False):
        """latent (b,4,h,w) -> uint8 images (b,8h,8w,3).  use_graph: replay a captured decode (the ~250 eager launches of
        a decode leave gaps a graph does not: 9.5 -> 8.4 ms at 768x768); the latent is copied into a static buffer and
        the image out of one, so the call stays re-entrant for the caller."""
        if not use_graph:
            return self.vae.decode_image_u8(1. / SCALE_FACTOR * latent)
        key = (tuple(latent.shape), str(latent.device), self.weights_signature(self.vae))
        if getattr(self, "_dkey", None) != key:
            self._din = torch.empty_like(latent, dtype=torch.float32)
            self._din.copy_(latent)
            self.vae.decode_image_u8(1. / SCALE_FACTOR * self._din)     # warm-up: weight pre-pack, allocator pools
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                out = self.vae.decode_image_u8(1. / SCALE_FACTOR * self._din)
            self._dgraph, self._dout, self._dkey = g, out, key
        self._din.copy_(latent)
        self._dgraph.replay()
        return self._dout.clone()
