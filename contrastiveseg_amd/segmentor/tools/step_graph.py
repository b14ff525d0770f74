"""hipGraph replay of the encoder's forward and backward inside the training step (round 4; VERDICT r3 "make the step
device-bound": the eager step issues ~2 900 launches through Python / ctypes / autograd -- 85 ms of host time per step at
batch 8, 50-70 ms at one image per GPU for ~25 ms of kernel work).

What is captured: the segmentor (`HRNet_W48_CONTRAST`, `HRNet_W48_OCR_CONTRAST`, `DeepLabV3Contrast`, or the `encoder_q` of a
memory model) as TWO hipGraphs sharing one memory pool -- forward, and backward through `torch.autograd.grad` -- the scheme of
`torch.cuda.make_graphed_callables`, written out here because this path needs four things that helper cannot know about:

  * the max|.| records of the f16x3 convolutions (kernels.amax_slot) are zero-filled ONCE per arena in eager mode; a replay must
    start from zeroed records, so the capture allocates its arenas INSIDE the graph (the fill becomes a memset node);
  * the packed convolution weights (kernels.SplitWeights) are refreshed by Python on the first use after an optimizer step; a
    replay runs no Python, so the refresh is issued explicitly before every forward replay;
  * the row-sparse gradient of the embedding (kernels.SparseGradSlot): the loss deposits <= max_samples rows + pixel indices,
    a data-dependent count; the captured backward reads a STATIC [capacity, D] row buffer and index buffer, zero-padded (a zero
    row at pixel 0 contributes exactly nothing to any gradient), which the replay fills from the step's deposits;
  * BatchNorm buffers and the RNG advance during the warm-up iterations that precede a capture; the buffers are restored.

What stays eager, around its one host synchronisation: the criterion (fused upsample + CE, anchor mining, the host-side
torch.randperm draws, the contrast kernels), the memory-bank update and the optimizer.

The reference semantics are unchanged (segmentor/trainer_contrastive.py:193-267): the same kernels run in the same order on the
same values; `Trainer.train_step` still does forward -> loss -> zero_grad -> backward -> step. Validation, eval mode,
no_grad calls, other input shapes than the captured ones and multi-rank runs (SyncBN / DDP collectives between the kernels)
take the eager path: REPLAY IS SINGLE-RANK ONLY (install() returns None under a process group), so the default "auto" mode speeds up
one-image-per-GPU runs of ONE process, not the ranks of a DDP job.

Findings of the capture probe on the MI355X (tools/graph_probe.py, profiles/r04_graph_probe.txt): capture + replay of forward
and backward are bit-identical to eager; an autograd graph of an EARLIER eager iteration that is still alive during the capture
(it pins the parameters' AccumulateGrad nodes to the stream they were created on) kills the process with SIGSEGV inside the
engine -- hence the warm-up on the capture stream and the garbage collection right before the capture.

Switches: CSEG_STEP_GRAPH=0 disables; CSEG_STEP_GRAPH_STREAMS=0 keeps the parallel HRNet branches on one stream inside the
capture (default: one side stream per branch, so that the graph has parallel paths)."""
import gc
import os

import torch

from contrastiveseg_amd import kernels as K
from contrastiveseg_amd.lib.utils.distributed import is_distributed
from contrastiveseg_amd.lib.utils.tools.logger import Logger as Log

# "auto" (default): replay where the eager step is host-bound -- per-GPU batches of at most AUTO_MAX_BATCH images (the 8-GPU
# strong-scaling point of BASELINE.json's bs-8 metric is ONE image per GPU) -- and stay eager where the GPU is the limit anyway:
# measured on the MI355X (profiles/r04_step_graph_ab.txt), ms/step eager vs replay: batch 8 95.1 vs 102.3 (110.2 without the forked
# branches), batch 2 57.1 vs 56.6, batch 1 62.0 vs 48.6 -- a hipGraph launch on ROCm 7.2 keeps a per-node cost of ~13 us and loses
# the back-to-back dispatch of an in-order queue, so it pays only where the host is the limit.
MODE = os.environ.get("CSEG_STEP_GRAPH", "auto")
ENABLED = MODE != "0"
AUTO_MAX_BATCH = int(os.environ.get("CSEG_STEP_GRAPH_MAX_BATCH", "1"))
BRANCH_STREAMS = os.environ.get("CSEG_STEP_GRAPH_STREAMS", "1") == "1"
MAX_SHAPES = 2                       # distinct input shapes that get their own pair of graphs

_CAPTURING = [False]                 # read by the model code (lib/models/...): "this forward / backward is being captured"


def capturing():
    return _CAPTURING[0]


def _set_state(text):
    os.environ["CSEG_STEP_GRAPH_STATE"] = text            # bench.py reports it in config.step_graph


class _Captured(object):
    """One input shape: static input, outputs, gradient buffers and the two graphs."""
    __slots__ = ("x", "out", "keys", "grad_keys", "g_out", "rows", "sel", "g_fwd", "g_bwd", "grads", "reach", "cap", "gen")


class _Replay(torch.autograd.Function):
    """forward: copy the input, replay the forward graph, hand out aliases of the static outputs. backward: stage the incoming
    gradients (dense ones by copy, the embedding's row-sparse one by filling the static row / index buffers), replay the backward
    graph, return the static parameter gradients."""

    @staticmethod
    def forward(ctx, cap, slot, x, *params):
        cap.x.copy_(x)
        K.SPLIT_WEIGHTS.refresh_all()          # the optimizer step made every packed weight stale; no Python runs inside a replay
        cap.g_fwd.replay()
        outs = tuple(cap.out[k].detach() for k in cap.keys)
        ctx.cap, ctx.slot = cap, slot
        ctx.set_materialize_grads(False)       # an unused output arrives as None, not as a zero-filled tensor
        ctx.mark_non_differentiable(*[o for k, o in zip(cap.keys, outs) if k not in cap.grad_keys])
        return outs

    @staticmethod
    def backward(ctx, *grads):
        cap = ctx.cap
        by_key = dict(zip(cap.keys, grads))
        for k in cap.grad_keys:
            g = by_key[k]
            if k == "embed" and cap.rows is not None:
                deposits = ctx.slot.take() if g is None or ctx.slot.is_standin(g) else None
                if deposits is None:
                    raise RuntimeError("step graph: the embedding received a dense gradient (a consumer other than the contrastive "
                                       "criterion); the captured backward covers the row-sparse hand-over only -- run with "
                                       "CSEG_STEP_GRAPH=0")
                cap.rows.zero_()
                cap.sel.zero_()
                off = 0
                for rows, sel in deposits:
                    n = rows.shape[0]
                    if off + n > cap.cap:
                        raise RuntimeError("step graph: %d anchor rows exceed the captured capacity %d" % (off + n, cap.cap))
                    cap.rows[off:off + n].copy_(rows)
                    cap.sel[off:off + n].copy_(sel)
                    off += n
            elif g is None:
                cap.g_out[k].zero_()
            else:
                cap.g_out[k].copy_(g)
        cap.g_bwd.replay()
        live = [k for k in cap.grad_keys if by_key[k] is not None]
        if len(live) == len(cap.grad_keys) or cap.reach is None:
            return (None, None, None) + tuple(None if g is None else g.detach() for g in cap.grads)
        # an output the criterion does not use (DeepLab's `seg_aux` under contrast_ce_loss): the eager step leaves the parameters
        # that feed only that output WITHOUT a gradient (the optimizer skips them: no weight decay, no momentum); a zero-filled
        # gradient from the replay would not be the same update
        keep = [any(cap.reach[k][i] for k in live) for i in range(len(cap.grads))]
        return (None, None, None) + tuple(g.detach() if (g is not None and keep[i]) else None for i, g in enumerate(cap.grads))


class GraphedEncoder(object):
    """Installs itself as `module.forward`; the original forward stays reachable for every case the graphs do not cover."""

    def __init__(self, module, max_rows):
        self.module = module
        self.eager_forward = module.forward
        self.max_rows = int(max_rows)
        self.captured = {}                # input shape -> _Captured
        self.failed = None
        self.params = None
        module.forward = self.forward
        module._cseg_step_graph = self

    # -- routing ---------------------------------------------------------------------------------------
    def _usable(self, x, is_eval):
        return (ENABLED and self.failed is None and not is_eval and self.module.training and torch.is_grad_enabled()
                and (MODE != "auto" or x.shape[0] <= AUTO_MAX_BATCH)
                and x.is_cuda and not x.requires_grad and not is_distributed() and x.dtype == torch.float32
                and not torch.cuda.is_current_stream_capturing())

    def forward(self, x_, with_embed=False, is_eval=False, **kw):
        if kw or not self._usable(x_, is_eval):
            return self.eager_forward(x_, with_embed=with_embed, is_eval=is_eval, **kw)
        key = (tuple(x_.shape), x_.device.index)
        cap = self.captured.get(key)
        if cap is not None and cap.gen != K.SPLIT_WEIGHTS.generation:
            # the max|w| records moved (another model registered its weights and the arena grew): the graphs hold the old addresses
            del self.captured[key]
            cap = None
        if cap is None:
            if len(self.captured) >= MAX_SHAPES:
                return self.eager_forward(x_, with_embed=with_embed, is_eval=is_eval)
            try:
                cap = self._capture(x_)
            except Exception as e:               # a failed capture must not cost the run: eager from here on
                # (_capture has already put the BatchNorm buffers / the RNG back and dropped the capture's max|.| arenas)
                self.failed = repr(e)
                _CAPTURING[0] = False
                _set_state("eager (capture failed: %s)" % self.failed[:120])
                Log.warn("step graph: capture failed, continuing eagerly: %s" % self.failed)
                if not torch.cuda.is_current_stream_capturing():      # a synchronize inside a live capture raises by itself
                    torch.cuda.synchronize()
                return self.eager_forward(x_, with_embed=with_embed, is_eval=is_eval)
            self.captured[key] = cap
        slot = K.SparseGradSlot() if cap.rows is not None else None
        out = dict(zip(cap.keys, _Replay.apply(cap, slot, x_, *self.params)))
        if slot is not None:
            out["embed"]._cseg_grad_slot = slot          # the criterion deposits its anchor rows here (lib/loss/loss_contrast.py)
        return out

    # -- capture ---------------------------------------------------------------------------------------
    def _capture(self, x):
        mod = self.module
        dev = x.device
        self.params = tuple(p for p in mod.parameters() if p.requires_grad)
        buffers = [(b, b.detach().clone()) for b in mod.buffers()]
        rng = torch.cuda.get_rng_state(dev)

        def restore():
            with torch.no_grad():
                for b, saved in buffers:
                    b.copy_(saved)
            torch.cuda.set_rng_state(rng, dev)

        cap = _Captured()
        cap.x = x.detach().clone()              # (a fresh tensor object: no max|.| record of an earlier forward hangs on it)
        cap.cap = self.max_rows
        main = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        done = False
        torch.cuda.synchronize(dev)
        gc.collect()                       # no autograd graph of an earlier iteration may survive into the capture (module docstring)
        side.wait_stream(main)
        _CAPTURING[0] = True
        try:
            with torch.cuda.stream(side):
                # warm-up on the capture stream: per-stream scratch, MIOpen solver selection, registration of the weight packs
                cap.reach = None
                for it in range(2):
                    out = self.eager_forward(cap.x, with_embed=True)
                    keys = [k for k, v in out.items() if torch.is_tensor(v)]
                    gk = [k for k in keys if out[k].requires_grad]
                    gout = [self._standin_grad(out, k) for k in gk]
                    if it == 0 and len(gk) > 1:
                        # which parameters each output reaches (one backward per output; together they run every backward kernel)
                        cap.reach = {}
                        for k, g in zip(gk, gout):
                            got = torch.autograd.grad([out[k]], self.params, [g], allow_unused=True, retain_graph=True)
                            cap.reach[k] = [t is not None for t in got]
                            del got
                    else:
                        torch.autograd.grad([out[k] for k in gk], self.params, gout, allow_unused=True)
                    del out, gout
                torch.cuda.synchronize(dev)
                restore()
                gc.collect()
                if hasattr(cap.x, "_cseg_amax"):
                    del cap.x._cseg_amax   # a record computed by the warm-up must not be baked into the graph: the replay recomputes it
                K._AMAX_ARENAS.clear()     # the capture allocates (and zero-fills, as a graph node) its own max|.| arenas
                pool = torch.cuda.graph_pool_handle()
                cap.g_fwd, cap.g_bwd = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(cap.g_fwd, pool=pool, stream=side):
                    out = self.eager_forward(cap.x, with_embed=True)
                cap.keys = [k for k, v in out.items() if torch.is_tensor(v)]
                cap.out = {k: out[k] for k in cap.keys}
                cap.grad_keys = [k for k in cap.keys if out[k].requires_grad]
                cap.rows = cap.sel = None
                cap.g_out = {}
                gout = []
                for k in cap.grad_keys:
                    slot = getattr(out[k], "_cseg_grad_slot", None) if k == "embed" else None
                    if slot is not None:
                        D = out[k].shape[1]
                        cap.rows = torch.zeros(cap.cap, D, dtype=torch.float32, device=dev)
                        cap.sel = torch.zeros(cap.cap, dtype=torch.int32, device=dev)
                        gout.append(slot.deposit(cap.rows, cap.sel, out[k].shape))
                    else:
                        cap.g_out[k] = torch.zeros_like(out[k])
                        gout.append(cap.g_out[k])
                with torch.cuda.graph(cap.g_bwd, pool=pool, stream=side):
                    cap.grads = torch.autograd.grad([out[k] for k in cap.grad_keys], self.params, gout, allow_unused=True)
            done = True
        finally:
            _CAPTURING[0] = False
            K._AMAX_ARENAS.clear()         # eager code gets arenas of its own again (the captured ones are re-zeroed by every replay)
            if not done:
                # the warm-up iterations (and a capture that broke half-way) have advanced the running statistics,
                # num_batches_tracked and the RNG -- the eager fallback must start from the state the caller handed in
                self._restore_after_failure(restore, main, side, dev)
        main.wait_stream(side)
        torch.cuda.synchronize(dev)
        restore()
        cap.gen = K.SPLIT_WEIGHTS.generation
        n_p = sum(1 for g in cap.grads if g is not None)
        _set_state("replay (forward + backward hipGraphs, %d parameter gradients, %s)"
                   % (n_p, "row-sparse embedding gradient, capacity %d" % cap.cap if cap.rows is not None else "dense gradients"))
        Log.info("step graph: captured forward + backward for input %s" % (tuple(x.shape),))
        return cap

    @staticmethod
    def _restore_after_failure(restore, main, side, dev):
        """Best effort, never raises (the original exception is the one the caller reports)."""
        try:
            if torch.cuda.is_current_stream_capturing():
                return                     # still inside a capture that could not be ended: nothing can be copied here
            main.wait_stream(side)
            torch.cuda.synchronize(dev)
            restore()
        except Exception as e:             # pragma: no cover  (a device that is gone)
            Log.warn("step graph: could not restore the BatchNorm buffers / RNG after a failed capture: %r" % (e,))

    def _standin_grad(self, out, k):
        """Gradient fed to output `k` during warm-up: zeros -- through the sparse slot for the embedding, so that the warm-up runs the
        same backward route the capture will record."""
        slot = getattr(out[k], "_cseg_grad_slot", None) if k == "embed" else None
        if slot is not None:
            D = out[k].shape[1]
            return slot.deposit(torch.zeros(self.max_rows, D, dtype=torch.float32, device=out[k].device),
                                torch.zeros(self.max_rows, dtype=torch.int32, device=out[k].device), out[k].shape)
        return torch.zeros_like(out[k])


def install(seg_net, configer):
    """Called by Trainer._init_model on the (unwrapped) segmentor. Returns the GraphedEncoder or None."""
    if not ENABLED:
        _set_state("eager (CSEG_STEP_GRAPH=0)")
        return None
    if MODE == "auto":
        _set_state("eager (auto: per-GPU batch above %d, the GPU is the limit)" % AUTO_MAX_BATCH)
    if not torch.cuda.is_available() or not next(seg_net.parameters()).is_cuda:
        return None
    if is_distributed():
        _set_state("eager (multi-rank run: SyncBN / DDP collectives between the kernels)")
        return None
    encoder = getattr(seg_net, "encoder_q", seg_net)       # memory models: the queues and key/lb_key stay outside
    if hasattr(encoder, "_cseg_step_graph"):
        return encoder._cseg_step_graph
    max_rows = configer.get("contrast", "max_samples") if configer.exists("contrast", "max_samples") else 1024
    if MODE != "auto":
        _set_state("eager (not captured yet)")
    return GraphedEncoder(encoder, max_rows)
