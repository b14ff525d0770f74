"""hipGraph capture probe (round 4, GPU call 1): does torch.cuda.graph capture -- and replay -- forward + backward of the product's
encoder (ctypes-launched HIP kernels inside torch.autograd.Functions, MIOpen for the stem) on this ROCm build? Round 2 saw a crash
"inside the autograd engine" when micro-benchmark loops were captured; this isolates the variants, each in its own child process with
a timeout and progress markers (a crash must not take the caller down, and the last marker says where it happened).

  python tools/graph_probe.py [variant ...]     variants: one (forward+backward in ONE capture), two (forward graph + backward graph
                                                through torch.autograd.grad, the make_graphed_callables scheme), relaxed (as `two`, capture
                                                error mode 'relaxed'), fwd (forward only, no_grad); suffix _st = autograd's
                                                multithreading off (backward on the calling thread)
Prints one JSON line per variant: {"variant", "ok", "eager_ms", "replay_ms", "max_abs_dev_vs_eager", "last_marker", ...}."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def mark(s):
    sys.stderr.write("MARK %s\n" % s)
    sys.stderr.flush()


def child(variant, batch):
    import torch
    from contrastiveseg_amd import kernels as K
    from contrastiveseg_amd.lib.models.model_manager import ModelManager
    from contrastiveseg_amd.lib.utils.tools.configer import Configer
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    cfg = Configer(configs=os.path.join(ROOT, "configs", "cityscapes", "H_48_D_4.json"))
    cfg.add(["network", "pretrained"], None)
    cfg.add(["network", "resume"], None)
    torch.manual_seed(304)
    net = ModelManager(cfg).semantic_segmentor().to(dev).train()
    for m in net.modules():                      # deterministic comparison: no dropout
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    # the row-sparse hand-over of the embedding gradient needs the loss' deposit: the probe feeds dense gradients
    K.SPARSE_EMBED_GRAD = False
    g = torch.Generator().manual_seed(1)
    x = torch.randn(batch, 3, 512, 1024, generator=g).to(dev)
    g_seg = (torch.randn(batch, 19, 128, 256, generator=g) * 1e-3).to(dev)
    g_emb = (torch.randn(batch, 256, 128, 256, generator=g) * 1e-4).to(dev)
    params = [p for p in net.parameters() if p.requires_grad]
    mark("model built")

    def fwd():
        out = net(x, with_embed=True)
        return out["seg"], out["embed"]

    def eager_step():
        seg, emb = fwd()
        return seg, emb, torch.autograd.grad((seg, emb), params, (g_seg, g_emb))

    def timed(fn, n=5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    # BN running statistics move with every forward: snapshot / restore so that eager and replay see the same state
    state = {k: v.clone() for k, v in net.state_dict().items()}

    def restore():
        with torch.no_grad():
            for k, v in net.state_dict().items():
                v.copy_(state[k])

    # EVERYTHING eager runs on the side stream the captures will use, and no autograd graph of an eager iteration survives into
    # the capture: a live graph keeps the parameters' AccumulateGrad nodes -- and the stream they were created on -- alive, and
    # the engine then synchronises the capture stream with that stream (GPU call 1: rc -11 in every backward variant, with
    # exactly that warning from torch/autograd/graph.py)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    torch.cuda.set_stream(side)
    if variant.endswith("st"):
        torch.autograd.set_multithreading_enabled(False)          # backward on the calling thread
    for _ in range(2):
        eager_step()
    restore()
    seg_e, emb_e, grads_e = eager_step()
    ref = [seg_e.detach().clone(), emb_e.detach().clone()] + [t.clone() for t in (grads_e[0], grads_e[len(grads_e) // 2], grads_e[-1])]
    del seg_e, emb_e, grads_e
    eager_ms = timed(eager_step)
    mark("eager done %.1f ms" % eager_ms)

    # hipGraph capture wants no pre-zeroed max|.| records from outside the capture pool: a fresh arena is created (and zeroed by a
    # captured memset) inside the capture, so every replay starts from zero records
    def fresh_arenas():
        K._AMAX_ARENAS.clear()
        K._BN_WS.clear()
    K.SPLIT_WEIGHTS.get  # (weights are fresh: no optimizer step since the last forward -> no pack launches inside the capture)

    import gc
    gc.collect()
    torch.cuda.synchronize()
    restore()
    mark("side-stream warm-up done")
    res = {"variant": variant, "batch": batch, "eager_ms": round(eager_ms, 2)}
    mode = "relaxed" if variant.startswith("relaxed") else "global"
    if variant == "fwd":
        g1 = torch.cuda.CUDAGraph()
        fresh_arenas()
        with torch.no_grad():
            mark("capture begin (fwd)")
            with torch.cuda.graph(g1, stream=side, capture_error_mode=mode):
                seg_s, emb_s = fwd()
            mark("capture end (fwd)")
        outs = [seg_s, emb_s]

        def replay():
            g1.replay()
    elif variant.startswith("one"):
        g1 = torch.cuda.CUDAGraph()
        fresh_arenas()
        mark("capture begin (one)")
        with torch.cuda.graph(g1, stream=side, capture_error_mode=mode):
            seg_s, emb_s = fwd()
            grads_s = torch.autograd.grad((seg_s, emb_s), params, (g_seg, g_emb))
        mark("capture end (one)")
        outs = [seg_s, emb_s, grads_s[0], grads_s[len(grads_s) // 2], grads_s[-1]]

        def replay():
            g1.replay()
    else:
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        pool = torch.cuda.graph_pool_handle()
        fresh_arenas()
        mark("capture begin (two: forward)")
        with torch.cuda.graph(g1, pool=pool, stream=side, capture_error_mode=mode):
            seg_s, emb_s = fwd()
        mark("capture end (two: forward)")
        with torch.cuda.graph(g2, pool=pool, stream=side, capture_error_mode=mode):
            grads_s = torch.autograd.grad((seg_s, emb_s), params, (g_seg, g_emb))
        mark("capture end (two: backward)")
        outs = [seg_s, emb_s, grads_s[0], grads_s[len(grads_s) // 2], grads_s[-1]]

        def replay():
            g1.replay()
            g2.replay()
    torch.cuda.synchronize()
    restore()
    replay()
    torch.cuda.synchronize()
    mark("first replay done")
    devs = [float((a - b).abs().max()) / max(float(b.abs().max()), 1e-30) for a, b in zip(outs, ref)]
    restore()
    replay_ms = timed(replay)
    mark("timed replays done")
    res.update(ok=True, replay_ms=round(replay_ms, 2), max_rel_dev_vs_eager=[float("%.2e" % d) for d in devs],
               pool_gb=round(torch.cuda.memory_reserved() / 2 ** 30, 2))
    print("GRAPH_PROBE " + json.dumps(res), flush=True)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]))
        return
    variants = [a for a in sys.argv[1:] if not a.startswith("-")] or ["two", "one", "two_st", "relaxed_st"]
    batch = int(os.environ.get("PROBE_BATCH", "1"))
    for v in variants:
        t0 = time.time()
        try:
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", v, str(batch)], capture_output=True, text=True,
                               timeout=240)
            rc, out, err = p.returncode, p.stdout, p.stderr
        except subprocess.TimeoutExpired as e:
            rc, out, err = "timeout", (e.stdout or b"").decode() if isinstance(e.stdout, bytes) else (e.stdout or ""), \
                (e.stderr or b"").decode() if isinstance(e.stderr, bytes) else (e.stderr or "")
        lines = [l for l in out.splitlines() if l.startswith("GRAPH_PROBE ")]
        marks = [l[5:] for l in err.splitlines() if l.startswith("MARK ")]
        if lines:
            d = json.loads(lines[-1][len("GRAPH_PROBE "):])
        else:
            tail = [l for l in err.splitlines() if not l.startswith("MARK ")][-6:]
            d = {"variant": v, "batch": batch, "ok": False, "rc": rc, "stderr_tail": tail}
        d["last_marker"] = marks[-1] if marks else None
        d["wall_s"] = round(time.time() - t0, 1)
        print(json.dumps(d), flush=True)


if __name__ == "__main__":
    main()
