"""Hardware checks of the DEFAULT kernel routes at the benched shapes (they sort last on purpose: the driver runs `pytest -x`
and a surprise here must not hide the result of any other test):

  * the split-operand weight gradient reached through autograd for the 48 / 96 / 192-channel branches and the 720-channel head,
  * module_helper.Conv3x3 / HeadConv3x3 at batch 8 and the benched resolutions (64 / 192 / 384 channels through the explicit
    tilings) against fp64 with MIOpen's fp32 kernel as the yardstick,
  * GATING since round 3: the reference's one-SGD-step goldens (fp64 truth) with the grid-fill thresholds lifted, so that every
    split-operand kernel of the default set is inside a reference-pinned forward + backward (at their own shapes the goldens
    stay below the thresholds and would run on MIOpen).
Non-gating, at the very end and only while the session is young: whole-step timings of the two arithmetics, one image per GPU,
a kernel trace by family and the one-rank RCCL check -- one `CSEG_ZZ {...}` line (tests/conftest.py) that lands in the driver's
log tail for the next round."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
REPORT = {}          # printed as one line by tests/conftest.py at the end of the session (lands in the driver's log tail)


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    if "dev" not in REPORT:
        try:
            p = torch.cuda.get_device_properties(0)
            REPORT["dev"] = [p.name[:24], p.multi_processor_count, int(getattr(p, "clock_rate", 0) // 1000)]
        except Exception:                           # noqa: BLE001 -- informational only
            pass
    return torch.device("cuda:0")


def _spy(monkeypatch, K, names):
    calls = []
    for name in names:
        fn = getattr(K, name)
        monkeypatch.setattr(K, name, (lambda fn, name: lambda *a, **k: (calls.append((name, a, k)), fn(*a, **k))[1])(fn, name))
    return calls


@pytest.mark.parametrize("case", [(2, 96, 96, 16, 64), (1, 720, 720, 8, 64), (1, 48, 48, 5, 64)])
def test_split_weight_gradient_through_autograd_matches_fp64(case, monkeypatch):
    from contrastiveseg_amd import kernels as K
    dev = _dev()
    assert K.CONV3X3_SB_WRW and case[1] in K.CONV3X3_SB_WRW_CHANNELS, "defaults changed: update this test"
    calls = _spy(monkeypatch, K, ["conv3x3_sb_wrw"])
    B, ci, co, H, W = case
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, ci, H, W, generator=g)
    w = torch.randn(co, ci, 3, 3, generator=g) / (3.0 * ci ** 0.5)
    b = torch.randn(co, generator=g)
    dy = torch.randn(B, co, H, W, generator=g)
    x64, w64, b64 = (t.double().requires_grad_(True) for t in (x, w, b))
    F.conv2d(x64, w64, b64, 1, 1).backward(dy.double())
    xd, wd, bd = (t.clone().to(dev).requires_grad_(True) for t in (x, w, b))
    K.conv3x3_split_bf16(xd, wd, bd).backward(dy.to(dev))
    xr, wr, br = (t.clone().to(dev).requires_grad_(True) for t in (x, w, b))
    F.conv2d(xr, wr, br, 1, 1).backward(dy.to(dev))
    assert len(calls) == 1, "the weight gradient did not take the split-bf16 route"
    for name, g64, got, fp32 in (("dx", x64.grad, xd.grad, xr.grad), ("dw", w64.grad, wd.grad, wr.grad),
                                 ("db", b64.grad, bd.grad, br.grad)):
        scale = float(g64.abs().max())
        err = float((got.cpu().double() - g64).abs().max())
        base = float((fp32.cpu().double() - g64).abs().max())
        assert err <= max(8.0 * base, 4e-6 * scale), (case, name, err, base, scale)


@pytest.mark.parametrize("channels,hw", [(48, (128, 256)), (64, (128, 256)), (96, (64, 128)), (192, (32, 64)), (384, (16, 32))])
def test_branch_convolution_routes_at_the_benched_shapes(channels, hw, monkeypatch):
    """module_helper.Conv3x3 at batch 8 and the benched resolution: the module must take the split-bf16 route (3 channel
    tiles per block at 192 channels; split weight gradient at 48 / 96) and still be the reference's nn.Conv2d."""
    from contrastiveseg_amd import kernels as K
    from contrastiveseg_amd.lib.models.tools.module_helper import Conv3x3
    dev = _dev()
    calls = _spy(monkeypatch, K, ["conv3x3_sb_run", "conv3x3_sb_wrw"])
    torch.manual_seed(channels)
    conv = Conv3x3(channels, channels).to(dev)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(8, channels, *hw, generator=g)
    dy = torch.randn(8, channels, *hw, generator=g)
    xd = x.clone().to(dev).requires_grad_(True)
    y = conv(xd)
    y.backward(dy.to(dev))
    runs = [c for c in calls if c[0] == "conv3x3_sb_run"]
    assert len(runs) == 2, "forward / backward-data did not take the split-bf16 route"
    if channels in K.CONV3X3_SB_PICK_NT_CHANNELS:
        assert all((c[1][4] if len(c[1]) > 4 else c[2].get("nt", 0)) == 3 for c in runs), "expected 3 channel tiles per block"
    assert (len([c for c in calls if c[0] == "conv3x3_sb_wrw"]) == 1) == (channels in K.CONV3X3_SB_WRW_CHANNELS and hw[1] % 32 == 0)
    # fp64 truth on the host, MIOpen's fp32 result as the yardstick (same rule as tests/test_gpu_conv3x3_sb.py)
    x64 = x.clone().double().requires_grad_(True)
    w64 = conv.weight.detach().cpu().double().requires_grad_(True)
    y64 = F.conv2d(x64, w64, None, 1, 1)
    y64.backward(dy.double())
    xr = x.clone().to(dev).requires_grad_(True)
    wr = conv.weight.detach().clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, 1, 1)
    yr.backward(dy.to(dev))
    for name, t64, got, fp32 in (("y", y64.detach(), y.detach(), yr.detach()), ("dx", x64.grad, xd.grad, xr.grad),
                                 ("dw", w64.grad, conv.weight.grad, wr.grad)):
        scale = float(t64.abs().max())
        err = float((got.cpu().double() - t64).abs().max())
        base = float((fp32.cpu().double() - t64).abs().max())
        assert err <= max(8.0 * base, 4e-6 * scale), (channels, name, err, base, scale)


def test_head_weight_gradient_route_at_the_benched_shape(monkeypatch):
    _head_case(monkeypatch, 8, 128, 256)


def _head_case(monkeypatch, B, H, W):
    """HeadConv3x3(720) at 8 x 128 x 256: the weight / bias gradient through autograd on the split-bf16 route against MIOpen's
    fp32 result (an fp64 evaluation of 2.4 TFLOP on the host is out of reach; tools/conv3x3_sb_wrw_probe.py measured 4.5e-6
    of the gradient scale between the two at this shape)."""
    from contrastiveseg_amd import kernels as K
    from contrastiveseg_amd.lib.models.tools.module_helper import HeadConv3x3
    dev = _dev()
    calls = _spy(monkeypatch, K, ["conv3x3_sb_wrw"])
    torch.manual_seed(7)
    conv = HeadConv3x3(720).to(dev)
    x = torch.randn(B, 720, H, W, device=dev)
    dy = torch.randn(B, 720, H, W, device=dev) / 64.0
    conv(x).backward(dy)
    assert len(calls) == 1
    ref_w, ref_b = torch.ops.aten.convolution_backward(dy, x, conv.weight.detach(), [720], [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                       [False, True, True])[1:]
    # dw: 262 144 products per entry on both sides; db: two different fp32 summation orders over 262 144 values (torch's
    # tree vs whatever MIOpen does): sqrt(N) * eps = 3e-5 of the scale is already legitimate there
    for name, got, ref, tol in (("dw", conv.weight.grad, ref_w, 4e-5), ("db", conv.bias.grad, ref_b, 1e-3)):
        scale = float(ref.abs().max())
        assert float((got - ref).abs().max()) <= tol * scale, (name, float((got - ref).abs().max()), scale)


# ---- non-gating information for the next round, collected in the driver's round-end pass (no builder GPU minutes) ----------

OPTIONAL_BUDGET_S = 900       # optional steps start only while the session is younger than this (driver limit: 1800 s)


def _session_age():
    import sys
    import time
    conf = sys.modules.get("conftest")
    return time.time() - getattr(conf, "SESSION_T0", time.time())


def _within_budget(name):
    if _session_age() > OPTIONAL_BUDGET_S:
        REPORT.setdefault("skipped_for_time", []).append(name)
        return False
    return True


def _attempt(name, body, mark_ok=False):
    """Runs `body`; a failure is recorded in REPORT instead of raised. -> the exception or None."""
    try:
        body()
    except Exception as e:                        # noqa: BLE001 -- anything, incl. assertion errors
        REPORT[name] = "FAIL " + repr(e)[:140]
        return e
    if mark_ok:
        REPORT[name] = "ok"


def _child(cmd, env, timeout):
    import os
    import signal
    import subprocess
    p = subprocess.Popen(cmd, env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True,
                         start_new_session=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        out, _ = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        out, _ = p.communicate()
        return None, out
    return p.returncode, out


@pytest.mark.parametrize("name,min_runs,min_wrw", [("step_hrnet48_contrast", 300, 30), ("step_hrnet48_mem", 300, 30),
                                                    ("step_hrnet48_ocr", 300, 0)])
def test_step_golden_with_the_split_kernels_engaged(name, min_runs, min_wrw, monkeypatch, golden_dir):
    """GATING. tests/test_step_golden.py at its own shapes never reaches the split-operand kernels on the GPU (the grid-fill
    thresholds keep launches of fewer than 256 blocks on MIOpen). Here the thresholds are lifted, so the reference's
    one-SGD-step goldens (fp64 truth, bounds relative to the reference's own fp32 noise) run through the default kernel set in
    the default arithmetic: 3x3 forward / backward-data on the 48 / 64 / 96 / 192 / 384-channel convolutions and the head, the
    split weight gradient where the width allows (the OCR golden's maps are narrower than 64 columns), the 1x1 kernels."""
    import numpy as np
    import os
    from contrastiveseg_amd import kernels as K
    import test_step_golden as T
    from oracle.make_golden import STEP_CASES
    _dev()
    monkeypatch.setattr(K, "CONV3X3_SB_MIN_TILES", 1)
    monkeypatch.setattr(K, "CONV1X1_SB_MIN_TILES", 1)
    calls = _spy(monkeypatch, K, ["conv3x3_sb_run", "conv3x3_sb_wrw", "conv1x1_sb_run", "conv3x3_group_run", "conv3x3_group_wrw"])
    torch.backends.cudnn.benchmark = False
    c = STEP_CASES[name]
    g = np.load(os.path.join(golden_dir, "%s.npz" % name))
    res = T._run(c, torch.device("cuda:0"))
    # convolutions on the split kernels, one-layer launches + the members of the grouped launches (round 6: the residual blocks of
    # HRNet's parallel branches run a depth at a time, kernels.BasicBlockGroup)
    n = [len([x for x in calls if x[0] == k]) for k in ("conv3x3_sb_run", "conv3x3_sb_wrw", "conv1x1_sb_run")]
    n[0] += sum(len(x[1][0]) for x in calls if x[0] == "conv3x3_group_run")
    n[1] += sum(len(x[1][0]) for x in calls if x[0] == "conv3x3_group_wrw")
    REPORT.setdefault("step_group_launches", {})[name.replace("step_", "")] = len([x for x in calls if x[0].startswith("conv3x3_group")])
    REPORT.setdefault("step_split_launches", {})[name.replace("step_", "")] = n
    assert n[0] > min_runs and n[1] >= min_wrw and n[2] > 0, n
    worst = T._compare(res, g, c, 1e-3, 1e-3, 5e-2)
    REPORT.setdefault("step_worst_x_noise", {})[name.replace("step_", "")] = round(
        max(v[0] / max(float(g["gradnoise_l2/" + k]), 1e-30) for k, v in worst.items()), 2)


def test_every_collective_through_rccl_with_one_rank():
    """tools/rccl_single_rank_check.py in a child: a process group of ONE RCCL rank on this GPU with the multi-rank code paths
    forced on (CSEG_DIST_SINGLE_RANK=1) -- the packed fp64 all-reduces of FusedSyncBatchNorm, the counts / anchor all-gathers
    of the cross-rank contrast set and DDP's bucket all-reduce run as RCCL operations and must reproduce the local paths
    (with one rank every exchange is an identity). The builder's boxes have one GPU: this is as close to RCCL as they get."""
    import json
    import sys
    _dev()
    if not _within_budget("rccl1"):
        pytest.skip("time budget")

    def body():
        rc, out = _child([sys.executable, "tools/rccl_single_rank_check.py", "--backend", "nccl"], {}, 180)
        lines = [l for l in out.splitlines() if l.startswith("{")]
        assert rc == 0 and lines, "rc=%s %s" % (rc, out[-700:])
        d = json.loads(lines[-1])
        assert d.get("ok") and d.get("backend") == "nccl"
        REPORT["rccl1"] = [d["syncbn_all_reduces"], d["cross_rank_collectives"], d["ddp"]]
    e = _attempt("rccl1", body)
    if e is not None:
        pytest.xfail(repr(e)[:300])


def test_whole_step_timings_for_the_next_round():
    """bench.py (4 timed steps, no extras) in child processes: the default, the round-2 arithmetic, the strict fp32 path and one
    image per GPU (what a rank of the 8-GPU strong-scaling run computes). Recorded in CSEG_ZZ as ms/step; never gating."""
    import json
    import sys
    _dev()
    groups = {"default": {}, "bf16x6": {"CSEG_SPLIT_ARITH": "bf16x6"}, "fp32": {"CSEG_CONV3X3_SPLIT_BF16": "0", "CSEG_CONV1X1_SPLIT_BF16": "0"},
              "b1": {}}
    ms, failed = {}, []
    for name, env in groups.items():
        def body(name=name, env=env):
            extra = ["--global-batch", "1", "--steps", "8"] if name == "b1" else []
            rc, out = _child([sys.executable, "bench.py", "--steps", "4", "--warmup", "2", "--no-kernels", "--no-cpu-baseline",
                              "--no-fp32-pass"] + extra, dict(env, CSEG_BENCH_GUARD="0"), 300)
            lines = [l for l in out.splitlines() if l.startswith("{")]
            assert rc == 0 and lines, "rc=%s %s" % (rc, out[-600:])
            d = json.loads(lines[-1])
            assert d["config"]["final_loss"] == d["config"]["final_loss"]
            ms[name] = round(d["ms_per_step"], 1)
        failed.append(_attempt("step_" + name, body) if _within_budget("step_" + name) else None)
    REPORT["step_ms"] = ms
    if any(e is not None for e in failed):
        pytest.xfail("; ".join(repr(e)[:200] for e in failed if e is not None))


_FAMILIES = [("conv3x3_group", "c3g"), ("conv3x3_wrw2_group", "c3gwrw"), ("sb_wrw_reduce_group", "c3gwrw"), ("bn_group", "bng"),   # grouped launches (round 6)
             ("conv3x3_sb_wrw_s2", "s2"), ("conv3x3_s2_", "s2"),                                   # stride-2 kernels (round 3), before "conv3x3_sb_wrw"
             ("conv3x3_sb_kernel", "c3"), ("conv3x3_sb16_kernel", "c3"), ("conv3x3_sb16p_kernel", "c3"), ("conv3x3_sb16r_kernel", "c3"),
             ("conv3x3_sb8_kernel", "c3"), ("conv3x3_sb8p_kernel", "c3"), ("conv3x3_sb_wrw", "c3wrw"), ("cls1x1_", "cls"),
             ("sb_wrw_reduce", "c3wrw"), ("pack_batch", "pack"), ("amax_batch", "amax"),
             ("pack_weights", "pack"), ("conv1x1_sb", "c1"), ("sb_wrw1_reduce", "c1"), ("amax_kernel", "amax"),
             ("conv3x3_wrw_kernel", "f32wrw"), ("wrw_reduce_kernel", "f32wrw"), ("conv3x3_kernel", "f32conv"),
             ("igemm_wrw", "mi_wrw"), ("igemm_fwd", "mi_ig"), ("igemm_bwd", "mi_ig"), ("miopenSp3AsmConv", "mi_wino"),
             ("batched_transpose", "transp"), ("SubTensorOp", "transp"), ("Cijk_", "rocblas"), ("bn_", "bn"),
             ("elementwise", "eltw"), ("multi_tensor", "eltw"), ("reduce_kernel", "eltw")]


def family_ms_per_step(trace_csv, ms_per_step, steps):
    """Kernel time per family (ms per step) over the last `steps` steps of a rocprofv3 kernel trace."""
    import csv
    rows = list(csv.DictReader(open(trace_csv)))
    t_end = max(int(r["End_Timestamp"]) for r in rows)
    win = steps * ms_per_step * 1e6
    acc = {}
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s >= t_end - win:
            fam = next((short for key, short in _FAMILIES if key in r["Kernel_Name"]), "other")
            acc[fam] = acc.get(fam, 0) + (e - s)
    return {k: round(v / 1e6 / steps, 1) for k, v in sorted(acc.items(), key=lambda kv: -kv[1])}


def test_kernel_trace_of_the_default_step(tmp_path):
    """rocprofv3 --kernel-trace around a short bench.py run of the default configuration (child process): kernel time per
    family and step in CSEG_ZZ. Never gating."""
    import glob
    import json
    import os
    import sys
    _dev()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def body():
        out_dir = str(tmp_path / "kt")
        rc, out = _child(["rocprofv3", "--kernel-trace", "-d", out_dir, "-o", "kt", "--output-format", "csv", "--",
                          sys.executable, os.path.join(root, "bench.py"), "--steps", "4", "--warmup", "3", "--no-cpu-baseline",
                          "--no-kernels", "--no-fp32-pass"], {"CSEG_BENCH_GUARD": "0", "TMPDIR": "/tmp"}, 300)
        lines = [l for l in out.splitlines() if l.startswith("{")]
        traces = glob.glob(os.path.join(out_dir, "**", "*kernel_trace.csv"), recursive=True)
        assert rc == 0 and lines and traces, "rc=%s traces=%d %s" % (rc, len(traces), out[-500:])
        ms = json.loads(lines[-1])["ms_per_step"]
        REPORT["trace_ms"] = dict(family_ms_per_step(traces[0], ms, 3), step=round(ms, 1))
    if not _within_budget("trace"):
        pytest.skip("time budget")
    e = _attempt("trace", body)
    if e is not None:
        pytest.xfail(repr(e)[:300])


