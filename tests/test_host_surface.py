"""Host-side drop-in surface: Configer / CLI override scheme (reference configer.py:47-145, main_contrastive.py:31-164),
checkpoint interchange with the reference (module_runner.py:78-119, 168-226), optimizer / lr policy
(optim_scheduler.py:46-98, trainer_contrastive.py:163-175). CPU only."""
import os

import pytest
import torch

from oracle import ref_shim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cli_overrides_and_free_form_pairs():
    from contrastiveseg_amd.lib.utils.tools.configer import Configer
    from contrastiveseg_amd.main_contrastive import build_parser
    args = build_parser().parse_args(["--configs", os.path.join(ROOT, "configs/cityscapes/H_48_D_4.json"), "--phase", "train",
                                      "--model_name", "hrnet_w48_mem", "--loss_type", "mem_contrast_ce_loss",
                                      "--max_iters", "123", "contrast.temperature", "0.07", "contrast.with_memory",
                                      "True", "network.new_key", "[1, 2]", "data.tag", "plain-string"])
    cfg = Configer(args_parser=args)
    assert cfg.get("network", "model_name") == "hrnet_w48_mem"            # 'section:key' dest overrides the JSON
    assert cfg.get("loss", "loss_type") == "mem_contrast_ce_loss"
    assert cfg.get("solver", "max_iters") == 123
    assert cfg.get("lr", "base_lr") == 0.01                                # None-valued flags keep the JSON value
    assert cfg.get("contrast", "temperature") == 0.07                      # literal_eval'ed free-form pairs
    assert cfg.get("contrast", "with_memory") is True
    assert cfg.get("network", "new_key") == [1, 2]
    assert cfg.get("data", "tag") == "plain-string"
    assert cfg.get("phase") == "train" and cfg.exists("contrast", "loss_weight")
    cfg.add(["iters"], 0)
    cfg.plus_one("iters")
    assert cfg.get("iters") == 1


def _trainer(tmp=None, **over):
    from contrastiveseg_amd.lib.utils.tools.configer import Configer
    from contrastiveseg_amd.segmentor.trainer_contrastive import Trainer
    cfg = Configer(configs=os.path.join(ROOT, "configs", "synthetic", "R_18_D_8_tiny.json"))
    cfg.add(["network", "pretrained"], None)
    cfg.add(["network", "resume"], over.get("resume"))
    cfg.add(["gpu"], None)
    cfg.update(["lr", "nbb_mult"], 10.0)
    if tmp is not None:
        cfg.update(["checkpoints", "checkpoints_dir"], str(tmp))
        cfg.get("checkpoints")["checkpoints_root"] = None
        cfg.add(["project_dir"], str(tmp))
    return Trainer(cfg, train_loader=[]), cfg


def test_param_groups_and_poly_schedule(monkeypatch):
    from oracle import cpu_port
    cpu_port.install(monkeypatch)
    tr, cfg = _trainer()
    g = tr.optimizer.param_groups
    assert len(g) == 2 and g[0]["lr"] == 0.01 and abs(g[1]["lr"] - 0.1) < 1e-12     # backbone / head * nbb_mult
    n_bb = sum(1 for k, _ in tr.seg_net.named_parameters() if "backbone" in k)
    assert len(g[0]["params"]) == n_bb and len(g[0]["params"]) + len(g[1]["params"]) == len(list(tr.seg_net.parameters()))
    assert g[0]["momentum"] == 0.9 and g[0]["weight_decay"] == 0.0005
    tr.scheduler.step(50)
    assert abs(tr.optimizer.param_groups[0]["lr"] - 0.01 * (1 - 50 / 100) ** 0.9) < 1e-12


def test_checkpoint_roundtrip(tmp_path, monkeypatch):
    from oracle import cpu_port
    cpu_port.install(monkeypatch)
    tr, cfg = _trainer(tmp_path)
    cfg.update(["performance"], 0.5)
    tr.module_runner.save_net(tr.seg_net, save_mode="performance")
    name = cfg.get("checkpoints", "checkpoints_name")
    paths = [os.path.join(dp, f) for dp, _, fs in os.walk(str(tmp_path)) for f in fs]
    assert any(p.endswith("_latest.pth") for p in paths) and any(p.endswith("_max_performance.pth") for p in paths)
    ck = torch.load([p for p in paths if p.endswith("_latest.pth")][0], map_location="cpu", weights_only=False)
    assert set(ck) == {"config_dict", "state_dict"}                        # the reference's checkpoint format
    tr2, _ = _trainer(resume=[p for p in paths if p.endswith("_latest.pth")][0])
    for (k1, v1), (k2, v2) in zip(tr.seg_net.state_dict().items(), tr2.seg_net.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2)


@pytest.mark.skipif(not ref_shim.available(), reason="reference tree not present")
def test_reference_checkpoint_loads_unchanged(tmp_path, monkeypatch):
    """A checkpoint written by the reference model ('module.' prefix of its DDP wrapper included) resumes here."""
    from oracle import cpu_port
    cpu_port.install(monkeypatch)
    ref_shim.install()
    from lib.models.model_manager import ModelManager as RefManager
    torch.manual_seed(1)
    ref = RefManager(ref_shim.configer(num_classes=19, model_name="hrnet_w48_contrast", backbone="hrnet48")) \
        .semantic_segmentor()
    path = os.path.join(str(tmp_path), "ref.pth")
    torch.save({"config_dict": {}, "state_dict": {"module." + k: v for k, v in ref.state_dict().items()}}, path)
    from contrastiveseg_amd.lib.models.model_manager import ModelManager
    from contrastiveseg_amd.lib.utils.tools.configer import Configer
    from contrastiveseg_amd.segmentor.tools.module_runner import ModuleRunner
    cfg = Configer(config_dict={"data": {"num_classes": 19}, "gpu": None,
                                "network": {"backbone": "hrnet48", "model_name": "hrnet_w48_contrast",
                                            "bn_type": "torchsyncbn", "resume": path, "resume_strict": True,
                                            "pretrained": None},
                                "contrast": {"proj_dim": 256}})
    torch.manual_seed(2)
    net = ModuleRunner(cfg).load_net(ModelManager(cfg).semantic_segmentor())
    for k, v in ref.state_dict().items():
        assert torch.equal(net.state_dict()[k], v), k


def test_distributed_flag_respawns_one_rank_per_gpu(monkeypatch):
    """`main_contrastive.py --distributed --gpu 0 1 2` (reference lib/utils/distributed.py:27-69): not yet a rank ->
    start one rank per listed GPU through torch.distributed.run on 127.0.0.1 and exit with the launcher's status; already
    a rank (torchrun environment) -> join the process group instead of respawning."""
    import subprocess
    import types
    from contrastiveseg_amd.lib.utils import distributed as D
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(k, raising=False)
    seen = {}

    class FakeProc(object):
        returncode = 0

        def __init__(self, cmd, env=None):
            seen["cmd"], seen["env"] = cmd, env

        def wait(self):
            return 0
    monkeypatch.setattr(subprocess, "Popen", FakeProc)
    args = types.SimpleNamespace(distributed=True, gpu=[0, 1, 2], local_rank=-1)
    with pytest.raises(SystemExit) as e:
        D.handle_distributed(args, module="contrastiveseg_amd.main_contrastive",
                             argv=["--configs", "x.json", "--distributed", "--gpu", "0", "1", "2"])
    assert e.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-u", "-m", "torch.distributed.run"] and "--nproc-per-node" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "3" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("-m", 3) + 1] == "contrastiveseg_amd.main_contrastive" and cmd[-4:] == ["--gpu", "0", "1", "2"]
    assert seen["env"]["HIP_VISIBLE_DEVICES"] == "0,1,2"
    # a rank started by torchrun joins instead of respawning
    called = {}
    monkeypatch.setattr(D, "setup_process_group", lambda backend=None: called.setdefault("joined", True))
    monkeypatch.setenv("RANK", "1")
    monkeypatch.setenv("WORLD_SIZE", "3")
    monkeypatch.setenv("LOCAL_RANK", "1")
    D.handle_distributed(args, module="contrastiveseg_amd.main_contrastive", argv=[])
    assert called.get("joined")
    # without --distributed the process is pinned to the listed GPUs (reference :28-30)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    args2 = types.SimpleNamespace(distributed=False, gpu=[2], local_rank=-1)
    D.handle_distributed(args2, module="m", argv=[])
    import os
    assert os.environ.get("HIP_VISIBLE_DEVICES") == "2"
    monkeypatch.delenv("HIP_VISIBLE_DEVICES", raising=False)


def test_bench_line_is_assembled_from_measurements(monkeypatch, tmp_path):
    """bench.assemble_line: the JSON line of the contract from fake measurements, for every workload and world size, with and
    without the optional objects. The line must stay under 2 KB whatever goes into it (BENCH_r03.parsed was null
    because the line outgrew the driver's stdout tail); the per-kernel tables live in bench_detail.json."""
    import json
    import sys
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from contrastiveseg_amd import kernels as Kn
    from contrastiveseg_amd.lib.utils.tools.configer import Configer
    monkeypatch.setenv("CSEG_BENCH_ROUTE_FALLBACK", "x" * 5000)        # even a pathological note must not push the head out
    for workload, wl in bench.WORKLOADS.items():
        cfg = Configer(configs=os.path.join(root, "configs", wl["config"]))
        for world, split_on, extras in ((1, True, True), (1, True, False), (2, True, True), (4, False, False), (8, True, True)):
            args = types.SimpleNamespace(steps=10, warmup=3, scaling="strong", workload=workload, labels=None, miopen_find=0,
                                         channels_last=0)
            kernels = {"upcat_fwd": {"us": 190.0, "note": "n" * 3000}} if extras else {"error": "boom"}
            # two kernel families: 30 rows of one entry point at 6.5 ms each, 30 of another at 3 ms -- the line names the FAMILY with
            # the larger time per step, flops and time summed over its launches
            rows = [{"kernel": "conv3x3 720->720 forward @8x128x256 " + "k" * (40 * i), "entry": "conv3x3_sb_run" if i % 2 == 0 else "conv3x3_sb_wrw",
                     "calls_per_step": 1, "us_per_launch": 6500.0 if i % 2 == 0 else 3000.0, "ms_per_step": 6.5 if i % 2 == 0 else 3.0,
                     "algorithmic_flops_per_launch": 2446118092800, "achieved_TFLOPs": 376.3, "peak_TFLOPs": 833.3,
                     "frac": 0.4516 if i else 0.2} for i in range(60)] if extras else None
            split_flops = 12.7e12 / world if split_on else 0.0
            line, detail = bench.assemble_line(
                args, wl, cfg, world, wl["batch"], 1.766, 1765.0, 2.34567, split_on, Kn, "nccl" if world > 1 else None,
                {"value": 40.0, "ms_per_step": 201.8} if extras else None,
                {"value": 300.0, "global_batch": 64, "ms_per_step": 210.0, "steps": 5} if (extras and world > 1) else None,
                {"value": 0.2, "unit": "images/sec", "cores": 16, "kind": "port", "sample": "s" * 900} if extras else None, kernels,
                rows, split_flops)
            text = json.dumps(line)
            assert len(text) < 2048, (workload, world, len(text))
            back = json.loads(text)
            for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                        "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
                assert key in back, key
            for key in ("workload", "global_batch", "arithmetic"):
                assert key in back["config"], key
            for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
                assert key in back["roofline"], key
            assert back["n_gpus"] == world and abs(back["value"] - wl["batch"] * 10 / 1.766) < 1e-2
            r = back["roofline"]
            assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and 0 < r["frac"]
            # the roof of this implementation lies between the fp32 MFMA peak and the split-operand peak of the N GPUs
            lo, hi = bench.PEAK_FP32_MFMA_TFLOPS * world, bench.peak_split_tflops(Kn.SPLIT_ARITH) * world
            assert lo - 0.1 <= r["peak"] <= hi + 0.1, (r["peak"], lo, hi)
            if split_on:
                # 12.7 of the step's flops on the split pipe: roof time = split / 833 + rest / 157 (per GPU)
                total = wl["tflop"] * wl["batch"]
                roof_s = (12.7 / bench.peak_split_tflops(Kn.SPLIT_ARITH) + (total - 12.7) / bench.PEAK_FP32_MFMA_TFLOPS) / world
                assert abs(r["peak"] - total / roof_s) < 0.2 and abs(r["frac"] - roof_s / 0.1765) < 2e-3
                assert back["dtype"].startswith("fp32 in/out, f16x3") or back["dtype"].startswith("fp32 in/out, bf16x6")
            else:
                assert back["dtype"] == "fp32" and abs(r["peak"] - lo) < 0.1
            assert ("dominant_kernel" in r) == (extras and split_on)
            if extras and split_on:
                assert set(r["dominant_kernel"]) >= {"name", "frac", "us"} and r["dominant_kernel"]["us"] == 6500.0
                assert r["dominant_kernel"]["calls_per_step"] == 30 and abs(r["dominant_kernel"]["ms_per_step"] - 195.0) < 1e-6
                assert abs(r["dominant_kernel"]["frac"] - 2446118092800 / 6.5e-3 * 1e-12 / 833.3) < 2e-3
                fam = detail["split_kernel_families"]
                assert [f["entry"] for f in fam] == ["conv3x3_sb_run", "conv3x3_sb_wrw"] and len(fam[0]["members"]) == 30
                assert detail["roofline"]["dominant_kernel"]["worst_member"]["frac"] == 0.2
                assert "traffic_source" in r or r["traffic"] is None
                assert back["cpu_baseline"]["kind"] == "port" and len(back["cpu_baseline"]["sample"]) <= 160
            # nothing is lost: the tables are in the detail object, which is what bench.py writes next to itself
            assert detail["kernels"] == kernels and detail["split_kernels"] == rows
            assert "conv3x3_arithmetic" in detail["config"]
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    os.makedirs(os.path.join(str(tmp_path), "gpurun_out"))
    bench.write_detail(detail)
    for path in (os.path.join(str(tmp_path), bench.DETAIL_NAME), os.path.join(str(tmp_path), "gpurun_out", bench.DETAIL_NAME)):
        assert json.load(open(path))["line"]["metric"] == line["metric"]


def test_bench_counts_split_operand_flops_from_a_tally():
    """bench.split_flops_of: fp32-equivalent flops of the split-operand launches of one step (what the blended roof is built from)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    counts = {("conv3x3_sb_run", (8, 720, 128, 256), (720, 720, 3, 3), False): 1,
              ("conv3x3_sb_run", (8, 720, 128, 256), (720, 720, 3, 3), True): 1,
              ("conv3x3_sb_wrw", (8, 48, 128, 256), (8, 48, 128, 256), False): 2,
              ("conv1x1_sb_run", (8, 720, 128, 256), (256, 720, 1, 1), False): 1,
              ("conv1x1_sb_wrw", (8, 720, 128, 256), (8, 256, 128, 256), False): 1,
              ("conv3x3_s2_run", (8, 48, 128, 256), (96, 48, 3, 3), False): 3,
              ("conv3x3_s2_bwd_run", (8, 96, 64, 128), (96, 48, 3, 3), False): 1,
              ("conv3x3_s2_wrw", (8, 48, 128, 256), (8, 96, 64, 128), False): 1}
    px = 8 * 128 * 256
    want = (2 * 2.0 * px * 720 * 720 * 9 + 2 * 2.0 * px * 48 * 48 * 9 + 2 * 2.0 * px * 720 * 256
            + 5 * 2.0 * (px // 4) * 48 * 96 * 9)
    assert abs(bench.split_flops_of(counts) - want) <= 1e-6 * want


def test_bench_guard_repeats_once_with_the_hardware_measured_routes(monkeypatch, capfd):
    """bench.run_guarded: a first attempt that dies (or exits 3 on a non-finite loss) is repeated once with SAFE_ROUTES and
    the reason in CSEG_BENCH_ROUTE_FALLBACK; a clean first attempt is not repeated; explicit route switches or
    CSEG_BENCH_GUARD=0 disable the guard."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    for k in list(bench.SAFE_ROUTES) + ["CSEG_BENCH_GUARD", "CSEG_BENCH_ROUTE_FALLBACK", "CSEG_BENCH_GUARDED"]:
        monkeypatch.delenv(k, raising=False)
    assert bench.guard_enabled()
    show = "import os; print(os.environ.get('CSEG_BENCH_GUARDED'), os.environ.get('CSEG_BRANCH_STREAMS'), " \
           "os.environ.get('CSEG_CONV_STATS'), os.environ.get('CSEG_BENCH_ROUTE_FALLBACK'))"
    attempts = []

    def cmd(attempt, first_rc):
        attempts.append(attempt)
        return [sys.executable, "-c", ("import sys; sys.exit(%d)" % first_rc) if (attempt == 0 and first_rc) else show]
    assert bench.run_guarded(lambda a: cmd(a, 3)) == 0
    out = capfd.readouterr()
    assert attempts == [0, 1] and "1 0 0 the first attempt" in out.out and "exit code 3" in out.err
    attempts.clear()
    assert bench.run_guarded(lambda a: cmd(a, 0)) == 0
    assert attempts == [0] and "1 None None None" in capfd.readouterr().out
    # a run that printed its result line and then died in teardown is not repeated
    attempts.clear()
    late = "import sys; print('{\"metric\": \"m\", \"value\": 1}'); sys.stdout.flush(); sys.exit(139)"
    assert bench.run_guarded(lambda a: (attempts.append(a), [sys.executable, "-c", late])[1]) == 0
    got = capfd.readouterr()
    assert attempts == [0] and got.out.count('"metric"') == 1 and "teardown" in got.err
    monkeypatch.setenv("CSEG_BRANCH_STREAMS", "1")
    assert not bench.guard_enabled()
    monkeypatch.delenv("CSEG_BRANCH_STREAMS")
    monkeypatch.setenv("CSEG_BENCH_GUARD", "0")
    assert not bench.guard_enabled()


def test_stub_kernel_host_benchmark_runs():
    """tools/host_null_bench.py (host cost of the residual-block path with stub kernels, no GPU) finishes and reports per-block times."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for extra in ([],):
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "host_null_bench.py"), "block"] + extra, capture_output=True,
                           text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-1500:]
        assert "of host time per block" in r.stdout, r.stdout[-500:]
