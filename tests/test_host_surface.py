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


def test_bench_line_is_assembled_from_measurements(monkeypatch):
    """bench.assemble_line / conv_arith_note / dominant_kernel: the JSON line of the contract from fake measurements, for
    every workload, with and without the optional objects (a formatting slip here would cost a GPU run its only output)."""
    import json
    import sys
    import types
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from contrastiveseg_amd import kernels as Kn
    from contrastiveseg_amd.lib.utils.tools.configer import Configer
    for workload, wl in bench.WORKLOADS.items():
        cfg = Configer(configs=os.path.join(root, "configs", wl["config"]))
        for world, split_on, extras in ((1, True, True), (8, False, False)):
            args = types.SimpleNamespace(steps=10, warmup=3, scaling="strong", workload=workload, labels=None, miopen_find=0,
                                         channels_last=0)
            kernels = {"upcat_fwd": {"us": 190.0}} if extras else {"error": "boom"}
            rows = [{"kernel": "conv3x3 720->720 forward @8x128x256", "entry": "conv3x3_sb_run", "calls_per_step": 1,
                     "us_per_launch": 6500.0, "ms_per_step": 6.5, "algorithmic_flops_per_launch": 2446118092800,
                     "achieved_TFLOPs": 376.3, "peak_TFLOPs": 833.3, "frac": 0.4516}] if extras else None
            line = bench.assemble_line(args, wl, cfg, world, wl["batch"], 1.766, 1765.0, 2.34567, split_on, Kn,
                                       "nccl" if world > 1 else None, {"value": 40.0} if extras else None, None,
                                       {"value": 0.2, "cores": 16, "kind": "port"} if extras else None, kernels,
                                       rows, 12.7e12 if extras else 0.0)
            back = json.loads(json.dumps(line))
            for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                        "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
                assert key in back, key
            assert back["n_gpus"] == world and abs(back["value"] - wl["batch"] * 10 / 1.766) < 1e-2
            assert abs(back["roofline"]["frac"] - back["roofline"]["achieved"] / back["roofline"]["peak"]) < 1e-3
            assert ("dominant_kernel" in back["roofline"]) == extras == ("blended_roof" in back["roofline"])
            if extras:
                assert back["roofline"]["dominant_kernel"]["name"] == rows[0]["kernel"]
                br = back["roofline"]["blended_roof"]
                assert abs(br["frac"] - br["roof_ms_per_step"] / 176.5) < 1e-3 and br["frac"] > 0
            assert (("split-fp16" in back["config"]["conv3x3_arithmetic"]) or ("split-bf16" in back["config"]["conv3x3_arithmetic"])) == split_on


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
    show = "import os; print(os.environ.get('CSEG_BENCH_GUARDED'), os.environ.get('CSEG_CONV3X3_SB_WRW'), " \
           "os.environ.get('CSEG_CONV3X3_SB_CHANNELS'), os.environ.get('CSEG_BENCH_ROUTE_FALLBACK'))"
    attempts = []

    def cmd(attempt, first_rc):
        attempts.append(attempt)
        return [sys.executable, "-c", ("import sys; sys.exit(%d)" % first_rc) if (attempt == 0 and first_rc) else show]
    assert bench.run_guarded(lambda a: cmd(a, 3)) == 0
    out = capfd.readouterr()
    assert attempts == [0, 1] and "1 0 48,96 the first attempt" in out.out and "exit code 3" in out.err
    attempts.clear()
    assert bench.run_guarded(lambda a: cmd(a, 0)) == 0
    assert attempts == [0] and "1 None None None" in capfd.readouterr().out
    # a run that printed its result line and then died in teardown is not repeated
    attempts.clear()
    late = "import sys; print('{\"metric\": \"m\", \"value\": 1}'); sys.stdout.flush(); sys.exit(139)"
    assert bench.run_guarded(lambda a: (attempts.append(a), [sys.executable, "-c", late])[1]) == 0
    got = capfd.readouterr()
    assert attempts == [0] and got.out.count('"metric"') == 1 and "teardown" in got.err
    monkeypatch.setenv("CSEG_CONV3X3_SB_WRW", "1")
    assert not bench.guard_enabled()
    monkeypatch.delenv("CSEG_CONV3X3_SB_WRW")
    monkeypatch.setenv("CSEG_BENCH_GUARD", "0")
    assert not bench.guard_enabled()
