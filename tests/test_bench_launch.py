"""bench.py's own rank launcher (`python bench.py --gpus N` without torch.distributed.run around it — the driver's
command): a line is printed only for the world size that was asked for."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=900):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "PCC_BENCH_BACKEND", "PCC_BENCH_TILED"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, cwd=ROOT, capture_output=True, text=True, timeout=timeout)


def test_foreign_world_size_prints_no_line():
    """a launcher that set WORLD_SIZE = 1 around `--gpus 2` used to get an N = 1 line labelled by WORLD_SIZE"""
    r = _run(["--gpus", "2"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "WORLD_SIZE 1 != --gpus 2" in r.stderr


def test_too_few_gpus_for_rccl_is_a_one_line_refusal():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs present")
    r = _run(["--gpus", "2"])
    assert r.returncode != 0 and r.stdout.strip() == ""
    lines = [ln for ln in r.stderr.splitlines() if ln.strip()]
    assert len(lines) == 1 and "--gpus 2 needs 2 GPUs" in lines[0], r.stderr


@pytest.mark.gpu
def test_plain_command_starts_its_own_ranks():
    """`python bench.py --gpus 2 ...` with no launcher: two gloo ranks share the box's GPU (RCCL refuses two ranks on
    one device), the tiled workload + the all-gather run, and the line says two ranks were seen"""
    r = _run(["--gpus", "2", "--points", "60000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-psnr"],
             {"PCC_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-3000:]
    out = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(out) == 1, r.stdout
    line = json.loads(out[0])
    assert line["n_gpus"] == line["ranks_seen"] == 2
    assert len(line["ms_per_step_per_rank"]) == 2
    assert line["backend"] == "gloo" and line["scaling"] == "weak"
    assert "configs[4]" in line["config"]["workload"]
    assert line["value"] > 0


@pytest.mark.gpu
def test_rccl_on_a_one_gpu_box_is_refused():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("two GPUs present")
    r = _run(["--gpus", "2", "--points", "60000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-psnr"])
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "needs 2 GPUs" in r.stderr


@pytest.mark.gpu
def test_driver_line_carries_every_config():
    """the N = 1 line: headline fields of the contract, roofline + cpu_baseline objects, and one entry per BASELINE.json
    configuration beside the headline's (C1, C3, C4, C5 on one GPU) plus the reference's operating point (a GOP of 5
    recorded ZED frames), the affordable ones checked against the oracle inside the run"""
    r = _run(["--steps", "2", "--warmup", "1", "--points", "60000", "--cpu-runs", "1", "--inflight", "0", "--config-steps", "1"])
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["roofline"]["frac"] > 0 and line["cpu_baseline"]["kind"] == "port"
    cfg = line["configs"]
    assert set(cfg) == {"C1", "C3", "C4", "C5_one_gpu", "zed_gop5"}
    for name, c in cfg.items():
        assert c["encode_ms"] > 0 and c["decode_ms"] > 0 and c["frames_per_s"] > 0 and c["bpp"], name
    assert all(cfg[k]["equals_oracle"] is True for k in ("C1", "C3", "C4", "zed_gop5"))
    assert cfg["C3"]["blob_version"] == 2 and cfg["C5_one_gpu"]["frames"] == 8 and cfg["zed_gop5"]["frames"] == 5


@pytest.mark.gpu
def test_rccl_leg_with_a_world_of_one():
    """the tiled (N > 1) workload and its exchange over the nccl (= RCCL) backend with ONE rank on the box's one GPU
    (PCC_BENCH_TILED=1): process-group set-up with a device id, the variable-length all-gather of device tensors, the
    barriers and the max-over-ranks reduction of the N > 1 path run for real — RCCL refuses two ranks on one device, so a
    world of one is the most of that leg a one-GPU box can execute.  The line says what it was: n_gpus 1, backend nccl."""
    r = _run(["--gpus", "1", "--points", "60000", "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-psnr"],
             {"PCC_BENCH_TILED": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == line["ranks_seen"] == 1 and line["backend"].startswith("nccl")
    assert "configs[4]" in line["config"]["workload"] and line["config"]["frames_per_step_per_gpu"] == 2
    assert len(line["ms_per_step_per_rank"]) == 1 and line["value"] > 0
