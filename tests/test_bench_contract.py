"""bench.py's output contract on a small workload (the full BASELINE config 3 is what the driver runs)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"}


def _run(*extra):
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--sites", "24", "--images", "3000", "--maxm", "12",
                          "--warmup", "2", *extra], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, run.stdout[-2000:]                       # ONE JSON line
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_json_line_whole_sweep():
    d = _run()
    assert REQUIRED <= set(d), REQUIRED - set(d)
    assert d["metric"] == "two-site bond updates/sec" and d["unit"] == "bond updates/s"
    assert d["steps"] == 2 * (24 - 1) and d["warmup"] == 2 and d["n_gpus"] == 1       # default: one whole sweep
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["dtype"] == "f64" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 78.6
    assert r["achieved"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["traffic"] is None                                       # the PMC record belongs to the full workload only
    rk = d["roofline_kernels"]                                        # every matrix-pipe kernel timed live (toy workloads: the forward kernel only)
    assert rk and all(v["bound"] == "mfma" and v["frac"] > 0 and v["launches"] > 0 for v in rk.values())
    assert r.get("dominant_by_ms_per_step") is True and "not a BASELINE configuration" in d["config"]["workload"]
    rs = d["roofline_step"]
    assert rs["achieved"] > 0 and abs(rs["frac"] - rs["achieved"] / rs["peak"]) < 1e-12 and rs["frac"] == rs["frac_executed"]      # the step figure is the EXECUTED one
    assert rs["algorithmic_gflop_per_step"] >= rs["executed_gflop_per_step"] > 0
    assert d["value_literal_order"] > 0 and d["value_literal_order"] < 1.2 * d["value"]
    assert d["gradient_phase_ms"] > 0 and d["parity"].startswith("unpinned-oracle")
    h = d["roofline_hbm"]
    assert h["bound"] == "hbm" and h["unit"] == "GB/s" and h["launches"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]
    assert c["cpu_model"] and c["image_bond_updates_per_s"] > 0
    assert d["last_cost_per_image"] > 0


@pytest.mark.gpu
def test_bench_window_and_variants():
    d = _run("--steps", "4", "--no-cpu-baseline")
    assert d["steps"] == 4 and "cpu_baseline" not in d
    e = _run("--steps", "4", "--no-cpu-baseline", "--dtype", "f64_e32")
    assert e["dtype"] == "f64_e32"
    s = _run("--steps", "4", "--no-cpu-baseline", "--single-label", "3")
    assert s["value"] > 0


def test_bench_refuses_without_a_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a HIP device is present")
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1"], capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert run.returncode != 0 and "no CPU fallback" in (run.stderr + run.stdout)


def test_bench_gpus_n_without_a_launcher_starts_n_ranks():
    """`python bench.py --gpus 2` (no torch.distributed.run around it) launches its own two ranks: world-size-2 gloo
    rendezvous on 127.0.0.1, shard bounds, max-over-ranks reduction and ONE JSON line from rank 0 (--dry-run: the control
    plane only, there is no GPU in this test)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--images", "61"],
                         capture_output=True, text=True, cwd=ROOT, timeout=600, env=env)
    assert run.returncode == 0, (run.stderr + run.stdout)[-2000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, run.stdout[-2000:]
    d = json.loads(lines[0])
    # ... and the SECOND transport's leg of the same invocation (one child process per rank with a rendezvous of its own, here --dry-run too):
    # `bench.py --gpus N` times its window over RCCL and over the one-shot all-reduce and reports both under `collectives`
    assert d == {"dry_run": True, "n_gpus": 2, "max_over_ranks": 2.0, "images_over_ranks": 61, "shard_of_rank0": [0, 30], "allreduce": "rccl", "handles_gathered": None,
                 "second_transport": {"transport": "oneshot", "n_gpus": 2, "handles_gathered": 2}}
    # --allreduce oneshot: the IPC handles of the cross-process one-shot all-reduce travel over the same gloo control plane
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run", "--images", "61", "--allreduce", "oneshot"],
                         capture_output=True, text=True, cwd=ROOT, timeout=600, env=env)
    assert run.returncode == 0, (run.stderr + run.stdout)[-2000:]
    d = json.loads([ln for ln in run.stdout.splitlines() if ln.startswith("{")][0])
    assert d["allreduce"] == "oneshot" and d["handles_gathered"] == 2 and d["second_transport"] == {"transport": "rccl", "n_gpus": 2, "handles_gathered": None}


def test_bench_under_the_launcher_form_the_driver_uses():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    run = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run"],
                         capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert run.returncode == 0, (run.stderr + run.stdout)[-2000:]
    d = json.loads([ln for ln in run.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 2 and d["images_over_ranks"] == 60000 and d["shard_of_rank0"] == [0, 30000]


@pytest.mark.gpu
def test_bench_gpus_2_reports_both_transports_in_one_line():
    """`bench.py --gpus 2` tells the whole multi-GPU story in ONE line: `collectives` carries a block per transport (RCCL and the
    cross-process one-shot all-reduce), each with its bond updates/s, ms per all-reduce, gradient-phase and split time; per-rank image
    counts; the communicator size.  On a one-GPU box both ranks share device 0 (--share-device): RCCL refuses two ranks on one device,
    so its block carries that as an error string and `value` comes from the one-shot transport -- the second-leg machinery itself
    (child process per rank, own rendezvous, merge on rank 0) is what test_bench_gpus_n_without_a_launcher_starts_n_ranks covers."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-device", "--sites", "24", "--images", "3000", "--maxm", "12",
                          "--steps", "6", "--warmup", "2", "--plain", "--no-cpu-baseline"], capture_output=True, text=True, cwd=ROOT, timeout=900, env=env)
    assert run.returncode == 0, (run.stderr + run.stdout)[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, run.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["images_per_rank"] == [1500, 1500] and d["config"]["rccl_ranks"] == 2
    c = d["collectives"]
    assert c["value_from"] == "oneshot" and "error" in c["rccl"]
    o = c["oneshot"]
    assert o["value"] == d["value"] and o["ranks"] == 2 and o["allreduces_per_bond_update"] >= 4 and o["ms_per_allreduce"] > 0
    assert o["gradient_phase_ms"] > 0 and o["svd_ms"] > 0 and "processes" in o["mode"]
    assert d["speculative_split"]["splits_in_timed_region"] >= 0 and "roll_backs_per_sweep" in d["speculative_split"]


@pytest.mark.gpu
def test_bench_gpus_8_runs_eight_ranks_over_the_one_shot_transport():
    """the shape of the driver's 8-GPU run (BASELINE config 3 = 8 shards) as far as a one-GPU box can take it: `bench.py --gpus 8` starts eight
    ranks, each a process with its own context, the cross-process one-shot all-reduce joins them (eight receive slots per rank, 7 peers written
    per collective), the line reports eight image counts and the communicator size.  All ranks share device 0 (--share-device), so the
    rate means nothing; that every rank finishes the same bond updates with the same bits is checked inside the library (check_replicas)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    run = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--share-device", "--sites", "24", "--images", "8000", "--maxm", "12",
                          "--steps", "6", "--warmup", "2", "--plain", "--no-cpu-baseline"], capture_output=True, text=True, cwd=ROOT, timeout=900, env=env)
    assert run.returncode == 0, (run.stderr + run.stdout)[-3000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, run.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["images_per_rank"] == [1000] * 8 and d["config"]["rccl_ranks"] == 8
    o = d["collectives"]["oneshot"]
    assert o["ranks"] == 8 and o["value"] == d["value"] and o["allreduces_per_bond_update"] >= 4
    assert d.get("replica_repairs", 0) == 0
