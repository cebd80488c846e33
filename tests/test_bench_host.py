"""CPU tests (-m "not gpu") of bench.py's host logic: the rank launcher, the kernel-timing sampling
period, JSON hygiene and the PMC-summary stamp.  No compute call, no GPU."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launcher_starts_n_ranks_by_itself():
    """`python bench.py --gpus 2` outside torchrun must run TWO ranks (it ran one and
    printed n_gpus 1).  The selftest does what a rank does minus the GPU: gloo group of WORLD_SIZE."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-launcher"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2
    # ONE N > 1 run carries the three legs -- no collation, the whole block, (reward, done) -- each payload
    # with its own xGMI fraction (here: gloo on the CPU, the structure of the line)
    legs = d["payload_legs"]
    assert set(legs) == {"compute_only", "full", "obs8", "reward_done"}
    assert legs["full"]["rows_per_step"] == 13.0 and legs["reward_done"]["rows_per_step"] == 2.0 and legs["obs8"]["rows_per_step"] == 10.0
    # ... and what xGMI allows each payload before anything is measured: kernel / max(kernel, link time of one block)
    exp = d["expected_payload_efficiency"]
    assert set(exp) == {"full", "obs8", "reward_done", "full_every_k"}
    assert abs(exp["full"]["link_us_per_step"] - 13 * 65536 * 4 / 76.5e9 * 1e6) < 1e-6
    assert exp["full"]["expected_efficiency"] < exp["obs8"]["expected_efficiency"] < exp["reward_done"]["expected_efficiency"] <= 1.0
    for k in ("full", "obs8", "reward_done"):
        x = legs[k]["xgmi_roofline"]
        assert x["bound"] == "xgmi" and x["bytes_received_per_step_per_gpu"] == 256 * 4 * legs[k]["rows_per_step"] and x["frac"] > 0
    assert all(v["value"] > 0 for v in legs.values())
    # the line explains itself -- value is the 'full' leg, compute_only beside it, expectation beside measurement
    note = d["scaling_note"]
    assert "payload 'full' at 2 GPUs" in note and "compute_only" in note and "full expected 0.18 measured" in note
    assert "obs8 expected 0.23" in note and "reward_done expected 1.00" in note


def test_launcher_propagates_a_failing_rank():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    code = "import sys, os, bench; sys.exit(bench.launch_ranks(2, ['--definitely-not-an-option']))"
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert p.returncode != 0


@pytest.mark.parametrize("steps,every,lo,hi", [(4000, 64, 62, 63), (300, 64, 8, 300), (256, 64, 4, 32), (20, 64, 4, 20)])
def test_every_run_records_kernel_samples(steps, every, lo, hi):
    """the driver's 5 + 20 steps recorded zero samples at --time-every 64 -> NaN.  Long
    windows carry bracketed launches inside; short ones (< 256 steps) are bracketed as a whole and get
    their per-launch samples right behind the window (run_rank) -- sample_period still never returns a
    period that would leave a window without samples."""
    p = bench.sample_period(every, steps)
    assert p >= 1
    n = steps // p
    assert lo <= n <= hi, (p, n)
    assert bench.sample_period(0, steps) == 0


def test_json_line_never_carries_nan():
    assert bench.finite(float("nan")) is None and bench.finite(float("inf")) is None and bench.finite(None) is None
    assert bench.finite(0.25) == 0.25
    with pytest.raises(ValueError):
        json.dumps({"x": float("nan")}, allow_nan=False)


def test_pmc_summary_counts_only_for_the_sources_it_was_measured_on(tmp_path, monkeypatch):
    sha = bench.kernel_source_hash()
    assert len(sha) == 16
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_hash", lambda: sha)
    (prof / "r09_pmc_x_traffic.json").write_text(json.dumps(
        {"kernel": "k", "envs": 8, "traffic_bytes_per_launch": 123.0, "kernel_src_sha": "stale"}))
    assert bench.pmc_traffic("k", 8) == (None, None)
    (prof / "r10_pmc_x_traffic.json").write_text(json.dumps(
        {"kernel": "k", "envs": 8, "traffic_bytes_per_launch": 456.0, "kernel_src_sha": sha}))
    assert bench.pmc_traffic("k", 8) == (456.0, "r10_pmc_x_traffic.json")
    assert bench.pmc_traffic("k", 16) == (None, None)


def test_visible_gpu_count_without_the_hip_runtime(tmp_path, monkeypatch):
    """the parent that starts the ranks counts GPUs from the environment / the KFD topology, not
    through torch.cuda (whatever that initialises)."""
    import bench
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        monkeypatch.delenv(var, raising=False)
    for k, simd in enumerate((0, 0, 1024, 1024, 1024)):          # two CPU nodes, three GPUs
        d = tmp_path / str(k)
        d.mkdir()
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\nmem_banks_count 1\n" % (64 if simd == 0 else 0, simd))
    assert bench.visible_gpu_count(str(tmp_path)) == 3
    assert bench.visible_gpu_count(str(tmp_path / "missing")) == -1
    monkeypatch.setenv("HIP_VISIBLE_DEVICES", "0,1")
    assert bench.visible_gpu_count(str(tmp_path)) == 2
    monkeypatch.setenv("ROCR_VISIBLE_DEVICES", "")
    assert bench.visible_gpu_count(str(tmp_path)) == 0
    src = open(bench.__file__).read()
    assert "torch.cuda.device_count" not in src
