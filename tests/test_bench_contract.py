"""The bench line's contract (the driver parses it): checked on the committed lines of rounds 2 and 3 (profiles/r0N_bench_*.json, written by
`python bench.py [--config N]` on an MI355X) and on bench.py's command line. No GPU needed."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


@pytest.mark.parametrize("name,config", [("r02_bench_n1.json", 2), ("r02_bench_config2.json", 2), ("r02_bench_config4.json", 4), ("r02_bench_config5.json", 5),
                                         ("r03_bench_n1.json", 2), ("r03_bench_config2.json", 2), ("r03_bench_config4.json", 4), ("r03_bench_config5.json", 5)])
def test_committed_bench_lines_keep_the_contract(name, config):
    d = _line(name)
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["unit"] == "Mrays/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] >= 1 and d["scaling"] == "strong" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert d["config"]["baseline_config"] == config and d["config"]["baseline_parameters"] is True and "workload" in d["config"] and "model" not in d["config"]
    # value = rays of a step / time of a step
    assert d["value"] == pytest.approx(d["config"]["rays_per_step"] / d["ms_per_step"] / 1e3, rel=1e-6)
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert 0.0 < r["frac"] <= 1.0 and r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-9)
    # achieved = queue bytes of the dominant class per launch / its launch time
    assert r["achieved"] == pytest.approx(r["queue_bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9, rel=1e-6)
    assert 0.0 < r["whole_step"]["frac"] <= 1.0
    assert r["traffic"] is None or (r["traffic_source"] and "profiles/" in r["traffic_source"])
    if config == 5:
        r["traffic"] = None            # lines written before bench.py stopped quoting the 8-spp PMC figure for the 64-spp run
    if r["traffic"] is not None:       # counted queue bytes against the PMC figure of the same launches: about equal when the tree is LDS-resident
        ratio = r["queue_bytes_per_launch"] / r["traffic"]        # (config 2); with the tree in global memory its nodes and triangles add HBM traffic
        assert (0.8 < ratio < 1.25) if config == 2 else (0.2 < ratio < 1.25), ratio
    assert set(r["kernels"]) >= {"wf_extend", "wf_shade", "wf_shadow", "wf_resolve"}
    assert "algorithmic_model" in r and r["algorithmic_model"]["bytes_per_step"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "Mrays/s" and c["sample"]
    assert d["one_frame_in_flight"]["ms_per_step"] >= d["ms_per_step"] * 0.9
    if name.startswith("r03"):
        # the dominant class is named from this run's times; the PMC figures it quotes are the round's own; the VALU counters say what they are read against
        assert r["kernel"] == max(r["kernels"], key=lambda k: r["kernels"][k]["ms_per_step"])
        assert r["traffic_source"] is None or "profiles/r03_pmc_traffic" in r["traffic_source"]
        assert r["valu"]["issue_rate_reference"]["full_rate_cycles"] == 2.2 and r["valu"]["issue_rate_reference"]["half_rate_cycles"] == 4.2
        assert "profiles/r03_pmc_valu" in r["valu"]["source"]


def test_bench_command_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--config", "--serial-kernels", "--frames-in-flight"):
        assert flag in out.stdout


def test_gpus_n_without_launcher_starts_fresh_ranks():
    """`python bench.py --gpus N` as the driver types it (no torch.distributed.run, WORLD_SIZE unset): the parent turns into the launcher
    BEFORE it imports torch or loads the HIP library (a process that touched the GPU must not fork ranks from itself), hands the ranks the
    unchanged command line and a 127.0.0.1 rendezvous, and exits with their status."""
    probe = r"""
import json, os, subprocess, sys
sys.path.insert(0, %r)
os.environ.pop("WORLD_SIZE", None)
import bench
seen = {}
def fake_call(cmd, env=None):
    seen["cmd"] = cmd; seen["env_ipc"] = env.get("HSA_ENABLE_IPC_MODE_LEGACY")
    return 7
subprocess.call = fake_call
try:
    bench.main(["--gpus", "4", "--steps", "3", "--warmup", "1"])
    code = None
except SystemExit as e:
    code = e.code
with open("/proc/self/maps") as f:
    maps = f.read()
print(json.dumps({"code": code, "cmd": seen.get("cmd"), "ipc": seen.get("env_ipc"),
                  "torch": any(m == "torch" or m.startswith("torch.") for m in sys.modules),
                  "hip": ("libhobbyrt_pt" in maps) or ("libamdhip64" in maps)}))
""" % ROOT
    out = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["code"] == 7                               # the ranks' exit status is the parent's
    assert d["torch"] is False and d["hip"] is False    # nothing GPU-related was imported on the way
    cmd = d["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert d["ipc"] == "0"


def test_gpus_n_under_a_launcher_does_not_relaunch():
    """With WORLD_SIZE set (torch.distributed.run started us) the rank path is taken: --help still parses, and _launch_ranks is not reached."""
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); import bench, subprocess\n"
                          "subprocess.call = lambda *a, **k: (_ for _ in ()).throw(AssertionError('relaunched'))\n"
                          "bench._launch_ranks = lambda *a: (_ for _ in ()).throw(AssertionError('relaunched'))\n"
                          "import argparse\n"
                          "try:\n    bench.main(['--gpus', '2', '--help'])\nexcept SystemExit as e:\n    print('exit', e.code)\n" % ROOT],
                         capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0 and "exit 0" in out.stdout, out.stderr
