"""bench.py starts its own ranks when `--gpus N` is given without a launcher: `python bench.py --gpus 8`
has the shape of the recorded N = 1 command, so the scaling run needs no wrapper.  The reference's counterpart is the
process pool of MultiprocessingDistributor (tsfresh/utilities/distribution.py:438-494)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_self_launch_command_for_eight_ranks():
    import bench
    port = bench.free_port()
    assert 1024 < port < 65536
    argv = ["--gpus", "8", "--steps", "5", "--warmup", "2"]
    cmd, env = bench.self_launch_command(8, argv, port)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd
    assert cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"      # the container hostname may not resolve
    assert cmd[cmd.index("--master-port") + 1] == str(port)
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == argv                                  # the ranks see the caller's flags unchanged
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and env["TSFA_BENCH_LAUNCHED"] == "1"


def test_a_launched_rank_does_not_launch_again(monkeypatch):
    """Under a launcher (WORLD_SIZE set) main() goes on to the device check instead of spawning: no GPU here, so it
    stops with the 'needs a HIP device' message -- not with a subprocess."""
    import bench
    calls = []
    monkeypatch.setattr(subprocess, "call", lambda *a, **k: calls.append(a) or 0)
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    import torch
    if torch.cuda.is_available():
        pytest.skip("needs a box without a GPU")
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert "HIP device" in str(e.value) and not calls


def test_without_a_launcher_gpus_n_spawns_n_ranks(monkeypatch):
    import bench
    calls = []
    monkeypatch.setattr(subprocess, "call", lambda cmd, env=None: calls.append((cmd, env)) or 7)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7 and len(calls) == 1                    # the job's exit code is handed back
    cmd, env = calls[0]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]


def test_strong_scaling_flag_reaches_the_ranks_unchanged():
    """`--strong --n-series TOTAL` (configs[3]: 1 M x 256 split over N): the self-launch hands the flag to every rank; a rank
    derives its share from WORLD_SIZE (bench.py main)."""
    import bench
    argv = ["--gpus", "8", "--strong", "--n-series", "1000000", "--length", "256"]
    cmd, _ = bench.self_launch_command(8, argv, 29999)
    assert cmd[-len(argv):] == argv


@pytest.mark.gpu
def test_bench_self_launch_path_on_one_gpu(gpu):
    """`python bench.py --gpus 1` THROUGH the self-launch path (torch.distributed.run, one rank, RCCL communicator of
    one): the JSON line comes back on stdout with the multi_gpu block saying what RCCL saw."""
    env = dict(os.environ, TSFA_BENCH_SELF_LAUNCH="1", TSFA_BENCH_FORCE_DIST="1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
                          "--n-series", "2000", "--no-cpu-baseline", "--no-e2e"], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 1 and doc["value"] > 0 and doc["outputs_finite"]
    assert doc["multi_gpu"]["world"] == 1 and doc["multi_gpu"]["ranks_seen_by_rccl"] == 1
    assert doc["multi_gpu"]["self_launched"] is True
    assert doc["parity_sample"] == "ok"
