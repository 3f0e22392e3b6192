"""bench.py --gpus N: a failed start-up of the process group ends in ONE parseable JSON line and a non-zero exit code (round 5).

The first RCCL collectives this code ever issues are the driver's scaling run (the pool hands out one-GPU boxes), so the start-up must
not be a traceback on one rank and a hang on the others.  Forced here over gloo on the CPU: a rank that finds nobody listening on the
rendezvous port, a rank 0 nobody joins, and a stage that hangs (the timer thread of bench.DistStage)."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _launch(rank):
    env = dict(os.environ, WORLD_SIZE="2", RANK=str(rank), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               GM_BENCH_BACKEND="gloo", GM_BENCH_DIST_TIMEOUT="4")
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"], cwd=ROOT, env=env,
                          capture_output=True, text=True, timeout=240)


def _error_line(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert len(lines) == 1, text[-1500:]
    return json.loads(lines[0])


def test_rank_without_a_rendezvous_reports_one_json_line():
    out = _launch(1)                                   # nobody listens on the port: connection refused until the timeout
    assert out.returncode == 2
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]          # stdout belongs to rank 0
    e = _error_line(out.stderr)
    assert e["stage"] == "gloo_init" and e["rank"] == 1 and e["value"] is None and e["n_gpus"] == 2
    assert "HSA_ENABLE_IPC_MODE_LEGACY" in e["rank_env"] and e["rank_env"]["MASTER_ADDR"] == "127.0.0.1"
    assert "timed out" in e["error"].lower() or "refused" in e["error"].lower(), e["error"]


def test_rank_zero_nobody_joins_reports_on_stdout():
    out = _launch(0)
    assert out.returncode == 2
    e = _error_line(out.stdout)
    assert e["stage"] == "gloo_init" and e["rank"] == 0 and "clients joined" in e["error"]


def test_a_hanging_stage_is_ended_by_the_timer():
    code = ("import sys, time; sys.argv = ['bench.py']; import bench\n"
            "with bench.DistStage('first_broadcast', 0, 8, 1.0):\n"
            "    time.sleep(60)\n")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=240)
    assert out.returncode == 3
    e = _error_line(out.stdout)
    assert e["stage"] == "first_broadcast" and "hang" in e["error"] and e["n_gpus"] == 8
