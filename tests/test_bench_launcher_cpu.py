"""bench.py's own launcher (`python bench.py --gpus N` with no torchrun around it), driven on CPU: `--selftest` keeps the launcher,
the rendezvous on 127.0.0.1, the barrier + max-over-ranks bracket and the one-JSON-line contract, and swaps the GPU work for a sleep
on the gloo backend.  What the driver runs at round end is exactly this path with the GPU modes instead of --selftest."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, f"stdout must carry exactly one JSON line, got {lines}"
    return json.loads(lines[0])


def test_gpus_2_spawns_two_ranks_and_prints_one_line():
    out = _run(["--gpus", "2", "--selftest", "--steps", "5", "--warmup", "1"])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["warmup"] == 1
    assert out["config"]["group_world_size"] == 2 and out["config"]["requested_gpus"] == 2
    # 2 ranks x 5 steps of >= 2 ms each, max over ranks: the aggregate cannot beat 2 / 2 ms
    assert 0 < out["value"] <= 1000.0 + 1e-6
    assert out["scaling"] == "weak" and out["higher_is_better"] is True
    # the command the driver runs (`bench.py --gpus N`, default mode) carries the SFT gradient exchange: on the GPU the real step with the
    # process group on every rank (bench.py sft_side_measurement), here its dry run over gloo
    sft = out["sft"]
    assert sft["world"] == 2 and "world=2" in sft["grad_exchange"] and sft["exchange_active"] is True and sft["exchange_ok"] is True
    assert sft["exchange_bytes"] > 0 and sft["buckets"] >= 8


def test_single_rank_needs_no_launcher():
    out = _run(["--gpus", "1", "--selftest", "--steps", "3", "--warmup", "0"])
    assert out["n_gpus"] == 1 and out["config"]["group_world_size"] == 1


def test_under_torchrun_the_environment_wins():
    """The driver's N > 1 form: torch.distributed.run sets RANK/WORLD_SIZE; bench.py must not spawn again."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest", "--steps", "4", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1
    assert json.loads(lines[0])["n_gpus"] == 2


def test_sft_mode_dry_run_on_two_ranks():
    """`python bench.py --mode sft --gpus 2` as the driver will run it, minus the kernels: the launcher, the trainer's flat buffers, the
    global token count and every gradient bucket's SUM all-reduce over the group."""
    out = _run(["--gpus", "2", "--mode", "sft", "--selftest", "--steps", "2", "--warmup", "1"])
    c = out["config"]
    assert out["n_gpus"] == 2 and c["group_world_size"] == 2 and c["mode"] == "sft dry run"
    assert c["global_num_items"] == 100 + 101 and c["exchange_ok"] is True and c["buckets"] >= 8
    assert "world=2" in c["grad_exchange"]


def test_stdout_redirect_also_catches_c_stdio_writes():
    """RCCL's version banner is written with C stdio while the communicator comes up; buffered, it used to surface at process exit BEHIND the JSON
    line (round 4, on hardware).  bench.stdout_to_stderr must flush libc's buffers before it restores fd 1: stdout then carries the line only."""
    code = ("import sys, ctypes; sys.path.insert(0, %r); import bench\n"
            "with bench.stdout_to_stderr():\n"
            "    ctypes.CDLL(None).printf(b'BANNER via C stdio\\n')\n"
            "print('{\"ok\": 1}')\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-1000:]
    assert r.stdout.strip() == '{"ok": 1}' and "BANNER via C stdio" in r.stderr


def test_deadline_guard_prints_the_decode_line_and_exits_when_the_side_measurement_hangs(capsys):
    """bench.py's `Deadline` around the multi-rank SFT side measurement: a rank that dies inside a data-parallel step leaves the others in its
    all-reduce; the decode line (complete by then) must still come out, once, as the last stdout line, with exit status 0."""
    import importlib.util
    import time
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    codes = []
    g = bench.Deadline(30.0, lambda: '{"never": 1}', _exit=codes.append)
    assert g.cancel() is True and codes == []                       # finished in time: nothing printed, nothing exits
    g = bench.Deadline(0.05, lambda: '{"value": 341.0, "sft": {"error": "deadline"}}', _exit=codes.append)
    time.sleep(0.4)
    assert codes == [0] and g.cancel() is False                      # fired: the caller is told to stay away from stdout
    out = capsys.readouterr()
    assert out.out.strip().splitlines() == ['{"value": 341.0, "sft": {"error": "deadline"}}'] and "deadline" in out.err
    g = bench.Deadline(0.05, None, _exit=codes.append)              # a rank other than 0: exits quietly
    time.sleep(0.4)
    assert codes == [0, 0] and capsys.readouterr().out == ""
