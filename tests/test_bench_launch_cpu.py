"""`python bench.py --gpus N` with WORLD_SIZE unset must launch its own ranks (the driver's SCALE
command) instead of exiting: the launcher branch is run here with a stubbed child command -- exit-code
propagation, exactly one JSON line on stdout, arguments passed through, no GPU touched."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = r"""
import json, os, sys
rc = int(os.environ.get("STUB_RC", "0"))
print("noise from some rank")                       # must not reach the launcher's stdout
if os.environ.get("STUB_SILENT") != "1":
    print(json.dumps({"metric": "stub", "value": 1.0, "n_gpus": 2, "argv": sys.argv[1:],
                      "world_env": os.environ.get("WORLD_SIZE")}))
print("more noise", file=sys.stderr)
sys.exit(rc)
"""


def run(extra_env, args=("--gpus", "2", "--steps", "3", "--warmup", "1", "--no-legs")):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["RAG_BENCH_LAUNCH_CMD"] = json.dumps([sys.executable, "-c", STUB])
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), env=env,
                          capture_output=True, text=True, timeout=120)


def test_launcher_relays_one_json_line_and_passes_arguments():
    r = run({})
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["metric"] == "stub" and out["n_gpus"] == 2
    assert out["argv"] == ["--gpus", "2", "--steps", "3", "--warmup", "1", "--no-legs"]
    assert "noise from some rank" in r.stderr and "more noise" in r.stderr


def test_launcher_propagates_the_exit_code():
    r = run({"STUB_RC": "7"})
    assert r.returncode == 7
    assert len([ln for ln in r.stdout.splitlines() if ln.strip()]) == 1


def test_launcher_fails_when_no_result_line_was_printed():
    r = run({"STUB_SILENT": "1"})
    assert r.returncode != 0
    assert r.stdout.strip() == ""


def test_launcher_never_imports_torch_or_the_engine():
    """The parent must not initialise the GPU before starting its children."""
    code = ("import sys, json; sys.argv = ['bench.py', '--gpus', '2']; import bench\n"
            "rc = bench.self_launch(2, ['--gpus', '2'], child_cmd=[sys.executable, '-c', "
            "'print(chr(123) + chr(34) + \"metric\" + chr(34) + \": 1\" + chr(125))'])\n"
            "assert rc == 0, rc\n"
            "assert 'torch' not in sys.modules and 'rag_system_amd' not in sys.modules, sorted(sys.modules)[:5]\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip() == '{"metric": 1}'


def test_default_launcher_command_is_torch_distributed_run(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    class FakeProc:
        stdout = iter(['{"metric": "x"}\n'])

        def wait(self):
            return 0

    def fake_popen(cmd, **kw):
        seen["cmd"] = cmd
        seen["env"] = kw["env"]
        return FakeProc()

    monkeypatch.setattr(subprocess, "Popen", fake_popen)
    monkeypatch.delenv("RAG_BENCH_LAUNCH_CMD", raising=False)
    assert bench.self_launch(4, ["--gpus", "4", "--steps", "5"]) == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "5"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
