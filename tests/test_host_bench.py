"""
bench.py --gpus N without a torch.distributed environment starts the N ranks itself, as a child process, before it
imports torch or the HIP library (the parent must never touch the GPU), and exits with the child's status.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_PROBE = r"""
import json, os, sys
sys.path.insert(0, %r)
os.environ.pop("WORLD_SIZE", None)
sys.argv = ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"]
import bench
seen = {}
def fake_call(cmd, env=None):
    seen["cmd"] = cmd
    seen["env_ipc"] = (env or {}).get("HSA_ENABLE_IPC_MODE_LEGACY")
    return 7
bench.subprocess.call = fake_call
try:
    bench.main()
    code = 0
except SystemExit as e:
    code = e.code
seen["code"] = code
seen["torch_loaded"] = "torch" in sys.modules
seen["lib_loaded"] = "libdmet_preview_amd._lib" in sys.modules
print(json.dumps(seen))
""" % ROOT


def test_bench_parent_spawns_ranks_without_touching_the_gpu():
    import json
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, "-c", _PROBE], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stderr
    seen = json.loads(r.stdout.strip().splitlines()[-1])
    cmd = seen["cmd"]
    assert seen["code"] == 7                                   # the child's exit status is propagated
    assert not seen["torch_loaded"] and not seen["lib_loaded"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1].isdigit()
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env_ipc"] == "0"


def test_bench_rank_uses_world_size_from_env():
    """Inside a torch.distributed.run launch (WORLD_SIZE set) bench.py must NOT spawn again."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'if a.gpus > 1 and "WORLD_SIZE" not in os.environ:' in src
