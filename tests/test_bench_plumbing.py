"""CPU: bench.py's rank plumbing fails cleanly (non-zero exit and a message, before any GPU call) when the launch does
not match `--gpus`, and the reducer's exposed-communication bookkeeping leaves CPU runs alone."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, cwd=ROOT,
                          capture_output=True, text=True, timeout=300)


@pytest.mark.parametrize("args,env,needle", [
    (["--gpus", "2"], {}, "WORLD_SIZE=1"),                                  # N > 1 without the launcher
    (["--gpus", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, "WORLD_SIZE=2"),
    (["--gpus", "4"], {"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1"}, "--gpus 4 but WORLD_SIZE=2"),
    (["--gpus", "2"], {"WORLD_SIZE": "2", "RANK": "5", "LOCAL_RANK": "0"}, "RANK=5"),
    (["--gpus", "0"], {}, "--gpus 0"),
])
def test_mismatched_launch_exits_with_a_message(args, env, needle):
    r = _bench(args, env)
    assert r.returncode != 0
    assert "bench.py:" in r.stderr and needle in r.stderr, r.stderr
    assert not r.stdout.strip()                       # no JSON line from a run that did not happen
