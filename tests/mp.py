"""TEST infrastructure: start `world` worker processes the way a launcher would (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT in the environment, rendezvous on 127.0.0.1) and hand back rank 0's last stdout line as JSON."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_ranks(tmp_path, name: str, source: str, world: int, args=(), env=None, timeout=900):
    script = tmp_path / f"{name}.py"
    script.write_text(source)
    port = free_port()
    procs = []
    for r in range(world):
        e = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                 OMP_NUM_THREADS="2" if world <= 2 else "1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        e.update(env or {})
        procs.append(subprocess.Popen([sys.executable, str(script), ROOT, *map(str, args)], env=e, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (o, e)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n" + e.decode()[-3000:]
    return json.loads(outs[0][0].decode().strip().splitlines()[-1])
