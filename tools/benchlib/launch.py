"""`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): the parent starts the N ranks itself as fresh
child processes - through torch.distributed.run, exactly the command the driver would have used - BEFORE anything touches the GPU
(the parent never initialises HIP: it only imports this module), forwards their output (rank 0 prints the judged line) and exits
with the launcher's code. The reference trains with one process per GPU the same way (run.py:50-52, strategy="ddp")."""
import os
import socket
import subprocess
import sys


def needs_spawn(n_gpus: int) -> bool:
    return n_gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_command(script: str, n_gpus: int, argv, port: int):
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port), script, *argv]


def spawn_ranks(script: str, n_gpus: int, argv) -> int:
    """Runs the N ranks to completion; returns the launcher's exit code (non-zero if any rank failed)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = launcher_command(script, n_gpus, list(argv), free_port())
    print("bench.py: no WORLD_SIZE in the environment - launching " + " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode
