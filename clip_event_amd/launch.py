"""One command -> one process per GPU.

The reference starts its data-parallel job with ONE command that spawns the ranks
(`train.sh:2`: ``python -m torch.distributed.launch --nproc_per_node=N --use_env train.py``) and each rank reads
RANK / WORLD_SIZE / LOCAL_RANK from the environment (`utils.py:562-570`, ``init_distributed_mode``).  ``spawn_ranks``
is that launcher for this package's entry points (``bench.py --gpus N``): it starts N children with the same environment
contract and relays rank 0's standard output.

The parent never touches the GPU: on this pool a process that has initialised HIP must not be replaced or forked into
workers, so the children are fresh interpreters (``subprocess``), started BEFORE any device call, and the parent only
waits for them.  Rendezvous is on 127.0.0.1 (the container host name may not resolve).
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import threading
from typing import Dict, List, Optional, Sequence


def visible_gpus() -> Optional[int]:
    """GPUs this process may use, counted WITHOUT touching HIP (``torch.cuda.device_count()`` falls back to
    ``hipGetDeviceCount`` -- which initialises the runtime in the parent -- when its amdsmi path fails): the visibility
    variables first, then the KFD topology in sysfs (nodes with SIMDs are GPUs).  ``None`` when neither is readable."""
    for var in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    root = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(root):
            with open(os.path.join(root, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        return n
    except OSError:
        return None


def free_port() -> int:
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def rank_env(rank: int, world: int, port: int, base: Optional[Dict[str, str]] = None) -> Dict[str, str]:
    """Environment of one rank: what torch.distributed.run would set for a one-node job."""
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this driver (RCCL needs it)
    return env


def _pump(stream, sink, prefix: str = "") -> None:
    for line in iter(stream.readline, ""):
        sink.write(prefix + line)
        sink.flush()
    stream.close()


def spawn_ranks(nproc: int, argv: Sequence[str], env_extra: Optional[Dict[str, str]] = None,
                stdout=None, stderr=None, timeout: Optional[float] = None) -> int:
    """Start ``nproc`` children running ``argv`` (rank r gets RANK = LOCAL_RANK = r), relay rank 0's stdout to
    ``stdout`` and every rank's stderr to ``stderr``, and return 0 only if every child exited 0 (otherwise the first
    non-zero exit code; the remaining children are terminated as soon as one fails, so a dead rank cannot leave the
    others waiting in a collective)."""
    if nproc < 1:
        raise ValueError(f"spawn_ranks: nproc={nproc}")
    stdout = sys.stdout if stdout is None else stdout
    stderr = sys.stderr if stderr is None else stderr
    port = free_port()
    procs: List[subprocess.Popen] = []
    pumps: List[threading.Thread] = []
    for r in range(nproc):
        env = rank_env(r, nproc, port)
        if env_extra:
            env.update(env_extra)
        p = subprocess.Popen(list(argv), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, bufsize=1)
        procs.append(p)
        if r == 0:
            pumps.append(threading.Thread(target=_pump, args=(p.stdout, stdout), daemon=True))
        else:       # other ranks print nothing on stdout by contract; whatever they do print must not corrupt rank 0's line
            pumps.append(threading.Thread(target=_pump, args=(p.stdout, stderr, f"[rank {r} stdout] "), daemon=True))
        pumps.append(threading.Thread(target=_pump, args=(p.stderr, stderr, f"[rank {r}] " if r else ""), daemon=True))
    for t in pumps:
        t.start()
    rc = 0
    import time
    t_end = None if timeout is None else time.time() + timeout
    alive = set(range(nproc))
    while alive:
        for r in sorted(alive):
            code = procs[r].poll()
            if code is None:
                continue
            alive.discard(r)
            if code != 0 and rc == 0:
                rc = code
                stderr.write(f"[launch] rank {r} exited with code {code}; stopping the other ranks\n")
                for q in alive:
                    procs[q].terminate()
        if t_end is not None and time.time() > t_end and alive:
            stderr.write(f"[launch] timeout after {timeout}s; stopping ranks {sorted(alive)}\n")
            for q in alive:
                procs[q].kill()
            rc = rc or 124
            t_end = None
        time.sleep(0.05)
    for p in procs:
        p.wait()
    for t in pumps:
        t.join(timeout=5)
    return rc
