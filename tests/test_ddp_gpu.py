"""W = 2 on one GPU: the HIP model through ``engine.train_step`` + the real ``GradSync`` in two fresh child
processes (gloo, both on cuda:0) against the single-process run on the concatenated batch (SURVEY.md 8(e), H3;
reference contract: DistributedDataParallel, train.py:222-225; gather pattern utils.py:192-206).  Cases: K = 1,
K = 5 hard negatives, alignment (two passes per tower), the region branch (three text passes), and config 4's
combination; and the reference's own loop on ``distributed.DistributedDataParallel(model)`` (local-batch loss, gradients
averaged by the time ``backward()`` returns)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_ranks(script, case, W=2, timeout=420, extra_env=None):
    port = _free_port()
    procs = []
    for r in range(W):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(W), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), CASE=case, HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", script)], env=env, cwd=ROOT,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=timeout)
            outs.append(out)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return [p.returncode for p in procs], outs


@pytest.mark.gpu
@pytest.mark.timeout(900)
@pytest.mark.parametrize("case", ["k1", "k5", "align", "region", "all", "wrapper", "sharded"])
def test_two_rank_step_equals_concatenated_batch(case):
    rcs, outs = run_ranks("ddp_child.py", case)
    print(outs[0][-3000:])
    assert rcs == [0, 0], "\n".join(o[-3000:] for o in outs)
    assert f"[{case}] OK" in outs[0]


@pytest.mark.gpu
@pytest.mark.timeout(900)
def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (the driver's N > 1 call when it does not use
    torch.distributed.run): bench.py starts the two ranks itself BEFORE touching the GPU (clip_event_amd/launch.py;
    reference contract train.sh:2, utils.py:541-616) and relays rank 0's ONE JSON line with n_gpus = 2 from the live
    process group.  On this one-GPU box both ranks share cuda:0 (CE_ALL_RANKS_ON_GPU0=1; RCCL refuses two ranks on one
    device, so the rehearsal's collectives run on gloo and rccl_ranks is 0; on an N-GPU node backend nccl = RCCL)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(CE_ALL_RANKS_ON_GPU0="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "64",
                        "--no-cpu-baseline", "--no-roofline", "--no-dense-compare"], env=env, cwd=ROOT, capture_output=True,
                       text=True, timeout=800)
    print(r.stderr[-3000:])
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 128 and d["config"]["parallelism"] == "dp2"
    assert d["rccl_ranks"] == 0 and d["value"] > 0 and d["scaling"] == "weak"
