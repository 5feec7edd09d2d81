"""The N > 1 path of bench.py on a one-GPU box: two (and four) ranks share cuda:0, process group over
gloo, per-round exchange through host shared memory, residual rounds inside the compiled call.
RCCL refuses several ranks on one device, so its path is covered by BN_FORCE_SHARDED=1 (world = 1)
below; everything else of the multi-GPU flow is exactly what the driver launches on 2/4/8 GPUs."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(cmd, env):
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=600)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and lines, out.stdout[-2000:] + out.stderr[-2000:]
    return json.loads(lines[-1])


@pytest.mark.parametrize("exchange", ["shm", "peer"])
@pytest.mark.parametrize("n_ranks", [2, 4])
def test_bench_two_and_four_ranks_on_one_gpu(n_ranks, exchange):
    """bench.py's N > 1 path: STRONG scaling of a fixed global instance (here 2^17 elements), gloo process group (RCCL
    refuses several ranks on one device); the partials meet in host shared memory (shm) or in the peers' hipIpc-mapped
    device mailboxes (peer, the default on a node); the other one is timed as alt_exchange and must give the same
    transcript."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", str(n_ranks), "--steps", "2", "--warmup", "1", "--n-vars", "17",
           "--no-cpu-baseline"]
    r = _run(cmd, {"BN_ALL_ON_GPU0": "1", "BN_PG_BACKEND": "gloo", "BN_EXCHANGE": exchange})
    assert r["n_gpus"] == n_ranks and r["verifier_check"] is True
    assert (" -- " not in r["config"]["sharding"]), "the requested exchange fell back: " + r["config"]["sharding"]
    others = [a for a in r["alt_exchange"] if "same_transcript_as_default" in a]
    assert others and all(a["same_transcript_as_default"] for a in others)
    assert r["config"]["n_vars_global"] == 17
    assert r["config"]["n_vars_local"] == 17 - (n_ranks.bit_length() - 1)
    assert r["scaling"] == "strong"


def test_bench_transcript_is_independent_of_the_number_of_ranks():
    """Strong scaling proves the SAME instance: bench.py prints a digest of the transcript it timed (all round
    polynomials + final evaluations); the 1-, 2- and 4-rank runs of the 2^17 instance must print the same one."""
    digests = []
    for n_ranks in (1, 2, 4):
        if n_ranks == 1:
            r = _run([sys.executable, "bench.py", "--steps", "1", "--warmup", "1", "--n-vars", "17", "--no-cpu-baseline"], {})
        else:
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1",
                   "--master-port", str(_free_port()), "bench.py", "--gpus", str(n_ranks), "--steps", "1", "--warmup", "1", "--n-vars", "17",
                   "--no-cpu-baseline"]
            r = _run(cmd, {"BN_ALL_ON_GPU0": "1", "BN_PG_BACKEND": "gloo", "BN_EXCHANGE": "shm"})
        assert r["verifier_check"] is True and r["config"]["n_vars_global"] == 17
        digests.append(r["transcript_digest"])
    assert len(digests[0]) == 32 and digests[0] == digests[1] == digests[2], digests


def test_bench_bare_gpus_2_spawns_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher (WORLD_SIZE unset) must be a two-rank run -- the
    process re-executes itself under torch.distributed.run -- and print `n_gpus: 2`, never an N = 1 line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"BN_ALL_ON_GPU0": "1", "BN_PG_BACKEND": "gloo", "BN_EXCHANGE": "shm"})
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "1", "--warmup", "1", "--n-vars", "17", "--no-cpu-baseline"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode == 0 and lines, out.stdout[-2000:] + out.stderr[-2000:]
    r = json.loads(lines[-1])
    assert r["n_gpus"] == 2 and r["verifier_check"] is True and r["config"]["n_vars_local"] == 16


def test_bench_bare_gpus_8_at_n20_on_one_device():
    """the command the driver's scaling run issues for eight GPUs -- `python bench.py --gpus 8`, no launcher --
    with all eight ranks on device 0 (BN_ALL_ON_GPU0=1; the only multi-rank run a one-GPU box allows): eight ranks are spawned,
    the line says n_gpus 8, the sharded prover's transcript satisfies the verifier and its digest is the single-GPU digest of the
    same 2^20 instance."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"BN_ALL_ON_GPU0": "1", "BN_PG_BACKEND": "gloo"})
    out = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--steps", "2", "--warmup", "1", "--n-vars", "20", "--no-cpu-baseline"],
                         cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert out.returncode == 0 and lines, out.stdout[-2000:] + out.stderr[-2000:]
    r = json.loads(lines[-1])
    assert r["n_gpus"] == 8 and r["verifier_check"] is True and r["config"]["n_vars_local"] == 17 and r["config"]["n_vars_global"] == 20
    one = _run([sys.executable, "bench.py", "--steps", "1", "--warmup", "1", "--n-vars", "20", "--no-cpu-baseline", "--no-claim-groups"], {})
    assert one["transcript_digest"] == r["transcript_digest"]


@pytest.mark.parametrize("exchange", ["shm", "rccl", "peer"])
def test_bench_sharded_code_path_world1(exchange):
    r = _run([sys.executable, "bench.py", "--steps", "2", "--warmup", "1", "--n-vars", "15", "--no-cpu-baseline"],
             {"BN_FORCE_SHARDED": "1", "BN_EXCHANGE": exchange})
    assert r["verifier_check"] is True and r["n_gpus"] == 1
