"""Multi-process test of the sharded sumcheck (binius_amd/distributed.py) on CPU: world_size 2 and 4
over the gloo backend.  The per-shard field work is done by the oracle here (no GPU in this
container); what is under test is the sharding scheme (low index bits = last-bound variables), the
one-collective-per-round XOR combine, the residual gather and the tail rounds -- the round
polynomials and final evaluations must equal the UNSHARDED oracle prover's, bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleCompute:
    def __init__(self, oracle, local_mls, comps, batch_coeff):
        self.o, self.mls, self.comps, self.bc = oracle, local_mls, comps, batch_coeff
        self.rem = None

    def round_evals(self, rem):
        views = [x[: 1 << rem] for x in self.mls]
        rc, ev = self.o.round_evals(views, rem, self.comps, self.bc)
        assert rc == 0
        self.rem = rem
        return ev

    def fold(self, z):
        for x in self.mls:
            self.o.fold_high(x, self.rem, z)

    def finals(self):
        return [int(x[0, 0]) | (int(x[0, 1]) << 64) for x in self.mls]

    def tail_prove(self, residual, n_tail, running, batch_coeff, challenges):
        o = self.o
        mls = [o.ints_to_arr(v) for v in residual]
        coeffs = []
        for r in range(n_tail):
            rem = n_tail - r
            rc, (y_1, y_inf) = o.round_evals([x[: 1 << rem] for x in mls], rem, self.comps, batch_coeff)
            c_0 = running ^ y_1
            c_2 = y_inf
            c_1 = y_1 ^ c_0 ^ c_2
            coeffs.append([c_0, c_1, c_2])
            running = o.evaluate_univariate([c_0, c_1, c_2], challenges[r])
            for x in mls:
                o.fold_high(x, rem, challenges[r])
        return coeffs, [int(x[0, 0]) | (int(x[0, 1]) << 64) for x in mls]


class OracleField:
    def __init__(self, oracle):
        self.o = oracle

    def mul(self, a, b):
        return self.o.mul(a, b)


def _worker(rank, world, port, n_global, q, use_shm=False):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    import oracle
    from binius_amd.distributed import ShardedBivariateSumcheck, TorchComm, shard_indices

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        m, comps = 3, [(0, 1), (2, 0)]
        log_world = world.bit_length() - 1
        n_local = n_global - log_world
        full = [oracle.random_b128(0xB1A50000 + j, 1 << n_global) for j in range(m)]
        idx = shard_indices(1 << n_global, world, rank)
        local = [np.ascontiguousarray(x[idx]) for x in full]
        stream = oracle.random_scalars(0xC4A1, n_global + 1)
        batch_coeff, challenges = stream[0], stream[1:]
        sums = [oracle.inner_product(full[i], 7, full[j])[1] for i, j in comps]
        if use_shm:
            from binius_amd._host import ShmExchange

            comm = ShmExchange(dist, rank, world)  # the exchange bench.py uses on a single node
        else:
            comm = TorchComm(dist, world)
        prover = ShardedBivariateSumcheck(comm, OracleCompute(oracle, local, comps, batch_coeff), OracleField(oracle), n_local, world, len(comps))
        coeffs, finals = prover.prove(sums, batch_coeff, challenges)
        want_coeffs, want_finals = oracle.bivariate_sumcheck_prove([x.copy() for x in full], n_global, comps, sums, batch_coeff, challenges)
        ok = coeffs == want_coeffs and finals == want_finals
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,n_global,use_shm", [(2, 9, False), (4, 10, False), (2, 9, True), (4, 8, True)])
def test_sharded_sumcheck_gloo(world, n_global, use_shm):
    import torch.multiprocessing as mp

    import oracle

    oracle.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_global, q, use_shm)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    results = dict(q.get(timeout=10) for _ in range(world))
    assert results == {r: True for r in range(world)}


def test_shard_layout_keeps_pairs_local():
    """Under High-to-Low binding round r pairs global indices (i, i + N/2); with the device id in
    the low bits both live on the same rank at adjacent-half local indices, for every round."""
    sys.path.insert(0, ROOT)
    from binius_amd.distributed import shard_indices

    n_global, world = 8, 4
    N = 1 << n_global
    for rank in range(world):
        idx = shard_indices(N, world, rank)
        local_n = len(idx)
        size = N
        while size > world:
            half_local = (size // world) // 2
            for li in range(half_local):
                gi = idx[li]
                assert gi + size // 2 == idx[li + half_local]
            size //= 2
        assert local_n == N // world


def _shm_worker(rank, world, port, q):
    import os

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from binius_amd._host import ShmExchange

        ex = ShmExchange(dist, rank, world)
        ok = True
        acc = 0
        for r in range(2000):  # many rounds: exercises the two-slot reuse
            mine = [(rank + 1) * 0x9E3779B97F4A7C15 * (r + 1) & ((1 << 128) - 1), (r << 64) | rank]
            got = ex.all_gather_scalars(mine)
            for w in range(world):
                want = [(w + 1) * 0x9E3779B97F4A7C15 * (r + 1) & ((1 << 128) - 1), (r << 64) | w]
                ok = ok and got[w] == want
            acc ^= ex.xor_scalars([rank + r])[0]
        ex.close()
        q.put((rank, ok, acc))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_shared_memory_exchange(world):
    """The intra-node exchange used by the sharded prover (bnh_shm_*): every rank sees every rank's
    words, round after round, with two slots per rank."""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_shm_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in ps:
        p.join(timeout=60)
    assert all(ok for _, ok, _ in res)
    want = 0
    for r in range(2000):
        x = 0
        for w in range(world):
            x ^= w + r
        want ^= x
    assert all(acc == want for _, _, acc in res)


def test_bench_refuses_a_world_size_that_disagrees_with_gpus():
    """VERDICT r4 item 3a (the half that needs no GPU): bench.py under a launcher whose WORLD_SIZE is not --gpus exits with
    an error before it touches a device -- it never prints a line labelled with the wrong number of GPUs."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.update({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "refusing" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
