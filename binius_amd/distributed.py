"""Multi-GPU sumcheck: one process per GPU, torch.distributed over RCCL (backend "nccl").

Sharding.  The v3 provers bind High-to-Low, so the pair (i, i + N/2) a
round touches stays device-local exactly when the device id is taken from the variables bound
LAST -- the low log2(G) index bits.  Rank g holds the elements with global index = g (mod G) as
one contiguous local array (local index = global index >> log2 G).

  rounds 0 .. n_local-1   every rank runs the same round on its shard; round evaluation is a sum
                          over hypercube points, so the G partial (y_1, y_inf) pairs are combined
                          with ONE collective per round; fold is element-wise, no communication.
  rounds n_local .. n-1   each rank is down to one element per multilinear; one all_gather of m
                          elements rebuilds the 2^(log2 G)-element residual multilinears (index =
                          rank) and the last log2(G) rounds run on that tiny instance.

RCCL has no XOR reduction (ncclSum on integers is the wrong field), so the combine is
all_gather(32 bytes per rank) + XOR.  The payload is latency-only over xGMI.

The class is backend-agnostic: `compute` does the per-shard field work (the HIP backend in
production and in bench.py; the CPU oracle in the gloo tests), `comm` moves 128-bit scalars.
"""
import numpy as np

MASK64 = (1 << 64) - 1


def shard_indices(n_global, world, rank):
    """Global indices owned by `rank`, in local order."""
    return np.arange(rank, n_global, world)


def _to_i64(vals):
    a = np.zeros(2 * len(vals), dtype=np.uint64)
    for i, v in enumerate(vals):
        a[2 * i] = v & MASK64
        a[2 * i + 1] = v >> 64
    return a.view(np.int64)


def _from_i64(arr, world, n_scalars):
    u = np.ascontiguousarray(arr).view(np.uint64).reshape(world, n_scalars, 2)
    return [[int(u[g, i, 0]) | (int(u[g, i, 1]) << 64) for i in range(n_scalars)] for g in range(world)]


class TorchComm:
    """128-bit scalars over torch.distributed.  device=None: CPU tensors (gloo); otherwise CUDA
    tensors on `device` (nccl = RCCL)."""

    def __init__(self, dist, world, device=None):
        import torch

        self.dist, self.world, self.torch, self.device = dist, world, torch, device

    def all_gather_scalars(self, vals):
        """Every rank contributes len(vals) field elements; returns [rank][i]."""
        t = self.torch.from_numpy(_to_i64(vals).copy())
        if self.device is not None:
            t = t.to(self.device)
        out = self.torch.zeros(t.numel() * self.world, dtype=self.torch.int64, device=t.device)
        self.dist.all_gather_into_tensor(out, t)
        return _from_i64(out.cpu().numpy(), self.world, len(vals))

    def xor_scalars(self, vals):
        per_rank = self.all_gather_scalars(vals)
        out = [0] * len(vals)
        for r in per_rank:
            for i, v in enumerate(r):
                out[i] ^= v
        return out


class ShardedBivariateSumcheck:
    """BivariateSumcheckProver (v3/bivariate_product.rs:27-254) over G shards.

    compute must provide:
      round_evals(n_vars_remaining) -> [y_1, y_inf] partials over this rank's shard (batch coeffs applied)
      fold(challenge)               -> folds every local multilinear once
      finals()                      -> the m fully folded local values (after n_local rounds)
      tail_prove(residual_multilins[m][G], n_tail, running_sum, batch_coeff, challenges)
                                    -> (round_coeffs, final_evals) of the last log2(G) rounds
    field.mul(a, b) handles the O(1) protocol scalars.
    """

    def __init__(self, comm, compute, field, n_local, world, n_comps=1):
        self.comm, self.compute, self.field = comm, compute, field
        self.n_local, self.world = n_local, world
        self.log_world = world.bit_length() - 1
        assert 1 << self.log_world == world, "number of shards must be a power of two"
        self.n_comps = n_comps

    def _eval_univariate(self, coeffs, x):
        e = 0
        for c in reversed(coeffs):
            e = self.field.mul(e, x) ^ c
        return e

    def prove(self, sums, batch_coeff, challenges):
        assert len(challenges) == self.n_local + self.log_world
        running = self._eval_univariate(list(sums), batch_coeff)
        round_coeffs = []
        for r in range(self.n_local):
            part = self.compute.round_evals(self.n_local - r)
            y_1, y_inf = self.comm.xor_scalars(part) if self.world > 1 else part
            c_0 = running ^ y_1
            c_2 = y_inf
            c_1 = y_1 ^ c_0 ^ c_2
            round_coeffs.append([c_0, c_1, c_2])
            running = self._eval_univariate([c_0, c_1, c_2], challenges[r])
            self.compute.fold(challenges[r])
        finals = self.compute.finals()
        if self.world == 1:
            return round_coeffs, finals
        per_rank = self.comm.all_gather_scalars(finals)  # [rank][multilinear]
        residual = [[per_rank[g][j] for g in range(self.world)] for j in range(len(finals))]
        tail_coeffs, tail_finals = self.compute.tail_prove(
            residual, self.log_world, running, batch_coeff, challenges[self.n_local :]
        )
        return round_coeffs + tail_coeffs, tail_finals


class ShardedRoundReducer:
    """Device-resident variant used by bench.py: the round-eval kernel leaves its partial
    (y_1, y_inf) in a CUDA tensor (bn_kernel_launch d_out), one RCCL all_gather on the same stream,
    XOR on the host."""

    def __init__(self, hal, dist, world):
        import torch

        self.hal, self.dist, self.world, self.torch = hal, dist, world, torch
        dev = torch.device("cuda", torch.cuda.current_device())
        self.local = torch.zeros(4, dtype=torch.int64, device=dev)  # (y_1, y_inf) = 2 x u128
        self.gathered = torch.zeros(4 * world, dtype=torch.int64, device=dev)
        self.comm = TorchComm(dist, world, dev)

    def gather_local(self):
        """Combine the partial (y_1, y_inf) a kernel just left in `self.local` (same stream)."""
        self.dist.all_gather_into_tensor(self.gathered, self.local)
        per_rank = _from_i64(self.gathered.cpu().numpy(), self.world, 2)
        out = [0, 0]
        for r in per_rank:
            out[0] ^= r[0]
            out[1] ^= r[1]
        return out

    def xor_scalars(self, scalars):
        return self.comm.xor_scalars(list(scalars))

    def all_gather_scalars(self, scalars):
        return self.comm.all_gather_scalars(list(scalars))
