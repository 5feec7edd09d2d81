"""Multi-GPU sumcheck: one process per GPU, torch.distributed over RCCL (backend "nccl").

Sharding: the v3 provers bind High-to-Low, so the pair (i, i + N/2) a round
touches stays device-local as long as the device id is taken from the variables bound LAST -- the
low log2(G) index bits.  Rank g holds the elements with global index = g (mod G) as one contiguous
local array; round evaluation is a sum over hypercube points, so each rank produces a partial
(y_1, y_inf) over its shard and the G partials are combined with one collective per round; fold
is element-wise and needs no communication.

RCCL has no XOR reduction (ncclSum on integers is the wrong field), so the combine is
all_gather(32 bytes per rank) + XOR.  The payload is latency-only (a few hundred bytes over xGMI).
"""
import numpy as np

from .sumcheck import round_eval_kernel


class ShardedRoundReducer:
    def __init__(self, hal, dist, world):
        import torch

        self.hal = hal
        self.dist = dist
        self.world = world
        self.torch = torch
        dev = torch.device("cuda", torch.cuda.current_device())
        self.local = torch.zeros(4, dtype=torch.int64, device=dev)          # (y_1, y_inf) = 2 x u128
        self.gathered = torch.zeros(4 * world, dtype=torch.int64, device=dev)
        self.scalar_local = torch.zeros(2, dtype=torch.int64, device=dev)
        self.scalar_gathered = torch.zeros(2 * world, dtype=torch.int64, device=dev)

    @staticmethod
    def _fold_xor(host, n_scalars, world):
        u = host.view(np.uint64).reshape(world, n_scalars, 2)
        x = np.bitwise_xor.reduce(u, axis=0)
        return [int(x[i, 0]) | (int(x[i, 1]) << 64) for i in range(n_scalars)]

    def round_evals(self, n_vars, multilins, expr):
        """Partial round evals stay on the device, one RCCL all_gather, XOR on the host."""
        kernel, maps = round_eval_kernel(n_vars, [1], multilins, [expr])
        ops, rets, lc = self.hal.record(kernel, maps)
        self.hal.kernel_launch(maps, ops, rets, lc, d_out=self.local.data_ptr(), want_host=False)
        self.dist.all_gather_into_tensor(self.gathered, self.local)
        host = self.gathered.cpu().numpy()
        return self._fold_xor(host, 2, self.world)

    def gather_local(self):
        """Combine the partial (y_1, y_inf) a kernel just left in `self.local` (same stream)."""
        self.dist.all_gather_into_tensor(self.gathered, self.local)
        return self._fold_xor(self.gathered.cpu().numpy(), 2, self.world)

    def xor_scalars(self, scalars):
        """XOR-combine one field element per rank (e.g. the claimed sum)."""
        out = []
        for s in scalars:
            v = np.array([s & ((1 << 64) - 1), s >> 64], dtype=np.uint64).view(np.int64)
            self.scalar_local.copy_(self.torch.from_numpy(v))
            self.dist.all_gather_into_tensor(self.scalar_gathered, self.scalar_local)
            out.append(self._fold_xor(self.scalar_gathered.cpu().numpy(), 1, self.world)[0])
        return out
