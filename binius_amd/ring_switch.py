"""Host-side mirror of the ring-switching equality indicator (the multilinear A of DP24 section 5) on the
ComputeLayer: crates/core/src/ring_switch/eq_ind.rs:39-141 (RingSwitchEqInd::new / precompute_values /
multilinear_extension) -- the caller of tensor_expand and fold_right in the PCS prover
(crates/core/src/ring_switch/prove.rs)."""
import numpy as np

from ._ffi import BN_ERR_INPUT_VALIDATION, BnError


class RingSwitchEqInd:
    """eq_ind.rs:39-50.  `kappa` = log2 of the extension degree of F over the packed subfield
    (tower_level = 7 - kappa): the evaluations are folded over their 2^kappa subfield limbs."""

    def __init__(self, z_vals, row_batch_coeffs, mixing_coeff, kappa):
        if len(row_batch_coeffs) < (1 << kappa):  # (:63-69)
            raise BnError(BN_ERR_INPUT_VALIDATION, "InvalidArgs(RingSwitchEqInd::new expects row_batch_coeffs length greater than or equal to the extension degree)")
        self.z_vals, self.row_batch_coeffs, self.mixing_coeff, self.kappa = list(z_vals), list(row_batch_coeffs), mixing_coeff, kappa

    def precompute_values(self, hal, dev_alloc):
        """eq_ind.rs:78-121: the three device buffers."""
        deg = 1 << self.kappa
        expansion = dev_alloc.alloc(deg)
        coeffs = np.zeros((deg, 2), dtype=np.uint64)
        for i, c in enumerate(self.row_batch_coeffs[:deg]):
            coeffs[i, 0], coeffs[i, 1] = c & ((1 << 64) - 1), c >> 64
        hal.copy_h2d(coeffs, expansion)
        evals = dev_alloc.alloc(1 << len(self.z_vals))
        hal.fill(evals, 0)  # (the reference's allocator hands out zeroed memory for the part tensor_expand grows into)
        hal.fill(evals.slice(0, 1), self.mixing_coeff)
        mle = dev_alloc.alloc(evals.len)
        return evals, expansion, mle

    def multilinear_extension(self, hal, precompute):
        """eq_ind.rs:123-141: tensor_expand(0, z_vals) then fold_right over the subfield limbs."""
        evals, expansion, mle = precompute
        hal.tensor_expand(0, self.z_vals, evals)
        hal.fold_right(evals, 7 - self.kappa, expansion, mle)
        return mle
