"""Host-side mirror of the reference's product-tree layer computation over a ComputeLayer:

  ProductCircuitLayers::compute     crates/core/src/protocols/prodcheck/prove.rs:24-77

Layer i of the binary product circuit has 2^i values; each layer is the element-wise product of the
two halves of the layer below (`compute_composite` with the BivariateProduct composition)."""
from ._ffi import BN_ERR_INPUT_VALIDATION, BnError
from .sumcheck import bivariate_product_expr


class ProductCircuitLayers:
    def __init__(self, layers, product):
        self._layers, self.product = layers, product

    @classmethod
    def compute(cls, evals, hal, dev_alloc):
        """evals: device slice of 2^log_n values.  Returns the layers (entry i has 2^(i+1) values, the
        last one is `evals` itself) and the product of all values."""
        n = evals.len
        if n == 0 or n & (n - 1):
            raise BnError(BN_ERR_INPUT_VALIDATION, "ExpectInputSlicePowerOfTwoLength")
        log_n = n.bit_length() - 1
        prod_expr = bivariate_product_expr(hal, 0, 1)
        last_layer = evals
        layers = []
        for i in reversed(range(log_n)):
            row_len = 1 << i
            lo_half, hi_half = last_layer.split_half()
            new_layer = dev_alloc.alloc(row_len)
            hal.compute_composite([lo_half, hi_half], new_layer, prod_expr)
            layers.append(last_layer)
            last_layer = new_layer
        h = hal.copy_d2h(last_layer)
        product = int(h[0, 0]) | (int(h[0, 1]) << 64)
        layers.reverse()
        return cls(layers, product)

    def layers(self):
        return self._layers
