"""Kernel-recording helpers of the ctypes binding: the shapes of work the reference's provers hand to a ComputeLayer,
expressed once so that tests, tools and bench.py issue exactly what the Rust host would.

  bivariate_product_expr    IndexComposition<BivariateProduct, 2>::expression (product_composition.rs:30-32)
  round_eval_kernel /
  calculate_round_evals     the accumulate_kernels closure of v3/bivariate_product.rs:303-408
                            (with eq_ind: v3/bivariate_mlecheck.rs:391-520)
  eq_ind_partial_eval       crates/compute/src/ops.rs:26-50

The provers themselves (execute / fold / finish loops, FRI, Merkle, prodcheck, ring switching) are mirrored ONCE, in
C++ (binius_amd/host/*.hpp), and reached from Python through binius_amd._host (SumcheckPlan, MlecheckPlan, FriPlan).
"""
def bivariate_product_expr(hal, i, j):
    """CompositionPoly::expression of IndexComposition<BivariateProduct, 2> {indices [i, j]}
    (crates/core/src/composition/product_composition.rs:30-32 remapped by index.rs:50-55)."""
    return hal.compile_expr([("var", i), ("var", j), ("mul", 0, 1)])


def round_eval_kernel(n_vars, batch_coeffs, multilins, compositions, eq_ind=None):
    """The kernel-spec closure and memory mappings of calculate_round_evals
    (v3/bivariate_product.rs:317-402; with eq_ind: v3/bivariate_mlecheck.rs:409-515)."""
    split_n_vars = n_vars - 1
    mem_maps = []
    for ml in multilins:
        lo, hi = ml.split_half()
        mem_maps += [("chunked", lo, 0), ("chunked", hi, 0), ("local", split_n_vars)]
    if eq_ind is not None:
        mem_maps.append(("chunked", eq_ind, 0))
    m = len(multilins)

    def kernel(local_exec, log_chunks, buffers):
        log_chunk_size = split_n_vars - log_chunks
        eq = buffers[-1] if eq_ind is not None else None
        acc_1 = local_exec.decl_value(0)
        eval_1s = [buffers[i * 3 + 1].to_ref() for i in range(m)]
        if eq is not None:
            eval_1s.append(eq.to_ref())
        for coeff, evaluator in zip(batch_coeffs, compositions):
            local_exec.sum_composition_evals(eval_1s, evaluator, coeff, acc_1)
        for i in range(m):
            local_exec.add(log_chunk_size, buffers[3 * i], buffers[3 * i + 1], buffers[3 * i + 2])
        acc_inf = local_exec.decl_value(0)
        eval_infs = [buffers[i * 3 + 2].to_ref() for i in range(m)]
        if eq is not None:
            eval_infs.append(eq.to_ref())
        for coeff, evaluator in zip(batch_coeffs, compositions):
            local_exec.sum_composition_evals(eval_infs, evaluator, coeff, acc_inf)
        return [acc_1, acc_inf]

    return kernel, mem_maps


def calculate_round_evals(hal, n_vars, batch_coeffs, multilins, compositions, eq_ind=None):
    """Returns [y_1, y_inf].  `compositions` are compiled exprs over the m (+1 with eq_ind) rows;
    `batch_coeffs[c]` = batch_coeff ** c (powers(), computed by the caller)."""
    kernel, mem_maps = round_eval_kernel(n_vars, batch_coeffs, multilins, compositions, eq_ind)
    return hal.accumulate_kernels(kernel, mem_maps)


def eq_ind_partial_eval(hal, dev_alloc, point):
    n_vars = len(point)
    out = dev_alloc.alloc(1 << n_vars)
    hal.fill(out.slice(0, 1), 1)
    hal.tensor_expand(0, point, out)
    return out


def bivariate_product_eq_expr(hal, i, j, m):
    """IndexComposition<BivariateProduct, 2>::expression() * var(m) -- the eq indicator is the last
    composition variable (v3/bivariate_mlecheck.rs:399-407)."""
    return hal.compile_expr([("var", i), ("var", j), ("mul", 0, 1), ("var", m), ("mul", 2, 3)])
