// The reference's backend-conformance suite (crates/compute_test_utils/src/layer.rs), instantiated for the
// MI355X layer the way crates/fast_compute/tests/layer.rs instantiates it for FastCpuLayer.  Needs a GPU and
// libbinius_amd.so (BINIUS_AMD_LIB_DIR).

use binius_compute_test_utils::layer::{
	test_extrapolate_line, test_generic_compute_composite, test_generic_fri_fold, test_generic_kernel_add,
	test_generic_map_with_multilinear_evaluations, test_generic_multiple_multilinear_evaluations, test_generic_pairwise_product_reduce,
	test_generic_single_inner_product, test_generic_single_inner_product_using_kernel_accumulator, test_generic_single_left_fold,
	test_generic_single_right_fold, test_generic_single_tensor_expand, test_map_kernels,
};
use binius_field::{BinaryField16b, BinaryField32b, BinaryField128b};
use binius_mi355x::Mi355xLayerHolder;

type F = BinaryField128b;

#[test]
fn test_exec_single_tensor_expand() {
	let n_vars = 8;
	test_generic_single_tensor_expand::<F, _, _>(Mi355xLayerHolder::new(1 << (n_vars + 1), 1 << n_vars), n_vars);
}

#[test]
fn test_exec_single_left_fold() {
	let n_vars = 8;
	test_generic_single_left_fold::<BinaryField16b, F, _, _>(Mi355xLayerHolder::new(1 << (n_vars + 1), 1 << n_vars), n_vars / 2, n_vars / 8);
}

#[test]
fn test_exec_single_right_fold() {
	let n_vars = 8;
	test_generic_single_right_fold::<BinaryField16b, F, _, _>(Mi355xLayerHolder::new(1 << (n_vars + 1), 1 << n_vars), n_vars / 2, n_vars / 8);
}

#[test]
fn test_exec_single_inner_product() {
	let n_vars = 8;
	test_generic_single_inner_product::<BinaryField16b, F, _, _>(Mi355xLayerHolder::new(1 << (n_vars + 2), 1 << (n_vars + 1)), n_vars);
}

#[test]
fn test_exec_multiple_multilinear_evaluations() {
	let n_vars = 8;
	test_generic_multiple_multilinear_evaluations::<BinaryField16b, BinaryField32b, F, _, _>(
		Mi355xLayerHolder::new(1 << (n_vars + 2), 1 << (n_vars + 2)),
		n_vars,
	);
}

#[test]
fn test_exec_map_with_multilinear_evaluations() {
	let n_vars = 8;
	test_generic_map_with_multilinear_evaluations::<F, _, _>(Mi355xLayerHolder::new(3 << n_vars, 3 << (n_vars + 1)), n_vars);
}

#[test]
fn test_exec_single_inner_product_using_kernel_accumulator() {
	let n_vars = 8;
	test_generic_single_inner_product_using_kernel_accumulator::<F, _, _>(Mi355xLayerHolder::new(1 << (n_vars + 2), 1 << (n_vars + 1)), n_vars);
}

#[test]
fn test_exec_fri_fold_non_zero_log_batch() {
	let (log_len, log_batch_size, log_fold_challenges) = (10, 4, 2);
	test_generic_fri_fold::<F, BinaryField16b, _, _>(
		Mi355xLayerHolder::new(1 << (log_len + log_batch_size + 2), 1 << (log_len + log_batch_size + 1)),
		log_len,
		log_batch_size,
		log_fold_challenges,
	);
}

#[test]
fn test_exec_fri_fold_zero_log_batch() {
	let (log_len, log_batch_size, log_fold_challenges) = (10, 0, 2);
	test_generic_fri_fold::<F, BinaryField16b, _, _>(
		Mi355xLayerHolder::new(1 << (log_len + log_batch_size + 2), 1 << (log_len + log_batch_size + 1)),
		log_len,
		log_batch_size,
		log_fold_challenges,
	);
}

#[test]
fn test_exec_kernel_add() {
	let log_len = 10;
	test_generic_kernel_add::<F, _, _>(Mi355xLayerHolder::new(1 << (log_len + 4), 1 << (log_len + 3)), log_len);
}

#[test]
fn test_exec_extrapolate_line() {
	let log_len = 10;
	test_extrapolate_line::<F, _, _>(Mi355xLayerHolder::new(1 << (log_len + 4), 1 << (log_len + 3)), log_len);
}

#[test]
fn test_exec_compute_composite() {
	let log_len = 10;
	test_generic_compute_composite::<F, _, _>(Mi355xLayerHolder::new(1 << (log_len + 4), 1 << (log_len + 3)), log_len);
}

#[test]
fn test_exec_map_kernels() {
	let log_len = 10;
	test_map_kernels::<F, _, _>(Mi355xLayerHolder::new(1 << (log_len + 4), 1 << (log_len + 3)), log_len);
}

#[test]
fn test_exec_pairwise_product_reduce() {
	let log_len = 8;
	test_generic_pairwise_product_reduce::<F, _, _>(Mi355xLayerHolder::new(1 << (log_len + 4), 1 << (log_len + 3)), log_len);
}
