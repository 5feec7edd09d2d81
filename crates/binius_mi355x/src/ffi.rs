//! Raw bindings of `include/binius_amd.h` -- one declaration per C entry point, same names, same
//! argument order, same widths (`tests/test_rust_shim_ffi.py` in the backend repository parses this block
//! and the header and compares them).
#![allow(non_camel_case_types)]

use std::os::raw::{c_char, c_int, c_void};

/// `bn_f128`: one little-endian u128 = `BinaryField128b` (crates/field/src/binary_field.rs:747).
#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct bn_f128 {
	pub lo: u64,
	pub hi: u64,
}

pub const BN_OK: c_int = 0;
pub const BN_ERR_INPUT_VALIDATION: c_int = 1;
pub const BN_ERR_ALLOC: c_int = 2;
pub const BN_ERR_DEVICE: c_int = 3;
pub const BN_ERR_CORE_LIB: c_int = 4;

/// Opaque context (stream, scratch, optional arena).
#[repr(C)]
pub struct bn_ctx {
	_private: [u8; 0],
}
/// Opaque compiled arithmetic circuit.
#[repr(C)]
pub struct bn_expr {
	_private: [u8; 0],
}

pub const BN_STEP_ADD: u32 = 0;
pub const BN_STEP_MUL: u32 = 1;
pub const BN_STEP_POW: u32 = 2;
pub const BN_STEP_CONST: u32 = 3;
pub const BN_STEP_VAR: u32 = 4;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct bn_step {
	pub kind: u32,
	pub a: u32,
	pub b: u64,
	pub cst: bn_f128,
}

pub const BN_NTT_MAX_DIM: usize = 64;

pub const BN_MAP_CHUNKED: u32 = 0;
pub const BN_MAP_CHUNKED_MUT: u32 = 1;
pub const BN_MAP_LOCAL: u32 = 2;

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct bn_memmap {
	pub kind: u32,
	pub log_min_chunk_size: u32,
	pub d_data: *mut c_void,
	pub len: u64,
	pub log_size: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct bn_kslice {
	pub buf: u32,
	pub off: u64,
	pub len: u64,
}

pub const BN_KOP_DECL_VALUE: u32 = 0;
pub const BN_KOP_SUM_COMPOSITION: u32 = 1;
pub const BN_KOP_ADD: u32 = 2;
pub const BN_KOP_ADD_ASSIGN: u32 = 3;

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct bn_kop {
	pub kind: u32,
	pub value: u32,
	pub scalar: bn_f128,
	pub expr: *const bn_expr,
	pub n_rows: u32,
	pub rows: *const bn_kslice,
	pub src1: bn_kslice,
	pub src2: bn_kslice,
	pub dst: bn_kslice,
}

pub const BN_PROF_N: usize = 11;
pub const BN_ARM_N: usize = 15;

// ---- the old HAL (binius_hal::ComputationBackend) on device-resident multilinears
pub const BN_ORDER_LOW_TO_HIGH: u32 = 0;
pub const BN_ORDER_HIGH_TO_LOW: u32 = 1;
pub const BN_HAL_ML_FOLDED: u32 = 0;
pub const BN_HAL_ML_TRANSPARENT: u32 = 1;

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct bn_hal_multilinear {
	pub kind: u32,
	pub tower_level: u32,
	pub d_evals: *const c_void,
	pub len: u64,
	pub suffix_eval: bn_f128,
	pub n_vars_ml: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct bn_hal_evaluator {
	pub composition: *const bn_expr,
	pub composition_at_infinity: *const bn_expr,
	pub eval_point_start: u32,
	pub eval_point_end: u32,
	pub d_eq_ind: *const c_void,
}

unsafe extern "C" {
	pub fn bn_last_error() -> *const c_char;
	pub fn bn_version() -> *const c_char;

	pub fn bn_ctx_create(device: c_int, arena_elems: u64, out: *mut *mut bn_ctx) -> c_int;
	pub fn bn_ctx_destroy(ctx: *mut bn_ctx) -> c_int;
	pub fn bn_arena_base(ctx: *mut bn_ctx, d_base: *mut *mut c_void, elems: *mut u64) -> c_int;
	pub fn bn_ctx_set_stream(ctx: *mut bn_ctx, hip_stream: *mut c_void) -> c_int;
	pub fn bn_sync(ctx: *mut bn_ctx) -> c_int;
	pub fn bn_ctx_get_stream(ctx: *mut bn_ctx, hip_stream: *mut *mut c_void) -> c_int;

	pub fn bn_copy_h2d(ctx: *mut bn_ctx, h_src: *const bn_f128, src_len: u64, d_dst: *mut c_void, dst_len: u64) -> c_int;
	pub fn bn_copy_d2h(ctx: *mut bn_ctx, d_src: *const c_void, src_len: u64, h_dst: *mut bn_f128, dst_len: u64) -> c_int;
	pub fn bn_copy_d2d(ctx: *mut bn_ctx, d_src: *const c_void, src_len: u64, d_dst: *mut c_void, dst_len: u64) -> c_int;
	pub fn bn_fill(ctx: *mut bn_ctx, d_dst: *mut c_void, n: u64, value: *const bn_f128) -> c_int;

	pub fn bn_expr_compile(ctx: *mut bn_ctx, steps: *const bn_step, n_steps: u64, out: *mut *mut bn_expr) -> c_int;
	pub fn bn_expr_free(expr: *mut bn_expr) -> c_int;
	pub fn bn_expr_n_vars(expr: *const bn_expr, n_vars: *mut u32) -> c_int;

	pub fn bn_extrapolate_line(ctx: *mut bn_ctx, d_evals_0: *mut c_void, n0: u64, d_evals_1: *const c_void, n1: u64, z: *const bn_f128) -> c_int;
	pub fn bn_extrapolate_line_batch(
		ctx: *mut bn_ctx,
		d_evals_0: *const *mut c_void,
		d_evals_1: *const *const c_void,
		count: u32,
		n: u64,
		z: *const bn_f128,
	) -> c_int;
	pub fn bn_extrapolate_line_batch_scaled(
		ctx: *mut bn_ctx,
		d_evals_0: *const *mut c_void,
		d_evals_1: *const *const c_void,
		count: u32,
		n: u64,
		z: *const bn_f128,
		scale_mask: u32,
		hi_scale: *const bn_f128,
	) -> c_int;
	pub fn bn_tensor_expand(ctx: *mut bn_ctx, d_data: *mut c_void, data_len: u64, log_n: u32, h_coords: *const bn_f128, k: u32) -> c_int;
	pub fn bn_inner_product(
		ctx: *mut bn_ctx,
		d_a: *const c_void,
		a_len: u64,
		tower_level: u32,
		d_b: *const c_void,
		b_len: u64,
		h_out: *mut bn_f128,
	) -> c_int;
	pub fn bn_fold_left(
		ctx: *mut bn_ctx,
		d_mat: *const c_void,
		mat_len: u64,
		tower_level: u32,
		d_vec: *const c_void,
		vec_len: u64,
		d_out: *mut c_void,
		out_len: u64,
	) -> c_int;
	pub fn bn_fold_right(
		ctx: *mut bn_ctx,
		d_mat: *const c_void,
		mat_len: u64,
		tower_level: u32,
		d_vec: *const c_void,
		vec_len: u64,
		d_out: *mut c_void,
		out_len: u64,
	) -> c_int;
	pub fn bn_fri_fold(
		ctx: *mut bn_ctx,
		h_s_evals: *const u64,
		tw_level: u32,
		log_domain: u32,
		log_len: u32,
		log_batch_size: u32,
		h_challenges: *const bn_f128,
		n_challenges: u32,
		d_in: *const c_void,
		in_len: u64,
		d_out: *mut c_void,
		out_len: u64,
	) -> c_int;
	pub fn bn_compute_composite(
		ctx: *mut bn_ctx,
		d_rows: *const *const c_void,
		n_rows: u32,
		row_len: u64,
		d_out: *mut c_void,
		out_len: u64,
		expr: *const bn_expr,
	) -> c_int;
	pub fn bn_pairwise_product_reduce(
		ctx: *mut bn_ctx,
		d_in: *const c_void,
		n: u64,
		d_round_outs: *const *mut c_void,
		round_lens: *const u64,
		n_rounds: u32,
	) -> c_int;

	pub fn bn_log_chunks_range(maps: *const bn_memmap, n_maps: u32, start: *mut u32, end: *mut u32) -> c_int;
	pub fn bn_pick_log_chunks(maps: *const bn_memmap, n_maps: u32, log_chunks: *mut u32) -> c_int;
	pub fn bn_kernel_launch(
		ctx: *mut bn_ctx,
		maps: *const bn_memmap,
		n_maps: u32,
		ops: *const bn_kop,
		n_ops: u32,
		ret_values: *const u32,
		n_ret: u32,
		log_chunks: u32,
		h_out: *mut bn_f128,
		d_out: *mut c_void,
	) -> c_int;

	pub fn bn_hal_round_evals(
		ctx: *mut bn_ctx,
		order: u32,
		n_vars: u32,
		d_tensor_query: *const c_void,
		query_vars: u32,
		mls: *const bn_hal_multilinear,
		n_mls: u32,
		evaluators: *const bn_hal_evaluator,
		n_evaluators: u32,
		h_nontrivial_points: *const bn_f128,
		n_points: u32,
		h_out: *mut bn_f128,
	) -> c_int;
	pub fn bn_hal_fold_multilinear(
		ctx: *mut bn_ctx,
		order: u32,
		n_vars: u32,
		ml: *const bn_hal_multilinear,
		challenge: *const bn_f128,
		d_tensor_query: *const c_void,
		query_vars: u32,
		d_out: *mut c_void,
		out_cap: u64,
		out_len: *mut u64,
	) -> c_int;

	pub fn bn_ntt_forward(
		ctx: *mut bn_ctx,
		d_data: *mut c_void,
		elem_level: u32,
		tw_level: u32,
		h_s_evals: *const u64,
		log_domain: u32,
		log_x: u32,
		log_y: u32,
		log_z: u32,
		coset: u64,
		coset_bits: u32,
		skip_rounds: u32,
	) -> c_int;
	pub fn bn_ntt_inverse(
		ctx: *mut bn_ctx,
		d_data: *mut c_void,
		elem_level: u32,
		tw_level: u32,
		h_s_evals: *const u64,
		log_domain: u32,
		log_x: u32,
		log_y: u32,
		log_z: u32,
		coset: u64,
		coset_bits: u32,
		skip_rounds: u32,
	) -> c_int;
	pub fn bn_ntt_s_evals(tw_level: u32, log_domain: u32, h_s_evals: *mut u64) -> c_int;

	pub fn bn_scalar_mul(a: *const bn_f128, b: *const bn_f128, out: *mut bn_f128) -> c_int;
	pub fn bn_scalar_invert(a: *const bn_f128, out: *mut bn_f128) -> c_int;

	pub fn bn_groestl256_leaves(ctx: *mut bn_ctx, d_elems: *const c_void, n_elems: u64, batch_size: u64, d_digests: *mut c_void) -> c_int;
	pub fn bn_groestl256_compress_layer(ctx: *mut bn_ctx, d_prev: *const c_void, n_out: u64, d_next: *mut c_void) -> c_int;
	pub fn bn_merkle_build(ctx: *mut bn_ctx, d_elems: *const c_void, n_elems: u64, batch_size: u64, d_nodes: *mut c_void) -> c_int;
	pub fn bn_gather_d2h(ctx: *mut bn_ctx, d_src: *const c_void, h_offsets: *const u64, n_items: u64, item_elems: u64, h_out: *mut bn_f128) -> c_int;

	pub fn bn_timer_begin(ctx: *mut bn_ctx) -> c_int;
	pub fn bn_timer_end_ms(ctx: *mut bn_ctx, ms: *mut f32) -> c_int;
	pub fn bn_host_scratch(ctx: *mut bn_ctx, h_ptr: *mut *mut c_void, d_ptr: *mut *mut c_void, elems: *mut u64) -> c_int;
	pub fn bn_device_numa_node(device: c_int, node: *mut c_int) -> c_int;
	pub fn bn_xor_reduce(ctx: *mut bn_ctx, d_vals: *const c_void, n_groups: u32, group_len: u32, h_out: *mut bn_f128) -> c_int;
	pub fn bn_prof_begin(ctx: *mut bn_ctx) -> c_int;
	pub fn bn_prof_end(ctx: *mut bn_ctx, ms_by_class: *mut f64, launches_by_class: *mut u64) -> c_int;
	pub fn bn_arm_counters(ctx: *mut bn_ctx, counters: *mut u64) -> c_int;
	pub fn bn_group_counters(ctx: *mut bn_ctx, counters: *mut u64) -> c_int;
	// cross-rank reduction of the round evaluations inside the kernels' finalize step (one process per GPU)
	pub fn bn_peer_create(ctx: *mut bn_ctx, world: u32, rank: u32, handle_out: *mut u8) -> c_int;
	pub fn bn_peer_connect(ctx: *mut bn_ctx, handles: *const u8) -> c_int;
	pub fn bn_peer_set_active(ctx: *mut bn_ctx, on: c_int) -> c_int;
	pub fn bn_peer_stats(ctx: *mut bn_ctx, stats: *mut u64) -> c_int;
	pub fn bn_host_tail_allow_peer(ctx: *mut bn_ctx, on: c_int) -> c_int;
	pub fn bn_host_tail_active(ctx: *mut bn_ctx, active: *mut c_int) -> c_int;
	pub fn bn_peer_destroy(ctx: *mut bn_ctx) -> c_int;
}
