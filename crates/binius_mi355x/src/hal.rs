//! The OLD hardware abstraction layer (`binius_hal::ComputationBackend`, crates/hal/src/backend.rs:35-84) on
//! device-resident multilinears: safe wrappers over `bn_hal_round_evals` / `bn_hal_fold_multilinear`.
//!
//! `ComputationBackend` itself cannot be implemented for a device without touching the trait: its `Vcs<P>` must
//! deref to a host slice (`HalSlice<P>: Deref<Target = [P]>`, backend.rs:23-33) and the multilinears arrive as
//! `M: MultilinearPoly<P>` host objects.  What a maintainer wires up instead is the caller of the backend,
//! `ProverState` (crates/core/src/protocols/sumcheck/prove/prover_state.rs:57-265): its `multilinears` become
//! [`DevMultilinear`]s, `calculate_round_evals` calls [`Mi355xLayer::hal_round_evals`], `fold` calls
//! [`Mi355xLayer::hal_fold_multilinear`] per multilinear -- the C++ mirror `binius_amd/host/hal_backend.hpp` is that
//! wiring written out and tested (tests/cpp/conformance.cpp, test_old_hal_prover_state).

use std::ptr;

use binius_compute::layer::Error;
use binius_field::BinaryField128b as B128;
use binius_math::EvaluationOrder;

use crate::{
	check,
	exec::Mi355xExpr,
	ffi::{self, bn_hal_evaluator, bn_hal_multilinear},
	from_ffi,
	memory::{DevSlice, DevSliceMut},
	to_ffi, Mi355xLayer,
};

/// `SumcheckMultilinear` (crates/hal/src/sumcheck_multilinear.rs:8-30) with the evaluations in device memory.
pub enum DevMultilinear<'a> {
	/// Packed values of the subfield of tower level `tower_level`; `n_vars` variables, none bound yet.
	Transparent {
		packed: DevSlice<'a>,
		tower_level: usize,
		n_vars: usize,
	},
	/// Large-field evaluations after the challenges so far; the rest of the cube equals `suffix_eval`.
	Folded { evals: DevSlice<'a>, suffix_eval: B128 },
}

impl DevMultilinear<'_> {
	fn raw(&self) -> bn_hal_multilinear {
		match self {
			Self::Transparent {
				packed,
				tower_level,
				n_vars,
			} => bn_hal_multilinear {
				kind: ffi::BN_HAL_ML_TRANSPARENT,
				tower_level: *tower_level as u32,
				d_evals: packed.ptr.cast(),
				len: packed.len as u64,
				suffix_eval: to_ffi(B128::default()),
				n_vars_ml: *n_vars as u32,
			},
			Self::Folded { evals, suffix_eval } => bn_hal_multilinear {
				kind: ffi::BN_HAL_ML_FOLDED,
				tower_level: 0,
				d_evals: evals.ptr.cast(),
				len: evals.len as u64,
				suffix_eval: to_ffi(*suffix_eval),
				n_vars_ml: 0,
			},
		}
	}
}

/// What one `SumcheckEvaluator` (crates/hal/src/sumcheck_evaluator.rs:16-77) contributes to a round.
pub struct DevEvaluator<'a> {
	pub composition: &'a Mi355xExpr,
	/// `ArithCircuit::leading_term` of the composition (regular_sumcheck.rs:199-200).
	pub composition_at_infinity: &'a Mi355xExpr,
	/// `eval_point_indices`: 0 -> X = 0, 1 -> X = 1, 2 -> infinity, 3 + k -> nontrivial point k.
	pub eval_point_indices: std::ops::Range<usize>,
	/// Partial evaluations of the equality indicator (`EqIndSumcheckEvaluator`, eq_ind.rs:676-704).
	pub eq_ind_partial_evals: Option<DevSlice<'a>>,
}

fn order_code(order: EvaluationOrder) -> u32 {
	match order {
		EvaluationOrder::LowToHigh => ffi::BN_ORDER_LOW_TO_HIGH,
		EvaluationOrder::HighToLow => ffi::BN_ORDER_HIGH_TO_LOW,
	}
}

impl Mi355xLayer {
	/// `ComputationBackend::sumcheck_compute_round_evals` (backend.rs:52-67): one vector of evaluations per evaluator.
	pub fn hal_round_evals(
		&self,
		evaluation_order: EvaluationOrder,
		n_vars: usize,
		tensor_query: Option<&DevSlice<'_>>,
		multilinears: &[DevMultilinear<'_>],
		evaluators: &[DevEvaluator<'_>],
		nontrivial_evaluation_points: &[B128],
	) -> Result<Vec<Vec<B128>>, Error> {
		let mls = multilinears.iter().map(DevMultilinear::raw).collect::<Vec<_>>();
		let evs = evaluators
			.iter()
			.map(|e| bn_hal_evaluator {
				composition: e.composition.as_ptr(),
				composition_at_infinity: e.composition_at_infinity.as_ptr(),
				eval_point_start: e.eval_point_indices.start as u32,
				eval_point_end: e.eval_point_indices.end as u32,
				d_eq_ind: e
					.eq_ind_partial_evals
					.as_ref()
					.map_or(ptr::null(), |s| s.ptr.cast()),
			})
			.collect::<Vec<_>>();
		let points = nontrivial_evaluation_points
			.iter()
			.copied()
			.map(to_ffi)
			.collect::<Vec<_>>();
		let total = evaluators
			.iter()
			.map(|e| e.eval_point_indices.len())
			.sum::<usize>();
		let mut flat = vec![to_ffi(B128::default()); total.max(1)];
		let (q_ptr, q_vars) = tensor_query.map_or((ptr::null(), 0), |q| (q.ptr.cast(), q.len.trailing_zeros()));
		check(unsafe {
			ffi::bn_hal_round_evals(
				self.ctx,
				order_code(evaluation_order),
				n_vars as u32,
				q_ptr,
				q_vars,
				mls.as_ptr(),
				mls.len() as u32,
				evs.as_ptr(),
				evs.len() as u32,
				points.as_ptr(),
				points.len() as u32,
				flat.as_mut_ptr(),
			)
		})?;
		let mut out = Vec::with_capacity(evaluators.len());
		let mut off = 0;
		for e in evaluators {
			let cnt = e.eval_point_indices.len();
			out.push(flat[off..off + cnt].iter().copied().map(from_ffi).collect());
			off += cnt;
		}
		Ok(out)
	}

	/// One multilinear of `ComputationBackend::sumcheck_fold_multilinears` (backend.rs:69-78); returns the number of
	/// evaluations written to `out`.  A `Transparent` multilinear is partially evaluated at `tensor_query` (which
	/// already holds this round's challenge), a `Folded` one is folded by linear interpolation at `challenge`.
	pub fn hal_fold_multilinear(
		&self,
		evaluation_order: EvaluationOrder,
		n_vars: usize,
		multilinear: &DevMultilinear<'_>,
		challenge: B128,
		tensor_query: Option<&DevSlice<'_>>,
		out: &mut DevSliceMut<'_>,
	) -> Result<usize, Error> {
		let raw = multilinear.raw();
		let z = to_ffi(challenge);
		let mut n_out = 0u64;
		let (q_ptr, q_vars) = tensor_query.map_or((ptr::null(), 0), |q| (q.ptr.cast(), q.len.trailing_zeros()));
		check(unsafe {
			ffi::bn_hal_fold_multilinear(
				self.ctx,
				order_code(evaluation_order),
				n_vars as u32,
				&raw,
				&z,
				q_ptr,
				q_vars,
				out.ptr.cast(),
				out.len as u64,
				&mut n_out,
			)
		})?;
		Ok(n_out as usize)
	}
}
