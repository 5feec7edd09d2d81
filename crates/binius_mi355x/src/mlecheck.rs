//! `SumcheckProver<B128>` for the eq-indicator bivariate-product claim, backed by the compiled prover of
//! `libbinius_amd_host.so` (`bnh_mlecheck_*`, include/binius_amd_host.h) -- the drop-in for
//! `BivariateMLEcheckProver` (crates/core/src/protocols/sumcheck/v3/bivariate_mlecheck.rs:27-389) when the layer is
//! `Mi355xLayer`.
//!
//! Same inputs, same transcript.  Behind the handle the indicator is multiplied into one factor of every composition
//! once and the rounds are plain bivariate product rounds on the matrix-core kernels; where that does
//! not apply (no proper 2-colouring of the compositions, an indicator coordinate equal to 0 or 1, too little scratch)
//! the same handle runs the literal trait-call sequence of the reference prover.
//!
//! Feature `provers` (pulls in `binius_core` for the trait and the claim types).

use std::{os::raw::c_int, ptr};

use binius_core::{
	composition::{BivariateProduct, IndexComposition},
	protocols::sumcheck::{prove::SumcheckProver, EqIndSumcheckClaim, Error as SumcheckError, RoundCoeffs},
};
use binius_field::BinaryField128b as B128;
use binius_math::EvaluationOrder;

use crate::{
	ffi::{bn_ctx, bn_f128},
	from_ffi,
	memory::{DevSlice, DevSliceMut},
	to_ffi, Mi355xLayer,
};

#[repr(C)]
pub struct bnh_mlecheck {
	_private: [u8; 0],
}

unsafe extern "C" {
	fn bnh_last_error() -> *const std::os::raw::c_char;
	fn bnh_mlecheck_new(
		ctx: *mut bn_ctx,
		n_vars: u32,
		m: u32,
		d_multilins: *const *const std::os::raw::c_void,
		d_eq_ind: *const std::os::raw::c_void,
		eq_ind_challenges: *const bn_f128,
		d_scratch: *mut std::os::raw::c_void,
		scratch_elems: u64,
		n_comps: u32,
		comp_indices: *const u32,
		sums: *const bn_f128,
		out: *mut *mut bnh_mlecheck,
	) -> c_int;
	fn bnh_mlecheck_execute(prover: *mut bnh_mlecheck, batch_coeff: *const bn_f128, coeffs_out: *mut bn_f128) -> c_int;
	fn bnh_mlecheck_fold(prover: *mut bnh_mlecheck, challenge: *const bn_f128) -> c_int;
	fn bnh_mlecheck_finish(prover: *mut bnh_mlecheck, final_evals_out: *mut bn_f128) -> c_int;
	fn bnh_mlecheck_free(prover: *mut bnh_mlecheck);
}

fn host_check(rc: c_int) -> Result<(), SumcheckError> {
	if rc == 0 {
		return Ok(());
	}
	let msg = unsafe { std::ffi::CStr::from_ptr(bnh_last_error()) }
		.to_string_lossy()
		.into_owned();
	// the phase errors of the reference prover keep their identity; everything else is a compute-layer error
	Err(match msg.as_str() {
		"ExpectedFold" => SumcheckError::ExpectedFold,
		"ExpectedExecution" => SumcheckError::ExpectedExecution,
		"ExpectedFinish" => SumcheckError::ExpectedFinish,
		"NumberOfVariablesMismatch" => SumcheckError::NumberOfVariablesMismatch,
		"IncorrectEqIndPartialEvalsSize" => SumcheckError::IncorrectEqIndPartialEvalsSize,
		_ => SumcheckError::Compute(binius_compute::Error::CoreLibError(msg.into())),
	})
}

/// The MLE-check prover of `bivariate_mlecheck.rs` for `Hal = Mi355xLayer`.
pub struct Mi355xMLEcheckProver<'a> {
	handle: *mut bnh_mlecheck,
	n_vars: usize,
	n_multilinears: usize,
	// the device memory the handle works on stays borrowed for its lifetime
	_multilins: Vec<DevSlice<'a>>,
	_eq_ind: DevSlice<'a>,
	_scratch: DevSliceMut<'a>,
}

impl<'a> Mi355xMLEcheckProver<'a> {
	/// Arguments as `BivariateMLEcheckProver::new` (:64-131); `scratch` replaces the device allocator: 2^n_vars
	/// elements per weighted multilinear and 2^(n_vars-1) per other one are enough (`(m + 1) * 2^(n_vars-1)` for the
	/// literal sequence).
	pub fn new(
		hal: &'a Mi355xLayer,
		claim: &EqIndSumcheckClaim<B128, IndexComposition<BivariateProduct, 2>>,
		multilins: Vec<DevSlice<'a>>,
		eq_ind_partial_evals: DevSlice<'a>,
		eq_ind_challenges: &[B128],
		scratch: DevSliceMut<'a>,
	) -> Result<Self, SumcheckError> {
		let n_vars = claim.n_vars();
		assert_eq!(claim.n_multilinears(), multilins.len());
		let ptrs = multilins.iter().map(|s| s.ptr.cast()).collect::<Vec<_>>();
		let (comp_indices, sums): (Vec<[u32; 2]>, Vec<bn_f128>) = claim
			.eq_ind_composite_sums()
			.iter()
			.map(|c| {
				let idx = c.composition.indices();
				([idx[0] as u32, idx[1] as u32], to_ffi(c.sum))
			})
			.unzip();
		let challenges = eq_ind_challenges.iter().copied().map(to_ffi).collect::<Vec<_>>();
		if challenges.len() != n_vars {
			return Err(SumcheckError::IncorrectEqIndChallengesLength);
		}
		let mut handle = ptr::null_mut();
		host_check(unsafe {
			bnh_mlecheck_new(
				hal.ctx,
				n_vars as u32,
				multilins.len() as u32,
				ptrs.as_ptr(),
				eq_ind_partial_evals.ptr.cast(),
				challenges.as_ptr(),
				scratch.ptr.cast(),
				scratch.len as u64,
				comp_indices.len() as u32,
				comp_indices.as_ptr().cast(),
				sums.as_ptr(),
				&mut handle,
			)
		})?;
		Ok(Self {
			handle,
			n_vars,
			n_multilinears: multilins.len(),
			_multilins: multilins,
			_eq_ind: eq_ind_partial_evals,
			_scratch: scratch,
		})
	}
}

impl Drop for Mi355xMLEcheckProver<'_> {
	fn drop(&mut self) {
		unsafe { bnh_mlecheck_free(self.handle) };
	}
}

impl SumcheckProver<B128> for Mi355xMLEcheckProver<'_> {
	fn n_vars(&self) -> usize {
		self.n_vars
	}

	fn evaluation_order(&self) -> EvaluationOrder {
		EvaluationOrder::HighToLow
	}

	fn execute(&mut self, batch_coeff: B128) -> Result<RoundCoeffs<B128>, SumcheckError> {
		let bc = to_ffi(batch_coeff);
		let mut out = [to_ffi(B128::default()); 4];
		host_check(unsafe { bnh_mlecheck_execute(self.handle, &bc, out.as_mut_ptr()) })?;
		Ok(RoundCoeffs(out.iter().copied().map(from_ffi).collect()))
	}

	fn fold(&mut self, challenge: B128) -> Result<(), SumcheckError> {
		let z = to_ffi(challenge);
		host_check(unsafe { bnh_mlecheck_fold(self.handle, &z) })
	}

	fn finish(self: Box<Self>) -> Result<Vec<B128>, SumcheckError> {
		let mut out = vec![to_ffi(B128::default()); self.n_multilinears + 1];
		host_check(unsafe { bnh_mlecheck_finish(self.handle, out.as_mut_ptr()) })?;
		Ok(out.into_iter().map(from_ffi).collect())
	}
}
