//! `ComputeLayerExecutor<B128>` over the C ABI: one forwarding call per trait method, no arithmetic.

use std::{marker::PhantomData, os::raw::c_void, ptr};

use binius_compute::{
	layer::{ComputeLayerExecutor, Error, KernelBuffer, KernelMemMap},
	memory::{ComputeMemory, SizedSlice, SlicesBatch, SubfieldSlice},
};
use binius_field::{BinaryField, BinaryField128b as B128, ExtensionField};
use binius_ntt::AdditiveNTT;

use crate::{
	check,
	ffi::{self, bn_ctx, bn_expr, bn_f128, bn_memmap},
	from_ffi,
	memory::{DevSlice, DevSliceMut, Mi355xMemory},
	recorder::{KSlice, KSliceMut, RecMem, Recorder, ValueId},
	to_ffi,
};

/// A compiled arithmetic circuit living in the backend (`bn_expr`).
pub struct Mi355xExpr(pub(crate) *mut bn_expr);

// The handle is immutable after compilation and the backend only reads it.
unsafe impl Send for Mi355xExpr {}
unsafe impl Sync for Mi355xExpr {}

impl Mi355xExpr {
	pub(crate) fn as_ptr(&self) -> *const bn_expr {
		self.0.cast_const()
	}
}

impl Drop for Mi355xExpr {
	fn drop(&mut self) {
		// nothing useful can be done with a failure here
		let _ = unsafe { ffi::bn_expr_free(self.0) };
	}
}

/// One `extrapolate_line` recorded inside a `map` scope.
struct PendingLine {
	evals_0: *mut c_void,
	evals_1: *const c_void,
	len: usize,
	z: bn_f128,
}

/// The executor handed to `ComputeLayer::execute` closures.
pub struct Mi355xExec<'a> {
	pub(crate) ctx: *mut bn_ctx,
	// extrapolate_line calls issued inside a `map` scope: the prover folds every multilinear of a round in
	// one `map` (core/src/protocols/sumcheck/v3/bivariate_product.rs:217-228); sent as ONE
	// bn_extrapolate_line_batch the backend can run together with the next round's evaluation
	lines: Vec<PendingLine>,
	map_depth: usize,
	pub(crate) _lifetime: PhantomData<&'a ()>,
}

unsafe impl Send for Mi355xExec<'_> {}

/// Largest batch `bn_extrapolate_line_batch` accepts (include/binius_amd.h BN_FOLD_CALL_MAX = binius_amd/csrc/internal.hpp
/// kFoldCallMax): the fold of a whole prover -- every multilinear of one size -- is one call, which the backend defers as one batch.
const FOLD_BATCH_MAX: usize = 256;

impl<'a> Mi355xExec<'a> {
	pub(crate) fn new(ctx: *mut bn_ctx) -> Self {
		Self {
			ctx,
			lines: Vec::new(),
			map_depth: 0,
			_lifetime: PhantomData,
		}
	}

	/// Sends the recorded folds, in issue order, batching runs with the same length and challenge.
	pub(crate) fn flush_lines(&mut self) -> Result<(), Error> {
		let lines = std::mem::take(&mut self.lines);
		let mut i = 0;
		while i < lines.len() {
			let mut j = i + 1;
			while j < lines.len() && j - i < FOLD_BATCH_MAX && lines[j].len == lines[i].len && lines[j].z == lines[i].z {
				j += 1;
			}
			let e0: Vec<*mut c_void> = lines[i..j].iter().map(|l| l.evals_0).collect();
			let e1: Vec<*const c_void> = lines[i..j].iter().map(|l| l.evals_1).collect();
			check(unsafe { ffi::bn_extrapolate_line_batch(self.ctx, e0.as_ptr(), e1.as_ptr(), (j - i) as u32, lines[i].len as u64, &lines[i].z) })?;
			i = j;
		}
		Ok(())
	}

	fn raw_maps(mem_maps: &[KernelMemMap<'_, B128, Mi355xMemory>]) -> Vec<bn_memmap> {
		mem_maps
			.iter()
			.map(|m| match m {
				KernelMemMap::Chunked { data, log_min_chunk_size } => bn_memmap {
					kind: ffi::BN_MAP_CHUNKED,
					log_min_chunk_size: *log_min_chunk_size as u32,
					d_data: data.as_ptr().cast_mut(),
					len: data.len() as u64,
					log_size: 0,
				},
				KernelMemMap::ChunkedMut { data, log_min_chunk_size } => bn_memmap {
					kind: ffi::BN_MAP_CHUNKED_MUT,
					log_min_chunk_size: *log_min_chunk_size as u32,
					d_data: data.as_const_ptr().cast_mut(),
					len: data.len() as u64,
					log_size: 0,
				},
				KernelMemMap::Local { log_size } => bn_memmap {
					kind: ffi::BN_MAP_LOCAL,
					log_min_chunk_size: 0,
					d_data: ptr::null_mut(),
					len: 0,
					log_size: *log_size as u32,
				},
			})
			.collect()
	}

	/// The kernel's view of the mappings for one chunk: symbolic slices `(mapping index, 0, chunk length)`.
	fn symbolic_buffers<'k>(mem_maps: &[KernelMemMap<'_, B128, Mi355xMemory>], log_chunks: usize) -> Vec<KernelBuffer<'k, B128, RecMem>> {
		mem_maps
			.iter()
			.enumerate()
			.map(|(i, m)| match m {
				KernelMemMap::Chunked { data, .. } => KernelBuffer::Ref(KSlice::new(i as u32, 0, data.len() >> log_chunks)),
				KernelMemMap::ChunkedMut { data, .. } => KernelBuffer::Mut(KSliceMut::new(i as u32, 0, data.len() >> log_chunks)),
				KernelMemMap::Local { log_size } => KernelBuffer::Mut(KSliceMut::new(i as u32, 0, 1 << (log_size - log_chunks))),
			})
			.collect()
	}

	/// Picks `log_chunks`, runs the kernel closure once against a recorder and launches what it recorded.
	fn launch_recorded<R>(
		&mut self,
		mem_maps: &[KernelMemMap<'_, B128, Mi355xMemory>],
		record: impl for<'k> FnOnce(&'k mut Recorder, usize, Vec<KernelBuffer<'k, B128, RecMem>>) -> Result<R, Error>,
		ret_ids: impl FnOnce(&R) -> Vec<u32>,
	) -> Result<Vec<B128>, Error> {
		self.flush_lines()?;
		let raw = Self::raw_maps(mem_maps);
		let mut log_chunks = 0u32;
		check(unsafe { ffi::bn_pick_log_chunks(raw.as_ptr(), raw.len() as u32, &mut log_chunks) })?;
		let mut recorder = Recorder::default();
		let buffers = Self::symbolic_buffers(mem_maps, log_chunks as usize);
		let recorded = record(&mut recorder, log_chunks as usize, buffers)?;
		let rets = ret_ids(&recorded);
		let mut out = vec![bn_f128::default(); rets.len()];
		check(unsafe {
			ffi::bn_kernel_launch(
				self.ctx,
				raw.as_ptr(),
				raw.len() as u32,
				recorder.ops.as_ptr(),
				recorder.ops.len() as u32,
				rets.as_ptr(),
				rets.len() as u32,
				log_chunks,
				if rets.is_empty() { ptr::null_mut() } else { out.as_mut_ptr() },
				ptr::null_mut(),
			)
		})?;
		Ok(out.into_iter().map(from_ffi).collect())
	}
}

impl<'a> ComputeLayerExecutor<B128> for Mi355xExec<'a> {
	type ExprEval = Mi355xExpr;
	type DevMem = Mi355xMemory;
	type OpValue = B128;
	type KernelExec = Recorder;

	fn join<Out1: Send, Out2: Send>(
		&mut self,
		op1: impl Send + FnOnce(&mut Self) -> Result<Out1, Error>,
		op2: impl Send + FnOnce(&mut Self) -> Result<Out2, Error>,
	) -> Result<(Out1, Out2), Error> {
		// one in-order stream per context: the two branches run back to back (the trait's default)
		let out1 = op1(self)?;
		let out2 = op2(self)?;
		Ok((out1, out2))
	}

	fn map<Out: Send, I: ExactSizeIterator<Item: Send> + Send>(
		&mut self,
		iter: I,
		map: impl Sync + Fn(&mut Self, I::Item) -> Result<Out, Error>,
	) -> Result<Vec<Out>, Error> {
		self.map_depth += 1;
		let out: Result<Vec<Out>, Error> = iter.map(|item| map(self, item)).collect();
		self.map_depth -= 1;
		if self.map_depth == 0 {
			self.flush_lines()?;
		}
		out
	}

	fn accumulate_kernels(
		&mut self,
		map: impl Sync
		+ for<'k> Fn(&'k mut Self::KernelExec, usize, Vec<KernelBuffer<'k, B128, RecMem>>) -> Result<Vec<ValueId>, Error>,
		mem_maps: Vec<KernelMemMap<'_, B128, Self::DevMem>>,
	) -> Result<Vec<Self::OpValue>, Error> {
		self.launch_recorded(&mem_maps, |rec, log_chunks, buffers| map(rec, log_chunks, buffers), |values| values.iter().map(|v| v.0).collect())
	}

	fn map_kernels(
		&mut self,
		map: impl Sync + for<'k> Fn(&'k mut Self::KernelExec, usize, Vec<KernelBuffer<'k, B128, RecMem>>) -> Result<(), Error>,
		mem_maps: Vec<KernelMemMap<'_, B128, Self::DevMem>>,
	) -> Result<(), Error> {
		self.launch_recorded(&mem_maps, |rec, log_chunks, buffers| map(rec, log_chunks, buffers), |_| Vec::new())
			.map(|_| ())
	}

	fn inner_product(&mut self, a_in: SubfieldSlice<'_, B128, Self::DevMem>, b_in: DevSlice<'_>) -> Result<Self::OpValue, Error> {
		self.flush_lines()?;
		let mut out = bn_f128::default();
		check(unsafe {
			ffi::bn_inner_product(
				self.ctx,
				a_in.slice.as_ptr(),
				a_in.slice.len() as u64,
				a_in.tower_level as u32,
				b_in.as_ptr(),
				b_in.len() as u64,
				&mut out,
			)
		})?;
		Ok(from_ffi(out))
	}

	fn tensor_expand(&mut self, log_n: usize, coordinates: &[B128], data: &mut DevSliceMut<'_>) -> Result<(), Error> {
		self.flush_lines()?;
		let coords: Vec<bn_f128> = coordinates.iter().copied().map(to_ffi).collect();
		check(unsafe { ffi::bn_tensor_expand(self.ctx, data.as_mut_ptr(), data.len() as u64, log_n as u32, coords.as_ptr(), coords.len() as u32) })
	}

	fn fold_left(&mut self, mat: SubfieldSlice<'_, B128, Self::DevMem>, vec: DevSlice<'_>, out: &mut DevSliceMut<'_>) -> Result<(), Error> {
		self.flush_lines()?;
		check(unsafe {
			ffi::bn_fold_left(
				self.ctx,
				mat.slice.as_ptr(),
				mat.slice.len() as u64,
				mat.tower_level as u32,
				vec.as_ptr(),
				vec.len() as u64,
				out.as_mut_ptr(),
				out.len() as u64,
			)
		})
	}

	fn fold_right(&mut self, mat: SubfieldSlice<'_, B128, Self::DevMem>, vec: DevSlice<'_>, out: &mut DevSliceMut<'_>) -> Result<(), Error> {
		self.flush_lines()?;
		check(unsafe {
			ffi::bn_fold_right(
				self.ctx,
				mat.slice.as_ptr(),
				mat.slice.len() as u64,
				mat.tower_level as u32,
				vec.as_ptr(),
				vec.len() as u64,
				out.as_mut_ptr(),
				out.len() as u64,
			)
		})
	}

	fn fri_fold<FSub>(
		&mut self,
		ntt: &(impl AdditiveNTT<FSub> + Sync),
		log_len: usize,
		log_batch_size: usize,
		challenges: &[B128],
		data_in: DevSlice<'_>,
		data_out: &mut DevSliceMut<'_>,
	) -> Result<(), Error>
	where
		FSub: BinaryField,
		B128: ExtensionField<FSub>,
	{
		self.flush_lines()?;
		// The twiddle basis of the NTT, in the row layout of OnTheFlyTwiddleAccess (crates/ntt/src/twiddle.rs:
		// 141-168): row `layer` holds W^_layer(beta_{layer+1+b}).  Every concrete NTT of the reference answers
		// get_subspace_eval(i, index) with the XOR-combination of row l - i selected by the bits of `index`
		// (single_threaded.rs:91-93), so index = 1 << b reads entry b.
		let log_domain = ntt.log_domain_size();
		if log_domain == 0 || log_domain > ffi::BN_NTT_MAX_DIM {
			return Err(Error::InputValidation("NTT domain size not supported by the backend".to_string()));
		}
		let mut s_evals = vec![0u64; ffi::BN_NTT_MAX_DIM * ffi::BN_NTT_MAX_DIM];
		for layer in 0..log_domain {
			for b in 0..(log_domain - 1 - layer) {
				let t: B128 = ntt.get_subspace_eval(log_domain - layer, 1 << b).into();
				s_evals[layer * ffi::BN_NTT_MAX_DIM + b] = u128::from(t) as u64;
			}
		}
		let tw_level = FSub::N_BITS.ilog2();
		let ch: Vec<bn_f128> = challenges.iter().copied().map(to_ffi).collect();
		check(unsafe {
			ffi::bn_fri_fold(
				self.ctx,
				s_evals.as_ptr(),
				tw_level,
				log_domain as u32,
				log_len as u32,
				log_batch_size as u32,
				ch.as_ptr(),
				ch.len() as u32,
				data_in.as_ptr(),
				data_in.len() as u64,
				data_out.as_mut_ptr(),
				data_out.len() as u64,
			)
		})
	}

	fn extrapolate_line(&mut self, evals_0: &mut DevSliceMut<'_>, evals_1: DevSlice<'_>, z: B128) -> Result<(), Error> {
		if evals_0.len() != evals_1.len() {
			return Err(Error::InputValidation("evals_0 and evals_1 must be the same length".to_string()));
		}
		if self.map_depth > 0 {
			self.lines.push(PendingLine {
				evals_0: evals_0.as_mut_ptr(),
				evals_1: evals_1.as_ptr(),
				len: evals_0.len(),
				z: to_ffi(z),
			});
			return Ok(());
		}
		self.flush_lines()?;
		check(unsafe { ffi::bn_extrapolate_line(self.ctx, evals_0.as_mut_ptr(), evals_0.len() as u64, evals_1.as_ptr(), evals_1.len() as u64, &to_ffi(z)) })
	}

	fn compute_composite(&mut self, inputs: &SlicesBatch<DevSlice<'_>>, output: &mut DevSliceMut<'_>, composition: &Self::ExprEval) -> Result<(), Error> {
		self.flush_lines()?;
		let rows: Vec<*const c_void> = inputs.iter().map(DevSlice::as_ptr).collect();
		check(unsafe {
			ffi::bn_compute_composite(
				self.ctx,
				rows.as_ptr(),
				rows.len() as u32,
				inputs.row_len() as u64,
				output.as_mut_ptr(),
				output.len() as u64,
				composition.as_ptr(),
			)
		})
	}

	fn pairwise_product_reduce(&mut self, input: DevSlice<'_>, round_outputs: &mut [DevSliceMut<'_>]) -> Result<(), Error> {
		self.flush_lines()?;
		let outs: Vec<*mut c_void> = round_outputs.iter_mut().map(DevSliceMut::as_mut_ptr).collect();
		let lens: Vec<u64> = round_outputs.iter().map(|o| o.len() as u64).collect();
		check(unsafe { ffi::bn_pairwise_product_reduce(self.ctx, input.as_ptr(), input.len() as u64, outs.as_ptr(), lens.as_ptr(), outs.len() as u32) })
	}
}
