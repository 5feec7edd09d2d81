//! The recording `KernelExecutor`.
//!
//! `accumulate_kernels` / `map_kernels` take the kernel as a Rust closure (crates/compute/src/layer.rs:
//! 140-216); a closure cannot cross an FFI.  The trait lets the layer call the closure itself -- possibly
//! several times, with whatever `log_chunks` it picks from `log_chunks_range` (layer.rs:171-177) -- so the
//! shim calls it ONCE against this executor.  Its memory model is symbolic: a slice is `(index of the memory
//! mapping, offset, length)` instead of data, a value is an id, and every `KernelExecutor` call appends one
//! `bn_kop` to a list that `bn_kernel_launch` then runs on the device, where the grid is the parallel
//! decomposition.

use std::{
	fmt::{self, Debug},
	marker::PhantomData,
	ops::RangeBounds,
};

use binius_compute::{
	layer::{Error, KernelExecutor},
	memory::{ComputeMemory, SizedSlice, SlicesBatch},
};
use binius_field::BinaryField128b as B128;

use crate::{
	exec::Mi355xExpr,
	ffi::{self, bn_f128, bn_kop, bn_kslice},
	memory::resolve_range,
	to_ffi,
};

/// Read-only symbolic slice of a kernel buffer.
#[derive(Clone, Copy)]
pub struct KSlice<'a> {
	pub(crate) buf: u32,
	pub(crate) off: u64,
	pub(crate) len: usize,
	_lifetime: PhantomData<&'a ()>,
}

/// Exclusive symbolic slice of a kernel buffer.
pub struct KSliceMut<'a> {
	pub(crate) buf: u32,
	pub(crate) off: u64,
	pub(crate) len: usize,
	_lifetime: PhantomData<&'a mut ()>,
}

impl<'a> KSlice<'a> {
	pub(crate) fn new(buf: u32, off: u64, len: usize) -> Self {
		Self {
			buf,
			off,
			len,
			_lifetime: PhantomData,
		}
	}

	fn raw(&self) -> bn_kslice {
		bn_kslice {
			buf: self.buf,
			off: self.off,
			len: self.len as u64,
		}
	}
}

impl<'a> KSliceMut<'a> {
	pub(crate) fn new(buf: u32, off: u64, len: usize) -> Self {
		Self {
			buf,
			off,
			len,
			_lifetime: PhantomData,
		}
	}

	fn raw(&self) -> bn_kslice {
		bn_kslice {
			buf: self.buf,
			off: self.off,
			len: self.len as u64,
		}
	}
}

impl Debug for KSlice<'_> {
	fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
		write!(f, "KSlice(buf {}, off {}, len {})", self.buf, self.off, self.len)
	}
}

impl Debug for KSliceMut<'_> {
	fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
		write!(f, "KSliceMut(buf {}, off {}, len {})", self.buf, self.off, self.len)
	}
}

impl SizedSlice for KSlice<'_> {
	fn len(&self) -> usize {
		self.len
	}
}

impl SizedSlice for KSliceMut<'_> {
	fn len(&self) -> usize {
		self.len
	}
}

/// The symbolic memory model of a recorded kernel.
pub struct RecMem;

impl ComputeMemory<B128> for RecMem {
	const ALIGNMENT: usize = 1;

	type FSlice<'a> = KSlice<'a>;
	type FSliceMut<'a> = KSliceMut<'a>;

	fn narrow<'a>(data: &'a Self::FSlice<'_>) -> Self::FSlice<'a> {
		KSlice::new(data.buf, data.off, data.len)
	}

	fn narrow_mut<'a, 'b: 'a>(data: Self::FSliceMut<'b>) -> Self::FSliceMut<'a> {
		KSliceMut::new(data.buf, data.off, data.len)
	}

	fn to_owned_mut<'a>(data: &'a mut Self::FSliceMut<'_>) -> Self::FSliceMut<'a> {
		KSliceMut::new(data.buf, data.off, data.len)
	}

	fn as_const<'a>(data: &'a Self::FSliceMut<'_>) -> Self::FSlice<'a> {
		KSlice::new(data.buf, data.off, data.len)
	}

	fn to_const(data: Self::FSliceMut<'_>) -> Self::FSlice<'_> {
		KSlice::new(data.buf, data.off, data.len)
	}

	fn slice(data: Self::FSlice<'_>, range: impl RangeBounds<usize>) -> Self::FSlice<'_> {
		let (start, end) = resolve_range(range, data.len);
		KSlice::new(data.buf, data.off + start as u64, end - start)
	}

	fn slice_mut<'a>(data: &'a mut Self::FSliceMut<'_>, range: impl RangeBounds<usize>) -> Self::FSliceMut<'a> {
		let (start, end) = resolve_range(range, data.len);
		KSliceMut::new(data.buf, data.off + start as u64, end - start)
	}

	fn split_at_mut(data: Self::FSliceMut<'_>, mid: usize) -> (Self::FSliceMut<'_>, Self::FSliceMut<'_>) {
		assert!(mid <= data.len, "split point {mid} out of bounds for a slice of length {}", data.len);
		(
			KSliceMut::new(data.buf, data.off, mid),
			KSliceMut::new(data.buf, data.off + mid as u64, data.len - mid),
		)
	}

	fn slice_chunks_mut<'a>(data: Self::FSliceMut<'a>, chunk_len: usize) -> impl Iterator<Item = Self::FSliceMut<'a>> {
		assert!(chunk_len > 0 && data.len % chunk_len == 0, "chunk length must divide the slice length");
		let (buf, off, n_chunks) = (data.buf, data.off, data.len / chunk_len);
		(0..n_chunks).map(move |i| KSliceMut::new(buf, off + (i * chunk_len) as u64, chunk_len))
	}
}

/// Id of a value declared inside a recorded kernel (`KernelExecutor::Value`).
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct ValueId(pub(crate) u32);

/// Records the operations of one kernel closure as a `bn_kop` list.
#[derive(Default)]
pub struct Recorder {
	pub(crate) ops: Vec<bn_kop>,
	// the row lists the SUM_COMPOSITION ops point into; boxed so the pointers stay valid while `ops` grows
	row_lists: Vec<Box<[bn_kslice]>>,
	n_values: u32,
}

impl Recorder {
	fn blank_op(kind: u32) -> bn_kop {
		bn_kop {
			kind,
			value: 0,
			scalar: bn_f128::default(),
			expr: std::ptr::null(),
			n_rows: 0,
			rows: std::ptr::null(),
			src1: bn_kslice::default(),
			src2: bn_kslice::default(),
			dst: bn_kslice::default(),
		}
	}
}

fn check_log_len(what: &str, len: usize, log_len: usize) -> Result<(), Error> {
	if len != 1 << log_len {
		return Err(Error::InputValidation(format!("{what} length must be equal to 2^log_len")));
	}
	Ok(())
}

impl KernelExecutor<B128> for Recorder {
	type Mem = RecMem;
	type Value = ValueId;
	type ExprEval = Mi355xExpr;

	fn decl_value(&mut self, init: B128) -> Result<Self::Value, Error> {
		let id = self.n_values;
		self.n_values += 1;
		let mut op = Self::blank_op(ffi::BN_KOP_DECL_VALUE);
		op.value = id;
		op.scalar = to_ffi(init);
		self.ops.push(op);
		Ok(ValueId(id))
	}

	fn sum_composition_evals(
		&mut self,
		inputs: &SlicesBatch<<Self::Mem as ComputeMemory<B128>>::FSlice<'_>>,
		composition: &Self::ExprEval,
		batch_coeff: B128,
		accumulator: &mut Self::Value,
	) -> Result<(), Error> {
		// A prover with k claims passes the SAME batch of m rows k times in a row (v3/bivariate_product.rs:355-399): the recorded
		// ops share one copy of it (compared by content) -- at keccak's width that is 100 x 100 slices per round otherwise.
		let same = self.row_lists.last().is_some_and(|last| {
			last.len() == inputs.n_rows()
				&& last.iter().zip(inputs.iter()).all(|(a, b)| {
					let b = b.raw();
					a.buf == b.buf && a.off == b.off && a.len == b.len
				})
		});
		if !same {
			let rows: Box<[bn_kslice]> = inputs.iter().map(KSlice::raw).collect();
			self.row_lists.push(rows);
		}
		let rows = self.row_lists.last().expect("pushed above");
		let mut op = Self::blank_op(ffi::BN_KOP_SUM_COMPOSITION);
		op.value = accumulator.0;
		op.scalar = to_ffi(batch_coeff);
		op.expr = composition.as_ptr();
		op.n_rows = rows.len() as u32;
		op.rows = rows.as_ptr();
		self.ops.push(op);
		Ok(())
	}

	fn add(
		&mut self,
		log_len: usize,
		src1: <Self::Mem as ComputeMemory<B128>>::FSlice<'_>,
		src2: <Self::Mem as ComputeMemory<B128>>::FSlice<'_>,
		dst: &mut <Self::Mem as ComputeMemory<B128>>::FSliceMut<'_>,
	) -> Result<(), Error> {
		check_log_len("src1", src1.len, log_len)?;
		check_log_len("src2", src2.len, log_len)?;
		check_log_len("dst", dst.len, log_len)?;
		let mut op = Self::blank_op(ffi::BN_KOP_ADD);
		op.src1 = src1.raw();
		op.src2 = src2.raw();
		op.dst = dst.raw();
		self.ops.push(op);
		Ok(())
	}

	fn add_assign(
		&mut self,
		log_len: usize,
		src: <Self::Mem as ComputeMemory<B128>>::FSlice<'_>,
		dst: &mut <Self::Mem as ComputeMemory<B128>>::FSliceMut<'_>,
	) -> Result<(), Error> {
		check_log_len("src", src.len, log_len)?;
		check_log_len("dst", dst.len, log_len)?;
		let mut op = Self::blank_op(ffi::BN_KOP_ADD_ASSIGN);
		op.src1 = src.raw();
		op.dst = dst.raw();
		self.ops.push(op);
		Ok(())
	}
}
