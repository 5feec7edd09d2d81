//! Device memory handles: `ComputeMemory<B128>` for HBM behind the C ABI.
//!
//! A slice is `(device pointer, length)`; every operation of the trait is O(1) host arithmetic with no
//! device call (crates/compute/src/memory.rs:69-234).  `ALIGNMENT = 1`: any element boundary is a valid
//! split point, exactly like `CpuMemory`, so `split_half` and the bump allocator keep their default
//! behaviour.

use std::{
	fmt::{self, Debug},
	marker::PhantomData,
	ops::{Bound, RangeBounds},
};

use binius_compute::memory::{ComputeMemory, SizedSlice};
use binius_field::BinaryField128b as B128;

const ELEM_BYTES: usize = 16;

/// Read-only view of `len` field elements in device memory.
#[derive(Clone, Copy)]
pub struct DevSlice<'a> {
	pub(crate) ptr: *const u8,
	pub(crate) len: usize,
	_lifetime: PhantomData<&'a [B128]>,
}

/// Exclusive view of `len` field elements in device memory.
pub struct DevSliceMut<'a> {
	pub(crate) ptr: *mut u8,
	pub(crate) len: usize,
	_lifetime: PhantomData<&'a mut [B128]>,
}

// The pointers are device addresses: the host never dereferences them, and the C ABI serialises the
// entry points of a context, so handing a slice to another host thread is sound.
unsafe impl Send for DevSlice<'_> {}
unsafe impl Sync for DevSlice<'_> {}
unsafe impl Send for DevSliceMut<'_> {}
unsafe impl Sync for DevSliceMut<'_> {}

impl<'a> DevSlice<'a> {
	pub(crate) fn from_raw(ptr: *const u8, len: usize) -> Self {
		Self {
			ptr,
			len,
			_lifetime: PhantomData,
		}
	}

	/// The device address, for the FFI layer.
	pub fn as_ptr(&self) -> *const std::os::raw::c_void {
		self.ptr.cast()
	}
}

impl<'a> DevSliceMut<'a> {
	pub(crate) fn from_raw(ptr: *mut u8, len: usize) -> Self {
		Self {
			ptr,
			len,
			_lifetime: PhantomData,
		}
	}

	/// The device address, for the FFI layer.
	pub fn as_mut_ptr(&mut self) -> *mut std::os::raw::c_void {
		self.ptr.cast()
	}

	pub(crate) fn as_const_ptr(&self) -> *const std::os::raw::c_void {
		self.ptr.cast_const().cast()
	}
}

impl Debug for DevSlice<'_> {
	fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
		write!(f, "DevSlice({:p}, len = {})", self.ptr, self.len)
	}
}

impl Debug for DevSliceMut<'_> {
	fn fmt(&self, f: &mut fmt::Formatter<'_>) -> fmt::Result {
		write!(f, "DevSliceMut({:p}, len = {})", self.ptr, self.len)
	}
}

impl SizedSlice for DevSlice<'_> {
	fn len(&self) -> usize {
		self.len
	}
}

impl SizedSlice for DevSliceMut<'_> {
	fn len(&self) -> usize {
		self.len
	}
}

/// `(start, end)` of `range` inside a slice of `len` elements; panics like `[T]` indexing does.
pub(crate) fn resolve_range(range: impl RangeBounds<usize>, len: usize) -> (usize, usize) {
	let start = match range.start_bound() {
		Bound::Included(&s) => s,
		Bound::Excluded(&s) => s + 1,
		Bound::Unbounded => 0,
	};
	let end = match range.end_bound() {
		Bound::Included(&e) => e + 1,
		Bound::Excluded(&e) => e,
		Bound::Unbounded => len,
	};
	assert!(start <= end && end <= len, "range {start}..{end} out of bounds for a slice of length {len}");
	(start, end)
}

/// The memory model of the MI355X backend.
pub struct Mi355xMemory;

impl ComputeMemory<B128> for Mi355xMemory {
	const ALIGNMENT: usize = 1;

	type FSlice<'a> = DevSlice<'a>;
	type FSliceMut<'a> = DevSliceMut<'a>;

	fn narrow<'a>(data: &'a Self::FSlice<'_>) -> Self::FSlice<'a> {
		DevSlice::from_raw(data.ptr, data.len)
	}

	fn narrow_mut<'a, 'b: 'a>(data: Self::FSliceMut<'b>) -> Self::FSliceMut<'a> {
		DevSliceMut::from_raw(data.ptr, data.len)
	}

	fn to_owned_mut<'a>(data: &'a mut Self::FSliceMut<'_>) -> Self::FSliceMut<'a> {
		DevSliceMut::from_raw(data.ptr, data.len)
	}

	fn as_const<'a>(data: &'a Self::FSliceMut<'_>) -> Self::FSlice<'a> {
		DevSlice::from_raw(data.ptr.cast_const(), data.len)
	}

	fn to_const(data: Self::FSliceMut<'_>) -> Self::FSlice<'_> {
		DevSlice::from_raw(data.ptr.cast_const(), data.len)
	}

	fn slice(data: Self::FSlice<'_>, range: impl RangeBounds<usize>) -> Self::FSlice<'_> {
		let (start, end) = resolve_range(range, data.len);
		DevSlice::from_raw(data.ptr.wrapping_add(start * ELEM_BYTES), end - start)
	}

	fn slice_mut<'a>(data: &'a mut Self::FSliceMut<'_>, range: impl RangeBounds<usize>) -> Self::FSliceMut<'a> {
		let (start, end) = resolve_range(range, data.len);
		DevSliceMut::from_raw(data.ptr.wrapping_add(start * ELEM_BYTES), end - start)
	}

	fn split_at_mut(data: Self::FSliceMut<'_>, mid: usize) -> (Self::FSliceMut<'_>, Self::FSliceMut<'_>) {
		assert!(mid <= data.len, "split point {mid} out of bounds for a slice of length {}", data.len);
		(
			DevSliceMut::from_raw(data.ptr, mid),
			DevSliceMut::from_raw(data.ptr.wrapping_add(mid * ELEM_BYTES), data.len - mid),
		)
	}

	fn slice_chunks_mut<'a>(data: Self::FSliceMut<'a>, chunk_len: usize) -> impl Iterator<Item = Self::FSliceMut<'a>> {
		assert!(chunk_len > 0 && data.len % chunk_len == 0, "chunk length must divide the slice length");
		let (ptr, n_chunks) = (data.ptr, data.len / chunk_len);
		(0..n_chunks).map(move |i| DevSliceMut::from_raw(ptr.wrapping_add(i * chunk_len * ELEM_BYTES), chunk_len))
	}
}
