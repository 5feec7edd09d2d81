//! `binius_compute::ComputeLayer<BinaryField128b>` on an AMD Instinct MI355X.
//!
//! A forwarding layer: every trait method is one call into `libbinius_amd.so` (hand-written gfx950 kernels
//! behind the C ABI of `include/binius_amd.h`).  No field arithmetic happens on this side; the prover code in
//! `binius_core` (generic over `Hal: ComputeLayer<F>`) is unchanged.
//!
//! ```ignore
//! use binius_mi355x::Mi355xLayerHolder;
//! let mut holder = Mi355xLayerHolder::new(1 << 20, 1 << 28);   // host / device elements
//! let compute_data = holder.to_data();                         // { hal, host_alloc, dev_alloc }
//! // ... binius_core::constraint_system::prove::<_, _, _, _, _, Mi355xLayer, _, _>(&mut compute_data, ...)
//! ```
//!
//! Trait surface implemented: `crates/compute/src/layer.rs:22-590` (`ComputeLayer`, `ComputeLayerExecutor`,
//! `KernelExecutor`), `memory.rs:69-234` (`ComputeMemory`), `layer.rs:732-776` (`ComputeHolder`).

pub mod exec;
pub mod ffi;
pub mod hal;
pub mod holder;
pub mod memory;
#[cfg(feature = "provers")]
pub mod mlecheck;
pub mod recorder;

use std::{ffi::CStr, os::raw::c_int, ptr};

use binius_compute::{
	alloc::Error as AllocError,
	layer::{ComputeLayer, ComputeLayerExecutor, Error, FSlice, FSliceMut},
	memory::{ComputeMemory, SizedSlice},
};
use binius_field::BinaryField128b as B128;
use binius_math::{ArithCircuit, ArithCircuitStep};

pub use crate::{
	exec::{Mi355xExec, Mi355xExpr},
	hal::{DevEvaluator, DevMultilinear},
	holder::Mi355xLayerHolder,
	memory::{DevSlice, DevSliceMut, Mi355xMemory},
};
use crate::ffi::{bn_ctx, bn_f128, bn_step};

pub(crate) fn to_ffi(x: B128) -> bn_f128 {
	let v = u128::from(x);
	bn_f128 {
		lo: v as u64,
		hi: (v >> 64) as u64,
	}
}

pub(crate) fn from_ffi(x: bn_f128) -> B128 {
	B128::new(u128::from(x.lo) | (u128::from(x.hi) << 64))
}

fn last_error() -> String {
	let p = unsafe { ffi::bn_last_error() };
	if p.is_null() {
		return String::new();
	}
	unsafe { CStr::from_ptr(p) }.to_string_lossy().into_owned()
}

/// `bn_status` -> `binius_compute::Error` (the backend's codes mirror the enum, layer.rs:706-716).
pub(crate) fn check(rc: c_int) -> Result<(), Error> {
	match rc {
		ffi::BN_OK => Ok(()),
		ffi::BN_ERR_INPUT_VALIDATION => {
			let msg = last_error();
			Err(Error::InputValidation(msg.strip_prefix("input validation: ").unwrap_or(&msg).to_string()))
		}
		ffi::BN_ERR_ALLOC => Err(Error::Alloc(AllocError::OutOfMemory)),
		ffi::BN_ERR_CORE_LIB => Err(Error::CoreLibError(last_error().into())),
		_ => Err(Error::DeviceError(last_error().into())),
	}
}

/// Device index used by `Default` and `Mi355xLayerHolder::new`: `BINIUS_MI355X_DEVICE`, else 0.
pub(crate) fn default_device() -> i32 {
	std::env::var("BINIUS_MI355X_DEVICE").ok().and_then(|v| v.parse().ok()).unwrap_or(0)
}

/// The compute layer: one backend context (device, stream, scratch, optional arena).
pub struct Mi355xLayer {
	pub(crate) ctx: *mut bn_ctx,
}

// Every entry point of the C ABI takes the context's lock and makes its device current; the trait lets the
// host call from several threads (rayon join/map).
unsafe impl Send for Mi355xLayer {}
unsafe impl Sync for Mi355xLayer {}

impl Mi355xLayer {
	/// A context without an arena: the caller brings device memory of its own.
	pub fn new(device: i32) -> Result<Self, Error> {
		Self::with_arena(device, 0)
	}

	/// A context that owns an arena of `arena_elems` field elements (see `Mi355xLayerHolder`).
	pub fn with_arena(device: i32, arena_elems: usize) -> Result<Self, Error> {
		let mut ctx = ptr::null_mut();
		check(unsafe { ffi::bn_ctx_create(device, arena_elems as u64, &mut ctx) })?;
		Ok(Self { ctx })
	}

	/// Base address and size (in elements) of the context's arena.
	pub(crate) fn arena(&self) -> (*mut u8, usize) {
		let mut base = ptr::null_mut();
		let mut elems = 0u64;
		check(unsafe { ffi::bn_arena_base(self.ctx, &mut base, &mut elems) }).expect("arena of a live context");
		(base.cast(), elems as usize)
	}

	/// Blocks until everything enqueued on the context's stream has finished.
	pub fn sync(&self) -> Result<(), Error> {
		check(unsafe { ffi::bn_sync(self.ctx) })
	}

	/// NUMA node of the host that `device` hangs off, if the platform says.  The thread that drives a context should run
	/// there: a small sumcheck round is a host -> device -> host round trip through pinned memory, and from the other
	/// socket of a 2-socket host every one of them also crosses the socket interconnect.  The
	/// library does not touch affinities; bind the prover thread with `sched_setaffinity` / `hwloc` before creating the layer.
	pub fn device_numa_node(device: i32) -> Result<Option<u32>, Error> {
		let mut node: std::os::raw::c_int = -1;
		check(unsafe { ffi::bn_device_numa_node(device, &mut node) })?;
		Ok(if node >= 0 { Some(node as u32) } else { None })
	}
}

impl Default for Mi355xLayer {
	/// `constraint_system::prove` asks for `Hal: ComputeLayer + Default` (core/src/constraint_system/prove.rs:96).
	fn default() -> Self {
		Self::new(default_device()).expect("MI355X context")
	}
}

impl Drop for Mi355xLayer {
	fn drop(&mut self) {
		let _ = unsafe { ffi::bn_ctx_destroy(self.ctx) };
	}
}

fn step_to_ffi(step: &ArithCircuitStep<B128>) -> bn_step {
	match *step {
		ArithCircuitStep::Add(l, r) => bn_step {
			kind: ffi::BN_STEP_ADD,
			a: l as u32,
			b: r as u64,
			cst: bn_f128::default(),
		},
		ArithCircuitStep::Mul(l, r) => bn_step {
			kind: ffi::BN_STEP_MUL,
			a: l as u32,
			b: r as u64,
			cst: bn_f128::default(),
		},
		ArithCircuitStep::Pow(base, exp) => bn_step {
			kind: ffi::BN_STEP_POW,
			a: base as u32,
			b: exp,
			cst: bn_f128::default(),
		},
		ArithCircuitStep::Const(c) => bn_step {
			kind: ffi::BN_STEP_CONST,
			a: 0,
			b: 0,
			cst: to_ffi(c),
		},
		ArithCircuitStep::Var(i) => bn_step {
			kind: ffi::BN_STEP_VAR,
			a: i as u32,
			b: 0,
			cst: bn_f128::default(),
		},
	}
}

impl ComputeLayer<B128> for Mi355xLayer {
	type DevMem = Mi355xMemory;
	type Exec<'a> = Mi355xExec<'a>;

	fn copy_h2d(&self, src: &[B128], dst: &mut FSliceMut<'_, B128, Self>) -> Result<(), Error> {
		// BinaryField128b is #[repr(transparent)] over u128 (crates/field/src/binary_field.rs:115-119) = bn_f128 on a
		// little-endian host
		check(unsafe { ffi::bn_copy_h2d(self.ctx, src.as_ptr().cast(), src.len() as u64, dst.as_mut_ptr(), dst.len() as u64) })
	}

	fn copy_d2h(&self, src: FSlice<'_, B128, Self>, dst: &mut [B128]) -> Result<(), Error> {
		check(unsafe { ffi::bn_copy_d2h(self.ctx, src.as_ptr(), src.len() as u64, dst.as_mut_ptr().cast(), dst.len() as u64) })
	}

	fn copy_d2d(&self, src: FSlice<'_, B128, Self>, dst: &mut FSliceMut<'_, B128, Self>) -> Result<(), Error> {
		check(unsafe { ffi::bn_copy_d2d(self.ctx, src.as_ptr(), src.len() as u64, dst.as_mut_ptr(), dst.len() as u64) })
	}

	fn compile_expr(&self, expr: &ArithCircuit<B128>) -> Result<<Self::Exec<'_> as ComputeLayerExecutor<B128>>::ExprEval, Error> {
		let steps: Vec<bn_step> = expr.steps().iter().map(step_to_ffi).collect();
		let mut handle = ptr::null_mut();
		check(unsafe { ffi::bn_expr_compile(self.ctx, steps.as_ptr(), steps.len() as u64, &mut handle) })?;
		Ok(Mi355xExpr(handle))
	}

	fn execute<'a, 'b>(
		&'b self,
		f: impl FnOnce(&mut Self::Exec<'a>) -> Result<Vec<<Self::Exec<'a> as ComputeLayerExecutor<B128>>::OpValue>, Error>,
	) -> Result<Vec<B128>, Error>
	where
		'b: 'a,
	{
		let mut exec = Mi355xExec::new(self.ctx);
		let out = f(&mut exec)?;
		exec.flush_lines()?;
		Ok(out)
	}

	fn fill(&self, slice: &mut <Self::DevMem as ComputeMemory<B128>>::FSliceMut<'_>, value: B128) -> Result<(), Error> {
		check(unsafe { ffi::bn_fill(self.ctx, slice.as_mut_ptr(), slice.len() as u64, &to_ffi(value)) })
	}
}
