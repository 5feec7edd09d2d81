//! `ComputeHolder`: owns the layer, a host scratch vector and the device arena, and hands out bump
//! allocators over both -- the counterpart of `FastCpuLayerHolder` (crates/fast_compute/src/layer.rs:905-952).

use binius_compute::{
	alloc::{BumpAllocator, HostBumpAllocator},
	layer::{ComputeData, ComputeHolder},
};
use binius_field::{BinaryField128b as B128, Field};

use crate::{
	Mi355xLayer,
	memory::{DevSliceMut, Mi355xMemory},
};

pub struct Mi355xLayerHolder {
	layer: Mi355xLayer,
	host_mem: Vec<B128>,
}

impl Mi355xLayerHolder {
	/// `host_mem_size` / `dev_mem_size` in field elements, as for the CPU holders.  The device arena is one
	/// `hipMalloc` owned by the backend context (`bn_ctx_create(device, arena_elems)`).
	pub fn new(host_mem_size: usize, dev_mem_size: usize) -> Self {
		Self::on_device(crate::default_device(), host_mem_size, dev_mem_size)
	}

	pub fn on_device(device: i32, host_mem_size: usize, dev_mem_size: usize) -> Self {
		let layer = Mi355xLayer::with_arena(device, dev_mem_size.max(1)).expect("MI355X context with a device arena");
		Self {
			layer,
			host_mem: vec![B128::ZERO; host_mem_size],
		}
	}
}

impl ComputeHolder<B128, Mi355xLayer> for Mi355xLayerHolder {
	type HostComputeAllocator<'a> = HostBumpAllocator<'a, B128>;
	type DeviceComputeAllocator<'a> = BumpAllocator<'a, B128, Mi355xMemory>;

	fn to_data<'a, 'b>(&'a mut self) -> ComputeData<'a, B128, Mi355xLayer, Self::HostComputeAllocator<'b>, Self::DeviceComputeAllocator<'b>>
	where
		'a: 'b,
	{
		let (base, elems) = self.layer.arena();
		ComputeData::new(
			&self.layer,
			BumpAllocator::new(self.host_mem.as_mut_slice()),
			BumpAllocator::new(DevSliceMut::from_raw(base, elems)),
		)
	}
}
