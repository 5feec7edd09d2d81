// binius_amd/host/compute_layer.hpp -- C++ host-side mirror of the reference's HAL trait surface
// over the C ABI of include/binius_amd.h.
//
// The reference's host code is Rust (crates/compute/src/{layer,memory,alloc,ops}.rs); there is no
// Rust toolchain in the build image, so the host side above the boundary is written here in C++
// with the same names, argument meaning and error behaviour, one-to-one:
//
//   reference (Rust)                                        this header
//   ------------------------------------------------------  -------------------------------------
//   ComputeMemory::{FSlice,FSliceMut,..}  memory.rs:69-234   FSlice / FSliceMut / ComputeMemory
//   SubfieldSlice                         memory.rs:257-281  SubfieldSlice
//   SlicesBatch                           memory.rs:29-66    SlicesBatch
//   ComputeAllocator / BumpAllocator      alloc.rs:10-105    ComputeAllocator / BumpAllocator
//   Error                                 layer.rs:706-716   Error
//   KernelMemMap / KernelBuffer           layer.rs:595-704   KernelMemMap / KernelBuffer
//   KernelExecutor                        layer.rs:518-590   KernelExecutor   (recording)
//   ComputeLayerExecutor                  layer.rs:100-510   ComputeLayerExecutor
//   ComputeLayer                          layer.rs:22-88     ComputeLayer
//   ComputeHolder / ComputeData           layer.rs:732-776   ComputeHolder / ComputeData
//   ArithCircuit{,Step}                   math/src/arith_expr.rs:200-226   ArithCircuit
//
// INTEGRATION.md shows the Rust shim (`impl ComputeLayer<B128> for Mi355xLayer`) that forwards to
// the same C entry points; this header is that shim in C++.  It contains no field arithmetic
// except O(1) protocol scalars through bn_scalar_mul.
#pragma once

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/binius_amd.h"

namespace binius_amd {

// ---------------------------------------------------------------------------------------- field
// BinaryField128b scalar on the host: one little-endian u128 (crates/field/src/binary_field.rs:747).
struct B128 {
	uint64_t lo = 0, hi = 0;
	constexpr B128() = default;
	constexpr B128(uint64_t l, uint64_t h = 0) : lo(l), hi(h) {}
	static constexpr B128 ZERO() { return B128(0, 0); }
	static constexpr B128 ONE() { return B128(1, 0); }
	friend B128 operator+(B128 a, B128 b) { return B128(a.lo ^ b.lo, a.hi ^ b.hi); }
	friend B128 operator-(B128 a, B128 b) { return a + b; }
	B128 &operator+=(B128 b)
	{
		lo ^= b.lo;
		hi ^= b.hi;
		return *this;
	}
	friend bool operator==(B128 a, B128 b) { return a.lo == b.lo && a.hi == b.hi; }
	friend bool operator!=(B128 a, B128 b) { return !(a == b); }
	friend B128 operator*(B128 a, B128 b)
	{
		bn_f128 x{a.lo, a.hi}, y{b.lo, b.hi}, o;
		bn_scalar_mul(&x, &y, &o);
		return B128(o.lo, o.hi);
	}
	B128 invert_or_zero() const
	{
		bn_f128 x{lo, hi}, o;
		bn_scalar_invert(&x, &o);
		return B128(o.lo, o.hi);
	}
	B128 dbl() const { return ZERO(); } // Field::double() in characteristic 2
	bn_f128 raw() const { return bn_f128{lo, hi}; }
};
static_assert(sizeof(B128) == 16, "B128 must be a plain u128");

// ---------------------------------------------------------------------------------------- errors
// binius_compute::Error (layer.rs:706-716) + alloc::Error::OutOfMemory (alloc.rs:110-113)
class Error : public std::runtime_error {
public:
	enum Kind { InputValidation = 1, Alloc = 2, DeviceError = 3, CoreLibError = 4 };
	Error(Kind k, const std::string &msg) : std::runtime_error(msg), kind_(k) {}
	Kind kind() const { return kind_; }

private:
	Kind kind_;
};

inline void check(int rc)
{
	if (rc != BN_OK)
		throw Error(static_cast<Error::Kind>(rc), bn_last_error());
}

// BNH_PROF=1: time spent inside the C ABI, by kind of call (diagnostic: what is left of a prover's wall time is this mirror)
struct AbiProf {
	enum Kind { LAUNCH = 0, LINE = 1, COPY = 2, SCALAR = 3, N = 4 };
	uint64_t ns[N] = {}, calls[N] = {};
	static bool on()
	{
		static const bool v = getenv("BNH_PROF") != nullptr;
		return v;
	}
	static AbiProf &get()
	{
		static thread_local AbiProf p;
		return p;
	}
};
template <class F>
inline int abi_timed(AbiProf::Kind k, F &&f)
{
	if (!AbiProf::on()) return f();
	const auto t0 = std::chrono::steady_clock::now();
	const int rc = f();
	AbiProf &p = AbiProf::get();
	p.ns[k] += (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
	p.calls[k]++;
	return rc;
}

// ---------------------------------------------------------------------------------------- memory
// Opaque handles to slices of F in device memory.  ALIGNMENT = 1: a handle is (pointer, len) and all
// splitting is O(1) host arithmetic with no device calls (memory.rs:69-234).
struct FSlice {
	const void *ptr = nullptr;
	size_t len_ = 0;
	size_t len() const { return len_; }
	bool is_empty() const { return len_ == 0; }
};
struct FSliceMut {
	void *ptr = nullptr;
	size_t len_ = 0;
	size_t len() const { return len_; }
	bool is_empty() const { return len_ == 0; }
};

struct ComputeMemory {
	static constexpr size_t ALIGNMENT = 1;
	static FSlice narrow(const FSlice &d) { return d; }
	static FSliceMut narrow_mut(FSliceMut d) { return d; }
	static FSliceMut to_owned_mut(FSliceMut &d) { return d; }
	static FSlice as_const(const FSliceMut &d) { return FSlice{d.ptr, d.len_}; }
	static FSlice to_const(FSliceMut d) { return FSlice{d.ptr, d.len_}; }
	static FSlice slice(FSlice d, size_t start, size_t end)
	{
		if (start > end || end > d.len_) throw std::out_of_range("slice range out of bounds");
		return FSlice{static_cast<const char *>(d.ptr) + 16 * start, end - start};
	}
	static FSliceMut slice_mut(FSliceMut &d, size_t start, size_t end)
	{
		if (start > end || end > d.len_) throw std::out_of_range("slice range out of bounds");
		return FSliceMut{static_cast<char *>(d.ptr) + 16 * start, end - start};
	}
	static std::pair<FSlice, FSlice> split_at(FSlice d, size_t mid) { return {slice(d, 0, mid), slice(d, mid, d.len_)}; }
	static std::pair<FSliceMut, FSliceMut> split_at_mut(FSliceMut d, size_t mid)
	{
		return {slice_mut(d, 0, mid), slice_mut(d, mid, d.len_)};
	}
	static std::vector<FSlice> slice_chunks(FSlice d, size_t chunk_len)
	{
		if (chunk_len == 0 || d.len_ % chunk_len) throw std::invalid_argument("length not a multiple of chunk_len");
		std::vector<FSlice> out;
		for (size_t i = 0; i < d.len_; i += chunk_len) out.push_back(slice(d, i, i + chunk_len));
		return out;
	}
	static std::vector<FSliceMut> slice_chunks_mut(FSliceMut d, size_t chunk_len)
	{
		if (chunk_len == 0 || d.len_ % chunk_len) throw std::invalid_argument("length not a multiple of chunk_len");
		std::vector<FSliceMut> out;
		for (size_t i = 0; i < d.len_; i += chunk_len) out.push_back(slice_mut(d, i, i + chunk_len));
		return out;
	}
	static std::pair<FSlice, FSlice> split_half(FSlice d)
	{
		if (d.len_ < 2 || (d.len_ & (d.len_ - 1))) throw std::invalid_argument("data length must be a power of two greater than 1");
		return split_at(d, d.len_ / 2);
	}
	static std::pair<FSliceMut, FSliceMut> split_half_mut(FSliceMut d)
	{
		if (d.len_ < 2 || (d.len_ & (d.len_ - 1))) throw std::invalid_argument("data length must be a power of two greater than 1");
		return split_at_mut(d, d.len_ / 2);
	}
	static FSliceMut slice_power_of_two_mut(FSliceMut &input, size_t n)
	{
		if (input.len_ <= n) return input;
		return slice_mut(input, 0, n);
	}
};

struct SubfieldSlice {
	FSlice slice;
	size_t tower_level;
	SubfieldSlice(FSlice s, size_t level) : slice(s), tower_level(level) {}
	size_t len() const { return slice.len_ << (7 - tower_level); }
};

template <class Slice>
class SlicesBatch {
public:
	SlicesBatch(std::vector<Slice> rows, size_t row_len) : rows_(std::move(rows)), row_len_(row_len), id_(next_id())
	{
		for (const auto &r : rows_)
			if (r.len() != row_len) throw std::invalid_argument("SlicesBatch: row length mismatch");
	}
	size_t n_rows() const { return rows_.size(); }
	size_t row_len() const { return row_len_; }
	const Slice &row(size_t i) const { return rows_[i]; }
	const std::vector<Slice> &rows() const { return rows_; }
	// A batch is immutable: equal ids <=> the same rows (copies keep the id of what they copy).  Lets a recorder see that k ops
	// in a row were handed the same batch without comparing k x m rows.
	uint64_t id() const { return id_; }

private:
	static uint64_t next_id()
	{
		static std::atomic<uint64_t> counter{0};
		return ++counter;
	}
	std::vector<Slice> rows_;
	size_t row_len_;
	uint64_t id_;
};

// ---------------------------------------------------------------------------------------- alloc
// ComputeAllocator + BumpAllocator (alloc.rs:10-105).  `Mem` is FSliceMut (device) or a host span.
struct HostSliceMut {
	B128 *ptr = nullptr;
	size_t len_ = 0;
	size_t len() const { return len_; }
	B128 &operator[](size_t i) { return ptr[i]; }
};

template <class SliceMut>
class BumpAllocator {
public:
	explicit BumpAllocator(SliceMut buffer) : buffer_(buffer) {}
	SliceMut alloc(size_t n)
	{
		std::lock_guard<std::mutex> g(mu_);
		if (buffer_.len_ < n) throw Error(Error::Alloc, "allocator is out of memory");
		SliceMut lhs = buffer_;
		lhs.len_ = n;
		buffer_.ptr = advance(buffer_.ptr, n);
		buffer_.len_ -= n;
		return lhs;
	}
	size_t capacity()
	{
		std::lock_guard<std::mutex> g(mu_);
		return buffer_.len_;
	}
	// remaining capacity as a new allocator with a limited scope (alloc.rs:26-28, 98-104)
	BumpAllocator subscope_allocator()
	{
		std::lock_guard<std::mutex> g(mu_);
		return BumpAllocator(buffer_);
	}
	BumpAllocator(BumpAllocator &&o) noexcept : buffer_(o.buffer_) {}

private:
	static void *advance(void *p, size_t n) { return static_cast<char *>(p) + 16 * n; }
	static B128 *advance(B128 *p, size_t n) { return p + n; }
	std::mutex mu_;
	SliceMut buffer_;
};
using DeviceBumpAllocator = BumpAllocator<FSliceMut>;
using HostBumpAllocator = BumpAllocator<HostSliceMut>;

// ---------------------------------------------------------------------------------------- circuits
// ArithCircuit as a step list (math/src/arith_expr.rs:200-226); evaluate order = step order.
class ArithCircuit {
public:
	static ArithCircuit var(size_t index)
	{
		ArithCircuit c;
		c.steps_.push_back(step(BN_STEP_VAR, (uint32_t)index, 0, B128()));
		return c;
	}
	static ArithCircuit constant(B128 v)
	{
		ArithCircuit c;
		c.steps_.push_back(step(BN_STEP_CONST, 0, 0, v));
		return c;
	}
	friend ArithCircuit operator+(const ArithCircuit &a, const ArithCircuit &b) { return combine(a, b, BN_STEP_ADD); }
	friend ArithCircuit operator*(const ArithCircuit &a, const ArithCircuit &b) { return combine(a, b, BN_STEP_MUL); }
	ArithCircuit &operator*=(const ArithCircuit &b) { return *this = *this * b; }
	ArithCircuit pow(uint64_t e) const
	{
		ArithCircuit c = *this;
		c.steps_.push_back(step(BN_STEP_POW, (uint32_t)steps_.size() - 1, e, B128()));
		return c;
	}
	ArithCircuit remap_vars(const std::vector<size_t> &indices) const
	{
		ArithCircuit c = *this;
		for (auto &s : c.steps_)
			if (s.kind == BN_STEP_VAR) {
				if (s.a >= indices.size()) throw std::out_of_range("remap_vars: index out of range");
				s.a = (uint32_t)indices[s.a];
			}
		return c;
	}
	size_t n_vars() const
	{
		size_t n = 0;
		for (const auto &s : steps_)
			if (s.kind == BN_STEP_VAR && (size_t)s.a + 1 > n) n = s.a + 1;
		return n;
	}
	const std::vector<bn_step> &steps() const { return steps_; }
	// a circuit given as its step list (ArithCircuit::steps, math/src/arith_expr.rs:200-226): what crosses the C boundary
	static ArithCircuit from_steps(std::vector<bn_step> steps)
	{
		ArithCircuit c;
		c.steps_ = std::move(steps);
		return c;
	}

private:
	static bn_step step(uint32_t kind, uint32_t a, uint64_t b, B128 c)
	{
		bn_step s;
		s.kind = kind;
		s.a = a;
		s.b = b;
		s.cst = c.raw();
		return s;
	}
	static ArithCircuit combine(const ArithCircuit &a, const ArithCircuit &b, uint32_t kind)
	{
		ArithCircuit c = a;
		const uint32_t off = (uint32_t)a.steps_.size();
		for (bn_step s : b.steps_) {
			if (s.kind == BN_STEP_ADD || s.kind == BN_STEP_MUL) {
				s.a += off;
				s.b += off;
			} else if (s.kind == BN_STEP_POW) {
				s.a += off;
			}
			c.steps_.push_back(s);
		}
		c.steps_.push_back(step(kind, off - 1, off + (uint32_t)b.steps_.size() - 1, B128()));
		return c;
	}
	std::vector<bn_step> steps_;
};

// ExprEval: the handle compile_expr returns (layer.rs:57-60)
class ExprEval {
public:
	ExprEval() = default;
	ExprEval(bn_expr *h, size_t n_vars) : h_(h, [](bn_expr *e) { bn_expr_free(e); }), n_vars_(n_vars) {}
	const bn_expr *handle() const { return h_.get(); }
	size_t n_vars() const { return n_vars_; }

private:
	std::shared_ptr<bn_expr> h_;
	size_t n_vars_ = 0;
};

// ---------------------------------------------------------------------------------------- kernels
// KernelMemMap (layer.rs:595-612)
struct KernelMemMap {
	enum Kind { Chunked = BN_MAP_CHUNKED, ChunkedMut = BN_MAP_CHUNKED_MUT, Local = BN_MAP_LOCAL } kind;
	FSliceMut data{}; // Chunked: read-only view; ChunkedMut: writable
	size_t log_min_chunk_size = 0;
	size_t log_size = 0; // Local
	static KernelMemMap chunked(FSlice d, size_t log_min_chunk_size)
	{
		return KernelMemMap{Chunked, FSliceMut{const_cast<void *>(d.ptr), d.len_}, log_min_chunk_size, 0};
	}
	static KernelMemMap chunked_mut(FSliceMut d, size_t log_min_chunk_size) { return KernelMemMap{ChunkedMut, d, log_min_chunk_size, 0}; }
	static KernelMemMap local(size_t log_size) { return KernelMemMap{Local, FSliceMut{}, 0, log_size}; }
	bn_memmap raw() const
	{
		bn_memmap m;
		m.kind = (uint32_t)kind;
		m.log_min_chunk_size = (uint32_t)log_min_chunk_size;
		m.d_data = data.ptr;
		m.len = data.len_;
		m.log_size = (uint32_t)log_size;
		return m;
	}
	// KernelMemMap::log_chunks_range (layer.rs:617-644); empty optional <=> no mappings
	static bool log_chunks_range(const std::vector<KernelMemMap> &maps, size_t &start, size_t &end)
	{
		if (maps.empty()) return false;
		std::vector<bn_memmap> raw;
		for (const auto &m : maps) raw.push_back(m.raw());
		uint32_t s = 0, e = 0;
		check(bn_log_chunks_range(raw.data(), (uint32_t)raw.size(), &s, &e));
		start = s;
		end = e;
		return true;
	}
};

// A kernel-local slice: chunk-relative view of mapped buffer `buf` (the kernel's FSlice type).
struct KSlice {
	uint32_t buf = 0;
	size_t off = 0, len_ = 0;
	size_t len() const { return len_; }
	KSlice slice(size_t start, size_t end) const
	{
		if (start > end || end > len_) throw std::out_of_range("kernel slice range out of bounds");
		return KSlice{buf, off + start, end - start};
	}
	bn_kslice raw() const { return bn_kslice{buf, off, len_}; }
};
using KSliceMut = KSlice;

// KernelBuffer::{Ref, Mut} (layer.rs:682-704)
struct KernelBuffer {
	KSlice s;
	bool is_mut = false;
	KSlice to_ref() const { return s; }
	KSliceMut &as_mut()
	{
		if (!is_mut) throw std::logic_error("KernelBuffer::Ref used as Mut");
		return s;
	}
	size_t len() const { return s.len_; }
};

struct KernelValue {
	uint32_t id;
};

// Recording KernelExecutor (layer.rs:518-590): the kernel-spec closure is run once against it.
class KernelExecutor {
public:
	using Value = KernelValue;
	Value decl_value(B128 init)
	{
		bn_kop op{};
		op.kind = BN_KOP_DECL_VALUE;
		op.value = n_values_;
		op.scalar = init.raw();
		ops_.push_back(op);
		return Value{n_values_++};
	}
	void sum_composition_evals(const SlicesBatch<KSlice> &inputs, const ExprEval &composition, B128 batch_coeff, Value &accumulator)
	{
		bn_kop op{};
		op.kind = BN_KOP_SUM_COMPOSITION;
		op.value = accumulator.id;
		op.scalar = batch_coeff.raw();
		op.expr = composition.handle();
		// (a prover with k claims passes the SAME batch of m rows k times in a row, v3/bivariate_product.rs:355-399: the recorded ops
		// share one copy of it -- told by the batch's id, not its address: a new batch may live where the last one did)
		const bool same = !rows_.empty() && last_batch_id_ == inputs.id();
		if (!same) {
			last_batch_id_ = inputs.id();
			rows_.emplace_back();
			rows_.back().reserve(inputs.n_rows());
			for (const auto &r : inputs.rows()) rows_.back().push_back(r.raw());
		}
		op.n_rows = (uint32_t)inputs.n_rows();
		row_index_.push_back({ops_.size(), rows_.size() - 1});
		ops_.push_back(op);
		keep_.push_back(composition);
	}
	void add(size_t log_len, KSlice src1, KSlice src2, KSliceMut &dst)
	{
		if (src1.len_ != (size_t)1 << log_len || src2.len_ != (size_t)1 << log_len || dst.len_ != (size_t)1 << log_len)
			throw std::logic_error("add: slice lengths must equal 1 << log_len");
		bn_kop op{};
		op.kind = BN_KOP_ADD;
		op.src1 = src1.raw();
		op.src2 = src2.raw();
		op.dst = dst.raw();
		ops_.push_back(op);
	}
	void add_assign(size_t log_len, KSlice src, KSliceMut &dst)
	{
		if (src.len_ != (size_t)1 << log_len || dst.len_ != (size_t)1 << log_len)
			throw std::logic_error("add_assign: slice lengths must equal 1 << log_len");
		bn_kop op{};
		op.kind = BN_KOP_ADD_ASSIGN;
		op.src1 = src.raw();
		op.dst = dst.raw();
		ops_.push_back(op);
	}
	// finalise row pointers (vectors may have moved while recording)
	std::vector<bn_kop> &finish()
	{
		for (const auto &ri : row_index_) ops_[ri.first].rows = rows_[ri.second].data();
		return ops_;
	}

private:
	std::vector<bn_kop> ops_;
	std::vector<std::vector<bn_kslice>> rows_;
	std::vector<std::pair<size_t, size_t>> row_index_; // (op, its rows_ entry)
	std::vector<ExprEval> keep_;
	uint64_t last_batch_id_ = 0;
	uint32_t n_values_ = 0;
};

// ---------------------------------------------------------------------------------------- executor
class ComputeLayer;

// ComputeLayerExecutor (layer.rs:100-510).  OpValue is a resolved scalar: ops that return scalars
// synchronise the stream (the ABI can also leave them on the device -- see bn_kernel_launch d_out).
class ComputeLayerExecutor {
public:
	using OpValue = B128;
	using KernelExec = KernelExecutor;
	explicit ComputeLayerExecutor(bn_ctx *ctx) : ctx_(ctx) {}

	template <class Op1, class Op2>
	auto join(Op1 &&op1, Op2 &&op2)
	{
		auto out1 = op1(*this);
		auto out2 = op2(*this);
		return std::make_pair(std::move(out1), std::move(out2));
	}
	// map (layer.rs:126-132): the items are independent, so extrapolate_line calls issued inside the
	// scope are deferred and launched together (one kernel per group of equal (len, z)); the
	// store-to-load order the executor contract requires (layer.rs:92-95) is kept because the
	// deferred work is flushed before map returns.
	template <class Iter, class Fn>
	auto map(Iter begin, Iter end, Fn &&f)
	{
		std::vector<decltype(f(*this, *begin))> out;
		const bool outer = !batching_;
		batching_ = true;
		try {
			for (auto it = begin; it != end; ++it) out.push_back(f(*this, *it));
		} catch (...) {
			if (outer) {
				batching_ = false;
				pending_.clear();
			}
			throw;
		}
		if (outer) {
			batching_ = false;
			flush_pending();
		}
		return out;
	}

	using KernelFn = std::function<std::vector<KernelValue>(KernelExecutor &, size_t, std::vector<KernelBuffer> &)>;
	using MapKernelFn = std::function<void(KernelExecutor &, size_t, std::vector<KernelBuffer> &)>;

	std::vector<OpValue> accumulate_kernels(const KernelFn &map, const std::vector<KernelMemMap> &mem_maps)
	{
		std::vector<bn_memmap> raw;
		size_t log_chunks;
		std::vector<KernelBuffer> bufs = setup(mem_maps, raw, log_chunks);
		KernelExecutor ke;
		std::vector<KernelValue> rets = map(ke, log_chunks, bufs);
		std::vector<uint32_t> ids;
		for (auto v : rets) ids.push_back(v.id);
		std::vector<bn_f128> out(ids.size() ? ids.size() : 1);
		auto &ops = ke.finish();
		check(abi_timed(AbiProf::LAUNCH, [&] {
			return bn_kernel_launch(ctx_, raw.data(), (uint32_t)raw.size(), ops.data(), (uint32_t)ops.size(), ids.data(), (uint32_t)ids.size(),
			                        (uint32_t)log_chunks, ids.empty() ? nullptr : out.data(), nullptr);
		}));
		std::vector<OpValue> res;
		for (size_t i = 0; i < ids.size(); i++) res.emplace_back(out[i].lo, out[i].hi);
		return res;
	}

	// Same as accumulate_kernels but the accumulated scalars stay on the device in d_out (no stream
	// synchronisation): a deferred OpValue, e.g. to feed an RCCL collective in the sharded prover.
	void accumulate_kernels_to_device(const KernelFn &map, const std::vector<KernelMemMap> &mem_maps, void *d_out)
	{
		std::vector<bn_memmap> raw;
		size_t log_chunks;
		std::vector<KernelBuffer> bufs = setup(mem_maps, raw, log_chunks);
		KernelExecutor ke;
		std::vector<KernelValue> rets = map(ke, log_chunks, bufs);
		std::vector<uint32_t> ids;
		for (auto v : rets) ids.push_back(v.id);
		auto &ops = ke.finish();
		check(bn_kernel_launch(ctx_, raw.data(), (uint32_t)raw.size(), ops.data(), (uint32_t)ops.size(), ids.data(), (uint32_t)ids.size(),
		                       (uint32_t)log_chunks, nullptr, d_out));
	}

	void map_kernels(const MapKernelFn &map, const std::vector<KernelMemMap> &mem_maps)
	{
		std::vector<bn_memmap> raw;
		size_t log_chunks;
		std::vector<KernelBuffer> bufs = setup(mem_maps, raw, log_chunks);
		KernelExecutor ke;
		map(ke, log_chunks, bufs);
		auto &ops = ke.finish();
		check(bn_kernel_launch(ctx_, raw.data(), (uint32_t)raw.size(), ops.data(), (uint32_t)ops.size(), nullptr, 0, (uint32_t)log_chunks,
		                       nullptr, nullptr));
	}

	OpValue inner_product(const SubfieldSlice &a_in, FSlice b_in)
	{
		bn_f128 out;
		check(bn_inner_product(ctx_, a_in.slice.ptr, a_in.slice.len_, (uint32_t)a_in.tower_level, b_in.ptr, b_in.len_, &out));
		return B128(out.lo, out.hi);
	}
	void tensor_expand(size_t log_n, const std::vector<B128> &coordinates, FSliceMut &data)
	{
		check(bn_tensor_expand(ctx_, data.ptr, data.len_, (uint32_t)log_n, reinterpret_cast<const bn_f128 *>(coordinates.data()),
		                       (uint32_t)coordinates.size()));
	}
	void fold_left(const SubfieldSlice &mat, FSlice vec, FSliceMut &out)
	{
		check(bn_fold_left(ctx_, mat.slice.ptr, mat.slice.len_, (uint32_t)mat.tower_level, vec.ptr, vec.len_, out.ptr, out.len_));
	}
	void fold_right(const SubfieldSlice &mat, FSlice vec, FSliceMut &out)
	{
		check(bn_fold_right(ctx_, mat.slice.ptr, mat.slice.len_, (uint32_t)mat.tower_level, vec.ptr, vec.len_, out.ptr, out.len_));
	}
	// `ntt` is passed as its twiddle basis (see AdditiveNTT below)
	template <class NTT>
	void fri_fold(const NTT &ntt, size_t log_len, size_t log_batch_size, const std::vector<B128> &challenges, FSlice data_in,
	              FSliceMut &data_out)
	{
		check(bn_fri_fold(ctx_, ntt.s_evals(), (uint32_t)ntt.tower_level(), (uint32_t)ntt.log_domain_size(), (uint32_t)log_len,
		                  (uint32_t)log_batch_size, reinterpret_cast<const bn_f128 *>(challenges.data()), (uint32_t)challenges.size(),
		                  data_in.ptr, data_in.len_, data_out.ptr, data_out.len_));
	}
	void extrapolate_line(FSliceMut &evals_0, FSlice evals_1, B128 z)
	{
		if (evals_0.len_ != evals_1.len_) throw Error(Error::InputValidation, "evals_0 and evals_1 must be the same length");
		if (batching_) {
			pending_.push_back(PendingLine{evals_0.ptr, evals_1.ptr, evals_0.len_, z, false, B128()});
			return;
		}
		bn_f128 zz = z.raw();
		check(bn_extrapolate_line(ctx_, evals_0.ptr, evals_0.len_, evals_1.ptr, evals_1.len_, &zz));
	}
	// Extension (bn_extrapolate_line_batch_scaled): extrapolate_line, then the upper half of evals_0 times hi_scale
	void extrapolate_line_scaled(FSliceMut &evals_0, FSlice evals_1, B128 z, B128 hi_scale)
	{
		if (evals_0.len_ != evals_1.len_) throw Error(Error::InputValidation, "evals_0 and evals_1 must be the same length");
		if (batching_) {
			pending_.push_back(PendingLine{evals_0.ptr, evals_1.ptr, evals_0.len_, z, true, hi_scale});
			return;
		}
		void *e0 = evals_0.ptr;
		const void *e1 = evals_1.ptr;
		const bn_f128 zz = z.raw(), hs = hi_scale.raw();
		check(bn_extrapolate_line_batch_scaled(ctx_, &e0, &e1, 1, evals_0.len_, &zz, 1, &hs));
	}
	void compute_composite(const SlicesBatch<FSlice> &inputs, FSliceMut &output, const ExprEval &composition)
	{
		std::vector<const void *> rows;
		for (const auto &r : inputs.rows()) rows.push_back(r.ptr);
		check(bn_compute_composite(ctx_, rows.data(), (uint32_t)rows.size(), inputs.row_len(), output.ptr, output.len_, composition.handle()));
	}
	void pairwise_product_reduce(FSlice input, std::vector<FSliceMut> &round_outputs)
	{
		std::vector<void *> outs;
		std::vector<uint64_t> lens;
		for (auto &r : round_outputs) {
			outs.push_back(r.ptr);
			lens.push_back(r.len_);
		}
		check(bn_pairwise_product_reduce(ctx_, input.ptr, input.len_, outs.data(), lens.data(), (uint32_t)outs.size()));
	}
	bn_ctx *raw_ctx() const { return ctx_; }

private:
	struct PendingLine {
		void *e0;
		const void *e1;
		size_t len;
		B128 z;
		bool scaled = false;
		B128 hi_scale{};
	};
	void flush_pending()
	{
		std::vector<PendingLine> todo;
		todo.swap(pending_);
		size_t i = 0;
		while (i < todo.size()) {
			std::vector<void *> e0;
			std::vector<const void *> e1;
			size_t j = i;
			uint32_t mask = 0;
			B128 hs{};
			// (BN_FOLD_CALL_MAX slices per call -- a whole prover's fold is ONE deferred batch on the backend; a batch that carries
			// scaled slices stays within the 32 bits of the mask)
			while (j < todo.size() && e0.size() < (size_t)BN_FOLD_CALL_MAX && !((mask || todo[j].scaled) && e0.size() >= 32) && todo[j].len == todo[i].len &&
			       todo[j].z == todo[i].z && !(todo[j].scaled && mask && !(todo[j].hi_scale == hs))) {
				if (todo[j].scaled) {
					mask |= 1u << e0.size();
					hs = todo[j].hi_scale;
				}
				e0.push_back(todo[j].e0);
				e1.push_back(todo[j].e1);
				j++;
			}
			const bn_f128 zz = todo[i].z.raw(), hsr = hs.raw();
			check(abi_timed(AbiProf::LINE, [&] {
				return bn_extrapolate_line_batch_scaled(ctx_, e0.data(), e1.data(), (uint32_t)e0.size(), todo[i].len, &zz, mask, mask ? &hsr : nullptr);
			}));
			i = j;
		}
	}
	bool batching_ = false;
	std::vector<PendingLine> pending_;
	std::vector<KernelBuffer> setup(const std::vector<KernelMemMap> &mem_maps, std::vector<bn_memmap> &raw, size_t &log_chunks)
	{
		if (mem_maps.empty()) throw Error(Error::InputValidation, "kernel launch needs at least one mapping");
		for (const auto &m : mem_maps) raw.push_back(m.raw());
		uint32_t lc = 0;
		check(bn_pick_log_chunks(raw.data(), (uint32_t)raw.size(), &lc));
		log_chunks = lc;
		std::vector<KernelBuffer> bufs;
		for (uint32_t i = 0; i < mem_maps.size(); i++) {
			const auto &m = mem_maps[i];
			const size_t total = m.kind == KernelMemMap::Local ? ((size_t)1 << m.log_size) : m.data.len_;
			bufs.push_back(KernelBuffer{KSlice{i, 0, total >> log_chunks}, m.kind != KernelMemMap::Chunked});
		}
		return bufs;
	}
	bn_ctx *ctx_;
};

// ---------------------------------------------------------------------------------------- layer
// ComputeLayer (layer.rs:22-88).  Non-copyable owner of the device context.
class ComputeLayer {
public:
	using Exec = ComputeLayerExecutor;
	explicit ComputeLayer(int device = 0, size_t device_arena_elems = 0)
	{
		check(bn_ctx_create(device, device_arena_elems, &ctx_));
	}
	// adopt a context created elsewhere (not destroyed by this object)
	explicit ComputeLayer(bn_ctx *external) : ctx_(external), owned_(false) {}
	~ComputeLayer()
	{
		if (ctx_ && owned_) bn_ctx_destroy(ctx_);
	}
	ComputeLayer(const ComputeLayer &) = delete;
	ComputeLayer &operator=(const ComputeLayer &) = delete;

	void copy_h2d(const B128 *src, size_t src_len, FSliceMut &dst)
	{
		check(bn_copy_h2d(ctx_, reinterpret_cast<const bn_f128 *>(src), src_len, dst.ptr, dst.len_));
	}
	void copy_h2d(const std::vector<B128> &src, FSliceMut &dst) { copy_h2d(src.data(), src.size(), dst); }
	void copy_d2h(FSlice src, B128 *dst, size_t dst_len)
	{
		check(abi_timed(AbiProf::COPY, [&] { return bn_copy_d2h(ctx_, src.ptr, src.len_, reinterpret_cast<bn_f128 *>(dst), dst_len); }));
	}
	void copy_d2h(FSlice src, std::vector<B128> &dst) { copy_d2h(src, dst.data(), dst.size()); }
	void copy_d2d(FSlice src, FSliceMut &dst)
	{
		check(abi_timed(AbiProf::COPY, [&] { return bn_copy_d2d(ctx_, src.ptr, src.len_, dst.ptr, dst.len_); }));
	}
	ExprEval compile_expr(const ArithCircuit &expr)
	{
		bn_expr *h = nullptr;
		check(bn_expr_compile(ctx_, expr.steps().data(), expr.steps().size(), &h));
		return ExprEval(h, expr.n_vars());
	}
	// execute (layer.rs:65-72): f receives the executor and returns the scalars it wants resolved
	template <class Fn>
	std::vector<B128> execute(Fn &&f)
	{
		ComputeLayerExecutor exec(ctx_);
		return f(exec);
	}
	void fill(FSliceMut &slice, B128 value)
	{
		bn_f128 v = value.raw();
		check(bn_fill(ctx_, slice.ptr, slice.len_, &v));
	}
	FSliceMut device_arena()
	{
		void *base = nullptr;
		uint64_t n = 0;
		check(bn_arena_base(ctx_, &base, &n));
		return FSliceMut{base, (size_t)n};
	}
	void sync() { check(bn_sync(ctx_)); }
	bn_ctx *raw_ctx() const { return ctx_; }

private:
	bn_ctx *ctx_ = nullptr;
	bool owned_ = true;
};

// ComputeHolder / ComputeData (layer.rs:732-776): the popular triple (hal, host_alloc, dev_alloc),
// constructed like FastCpuLayerHolder::new(host_mem_size, dev_mem_size).
struct ComputeData {
	ComputeLayer *hal;
	HostBumpAllocator host_alloc;
	DeviceBumpAllocator dev_alloc;
};

class ComputeHolder {
public:
	ComputeHolder(size_t host_mem_size, size_t dev_mem_size, int device = 0) : layer_(device, dev_mem_size), host_mem_(host_mem_size) {}
	ComputeData to_data()
	{
		return ComputeData{&layer_, HostBumpAllocator(HostSliceMut{host_mem_.data(), host_mem_.size()}),
		                   DeviceBumpAllocator(layer_.device_arena())};
	}
	ComputeLayer &layer() { return layer_; }

private:
	ComputeLayer layer_;
	std::vector<B128> host_mem_;
};

// ---------------------------------------------------------------------------------------- NTT
// AdditiveNTT over the canonical subspace with on-the-fly twiddles
// (SingleThreadedNTT::new, crates/ntt/src/single_threaded.rs:27-30; trait additive_ntt.rs:58-166).
struct NTTShape {
	size_t log_x = 0, log_y = 0, log_z = 0;
};

class AdditiveNTT {
public:
	AdditiveNTT(ComputeLayer &hal, size_t tower_level, size_t log_domain_size)
	    : hal_(&hal), level_(tower_level), log_domain_(log_domain_size), s_evals_(BN_NTT_MAX_DIM * BN_NTT_MAX_DIM)
	{
		check(bn_ntt_s_evals((uint32_t)tower_level, (uint32_t)log_domain_size, s_evals_.data()));
	}
	size_t log_domain_size() const { return log_domain_; }
	size_t tower_level() const { return level_; }
	const uint64_t *s_evals() const { return s_evals_.data(); }
	// get_subspace_eval(i, j) = s_evals[log_domain - i].get(j)  (single_threaded.rs:91-93)
	uint64_t get_subspace_eval(size_t i, size_t j) const
	{
		const size_t layer = log_domain_ - i;
		const uint64_t *row = &s_evals_[layer * BN_NTT_MAX_DIM];
		uint64_t t = 0;
		for (size_t b = 0; b + 1 + layer < log_domain_; b++)
			if ((j >> b) & 1) t ^= row[b];
		return t;
	}
	// data: device pointer to 2^(log_x+log_y+log_z) elements of T_elem_level
	void forward_transform(void *d_data, size_t elem_level, NTTShape shape, size_t coset, size_t coset_bits, size_t skip_rounds) const
	{
		check(bn_ntt_forward(hal_->raw_ctx(), d_data, (uint32_t)elem_level, (uint32_t)level_, s_evals_.data(), (uint32_t)log_domain_,
		                     (uint32_t)shape.log_x, (uint32_t)shape.log_y, (uint32_t)shape.log_z, coset, (uint32_t)coset_bits,
		                     (uint32_t)skip_rounds));
	}
	void inverse_transform(void *d_data, size_t elem_level, NTTShape shape, size_t coset, size_t coset_bits, size_t skip_rounds) const
	{
		check(bn_ntt_inverse(hal_->raw_ctx(), d_data, (uint32_t)elem_level, (uint32_t)level_, s_evals_.data(), (uint32_t)log_domain_,
		                     (uint32_t)shape.log_x, (uint32_t)shape.log_y, (uint32_t)shape.log_z, coset, (uint32_t)coset_bits,
		                     (uint32_t)skip_rounds));
	}

private:
	ComputeLayer *hal_;
	size_t level_, log_domain_;
	std::vector<uint64_t> s_evals_;
};

} // namespace binius_amd
