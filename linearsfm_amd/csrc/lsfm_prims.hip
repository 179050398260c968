// Context, arenas and the two library primitives used off the hot path (prefix sum, radix sort: rocPRIM).
#include <dlfcn.h>
#include <algorithm>
#include <chrono>
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include "lsfm_internal.hpp"

namespace lsfm {

void Arena::init(size_t bytes)
{
	destroy();
	LSFM_CHECK_HIP(hipMalloc(&base, bytes));
	cap = bytes; off = 0; high = 0;
}
void Arena::destroy()
{
	if (base) (void)hipFree(base);
	base = nullptr; cap = off = 0;
}
// Fills and the small host -> device copies of the path as kernels of the library (round 5).  hipMemsetAsync / hipMemcpyAsync are
// blit kernels or SDMA transfers of the runtime with their own hand-over on the queue: the kernel trace of a tree showed the main stream
// idle for 10-50 us around each of them (~100 fills and ~100 small copies per tree), while kernels that follow kernels start
// without a gap.  A small copy is read by the kernel straight from the pinned ring (device-visible host memory).
__global__ void __launch_bounds__(256) k_fill_words(uint4* __restrict__ d16, size_t n16, unsigned* __restrict__ tail, int ntail, unsigned v)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	const uint4 v4 = make_uint4(v, v, v, v);
	for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n16; q += stride) d16[q] = v4;
	if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = v;
}
__global__ void __launch_bounds__(256) k_fill_bytes(unsigned char* __restrict__ d, size_t n, unsigned char v)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x;
	for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n; q += stride) d[q] = v;
}
void fill_async(hipStream_t s, void* d, int byte, size_t bytes)
{
	if (!bytes) return;
	static const bool runtime_fill = getenv("LSFM_RUNTIME_FILL") != nullptr; // (as before: hipMemsetAsync)
	if (runtime_fill) { LSFM_CHECK_HIP(hipMemsetAsync(d, byte, bytes, s)); return; }
	const unsigned b = (unsigned)byte & 0xffu, v = b * 0x01010101u;
	char* p = static_cast<char*>(d);
	// (head up to a 16-byte boundary and the last bytes one by one: the arenas hand out 256-byte aligned arrays of 4- and 8-byte elements,
	// so this is the rare case)
	const size_t head = std::min(bytes, (size_t)((16 - ((size_t)p & 15)) & 15));
	if (head) hipLaunchKernelGGL(k_fill_bytes, dim3(1), dim3(64), 0, s, reinterpret_cast<unsigned char*>(p), head, (unsigned char)b);
	p += head; bytes -= head;
	const size_t n16 = bytes / 16, rest = bytes - 16 * n16, nw = rest / 4, nb = rest - 4 * nw;
	if (n16 || nw)
	{
		const unsigned grid = (unsigned)std::min<size_t>(2048, std::max<size_t>(1, (n16 + 1023) / 1024));
		hipLaunchKernelGGL(k_fill_words, dim3(grid), dim3(256), 0, s, reinterpret_cast<uint4*>(p), n16, reinterpret_cast<unsigned*>(p + 16 * n16), (int)nw, v);
	}
	if (nb) hipLaunchKernelGGL(k_fill_bytes, dim3(1), dim3(64), 0, s, reinterpret_cast<unsigned char*>(p + 16 * n16 + 4 * nw), nb, (unsigned char)b);
}
__global__ void __launch_bounds__(256) k_copy_words(unsigned* __restrict__ d, const unsigned* __restrict__ h, size_t nw)
{
	const size_t stride = (size_t)gridDim.x * blockDim.x, first = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if ((((size_t)d | (size_t)h) & 15) == 0)
	{
		const size_t n16 = nw / 4;
		uint4* o = reinterpret_cast<uint4*>(d);
		const uint4* i = reinterpret_cast<const uint4*>(h);
		for (size_t q = first; q < n16; q += stride) o[q] = i[q];
		for (size_t q = 4 * n16 + first; q < nw; q += stride) d[q] = h[q];
	}
	else
		for (size_t q = first; q < nw; q += stride) d[q] = h[q];
}
void ZeroSpan::zero(hipStream_t s) const
{
	if (ar.off > from) fill_async(s, ar.base + from, 0, ar.off - from);
}
void* Arena::alloc_bytes(size_t bytes)
{
	size_t a = (off + 255) & ~size_t(255);
	if (bytes == 0) bytes = 8;
	if (a + bytes > cap)
		LSFM_FAIL(LSFM_ERR_OOM, "device arena exhausted (need " + std::to_string((a + bytes) >> 20) + " MiB of " + std::to_string(cap >> 20) +
		                            " MiB); create the context with a larger arena_bytes");
	off = a + bytes;
	if (off > high) high = off;
	return base + a;
}

namespace {
struct IndexHold {
	std::vector<void*> p;
	~IndexHold() { for (void* q : p) (void)hipFree(q); }
};
} // namespace
const int* level_index_keep(lsfm_context* ctx, LevelIndex& li, const int* src, size_t n)
{
	if (!li.own) li.own = std::make_shared<IndexHold>();
	IndexHold* h = static_cast<IndexHold*>(li.own.get());
	void* d = nullptr;
	LSFM_CHECK_HIP(hipMalloc(&d, std::max<size_t>(n, 1) * sizeof(int)));
	h->p.push_back(d);
	if (n) LSFM_CHECK_HIP(hipMemcpyAsync(d, src, n * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
	return static_cast<const int*>(d);
}

void dev_exclusive_scan(lsfm_context* ctx, const int* in, int* out, size_t n)
{
	// scans n+1 entries (callers keep one trailing zero in `in`) so that out[n] is the total
	size_t tb = 0;
	LSFM_CHECK_HIP(rocprim::exclusive_scan(nullptr, tb, in, out, 0, n + 1, rocprim::plus<int>(), ctx->stream));
	size_t mk = ctx->scratch.mark();
	void* tmp = ctx->scratch.alloc_bytes(tb);
	LSFM_CHECK_HIP(rocprim::exclusive_scan(tmp, tb, in, out, 0, n + 1, rocprim::plus<int>(), ctx->stream));
	// the temporary may be reused by later allocations only after this launch is enqueued on the same stream
	ctx->scratch.release(mk);
}

void dev_sort_keys_u64(lsfm_context* ctx, unsigned long long* keys, size_t n, int begin_bit, int end_bit)
{
	if (n == 0 || end_bit <= begin_bit) return;
	size_t mk = ctx->scratch.mark();
	unsigned long long* k2 = ctx->scratch.alloc<unsigned long long>(n);
	size_t tb = 0;
	LSFM_CHECK_HIP(rocprim::radix_sort_keys(nullptr, tb, keys, k2, n, begin_bit, end_bit, ctx->stream));
	void* tmp = ctx->scratch.alloc_bytes(tb);
	LSFM_CHECK_HIP(rocprim::radix_sort_keys(tmp, tb, keys, k2, n, begin_bit, end_bit, ctx->stream));
	LSFM_CHECK_HIP(hipMemcpyAsync(keys, k2, n * sizeof(unsigned long long), hipMemcpyDeviceToDevice, ctx->stream));
	ctx->scratch.release(mk);
}

void dev_sort_pairs_u64(lsfm_context* ctx, unsigned long long* keys, int* vals, size_t n, int end_bit, int begin_bit)
{
	if (n == 0 || end_bit <= begin_bit) return;
	size_t mk = ctx->scratch.mark();
	unsigned long long* k2 = ctx->scratch.alloc<unsigned long long>(n);
	int* v2 = ctx->scratch.alloc<int>(n);
	size_t tb = 0;
	LSFM_CHECK_HIP(rocprim::radix_sort_pairs(nullptr, tb, keys, k2, vals, v2, n, begin_bit, end_bit, ctx->stream));
	void* tmp = ctx->scratch.alloc_bytes(tb);
	LSFM_CHECK_HIP(rocprim::radix_sort_pairs(tmp, tb, keys, k2, vals, v2, n, begin_bit, end_bit, ctx->stream));
	LSFM_CHECK_HIP(hipMemcpyAsync(keys, k2, n * sizeof(unsigned long long), hipMemcpyDeviceToDevice, ctx->stream));
	LSFM_CHECK_HIP(hipMemcpyAsync(vals, v2, n * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
	ctx->scratch.release(mk);
}

int d2h_int(lsfm_context* ctx, const int* dptr)
{
	LSFM_CHECK_HIP(hipMemcpyAsync(ctx->h_pinned, dptr, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
	LSFM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
	return ctx->h_pinned[0];
}
void d2h_ints(lsfm_context* ctx, const int* dptr, int* h, size_t n)
{
	LSFM_CHECK_HIP(hipMemcpyAsync(h, dptr, n * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
	LSFM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
}
void h2d(lsfm_context* ctx, void* d, const void* h, size_t bytes)
{
	if (!bytes) return;
	// Callers pass pageable buffers that may go out of scope.  Small copies (index arrays, per-map parameters: a dozen
	// per tree level) are staged through a pinned ring and only ENQUEUED -- stream order keeps them ahead of their
	// consumers and the host does not stop for a round trip each time.  The ring is reused only after a
	// synchronisation; larger copies wait for completion as before.
	if (ctx->h_stage && bytes <= ctx->stage_size / 4)
	{
		const size_t need = (bytes + 63) & ~(size_t)63;
		if (ctx->stage_off + need > ctx->stage_size)
		{
			LSFM_CHECK_HIP(hipDeviceSynchronize()); // (every stream of the context may have copies from the ring in flight)
			ctx->stage_off = 0;
		}
		char* slot = ctx->h_stage + ctx->stage_off;
		memcpy(slot, h, bytes);
		ctx->stage_off += need;
		static const bool runtime_copy = getenv("LSFM_RUNTIME_COPY") != nullptr; // (as before: hipMemcpyAsync from the ring)
		if (!runtime_copy && ctx->d_stage && bytes <= ((size_t)1 << 20) && bytes % 4 == 0 && ((size_t)d & 3) == 0)
		{
			// (the kernel reads the ring slot over the bus: tables of a few KB)
			const size_t nw = bytes / 4;
			const unsigned grid = (unsigned)std::min<size_t>(64, std::max<size_t>(1, (nw + 1023) / 1024));
			hipLaunchKernelGGL(k_copy_words, dim3(grid), dim3(256), 0, ctx->stream, static_cast<unsigned*>(d),
			                   reinterpret_cast<const unsigned*>(ctx->d_stage + (slot - ctx->h_stage)), nw);
			return;
		}
		LSFM_CHECK_HIP(hipMemcpyAsync(d, slot, bytes, hipMemcpyHostToDevice, ctx->stream));
		return;
	}
	LSFM_CHECK_HIP(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, ctx->stream));
	LSFM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
}
void h2d_gather(lsfm_context* ctx, void* d, const std::vector<HostPiece>& pieces)
{
	size_t total = 0;
	for (const HostPiece& pc : pieces) total += pc.bytes;
	if (!total) return;
	LSFM_CHECK_HIP(hipStreamSynchronize(ctx->stream)); // the ring is also the staging area of the small enqueued copies
	ctx->stage_off = 0;
	const size_t half = ctx->stage_size / 2;
	for (auto& e : ctx->ev_half) if (!e) LSFM_CHECK_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
	bool used[2] = { false, false };
	size_t done = 0, ip = 0, ioff = 0;
	int h = 0;
	while (done < total)
	{
		if (used[h]) LSFM_CHECK_HIP(hipEventSynchronize(ctx->ev_half[h]));
		char* buf = ctx->h_stage + (size_t)h * half;
		size_t fill = 0;
		while (fill < half && ip < pieces.size())
		{
			const size_t n = std::min(half - fill, pieces[ip].bytes - ioff);
			if (n) memcpy(buf + fill, static_cast<const char*>(pieces[ip].p) + ioff, n);
			fill += n; ioff += n;
			if (ioff == pieces[ip].bytes) { ip++; ioff = 0; }
		}
		LSFM_CHECK_HIP(hipMemcpyAsync(static_cast<char*>(d) + done, buf, fill, hipMemcpyHostToDevice, ctx->stream));
		LSFM_CHECK_HIP(hipEventRecord(ctx->ev_half[h], ctx->stream));
		used[h] = true;
		done += fill;
		h ^= 1;
	}
	LSFM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
}
struct CopyDesc { unsigned long long dst, src, bytes, pad; };
__global__ void __launch_bounds__(256) k_copy_many(const CopyDesc* __restrict__ descs)
{
	const CopyDesc d = descs[blockIdx.x];
	const size_t first = (size_t)blockIdx.y * blockDim.x + threadIdx.x, stride = (size_t)gridDim.y * blockDim.x;
	if (((d.dst | d.src | d.bytes) & 15ull) == 0)
	{
		uint4* o = reinterpret_cast<uint4*>(d.dst);
		const uint4* i = reinterpret_cast<const uint4*>(d.src);
		for (size_t q = first; q < d.bytes / 16; q += stride) o[q] = i[q];
	}
	else
	{
		unsigned* o = reinterpret_cast<unsigned*>(d.dst);
		const unsigned* i = reinterpret_cast<const unsigned*>(d.src);
		for (size_t q = first; q < d.bytes / 4; q += stride) o[q] = i[q];
	}
}
void CopyBatch::flush()
{
	if (items.empty()) return;
	if (items.size() == 1)
	{
		const Item& it = items[0];
		if (it.host) lsfm::h2d(ctx, it.dst, it.src, it.bytes);
		else LSFM_CHECK_HIP(hipMemcpyAsync(it.dst, it.src, it.bytes, hipMemcpyDeviceToDevice, ctx->stream));
		items.clear();
		return;
	}
	size_t table = items.size() * sizeof(CopyDesc), host_bytes = 0, most = 0;
	for (const Item& it : items)
	{
		if (it.bytes % 4) LSFM_FAIL(LSFM_ERR_INTERNAL, "CopyBatch: sizes must be multiples of 4 bytes");
		if (it.host) host_bytes += (it.bytes + 15) & ~(size_t)15;
		most = std::max(most, it.bytes);
	}
	const size_t total = table + host_bytes;
	if (!ctx->h_stage || total > ctx->stage_size / 4)
	{
		// too large to stage in one piece: one by one
		for (const Item& it : items)
		{
			if (it.host) lsfm::h2d(ctx, it.dst, it.src, it.bytes);
			else LSFM_CHECK_HIP(hipMemcpyAsync(it.dst, it.src, it.bytes, hipMemcpyDeviceToDevice, ctx->stream));
		}
		items.clear();
		return;
	}
	if (ctx->stage_off + total + 64 > ctx->stage_size)
	{
		LSFM_CHECK_HIP(hipDeviceSynchronize());
		ctx->stage_off = 0;
	}
	char* slot = ctx->h_stage + ctx->stage_off;
	ctx->stage_off += (total + 63) & ~(size_t)63;
	// (the table and the host pieces: read by the kernel from the ring itself where it is device-visible, else through a copy of the slot)
	static const bool runtime_copy = getenv("LSFM_RUNTIME_COPY") != nullptr;
	const bool direct = ctx->d_stage && !runtime_copy;
	char* dev = direct ? ctx->d_stage + (slot - ctx->h_stage) : static_cast<char*>(ctx->scratch.alloc_bytes(total)); // (256-byte aligned / 64: the 16-byte path applies to aligned pieces)
	CopyDesc* desc = reinterpret_cast<CopyDesc*>(slot);
	size_t off = table;
	for (size_t i = 0; i < items.size(); i++)
	{
		const Item& it = items[i];
		desc[i].dst = (unsigned long long)(size_t)it.dst; desc[i].bytes = it.bytes; desc[i].pad = 0;
		if (it.host)
		{
			memcpy(slot + off, it.src, it.bytes);
			desc[i].src = (unsigned long long)(size_t)(dev + off);
			off += (it.bytes + 15) & ~(size_t)15;
		}
		else desc[i].src = (unsigned long long)(size_t)it.src;
	}
	if (!direct) LSFM_CHECK_HIP(hipMemcpyAsync(dev, slot, total, hipMemcpyHostToDevice, ctx->stream));
	const unsigned ny = (unsigned)std::min<size_t>(256, std::max<size_t>(1, most / (64 * 1024)));
	hipLaunchKernelGGL(k_copy_many, dim3((unsigned)items.size(), ny), dim3(256), 0, ctx->stream, reinterpret_cast<const CopyDesc*>(dev));
	items.clear();
}
void d2h(lsfm_context* ctx, void* h, const void* d, size_t bytes)
{
	if (bytes) LSFM_CHECK_HIP(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, ctx->stream));
	LSFM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
}
void dev_zero(lsfm_context* ctx, void* d, size_t bytes)
{
	fill_async(ctx->stream, d, 0, bytes);
}

// ---- the sums of a feature-sharded run (Comm, lsfm_internal.hpp) ---------------------------------------------------------------
__global__ void k_comm_header(long long* h, long long failed, long long healthy, long long count, long long what)
{
	h[0] = failed; h[1] = healthy; h[2] = count; h[3] = what;
}
void Comm::allreduce(hipStream_t s, void* p, size_t count, int dtype, int kind)
{
	if (!count) return;
	if (broken) throw Error{ LSFM_ERR_INTERNAL, "the caller's all-reduce failed earlier in this run" };
	hipLaunchKernelGGL(k_comm_header, dim3(1), dim3(1), 0, s, reinterpret_cast<long long*>(buf), 0ll, 1ll, (long long)count, (long long)(dtype | (kind << 8)));
	call(s, 0, 4, LSFM_DTYPE_I64);
	call(s, (size_t)(static_cast<char*>(p) - buf), count, dtype);
}
bool Comm::follow(hipStream_t s)
{
	if (broken || !fn) return false;
	// (at most as many sums as a run of a deep tree can hold: a peer that never reaches its exchange is a bug, not a reason to spin)
	for (int guard = 0; guard < (1 << 20); guard++)
	{
		long long h[4] = { 0, 0, 0, 0 };
		hipLaunchKernelGGL(k_comm_header, dim3(1), dim3(1), 0, s, reinterpret_cast<long long*>(buf), 1ll, 0ll, 0ll, 0ll);
		call(s, 0, 4, LSFM_DTYPE_I64);
		LSFM_CHECK_HIP(hipMemcpyAsync(h, buf, sizeof h, hipMemcpyDeviceToHost, s));
		LSFM_CHECK_HIP(hipStreamSynchronize(s));
		if (h[1] <= 0) return false; // every rank is following: there is nobody left to follow
		const long long count = h[2] / h[1], what = h[3] / h[1];
		if ((what >> 8) == KIND_FINAL) return true;
		if (count <= 0 || HDR_BYTES + (size_t)count * 8 > cap) { broken = true; return false; }
		fill_async(s, buf + HDR_BYTES, 0, (size_t)count * 8); // (this rank's part of the sum: nothing)
		call(s, HDR_BYTES, (size_t)count, (int)(what & 0xff));
	}
	broken = true;
	return false;
}

} // namespace lsfm

namespace lsfm {
Roctx::Roctx()
{
	if (!getenv("LSFM_ROCTX")) return;
	for (const char* name : { "librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4" })
	{
		void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
		if (!h) continue;
		push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
		pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
		mark = reinterpret_cast<void (*)(const char*)>(dlsym(h, "roctxMarkA"));
		if (push && pop) return;
		push = nullptr; pop = nullptr; mark = nullptr;
	}
	fprintf(stderr, "liblsfm_hip: LSFM_ROCTX is set but no roctx library could be loaded -- no ranges\n");
}
Roctx& roctx() { static Roctx r; return r; }
} // namespace lsfm

void lsfm_context::mark(const char* what)
{
	if (lsfm::roctx().mark) lsfm::roctx().mark(what);
	if (!timeline_on) return;
	timeline.emplace_back(what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count());
}
namespace lsfm {
// The events of the path are recorded some 250 times per tree, between kernels of one stream.  A default event ends with a system-scope
// release -- cache write-back and invalidation for the HOST's sake -- and the kernel behind it starts 5-15 us late (rocprofv3
// kernel trace of round 5: every gap of that size on the main queue sat at a hipEventRecord).  Nothing here needs that: the events that
// bracket phases are read by hipEventElapsedTime after the run's stream synchronisation, the others order streams of ONE device
// (device-scope release); what the host reads it reads behind a stream synchronisation of its own.  LSFM_EVENTS_DEFAULT=1: as before.
static bool events_default() { static const bool v = getenv("LSFM_EVENTS_DEFAULT") != nullptr; return v; }
bool no_timing_events() { static const bool v = getenv("LSFM_NO_TIMING_EVENTS") != nullptr; return v; }
unsigned timing_event_flags() { return events_default() ? hipEventDefault : hipEventDisableSystemFence; }
unsigned order_event_flags() { return events_default() ? hipEventDisableTiming : (hipEventDisableTiming | hipEventReleaseToDevice); }
}
hipEvent_t lsfm_context::pool_event()
{
	if (ev_next == ev_pool.size())
	{
		hipEvent_t e = nullptr;
		LSFM_CHECK_HIP(hipEventCreateWithFlags(&e, lsfm::timing_event_flags()));
		ev_pool.push_back(e);
	}
	return ev_pool[ev_next++];
}
void lsfm_context::flush_times()
{
	for (const Timed& t : timed)
	{
		float ms = 0;
		if (hipEventElapsedTime(&ms, t.a, t.b) == hipSuccess && t.sink) *t.sink += ms;
	}
	timed.clear();
	ev_next = 0;
	(void)hipGetLastError();
}

static void alloc_arenas(lsfm_context* c, size_t bytes_each)
{
	// The helper thread may still be analysing the level an error interrupted (prefetch_next_level hands it a raw pointer into what
	// ctx->pre / ctx->pre_pending keep alive, and its index arrays live in sarena[]): wait for it and forget what was prepared BEFORE
	// anything is freed -- grow_arenas() is called from the handler of an LSFM_ERR_OOM thrown mid-level
	c->drop_prepared();
	c->early.reset(); c->solved_keys = nullptr; c->solved_nnzb = 0; // (pointers into the arenas about to go)
	c->arena[0].destroy(); c->arena[1].destroy(); c->arena[2].destroy(); c->scratch.destroy(); c->sarena[0].destroy(); c->sarena[1].destroy();
	c->pre.reset();
	c->arena[0].init(bytes_each);
	c->arena[1].init(bytes_each);
	c->arena[2].init(bytes_each);
	c->scratch.init(bytes_each);
	c->sarena[0].init(std::max<size_t>((size_t)96 << 20, bytes_each / 6));
	c->sarena[1].init(std::max<size_t>((size_t)96 << 20, bytes_each / 6));
	c->arena_bytes = bytes_each;
}
void lsfm_context::ensure_arenas(size_t bytes_each, bool start_small)
{
	generation++; // every caller is about to reset the arenas
	// (what is there suffices: by its size -- or, for a caller that can grow the arenas, by the bound they were made for)
	if (arena[0].base && (arena_bytes >= bytes_each || (start_small && arena_req >= bytes_each))) return;
	LSFM_CHECK_HIP(hipDeviceSynchronize());
	const size_t requested = bytes_each;
	// the estimate is an upper bound that ignores the merging of common features (an order of magnitude at depth): never
	// ask for more than a share of what the device has free; a tree that really needs more fails with LSFM_ERR_OOM at
	// the allocation that overflows its arena
	drop_prepared(); // (waits for the helper thread: nothing it reads may be freed under it)
	arena[0].destroy(); arena[1].destroy(); arena[2].destroy(); scratch.destroy(); sarena[0].destroy(); sarena[1].destroy();
	size_t free_b = 0, total_b = 0;
	if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > ((size_t)4 << 30))
		bytes_each = std::min(bytes_each, (free_b - ((size_t)2 << 30)) / 5);
	arena_bound = bytes_each;
	static const int div = getenv("LSFM_ARENA_DIV") ? std::max(1, atoi(getenv("LSFM_ARENA_DIV"))) : 8;
	arena_small = start_small && div > 1;
	static const size_t floor_b = (size_t)(getenv("LSFM_ARENA_MIN_MB") ? std::max(1, atoi(getenv("LSFM_ARENA_MIN_MB"))) : 1024) << 20; // (tests: a small floor)
	if (arena_small) bytes_each = std::min(bytes_each, std::max<size_t>(floor_b, bytes_each / div));
	alloc_arenas(this, bytes_each);
	arena_req = requested;
}
bool lsfm_context::grow_arenas()
{
	if (!arena_small || arena_bytes >= arena_bound) return false;
	(void)hipDeviceSynchronize();
	(void)hipGetLastError();
	alloc_arenas(this, std::min(arena_bound, 2 * arena_bytes));
	generation++;
	return true;
}

extern "C" {

int lsfm_context_create(int device, size_t arena_bytes, lsfm_context** out)
{
	if (!out) return LSFM_ERR_ARG;
	*out = nullptr;
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev)
	{
		fprintf(stderr, "liblsfm_hip: no usable HIP device (count=%d, requested %d) -- this library has no CPU path\n", ndev, device);
		return LSFM_ERR_NO_DEVICE;
	}
	lsfm_context* c = new lsfm_context();
	try
	{
		c->device = device;
		if (getenv("LSFM_SMALL_MAX")) c->small_max = std::max(0, std::min(16, atoi(getenv("LSFM_SMALL_MAX")))); // (measurements: tools/small_levels.py)
		LSFM_CHECK_HIP(hipSetDevice(device));
		LSFM_CHECK_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
		LSFM_CHECK_HIP(hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking));
		{
			// the early pattern's stream: its small kernels must not queue behind the work-groups of the transform's block kernels
			int least = 0, greatest = 0;
			(void)hipDeviceGetStreamPriorityRange(&least, &greatest);
			LSFM_CHECK_HIP(hipStreamCreateWithPriority(&c->stream3, hipStreamNonBlocking, greatest));
		}
		for (auto& e : c->ev_k9) LSFM_CHECK_HIP(hipEventCreateWithFlags(&e, lsfm::order_event_flags()));
		LSFM_CHECK_HIP(hipEventCreateWithFlags(&c->evA, lsfm::order_event_flags()));
		LSFM_CHECK_HIP(hipEventCreateWithFlags(&c->evB, lsfm::order_event_flags()));
		LSFM_CHECK_HIP(hipEventCreateWithFlags(&c->evC, lsfm::order_event_flags()));
		LSFM_CHECK_HIP(hipEventCreateWithFlags(&c->evY, lsfm::order_event_flags()));
		LSFM_CHECK_HIP(hipEventCreateWithFlags(&c->evP, lsfm::order_event_flags()));
		LSFM_CHECK_HIP(hipEventCreateWithFlags(&c->evK, lsfm::order_event_flags()));
		LSFM_CHECK_HIP(hipEventCreateWithFlags(&c->evU, lsfm::order_event_flags()));
		LSFM_CHECK_HIP(hipHostMalloc((void**)&c->h_pinned, 4096));
		c->stage_size = (size_t)64 << 20;
		LSFM_CHECK_HIP(hipHostMalloc((void**)&c->h_stage, c->stage_size));
		{
			void* dp = nullptr;
			if (hipHostGetDevicePointer(&dp, c->h_stage, 0) == hipSuccess) c->d_stage = static_cast<char*>(dp);
			else (void)hipGetLastError();
		}
		LSFM_CHECK_HIP(hipEventCreateWithFlags(&c->ev0, lsfm::timing_event_flags()));
		LSFM_CHECK_HIP(hipEventCreateWithFlags(&c->ev1, lsfm::timing_event_flags()));
		LSFM_CHECK_HIP(hipEventCreateWithFlags(&c->ev2, lsfm::timing_event_flags()));
		LSFM_CHECK_HIP(hipEventCreateWithFlags(&c->ev3, lsfm::timing_event_flags()));
		for (auto& e : c->evs) LSFM_CHECK_HIP(hipEventCreateWithFlags(&e, lsfm::timing_event_flags()));
		LSFM_CHECK_HIP(hipMalloc((void**)&c->d_run, sizeof(lsfm::RunStatsDev)));
		LSFM_CHECK_HIP(hipMemset(c->d_run, 0, sizeof(lsfm::RunStatsDev)));
		if (arena_bytes) c->ensure_arenas(arena_bytes);
	}
	catch (const lsfm::Error& e)
	{
		fprintf(stderr, "liblsfm_hip: %s\n", e.msg.c_str());
		int code = e.code;
		delete c;
		return code;
	}
	*out = c;
	return LSFM_OK;
}

void lsfm_context_destroy(lsfm_context* c)
{
	if (!c) return;
	c->drop_prepared();
	c->worker.reset(); // (joins the helper thread)
	(void)hipSetDevice(c->device);
	if (c->stream) (void)hipStreamSynchronize(c->stream);
	c->arena[0].destroy(); c->arena[1].destroy(); c->arena[2].destroy(); c->scratch.destroy();
	if (c->h_pinned) (void)hipHostFree(c->h_pinned);
	if (c->h_stage) (void)hipHostFree(c->h_stage);
	for (auto& e : c->ev_half) if (e) (void)hipEventDestroy(e);
	if (c->ev0) (void)hipEventDestroy(c->ev0);
	if (c->ev1) (void)hipEventDestroy(c->ev1);
	if (c->ev2) (void)hipEventDestroy(c->ev2);
	if (c->ev3) (void)hipEventDestroy(c->ev3);
	for (auto& e : c->evs) if (e) (void)hipEventDestroy(e);
	for (auto& e : c->ev_pool) if (e) (void)hipEventDestroy(e);
	if (c->d_run) (void)hipFree(c->d_run);
	if (c->stream) (void)hipStreamDestroy(c->stream);
	if (c->stream2) (void)hipStreamDestroy(c->stream2);
	if (c->stream3) (void)hipStreamDestroy(c->stream3);
	for (auto& e : c->ev_k9) if (e) (void)hipEventDestroy(e);
	if (c->evA) (void)hipEventDestroy(c->evA);
	if (c->evB) (void)hipEventDestroy(c->evB);
	if (c->evC) (void)hipEventDestroy(c->evC);
	if (c->evY) (void)hipEventDestroy(c->evY);
	if (c->evP) (void)hipEventDestroy(c->evP);
	if (c->evK) (void)hipEventDestroy(c->evK);
	if (c->evU) (void)hipEventDestroy(c->evU);
	c->pre.reset();
	c->sarena[0].destroy(); c->sarena[1].destroy();
	c->early.reset();
	delete c;
}

int lsfm_set_pcg(lsfm_context* ctx, double rel_tol, int max_steps)
{
	if (!ctx || !(rel_tol > 0) || max_steps > 50) return LSFM_ERR_ARG;
	ctx->pcg.rel_tol = rel_tol;
	ctx->pcg.max_steps = max_steps <= 0 ? 50 : max_steps;
	return LSFM_OK;
}

int lsfm_set_precision(lsfm_context* ctx, int mode)
{
	if (!ctx || (mode != 0 && mode != 1)) return LSFM_ERR_ARG;
	ctx->pcg.mixed = mode == 1;
	return LSFM_OK;
}

int lsfm_set_small_solve(lsfm_context* ctx, int max_poses)
{
	if (!ctx || max_poses < 0 || max_poses > 16) return LSFM_ERR_ARG;
	ctx->small_max = max_poses;
	return LSFM_OK;
}

int lsfm_set_spmv_variant(lsfm_context* ctx, int variant)
{
	if (!ctx || variant < 0 || variant > 2) return LSFM_ERR_ARG;
	ctx->pcg.spmv_variant = variant;
	return LSFM_OK;
}

const char* lsfm_last_error(lsfm_context* ctx) { return ctx ? ctx->last_error.c_str() : "null context"; }
void* lsfm_stream(lsfm_context* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

} // extern "C"
