// Host-side internals of liblsfm_hip: context, device arenas, the batched map container and stage entry points.
//
// Design: all maps of one tree level live in ONE flat structure-of-arrays container on the device
// ("DevBatch"); pose / feature indices inside U and W are GLOBAL indices into that container, so a kernel
// processes every map of the level in a single launch and a pairwise join is mostly a re-indexing of features.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <condition_variable>
#include <cstring>
#include <exception>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

#include "../../include/lsfm.h"

namespace lsfm {

struct Error { int code; std::string msg; };
struct DevBatch;

#define LSFM_CHECK_HIP(expr)                                                                                   \
	do {                                                                                                        \
		hipError_t e__ = (expr);                                                                                \
		if (e__ != hipSuccess)                                                                                  \
			throw ::lsfm::Error{ LSFM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__) + " at " + __FILE__ + ":" + std::to_string(__LINE__) }; \
	} while (0)
#define LSFM_FAIL(code, m) throw ::lsfm::Error{ (code), std::string(m) + " at " + __FILE__ + ":" + std::to_string(__LINE__) }

// bump allocator over one hipMalloc'ed slab: maps and work space of a level are carved out without any
// hipMalloc/hipFree (both synchronise) inside the timed region.
struct Arena {
	char* base = nullptr;
	size_t cap = 0, off = 0, high = 0;
	void init(size_t bytes);
	void destroy();
	void* alloc_bytes(size_t bytes);
	template <class T> T* alloc(size_t n) { return static_cast<T*>(alloc_bytes(n * sizeof(T))); }
	void reset() { off = 0; }
	size_t mark() const { return off; }
	void release(size_t m) { off = m; }
};

// Zeroes everything allocated from an arena between construction and zero(): one memset for a run of accumulators
// instead of one each (a level has ~25 of them)
struct ZeroSpan {
	Arena& ar;
	size_t from;
	explicit ZeroSpan(Arena& a) : ar(a), from((a.off + 255) & ~size_t(255)) {}
	void zero(hipStream_t s) const;
};

// One tree level's maps, flat SoA on the device.  B maps; M poses, NF features, NU U blocks, NW W blocks in total.
struct DevBatch {
	int B = 0, M = 0, NF = 0, NU = 0, NW = 0;
	// per map, host mirrors (offsets have B+1 entries)
	std::vector<int> pose_off, feat_off, u_off, w_off;
	std::vector<int> Ref, FRef, ScaP, Fix, Sign, FScaP, FFix;
	// per map, device copies of the offsets
	int *d_pose_off = nullptr, *d_feat_off = nullptr;
	// poses
	double* pose = nullptr; // [M*6]
	int* pose_id = nullptr; // [M]  = -stno
	int* pose_map = nullptr; // [M]
	int* pose_origin = nullptr; // [M] index of the local map that brought the pose (drives the elimination order)
	// features
	double* feat = nullptr; // [NF*3]
	int* feat_id = nullptr; // [NF]
	int* feat_map = nullptr; // [NF]
	// information blocks (global pose / feature indices)
	double* U = nullptr; int *Ui = nullptr, *Uj = nullptr;
	double* W = nullptr; int *photo = nullptr, *feature = nullptr;
	int* fptr = nullptr;    // [NF+1] W run of each feature
	double* V = nullptr;    // [NF*9]
	// W blocks of maps a transform passed through unchanged are not copied when the batch only feeds a join: block j of
	// such a map is W_alias[j + d_alias[map]] in the transform's INPUT (d_alias[map] == INT_MIN: materialised in W)
	const double* W_alias = nullptr;
	const int* d_alias = nullptr;
	// Stereo tree levels that analyse: the sorted upper block pattern of the camera system this batch was solved with (null: not
	// kept).  The next level's pattern contains it -- a joint feature is seen by everything its sources were seen by -- so it
	// only adds the hub links and the pose pairs across the two maps of a pair (schur_pattern_early_issue)
	const unsigned long long* s_keys = nullptr;
	int s_nnzb = 0;
};

struct PcgOptions { double rel_tol = 1e-12; int max_steps = 50; bool mixed = false; int spmv_variant = 0; };

// What the first run of a tree level leaves behind for the next runs of the SAME resident tree.  Everything here is
// structure: it depends on the labels, the index arrays and the join tree of the uploaded local maps, which no run
// changes -- sizes the host needs to carve the next container (the first run reads them back from the device: a
// round trip each), the block pattern of S with its hash index, the ordering / elimination structure / supernode
// groups of the factorisation (the reference redoes its symbolic analysis in every join: cholmod_analyze_p,
// Imp.cpp:2440; here it is done once per tree shape).  With a valid plan a level is enqueued without a single
// host <-> device synchronisation.
// Index arrays of a level that depend on its structure alone and that transform and join would otherwise work out again: the
// exclusive scans of the transform's kept-block flags (k_tr_flags + two scans) and the join's common-feature matches with the ranks
// of the unmatched features (hash insert + probe + a scan).  A run that analyses finds them where the level below's preparation
// one level ahead left them (schur_pattern_prefetch computes exactly these for the counts and the pattern: they were computed twice
// until round 6); a resident tree's plan keeps copies of its own (`own`).
struct LevelIndex {
	int NU = -1, NW = -1, NF = -1;             // sizes of the level's INPUT batch they belong to
	const int *KU = nullptr, *KW = nullptr;    // [NU + 2] / [NW + 2] (transform_batch)
	const int *match = nullptr, *R = nullptr;  // [NF + 1] / [NF + 2] (join_stereo_prepare)
	std::shared_ptr<void> own;                 // plan-owned device memory; null: the arrays live in the preparer's arena
};
const int* level_index_keep(lsfm_context* ctx, LevelIndex& li, const int* src, size_t n); // a copy of src[0..n) in memory li owns

struct LevelPlan {
	bool valid = false;
	LevelIndex idx;
	std::vector<int> tr_cnt;     // transform: kept-block prefix values at the map boundaries (U then W)
	// Mono: sign of the new scale of every transformed map.  This one depends on VALUES (sign of a pose component): a
	// planned level compares it with what the device computes from the current values and flags the run when they differ
	std::vector<int> tr_sign;
	std::vector<int> join_rb;    // join: ranks of the unmatched features at the map boundaries
	std::vector<int> join_uo, join_wo; // Mono join: kept-U prefix at the map boundaries, W offsets of the joint maps
	std::shared_ptr<void> solve; // pattern of S + symbolic factorisation + iteration count (lsfm_pcg.hip)
};

// per-run accumulators on the device, read back once at the end of a run (a warm level does not stop for them)
struct RunStatsDev {
	int chol_err;          // 1 + block column of a non-positive pivot (first one wins)
	int not_converged;     // systems left above the residual bound
	int tr_err;            // 1 + map whose transform target was not found
	int plan_stale;        // a planned level met VALUES the plan does not fit (Mono: the sign of a new scale): the run is repeated without plans
	int undone;            // systems whose refinement was enqueued with a step count from an earlier run and had not met its stopping rule when the steps ran out
	int floored;           // pivots of the separators replaced by their lower bound (static pivoting, lsfm_pcg.hip k_sn_panel)
	double max_rel_residual;
	unsigned long long s_digest, factor_digest; // LSFM_FACTOR_DIGEST=1 (lsfm_stats)
	int refactor_mismatch;                      // ... systems whose second factorisation gave other bits than the first
	int s_rebuild_mismatch;                     // ... levels whose camera systems (S, E), assembled a second time from the same joint maps, were other bits
	unsigned long long k2;  // sum over the levels of sum over the features of (W run length)^2: K9's pose pairs, for its algorithmic flop count
};

// Feature-sharded joins (include/lsfm.h, lsfm_tree_set_comm): the sums that cross the GPUs.  The arrays to be reduced are
// carved out of the caller's buffer (bump allocation, restarted by every stage: a stage's arrays are consumed in stream order
// before the next stage zeroes its own), the caller's function sums them over the ranks in place.
struct Comm {
	int rank = 0, world = 1;
	lsfm_allreduce_fn fn = nullptr;
	void* user = nullptr;
	char* buf = nullptr;
	size_t cap = 0, off = 256; // (the first HDR_BYTES hold the header of the sum under way)
	// > 0: the pose-side factorisation is distributed too -- rank r owns what lies inside block r of block_maps consecutive local
	// maps of the whole tree (lsfm_tree_set_comm_blocks); 0: every rank factors everything
	int block_maps = 0;
	// Every sum is announced by a HEADER sum of 4 x int64 in the first bytes of the buffer: {ranks that failed, ranks that have not,
	// count, dtype | kind << 8} (the last two from the healthy ranks, times their number).  A healthy rank never reads it -- it
	// enqueues header and payload and goes on.  A rank whose pass threw between two sums (an error of its own: out of memory, a HIP
	// error, a buffer too small) cannot know what its peers will sum next; it FOLLOWS them instead (follow(), lsfm_prims.hip): header
	// after header it learns count and dtype of the payload, contributes zeros, and so reaches the exchange of the run's flags
	// (kind FINAL) with every collective matched -- there all ranks learn of the failure and every one of them returns an error.
	// Nobody waits for a sum the failed rank never joins (advisor, round 4).
	enum { HDR_BYTES = 256, KIND_DATA = 0, KIND_FINAL = 1 };
	bool broken = false;   // the caller's function itself failed: the communicator cannot be trusted to match anything any more
	void restart() { off = HDR_BYTES; }
	void* alloc_bytes(size_t bytes)
	{
		const size_t a = (off + 255) & ~size_t(255);
		if (a + bytes > cap)
			throw Error{ LSFM_ERR_ARG, "the buffer handed to lsfm_tree_set_comm is too small (" + std::to_string(a + bytes) + " bytes needed, " + std::to_string(cap) + " given)" };
		off = a + bytes;
		return buf + a;
	}
	template <class T> T* alloc(size_t n) { return static_cast<T*>(alloc_bytes(n * sizeof(T))); }
	// p: inside the buffer; count elements of 8 bytes (the same count on every rank)
	void allreduce(hipStream_t s, void* p, size_t count, int dtype, int kind = KIND_DATA);
	void call(hipStream_t s, size_t offset, size_t count, int dtype)
	{
		const int rc = fn(user, offset, count, dtype, (void*)s);
		if (rc) { broken = true; throw Error{ LSFM_ERR_INTERNAL, "the caller's all-reduce failed with code " + std::to_string(rc) }; }
	}
	// A rank whose pass failed: takes part in its peers' sums with zeros until they reach the exchange of the run's flags.
	// true: the peers are at that exchange (the caller now sums its flags WITHOUT a header); false: no rank is healthy any more (all
	// of them are following: nobody exchanges anything) or the communicator is broken
	bool follow(hipStream_t s);
};

// One helper thread per context for host work that the enqueuing thread need not wait for at once (the symbolic factorisation
// of the next level: lsfm_pcg.hip prefetch_next_level).  One job at a time; wait() returns when it is done and rethrows what
// it threw.  The thread lives as long as the context, so that what it keeps per thread (the symbolic analysis' workspace)
// is kept between jobs.
struct HostWorker {
	std::thread th;
	std::mutex m;
	std::condition_variable cv;
	std::function<void()> job;
	std::exception_ptr err;
	bool busy = false, quit = false;
	~HostWorker()
	{
		{ std::lock_guard<std::mutex> l(m); quit = true; }
		cv.notify_all();
		if (th.joinable()) th.join();
	}
	void run(std::function<void()> f)
	{
		std::unique_lock<std::mutex> l(m);
		cv.wait(l, [&] { return !busy; });
		job = std::move(f); busy = true; err = nullptr;
		if (!th.joinable())
			th = std::thread([this] {
				std::unique_lock<std::mutex> lk(m);
				for (;;)
				{
					cv.wait(lk, [&] { return quit || (busy && job); });
					if (quit) return;
					std::function<void()> j = std::move(job);
					job = nullptr;
					lk.unlock();
					std::exception_ptr e;
					try { j(); } catch (...) { e = std::current_exception(); }
					lk.lock();
					err = e; busy = false;
					cv.notify_all();
				}
			});
		cv.notify_all();
	}
	void wait()
	{
		std::unique_lock<std::mutex> l(m);
		cv.wait(l, [&] { return !busy; });
		if (err) { std::exception_ptr e = err; err = nullptr; std::rethrow_exception(e); }
	}
};

} // namespace lsfm

namespace lsfm {
// LSFM_ROCTX=1: the phases of a tree run as roctx ranges (tree run > level k > transform | join + solve; the marks of LSFM_TIMELINE as
// roctx marks), for `rocprofv3 --marker-trace --kernel-trace`.  The library is looked up at run time (librocprofiler-sdk-roctx /
// libroctx64): nothing is linked, nothing happens without the variable.
struct Roctx {
	int (*push)(const char*) = nullptr;
	int (*pop)() = nullptr;
	void (*mark)(const char*) = nullptr;
	Roctx();
};
Roctx& roctx();
struct Range {
	bool on;
	explicit Range(const char* name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
	~Range() { if (on) roctx().pop(); }
	Range(const Range&) = delete;
	Range& operator=(const Range&) = delete;
};
} // namespace lsfm

struct lsfm_context {
	int device = 0;
	hipStream_t stream = nullptr;
	lsfm::Arena arena[3];   // rotation: level input / transformed level / joined level
	lsfm::Arena scratch;    // per-stage work space
	int cur = 0;
	size_t arena_bytes = 0, arena_req = 0; // actual size of each arena / the request it answers (may have been capped)
	lsfm::PcgOptions pcg;
	// levels whose systems have at most this many poses take the one-launch dense path (lsfm_small.hip); 0: none.  The kernel holds 16;
	// the default is where it beats the level pipeline on the NC3500-like set (per-level measurements).  lsfm_set_small_solve
	int small_max = 5;
	std::string last_error;
	int* h_pinned = nullptr; // small pinned staging buffer for counters
	char* h_stage = nullptr; // pinned ring for small host->device copies: they are enqueued, not waited for
	char* d_stage = nullptr; // ... as the device sees it (null: not mapped -- copies then go through hipMemcpyAsync)
	size_t stage_size = 0, stage_off = 0;
	hipEvent_t ev_half[2] = { nullptr, nullptr }; // h2d_gather: a half of the ring may be refilled once its copy has left
	hipEvent_t ev0 = nullptr, ev1 = nullptr, ev2 = nullptr, ev3 = nullptr;
	hipEvent_t evs[4] = { nullptr, nullptr, nullptr, nullptr }; // stage brackets of a solve (owned here: nothing to leak on an error path)
	// Every entry point that resets or reallocates the arenas bumps this; a finished tree remembers the value it ended with,
	// and lsfm_tree_download refuses a result that a later call on the same context has overwritten
	unsigned long long generation = 0;
	// side stream: the pattern of S is built there while the caller's right-hand-side kernels run on the main stream.
	// evA = point of the main stream after which the index arrays of the joint map are complete (recorded by the caller,
	// pattern_dep set), evB = pattern ready
	hipStream_t stream2 = nullptr;
	hipEvent_t evA = nullptr, evB = nullptr;
	bool pattern_dep = false;
	// Early pattern of S (Stereo tree levels that analyse): the pose pairs of the JOINT map follow from the level's input index
	// arrays, the feature matches and the hub pose of every transformed map, all known before the transform's heavy kernels
	// run -- the pattern is put together on the side stream while those run, and the host's symbolic analysis no longer
	// waits for them (lsfm_solve.hip: schur_pattern_early_*).  tr_in / tr_hub: set by transform_batch around its hook.
	const lsfm::DevBatch* tr_in = nullptr;
	const int* tr_hub = nullptr;        // [B] global pose index of the hub column of every transformed map, -1: passed through
	std::shared_ptr<void> early;        // the build in flight (null: none)
	hipEvent_t ev_k9[2] = { nullptr, nullptr }; // K9: the 32-slot panel variant of a level runs on the side stream, beside the others (lsfm_schur_panel.hip)
	hipStream_t stream3 = nullptr;      // its own stream: the side stream carries the transform's U stage, which waits for the block kernel
	// One level ahead (Stereo tree runs that analyse): while the device factors and refines level L, the pattern of level
	// L + 1's system is put together on stream3 from level L's joint maps (their index arrays are final long before the
	// solve ends) and the host analyses it -- level L + 1 then finds its pattern and its symbolic factorisation waiting
	// (lsfm_pcg.hip: prefetch_next_level).  Their arrays live in two small arenas used in turn.
	lsfm::Arena sarena[2];
	std::shared_ptr<void> pre;          // what was prepared for the level about to run (null: nothing)
	// ... and, when an earlier run of the tree has left the refinement step count of that level, everything else the level
	// would stop for (kept-block counts of its transform, unmatched-feature ranks of its join): a plan of the level made
	// one level ahead -- the level then runs like a planned one, without a single host <-> device round trip
	lsfm::LevelPlan pre_plan;
	int pre_plan_level = -1;
	// ... whose solve part (symbolic factorisation: host work) may still be under way on the helper thread when the level starts:
	// its transform, join and Schur assembly are enqueued meanwhile, solve_batch completes the plan (lsfm_pcg.hip)
	std::shared_ptr<void> pre_pending;
	std::unique_ptr<lsfm::HostWorker> worker;
	// nothing the helper thread still reads may be dropped: wait for it, then forget what was prepared
	void drop_prepared()
	{
		if (worker) { try { worker->wait(); } catch (...) {} }
		pre.reset(); pre_pending.reset(); pre_plan = lsfm::LevelPlan(); pre_plan_level = -1;
	}
	hipEvent_t ev_solve_end = nullptr; // (LSFM_LEVEL_GAPS=1: the event behind the last solve, against the next level's first)
	double dbg_gap_ms = 0.0;
	hipEvent_t evY = nullptr, evP = nullptr; // joint index arrays of the level final (main stream) / prefetch complete (stream3)
	hipEvent_t evU = nullptr;                // the transform's U stage may start: everything it reads is final, the block kernel of the features has not begun
	hipEvent_t evK = nullptr;                // the level's Schur assembly (K9) has left the main stream: the chain of the factorisation starts
	const unsigned long long* solved_keys = nullptr; // left by solve_batch: sorted upper pattern of the system it just solved (scratch arena)
	int solved_nnzb = 0;
	// LSFM_TIMELINE=1: host wall-clock marks of a tree run (where the enqueuing thread is when), printed at the end of the run
	std::vector<std::pair<const char*, double>> timeline;
	bool timeline_on = false;
	void mark(const char* what);
	bool in_tree_run = false;
	int inject_level = -1; // tests: the level in which this rank's pass fails (LSFM_TEST_FAIL_RANK, lsfm_capi.hip), -1: none
	// refinement steps of the level being run: step_hint > 0 = what an earlier run of this tree needed here (the steps are then
	// enqueued without asking the device after each one; whether they sufficed is read at the end of the run), steps_used = what
	// a level that did ask needed
	bool level_syncs = false;                    // the level under way waits for the device at its end anyway (a Mono level that analyses): a hinted refinement may ask once
	int step_hint = 0, steps_used = 0;           // lsfm_tree_run: errors of a level may be left in d_run and read at the end of the run
	hipEvent_t evC = nullptr;
	lsfm_stats* stats = nullptr; // optional sink during a tree run
	lsfm::LevelPlan* plan = nullptr; // plan of the tree level being run (null: stage-level calls, nothing is recorded or reused)
	bool warm() const { return plan && plan->valid; }
	lsfm::RunStatsDev* d_run = nullptr; // device accumulators of the current run
	lsfm::Comm* comm = nullptr;         // feature-sharded tree run (set for the duration of lsfm_tree_run): sums cross the GPUs
	// timing of stages without stopping for them: events from a pool, elapsed times added to their sinks by flush_times()
	std::vector<hipEvent_t> ev_pool;
	size_t ev_next = 0;
	struct Timed { hipEvent_t a, b; double* sink; };
	std::vector<Timed> timed;
	hipEvent_t pool_event();
	void defer_time(hipEvent_t a, hipEvent_t b, double* sink) { timed.push_back(Timed{ a, b, sink }); }
	void flush_times(); // after the stream has been synchronised
	// bytes_each: an upper bound of what a call may need (the estimates ignore that joins merge their common features: an order of
	// magnitude at depth).  start_small: allocate an eighth of it and let grow_arenas() double it when a run exhausts an arena --
	// a cold process paid 2.5-3 s of hipMalloc for the 4 x 26 GB the NC3500-like estimate asks for, of which a run touches 3 GB
	void ensure_arenas(size_t bytes_each, bool start_small = false);
	bool grow_arenas(); // false: already at the bound
	bool arena_small = false;
	size_t arena_bound = 0; // what the arenas may grow to (the capped request)
};

namespace lsfm {

// ---- primitives (lsfm_prims.hip; rocPRIM scan / radix sort on the context stream) ------------------------
void dev_exclusive_scan(lsfm_context* ctx, const int* in, int* out, size_t n); // out[n] = total (n+1 entries written)
void dev_sort_pairs_u64(lsfm_context* ctx, unsigned long long* keys, int* vals, size_t n, int end_bit, int begin_bit = 0); // stable, bits [begin, end)
void dev_sort_keys_u64(lsfm_context* ctx, unsigned long long* keys, size_t n, int begin_bit, int end_bit);
int d2h_int(lsfm_context* ctx, const int* dptr);
void d2h_ints(lsfm_context* ctx, const int* dptr, int* h, size_t n);
void h2d(lsfm_context* ctx, void* d, const void* h, size_t bytes);
// many host pieces -> ONE contiguous device range, in order, streamed through the two halves of the pinned ring (the host
// fills one half while the other one is on the wire): a set of N local maps arrives as a dozen large copies instead of
// ~3 N pageable ones.  Synchronises before and after.
struct HostPiece { const void* p; size_t bytes; };
void h2d_gather(lsfm_context* ctx, void* d, const std::vector<HostPiece>& pieces);
// Several small copies of a stage as ONE staged transfer + ONE kernel: a level makes ~20 of them (map records, offsets, copies
// of label arrays), each a 10 us blit of its own when issued one by one -- 320 per tree, 3.4 ms of the NC3500-like tree's 41.
// Host pieces are packed into the pinned ring behind a table of (destination, source, bytes), the whole goes to the device in
// one copy, k_copy_many moves every piece (device-to-device pieces straight from their source).  Sizes are multiples of 4.
struct CopyBatch {
	lsfm_context* ctx;
	struct Item { void* dst; const void* src; size_t bytes; bool host; };
	std::vector<Item> items;
	explicit CopyBatch(lsfm_context* c) : ctx(c) {}
	void h2d(void* d, const void* h, size_t bytes) { if (bytes) items.push_back(Item{ d, h, bytes, true }); }
	void d2d(void* d, const void* s, size_t bytes) { if (bytes) items.push_back(Item{ d, s, bytes, false }); }
	void flush(); // enqueues on ctx->stream; the host sources may be freed afterwards
};
void d2h(lsfm_context* ctx, void* h, const void* d, size_t bytes);
void dev_zero(lsfm_context* ctx, void* d, size_t bytes);
void fill_async(hipStream_t s, void* d, int byte, size_t bytes); // (a kernel of the library, not hipMemsetAsync: lsfm_prims.hip)

// ---- batches (lsfm_batch.hip) -------------------------------------------------------------------------------
void batch_upload(lsfm_context* ctx, Arena& ar, const lsfm_map* maps, int N, bool mono, DevBatch& out);
void batch_download_map(lsfm_context* ctx, const DevBatch& b, int k, bool mono, lsfm_map* out);
struct CopyBatch;
// uploads pose_off / feat_off, fills pose_map / feat_map (cb != null: the uploads join the caller's batch, which then calls batch_fill_maps)
void batch_set_offsets(lsfm_context* ctx, Arena& ar, DevBatch& b, CopyBatch* cb = nullptr);
void batch_fill_maps(lsfm_context* ctx, DevBatch& b);
// A map packed into one contiguous device buffer for the hand-off between GPUs (include/lsfm.h): header, then the arrays
// with map-local indices, each 256-byte aligned.
struct PackHeader {
	int magic, version, mono, m, n, nU, nW, Ref, FRef, ScaP, Fix, Sign, FScaP, FFix, pad0, pad1;
	unsigned long long total;
	unsigned long long off[12]; // pose feat U W V | pose_id pose_origin feat_id Ui Uj photo fptr
	unsigned long long pad[11];
};
static_assert(sizeof(PackHeader) == 256, "the header is the first 256 bytes of a pack");
#define LSFM_PACK_MAGIC 0x4d46534c
size_t pack_layout(PackHeader& h); // fills off[] and total from the counts
void batch_pack_map(lsfm_context* ctx, const DevBatch& b, int k, bool mono, void* dst, size_t cap);
// the single map of `b` cut by feature label: slice g holds the features with feat_id % nslices == g (order kept) with their V
// and W blocks, and ALL poses and U blocks.  batch_slice_counts: features / W blocks of every slice (synchronises).
void batch_slice_counts(lsfm_context* ctx, const DevBatch& b, int nslices, std::vector<int>& nf, std::vector<int>& nw);
void batch_pack_slice(lsfm_context* ctx, const DevBatch& b, bool mono, int nslices, int slice, int nf, int nw, void* dst, size_t cap);
void batch_unpack_maps(lsfm_context* ctx, Arena& ar, const void* const* packed, const PackHeader* hdr, int N, bool mono, DevBatch& out);
// digest of everything a tree's plans are derived from: labels, index arrays, pose origins (synchronises)
unsigned long long batch_structure_digest(lsfm_context* ctx, const DevBatch& b);

// ---- transform (lsfm_transform.hip): K1-K4 ------------------------------------------------------------------
// target_ref[b] < 0 ... map b is passed through unchanged; otherwise the pose id the map is re-expressed in
// (Mono: target_scap / target_fix as well).  out is allocated from `ar`.
// Where the transform's W stage writes when its consumer has already laid out the next container (a join): the run of
// input feature f starts at wbase[f] of W / photo / feature, blocks are labelled newf[f], srcf[] records the input feature.
struct TrRedirect {
	const int* wbase = nullptr;
	const int* newf = nullptr;
	double* W = nullptr;
	int *photo = nullptr, *feature = nullptr, *srcf = nullptr;
};
// hook: called once everything of `out` except the information blocks exists (poses, feature values, V', run pointers,
// offsets); its answer redirects the W blocks.  keep_scratch: the caller releases the scratch arena (allocations made in
// the hook outlive the call).
void transform_batch(lsfm_context* ctx, Arena& ar, const DevBatch& in, const std::vector<int>& target_ref,
                     const std::vector<int>& target_scap, const std::vector<int>& target_fix, bool mono, DevBatch& out,
                     bool alias_passthrough = false, const std::function<TrRedirect(DevBatch&)>* hook = nullptr);

// ---- join + solve (lsfm_join.hip, lsfm_solve.hip): K5-K11 ---------------------------------------------------
struct JoinWork; // device work arrays shared between assembly and solve
// groups: consecutive maps (2g, 2g+1) of `in` are joined, a trailing unpaired map is carried over unchanged.
// Produces `out` (ceil(B/2) maps) with the solved state.  eP_out / eF_out (host, optional) receive the right-hand
// sides of group 0 when B <= 2 (stage-level C ABI).
void join_batch_stereo(lsfm_context* ctx, Arena& ar, const DevBatch& in, DevBatch& out, double* eP_out, double* eF_out);
void join_batch_mono(lsfm_context* ctx, Arena& ar, const DevBatch& in, DevBatch& out, double* eP_out, double* eF_out);

struct SolveIO {
	// system of `nseg` independent camera systems laid out back to back (block rows = poses of the batch)
	int M = 0, NF = 0, NU = 0, NW = 0, nseg = 0;
	const int* d_pose_seg = nullptr;   // [M] segment of each pose
	const int* d_feat_seg = nullptr;   // [NF]
	const unsigned char* d_seg_active = nullptr; // [nseg] 0 = carried map: state is not touched
	const double* U = nullptr; const int *Ui = nullptr, *Uj = nullptr;
	const double* W = nullptr; const int *photo = nullptr, *fptr = nullptr;
	const double* V = nullptr;
	const double* ea = nullptr;        // [M*6]
	const double* eb = nullptr;        // [NF*3]
	const double* x0 = nullptr;        // [M*6] initial guess (may be null -> zero)
	const unsigned char* d_fixed = nullptr; // [M*6] optional: 1 = scalar removed from the system (Mono gauge)
	const int* d_pose_origin = nullptr;     // [M] optional: index of the local map that brought the pose (null: its position)
	double* x_pose = nullptr;          // [M*6] out
	double* x_feat = nullptr;          // [NF*3] out
	std::vector<int> seg_rows;         // host: block rows per segment
	// optional: [nseg + 1] first pose / feature / U block of every segment (segments are contiguous ranges of the batch).  With them,
	// a level whose systems have at most 16 poses is solved by the one-launch dense path (lsfm_small.hip)
	const int *d_pose_off = nullptr, *d_feat_off = nullptr, *d_u_off = nullptr;
	// optional (Stereo tree levels on the sparse pipeline): the W part of the right-hand sides is left to the Schur assembly -- ea
	// holds U's part only, eb the V part (lsfm_solve.hpp RhsFused; null: ea / eb are complete)
	const struct RhsFused* rhs = nullptr;
	// optional (Mono tree levels that analyse): the pattern of S follows from the pattern of the level below instead of being built
	// from every pose pair of every feature again (lsfm_solve.hpp PatternSeed; null: from scratch)
	const struct PatternSeed* seed = nullptr;
};
int small_solve_strips(int most_poses, int cap); // 16-row strips of the dense path's panel; 0: the systems are too large for it (cap: lsfm_context::small_max)
void small_solve_launch(lsfm_context* ctx, const SolveIO& io, int strips, int* status, double* max_rel);
int solve_batch(lsfm_context* ctx, const SolveIO& io);
// Gauss-Newton polish of the map-joining objective over all local maps at once (lsfm_gn.hip; C ABI: lsfm_gn_polish)
int gn_polish(lsfm_context* ctx, const lsfm_map* maps, int N, bool mono, lsfm_map* x, int iters, double* obj, double* gnorm, int* halvings);
// the two feature-side pieces of the solve on their own (C ABI: lsfm_inverse_v / lsfm_solve_features); device pointers
void vinv_only(lsfm_context* ctx, int NF, const double* V, double* IV);
void backsub_only(lsfm_context* ctx, int NF, const int* fptr, const int* photo, const double* W, const double* IV, const double* eb, const double* xp, double* xf);
// what the early pattern is made from: X = the level's input batch (index arrays only), per joint feature its source
// features in X (srcE / srcC, -1: none), per map of X its hub pose
struct EarlyPatternIn {
	int M = 0, NFY = 0, NU = 0;
	const int *Ui = nullptr, *Uj = nullptr, *pose_map = nullptr, *hub = nullptr;
	const int *fptr = nullptr, *photo = nullptr, *feat_map = nullptr, *srcE = nullptr, *srcC = nullptr;
	const unsigned long long* prev_keys = nullptr; // pattern of the level below (DevBatch::s_keys), null: none
	int prev_nnzb = 0;
};
void schur_pattern_early_issue(lsfm_context* ctx, const EarlyPatternIn& in); // enqueues on the side stream; the caller has recorded evC
void schur_pattern_early_drop(lsfm_context* ctx);
// target_ref[b] of the NEXT level's transform for every map of `Y` (-1: passed through), as run_level will compute it
bool no_timing_events(); // (LSFM_NO_TIMING_EVENTS=1: the phase brackets are not recorded -- the stage times of lsfm_stats stay zero)
#define LSFM_REC_T(e, s) do { if (!lsfm::no_timing_events()) LSFM_CHECK_HIP(hipEventRecord(e, s)); } while (0)
unsigned timing_event_flags(); // (lsfm_prims.hip: events without the system-scope fence)
unsigned order_event_flags();
void prefetch_next_level(lsfm_context* ctx, const DevBatch& Y, const std::vector<int>& target_ref, int next_level, int step_hint);
// the block pattern of S alone (K8), from the index members of io: upper block CSR left in the scratch arena
void schur_pattern_only(lsfm_context* ctx, const SolveIO& io, int* nnzb, const int** rowptr, const int** colidx);
int spmv_external(lsfm_context* ctx, int m, const int* rowptr, const int* colidx, const double* val, const double* x, double* y, int reps,
                  double* avg_ms, double* bytes);

int wstream_bench(lsfm_context* ctx, long long nblocks, int mode, int reps, double* avg_ms); // measurement: W access patterns vs stream copy
} // namespace lsfm
