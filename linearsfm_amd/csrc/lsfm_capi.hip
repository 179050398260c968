// C ABI (include/lsfm.h) and the tree scheduler that replaces lmj_PF3D_Divide_Conquer{Stereo,Mono}
// (Imp.cpp:1926-2063 / 6511-6630): every level of the reference's binary join tree runs as ONE batch.
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "lsfm_internal.hpp"
#include "lsfm_join.hpp"

using namespace lsfm;

struct lsfm_tree {
	bool mono = false;
	int N = 0;
	Arena input_arena; // pristine copy of the N local maps, resident in HBM; lsfm_tree_run starts from a device copy of it
	DevBatch input;
	DevBatch level;    // current level (lives in ctx->arena[slot]; slot -1: the resident inputs)
	int slot = 0;
	bool done = false;
	bool final_reanchor = true;
	int stop_level = 0; // > 0: a run ends after this many tree levels (lsfm_tree_set_stop_level)
	unsigned long long generation = 0; // ctx->generation when the run ended: the result lives in the context's arenas
	// what the first run leaves for the next ones (structure only: the resident inputs never change): one plan per tree
	// level + one for the final re-anchoring transform
	std::vector<LevelPlan> plans;
	bool use_plans = true;
	double upload_ms = 0; // wall time of lsfm_tree_upload (reported in lsfm_stats)
	// per level: the refinement steps the level's systems needed in an earlier run (0: not known).  Not structure -- a guess about
	// values that lets a run enqueue the steps of a level without stopping to ask; checked at the end of every run that uses it
	std::vector<int> step_hint;
	unsigned long long digest = 0; // of the resident inputs' labels and index arrays (trees built from packed maps: reload compares)
	// feature-sharded tree (lsfm_tree_set_comm): this process holds one slice of every map; comm.fn == null: off
	Comm comm;
	// sizes of the slice packs of the final map (lsfm_tree_export_slice_*): structure, learnt at the first export
	int slice_n = 0;
	std::vector<int> slice_nf, slice_nw;
};

namespace {

double now_ms()
{
	using namespace std::chrono;
	return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

template <class F> int guarded(lsfm_context* ctx, F&& f)
{
	if (!ctx) return LSFM_ERR_ARG;
	struct Reset { // per-call state that must not outlive the call (sinks of deferred timings point into the caller's frame)
		lsfm_context* c;
		~Reset() { c->timed.clear(); c->ev_next = 0; c->plan = nullptr; }
	} reset{ ctx };
	try
	{
		if (hipSetDevice(ctx->device) != hipSuccess) return LSFM_ERR_NO_DEVICE;
		return f();
	}
	catch (const Error& e)
	{
		ctx->last_error = e.msg;
		fprintf(stderr, "liblsfm_hip: %s\n", e.msg.c_str());
		(void)hipGetLastError();
		return e.code;
	}
	catch (const std::exception& e)
	{
		ctx->last_error = e.what();
		return LSFM_ERR_INTERNAL;
	}
}

size_t estimate_arena(const lsfm_map* maps, int N, int levels)
{
	size_t nw = 0, nf = 0, nu = 0, m = 0;
	for (int k = 0; k < N; k++) { nw += maps[k].nW; nf += maps[k].n; nu += maps[k].nU; m += maps[k].m; }
	const size_t L = levels + 1;
	size_t e = (nw + 2 * L * nf) * 160 * 3 + (nu + 3 * L * m) * 320 * 3 + nf * 400 + ((size_t)256 << 20);
	return e;
}

int tree_levels(int N)
{
	int L = 0;
	while (N > 1) { N = (N + 1) / 2; L++; }
	return L;
}

// one level: transform the maps that need it, then join the pairs
void run_level(lsfm_context* ctx, lsfm_tree* t, lsfm_stats* st, int level)
{
	ctx->plan = (t->use_plans && level < (int)t->plans.size()) ? &t->plans[level] : nullptr;
	const bool analysing = !(ctx->plan && ctx->plan->valid); // this run does the level's symbolic work (it may have done it one level ahead)
	if (!ctx->plan && !t->mono && ctx->pre_plan.valid && ctx->pre_plan_level == level)
	{
		// ... all of it: the plan of this level was made while the level below was being solved (prefetch_next_level)
		ctx->plan = &ctx->pre_plan;
		LSFM_CHECK_HIP(hipStreamWaitEvent(ctx->stream, ctx->evP, 0));
	}
	char rname[48];
	snprintf(rname, sizeof rname, "lsfm level %d (%d maps)", level, t->level.B);
	Range rlevel(rname);
	ctx->mark("level");
	if ((int)t->step_hint.size() <= level) t->step_hint.resize(level + 1, 0);
	ctx->step_hint = t->step_hint[level];
	ctx->steps_used = 0;
	DevBatch& X = t->level;
	const int B = X.B, npairs = B / 2;
	std::vector<int> tref(B, -1), tscap(B, 0), tfix(B, 0);
	int ntr = 0;
	for (int i = 0; i < npairs; i++)
	{
		const int e = 2 * i, c = 2 * i + 1;
		// odd outputs of the previous level go back to their first frame (Imp.cpp:1997-2025 / 6576-6602) ...
		const bool re = X.Ref[c] > X.FRef[c];
		int cref = X.Ref[c], cscap = X.ScaP[c], cfix = X.Fix[c];
		if (re) { cref = X.FRef[c]; cscap = X.FScaP[c]; cfix = X.FFix[c]; tref[c] = cref; tscap[c] = cscap; tfix[c] = cfix; ntr++; }
		// ... and End is expressed in Cur's frame (Imp.cpp:1964 / 6549)
		tref[e] = cref; tscap[e] = cscap; tfix[e] = cfix; ntr++;
	}
	// stage times from events on the stream (a warm level is only enqueued: host clocks say nothing about it)
	hipEvent_t e_t0 = ctx->pool_event(), e_t1 = ctx->pool_event(), e_t2 = ctx->pool_event();
	LSFM_REC_T(e_t0, ctx->stream);
	{
		// LSFM_LEVEL_GAPS=1: what the device waited for the host between the levels of a run (the event behind a level's solve was
		// handed over long before the host got here; this one is stamped when the stream reaches it, or when it arrives)
		static const bool gaps = getenv("LSFM_LEVEL_GAPS") != nullptr;
		if (gaps && ctx->ev_solve_end && ctx->in_tree_run) ctx->defer_time(ctx->ev_solve_end, e_t0, &ctx->dbg_gap_ms);
		ctx->ev_solve_end = nullptr;
	}
	// three arenas in rotation: X (this level; slot -1 = the resident inputs, never written) stays alive until the join
	// is done, because the W blocks of the maps the transform passes through are read from X, not copied (W_alias)
	const int so = t->slot < 0 ? 0 : (t->slot + 1) % 3, sm = t->slot < 0 ? 1 : (t->slot + 2) % 3;
	Arena& other = ctx->arena[so];
	Arena& mine = ctx->arena[sm];
	other.reset();
	mine.reset();
	DevBatch Xt, Y;
	// tests (tests/test_gpu_sharded.py): ONE rank of a feature-sharded run fails in the middle of a level, between two sums
	const bool inject = ctx->inject_level == level;
	if (t->mono)
	{
		{ Range r("lsfm transform"); transform_batch(ctx, other, X, tref, tscap, tfix, true, Xt, true); }
		if (inject) LSFM_FAIL(LSFM_ERR_INTERNAL, "injected failure of this rank (LSFM_TEST_FAIL_RANK)");
		LSFM_REC_T(e_t1, ctx->stream);
		Range r("lsfm join + solve");
		join_batch_mono(ctx, mine, Xt, Y, nullptr, nullptr);
	}
	else
	{
		// Stereo: the joint map is laid out in the middle of the transform (labels, V' and run lengths are known before the
		// W stage), and the transform's block kernel writes every W' block straight to its place in the joint map
		JoinState js;
		const size_t smark = ctx->scratch.mark();
		std::function<TrRedirect(DevBatch&)> hook = [&](DevBatch& mid) {
			join_stereo_prepare(ctx, mine, mid, Y, js);
			TrRedirect rd;
			rd.wbase = js.wbase; rd.newf = js.newf; rd.W = Y.W; rd.photo = Y.photo; rd.feature = Y.feature; rd.srcf = js.srcf;
			return rd;
		};
		{ Range r("lsfm transform"); transform_batch(ctx, other, X, tref, tscap, tfix, false, Xt, false, &hook); } // (the join's layout kernels run inside)
		if (inject) LSFM_FAIL(LSFM_ERR_INTERNAL, "injected failure of this rank (LSFM_TEST_FAIL_RANK)");
		LSFM_REC_T(e_t1, ctx->stream);
		Range r("lsfm join + solve");
		js.smark = smark; // everything of this level goes at once
		join_stereo_finish(ctx, Xt, Y, js, nullptr, nullptr);
		if (ctx->pre_pending) ctx->drop_prepared(); // (a plan the level's solve did not take up)
		ctx->pre_plan = LevelPlan(); // (consumed, if it was this level's)
		ctx->pre_plan_level = -1;
		if (analysing && Y.B > 1 && !ctx->comm)
		{
			// while the device solves this level: the next level's pattern and symbolic factorisation (lsfm_pcg.hip)
			const int nb = Y.B;
			std::vector<int> nref(nb, -1);
			for (int i = 0; i < nb / 2; i++)
			{
				const int e2 = 2 * i, c2 = 2 * i + 1;
				const bool re = Y.Ref[c2] > Y.FRef[c2];
				const int cref = re ? Y.FRef[c2] : Y.Ref[c2];
				if (re) nref[c2] = cref;
				nref[e2] = cref;
			}
			for (int b = 0; b < nb; b++) if (nref[b] >= 0 && Y.Ref[b] == nref[b]) nref[b] = -1; // (same frame: passed through, Imp.cpp:352)
			prefetch_next_level(ctx, Y, nref, level + 1, level + 1 < (int)t->step_hint.size() ? t->step_hint[level + 1] : 0);
		}
		else ctx->drop_prepared();
	}
	LSFM_REC_T(e_t2, ctx->stream);
	if (ctx->steps_used > 0) t->step_hint[level] = ctx->steps_used;
	ctx->step_hint = 0;
	ctx->plan = nullptr;
	t->level = Y;
	t->slot = sm;
	if (st)
	{
		ctx->defer_time(e_t0, e_t1, &st->t_transform_ms);
		ctx->defer_time(e_t1, e_t2, &st->t_join_ms);
		st->levels++; st->joins += npairs; st->transforms += ntr;
	}
}

size_t input_bytes(const lsfm_map* maps, int N)
{
	size_t nw = 0, nf = 0, nu = 0, m = 0;
	for (int k = 0; k < N; k++) { nw += maps[k].nW; nf += maps[k].n; nu += maps[k].nU; m += maps[k].m; }
	return m * 64 + nf * 120 + nu * 300 + nw * 156 + (size_t)N * 16 + ((size_t)1 << 20);
}

} // namespace

extern "C" {

int lsfm_tree_upload(lsfm_context* ctx, const lsfm_map* maps, int N, int mono, lsfm_tree** out)
{
	if (!out || !maps || N <= 0) return LSFM_ERR_ARG;
	*out = nullptr;
	return guarded(ctx, [&]() {
		ctx->ensure_arenas(estimate_arena(maps, N, tree_levels(N)), true);
		lsfm_tree* t = new lsfm_tree();
		t->mono = mono != 0; t->N = N; t->slot = 0;
		ctx->arena[0].reset(); ctx->arena[1].reset(); ctx->scratch.reset();
		try
		{
			t->input_arena.init(input_bytes(maps, N));
			const double t0 = now_ms();
			batch_upload(ctx, t->input_arena, maps, N, t->mono, t->input);
			LSFM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
			t->upload_ms = now_ms() - t0;
		}
		catch (...) { t->input_arena.destroy(); delete t; throw; }
		*out = t;
		return LSFM_OK;
	});
}

// one pass over the tree; with valid plans nothing in here waits for the device before the final synchronisation
static void tree_pass(lsfm_context* ctx, lsfm_tree* t, lsfm_stats* st)
{
	Range rrun("lsfm tree run");
	// level 0 reads the resident inputs where they are (no level writes its input), so a tree can be run repeatedly
	t->slot = -1;
	t->done = false;
	ctx->generation++;
	ctx->arena[0].reset(); ctx->arena[1].reset(); ctx->arena[2].reset(); ctx->scratch.reset();
	ctx->stage_off = 0; // the stream is idle: the staging ring starts over
	ctx->drop_prepared(); ctx->early.reset(); ctx->solved_keys = nullptr; ctx->solved_nnzb = 0; // nothing prepared by an earlier run
	ctx->ev_solve_end = nullptr;
	LSFM_CHECK_HIP(hipMemsetAsync(ctx->d_run, 0, sizeof(RunStatsDev), ctx->stream));
	static const bool poison = getenv("LSFM_POISON") != nullptr; // debug: every byte a run has not written itself reads as NaN / -1
	if (poison)
	{
		for (int i = 0; i < 3; i++) LSFM_CHECK_HIP(hipMemsetAsync(ctx->arena[i].base, 0xFF, ctx->arena[i].cap, ctx->stream));
		LSFM_CHECK_HIP(hipMemsetAsync(ctx->scratch.base, 0xFF, ctx->scratch.cap, ctx->stream));
		for (int i = 0; i < 2; i++) if (ctx->sarena[i].base) LSFM_CHECK_HIP(hipMemsetAsync(ctx->sarena[i].base, 0xFF, ctx->sarena[i].cap, ctx->stream));
	}
	t->level = t->input;
	const int nlev = tree_levels(t->N);
	if ((int)t->plans.size() != nlev + 1) t->plans.assign(nlev + 1, LevelPlan());
	int level = 0;
	while (t->level.B > 1 && (t->stop_level <= 0 || level < t->stop_level)) run_level(ctx, t, st, level++);
	// final map back to its first frame (Imp.cpp:2039-2063 / 6613-6630)
	DevBatch& X = t->level;
	if (t->final_reanchor && X.B == 1 && X.Ref[0] > X.FRef[0])
	{
		std::vector<int> tref(1, X.FRef[0]), tscap(1, X.FScaP[0]), tfix(1, X.FFix[0]);
		const int so = t->slot < 0 ? 0 : (t->slot + 1) % 3;
		Arena& other = ctx->arena[so];
		other.reset();
		DevBatch Xt;
		hipEvent_t e0 = ctx->pool_event(), e1 = ctx->pool_event();
		LSFM_REC_T(e0, ctx->stream);
		ctx->plan = t->use_plans ? &t->plans[nlev] : nullptr;
		transform_batch(ctx, other, X, tref, tscap, tfix, t->mono, Xt);
		if (ctx->plan) ctx->plan->valid = true;
		ctx->plan = nullptr;
		LSFM_REC_T(e1, ctx->stream);
		ctx->defer_time(e0, e1, &st->t_transform_ms);
		st->transforms++;
		t->level = Xt;
		t->slot = so;
	}
	LSFM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
}

int lsfm_tree_run(lsfm_context* ctx, lsfm_tree* t, lsfm_stats* stats)
{
	if (!t) return LSFM_ERR_ARG;
	return guarded(ctx, [&]() {
		lsfm_stats local;
		memset(&local, 0, sizeof local);
		lsfm_stats* st = stats ? stats : &local;
		ctx->stats = st;
		struct InRun { lsfm_context* c; InRun(lsfm_context* x, Comm* cm) : c(x) { c->in_tree_run = true; c->comm = cm; } ~InRun() { c->in_tree_run = false; c->comm = nullptr; } }
			in_run(ctx, t->comm.fn ? &t->comm : nullptr);
		LSFM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
		try
		{
			const double t_begin = now_ms();
			for (int attempt = 0;; attempt++)
			{
				memset(st, 0, sizeof *st);
				st->attempts = attempt + 1;
				ctx->timed.clear(); ctx->ev_next = 0;
				static const bool no_hints = getenv("LSFM_NO_STEP_HINTS") != nullptr; // debug: every run asks after every refinement step
				if (no_hints) t->step_hint.clear();
				ctx->timeline_on = getenv("LSFM_TIMELINE") != nullptr;
				ctx->timeline.clear();
				ctx->mark("run");
				// feature-sharded run: an error of this rank alone (LSFM_FAIL inside the pass) must still reach the exchange of the flags
				// below, or its peers would wait there for a sum this rank never joins; it is rethrown after the exchange
				std::unique_ptr<Error> pass_error;
				ctx->inject_level = -1;
				bool inject_undone = false;
				if (ctx->comm)
				{
					// tests: LSFM_TEST_FAIL_RANK=r makes rank r's FIRST attempt fail -- LSFM_TEST_FAIL_KIND=throw (default): an error in the
					// middle of level LSFM_TEST_FAIL_LEVEL (default 0); undone: a system reported above its bound at the end of the pass
					static const char* frank = getenv("LSFM_TEST_FAIL_RANK");
					static int injected = 0; // (once per process: the run after the failed one must go through)
					if (frank && attempt == 0 && atoi(frank) == ctx->comm->rank && !injected++)
					{
						const char* kind = getenv("LSFM_TEST_FAIL_KIND");
						if (kind && !strcmp(kind, "undone")) inject_undone = true;
						else ctx->inject_level = getenv("LSFM_TEST_FAIL_LEVEL") ? atoi(getenv("LSFM_TEST_FAIL_LEVEL")) : 0;
					}
					try { tree_pass(ctx, t, st); }
					catch (const Error& e)
					{
						pass_error.reset(new Error(e));
						ctx->inject_level = -1;
						(void)hipStreamSynchronize(ctx->stream); (void)hipGetLastError();
						ctx->drop_prepared();
					}
				}
				else
				{
					// (the arenas start at an eighth of the upper bound the upload asked for: a run that exhausts one doubles them and starts over)
					try { tree_pass(ctx, t, st); }
					catch (const Error& e)
					{
						// a level that was recording its plan found a pivot far below zero itself (lsfm_pcg.hip check_factor): treated like the
						// same finding at the end of a run, below -- the tree is joined again while attempts are left
						if (e.code == LSFM_ERR_NOT_SPD && attempt < 3)
						{
							(void)hipStreamSynchronize(ctx->stream); (void)hipGetLastError();
							ctx->stats = st; ctx->plan = nullptr; // (tree_pass was left mid-level)
							t->step_hint.clear(); t->plans.clear();
							if (getenv("LSFM_DEBUG_CONV")) fprintf(stderr, "[lsfm conv] attempt %d: %s -- joining the tree again\n", attempt, e.msg.c_str());
							continue;
						}
						if (e.code != LSFM_ERR_OOM || !ctx->grow_arenas()) throw;
						if (getenv("LSFM_DEBUG")) fprintf(stderr, "[lsfm] arenas grown to %zu MiB each after: %s\n", ctx->arena_bytes >> 20, e.msg.c_str());
						attempt--; // (not a numerical repeat)
						continue;
					}
				}
				st->t_total_ms = now_ms() - t_begin; // (repeated attempts included; the stage times below are the last attempt's)
				ctx->mark("end");
				if (ctx->timeline_on)
				{
					double prev = ctx->timeline.empty() ? 0 : ctx->timeline[0].second;
					for (const auto& m : ctx->timeline)
					{
						if (!strcmp(m.first, "level")) fprintf(stderr, "\n[tl]");
						fprintf(stderr, " %s+%.0f", m.first, 1e3 * (m.second - prev));
						prev = m.second;
					}
					fprintf(stderr, "\n");
				}
				// what the warm levels left in the device accumulators instead of stopping for it
				RunStatsDev rs;
				LSFM_CHECK_HIP(hipStreamSynchronize(ctx->stream2)); // (the side stream's share of the record: k_sum_run_squares)
				LSFM_CHECK_HIP(hipMemcpy(&rs, ctx->d_run, sizeof rs, hipMemcpyDeviceToHost));
				if (ctx->comm)
				{
					// feature-sharded run: whether the run is repeated (below) must be decided alike on every rank -- a rank that
					// went on alone would wait for sums nobody else takes part in
					Comm& cm = *ctx->comm;
					if (inject_undone) rs.undone++;
					if (pass_error)
					{
						// this rank left the pass alone, somewhere between two sums: it takes part in its peers' sums (with zeros) until they
						// are here too -- see Comm in lsfm_internal.hpp.  No healthy rank left, or the communicator itself failed: nothing to
						// exchange, the error is this rank's own
						bool there = false;
						try { there = cm.follow(ctx->stream); } catch (const Error&) { there = false; }
						if (!there) throw *pass_error;
					}
					cm.restart();
					long long* d_fl = cm.alloc<long long>(8);
					// (st->not_converged: what the levels that recorded a plan reported through the stats; fl[6]: this rank's pass threw)
					long long fl[8] = { rs.tr_err != 0, rs.chol_err != 0, rs.plan_stale != 0, rs.not_converged, rs.undone, st->not_converged, pass_error ? 1 : 0, 0 };
					LSFM_CHECK_HIP(hipMemcpyAsync(d_fl, fl, sizeof fl, hipMemcpyHostToDevice, ctx->stream));
					LSFM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
					if (pass_error) cm.call(ctx->stream, (size_t)(reinterpret_cast<char*>(d_fl) - cm.buf), 8, LSFM_DTYPE_I64); // (its header went with follow())
					else cm.allreduce(ctx->stream, d_fl, 8, LSFM_DTYPE_I64, Comm::KIND_FINAL);
					LSFM_CHECK_HIP(hipMemcpyAsync(fl, d_fl, sizeof fl, hipMemcpyDeviceToHost, ctx->stream));
					LSFM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
					if (fl[0] && !rs.tr_err) rs.tr_err = 1;
					if (fl[1] && !rs.chol_err) rs.chol_err = 1;
					rs.plan_stale = fl[2] != 0;
					rs.not_converged = (int)((fl[3] + cm.world - 1) / cm.world); // (every rank solves every system: the count, not its multiple)
					rs.undone = (int)((fl[4] + cm.world - 1) / cm.world);
					st->not_converged = (int)((fl[5] + cm.world - 1) / cm.world); // (the same verdict on every rank: the repeat test below reads it)
					// (a failed pass may have left plans and step counts of levels it ran on zeros: the next run starts without them, alike
					// on every rank)
					if (pass_error || fl[6]) { t->plans.clear(); t->step_hint.clear(); }
					if (pass_error) throw *pass_error;
					if (fl[6]) LSFM_FAIL(LSFM_ERR_INTERNAL, "another rank of the feature-sharded run failed");
				}
				if (rs.floored && getenv("LSFM_DEBUG_CONV")) fprintf(stderr, "[lsfm conv] %d pivot(s) of the separators held at their lower bound in this run\n", rs.floored);
				if (rs.tr_err) LSFM_FAIL(LSFM_ERR_ARG, "transform: target pose id not found in map " + std::to_string(rs.tr_err - 1));
				// (const bool more, below: attempts left)
				if (rs.chol_err && !(attempt < 3))
					LSFM_FAIL(LSFM_ERR_NOT_SPD, "Schur system is not positive definite (block column " + std::to_string(rs.chol_err - 1) + " of the factor)");
				// Repeating a run.  A plan that met values it does not fit (LevelPlan::tr_sign), refinement steps enqueued by a count
				// from an earlier run that did not suffice this time (`undone`), or a system left above its bound although every level
				// asked after every step -- seen in one synth-16k Mono tree out of fifteen: the last 6x6 block of the root's top
				// separator, what is left of 1e6..1e8-sized entries after 16 000 columns of updates whose atomic sums land in another
				// order every run, came out slightly indefinite and its factor, taken by magnitude (k_sn_panel), was too poor a
				// preconditioner.  In each case the tree is joined again without what the earlier runs left (plans, step counts): the
				// rounding falls differently.  At most three times; what is still not converged then is reported (LSFM_NOT_CONVERGED).
				const bool more = attempt < 3;
				if (rs.plan_stale)
				{
					if (!more) LSFM_FAIL(LSFM_ERR_INTERNAL, "level plans kept being reported stale");
					t->plans.clear();
					continue;
				}
				// A pivot far below zero (k_sn_panel: more than 1 % of the diagonal entry S had) is reported as "not positive definite" --
				// after the other attempts: it was seen once in ~400 runs of the synth-16k Mono tree, at the last block of the root
				// (16 382 columns of updates above it), where a run before or after it factors a system that differs in the last bits
				// of S (K9's sums are floating-point atomics) without complaint.  A system that IS indefinite fails three times.
				if ((rs.chol_err || rs.not_converged || rs.undone || st->not_converged) && more) // (a level that records its plan reports through the stats, not the device record)
				{
					if (rs.chol_err && getenv("LSFM_DEBUG_CONV")) fprintf(stderr, "[lsfm conv] attempt %d: pivot of block column %d far below zero, joining the tree again\n", attempt, rs.chol_err - 1);
					t->step_hint.clear();
					t->plans.clear();
					continue;
				}
				st->not_converged += rs.not_converged;
				st->max_rel_residual = std::max(st->max_rel_residual, rs.max_rel_residual);
				st->upload_ms = t->upload_ms;
				st->schur_flops += 108.0 * (double)rs.k2;
				st->s_digest = rs.s_digest; st->factor_digest = rs.factor_digest; st->refactor_mismatch = rs.refactor_mismatch; st->s_rebuild_mismatch = rs.s_rebuild_mismatch;
				break;
			}
			ctx->flush_times();
			if (getenv("LSFM_LEVEL_GAPS")) { fprintf(stderr, "[lsfm] device idle between the levels of this run: %.3f ms\n", ctx->dbg_gap_ms); ctx->dbg_gap_ms = 0.0; }
		}
		catch (...) { ctx->stats = nullptr; throw; }
		ctx->stats = nullptr;
		t->done = true;
		t->generation = ctx->generation;
		return st->not_converged ? LSFM_NOT_CONVERGED : LSFM_OK;
	});
}

int lsfm_tree_set_plans(lsfm_tree* t, int on)
{
	if (!t) return LSFM_ERR_ARG;
	t->use_plans = on != 0;
	if (!on) t->plans.clear();
	return LSFM_OK;
}

int lsfm_tree_download(lsfm_context* ctx, lsfm_tree* t, lsfm_map* out)
{
	if (!t || !out) return LSFM_ERR_ARG;
	return guarded(ctx, [&]() {
		if (!t->done || t->level.B != 1) LSFM_FAIL(LSFM_ERR_ARG, "tree has not been run");
		if (t->generation != ctx->generation)
			LSFM_FAIL(LSFM_ERR_ARG, "the result of this tree was overwritten by a later call on the same context (it lives in the context's arenas): "
			                        "download a tree before the context is used for anything else, or run it again");
		batch_download_map(ctx, t->level, 0, t->mono, out);
		return LSFM_OK;
	});
}

int lsfm_tree_download_state(lsfm_context* ctx, lsfm_tree* t, int* m, int* n, int* stno, double* stVal, size_t cap)
{
	if (!t || !m || !n) return LSFM_ERR_ARG;
	return guarded(ctx, [&]() {
		if (!t->done || t->level.B != 1) LSFM_FAIL(LSFM_ERR_ARG, "tree has not been run");
		if (t->generation != ctx->generation) LSFM_FAIL(LSFM_ERR_ARG, "the result of this tree was overwritten by a later call on the same context");
		const DevBatch& b = t->level;
		*m = b.M; *n = b.NF;
		if (!stno && !stVal) return LSFM_OK;
		const size_t r = (size_t)6 * b.M + (size_t)3 * b.NF;
		if (cap < r) LSFM_FAIL(LSFM_ERR_ARG, "state arrays too small");
		if (stVal)
		{
			d2h(ctx, stVal, b.pose, (size_t)b.M * 6 * sizeof(double));
			d2h(ctx, stVal + (size_t)6 * b.M, b.feat, (size_t)b.NF * 3 * sizeof(double));
		}
		if (stno)
		{
			std::vector<int> pid(b.M), fid(b.NF);
			d2h(ctx, pid.data(), b.pose_id, (size_t)b.M * sizeof(int));
			d2h(ctx, fid.data(), b.feat_id, (size_t)b.NF * sizeof(int));
			for (int i = 0; i < b.M; i++) for (int c = 0; c < 6; c++) stno[6 * (size_t)i + c] = -pid[i];
			for (int i = 0; i < b.NF; i++) for (int c = 0; c < 3; c++) stno[6 * (size_t)b.M + 3 * (size_t)i + c] = fid[i];
		}
		return LSFM_OK;
	});
}

size_t lsfm_tree_export_size(lsfm_context* ctx, lsfm_tree* t)
{
	if (!ctx || !t || !t->done || t->level.B != 1 || t->generation != ctx->generation) return 0;
	PackHeader h;
	memset(&h, 0, sizeof h);
	const DevBatch& b = t->level;
	h.m = b.M; h.n = b.NF; h.nU = b.NU; h.nW = b.NW;
	return pack_layout(h);
}

int lsfm_tree_export_dev(lsfm_context* ctx, lsfm_tree* t, void* dst, size_t cap)
{
	if (!t || !dst) return LSFM_ERR_ARG;
	return guarded(ctx, [&]() {
		if (!t->done || t->level.B != 1) LSFM_FAIL(LSFM_ERR_ARG, "tree has not been run");
		if (t->generation != ctx->generation) LSFM_FAIL(LSFM_ERR_ARG, "the result of this tree was overwritten by a later call on the same context");
		batch_pack_map(ctx, t->level, 0, t->mono, dst, cap);
		LSFM_CHECK_HIP(hipStreamSynchronize(ctx->stream)); // the caller hands dst to another library / stream next
		return LSFM_OK;
	});
}

static_assert(sizeof(PackHeader) == LSFM_PACK_HEADER_BYTES, "include/lsfm.h documents the header size");
size_t lsfm_packed_size(const void* host_header)
{
	if (!host_header) return 0;
	PackHeader h;
	memcpy(&h, host_header, sizeof h);
	if (h.magic != LSFM_PACK_MAGIC || h.version != 1 || h.m < 0 || h.n < 0 || h.nU < 0 || h.nW < 0) return 0;
	// the offsets an unpack follows are the ones the sizes imply, never just what the buffer says
	PackHeader c = h;
	if (pack_layout(c) != h.total) return 0;
	for (int i = 0; i < 12; i++) if (c.off[i] != h.off[i]) return 0;
	return (size_t)h.total;
}

int lsfm_tree_upload_dev(lsfm_context* ctx, const void* const* packed, int N, int mono, lsfm_tree** out)
{
	if (!out || !packed || N <= 0) return LSFM_ERR_ARG;
	*out = nullptr;
	return guarded(ctx, [&]() {
		std::vector<PackHeader> hdr(N);
		size_t nw = 0, nf = 0, nu = 0, m = 0, bytes = 0;
		for (int k = 0; k < N; k++)
		{
			if (!packed[k]) LSFM_FAIL(LSFM_ERR_ARG, "null packed map");
			LSFM_CHECK_HIP(hipMemcpyAsync(&hdr[k], packed[k], sizeof(PackHeader), hipMemcpyDeviceToHost, ctx->stream));
		}
		LSFM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
		for (int k = 0; k < N; k++)
		{
			if (lsfm_packed_size(&hdr[k]) == 0) LSFM_FAIL(LSFM_ERR_ARG, "buffer " + std::to_string(k) + " is not a packed map");
			if ((hdr[k].mono != 0) != (mono != 0)) LSFM_FAIL(LSFM_ERR_ARG, "packed map of the other camera type");
			nw += hdr[k].nW; nf += hdr[k].n; nu += hdr[k].nU; m += hdr[k].m; bytes += hdr[k].total;
		}
		const size_t L = tree_levels(N) + 1;
		ctx->ensure_arenas((nw + 2 * L * nf) * 160 * 3 + (nu + 3 * L * m) * 320 * 3 + nf * 400 + ((size_t)256 << 20));
		lsfm_tree* t = new lsfm_tree();
		t->mono = mono != 0; t->N = N; t->slot = 0;
		ctx->arena[0].reset(); ctx->arena[1].reset(); ctx->scratch.reset();
		try
		{
			t->input_arena.init(bytes + (m + nf + nw) * 8 + (size_t)N * 4096 + ((size_t)1 << 20));
			batch_unpack_maps(ctx, t->input_arena, packed, hdr.data(), N, t->mono, t->input);
			t->digest = batch_structure_digest(ctx, t->input);
			LSFM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
		}
		catch (...) { t->input_arena.destroy(); delete t; throw; }
		*out = t;
		return LSFM_OK;
	});
}

int lsfm_tree_reload_dev(lsfm_context* ctx, lsfm_tree* t, const void* const* packed, int N)
{
	if (!t || !packed || N != t->N) return LSFM_ERR_ARG;
	return guarded(ctx, [&]() {
		std::vector<PackHeader> hdr(N);
		for (int k = 0; k < N; k++)
		{
			if (!packed[k]) LSFM_FAIL(LSFM_ERR_ARG, "null packed map");
			LSFM_CHECK_HIP(hipMemcpyAsync(&hdr[k], packed[k], sizeof(PackHeader), hipMemcpyDeviceToHost, ctx->stream));
		}
		LSFM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
		const DevBatch& b = t->input;
		for (int k = 0; k < N; k++)
		{
			const PackHeader& h = hdr[k];
			if (lsfm_packed_size(&h) == 0 || (h.mono != 0) != t->mono || h.m != b.pose_off[k + 1] - b.pose_off[k] || h.n != b.feat_off[k + 1] - b.feat_off[k] ||
			    h.nU != b.u_off[k + 1] - b.u_off[k] || h.nW != b.w_off[k + 1] - b.w_off[k])
				LSFM_FAIL(LSFM_ERR_ARG, "packed map " + std::to_string(k) + " does not have the sizes of the tree's resident map (reload keeps the structure)");
		}
		ctx->generation++;
		t->done = false;
		t->input_arena.reset(); // same sizes, same order: every array lands where it was
		batch_unpack_maps(ctx, t->input_arena, packed, hdr.data(), N, t->mono, t->input);
		ctx->scratch.reset();
		const unsigned long long dg = batch_structure_digest(ctx, t->input);
		if (dg != t->digest)
		{
			// same sizes, other labels / index arrays: everything the plans hold (S pattern, K9 slots, join offsets) is void
			t->plans.clear();
			t->digest = dg;
		}
		LSFM_CHECK_HIP(hipStreamSynchronize(ctx->stream));
		return LSFM_OK;
	});
}

int lsfm_tree_set_comm(lsfm_tree* t, int rank, int world, lsfm_allreduce_fn fn, void* user, void* dev_buf, size_t dev_bytes)
{
	if (!t) return LSFM_ERR_ARG;
	if (!fn) { t->comm = Comm(); return LSFM_OK; }
	if (world < 1 || rank < 0 || rank >= world || !dev_buf || dev_bytes < 4096) return LSFM_ERR_ARG;
	if (t->comm.rank != rank || t->comm.world != world) t->plans.clear();
	t->comm.rank = rank; t->comm.world = world; t->comm.fn = fn; t->comm.user = user;
	t->comm.buf = static_cast<char*>(dev_buf); t->comm.cap = dev_bytes; t->comm.restart(); t->comm.broken = false;
	return LSFM_OK;
}

int lsfm_tree_set_comm_blocks(lsfm_tree* t, int block_maps)
{
	if (!t || block_maps < 0 || (block_maps & (block_maps - 1))) return LSFM_ERR_ARG; // (blocks of the reference's pairing are 2^k local maps)
	if (t->comm.block_maps != block_maps) t->plans.clear(); // (the plans hold the ownership of every factorisation)
	t->comm.block_maps = block_maps;
	return LSFM_OK;
}

int lsfm_tree_export_slice_sizes(lsfm_context* ctx, lsfm_tree* t, int nslices, size_t* sizes)
{
	if (!t || !sizes || nslices <= 0) return LSFM_ERR_ARG;
	return guarded(ctx, [&]() {
		if (!t->done || t->level.B != 1) LSFM_FAIL(LSFM_ERR_ARG, "tree has not been run");
		if (t->generation != ctx->generation) LSFM_FAIL(LSFM_ERR_ARG, "the result of this tree was overwritten by a later call on the same context");
		if (t->slice_n != nslices)
		{
			// how many features and W blocks every slice holds: structure -- counted once, kept for the later runs of the tree
			batch_slice_counts(ctx, t->level, nslices, t->slice_nf, t->slice_nw);
			t->slice_n = nslices;
		}
		const DevBatch& b = t->level;
		for (int g = 0; g < nslices; g++)
		{
			PackHeader h;
			memset(&h, 0, sizeof h);
			h.m = b.M; h.n = t->slice_nf[g]; h.nU = b.NU; h.nW = t->slice_nw[g];
			sizes[g] = pack_layout(h);
		}
		return LSFM_OK;
	});
}

int lsfm_tree_export_slice_dev(lsfm_context* ctx, lsfm_tree* t, int nslices, int slice, void* dst, size_t cap)
{
	if (!t || !dst || nslices <= 0 || slice < 0 || slice >= nslices) return LSFM_ERR_ARG;
	return guarded(ctx, [&]() {
		if (!t->done || t->level.B != 1) LSFM_FAIL(LSFM_ERR_ARG, "tree has not been run");
		if (t->generation != ctx->generation) LSFM_FAIL(LSFM_ERR_ARG, "the result of this tree was overwritten by a later call on the same context");
		if (t->slice_n != nslices) LSFM_FAIL(LSFM_ERR_ARG, "call lsfm_tree_export_slice_sizes with the same number of slices first");
		const size_t mk = ctx->scratch.mark();
		batch_pack_slice(ctx, t->level, t->mono, nslices, slice, t->slice_nf[slice], t->slice_nw[slice], dst, cap);
		LSFM_CHECK_HIP(hipStreamSynchronize(ctx->stream)); // the caller hands dst to another library / stream next
		ctx->scratch.release(mk);
		return LSFM_OK;
	});
}

int lsfm_tree_set_stop_level(lsfm_tree* t, int levels)
{
	if (!t || levels < 0) return LSFM_ERR_ARG;
	t->stop_level = levels;
	return LSFM_OK;
}

int lsfm_tree_node_count(lsfm_context* ctx, lsfm_tree* t)
{
	if (!ctx || !t || !t->done || t->generation != ctx->generation) return 0;
	return t->level.B;
}

int lsfm_tree_download_node(lsfm_context* ctx, lsfm_tree* t, int k, lsfm_map* out)
{
	if (!t || !out) return LSFM_ERR_ARG;
	return guarded(ctx, [&]() {
		if (!t->done) LSFM_FAIL(LSFM_ERR_ARG, "tree has not been run");
		if (t->generation != ctx->generation) LSFM_FAIL(LSFM_ERR_ARG, "the result of this tree was overwritten by a later call on the same context");
		if (k < 0 || k >= t->level.B) LSFM_FAIL(LSFM_ERR_ARG, "node index out of range");
		batch_download_map(ctx, t->level, k, t->mono, out);
		return LSFM_OK;
	});
}

int lsfm_tree_set_final_reanchor(lsfm_tree* t, int on)
{
	if (!t) return LSFM_ERR_ARG;
	t->final_reanchor = on != 0;
	return LSFM_OK;
}

void lsfm_tree_free(lsfm_context* ctx, lsfm_tree* t)
{
	if (!t) return;
	if (ctx) { (void)hipSetDevice(ctx->device); (void)hipStreamSynchronize(ctx->stream); }
	t->input_arena.destroy();
	delete t;
}

int lsfm_divide_conquer(lsfm_context* ctx, const lsfm_map* maps, int N, int mono, lsfm_map* out, lsfm_stats* stats)
{
	lsfm_tree* t = nullptr;
	int rc = lsfm_tree_upload(ctx, maps, N, mono, &t);
	if (rc) return rc;
	t->use_plans = false; // one run, then the tree is gone: nothing to record for a next one
	int rrc = lsfm_tree_run(ctx, t, stats);
	if (rrc < 0) { lsfm_tree_free(ctx, t); return rrc; }
	rc = lsfm_tree_download(ctx, t, out);
	lsfm_tree_free(ctx, t);
	return rc ? rc : rrc;
}

static int transform_one(lsfm_context* ctx, const lsfm_map* in, int Ref, int ScaP, int Fix, bool mono, lsfm_map* out)
{
	if (!in || !out) return LSFM_ERR_ARG;
	return guarded(ctx, [&]() {
		ctx->ensure_arenas(estimate_arena(in, 1, 1));
		ctx->arena[0].reset(); ctx->arena[1].reset(); ctx->scratch.reset();
		DevBatch X, Y;
		batch_upload(ctx, ctx->arena[0], in, 1, mono, X);
		std::vector<int> tref(1, Ref), tscap(1, ScaP), tfix(1, Fix);
		transform_batch(ctx, ctx->arena[1], X, tref, tscap, tfix, mono, Y);
		batch_download_map(ctx, Y, 0, mono, out);
		return LSFM_OK;
	});
}

int lsfm_transform_stereo(lsfm_context* ctx, const lsfm_map* in, int Ref, lsfm_map* out) { return transform_one(ctx, in, Ref, 0, 0, false, out); }
int lsfm_transform_mono(lsfm_context* ctx, const lsfm_map* in, int Ref, int ScaP, int Fix, lsfm_map* out)
{
	return transform_one(ctx, in, Ref, ScaP, Fix, true, out);
}

int lsfm_join_stereo(lsfm_context* ctx, const lsfm_map* End, const lsfm_map* Cur, lsfm_map* joint, double* eP_out, double* eF_out)
{
	if (!End || !Cur || !joint) return LSFM_ERR_ARG;
	return guarded(ctx, [&]() {
		lsfm_map two[2] = { *End, *Cur };
		ctx->ensure_arenas(estimate_arena(two, 2, 1));
		ctx->arena[0].reset(); ctx->arena[1].reset(); ctx->scratch.reset();
		DevBatch X, Y;
		batch_upload(ctx, ctx->arena[0], two, 2, false, X);
		lsfm_stats st;
		memset(&st, 0, sizeof st);
		ctx->stats = &st;
		try { join_batch_stereo(ctx, ctx->arena[1], X, Y, eP_out, eF_out); }
		catch (...) { ctx->stats = nullptr; throw; }
		ctx->stats = nullptr;
		batch_download_map(ctx, Y, 0, false, joint);
		return st.not_converged ? LSFM_NOT_CONVERGED : LSFM_OK;
	});
}

int lsfm_join_mono(lsfm_context* ctx, const lsfm_map* End, const lsfm_map* Cur, lsfm_map* joint, double* eP_out, double* eF_out)
{
	if (!End || !Cur || !joint) return LSFM_ERR_ARG;
	return guarded(ctx, [&]() {
		lsfm_map two[2] = { *End, *Cur };
		ctx->ensure_arenas(estimate_arena(two, 2, 1));
		ctx->arena[0].reset(); ctx->arena[1].reset(); ctx->scratch.reset();
		DevBatch X, Y;
		batch_upload(ctx, ctx->arena[0], two, 2, true, X);
		lsfm_stats st;
		memset(&st, 0, sizeof st);
		ctx->stats = &st;
		try { join_batch_mono(ctx, ctx->arena[1], X, Y, eP_out, eF_out); }
		catch (...) { ctx->stats = nullptr; throw; }
		ctx->stats = nullptr;
		batch_download_map(ctx, Y, 0, true, joint);
		return st.not_converged ? LSFM_NOT_CONVERGED : LSFM_OK;
	});
}

// raw-pointer solver with the reference's argument list (Imp.h:209 / 223); fixed_blk / fixed_scalar < 0: Stereo
static int solve_raw(lsfm_context* ctx, double* stVal, const double* eb, const double* ea, const double* U, const double* W,
                     const double* V, const int* Ui, const int* Uj, const int* photo, const int* feature, int m, int n, int nU,
                     int nW, const double* x0, int fixed_blk, int fixed_scalar)
{
	if (!stVal || m <= 0 || n < 0 || nU < 0 || nW < 0) return LSFM_ERR_ARG;
	return guarded(ctx, [&]() {
		size_t need = ((size_t)nW * 200 + (size_t)nU * 400 + (size_t)n * 300 + (size_t)m * 4000) * 3 + ((size_t)128 << 20);
		ctx->ensure_arenas(need);
		ctx->arena[0].reset(); ctx->scratch.reset();
		Arena& ar = ctx->arena[0];
		std::vector<int> fptr(n + 1);
		{
			int j = 0;
			for (int f = 0; f < n; f++)
			{
				fptr[f] = j;
				while (j < nW && feature[j] == f) j++;
				if (j == fptr[f]) LSFM_FAIL(LSFM_ERR_ARG, "every feature needs at least one W block, W sorted by feature");
			}
			if (j != nW) LSFM_FAIL(LSFM_ERR_ARG, "W is not sorted by feature");
			fptr[n] = nW;
		}
		double* dU = ar.alloc<double>((size_t)nU * 36); int* dUi = ar.alloc<int>(nU); int* dUj = ar.alloc<int>(nU);
		double* dW = ar.alloc<double>((size_t)nW * 18); int* dph = ar.alloc<int>(nW); int* dfp = ar.alloc<int>(n + 1);
		double* dV = ar.alloc<double>((size_t)n * 9); double* dea = ar.alloc<double>((size_t)m * 6); double* deb = ar.alloc<double>((size_t)n * 3);
		double* dx0 = x0 ? ar.alloc<double>((size_t)m * 6) : nullptr;
		double* dxp = ar.alloc<double>((size_t)m * 6); double* dxf = ar.alloc<double>((size_t)n * 3);
		int* dseg = ar.alloc<int>(m + n + 1);
		h2d(ctx, dU, U, (size_t)nU * 36 * sizeof(double)); h2d(ctx, dUi, Ui, nU * sizeof(int)); h2d(ctx, dUj, Uj, nU * sizeof(int));
		h2d(ctx, dW, W, (size_t)nW * 18 * sizeof(double)); h2d(ctx, dph, photo, nW * sizeof(int)); h2d(ctx, dfp, fptr.data(), (n + 1) * sizeof(int));
		h2d(ctx, dV, V, (size_t)n * 9 * sizeof(double)); h2d(ctx, dea, ea, (size_t)m * 6 * sizeof(double)); h2d(ctx, deb, eb, (size_t)n * 3 * sizeof(double));
		if (x0) h2d(ctx, dx0, x0, (size_t)m * 6 * sizeof(double));
		dev_zero(ctx, dseg, (m + n + 1) * sizeof(int));
		SolveIO io;
		io.M = m; io.NF = n; io.NU = nU; io.NW = nW; io.nseg = 1;
		io.d_pose_seg = dseg; io.d_feat_seg = dseg + m; io.d_seg_active = nullptr;
		io.U = dU; io.Ui = dUi; io.Uj = dUj; io.W = dW; io.photo = dph; io.fptr = dfp; io.V = dV;
		io.ea = dea; io.eb = deb; io.x0 = dx0; io.x_pose = dxp; io.x_feat = dxf;
		io.seg_rows.assign(1, m);
		if (ctx->small_max > 0 && small_solve_strips(m, ctx->small_max))
		{
			const int offs[6] = { 0, m, 0, n, 0, nU };
			int* d_offs = ar.alloc<int>(6);
			h2d(ctx, d_offs, offs, sizeof offs);
			io.d_pose_off = d_offs; io.d_feat_off = d_offs + 2; io.d_u_off = d_offs + 4;
		}
		if (fixed_blk >= 0 || fixed_scalar >= 0)
		{
			std::vector<unsigned char> fx((size_t)m * 6, 0);
			if (fixed_blk >= 0 && fixed_blk < m) for (int i = 0; i < 6; i++) fx[(size_t)fixed_blk * 6 + i] = 1;
			if (fixed_scalar >= 0 && fixed_scalar < 6 * m) fx[fixed_scalar] = 1;
			unsigned char* dfx = ar.alloc<unsigned char>((size_t)m * 6);
			h2d(ctx, dfx, fx.data(), fx.size());
			io.d_fixed = dfx;
		}
		int rc = solve_batch(ctx, io);
		d2h(ctx, stVal, dxp, (size_t)m * 6 * sizeof(double));
		d2h(ctx, stVal + 6 * m, dxf, (size_t)n * 3 * sizeof(double));
		return rc ? LSFM_NOT_CONVERGED : LSFM_OK;
	});
}

int lsfm_solve_stereo(lsfm_context* ctx, double* stVal, const double* eb, const double* ea, const double* U, const double* W,
                      const double* V, const int* Ui, const int* Uj, const int* photo, const int* feature, int m, int n, int nU,
                      int nW, const double* x0)
{
	return solve_raw(ctx, stVal, eb, ea, U, W, V, Ui, Uj, photo, feature, m, n, nU, nW, x0, -1, -1);
}

// Imp.cpp:6756-7041: the 6 scalars of block `Ref` (= scalars ScaP..ScaP+5, the call site passes ScaP = 6*Ref) and scalar
// `Fix` are removed from the system, the solution is 0 there, and finally stVal[Fix] = Sign (Imp.cpp:7026)
int lsfm_solve_mono(lsfm_context* ctx, double* stVal, const double* eb, const double* ea, const double* U, const double* W,
                    const double* V, const int* Ui, const int* Uj, const int* photo, const int* feature, int m, int n, int nU,
                    int nW, int Ref, int ScaP, int Fix, int Sign, int FixBlk, const double* x0)
{
	(void)FixBlk;
	if (ScaP != 6 * Ref || Ref < 0 || Ref >= m || Fix < 0 || Fix >= 6 * m) return LSFM_ERR_ARG;
	int rc = solve_raw(ctx, stVal, eb, ea, U, W, V, Ui, Uj, photo, feature, m, n, nU, nW, x0, Ref, Fix);
	if (rc >= 0) stVal[Fix] = Sign;
	return rc;
}

int lsfm_gn_polish(lsfm_context* ctx, const lsfm_map* maps, int N, int type, lsfm_map* x, int iters, double* obj, double* gnorm, int* halvings)
{
	if (!maps || N <= 0 || !x || iters < 0 || !obj || !gnorm || (type != 0 && type != 1)) return LSFM_ERR_ARG;
	return guarded(ctx, [&]() { return gn_polish(ctx, maps, N, type == 1, x, iters, obj, gnorm, halvings); });
}

int lsfm_inverse_v(lsfm_context* ctx, double* V, int m, int n)
{
	(void)m;
	if (n < 0 || (n && !V)) return LSFM_ERR_ARG;
	return guarded(ctx, [&]() {
		ctx->ensure_arenas((size_t)n * 400 + ((size_t)64 << 20));
		ctx->arena[0].reset(); ctx->scratch.reset();
		double* dV = ctx->arena[0].alloc<double>((size_t)n * 9);
		double* dIV = ctx->arena[0].alloc<double>((size_t)n * 9);
		h2d(ctx, dV, V, (size_t)n * 9 * sizeof(double));
		vinv_only(ctx, n, dV, dIV);
		d2h(ctx, V, dIV, (size_t)n * 9 * sizeof(double));
		return LSFM_OK;
	});
}

int lsfm_solve_features(lsfm_context* ctx, const double* W, const double* IV, const double* ea, const double* eb, const double* dpa, double* dpb,
                        int m, int n, const int* mapCor, const int* photo)
{
	(void)ea;
	if (m <= 0 || n < 0 || !dpa || (n && (!W || !IV || !eb || !dpb || !mapCor || !photo))) return LSFM_ERR_ARG;
	return guarded(ctx, [&]() {
		std::vector<int> fptr((size_t)n + 1, 0);
		for (int f = 0; f < n; f++)
		{
			if (mapCor[f] < 0) LSFM_FAIL(LSFM_ERR_ARG, "negative block count of a feature");
			fptr[f + 1] = fptr[f] + mapCor[f];
		}
		const int nW = fptr[n];
		for (int j = 0; j < nW; j++) if (photo[j] < 0 || photo[j] >= m) LSFM_FAIL(LSFM_ERR_ARG, "photo index out of range");
		ctx->ensure_arenas((size_t)nW * 200 + (size_t)n * 300 + (size_t)m * 100 + ((size_t)64 << 20));
		ctx->arena[0].reset(); ctx->scratch.reset();
		Arena& ar = ctx->arena[0];
		double* dW = ar.alloc<double>((size_t)nW * 18); int* dph = ar.alloc<int>(nW); int* dfp = ar.alloc<int>((size_t)n + 1);
		double* dIV = ar.alloc<double>((size_t)n * 9); double* deb = ar.alloc<double>((size_t)n * 3);
		double* dxp = ar.alloc<double>((size_t)m * 6); double* dxf = ar.alloc<double>((size_t)n * 3);
		h2d(ctx, dW, W, (size_t)nW * 18 * sizeof(double)); h2d(ctx, dph, photo, (size_t)nW * sizeof(int));
		h2d(ctx, dfp, fptr.data(), ((size_t)n + 1) * sizeof(int)); h2d(ctx, dIV, IV, (size_t)n * 9 * sizeof(double));
		h2d(ctx, deb, eb, (size_t)n * 3 * sizeof(double)); h2d(ctx, dxp, dpa, (size_t)m * 6 * sizeof(double));
		backsub_only(ctx, n, dfp, dph, dW, dIV, deb, dxp, dxf);
		d2h(ctx, dpb, dxf, (size_t)n * 3 * sizeof(double));
		return LSFM_OK;
	});
}

int lsfm_schur_pattern(lsfm_context* ctx, const int* Ui, const int* Uj, const int* photo, const int* feature, int m, int n, int nU, int nW, int* rowptr,
                       int* colidx, int cap, int* nnzb)
{
	if (m <= 0 || n < 0 || nU < 0 || nW < 0 || !rowptr || !colidx || !nnzb || (nU && (!Ui || !Uj)) || (nW && (!photo || !feature))) return LSFM_ERR_ARG;
	return guarded(ctx, [&]() {
		ctx->ensure_arenas(((size_t)nW * 64 + (size_t)nU * 64 + (size_t)m * 4096) * 2 + ((size_t)64 << 20));
		ctx->arena[0].reset(); ctx->scratch.reset();
		Arena& ar = ctx->arena[0];
		std::vector<int> fptr(n + 1);
		{
			int j = 0;
			for (int f = 0; f < n; f++)
			{
				fptr[f] = j;
				while (j < nW && feature[j] == f) j++;
			}
			if (j != nW) LSFM_FAIL(LSFM_ERR_ARG, "W is not sorted by feature");
			fptr[n] = nW;
		}
		for (int i = 0; i < nU; i++) if (Ui[i] < 0 || Uj[i] >= m || Ui[i] > Uj[i]) LSFM_FAIL(LSFM_ERR_ARG, "U block coordinates must satisfy 0 <= Ui <= Uj < m");
		for (int j = 0; j < nW; j++) if (photo[j] < 0 || photo[j] >= m) LSFM_FAIL(LSFM_ERR_ARG, "photo index out of range");
		int* dUi = ar.alloc<int>(nU); int* dUj = ar.alloc<int>(nU); int* dph = ar.alloc<int>(nW); int* dfp = ar.alloc<int>(n + 1);
		h2d(ctx, dUi, Ui, nU * sizeof(int)); h2d(ctx, dUj, Uj, nU * sizeof(int)); h2d(ctx, dph, photo, nW * sizeof(int));
		h2d(ctx, dfp, fptr.data(), (n + 1) * sizeof(int));
		SolveIO io;
		io.M = m; io.NF = n; io.NU = nU; io.NW = nW; io.nseg = 1;
		io.Ui = dUi; io.Uj = dUj; io.photo = dph; io.fptr = dfp;
		int cnt = 0;
		const int* d_rowptr = nullptr;
		const int* d_colidx = nullptr;
		schur_pattern_only(ctx, io, &cnt, &d_rowptr, &d_colidx);
		*nnzb = cnt;
		if (cnt > cap) LSFM_FAIL(LSFM_ERR_ARG, "colidx too small for the pattern (" + std::to_string(cnt) + " blocks)");
		d2h(ctx, rowptr, d_rowptr, (size_t)(m + 1) * sizeof(int));
		d2h(ctx, colidx, d_colidx, (size_t)cnt * sizeof(int));
		return LSFM_OK;
	});
}

int lsfm_spmv_bench(lsfm_context* ctx, int m, const int* rowptr, const int* colidx, const double* val, const double* x, double* y,
                    int reps, double* avg_ms, double* algorithmic_bytes)
{
	if (!rowptr || !colidx || !val || !x || !y || m <= 0) return LSFM_ERR_ARG;
	return guarded(ctx, [&]() {
		ctx->ensure_arenas(((size_t)rowptr[m] * 480 + (size_t)m * 1000) * 2 + ((size_t)64 << 20));
		ctx->scratch.reset();
		return spmv_external(ctx, m, rowptr, colidx, val, x, y, reps, avg_ms, algorithmic_bytes);
	});
}

int lsfm_wstream_bench(lsfm_context* ctx, long long nblocks, int mode, int reps, double* avg_ms)
{
	if (nblocks <= 0 || mode < 0 || mode > 2 || reps <= 0 || !avg_ms) return LSFM_ERR_ARG;
	return guarded(ctx, [&]() {
		ctx->ensure_arenas((size_t)nblocks * 144 * 2 + ((size_t)64 << 20));
		ctx->scratch.reset();
		return wstream_bench(ctx, nblocks, mode, reps, avg_ms);
	});
}

int lsfm_selftest_prims(lsfm_context* ctx, int cases, unsigned seed)
{
	if (cases <= 0) return LSFM_ERR_ARG;
	return guarded(ctx, [&]() {
		ctx->ensure_arenas((size_t)64 << 20);
		ctx->scratch.reset();
		const size_t cap = (size_t)1 << 20;
		unsigned char* buf = ctx->scratch.alloc<unsigned char>(cap);
		std::vector<unsigned char> ref(cap), got(cap);
		unsigned long long st = seed * 2654435761ull + 12345ull;
		auto rnd = [&]() { st = st * 6364136223846793005ull + 1442695040888963407ull; return (unsigned)(st >> 33); };
		LSFM_CHECK_HIP(hipMemset(buf, 0xA5, cap));
		std::fill(ref.begin(), ref.end(), (unsigned char)0xA5);
		auto compare = [&](const char* what, int c) {
			d2h(ctx, got.data(), buf, cap);
			for (size_t i = 0; i < cap; i++)
				if (got[i] != ref[i])
					LSFM_FAIL(LSFM_ERR_INTERNAL, std::string("lsfm_selftest_prims: ") + what + " case " + std::to_string(c) + ": byte " + std::to_string(i) + " is " +
					                                 std::to_string(got[i]) + ", expected " + std::to_string(ref[i]));
		};
		for (int c = 0; c < cases; c++)
		{
			// fills: short and long, any alignment of either end
			const size_t len = (c % 3 == 0) ? rnd() % 70 : rnd() % (cap / 2);
			const size_t off = rnd() % (cap - len);
			const int byte = (c % 4 == 0) ? 0 : (c % 4 == 1 ? 0xff : (int)(rnd() & 0xff));
			fill_async(ctx->stream, buf + off, byte, len);
			std::fill(ref.begin() + off, ref.begin() + off + len, (unsigned char)byte);
			if (c % 8 == 7 || c + 1 == cases) compare("fill", c);
		}
		std::vector<unsigned> src(cap / 4);
		for (unsigned& v : src) v = rnd();
		for (int c = 0; c < cases; c++)
		{
			// copies of whole words (what the path copies: index tables, records), one by one ...
			const size_t nw = 1 + rnd() % (c % 2 ? 300 : 60000), offw = rnd() % (cap / 4 - nw), from = rnd() % (cap / 4 - nw);
			h2d(ctx, buf + 4 * offw, src.data() + from, 4 * nw);
			memcpy(ref.data() + 4 * offw, src.data() + from, 4 * nw);
			if (c % 8 == 7 || c + 1 == cases) compare("copy", c);
		}
		for (int c = 0; c < cases; c += 4)
		{
			// ... and as a batch of host pieces and device-to-device pieces (CopyBatch: one table, one launch)
			CopyBatch cb(ctx);
			size_t at = 0;
			const size_t half = cap / 2;
			for (int i = 0; i < 5 && at + 70000 < half; i++)
			{
				const size_t nw = 4 * (1 + rnd() % 4000), from = rnd() % (cap / 4 - nw);
				if (i % 2 == 0) { cb.h2d(buf + at, src.data() + from, 4 * nw); memcpy(ref.data() + at, src.data() + from, 4 * nw); }
				else { cb.d2d(buf + at, buf + half + 4 * (from % (half / 4 - nw)), 4 * nw); memcpy(ref.data() + at, ref.data() + half + 4 * (from % (half / 4 - nw)), 4 * nw); }
				at += 4 * nw + 4 * (rnd() % 5);
			}
			cb.flush();
			compare("batch", c);
		}
		return LSFM_OK;
	});
}

} // extern "C"
