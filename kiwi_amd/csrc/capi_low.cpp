// Low-level C ABI (include/kiwi_amd.h) over kamd::Engine.  Error convention mirrors the reference's C API:
// nothing throws across the boundary; failures return NULL / negative and leave a thread-local message
// (/root/reference/src/capi/kiwi_c.cpp:84, 95-114).
#include <cstddef>
#include <cstring>
#include <memory>
#include <mutex>
#include <string>
#include "../../include/kiwi_amd.h"
#include "engine.hpp"
#include "sbg_eval.hpp"
#include "typo.hpp"

using namespace kamd;

struct kamd_engine { std::unique_ptr<Engine> e; };
struct kamd_batch { std::shared_ptr<StagedBatch> b; };
struct kamd_results { BatchResults r; };   // flat segments; the accessors below index into them
static_assert(sizeof(kamd_token_t) == sizeof(FlatToken) && offsetof(kamd_token_t, form_off) == offsetof(FlatToken, formOff) && offsetof(kamd_token_t, morph_id) == offsetof(FlatToken, morph),
	"kamd_token_t is the layout of kamd::FlatToken");

namespace kamd { void exactMathProbe(const float* x, float* e, float* l, uint32_t n); }

namespace
{
	thread_local std::string lastError;
	template<class Fn, class R> R guarded(Fn&& fn, R fail)
	{
		try { return fn(); }
		catch (const std::exception& e) { lastError = e.what(); }
		catch (...) { lastError = "unknown error"; }
		return fail;
	}

	std::vector<std::pair<const char16_t*, size_t>> views(const uint16_t* texts, const uint64_t* offsets, uint32_t n)
	{
		std::vector<std::pair<const char16_t*, size_t>> v(n);
		for (uint32_t i = 0; i < n; ++i) v[i] = { (const char16_t*)texts + offsets[i], (size_t)(offsets[i + 1] - offsets[i]) };
		return v;
	}

	kamd_results* pack(BatchResults&& res) { return new kamd_results{ std::move(res) }; }

	// analysis `i` of text `t`: segment + global analysis index, or null
	const ResultSegment* ana(kamd_results_h r, uint32_t t, uint32_t i, uint32_t& a)
	{
		if (!r || t >= r->r.nTexts) return nullptr;
		size_t local;
		const ResultSegment& seg = r->r.locate(t, local);
		if (i >= seg.textAna[local + 1] - seg.textAna[local]) return nullptr;
		a = seg.textAna[local] + i;
		return &seg;
	}
}

extern "C"
{
	kamd_engine_h kamd_open(const char* path, int device)
	{
		return guarded([&]() { auto h = std::make_unique<kamd_engine>(); h->e.reset(new Engine(path, device)); return h.release(); }, (kamd_engine*)nullptr);
	}
	void kamd_close(kamd_engine_h h) { delete h; }
	const char* kamd_last_error(void) { return lastError.c_str(); }

	int kamd_set_config(kamd_engine_h h, float cutOff, float spacePenalty, float typoCostWeight, uint32_t maxUnk, uint32_t maxUnkJ, uint32_t spaceTol, int integrateAllomorph)
	{
		if (!h) return -2;
		auto& c = h->e->config;
		c.cutOffThreshold = cutOff; c.spacePenalty = spacePenalty; c.typoCostWeight = typoCostWeight;
		c.maxUnkFormSize = maxUnk; c.maxUnkFormSizeFollowedByJClass = maxUnkJ; c.spaceTolerance = spaceTol; c.integrateAllomorph = !!integrateAllomorph;
		return 0;
	}

	kamd_results_h kamd_analyze_batch(kamd_engine_h h, const uint16_t* texts, const uint64_t* offsets, uint32_t n, uint32_t topN, uint64_t match, int openEnding, int hostThreads)
	{
		if (!h) { lastError = "invalid handle"; return nullptr; }
		return guarded([&]() { return pack(h->e->analyzeBatch(views(texts, offsets, n), topN, match, !!openEnding, hostThreads)); }, (kamd_results*)nullptr);
	}

	kamd_batch_h kamd_stage(kamd_engine_h h, const uint16_t* texts, const uint64_t* offsets, uint32_t n, uint64_t match, int openEnding, int hostThreads)
	{
		if (!h) { lastError = "invalid handle"; return nullptr; }
		return guarded([&]() { auto b = std::make_unique<kamd_batch>(); b->b = h->e->stage(views(texts, offsets, n), match, !!openEnding, hostThreads); return b.release(); }, (kamd_batch*)nullptr);
	}
	int kamd_run(kamd_engine_h h, kamd_batch_h b, float* ms)
	{
		if (!h || !b) return -2;
		return guarded([&]() { auto t = h->e->run(*b->b); if (ms) { ms[0] = t.scanMs; ms[1] = t.latticeMs; ms[2] = t.searchMs; ms[3] = t.finishMs; } return 0; }, -1);
	}
	kamd_results_h kamd_fetch(kamd_engine_h h, kamd_batch_h b, uint32_t topN)
	{
		if (!h || !b) { lastError = "invalid handle"; return nullptr; }
		return guarded([&]() { return pack(h->e->fetch(*b->b, topN)); }, (kamd_results*)nullptr);
	}
	int kamd_batch_info(kamd_batch_h b, uint64_t* info)
	{
		if (!b) return -2;
		info[0] = Engine::stagedChunks(*b->b); info[1] = Engine::stagedUnits(*b->b); info[2] = Engine::stagedDeviceBytes(*b->b);
		return 0;
	}
	void kamd_batch_close(kamd_batch_h b) { delete b; }

	uint32_t kamd_res_texts(kamd_results_h r) { return r ? (uint32_t)r->r.nTexts : 0; }
	uint32_t kamd_res_size(kamd_results_h r, uint32_t t)
	{
		if (!r || t >= r->r.nTexts) return 0;
		size_t local; const ResultSegment& seg = r->r.locate(t, local);
		return seg.textAna[local + 1] - seg.textAna[local];
	}
	float kamd_res_prob(kamd_results_h r, uint32_t t, uint32_t i) { uint32_t a; const ResultSegment* s = ana(r, t, i, a); return s ? s->anaScore[a] : 0.f; }
	uint32_t kamd_res_token_num(kamd_results_h r, uint32_t t, uint32_t i) { uint32_t a; const ResultSegment* s = ana(r, t, i, a); return s ? s->anaTok[a + 1] - s->anaTok[a] : 0; }
	const kamd_token_t* kamd_res_tokens(kamd_results_h r, uint32_t t, uint32_t i)
	{
		uint32_t a; const ResultSegment* s = ana(r, t, i, a);
		return (s && s->anaTok[a + 1] > s->anaTok[a]) ? reinterpret_cast<const kamd_token_t*>(s->toks.data() + s->anaTok[a]) : nullptr;
	}
	const uint16_t* kamd_res_forms(kamd_results_h r, uint32_t t)
	{
		if (!r || t >= r->r.nTexts) return nullptr;
		size_t local; const ResultSegment& seg = r->r.locate(t, local);
		return reinterpret_cast<const uint16_t*>(seg.forms.data());
	}
	uint64_t kamd_res_d2h_bytes(kamd_results_h r) { return r ? r->r.d2hBytes : 0; }
	void kamd_res_close(kamd_results_h r) { delete r; }

	int kamd_debug_exact_math(const float* x, float* exp_out, float* log_out, uint32_t n)
	{
		return guarded([&]() { kamd::exactMathProbe(x, exp_out, log_out, n); return 0; }, -1);
	}

	// Host-side run of the SkipBigram step the search kernel uses (sbg_eval.hpp, shared source): one LmState::next on top of a
	// Knlm log-likelihood the caller supplies.  No device involved; tests compare it with the reference's SbgState::next.
	int kamd_debug_sbg_next(const char* raw_model_path, uint32_t* hist8, uint32_t* pos, uint32_t wid, float knlm_ll, float* ll_out)
	{
		return guarded([&]()
		{
			static std::mutex mu; static std::string cachedPath; static std::unique_ptr<FlatModel> cached;
			std::lock_guard<std::mutex> lk{ mu };
			if (!cached || cachedPath != raw_model_path) { cached.reset(new FlatModel); bakeModel(*cached, raw_model_path); cachedPath = raw_model_path; }
			const SbgView sv = cached->sbgView();
			if (!sv.present()) throw std::runtime_error{ "kamd_debug_sbg_next: the model has no SkipBigram tables" };
			uint32_t ring[8]; for (int i = 0; i < 8; ++i) ring[i] = hist8[i];
			uint32_t p = *pos & 7u;
			*ll_out = sbgNext(sv, ring, p, wid, knlm_ll);
			for (int i = 0; i < 8; ++i) hist8[i] = ring[i];
			*pos = p;
			return 0;
		}, -1);
	}

	// ---- typo transformers (typo.hpp): host-side building block; analyze does not take them yet ---------------------------------------------
	struct kamd_typo { TypoTransformer tt; std::unique_ptr<PreparedTypo> prepared; };
	kamd_typo* kamd_typo_new(float continual_cost, float lengthening_cost)
	{
		return guarded([&]() { auto* t = new kamd_typo; t->tt.setContinualCost(continual_cost); t->tt.setLengtheningCost(lengthening_cost); return t; }, (kamd_typo*)nullptr);
	}
	void kamd_typo_close(kamd_typo* t) { delete t; }
	// a copy of one of Kiwi's built-in sets (DefaultTypoSet id 0..6): the caller owns it
	kamd_typo* kamd_typo_default(int set)
	{
		return guarded([&]() { auto* t = new kamd_typo; t->tt = defaultTypoSet(set); return t; }, (kamd_typo*)nullptr);
	}
	int kamd_typo_add(kamd_typo* t, const uint16_t* orig, uint32_t n_orig, const uint16_t* error, uint32_t n_error, float cost, int left_cond, int dialect)
	{
		if (!t) return -2;
		return guarded([&]() { t->tt.add(std::u16string{ (const char16_t*)orig, n_orig }, std::u16string{ (const char16_t*)error, n_error }, cost, (uint8_t)left_cond, (uint16_t)dialect); return 0; }, -1);
	}
	int kamd_typo_add_entry(kamd_typo* t, const uint16_t* orig, uint32_t n_orig, const uint16_t* error, uint32_t n_error, float cost, int left_cond, int dialect)
	{
		if (!t) return -2;
		return guarded([&]() { t->tt.addEntry(std::u16string{ (const char16_t*)orig, n_orig }, std::u16string{ (const char16_t*)error, n_error }, cost, (uint8_t)left_cond, (uint16_t)dialect); return 0; }, -1);
	}
	int kamd_typo_set_costs(kamd_typo* t, float continual_cost, float lengthening_cost)
	{
		if (!t) return -2;
		t->tt.setContinualCost(continual_cost); t->tt.setLengtheningCost(lengthening_cost);
		return 0;
	}
	int kamd_typo_scale(kamd_typo* t, float scale)
	{
		if (!t) return -2;
		return guarded([&]() { t->tt.scaleCost(scale); return 0; }, -1);
	}
	int kamd_typo_prepare(kamd_typo* t, int inverse)
	{
		if (!t) return -2;
		return guarded([&]() { t->prepared.reset(new PreparedTypo{ t->tt, inverse != 0 }); return 0; }, -1);
	}
	size_t kamd_typo_graph(kamd_typo* t, const uint16_t* text, uint32_t len, int allowed_dialect, int normalize_coda, uint8_t* out, size_t cap)
	{
		if (!t || !t->prepared) { lastError = "typo transformer not prepared"; return 0; }
		return guarded([&]()
		{
			U16 norm; std::vector<uint32_t> pos;
			normalizeWithPosition((const char16_t*)text, len, norm, pos);
			if (normalize_coda) normalizeCoda(norm);
			std::vector<TypoGraphNode> g;
			const size_t maxIdx = t->prepared->graph((const char16_t*)norm.data(), norm.size(), (uint16_t)allowed_dialect, g);
			size_t need = 0; uint8_t* p = out;
			auto put = [&](const void* v, size_t n) { need += n; if (p && need <= cap) { std::memcpy(p, v, n); p += n; } };
			auto put32 = [&](uint32_t v) { put(&v, 4); };
			put32((uint32_t)norm.size()); put(norm.data(), 2 * norm.size());
			put32((uint32_t)g.size());
			for (auto& n : g)
			{
				const std::u16string f = t->prepared->formOf(n, (const char16_t*)norm.data());
				put32((uint32_t)f.size()); put(f.data(), 2 * f.size());
				put32(n.endPos); put(&n.typoCost, 4); put32(n.prevOffset); put32(n.siblingOffset); put(&n.continualTypoIdx, 1); put(&n.dialect, 2);
			}
			put32((uint32_t)maxIdx);
			return need;
		}, (size_t)0);
	}

	kamd_results_h kamd_analyze_batch_typo(kamd_engine_h h, kamd_typo* t, float threshold, int allowed_dialect, const uint16_t* texts, const uint64_t* offsets, uint32_t n, uint32_t topN, uint64_t match, int openEnding, int hostThreads)
	{
		if (!h || !t || !t->prepared) { lastError = "invalid handle / typo transformer not prepared"; return nullptr; }
		return guarded([&]() { return pack(h->e->analyzeBatch(views(texts, offsets, n), topN, match, !!openEnding, hostThreads, TypoOption{ t->prepared.get(), threshold, (uint16_t)allowed_dialect })); }, (kamd_results*)nullptr);
	}
	kamd_batch_h kamd_stage_typo(kamd_engine_h h, kamd_typo* t, float threshold, int allowed_dialect, const uint16_t* texts, const uint64_t* offsets, uint32_t n, uint64_t match, int openEnding, int hostThreads)
	{
		if (!h || !t || !t->prepared) { lastError = "invalid handle / typo transformer not prepared"; return nullptr; }
		return guarded([&]() { auto b = std::make_unique<kamd_batch>(); b->b = h->e->stage(views(texts, offsets, n), match, !!openEnding, hostThreads, TypoOption{ t->prepared.get(), threshold, (uint16_t)allowed_dialect }); return b.release(); }, (kamd_batch*)nullptr);
	}
	size_t kamd_typo_lattices(kamd_engine_h h, kamd_typo* t, float threshold, int allowed_dialect, const uint16_t* text, uint32_t len, uint64_t match, uint8_t* out, size_t cap)
	{
		if (!h || !t || !t->prepared) { lastError = "invalid handle / typo transformer not prepared"; return 0; }
		return guarded([&]() { auto d = h->e->dumpTypoLattices(*t->prepared, threshold, (uint16_t)allowed_dialect, (const char16_t*)text, len, match); if (d.size() <= cap) std::memcpy(out, d.data(), d.size()); return d.size(); }, (size_t)0);
	}

	size_t kamd_dump_dict(kamd_engine_h h, uint8_t* out, size_t cap)
	{
		if (!h) return 0;
		return guarded([&]() { auto d = dumpDict(h->e->model()); if (d.size() <= cap) std::memcpy(out, d.data(), d.size()); return d.size(); }, (size_t)0);
	}
	size_t kamd_dump_lattices(kamd_engine_h h, const uint16_t* text, uint32_t len, uint64_t match, uint8_t* out, size_t cap)
	{
		if (!h) return 0;
		return guarded([&]() { auto d = h->e->dumpLattices((const char16_t*)text, len, match); if (d.size() <= cap) std::memcpy(out, d.data(), d.size()); return d.size(); }, (size_t)0);
	}
}
