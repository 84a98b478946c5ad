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

namespace kamd { void exactMathProbe(const float* x, float* e, float* l, float* t, uint32_t n); }
namespace kamd { void conggProbe(const FlatModel& m, const uint32_t* ctx, const uint32_t* hist, const uint32_t* next, const uint8_t* flags, float* out, uint32_t n); }

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
	kamd_engine_h kamd_open_dialects(const char* path, int device, int enabled_dialects)
	{
		return guarded([&]() { auto h = std::make_unique<kamd_engine>(); h->e.reset(new Engine(path, device, Engine::LmMode::Auto, (uint32_t)enabled_dialects)); return h.release(); }, (kamd_engine*)nullptr);
	}
	kamd_engine_h kamd_open_mode(const char* path, int device, int lm_mode, int enabled_dialects)
	{
		return guarded([&]()
		{
			if (lm_mode < 0 || lm_mode > 4) throw std::invalid_argument{ "kamd_open_mode: lm_mode must be 0 .. 4" };
			static const Engine::LmMode modes[5] = { Engine::LmMode::Auto, Engine::LmMode::Knlm, Engine::LmMode::Sbg, Engine::LmMode::Cong, Engine::LmMode::CongGlobal };
			auto h = std::make_unique<kamd_engine>(); h->e.reset(new Engine(path, device, modes[lm_mode], (uint32_t)enabled_dialects)); return h.release();
		}, (kamd_engine*)nullptr);
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

	int kamd_set_oov_chr_bias(kamd_engine_h h, float bias)
	{
		if (!h) return -2;
		h->e->config.oovChrBias = bias;
		return 0;
	}

	int kamd_set_oov_freq_params(kamd_engine_h h, float global_weight, float local_weight, float global_min_freq)
	{
		if (!h) return -2;
		h->e->config.oovGlobalWeight = global_weight; h->e->config.oovLocalWeight = local_weight; h->e->config.oovGlobalMinFreq = global_min_freq;
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
	int kamd_batch_reruns(kamd_batch_h b, float* ms)
	{
		if (!b) return -2;
		return (int)Engine::rerunChunks(*b->b, ms);
	}
	int kamd_batch_pool(kamd_batch_h b, uint64_t* out3)
	{
		if (!b) return -2;
		Engine::stagedPool(*b->b, out3);
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

	// ---- packed, position-independent form of a batch's results: what a rank ships to the gathering rank -----
	// {u32 magic 'KRES', u32 nTexts, u32 nAna, u32 pad, u64 nTok, u64 nFormUnits} u32 textAna[nTexts+1] u32 anaTok[nAna+1] f32 anaScore[nAna]
	// kamd_token_t tok[nTok] (form_off into the one pool that follows) u16 forms[nFormUnits]; every section starts at a multiple of 8 bytes
	size_t kamd_res_pack(kamd_results_h r, uint8_t* out, size_t cap)
	{
		if (!r) return 0;
		return guarded([&]()
		{
			const BatchResults& R = r->r;
			uint64_t nAna = 0, nTok = 0, nForms = 0;
			for (size_t t = 0; t < R.nTexts; ++t)
			{
				size_t l; const ResultSegment& s = R.locate(t, l);
				const uint32_t a0 = s.textAna[l], a1 = s.textAna[l + 1];
				nAna += a1 - a0;
				if (a1 > a0)
				{
					const uint32_t t0 = s.anaTok[a0], t1 = s.anaTok[a1];
					nTok += t1 - t0;
					if (t1 > t0) nForms += s.toks[t1 - 1].formOff + s.toks[t1 - 1].formLen + 1 - s.toks[t0].formOff;
				}
			}
			auto pad8 = [](size_t n) { return (n + 7) & ~(size_t)7; };
			const size_t oTextAna = 32, oAnaTok = pad8(oTextAna + 4 * (R.nTexts + 1)), oScore = pad8(oAnaTok + 4 * (nAna + 1)), oTok = pad8(oScore + 4 * nAna),
				oForms = oTok + sizeof(FlatToken) * nTok, total = pad8(oForms + 2 * nForms);
			if (!out || cap < total) return total;
			std::memset(out, 0, total);
			uint32_t* hd = reinterpret_cast<uint32_t*>(out);
			hd[0] = 0x5345524Bu; hd[1] = (uint32_t)R.nTexts; hd[2] = (uint32_t)nAna; hd[3] = 0;
			reinterpret_cast<uint64_t*>(out)[2] = nTok; reinterpret_cast<uint64_t*>(out)[3] = nForms;
			uint32_t* textAna = reinterpret_cast<uint32_t*>(out + oTextAna); uint32_t* anaTok = reinterpret_cast<uint32_t*>(out + oAnaTok);
			float* score = reinterpret_cast<float*>(out + oScore); FlatToken* tok = reinterpret_cast<FlatToken*>(out + oTok); char16_t* forms = reinterpret_cast<char16_t*>(out + oForms);
			uint64_t a = 0, k = 0, f = 0;
			textAna[0] = 0; anaTok[0] = 0;
			for (size_t t = 0; t < R.nTexts; ++t)
			{
				size_t l; const ResultSegment& s = R.locate(t, l);
				const uint32_t a0 = s.textAna[l], a1 = s.textAna[l + 1];
				if (a1 > a0)
				{
					const uint32_t t0 = s.anaTok[a0], t1 = s.anaTok[a1];
					const uint64_t f0 = t1 > t0 ? s.toks[t0].formOff : 0, f1 = t1 > t0 ? s.toks[t1 - 1].formOff + s.toks[t1 - 1].formLen + 1 : 0;
					for (uint32_t i = a0; i < a1; ++i) { score[a] = s.anaScore[i]; anaTok[a + 1] = (uint32_t)(k + (s.anaTok[i + 1] - t0)); ++a; }
					for (uint32_t i = t0; i < t1; ++i) { tok[k] = s.toks[i]; tok[k].formOff = s.toks[i].formOff - f0 + f; ++k; }
					if (f1 > f0) std::memcpy(forms + f, s.forms.data() + f0, 2 * (f1 - f0));
					f += f1 - f0;
				}
				textAna[t + 1] = (uint32_t)a;
			}
			return total;
		}, (size_t)0);
	}

	// Results of one corpus that was sharded over `n_parts` ranks by index (text g -> part g % n_parts, local index g / n_parts: kiwi_amd.dist.shard_indices),
	// every part packed by kamd_res_pack: merged back into input order.  The handle answers kamd_res_* like any other.
	kamd_results_h kamd_res_merge_strided(const uint8_t* const* parts, const size_t* sizes, uint32_t n_parts)
	{
		return guarded([&]()
		{
			struct View { uint32_t nTexts, nAna; const uint32_t* textAna; const uint32_t* anaTok; const float* score; const FlatToken* tok; const char16_t* forms; };
			auto pad8 = [](size_t n) { return (n + 7) & ~(size_t)7; };
			std::vector<View> v(n_parts);
			size_t total = 0;
			for (uint32_t p = 0; p < n_parts; ++p)
			{
				const uint8_t* b = parts[p];
				if (sizes[p] < 32 || reinterpret_cast<const uint32_t*>(b)[0] != 0x5345524Bu) throw std::runtime_error{ "kamd_res_merge_strided: not a packed result" };
				View& w = v[p];
				w.nTexts = reinterpret_cast<const uint32_t*>(b)[1]; w.nAna = reinterpret_cast<const uint32_t*>(b)[2];
				const uint64_t nTok = reinterpret_cast<const uint64_t*>(b)[2], nForms = reinterpret_cast<const uint64_t*>(b)[3];
				const size_t oTextAna = 32, oAnaTok = pad8(oTextAna + 4 * ((size_t)w.nTexts + 1)), oScore = pad8(oAnaTok + 4 * ((size_t)w.nAna + 1)), oTok = pad8(oScore + 4 * (size_t)w.nAna),
					oForms = oTok + sizeof(FlatToken) * nTok;
				if (pad8(oForms + 2 * nForms) > sizes[p]) throw std::runtime_error{ "kamd_res_merge_strided: truncated part" };
				w.textAna = reinterpret_cast<const uint32_t*>(b + oTextAna); w.anaTok = reinterpret_cast<const uint32_t*>(b + oAnaTok); w.score = reinterpret_cast<const float*>(b + oScore);
				w.tok = reinterpret_cast<const FlatToken*>(b + oTok); w.forms = reinterpret_cast<const char16_t*>(b + oForms);
				// the index tables come from another rank's buffer: every offset is checked before it is followed
				bool ok = w.textAna[0] == 0 && w.textAna[w.nTexts] == w.nAna && w.anaTok[0] == 0 && w.anaTok[w.nAna] == nTok;
				for (uint32_t i = 0; ok && i < w.nTexts; ++i) ok = w.textAna[i] <= w.textAna[i + 1];
				for (uint32_t i = 0; ok && i < w.nAna; ++i) ok = w.anaTok[i] <= w.anaTok[i + 1];
				for (uint64_t i = 0; ok && i < nTok; ++i) ok = w.tok[i].formOff <= nForms && (uint64_t)w.tok[i].formLen + 1 <= nForms - w.tok[i].formOff;
				if (!ok) throw std::runtime_error{ "kamd_res_merge_strided: corrupt part (index tables out of range)" };
				total += w.nTexts;
			}
			for (uint32_t p = 0; p < n_parts; ++p) if (v[p].nTexts != (total + n_parts - 1 - p) / n_parts) throw std::runtime_error{ "kamd_res_merge_strided: parts are not an index-strided split" };
			auto res = std::make_unique<kamd_results>();
			BatchResults& R = res->r;
			R.nTexts = total;
			R.segs.resize((total + BatchResults::kSegTexts - 1) / BatchResults::kSegTexts);
			for (size_t g = 0; g < total; ++g)
			{
				const View& w = v[g % n_parts]; const size_t l = g / n_parts;
				ResultSegment& seg = R.segs[g / BatchResults::kSegTexts];
				for (uint32_t a = w.textAna[l]; a < w.textAna[l + 1]; ++a)
				{
					for (uint32_t i = w.anaTok[a]; i < w.anaTok[a + 1]; ++i)
					{
						FlatToken t = w.tok[i];
						const char16_t* fs = w.forms + t.formOff;
						t.formOff = seg.forms.size();
						seg.forms.insert(seg.forms.end(), fs, fs + t.formLen + 1);
						seg.toks.push_back(t);
					}
					seg.anaScore.push_back(w.score[a]);
					seg.anaTok.push_back((uint32_t)seg.toks.size());
				}
				seg.textAna.push_back((uint32_t)seg.anaScore.size());
			}
			return res.release();
		}, (kamd_results*)nullptr);
	}
	void kamd_res_close(kamd_results_h r) { delete r; }

	// developer probe, host side: the pattern recogniser of the text preparation at one position (textprep.cpp matchPattern): length | tag << 32
	uint64_t kamd_debug_match_pattern(uint16_t left, const uint16_t* text, uint32_t len, uint64_t match_options)
	{
		const auto r = kamd::matchPattern((char16_t)left, (const char16_t*)text, (const char16_t*)text + len, match_options);
		return (uint64_t)r.first | ((uint64_t)r.second << 32);
	}

	int kamd_debug_exact_math(const float* x, float* exp_out, float* log_out, uint32_t n)
	{
		return guarded([&]() { kamd::exactMathProbe(x, exp_out, log_out, nullptr, n); return 0; }, -1);
	}
	int kamd_debug_exact_tanh(const float* x, float* tanh_out, uint32_t n)
	{
		return guarded([&]() { kamd::exactMathProbe(x, nullptr, nullptr, tanh_out, n); return 0; }, -1);
	}

	int kamd_debug_cong_global(kamd_engine_h h, const uint32_t* ctx, const uint32_t* hist7, const uint32_t* next, const uint8_t* flags, float* out, uint32_t n)
	{
		if (!h) return -1;
		return guarded([&]() { kamd::conggProbe(h->e->model(), ctx, hist7, next, flags, out, n); return 0; }, -1);
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
	size_t kamd_typo_graph_device(kamd_engine_h h, kamd_typo* t, const uint16_t* text, uint32_t len, int allowed_dialect, int normalize_coda, int use_device, uint8_t* out, size_t cap)
	{
		if (!h || !t || !t->prepared) { lastError = "invalid handle / typo transformer not prepared"; return 0; }
		return guarded([&]() { auto d = h->e->dumpTypoGraph(*t->prepared, (uint16_t)allowed_dialect, (const char16_t*)text, len, normalize_coda != 0, use_device != 0); if (out && d.size() <= cap) std::memcpy(out, d.data(), d.size()); return d.size(); }, (size_t)0);
	}
	size_t kamd_typo_lattices(kamd_engine_h h, kamd_typo* t, float threshold, int allowed_dialect, const uint16_t* text, uint32_t len, uint64_t match, uint8_t* out, size_t cap)
	{
		if (!h || !t || !t->prepared) { lastError = "invalid handle / typo transformer not prepared"; return 0; }
		return guarded([&]() { auto d = h->e->dumpTypoLattices(*t->prepared, threshold, (uint16_t)allowed_dialect, (const char16_t*)text, len, match); if (d.size() <= cap) std::memcpy(out, d.data(), d.size()); return d.size(); }, (size_t)0);
	}

	// ---- morpheme sets (AnalyzeOption::blocklist)
	struct kamd_morphset_impl { kamd_engine_h owner; std::vector<uint32_t> ids; std::vector<uint32_t> bits; };
	kamd_morphset_h kamd_morphset_new(kamd_engine_h h)
	{
		if (!h) { lastError = "invalid handle"; return nullptr; }
		return guarded([&]() { auto m = std::make_unique<kamd_morphset_impl>(); m->owner = h; return reinterpret_cast<kamd_morphset_h>(m.release()); }, (kamd_morphset_h)nullptr);
	}
	int kamd_morphset_add(kamd_morphset_h mh, const uint16_t* form, uint32_t len, int tag)
	{
		auto* m = reinterpret_cast<kamd_morphset_impl*>(mh);
		if (!m || !form) return -2;
		return guarded([&]()
		{
			const auto found = findMorphemes(m->owner->e->model(), (const char16_t*)form, len, (uint8_t)(tag < 0 ? 0 : tag));
			m->ids.insert(m->ids.end(), found.begin(), found.end());
			m->bits = blockBitsOf(m->owner->e->model(), m->ids);
			return (int)found.size();
		}, -1);
	}
	void kamd_morphset_close(kamd_morphset_h mh) { delete reinterpret_cast<kamd_morphset_impl*>(mh); }
	kamd_results_h kamd_analyze_batch_opt(kamd_engine_h h, kamd_typo* t, float threshold, int allowed_dialect, kamd_morphset_h blocklist, const uint16_t* texts, const uint64_t* offsets, uint32_t n, uint32_t topN, uint64_t match, int openEnding, int hostThreads)
	{
		if (!h || (t && !t->prepared)) { lastError = "invalid handle / typo transformer not prepared"; return nullptr; }
		auto* m = reinterpret_cast<kamd_morphset_impl*>(blocklist);
		if (m && m->owner != h) { lastError = "the morpheme set belongs to another engine"; return nullptr; }
		return guarded([&]()
		{
			TypoOption o; if (t) { o.typo = t->prepared.get(); o.threshold = threshold; o.allowedDialect = (uint16_t)allowed_dialect; }
			if (m && !m->ids.empty()) o.blocked = &m->bits;
			return pack(h->e->analyzeBatch(views(texts, offsets, n), topN, match, !!openEnding, hostThreads, o));
		}, (kamd_results*)nullptr);
	}

	kamd_results_h kamd_analyze_batch_dialect(kamd_engine_h h, kamd_typo* t, float threshold, int allowed_dialect, float dialect_cost, kamd_morphset_h blocklist, const uint16_t* texts, const uint64_t* offsets, uint32_t n, uint32_t topN, uint64_t match, int openEnding, int hostThreads)
	{
		if (!h || (t && !t->prepared)) { lastError = "invalid handle / typo transformer not prepared"; return nullptr; }
		auto* m = reinterpret_cast<kamd_morphset_impl*>(blocklist);
		if (m && m->owner != h) { lastError = "the morpheme set belongs to another engine"; return nullptr; }
		return guarded([&]()
		{
			TypoOption o; o.allowedDialect = (uint16_t)allowed_dialect; o.dialectCost = dialect_cost;
			if (t) { o.typo = t->prepared.get(); o.threshold = threshold; }
			else if (allowed_dialect) { o.typo = &defaultDialectTypo(); o.threshold = 2.5f; }      // src/Kiwi.cpp:1037-1041
			if (m && !m->ids.empty()) o.blocked = &m->bits;
			return pack(h->e->analyzeBatch(views(texts, offsets, n), topN, match, !!openEnding, hostThreads, o));
		}, (kamd_results*)nullptr);
	}

	// Kiwi::analyze(text, option, pretokenized) for ONE text (src/Kiwi.cpp:1014-1158 with :1043-1051, 785-946): `spans` = per span {begin, end, nTokens} followed by
	// nTokens x {offset of the form in `forms`, its length, begin, end (relative to the span), tag id, inferRegularity}, offsets in UTF-16 units of `text`
	kamd_results_h kamd_analyze_pretokenized(kamd_engine_h h, const uint16_t* text, uint32_t len, uint32_t topN, uint64_t match, int openEnding, const uint32_t* spans, uint32_t nSpans, const uint16_t* forms)
	{
		if (!h) { lastError = "invalid handle"; return nullptr; }
		return guarded([&]()
		{
			std::vector<PtSpan> pt;
			const uint32_t* p = spans;
			for (uint32_t i = 0; i < nSpans; ++i)
			{
				PtSpan sp{ p[0], p[1], {} };
				const uint32_t nTok = p[2];
				p += 3;
				for (uint32_t k = 0; k < nTok; ++k, p += 6) sp.tokens.push_back(PtToken{ std::u16string{ (const char16_t*)forms + p[0], (const char16_t*)forms + p[0] + p[1] }, p[2], p[3], (uint8_t)p[4], p[5] != 0 });
				pt.push_back(std::move(sp));
			}
			return pack(h->e->analyzePretokenized((const char16_t*)text, len, pt, topN, match, !!openEnding, 1));
		}, (kamd_results*)nullptr);
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
