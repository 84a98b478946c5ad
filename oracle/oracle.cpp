// TEST INFRASTRUCTURE ONLY -- never linked into the product.
// C entry points of the CPU oracle (lattice_oracle.hpp + viterbi_oracle.hpp).  The oracle shares the
// host-side model baker and text preparation / result assembly code with the product (those are not the
// hot path); the lattice construction and the best-path search are restated here independently of the
// HIP kernels.  Output buffers use the same byte layout as oracle/ref_bridge.cpp so tests diff them directly.
#include "timed_pool.hpp"
#include <chrono>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <thread>
#include <atomic>
#include "viterbi_oracle.hpp"
#include "typo_oracle.hpp"
#include "typo_lattice_oracle.hpp"

using namespace korc;

struct OracleHandle
{
	FlatModel model;
	ModelView view;
	SbgView sbg;          // present when the raw model carries SkipBigram tables
	CongView cong;        // present when the raw model carries a CoNgram blob: the search is scored with it (the reference's default model type)
	PersistentContainers persistent;   // faithfulOrder: what the reference keeps thread_local (single-threaded entry points only)
	SplitConfig scfg{ 0, 6, 0xFFFFFFFFu, 0 };
	BestPathConfig bcfg;
	bool integrateAllomorph = true;
	float oovChrBias = 0;      // KiwiConfig::oovChrBias
	korc::ChrFreqConfig freq;  // KiwiConfig::oovGlobalWeight / oovLocalWeight / oovGlobalMinFreq
	Counters counters;
	std::vector<uint32_t> blockIds, blockBits;      // AnalyzeOption::blocklist of the following analyses (korc_blocklist_*)
	std::string rawPath; uint32_t enabledDialects = 0;      // where the model came from: a call with pretokenized spans that need temporary entries bakes it again with them
};

namespace
{
	struct Writer
	{
		uint8_t* p; uint8_t* end; size_t need = 0;
		template<class T> void put(T v) { need += sizeof(T); if (p + sizeof(T) <= end) { std::memcpy(p, &v, sizeof(T)); p += sizeof(T); } }
		void putStr(const U16& s) { put<uint32_t>((uint32_t)s.size()); for (auto c : s) put<uint16_t>(c); }
	};

	// a prepared typo transformer applied to an analysis (AnalyzeOption::typoTransformer / typoThreshold / allowedDialects)
	struct TypoOpt { const korc::typo::Prepared* prepared = nullptr; float threshold = 2.5f; uint16_t dialect = 0; float dialectCost = 3.f; };      // (dialect / dialectCost: AnalyzeOption::allowedDialects / dialectCost)

	// Pretokenized spans (Kiwi::analyze(..., pretokenized): src/Kiwi.cpp:785-946, 1043-1051, 1120, 745-756; KTrie.cpp:782-790, 1177-1210): a span without
	// tokens and a span of one token that IS a single-candidate dictionary entry point at a form of the model; any other single token and a span of several
	// tokens get TEMPORARY forms / morphemes (makePretokenizedSpanGroup's ret.forms / ret.morphemes / ret.formStrs), for which the model is baked once more
	// with them behind its own entries (bakeModelWithTemps).  Pinned by tests/test_pretokenized_golden.py against tests/golden/pretokenized_small.json (the
	// real reference's answers).
	struct PtToken { U16 form; uint32_t begin, end; uint8_t tag, infer; };
	struct PtSpan { uint32_t begin, end; std::vector<PtToken> toks; };

	// findForm (src/KTrie.cpp:2172-2192): the form whose string is exactly `nrm`, -1 if none (or only a submatch marker) ends there
	int32_t findFormId(const FlatModel& m, const U16& nrm)
	{
		uint32_t node = 0;
		for (size_t i = 0; i < nrm.size(); ++i)
		{
			const uint16_t c = (uint16_t)nrm[i];
			if (node == 0) { node = m.trieRoot[c]; if (!node) return -1; continue; }
			const TrieNodeRec& t = m.trie[node];
			const uint16_t* kb = m.trieKeys.data() + t.edgeOff;
			const uint16_t* it = std::lower_bound(kb, kb + t.numNexts, c);
			if (it == kb + t.numNexts || *it != c) return -1;
			node = m.trieChild[t.edgeOff + (it - kb)];
		}
		if (node == 0 || m.trie[node].value < 0) return -1;
		return m.trie[node].value;
	}

	std::vector<TokenResult> analyzeOne(OracleHandle& h, Counters& cnt, PersistentContainers* persistent, const char16_t* text, uint32_t len, uint32_t topN, uint64_t match, bool openEnding,
		std::vector<std::vector<LNode>>* latticesOut = nullptr, TypoOpt typo = {}, const std::vector<PtSpan>* pretok = nullptr)
	{
		if (topN < 1 || topN > 16) throw std::runtime_error{ "oracle: top_n must be 1..16" };
		PreparedText pt;
		std::vector<std::pair<uint32_t, uint32_t>> spanCut;      // the spans in normalised offsets
		std::vector<LatticeBuilder::SpanNode> spanNodes;          // ... with their forms (offsets still text-relative here)
		TempEntries temps;                                        // ... and the temporary forms / morphemes they need
		const uint32_t nModelForms = (uint32_t)h.model.h.nForms, nModelMorphs = (uint32_t)h.model.h.nMorphs;
		if (pretok && !pretok->empty())
		{
			if (typo.prepared) throw std::runtime_error{ "oracle: pretokenized spans with a typo transformer are not restated" };
			U16 norm; std::vector<uint32_t> pos;
			normalizeWithPosition(text, len, norm, pos);
			if (match & M_NORMALIZE_CODA) normalizeCoda(norm);
			for (const PtSpan& sp : *pretok)
			{
				if (sp.begin >= sp.end || sp.end > len) throw std::runtime_error{ "oracle: bad pretokenized span" };
				const uint32_t b = pos[sp.begin], e = pos[sp.end];
				LatticeBuilder::SpanNode sn{ b, e, 0, false };
				if (sp.toks.empty())
				{
					const int32_t f = findFormId(h.model, norm.substr(b, e - b));
					if (f >= 0) sn.form = (uint32_t)f;
					else sn.form = (uint32_t)T_NNP - 1u;      // formTrie.value(POSTag::nnp): the default form of the tag
				}
				else if (sp.toks.size() == 1)
				{
					U16 fs, dummy; std::vector<uint32_t> dp;
					normalizeWithPosition(sp.toks[0].form.data(), sp.toks[0].form.size(), fs, dp);
					const int32_t f = findFormId(h.model, fs);
					bool reuse = false;
					if (f >= 0 && h.model.forms[f].candCnt == 1)
					{
						const uint8_t mt = h.model.morphs[h.model.formCand[h.model.forms[f].candOff]].tag, tt = sp.toks[0].tag;
						reuse = sp.toks[0].infer ? ((mt & 0x7F) == (tt & 0x7F)) : (mt == tt);      // areTagsEqual (include/kiwi/Types.h:249-252)
					}
					if (reuse) sn.form = (uint32_t)f;
					else
					{
						// a new form whose candidates are the entry's morphemes of that tag (at most two), or a new morpheme (src/Kiwi.cpp:838-870)
						TempEntries::Form tf; tf.str = fs;
						if (f >= 0)
							for (uint32_t ci = 0; ci < h.model.forms[f].candCnt && tf.cands.size() < 2; ++ci)
							{
								const uint32_t mi = h.model.formCand[h.model.forms[f].candOff + ci];
								const uint8_t mt = h.model.morphs[mi].tag, tt = sp.toks[0].tag;
								if (sp.toks[0].infer ? ((mt & 0x7F) == (tt & 0x7F)) : (mt == tt)) tf.cands.push_back(mi);
							}
						if (tf.cands.empty())
						{
							tf.cands.push_back(nModelMorphs + (uint32_t)temps.morphs.size());
							temps.morphs.push_back(TempEntries::Morph{ (uint32_t)temps.forms.size(), sp.toks[0].tag, (uint32_t)(sp.toks[0].tag & 0x7F) + 1u, {} });      // getDefaultMorphemeId (include/kiwi/Kiwi.h:64-67)
						}
						sn.form = nModelForms + (uint32_t)temps.forms.size();
						temps.forms.push_back(std::move(tf));
					}
				}
				else
				{
					// several tokens: one new form with one new morpheme whose chunks are the tokens -- dictionary morphemes of exactly that form and tag, or new
					// ones with the tag's default LM id (src/Kiwi.cpp:872-934)
					TempEntries::Morph whole{ 0, 0 /* POSTag::unknown */, 0, {} };
					std::vector<TempEntries::Morph> news; std::vector<TempEntries::Form> newForms;
					const uint32_t wholeForm = (uint32_t)temps.forms.size();
					const uint32_t wholeMorph = nModelMorphs + (uint32_t)temps.morphs.size();
					temps.forms.push_back(TempEntries::Form{ U16{}, { wholeMorph } });
					temps.morphs.push_back(whole);      // (filled below: the vector may grow in between)
					const size_t wholeAt = temps.morphs.size() - 1;
					std::vector<TempEntries::Chunk> chunks;
					for (const PtToken& t : sp.toks)
					{
						U16 fs; std::vector<uint32_t> dp;
						normalizeWithPosition(t.form.data(), t.form.size(), fs, dp);
						const int32_t f = findFormId(h.model, fs);
						uint32_t found = 0xFFFFFFFFu;
						if (f >= 0)
							for (uint32_t ci = 0; ci < h.model.forms[f].candCnt; ++ci)
							{
								const uint32_t mi = h.model.formCand[h.model.forms[f].candOff + ci];
								if (h.model.morphs[mi].tag == t.tag) { found = mi; break; }
							}
						if (found == 0xFFFFFFFFu)
						{
							found = nModelMorphs + (uint32_t)temps.morphs.size();
							temps.morphs.push_back(TempEntries::Morph{ (uint32_t)temps.forms.size(), t.tag, (uint32_t)(t.tag & 0x7F) + 1u, {} });
							temps.forms.push_back(TempEntries::Form{ fs, {} });      // (formStrs: the string alone, no candidates)
						}
						if (sp.begin + t.end > len) throw std::runtime_error{ "oracle: bad token range in a pretokenized span" };
						chunks.push_back(TempEntries::Chunk{ found, (uint8_t)(pos[sp.begin + t.begin] - b), (uint8_t)(pos[sp.begin + t.end] - b) });
					}
					temps.morphs[wholeAt].tempForm = wholeForm;
					temps.morphs[wholeAt].chunks = std::move(chunks);
					sn.form = nModelForms + wholeForm;
				}
				sn.fallback = sn.form + 1 >= (uint32_t)T_NNG && sn.form + 1 < (uint32_t)T_MAX;      // within(form, value(nng), value(max)): KTrie.cpp:1197
				spanCut.emplace_back(b, e);
				spanNodes.push_back(sn);
			}
		}
		// temporary entries: this call runs on the model baked once more with them behind its own (restored when the call returns)
		struct ModelSwap      // (moving a FlatModel moves its vectors: the saved views stay valid for the saved model)
		{
			OracleHandle& h; FlatModel saved; ModelView view; SbgView sbg; CongView cong; bool on = false;
			~ModelSwap() { if (on) { h.model = std::move(saved); h.view = view; h.sbg = sbg; h.cong = cong; } }
		} swap{ h, {}, h.view, h.sbg, h.cong };
		if (!temps.forms.empty())
		{
			if (h.rawPath.empty()) throw std::runtime_error{ "oracle: temporary morphemes need the path the model was opened from" };
			FlatModel tm;
			bakeModelWithTemps(tm, h.rawPath, h.enabledDialects, temps);
			// (test hook, KORC_CHECK_OVERLAY=1: the product's per-batch overlay -- bakeTempsOverlay, computed from the baked model alone -- must be exactly what the
			// second bake appended to the tables; tests/test_pretokenized_golden.py runs every golden and live case with it)
			if (std::getenv("KORC_CHECK_OVERLAY"))
			{
				TempOverlay o; bakeTempsOverlay(h.model, temps, o);
				const FlatModel& b0 = h.model;
				auto fail = [](const char* what) { throw std::runtime_error{ std::string{ "overlay != second bake: " } + what }; };
				auto tailEq = [&](const auto& full, const auto& base, const auto& ov, size_t drop, const char* what)
				{
					if (full.size() != base.size() - drop + ov.size() || (ov.size() && std::memcmp(full.data() + (base.size() - drop), ov.data(), ov.size() * sizeof(ov[0])))) fail(what);
				};
				tailEq(tm.morphs, b0.morphs, o.morphs, 0, "morphs"); tailEq(tm.morphKform, b0.morphKform, o.morphKform, 0, "morphKform"); tailEq(tm.sbInfo, b0.sbInfo, o.sbInfo, 0, "sbInfo");
				tailEq(tm.morphPath, b0.morphPath, o.morphPath, 0, "morphPath"); tailEq(tm.chunkMorph, b0.chunkMorph, o.chunkMorph, 0, "chunkMorph"); tailEq(tm.chunkLm, b0.chunkLm, o.chunkLm, 0, "chunkLm");
				tailEq(tm.chunkPos, b0.chunkPos, o.chunkPos, 0, "chunkPos"); tailEq(tm.formCand, b0.formCand, o.formCand, 0, "formCand"); tailEq(tm.formChars, b0.formChars, o.formChars, 1, "formChars");
				tailEq(tm.forms, b0.forms, o.forms, 1, "forms");
				if (!b0.formUnkChr.empty()) { tailEq(tm.formUnkChr, b0.formUnkChr, o.formUnkChr, 0, "formUnkChr"); tailEq(tm.formChrTok, b0.formChrTok, o.formChrTok, 1, "formChrTok"); }
				if (std::memcmp(tm.morphs.data(), b0.morphs.data(), b0.morphs.size() * sizeof(MorphRec)) || std::memcmp(tm.forms.data(), b0.forms.data(), (b0.forms.size() - 1) * sizeof(FormRec))) fail("the model's own entries moved");
				if (const char* log = std::getenv("KORC_CHECK_OVERLAY_LOG")) { if (FILE* f = std::fopen(log, "a")) { std::fprintf(f, "%zu %zu\n", temps.forms.size(), temps.morphs.size()); std::fclose(f); } }
			}
			swap.saved = std::move(h.model); swap.on = true;
			h.model = std::move(tm);
			h.view = h.model.view();
			h.sbg = swap.sbg.present() ? h.model.sbgView() : SbgView{};      // (the language model the handle scores with stays what it was)
			h.cong = swap.cong.present() ? h.model.congView() : CongView{};
		}
		prepareText(pt, text, len, match, 0, spanCut.data(), spanCut.size());
		SplitConfig sc = h.scfg; sc.match = match;
		BestPathConfig bc = h.bcfg;
		bc.topN = topN;
		bc.allowedDialect = typo.dialect; bc.dialectCost = typo.dialectCost;
		bc.splitComplex = match & M_SPLIT_COMPLEX; bc.splitSaisiot = match & M_SPLIT_SAISIOT; bc.mergeSaisiot = match & M_MERGE_SAISIOT;
		bc.spaceTolerance = sc.spaceTol;
		// Match::oovMask (include/kiwi/PatternMatcher.h:20-24): 1 = the character model scores unknown forms; 2 / 3: mixed with the substring counts of
		// the filtered text (Kiwi.cpp:1058-1086, 1138; 3 -- "branch" -- evaluates the same expression: src/UnkFormScorer.cpp:118-121)
		const kamd::ChrView chrV = h.model.chrView();
		korc::SubstringCounts substr;
		if ((match >> 8) & 3)
		{
			if (!chrV.present()) throw std::runtime_error{ "`oovChrModel` option is set but the character-level noun model is not loaded." };      // Kiwi.cpp:1032-1035
			bc.chr = &chrV; bc.oovChrBias = h.oovChrBias;
			if (((match >> 8) & 3) > 1)
			{
				const std::u16string filtered = korc::filteredText((const char16_t*)pt.norm.data(), pt.norm.size());
				substr.build(filtered.data(), filtered.size());
				bc.substr = &substr; bc.freq = h.freq;
			}
		}
		LatticeBuilder lb{ h.view, sc, cnt };
		TypoLatticeBuilder tlb{ h.view, sc };
		tlb.countInto(&cnt);
		ResultBuilder rb{ h.model, topN, match, h.integrateAllomorph };
		rb.begin(text, len, pt.position.data(), pt.position.size());
		std::vector<LNode> nodes;
		std::vector<PathResult> paths;
		for (auto& ch : pt.chunks)
		{
			if (ch.empty) continue;
			// the spans of this chunk (the cut never ends inside one), chunk-relative
			std::vector<LatticeBuilder::SpanNode> chSpans;
			for (const auto& sn : spanNodes) if (sn.begin >= ch.startOffset && sn.begin < ch.startOffset + ch.nChars) chSpans.push_back({ sn.begin - ch.startOffset, sn.end - ch.startOffset, sn.form, sn.fallback });
			const bool ok = typo.prepared
				? tlb.build(nodes, pt.norm.data() + ch.startOffset, ch.nChars, pt.patterns.data() + ch.patBegin, pt.patterns.data() + ch.patEnd, ch.startOffset, *typo.prepared, typo.threshold, typo.dialect)
				: lb.build(nodes, pt.norm.data() + ch.startOffset, ch.nChars, pt.cls.data() + ch.startOffset, pt.script.data() + ch.startOffset,
				pt.patterns.data() + ch.patBegin, pt.patterns.data() + ch.patEnd, ch.startOffset, chSpans.data(), chSpans.data() + chSpans.size());
			if (latticesOut) latticesOut->push_back(nodes);
			if (!ok) continue;
			BestPathConfig bc2 = bc;
			bc2.openEnding = openEnding && ch.nextOffset == pt.norm.size();
			BestPathSearch bp{ h.view, bc2, cnt, h.sbg, persistent, h.cong };
			bp.run(paths, pt.norm, pt.cls, nodes.data(), (uint32_t)nodes.size(), rb.spStates());
			if (!chSpans.empty())
			{
				// findPretokenizedGroupOfNode (src/Kiwi.cpp:949-969) + Kiwi.cpp:745-750: a token of a node inside span i of the CHUNK reports i + 1 as its typoFormId
				std::vector<uint32_t> group(nodes.size(), 0);
				size_t cur = 0;
				for (size_t i = 0; i < nodes.size(); ++i)
				{
					while (cur < chSpans.size() && nodes[i].startPos >= chSpans[cur].end + ch.startOffset) ++cur;
					if (cur == chSpans.size()) break;
					if (chSpans[cur].begin + ch.startOffset <= nodes[i].startPos && nodes[i].endPos <= chSpans[cur].end + ch.startOffset) group[i] = (uint32_t)cur + 1;
				}
				for (auto& p : paths) for (auto& t : p.path) if (t.nodeId < group.size() && group[t.nodeId]) t.typoFormId = group[t.nodeId];
			}
			rb.insertPaths(paths);
		}
		auto res = rb.finish(text, len);
		if (swap.on) for (auto& r : res) for (auto& t : r.first) if (t.morph >= (int32_t)nModelMorphs) t.morph = -1;      // (token.morph = nullptr for the span group's own morphemes, src/Kiwi.cpp:733)
		return res;
	}
}

extern "C"
{
	void* korc_open(const char* rawModelPath)
	{
		try
		{
			auto h = std::make_unique<OracleHandle>();
			bakeModel(h->model, rawModelPath);
			h->rawPath = rawModelPath;
			h->view = h->model.view();
			h->sbg = h->model.sbgView();
			h->cong = h->model.congView();
			if (h->cong.present()) h->sbg = SbgView{};
			return h.release();
		}
		catch (const std::exception& e) { fprintf(stderr, "korc_open: %s\n", e.what()); return nullptr; }
	}
	// the bake with KiwiBuilder's enabledDialects (kiwi_init's last argument)
	void* korc_open_dialects(const char* rawModelPath, int enabledDialects)
	{
		try
		{
			auto h = std::make_unique<OracleHandle>();
			bakeModel(h->model, rawModelPath, (uint32_t)enabledDialects);
			h->rawPath = rawModelPath; h->enabledDialects = (uint32_t)enabledDialects;
			h->view = h->model.view();
			h->sbg = h->model.sbgView();
			h->cong = h->model.congView();
			if (h->cong.present()) h->sbg = SbgView{};
			return h.release();
		}
		catch (const std::exception& e) { fprintf(stderr, "korc_open_dialects: %s\n", e.what()); return nullptr; }
	}
	void korc_close(void* h) { delete (OracleHandle*)h; }

	void korc_set_config(void* hp, float cutOff, float spacePenalty, float typoCostWeight, uint32_t maxUnk, uint32_t maxUnkJ, uint32_t spaceTol, int integrateAllomorph)
	{
		auto& h = *(OracleHandle*)hp;
		h.bcfg.cutOff = cutOff; h.bcfg.spacePenalty = spacePenalty; h.bcfg.typoCostWeight = typoCostWeight;
		h.scfg.maxUnk = maxUnk; h.scfg.maxUnkJ = maxUnkJ; h.scfg.spaceTol = spaceTol; h.integrateAllomorph = !!integrateAllomorph;
	}

	// probe of csrc/cong_global.hpp's four 8-term kernels (tests/test_cong_global.py, against the reference's own lm::logSoftmax / logSumExp / ...Transposed
	// of its SSE4.1 build): which 0 = logSoftmax (in place), 1 = logSumExp, 2 / 3 = the transposed (per-lane) forms; returns the scalar result of 1 / 3
	float korc_congg_math(int which, float* w8)
	{
		switch (which)
		{
		case 0: congg::logSoftmax8(w8); return 0;
		case 1: return congg::logSumExp8(w8);
		case 2: congg::logSoftmaxT8(w8); return 0;
		default: return congg::logSumExpT8(w8);
		}
	}

	// the host evaluation behind kamd_debug_cong_global's device probe: same arguments, the oracle's own model (congGlobal must be on)
	int korc_congg_scores(void* hp, const uint32_t* ctx, const uint32_t* hist7, const uint32_t* next, const uint8_t* flags, float* out, uint32_t n)
	{
		auto& h = *(OracleHandle*)hp;
		const CongView& C = h.cong;
		if (!C.window) return -1;
		for (uint32_t i = 0; i < n; ++i)
		{
			const bool matrix = flags[i] & 1, outFirst = flags[i] & 2;
			if (C.distant(next[i])) out[i] = matrix ? congg::scoreMatrix(C, ctx[i], hist7 + 7ull * i, next[i], outFirst) : congg::scoreSingle(C, ctx[i], hist7 + 7ull * i, next[i]);
			else out[i] = outFirst ? congScoreOutputFirst(C, ctx[i], next[i]) : congScore(C, ctx[i], next[i]);
		}
		return 0;
	}

	// ModelType::congGlobal: score with the window sections of the CoNgram file (0 = ok, -1 = the model has none)
	int korc_set_cong_global(void* hp, int on)
	{
		auto& h = *(OracleHandle*)hp;
		if (on && !(h.model.congDim && h.model.congWindow)) return -1;
		h.model.congGlobal = !!on;
		h.cong = h.model.congView();
		return 0;
	}

	// test hook: hand kept paths on in the reference's own (history-dependent) container order instead of insertion order
	void korc_set_faithful_order(void* hp, int on)
	{
		auto& h = *(OracleHandle*)hp;
		h.bcfg.faithfulOrder = !!on;
		h.persistent = PersistentContainers{};
	}

	// AnalyzeOption::blocklist for every analysis that follows: kiwi_morphset_add (src/capi/kiwi_c.cpp:1779-1794) = Kiwi::findMorphemes(form, tag)
	void korc_blocklist_clear(void* hp)
	{
		auto& h = *(OracleHandle*)hp;
		h.blockIds.clear(); h.blockBits.clear(); h.bcfg.blockBits = nullptr;
	}
	int korc_blocklist_add(void* hp, const uint16_t* form, uint32_t len, int tag)
	{
		auto& h = *(OracleHandle*)hp;
		const auto found = findMorphemes(h.model, (const char16_t*)form, len, (uint8_t)(tag < 0 ? 0 : tag));
		h.blockIds.insert(h.blockIds.end(), found.begin(), found.end());
		h.blockBits = blockBitsOf(h.model, h.blockIds);
		h.bcfg.blockBits = h.blockIds.empty() ? nullptr : h.blockBits.data();
		return (int)found.size();
	}

	// test hook: container selection limits (defaults 128, 512, 128)
	void korc_set_container_limits(void* hp, uint32_t smallMax, uint32_t mediumMax, uint32_t bucketCap)
	{
		auto& h = *(OracleHandle*)hp;
		h.bcfg.smallMax = smallMax; h.bcfg.mediumMax = mediumMax; h.bcfg.bucketCap = bucketCap;
	}

	size_t korc_dump_dict(void* hp, uint8_t* out, size_t cap)
	{
		auto d = dumpDict(((OracleHandle*)hp)->model);
		if (d.size() <= cap) std::memcpy(out, d.data(), d.size());
		return d.size();
	}

	// One CoNgram step (CoNgramState::next): node and context id in / out
	float korc_cong_next(void* hp, int32_t* node, uint32_t* ctx, uint32_t wid)
	{
		auto& h = *(OracleHandle*)hp;
		BestPathConfig bc; Counters c;
		BestPathSearch bp{ h.view, bc, c, SbgView{}, nullptr, h.cong };
		WPath st; st.lmNode = *node; st.ctx = *ctx;
		const float ll = bp.lmNext(st, wid);
		*node = st.lmNode; *ctx = st.ctx;
		return ll;
	}
	// the same with the global model's history (CoNgramState<7>::history, 8 words) in / out
	float korc_cong_next_hist(void* hp, int32_t* node, uint32_t* ctx, uint32_t* hist8, uint32_t wid)
	{
		auto& h = *(OracleHandle*)hp;
		BestPathConfig bc; Counters c;
		BestPathSearch bp{ h.view, bc, c, SbgView{}, nullptr, h.cong };
		WPath st; st.lmNode = *node; st.ctx = *ctx;
		for (int i = 0; i < 8; ++i) st.hist[i] = hist8[i];
		const float ll = bp.lmNext(st, wid);
		*node = st.lmNode; *ctx = st.ctx;
		for (int i = 0; i < 8; ++i) hist8[i] = st.hist[i];
		return ll;
	}

	float korc_lm_progress(void* hp, int32_t* node, uint32_t wid)
	{
		auto& h = *(OracleHandle*)hp;
		BestPathConfig bc; Counters c;
		BestPathSearch bp{ h.view, bc, c };
		return bp.lmProgress(*node, wid);
	}

	// One LM step of the model's state type (Knlm, or SkipBigram on top of it): node, ring position and 8 history words in / out
	float korc_lm_next(void* hp, int32_t* node, uint32_t* pos, uint32_t* hist8, uint32_t wid)
	{
		auto& h = *(OracleHandle*)hp;
		BestPathConfig bc; Counters c;
		BestPathSearch bp{ h.view, bc, c, h.sbg };
		WPath st; st.lmNode = *node; st.histPos = (uint8_t)*pos; for (int i = 0; i < 8; ++i) st.hist[i] = hist8[i];
		const float ll = bp.lmNext(st, wid);
		*node = st.lmNode; *pos = st.histPos; for (int i = 0; i < 8; ++i) hist8[i] = st.hist[i];
		return ll;
	}

	// UnkFormScorer::chrBasedScore of a (normalised) string with bias 0; NaN when the model has no character model
	float korc_unk_chr_score(void* hp, const uint16_t* s, uint32_t len)
	{
		auto& h = *(OracleHandle*)hp;
		const kamd::ChrView C = h.model.chrView();
		if (!C.present()) return 0.f / 0.f;
		return kamd::chrScoreHost(C, s, len);
	}
	void korc_set_oov_chr_bias(void* hp, float bias) { ((OracleHandle*)hp)->oovChrBias = bias; }
	void korc_set_oov_freq_params(void* hp, float globalWeight, float localWeight, float globalMinFreq) { ((OracleHandle*)hp)->freq = korc::ChrFreqConfig{ globalWeight, localWeight, globalMinFreq }; }
	// UnkFormScorer::chrFreqBasedScore of a (normalised) string against the substring counts of an (already filtered) text, bias 0
	float korc_unk_chr_freq_score(void* hp, const uint16_t* text, uint32_t textLen, const uint16_t* s, uint32_t len)
	{
		auto& h = *(OracleHandle*)hp;
		const kamd::ChrView C = h.model.chrView();
		if (!C.present()) return 0.f / 0.f;
		korc::SubstringCounts sc; sc.build((const char16_t*)text, textLen);
		return korc::chrFreqScoreOracle(C, sc, h.freq, 0.f, s, len);
	}

	struct TypoHandle { korc::typo::Rules rules; std::unique_ptr<korc::typo::Prepared> prepared; };
	size_t korc_split_typo(void* hp, void* typoHp, float typoThreshold, int allowedDialect, const uint16_t* text, uint32_t len, uint64_t match, uint8_t* out, size_t cap);
	// same layout as kref_split (oracle/ref_bridge.cpp)
	size_t korc_split(void* hp, const uint16_t* text, uint32_t len, uint64_t match, uint8_t* out, size_t cap)
	{
		return korc_split_typo(hp, nullptr, 2.5f, 0, text, len, match, out, cap);
	}
	size_t korc_split_typo(void* hp, void* typoHp, float typoThreshold, int allowedDialect, const uint16_t* text, uint32_t len, uint64_t match, uint8_t* out, size_t cap)
	{
		auto& h = *(OracleHandle*)hp;
		const korc::typo::Prepared* prepared = typoHp ? ((TypoHandle*)typoHp)->prepared.get() : nullptr;
		Writer w{ out, out + cap };
		try
		{
			PreparedText pt;
			prepareText(pt, (const char16_t*)text, len, match, 0);
			SplitConfig sc = h.scfg; sc.match = match;
			Counters cnt;
			LatticeBuilder lb{ h.view, sc, cnt };
			TypoLatticeBuilder tlb{ h.view, sc };
			w.put<uint32_t>((uint32_t)pt.chunks.size());
			std::vector<LNode> nodes;
			for (auto& ch : pt.chunks)
			{
				if (ch.empty) { nodes.assign(2, LNode{}); }
				else if (prepared) tlb.build(nodes, pt.norm.data() + ch.startOffset, ch.nChars, pt.patterns.data() + ch.patBegin, pt.patterns.data() + ch.patEnd, ch.startOffset, *prepared, typoThreshold, (uint16_t)allowedDialect);
				else lb.build(nodes, pt.norm.data() + ch.startOffset, ch.nChars, pt.cls.data() + ch.startOffset, pt.script.data() + ch.startOffset,
					pt.patterns.data() + ch.patBegin, pt.patterns.data() + ch.patEnd, ch.startOffset);
				w.put<uint32_t>((uint32_t)nodes.size());
				w.put<uint32_t>(ch.nextOffset);
				for (auto& n : nodes)
				{
					w.put<uint32_t>(n.startPos); w.put<uint32_t>(n.endPos); w.put<uint32_t>(n.prev); w.put<uint32_t>(n.sibling);
					w.put<int32_t>(n.form == NOFORM ? -1 : (int32_t)n.form);
					w.put<uint32_t>(n.uformLen); w.put<uint32_t>(n.uformLen ? n.uformOff : 0);
					w.put<uint32_t>(n.spaceErrors); w.put<float>(n.typoCost);
				}
			}
		}
		catch (const std::exception& e) { fprintf(stderr, "korc_split: %s\n", e.what()); return 0; }
		return w.need;
	}

	static void writeResults(Writer& w, const std::vector<TokenResult>& res)
	{
		w.put<uint32_t>((uint32_t)res.size());
		for (auto& r : res)
		{
			w.put<float>(r.second);
			w.put<uint32_t>((uint32_t)r.first.size());
			for (auto& t : r.first)
			{
				w.putStr(t.str);
				w.put<uint32_t>(t.position); w.put<uint32_t>(t.wordPosition); w.put<uint32_t>(t.sentPosition); w.put<uint32_t>(t.lineNumber);
				w.put<uint16_t>(t.length); w.put<uint8_t>(t.tag); w.put<uint8_t>(t.senseId);
				w.put<float>(t.score); w.put<float>(t.typoCost); w.put<uint32_t>(t.typoFormId); w.put<uint32_t>(t.pairedToken);
				w.put<uint32_t>(t.subSentPosition); w.put<uint16_t>(t.dialect); w.put<int32_t>(t.morph);
			}
		}
	}

	size_t korc_analyze_typo(void* hp, void* typoHp, float typoThreshold, int allowedDialect, const uint16_t* text, uint32_t len, uint32_t topN, uint64_t match, int openEnding, uint8_t* out, size_t cap);
	size_t korc_analyze(void* hp, const uint16_t* text, uint32_t len, uint32_t topN, uint64_t match, int openEnding, uint8_t* out, size_t cap)
	{
		return korc_analyze_typo(hp, nullptr, 2.5f, 0, text, len, topN, match, openEnding, out, cap);
	}
	size_t korc_analyze_typo(void* hp, void* typoHp, float typoThreshold, int allowedDialect, const uint16_t* text, uint32_t len, uint32_t topN, uint64_t match, int openEnding, uint8_t* out, size_t cap)
	{
		auto& h = *(OracleHandle*)hp;
		Writer w{ out, out + cap };
		try
		{
			TypoOpt typo;
			if (typoHp) { typo.prepared = ((TypoHandle*)typoHp)->prepared.get(); typo.threshold = typoThreshold; typo.dialect = (uint16_t)allowedDialect; }
			auto res = analyzeOne(h, h.counters, &h.persistent, (const char16_t*)text, len, topN, match, !!openEnding, nullptr, typo);
			writeResults(w, res);
		}
		catch (const std::exception& e) { fprintf(stderr, "korc_analyze: %s\n", e.what()); return 0; }
		return w.need;
	}

	// AnalyzeOption::allowedDialects / dialectCost.  The caller hands in the typo transformer -- the reference takes its built-in `dialect` set by itself when a
	// dialect is allowed and none is given (src/Kiwi.cpp:1037-1041): oraclelib.OracleKiwi.analyze_dialect does the same with that set's entries.
	size_t korc_analyze_dialect(void* hp, void* typoHp, float typoThreshold, int allowedDialect, float dialectCost, const uint16_t* text, uint32_t len, uint32_t topN, uint64_t match, int openEnding, uint8_t* out, size_t cap)
	{
		auto& h = *(OracleHandle*)hp;
		Writer w{ out, out + cap };
		try
		{
			TypoOpt typo;
			if (typoHp) { typo.prepared = ((TypoHandle*)typoHp)->prepared.get(); typo.threshold = typoThreshold; }
			typo.dialect = (uint16_t)allowedDialect; typo.dialectCost = dialectCost;
			auto res = analyzeOne(h, h.counters, &h.persistent, (const char16_t*)text, len, topN, match, !!openEnding, nullptr, typo);
			writeResults(w, res);
		}
		catch (const std::exception& e) { fprintf(stderr, "korc_analyze_dialect: %s\n", e.what()); return 0; }
		return w.need;
	}

	// Kiwi::analyze with pretokenized spans, as far as restated (see analyzeOne); the argument layout of kref_analyze_pretokenized (oracle/ref_bridge.cpp).
	// Returns 0 -- with the reason on stderr -- for a span that is not restated.
	size_t korc_analyze_pretokenized(void* hp, const uint16_t* text, uint32_t len, uint32_t topN, uint64_t match, const uint32_t* spans, uint32_t nSpans, const uint16_t* forms, uint8_t* out, size_t cap)
	{
		auto& h = *(OracleHandle*)hp;
		Writer w{ out, out + cap };
		try
		{
			std::vector<PtSpan> pt;
			const uint32_t* p = spans;
			for (uint32_t i = 0; i < nSpans; ++i)
			{
				PtSpan sp{ p[0], p[1], {} };
				const uint32_t nTok = p[2];
				p += 3;
				for (uint32_t t = 0; t < nTok; ++t, p += 6) sp.toks.push_back(PtToken{ U16{ (const char16_t*)forms + p[0], (const char16_t*)forms + p[0] + p[1] }, p[2], p[3], (uint8_t)p[4], (uint8_t)p[5] });
				pt.push_back(std::move(sp));
			}
			auto res = analyzeOne(h, h.counters, &h.persistent, (const char16_t*)text, len, topN, match, false, nullptr, TypoOpt{}, &pt);
			writeResults(w, res);
		}
		catch (const std::exception& e) { if (!std::getenv("KORC_QUIET")) fprintf(stderr, "korc_analyze_pretokenized: %s\n", e.what()); return 0; }
		return w.need;
	}

	// How many cores this process really gets: `threads` threads each run the same fixed amount of register-only integer work (no memory, no
	// allocation, nothing shared); returns aggregate work per second relative to one thread doing it alone.  A container may see 256 logical CPUs
	// and be scheduled on a fraction of them (CPU quota): the multi-thread baseline cannot scale beyond this number, whatever the code does.
	double korc_cpu_capacity(int threads, double minSeconds)
	{
		auto spin = [](uint64_t seed) { uint64_t x = seed | 1; for (uint32_t i = 0; i < 20000000u; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; } return x; };
		std::atomic<uint64_t> sink{ 0 };
		uint32_t p1 = 0, pn = 0;
		const double s1 = timedpool::run(1, 4, minSeconds / 2, &p1, [&](int, uint32_t i) { sink.fetch_xor(spin(i + 1), std::memory_order_relaxed); });
		const double sn = timedpool::run(threads, 4u * (uint32_t)threads, minSeconds / 2, &pn, [&](int, uint32_t i) { sink.fetch_xor(spin(i + 1), std::memory_order_relaxed); });
		const double r1 = 4.0 * p1 / s1, rn = 4.0 * threads * pn / sn;
		return sink.load() == 0x1234567ull ? 0.0 : rn / r1;
	}

	// CPU baseline ("port" kind): batch over `threads` workers; returns wall seconds
	// the batch timed soundly (timed_pool.hpp): persistent threads, one untimed warm-up pass, whole passes until >= minSeconds of wall
	double korc_analyze_batch_timed(void* hp, void* typoHp, float typoThreshold, const uint16_t* texts, const uint64_t* offsets, uint32_t n, uint32_t topN, uint64_t match, int threads, double minSeconds, uint32_t* passesOut, uint64_t* tokensOut)
	{
		auto& h = *(OracleHandle*)hp;
		TypoOpt typo;
		if (typoHp) { typo.prepared = ((TypoHandle*)typoHp)->prepared.get(); typo.threshold = typoThreshold; }
		std::atomic<uint64_t> tokens{ 0 };
		std::vector<Counters> cnts(std::max(threads, 1));
		uint32_t passes = 0;
		const double sec = timedpool::run(threads, n, minSeconds, &passes, [&](int tid, uint32_t i)
		{
			auto res = analyzeOne(h, cnts[tid], nullptr, (const char16_t*)texts + offsets[i], (uint32_t)(offsets[i + 1] - offsets[i]), topN, match, false, nullptr, typo);
			tokens.fetch_add(res[0].first.size(), std::memory_order_relaxed);
		});
		if (passesOut) *passesOut = passes;
		if (tokensOut) *tokensOut = tokens.load() / (passes + 1);
		return sec;
	}
	double korc_analyze_batch_typo(void* hp, void* typoHp, float typoThreshold, const uint16_t* texts, const uint64_t* offsets, uint32_t n, uint32_t topN, uint64_t match, int threads, uint64_t* tokensOut);
	double korc_analyze_batch(void* hp, const uint16_t* texts, const uint64_t* offsets, uint32_t n, uint32_t topN, uint64_t match, int threads, uint64_t* tokensOut)
	{
		return korc_analyze_batch_typo(hp, nullptr, 2.5f, texts, offsets, n, topN, match, threads, tokensOut);
	}
	double korc_analyze_batch_typo(void* hp, void* typoHp, float typoThreshold, const uint16_t* texts, const uint64_t* offsets, uint32_t n, uint32_t topN, uint64_t match, int threads, uint64_t* tokensOut)
	{
		auto& h = *(OracleHandle*)hp;
		TypoOpt typo;
		if (typoHp) { typo.prepared = ((TypoHandle*)typoHp)->prepared.get(); typo.threshold = typoThreshold; }
		std::atomic<uint32_t> next{ 0 };
		std::atomic<uint64_t> tokens{ 0 };
		std::vector<Counters> cnts(std::max(threads, 1));
		auto work = [&](int tid)
		{
			uint64_t local = 0;
			for (;;)
			{
				const uint32_t i = next.fetch_add(1);
				if (i >= n) break;
				auto res = analyzeOne(h, cnts[tid], nullptr, (const char16_t*)texts + offsets[i], (uint32_t)(offsets[i + 1] - offsets[i]), topN, match, false, nullptr, typo);
				local += res[0].first.size();
			}
			tokens += local;
		};
		auto t0 = std::chrono::steady_clock::now();
		if (threads <= 1) work(0);
		else
		{
			std::vector<std::thread> ts;
			for (int t = 0; t < threads; ++t) ts.emplace_back(work, t);
			for (auto& t : ts) t.join();
		}
		auto t1 = std::chrono::steady_clock::now();
		if (tokensOut) *tokensOut = tokens.load();
		// fold counters
		for (auto& c : cnts)
		{
			uint64_t* d = (uint64_t*)&h.counters; const uint64_t* s = (const uint64_t*)&c;
			for (size_t k = 0; k < sizeof(Counters) / 8; ++k) { if (k == 13) d[k] = std::max(d[k], s[k]); else d[k] += s[k]; }
		}
		return std::chrono::duration<double>(t1 - t0).count();
	}

	void korc_counters(void* hp, uint64_t* out21, int reset)
	{
		auto& h = *(OracleHandle*)hp;
		std::memcpy(out21, &h.counters, sizeof(Counters));
		if (reset) h.counters = Counters{};
	}

	// ---- typo graphs (typo_oracle.hpp); byte layouts as oracle/ref_bridge.cpp kref_typo_* ------------------------------------------------
	void* korc_typo_new(float continualCost, float lengtheningCost)
	{
		auto* h = new TypoHandle; h->rules.continualCost = continualCost; h->rules.lengtheningCost = lengtheningCost; return h;
	}
	void korc_typo_close(void* hp) { delete (TypoHandle*)hp; }
	int korc_typo_add(void* hp, const uint16_t* orig, uint32_t nOrig, const uint16_t* err, uint32_t nErr, float cost, int cond, int dialect)
	{
		try { ((TypoHandle*)hp)->rules.add(std::u16string{ (const char16_t*)orig, nOrig }, std::u16string{ (const char16_t*)err, nErr }, cost, (uint8_t)cond, (uint16_t)dialect); return 0; }
		catch (const std::exception&) { return -1; }
	}
	void korc_typo_add_entry(void* hp, const uint16_t* orig, uint32_t nOrig, const uint16_t* err, uint32_t nErr, float cost, int cond, int dialect)
	{
		((TypoHandle*)hp)->rules.addEntry(std::u16string{ (const char16_t*)orig, nOrig }, std::u16string{ (const char16_t*)err, nErr }, cost, (uint8_t)cond, (uint16_t)dialect);
	}
	void korc_typo_set_costs(void* hp, float continualCost, float lengtheningCost) { auto* h = (TypoHandle*)hp; h->rules.continualCost = continualCost; h->rules.lengtheningCost = lengtheningCost; }
	void korc_typo_prepare(void* hp, int inverse) { auto* h = (TypoHandle*)hp; h->prepared.reset(new korc::typo::Prepared{ h->rules, inverse != 0 }); }
	size_t korc_typo_graph(void* hp, const uint16_t* text, uint32_t len, int allowedDialect, int normCoda, uint8_t* out, size_t cap)
	{
		auto* h = (TypoHandle*)hp;
		U16 norm; std::vector<uint32_t> pos;
		normalizeWithPosition((const char16_t*)text, len, norm, pos);
		if (normCoda) normalizeCoda(norm);
		size_t maxIdx = 0;
		const auto g = h->prepared->graph(std::u16string{ (const char16_t*)norm.data(), norm.size() }, (uint16_t)allowedDialect, maxIdx);
		Writer w{ out, out + cap };
		w.put<uint32_t>((uint32_t)norm.size()); for (auto c : norm) w.put<uint16_t>((uint16_t)c);
		w.put<uint32_t>((uint32_t)g.size());
		for (auto& n : g)
		{
			w.put<uint32_t>((uint32_t)n.form.size()); for (auto c : n.form) w.put<uint16_t>((uint16_t)c);
			w.put<uint32_t>(n.endPos); w.put<float>(n.typoCost); w.put<uint32_t>(n.prevOffset); w.put<uint32_t>(n.siblingOffset);
			w.put<uint8_t>(n.continualTypoIdx); w.put<uint16_t>(n.dialect);
		}
		w.put<uint32_t>((uint32_t)maxIdx);
		return w.need;
	}
}

#ifdef KORC_HAZARD_STATS
extern "C" void korc_haz_stats(uint64_t* out12, int reset)
{
	auto& h = korc::hazStats();
	const uint64_t v[12] = { h.chunks, h.chunksHaz, h.live, h.h1enter, h.h1act, h.h2, h.h3, h.h4, h.ops, h.appFailReach, h.steps, h.maxCand };
	std::memcpy(out12, v, sizeof(v));
	if (reset) h = korc::HazStats{};
}
#endif
