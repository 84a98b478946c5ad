// TEST INFRASTRUCTURE ONLY -- never linked into the product.
//
// Bridge that lets the REAL reference translation units (compiled where they lie under
// /root/reference by oracle/Makefile into oracle/_ref/libkiwi_ref.so) run on the synthetic
// "raw model" this repo generates (kiwi_amd/synth.py).  The reference cannot load its own
// models here: the binaries are git-LFS pointers and src/KiwiBuilder.cpp does not compile
// without Eigen.  So this file re-does, with the reference's own functions,
// the part of KiwiBuilder::build() that bakes a Kiwi object
// (/root/reference/src/KiwiBuilder.cpp:2385-2640): bake() (src/Form.cpp:93-141),
// utils::sortWriteInvIdx (src/SortUtils.hpp:235), ContinuousTrie::buildWithCaching
// (include/kiwi/Trie.hpp:444), utils::freezeTrie (src/FrozenTrie.hpp:176) and
// lm::KnLangModelBase::create (src/Knlm.cpp:176).  Everything downstream -- Kiwi::analyze,
// splitByTrie, BestPathFinder<KnLangModel>::findBestPath -- is the untouched reference.
//
// Private members of kiwi::Kiwi are reached through an explicit specialisation of
// BestPathFinder<>, which Kiwi befriends for every template argument (include/kiwi/Kiwi.h:176).
#include <cstring>
#include <chrono>
#include <fstream>
#include <thread>
#include "timed_pool.hpp"
#include <atomic>
#include <malloc.h>
#include <numeric>

#include <kiwi/Kiwi.h>
#include <kiwi/PatternMatcher.h>
#include <kiwi/Utils.h>
#include <kiwi/Form.h>
#include <kiwi/Knlm.h>
#include <kiwi/SkipBigramModel.h>
#include "SkipBigramModel.hpp"
#ifdef KREF_X86
#include <kiwi/CoNgramModel.h>
#endif
#include <kiwi/Dataset.h>
#include "ArchAvailable.h"
#include "KTrie.h"
#include "FrozenTrie.hpp"
#include "StrUtils.h"
#include "SortUtils.hpp"
#include "PathEvaluator.h"
#include "MathFunc.hpp"
#include "serializer.hpp"
#include "UnkFormScorer.h"
#include "SubstringCounter.hpp"

#include "../kiwi_amd/csrc/container.hpp"
#include "../kiwi_amd/csrc/raw_model.hpp"

namespace kiwi
{
	namespace lm
	{
		// src/archImpl/none.cpp:8-16 cannot be compiled (pulls Eigen); the scalar LSE it would
		// instantiate is header-only (src/MathFunc.hpp:35-55), so instantiate it here.
		template float logSumExp<ArchType::none>(const float* arr, size_t size);
		template float logSumExp<ArchType::balanced>(const float* arr, size_t size);
		template float logSumExp<ArchType::sse2>(const float* arr, size_t size);
	}
#ifndef KREF_X86
	// src/Dataset.cpp:805 (needs Eigen-dependent headers); only used by the optional OOV character model, which only the x86 build of the bridge
	// enables -- that build compiles the real src/Dataset.cpp over the Eigen stand-in
	size_t ChrTokenizer::encodeOne(char32_t) const { throw std::runtime_error{ "ChrTokenizer unavailable in ref bridge" }; }
#endif
}

namespace kamd_ref { struct Access; }

namespace kiwi
{
	static constexpr size_t defaultFormSize = defaultTagSize + 26; // src/KiwiBuilder.cpp:40

	// restated file-local helpers of src/KiwiBuilder.cpp (not reachable from outside that TU)
	static CondVowel reduceVowelR(CondVowel v, const Morpheme* m) // KiwiBuilder.cpp:2258-2278
	{
		if (v == m->vowel) return v;
		if (CondVowel::vowel <= v && v <= CondVowel::vocalic_h)
		{
			if (CondVowel::vowel <= m->vowel && m->vowel <= CondVowel::vocalic_h) return std::max(v, m->vowel);
			return CondVowel::none;
		}
		else if (CondVowel::non_vowel <= v && v <= CondVowel::non_vocalic_h)
		{
			if (CondVowel::non_vowel <= m->vowel && m->vowel <= CondVowel::non_vocalic_h) return std::min(v, m->vowel);
			return CondVowel::none;
		}
		return CondVowel::none;
	}
	static CondPolarity reducePolarR(CondPolarity p, const Morpheme* m) { return p == m->polar ? p : CondPolarity::none; }
	static Dialect reduceDialectR(Dialect d, const Morpheme* m)
	{
		if (d == Dialect::standard || m->dialect == Dialect::standard) return Dialect::standard;
		return d | m->dialect;
	}
	static bool zCodaAppendableR(const KString& form, const Vector<uint32_t>& cand, const Vector<MorphemeRaw>& ms) // :2294-2323
	{
		if (form.empty() || !isHangulSyllable(form.back())) return false;
		for (auto i : cand)
		{
			auto tag = ms[i].tag;
			if (tag == POSTag::unknown && !ms[i].chunks.empty()) tag = ms[ms[i].chunks.back()].tag;
			if (isJClass(tag) || isEClass(tag)) return true;
		}
		return false;
	}
	static bool zSiotAppendableR(const KString& form, const Vector<uint32_t>& cand, const Vector<MorphemeRaw>& ms) // :2325-2352
	{
		if (form.empty() || !isHangulSyllable(form.back()) || isHangulCoda(form.back())) return false;
		for (auto i : cand)
		{
			const auto tag = ms[i].tag;
			if (!isNNClass(tag)) continue;
			if (ms[i].lmMorphemeId != getDefaultMorphemeId(tag)) return true;
		}
		return false;
	}

	template<>
	struct BestPathFinder<kamd_ref::Access>
	{
		// --- raw records -> FormRaw / MorphemeRaw (what KiwiBuilder holds before build()) ---
		static void fromRaw(const kamd::RawModel& raw, Vector<FormRaw>& forms, Vector<MorphemeRaw>& morphemes)
		{
			forms.resize(raw.nForms());
			morphemes.resize(raw.nMorphs());
			for (size_t i = 0; i < raw.nForms(); ++i)
			{
				forms[i].form = KString{ (const char16_t*)raw.formChars + raw.formPtr[i], (const char16_t*)raw.formChars + raw.formPtr[i + 1] };
				forms[i].candidate.assign(raw.formCand + raw.formCandPtr[i], raw.formCand + raw.formCandPtr[i + 1]);
			}
			for (size_t i = 0; i < raw.nMorphs(); ++i)
			{
				auto& r = raw.morph[i];
				auto& m = morphemes[i];
				m.kform = r.kform; m.tag = (POSTag)r.tag; m.vpPack = r.vpPack; m.senseId = r.senseId;
				m.combineSocket = r.socket; m.combined = r.combined; m.userScore = r.userScore;
				m.lmMorphemeId = r.lmId; m.origMorphemeId = r.origId; m.dialect = (Dialect)r.dialect;
				for (size_t c = 0; c < r.nChunks; ++c)
				{
					m.chunks.emplace_back(raw.chunkIds[r.chunkPtr + c]);
					m.chunkPositions.emplace_back(raw.chunkPos[(r.chunkPtr + c) * 2], raw.chunkPos[(r.chunkPtr + c) * 2 + 1]);
				}
			}
		}

#ifdef KREF_X86
		static void attachChr(Kiwi& k, const uint8_t* blob, size_t size, ArchType arch)
		{
			utils::MemoryOwner mem{ size };
			std::memcpy(mem.get(), blob, size);
			k.nounChrMdl = lm::CoNgramModelBase::create(utils::MemoryObject{ std::move(mem) }, arch, false, true);
		}
		static float unkChrScore(const Kiwi& k, const char16_t* s, size_t n)
		{
			UnkFormScorer sc{ 0.f, 0.f, k.nounChrMdl.get(), 0.f, nullptr };
			return sc(U16StringView{ s, n });
		}
		static float unkChrFreqScore(const Kiwi& k, const char16_t* text, size_t textLen, const char16_t* s, size_t n)
		{
			const SubstringCounter counter{ text, textLen };
			const KiwiConfig& c = k.globalConfig;
			UnkFormScorer sc{ 0.f, 0.f, k.nounChrMdl.get(), 0.f, &counter, c.oovGlobalWeight, c.oovLocalWeight, c.oovGlobalMinFreq, false };
			return sc(U16StringView{ s, n });
		}
		static bool hasChr(const Kiwi& k) { return !!k.nounChrMdl; }
#endif
		static bool& congGlobalWanted() { static bool v = false; return v; }
		static Dialect& enabledDialects() { static Dialect v = Dialect::standard; return v; }      // KiwiBuilder::enabledDialects of the bake below (kref_open_dialects)
		static Kiwi build(const kamd::RawModel& raw, ArchType arch)
		{
			Vector<FormRaw> forms; Vector<MorphemeRaw> morphemes;
			fromRaw(raw, forms, morphemes);
#ifdef KREF_X86
			// a container that carries a CoNgram blob is analysed with it (the reference's default model type): CoNgramModelBase::create with
			// quantized = true exists for the SIMD architectures only (src/ArchAvailable.h:50-78), which is why this is the x86 build's job
			if (raw.cong)
			{
				Kiwi k = build(forms, morphemes, raw.cong, raw.congSize, nullptr, 0, arch, true);
				// the character model of Match::oovChrModel next to a CoNgram model: loaded quantised (KiwiBuilder.cpp:1094-1100)
				if (raw.nounchr) attachChr(k, raw.nounchr, raw.nounchrSize, arch);
				return k;
			}
#endif
			return build(forms, morphemes, raw.knlm, raw.knlmSize, raw.sbg, raw.sbgSize, arch);
		}

		// the on-disk model files of the reference: sj.morph through its own serializer (KiwiBuilder::loadMorphBin / saveMorphBin,
		// src/KiwiBuilder.cpp:923-937: the "KIWI" key, the raw forms, the raw morphemes); sj.knlm and skipbigram.mdl are memory images
		static void writeMorph(std::ostream& os, const Vector<FormRaw>& forms, const Vector<MorphemeRaw>& morphemes)
		{
			serializer::writeMany(os, serializer::toKey("KIWI"), forms, morphemes);
		}
		static void readMorph(std::istream& is, Vector<FormRaw>& forms, Vector<MorphemeRaw>& morphemes)
		{
			serializer::readMany(is, serializer::toKey("KIWI"), forms, morphemes);
		}

		static Kiwi build(const Vector<FormRaw>& forms, const Vector<MorphemeRaw>& morphemes, const uint8_t* knlm, size_t knlmSize, const uint8_t* sbg, size_t sbgSize, ArchType arch, bool cong = false)
		{
			utils::MemoryOwner mem{ knlmSize };
			std::memcpy(mem.get(), knlm, knlmSize);
			std::shared_ptr<lm::ILangModel> langMdl;
#ifdef KREF_X86
			// quantised; local (window 0: ModelType::cong) or, for kref_open_cong_global, with distant tokens (window 7: ModelType::congGlobal, KiwiBuilder.cpp:1018-1031)
			if (cong) langMdl = lm::CoNgramModelBase::create(utils::MemoryObject{ std::move(mem) }, arch, congGlobalWanted(), true);
			else
#endif
			if (sbg)
			{
				// SkipBigram on top of the same Knlm (KiwiBuilder.cpp: ModelType::sbg): SkipBigramModelBase::create (src/SkipBigramModel.cpp:99)
				utils::MemoryOwner smem{ sbgSize };
				std::memcpy(smem.get(), sbg, sbgSize);
				langMdl = lm::SkipBigramModelBase::create(utils::MemoryObject{ std::move(mem) }, utils::MemoryObject{ std::move(smem) }, arch);
			}
			else langMdl = lm::KnLangModelBase::create(utils::MemoryObject{ std::move(mem) }, arch);

			// --- from here on: KiwiBuilder::build() with typos.empty() and no rule-combined morphemes ---
			Kiwi ret{ arch, langMdl, false, false, false };
			ret.forms.reserve(forms.size() + 1);
			ret.morphemes.reserve(morphemes.size());
			ret.globalConfig.integrateAllomorph = true;

			for (auto& f : forms)
			{
				ret.forms.emplace_back(bake(f, ret.morphemes.data(), zCodaAppendableR(f.form, f.candidate, morphemes), zSiotAppendableR(f.form, f.candidate, morphemes)));
			}
			Vector<size_t> newFormIdMapper(ret.forms.size());
			std::iota(newFormIdMapper.begin(), newFormIdMapper.begin() + defaultFormSize, 0);
			utils::sortWriteInvIdx(ret.forms.begin() + defaultFormSize, ret.forms.end(), newFormIdMapper.begin() + defaultFormSize, defaultFormSize);
			ret.forms.emplace_back();

			uint8_t formHash = 0;
			for (size_t i = 1; i < ret.forms.size(); ++i)
			{
				if (!ComparatorIgnoringSpace::equal(ret.forms[i].form, ret.forms[i - 1].form)) ++formHash;
				ret.forms[i].formHash = formHash;
			}
			for (auto& m : morphemes)
			{
				ret.morphemes.emplace_back(bake(m, ret.morphemes.data(), ret.forms.data(), newFormIdMapper));
			}

			utils::ContinuousTrie<KTrie> formTrie{ defaultFormSize + 1 };
			for (size_t i = 0; i < defaultFormSize; ++i) formTrie[i + 1].val = &ret.forms[i];

			Vector<const Form*> sortedForms;
			for (size_t i = defaultFormSize; i < ret.forms.size() - 1; ++i)
			{
				auto& f = ret.forms[i];
				if (f.candidate.empty()) continue;
				if (f.candidate[0]->vowel != CondVowel::none)
					f.vowel = std::accumulate(f.candidate.begin(), f.candidate.end(), f.candidate[0]->vowel, reduceVowelR);
				if (f.candidate[0]->polar != CondPolarity::none)
					f.polar = std::accumulate(f.candidate.begin(), f.candidate.end(), f.candidate[0]->polar, reducePolarR);
				f.hasJClass = std::any_of(f.candidate.begin(), f.candidate.end(), [&](const Morpheme* m)
				{
					return isJClass(m->tag) || m->tag == POSTag::ec || m->tag == POSTag::ef;
				});
				f.hasAnyFullMorphemes = std::any_of(f.candidate.begin(), f.candidate.end(), [&](const Morpheme* m)
				{
					const auto tag = clearIrregular(m->tag);
					return m->dialect == Dialect::standard && tag != POSTag::unknown && tag != POSTag::pa && tag != POSTag::pv;
				});
				f.dialect = std::accumulate(f.candidate.begin(), f.candidate.end(), f.candidate[0]->dialect, reduceDialectR);
				if (f.dialect != Dialect::standard && !(enabledDialects() & f.dialect)) continue;      // (KiwiBuilder.cpp:2500-2504)
				sortedForms.emplace_back(&f);
			}
			std::sort(sortedForms.begin(), sortedForms.end(), [](const Form* a, const Form* b)
			{
				return ComparatorIgnoringSpace::less(a->form, b->form);
			});
			size_t estimated = 0;
			for (auto f : sortedForms) estimated += f->form.size();
			formTrie.reserveMore(estimated);
			decltype(formTrie)::CacheStore<KString> cache;
			for (auto f : sortedForms) formTrie.buildWithCaching(removeSpace(f->form), f, cache);
			ret.formTrie = utils::freezeTrie(std::move(formTrie), arch);

			// KiwiBuilder::getSpecialMorphs (KiwiBuilder.cpp:2642-2662)
			for (size_t i = 0; i < morphemes.size(); ++i)
			{
				auto& m = morphemes[i];
				auto& fs = forms[m.kform].form;
				size_t base;
				if (fs == u"'") base = 0; else if (fs == u"\"") base = 3; else continue;
				if (m.tag == POSTag::sso) ret.specialMorphIds[base + 0] = i;
				else if (m.tag == POSTag::ssc) ret.specialMorphIds[base + 1] = i;
				else if (m.tag == POSTag::ss) ret.specialMorphIds[base + 2] = i;
			}
			return ret;
		}

		static const Vector<Form>& forms(const Kiwi& k) { return k.forms; }
		static const Vector<Morpheme>& morphemes(const Kiwi& k) { return k.morphemes; }
		static KiwiConfig& config(Kiwi& k) { return k.globalConfig; }
		static const lm::ILangModel* lm(const Kiwi& k) { return k.langMdl.get(); }
		static const std::array<size_t, 6>& specialMorphs(const Kiwi& k) { return k.specialMorphIds; }

		static size_t split(const Kiwi& k, Vector<KGraphNode>& nodes, U16StringView str, size_t startOffset, Match m, const KiwiConfig& cfg,
			const PreparedTypoTransformer* typo = nullptr, float typoThreshold = 2.5f, Dialect allowedDialect = Dialect::standard)
		{
			const PretokenizedSpanGroup::Span* pf = nullptr;
			return (*reinterpret_cast<FnSplitByTrie>(k.dfSplitByTrie))(nodes, k.forms.data(), k.typoPtrs.data(), k.formTrie, str, startOffset,
				m, allowedDialect, cfg.maxUnkFormSize, cfg.maxUnkFormSizeFollowedByJClass, cfg.spaceTolerance,
				typo, typoThreshold, k.continualTypoCost, k.lengtheningTypoCost, pf, pf);
		}
	};
}

using Acc = kiwi::BestPathFinder<kamd_ref::Access>;

struct TypoHandle { kiwi::TypoTransformer tt; std::unique_ptr<kiwi::PreparedTypoTransformer> ptt; };

struct RefHandle
{
	kamd::Container file;
	kamd::RawModel raw;
	kiwi::Kiwi kw;
	std::unordered_set<const kiwi::Morpheme*> blocklist;      // AnalyzeOption::blocklist of the following analyses (kref_blocklist_*)
};

namespace
{
	struct Writer
	{
		uint8_t* p; uint8_t* end; bool overflow = false; size_t need = 0;
		template<class T> void put(T v)
		{
			need += sizeof(T);
			if (p + sizeof(T) > end) { overflow = true; return; }
			std::memcpy(p, &v, sizeof(T)); p += sizeof(T);
		}
		void putStr(const std::u16string& s)
		{
			put<uint32_t>((uint32_t)s.size());
			for (auto c : s) put<uint16_t>(c);
		}
	};

	kiwi::ArchType toArch(int a)
	{
		switch (a)
		{
		case 1: return kiwi::ArchType::balanced;
		case 2: return kiwi::ArchType::sse2;
#ifdef KREF_X86
		case 3: return kiwi::ArchType::sse4_1;
		case 4: return kiwi::ArchType::avx2;
		case 5: return kiwi::ArchType::avx512bw;
		case 6: return kiwi::ArchType::avx512vnni;
#endif
		default: return kiwi::ArchType::none;
		}
	}
}


extern "C"
{
	void* kref_open(const char* rawModelPath, int arch)
	{
		try
		{
			auto h = std::make_unique<RefHandle>();
			h->file.load(rawModelPath);
			h->raw.bind(h->file);
			h->kw = Acc::build(h->raw, toArch(arch));
			return h.release();
		}
		catch (const std::exception& e)
		{
			fprintf(stderr, "kref_open: %s\n", e.what());
			return nullptr;
		}
	}

	// the same with KiwiBuilder's enabledDialects (kiwi_init's last argument): forms whose candidates are all of disabled dialects stay out of the trie
	void* kref_open_dialects(const char* rawModelPath, int arch, int enabledDialects)
	{
		Acc::enabledDialects() = (kiwi::Dialect)enabledDialects;
		void* h = kref_open(rawModelPath, arch);
		Acc::enabledDialects() = kiwi::Dialect::standard;
		return h;
	}

#ifdef KREF_X86
}
// (instantiated by src/archImpl/sse4_1.cpp with its own ISA flags: not to be instantiated again by this translation unit)
extern template float kiwi::lm::logSumExp<kiwi::ArchType::sse4_1>(const float*, size_t);
extern template void kiwi::lm::logSoftmax<kiwi::ArchType::sse4_1>(float*, size_t);
extern template void kiwi::lm::logSumExpTransposed<kiwi::ArchType::sse4_1>(float*, size_t, size_t, size_t);
extern template void kiwi::lm::logSoftmaxTransposed<kiwi::ArchType::sse4_1>(float*, size_t, size_t, size_t);
extern "C"
{
	// lm::logSoftmax / logSumExp / logSoftmaxTransposed / logSumExpTransposed of the SSE4.1 build (src/MathFunc.hpp, instantiated by src/archImpl/sse4_1.cpp)
	// over 8 terms; the transposed forms run on lane 0 of a 32-wide batch like progressMatrix's score buffer.  which as korc_congg_math.
	float kref_congg_math(int which, float* w8)
	{
		using kiwi::ArchType;
		if (which == 0) { kiwi::lm::logSoftmax<ArchType::sse4_1>(w8, 8); return 0; }
		if (which == 1) return kiwi::lm::logSumExp<ArchType::sse4_1>(w8, 8);
		float buf[32 * 8] = { 0 };
		for (int k = 0; k < 8; ++k) for (int l = 0; l < 4; ++l) buf[k * 32 + l] = w8[k];
		if (which == 2) { kiwi::lm::logSoftmaxTransposed<ArchType::sse4_1>(buf, 8, 1, 32); for (int k = 0; k < 8; ++k) w8[k] = buf[k * 32]; return 0; }
		kiwi::lm::logSumExpTransposed<ArchType::sse4_1>(buf, 8, 1, 32);
		return buf[0];
	}

	// the same with the container's CoNgram blob loaded as the GLOBAL model (CoNgramModelBase::create(useDistantTokens = true): ModelType::congGlobal)
	void* kref_open_cong_global(const char* rawModelPath, int arch)
	{
		Acc::congGlobalWanted() = true;
		void* h = kref_open(rawModelPath, arch);
		Acc::congGlobalWanted() = false;
		return h;
	}
#endif

	// The synthetic model written as the reference's own model FILES: sj.morph by the reference's serializer, sj.knlm / skipbigram.mdl as the
	// memory images they are.  Returns 0 on success.
	int kref_write_model_dir(const char* rawModelPath, const char* dir)
	{
		try
		{
			kamd::Container file; file.load(rawModelPath);
			kamd::RawModel raw; raw.bind(file);
			kiwi::Vector<kiwi::FormRaw> forms; kiwi::Vector<kiwi::MorphemeRaw> morphemes;
			Acc::fromRaw(raw, forms, morphemes);
			const std::string d = dir;
			{ std::ofstream os{ d + "/sj.morph", std::ios::binary }; Acc::writeMorph(os, forms, morphemes); if (!os) return -2; }
			if (raw.knlm) { std::ofstream os{ d + "/sj.knlm", std::ios::binary }; os.write((const char*)raw.knlm, (std::streamsize)raw.knlmSize); if (!os) return -2; }
			if (raw.sbg) { std::ofstream os{ d + "/skipbigram.mdl", std::ios::binary }; os.write((const char*)raw.sbg, (std::streamsize)raw.sbgSize); if (!os) return -2; }
			if (raw.cong) { std::ofstream os{ d + "/cong.mdl", std::ios::binary }; os.write((const char*)raw.cong, (std::streamsize)raw.congSize); if (!os) return -2; }
			if (raw.nounchr) { std::ofstream os{ d + "/nounchr.mdl", std::ios::binary }; os.write((const char*)raw.nounchr, (std::streamsize)raw.nounchrSize); if (!os) return -2; }
			return 0;
		}
		catch (const std::exception& e) { fprintf(stderr, "kref_write_model_dir: %s\n", e.what()); return -1; }
	}

	// The reference loading those files: sj.morph through its serializer, the language model through KnLangModelBase / SkipBigramModelBase::create;
	// useSbg: 1 = skipbigram.mdl as well (ModelType::sbg), 2 = cong.mdl (ModelType::cong), else Knlm only (KiwiBuilder.cpp:939-961)
	void* kref_open_dir(const char* dir, int arch, int useSbg)
	{
		try
		{
			const std::string d = dir;
			auto slurp = [](const std::string& p) { std::ifstream is{ p, std::ios::binary }; if (!is) throw std::runtime_error{ "cannot open " + p }; return std::vector<uint8_t>{ std::istreambuf_iterator<char>{ is }, std::istreambuf_iterator<char>{} }; };
			kiwi::Vector<kiwi::FormRaw> forms; kiwi::Vector<kiwi::MorphemeRaw> morphemes;
			{ std::ifstream is{ d + "/sj.morph", std::ios::binary }; if (!is) throw std::runtime_error{ "cannot open sj.morph" }; Acc::readMorph(is, forms, morphemes); }
			auto h = std::make_unique<RefHandle>();
			if (useSbg == 2)      // ModelType::cong: cong.mdl alone (KiwiBuilder.cpp:1018-1031); the quantised path needs a SIMD architecture
			{
				const auto cong = slurp(d + "/cong.mdl");
				h->kw = Acc::build(forms, morphemes, cong.data(), cong.size(), nullptr, 0, toArch(arch), true);
#ifdef KREF_X86
				{
					// the optional character model (KiwiBuilder.cpp:1094-1100)
					std::ifstream is{ d + "/nounchr.mdl", std::ios::binary };
					if (is) { const std::vector<uint8_t> chr{ std::istreambuf_iterator<char>{ is }, std::istreambuf_iterator<char>{} }; Acc::attachChr(h->kw, chr.data(), chr.size(), toArch(arch)); }
				}
#endif
				return h.release();
			}
			const auto knlm = slurp(d + "/sj.knlm");
			std::vector<uint8_t> sbg;
			if (useSbg) sbg = slurp(d + "/skipbigram.mdl");
			h->kw = Acc::build(forms, morphemes, knlm.data(), knlm.size(), sbg.empty() ? nullptr : sbg.data(), sbg.size(), toArch(arch));
			return h.release();
		}
		catch (const std::exception& e)
		{
			fprintf(stderr, "kref_open_dir: %s\n", e.what());
			return nullptr;
		}
	}

#ifdef KREF_X86
	// The REAL builder, unmodified (src/KiwiBuilder.cpp is a translation unit of the x86 library): KiwiBuilder{ dir, ... } loads the directory as
	// Kiwi ships it -- sj.morph, the language model, extract.mdl, combiningRule.txt and, per `options` (BuildOption bits), default.dict / typo.dict /
	// multi.dict -- and build() bakes it (combined morphemes, dictionary entries).  Every other entry point of this bridge bakes with its own
	// restatement of build() (Acc::build above, written while that file did not compile): this one is what pins the restatement.
	// extract.mdl of the reference checkout is a git-LFS pointer; kref_write_empty_extract writes one with empty tables through the reference's serializer
	// (the word detector it feeds is not on the analysis path).
	int kref_write_empty_extract(const char* dir)
	{
		try
		{
			std::map<std::pair<kiwi::POSTag, bool>, std::map<char16_t, float>> posScore;
			std::map<std::u16string, float> nounTailScore;
			std::ofstream os{ std::string{ dir } + "/extract.mdl", std::ios::binary };
			kiwi::serializer::writeMany(os, posScore, nounTailScore);
			return os ? 0 : -2;
		}
		catch (const std::exception& e) { fprintf(stderr, "kref_write_empty_extract: %s\n", e.what()); return -1; }
	}
	// (the export of the builder's tables after buildCombinedMorphemes is tools/export_built.cpp -- a program of its own, the one a Kiwi maintainer would build)
	void* kref_open_built(const char* dir, int modelType, int options)
	{
		try
		{
			auto h = std::make_unique<RefHandle>();
			kiwi::KiwiBuilder kb{ std::string{ dir }, 1, (kiwi::BuildOption)options, (kiwi::ModelType)modelType };
			h->kw = kb.build();
			return h.release();
		}
		catch (const std::exception& e)
		{
			fprintf(stderr, "kref_open_built: %s\n", e.what());
			return nullptr;
		}
	}
#endif

	void kref_close(void* h) { delete (RefHandle*)h; }

	// 13 floats/ints mirroring KiwiConfig (include/kiwi/Kiwi.h:150-167); negative index = no change
	void kref_set_config(void* hp, float cutOff, float spacePenalty, float typoCostWeight, uint32_t maxUnk, uint32_t maxUnkJ, uint32_t spaceTol, int integrateAllomorph)
	{
		auto& c = Acc::config(((RefHandle*)hp)->kw);
		c.cutOffThreshold = cutOff; c.spacePenalty = spacePenalty; c.typoCostWeight = typoCostWeight;
		c.maxUnkFormSize = maxUnk; c.maxUnkFormSizeFollowedByJClass = maxUnkJ; c.spaceTolerance = spaceTol;
		c.integrateAllomorph = !!integrateAllomorph;
	}

	// Serialises the baked dictionary so that the product's baker can be compared field by field.
	// forms: {u32 nChars, u16 chars[], u32 numSpaces, u8 vowel, u8 polar, u8 formHash, u8 flags, u16 dialect, u32 nCand, u32 cand[]}
	// morphs: {u32 kformId, u8 tag, u8 vowel, u8 polar, u8 complex, u8 saisiot, u8 senseId, u8 socket, i32 combined, f32 userScore,
	//          u32 lmId, u32 origId, u16 dialect, u32 nChunks, {u32 id, u8 b, u8 e}[]}
	size_t kref_dump_dict(void* hp, uint8_t* out, size_t cap)
	{
		auto& kw = ((RefHandle*)hp)->kw;
		auto& forms = Acc::forms(kw);
		auto& morphs = Acc::morphemes(kw);
		Writer w{ out, out + cap };
		w.put<uint32_t>((uint32_t)forms.size());
		w.put<uint32_t>((uint32_t)morphs.size());
		for (auto& f : forms)
		{
			w.putStr(f.form);
			w.put<uint32_t>(f.numSpaces);
			w.put<uint8_t>((uint8_t)f.vowel); w.put<uint8_t>((uint8_t)f.polar); w.put<uint8_t>(f.formHash);
			w.put<uint8_t>(f.zCodaAppendable | (f.zSiotAppendable << 1) | (f.hasJClass << 2) | (f.hasAnyFullMorphemes << 3));
			w.put<uint16_t>((uint16_t)f.dialect);
			w.put<uint32_t>((uint32_t)f.candidate.size());
			for (auto c : f.candidate) w.put<uint32_t>((uint32_t)(c - morphs.data()));
		}
		for (auto& m : morphs)
		{
			uint32_t kf = 0;
			for (; kf < forms.size(); ++kf) if (&forms[kf].form == m.kform) break; // linear; dumps are for small models
			w.put<uint32_t>(kf);
			w.put<uint8_t>((uint8_t)m.tag); w.put<uint8_t>((uint8_t)m.vowel); w.put<uint8_t>((uint8_t)m.polar);
			w.put<uint8_t>(m.complex); w.put<uint8_t>(m.saisiot); w.put<uint8_t>(m.senseId); w.put<uint8_t>(m.combineSocket);
			w.put<int32_t>(m.combined); w.put<float>(m.userScore);
			w.put<uint32_t>(m.lmMorphemeId); w.put<uint32_t>(m.origMorphemeId); w.put<uint16_t>((uint16_t)m.dialect);
			w.put<uint32_t>((uint32_t)m.chunks.size());
			for (size_t c = 0; c < m.chunks.size(); ++c)
			{
				w.put<uint32_t>((uint32_t)(m.chunks[c] - morphs.data()));
				w.put<uint8_t>(m.chunks.getSecond(c).first); w.put<uint8_t>(m.chunks.getSecond(c).second);
			}
		}
		for (auto v : Acc::specialMorphs(kw)) w.put<uint32_t>((uint32_t)v);
		return w.need;
	}

	// One Knlm step through the reference (src/Knlm.cpp:44-130). node in/out, returns ll.
	float kref_lm_progress(void* hp, int32_t* node, uint32_t wid)
	{
		auto* lm = dynamic_cast<const kiwi::lm::KnLangModelBase*>(Acc::lm(((RefHandle*)hp)->kw));
		ptrdiff_t n = *node;
		float ll = lm->progress(n, wid);
		*node = (int32_t)n;
		return ll;
	}

#ifdef KREF_X86
	// One CoNgram state step through the reference (CoNgramModelBase::progressOneStep, src/CoNgramModel.cpp:922-927 -> progress()): node and context id in / out
	float kref_cong_next(void* hp, int32_t* node, uint32_t* ctx, uint32_t wid)
	{
		auto* lm = dynamic_cast<const kiwi::lm::CoNgramModelBase*>(Acc::lm(((RefHandle*)hp)->kw));
		if (!lm) return 0.f / 0.f;
		return lm->progressOneStep(*node, *ctx, wid);
	}
#endif

#ifdef KREF_X86
	// UnkFormScorer::chrBasedScore (src/UnkFormScorer.cpp:53-66) of a normalised string through the reference's own scorer, bias 0
	float kref_unk_chr_score(void* hp, const uint16_t* s, uint32_t len)
	{
		auto& kw = ((RefHandle*)hp)->kw;
		if (!Acc::hasChr(kw)) return 0.f / 0.f;
		return Acc::unkChrScore(kw, (const char16_t*)s, len);
	}
	void kref_set_oov_chr_bias(void* hp, float bias) { Acc::config(((RefHandle*)hp)->kw).oovChrBias = bias; }
	void kref_set_oov_freq_params(void* hp, float globalWeight, float localWeight, float globalMinFreq)
	{
		auto& c = Acc::config(((RefHandle*)hp)->kw);
		c.oovGlobalWeight = globalWeight; c.oovLocalWeight = localWeight; c.oovGlobalMinFreq = globalMinFreq;
	}
	// UnkFormScorer::chrFreqBasedScore (src/UnkFormScorer.cpp:68-116) of a normalised string through the reference's own scorer and its own SubstringCounter
	// over an already filtered text, bias 0
	float kref_unk_chr_freq_score(void* hp, const uint16_t* text, uint32_t textLen, const uint16_t* s, uint32_t len)
	{
		auto& kw = ((RefHandle*)hp)->kw;
		if (!Acc::hasChr(kw)) return 0.f / 0.f;
		return Acc::unkChrFreqScore(kw, (const char16_t*)text, textLen, (const char16_t*)s, len);
	}
#endif

	// One SkipBigram state step through the reference (SbgState::nextImpl, src/SkipBigramModel.hpp:169-182); 16-bit vocabulary only.
	float kref_sbg_next(void* hp, int32_t* node, uint32_t* pos, uint32_t* hist8, uint32_t wid)
	{
		using Model = kiwi::lm::SkipBigramModel<kiwi::ArchType::none, uint16_t, 8>;
		auto* lm = dynamic_cast<const Model*>(Acc::lm(((RefHandle*)hp)->kw));
		if (!lm) return NAN;
		typename Model::LmStateType st{ lm };
		st.knlm.node = *node; st.historyPos = *pos; for (int i = 0; i < 8; ++i) st.history[i] = (uint16_t)hist8[i];
		const float ll = st.next(lm, (uint16_t)wid);
		*node = (int32_t)st.knlm.node; *pos = (uint32_t)st.historyPos; for (int i = 0; i < 8; ++i) hist8[i] = st.history[i];
		return ll;
	}

	// Lattice of every chunk of `text` (already raw UTF-16; normalised inside exactly as Kiwi::analyze does,
	// src/Kiwi.cpp:1028-1030,1095-1117).  Per chunk: {u32 nNodes, u32 splitEnd, nodes{u32 start,end,prev,sibling,
	// i32 formId, u32 uformLen, u32 uformOff(in normalised str), u32 spaceErrors, f32 typoCost}[]} ; leading u32 nChunks.
	size_t kref_split_typo(void* hp, void* typoHp, float typoThreshold, int allowedDialect, const uint16_t* text, uint32_t len, uint64_t match, uint8_t* out, size_t cap);
	size_t kref_split(void* hp, const uint16_t* text, uint32_t len, uint64_t match, uint8_t* out, size_t cap)
	{
		return kref_split_typo(hp, nullptr, 2.5f, 0, text, len, match, out, cap);
	}
	// ... with a prepared typo transformer (kref_typo_*): the lattice over the typo graph of every chunk (KTrie.cpp:873-895, 998-1464)
	size_t kref_split_typo(void* hp, void* typoHp, float typoThreshold, int allowedDialect, const uint16_t* text, uint32_t len, uint64_t match, uint8_t* out, size_t cap)
	{
		using namespace kiwi;
		auto& kw = ((RefHandle*)hp)->kw;
		const PreparedTypoTransformer* typo = typoHp ? ((TypoHandle*)typoHp)->ptt.get() : nullptr;
		KString norm; Vector<uint32_t> pos;
		normalizeHangulWithPosition((const char16_t*)text, (const char16_t*)text + len, std::back_inserter(norm), std::back_inserter(pos));
		if (!!((Match)match & Match::normalizeCoda)) normalizeCoda(norm.begin(), norm.end());
		Writer w{ out, out + cap };
		uint8_t* countPos = w.p; w.put<uint32_t>(0);
		uint32_t nChunks = 0;
		size_t splitEnd = 0;
		Vector<KGraphNode> nodes;
		auto& forms = Acc::forms(kw);
		while (splitEnd < norm.size())
		{
			nodes.clear();
			splitEnd = Acc::split(kw, nodes, U16StringView{ norm.data() + splitEnd, norm.size() - splitEnd }, splitEnd, (Match)match, Acc::config(kw), typo, typoThreshold, (Dialect)allowedDialect);
			++nChunks;
			w.put<uint32_t>((uint32_t)nodes.size());
			w.put<uint32_t>((uint32_t)splitEnd);
			for (auto& n : nodes)
			{
				w.put<uint32_t>(n.startPos); w.put<uint32_t>(n.endPos); w.put<uint32_t>(n.prev); w.put<uint32_t>(n.sibling);
				w.put<int32_t>(n.form ? (int32_t)(n.form - forms.data()) : -1);
				w.put<uint32_t>((uint32_t)n.uform.size());
				w.put<uint32_t>(n.uform.empty() ? 0 : (uint32_t)(n.uform.data() - norm.data()));
				w.put<uint32_t>(n.spaceErrors);
				w.put<float>(n.typoCost);
			}
		}
		if (!w.overflow || countPos + 4 <= out + cap) std::memcpy(countPos, &nChunks, 4);
		return w.need;
	}

	static void writeResults(Writer& w, const std::vector<kiwi::TokenResult>& res, const kiwi::Kiwi& kw)
	{
		auto& morphs = Acc::morphemes(kw);
		w.put<uint32_t>((uint32_t)res.size());
		for (auto& r : res)
		{
			w.put<float>(r.second);
			w.put<uint32_t>((uint32_t)r.first.size());
			for (auto& t : r.first)
			{
				w.putStr(t.str);
				w.put<uint32_t>(t.position); w.put<uint32_t>(t.wordPosition); w.put<uint32_t>(t.sentPosition); w.put<uint32_t>(t.lineNumber);
				w.put<uint16_t>(t.length); w.put<uint8_t>((uint8_t)t.tag); w.put<uint8_t>(t.senseId);
				w.put<float>(t.score); w.put<float>(t.typoCost); w.put<uint32_t>(t.typoFormId); w.put<uint32_t>(t.pairedToken);
				w.put<uint32_t>(t.subSentPosition); w.put<uint16_t>((uint16_t)t.dialect);
				w.put<int32_t>(t.morph ? (int32_t)(t.morph - morphs.data()) : -1);
			}
		}
	}

	// Kiwi::analyze (src/Kiwi.cpp:1014-1158) on one text.  Returns bytes needed.
	size_t kref_analyze_typo(void* hp, void* typoHp, float typoThreshold, int allowedDialect, const uint16_t* text, uint32_t len, uint32_t topN, uint64_t match, int openEnding, uint8_t* out, size_t cap);
	// the blocklist of the following kref_analyze* calls, filled the way kiwi_morphset_add does it (src/capi/kiwi_c.cpp:1779-1794): the reference's own
	// Kiwi::findMorphemes(form, tag) -> set of Morpheme pointers
	void kref_blocklist_clear(void* hp) { ((RefHandle*)hp)->blocklist.clear(); }
	int kref_blocklist_add(void* hp, const uint16_t* form, uint32_t len, int tag)
	{
		auto& h = *(RefHandle*)hp;
		try
		{
			auto found = h.kw.findMorphemes(std::u16string{ (const char16_t*)form, (const char16_t*)form + len }, tag < 0 ? kiwi::POSTag::unknown : (kiwi::POSTag)tag);
			h.blocklist.insert(found.begin(), found.end());
			return (int)found.size();
		}
		catch (const std::exception& e) { fprintf(stderr, "kref_blocklist_add: %s\n", e.what()); return -1; }
	}

	size_t kref_analyze(void* hp, const uint16_t* text, uint32_t len, uint32_t topN, uint64_t match, int openEnding, uint8_t* out, size_t cap)
	{
		return kref_analyze_typo(hp, nullptr, 2.5f, 0, text, len, topN, match, openEnding, out, cap);
	}
	size_t kref_analyze_dialect(void* hp, void* typoHp, float typoThreshold, int allowedDialect, float dialectCost, const uint16_t* text, uint32_t len, uint32_t topN, uint64_t match, int openEnding, uint8_t* out, size_t cap);
	size_t kref_analyze_typo(void* hp, void* typoHp, float typoThreshold, int allowedDialect, const uint16_t* text, uint32_t len, uint32_t topN, uint64_t match, int openEnding, uint8_t* out, size_t cap)
	{
		if (!typoHp && allowedDialect) return kref_analyze_dialect(hp, nullptr, typoThreshold, allowedDialect, 3.f, text, len, topN, match, openEnding, out, cap);
		auto& kw = ((RefHandle*)hp)->kw;
		Writer w{ out, out + cap };
		try
		{
			kiwi::AnalyzeOption opt{ (kiwi::Match)match };
			opt.openEnding = !!openEnding;
			if (typoHp) { opt.typoTransformer = ((TypoHandle*)typoHp)->ptt.get(); opt.typoThreshold = typoThreshold; opt.allowedDialects = (kiwi::Dialect)allowedDialect; }
			if (!((RefHandle*)hp)->blocklist.empty()) opt.blocklist = &((RefHandle*)hp)->blocklist;
			auto res = kw.analyze(std::u16string{ (const char16_t*)text, (const char16_t*)text + len }, topN, opt);
			writeResults(w, res, kw);
		}
		catch (const std::exception& e)
		{
			fprintf(stderr, "kref_analyze: %s\n", e.what());
			return 0;
		}
		return w.need;
	}

	// AnalyzeOption::allowedDialects / dialectCost (include/kiwi/Kiwi.h); without a typo transformer the reference takes its built-in `dialect` set itself
	// (src/Kiwi.cpp:1037-1041)
	size_t kref_analyze_dialect(void* hp, void* typoHp, float typoThreshold, int allowedDialect, float dialectCost, const uint16_t* text, uint32_t len, uint32_t topN, uint64_t match, int openEnding, uint8_t* out, size_t cap)
	{
		auto& kw = ((RefHandle*)hp)->kw;
		Writer w{ out, out + cap };
		try
		{
			kiwi::AnalyzeOption opt{ (kiwi::Match)match };
			opt.openEnding = !!openEnding;
			if (typoHp) { opt.typoTransformer = ((TypoHandle*)typoHp)->ptt.get(); opt.typoThreshold = typoThreshold; }
			opt.allowedDialects = (kiwi::Dialect)allowedDialect; opt.dialectCost = dialectCost;
			if (!((RefHandle*)hp)->blocklist.empty()) opt.blocklist = &((RefHandle*)hp)->blocklist;
			auto res = kw.analyze(std::u16string{ (const char16_t*)text, (const char16_t*)text + len }, topN, opt);
			writeResults(w, res, kw);
		}
		catch (const std::exception& e)
		{
			fprintf(stderr, "kref_analyze_dialect: %s\n", e.what());
			return 0;
		}
		return w.need;
	}

	// Kiwi::analyze with pretokenized spans (src/Kiwi.cpp:785-946, 1043-1051; the argument of kiwi_analyze* that this repo's product still refuses): the
	// reference's own answer, for golden vectors (tools/make_golden_pretokenized.py) and for the restatement to come.  spans: per span {begin, end, nTokens}
	// (UTF-16 offsets into `text`), then per token {formOff, formLen (into `forms`), begin, end (relative to the span), tag, inferRegularity}.
	size_t kref_analyze_pretokenized(void* hp, const uint16_t* text, uint32_t len, uint32_t topN, uint64_t match, const uint32_t* spans, uint32_t nSpans, const uint16_t* forms, uint8_t* out, size_t cap)
	{
		auto& kw = ((RefHandle*)hp)->kw;
		Writer w{ out, out + cap };
		try
		{
			std::vector<kiwi::PretokenizedSpan> pt;
			const uint32_t* p = spans;
			for (uint32_t i = 0; i < nSpans; ++i)
			{
				kiwi::PretokenizedSpan sp{ p[0], p[1] };
				const uint32_t nTok = p[2];
				p += 3;
				for (uint32_t t = 0; t < nTok; ++t, p += 6)
					sp.tokenization.emplace_back(std::u16string{ (const char16_t*)forms + p[0], (const char16_t*)forms + p[0] + p[1] }, p[2], p[3], (kiwi::POSTag)p[4], (uint8_t)p[5]);
				pt.push_back(std::move(sp));
			}
			kiwi::AnalyzeOption opt{ (kiwi::Match)match };
			auto res = kw.analyze(std::u16string{ (const char16_t*)text, (const char16_t*)text + len }, topN, opt, pt);
			writeResults(w, res, kw);
		}
		catch (const std::exception& e)
		{
			fprintf(stderr, "kref_analyze_pretokenized: %s\n", e.what());
			return 0;
		}
		return w.need;
	}

	// CPU baseline: the reference analysing a batch with `threads` worker threads, like
	// Kiwi::analyze(topN, reader, receiver) does with its pool (include/kiwi/Kiwi.h:402-454).
	// texts are concatenated UTF-16 with offsets[n+1]. Returns wall seconds; *tokensOut = total top-1 tokens.
	double kref_analyze_batch_typo(void* hp, void* typoHp, float typoThreshold, const uint16_t* texts, const uint64_t* offsets, uint32_t n, uint32_t topN, uint64_t match, int threads, uint64_t* tokensOut);
	double kref_analyze_batch(void* hp, const uint16_t* texts, const uint64_t* offsets, uint32_t n, uint32_t topN, uint64_t match, int threads, uint64_t* tokensOut)
	{
		return kref_analyze_batch_typo(hp, nullptr, 2.5f, texts, offsets, n, topN, match, threads, tokensOut);
	}
	double kref_analyze_batch_typo(void* hp, void* typoHp, float typoThreshold, const uint16_t* texts, const uint64_t* offsets, uint32_t n, uint32_t topN, uint64_t match, int threads, uint64_t* tokensOut)
	{
		auto& kw = ((RefHandle*)hp)->kw;
		std::atomic<uint32_t> next{ 0 };
		std::atomic<uint64_t> tokens{ 0 };
		auto work = [&]()
		{
			kiwi::AnalyzeOption opt{ (kiwi::Match)match };
			if (typoHp) { opt.typoTransformer = ((TypoHandle*)typoHp)->ptt.get(); opt.typoThreshold = typoThreshold; }
			uint64_t local = 0;
			for (;;)
			{
				uint32_t i = next.fetch_add(1);
				if (i >= n) break;
				auto res = kw.analyze(std::u16string{ (const char16_t*)texts + offsets[i], (const char16_t*)texts + offsets[i + 1] }, topN, opt);
				local += res[0].first.size();
			}
			tokens += local;
		};
		auto t0 = std::chrono::steady_clock::now();
		if (threads <= 1) work();
		else
		{
			std::vector<std::thread> ts;
			for (int t = 0; t < threads; ++t) ts.emplace_back(work);
			for (auto& t : ts) t.join();
		}
		auto t1 = std::chrono::steady_clock::now();
		if (tokensOut) *tokensOut = tokens.load();
		return std::chrono::duration<double>(t1 - t0).count();
	}

	// The same batch, timed soundly (timed_pool.hpp): persistent threads, one untimed warm-up pass, whole passes until >= minSeconds of wall.
	// Returns wall seconds of the timed passes; *passesOut = passes, *tokensOut = top-1 tokens of one pass.
	double kref_analyze_batch_timed(void* hp, void* typoHp, float typoThreshold, const uint16_t* texts, const uint64_t* offsets, uint32_t n, uint32_t topN, uint64_t match, int threads, double minSeconds, uint32_t* passesOut, uint64_t* tokensOut)
	{
		auto& kw = ((RefHandle*)hp)->kw;
		std::atomic<uint64_t> tokens{ 0 };
		kiwi::AnalyzeOption opt{ (kiwi::Match)match };
		if (typoHp) { opt.typoTransformer = ((TypoHandle*)typoHp)->ptt.get(); opt.typoThreshold = typoThreshold; }
		// The reference is built with mimalloc by default (KIWI_USE_MIMALLOC); this build of its translation units allocates through glibc, whose
		// default policy maps and unmaps every large block -- at 256 threads the analyses then queue on the process's address-space lock instead of
		// running (measured on the MI355X host: 6.7 % scaling efficiency).  Keep freed memory in the arenas, as mimalloc would: a fair baseline.
		static const bool tuned = [] { mallopt(M_MMAP_THRESHOLD, 1 << 30); mallopt(M_TRIM_THRESHOLD, 1 << 30); mallopt(M_TOP_PAD, 64 << 20); return true; }();
		(void)tuned;
		uint32_t passes = 0;
		const double sec = timedpool::run(threads, n, minSeconds, &passes, [&](int, uint32_t i)
		{
			auto res = kw.analyze(std::u16string{ (const char16_t*)texts + offsets[i], (const char16_t*)texts + offsets[i + 1] }, topN, opt);
			tokens.fetch_add(res[0].first.size(), std::memory_order_relaxed);
		});
		if (passesOut) *passesOut = passes;
		if (tokensOut) *tokensOut = tokens.load() / (passes + 1);
		return sec;
	}

	// ---- typo graphs: the reference's TypoTransformer / PreparedTypoTransformer, public API only --------
	// A transformer is filled either rule by rule (TypoTransformer::addTypo: normalisation + jamo expansion inside) or by replaying,
	// entry by entry, one of the built-in sets (TypoTransformer::update, which inserts in the iteration order of the source map).
	// the reference's pattern recogniser at one position (src/PatternMatcher.cpp:380): length | tag << 32, 0 = no pattern
	uint64_t kref_match_pattern(uint16_t left, const uint16_t* text, uint32_t len, uint64_t match)
	{
		const auto r = kiwi::matchPattern((char16_t)left, (const char16_t*)text, (const char16_t*)text + len, (kiwi::Match)match);
		return (uint64_t)r.first | ((uint64_t)(uint8_t)r.second << 32);
	}

	void* kref_typo_new(float continualCost, float lengtheningCost)
	{
		auto* h = new TypoHandle;
		h->tt.setContinualTypoCost(continualCost); h->tt.setLengtheningTypoCost(lengtheningCost);
		return h;
	}
	void kref_typo_close(void* hp) { delete (TypoHandle*)hp; }
	int kref_typo_add(void* hp, const uint16_t* orig, uint32_t nOrig, const uint16_t* err, uint32_t nErr, float cost, int cond, int dialect)
	{
		try { ((TypoHandle*)hp)->tt.addTypo(std::u16string{ (const char16_t*)orig, nOrig }, std::u16string{ (const char16_t*)err, nErr }, cost, (kiwi::CondVowel)cond, (kiwi::Dialect)dialect); return 0; }
		catch (const std::exception&) { return -1; }
	}
	// the entries of a built-in set in the iteration order of its map: {u32 n; per entry: u32 nOrig, u16[], u32 nErr, u16[], f32 cost, u8 cond, u16 dialect}; f32 continual, f32 lengthening
	size_t kref_typo_default_entries(int set, uint8_t* out, size_t cap)
	{
		const auto& tt = kiwi::getDefaultTypoSet((kiwi::DefaultTypoSet)set);
		Writer w{ out, out + cap };
		w.put<uint32_t>((uint32_t)tt.getTypos().size());
		for (auto& p : tt.getTypos())
		{
			const auto& o = std::get<0>(p.first); const auto& e = std::get<1>(p.first);
			w.put<uint32_t>((uint32_t)o.size()); for (auto c : o) w.put<uint16_t>((uint16_t)c);
			w.put<uint32_t>((uint32_t)e.size()); for (auto c : e) w.put<uint16_t>((uint16_t)c);
			w.put<float>(p.second); w.put<uint8_t>((uint8_t)std::get<2>(p.first)); w.put<uint16_t>((uint16_t)std::get<3>(p.first));
		}
		w.put<float>(tt.getContinualTypoCost()); w.put<float>(tt.getLengtheningTypoCost());
		return w.need;
	}
	// a copy of a built-in set itself (what kiwi_typo_get_default hands to kiwi_typo_prepare)
	void* kref_typo_from_default(int set) { auto* h = new TypoHandle; h->tt = kiwi::getDefaultTypoSet((kiwi::DefaultTypoSet)set); return h; }
	// update(built-in set): what a client does with `TypoTransformer tt; tt |= getDefaultTypoSet(set)`
	void kref_typo_update_default(void* hp, int set) { ((TypoHandle*)hp)->tt.update(kiwi::getDefaultTypoSet((kiwi::DefaultTypoSet)set)); }
	void kref_typo_prepare(void* hp, int inverse) { auto* h = (TypoHandle*)hp; h->ptt.reset(new kiwi::PreparedTypoTransformer{ h->tt, inverse != 0 }); }
	// generateGraph over the normalised text (normalizeHangul; normalizeCoda when `normCoda`).
	// {u32 normLen, u16[]; u32 nNodes; per node: u32 formLen, u16[], u32 endPos, f32 typoCost, u32 prevOffset, u32 siblingOffset, u8 continualTypoIdx, u16 dialect}; u32 maxContinualTypoIdx
	size_t kref_typo_graph(void* hp, const uint16_t* text, uint32_t len, int allowedDialect, int normCoda, uint8_t* out, size_t cap)
	{
		using namespace kiwi;
		auto* h = (TypoHandle*)hp;
		KString norm = normalizeHangul(std::u16string{ (const char16_t*)text, len });
		if (normCoda) normalizeCoda(norm.begin(), norm.end());
		std::vector<TypoGraphNode> g;
		size_t maxIdx = 0;
		h->ptt->generateGraph(U16StringView{ norm.data(), norm.size() }, g, (Dialect)allowedDialect, nullptr, nullptr, &maxIdx);
		Writer w{ out, out + cap };
		w.put<uint32_t>((uint32_t)norm.size()); for (auto c : norm) w.put<uint16_t>((uint16_t)c);
		w.put<uint32_t>((uint32_t)g.size());
		for (auto& n : g)
		{
			w.put<uint32_t>((uint32_t)n.form.size()); for (auto c : n.form) w.put<uint16_t>((uint16_t)c);
			w.put<uint32_t>(n.endPos); w.put<float>(n.typoCost); w.put<uint32_t>(n.prevOffset); w.put<uint32_t>(n.siblingOffset);
			w.put<uint8_t>(n.continualTypoIdx); w.put<uint16_t>((uint16_t)n.dialect);
		}
		w.put<uint32_t>((uint32_t)maxIdx);
		return w.need;
	}
}
