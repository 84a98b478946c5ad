// export_built: the one translation unit a Kiwi maintainer adds next to the reference to hand a BUILT model to this library (SURVEY.md
// section 7 step 2).  kiwi_init reads the reference's binary model files itself, but not the inputs of KiwiBuilder's build step -- combiningRule.txt and the
// *.dict files of a model directory as Kiwi ships it (src/KiwiBuilder.cpp:1035-1092, 2385-2464, src/Combiner.cpp).  This program lets the reference do that step:
//
//     KiwiBuilder{ dir, 1, options, modelType }                  loads sj.morph, the dictionaries the options ask for, the language model
//     KiwiBuilder::buildCombinedMorphemes(...)                   the rule-combined morphemes, exactly as build() makes them (:2397)
//
// and writes the builder's tables AFTER that step -- forms, candidates, morphemes with chunks / combined links -- plus the directory's language-model files
// into one raw-model container (kiwi_amd/csrc/container.hpp, kind KAMDRAW1), which kiwi_init / kamd_open load like any other model; the bake (form order,
// trie, allomorph groups: the rest of build()) is this library's own, pinned byte for byte against KiwiBuilder::build() by tests/test_built_model.py.
//
//     usage:  export_built <model directory> <out.raw> [model type: 1 largest, 2 knlm, 3 sbg, 4 cong, 5 congGlobal; default 2] [BuildOption bits; default 1|2|4|8]
//
// Build (a maintainer): c++ -std=c++17 -O2 -I<kiwi>/include -I<kiwi>/src -I<this repo> tools/export_built.cpp -lkiwi -o export_built
// Build (this repository's tests, against the reference objects of oracle/_ref): tests/test_built_model.py::test_the_exporter_tool.
// The builder keeps its tables private and befriends nobody; an explicit template instantiation may name private members, which is how they are reached
// without touching the reference's sources.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <stdexcept>
#include <string>
#include <vector>

#include <kiwi/Kiwi.h>
#include <kiwi/Form.h>
#include <kiwi/Knlm.h>
#include <kiwi/CoNgramModel.h>

#include "kiwi_amd/csrc/container.hpp"
#include "kiwi_amd/csrc/raw_model.hpp"

namespace
{
	template<class Tag, typename Tag::type M> struct Reach { friend typename Tag::type reach(Tag) { return M; } };
	struct KbForms { using type = kiwi::Vector<kiwi::FormRaw> kiwi::KiwiBuilder::*; friend type reach(KbForms); };
	struct KbMorphs { using type = kiwi::Vector<kiwi::MorphemeRaw> kiwi::KiwiBuilder::*; friend type reach(KbMorphs); };
	struct KbCombine
	{
		using type = void (kiwi::KiwiBuilder::*)(kiwi::Vector<kiwi::FormRaw>&, kiwi::UnorderedMap<kiwi::KString, size_t>&, kiwi::Vector<kiwi::MorphemeRaw>&,
			kiwi::UnorderedMap<size_t, kiwi::Vector<uint32_t>>&, kiwi::Map<int, int>*) const;
		friend type reach(KbCombine);
	};
	template struct Reach<KbForms, &kiwi::KiwiBuilder::forms>;
	template struct Reach<KbMorphs, &kiwi::KiwiBuilder::morphemes>;
	template struct Reach<KbCombine, &kiwi::KiwiBuilder::buildCombinedMorphemes>;

	std::vector<uint8_t> slurp(const std::string& path)
	{
		std::ifstream is{ path, std::ios::binary };
		if (!is) return {};
		return std::vector<uint8_t>{ std::istreambuf_iterator<char>{ is }, std::istreambuf_iterator<char>{} };
	}
}

int main(int argc, char** argv)
{
	if (argc < 3) { std::fprintf(stderr, "usage: %s <model directory> <out.raw> [model type] [build options]\n", argv[0]); return 2; }
	try
	{
		using namespace kiwi;
		const std::string dir = argv[1];
		const int modelType = argc > 3 ? std::atoi(argv[3]) : 2, options = argc > 4 ? std::atoi(argv[4]) : (1 | 2 | 4 | 8);
		KiwiBuilder kb{ dir, 1, (BuildOption)options, (ModelType)modelType };
		const auto& forms = kb.*reach(KbForms{});
		const auto& morphemes = kb.*reach(KbMorphs{});
		Vector<FormRaw> cForms; Vector<MorphemeRaw> cMorphs;
		UnorderedMap<KString, size_t> newFormMap; UnorderedMap<size_t, Vector<uint32_t>> newFormCands;
		(kb.*reach(KbCombine{}))(cForms, newFormMap, cMorphs, newFormCands, nullptr);

		// forms with their candidate lists (a rule product may add candidates to an existing form: newFormCands), morphemes with their chunks
		std::vector<uint32_t> formPtr{ 0 }, candPtr{ 0 }, cands, chunkIds;
		std::vector<uint16_t> chars; std::vector<uint8_t> chunkPos;
		const size_t nF = forms.size() + cForms.size(), nM = morphemes.size() + cMorphs.size();
		for (size_t i = 0; i < nF; ++i)
		{
			const FormRaw& f = i < forms.size() ? forms[i] : cForms[i - forms.size()];
			chars.insert(chars.end(), f.form.begin(), f.form.end());
			formPtr.push_back((uint32_t)chars.size());
			cands.insert(cands.end(), f.candidate.begin(), f.candidate.end());
			auto it = newFormCands.find(i);
			if (it != newFormCands.end()) cands.insert(cands.end(), it->second.begin(), it->second.end());
			candPtr.push_back((uint32_t)cands.size());
		}
		std::vector<kamd::RawMorph> recs;
		for (size_t i = 0; i < nM; ++i)
		{
			const MorphemeRaw& m = i < morphemes.size() ? morphemes[i] : cMorphs[i - morphemes.size()];
			if (m.chunks.size() > 255) throw std::runtime_error{ "a morpheme of more than 255 chunks" };
			kamd::RawMorph r{};
			r.kform = m.kform; r.lmId = m.lmMorphemeId; r.origId = m.origMorphemeId; r.combined = m.combined; r.userScore = m.userScore;
			r.chunkPtr = (uint32_t)chunkIds.size(); r.tag = (uint8_t)m.tag; r.vpPack = m.vpPack; r.senseId = m.senseId; r.socket = m.combineSocket;
			r.dialect = (uint16_t)m.dialect; r.nChunks = (uint8_t)m.chunks.size();
			for (size_t c = 0; c < m.chunks.size(); ++c)
			{
				chunkIds.push_back(m.chunks[c]);
				chunkPos.push_back((uint8_t)m.chunkPositions[c].first); chunkPos.push_back((uint8_t)m.chunkPositions[c].second);
			}
			recs.push_back(r);
		}
		// the language-model files as they lie in the directory (KiwiBuilder.cpp:939-1031); the vocabulary size is the language model's
		const auto knlm = slurp(dir + "/sj.knlm"), sbg = slurp(dir + "/skipbigram.mdl"), cong = slurp(dir + "/cong.mdl"), nounchr = slurp(dir + "/nounchr.mdl");
		uint64_t vocab = 0;
		if (knlm.size() >= sizeof(lm::KnLangModelHeader)) { lm::KnLangModelHeader hd; std::memcpy(&hd, knlm.data(), sizeof(hd)); vocab = hd.vocab_size; }
		else if (cong.size() >= sizeof(lm::CoNgramModelHeader)) { lm::CoNgramModelHeader hd; std::memcpy(&hd, cong.data(), sizeof(hd)); vocab = hd.vocabSize; }
		else throw std::runtime_error{ "neither sj.knlm nor cong.mdl in " + dir };
		const uint32_t meta[4] = { (uint32_t)nF, (uint32_t)nM, (uint32_t)vocab, 0 };
		kamd::ContainerWriter w;
		w.add("meta", meta, sizeof(meta));
		w.add("form_ptr", formPtr); w.add("form_chars", chars); w.add("form_cand_ptr", candPtr); w.add("form_cand", cands);
		w.add("morph", recs); w.add("chunk_ids", chunkIds); w.add("chunk_pos", chunkPos);
		if (!knlm.empty()) w.add("knlm", knlm.data(), knlm.size());
		if (!sbg.empty()) w.add("sbg", sbg.data(), sbg.size());
		if (!cong.empty()) w.add("cong", cong.data(), cong.size());
		if (!nounchr.empty()) w.add("nounchr", nounchr.data(), nounchr.size());
		w.save(argv[2], "KAMDRAW1");
		std::printf("%zu forms, %zu morphemes (%zu rule-combined), vocabulary %llu -> %s\n", nF, nM, cMorphs.size(), (unsigned long long)vocab, argv[2]);
		return 0;
	}
	catch (const std::exception& e) { std::fprintf(stderr, "export_built: %s\n", e.what()); return 1; }
}
