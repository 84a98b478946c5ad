// TEST INFRASTRUCTURE ONLY -- never linked into the product.
//
// CPU restatement of Match::oovChrFreqModel / oovChrFreqBranchModel:
//   the filtered text                       /root/reference/src/Kiwi.cpp:1058-1086
//   SubstringCounter (ctor, count)          /root/reference/src/SubstringCounter.hpp:84-151
//   UnkFormScorer::chrFreqBasedScore        /root/reference/src/UnkFormScorer.cpp:68-116
//   UnkFormScorer::chrFreqBranchBasedScore  /root/reference/src/UnkFormScorer.cpp:118-121 (returns chrFreqBasedScore: what follows its first line is dead code)
//   CoNgramModel::getNodeDepth / getContextFrequency / dequantizeFrequencyScale   /root/reference/src/CoNgramModel.cpp:929-958, CoNgramModel.hpp:34-39
// Pinned by tests/test_chr_oracle.py against the real translation units (oracle/_ref).  The table of substring counts is kept by CONTENT (the
// reference's open-addressing table compares content too; its hash only places entries), counters are 16 bits wide and wrap like the reference's;
// tanhf / expf / logf are libm's, as in the reference.  The device path (kiwi_amd/csrc/chr_freq.hpp) counts differently -- one pass over the text per
// form -- and evaluates the same expression with the glibc algorithms restated in exact_math.hpp.
#pragma once
#include <cmath>
#include <string>
#include <unordered_map>
#include <vector>
#include "../kiwi_amd/csrc/flat_model.hpp"
#include "../kiwi_amd/csrc/hostutil.hpp"

namespace korc
{
	struct SubstringCounts
	{
		std::unordered_map<std::u16string, uint16_t> table;

		void build(const char16_t* data, size_t size, size_t maxLen = 32)
		{
			table.clear();
			size_t segStart = 0;
			for (size_t s = 0; s <= size; ++s)
			{
				if (s != size && data[s] != u' ') continue;
				for (size_t i = segStart; i < s; ++i)
				{
					const size_t jEnd = std::min(i + maxLen, s);
					std::u16string key;
					for (size_t j = i; j < jEnd; ++j) { key.push_back(data[j]); ++table[key]; }      // (uint16_t: wraps)
				}
				segStart = s + 1;
			}
		}
		size_t count(const uint16_t* s, size_t len) const
		{
			auto it = table.find(std::u16string{ (const char16_t*)s, len });
			return it == table.end() ? 0 : it->second;
		}
	};

	// Kiwi.cpp:1064-1084: identifySpecialChr of every UTF-16 unit on its own
	inline std::u16string filteredText(const char16_t* norm, size_t n)
	{
		std::u16string out{ norm, norm + n };
		for (auto& c : out)
		{
			switch (kamd::identifySpecialChr(c))
			{
			case kamd::T_UNKNOWN: case kamd::T_SF: case kamd::T_SP: case kamd::T_SS: case kamd::T_SSO: case kamd::T_SSC: case kamd::T_SE: case kamd::T_SO: case kamd::T_SW: case kamd::T_SB:
				c = u' ';
				break;
			default: break;
			}
		}
		return out;
	}

	struct ChrFreqConfig { float globalWeight = 35, localWeight = 3, globalMinFreq = 4; };      // include/kiwi/Kiwi.h:157-159

	// chrFreqBasedScore, `score -= chrBias` included
	inline float chrFreqScoreOracle(const kamd::ChrView& C, const SubstringCounts& sc, const ChrFreqConfig& Q, float chrBias, const uint16_t* form, uint32_t len)
	{
		int32_t nodeIdx = C.bosNode; uint32_t contextIdx = C.bosCtxPacked;
		float score = 0;
		for (size_t i = 0; i < len; ++i)
		{
			const size_t depth = C.depth[nodeIdx];
			const float contextFreq = C.hasFreq ? C.freqTab[contextIdx >> 24] : 0.f;
			const float globalContextFreq = depth < i ? Q.globalMinFreq : std::max(contextFreq, Q.globalMinFreq);
			const float globalContextFreqSat = tanhf(globalContextFreq / Q.globalWeight) * Q.globalWeight;
			const float lprob = kamd::chrProgressPacked(C, nodeIdx, contextIdx, kamd::chrToken(form[i], kamd::identifySpecialChr(form[i])));
			if (i == 0) { score += lprob; continue; }
			const float localContextFreq = (float)sc.count(form, i) - 1;
			if (localContextFreq > 0)
			{
				const float curFreq = (float)sc.count(form, i + 1) - 1;
				if (curFreq < 0) return -99999.f;
				const float localContextFreqSat = tanhf(localContextFreq / Q.localWeight) * Q.localWeight;
				const float localFreq = curFreq * (localContextFreqSat / localContextFreq);
				const float globalFreq = globalContextFreqSat * expf(lprob);
				const float mixedProb = logf((localFreq + globalFreq) / (localContextFreqSat + globalContextFreqSat));
				score += mixedProb;
			}
			else score += lprob;
		}
		score += kamd::chrProgressPacked(C, nodeIdx, contextIdx, 0);
		score -= chrBias;
		return score;
	}
}
