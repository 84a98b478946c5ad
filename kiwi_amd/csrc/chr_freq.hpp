// Match::oovChrFreqModel / oovChrFreqBranchModel (include/kiwi/PatternMatcher.h:20-24): the character model's score of an
// unknown form mixed with how often the form's prefixes occur in the text under analysis (UnkFormScorer::chrFreqBasedScore,
// /root/reference/src/UnkFormScorer.cpp:68-116; chrFreqBranchBasedScore :118-121 returns the same value -- the code behind its first line is unreachable).
//
// The reference counts every substring of at most 32 units of the FILTERED text (Kiwi.cpp:1058-1086: special characters and spaces become ' ', and a
// substring never spans a ' ') in a hash table (src/SubstringCounter.hpp) and looks the prefixes of a form up by content.  Here a form's counts come from one
// pass over the filtered text: the longest common prefix of the form and the text at every position (substringCounts).  Device code (k_unk_chr_freq) and
// the host side of the engine share this file; the test oracle restates the scorer on its own, with libm and a content-keyed table.
#pragma once
#include "flat_model.hpp"
#include "exact_math.hpp"

namespace kamd
{
	constexpr uint32_t kSubstrMaxLen = 32;      // SubstringCounter's maxLen (src/SubstringCounter.hpp:87)

	struct ChrFreqParams { float globalWeight, localWeight, globalMinFreq; };      // KiwiConfig::oovGlobalWeight / oovLocalWeight / oovGlobalMinFreq (include/kiwi/Kiwi.h:157-159)

	// Kiwi.cpp:1066-1082: the character types whose units are blanked in the filtered text (`type` = identifySpecialChr of the UTF-16 UNIT, surrogates unmerged)
	KAMD_HD bool chrFreqFiltered(uint8_t type)
	{
		switch (type)
		{
		case T_UNKNOWN: case T_SF: case T_SP: case T_SS: case T_SSO: case T_SSC: case T_SE: case T_SO: case T_SW: case T_SB: return true;
		default: return false;
		}
	}

	// cnt[j - 1] = SubstringCounter::count of form[0 .. j) for 1 <= j <= min(len, 32), as the reference's 16-bit counters hold it (they wrap);
	// a ' ' inside the form matches nothing (no counted substring holds one).  `stride`: distance between a lane's counters (LDS layout of the kernel).
	KAMD_HD void substringCounts(const uint16_t* text, uint32_t textLen, const uint16_t* form, uint32_t len, uint16_t* cnt, uint32_t stride)
	{
		const uint32_t L = len < kSubstrMaxLen ? len : kSubstrMaxLen;
		for (uint32_t j = 0; j < L; ++j) cnt[j * stride] = 0;
		if (!L) return;
		const uint16_t f0 = form[0];
		if (f0 == u' ') return;
		for (uint32_t p = 0; p < textLen; ++p)
		{
			if (text[p] != f0) continue;
			uint32_t m = 1;
			const uint32_t lim = (textLen - p) < L ? (textLen - p) : L;
			while (m < lim && form[m] != u' ' && text[p + m] == form[m]) ++m;
			for (uint32_t j = 0; j < m; ++j) cnt[j * stride] = (uint16_t)(cnt[j * stride] + 1);
		}
	}

	// chrFreqBasedScore before `score -= chrBias` (the early return of :101 included: its value leaves without the bias).  tokAt(i) = the character
	// model's token of unit i (ChrTokenizer::encodeOne); cnt as substringCounts leaves it.  Every float operation in the original's order;
	// tanhf / expf / logf are the glibc algorithms of exact_math.hpp (the reference calls libm).  `biased` tells the caller whether the bias applies.
	template<class TokAt>
	KAMD_HD float chrFreqScore(const ChrView& C, const ChrFreqParams& Q, uint32_t len, TokAt tokAt, const uint16_t* cnt, uint32_t stride, bool& biased)
	{
		using namespace exact;
		int32_t node = C.bosNode; uint32_t ctx = C.bosCtxPacked;
		float score = 0;
		biased = true;
		for (uint32_t i = 0; i < len; ++i)
		{
			const uint32_t depth = C.depth[node];
			float globalContextFreq = Q.globalMinFreq;
			if (!(depth < i))
			{
				const float f = C.hasFreq ? C.freqTab[ctx >> 24] : 0.f;
				globalContextFreq = (f < Q.globalMinFreq) ? Q.globalMinFreq : f;      // std::max(f, globalMinFreq)
			}
			const float globalContextFreqSat = tanhf_glibc(globalContextFreq / Q.globalWeight) * Q.globalWeight;
			const float lprob = chrProgressPacked(C, node, ctx, tokAt(i));
			if (i == 0) { score += lprob; continue; }
			const float localContextFreq = (float)(i <= kSubstrMaxLen ? cnt[(i - 1) * stride] : 0) - 1;
			if (localContextFreq > 0)
			{
				const float curFreq = (float)(i + 1 <= kSubstrMaxLen ? cnt[i * stride] : 0) - 1;
				if (curFreq < 0) { biased = false; return -99999.f; }
				const float localContextFreqSat = tanhf_glibc(localContextFreq / Q.localWeight) * Q.localWeight;
				const float localFreq = curFreq * (localContextFreqSat / localContextFreq);
				const float globalFreq = globalContextFreqSat * expf_glibc(lprob);
				const float mixedProb = logf_glibc((localFreq + globalFreq) / (localContextFreqSat + globalContextFreqSat));
				score += mixedProb;
			}
			else score += lprob;
		}
		score += chrProgressPacked(C, node, ctx, 0);
		return score;
	}
}
