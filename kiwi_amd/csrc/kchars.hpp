// Character-class helpers shared by host code and HIP kernels.
// Semantics follow the reference's helpers (cited per function); the code is written for this repo.
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define KAMD_HD __host__ __device__ inline
#else
#define KAMD_HD inline
#endif

namespace kamd
{
	// POSTag numbering: /root/reference/include/kiwi/Types.h:195-227 (part of the C ABI: kiwi_token_info_t.tag)
	enum Tag : uint8_t
	{
		T_UNKNOWN, T_NNG, T_NNP, T_NNB, T_VV, T_VA, T_MAG, T_NR, T_NP, T_VX, T_MM, T_MAJ, T_IC,
		T_XPN, T_XSN, T_XSV, T_XSA, T_XSM, T_XR, T_VCP, T_VCN,
		T_SF, T_SP, T_SS, T_SSO, T_SSC, T_SE, T_SO, T_SW, T_SB, T_SL, T_SH, T_SN,
		T_W_URL, T_W_EMAIL, T_W_MENTION, T_W_HASHTAG, T_W_SERIAL, T_W_EMOJI,
		T_JKS, T_JKC, T_JKG, T_JKO, T_JKB, T_JKV, T_JKQ, T_JX, T_JC,
		T_EP, T_EF, T_EC, T_ETN, T_ETM, T_Z_CODA, T_Z_SIOT,
		T_USER0, T_USER1, T_USER2, T_USER3, T_USER4, T_P, T_MAX,
		T_IRREGULAR = 0x80,
	};
	constexpr uint32_t kDefaultTagSize = T_P;                  // Types.h:257
	constexpr uint32_t kDefaultFormSize = kDefaultTagSize + 26; // src/KiwiBuilder.cpp:40

	// CondVowel / CondPolarity: Types.h:263-288
	enum CondV : uint8_t { CV_NONE, CV_ANY, CV_VOWEL, CV_VOCALIC, CV_VOCALIC_H, CV_NON_VOWEL, CV_NON_VOCALIC, CV_NON_VOCALIC_H, CV_APPLOSIVE, CV_COUNT };
	enum CondP : uint8_t { CP_NONE, CP_POSITIVE, CP_NEGATIVE, CP_NON_ADJ };

	// Match bit flags: /root/reference/include/kiwi/PatternMatcher.h
	enum MatchBits : uint64_t
	{
		M_URL = 1 << 0, M_EMAIL = 1 << 1, M_HASHTAG = 1 << 2, M_MENTION = 1 << 3, M_SERIAL = 1 << 4, M_EMOJI = 1 << 5,
		M_NORMALIZE_CODA = 1 << 16, M_JOIN_NOUN_PREFIX = 1 << 17, M_JOIN_NOUN_SUFFIX = 1 << 18,
		M_JOIN_VERB_SUFFIX = 1 << 19, M_JOIN_ADJ_SUFFIX = 1 << 20, M_JOIN_ADV_SUFFIX = 1 << 21,
		M_SPLIT_COMPLEX = 1 << 22, M_Z_CODA = 1 << 23, M_COMPATIBLE_JAMO = 1 << 24,
		M_SPLIT_SAISIOT = 1 << 25, M_MERGE_SAISIOT = 1 << 26,
	};

	KAMD_HD uint8_t clearIrregular(uint8_t t) { return t & 0x7F; }
	KAMD_HD bool isIrregularTag(uint8_t t) { return (t & 0x80) != 0; }
	// include/kiwi/TagUtils.h:25-49
	KAMD_HD bool isEClass(uint8_t t) { return T_EP <= t && t <= T_ETM; }
	KAMD_HD bool isJClass(uint8_t t) { return T_JKS <= t && t <= T_JC; }
	KAMD_HD bool isNNClass(uint8_t t) { return T_NNG <= t && t <= T_NNB; }
	KAMD_HD bool isSuffixTag(uint8_t t) { t = clearIrregular(t); return T_XSN <= t && t <= T_XSM; }
	// src/TagUtils.cpp:20-24
	KAMD_HD bool isVerbClass(uint8_t t)
	{
		t = clearIrregular(t);
		return t == T_VV || t == T_VA || t == T_VX || t == T_XSV || t == T_XSA || t == T_VCP || t == T_VCN;
	}

	KAMD_HD bool isHangulSyllable(uint32_t c) { return 0xAC00 <= c && c < 0xD7A4; }
	KAMD_HD bool isHangulCoda(uint32_t c) { return 0x11A8 <= c && c < 0x11A8 + 27; }
	KAMD_HD bool isHighSurrogate(uint32_t c) { return (c & 0xFC00) == 0xD800; }
	KAMD_HD bool isLowSurrogate(uint32_t c) { return (c & 0xFC00) == 0xDC00; }
	KAMD_HD uint32_t mergeSurrogate(uint32_t h, uint32_t l) { return (((h & 0x3FF) << 10) | (l & 0x3FF)) + 0x10000; }

	// include/kiwi/Utils.h:296-327
	KAMD_HD bool isSpace(uint32_t c)
	{
		switch (c)
		{
		case 0x20: case 0x0C: case 0x0A: case 0x0D: case 0x09: case 0x0B: case 0xA0: case 0x1680:
		case 0x2000: case 0x2001: case 0x2002: case 0x2003: case 0x2004: case 0x2005: case 0x2006:
		case 0x2007: case 0x2008: case 0x2009: case 0x200A: case 0x202F: case 0x205F: case 0x2800: case 0x3000:
			return true;
		}
		return false;
	}
}
