// Left-context feature tests (vowel / polarity agreement) used by the lattice search.
// Behaviour follows /root/reference/src/FeatureTestor.cpp:6-104; here they are folded into a
// 13-bit mask per left string so that the Viterbi kernel tests a condition with one AND.
#pragma once
#include "kchars.hpp"

namespace kamd
{
	// FeatureTestor::isMatched(begin, end, CondVowel)  (FeatureTestor.cpp:6-58)
	KAMD_HD bool matchVowel(const uint16_t* s, uint32_t n, uint8_t cond)
	{
		if (cond == CV_NONE) return true;
		if (n == 0) return false;
		if (cond == CV_ANY) return true;
		const uint32_t e = s[n - 1];
		if (cond == CV_APPLOSIVE)
		{
			switch (e)
			{
			case 0x11A8: case 0x11A9: case 0x11AA: case 0x11AE: case 0x11B8: case 0x11B9: case 0x11BA:
			case 0x11BB: case 0x11BD: case 0x11BE: case 0x11BF: case 0x11C0: case 0x11C1:
				return true;
			}
			return false;
		}
		const bool syl = 0xAC00 <= e && e <= 0xD7A4, cod = 0x11A8 <= e && e <= 0x11C2;
		if (!syl && !cod) return true;
		switch (cond)
		{
		case CV_VOCALIC_H: if (e == 0x11C2) return true; // fallthrough
		case CV_VOCALIC: if (e == 0x11AF) return true;   // fallthrough
		case CV_VOWEL: return !cod;
		case CV_NON_VOCALIC_H: if (e == 0x11C2) return false; // fallthrough
		case CV_NON_VOCALIC: if (e == 0x11AF) return false;   // fallthrough
		case CV_NON_VOWEL: return !syl;
		}
		return false;
	}

	// FeatureTestor::isMatched(begin, end, CondPolarity)  (FeatureTestor.cpp:60-78)
	KAMD_HD bool matchPolar(const uint16_t* s, uint32_t n, uint8_t polar)
	{
		if (polar == CP_NONE || polar == CP_NON_ADJ) return true;
		if (n == 0) return true;
		for (int32_t i = (int32_t)n - 1; i >= 0; --i)
		{
			const uint32_t c = s[i];
			if (0x11A8 <= c && c <= 0x11C2) continue;
			if (c == 0x1161 || c == 0x1163 || c == 0x1169 || c == 0x116D || c == 0x119E) return polar == CP_POSITIVE;
			if (!(0xAC00 <= c && c <= 0xD7A4)) break;
			const int v = ((c - 0xAC00) / 28) % 21;
			if (v == 0 || v == 2 || v == 8 || v == 12) return polar == CP_POSITIVE;
			if (v == 18 && i == (int32_t)n - 1) continue;
			return polar == CP_NEGATIVE;
		}
		return polar == CP_NEGATIVE;
	}

	// bit v (0..8): matchVowel(s, v); bit 9+p (0..3): matchPolar(s, p)
	KAMD_HD uint16_t featMask(const uint16_t* s, uint32_t n)
	{
		uint16_t m = 0;
		for (uint8_t v = 0; v < CV_COUNT; ++v) if (matchVowel(s, n, v)) m |= (uint16_t)(1u << v);
		for (uint8_t p = 0; p < 4; ++p) if (matchPolar(s, n, p)) m |= (uint16_t)(1u << (9 + p));
		return m;
	}

	// featMask in ONE backward pass over the string (the kernels call it once per lattice node with a surface string: the thirteen
	// predicate calls above re-read the string's tail thirteen times).  The nine vowel conditions are functions of the last unit alone,
	// the polarity conditions of one backward scan; tests/test_feature_mask.py checks it against featMask unit by unit.
	KAMD_HD uint16_t featMaskFast(const uint16_t* s, uint32_t n)
	{
		if (n == 0) return (uint16_t)((1u << CV_NONE) | (0xFu << 9));      // (no string: only "no condition" matches; every polarity does)
		const uint32_t e = s[n - 1];
		uint32_t m = (1u << CV_NONE) | (1u << CV_ANY);
		switch (e)
		{
		case 0x11A8: case 0x11A9: case 0x11AA: case 0x11AE: case 0x11B8: case 0x11B9: case 0x11BA:
		case 0x11BB: case 0x11BD: case 0x11BE: case 0x11BF: case 0x11C0: case 0x11C1:
			m |= 1u << CV_APPLOSIVE;
		}
		const bool syl = 0xAC00 <= e && e <= 0xD7A4, cod = 0x11A8 <= e && e <= 0x11C2;
		if (!syl && !cod) m |= (1u << CV_VOWEL) | (1u << CV_VOCALIC) | (1u << CV_VOCALIC_H) | (1u << CV_NON_VOWEL) | (1u << CV_NON_VOCALIC) | (1u << CV_NON_VOCALIC_H);
		else
		{
			if (!cod) m |= 1u << CV_VOWEL;
			if (!cod || e == 0x11AF) m |= 1u << CV_VOCALIC;
			if (!cod || e == 0x11AF || e == 0x11C2) m |= 1u << CV_VOCALIC_H;
			if (!syl) m |= 1u << CV_NON_VOWEL;
			if (!syl && e != 0x11AF) m |= 1u << CV_NON_VOCALIC;
			if (!syl && e != 0x11AF && e != 0x11C2) m |= 1u << CV_NON_VOCALIC_H;
		}
		// polarity of the last vowel (FeatureTestor.cpp:60-78): positive / negative
		bool positive = false;
		for (int32_t i = (int32_t)n - 1; i >= 0; --i)
		{
			const uint32_t c = s[i];
			if (0x11A8 <= c && c <= 0x11C2) continue;
			if (c == 0x1161 || c == 0x1163 || c == 0x1169 || c == 0x116D || c == 0x119E) { positive = true; break; }
			if (!(0xAC00 <= c && c <= 0xD7A4)) break;
			const int v = ((c - 0xAC00) / 28) % 21;
			if (v == 0 || v == 2 || v == 8 || v == 12) { positive = true; break; }
			if (v == 18 && i == (int32_t)n - 1) continue;
			break;
		}
		m |= (1u << (9 + CP_NONE)) | (1u << (9 + CP_NON_ADJ)) | (positive ? (1u << (9 + CP_POSITIVE)) : (1u << (9 + CP_NEGATIVE)));
		return (uint16_t)m;
	}

	KAMD_HD bool featTest(uint16_t mask, uint8_t vowel, uint8_t polar)
	{
		if (vowel >= CV_COUNT) return false; // typo-only conditions never match (FeatureTestor.cpp:55-57)
		return ((mask >> vowel) & 1) && ((mask >> (9 + (polar & 3))) & 1);
	}
}
