// Host-side string helpers of the analyze path (normalisation, character typing, Hangul re-joining).
// Each function cites the reference behaviour it reproduces; code is this repo's own.
#pragma once
#include <string>
#include <string_view>
#include <vector>
#include "kchars.hpp"

namespace kamd
{
	using U16 = std::u16string;

	inline bool isChineseChr(uint32_t c) // src/StrUtils.h:706-721
	{
		static const uint32_t r[][2] = {
			{0x4E00, 0x9FFF}, {0x3400, 0x4DBF}, {0x20000, 0x2A6DF}, {0x2A700, 0x2B73F}, {0x2B820, 0x2CEAF},
			{0x2CEB0, 0x2EBEF}, {0x30000, 0x3134F}, {0x31350, 0x323AF}, {0xF900, 0xFAFF}, {0x2F800, 0x2FA1F},
			{0x2F00, 0x2FDF}, {0x2E80, 0x2EFF} };
		for (auto& p : r) if (p[0] <= c && c <= p[1]) return true;
		return false;
	}

	// src/Utils.cpp:76-183 — character type as a POSTag: spaces -> unknown, Hangul -> max, digits -> sn ...
	inline uint8_t identifySpecialChr(uint32_t c)
	{
		if (c < 0x10000 && isSpace(c)) return T_UNKNOWN;
		if (0x2000 <= c && c <= 0x200F) return T_UNKNOWN;
		if ('0' <= c && c <= '9') return T_SN;
		if (('A' <= c && c <= 'Z') || ('a' <= c && c <= 'z')) return T_SL;
		if (0xAC00 <= c && c < 0xD7A4) return T_MAX;
		if (c < 0x10000)
		{
			const bool oldOnset = (0x1100 <= c && c < 0x1160) || (0xA960 <= c && c < 0xA980);
			const bool oldVowel = (0x1160 <= c && c < 0x11A8) || (0xD7B0 <= c && c < 0xD7CB);
			const bool oldCoda = (0x11A8 <= c && c < 0x1200) || (0xD7CB <= c && c < 0xD800);
			const bool tone = 0x302E <= c && c < 0x3030;
			if (oldOnset || oldVowel || oldCoda || tone) return T_MAX;
		}
		switch (c)
		{
		case '.': case '!': case '?': case 0x2047: case 0x2048: case 0x2049: case 0x3002: case 0xff01: case 0xff0e: case 0xff1f: case 0xff61:
			return T_SF;
		case '-': case '~': case 0x223c: case 0x301c: case 0xff5e:
			return T_SO;
		case 0x2026: case 0x205d:
			return T_SE;
		case ',': case ';': case ':': case '/': case 0xb7: case 0x3001: case 0xff0c: case 0xff1a: case 0xff1b: case 0xff64:
			return T_SP;
		case '(': case '<': case '[': case '{': case 0x2018: case 0x201c: case 0x226a: case 0x3008: case 0x300a: case 0x300c:
		case 0x300e: case 0x3010: case 0x3014: case 0x3016: case 0x3018: case 0x301a: case 0xff08: case 0xff1c: case 0xff3b:
		case 0xff5b: case 0xff5f: case 0xff62:
			return T_SSO;
		case ')': case '>': case ']': case '}': case 0x2019: case 0x201d: case 0x226b: case 0x3009: case 0x300b: case 0x300d:
		case 0x300f: case 0x3011: case 0x3015: case 0x3017: case 0x3019: case 0x301b: case 0xff09: case 0xff1e: case 0xff3d:
		case 0xff5d: case 0xff60: case 0xff63:
			return T_SSC;
		case '"': case '\'': case 0xad: case 0x2015: case 0x2500: case 0xff0d:
			return T_SS;
		}
		if (isChineseChr(c)) return T_SH;
		if (0xd800 <= c && c <= 0xdfff) return T_SH;
		return T_SW;
	}

	// include/kiwi/Utils.h:167-206 — syllable + separated coda -> composed syllable
	inline U16 joinHangul(const char16_t* s, size_t n)
	{
		U16 ret;
		ret.reserve(n);
		for (size_t i = 0; i < n; ++i)
		{
			const char16_t c = s[i];
			if (!ret.empty() && isHangulSyllable(ret.back()))
			{
				const bool hasCoda = (ret.back() - 0xAC00) % 28 != 0;
				const bool oldCoda = (0x11A8 <= c && c < 0x1200) || (0xD7CB <= c && c < 0xD800);
				if (hasCoda) ret.push_back(c);
				else if (isHangulCoda(c)) ret.back() += c - 0x11A7;
				else if (oldCoda)
				{
					const int onset = (ret.back() - 0xAC00) / 28 / 21, vowel = (ret.back() - 0xAC00) / 28 % 21;
					ret.back() = (char16_t)(0x1100 + onset);
					ret.push_back((char16_t)(0x1161 + vowel));
					ret.push_back(c);
				}
				else ret.push_back(c);
			}
			else ret.push_back(c);
		}
		return ret;
	}
	inline U16 joinHangul(const U16& s) { return joinHangul(s.data(), s.size()); }

	// src/Utils.cpp:264-298 — bullet ("SB") shape class of a form; an empty form yields 0 there as well.
	inline uint32_t getSBType(std::u16string_view form)
	{
		if (form.empty()) return 0;
		uint32_t format = 0, group = 0;
		uint32_t chr = form[0];
		if (form.back() == u'.') format = 1;
		else if (form.back() == u')')
		{
			if (form[0] == u'(') { chr = form.size() > 1 ? form[1] : 0; format = 2; }
			else format = 3;
		}
		if (0xAC00 <= chr && chr <= 0xD7A3) group = 1;
		else if (0x3131 <= chr && chr <= 0x314E) group = 2;
		else if ('0' <= chr && chr <= '9') group = 3;
		else if (0x2160 <= chr && chr <= 0x216B) group = 4;
		else if (0x2170 <= chr && chr <= 0x217B) group = 5;
		else if (0x2460 <= chr && chr <= 0x2473) return 24;
		else if (0x2780 <= chr && chr <= 0x2789) return 24;
		else if (0x2776 <= chr && chr <= 0x277F) return 25;
		else if (0x278A <= chr && chr <= 0x2793) return 25;
		else if (0x2474 <= chr && chr <= 0x2487) return 26;
		else if (0x2488 <= chr && chr <= 0x249B) return 27;
		return format | (group << 2);
	}

	// src/Utils.cpp:423-459 — lexicographic order of strings with spaces skipped
	inline int cmpIgnoringSpace(const U16& a, const U16& b)
	{
		size_t i = 0, j = 0;
		while (i < a.size() && j < b.size())
		{
			if (a[i] == u' ' && b[j] == u' ') { ++i; ++j; continue; }
			if (a[i] == u' ') { ++i; continue; }
			if (b[j] == u' ') { ++j; continue; }
			if (a[i] == b[j]) { ++i; ++j; continue; }
			return a[i] < b[j] ? -1 : 2; // 2: differ, not less
		}
		if (i >= a.size() && j >= b.size()) return 0;
		if (i >= a.size()) return -1;
		return 1; // a has leftovers: neither less nor equal in the reference's pair of predicates
	}
	inline bool lessIgnoringSpace(const U16& a, const U16& b) { return cmpIgnoringSpace(a, b) == -1; }
	inline bool equalIgnoringSpace(const U16& a, const U16& b) { return cmpIgnoringSpace(a, b) == 0; }
}
