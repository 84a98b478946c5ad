// Host-side text preparation (see textprep.hpp).  The pattern recognisers accept the regular languages of
// /root/reference/src/PatternMatcher.cpp:53-364 (URL, e-mail, mention, hashtag, number, serial, abbreviation, emoji), written over a small cursor type.
#include <algorithm>
#include <cstring>
#include <stdexcept>
#include "textprep.hpp"

namespace kamd
{
	namespace
	{
		struct ScriptRange { uint32_t lo, hi; uint8_t id; };
		struct CpRange { uint32_t lo, hi; };
#include "unicode_tables.inc"

		template<class R, size_t N>
		const R* findRange(const R(&tab)[N], uint32_t c)
		{
			size_t lo = 0, hi = N;
			while (lo < hi)
			{
				const size_t mid = (lo + hi) / 2;
				if (tab[mid].hi < c) lo = mid + 1; else hi = mid;
			}
			if (lo < N && tab[lo].lo <= c) return &tab[lo];
			return nullptr;
		}

		inline bool isAlpha(uint32_t c) { return ('A' <= c && c <= 'Z') || ('a' <= c && c <= 'z'); }
		inline bool isUpper(uint32_t c) { return 'A' <= c && c <= 'Z'; }
		inline bool isDigit(uint32_t c) { return ('0' <= c && c <= '9') || (0xff10 <= c && c <= 0xff19); }
		inline bool isAlnum(uint32_t c) { return isAlpha(c) || ('0' <= c && c <= '9'); }
		inline bool inSet(uint32_t c, const char* lits) { return c < 128 && c != 0 && std::strchr(lits, (int)c) != nullptr; }
		inline bool csEmailAccount(uint32_t c) { return isAlnum(c) || inSet(c, "-._%+"); }
		inline bool csAlnumDotDash(uint32_t c) { return isAlnum(c) || inSet(c, "-."); }
		inline bool csDomain(uint32_t c) { return isAlnum(c) || inSet(c, "-@:%._+~#="); }
		inline bool csPath(uint32_t c) { return isAlnum(c) || inSet(c, "-()@:%_+.~#!?&/="); }
		inline bool csHashtag(uint32_t c) { return !inSet(c, "# \t\n\r\v\f.,()[]<>{}"); }
		inline bool csSpace(uint32_t c) { return inSet(c, " \t\n\r\v\f"); }

		using P = const char16_t*;

		// The recognisers below are written over a cursor with the four moves they are made of -- take one unit of a class, take a given unit, take
		// a run of a class, take a literal -- so that each of them reads as the shape of the pattern it accepts (the reference states the same
		// languages as hand-expanded scanners over iterators, src/PatternMatcher.cpp:53-364; what is matched, and how much, is the same to the unit).
		struct Cursor
		{
			P at, end;
			bool more() const { return at != end; }
			size_t left() const { return (size_t)(end - at); }
			template<class Cls> bool peek(Cls&& cls) const { return at != end && cls(*at); }
			bool peekUnit(char16_t u) const { return at != end && *at == u; }
			template<class Cls> bool one(Cls&& cls) { if (!peek(cls)) return false; ++at; return true; }
			bool unit(char16_t u) { if (!peekUnit(u)) return false; ++at; return true; }
			template<class Cls> size_t run(Cls&& cls) { const P from = at; while (peek(cls)) ++at; return (size_t)(at - from); }
			bool literal(const char* ascii)
			{
				const size_t n = std::strlen(ascii);
				if (left() < n) return false;
				for (size_t i = 0; i < n; ++i) if (at[i] != (char16_t)ascii[i]) return false;
				at += n;
				return true;
			}
			// steps back over ONE trailing unit of the given set (a pattern never ends in its own punctuation)
			void unlessEndsIn(const char16_t* set) { for (; *set; ++set) if (at[-1] == *set) { --at; return; } }
		};

		// The longest prefix of the run of `cls` at `from` that ends in a top-level-domain-like tail: a '.', then two or more letters with nothing but
		// letters since the dot.  Null when the run has no such point.
		template<class Cls>
		P domainTail(P from, P end, Cls&& cls)
		{
			P best = nullptr;
			int lettersSinceDot = -1;      // -1: no dot yet, or something other than letters after it
			for (P p = from; p != end && cls(*p); ++p)
			{
				if (*p == u'.') lettersSinceDot = 0;
				else if (!isAlpha(*p)) lettersSinceDot = -1;
				else if (lettersSinceDot >= 0 && ++lettersSinceDot >= 2) best = p + 1;
			}
			return best;
		}

		// http(s)://host.tld[:port][/path]
		size_t testUrl(P first, P last)
		{
			Cursor c{ first, last };
			if (!c.literal("http://") && !c.literal("https://")) return 0;
			if (!c.one(csDomain)) return 0;
			const P host = domainTail(c.at, last, csDomain);
			if (!host) return 0;
			c.at = host;
			if (c.unit(u':') && !c.run(isDigit)) return 0;
			if (c.unit(u'/')) c.run(csPath);
			else if (c.more() && !csSpace(*c.at)) return 0;
			c.unlessEndsIn(u".:");
			return (size_t)(c.at - first);
		}

		// account@host.tld
		size_t testEmail(P first, P last)
		{
			Cursor c{ first, last };
			if (!c.run(csEmailAccount) || !c.unit(u'@') || !c.one(csAlnumDotDash)) return 0;
			const P tail = domainTail(c.at, last, csAlnumDotDash);
			return tail ? (size_t)(tail - first) : 0;
		}

		// @name: a letter, then account characters; at least four units in all
		size_t testMention(P first, P last)
		{
			Cursor c{ first, last };
			if (!c.unit(u'@') || !c.one(isAlpha)) return 0;
			c.run(csEmailAccount);
			c.unlessEndsIn(u".%+-");
			const size_t n = (size_t)(c.at - first);
			return n > 3 ? n : 0;
		}

		// #tag
		size_t testHashtag(P first, P last)
		{
			Cursor c{ first, last };
			if (!c.unit(u'#') || !c.run(csHashtag)) return 0;
			return (size_t)(c.at - first);
		}

		// digits[,ddd]*[.digits] -- not when a further '.' follows (that is a serial number); `left` is the unit before `first`
		size_t testNumeric(char16_t left, P first, P last)
		{
			Cursor c{ first, last };
			if (!c.run(isDigit)) return 0;
			bool grouped = false;
			while (c.unit(u','))
			{
				if (c.left() < 3 || !isDigit(c.at[0]) || !isDigit(c.at[1]) || !isDigit(c.at[2])) return (size_t)(c.at - 1 - first);      // the comma was punctuation
				c.at += 3;
				grouped = true;
			}
			if (!c.more() || isSpace(*c.at) || isHangulSyllable(*c.at)) return (size_t)(c.at - first);
			if (c.unit(u'.'))
			{
				const bool standsAlone = !grouped && !csAlnumDotDash(left) && !c.peek(csAlnumDotDash);      // "3." of an enumeration: the point belongs to it
				if (standsAlone) return (size_t)(c.at - first);
				if (!c.run(isDigit)) return (size_t)(c.at - 1 - first);      // the point was punctuation
			}
			return c.peekUnit(u'.') ? 0 : (size_t)(c.at - first);
		}

		// number SEP number [SEP number]*, SEP one of : . - / (the same throughout), a single blank allowed after each; with '.' three numbers at least
		size_t testSerial(P first, P last)
		{
			Cursor c{ first, last };
			if (!c.run(isDigit) || !c.more()) return 0;
			const char16_t sep = *c.at;
			if (sep != u':' && sep != u'.' && sep != u'-' && sep != u'/') return 0;
			++c.at;
			c.unit(u' ');
			if (!c.run(isDigit)) return 0;
			if (sep == u'.' && !c.peekUnit(sep)) return 0;
			while (c.unit(sep))
			{
				c.unit(u' ');
				if (!c.run(isDigit)) break;
			}
			c.unlessEndsIn(u" ");
			return (size_t)(c.at - first);
		}

		// an abbreviation: letters '.', either followed by a blank (short word) or continued as letters '.' letters '.' ... (each part at most five letters)
		size_t testAbbr(P first, P last)
		{
			Cursor c{ first, last };
			size_t letters = c.run(isAlpha);
			if (!letters || !c.unit(u'.')) return 0;
			if (c.peekUnit(u' ')) return letters > (isUpper(*first) ? 5u : 3u) ? 0 : (size_t)(c.at - first);
			if (letters > 5) return 0;
			while (c.peek(isAlpha))
			{
				letters = c.run(isAlpha);
				if (letters > 5) return 0;
				if (!c.unit(u'.')) return (size_t)(c.at - first);
			}
			c.unlessEndsIn(u" ");
			return (size_t)(c.at - first);
		}

		// one code point at p (a surrogate pair when it is whole), 0 at the end; `next` = where the following one starts
		inline uint32_t codePointAt(P p, P end, bool pairNeedsBoth, P& next)
		{
			if (p >= end) { next = p; return 0; }
			if (isHighSurrogate(*p) && (!pairNeedsBoth || p + 1 < end)) { next = p + 2; return mergeSurrogate(p[0], p[1]); }
			next = p + 1;
			return *p;
		}

		// emoji [variation selector | skin tone] [ZWJ emoji ...]
		size_t testEmoji(P first, P last)
		{
			P p = first;
			while (p + 1 < last)
			{
				P afterFirst, afterSecond;
				const uint32_t c0 = codePointAt(p, last, false, afterFirst);
				const uint32_t c1 = codePointAt(afterFirst, last, true, afterSecond);
				const int kind = isEmoji(c0, c1);      // 1: c0 alone is an emoji; 2: only together with the selector / skin tone c1
				if (kind == 1) p = afterFirst; else if (kind == 2) p = afterSecond; else break;
				if (p == last) break;
				if (0xfe00 <= *p && *p <= 0xfe0f) ++p;      // variation selector
				else if (p + 1 < last && isHighSurrogate(*p))
				{
					const uint32_t tone = mergeSurrogate(p[0], p[1]);
					if (0x1f3fb <= tone && tone <= 0x1f3ff) p += 2;
				}
				if (p == last || *p != 0x200d) break;
				++p;      // zero-width joiner: another emoji may follow
			}
			return (size_t)(p - first);
		}
	}

	uint8_t chr2ScriptType(uint32_t c)
	{
		const ScriptRange* r = findRange(kScriptRanges, c);
		return r ? r->id : 0;
	}

	const char* scriptName(uint8_t s) { return s < kScriptCount ? kScriptNames[s] : "Unknown"; }

	int isEmoji(uint32_t c0, uint32_t c1)
	{
		if (findRange(kEmoji1, c0)) return 1;
		if (!(c1 == 0xfe0f || (0x1f3fb <= c1 && c1 <= 0x1f3ff))) return 0;
		if (findRange(kEmoji2, c0)) return 2;
		return 0;
	}

	std::pair<size_t, uint8_t> matchPattern(char16_t left, const char16_t* first, const char16_t* last, uint64_t mo)
	{
		size_t n;
		if ((mo & M_SERIAL) && (n = testSerial(first, last))) return { n, T_W_SERIAL };
		if ((n = testNumeric(left, first, last))) return { n, T_SN };
		if ((mo & M_HASHTAG) && (n = testHashtag(first, last))) return { n, T_W_HASHTAG };
		if ((mo & M_EMAIL) && (n = testEmail(first, last))) return { n, T_W_EMAIL };
		if ((mo & M_MENTION) && (n = testMention(first, last))) return { n, T_W_MENTION };
		if ((mo & M_URL) && (n = testUrl(first, last))) return { n, T_W_URL };
		if ((mo & M_EMOJI) && (n = testEmoji(first, last))) return { n, T_W_EMOJI };
		if ((n = testAbbr(first, last))) return { n, T_SL };
		return { 0, T_UNKNOWN };
	}

	// appends the normalised form of s[0, n) to `out` and its position table (offsets relative to the text's first unit) to `pos`
	static void appendNormalized(const char16_t* s, size_t n, U16& out, std::vector<uint32_t>& pos)
	{
		// (written through pointers into arrays sized for the worst case -- every syllable with a coda -- and cut back: two push_backs per unit were a
		// third of the preparation's time)
		const size_t base = out.size(), pbase = pos.size();
		out.resize(base + 2 * n); pos.resize(pbase + n + 1);
		char16_t* o = n ? &out[base] : nullptr; uint32_t* p = &pos[pbase];
		size_t k = 0;
		for (size_t i = 0; i < n; ++i)
		{
			char16_t c = s[i];
			p[i] = (uint32_t)k;
			if (c == 0xB42C) c = 0xB410;
			if (0xAC00 <= c && c < 0xD7A4)
			{
				const int coda = (c - 0xAC00) % 28;
				o[k++] = (char16_t)(c - coda);
				if (coda) o[k++] = (char16_t)(coda + 0x11A7);
			}
			else o[k++] = c;
		}
		p[n] = (uint32_t)k;
		out.resize(base + k);
	}

	void normalizeWithPosition(const char16_t* s, size_t n, U16& out, std::vector<uint32_t>& pos)
	{
		out.clear(); pos.clear();
		out.reserve(n + n / 2); pos.reserve(n + 1);
		appendNormalized(s, n, out, pos);
	}

	void normalizeCoda(U16& s) { if (!s.empty()) normalizeCoda(&s[0], s.size()); }

	void normalizeCoda(char16_t* s, size_t n) // src/StrUtils.h:637-703: "받침 + 같은 초성체" -> merged
	{
		static const char16_t toOnset[27] = { 0x3131, 0x3131, 0x3145, 0x3134, 0x3148, 0x314E, 0x3137, 0x3139, 0x3131, 0x3141, 0x3142, 0x3145, 0x314C, 0x314D,
			0x314E, 0x3141, 0x3142, 0x3145, 0x3145, 0x3145, 0x3147, 0x3148, 0x314A, 0x314B, 0x314C, 0x314D, 0x314E };
		static const char16_t conv[27] = { 0, 0x11A8, 0x11A8, 0, 0x11AB, 0x11AB, 0, 0, 0x11AF, 0x11AF, 0x11AF, 0x11AF, 0x11AF, 0x11AF, 0x11AF,
			0, 0, 0x11B8, 0, 0x11BA, 0, 0, 0, 0, 0, 0, 0 };
		char16_t before = 0;
		for (size_t i = 0; i < n; ++i)
		{
			if (0x11A8 <= before && before <= 0x11C2)
			{
				const int off = before - 0x11A8;
				if (s[i] == toOnset[off]) s[i - 1] = conv[off] ? conv[off] : s[i];
			}
			before = s[i];
		}
	}

	void prepareText(PreparedText& o, const char16_t* raw, size_t n, uint64_t mo, uint32_t textId, const std::pair<uint32_t, uint32_t>* spans, size_t nSpans)
	{
		PrepBlock blk;
		blk.append(raw, n, mo, textId, spans, nSpans);
		o.norm = std::move(blk.norm); o.position = std::move(blk.position); o.cls = std::move(blk.cls); o.script = std::move(blk.script);
		o.chunks = std::move(blk.chunks); o.patterns = std::move(blk.patterns);
	}

	namespace
	{
		// Per-character facts of the preparation loop as tables: type and script of every ASCII character and of the two Hangul blocks the
		// normalised text consists of, and which characters can START a pattern at all -- matchPattern's recognisers begin with a digit
		// (ASCII or full-width), an ASCII letter, '#', '@', an e-mail account character or an emoji base; everything else (Hangul above all)
		// is rejected without running them.  Built once from the very functions they stand for.
		struct PrepTables
		{
			uint8_t cls[128], script[128]; bool canStart[128];
			uint8_t hangulSyllableScript, hangulJamoScript;
			PrepTables()
			{
				for (uint32_t c = 0; c < 128; ++c)
				{
					cls[c] = identifySpecialChr(c); script[c] = chr2ScriptType(c);
					canStart[c] = isDigit(c) || isAlpha(c) || c == '#' || c == '@' || csEmailAccount(c) || findRange(kEmoji1, c) || findRange(kEmoji2, c);
				}
				hangulSyllableScript = chr2ScriptType(0xAC00); hangulJamoScript = chr2ScriptType(0x11A8);
			}
		};
		const PrepTables& prepTables() { static const PrepTables t; return t; }
	}

	void PrepBlock::append(const char16_t* raw, size_t n, uint64_t mo, uint32_t textId, const std::pair<uint32_t, uint32_t>* spans, size_t nSpans)
	{
		size_t spanAt = 0;      // next pretokenized span (they are consumed in order, across the chunks of the text: KTrie.cpp:769, 788)
		const PrepTables& T = prepTables();
		Idx x{};
		x.normOff = norm.size(); x.posOff = position.size(); x.chunkOff = chunks.size(); x.patOff = patterns.size();
		// normalisation (src/StrUtils.h:494-521: a syllable with a coda becomes syllable + coda jamo) and the typing of what it makes -- ASCII, Hangul syllables,
		// coda jamo: tables -- in ONE pass over the raw text, written through pointers into arrays sized for the worst case (two units per raw unit) and cut
		// back; anything else in the text (other scripts, surrogates, emoji: the typing needs the following code point), or a compatibility jamo behind a coda
		// (normalizeCoda rewrites the coda), sends the whole text through the general typing loop below
		const size_t pbase = position.size();
		norm.resize(x.normOff + 2 * n); position.resize(pbase + n + 1);
		cls.resize(x.normOff + 2 * n); script.resize(x.normOff + 2 * n);
		char16_t* nrm = n ? &norm[x.normOff] : nullptr;
		uint8_t* ocls = cls.data() + x.normOff; uint8_t* oscript = script.data() + x.normOff;
		size_t L = 0;
		bool general = false, compatJamo = false;
		{
			uint32_t* p = &position[pbase];
			for (size_t i = 0; i < n; ++i)
			{
				char16_t c = raw[i];
				p[i] = (uint32_t)L;
				if (c < 128) { nrm[L] = c; ocls[L] = T.cls[c]; oscript[L] = T.script[c]; ++L; continue; }
				if (c == 0xB42C) c = 0xB410;
				if (0xAC00 <= c && c < 0xD7A4)
				{
					const int coda = (c - 0xAC00) % 28;
					nrm[L] = (char16_t)(c - coda); ocls[L] = T_MAX; oscript[L] = T.hangulSyllableScript; ++L;
					if (coda) { nrm[L] = (char16_t)(coda + 0x11A7); ocls[L] = T_MAX; oscript[L] = T.hangulJamoScript; ++L; }
					continue;
				}
				if (0x11A8 <= c && c <= 0x11C2) { nrm[L] = c; ocls[L] = T_MAX; oscript[L] = T.hangulJamoScript; ++L; continue; }
				if (0x3131 <= c && c <= 0x314E) compatJamo = true;
				general = true;
				nrm[L++] = c;
			}
			p[n] = (uint32_t)L;
		}
		norm.resize(x.normOff + L); cls.resize(x.normOff + L); script.resize(x.normOff + L);
		nrm = L ? &norm[x.normOff] : nullptr; ocls = cls.data() + x.normOff; oscript = script.data() + x.normOff;
		if ((mo & M_NORMALIZE_CODA) && compatJamo) normalizeCoda(nrm, L);
		if (general) for (size_t i = 0; i < L; ++i)
		{
			uint32_t c = nrm[i];
			if (c < 128) { ocls[i] = T.cls[c]; oscript[i] = T.script[c]; continue; }                                  // (no emoji starts below U+0080 is flagged here: see below)
			if (0xAC00 <= c && c < 0xD7A4) { ocls[i] = T_MAX; oscript[i] = T.hangulSyllableScript; continue; }
			if (0x11A8 <= c && c <= 0x11C2) { ocls[i] = T_MAX; oscript[i] = T.hangulJamoScript; continue; }
			uint32_t c1 = 0;
			size_t nx = i + 1;
			if (isHighSurrogate(c) && i + 1 < L) { c = mergeSurrogate(c, nrm[i + 1]); nx = i + 2; }
			if (nx < L)
			{
				c1 = nrm[nx];
				if (isHighSurrogate(c1) && nx + 1 < L) c1 = mergeSurrogate(c1, nrm[nx + 1]);
			}
			ocls[i] = identifySpecialChr(c);
			oscript[i] = chr2ScriptType(c);
			if (c >= 0x80 && isEmoji(c, c1)) ocls[i] |= 0x80;
		}
		size_t splitEnd = 0;
		while (splitEnd < L)
		{
			const char16_t* str = nrm + splitEnd;
			const size_t sz = L - splitEnd;
			ChunkDesc ch{};
			ch.textId = textId; ch.startOffset = (uint32_t)splitEnd; ch.patBegin = (uint32_t)(patterns.size() - x.patOff);
			size_t k = 0, contNonSpace = 0;
			uint8_t lastType = T_UNKNOWN;
			bool anyNonSpace = false;
			const uint8_t* ccls = ocls + splitEnd;
			for (; k < sz; ++k)
			{
				if (spanAt < nSpans && spans[spanAt].first == splitEnd + k)      // a pretokenized span: stepped over whole (KTrie.cpp:782-790)
				{
					k += (size_t)(spans[spanAt].second - spans[spanAt].first) - 1;
					++spanAt;
					continue;
				}
				const char16_t c0 = str[k];
				if (c0 < 128 ? T.canStart[c0] : ((ccls[k] & 0x80) || (0xff10 <= c0 && c0 <= 0xff19)))
				{
					auto pm = matchPattern(k ? str[k - 1] : u' ', str + k, str + sz, mo);
					if (pm.second != T_UNKNOWN)
					{
						patterns.push_back(PatternSpan{ (uint32_t)(k + pm.first), (uint32_t)pm.first, pm.second });
						k += pm.first - 1;
						continue;
					}
				}
				uint32_t c32 = c0;
				if (isHighSurrogate(c32) && k + 1 < sz) c32 = mergeSurrogate(c32, str[k + 1]);
				const uint8_t t = ccls[k] & 0x7F;                      // == identifySpecialChr(c32): typed above, surrogate pairs merged at their first unit
				if (t == T_UNKNOWN) contNonSpace = 0; else ++contNonSpace;
				if (t == T_UNKNOWN && k >= (lastType == T_SF ? 4u : 4096u))
				{
					if (!isSpace(str[k - 3]) && !isSpace(str[k - 2])) break;
				}
				else if (contNonSpace >= 1024) break;
				if (c32 >= 0x10000) ++k;
				lastType = t;
			}
			if (k > sz) k = sz;
			for (size_t i = 0; i < k; ++i) if (!isSpace(str[i])) { anyNonSpace = true; break; }
			std::sort(patterns.begin() + x.patOff + ch.patBegin, patterns.end(), [](const PatternSpan& a, const PatternSpan& b)
			{
				if (a.end != b.end) return a.end < b.end;
				if (a.length != b.length) return a.length < b.length;
				return a.tag < b.tag;
			});
			ch.patEnd = (uint32_t)(patterns.size() - x.patOff);
			ch.nChars = (uint32_t)k;
			ch.empty = !anyNonSpace;
			size_t stop = k;
			if (ch.empty) while (stop < sz && isSpace(str[stop])) ++stop;   // KTrie.cpp:1505-1506
			ch.nextOffset = (uint32_t)(splitEnd + stop);
			chunks.push_back(ch);
			if (stop == 0) throw std::runtime_error{ "prepareText: chunker made no progress" };
			splitEnd += stop;
		}
		x.normLen = L; x.posLen = position.size() - x.posOff; x.nChunks = chunks.size() - x.chunkOff; x.nPat = patterns.size() - x.patOff;
		idx.push_back(x);
	}
}
