// Result assembly, the parts that only look at a text's finished token list -- bracket pairing, the sentence boundary automaton, sentence / line / word
// numbers -- written once over ANY token record: post.cpp runs them over Token (the general path: any number of chunks and analyses), post_fast.hpp over
// the packed FlatToken records of a result segment (one chunk, one analysis: the common case).  `strOf(token)` yields the token's form as a
// std::u16string_view.  Reference functions: fillPairedTokenInfo src/Kiwi.cpp:98-143, SentenceParser :145-312, fillSentLineInfo :325-415.
#pragma once
#include <cstdint>
#include <string_view>
#include <utility>
#include <vector>
#include "flat_model.hpp"
#include "hostutil.hpp"
#include "kchars.hpp"

namespace kamd
{
	namespace postc
	{
		inline uint32_t getSSType(char16_t c) // src/Utils.cpp:185-262: bracket family id, open/close share an id
		{
			static const char16_t pairs[][2] = {
				{'(', ')'}, {'<', '>'}, {'[', ']'}, {'{', '}'}, {0x2018, 0x2019}, {0x201c, 0x201d}, {0x226a, 0x226b}, {0x3008, 0x3009},
				{0x300a, 0x300b}, {0x300c, 0x300d}, {0x300e, 0x300f}, {0x3010, 0x3011}, {0x3014, 0x3015}, {0x3016, 0x3017}, {0x3018, 0x3019},
				{0x301a, 0x301b}, {0xff08, 0xff09}, {0xff1c, 0xff1e}, {0xff3b, 0xff3d}, {0xff5b, 0xff5d}, {0xff5f, 0xff60}, {0xff62, 0xff63} };
			if (c == '\'') return 1;
			if (c == '"') return 2;
			for (uint32_t i = 0; i < sizeof(pairs) / sizeof(pairs[0]); ++i) if (c == pairs[i][0] || c == pairs[i][1]) return 3 + i;
			return 0;
		}

		inline char16_t toCompatibleConsonant(char16_t c) // src/Utils.cpp toCompatibleHangulConsonant: conjoining jamo -> compatibility jamo
		{
			static const char16_t onset[19] = { 0x3131, 0x3132, 0x3134, 0x3137, 0x3138, 0x3139, 0x3141, 0x3142, 0x3143, 0x3145, 0x3146, 0x3147, 0x3148, 0x3149, 0x314A, 0x314B, 0x314C, 0x314D, 0x314E };
			static const char16_t coda[27] = { 0x3131, 0x3132, 0x3133, 0x3134, 0x3135, 0x3136, 0x3137, 0x3139, 0x313A, 0x313B, 0x313C, 0x313D, 0x313E, 0x313F, 0x3140,
				0x3141, 0x3142, 0x3144, 0x3145, 0x3146, 0x3147, 0x3148, 0x314A, 0x314B, 0x314C, 0x314D, 0x314E };
			if (0x1100 <= c && c < 0x1100 + 19) return onset[c - 0x1100];
			if (0x11A8 <= c && c < 0x11A8 + 27) return coda[c - 0x11A8];
			return c;
		}

		// (the two stacks live as long as their thread: a host worker assembles thousands of texts)
		template<class Tok, class StrOf> void fillPaired(Tok* tokens, size_t n, StrOf&& strOf)
		{
			thread_local std::vector<std::pair<uint32_t, uint32_t>> pStack, bStack;
			pStack.clear(); bStack.clear();
			for (uint32_t i = 0; i < n; ++i)
			{
				Tok& t = tokens[i];
				if (t.tag == T_SSO)
				{
					const uint32_t type = getSSType(strOf(t)[0]);
					if (type) pStack.emplace_back(i, type);
				}
				else if (t.tag == T_SSC)
				{
					const uint32_t type = getSSType(strOf(t)[0]);
					if (!type) continue;
					for (size_t j = pStack.size(); j-- > 0;)
					{
						if (pStack[j].second != type) continue;
						t.pairedToken = pStack[j].first;
						tokens[pStack[j].first].pairedToken = i;
						pStack.resize(j);
						break;
					}
				}
				else if (t.tag == T_SB)
				{
					const uint32_t type = getSBType(strOf(t));
					if (!type) continue;
					for (size_t j = bStack.size(); j-- > 0;)
					{
						if (bStack[j].second != type) continue;
						tokens[bStack[j].first].pairedToken = i;
						bStack.resize(j);
						break;
					}
					bStack.emplace_back(i, type);
				}
			}
		}

		// Sentence boundary automaton (Kiwi.cpp:145-312): EF (요)? (z_coda)? trailing-symbols* | SF trailing-symbols*
		struct SentenceParser
		{
			enum { NONE, EF, EFJX, ZCODA, SF } state = NONE;
			size_t lastPosition = 0, lastLine = 0;
			const FlatModel* mdl;
			explicit SentenceParser(const FlatModel* m) : mdl(m) {}

			bool isYoMorph(int32_t morph) const
			{
				if (morph < 0) return false;
				const FormRec& f = mdl->forms[mdl->morphKform[morph]];
				return f.len == 1 && mdl->formChars[f.charOff] == 0xC694;
			}
			template<class Tok> bool isYo(const Tok& t) const { return isYoMorph(t.morph); }

			// one token: its tag, morpheme, position and length; `end` = true for the step past the last token (a token of tag 0)
			bool step(uint8_t tag, int32_t morph, size_t position, size_t length, size_t line, bool force = false)
			{
				bool ret = false;
				if (force) { state = NONE; lastPosition = position + length; return true; }
				auto closeOrBreak = [&](bool breakOnSameLineSso) // default branch shared by three states
				{
					if (tag == T_SSO && breakOnSameLineSso && line == lastLine) return;
					ret = true; state = NONE;
				};
				switch (state)
				{
				case NONE:
					if (tag == T_EF) state = EF; else if (tag == T_SF) state = SF;
					break;
				case EF:
					if (tag == T_VX) { state = NONE; break; }
					// fallthrough
				case EFJX:
					if (tag == T_Z_CODA) state = ZCODA;
					else if (isJClass(tag) || tag == T_VCP || tag == T_ETM || tag == T_EC)
					{
						if (tag == T_JX && isYoMorph(morph))
						{
							if (state == EF) state = EFJX; else { ret = true; state = NONE; }
						}
						else state = NONE;
					}
					else if (tag == T_SO || tag == T_SW || tag == T_SH || tag == T_SP || tag == T_SE || tag == T_SSC) {}
					else if (tag == T_SF) state = SF;
					else closeOrBreak(true);
					break;
				case ZCODA:
					if (tag == T_SO || tag == T_SW || tag == T_SH || tag == T_SP || tag == T_SE || tag == T_SF || tag == T_SSC) {}
					else closeOrBreak(true);
					break;
				case SF:
					if (tag == T_SO || tag == T_SW || tag == T_SH || tag == T_SE || tag == T_SP || tag == T_SSC) {}
					else if (tag == T_SSO) { if (line != lastLine) { ret = true; state = NONE; } }
					else if ((tag == T_SL || tag == T_SN) && lastPosition == position) state = NONE;
					else { ret = true; state = NONE; }
					break;
				}
				lastPosition = position + length;
				lastLine = line;
				return ret;
			}
			template<class Tok> bool next(const Tok& t, size_t line, bool force = false) { return step(t.tag, t.morph, t.position, t.length, line, force); }
			bool nextEnd(size_t line) { return step(0, -1, 0, 0, line); }      // (the reference feeds a default-constructed token behind the last one)
		};

		template<class Tok> bool hasSentences(const FlatModel* m, const Tok* first, const Tok* last)
		{
			SentenceParser sp{ m };
			for (; first != last; ++first) if (sp.next(*first, 0)) return true;
			return sp.nextEnd(0);
		}

		inline bool nestedLeft(uint8_t tag) { return isJClass(tag) || (isEClass(tag) && tag != T_EF) || tag == T_SP; }
		inline bool nestedRight(uint8_t tag, std::u16string_view str)
		{
			return isJClass(tag) || isEClass(tag) || (isVerbClass(tag) && str.size() == 1 && str[0] == 0xD558) || tag == T_VCP || tag == T_SP;
		}

		// Sentence, sub-sentence, line and word numbers of the tokens of one analysis (what the reference's fillSentLineInfo leaves in them,
		// src/Kiwi.cpp:325-415), as three passes over arrays instead of one loop over interleaved counters:
		//   1. line of every token        -- a merge of the (sorted) newline offsets with the token positions;
		//   2. sentence / sub-sentence    -- the boundary automaton: SentenceParser decides where a sentence ends; a bracketed span either hides its
		//                                    inside from it (no sentence in there) or numbers the sentences inside as sub-sentences; a gap of more
		//                                    than one line starts a sentence too.  A boundary may claim the symbol glued to the front of the token that
		//                                    opens the next sentence: such late claims are collected and applied after the pass;
		//   3. word index inside a sentence -- a running count of the changes of the tokens' original word index, restarted per sentence.
		// Passes 1 and 3 are prefix scans; pass 2 carries the parser's state from token to token.
		struct LateClaim { size_t token; uint32_t value; bool sentence; };

		template<class Tok, class StrOf> void fillSentLine(const FlatModel* m, Tok* tokens, size_t n, const std::vector<size_t>& newlines, StrOf&& strOf)
		{
			if (!n) return;
			// (three per-token arrays in one scratch block that lives as long as its thread: a host worker assembles thousands of texts)
			thread_local std::vector<uint32_t> scratch;
			scratch.assign(3 * n, 0u);
			uint32_t* const line = scratch.data(); uint32_t* const sent = line + n; uint32_t* const sub = sent + n;
			auto endOf = [](const Tok& t) { return (uint32_t)(t.position + t.length); };

			// ---- 1: lines ----
			{
				size_t seen = 0;
				for (size_t i = 0; i < n; ++i)
				{
					while (seen < newlines.size() && newlines[seen] < tokens[i].position) ++seen;
					line[i] = (uint32_t)seen;
				}
			}

			// ---- 2: sentences ----
			enum class Span { None, Opaque, SubSentences };      // what an open bracket pair is to the sentence count
			Span span = Span::None; size_t spanEnd = 0;          // its closing token
			thread_local std::vector<LateClaim> claims;
			claims.clear();
			{
				SentenceParser parser{ m };
				uint32_t curSent = 0, curSub = 0, subsSoFar = 1;      // subsSoFar: sub-sentence number the next bracketed span of this sentence starts at
				auto gluedSymbolBefore = [&](size_t i)      // token i - 1 is a symbol written onto token i, with a gap before it: it opens the new sentence
				{
					if (i < 2) return false;
					const Tok& p = tokens[i - 1];
					const bool symbol = p.tag == T_SO || p.tag == T_SW || p.tag == T_SP || p.tag == T_SE || p.tag == T_SSO;
					return symbol && endOf(p) == tokens[i].position && p.position > endOf(tokens[i - 2]);
				};
				for (size_t i = 0; i < n; ++i)
				{
					const Tok& t = tokens[i];
					const uint32_t sentBefore = curSent;
					const bool hidden = span == Span::Opaque && i < spanEnd;
					const bool closesSubs = span == Span::SubSentences && i == spanEnd;
					if (!hidden && parser.next(t, i ? line[i - 1] : 0, closesSubs))
					{
						if (span == Span::SubSentences)
						{
							++curSub; ++subsSoFar;
							if (gluedSymbolBefore(i)) claims.push_back({ i - 1, curSub, false });
						}
						else
						{
							++curSent; subsSoFar = 1;
							if (gluedSymbolBefore(i)) claims.push_back({ i - 1, curSent, true });
						}
					}
					if (span == Span::None)
					{
						if (t.tag == T_SSO && t.pairedToken != (uint32_t)-1)
						{
							const size_t close = t.pairedToken;
							if (!hasSentences(m, &tokens[i], &tokens[close])) { span = Span::Opaque; spanEnd = close; curSub = 0; }
							else if ((close + 1 < n && nestedRight(tokens[close + 1].tag, strOf(tokens[close + 1]))) || (i > 0 && nestedLeft(tokens[i - 1].tag))) { span = Span::SubSentences; spanEnd = close; curSub = subsSoFar; }
						}
					}
					else if ((span == Span::SubSentences && i > spanEnd) || (span == Span::Opaque && i >= spanEnd)) { span = Span::None; spanEnd = 0; curSub = 0; }

					// (a span that closes at token 0 cannot exist, so "no span" and "span ending at 0" coincide as in the reference's counters)
					const size_t subsEnd = span == Span::SubSentences ? spanEnd : 0;
					if (line[i] > (i ? line[i - 1] : 0) + 1 && curSent == sentBefore && span != Span::SubSentences) ++curSent;      // an empty line in between
					sent[i] = curSent;
					sub[i] = (i == subsEnd || i == tokens[subsEnd].pairedToken) ? 0 : curSub;
					if (curSent != (i ? sent[i - 1] : 0)) subsSoFar = 1;
				}
			}

			// ---- 3: word indices, then everything into the tokens ----
			{
				uint32_t word = 0, lastOriginal = 0;
				for (size_t i = 0; i < n; ++i)
				{
					Tok& t = tokens[i];
					if (sent[i] != (i ? sent[i - 1] : 0)) word = 0;
					else if (t.wordPosition != lastOriginal) ++word;
					lastOriginal = t.wordPosition;
					t.wordPosition = word; t.sentPosition = sent[i]; t.subSentPosition = sub[i]; t.lineNumber = line[i];
				}
				for (const LateClaim& c : claims)
				{
					if (c.sentence) { tokens[c.token].sentPosition = c.value; tokens[c.token].wordPosition = 0; }
					else tokens[c.token].subSentPosition = c.value;
				}
			}
		}

		// allNewLinePositions (Kiwi.cpp:70-96)
		inline void newLinePositions(const char16_t* raw, size_t n, std::vector<size_t>& newlines)
		{
			newlines.clear();
			bool isCR = false;
			for (size_t i = 0; i < n; ++i)
			{
				switch (raw[i])
				{
				case 0x0D: isCR = true; newlines.push_back(i); break;
				case 0x0A: if (!isCR) newlines.push_back(i); isCR = false; break;
				case 0x0B: case 0x0C: case 0x85: case 0x2028: case 0x2029: isCR = false; newlines.push_back(i); break;
				}
			}
		}
	}
}
