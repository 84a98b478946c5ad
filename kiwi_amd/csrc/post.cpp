// Host-side result assembly (see post.hpp for the reference functions each part reproduces).
#include <algorithm>
#include <numeric>
#include <stdexcept>
#include "post.hpp"
#include <string_view>
#include "feature.hpp"
#include "textprep.hpp"

namespace kamd
{
	namespace
	{
		uint32_t getSSType(char16_t c) // src/Utils.cpp:185-262: bracket family id, open/close share an id
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

		char16_t toCompatibleConsonant(char16_t c);

		void fillPaired(std::vector<Token>& tokens)
		{
			std::vector<std::pair<uint32_t, uint32_t>> pStack, bStack;
			for (uint32_t i = 0; i < tokens.size(); ++i)
			{
				Token& t = tokens[i];
				if (t.tag == T_SSO)
				{
					const uint32_t type = getSSType(t.str[0]);
					if (type) pStack.emplace_back(i, type);
				}
				else if (t.tag == T_SSC)
				{
					const uint32_t type = getSSType(t.str[0]);
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
					const uint32_t type = getSBType(t.str);
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

			bool isYo(const Token& t) const
			{
				if (t.morph < 0) return false;
				const U16 kf = mdl->formStr(mdl->morphKform[t.morph]);
				return kf.size() == 1 && kf[0] == 0xC694;
			}

			bool next(const Token& t, size_t line, bool force = false)
			{
				bool ret = false;
				if (force) { state = NONE; lastPosition = t.position + t.length; return true; }
				auto closeOrBreak = [&](bool breakOnSameLineSso) // default branch shared by three states
				{
					if (t.tag == T_SSO && breakOnSameLineSso && line == lastLine) return;
					ret = true; state = NONE;
				};
				switch (state)
				{
				case NONE:
					if (t.tag == T_EF) state = EF; else if (t.tag == T_SF) state = SF;
					break;
				case EF:
					if (t.tag == T_VX) { state = NONE; break; }
					// fallthrough
				case EFJX:
					if (t.tag == T_Z_CODA) state = ZCODA;
					else if (isJClass(t.tag) || t.tag == T_VCP || t.tag == T_ETM || t.tag == T_EC)
					{
						if (t.tag == T_JX && isYo(t))
						{
							if (state == EF) state = EFJX; else { ret = true; state = NONE; }
						}
						else state = NONE;
					}
					else if (t.tag == T_SO || t.tag == T_SW || t.tag == T_SH || t.tag == T_SP || t.tag == T_SE || t.tag == T_SSC) {}
					else if (t.tag == T_SF) state = SF;
					else closeOrBreak(true);
					break;
				case ZCODA:
					if (t.tag == T_SO || t.tag == T_SW || t.tag == T_SH || t.tag == T_SP || t.tag == T_SE || t.tag == T_SF || t.tag == T_SSC) {}
					else closeOrBreak(true);
					break;
				case SF:
					if (t.tag == T_SO || t.tag == T_SW || t.tag == T_SH || t.tag == T_SE || t.tag == T_SP || t.tag == T_SSC) {}
					else if (t.tag == T_SSO) { if (line != lastLine) { ret = true; state = NONE; } }
					else if ((t.tag == T_SL || t.tag == T_SN) && lastPosition == t.position) state = NONE;
					else { ret = true; state = NONE; }
					break;
				}
				lastPosition = t.position + t.length;
				lastLine = line;
				return ret;
			}
		};

		bool hasSentences(const FlatModel* m, const Token* first, const Token* last)
		{
			SentenceParser sp{ m };
			for (; first != last; ++first) if (sp.next(*first, 0)) return true;
			return sp.next(Token{}, 0);
		}

		bool nestedLeft(const Token& t) { return isJClass(t.tag) || (isEClass(t.tag) && t.tag != T_EF) || t.tag == T_SP; }
		bool nestedRight(const Token& t)
		{
			return isJClass(t.tag) || isEClass(t.tag) || (isVerbClass(t.tag) && t.str.size() == 1 && t.str[0] == 0xD558) || t.tag == T_VCP || t.tag == T_SP;
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

		void fillSentLine(const FlatModel* m, std::vector<Token>& tokens, const std::vector<size_t>& newlines)
		{
			const size_t n = tokens.size();
			if (!n) return;
			// (three per-token arrays in one scratch block that lives as long as its thread: a host worker assembles thousands of texts)
			thread_local std::vector<uint32_t> scratch;
			scratch.assign(3 * n, 0u);
			uint32_t* const line = scratch.data(); uint32_t* const sent = line + n; uint32_t* const sub = sent + n;

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
			std::vector<LateClaim> claims;
			{
				SentenceParser parser{ m };
				uint32_t curSent = 0, curSub = 0, subsSoFar = 1;      // subsSoFar: sub-sentence number the next bracketed span of this sentence starts at
				auto gluedSymbolBefore = [&](size_t i)      // token i - 1 is a symbol written onto token i, with a gap before it: it opens the new sentence
				{
					if (i < 2) return false;
					const Token& p = tokens[i - 1];
					const bool symbol = p.tag == T_SO || p.tag == T_SW || p.tag == T_SP || p.tag == T_SE || p.tag == T_SSO;
					return symbol && p.endPos() == tokens[i].position && p.position > tokens[i - 2].endPos();
				};
				for (size_t i = 0; i < n; ++i)
				{
					const Token& t = tokens[i];
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
							else if ((close + 1 < n && nestedRight(tokens[close + 1])) || (i > 0 && nestedLeft(tokens[i - 1]))) { span = Span::SubSentences; spanEnd = close; curSub = subsSoFar; }
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
					Token& t = tokens[i];
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

		void concatTokens(Token& dest, const Token& src, uint8_t tag)
		{
			dest.tag = tag; dest.morph = -1;
			dest.length = (uint16_t)(src.position + src.length - dest.position);
			dest.str += src.str;
		}

		size_t joinAffix(const FlatModel* m, std::vector<Token>& v, uint64_t mo)
		{
			constexpr uint64_t M_JOIN_PARTICLE_YO = 1ull << 27;
			if (!(mo & (M_JOIN_NOUN_PREFIX | M_JOIN_NOUN_SUFFIX | M_JOIN_VERB_SUFFIX | M_JOIN_ADJ_SUFFIX | M_JOIN_ADV_SUFFIX | M_MERGE_SAISIOT | M_JOIN_PARTICLE_YO))) return v.size();
			if (v.size() < 2) return v.size();
			size_t first = 0, next = 1;
			const size_t last = v.size();
			SentenceParser yo{ m };
			while (next != last)
			{
				Token& cur = v[first];
				Token& nx = v[next];
				if ((mo & M_JOIN_NOUN_PREFIX) && cur.tag == T_XPN && (isNNClass(nx.tag) || nx.tag == T_SN)) { concatTokens(cur, nx, nx.tag); ++next; }
				else if ((mo & M_JOIN_NOUN_SUFFIX) && nx.tag == T_XSN && (isNNClass(cur.tag) || cur.tag == T_SN)) { concatTokens(cur, nx, cur.tag); ++next; }
				else if ((mo & M_JOIN_VERB_SUFFIX) && clearIrregular(nx.tag) == T_XSV && (isNNClass(cur.tag) || cur.tag == T_XR)) { concatTokens(cur, nx, (uint8_t)(T_VV | (nx.tag & 0x80))); ++next; }
				else if ((mo & M_JOIN_ADJ_SUFFIX) && clearIrregular(nx.tag) == T_XSA && (isNNClass(cur.tag) || cur.tag == T_XR)) { concatTokens(cur, nx, (uint8_t)(T_VA | (nx.tag & 0x80))); ++next; }
				else if ((mo & M_JOIN_ADV_SUFFIX) && nx.tag == T_XSM && (isNNClass(cur.tag) || cur.tag == T_XR)) { concatTokens(cur, nx, T_MAG); ++next; }
				else if ((mo & M_MERGE_SAISIOT) && nx.tag == T_Z_SIOT && isNNClass(cur.tag) && next + 1 != last && isNNClass(v[next + 1].tag))
				{
					cur.str.back() += (0x11BA - 0x11A7);
					concatTokens(cur, v[next + 1], T_NNG);
					next += 2;
				}
				else if ((mo & M_JOIN_PARTICLE_YO) && nx.tag == T_JX && yo.isYo(nx) && (cur.tag == T_EC || cur.tag == T_EF)) { concatTokens(cur, nx, cur.tag); ++next; }
				else
				{
					++first;
					if (first != next) v[first] = std::move(v[next]);
					++next;
				}
			}
			return first + 1;
		}

		char16_t toCompatibleConsonant(char16_t c) // src/Utils.cpp toCompatibleHangulConsonant: conjoining jamo -> compatibility jamo
		{
			static const char16_t onset[19] = { 0x3131, 0x3132, 0x3134, 0x3137, 0x3138, 0x3139, 0x3141, 0x3142, 0x3143, 0x3145, 0x3146, 0x3147, 0x3148, 0x3149, 0x314A, 0x314B, 0x314C, 0x314D, 0x314E };
			static const char16_t coda[27] = { 0x3131, 0x3132, 0x3133, 0x3134, 0x3135, 0x3136, 0x3137, 0x3139, 0x313A, 0x313B, 0x313C, 0x313D, 0x313E, 0x313F, 0x3140,
				0x3141, 0x3142, 0x3144, 0x3145, 0x3146, 0x3147, 0x3148, 0x314A, 0x314B, 0x314C, 0x314D, 0x314E };
			if (0x1100 <= c && c < 0x1100 + 19) return onset[c - 0x1100];
			if (0x11A8 <= c && c < 0x11A8 + 27) return coda[c - 0x11A8];
			return c;
		}
	}

	void ResultSegment::appendText(const std::vector<TokenResult>& analyses)
	{
		for (const auto& a : analyses)
		{
			for (const Token& t : a.first)
			{
				FlatToken o{};
				o.position = t.position; o.wordPosition = t.wordPosition; o.sentPosition = t.sentPosition; o.lineNumber = t.lineNumber;
				o.length = t.length; o.tag = t.tag; o.senseOrScript = t.senseId; o.score = t.score; o.typoCost = t.typoCost;
				o.typoFormId = t.typoFormId; o.pairedToken = t.pairedToken; o.subSentPosition = t.subSentPosition; o.dialect = t.dialect;
				o.morph = t.morph; o.formLen = (uint16_t)t.str.size(); o.formOff = forms.size();
				forms.insert(forms.end(), t.str.begin(), t.str.end());
				forms.push_back(0);
				toks.push_back(o);
			}
			anaScore.push_back(a.second);
			anaTok.push_back((uint32_t)toks.size());
		}
		textAna.push_back((uint32_t)anaScore.size());
	}

	void ResultBuilder::begin(const char16_t* raw, size_t n, const uint32_t* pt, size_t ptLen)
	{
		ret.clear(); spStatesByRet.clear();
		positionTable = pt; positionLen = ptLen;
		// getWordPositions (Kiwi.cpp:465-487)
		wordPositions.resize(n);
		uint32_t position = 0; bool contSpace = false;
		for (size_t i = 0; i < n; ++i)
		{
			wordPositions[i] = (uint16_t)position;
			if (isSpace(raw[i])) { if (!contSpace) ++position; contSpace = true; }
			else contSpace = false;
		}
	}

	void ResultBuilder::insertPaths(const std::vector<PathResult>& pathes)
	{
		parentMap.clear();
		if (ret.empty())
		{
			const size_t n = std::min(pathes.size(), topN * 2);
			ret.resize(n); spStatesByRet.resize(n); parentMap.resize(n);
			std::iota(parentMap.begin(), parentMap.end(), 0);
		}
		else
		{
			uint32_t prevParents[256] = { 0 };
			std::vector<uint8_t> selected(pathes.size());
			const size_t nRet = ret.size();
			for (size_t i = 0; i < nRet; ++i)
			{
				const uint8_t st = spStatesByRet[i];
				auto findFrom = [&](size_t from) { size_t p = from; while (p < pathes.size() && pathes[p].prevState != st) ++p; return p; };
				size_t parent = findFrom(prevParents[st]);
				if (parent >= pathes.size() && prevParents[st]) parent = findFrom(0);
				parentMap.push_back(parent);
				if (parent < pathes.size()) { selected[parent] = 1; prevParents[st] = (uint32_t)parent + 1; }
			}
			for (size_t i = 0; i < pathes.size(); ++i)
			{
				if (selected[i]) continue;
				const size_t parent = std::find(spStatesByRet.begin(), spStatesByRet.end(), pathes[i].prevState) - spStatesByRet.begin();
				if (parent >= ret.size()) throw std::runtime_error{ "result merge: path with no matching predecessor state" };
				ret.push_back(ret[parent]);
				spStatesByRet.push_back(spStatesByRet[parent]);
				parentMap.push_back(i);
			}
		}

		uint32_t spStateCnt[256] = { 0 };
		size_t valid = 0;
		const uint32_t* ptBegin = positionTable; const uint32_t* ptEnd = positionTable + positionLen;
		for (size_t i = 0; i < ret.size(); ++i)
		{
			if (!(parentMap[i] < pathes.size() && spStateCnt[pathes[parentMap[i]].curState] < topN)) continue;
			if (valid != i) ret[valid] = std::move(ret[i]);
			const PathResult& r = pathes[parentMap[i]];
			auto& rarr = ret[valid].first;
			const size_t firstNew = rarr.size();
			rarr.reserve(rarr.size() + r.path.size());
			int32_t prevMorph = -1;
			for (auto& s : r.path)
			{
				if (!s.str.empty() && s.str[0] == u' ') continue;
				const MorphRec& mr = mdl.morphs[s.morph];
				// (the morpheme's form as a view into the model's character array: no copy per token)
				const FormRec& kfr = mdl.forms[mdl.morphKform[s.morph]];
				const std::u16string_view kform{ (const char16_t*)mdl.formChars.data() + kfr.charOff, kfr.len };
				U16 joined;
				bool done = false;
				if (!integrateAllomorph && T_EP <= mr.tag && mr.tag <= T_ETM && !kform.empty() && kform[0] == 0xC5B4)
				{
					U16 pk; if (prevMorph >= 0) pk = mdl.formStr(mdl.morphKform[prevMorph]);
					if (prevMorph >= 0 && !pk.empty() && pk.back() == 0xD558) { joined = joinHangul(U16(1, (char16_t)0xC5EC) + U16(kform.substr(1))); done = true; }
					else if (matchPolar((const uint16_t*)pk.data(), (uint32_t)pk.size(), CP_POSITIVE)) { joined = joinHangul(U16(1, (char16_t)0xC544) + U16(kform.substr(1))); done = true; }
				}
				if (!done) joined = s.str.empty() ? joinHangul(kform.data(), kform.size()) : joinHangul(s.str);
				if (match & M_COMPATIBLE_JAMO) for (auto& c : joined) c = toCompatibleConsonant(c);
				rarr.emplace_back();
				Token& tk = rarr.back();
				tk.str = std::move(joined); tk.tag = mr.tag; tk.morph = (int32_t)s.morph;
				const size_t b = (std::upper_bound(ptBegin, ptEnd, s.begin) - ptBegin) - 1;
				const size_t e = std::lower_bound(ptBegin, ptEnd, s.end) - ptBegin;
				tk.position = (uint32_t)b; tk.length = (uint16_t)(e - b);
				tk.score = s.wordScore; tk.typoCost = s.typoCost; tk.typoFormId = s.typoFormId;
				tk.senseId = mr.senseId;
				if ((mr.tag == T_NNG || mr.tag == T_NNP) && !s.str.empty()) tk.senseId = 0xFF;
				// updateTokenInfoScript (Kiwi.cpp:590-605)
				if ((tk.tag == T_SL || tk.tag == T_SH || tk.tag == T_SW || tk.tag == T_W_EMOJI) && kform.empty() && !tk.str.empty())
				{
					uint32_t c = tk.str[0];
					if (isHighSurrogate(c)) c = mergeSurrogate(c, tk.str.size() > 1 ? tk.str[1] : 0);
					tk.senseId = chr2ScriptType(c);
					if (tk.senseId == 1 /* latin */) tk.tag = T_SL;
				}
				tk.dialect = mdl.morphDialect.empty() ? (uint16_t)0 : mdl.morphDialect[s.morph];      // Kiwi.cpp:747
				tk.wordPosition = wordPositions[tk.position];
				prevMorph = (int32_t)s.morph;
			}
			(void)firstNew;
			rarr.resize(joinAffix(&mdl, rarr, match));
			ret[valid].second += r.score;
			spStatesByRet[valid] = r.curState;
			spStateCnt[r.curState]++;
			valid++;
		}
		if (valid <= 1)      // (one analysis carried on -- every top-1 text of one chunk: nothing to order, nothing to cut)
		{
			ret.resize(valid); spStatesByRet.resize(valid);
			return;
		}
		std::vector<size_t> idx(valid);
		std::iota(idx.begin(), idx.end(), 0);
		std::sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return ret[a].second > ret[b].second; });
		const size_t maxCands = std::min(topN * 2, valid);
		std::vector<TokenResult> sorted; std::vector<uint8_t> sortedSt;
		for (size_t i = 0; i < maxCands; ++i) { sorted.emplace_back(std::move(ret[idx[i]])); sortedSt.push_back(spStatesByRet[idx[i]]); }
		ret = std::move(sorted); spStatesByRet = std::move(sortedSt);
	}

	std::vector<TokenResult> ResultBuilder::finish(const char16_t* raw, size_t n)
	{
		std::sort(ret.begin(), ret.end(), [](const TokenResult& a, const TokenResult& b) { return a.second > b.second; });
		if (ret.size() > topN) ret.erase(ret.begin() + topN, ret.end());
		std::vector<size_t> newlines;
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
		for (auto& r : ret) { fillPaired(r.first); fillSentLine(&mdl, r.first, newlines); }
		if (ret.empty()) ret.emplace_back();
		return std::move(ret);
	}
}
