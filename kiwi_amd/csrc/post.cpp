// Host-side result assembly (see post.hpp for the reference functions each part reproduces).
#include <algorithm>
#include <numeric>
#include <stdexcept>
#include "post.hpp"
#include <string_view>
#include "feature.hpp"
#include "textprep.hpp"
#include "post_common.hpp"

namespace kamd
{
	namespace
	{
		inline std::u16string_view strOfToken(const Token& t) { return t.str; }

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
			postc::SentenceParser yo{ m };
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
				if (match & M_COMPATIBLE_JAMO) for (auto& c : joined) c = postc::toCompatibleConsonant(c);
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
		postc::newLinePositions(raw, n, newlines);
		for (auto& r : ret) { postc::fillPaired(r.first.data(), r.first.size(), strOfToken); postc::fillSentLine(&mdl, r.first.data(), r.first.size(), newlines, strOfToken); }
		if (ret.empty()) ret.emplace_back();
		return std::move(ret);
	}
}
