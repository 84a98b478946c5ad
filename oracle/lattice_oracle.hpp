// TEST INFRASTRUCTURE ONLY -- never linked into the product (tests/, bench.py's cpu_baseline leg and
// __graft_entry__.smoke() are the only users).
//
// CPU restatement of the reference's lattice construction for one chunk, typo-free path:
//   Splitter::{buildTypoGraph, search, progressNode, flushCandidates, insertUnkForm, hasFormAlready,
//   isZFollowable, writeResult}   /root/reference/src/KTrie.cpp:873-1464
//   appendNewNode / removeUnconnected / countSpaceErrors   /root/reference/src/KTrie.cpp:15-43, 240-299, 316-328
// Written against the flat (index-based) model; sequential, one chunk at a time.  Parity of this
// restatement is pinned against the real reference TUs through oracle/_ref (tests/test_oracle_vs_ref.py).
#pragma once
#include <algorithm>
#include <cstdint>
#include <deque>
#include <vector>
#include "../kiwi_amd/csrc/flat_model.hpp"
#include "../kiwi_amd/csrc/textprep.hpp"

namespace korc
{
	using namespace kamd;
	constexpr uint32_t NOFORM = 0xFFFFFFFFu;

	struct LNode
	{
		uint32_t form = NOFORM;
		uint32_t prev = 0, sibling = 0;
		uint32_t startPos = 0, endPos = 0;   // ns positions while building, string offsets after finish()
		uint32_t uformOff = 0, uformLen = 0; // chunk-relative
		uint32_t spaceErrors = 0;
		float typoCost = 0;
	};

	struct Counters   // ALG_BYTES v1 events
	{
		uint64_t inputUnits = 0, trieProbes = 0, trieProbeKeyBytes = 0, failHops = 0, candEmits = 0, otherNodes = 0;
		uint64_t transitions = 0, candMorphs = 0, statesWritten = 0, lmProbes = 0, lmProbeKeyBytes = 0, lmRootProbes = 0, tokens = 0;
		uint64_t maxPrevPaths = 0, nodesOver128 = 0, nodesOver512 = 0, lattNodes = 0;
		// SkipBigram extra (SURVEY.md §8(d) "SBG extra"): evaluate() calls, key bytes of their 8 partner searches, hits; sbgModel = the model has the tables
		uint64_t sbgEvals = 0, sbgProbeKeyBytes = 0, sbgHits = 0, sbgModel = 0;
		// CoNgram (SURVEY.md §8(d) "CoNgram"): embedding rows gathered -- per candidate evaluation the UNIQUE context / output rows of its score
		// matrix, one pair per later chunk step --, scores written, context-trie probes (non-root levels visited) with their key bytes, root
		// probes; congDim = the embedding dimension (0: not a CoNgram model)
		uint64_t congCtxRows = 0, congOutRows = 0, congScores = 0, congProbes = 0, congProbeKeyBytes = 0, congRootProbes = 0, congDim = 0;
		uint64_t congGlobalScores = 0;      // of congScores: mixtures over the history window (global model, valid distant tokens)
		// typo lattices (round 5; the plain counters above count a typo lattice's events too): graph nodes visited (28-byte records), search-state transitions
		// (progressNode calls: the 44-byte fixed part of a state read, and written for every state that lives on)
		uint64_t typoGraphNodes = 0, typoStateSteps = 0, typoStatesKept = 0;
		uint64_t congPast64 = 0;            // global model: insertions into a path container that already holds 64 entries (the reference's SIMD lookup misbehaves there)
	};

	struct SplitConfig { uint64_t match; uint32_t maxUnk, maxUnkJ, spaceTol; };
#ifdef KORC_HAZARD_STATS
	// developer statistics (oracle/_build/liboracle_haz.so only): how often the lattice build meets the cases that make it order-dependent
	struct HazStats { uint64_t chunks = 0, chunksHaz = 0, live = 0, h1enter = 0, h1act = 0, h2 = 0, h3 = 0, h4 = 0, ops = 0, appFailReach = 0, steps = 0, maxCand = 0; bool cur = false; };
	inline HazStats& hazStats() { static HazStats h; return h; }
#define KORC_HAZ(x) x
#else
#define KORC_HAZ(x)
#endif

	class LatticeBuilder
	{
		const ModelView& M;
		const SplitConfig& cfg;
		Counters& cnt;
		const char16_t* str; uint32_t n;
		const uint8_t* cls; const uint8_t* script;
		std::vector<uint32_t> nsToPos, posToNs;
		std::vector<std::pair<uint32_t, uint32_t>> endPosMap;
		std::vector<LNode> out;

		static uint32_t log2c(uint32_t v) { uint32_t l = 0; while ((1u << l) < v + 1) ++l; return l; }

		bool append(uint32_t s, uint32_t e, uint32_t form, uint32_t uOff, uint32_t uLen)
		{
			if (endPosMap[s].first == endPosMap[s].second) return false;
			const uint32_t id = (uint32_t)out.size();
			LNode nn; nn.startPos = s; nn.endPos = e; nn.form = form; nn.uformOff = uOff; nn.uformLen = uLen;
			nn.prev = id - endPosMap[s].first;
			out.push_back(nn);
			if (e >= endPosMap.size()) return true;
			auto& m = endPosMap[e];
			if (m.first == m.second) { m.first = id; m.second = id + 1; }
			else { out[m.second - 1].sibling = id - (m.second - 1); m.second = id + 1; }
			return true;
		}

		uint32_t nodeLen(const LNode& g) const
		{
			if (g.uformLen) return g.uformLen;
			const FormRec& f = M.forms[g.form];
			return f.len - f.numSpaces;
		}

		bool hasFormAlready(uint32_t s, uint32_t e) const
		{
			const uint32_t a = std::max(endPosMap[e].first, 1u), b = endPosMap[e].second;
			for (uint32_t i = a; i < b; ++i)
			{
				const LNode& g = out[i];
				if (g.endPos == e && g.endPos - nodeLen(g) == s && g.typoCost == 0 && (g.form == NOFORM || (M.forms[g.form].flags & FF_HAS_ANY_FULL))) return true;
			}
			return false;
		}

		void trimmed(uint32_t off, uint32_t len, uint32_t& oOff, uint32_t& oLen) const
		{
			while (len && isSpace(str[off + len - 1])) --len;
			oOff = off; oLen = len;
		}

		void insertUnk(uint32_t s, uint32_t e, bool hasJ)
		{
			if (s >= e || hasFormAlready(s, e)) return;
			KORC_HAZ(hazStats().live++; if (e - s > 64) { hazStats().h3++; hazStats().cur = true; })
			uint32_t lastPos = out.back().endPos;
			if (lastPos < e)
			{
				KORC_HAZ(hazStats().h1enter++;)
				if (lastPos && isHangulCoda(str[nsToPos[lastPos]])) lastPos--;
				if (lastPos != s && !hasFormAlready(lastPos, e))
				{
					KORC_HAZ(hazStats().h1act++; hazStats().cur = true;)
					uint32_t o, l; trimmed(nsToPos[lastPos], nsToPos[e - 1] + 1 - nsToPos[lastPos], o, l);
					if (append(lastPos, e, NOFORM, o, l)) cnt.otherNodes++;
				}
			}
			const uint32_t limit = hasJ ? cfg.maxUnkJ : cfg.maxUnk;
			if (e - s <= limit)
			{
				uint32_t o, l; trimmed(nsToPos[s], nsToPos[e - 1] + 1 - nsToPos[s], o, l);
				KORC_HAZ(if (l != e - s) { hazStats().h4++; hazStats().cur = true; })
				if (append(s, e, NOFORM, o, l)) cnt.otherNodes++;
				KORC_HAZ(else { hazStats().h2++; hazStats().cur = true; })
			}
		}

		void unkPair(uint32_t boundary, uint32_t unkStart, uint32_t e, bool hasJ)
		{
			if (boundary < unkStart) insertUnk(boundary, e, hasJ);
			insertUnk(unkStart, e, hasJ);
		}

		uint32_t spaceErrors(const FormRec& f, uint32_t b, uint32_t e) const
		{
			const uint16_t* fs = M.formChars + f.charOff;
			uint32_t nErr = 0, off = 0;
			for (uint32_t i = 1; i < e - b; ++i)
			{
				const bool hasSpace = nsToPos[b + i] - nsToPos[b + i - 1] > 1;
				const uint16_t fc = (i + off < f.len) ? fs[i + off] : 0;
				if (hasSpace && fc != u' ') ++nErr;
				if (fc == u' ') ++off;
			}
			return nErr;
		}

		int32_t trieNext(uint32_t node, uint16_t c)
		{
			cnt.trieProbes++;
			if (node == 0) { cnt.trieProbeKeyBytes += 4; const uint32_t r = M.trieRoot[c]; return r ? (int32_t)r : -1; }
			const TrieNodeRec& t = M.trie[node];
			cnt.trieProbeKeyBytes += 2 * log2c(t.numNexts);
			const uint16_t* kb = M.trieKeys + t.edgeOff;
			const uint16_t* it = std::lower_bound(kb, kb + t.numNexts, c);
			if (it == kb + t.numNexts || *it != c) return -1;
			return (int32_t)M.trieChild[t.edgeOff + (it - kb)];
		}

	public:
		LatticeBuilder(const ModelView& m, const SplitConfig& c, Counters& k) : M(m), cfg(c), cnt(k) {}

		// Builds the lattice of chunk str[0..n).  `patterns` are chunk-relative, sorted.  Returns false if
		// the chunk has no lattice (<= 2 nodes).  Output node positions are offsets into the *text* (startOffset added).
		// a pretokenized span of the chunk (chunk-relative units), the form makePretokenizedSpanGroup gave it, and whether that is a fallback form
		// (one of the default tag forms: the node then carries the text as its own string, KTrie.cpp:1197-1200)
		struct SpanNode { uint32_t begin, end, form; bool fallback; };
		bool build(std::vector<LNode>& ret, const char16_t* s, uint32_t len, const uint8_t* c, const uint8_t* sc,
			const PatternSpan* pat, const PatternSpan* patEnd, uint32_t startOffset, const SpanNode* span = nullptr, const SpanNode* spanEnd = nullptr)
		{
			str = s; n = len; cls = c; script = sc;
			nsToPos.clear(); posToNs.clear(); out.clear();
			for (uint32_t i = 0; i < n; ++i)
			{
				posToNs.push_back((uint32_t)nsToPos.size());
				if (!isSpace(str[i]))
				{
					nsToPos.push_back(i);
					if (isHighSurrogate(str[i]) && i + 1 < n) { posToNs.push_back((uint32_t)nsToPos.size()); nsToPos.push_back(++i); }
				}
			}
			posToNs.push_back((uint32_t)nsToPos.size());
			const uint32_t nNs = (uint32_t)nsToPos.size();
			cnt.inputUnits += n;
			endPosMap.assign(nNs + 1, { 0xFFFFFFFFu, 0xFFFFFFFFu });
			endPosMap[0] = { 0, 1 };
			out.emplace_back();

			uint8_t lastType = T_UNKNOWN, lastScript = 0;
			uint32_t specialStart = 0, unkStart = 0, boundary = 0;
			uint32_t cur = 0;
			std::vector<uint32_t> cands;
			const uint8_t scriptVS = 98; // ScriptType::variation_selectors
			for (uint32_t j = 0; j < n; ++j)
			{
				const uint16_t ch = str[j];
				uint32_t c32 = ch;
				if (isHighSurrogate(c32) && j + 1 < n) c32 = mergeSurrogate(c32, str[j + 1]);
				const bool inPattern = pat != patEnd && j >= pat->end - pat->length;
				uint8_t type = cls[j] & 0x3F, sct = script[j];
				if (lastType == T_SW && (c32 == 0x200d || (0x1f3fb <= c32 && c32 <= 0x1f3ff) || sct == scriptVS)) { type = lastType; sct = lastScript; }
				const uint8_t curT = inPattern ? (uint8_t)T_UNKNOWN : type;
				bool discont;
				{
					auto sym = [](uint8_t t) { return t == T_SL || t == T_SH || t == T_SW; };
					discont = (sym(lastType) && sym(curT)) ? (lastScript != sct) : (lastType != curT);
				}
				if (discont || lastType == T_SSO || lastType == T_SSC)
				{
					if (lastType != T_MAX && lastType != T_UNKNOWN && lastType != T_SS)
					{
						const bool sj = T_SF <= lastType && lastType <= T_SW;
						unkPair(boundary, unkStart, specialStart, sj);
						uint32_t o, l; trimmed(nsToPos[specialStart], j - nsToPos[specialStart], o, l);
						if (append(specialStart, posToNs[j], lastType - 1u, o, l)) cnt.otherNodes++;
					}
					unkStart = specialStart;
					specialStart = posToNs[j];
					if (T_SF <= lastType && lastType <= T_SW) boundary = specialStart;
				}
				else if (type == T_MAX) unkStart = specialStart;
				lastType = curT; lastScript = sct;

				if (c32 < 0x10000)
				{
					if (type == T_UNKNOWN)
					{
						unkPair(boundary, unkStart, posToNs[j + 1], true);
						boundary = specialStart = unkStart = posToNs[j + 1];
						continue;
					}
					// z-coda / saisiot shortcuts (KTrie.cpp:1126-1137)
					bool zc = false, zs = false;
					{
						const uint32_t p = posToNs[j];
						if (p < nNs)
						{
							const uint32_t a = endPosMap[p].first, b = endPosMap[p].second;
							for (uint32_t i = a; i < b; ++i)
							{
								if (out[i].endPos != p || out[i].form == NOFORM) continue;
								zc = zc || (M.forms[out[i].form].flags & FF_ZCODA_APPENDABLE);
								zs = zs || (M.forms[out[i].form].flags & FF_ZSIOT_APPENDABLE);
							}
						}
					}
					if ((cfg.match & M_Z_CODA) && zc && isHangulCoda(ch) && (j + 1 >= n || !isHangulSyllable(str[j + 1])))
						cands.push_back(kDefaultTagSize + (ch - 0x11A8) - 1);
					else if ((cfg.match & (M_SPLIT_SAISIOT | M_MERGE_SAISIOT)) && zs && ch == 0x11BA && j + 1 < n && isHangulSyllable(str[j + 1]))
						cands.push_back(kDefaultTagSize + (0x11BA - 0x11A8) - 1);
				}
				if (pat != patEnd)
				{
					const uint32_t curEnd = j + (c32 >= 0x10000 ? 2 : 1);
					while (pat != patEnd && pat->end == curEnd)
					{
						const uint32_t ms = pat->end - pat->length;
						const bool wj = T_W_URL <= pat->tag && pat->tag <= T_W_EMOJI;
						unkPair(boundary, unkStart, posToNs[ms], wj);
						if (append(posToNs[ms], posToNs[pat->end], pat->tag - 1u, ms, pat->length)) cnt.otherNodes++;
						++pat;
					}
				}
				// a pretokenized span begins here (KTrie.cpp:1177-1210): the pending unknown-form spans are closed, ONE node with the span's form is appended,
				// the walk restarts behind the span
				if (span != spanEnd && span->begin == j)
				{
					unkPair(boundary, unkStart, posToNs[span->begin], false);
					append(posToNs[span->begin], posToNs[span->end], span->form, span->fallback ? span->begin : 0, span->fallback ? span->end - span->begin : 0);
					j += (span->end - span->begin) - 1;
					++span;
					lastType = T_UNKNOWN;
					cur = 0;
					specialStart = unkStart = boundary = posToNs[j + 1];
					continue;
				}
				if (c32 >= 0x10000) { ++j; continue; }

				// Aho-Corasick step (KTrie.cpp:1282-1311)
				int32_t nx = trieNext(cur, ch);
				while (nx < 0)
				{
					const int32_t f = M.trie[cur].fail;
					if (f < 0) break;
					cnt.failHops++;
					cur = (uint32_t)f;
					nx = trieNext(cur, ch);
				}
				if (nx >= 0)
				{
					cur = (uint32_t)nx;
					for (int32_t sm = (int32_t)cur; sm >= 0; sm = M.trie[sm].fail)
					{
						cnt.failHops++;
						const int32_t v = M.trie[sm].value;
						if (v == TRIE_NONE) break;
						if (v != TRIE_SUBMATCH) cands.push_back((uint32_t)v);
					}
				}
				else cur = 0;

				// flushCandidates (KTrie.cpp:955-996)
				const uint32_t endNs = posToNs[j + 1];
				KORC_HAZ(if (!cands.empty()) { hazStats().steps++; hazStats().ops += cands.size(); if (cands.size() > hazStats().maxCand) hazStats().maxCand = cands.size(); })
				for (uint32_t fi : cands)
				{
					const FormRec& f = M.forms[fi];
					const uint32_t nb = endNs - (f.len - f.numSpaces), ne = endNs;
					if (!(f.flags & FF_FIRST_IS_CODA))
					{
						const bool hj = (f.flags & FF_HAS_JCLASS) || (f.flags & FF_IS_STAG);
						if (boundary < nb) insertUnk(boundary, nb, hj);
						insertUnk(unkStart, nb, hj);
					}
					const uint32_t se = spaceErrors(f, nb, ne);
					if (se <= cfg.spaceTol)
					{
						if (append(nb, ne, fi, 0, 0)) { out.back().spaceErrors = se; cnt.candEmits++; }
						KORC_HAZ(else hazStats().appFailReach++;)
					}
				}
				cands.clear();
			}
			if (lastType != T_MAX && lastType != T_UNKNOWN && lastType != T_SS)
			{
				const bool sj = T_SF <= lastType && lastType <= T_SW;
				unkPair(boundary, unkStart, specialStart, sj);
				uint32_t o, l; trimmed(nsToPos[specialStart], n - nsToPos[specialStart], o, l);
				if (append(specialStart, posToNs[n], lastType - 1u, o, l)) cnt.otherNodes++;
				unkStart = specialStart;
				if (sj) boundary = posToNs[n];
			}
			const uint32_t totEnd = nsToPos.back() + 1;
			if (n == totEnd) unkPair(boundary, unkStart, posToNs[totEnd], true);
			append(nNs, nNs + 1, NOFORM, 0, 0);
			out.back().endPos = nNs;
			KORC_HAZ(hazStats().chunks++; if (hazStats().cur) hazStats().chunksHaz++; hazStats().cur = false;)

			// removeUnconnected (KTrie.cpp:240-299)
			const uint32_t G = (uint32_t)out.size();
			std::vector<uint8_t> conn(G, 0);
			std::deque<uint32_t> dq{ G - 1 };
			conn[G - 1] = 1;
			while (!dq.empty())
			{
				const uint32_t id = dq.front(); dq.pop_front();
				const auto& mp = endPosMap[out[id].startPos];
				for (uint32_t i = mp.first; i < mp.second; ++i)
				{
					if (out[i].endPos != out[id].startPos || conn[i]) continue;
					conn[i] = 1; dq.push_back(i);
				}
			}
			std::vector<uint32_t> sorted(G), inv(G);
			for (uint32_t i = 0; i < G; ++i) sorted[i] = i;
			std::stable_sort(sorted.begin(), sorted.end(), [&](uint32_t a, uint32_t b)
			{
				if (conn[a] != conn[b]) return conn[a] > conn[b];
				return out[a].endPos < out[b].endPos;
			});
			for (uint32_t i = 0; i < G; ++i) inv[sorted[i]] = i;
			uint32_t nConn = 0;
			for (auto v : conn) nConn += v;
			ret.clear();
			for (uint32_t i = 0; i < nConn; ++i)
			{
				const uint32_t idx = sorted[i];
				LNode nn = out[idx];
				if (nn.prev) nn.prev = i - inv[idx - nn.prev];
				if (nn.sibling)
				{
					const uint32_t ns = inv[idx + nn.sibling];
					nn.sibling = ns >= nConn ? 0 : ns - i;
				}
				ret.push_back(nn);
			}
			for (uint32_t i = 1; i + 1 < ret.size(); ++i)
			{
				ret[i].startPos = nsToPos[ret[i].startPos] + startOffset;
				ret[i].endPos = nsToPos[ret[i].endPos - 1] + 1 + startOffset;
				if (ret[i].uformLen) ret[i].uformOff += startOffset;   // text-relative from here on
			}
			ret.back().startPos = ret.back().endPos = startOffset + n;
			cnt.lattNodes += ret.size();
			return ret.size() > 2;
		}
	};
}
