// TEST INFRASTRUCTURE -- CPU restatement of the reference's lattice construction OVER A TYPO GRAPH:
//   Splitter::buildTypoGraph / search / progressNode / flushCandidates / insertUnkForm / hasFormAlready / isZFollowable / writeResult
//   /root/reference/src/KTrie.cpp:873-996, 998-1464 (the general case; lattice_oracle.hpp is the same code specialised to the two-node
//   graph of a text without a typo transformer, and stays the oracle of the product's lattice kernels).
// One search state per (typo-graph node, way of reaching it): trie node, accumulated typo cost, the minimal form length that still covers
// the typo, start-position offset (a correction may be longer or shorter than what it replaces), the unknown-form / special-character
// bookkeeping positions, the last character and the continual-typo index the state started in.  Lattice positions are multiplied by
// 2^posMultiplierBit so that the halves of a continual typo (a coda carried over to the next syllable) get positions of their own.
// With a finite lengthening cost (SearchState<true>) a state also carries the trie nodes reached by skipping 1..8 syllables that merely
// lengthen the preceding vowel ("아아아"); a form found through one of them starts that many positions earlier and costs cost * (3 + skipped).
// Pinned against the real translation unit by tests/test_typo_oracle.py through kref_split_typo.
#pragma once
#include <deque>
#include "lattice_oracle.hpp"
#include "typo_oracle.hpp"

namespace korc
{
	class TypoLatticeBuilder
	{
		const ModelView& M;
		const SplitConfig& cfg;
		const char16_t* str; uint32_t n;
		std::vector<uint32_t> nsToPos, posToNs;
		std::vector<std::pair<uint32_t, uint32_t>> endPosMap;
		std::vector<LNode> out;
		std::vector<typo::GraphNode> graph;
		uint32_t pmb = 0;
		float typoThreshold = 2.5f, lengtheningCost = INFINITY;
		bool lengthening = false;
		const PatternSpan* pat = nullptr; const PatternSpan* patEnd = nullptr;
		Counters* cnt = nullptr;      // ALG_BYTES events, as LatticeBuilder counts them + the typo-graph / search-state events

		struct SState
		{
			int32_t node = 0; float cost = 0; uint32_t minFormLen = 0; int32_t startPosOffset = 0;
			uint32_t specialStart = 0, unkStart = 0, boundary = 0; uint32_t lastChr = 0; uint16_t startCti = 0;
			std::vector<std::pair<uint32_t, int32_t>> lnodes;      // (syllables skipped, trie node): LengtheningTypoNodes<true>
		};
		struct Cand { uint32_t form; uint32_t lengthened; };

		bool append(uint32_t s, uint32_t e, uint32_t form, uint32_t uOff, uint32_t uLen, float typoCost = 0)
		{
			if (endPosMap[s].first == endPosMap[s].second) return false;
			const uint32_t id = (uint32_t)out.size();
			LNode nn; nn.startPos = s; nn.endPos = e; nn.form = form; nn.uformOff = uOff; nn.uformLen = uLen; nn.typoCost = typoCost;
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
		bool hasFormAlready(uint32_t ms, uint32_t me) const      // multiplied positions
		{
			const uint32_t a = std::max(endPosMap[me].first, 1u), b = endPosMap[me].second;
			if (endPosMap[me].first == 0xFFFFFFFFu) return false;
			for (uint32_t i = a; i < b; ++i)
			{
				const LNode& g = out[i];
				if (g.endPos == me && g.endPos - (nodeLen(g) << pmb) == ms && g.typoCost == 0 && (g.form == NOFORM || (M.forms[g.form].flags & FF_HAS_ANY_FULL))) return true;
			}
			return false;
		}
		void trimmed(uint32_t off, uint32_t len, uint32_t& oOff, uint32_t& oLen) const
		{
			while (len && isSpace(str[off + len - 1])) --len;
			oOff = off; oLen = len;
		}
		void insertUnk(uint32_t s, uint32_t e, bool hasJ)      // ns positions (KTrie.cpp:923-953)
		{
			if (s >= e || hasFormAlready(s << pmb, e << pmb)) return;
			uint32_t lastPos = out.back().endPos;      // (a multiplied position compared with plain ones: as in the reference)
			if (lastPos < e)
			{
				if (lastPos && isHangulCoda(str[nsToPos[lastPos]])) lastPos--;
				if (lastPos != s && !hasFormAlready(lastPos << pmb, e << pmb))
				{
					uint32_t o, l; trimmed(nsToPos[lastPos], nsToPos[e - 1] + 1 - nsToPos[lastPos], o, l);
					if (append(lastPos << pmb, e << pmb, NOFORM, o, l) && cnt) cnt->otherNodes++;
				}
			}
			const uint32_t limit = hasJ ? cfg.maxUnkJ : cfg.maxUnk;
			if (e - s <= limit)
			{
				uint32_t o, l; trimmed(nsToPos[s], nsToPos[e - 1] + 1 - nsToPos[s], o, l);
				if (append(s << pmb, e << pmb, NOFORM, o, l) && cnt) cnt->otherNodes++;
			}
		}
		void unkPair(uint32_t boundary, uint32_t unkStart, uint32_t e, bool hasJ)
		{
			if (boundary < unkStart) insertUnk(boundary, e, hasJ);
			insertUnk(unkStart, e, hasJ);
		}
		uint32_t spaceErrors(const FormRec& f, uint32_t b, uint32_t e) const      // countSpaceErrors (KTrie.cpp:318-328)
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
		int32_t trieNext(uint32_t node, uint16_t c) const
		{
			if (cnt) cnt->trieProbes++;
			if (node == 0) { if (cnt) cnt->trieProbeKeyBytes += 4; const uint32_t r = M.trieRoot[c]; return r ? (int32_t)r : -1; }
			const TrieNodeRec& t = M.trie[node];
			if (cnt) { uint32_t l = 0; while ((1u << l) < t.numNexts) ++l; cnt->trieProbeKeyBytes += 2 * l; }
			const uint16_t* kb = M.trieKeys + t.edgeOff;
			const uint16_t* it = std::lower_bound(kb, kb + t.numNexts, c);
			if (it == kb + t.numNexts || *it != c) return -1;
			return (int32_t)M.trieChild[t.edgeOff + (it - kb)];
		}

		// flushCandidates (KTrie.cpp:955-996)
		void flush(std::vector<Cand>& cands, uint32_t endNs, int32_t startPosOffset, uint32_t unkStart, uint32_t boundary, float typoCost, uint32_t startCti, uint32_t endCti)
		{
			for (const Cand& cd : cands)
			{
				const uint32_t fi = cd.form;
				const FormRec& f = M.forms[fi];
				const uint32_t nb = (uint32_t)((int64_t)endNs - (int64_t)(f.len - f.numSpaces) - (int64_t)cd.lengthened + startPosOffset), ne = endNs;
				if (startCti == 0 && !(f.flags & FF_FIRST_IS_CODA))
				{
					const bool hj = (f.flags & FF_HAS_JCLASS) || (f.flags & FF_IS_STAG);
					if (boundary < nb) insertUnk(boundary, nb, hj);
					insertUnk(unkStart, nb, hj);
				}
				const uint32_t se = spaceErrors(f, nb, ne);
				if (se <= cfg.spaceTol)
				{
					const uint32_t b2 = startCti ? (nb << pmb) + startCti : nb << pmb;
					const uint32_t e2 = endCti ? ((ne - 1) << pmb) + endCti : ne << pmb;
					if (append(b2, e2, fi, 0, 0, typoCost + (cd.lengthened ? lengtheningCost * (float)(3 + cd.lengthened) : 0.f))) { out.back().spaceErrors = se; if (cnt) cnt->candEmits++; }
				}
			}
			cands.clear();
		}

		// progressNode (KTrie.cpp:998-1412), lengtheningTypoTolerant = false, no pretokenized spans
		void progress(const typo::GraphNode& prevT, const typo::GraphNode& tn, const SState& st, std::vector<SState>& cur)
		{
			if (cnt) cnt->typoStateSteps++;
			float typoCost = st.cost + tn.typoCost;
			if (typoCost > typoThreshold) return;
			uint32_t prevChr = st.lastChr;
			uint8_t lastType = prevChr ? identifySpecialChr(prevChr) : (uint8_t)T_UNKNOWN;
			uint8_t lastScript = prevChr ? chr2ScriptType(prevChr) : 0;
			uint32_t specialStart = st.specialStart, unkStart = st.unkStart, boundary = st.boundary;
			uint32_t minFormLen = st.minFormLen;
			int32_t startPosOffset = st.startPosOffset;
			const uint32_t fsz = (uint32_t)tn.form.size();
			if (tn.typoCost > 0) startPosOffset += (int32_t)fsz - (int32_t)(tn.endPos - prevT.endPos);
			int32_t curNode = st.node;      // -1 = none
			const uint8_t scriptVS = 98;    // ScriptType::variation_selectors
			std::vector<Cand> cands;
			auto lnodes = st.lnodes;
			const uint32_t nNs = (uint32_t)nsToPos.size();
			for (uint32_t j = 0; j < fsz; ++j)
			{
				const uint16_t ch = tn.form[j];
				uint32_t c32 = ch;
				if (isHighSurrogate(c32) && j + 1 < fsz) c32 = mergeSurrogate(c32, tn.form[j + 1]);
				const uint32_t pos = tn.endPos + j - fsz;      // position of this character in the text (meaningful for zero-cost nodes)
				if (typoCost == 0)
				{
					const bool inPattern = pat != patEnd && pos >= pat->end - pat->length;
					uint8_t type = identifySpecialChr(c32), sct = chr2ScriptType(c32);
					if (lastType == T_SW && (c32 == 0x200d || (0x1f3fb <= c32 && c32 <= 0x1f3ff) || sct == scriptVS)) { type = lastType; sct = lastScript; }
					const uint8_t curT = inPattern ? (uint8_t)T_UNKNOWN : type;
					auto sym = [](uint8_t t) { return t == T_SL || t == T_SH || t == T_SW; };
					const bool discont = (sym(lastType) && sym(curT)) ? (lastScript != sct) : (lastType != curT);
					if (discont || lastType == T_SSO || lastType == T_SSC)
					{
						if (lastType != T_MAX && lastType != T_UNKNOWN && lastType != T_SS)
						{
							const bool sj = T_SF <= lastType && lastType <= T_SW;
							unkPair(boundary, unkStart, specialStart, sj);
							uint32_t o, l; trimmed(nsToPos[specialStart], pos - nsToPos[specialStart], o, l);
							if (append(specialStart << pmb, posToNs[pos] << pmb, lastType - 1u, o, l) && cnt) cnt->otherNodes++;
						}
						unkStart = specialStart;
						specialStart = posToNs[pos];
						if (T_SF <= lastType && lastType <= T_SW) boundary = specialStart;
					}
					else if (type == T_MAX) unkStart = specialStart;
					lastType = curT; lastScript = sct;
					if (c32 < 0x10000)
					{
						if (type == T_UNKNOWN)
						{
							unkPair(boundary, unkStart, posToNs[pos + 1], true);
							boundary = specialStart = unkStart = posToNs[pos + 1];
							prevChr = c32;
							continue;
						}
						bool zc = false, zs = false;
						{
							const uint32_t p = posToNs[pos];
							if (p < nNs)
							{
								const uint32_t a = endPosMap[p << pmb].first, b = endPosMap[p << pmb].second;
								if (a != 0xFFFFFFFFu) for (uint32_t i = a; i < b; ++i)
								{
									if (out[i].endPos != (p << pmb) || out[i].form == NOFORM) continue;
									zc = zc || (M.forms[out[i].form].flags & FF_ZCODA_APPENDABLE);
									zs = zs || (M.forms[out[i].form].flags & FF_ZSIOT_APPENDABLE);
								}
							}
						}
						if ((cfg.match & M_Z_CODA) && zc && isHangulCoda(ch) && (pos + 1 >= n || !isHangulSyllable(str[pos + 1])))
							cands.push_back(Cand{ kDefaultTagSize + (ch - 0x11A8) - 1u, 0 });
						else if ((cfg.match & (M_SPLIT_SAISIOT | M_MERGE_SAISIOT)) && zs && ch == 0x11BA && pos + 1 < n && isHangulSyllable(str[pos + 1]))
							cands.push_back(Cand{ kDefaultTagSize + (0x11BA - 0x11A8) - 1u, 0 });
					}
				}
				else if (isSpace(c32))
				{
					boundary = specialStart = unkStart = posToNs[pos + 1];
					prevChr = c32;
					continue;
				}
				if (tn.typoCost == 0 && pat != patEnd)
				{
					const uint32_t curEnd = pos + (c32 >= 0x10000 ? 2 : 1);
					while (pat != patEnd && pat->end == curEnd)
					{
						const uint32_t ms = pat->end - pat->length;
						const bool wj = T_W_URL <= pat->tag && pat->tag <= T_W_EMOJI;
						unkPair(boundary, unkStart, posToNs[ms], wj);
						if (append(posToNs[ms] << pmb, posToNs[pat->end] << pmb, pat->tag - 1u, ms, pat->length) && cnt) cnt->otherNodes++;
						++pat;
					}
				}
				if (c32 >= 0x10000) { ++j; prevChr = c32; continue; }
				if (lengthening)      // KTrie.cpp:1215-1270
				{
					static const uint8_t lengtheningVowel[21] = { 0, 1, 0, 1, 4, 5, 4, 5, 8, 0, 1, 1, 8, 13, 4, 5, 20, 13, 18, 20, 20 };
					const size_t prevSize = lnodes.size();
					if (prevChr && prevChr < 0x10000 && isHangulSyllable((char16_t)prevChr) && (0xC544 <= ch && ch < 0xC790)
						&& lengtheningVowel[((prevChr - 0xAC00) / 28) % 21] == ((ch - 0xAC00) / 28) % 21)
					{
						lnodes.emplace_back(1u, curNode);
						for (size_t i = 0; i < prevSize; ++i) { const auto nd = lnodes[i]; if (nd.first < 8) lnodes.emplace_back(nd.first + 1, nd.second); }
					}
					size_t outIdx = 0;
					for (size_t i = 0; i < prevSize; ++i)
					{
						auto nd = lnodes[i];
						nd.second = trieNext((uint32_t)nd.second, ch);
						lnodes[i] = nd;
						if (nd.second < 0) continue;
						if (std::find(lnodes.begin(), lnodes.begin() + outIdx, nd) != lnodes.begin() + outIdx) continue;
						lnodes[outIdx++] = nd;
					}
					for (size_t i = prevSize; i < lnodes.size(); ++i)
					{
						const auto nd = lnodes[i];
						if (std::find(lnodes.begin(), lnodes.begin() + outIdx, nd) != lnodes.begin() + outIdx) continue;
						lnodes[outIdx++] = nd;
					}
					lnodes.resize(outIdx);
				}
				prevChr = c32;

				if (minFormLen > 0 || tn.typoCost > 0) ++minFormLen;
				int32_t nx = curNode >= 0 ? trieNext((uint32_t)curNode, ch) : -1;
				while (nx < 0 && curNode >= 0)
				{
					curNode = M.trie[curNode].fail;      // -1 at the root
					if (cnt) cnt->failHops++;
					if (curNode < 0) break;
					nx = trieNext((uint32_t)curNode, ch);
				}
				if (nx >= 0)
				{
					curNode = nx;
					// with a typo in the node only forms that contain the whole correction are looked for
					if (tn.typoCost == 0 || j == fsz - 1)
					{
						if (typoCost > 0 && M.trie[curNode].depth < minFormLen) {}      // early pruning
						else for (int32_t sm = curNode; sm >= 0; sm = M.trie[sm].fail)
						{
							const int32_t v = M.trie[sm].value;
							if (v == TRIE_NONE) break;
							if (v != TRIE_SUBMATCH)
							{
								if (M.forms[v].len < minFormLen) break;
								cands.push_back(Cand{ (uint32_t)v, 0 });
							}
						}
						for (auto& ln : lnodes)
						{
							const int32_t v = M.trie[ln.second].value;
							if (v >= 0)
							{
								if (M.forms[v].len < minFormLen) continue;
								cands.push_back(Cand{ (uint32_t)v, ln.first });
							}
						}
					}
				}
				else
				{
					lnodes.clear();
					if (typoCost == 0) curNode = 0;
					else return;
				}
				flush(cands, posToNs[tn.endPos + j + 1 - fsz], startPosOffset, unkStart, boundary, typoCost, st.startCti, tn.continualTypoIdx);
			}
			if (typoCost == 0 && lastType != T_MAX && lastType != T_UNKNOWN)
			{
				if (lastType != T_SS)
				{
					const bool sj = T_SF <= lastType && lastType <= T_SW;
					unkPair(boundary, unkStart, specialStart, sj);
					uint32_t o, l; trimmed(nsToPos[specialStart], tn.endPos - nsToPos[specialStart], o, l);
					if (append(specialStart << pmb, posToNs[tn.endPos] << pmb, lastType - 1u, o, l) && cnt) cnt->otherNodes++;
					unkStart = specialStart;
					if (sj) boundary = posToNs[tn.endPos];
				}
			}
			if (curNode >= 0)
			{
				if (tn.continualTypoIdx)
				{
					curNode = 0; typoCost = 0; minFormLen = 0; startPosOffset = -1;
					if (!cur.empty()) return;
					lnodes.clear();
				}
				if (typoCost > 0 && M.trie[curNode].depth < minFormLen && lnodes.empty()) {}      // early pruning
				else
				{
					SState ns; ns.node = curNode; ns.cost = typoCost; ns.minFormLen = minFormLen; ns.startPosOffset = startPosOffset;
					ns.specialStart = specialStart; ns.unkStart = unkStart; ns.boundary = boundary; ns.lastChr = prevChr;
					ns.startCti = tn.continualTypoIdx ? tn.continualTypoIdx : st.startCti;
					ns.lnodes = std::move(lnodes);
					cur.push_back(std::move(ns));
					if (cnt) cnt->typoStatesKept++;
				}
			}
		}

	public:
		TypoLatticeBuilder(const ModelView& m, const SplitConfig& c) : M(m), cfg(c) {}
		void countInto(Counters* c) { cnt = c; }

		// The lattice of chunk str[0..len) over the typo graph `prepared` generates for it.  Same output conventions as LatticeBuilder::build.
		bool build(std::vector<LNode>& ret, const char16_t* s, uint32_t len, const PatternSpan* patBegin, const PatternSpan* patEndIn, uint32_t startOffset,
			const typo::Prepared& prepared, float threshold, uint16_t allowedDialect)
		{
			lengtheningCost = prepared.lengthening(); lengthening = std::isfinite(lengtheningCost);
			str = s; n = len; pat = patBegin; patEnd = patEndIn; typoThreshold = threshold;
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
			// buildTypoGraph (KTrie.cpp:873-895)
			size_t maxCti = 0;
			graph = prepared.graph(std::u16string{ str, n }, allowedDialect, maxCti);
			if (cnt) { cnt->inputUnits += n; cnt->typoGraphNodes += graph.size(); }
			pmb = 0;
			if (maxCti > 1) { size_t v = maxCti - 1; while (v > 0) { v >>= 1; ++pmb; } }
			endPosMap.assign(((size_t)nNs << pmb) + 1, { 0xFFFFFFFFu, 0xFFFFFFFFu });
			endPosMap[0] = { 0, 1 };
			out.emplace_back();

			// search (KTrie.cpp:1414-1452)
			const uint32_t totEnd = nsToPos.back() + 1;
			std::vector<std::vector<SState>> states(graph.size());
			states[0].emplace_back();
			for (size_t i = 1; i < graph.size(); ++i)
			{
				const auto& tn = graph[i];
				auto& cur = states[i];
				for (size_t p = tn.prevOffset ? i - tn.prevOffset : (size_t)-1; p != (size_t)-1; p = graph[p].siblingOffset ? p + graph[p].siblingOffset : (size_t)-1)
				{
					const std::vector<SState> prevStates = states[p];      // (copy: `cur` may be states[p]'s neighbour in memory, never itself)
					for (auto& st : prevStates) progress(graph[p], tn, st, cur);
				}
				if (tn.typoCost == 0 && tn.endPos == totEnd)
				{
					for (auto& st : cur) unkPair(st.boundary, st.unkStart, posToNs[totEnd], true);
				}
			}
			append(nNs << pmb, (nNs << pmb) + 1, NOFORM, 0, 0);
			out.back().endPos = nNs << pmb;

			// removeUnconnected (KTrie.cpp:240-299) + writeResult (:1454-1464)
			const uint32_t G = (uint32_t)out.size();
			std::vector<uint8_t> conn(G, 0);
			std::deque<uint32_t> dq{ G - 1 };
			conn[G - 1] = 1;
			while (!dq.empty())
			{
				const uint32_t id = dq.front(); dq.pop_front();
				const auto& mp = endPosMap[out[id].startPos];
				if (mp.first == 0xFFFFFFFFu) continue;
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
				ret[i].startPos = nsToPos[ret[i].startPos >> pmb] + startOffset;
				ret[i].endPos = nsToPos[((ret[i].endPos + (1u << pmb) - 1) >> pmb) - 1] + 1 + startOffset;
				if (ret[i].uformLen) ret[i].uformOff += startOffset;
			}
			ret.back().startPos = ret.back().endPos = startOffset + n;
			if (cnt) cnt->lattNodes += ret.size();
			return ret.size() > 2;
		}
	};
}
