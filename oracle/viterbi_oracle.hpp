// TEST INFRASTRUCTURE ONLY -- never linked into the product.
//
// CPU restatement of the reference's best-path search for one lattice, Knlm scoring, top-1:
//   BestPathFinder<KnLangModel>::findBestPath      /root/reference/src/PathEvaluator.hpp:1178-1419
//   PathEvaluator::operator() / evalSingleMorpheme  /root/reference/src/PathEvaluator.hpp:347-635
//   RuleBasedScorer / insertToPathContainer / FormEvaluator  /root/reference/src/PathEvaluator.hpp:88-311
//   BestPathConatiner top1 / top1Small / top1Medium /root/reference/src/BestPathContainer.hpp:230-483
//   generateTokenList                               /root/reference/src/PathEvaluator.hpp:1038-1157
//   KnLangModel::progress                           /root/reference/src/Knlm.cpp:44-130
// top-N (> 1): the reference keeps, per candidate morpheme, a min-heap of the N best paths per key in a thread_local
// std::unordered_map (BestPathContainer.hpp:151-222); the ORDER in which it hands them on is the map's bucket order, which
// depends on everything the thread analysed before (the map is never shrunk).  Only that order is not restated: here (and on
// the device) the kept paths are handed on in insertion order.  Which paths are kept -- the N best per key, the earlier one
// on equal scores -- and every score is the reference's; results can differ from a given reference run only where two
// paths tie exactly.
#pragma once
#include <algorithm>
#include <cmath>
#include <unordered_map>
#include <unordered_set>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "lattice_oracle.hpp"
#include "../kiwi_amd/csrc/feature.hpp"
#include "../kiwi_amd/csrc/post.hpp"
#include "../kiwi_amd/csrc/cong_global.hpp"
#include "unk_freq_oracle.hpp"

namespace korc
{
	struct BestPathConfig
	{
		float cutOff = 8, spacePenalty = 7, typoCostWeight = 6, oovRuleScale = 4, oovRuleBias = 4;
		// Match::oovChrModel: unknown forms are scored by the character model instead of the length rule (UnkFormScorer::operator(), src/UnkFormScorer.h:40-58)
		const kamd::ChrView* chr = nullptr; float oovChrBias = 0;
		// Match::oovChrFreqModel / oovChrFreqBranchModel: ... mixed with the substring counts of the (filtered) text under analysis (unk_freq_oracle.hpp)
		const SubstringCounts* substr = nullptr; ChrFreqConfig freq;
		uint32_t spaceTolerance = 0;
		uint32_t topN = 1;
		// container selection by number of incoming paths and the per-bucket key cap (BestPathContainer.hpp:275-277, 363-367);
		// tests shrink them to drive the medium / large containers on small lattices
		uint32_t smallMax = 128, mediumMax = 512, bucketCap = 128;
		// AnalyzeOption::blocklist as one bit per morpheme id, Morpheme::hasMorpheme already applied (flat_model.hpp blockBitsOf); null = none
		const uint32_t* blockBits = nullptr;
		// AnalyzeOption::allowedDialects (Dialect bits) / dialectCost: a candidate of a dialect that is neither standard nor allowed is not a candidate, one of
		// an allowed dialect costs dialectCost (src/PathEvaluator.hpp:231-236, 386, 893); only models with dialect morphemes (ModelView::morphDialect) care
		uint32_t allowedDialect = 0; float dialectCost = 3.f;
		// `faithfulOrder`: the large top-1 container and the top-N container are the reference's own -- std::unordered_set / std::unordered_map
		// + std heap algorithms of this libstdc++, PERSISTENT across calls like the reference's thread_local ones -- so that the order in
		// which kept paths are handed on is the reference's as long as both sides analyse the same texts in the same sequence from a
		// fresh state.  Off (default): insertion order, the order the device implements.
		bool faithfulOrder = false;
		bool openEnding = false, splitComplex = false, splitSaisiot = false, mergeSaisiot = false;
	};

	constexpr uint8_t COMMON_ROOT = 0xFF;

	struct WPath   // WordLL<KnLMState> (BestPathContainer.hpp:21-67)
	{
		uint32_t ctx = 0;              // CoNgram models: CoNgramState::contextIdx (the LM state proper is lmNode: CoNgramModel.hpp:470-473)
		int32_t lmNode = 0;
		// SkipBigram state on top of the Knlm node (SbgState, SkipBigramModel.hpp:141-182): ring of the last 8 valid word ids
		uint32_t hist[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }; uint8_t histPos = 0;
		uint8_t prevRootId = 0, spState = 0, rootId = 0;
		uint32_t morph = 0;
		float accScore = 0, firstChunkScore = 0, accTypoCost = 0;
		int32_t parentNode = -1, parentIdx = -1;
		uint32_t wid = 0;
		uint16_t ownFormId = 0;
		uint8_t combineSocket = 0;
	};

	struct WPathSetHash
	{
		size_t operator()(const WPath& p) const   // Hash<WordLL<LmState>> (BestPathContainer.hpp:69-86) with Hash<KnLMState> / Hash<SbgState>
		{
			size_t r = (size_t)(int64_t)p.lmNode;
			if (useHist) for (int i = 0; i < 8; ++i) r = (size_t)p.hist[i] ^ ((r << 3) | (r >> 61));
			return ((uint16_t)p.prevRootId | ((uint16_t)p.spState << 8)) ^ ((r << 3) | (r >> 61));
		}
		bool useHist = false;
	};
	struct WPathSetEq
	{
		bool operator()(const WPath& a, const WPath& b) const
		{
			if (a.prevRootId != b.prevRootId || a.spState != b.spState || a.lmNode != b.lmNode) return false;
			if (!useHist) return true;
			if (a.histPos != b.histPos) return false;
			for (int i = 0; i < 8; ++i) if (a.hist[i] != b.hist[i]) return false;
			return true;
		}
		bool useHist = false;
	};
	struct TopNKey { int32_t lm; uint8_t rootId, sp; bool operator==(const TopNKey& o) const { return lm == o.lm && rootId == o.rootId && sp == o.sp; } };
	struct TopNKeyHash   // Hash<PathHash<LmState>> (BestPathContainer.hpp:113-121), Knlm state
	{
		size_t operator()(const TopNKey& k) const { size_t r = (size_t)(int64_t)k.lm; return ((uint16_t)k.rootId | ((uint16_t)k.sp << 8)) ^ ((r << 3) | (r >> 61)); }
	};
	// what the reference keeps in thread_local storage: never shrunk, so their bucket counts carry the history of the thread
	struct PersistentContainers
	{
		std::unordered_set<WPath, WPathSetHash, WPathSetEq> large{ 0, WPathSetHash{}, WPathSetEq{} };
		std::unordered_map<TopNKey, std::pair<uint32_t, uint32_t>, TopNKeyHash> topIndex;
		std::vector<WPath> topValues;
		bool histInit = false;
	};

	class BestPathSearch
	{
		const ModelView& M;
		SbgView S;                       // absent (vocabSize == 0): plain Knlm scoring
		CongView C;                      // present: CoNgram scoring (local, quantised) through the transposed evaluator
		const BestPathConfig& cfg;
		Counters& cnt;
		const U16* norm = nullptr;   // normalised text
		const LNode* graph = nullptr;
		uint32_t G = 0;
		std::vector<std::vector<WPath>> cache;
		struct OwnForm { uint32_t off, len; };   // offsets into the normalised text, or into a form string (kind 1)
		std::vector<std::pair<int, OwnForm>> ownForms;  // kind 0: text substring, kind 1: dictionary form (id in off)
		std::vector<uint8_t> uniqStates;
		float leftBoundary[2 * T_MAX + 1];   // [2][max] followed by `weight`: tag PA (== max) indexes one past a row, as in the reference

		static uint32_t log2c(uint32_t v) { uint32_t l = 0; while ((1u << l) < v + 1) ++l; return l; }

		bool lmSearch(const LmNodeRec& nd, uint32_t key, int32_t& v)
		{
			cnt.lmProbeKeyBytes += 2 * log2c(nd.numNexts) * M.h.lmKeyBytes;
			const uint32_t* k = M.lmKeys + nd.nextOff;
			const uint32_t* it = std::lower_bound(k, k + nd.numNexts, key);
			if (it == k + nd.numNexts || *it != key) return false;
			v = M.lmValues[nd.nextOff + (it - k)];
			return true;
		}
		static float asFloat(int32_t v) { float f; std::memcpy(&f, &v, 4); return f; }

	public:
		float lmProgress(int32_t& node, uint32_t next)
		{
			float acc = 0;
			for (;;)
			{
				int32_t v;
				const LmNodeRec* nd = &M.lmNodes[node];
				if (node == 0)
				{
					cnt.lmRootProbes++;
					v = M.lmRoot[next];
					// history-transformed model (Knlm.cpp:61-70): an unseen word still moves the state to the root's child for its transformed id
					if (v == 0) { node = M.lmHtxNode ? M.lmHtxNode[next] : 0; return acc + M.h.unkLl; }
				}
				else
				{
					cnt.lmProbes++;
					if (!lmSearch(*nd, next, v)) { acc += nd->gamma; node += nd->lower; continue; }
				}
				if (v > 0) { node += v; return acc + M.lmNodes[node].ll; }
				int32_t cur = node;
				while (M.lmNodes[cur].lower)
				{
					cur += M.lmNodes[cur].lower;
					int32_t lv;
					cnt.lmProbes++;
					if (lmSearch(M.lmNodes[cur], next, lv) && lv > 0) { node = cur + lv; return acc + asFloat(v); }
				}
				node = M.lmHtxNode ? M.lmHtxNode[next] : 0;      // (Knlm.cpp:116-126)
				return acc + asFloat(v);
			}
		}

		// LmState::next: Knlm alone, or SbgState::nextImpl (SkipBigramModel.hpp:169-182) + SkipBigramModel::evaluate (:113-139) with
		// the scalar logSumExp of MathFunc.hpp:43-56 (ArchType::none / balanced): same operations, same order, fp32
		// CoNgramModel::progressContextNodeVl (src/CoNgramModel.hpp:306-385): the context id the history + `next` maps to; moves `node`
		bool congSearch(const CongNodeRec& nd, uint32_t key, int32_t& v) const
		{
			cnt.congProbes++; cnt.congProbeKeyBytes += 2 * log2c(nd.numNexts) * 4;
			const uint32_t* k = C.keys + nd.nextOff;
			const uint32_t* it = std::lower_bound(k, k + nd.numNexts, key);
			if (it == k + nd.numNexts || *it != key) return false;
			v = C.values[nd.nextOff + (it - k)];
			return v != 0;
		}
		// CoNgramModel::progressContextNode (src/CoNgramModel.hpp:271-300): with variable-length keys (cong.mdl keySize 3) a word id >= tMax takes two
		// steps through the trie, the first one's context id is dropped
		uint32_t congContext(int32_t& nodeIdx, uint32_t next) const
		{
			if (next < C.vlTMax) return congContextVl(nodeIdx, next);
			const uint32_t r = next - C.vlTMax;
			congContextVl(nodeIdx, C.vlTMax + (r >> C.vlBits));
			return congContextVl(nodeIdx, C.vlTMax + (1u << C.vlBits) + (r & ((1u << C.vlBits) - 1)));
		}
		uint32_t congContextVl(int32_t& nodeIdx, uint32_t next) const
		{
			for (;;)
			{
				int32_t v;
				const CongNodeRec* node = &C.nodes[nodeIdx];
				if (nodeIdx != 0)
				{
					if (!congSearch(*node, next, v))
					{
						if (!node->lower) return 0;
						nodeIdx += node->lower;
						continue;
					}
				}
				else
				{
					cnt.congRootProbes++;
					v = next < C.rootSize ? C.root[next] : 0;
					if (v == 0) return 0;
				}
				if (v > 0) { nodeIdx += v; return C.nodes[nodeIdx].value; }
				while (node->lower)
				{
					node += node->lower;
					int32_t lv;
					if (node != C.nodes)
					{
						if (congSearch(*node, next, lv) && lv > 0) { nodeIdx = (int32_t)(node + lv - C.nodes); return (uint32_t)-v; }
					}
					else
					{
						lv = next < C.rootSize ? C.root[next] : 0;
						if (lv > 0) { nodeIdx = lv; return (uint32_t)-v; }
					}
				}
				nodeIdx = 0;
				return (uint32_t)-v;
			}
		}
		// CoNgramModel::progress, window 0, quantised (src/CoNgramModel.cpp:869-908): the score in the CURRENT context, then the context moves on
		// outputFirst: the batched path of the reference's SSE4.1 build multiplies the output scale in first (src/archImpl/sse4_1.cpp:116,
		// scatteredGEMV_128: ((x * outputScale) * contextScale) + bias) where progress() and the baseline kernel (src/qgemm.hpp:73-80) multiply the
		// context scale first -- one rounding apart
		// Global model (C.window == 7; src/CoNgramModel.cpp:802-868): a valid distant token is scored as a mixture over the context and the state's seven
		// history words (csrc/cong_global.hpp), and every step pushes `next` (or 0) into the history.  `matrix`: the entry of progressMatrixWSort / WOSort
		// (:1037-1466) instead of state.next() -- other roundings, see cong_global.hpp.
		float congNext(WPath& st, uint32_t next, bool outputFirst = false, bool countRows = true, bool matrix = false) const
		{
			if (countRows) { cnt.congCtxRows++; cnt.congOutRows++; cnt.congScores++; }
			float ll;
			if (C.window && C.distant(next))
			{
				cnt.congGlobalScores++;
				ll = matrix ? congg::scoreMatrix(C, st.ctx, st.hist, next, outputFirst) : congg::scoreSingle(C, st.ctx, st.hist, next);
			}
			else ll = outputFirst ? congScoreOutputFirst(C, st.ctx, next) : congScore(C, st.ctx, next);
			st.ctx = congContext(st.lmNode, next);
			if (C.window) congg::pushHistory(C, st.hist, next);
			return ll;
		}

		float lmNext(WPath& st, uint32_t next)
		{
			if (C.present()) return congNext(st, next);
			float ll = lmProgress(st.lmNode, next);
			if (!S.present()) return ll;
			if (next < S.vocabSize && S.valid[next])
			{
				if (ll > -13)
				{
					float arr[16];
					for (int i = 0; i < 8; ++i) { arr[i] = ll; arr[8 + i] = -INFINITY; }
					const uint32_t* kb = S.keys + S.ptrs[next]; const uint32_t* ke = S.keys + S.ptrs[next + 1];
					cnt.sbgEvals++; cnt.sbgModel = 1;
					cnt.sbgProbeKeyBytes += 8ull * 2 * log2c((uint32_t)(ke - kb)) * M.h.lmKeyBytes;
					for (int i = 0; i < 8; ++i)
					{
						arr[i] = S.discnts[st.hist[i]] + ll;
						const uint32_t* it = std::lower_bound(kb, ke, st.hist[i]);
						if (it != ke && *it == st.hist[i]) { arr[8 + i] = S.comps[S.ptrs[next] + (it - kb)]; cnt.sbgHits++; }
					}
					const float mx = *std::max_element(arr, arr + 16);
					float sum = 0;
					for (int i = 0; i < 16; ++i) sum += std::exp(arr[i] - mx);
					ll = (std::log(sum) + mx) - S.logWindowSize;
				}
				st.hist[st.histPos] = next;
				st.histPos = (uint8_t)((st.histPos + 1) % 8);
			}
			return ll;
		}
		bool sameLm(const WPath& a, const WPath& b) const   // LmState::operator== (Knlm node; + history ring and position for SBG)
		{
			// CoNgramState<7>::operator== (src/CoNgramModel.hpp:452-461): the node and history[3..6] -- not the newest word, not the three oldest
			if (C.window) return a.lmNode == b.lmNode && a.hist[3] == b.hist[3] && a.hist[4] == b.hist[4] && a.hist[5] == b.hist[5] && a.hist[6] == b.hist[6];
			if (a.lmNode != b.lmNode || a.histPos != b.histPos) return false;
			for (int i = 0; i < 8; ++i) if (a.hist[i] != b.hist[i]) return false;
			return true;
		}
		// PathHash equality of the top-N container (BestPathContainer.hpp:89-111; SBG: SkipBigramModel.cpp:8-35 compares the Knlm
		// state and the LAST FOUR history words only)
		bool sameTopNKey(const WPath& a, const WPath& b) const
		{
			if (a.prevRootId != b.prevRootId && !S.present()) return false;    // the SBG PathHash::operator== does not compare rootId
			if (a.spState != b.spState || a.lmNode != b.lmNode) return false;
			if (C.window) return sameLm(a, b);
			if (!S.present()) return true;
			for (int i = 0; i < 4; ++i) if (a.hist[(a.histPos + 8 + i - 4) % 8] != b.hist[(b.histPos + 8 + i - 4) % 8]) return false;
			return true;
		}

	private:
		const uint16_t* ownStr(uint16_t id, uint32_t& len) const
		{
			const auto& o = ownForms[id - 1];
			len = o.second.len;
			if (o.first == 1) return M.formChars + M.forms[o.second.off].charOff;
			return (const uint16_t*)norm->data() + o.second.off;
		}

		bool hasLeftBoundary(const LNode* node) const  // PathEvaluator.hpp:24-44
		{
			const LNode* p = node - node->prev;
			if (p->endPos == 0) return true;
			if (p->endPos < node->startPos) return true;
			if (p->uformLen)
			{
				const uint16_t c = (*norm)[p->uformOff + p->uformLen - 1];
				const uint8_t tag = identifySpecialChr(c);
				if (tag == T_SSC || c == u'"' || c == u'\'') return false;
				if (T_SF <= tag && tag <= T_SB) return true;
			}
			return false;
		}

		struct Rule   // RuleBasedScorer (PathEvaluator.hpp:88-184)
		{
			uint8_t special; uint32_t sbType; int sbOrder;
			bool vowelE, infJ, badPairOfL, positiveE, contractableE, snEndsWithPoint; uint8_t condP;
			float operator()(const MorphRec& prev, uint8_t sp) const
			{
				float a = 0;
				if (vowelE && (prev.prevFlags & PF_IRREGULAR)) a -= 10;
				if (infJ && (prev.prevFlags & PF_INFLECTENDA_NP)) a -= 5;
				if (badPairOfL && (prev.prevFlags & PF_VERB_L)) a -= 7;
				if (positiveE && !(prev.prevFlags & PF_POSITIVE_VERB)) a -= 100;
				if (contractableE && (prev.prevFlags & PF_VERB_VOWEL)) a -= 3;
				if (condP == CP_NON_ADJ && (prev.prevFlags & PF_VA_OR_XSA)) a -= 10;
				if (special <= 2) { if (special != (sp & 1)) a -= 2; }
				else if (special <= 5) { if ((uint8_t)(special - 3) != ((sp >> 1) & 1)) a -= 2; }
				if (sbType == 5) a -= 5;
				if (sbType && (prev.prevFlags & PF_E_NOT_EF)) a -= 10;
				if (sbType && (sp >> 2) == hashSb((uint8_t)sbType, (uint8_t)sbOrder)) a += 3;
				if (snEndsWithPoint && (prev.prevFlags & PF_UNK_EF_SF)) a -= 5;
				return a;
			}
			static uint8_t hashSb(uint8_t type, uint8_t order) { return (uint8_t)((((int)type << 1) ^ (type >> 7) ^ order) % 63 + 1); } // PathEvaluator.hpp:83-86
		};

		// ---- the three top-1 containers; `mode` 0 small, 1 medium, 2 hash set -----------------------
		struct Key { int32_t lm; uint8_t prevRoot, sp; };
		static size_t keyHash(const Key& k)   // Hash<WordLL>(lmState, prevRootId, spState) (BestPathContainer.hpp:80-85)
		{
			size_t r = (size_t)(int64_t)k.lm;  // std::hash<int32_t>
			return ((uint16_t)k.prevRoot | ((uint16_t)k.sp << 8)) ^ ((r << 3) | (r >> 61));
		}
		std::vector<WPath> bucket[4];
		std::vector<uint8_t> bucketHash[4];      // low hash byte per entry (BucketedHashContainer::hashes), kept for the global CoNgram model's lookups
		std::vector<WPath> lset;     // mode 2 (large): distinct keys in insertion order

		std::vector<WPath> titems;   // mode 3 (top-N): every inserted path of the current candidate, in insertion order
		PersistentContainers* pc = nullptr;   // faithfulOrder only

		void contClear()
		{
			for (auto& b : bucket) b.clear();
			for (auto& b : bucketHash) b.clear();
			lset.clear(); titems.clear();
			if (pc) { pc->large.clear(); pc->topIndex.clear(); pc->topValues.clear(); }
		}
		void contInsert(int mode, const WPath& np)
		{
			if (mode == 3 && pc && !S.present())
			{
				// BestPathConatiner<topN>::insert (BestPathContainer.hpp:167-203), verbatim semantics
				const size_t topN = cfg.topN;
				auto ins = pc->topIndex.emplace(TopNKey{ np.lmNode, np.prevRootId, np.spState }, std::make_pair((uint32_t)pc->topValues.size(), 1u));
				auto greater = [](const WPath& a, const WPath& b) { return a.accScore > b.accScore; };
				if (ins.second)
				{
					pc->topValues.push_back(np);
					pc->topValues.resize(pc->topValues.size() + topN - 1);
				}
				else
				{
					auto first = pc->topValues.begin() + ins.first->second.first;
					auto last = first + ins.first->second.second;
					if ((size_t)(last - first) < topN)
					{
						*last = np;
						std::push_heap(first, last + 1, greater);
						++ins.first->second.second;
					}
					else if (np.accScore > first->accScore)
					{
						std::pop_heap(first, last, greater);
						*(last - 1) = np;
						std::push_heap(first, last, greater);
					}
				}
				return;
			}
			if (mode == 3) { titems.push_back(np); return; }
			if (mode == 2 && pc)
			{
				// BestPathConatiner<top1>::insert (BestPathContainer.hpp:238-257)
				auto ins = pc->large.emplace(np);
				if (!ins.second && np.accScore > ins.first->accScore) const_cast<WPath&>(*ins.first) = np;
				return;
			}
			if (mode == 2)
			{
				// the reference's large container is a thread_local std::unordered_set that is never shrunk: its iteration order
				// depends on what the thread analysed before.  Restated with insertion order (as for top-N): same paths, same
				// scores; only the hand-on order among the kept paths -- i.e. tie-breaking further down -- can differ.
				for (auto& t : lset)
				{
					if (t.prevRootId == np.prevRootId && t.spState == np.spState && sameLm(t, np))
					{
						if (np.accScore > t.accScore) t = np;
						return;
					}
				}
				lset.push_back(np);
				return;
			}
			size_t h;
			if (C.present())
			{
				// Hash<CoNgramState<0>> = Hash<uint32_t>(node) (src/CoNgramModel.hpp:505-541), then Hash<WordLL>'s mix
				const size_t v = (size_t)(uint32_t)np.lmNode;
				size_t r = (v * (size_t)2305843009213693951ull) ^ ((v << 33) | (v >> 31));
				if (C.window)
				{
					// Hash<CoNgramState<7>> (src/CoNgramModel.hpp:520-532): the last 8 BYTES of history[0..6] read as one word -- four 16-bit or two 32-bit ids
					size_t hh = C.keyBytes == 2 ? ((size_t)(uint16_t)np.hist[3] | ((size_t)(uint16_t)np.hist[4] << 16) | ((size_t)(uint16_t)np.hist[5] << 32) | ((size_t)(uint16_t)np.hist[6] << 48))
						: ((size_t)np.hist[5] | ((size_t)np.hist[6] << 32));
					hh = (hh * (size_t)2305843009213693951ull) ^ ((hh << 31) | (hh >> 33));
					r = hh ^ ((r << 3) | (r >> 61));
				}
				h = ((uint16_t)np.prevRootId | ((uint16_t)np.spState << 8)) ^ ((r << 3) | (r >> 61));
			}
			else if (!S.present()) h = keyHash(Key{ np.lmNode, np.prevRootId, np.spState });
			else
			{
				// Hash<SbgState> (SkipBigramModel.hpp:187-201): Knlm hash folded with the 8 history words, then Hash<WordLL>'s mix
				size_t r = (size_t)(int64_t)np.lmNode;
				for (int i = 0; i < 8; ++i) r = (size_t)np.hist[i] ^ ((r << 3) | (r >> 61));
				h = ((uint16_t)np.prevRootId | ((uint16_t)np.spState << 8)) ^ ((r << 3) | (r >> 61));
			}
			auto& b = bucket[mode == 1 ? ((h >> 8) & 3) : 0];
			if (C.window && b.size() >= 64)
			{
				// BucketedHashContainer::insertOptimized of the SIMD builds (BestPathContainer.hpp:316-384) once a bucket holds 64 entries -- two defects
				// of the reference, reproduced because with the global model equal states (node + history[3..6]) are not identical ones and the survivor's
				// other history words change later scores (with the local model an equal state is the same state: a dominated duplicate changes nothing,
				// and device and oracle keep merging there):
				//   * nst::findAll<sse2 / sse4_1> masks its result with ((size_t)1 << size) - 1 (src/search.cpp:555-597): 0 for size == 64, the shift count
				//     wraps -- entries 0..63 are never candidates again, and with 128 entries none is;
				//   * a candidate position j of the second half (hash byte of entry 64 + j equals the new one) is tested with value[j].equalTo(...) -- the
				//     entry of the FIRST half -- and on success entry 64 + j is the one that is compared by score and overwritten (:341-350).
				auto& hb = bucketHash[mode == 1 ? ((h >> 8) & 3) : 0];
				const size_t n2 = b.size() - 64;
				cnt.congPast64++;
				if (n2 < 64)
					for (size_t bj = 0; bj < n2; ++bj)
					{
						if (hb[64 + bj] != (uint8_t)h) continue;
						const WPath& probe = b[bj];
						if (!(probe.prevRootId == np.prevRootId && probe.spState == np.spState && sameLm(probe, np))) continue;
						WPath& t = b[64 + bj];
						if (np.accScore > t.accScore) { const uint8_t pr = t.prevRootId; t = np; t.prevRootId = pr; }
						return;
					}
				if (b.size() < (mode == 1 ? cfg.bucketCap : 128u)) { b.push_back(np); hb.push_back((uint8_t)h); }
				return;
			}
			for (auto& t : b)
			{
				if (t.prevRootId == np.prevRootId && t.spState == np.spState && sameLm(t, np))
				{
					if (np.accScore > t.accScore) { const uint8_t pr = t.prevRootId; t = np; t.prevRootId = pr; }
					return;
				}
			}
			if (b.size() < (mode == 1 ? cfg.bucketCap : 128u)) { b.push_back(np); bucketHash[mode == 1 ? ((h >> 8) & 3) : 0].push_back((uint8_t)h); }
		}
		template<class Fn> void contEach(int mode, Fn&& fn)
		{
			if (mode == 3 && pc && !S.present())
			{
				for (auto& kv : pc->topIndex) for (uint32_t i = 0; i < kv.second.second; ++i) fn(pc->topValues[kv.second.first + i]);
				return;
			}
			if (mode == 2 && pc) { for (auto& p : pc->large) fn(p); return; }
			if (mode == 3)
			{
				// keep a path iff fewer than N paths of its key beat it (higher score, or equal score and inserted earlier)
				for (size_t i = 0; i < titems.size(); ++i)
				{
					const WPath& a = titems[i];
					uint32_t rank = 0;
					for (size_t j = 0; j < titems.size(); ++j)
					{
						const WPath& b = titems[j];
						if (j == i || !sameTopNKey(a, b)) continue;
						if (b.accScore > a.accScore || (b.accScore == a.accScore && j < i)) ++rank;
					}
					if (rank < cfg.topN) fn(a);
				}
				return;
			}
			if (mode == 2) { for (auto& p : lset) fn(p); return; }
			for (auto& b : bucket) for (auto& p : b) fn(p);
		}

		void evalSingle(int mode, std::vector<WPath>& outv, uint32_t nodeIdx, uint16_t ownFormId, uint32_t morphId, float ignoreCondScore, float nodeLevelDiscount)
		{
			const LNode* node = graph + nodeIdx;
			const MorphRec& cm = M.morphs[morphId];
			const bool single = cm.flags & MF_SINGLE;
			uint32_t firstWid = single ? cm.lmId : M.chunkLm[cm.chunkOff];
			contClear();
			cnt.candMorphs++;
			const float additional = cm.userScore + nodeLevelDiscount + leftBoundary[(hasLeftBoundary(node) ? T_MAX : 0) + clearIrregular(cm.tag)] * 5.f;
			Rule rule;
			rule.special = cm.special;
			rule.sbType = cm.tag == T_SB ? M.sbInfo[morphId] : 0;
			rule.sbOrder = rule.sbType ? cm.senseId : 0;
			rule.vowelE = cm.flags & MF_VOWEL_E; rule.infJ = cm.flags & MF_INF_J; rule.badPairOfL = cm.flags & MF_BAD_PAIR_OF_L;
			rule.positiveE = isEClass(cm.tag) && node->form != NOFORM && (M.forms[node->form].flags & FF_STARTS_WITH_A);
			rule.contractableE = cm.flags & MF_CONTRACTABLE_E;
			rule.snEndsWithPoint = cm.tag == T_SN && node->uformLen && (*norm)[node->uformOff + node->uformLen - 1] == u'.';
			rule.condP = cm.polar;

			const LNode* pfirst = node - node->prev;
			for (const LNode* prev = node->prev ? pfirst : nullptr; prev; prev = prev->sibling ? prev + prev->sibling : nullptr)
			{
				const auto& pc = cache[prev - graph];
				for (uint32_t pi = 0; pi < pc.size(); ++pi)
				{
					const WPath& pp = pc[pi];
					const MorphRec& pm = M.morphs[pp.morph];
					if (pm.tag == T_Z_SIOT && (!isNNClass(cm.tag) || prev->endPos < node->startPos)) continue;
					float cand = pp.accScore + additional;
					float firstChunk = additional;
					if (pp.combineSocket)
					{
						if (pp.combineSocket != cm.socket || single) continue;
						if (prev->endPos < node->startPos)
						{
							if (cfg.spaceTolerance > 0) cand -= cfg.spacePenalty; else continue;
						}
						firstWid = M.morphs[M.morphs[pp.wid].combinedId].lmId;   // persists for later predecessors, as in the reference
					}
					// FormEvaluator (PathEvaluator.hpp:253-311)
					{
						uint16_t feat; bool sscLeft;
						const MorphRec& wm = M.morphs[pp.wid];
						if (pp.ownFormId)
						{
							uint32_t l; const uint16_t* s = ownStr(pp.ownFormId, l);
							feat = featMask(s, l);
							sscLeft = l && identifySpecialChr(s[l - 1]) == T_SSC;
						}
						else if (!(wm.flags & MF_KFORM_EMPTY)) { feat = wm.feat; sscLeft = wm.flags & MF_ENDS_WITH_SSC; }
						else if (pm.tag == T_UNKNOWN && pm.nChunks)
						{
							const MorphRec& lm = M.morphs[M.chunkMorph[pm.chunkOff + pm.nChunks - 1]];
							feat = lm.feat; sscLeft = lm.flags & MF_ENDS_WITH_SSC;
						}
						else { feat = pm.feat; sscLeft = pm.flags & MF_ENDS_WITH_SSC; }
						if (pm.tag == T_SSC || sscLeft) {}
						else if (ignoreCondScore != 0) cand += featTest(feat, cm.vowel, cm.polar) ? 0 : ignoreCondScore;
						else if (!featTest(feat, cm.vowel, cm.polar)) continue;
					}
					WPath lmSt = pp;               // the LM part (Knlm node [+ SkipBigram history]) advances on a copy of the incoming path
					if (cm.socket && single) {}
					else
					{
						if (M.morphs[firstWid].tag == T_P) continue;
						float ll = lmNext(lmSt, firstWid);
						cand += ll; firstChunk += ll;
						if (!single)
						{
							bool bad = false;
							for (uint32_t c = 1; c < cm.nChunks; ++c)
							{
								const uint32_t wid = M.chunkLm[cm.chunkOff + c];
								if (M.morphs[wid].tag == T_P) { bad = true; break; }
								ll = lmNext(lmSt, wid);
								cand += ll;
							}
							if (bad) continue;
						}
					}
					cnt.transitions++;
					// insertToPathContainer (PathEvaluator.hpp:193-251)
					auto insert = [&](uint8_t rootId)
					{
						uint8_t sp = pp.spState;
						if (rootId != COMMON_ROOT) sp = uniqStates[rootId];
						const float rs = rule(M.morphs[pp.wid], sp);
						if (rule.special == 0) sp |= 1; else if (rule.special == 1) sp &= ~1; else if (rule.special == 3) sp |= 2; else if (rule.special == 4) sp &= ~2;
						if (rule.sbType) sp = (uint8_t)((sp & 3) | (Rule::hashSb((uint8_t)rule.sbType, (uint8_t)(rule.sbOrder + 1)) << 2));
						WPath np;
						np.morph = morphId; np.accScore = (cand + rs) - dialectCostOf(morphId); np.firstChunkScore = (firstChunk + rs) - dialectCostOf(morphId);
						np.accTypoCost = pp.accTypoCost + node->typoCost;
						np.parentNode = (int32_t)(prev - graph); np.parentIdx = (int32_t)pi;
						np.lmNode = lmSt.lmNode; np.histPos = lmSt.histPos; for (int hi = 0; hi < 8; ++hi) np.hist[hi] = lmSt.hist[hi];
						np.spState = sp;
						np.rootId = pp.rootId; np.prevRootId = pp.rootId;
						if (rootId != COMMON_ROOT) np.rootId = rootId;
						contInsert(mode, np);
					};
					const bool quote = rule.special == 0 || rule.special == 1 || rule.special == 3 || rule.special == 4;
					if ((rule.sbType || quote) && pp.rootId == COMMON_ROOT) for (uint8_t r = 0; r < uniqStates.size(); ++r) insert(r);
					else insert(COMMON_ROOT);
				}
			}
			contEach(mode, [&](const WPath& p)
			{
				outv.push_back(p);
				WPath& q = outv.back();
				q.wid = cm.lastSeqId;
				if (single) { q.combineSocket = cm.socket; q.ownFormId = ownFormId; }
			});
		}

		// MorphemeEvaluator<CoNgramState>::eval (src/CoNgramModel.cpp:18-318), the candidate side of the transposed PathEvaluator: all regular
		// candidates first (their scores against every socket-free incoming path come from progressMatrix: per pair, the same arithmetic as
		// progress()), then the left halves of split stems, then the right halves; one path container per candidate as in evalSingle.
		void evalCong(int mode, std::vector<WPath>& outv, uint32_t nodeIdx, uint16_t ownFormId, const std::vector<uint32_t>& morphs, float ignoreCondScore, float nodeLevelDiscount)
		{
			const LNode* node = graph + nodeIdx;
			struct Prev { const LNode* node; uint32_t idx; };
			std::vector<Prev> regularPrev, combiningPrev;
			const LNode* pfirst = node - node->prev;
			for (const LNode* prev = node->prev ? pfirst : nullptr; prev; prev = prev->sibling ? prev + prev->sibling : nullptr)
			{
				const auto& pcs = cache[prev - graph];
				for (uint32_t pi = 0; pi < pcs.size(); ++pi) (pcs[pi].combineSocket ? combiningPrev : regularPrev).push_back(Prev{ prev, pi });
			}
			std::vector<uint32_t> regular, regularDistant, combL, combR;
			for (uint32_t mid : morphs)
			{
				const MorphRec& cm = M.morphs[mid];
				const bool single = cm.flags & MF_SINGLE;
				if (cm.socket) { (single ? combL : combR).push_back(mid); continue; }
				const uint32_t firstWid = single ? cm.lmId : M.chunkLm[cm.chunkOff];
				if (M.morphs[firstWid].tag == T_P) continue;
				(C.window && C.distant(firstWid) ? regularDistant : regular).push_back(mid);
			}
			regular.insert(regular.end(), regularDistant.begin(), regularDistant.end());      // valid distant tokens last (src/CoNgramModel.cpp:105-124)
			auto ruleOf = [&](uint32_t morphId)
			{
				const MorphRec& cm = M.morphs[morphId];
				Rule rule;
				rule.special = cm.special;
				rule.sbType = cm.tag == T_SB ? M.sbInfo[morphId] : 0;
				rule.sbOrder = rule.sbType ? cm.senseId : 0;
				rule.vowelE = cm.flags & MF_VOWEL_E; rule.infJ = cm.flags & MF_INF_J; rule.badPairOfL = cm.flags & MF_BAD_PAIR_OF_L;
				rule.positiveE = isEClass(cm.tag) && node->form != NOFORM && (M.forms[node->form].flags & FF_STARTS_WITH_A);
				rule.contractableE = cm.flags & MF_CONTRACTABLE_E;
				rule.snEndsWithPoint = cm.tag == T_SN && node->uformLen && (*norm)[node->uformOff + node->uformLen - 1] == u'.';
				rule.condP = cm.polar;
				return rule;
			};
			// FormEvaluator (PathEvaluator.hpp:253-311): false = the path is excluded; may add ignoreCondScore to `score`
			auto formOk = [&](const WPath& pp, const MorphRec& cm, float& score) -> bool
			{
				const MorphRec& pm = M.morphs[pp.morph];
				uint16_t feat; bool sscLeft;
				const MorphRec& wm = M.morphs[pp.wid];
				if (pp.ownFormId)
				{
					uint32_t l; const uint16_t* sp = ownStr(pp.ownFormId, l);
					feat = featMask(sp, l);
					sscLeft = l && identifySpecialChr(sp[l - 1]) == T_SSC;
				}
				else if (!(wm.flags & MF_KFORM_EMPTY)) { feat = wm.feat; sscLeft = wm.flags & MF_ENDS_WITH_SSC; }
				else if (pm.tag == T_UNKNOWN && pm.nChunks)
				{
					const MorphRec& lm = M.morphs[M.chunkMorph[pm.chunkOff + pm.nChunks - 1]];
					feat = lm.feat; sscLeft = lm.flags & MF_ENDS_WITH_SSC;
				}
				else { feat = pm.feat; sscLeft = pm.flags & MF_ENDS_WITH_SSC; }
				if (pm.tag == T_SSC || sscLeft) return true;
				if (ignoreCondScore != 0) { score += featTest(feat, cm.vowel, cm.polar) ? 0 : ignoreCondScore; return true; }
				return featTest(feat, cm.vowel, cm.polar);
			};
			// insertToPathContainer (PathEvaluator.hpp:193-251)
			auto insertAll = [&](uint32_t morphId, const Rule& rule, const Prev& pr, const WPath& pp, const WPath& lmSt, float cand, float firstChunk)
			{
				cnt.transitions++;
				auto insert = [&](uint8_t rootId)
				{
					uint8_t sp = pp.spState;
					if (rootId != COMMON_ROOT) sp = uniqStates[rootId];
					const float rs = rule(M.morphs[pp.wid], sp);
					if (rule.special == 0) sp |= 1; else if (rule.special == 1) sp &= ~1; else if (rule.special == 3) sp |= 2; else if (rule.special == 4) sp &= ~2;
					if (rule.sbType) sp = (uint8_t)((sp & 3) | (Rule::hashSb((uint8_t)rule.sbType, (uint8_t)(rule.sbOrder + 1)) << 2));
					WPath np;
					np.morph = morphId; np.accScore = (cand + rs) - dialectCostOf(morphId); np.firstChunkScore = (firstChunk + rs) - dialectCostOf(morphId);
					np.accTypoCost = pp.accTypoCost + node->typoCost;
					np.parentNode = (int32_t)(pr.node - graph); np.parentIdx = (int32_t)pr.idx;
					np.lmNode = lmSt.lmNode; np.ctx = lmSt.ctx; for (int hi = 0; hi < 8; ++hi) np.hist[hi] = lmSt.hist[hi];
					np.spState = sp;
					np.rootId = pp.rootId; np.prevRootId = pp.rootId;
					if (rootId != COMMON_ROOT) np.rootId = rootId;
					contInsert(mode, np);
				};
				const bool quote = rule.special == 0 || rule.special == 1 || rule.special == 3 || rule.special == 4;
				if ((rule.sbType || quote) && pp.rootId == COMMON_ROOT) for (uint8_t r = 0; r < uniqStates.size(); ++r) insert(r);
				else insert(COMMON_ROOT);
			};
			auto writeOut = [&](uint32_t morphId)
			{
				const MorphRec& cm = M.morphs[morphId];
				const bool single = cm.flags & MF_SINGLE;
				contEach(mode, [&](const WPath& p)
				{
					outv.push_back(p);
					WPath& q = outv.back();
					q.wid = cm.lastSeqId;
					if (single) { q.combineSocket = cm.socket; q.ownFormId = ownFormId; }
				});
			};
			// Which kernel scores the (incoming path x regular candidate) matrix decides the rounding of every entry.  One path and one candidate:
			// state.next() = progress().  Otherwise progressMatrixNoWindow (src/CoNgramModel.cpp:1495-1611) over the m UNIQUE context ids and the
			// n UNIQUE first word ids: qgemm::scatteredGEMMOpt<sse4_1> (src/qgemm.hpp:157-205; pin = the reference's SSE4.1 build, the simplest
			// dispatch): m <= 3 and n <= 3 -> baseline; n == 1 -> scatteredGEMV (specialised, output scale first) unless m == 8 (scatteredGEMV8x1:
			// baseline there); everything else -> baseline.
			// Global model: progressMatrixWOSort for <= 16 paths and <= 16 candidates (:1470-1480) puts every path's context row, then EVERY non-empty history
			// slot's distant row (no de-duplication) against every candidate's row: m = paths + slots, n = candidates; progressMatrixWSort otherwise:
			// m = unique contexts + unique history words, n = unique first words.
			bool outputFirst = false;
			const bool matrix = !(regularPrev.size() == 1 && regular.size() == 1);
			cnt.congDim = C.dim;
			if (!regularPrev.empty() && !regular.empty())
			{
				std::vector<uint32_t> uc, uw, uh;
				for (const Prev& pr : regularPrev) uc.push_back(cache[pr.node - graph][pr.idx].ctx);
				for (uint32_t mid : regular) { const MorphRec& cm = M.morphs[mid]; uw.push_back((cm.flags & MF_SINGLE) ? cm.lmId : M.chunkLm[cm.chunkOff]); }
				if (C.window) for (const Prev& pr : regularPrev) for (uint32_t k = 0; k < congg::WINDOW; ++k) { const uint32_t t = cache[pr.node - graph][pr.idx].hist[k]; if (t) uh.push_back(t); }
				size_t m, n;
				if (C.window && regularPrev.size() <= 16 && regular.size() <= 16) { m = regularPrev.size() + uh.size(); n = regular.size(); }
				else
				{
					std::sort(uc.begin(), uc.end()); uc.erase(std::unique(uc.begin(), uc.end()), uc.end());
					std::sort(uw.begin(), uw.end()); uw.erase(std::unique(uw.begin(), uw.end()), uw.end());
					std::sort(uh.begin(), uh.end()); uh.erase(std::unique(uh.begin(), uh.end()), uh.end());
					m = uc.size() + uh.size(); n = uw.size();
				}
				cnt.congCtxRows += m; cnt.congOutRows += n; cnt.congScores += regularPrev.size() * regular.size();
				outputFirst = matrix && !(m <= 3 && n <= 3) && n == 1 && m != 8;
			}
			for (uint32_t mid : regular)
			{
				const MorphRec& cm = M.morphs[mid];
				const bool single = cm.flags & MF_SINGLE;
				const uint32_t firstWid = single ? cm.lmId : M.chunkLm[cm.chunkOff];
				const uint32_t length = single ? 1u : cm.nChunks;
				contClear();
				cnt.candMorphs++;
				const Rule rule = ruleOf(mid);
				const float morphScore = cm.userScore + nodeLevelDiscount + leftBoundary[(hasLeftBoundary(node) ? T_MAX : 0) + clearIrregular(cm.tag)] * 5.f;
				for (const Prev& pr : regularPrev)
				{
					const WPath& pp = cache[pr.node - graph][pr.idx];
					WPath lmSt = pp;
					const float ll = congNext(lmSt, firstWid, outputFirst, false, matrix);   // progressMatrix / next(): scores[prev][cur] and the moved-on state
					float score = pp.accScore + morphScore + ll;
					const float firstChunk = morphScore + ll;
					if (!formOk(pp, cm, score)) continue;
					if (M.morphs[pp.morph].tag == T_Z_SIOT && (!isNNClass(cm.tag) || pr.node->endPos < node->startPos)) continue;
					bool bad = false;
					for (uint32_t c = 1; c < length; ++c)
					{
						const uint32_t wid = M.chunkLm[cm.chunkOff + c];
						if (M.morphs[wid].tag == T_P) { bad = true; break; }
						score += congNext(lmSt, wid);
					}
					if (bad) continue;
					insertAll(mid, rule, pr, pp, lmSt, score, firstChunk);
				}
				writeOut(mid);
			}
			for (uint32_t mid : combL)
			{
				const MorphRec& cm = M.morphs[mid];
				contClear();
				cnt.candMorphs++;
				const Rule rule = ruleOf(mid);
				const float morphScore = cm.userScore + nodeLevelDiscount + leftBoundary[(hasLeftBoundary(node) ? T_MAX : 0) + clearIrregular(cm.tag)] * 5.f;
				for (const Prev& pr : regularPrev)
				{
					const WPath& pp = cache[pr.node - graph][pr.idx];
					float score = pp.accScore + morphScore;
					if (!formOk(pp, cm, score)) continue;
					insertAll(mid, rule, pr, pp, pp, score, morphScore);
				}
				writeOut(mid);
			}
			for (uint32_t mid : combR)
			{
				const MorphRec& cm = M.morphs[mid];
				const bool single = cm.flags & MF_SINGLE;
				const uint32_t length = single ? 1u : cm.nChunks;
				contClear();
				cnt.candMorphs++;
				const Rule rule = ruleOf(mid);
				const float morphScore = cm.userScore + nodeLevelDiscount + leftBoundary[(hasLeftBoundary(node) ? T_MAX : 0) + clearIrregular(cm.tag)] * 5.f;
				for (const Prev& pr : combiningPrev)
				{
					const WPath& pp = cache[pr.node - graph][pr.idx];
					float score = pp.accScore + morphScore;
					float firstChunk = 0;
					if (pp.combineSocket != cm.socket || single) continue;
					if (pr.node->endPos < node->startPos)
					{
						if (cfg.spaceTolerance > 0) score -= cfg.spacePenalty; else continue;
					}
					const uint32_t firstWid = M.morphs[M.morphs[pp.wid].combinedId].lmId;
					if (!formOk(pp, cm, score)) continue;
					WPath lmSt = pp;
					score += (firstChunk = congNext(lmSt, firstWid));
					firstChunk += morphScore;
					bool bad = false;
					for (uint32_t c = 1; c < length; ++c)
					{
						const uint32_t wid = M.chunkLm[cm.chunkOff + c];
						if (M.morphs[wid].tag == T_P) { bad = true; break; }
						score += congNext(lmSt, wid);
					}
					if (bad) continue;
					insertAll(mid, rule, pr, pp, lmSt, score, firstChunk);
				}
				writeOut(mid);
			}
		}

		// the transposed PathEvaluator::operator() (src/PathEvaluator.hpp:859-1035): candidate filter, z-coda / z-siot shortcuts FIRST, then the
		// candidate evaluator above; pruning is the caller's (shared with the row-major evaluator)
		void evaluateCongNode(std::vector<WPath>& nCache, uint32_t nodeIdx, uint16_t ownFormId, const uint32_t* cands, uint32_t nCands, float nodeLevelDiscount, int mode)
		{
			const LNode* node = graph + nodeIdx;
			const LNode* pfirst = node - node->prev;
			uint32_t zCoda = 0, zSiot = 0; bool hasZCoda = false, hasZSiot = false;
			std::vector<uint32_t> valid;
			for (uint32_t ci = 0; ci < nCands; ++ci)
			{
				const uint32_t mid = cands[ci];
				const MorphRec& cm = M.morphs[mid];
				if (cfg.splitComplex && (cm.flags & MF_HAS_COMPLEX)) continue;
				if (cm.tag == T_Z_CODA) { zCoda = mid; hasZCoda = true; continue; }
				if (cm.tag == T_Z_SIOT) { zSiot = mid; hasZSiot = true; continue; }
				if (!(cm.flags & MF_SINGLE) && (cm.flags & MF_HA_CONTRACTION) && node->prev && (node - node->prev)->endPos < node->startPos) continue;
				valid.push_back(mid);
			}
			auto shortcut = [&](uint32_t mid, bool coda)
			{
				const MorphRec& cm = M.morphs[mid];
				for (const LNode* prev = node->prev ? pfirst : nullptr; prev; prev = prev->sibling ? prev + prev->sibling : nullptr)
				{
					const auto& pcs = cache[prev - graph];
					for (uint32_t pi = 0; pi < pcs.size(); ++pi)
					{
						const uint8_t lastTag = M.morphs[pcs[pi].wid].tag;
						if (coda ? (!isJClass(lastTag) && !isEClass(lastTag)) : !isNNClass(lastTag)) continue;
						WPath np = pcs[pi];
						np.accScore += cm.userScore * cfg.typoCostWeight;
						np.accTypoCost -= cm.userScore;
						np.parentNode = (int32_t)(prev - graph); np.parentIdx = (int32_t)pi;
						np.morph = cm.lmId; np.wid = cm.lmId;
						nCache.push_back(np);
					}
				}
			};
			for (int ignoreCond = 0; ignoreCond < 2; ++ignoreCond)
			{
				if (hasZCoda) shortcut(zCoda, true);
				if (hasZSiot && (cfg.splitSaisiot || cfg.mergeSaisiot)) shortcut(zSiot, false);
				evalCong(mode, nCache, nodeIdx, ownFormId, valid, ignoreCond ? -10.f : 0.f, nodeLevelDiscount);
				if (!nCache.empty()) break;
			}
		}

		// curDialectCost (src/PathEvaluator.hpp:231)
		float dialectCostOf(uint32_t morphId) const { return (M.morphDialect && M.morphDialect[morphId]) ? cfg.dialectCost : 0.f; }
		void evaluate(uint32_t nodeIdx, uint16_t ownFormId, const uint32_t* cands, uint32_t nCands, float unkDiscount)
		{
			// `if (blocklist && curMorph->hasMorpheme(*blocklist)) continue;` is the first statement of both candidate loops
			// (src/PathEvaluator.hpp:385, 892): a blocked candidate is not a candidate
			std::vector<uint32_t> kept;
			if (cfg.blockBits || M.morphDialect)
			{
				for (uint32_t k = 0; k < nCands; ++k)
				{
					if (cfg.blockBits && ((cfg.blockBits[cands[k] >> 5] >> (cands[k] & 31)) & 1)) continue;
					// `if (curMorph->dialect != Dialect::standard && !(curMorph->dialect & allowedDialect)) continue;` -- the statement after it (:386, 893)
					if (M.morphDialect && M.morphDialect[cands[k]] && !(M.morphDialect[cands[k]] & cfg.allowedDialect)) continue;
					kept.push_back(cands[k]);
				}
				cands = kept.data(); nCands = (uint32_t)kept.size();
			}
			const LNode* node = graph + nodeIdx;
			auto& nCache = cache[nodeIdx];
			float wsDiscount = 0;
			if (!node->uformLen && node->form != NOFORM && M.forms[node->form].len && node->spaceErrors) wsDiscount = -cfg.spacePenalty * node->spaceErrors;
			const float typoDiscount = -node->typoCost * cfg.typoCostWeight;
			const float nodeLevelDiscount = wsDiscount + typoDiscount + unkDiscount;
			size_t totalPrev = 0;
			const LNode* pfirst = node - node->prev;
			for (const LNode* prev = node->prev ? pfirst : nullptr; prev; prev = prev->sibling ? prev + prev->sibling : nullptr) totalPrev += cache[prev - graph].size();
			cnt.maxPrevPaths = std::max<uint64_t>(cnt.maxPrevPaths, totalPrev);
			if (totalPrev > 128) cnt.nodesOver128++;
			if (totalPrev > 512) cnt.nodesOver512++;
			const int mode = cfg.topN > 1 ? 3 : totalPrev <= cfg.smallMax ? 0 : totalPrev <= cfg.mediumMax ? 1 : 2;

			int statIgnore = 0; uint32_t statZ = 0, statReg = 0, statR = 0; const size_t statBefore = nCache.size();
			if (C.present()) evaluateCongNode(nCache, nodeIdx, ownFormId, cands, nCands, nodeLevelDiscount, mode);
			else
			for (int ignoreCond = 0; ignoreCond < 2; ++ignoreCond)
			{
				statIgnore = ignoreCond; statZ = statReg = statR = 0;
				for (uint32_t ci = 0; ci < nCands; ++ci)
				{
					const uint32_t mid = cands[ci];
					const MorphRec& cm = M.morphs[mid];
					if (cfg.splitComplex && (cm.flags & MF_HAS_COMPLEX)) continue;
					if (cm.tag == T_Z_CODA || cm.tag == T_Z_SIOT)
					{
						if (cm.tag == T_Z_SIOT && !(cfg.splitSaisiot || cfg.mergeSaisiot)) continue;
						++statZ;
						for (const LNode* prev = node->prev ? pfirst : nullptr; prev; prev = prev->sibling ? prev + prev->sibling : nullptr)
						{
							const auto& pc = cache[prev - graph];
							for (uint32_t pi = 0; pi < pc.size(); ++pi)
							{
								const uint8_t lastTag = M.morphs[pc[pi].wid].tag;
								if (cm.tag == T_Z_CODA ? (!isJClass(lastTag) && !isEClass(lastTag)) : !isNNClass(lastTag)) continue;
								WPath np = pc[pi];
								np.accScore += cm.userScore * cfg.typoCostWeight;
								np.accTypoCost -= cm.userScore;
								np.parentNode = (int32_t)(prev - graph); np.parentIdx = (int32_t)pi;
								np.morph = cm.lmId; np.wid = cm.lmId;
								nCache.push_back(np);
							}
						}
						continue;
					}
					if (!(cm.flags & MF_SINGLE) && (cm.flags & MF_HA_CONTRACTION) && node->prev && (node - node->prev)->endPos < node->startPos) continue;
					{ ++statReg; const bool q_ = cm.special == 0 || cm.special == 1 || cm.special == 3 || cm.special == 4; statR += ((cm.tag == T_SB && M.sbInfo[mid]) || q_) ? (uint32_t)uniqStates.size() : 1u; }
					evalSingle(mode, nCache, nodeIdx, ownFormId, mid, ignoreCond ? -10.f : 0.f, nodeLevelDiscount);
				}
				if (!nCache.empty()) break;
			}
			const size_t statMid = nCache.size();
			// pruning threshold per root: the N-th best score (-inf while a root has fewer than N paths), PathEvaluator.hpp:475-503
			const size_t N = cfg.topN;
			std::vector<float> maxScores(1 + uniqStates.size(), -INFINITY);
			{
				std::vector<std::vector<float>> best(1 + uniqStates.size());
				for (auto& c : nCache)
				{
					if (M.morphs[c.morph].socket) continue;
					best[c.rootId == COMMON_ROOT ? 0 : c.rootId + 1].push_back(c.accScore);
				}
				for (size_t r = 0; r < best.size(); ++r)
				{
					if (best[r].size() < N) continue;
					std::sort(best[r].begin(), best[r].end(), std::greater<float>{});
					maxScores[r] = best[r][N - 1];
				}
			}
			size_t valid = 0;
			for (size_t i = 0; i < nCache.size(); ++i)
			{
				const size_t r = nCache[i].rootId == COMMON_ROOT ? 0 : nCache[i].rootId + 1;
				if (nCache[i].accScore + cfg.cutOff < maxScores[r]) continue;
				if (valid != i) nCache[valid] = nCache[i];
				valid++;
			}
			nCache.resize(valid);
			if (statFile) fprintf(statFile, "E %u %u %u %zu %d %u %u %u %d %zu %zu %zu\n", statSent, nodeIdx, (unsigned)node->endPos, totalPrev, mode, statReg, statR, statZ, statIgnore, statBefore, statMid, valid);
		}
		FILE* statFile = getenv("KORC_STATS") ? fopen(getenv("KORC_STATS"), "a") : nullptr; uint32_t statSent = 0;

		float unkScore(uint32_t len, bool emojiStart) const { return (emojiStart ? -10.f : 0.f) - (len * cfg.oovRuleScale + cfg.oovRuleBias); }
		// UnkFormScorer::operator(): chrBasedScore (src/UnkFormScorer.cpp:53-66: one model step per UTF-16 unit, </s>, minus the bias) when the
		// character model is in use, else ruleBasedScore (:27-51)
		float unkScoreOf(const uint16_t* s, uint32_t len, bool emojiStart) const
		{
			if (cfg.chr && cfg.substr) return chrFreqScoreOracle(*cfg.chr, *cfg.substr, cfg.freq, cfg.oovChrBias, s, len);      // chrFreqBasedScore (src/UnkFormScorer.cpp:68-121)
			if (cfg.chr) { float sc = kamd::chrScoreHost(*cfg.chr, s, len); sc -= cfg.oovChrBias; return sc; }
			return unkScore(len, emojiStart);
		}

		bool disconnected(std::vector<uint8_t>& reach, uint32_t scanStart) const   // PathEvaluator.hpp:1159-1176
		{
			if (reach[scanStart - 1]) return false;
			std::fill(reach.begin() + scanStart, reach.end(), 0);
			for (uint32_t i = scanStart; i < G; ++i)
			{
				const LNode* nd = graph + i;
				for (const LNode* prev = nd->prev ? nd - nd->prev : nullptr; prev; prev = prev->sibling ? prev + prev->sibling : nullptr)
					if (reach[prev - graph]) { reach[i] = 1; break; }
			}
			return reach[G - 1] == 0;
		}

		void tokenList(std::vector<PathTok>& ret, const WPath* result) const   // generateTokenList
		{
			std::vector<std::pair<const WPath*, uint32_t>> steps;   // (path, node index it lives in)
			{
				const WPath* s = result; int32_t nodeOf = -1;
				for (;;)
				{
					const int32_t pn = s->parentNode, pidx = s->parentIdx;
					if (pn < 0) break;
					const WPath* par = &cache[pn][pidx];
					if (par->parentNode < 0) break;
					steps.emplace_back(par, (uint32_t)pn);
					s = par; (void)nodeOf;
				}
			}
			if (steps.empty()) return;
			const WPath* prev = &cache[steps.back().first->parentNode][steps.back().first->parentIdx];
			const uint32_t vocab = M.h.vocabSize;
			auto unify = [&](uint32_t m) -> uint32_t
			{
				if (m >= vocab || M.morphs[m].combinedId != (int32_t)m) return m;
				return M.morphs[m].lmId;
			};
			for (size_t si = steps.size(); si-- > 0;)
			{
				const WPath* cur = steps[si].first;
				const LNode& g = graph[steps[si].second];
				const MorphRec& mm = M.morphs[cur->morph];
				const float scoreDiff = cur->accScore - prev->accScore;
				float typoDiff = cur->accTypoCost - prev->accTypoCost;
				const bool single = mm.flags & MF_SINGLE;
				const bool saisiotSplit = cfg.splitSaisiot && (mm.flags & MF_SAISIOT);
				const uint32_t numNew = (saisiotSplit || !single) ? mm.nChunks : 1;
				const float firstScore = cur->firstChunkScore + typoDiff * cfg.typoCostWeight;
				const float restScores = numNew > 1 ? (scoreDiff - cur->firstChunkScore) / (numNew - 1) : 0;
				typoDiff /= numNew;
				auto emit = [&](uint32_t morph, const U16& str, uint32_t b, uint32_t e, float sc)
				{
					PathTok t; t.morph = morph; t.str = str; t.begin = b; t.end = e; t.wordScore = sc; t.typoCost = typoDiff; t.typoFormId = 0; t.nodeId = steps[si].second;
					ret.push_back(std::move(t));
				};
				auto chunkTok = [&](uint32_t c, float sc)
				{
					emit(unify(M.chunkMorph[mm.chunkOff + c]), U16{}, g.startPos + M.chunkPos[2 * (mm.chunkOff + c)], g.startPos + M.chunkPos[2 * (mm.chunkOff + c) + 1], sc);
				};
				if (saisiotSplit)
				{
					for (uint32_t c = 0; c < numNew; ++c) chunkTok(c, c == 0 ? firstScore : restScores);
					ret.back().end = g.endPos;
				}
				else if (single)
				{
					U16 own;
					if (cur->ownFormId) { uint32_t l; const uint16_t* s = ownStr(cur->ownFormId, l); own.assign((const char16_t*)s, l); }
					emit(unify(cur->morph), own, g.startPos, g.endPos, firstScore);
				}
				else if (mm.socket)
				{
					PathTok& b = ret.back();
					b.morph = (uint32_t)M.morphs[b.morph].combinedId;
					b.end = g.startPos + M.chunkPos[2 * mm.chunkOff + 1];
					b.wordScore = firstScore; b.typoCost = typoDiff; b.typoFormId = 0;
					for (uint32_t c = 1; c < numNew; ++c) chunkTok(c, restScores);
					ret.back().end = g.endPos;
				}
				else
				{
					for (uint32_t c = 0; c < numNew; ++c) chunkTok(c, c == 0 ? firstScore : restScores);
					ret.back().end = g.endPos;
				}
				prev = cur;
			}
		}

	public:
		BestPathSearch(const ModelView& m, const BestPathConfig& c, Counters& k, const SbgView& sbg = SbgView{}, PersistentContainers* persistent = nullptr, const CongView& cong = CongView{}) : M(m), S(sbg), C(cong), cfg(c), cnt(k)
		{
			if (c.faithfulOrder && persistent)
			{
				pc = persistent;
				if (!pc->histInit)
				{
					WPathSetHash hh; hh.useHist = S.present(); WPathSetEq ee; ee.useHist = S.present();
					pc->large = std::unordered_set<WPath, WPathSetHash, WPathSetEq>{ 0, hh, ee };
					pc->histInit = true;
				}
			}
			// TagSequenceScorer (src/TagUtils.cpp:49-62), weight 5
			for (auto& v : leftBoundary) v = 0;
			leftBoundary[T_NNP] = leftBoundary[T_NP] = leftBoundary[T_IC] = -1; leftBoundary[T_SB] = -3;
			for (uint8_t r = 0; r < T_MAX; ++r) leftBoundary[T_MAX + r] = (isEClass(r) || isJClass(r) || isSuffixTag(r) || r == T_VCP) ? -1.f : 0.f;
			leftBoundary[2 * T_MAX] = 5.f;   // include/kiwi/TagUtils.h:10-12: the member after the table is `weight`
		}

		const std::vector<std::vector<WPath>>& states() const { return cache; }

		void run(std::vector<PathResult>& ret, const U16& normText, const std::vector<uint8_t>& cls, const LNode* g, uint32_t gsize, const std::vector<uint8_t>& prevSpStates)
		{
			norm = &normText; graph = g; G = gsize;
			{ static uint32_t sentCounter = 0; statSent = sentCounter++; }
			cache.assign(G, {}); ownForms.clear();
			std::vector<uint8_t> reach(G, 0);
			uniqStates = prevSpStates;
			std::sort(uniqStates.begin(), uniqStates.end());
			uniqStates.erase(std::unique(uniqStates.begin(), uniqStates.end()), uniqStates.end());
			if (prevSpStates.empty()) uniqStates.push_back(0);

			WPath bos; bos.morph = 0; bos.lmNode = C.present() ? 0 : M.h.bosNode; bos.rootId = COMMON_ROOT;      // CoNgramState(): node 0, context 0
			cache[0].push_back(bos);
			reach[0] = 1;
			const uint32_t unkCands[2] = { T_NNG + 1u, T_NNP + 1u }, unkLCands[1] = { T_NNP + 1u };

			for (uint32_t i = 1; i + 1 < G; ++i)
			{
				const LNode* node = g + i;
				uint16_t ownFormId = 0;
				if (node->uformLen) { ownForms.push_back({ 0, OwnForm{ node->uformOff, node->uformLen } }); ownFormId = (uint16_t)ownForms.size(); }
				auto emojiAt = [&](uint32_t off) { return (cls[off] & 0x80) != 0; };
				if (node->form != NOFORM)
				{
					const FormRec& f = M.forms[node->form];
					evaluate(i, ownFormId, M.formCand + f.candOff, f.candCnt, 0.f);
					// "isPretokenizedNode" (PathEvaluator.hpp:1258-1263) is also true for a form whose only candidate is a chunked UNKNOWN-tag morpheme
					const bool pretokLike = f.candCnt == 1 && M.morphs[M.formCand[f.candOff]].tag == T_UNKNOWN && M.morphs[M.formCand[f.candOff]].nChunks;
					bool allPartial = node->typoCost == 0 && !pretokLike;
					for (uint32_t c = 0; c < f.candCnt && allPartial; ++c)
					{
						const MorphRec& m = M.morphs[M.formCand[f.candOff + c]];
						if (!(m.socket || !(m.flags & MF_SINGLE))) allPartial = false;
					}
					if (allPartial)
					{
						ownForms.push_back({ 1, OwnForm{ node->form, f.len } });
						ownFormId = (uint16_t)ownForms.size();
						const uint16_t* fs = M.formChars + f.charOff;
						const bool emo = f.len && fs[0] >= 0x80 && isEmoji(fs[0], f.len > 1 ? fs[1] : 0);
						evaluate(i, ownFormId, unkLCands, 1, unkScoreOf(fs, f.len, emo));
					}
					bool any = false;
					for (auto& p : cache[i]) if (!p.combineSocket) { any = true; break; }
					reach[i] = any;
					if (disconnected(reach, i + 1))
					{
						if (statFile) fprintf(statFile, "D %u %u\n", statSent, i);
						ownForms.push_back({ 0, OwnForm{ node->startPos, node->endPos - node->startPos } });
						ownFormId = (uint16_t)ownForms.size();
						evaluate(i, ownFormId, unkCands, 2, unkScoreOf((const uint16_t*)norm->data() + node->startPos, node->endPos - node->startPos, emojiAt(node->startPos)));
					}
				}
				else evaluate(i, ownFormId, unkCands, 2, unkScoreOf((const uint16_t*)norm->data() + node->uformOff, node->uformLen, emojiAt(node->uformOff)));
				cnt.statesWritten += cache[i].size();
				if (getenv("KORC_DEBUG")) { fprintf(stderr, "node %u:", i); for (auto& p : cache[i]) fprintf(stderr, " [m%u w%u lm%d s%.9g par(%d,%d) r%u c%u h%u,%u,%u,%u,%u,%u,%u,%u]", p.morph, p.wid, p.lmNode, p.accScore, p.parentNode, p.parentIdx, p.rootId, p.ctx, p.hist[0], p.hist[1], p.hist[2], p.hist[3], p.hist[4], p.hist[5], p.hist[6], p.hist[7]); fprintf(stderr, "\n"); }
			}

			// end node (PathEvaluator.hpp:1320-1357)
			auto& cand = cache[G - 1];
			const LNode* endNode = g + G - 1;
			for (const LNode* prev = endNode->prev ? endNode - endNode->prev : nullptr; prev; prev = prev->sibling ? prev + prev->sibling : nullptr)
			{
				const auto& pc = cache[prev - g];
				for (uint32_t pi = 0; pi < pc.size(); ++pi)
				{
					const WPath& p = pc[pi];
					if (p.combineSocket) continue;
					const MorphRec& pm = M.morphs[p.morph];
					if (!(pm.flags & MF_SINGLE) && pm.nChunks <= (pm.socket ? 2u : 1u) && !matchVowel(nullptr, 0, pm.vowel)) continue;
					if (pm.tag == T_Z_SIOT) continue;
					float c = p.accScore, first = 0;
					WPath lmSt = p;
					if (!cfg.openEnding)
					{
						c += (first = lmNext(lmSt, 1));
						if (p.spState & 1) c -= 2;
						if (p.spState & 2) c -= 2;
					}
					WPath np;
					np.accScore = c; np.firstChunkScore = first; np.accTypoCost = p.accTypoCost;
					np.parentNode = (int32_t)(prev - g); np.parentIdx = (int32_t)pi; np.lmNode = lmSt.lmNode; np.ctx = lmSt.ctx;
					if (p.rootId == COMMON_ROOT)
					{
						for (size_t r = 0; r < uniqStates.size(); ++r) { np.spState = uniqStates[r]; np.rootId = (uint8_t)r; cand.push_back(np); }
					}
					else { np.spState = p.spState; np.rootId = p.rootId; cand.push_back(np); }
					if (getenv("KORC_DEBUG")) fprintf(stderr, "end par(%d,%d) acc %.9g eos %.9g -> %.9g\n", np.parentNode, np.parentIdx, p.accScore, first, c);
				}
			}
			std::sort(cand.begin(), cand.end(), [](const WPath& a, const WPath& b)
			{
				if (a.rootId < b.rootId) return true;
				if (a.rootId > b.rootId) return false;
				if (a.spState < b.spState) return true;
				if (a.spState > b.spState) return false;
				return a.accScore > b.accScore;
			});
			size_t numUniq = 0;
			{
				std::vector<uint32_t> seen;
				for (auto& c : cand) { const uint32_t k = (uint32_t)c.rootId << 8 | c.spState; if (std::find(seen.begin(), seen.end(), k) == seen.end()) seen.push_back(k); }
				numUniq = seen.size();
			}
			ret.clear();
			if (cand.empty()) return;
			const size_t perGroup = (size_t)std::ceil(cfg.topN * 2 / (double)numUniq);
			size_t startIdx = 0;
			uint32_t prevKey = (uint32_t)cand[0].rootId << 8 | cand[0].spState;
			for (size_t i = 0; i < cand.size(); ++i)
			{
				const uint32_t k = (uint32_t)cand[i].rootId << 8 | cand[i].spState;
				if (k != prevKey) { startIdx = i; prevKey = k; }
				if (i - startIdx < perGroup)
				{
					PathResult pr;
					tokenList(pr.path, &cand[i]);
					pr.score = cand[i].accScore; pr.prevState = uniqStates[cand[i].rootId]; pr.curState = cand[i].spState;
					cnt.tokens += pr.path.size();
					ret.push_back(std::move(pr));
				}
			}
			std::sort(ret.begin(), ret.end(), [](const PathResult& a, const PathResult& b) { return a.score > b.score; });
		}
	};
}
