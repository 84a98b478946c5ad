// HIP kernel for the lattice best-path search (Viterbi over morpheme candidates, Knlm scoring) on gfx950.
//
// One wavefront owns one chunk at a time (persistent waves pull chunk ids from an atomic counter) and
// sweeps its lattice in stored order (nodes sorted by end position = topological order).  For a node the
// work items are (candidate morpheme, incoming path[, root]) tuples; 64 of them are scored per step, one
// per lane: side terms + a Knlm back-off walk (dependent global loads, the memory-bound core).  Tuples of
// several candidates are flattened into one batch so lanes stay busy when a node has few incoming paths.
// De-duplication per (candidate, LM state, root, special state) -- the reference's per-morpheme hash
// container (src/BestPathContainer.hpp:279-483) -- is an LDS open-addressing table updated with 64-bit
// ds atomics: max over (score, -insertion index) reproduces "first inserted wins on ties", min over the
// insertion index gives the container's iteration order.  Pruning is a wave max-reduction + ballot
// compaction.  States stream to a per-chunk arena in HBM (40 B each), tokens are produced by a back-trace.
//
// Reference behaviour reproduced: BestPathFinder::findBestPath (src/PathEvaluator.hpp:1178-1419),
// PathEvaluator::operator()/evalSingleMorpheme (:347-635), RuleBasedScorer/insertToPathContainer/
// FormEvaluator (:88-311), generateTokenList (:1038-1157), KnLangModel::progress (src/Knlm.cpp:44-130).
// top-N > 1 is not implemented here (the host refuses it).
#include <hip/hip_runtime.h>
#include "device_types.hpp"
#include "feature.hpp"

namespace kamd
{
	constexpr uint32_t SMALL_Q = 128, SMALL_H = 256;
	constexpr uint32_t BIG_Q = 4096, BIG_H = 8192;
	constexpr uint32_t MAX_BATCH_CANDS = 16;
	constexpr uint64_t HEMPTY = ~0ull;
	constexpr uint16_t LF_STR_SSC = 1u << 13, LF_TAG_SSC = 1u << 15;

	struct QArrays   // per work-item results of one batch (LDS for <= SMALL_Q items, HBM scratch otherwise)
	{
		int32_t* lm; float* score; float* fcs; uint32_t* meta; uint32_t* slot;
		uint64_t* hKey; uint64_t* hBest; uint32_t* hFirst; uint32_t hMask;
	};

	struct CandInfo   // per candidate of the current batch (LDS)
	{
		MorphRec rec; uint32_t morph; uint32_t qOff; uint32_t R; float additional;
		uint16_t leftFeat; uint8_t prevFlags, sbType; uint8_t ruleBits; uint8_t pad[3];
	};
	enum { RB_POSITIVE_E = 1, RB_SN_POINT = 2 };

	__device__ __forceinline__ float asFloat(int32_t v) { return __int_as_float(v); }

	__device__ __forceinline__ bool lmSearch(const ModelView& M, const LmNodeRec& nd, uint32_t key, int32_t& v)
	{
		const uint32_t* k = M.lmKeys + nd.nextOff;
		uint32_t lo = 0, hi = nd.numNexts;
		while (lo < hi)
		{
			const uint32_t mid = (lo + hi) >> 1;
			if (k[mid] < key) lo = mid + 1; else hi = mid;
		}
		if (lo < nd.numNexts && k[lo] == key) { v = M.lmValues[nd.nextOff + lo]; return true; }
		return false;
	}

	// KnLangModel::progress (src/Knlm.cpp:44-130); float additions in the same order
	__device__ float lmProgress(const ModelView& M, int32_t& node, uint32_t next)
	{
		float acc = 0;
		for (;;)
		{
			int32_t v;
			if (node == 0)
			{
				v = M.lmRoot[next];
				if (v == 0) return acc + M.h.unkLl;
			}
			else
			{
				const LmNodeRec nd = M.lmNodes[node];
				if (!lmSearch(M, nd, next, v)) { acc += nd.gamma; node += nd.lower; continue; }
			}
			if (v > 0) { node += v; return acc + M.lmNodes[node].ll; }
			int32_t cur = node;
			for (;;)
			{
				const LmNodeRec nd = M.lmNodes[cur];
				if (!nd.lower) break;
				cur += nd.lower;
				int32_t lv;
				if (lmSearch(M, M.lmNodes[cur], next, lv) && lv > 0) { node = cur + lv; return acc + asFloat(v); }
			}
			node = 0;
			return acc + asFloat(v);
		}
	}

	__device__ __forceinline__ uint32_t orderedFloat(float f)
	{
		const uint32_t u = __float_as_uint(f);
		return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
	}

	__device__ __forceinline__ uint8_t hashSb(uint32_t type, uint32_t order)   // PathEvaluator.hpp:83-86
	{
		type &= 0xFF; order &= 0xFF;
		return (uint8_t)((((int)type << 1) ^ (int)(type >> 7) ^ (int)order) % 63 + 1);
	}

	// RuleBasedScorer::operator() (PathEvaluator.hpp:111-181)
	__device__ __forceinline__ float ruleScore(const CandInfo& c, uint8_t prevFlags, uint8_t sp)
	{
		float a = 0;
		const uint16_t fl = c.rec.flags;
		if ((fl & MF_VOWEL_E) && (prevFlags & PF_IRREGULAR)) a -= 10;
		if ((fl & MF_INF_J) && (prevFlags & PF_INFLECTENDA_NP)) a -= 5;
		if ((fl & MF_BAD_PAIR_OF_L) && (prevFlags & PF_VERB_L)) a -= 7;
		if ((c.ruleBits & RB_POSITIVE_E) && !(prevFlags & PF_POSITIVE_VERB)) a -= 100;
		if ((fl & MF_CONTRACTABLE_E) && (prevFlags & PF_VERB_VOWEL)) a -= 3;
		if (c.rec.polar == CP_NON_ADJ && (prevFlags & PF_VA_OR_XSA)) a -= 10;
		const uint8_t special = c.rec.special;
		if (special <= 2) { if (special != (sp & 1)) a -= 2; }
		else if (special <= 5) { if ((uint8_t)(special - 3) != ((sp >> 1) & 1)) a -= 2; }
		if (c.sbType == 5) a -= 5;
		if (c.sbType && (prevFlags & PF_E_NOT_EF)) a -= 10;
		if (c.sbType && (sp >> 2) == hashSb(c.sbType, c.rec.senseId)) a += 3;
		if ((c.ruleBits & RB_SN_POINT) && (prevFlags & PF_UNK_EF_SF)) a -= 5;
		return a;
	}

	__device__ __forceinline__ uint8_t nextSpState(const CandInfo& c, uint8_t sp)   // PathEvaluator.hpp:222-231
	{
		const uint8_t special = c.rec.special;
		if (special == 0) sp |= 1; else if (special == 1) sp &= ~1; else if (special == 3) sp |= 2; else if (special == 4) sp &= ~2;
		if (c.sbType) sp = (uint8_t)((sp & 3) | (hashSb(c.sbType, (uint32_t)c.rec.senseId + 1) << 2));
		return sp;
	}

	// left-string features a state exposes to its successors (FormEvaluator ctor, PathEvaluator.hpp:261-291), for
	// states that carry no own form: derived from the recorded word id / morpheme.
	__device__ uint16_t leftFeatOfMorph(const ModelView& M, const MorphRec& cm, uint32_t wid)
	{
		const MorphRec wm = M.morphs[wid];
		uint16_t f;
		if (!(wm.flags & MF_KFORM_EMPTY)) f = wm.feat | ((wm.flags & MF_ENDS_WITH_SSC) ? LF_STR_SSC : 0);
		else if (cm.tag == T_UNKNOWN && cm.nChunks)
		{
			const MorphRec lm = M.morphs[M.chunkMorph[cm.chunkOff + cm.nChunks - 1]];
			f = lm.feat | ((lm.flags & MF_ENDS_WITH_SSC) ? LF_STR_SSC : 0);
		}
		else f = cm.feat | ((cm.flags & MF_ENDS_WITH_SSC) ? LF_STR_SSC : 0);
		if (cm.tag == T_SSC) f |= LF_TAG_SSC;
		if (cm.tag == T_Z_SIOT) f |= LF_PREV_ZSIOT;
		return f;
	}

	struct WaveCtx
	{
		const ModelView* M; const SearchParams* P;
		const DevNode* nodes; uint32_t G;
		const uint16_t* str; const uint8_t* cls;
		DevState* st; uint32_t stCap, stTop;
		uint32_t* nodeStOff; uint32_t* nodeStCnt;
		const uint8_t* uniq; uint32_t nUniq;
		uint32_t lane;
		bool overflow, pairOverflow;
	};

	__device__ __forceinline__ uint32_t waveExclusiveCount(uint64_t ballot, uint32_t lane) { return __popcll(ballot & ((1ull << lane) - 1)); }

	// One batch of regular candidates [cands[0..nC)) evaluated against the incoming paths [pBeg, pBeg+nP).
	// mode: 0 small container, 1 medium (4 buckets), 2 large.  Appends the surviving paths to the node's state list.
	__device__ void evalBatch(WaveCtx& X, const QArrays& A, CandInfo* ci, uint32_t nC, uint32_t Qtot,
		uint32_t nodeIdx, uint32_t pBeg, uint32_t nP, bool spaceBefore, float ignoreCondScore, uint8_t ownKind, uint16_t ownFeat, int mode)
	{
		const ModelView& M = *X.M;
		const uint32_t lane = X.lane;
		for (uint32_t h = lane; h <= A.hMask; h += 64) { A.hKey[h] = HEMPTY; A.hBest[h] = 0; A.hFirst[h] = 0xFFFFFFFFu; }
		__threadfence_block();

		// ---- scoring pass -------------------------------------------------------------------------------
		for (uint32_t qb = 0; qb < Qtot; qb += 64)
		{
			const uint32_t q = qb + lane;
			bool valid = q < Qtot;
			uint32_t k = 0;
			if (valid) { while (k + 1 < nC && q >= ci[k + 1].qOff) ++k; }
			float cand = 0, firstChunk = 0; int32_t lmNode = 0; uint8_t rootKey = 0, newRoot = 0, sp = 0;
			if (valid)
			{
				const CandInfo& c = ci[k];
				const MorphRec& cm = c.rec;
				const uint32_t local = q - c.qOff;
				const uint32_t p = local / c.R, r = local % c.R;
				const DevState ps = X.st[pBeg + p];
				const bool single = cm.flags & MF_SINGLE;
				uint32_t firstWid = single ? cm.lmId : M.chunkLm[cm.chunkOff];
				do
				{
					if ((ps.leftFeat & LF_PREV_ZSIOT) && (!isNNClass(cm.tag) || spaceBefore)) { valid = false; break; }
					cand = ps.accScore + c.additional;
					firstChunk = c.additional;
					if (ps.socket)
					{
						if (ps.socket != cm.socket || single) { valid = false; break; }
						if (spaceBefore)
						{
							if (X.P->spaceTol > 0) cand -= X.P->spacePenalty; else { valid = false; break; }
						}
					}
					if (cm.socket && !single)
					{
						// the reference keeps the combined word id of the latest matching split stem for all later predecessors
						// (PathEvaluator.hpp:578-591: `firstWid` is assigned inside the loop and never reset)
						for (int32_t pp = (int32_t)p; pp >= 0; --pp)
						{
							const DevState qs = X.st[pBeg + pp];
							if (!qs.socket || qs.socket != cm.socket) continue;
							if ((qs.leftFeat & LF_PREV_ZSIOT) && (!isNNClass(cm.tag) || spaceBefore)) continue;
							if (spaceBefore && !(X.P->spaceTol > 0)) continue;
							firstWid = M.morphs[M.morphs[qs.wid].combinedId].lmId;
							break;
						}
					}
					// FormEvaluator (PathEvaluator.hpp:293-310)
					if (!(ps.leftFeat & (LF_STR_SSC | LF_TAG_SSC)))
					{
						const bool ok = featTest(ps.leftFeat & 0x1FFF, cm.vowel, cm.polar);
						if (ignoreCondScore != 0) cand += ok ? 0 : ignoreCondScore;
						else if (!ok) { valid = false; break; }
					}
					lmNode = ps.lmNode;
					if (!(cm.socket && single))
					{
						if (M.morphs[firstWid].tag == T_P) { valid = false; break; }
						float ll = lmProgress(M, lmNode, firstWid);
						cand += ll; firstChunk += ll;
						if (!single)
						{
							for (uint32_t ch = 1; ch < cm.nChunks; ++ch)
							{
								const uint32_t wid = M.chunkLm[cm.chunkOff + ch];
								if (M.morphs[wid].tag == T_P) { valid = false; break; }
								ll = lmProgress(M, lmNode, wid);
								cand += ll;
							}
							if (!valid) break;
						}
					}
					// insertToPathContainer (PathEvaluator.hpp:193-251)
					sp = ps.spState;
					rootKey = ps.rootId; newRoot = ps.rootId;
					if (c.R > 1 || (c.R == 1 && (c.sbType || cm.special == 0 || cm.special == 1 || cm.special == 3 || cm.special == 4) && ps.rootId == COMMON_ROOT))
					{
						if (ps.rootId == COMMON_ROOT) { newRoot = (uint8_t)r; sp = X.uniq[r]; }
						else if (r != 0) { valid = false; break; }   // paths already bound to a root are inserted once
					}
					const float rs = ruleScore(c, ps.prevFlags, sp);
					cand = cand + rs; firstChunk = firstChunk + rs;
					sp = nextSpState(c, sp);
				} while (0);
			}
			uint32_t slot = 0;
			if (valid)
			{
				const uint64_t key = (uint64_t)(uint32_t)lmNode | ((uint64_t)((k << 16) | ((uint32_t)rootKey << 8) | sp) << 32);
				uint32_t h = (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> 40) & A.hMask;
				for (;;)
				{
					const uint64_t old = atomicCAS((unsigned long long*)&A.hKey[h], (unsigned long long)HEMPTY, (unsigned long long)key);
					if (old == HEMPTY || old == key) break;
					h = (h + 1) & A.hMask;
				}
				slot = h;
				atomicMax((unsigned long long*)&A.hBest[h], ((unsigned long long)orderedFloat(cand) << 32) | (0xFFFFFFFFu - q));
				atomicMin(&A.hFirst[h], q);
				A.lm[q] = lmNode; A.score[q] = cand; A.fcs[q] = firstChunk;
				A.meta[q] = (uint32_t)newRoot | ((uint32_t)sp << 8) | (k << 16);
			}
			if (q < Qtot) A.slot[q] = valid ? slot : 0xFFFFFFFFu;
		}
		__threadfence_block();

		// ---- emission pass: container iteration order ----------------------------------------------------
		const int nBuckets = mode == 1 ? 4 : 1;
		for (uint32_t k0 = 0; k0 < nC; ++k0)
		{
			// medium mode emits bucket by bucket per candidate; otherwise one sweep over the whole batch
			if (mode != 1 && k0 > 0) break;
			const uint32_t qLo = mode == 1 ? ci[k0].qOff : 0, qHi = mode == 1 ? (k0 + 1 < nC ? ci[k0 + 1].qOff : Qtot) : Qtot;
			for (int b = 0; b < nBuckets; ++b)
			{
				uint32_t emittedInBucket = 0;
				for (uint32_t qb = qLo; qb < qHi; qb += 64)
				{
					const uint32_t q = qb + lane;
					bool rep = false; uint32_t slot = 0xFFFFFFFFu;
					if (q < qHi) { slot = A.slot[q]; rep = slot != 0xFFFFFFFFu && A.hFirst[slot] == q; }
					if (rep && mode == 1)
					{
						// bucket = (h >> 8) & 3 of Hash<WordLL> (BestPathContainer.hpp:80-85, 323)
						const uint64_t key = A.hKey[slot];
						const uint64_t lmv = (uint64_t)(int64_t)(int32_t)(uint32_t)key;
						const uint32_t hi = (uint32_t)(key >> 32);
						const uint64_t hh = (uint64_t)(((hi >> 8) & 0xFF) | ((hi & 0xFF) << 8)) ^ ((lmv << 3) | (lmv >> 61));
						rep = (int)((hh >> 8) & 3) == b;
					}
					const uint64_t bal = __ballot(rep);
					const uint32_t rank = emittedInBucket + waveExclusiveCount(bal, lane);
					const bool keep = rep && (mode == 2 || rank < 128);   // a full bucket drops later keys (BestPathContainer.hpp:363-367)
					const uint64_t kbal = __ballot(keep);
					if (keep)
					{
						const uint32_t pos = X.stTop + waveExclusiveCount(kbal, lane);
						if (pos < X.stCap)
						{
							const uint32_t qw = 0xFFFFFFFFu - (uint32_t)A.hBest[slot];
							const uint32_t meta = A.meta[qw];
							const CandInfo& c = ci[meta >> 16];
							const uint32_t local = qw - c.qOff;
							const uint32_t parent = pBeg + local / c.R;
							const bool single = c.rec.flags & MF_SINGLE;
							DevState ns;
							ns.lmNode = A.lm[qw]; ns.accScore = A.score[qw]; ns.firstChunkScore = A.fcs[qw];
							ns.accTypoCost = X.st[parent].accTypoCost + 0.f;
							ns.parent = parent; ns.morph = c.morph; ns.wid = c.rec.lastSeqId; ns.nodeId = (uint16_t)nodeIdx;
							ns.rootId = (uint8_t)(meta & 0xFF); ns.spState = (uint8_t)((meta >> 8) & 0xFF);
							ns.socket = single ? c.rec.socket : 0;
							ns.ownKind = single ? ownKind : 0; ns.ownNode = (single && ownKind) ? (uint16_t)nodeIdx : 0;
							ns.leftFeat = (single && ownKind) ? (uint16_t)(ownFeat | (c.leftFeat & (LF_TAG_SSC | LF_PREV_ZSIOT))) : c.leftFeat;
							ns.prevFlags = c.prevFlags; ns.pad = 0;
							X.st[pos] = ns;
						}
						else X.overflow = true;
					}
					const uint32_t nk = __popcll(kbal);
					X.stTop += nk;
					emittedInBucket += __popcll(bal);
				}
			}
		}
		X.overflow = __any(X.overflow);
		if (X.stTop > X.stCap) X.stTop = X.stCap;
		__threadfence_block();
	}

	// z_coda / z_siot shortcut (PathEvaluator.hpp:389-432): copies of the qualifying incoming paths
	__device__ void evalZShortcut(WaveCtx& X, const MorphRec& cm, uint32_t nodeIdx, uint32_t pBeg, uint32_t nP)
	{
		const ModelView& M = *X.M;
		const uint32_t newMorph = cm.lmId;
		const MorphRec nm = M.morphs[newMorph];
		const uint16_t lfMorph = leftFeatOfMorph(M, nm, newMorph);
		for (uint32_t pb = 0; pb < nP; pb += 64)
		{
			const uint32_t p = pb + X.lane;
			bool keep = false; DevState ns;
			if (p < nP)
			{
				ns = X.st[pBeg + p];
				const uint8_t lastTag = M.morphs[ns.wid].tag;
				keep = cm.tag == T_Z_CODA ? (isJClass(lastTag) || isEClass(lastTag)) : isNNClass(lastTag);
			}
			const uint64_t bal = __ballot(keep);
			if (keep)
			{
				const uint32_t pos = X.stTop + waveExclusiveCount(bal, X.lane);
				if (pos < X.stCap)
				{
					ns.accScore += cm.userScore * X.P->typoCostWeight;
					ns.accTypoCost -= cm.userScore;
					ns.parent = pBeg + p; ns.morph = newMorph; ns.wid = newMorph; ns.nodeId = (uint16_t)nodeIdx;
					ns.leftFeat = ns.ownKind ? (uint16_t)((ns.leftFeat & (0x1FFF | LF_STR_SSC)) | (lfMorph & (LF_TAG_SSC | LF_PREV_ZSIOT))) : lfMorph;
					ns.prevFlags = nm.prevFlags;
					X.st[pos] = ns;
				}
				else X.overflow = true;
			}
			X.stTop += __popcll(bal);
		}
		X.overflow = __any(X.overflow);
		if (X.stTop > X.stCap) X.stTop = X.stCap;
		__threadfence_block();
	}

	struct NodeEnv { uint32_t pBeg, nP; bool spaceBefore, leftBoundary; };

	// PathEvaluator::operator() (PathEvaluator.hpp:347-512) for one candidate list
	__device__ void evaluateNode(WaveCtx& X, const QArrays& As, const QArrays& Ab, CandInfo* ci, uint32_t nodeIdx, const NodeEnv& E,
		const uint32_t* cands, uint32_t nCands, uint8_t ownKind, uint16_t ownFeat, float unkDiscount, const float* lbTable)
	{
		const ModelView& M = *X.M;
		const SearchParams& P = *X.P;
		const DevNode node = X.nodes[nodeIdx];
		const uint32_t nodeStart = X.nodeStOff[nodeIdx];
		float ws = 0;
		if (!node.uformLen && node.form != NOFORM && M.forms[node.form].len && node.spaceErrors) ws = -P.spacePenalty * (float)node.spaceErrors;
		const float typoDiscount = -0.f * P.typoCostWeight;
		const float nodeLevelDiscount = ws + typoDiscount + unkDiscount;
		const int mode = E.nP <= 128 ? 0 : E.nP <= 512 ? 1 : 2;
		const bool formStartsA = node.form != NOFORM && (M.forms[node.form].flags & FF_STARTS_WITH_A);
		const bool uformEndsPoint = node.uformLen && X.str[node.uformOff + node.uformLen - 1] == u'.';

		for (int ignoreCond = 0; ignoreCond < 2; ++ignoreCond)
		{
			uint32_t c = 0;
			while (c < nCands)
			{
				// ---- gather the next batch of regular candidates (lane 0 decides, all lanes follow) ----------
				uint32_t nC = 0, Qtot = 0; bool zShortcut = false; uint32_t zMorph = 0;
				while (c < nCands && nC < MAX_BATCH_CANDS)
				{
					const uint32_t mid = cands[c];
					const MorphRec cm = M.morphs[mid];
					if (P.splitComplex && (cm.flags & MF_HAS_COMPLEX)) { ++c; continue; }
					if (cm.tag == T_Z_CODA || cm.tag == T_Z_SIOT)
					{
						if (cm.tag == T_Z_SIOT && !(P.splitSaisiot || P.mergeSaisiot)) { ++c; continue; }
						if (nC) break;           // flush the batch first, keep order
						zShortcut = true; zMorph = mid; ++c;
						break;
					}
					if (!(cm.flags & MF_SINGLE) && (cm.flags & MF_HA_CONTRACTION) && node.prev && E.spaceBefore) { ++c; continue; }
					const uint8_t sbType = cm.tag == T_SB ? M.sbInfo[mid] : 0;
					const bool quote = cm.special == 0 || cm.special == 1 || cm.special == 3 || cm.special == 4;
					const uint32_t R = ((sbType || quote) && X.nUniq > 1) ? X.nUniq : 1;
					const uint32_t Q = E.nP * R;
					const uint32_t limit = (mode == 0 && Qtot + Q <= SMALL_Q) ? SMALL_Q : BIG_Q;
					if (nC && (Qtot + Q > SMALL_Q || mode != 0)) break;   // only small nodes share a batch
					if (Q > limit) { X.pairOverflow = true; ++c; continue; }
					if (X.lane == 0)
					{
						CandInfo& o = ci[nC];
						o.rec = cm; o.morph = mid; o.qOff = Qtot; o.R = R;
						o.additional = cm.userScore + nodeLevelDiscount + lbTable[(E.leftBoundary ? T_MAX : 0) + clearIrregular(cm.tag)] * 5.f;
						o.leftFeat = leftFeatOfMorph(M, cm, cm.lastSeqId);
						o.prevFlags = M.morphs[cm.lastSeqId].prevFlags;
						o.sbType = sbType;
						o.ruleBits = ((isEClass(cm.tag) && formStartsA) ? RB_POSITIVE_E : 0) | ((cm.tag == T_SN && uformEndsPoint) ? RB_SN_POINT : 0);
					}
					++nC; Qtot += Q; ++c;
					if (mode != 0 || Qtot > SMALL_Q) break;
				}
				__threadfence_block();
				if (nC)
				{
					if (Qtot <= SMALL_Q) evalBatch(X, As, ci, nC, Qtot, nodeIdx, E.pBeg, E.nP, E.spaceBefore, ignoreCond ? -10.f : 0.f, ownKind, ownFeat, mode);
					else evalBatch(X, Ab, ci, nC, Qtot, nodeIdx, E.pBeg, E.nP, E.spaceBefore, ignoreCond ? -10.f : 0.f, ownKind, ownFeat, mode);
				}
				if (zShortcut) evalZShortcut(X, M.morphs[zMorph], nodeIdx, E.pBeg, E.nP);
			}
			if (X.stTop > nodeStart) break;
		}

		// ---- pruning (PathEvaluator.hpp:475-511): keep paths within cutOff of the best of their root ------
		const uint32_t cnt = X.stTop - nodeStart;
		if (!cnt) return;
		const uint32_t nRootSlots = 1 + X.nUniq;
		// per-root maxima; roots are few, so one wave reduction per root slot
		uint32_t out = 0;
		for (uint32_t rs = 0; rs < nRootSlots; ++rs)
		{
			float mx = -INFINITY; bool anyOfRoot = false;
			for (uint32_t b = 0; b < cnt; b += 64)
			{
				const uint32_t i = b + X.lane;
				if (i < cnt)
				{
					const DevState s = X.st[nodeStart + i];
					const uint32_t slot = s.rootId == COMMON_ROOT ? 0 : s.rootId + 1u;
					if (slot == rs) { anyOfRoot = true; if (!M.morphs[s.morph].socket) mx = fmaxf(mx, s.accScore); }
				}
			}
			for (uint32_t d = 32; d; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d));
			if (!__any(anyOfRoot)) continue;
			// mark survivors of this root: accTypoCost's sign bit is free? no -- use the pad byte as a keep flag
			for (uint32_t b = 0; b < cnt; b += 64)
			{
				const uint32_t i = b + X.lane;
				if (i < cnt)
				{
					DevState* s = &X.st[nodeStart + i];
					const uint32_t slot = s->rootId == COMMON_ROOT ? 0 : s->rootId + 1u;
					if (slot == rs) s->pad = (s->accScore + P.cutOff < mx) ? 0 : 1;
				}
			}
		}
		__threadfence_block();
		for (uint32_t b = 0; b < cnt; b += 64)
		{
			const uint32_t i = b + X.lane;
			DevState s; bool keep = false;
			if (i < cnt) { s = X.st[nodeStart + i]; keep = s.pad != 0; }
			const uint64_t bal = __ballot(keep);
			__threadfence_block();
			if (keep) { s.pad = 0; X.st[nodeStart + out + waveExclusiveCount(bal, X.lane)] = s; }
			out += __popcll(bal);
			__threadfence_block();
		}
		X.stTop = nodeStart + out;
	}

	// libstdc++'s std::sort restated for the end-node candidate list (the reference sorts it with an unstable
	// std::sort, PathEvaluator.hpp:1359-1368; equal keys must land where introsort puts them).  Runs on one lane.
	struct EndCand { float score, fcs, typo; uint32_t parent; uint8_t rootId, sp; uint16_t pad; };
	__device__ __forceinline__ bool endLess(const EndCand& a, const EndCand& b)
	{
		if (a.rootId < b.rootId) return true;
		if (a.rootId > b.rootId) return false;
		if (a.sp < b.sp) return true;
		if (a.sp > b.sp) return false;
		return a.score > b.score;
	}
	__device__ void sortEndCands(EndCand* v, int n)
	{
		// insertion sort == std::sort for n <= 16 (std::__insertion_sort); larger inputs run the same introsort
		// phases: median-of-three quick partitions down to 16-element runs, then one final insertion sort.
		struct Range { int lo, hi, depth; };
		Range stack[48]; int sp = 0;
		if (n > 16)
		{
			int depth = 0; for (int t = n; t > 1; t >>= 1) ++depth; depth *= 2;
			stack[sp++] = Range{ 0, n, depth };
			while (sp)
			{
				Range r = stack[--sp];
				while (r.hi - r.lo > 16)
				{
					if (r.depth == 0)
					{
						// heap sort fallback (std::__partial_sort(first,last,last)): make_heap + sort_heap
						const int len = r.hi - r.lo; EndCand* a = v + r.lo;
						auto sift = [&](int hole, int top, int length, EndCand val)
						{
							int child = hole;
							const int start = hole;
							while (child < (length - 1) / 2)
							{
								child = 2 * (child + 1);
								if (endLess(a[child], a[child - 1])) --child;
								a[hole] = a[child]; hole = child;
							}
							if ((length & 1) == 0 && child == (length - 2) / 2) { child = 2 * (child + 1); a[hole] = a[child - 1]; hole = child - 1; }
							int parent = (hole - 1) / 2;
							while (hole > start && endLess(a[parent], val)) { a[hole] = a[parent]; hole = parent; parent = (hole - 1) / 2; }
							(void)top;
							a[hole] = val;
						};
						for (int parent = (len - 2) / 2; parent >= 0; --parent) sift(parent, parent, len, a[parent]);
						for (int last = len - 1; last > 0; --last) { EndCand val = a[last]; a[last] = a[0]; sift(0, 0, last, val); }
						break;
					}
					--r.depth;
					// __move_median_to_first(first, first+1, mid, last-1)
					EndCand* a = v;
					const int first = r.lo, mid = r.lo + (r.hi - r.lo) / 2, ia = first + 1, ic = r.hi - 1;
					int med;
					if (endLess(a[ia], a[mid])) { if (endLess(a[mid], a[ic])) med = mid; else if (endLess(a[ia], a[ic])) med = ic; else med = ia; }
					else if (endLess(a[ia], a[ic])) med = ia; else if (endLess(a[mid], a[ic])) med = ic; else med = mid;
					{ EndCand t = a[first]; a[first] = a[med]; a[med] = t; }
					// __unguarded_partition(first+1, last, first)
					int i = first + 1, j = r.hi;
					for (;;)
					{
						while (endLess(a[i], a[first])) ++i;
						--j;
						while (endLess(a[first], a[j])) --j;
						if (!(i < j)) break;
						EndCand t = a[i]; a[i] = a[j]; a[j] = t;
						++i;
					}
					stack[sp++] = Range{ i, r.hi, r.depth };
					r.hi = i;
				}
			}
			// __final_insertion_sort: guarded on the first 16, unguarded on the rest
			for (int i = 1; i < 16; ++i)
			{
				EndCand val = v[i];
				if (endLess(val, v[0])) { for (int j = i; j > 0; --j) v[j] = v[j - 1]; v[0] = val; }
				else { int j = i; while (endLess(val, v[j - 1])) { v[j] = v[j - 1]; --j; } v[j] = val; }
			}
			for (int i = 16; i < n; ++i) { EndCand val = v[i]; int j = i; while (endLess(val, v[j - 1])) { v[j] = v[j - 1]; --j; } v[j] = val; }
			return;
		}
		for (int i = 1; i < n; ++i)
		{
			EndCand val = v[i];
			if (endLess(val, v[0])) { for (int j = i; j > 0; --j) v[j] = v[j - 1]; v[0] = val; }
			else { int j = i; while (endLess(val, v[j - 1])) { v[j] = v[j - 1]; --j; } v[j] = val; }
		}
	}

	__device__ uint32_t unifyMorpheme(const ModelView& M, uint32_t m)   // PathEvaluator.hpp:1054-1058
	{
		if (m >= M.h.vocabSize || M.morphs[m].combinedId != (int32_t)m) return m;
		return M.morphs[m].lmId;
	}

	// generateTokenList (PathEvaluator.hpp:1038-1157) for one end candidate; single lane. Returns the token count or -1.
	__device__ int backTrace(const WaveCtx& X, const EndCand& ec, DevToken* out, uint32_t cap)
	{
		const ModelView& M = *X.M;
		// first pass: count steps
		uint32_t nSteps = 0;
		for (uint32_t s = ec.parent; X.st[s].parent != 0xFFFFFFFFu; s = X.st[s].parent) ++nSteps;
		if (!nSteps) return 0;
		// tokens are produced oldest step first: walk the chain once per step (chains are short) -- O(n^2) but n ~ 20
		int nTok = 0;
		uint32_t prevIdx;
		{
			uint32_t s = ec.parent;
			for (uint32_t k = 1; k < nSteps; ++k) s = X.st[s].parent;
			prevIdx = X.st[s].parent;
		}
		for (uint32_t step = nSteps; step-- > 0;)
		{
			uint32_t s = ec.parent;
			for (uint32_t k = 0; k < step; ++k) s = X.st[s].parent;
			const DevState cur = X.st[s];
			const DevState prev = X.st[prevIdx];
			const DevNode g = X.nodes[cur.nodeId];
			const MorphRec mm = M.morphs[cur.morph];
			const float scoreDiff = cur.accScore - prev.accScore;
			float typoDiff = cur.accTypoCost - prev.accTypoCost;
			const bool single = mm.flags & MF_SINGLE;
			const bool saisiotSplit = X.P->splitSaisiot && (mm.flags & MF_SAISIOT);
			const uint32_t numNew = (saisiotSplit || !single) ? mm.nChunks : 1;
			const float firstScore = cur.firstChunkScore + typoDiff * X.P->typoCostWeight;
			const float restScores = numNew > 1 ? (scoreDiff - cur.firstChunkScore) / (float)(numNew - 1) : 0.f;
			typoDiff /= (float)numNew;
			auto emit = [&](uint32_t morph, uint32_t b, uint32_t e, float sc, uint8_t ownKind, uint32_t ownA, uint32_t ownLen) -> bool
			{
				if ((uint32_t)nTok >= cap) return false;
				DevToken t; t.morph = morph; t.begin = (uint16_t)b; t.end = (uint16_t)e; t.wordScore = sc; t.typoCost = typoDiff;
				t.ownKind = ownKind; t.ownA = ownA; t.ownLen = (uint16_t)ownLen; t.pad = 0;
				out[nTok++] = t;
				return true;
			};
			auto chunkTok = [&](uint32_t c, float sc) -> bool
			{
				return emit(unifyMorpheme(M, M.chunkMorph[mm.chunkOff + c]), g.startPos + M.chunkPos[2 * (mm.chunkOff + c)], g.startPos + M.chunkPos[2 * (mm.chunkOff + c) + 1], sc, 0, 0, 0);
			};
			bool ok = true;
			if (saisiotSplit || (!single && !mm.socket))
			{
				for (uint32_t c = 0; c < numNew && ok; ++c) ok = chunkTok(c, c == 0 ? firstScore : restScores);
				if (ok && nTok) out[nTok - 1].end = g.endPos;
			}
			else if (single)
			{
				uint8_t ok2 = cur.ownKind; uint32_t oa = 0, ol = 0;
				if (ok2)
				{
					const DevNode on = X.nodes[cur.ownNode];
					if (ok2 == 1) { oa = on.uformOff; ol = on.uformLen; }
					else if (ok2 == 2) { oa = on.form; ol = M.forms[on.form].len; }
					else { oa = on.startPos; ol = on.endPos - on.startPos; }
				}
				ok = emit(unifyMorpheme(M, cur.morph), g.startPos, g.endPos, firstScore, ok2, oa, ol);
			}
			else
			{
				if (!nTok) return -2;
				DevToken& b = out[nTok - 1];
				b.morph = (uint32_t)M.morphs[b.morph].combinedId;
				b.end = (uint16_t)(g.startPos + M.chunkPos[2 * mm.chunkOff + 1]);
				b.wordScore = firstScore; b.typoCost = typoDiff;
				for (uint32_t c = 1; c < numNew && ok; ++c) ok = chunkTok(c, restScores);
				if (ok) out[nTok - 1].end = g.endPos;
			}
			if (!ok) return -1;
			prevIdx = s;
		}
		return nTok;
	}

	__global__ void __launch_bounds__(64) k_best_path(ModelView M, BatchView B, WorkView W, SearchParams P, uint32_t* chunkCounter, const uint32_t* chunkOrder)
	{
		__shared__ uint64_t sKey[SMALL_H]; __shared__ uint64_t sBest[SMALL_H]; __shared__ uint32_t sFirst[SMALL_H];
		__shared__ int32_t sLm[SMALL_Q]; __shared__ float sScore[SMALL_Q]; __shared__ float sFcs[SMALL_Q]; __shared__ uint32_t sMeta[SMALL_Q]; __shared__ uint32_t sSlot[SMALL_Q];
		__shared__ CandInfo sCand[MAX_BATCH_CANDS];
		__shared__ float sLb[2 * T_MAX + 1];
		__shared__ EndCand sEnd[256];
		__shared__ uint32_t sChunk;

		const uint32_t lane = threadIdx.x;
		// TagSequenceScorer tables (src/TagUtils.cpp:49-62): [0..T_MAX) without, [T_MAX..2*T_MAX) with a left boundary
		// tag PA (== T_MAX) indexes one past a row in the reference (include/kiwi/TagUtils.h:10-18): row 0 spills into
		// row 1, row 1 spills into the `weight` member (5.0) -- reproduced by the flat layout plus one extra slot.
		for (uint32_t t = lane; t < 2 * T_MAX + 1; t += 64)
		{
			float v = 0;
			if (t == 2 * T_MAX) v = 5.f;
			else if (t < T_MAX) { if (t == T_NNP || t == T_NP || t == T_IC) v = -1; else if (t == T_SB) v = -3; }
			else { const uint8_t r = (uint8_t)(t - T_MAX); v = (isEClass(r) || isJClass(r) || isSuffixTag(r) || r == T_VCP) ? -1.f : 0.f; }
			sLb[t] = v;
		}
		// LDS addresses must not be folded into a constant aggregate (lld rejects addrspacecasts in static initialisers)
		uint32_t opaqueZero = 0;
		asm volatile("" : "+v"(opaqueZero));
		QArrays As;
		As.lm = sLm + opaqueZero; As.score = sScore + opaqueZero; As.fcs = sFcs + opaqueZero; As.meta = sMeta + opaqueZero; As.slot = sSlot + opaqueZero;
		As.hKey = sKey + opaqueZero; As.hBest = sBest + opaqueZero; As.hFirst = sFirst + opaqueZero; As.hMask = SMALL_H - 1;
		QArrays Ab;
		{
			uint8_t* base = W.bigScratch + (size_t)blockIdx.x * W.bigScratchBytes;
			Ab.hKey = (uint64_t*)base; base += sizeof(uint64_t) * BIG_H;
			Ab.hBest = (uint64_t*)base; base += sizeof(uint64_t) * BIG_H;
			Ab.hFirst = (uint32_t*)base; base += sizeof(uint32_t) * BIG_H;
			Ab.lm = (int32_t*)base; base += 4 * BIG_Q; Ab.score = (float*)base; base += 4 * BIG_Q; Ab.fcs = (float*)base; base += 4 * BIG_Q;
			Ab.meta = (uint32_t*)base; base += 4 * BIG_Q; Ab.slot = (uint32_t*)base;
			Ab.hMask = BIG_H - 1;
		}
		__syncthreads();

		for (;;)
		{
			if (lane == 0) sChunk = atomicAdd(chunkCounter, 1u);
			__syncthreads();
			const uint32_t ci = sChunk;
			__syncthreads();
			if (ci >= B.nChunks) break;
			const uint32_t chunk = chunkOrder ? chunkOrder[ci] : ci;
			DevChunkResult* res = &W.results[chunk];
			if (res->status != CS_OK) { if (lane == 0) res->nPaths = 0; continue; }

			const uint32_t cOff = B.charOff[chunk];
			WaveCtx X;
			X.M = &M; X.P = &P; X.lane = lane;
			X.nodes = W.nodes + W.nodeBase[chunk]; X.G = W.nNodes[chunk];
			X.str = B.chars + cOff; X.cls = B.cls + cOff;
			X.st = W.states + W.stateBase[chunk]; X.stCap = (uint32_t)(W.stateBase[chunk + 1] - W.stateBase[chunk]); X.stTop = 0;
			X.nodeStOff = W.nodeStateOff + W.nodeBase[chunk]; X.nodeStCnt = W.nodeStateCnt + W.nodeBase[chunk];
			X.uniq = B.spStates + B.spOff[chunk]; X.nUniq = B.spOff[chunk + 1] - B.spOff[chunk];
			X.overflow = false; X.pairOverflow = false;
			const uint32_t G = X.G;
			const bool openEnding = B.chunkFlags[chunk] & 1;

			// start node (PathEvaluator.hpp:1224-1226)
			if (lane == 0)
			{
				DevState bos;
				bos.lmNode = M.h.bosNode; bos.accScore = 0; bos.firstChunkScore = 0; bos.accTypoCost = 0; bos.parent = 0xFFFFFFFFu;
				bos.morph = 0; bos.wid = 0; bos.nodeId = 0; bos.rootId = COMMON_ROOT; bos.spState = 0; bos.socket = 0; bos.ownKind = 0;
				const MorphRec m0 = M.morphs[0];
				bos.leftFeat = leftFeatOfMorph(M, m0, 0); bos.prevFlags = m0.prevFlags; bos.pad = 0; bos.ownNode = 0;
				X.st[0] = bos;
				X.nodeStOff[0] = 0; X.nodeStCnt[0] = 1;
			}
			X.stTop = 1;
			__threadfence_block();

			uint8_t* reach = W.reach + W.nodeBase[chunk];
			for (uint32_t k = lane; k < G; k += 64) reach[k] = k == 0 ? 1 : 0;
			__threadfence_block();
			const uint32_t unkCands[2] = { T_NNG + 1u, T_NNP + 1u };
			const uint32_t unkLCands[1] = { T_NNP + 1u };

			for (uint32_t i = 1; i + 1 < G; ++i)
			{
				const DevNode node = X.nodes[i];
				NodeEnv E;
				const uint32_t firstPrev = i - node.prev;
				uint32_t lastPrev = firstPrev;
				while (X.nodes[lastPrev].sibling) lastPrev += X.nodes[lastPrev].sibling;
				E.pBeg = X.nodeStOff[firstPrev];
				E.nP = X.nodeStOff[lastPrev] + X.nodeStCnt[lastPrev] - E.pBeg;
				const DevNode pn = X.nodes[firstPrev];
				E.spaceBefore = pn.endPos < node.startPos;
				// hasLeftBoundary (PathEvaluator.hpp:24-44)
				E.leftBoundary = false;
				if (firstPrev == 0 || pn.endPos == 0 || E.spaceBefore) E.leftBoundary = true;
				else if (pn.uformLen)
				{
					const uint32_t lp = pn.uformOff + pn.uformLen - 1;
					const uint16_t c = X.str[lp];
					// character type of the single UTF-16 unit (a trailing low surrogate types as SH, src/Utils.cpp:181)
					const uint8_t tag = (isLowSurrogate(c) || isHighSurrogate(c)) ? (uint8_t)T_SH : (uint8_t)(X.cls[lp] & 0x3F);
					if (tag == T_SSC || c == u'"' || c == u'\'') E.leftBoundary = false;
					else if (T_SF <= tag && tag <= T_SB) E.leftBoundary = true;
				}
				if (lane == 0) X.nodeStOff[i] = X.stTop;
				__threadfence_block();

				auto unkScore = [&](uint32_t len, bool emoji) { return (emoji ? -10.f : 0.f) - ((float)len * P.oovRuleScale + P.oovRuleBias); };
				uint8_t ownKind = 0; uint16_t ownFeat = 0;
				auto setOwn = [&](uint8_t kind, const uint16_t* s, uint32_t len, bool typeFromCls, uint32_t clsOff)
				{
					ownKind = kind;
					ownFeat = featMask(s, len) & 0x1FFF;
					if (len)
					{
						const uint16_t c = s[len - 1];
						uint8_t tag;
						if (typeFromCls) tag = (isLowSurrogate(c) || isHighSurrogate(c)) ? (uint8_t)T_SH : (uint8_t)(X.cls[clsOff + len - 1] & 0x3F);
						else tag = T_MAX;   // dictionary form strings never end in a closing bracket unless typed so below
						if (tag == T_SSC) ownFeat |= LF_STR_SSC;
					}
				};
				if (node.uformLen) setOwn(1, X.str + node.uformOff, node.uformLen, true, node.uformOff);

				if (node.form != NOFORM)
				{
					const FormRec f = M.forms[node.form];
					evaluateNode(X, As, Ab, sCand, i, E, M.formCand + f.candOff, f.candCnt, ownKind, ownFeat, 0.f, sLb);
					const bool pretokLike = f.candCnt == 1 && M.morphs[M.formCand[f.candOff]].tag == T_UNKNOWN && M.morphs[M.formCand[f.candOff]].nChunks;
					bool allPartial = !pretokLike;
					for (uint32_t c = 0; c < f.candCnt && allPartial; ++c)
					{
						const MorphRec m = M.morphs[M.formCand[f.candOff + c]];
						if (!(m.socket || !(m.flags & MF_SINGLE))) allPartial = false;
					}
					if (allPartial)
					{
						const uint16_t* fs = M.formChars + f.charOff;
						ownKind = 2; ownFeat = featMask(fs, f.len) & 0x1FFF;
						if (f.flags & FF_ENDS_WITH_SSC) ownFeat |= LF_STR_SSC;
						evaluateNode(X, As, Ab, sCand, i, E, unkLCands, 1, ownKind, ownFeat, unkScore(f.len, false), sLb);
					}
					// reachable[i] and the forward reachability scan (PathEvaluator.hpp:1159-1176, 1286-1299)
					const uint32_t cntNow = X.stTop - X.nodeStOff[i];
					if (lane == 0) X.nodeStCnt[i] = cntNow;
					__threadfence_block();
					bool any = false;
					for (uint32_t b = 0; b < cntNow; b += 64) { const uint32_t k = b + lane; if (k < cntNow && !X.st[X.nodeStOff[i] + k].socket) any = true; }
					any = __any(any);
					if (lane == 0) reach[i] = any ? 1 : 0;
					if (!any)
					{
						// isDisconnected (PathEvaluator.hpp:1159-1176): forward re-scan of the persistent reachability flags
						bool disconnected;
						if (lane == 0)
						{
							for (uint32_t k = i + 1; k < G; ++k)
							{
								const DevNode nk = X.nodes[k];
								uint8_t r = 0;
								if (nk.prev)
								{
									for (uint32_t pj = k - nk.prev;;)
									{
										if (reach[pj]) { r = 1; break; }
										const uint32_t sb = X.nodes[pj].sibling;
										if (!sb) break;
										pj += sb;
									}
								}
								reach[k] = r;
							}
							sChunk = reach[G - 1] ? 0 : 1;
						}
						__syncthreads();
						disconnected = sChunk != 0;
						__syncthreads();
						if (disconnected)
						{
							const uint32_t len = node.endPos - node.startPos;
							setOwn(3, X.str + node.startPos, len, true, node.startPos);
							evaluateNode(X, As, Ab, sCand, i, E, unkCands, 2, ownKind, ownFeat, unkScore(len, (X.cls[node.startPos] & 0x80) != 0), sLb);
						}
					}
				}
				else evaluateNode(X, As, Ab, sCand, i, E, unkCands, 2, ownKind, ownFeat, unkScore(node.uformLen, (X.cls[node.uformOff] & 0x80) != 0), sLb);
				if (lane == 0) X.nodeStCnt[i] = X.stTop - X.nodeStOff[i];
				__threadfence_block();
				if (X.overflow || X.pairOverflow) break;
			}
			if (X.overflow || X.pairOverflow)
			{
				if (lane == 0) { res->status = X.overflow ? CS_ERR_STATE_OVERFLOW : CS_ERR_PAIR_OVERFLOW; res->nPaths = 0; }
				continue;
			}

			// ---- end node (PathEvaluator.hpp:1320-1357): EOS transition for every surviving path ------------
			uint32_t nEnd = 0; bool endOverflow = false;
			{
				const DevNode en = X.nodes[G - 1];
				const uint32_t firstPrev = G - 1 - en.prev;
				uint32_t lastPrev = firstPrev;
				if (en.prev) while (X.nodes[lastPrev].sibling) lastPrev += X.nodes[lastPrev].sibling;
				const uint32_t pBeg = X.nodeStOff[firstPrev];
				const uint32_t nP = en.prev ? X.nodeStOff[lastPrev] + X.nodeStCnt[lastPrev] - pBeg : 0;
				for (uint32_t pb = 0; pb < nP; pb += 64)
				{
					const uint32_t p = pb + lane;
					bool ok = false; DevState ps; float c = 0, first = 0;
					if (p < nP)
					{
						ps = X.st[pBeg + p];
						const MorphRec pm = M.morphs[ps.morph];
						ok = !ps.socket;
						if (ok && !(pm.flags & MF_SINGLE) && pm.nChunks <= (pm.socket ? 2u : 1u) && pm.vowel != CV_NONE) ok = false;   // isMatched(nullptr, vowel)
						if (ok && pm.tag == T_Z_SIOT) ok = false;
						if (ok)
						{
							c = ps.accScore;
							if (!openEnding)
							{
								int32_t ln = ps.lmNode;
								first = lmProgress(M, ln, 1);
								c += first;
								if (ps.spState & 1) c -= 2;
								if (ps.spState & 2) c -= 2;
							}
						}
					}
					const uint32_t mult = (ok && ps.rootId == COMMON_ROOT) ? X.nUniq : (ok ? 1u : 0u);
					// exclusive scan of mult across lanes (paths keep their order; a common-root path expands to one entry per root)
					uint32_t incl = mult;
					for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d); if (lane >= d) incl += v; }
					const uint32_t base = nEnd + incl - mult;
					for (uint32_t r = 0; r < mult; ++r)
					{
						if (base + r < 256)
						{
							EndCand e; e.score = c; e.fcs = first; e.typo = ps.accTypoCost; e.parent = pBeg + p; e.pad = 0;
							if (ps.rootId == COMMON_ROOT) { e.rootId = (uint8_t)r; e.sp = X.uniq[r]; } else { e.rootId = ps.rootId; e.sp = ps.spState; }
							sEnd[base + r] = e;
						}
						else endOverflow = true;
					}
					nEnd += __shfl(incl, 63);
				}
			}
			endOverflow = __any(endOverflow);
			__syncthreads();
			if (lane == 0)
			{
				uint32_t status = CS_OK; uint32_t nPaths = 0;
				if (endOverflow) status = CS_ERR_PATH_OVERFLOW;
				else
				{
					sortEndCands(sEnd, (int)nEnd);
					// group bookkeeping (PathEvaluator.hpp:1380-1413)
					uint32_t numUniq = 0;
					for (uint32_t a = 0; a < nEnd; ++a)
					{
						bool seen = false;
						for (uint32_t b = 0; b < a && !seen; ++b) seen = sEnd[b].rootId == sEnd[a].rootId && sEnd[b].sp == sEnd[a].sp;
						if (!seen) ++numUniq;
					}
					const uint32_t perGroup = numUniq ? (2 + numUniq - 1) / numUniq : 0;   // ceil(topN*2 / numUniq), topN = 1
					DevToken* tok = W.tokens + W.tokenBase[chunk];
					const uint32_t tokCap = (uint32_t)(W.tokenBase[chunk + 1] - W.tokenBase[chunk]);
					uint32_t tokTop = 0, startIdx = 0;
					for (uint32_t a = 0; a < nEnd && status == CS_OK; ++a)
					{
						if (a && (sEnd[a].rootId != sEnd[a - 1].rootId || sEnd[a].sp != sEnd[a - 1].sp)) startIdx = a;
						if (a - startIdx >= perGroup) continue;
						if (nPaths >= kMaxPathsPerChunk) { status = CS_ERR_PATH_OVERFLOW; break; }
						const int nt = backTrace(X, sEnd[a], tok + tokTop, tokCap - tokTop);
						if (nt < 0) { status = CS_ERR_TOKEN_OVERFLOW; break; }
						DevPathHeader& ph = res->paths[nPaths++];
						ph.score = sEnd[a].score; ph.tokOff = tokTop; ph.nTokens = (uint16_t)nt;
						ph.prevState = X.uniq[sEnd[a].rootId]; ph.curState = sEnd[a].sp;
						tokTop += (uint32_t)nt;
					}
				}
				res->status = status; res->nPaths = status == CS_OK ? nPaths : 0;
			}
			__syncthreads();
		}
	}
}
