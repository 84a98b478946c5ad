// HIP kernel for the lattice best-path search (Viterbi over morpheme candidates, Knlm scoring) on gfx950.
//
// Work decomposition.  A lattice node has on average < 2 candidate morphemes and < 5 incoming paths, so a
// whole wavefront per chunk leaves > 90 % of its lanes idle and the search is bound by the latency of its
// dependent loads (Knlm back-off walk, state and morpheme records), not by bandwidth.  The kernel therefore
// splits every 64-wide wavefront into NG = 64/G independent *lane groups* of G lanes; each group owns one
// chunk at a time (pulled longest-first from an atomic counter, so the groups of a wave see similar work) and
// sweeps its lattice in stored order (nodes sorted by end position = topological order).  Within a group the
// work items of a node -- (candidate morpheme, incoming path[, root]) tuples, flattened over consecutive
// candidates -- are scored G at a time, one per lane: side terms + the Knlm walk.  That puts 64 independent
// dependent-load chains in flight per wave instead of ~5.
//
// De-duplication per (candidate, LM state, root, special state) -- the reference's per-morpheme hash
// container (src/BestPathContainer.hpp:279-483) -- works on the scored items staged in LDS: an item is the
// representative of its key iff no earlier item of the same candidate carries the key; the winner of a key is
// the first item with the maximal score ("first inserted wins on ties").  Items are few (<= 64 per batch), so
// this is a short broadcast scan of LDS, no atomics.  Group ballots (slices of the wave ballot) give container
// order; pruning is a group max-reduction + ballot compaction.  States stream to a per-chunk arena in HBM
// (40 B each); the end node, the restated std::sort, group selection and the back-trace run on the group's
// first lane and emit 24-byte tokens.
//
// Reference behaviour reproduced: BestPathFinder::findBestPath (src/PathEvaluator.hpp:1178-1419),
// PathEvaluator::operator()/evalSingleMorpheme (:347-635), RuleBasedScorer/insertToPathContainer/
// FormEvaluator (:88-311), generateTokenList (:1038-1157), KnLangModel::progress (src/Knlm.cpp:44-130).
// top-N > 1 is not implemented here (the host refuses it).
#include <hip/hip_runtime.h>
#include "device_types.hpp"
#include "feature.hpp"
#include "viterbi_kernel.hpp"

namespace kamd
{
	constexpr uint64_t KINVALID = ~0ull;

	struct CandInfo   // per candidate of the current batch (LDS)
	{
		MorphRec rec; uint32_t morph; uint32_t qOff; uint32_t R; float additional;
		uint16_t leftFeat; uint8_t prevFlags, sbType; uint8_t ruleBits; uint8_t pad[3];
	};
	enum { RB_POSITIVE_E = 1, RB_SN_POINT = 2 };

	__device__ __forceinline__ float asFloat(int32_t v) { return __int_as_float(v); }

	__device__ __forceinline__ bool lmSearch(const ModelView& M, uint32_t nextOff, uint32_t numNexts, uint32_t key, int32_t& v)
	{
		const uint32_t* k = M.lmKeys + nextOff;
		uint32_t lo = 0, hi = numNexts;
		while (lo < hi)
		{
			const uint32_t mid = (lo + hi) >> 1;
			if (k[mid] < key) lo = mid + 1; else hi = mid;
		}
		if (lo < numNexts && k[lo] == key) { v = M.lmValues[nextOff + lo]; return true; }
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
				if (!lmSearch(M, nd.nextOff, nd.numNexts, next, v)) { acc += nd.gamma; node += nd.lower; continue; }
			}
			if (v > 0) { node += v; return acc + M.lmNodes[node].ll; }
			int32_t cur = node;
			for (;;)
			{
				const int32_t lower = M.lmNodes[cur].lower;
				if (!lower) break;
				cur += lower;
				const LmNodeRec nd = M.lmNodes[cur];
				int32_t lv;
				if (lmSearch(M, nd.nextOff, nd.numNexts, next, lv) && lv > 0) { node = cur + lv; return acc + asFloat(v); }
			}
			node = 0;
			return acc + asFloat(v);
		}
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

	__device__ __forceinline__ bool isQuoteOrBullet(const CandInfo& c)
	{
		const uint8_t s = c.rec.special;
		return c.sbType || s == 0 || s == 1 || s == 3 || s == 4;
	}

	// ---------------------------------------------------------------------------------------------------
	template<int G>
	struct GroupCtx
	{
		static constexpr uint64_t GMASK = G == 64 ? ~0ull : ((1ull << G) - 1);
		const ModelView* M; const SearchParams* P; const float* lb;
		uint32_t gl, gshift;
		const DevNode* nodes; uint32_t Gn;
		const uint16_t* str; const uint8_t* cls;
		DevState* st; uint32_t stCap, stTop;
		uint32_t* nodeStOff; uint32_t* nodeStCnt;
		const uint8_t* uniq; uint32_t nUniq;
		bool overflow, pairOverflow;
		uint64_t* qKey; float* qScore; float* qFcs; CandInfo* ci;   // LDS, this group's slices
		GroupScratch* scratch;

		__device__ __forceinline__ uint64_t ballot(bool p) const { return (__ballot(p) >> gshift) & GMASK; }
		__device__ __forceinline__ bool any(bool p) const { return ballot(p) != 0; }
		__device__ __forceinline__ uint32_t prefix(uint64_t b) const { return __popcll(b & ((1ull << gl) - 1)); }
		template<class T> __device__ __forceinline__ T bcast(T v, int srcLane) const { return __shfl(v, srcLane, G); }
	};

	struct NodeEnv { uint32_t pBeg, nP; bool spaceBefore, leftBoundary, formStartsA, uformEndsPoint; };

	// One batch of regular candidates ci[0..nC) against the incoming paths [pBeg, pBeg+nP): Qtot work items.
	// mode: 0 small container, 1 medium (4 hash buckets), 2 large (PathEvaluator.hpp:447-466).
	template<int G>
	__device__ void evalBatch(GroupCtx<G>& X, uint32_t nC, uint32_t Qtot, uint32_t nodeIdx, const NodeEnv& E,
		float ignoreCondScore, uint8_t ownKind, uint16_t ownFeat, int mode)
	{
		const ModelView& M = *X.M;
		const bool big = Qtot > QCAP;
		uint64_t* qKey = big ? X.scratch->key : X.qKey;
		float* qScore = big ? X.scratch->score : X.qScore;
		float* qFcs = big ? X.scratch->fcs : X.qFcs;
		const uint32_t pBeg = E.pBeg;

		// ---- scoring pass: one work item per lane -------------------------------------------------------
		for (uint32_t qb = 0; qb < Qtot; qb += G)
		{
			const uint32_t q = qb + X.gl;
			bool valid = q < Qtot;
			uint32_t k = 0;
			if (valid) { while (k + 1 < nC && q >= X.ci[k + 1].qOff) ++k; }
			float cand = 0, firstChunk = 0; int32_t lmNode = 0; uint8_t rootKey = 0, sp = 0;
			if (valid)
			{
				const CandInfo& c = X.ci[k];
				const MorphRec& cm = c.rec;
				const uint32_t local = q - c.qOff;
				const uint32_t p = local / c.R, r = local % c.R;
				const DevState ps = X.st[pBeg + p];
				const bool single = cm.flags & MF_SINGLE;
				uint32_t firstWid = single ? cm.lmId : M.chunkLm[cm.chunkOff];
				do
				{
					if ((ps.leftFeat & LF_PREV_ZSIOT) && (!isNNClass(cm.tag) || E.spaceBefore)) { valid = false; break; }
					cand = ps.accScore + c.additional;
					firstChunk = c.additional;
					if (ps.socket)
					{
						if (ps.socket != cm.socket || single) { valid = false; break; }
						if (E.spaceBefore)
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
							if ((qs.leftFeat & LF_PREV_ZSIOT) && (!isNNClass(cm.tag) || E.spaceBefore)) continue;
							if (E.spaceBefore && !(X.P->spaceTol > 0)) continue;
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
					rootKey = ps.rootId;
					if (isQuoteOrBullet(c))
					{
						if (ps.rootId == COMMON_ROOT) sp = X.uniq[r];
						else if (r != 0) { valid = false; break; }   // a path already bound to a root is inserted once
					}
					const float rs = ruleScore(c, ps.prevFlags, sp);
					cand = cand + rs; firstChunk = firstChunk + rs;
					sp = nextSpState(c, sp);
				} while (0);
			}
			if (q < Qtot)
			{
				// key: LM node | new special state | previous root | candidate ; r is recoverable from q
				qKey[q] = valid ? ((uint64_t)(uint32_t)lmNode | ((uint64_t)sp << 32) | ((uint64_t)rootKey << 40) | ((uint64_t)k << 48)) : KINVALID;
				qScore[q] = cand; qFcs[q] = firstChunk;
			}
		}
		__threadfence_block();

		// ---- emission pass: representatives in container iteration order, each carrying its key's winner ----
		const int nBuckets = mode == 1 ? 4 : 1;
		for (int b = 0; b < nBuckets; ++b)
		{
			uint32_t emittedInBucket = 0;
			for (uint32_t qb = 0; qb < Qtot; qb += G)
			{
				const uint32_t q = qb + X.gl;
				bool rep = false; uint32_t qw = q; uint64_t key = KINVALID; uint32_t k = 0;
				if (q < Qtot) key = qKey[q];
				if (key != KINVALID)
				{
					k = (uint32_t)(key >> 48);
					const uint32_t lo = X.ci[k].qOff, hi = (k + 1 < nC) ? X.ci[k + 1].qOff : Qtot;
					rep = true;
					float best = -INFINITY; bool haveBest = false;
					for (uint32_t j = lo; j < hi; ++j)
					{
						if (qKey[j] != key) continue;
						if (j < q) { rep = false; break; }
						const float s = qScore[j];
						if (!haveBest || s > best) { best = s; qw = j; haveBest = true; }
					}
					if (rep && mode == 1)
					{
						// bucket = (h >> 8) & 3 of Hash<WordLL> (BestPathContainer.hpp:80-85, 323)
						const uint64_t lmv = (uint64_t)(int64_t)(int32_t)(uint32_t)key;
						const uint64_t hh = (uint64_t)(((key >> 40) & 0xFF) | (((key >> 32) & 0xFF) << 8)) ^ ((lmv << 3) | (lmv >> 61));
						rep = (int)((hh >> 8) & 3) == b;
					}
				}
				const uint64_t bal = X.ballot(rep);
				// mode 1 runs exactly one candidate per batch, so the per-bucket rank is the container's per-bucket fill
				const uint32_t rank = emittedInBucket + X.prefix(bal);
				const bool keep = rep && (mode == 2 || rank < 128);   // a full bucket drops later keys (BestPathContainer.hpp:363-367)
				const uint64_t kbal = X.ballot(keep);
				if (keep)
				{
					const uint32_t pos = X.stTop + X.prefix(kbal);
					if (pos < X.stCap)
					{
						const CandInfo& c = X.ci[k];
						const uint64_t wkey = qKey[qw];
						const uint32_t local = qw - c.qOff;
						const uint32_t parent = pBeg + local / c.R, r = local % c.R;
						const bool single = c.rec.flags & MF_SINGLE;
						const uint8_t rootKey = (uint8_t)(wkey >> 40);
						DevState ns;
						ns.lmNode = (int32_t)(uint32_t)wkey; ns.accScore = qScore[qw]; ns.firstChunkScore = qFcs[qw];
						ns.accTypoCost = X.st[parent].accTypoCost + 0.f;
						ns.parent = parent; ns.morph = c.morph; ns.wid = c.rec.lastSeqId; ns.nodeId = (uint16_t)nodeIdx;
						ns.rootId = (isQuoteOrBullet(c) && rootKey == COMMON_ROOT) ? (uint8_t)r : rootKey;
						ns.spState = (uint8_t)(wkey >> 32);
						ns.socket = single ? c.rec.socket : 0;
						ns.ownKind = single ? ownKind : 0; ns.ownNode = (single && ownKind) ? (uint16_t)nodeIdx : 0;
						ns.leftFeat = (single && ownKind) ? (uint16_t)(ownFeat | (c.leftFeat & (LF_TAG_SSC | LF_PREV_ZSIOT))) : c.leftFeat;
						ns.prevFlags = c.prevFlags; ns.pad = 0;
						X.st[pos] = ns;
					}
					else X.overflow = true;
				}
				X.stTop += __popcll(kbal);
				emittedInBucket += __popcll(bal);
			}
		}
		X.overflow = X.any(X.overflow);
		if (X.stTop > X.stCap) X.stTop = X.stCap;
		__threadfence_block();
	}

	// z_coda / z_siot shortcut (PathEvaluator.hpp:389-432): copies of the qualifying incoming paths
	template<int G>
	__device__ void evalZShortcut(GroupCtx<G>& X, uint32_t zMorph, uint32_t nodeIdx, const NodeEnv& E)
	{
		const ModelView& M = *X.M;
		const MorphRec cm = M.morphs[zMorph];
		const uint32_t newMorph = cm.lmId;
		const uint32_t mp = M.morphPath[newMorph];
		const uint16_t lfMorph = (uint16_t)mp;
		for (uint32_t pb = 0; pb < E.nP; pb += G)
		{
			const uint32_t p = pb + X.gl;
			bool keep = false; DevState ns;
			if (p < E.nP)
			{
				ns = X.st[E.pBeg + p];
				const uint8_t lastTag = M.morphs[ns.wid].tag;
				keep = cm.tag == T_Z_CODA ? (isJClass(lastTag) || isEClass(lastTag)) : isNNClass(lastTag);
			}
			const uint64_t bal = X.ballot(keep);
			if (keep)
			{
				const uint32_t pos = X.stTop + X.prefix(bal);
				if (pos < X.stCap)
				{
					ns.accScore += cm.userScore * X.P->typoCostWeight;
					ns.accTypoCost -= cm.userScore;
					ns.parent = E.pBeg + p; ns.morph = newMorph; ns.wid = newMorph; ns.nodeId = (uint16_t)nodeIdx;
					ns.leftFeat = ns.ownKind ? (uint16_t)((ns.leftFeat & (0x1FFF | LF_STR_SSC)) | (lfMorph & (LF_TAG_SSC | LF_PREV_ZSIOT))) : lfMorph;
					ns.prevFlags = (uint8_t)(mp >> 16);
					X.st[pos] = ns;
				}
				else X.overflow = true;
			}
			X.stTop += __popcll(bal);
		}
		X.overflow = X.any(X.overflow);
		if (X.stTop > X.stCap) X.stTop = X.stCap;
		__threadfence_block();
	}

	// PathEvaluator::operator() (PathEvaluator.hpp:347-512) for one candidate list
	template<int G>
	__device__ void evaluateNode(GroupCtx<G>& X, uint32_t nodeIdx, const DevNode& node, const NodeEnv& E,
		const uint32_t* cands, uint32_t nCands, uint8_t ownKind, uint16_t ownFeat, float unkDiscount)
	{
		const ModelView& M = *X.M;
		const SearchParams& P = *X.P;
		const uint32_t nodeStart = X.nodeStOff[nodeIdx];
		float ws = 0;
		if (!node.uformLen && node.form != NOFORM && M.forms[node.form].len && node.spaceErrors) ws = -P.spacePenalty * (float)node.spaceErrors;
		const float typoDiscount = -0.f * P.typoCostWeight;
		const float nodeLevelDiscount = ws + typoDiscount + unkDiscount;
		const int mode = E.nP <= 128 ? 0 : E.nP <= 512 ? 1 : 2;
		enum { K_NONE = 0, K_SKIP = 1, K_Z = 2, K_REG = 3 };

		for (int ignoreCond = 0; ignoreCond < 2; ++ignoreCond)
		{
			uint32_t c = 0;
			while (c < nCands)
			{
				// ---- lane j classifies candidate c+j; then the group agrees on the next batch ----------------
				const uint32_t idx = c + X.gl;
				uint32_t kind = K_NONE, Q = 0, R = 1, mid = 0;
				MorphRec cm{}; uint8_t sbType = 0;
				if (idx < nCands)
				{
					mid = cands[idx];
					cm = M.morphs[mid];
					if (P.splitComplex && (cm.flags & MF_HAS_COMPLEX)) kind = K_SKIP;
					else if (cm.tag == T_Z_CODA || cm.tag == T_Z_SIOT) kind = (cm.tag == T_Z_SIOT && !(P.splitSaisiot || P.mergeSaisiot)) ? K_SKIP : K_Z;
					else if (!(cm.flags & MF_SINGLE) && (cm.flags & MF_HA_CONTRACTION) && node.prev && E.spaceBefore) kind = K_SKIP;
					else
					{
						kind = K_REG;
						sbType = cm.tag == T_SB ? M.sbInfo[mid] : 0;
						const bool quote = cm.special == 0 || cm.special == 1 || cm.special == 3 || cm.special == 4;
						R = ((sbType || quote) && X.nUniq > 1) ? X.nUniq : 1;
						Q = E.nP * R;
					}
				}
				uint32_t nTake = 0, nC = 0, Qtot = 0, myK = 0xFFFFFFFFu, myOff = 0, zMorph = 0;
				bool zShortcut = false;
				for (int j = 0; j < G; ++j)
				{
					const uint32_t kj = X.bcast(kind, j);
					const uint32_t Qj = X.bcast(Q, j);
					const uint32_t midj = X.bcast(mid, j);
					if (kj == K_NONE) break;
					if (kj == K_SKIP) { ++nTake; continue; }
					if (kj == K_Z)
					{
						if (nC) break;                       // flush the regular batch first: container results keep candidate order
						zShortcut = true; zMorph = midj; ++nTake;
						break;
					}
					if (nC && (mode != 0 || Qtot + Qj > QCAP)) break;   // only small-container nodes share a batch
					if (Qj > BIGQ) { X.pairOverflow = true; ++nTake; continue; }
					if ((uint32_t)j == X.gl) { myK = nC; myOff = Qtot; }
					++nC; Qtot += Qj; ++nTake;
					if (mode != 0 || Qtot > QCAP) break;
				}
				if (myK != 0xFFFFFFFFu)
				{
					CandInfo& o = X.ci[myK];
					const uint32_t mp = M.morphPath[mid];
					o.rec = cm; o.morph = mid; o.qOff = myOff; o.R = R;
					o.additional = cm.userScore + nodeLevelDiscount + X.lb[(E.leftBoundary ? T_MAX : 0) + clearIrregular(cm.tag)] * 5.f;
					o.leftFeat = (uint16_t)mp; o.prevFlags = (uint8_t)(mp >> 16);
					o.sbType = sbType;
					o.ruleBits = ((isEClass(cm.tag) && E.formStartsA) ? RB_POSITIVE_E : 0) | ((cm.tag == T_SN && E.uformEndsPoint) ? RB_SN_POINT : 0);
				}
				__threadfence_block();
				c += nTake;
				if (nC) evalBatch<G>(X, nC, Qtot, nodeIdx, E, ignoreCond ? -10.f : 0.f, ownKind, ownFeat, mode);
				if (zShortcut) evalZShortcut<G>(X, zMorph, nodeIdx, E);
			}
			if (X.stTop > nodeStart) break;
		}

		// ---- pruning (PathEvaluator.hpp:475-511): keep paths within cutOff of the best of their root ------
		const uint32_t cnt = X.stTop - nodeStart;
		if (!cnt) return;
		const uint32_t nRootSlots = 1 + X.nUniq;
		for (uint32_t rs = 0; rs < nRootSlots; ++rs)
		{
			float mx = -INFINITY; bool anyOfRoot = false;
			for (uint32_t b = 0; b < cnt; b += G)
			{
				const uint32_t i = b + X.gl;
				if (i < cnt)
				{
					const DevState* s = &X.st[nodeStart + i];
					const uint32_t slot = s->rootId == COMMON_ROOT ? 0 : s->rootId + 1u;
					if (slot == rs) { anyOfRoot = true; if (!M.morphs[s->morph].socket) mx = fmaxf(mx, s->accScore); }
				}
			}
			for (int d = G / 2; d; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, G));
			if (!X.any(anyOfRoot)) continue;
			for (uint32_t b = 0; b < cnt; b += G)
			{
				const uint32_t i = b + X.gl;
				if (i < cnt)
				{
					DevState* s = &X.st[nodeStart + i];
					const uint32_t slot = s->rootId == COMMON_ROOT ? 0 : s->rootId + 1u;
					if (slot == rs) s->pad = (s->accScore + P.cutOff < mx) ? 0 : 1;
				}
			}
		}
		__threadfence_block();
		uint32_t out = 0;
		for (uint32_t b = 0; b < cnt; b += G)
		{
			const uint32_t i = b + X.gl;
			DevState s; bool keep = false;
			if (i < cnt) { s = X.st[nodeStart + i]; keep = s.pad != 0; }
			const uint64_t bal = X.ballot(keep);
			__threadfence_block();
			if (keep) { s.pad = 0; X.st[nodeStart + out + X.prefix(bal)] = s; }
			out += __popcll(bal);
			__threadfence_block();
		}
		X.stTop = nodeStart + out;
	}

	// libstdc++'s std::sort restated for the end-node candidate list (the reference sorts it with an unstable
	// std::sort, PathEvaluator.hpp:1359-1368; equal keys must land where introsort puts them).  Runs on one lane.
	__device__ __forceinline__ bool endLess(const EndCand& a, const EndCand& b)
	{
		if (a.rootId < b.rootId) return true;
		if (a.rootId > b.rootId) return false;
		if (a.sp < b.sp) return true;
		if (a.sp > b.sp) return false;
		return a.score > b.score;
	}
	__device__ void insertionSortEnd(EndCand* v, int lo, int hi, bool guarded)
	{
		for (int i = lo; i < hi; ++i)
		{
			const EndCand val = v[i];
			if (guarded && endLess(val, v[0])) { for (int j = i; j > 0; --j) v[j] = v[j - 1]; v[0] = val; }
			else { int j = i; while (endLess(val, v[j - 1])) { v[j] = v[j - 1]; --j; } v[j] = val; }
		}
	}
	__device__ __noinline__ void sortEndCands(EndCand* v, int n)
	{
		if (n <= 16) { insertionSortEnd(v, 1, n, true); return; }
		// introsort: median-of-three quick partitions down to 16-element runs (heap sort when the depth budget runs out),
		// then the final insertion sort -- std::__introsort_loop / std::__final_insertion_sort
		int stLo[40], stHi[40], stDepth[40]; int sp = 0;
		int depth = 0; for (int t = n; t > 1; t >>= 1) ++depth; depth *= 2;
		stLo[sp] = 0; stHi[sp] = n; stDepth[sp] = depth; ++sp;
		while (sp)
		{
			--sp;
			int lo = stLo[sp], hi = stHi[sp], dl = stDepth[sp];
			while (hi - lo > 16)
			{
				if (dl == 0)
				{
					// std::__partial_sort(first, last, last): make_heap + sort_heap with std::__adjust_heap
					const int len = hi - lo; EndCand* a = v + lo;
					auto adjust = [&](int hole, int length, EndCand val)
					{
						const int top = hole;
						int child = hole;
						while (child < (length - 1) / 2)
						{
							child = 2 * (child + 1);
							if (endLess(a[child], a[child - 1])) --child;
							a[hole] = a[child]; hole = child;
						}
						if ((length & 1) == 0 && child == (length - 2) / 2) { child = 2 * (child + 1); a[hole] = a[child - 1]; hole = child - 1; }
						int parent = (hole - 1) / 2;
						while (hole > top && endLess(a[parent], val)) { a[hole] = a[parent]; hole = parent; parent = (hole - 1) / 2; }
						a[hole] = val;
					};
					for (int parent = (len - 2) / 2; parent >= 0; --parent) adjust(parent, len, a[parent]);
					for (int last = len - 1; last > 0; --last) { const EndCand val = a[last]; a[last] = a[0]; adjust(0, last, val); }
					break;
				}
				--dl;
				const int first = lo, mid = lo + (hi - lo) / 2, ia = first + 1, ic = hi - 1;
				int med;
				if (endLess(v[ia], v[mid])) { if (endLess(v[mid], v[ic])) med = mid; else if (endLess(v[ia], v[ic])) med = ic; else med = ia; }
				else if (endLess(v[ia], v[ic])) med = ia; else if (endLess(v[mid], v[ic])) med = ic; else med = mid;
				{ const EndCand t = v[first]; v[first] = v[med]; v[med] = t; }
				int i = first + 1, j = hi;
				for (;;)
				{
					while (endLess(v[i], v[first])) ++i;
					--j;
					while (endLess(v[first], v[j])) --j;
					if (!(i < j)) break;
					const EndCand t = v[i]; v[i] = v[j]; v[j] = t;
					++i;
				}
				stLo[sp] = i; stHi[sp] = hi; stDepth[sp] = dl; ++sp;
				hi = i;
			}
		}
		insertionSortEnd(v, 1, 16, true);
		insertionSortEnd(v, 16, n, false);
	}

	__device__ __forceinline__ uint32_t unifyMorpheme(const ModelView& M, uint32_t m)   // PathEvaluator.hpp:1054-1058
	{
		if (m >= M.h.vocabSize || M.morphs[m].combinedId != (int32_t)m) return m;
		return M.morphs[m].lmId;
	}

	// generateTokenList (PathEvaluator.hpp:1038-1157) for one end candidate; single lane. Returns the token count or < 0.
	__device__ __noinline__ int backTrace(const ModelView& M, const SearchParams& P, const DevNode* nodes, const DevState* st, uint32_t endParent, DevToken* out, uint32_t cap)
	{
		uint32_t nSteps = 0;
		for (uint32_t s = endParent; st[s].parent != 0xFFFFFFFFu; s = st[s].parent) ++nSteps;
		if (!nSteps) return 0;
		int nTok = 0;
		uint32_t prevIdx;
		{
			uint32_t s = endParent;
			for (uint32_t k = 1; k < nSteps; ++k) s = st[s].parent;
			prevIdx = st[s].parent;
		}
		for (uint32_t step = nSteps; step-- > 0;)
		{
			uint32_t s = endParent;
			for (uint32_t k = 0; k < step; ++k) s = st[s].parent;
			const DevState cur = st[s];
			const DevState prev = st[prevIdx];
			const DevNode g = nodes[cur.nodeId];
			const MorphRec mm = M.morphs[cur.morph];
			const float scoreDiff = cur.accScore - prev.accScore;
			float typoDiff = cur.accTypoCost - prev.accTypoCost;
			const bool single = mm.flags & MF_SINGLE;
			const bool saisiotSplit = P.splitSaisiot && (mm.flags & MF_SAISIOT);
			const uint32_t numNew = (saisiotSplit || !single) ? mm.nChunks : 1;
			const float firstScore = cur.firstChunkScore + typoDiff * P.typoCostWeight;
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
				const uint8_t ok2 = cur.ownKind; uint32_t oa = 0, ol = 0;
				if (ok2)
				{
					const DevNode on = nodes[cur.ownNode];
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

	// End node (PathEvaluator.hpp:1320-1418): EOS transition, candidate sort, per-(root,state) selection, back-trace.
	template<int G>
	__device__ void finishChunk(GroupCtx<G>& X, const WorkView& W, uint32_t chunk, bool openEnding, DevChunkResult* res)
	{
		const ModelView& M = *X.M;
		const uint32_t Gn = X.Gn;
		const DevNode en = X.nodes[Gn - 1];
		const uint32_t firstPrev = Gn - 1 - en.prev;
		const uint32_t pBeg = X.nodeStOff[firstPrev];
		const uint32_t nP = (en.prev && en.nPrev) ? X.nodeStOff[firstPrev + en.nPrev - 1] + X.nodeStCnt[firstPrev + en.nPrev - 1] - pBeg : 0;
		EndCand* endBuf = X.scratch->end;
		uint32_t nEnd = 0; bool endOverflow = false;
		for (uint32_t pb = 0; pb < nP; pb += G)
		{
			const uint32_t p = pb + X.gl;
			bool ok = false; DevState ps{}; float c = 0, first = 0;
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
			uint32_t incl = mult;
			for (int d = 1; d < G; d <<= 1) { const uint32_t v = __shfl_up(incl, d, G); if ((int)X.gl >= d) incl += v; }
			const uint32_t base = nEnd + incl - mult;
			for (uint32_t r = 0; r < mult; ++r)
			{
				if (base + r < ENDCAP)
				{
					EndCand e; e.score = c; e.fcs = first; e.typo = ps.accTypoCost; e.parent = pBeg + p; e.pad = 0;
					if (ps.rootId == COMMON_ROOT) { e.rootId = (uint8_t)r; e.sp = X.uniq[r]; } else { e.rootId = ps.rootId; e.sp = ps.spState; }
					endBuf[base + r] = e;
				}
				else endOverflow = true;
			}
			nEnd += X.bcast(incl, G - 1);
		}
		endOverflow = X.any(endOverflow);
		__threadfence_block();
		if (X.gl == 0)
		{
			uint32_t status = CS_OK; uint32_t nPaths = 0;
			if (endOverflow) status = CS_ERR_PATH_OVERFLOW;
			else
			{
				sortEndCands(endBuf, (int)nEnd);
				uint32_t numUniq = 0;
				for (uint32_t a = 0; a < nEnd; ++a)
				{
					bool seen = false;
					for (uint32_t b = 0; b < a && !seen; ++b) seen = endBuf[b].rootId == endBuf[a].rootId && endBuf[b].sp == endBuf[a].sp;
					if (!seen) ++numUniq;
				}
				const uint32_t perGroup = numUniq ? (2 + numUniq - 1) / numUniq : 0;   // ceil(topN*2 / numUniq), topN = 1
				DevToken* tok = W.tokens + W.tokenBase[chunk];
				const uint32_t tokCap = (uint32_t)(W.tokenBase[chunk + 1] - W.tokenBase[chunk]);
				uint32_t tokTop = 0, startIdx = 0;
				for (uint32_t a = 0; a < nEnd && status == CS_OK; ++a)
				{
					if (a && (endBuf[a].rootId != endBuf[a - 1].rootId || endBuf[a].sp != endBuf[a - 1].sp)) startIdx = a;
					if (a - startIdx >= perGroup) continue;
					if (nPaths >= kMaxPathsPerChunk) { status = CS_ERR_PATH_OVERFLOW; break; }
					const int nt = backTrace(M, *X.P, X.nodes, X.st, endBuf[a].parent, tok + tokTop, tokCap - tokTop);
					if (nt < 0) { status = CS_ERR_TOKEN_OVERFLOW; break; }
					DevPathHeader& ph = res->paths[nPaths++];
					ph.score = endBuf[a].score; ph.tokOff = tokTop; ph.nTokens = (uint16_t)nt;
					ph.prevState = X.uniq[endBuf[a].rootId]; ph.curState = endBuf[a].sp;
					tokTop += (uint32_t)nt;
				}
			}
			res->status = status; res->nPaths = status == CS_OK ? nPaths : 0;
		}
	}

	template<int G>
	__device__ void searchChunk(GroupCtx<G>& X, const BatchView& B, const WorkView& W, uint32_t chunk)
	{
		const ModelView& M = *X.M;
		const SearchParams& P = *X.P;
		DevChunkResult* res = &W.results[chunk];
		if (res->status != CS_OK) { if (X.gl == 0) res->nPaths = 0; return; }
		const uint32_t cOff = B.charOff[chunk];
		const uint32_t nBase = W.nodeBase[chunk];
		X.nodes = W.nodes + nBase; X.Gn = W.nNodes[chunk];
		X.str = B.chars + cOff; X.cls = B.cls + cOff;
		X.st = W.states + W.stateBase[chunk]; X.stCap = (uint32_t)(W.stateBase[chunk + 1] - W.stateBase[chunk]); X.stTop = 0;
		X.nodeStOff = W.nodeStateOff + nBase; X.nodeStCnt = W.nodeStateCnt + nBase;
		X.uniq = B.spStates + B.spOff[chunk]; X.nUniq = B.spOff[chunk + 1] - B.spOff[chunk];
		X.overflow = false; X.pairOverflow = false;
		const uint32_t Gn = X.Gn;
		const bool openEnding = B.chunkFlags[chunk] & 1;
		uint8_t* reach = W.reach + nBase;

		// start node (PathEvaluator.hpp:1224-1226)
		if (X.gl == 0)
		{
			DevState bos;
			bos.lmNode = M.h.bosNode; bos.accScore = 0; bos.firstChunkScore = 0; bos.accTypoCost = 0; bos.parent = 0xFFFFFFFFu;
			bos.morph = 0; bos.wid = 0; bos.nodeId = 0; bos.rootId = COMMON_ROOT; bos.spState = 0; bos.socket = 0; bos.ownKind = 0;
			const uint32_t mp = M.morphPath[0];
			bos.leftFeat = (uint16_t)mp; bos.prevFlags = (uint8_t)(mp >> 16); bos.pad = 0; bos.ownNode = 0;
			X.st[0] = bos;
			X.nodeStOff[0] = 0; X.nodeStCnt[0] = 1;
		}
		X.stTop = 1;
		for (uint32_t k = X.gl; k < Gn; k += G) reach[k] = k == 0 ? 1 : 0;
		__threadfence_block();

		const uint32_t unkCands[2] = { T_NNG + 1u, T_NNP + 1u };
		for (uint32_t i = 1; i + 1 < Gn; ++i)
		{
			const DevNode node = X.nodes[i];
			NodeEnv E;
			const uint32_t firstPrev = i - node.prev, lastPrev = firstPrev + node.nPrev - 1;
			E.pBeg = X.nodeStOff[firstPrev];
			E.nP = X.nodeStOff[lastPrev] + X.nodeStCnt[lastPrev] - E.pBeg;
			E.spaceBefore = node.nflags & NF_SPACE_BEFORE; E.leftBoundary = node.nflags & NF_LEFT_BOUNDARY;
			E.uformEndsPoint = node.nflags & NF_UFORM_ENDS_POINT;
			E.formStartsA = false;
			if (X.gl == 0) X.nodeStOff[i] = X.stTop;
			__threadfence_block();

			uint8_t ownKind = 0; uint16_t ownFeat = 0;
			if (node.uformLen)
			{
				ownKind = 1;
				ownFeat = featMask(X.str + node.uformOff, node.uformLen) & 0x1FFF;
				const uint32_t lp = node.uformOff + node.uformLen - 1;
				const uint16_t c = X.str[lp];
				const uint8_t tag = (isLowSurrogate(c) || isHighSurrogate(c)) ? (uint8_t)T_SH : (uint8_t)(X.cls[lp] & 0x3F);
				if (tag == T_SSC) ownFeat |= LF_STR_SSC;
			}
			if (node.form != NOFORM)
			{
				const FormRec f = M.forms[node.form];
				E.formStartsA = f.flags & FF_STARTS_WITH_A;
				evaluateNode<G>(X, i, node, E, M.formCand + f.candOff, f.candCnt, ownKind, ownFeat, 0.f);
				// forms whose candidates are all partial morphemes also get an unknown proper-noun reading (PathEvaluator.hpp:1277-1287)
				bool notPartial = false;
				for (uint32_t cb = 0; cb < f.candCnt; cb += G)
				{
					const uint32_t ci = cb + X.gl;
					if (ci < f.candCnt)
					{
						const MorphRec m = M.morphs[M.formCand[f.candOff + ci]];
						if (!(m.socket || !(m.flags & MF_SINGLE))) notPartial = true;
						if (f.candCnt == 1 && m.tag == T_UNKNOWN && m.nChunks) notPartial = true;   // "isPretokenizedNode" (:1258-1263)
					}
				}
				if (!X.any(notPartial))
				{
					const uint16_t* fs = M.formChars + f.charOff;
					uint16_t of = featMask(fs, f.len) & 0x1FFF;
					if (f.flags & FF_ENDS_WITH_SSC) of |= LF_STR_SSC;
					evaluateNode<G>(X, i, node, E, &unkCands[1], 1, 2, of, -((float)f.len * P.oovRuleScale + P.oovRuleBias));
				}
				// reachable[i] and the forward re-scan of the persistent flags (PathEvaluator.hpp:1159-1176, 1286-1299)
				const uint32_t cntNow = X.stTop - X.nodeStOff[i];
				bool anyFree = false;
				for (uint32_t b = 0; b < cntNow; b += G) { const uint32_t k = b + X.gl; if (k < cntNow && !X.st[X.nodeStOff[i] + k].socket) anyFree = true; }
				anyFree = X.any(anyFree);
				if (X.gl == 0) reach[i] = anyFree ? 1 : 0;
				if (!anyFree)
				{
					uint32_t disc = 0;
					if (X.gl == 0)
					{
						for (uint32_t k = i + 1; k < Gn; ++k)
						{
							const DevNode nk = X.nodes[k];
							uint8_t r = 0;
							if (nk.prev) for (uint32_t pj = k - nk.prev, e = pj + nk.nPrev; pj < e; ++pj) if (reach[pj]) { r = 1; break; }
							reach[k] = r;
						}
						disc = reach[Gn - 1] ? 0 : 1;
					}
					disc = X.bcast(disc, 0);
					if (disc)
					{
						const uint32_t len = node.endPos - node.startPos;
						uint16_t of = featMask(X.str + node.startPos, len) & 0x1FFF;
						if (len)
						{
							const uint16_t c = X.str[node.endPos - 1];
							const uint8_t tag = (isLowSurrogate(c) || isHighSurrogate(c)) ? (uint8_t)T_SH : (uint8_t)(X.cls[node.endPos - 1] & 0x3F);
							if (tag == T_SSC) of |= LF_STR_SSC;
						}
						const float emo = (X.cls[node.startPos] & 0x80) ? -10.f : 0.f;
						evaluateNode<G>(X, i, node, E, unkCands, 2, 3, of, emo - ((float)len * P.oovRuleScale + P.oovRuleBias));
					}
				}
			}
			else
			{
				const float emo = (X.cls[node.uformOff] & 0x80) ? -10.f : 0.f;
				evaluateNode<G>(X, i, node, E, unkCands, 2, ownKind, ownFeat, emo - ((float)node.uformLen * P.oovRuleScale + P.oovRuleBias));
			}
			if (X.gl == 0) X.nodeStCnt[i] = X.stTop - X.nodeStOff[i];
			__threadfence_block();
			if (X.overflow || X.pairOverflow) break;
		}
		if (X.overflow || X.pairOverflow)
		{
			if (X.gl == 0) { res->status = X.overflow ? CS_ERR_STATE_OVERFLOW : CS_ERR_PAIR_OVERFLOW; res->nPaths = 0; }
			return;
		}
		finishChunk<G>(X, W, chunk, openEnding, res);
	}

	template<int G>
	__global__ void __launch_bounds__(64) k_best_path(ModelView M, BatchView B, WorkView W, SearchParams P, uint32_t* chunkCounter, const uint32_t* chunkOrder)
	{
		constexpr int NG = 64 / G;
		__shared__ uint64_t sKey[NG * QCAP];
		__shared__ float sScore[NG * QCAP];
		__shared__ float sFcs[NG * QCAP];
		__shared__ CandInfo sCand[NG * G];
		__shared__ float sLb[2 * T_MAX + 1];

		const uint32_t lane = threadIdx.x;
		// TagSequenceScorer tables (src/TagUtils.cpp:49-62): [0..T_MAX) without, [T_MAX..2*T_MAX) with a left boundary.
		// Tag PA (== T_MAX) indexes one past a row in the reference (include/kiwi/TagUtils.h:10-18): row 0 spills into
		// row 1, row 1 spills into the `weight` member (5.0) -- reproduced by the flat layout plus one extra slot.
		for (uint32_t t = lane; t < 2 * T_MAX + 1; t += 64)
		{
			float v = 0;
			if (t == 2 * T_MAX) v = 5.f;
			else if (t < T_MAX) { if (t == T_NNP || t == T_NP || t == T_IC) v = -1; else if (t == T_SB) v = -3; }
			else { const uint8_t r = (uint8_t)(t - T_MAX); v = (isEClass(r) || isJClass(r) || isSuffixTag(r) || r == T_VCP) ? -1.f : 0.f; }
			sLb[t] = v;
		}
		__syncthreads();

		// LDS addresses must not be folded into a constant aggregate (lld rejects addrspacecasts in static initialisers)
		uint32_t opaqueZero = 0;
		asm volatile("" : "+v"(opaqueZero));
		GroupCtx<G> X;
		const uint32_t gid = lane / G;
		X.M = &M; X.P = &P; X.lb = sLb + opaqueZero;
		X.gl = lane % G; X.gshift = gid * G;
		X.qKey = sKey + gid * QCAP + opaqueZero; X.qScore = sScore + gid * QCAP + opaqueZero; X.qFcs = sFcs + gid * QCAP + opaqueZero;
		X.ci = sCand + gid * G + opaqueZero;
		X.scratch = reinterpret_cast<GroupScratch*>(W.bigScratch) + ((size_t)blockIdx.x * NG + gid);

		for (;;)
		{
			uint32_t ci = 0;
			if (X.gl == 0) ci = atomicAdd(chunkCounter, 1u);
			ci = X.bcast(ci, 0);
			if (ci >= B.nChunks) break;
			searchChunk<G>(X, B, W, chunkOrder ? chunkOrder[ci] : ci);
		}
	}

	template __global__ void k_best_path<4>(ModelView, BatchView, WorkView, SearchParams, uint32_t*, const uint32_t*);
	template __global__ void k_best_path<8>(ModelView, BatchView, WorkView, SearchParams, uint32_t*, const uint32_t*);
	template __global__ void k_best_path<16>(ModelView, BatchView, WorkView, SearchParams, uint32_t*, const uint32_t*);
	template __global__ void k_best_path<64>(ModelView, BatchView, WorkView, SearchParams, uint32_t*, const uint32_t*);
}
