// HIP kernels for lattice construction on gfx950 (MI355X).
//
//  k_dict_scan   : one wavefront per chunk.  Lane s owns dictionary-scan start position s (looping when the
//                  chunk has more than 64 positions) and walks the CSR-flattened form trie from the root along
//                  str[s..]; the root transition is a direct table, deeper transitions a binary search over the
//                  node's sorted child keys.  Every terminal reached at depth d ending at position e sets bit
//                  d-1 of matchMask[e]; a wave-wide scan over popcounts turns the masks into packed per-end
//                  form lists (longest first), which is exactly the candidate order the reference's
//                  Aho-Corasick walk produces per character (goto/fail + submatch chain,
//                  /root/reference/src/KTrie.cpp:1283-1311).  Order independent => embarrassingly parallel.
//  k_build_lattice : one WAVE per chunk.  The reference's lattice bookkeeping is inherently sequential
//                  (OOV insertion depends on the end of the most recently appended node, appends depend on
//                  reachability: /root/reference/src/KTrie.cpp:15-43, 921-996, 1040-1137, 240-299), so lane 0
//                  replays it -- over an LDS-resident copy of the chunk's text, index maps, match masks and
//                  packed matches that all lanes stage first, growing the node list in LDS -- and the final
//                  per-node facts / re-ordering run one node per lane.  One launch per LDS size class.
//  k_build_lattice_big : one thread per chunk, arrays in HBM: chunks beyond the LDS budget, or that outgrew
//                  their LDS copy at run time.
//  k_expand_cands : static candidate records per (node, candidate) for the search kernel.
// Memory-bound integer work: no MFMA; see DESIGN.md for the roofline accounting.
#include <hip/hip_runtime.h>
#include "device_types.hpp"
#include "feature.hpp"
#include "lattice_connect.hpp"
#include "lattice_expand.hpp"
#include "chr_freq.hpp"

namespace kamd
{
	// child of `node` over `c` (0: none): the root's direct table, below it the edge hash -- one 16-byte load per probe, 1.1 probes on average (flat_model.hpp
	// TrieEdgeSlot); rounds 1 - 5 searched the node's sorted keys: record + log2(fan-out) halving steps + child, each a dependent load
	// `value`: the child's TrieNodeRec::value -- a slot carries it (TrieEdgeSlot::value), so that a step below the root is ONE load; below the root's table it is read
	__device__ __forceinline__ uint32_t trieChild(const ModelView& M, uint32_t node, uint16_t c, int32_t& value)
	{
		if (node == 0) { const uint32_t ch = M.trieRoot[c]; value = ch ? M.trie[ch].value : TRIE_NONE; return ch; }
		uint32_t h = trieEdgeHash(node, c) & M.trieEdgeMask;
		for (;;)
		{
			const uint4 s = reinterpret_cast<const uint4*>(M.trieEdges)[h];
			if (s.x == node && s.y == (uint32_t)c) { value = (int32_t)s.w; return s.z; }
			if (s.x == TRIE_EDGE_EMPTY) { value = TRIE_NONE; return 0; }
			h = (h + 1) & M.trieEdgeMask;
		}
	}

	__global__ void __launch_bounds__(256) k_dict_scan(ModelView M, BatchView B, WorkView W, uint32_t chunkBegin, uint32_t chunkCount)
	{
		const uint32_t lane = threadIdx.x & 63;
		const uint32_t local = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
		if (local >= chunkCount) return;
		const uint32_t chunk = chunkBegin + local;
		const uint32_t cOff = B.charOff[chunk], n = B.charOff[chunk + 1] - cOff;
		const uint16_t* str = B.chars + cOff;
		const uint8_t* cls = B.cls + cOff;
		uint16_t* nsToPos = W.nsToPos + cOff + chunk;
		uint16_t* posToNs = W.posToNs + cOff + chunk;
		uint8_t* cflag = W.cflag + cOff;
		uint64_t* mask = W.matchMask + cOff + chunk;
		uint32_t* moff = W.matchOff + cOff + chunk;

		// ---- 1. non-space index maps (Splitter::preparePattern tail, KTrie.cpp:837-854) ----------------
		uint32_t nsCount = 0;
		for (uint32_t base = 0; base < n; base += 64)
		{
			const uint32_t i = base + lane;
			bool isNs = false, skip = false;
			if (i < n)
			{
				const uint16_t c = str[i];
				// a unit right after an (unpaired-so-far) high surrogate is always kept with it
				uint32_t run = 0;
				for (int32_t j = (int32_t)i - 1; j >= 0 && isHighSurrogate(str[j]); --j) ++run;
				const bool second = run & 1;
				isNs = second || !isSpace(c);
				const bool pairStart = !second && isHighSurrogate(c) && i + 1 < n;
				// units the dictionary walk never feeds to the trie: surrogate pairs, and space-class units
				// (KTrie.cpp:1099-1116, 1218-1223); the ZWJ-after-symbol exception is resolved in k_build_lattice
				skip = second || pairStart || ((cls[i] & 0x3F) == T_UNKNOWN);
			}
			const uint64_t bal = __ballot(isNs);
			const uint32_t rank = nsCount + __popcll(bal & ((1ull << lane) - 1));
			if (i < n)
			{
				posToNs[i] = (uint16_t)rank;
				if (isNs) nsToPos[rank] = (uint16_t)i;
				cflag[i] = (isNs ? 1 : 0) | (skip ? 2 : 0);
			}
			nsCount += __popcll(bal);
		}
		if (lane == 0) { posToNs[n] = (uint16_t)nsCount; W.nNs[chunk] = nsCount; }
		const uint32_t nNs = nsCount;
		for (uint32_t e = lane; e <= nNs; e += 64) mask[e] = 0;
		waveSync();

		// ---- 2. trie walk, pass 1: mark terminals.  A start position's first three terminals are remembered (form, end position, depth) so that the
		// second pass need not walk the trie again; a start with more of them -- or a form id beyond 24 bits -- is walked twice as before ----
		constexpr uint32_t KEEP = 3;
		uint32_t kForm[KEEP], kEnd[KEEP];      // (end position | depth << 16)
		uint32_t nKept = 0; bool rewalk = false;
		const bool oneBlock = nNs <= 64;      // (the remembered terminals belong to one start position per lane)
		for (uint32_t base = 0; base < nNs; base += 64)
		{
			const uint32_t s = base + lane;
			if (s >= nNs) continue;
			if (cflag[nsToPos[s]] & 2) continue;
			uint32_t node = 0, depth = 0;
			for (uint32_t i = s; i < nNs; ++i)
			{
				const uint32_t p = nsToPos[i];
				if (cflag[p] & 2) continue;
				int32_t v;
				node = trieChild(M, node, str[p], v);
				if (!node) break;
				++depth;
				if (v >= 0)
				{
					atomicOr((unsigned long long*)&mask[i + 1], 1ull << (depth - 1));
					if (oneBlock)
					{
						if (nKept < KEEP && (uint32_t)v < (1u << 24) && depth < 64) { kForm[nKept] = (uint32_t)v; kEnd[nKept] = (i + 1) | (depth << 16); ++nKept; }
						else rewalk = true;
					}
				}
			}
		}
		waveSync();

		// ---- 3. exclusive scan of popcounts -> per-end offsets ----------------------------------------
		uint32_t total = 0;
		for (uint32_t base = 0; base <= nNs; base += 64)
		{
			const uint32_t e = base + lane;
			const uint32_t cnt = e <= nNs ? __popcll(mask[e]) : 0;
			uint32_t incl = cnt;
			for (uint32_t d = 1; d < 64; d <<= 1)
			{
				const uint32_t v = __shfl_up(incl, d);
				if (lane >= d) incl += v;
			}
			if (e <= nNs) moff[e] = total + incl - cnt;
			total += __shfl(incl, 63);
		}
		const uint32_t mBase = W.matchBase[chunk], mCap = W.matchBase[chunk + 1] - mBase;
		if (total > mCap)
		{
			if (lane == 0) W.results[chunk].status = CS_ERR_MATCH_OVERFLOW;
			return;
		}
		waveSync();

		// ---- 4. pass 2: fill packed form lists, longest form first within an end position -------------
		uint32_t* forms = W.matchForm + mBase;
		if (oneBlock && !rewalk)
		{
			for (uint32_t t = 0; t < nKept; ++t)
			{
				const uint32_t e = kEnd[t] & 0xFFFFu, depth = kEnd[t] >> 16;
				forms[moff[e] + (uint32_t)__popcll(mask[e] >> depth)] = kForm[t];      // forms longer than this one come first
			}
		}
		else
		for (uint32_t base = 0; base < nNs; base += 64)
		{
			const uint32_t s = base + lane;
			if (s >= nNs) continue;
			if (cflag[nsToPos[s]] & 2) continue;
			uint32_t node = 0, depth = 0;
			for (uint32_t i = s; i < nNs; ++i)
			{
				const uint32_t p = nsToPos[i];
				if (cflag[p] & 2) continue;
				int32_t v;
				node = trieChild(M, node, str[p], v);
				if (!node) break;
				++depth;
				if (v >= 0)
				{
					const uint64_t mk = mask[i + 1];
					const uint32_t rank = depth >= 64 ? 0 : __popcll(mk >> depth);   // forms longer than this one come first
					forms[moff[i + 1] + rank] = (uint32_t)v;
				}
			}
		}
	}

	// ------------------------------------------------------------------------------------------------
	// lattice node while the build runs: the 32-byte record of the search kernel (HBM variant), or the 16 bytes of it that the
	// build itself fills (LDS variant: the node list is the largest LDS array of a chunk; spaceErrors sits in a byte array beside it)
	struct BuildNode16 { uint32_t form; uint16_t startPos, endPos, prev, sibling, uformOff, uformLen; };
	template<class NodeT>
	struct LatticeCtxT
	{
		using Node = NodeT;
		uint8_t* spaceErr;      // BuildNode16 only
		const ModelView* M; const SearchParams* P;
		const uint16_t* str; const uint16_t* nsToPos; const uint16_t* posToNs;
		NodeT* out; uint32_t* endPosMap; uint64_t* fullMask; uint8_t* zAt; uint32_t nOut, cap; bool overflow;
		uint32_t lastEnd = 0;   // end position of the most recently appended node (what insertUnkForm asks for), kept out of the node list
		__device__ __forceinline__ void setSpaceErrors(uint32_t id, uint8_t v) { if constexpr (sizeof(NodeT) == sizeof(DevNode)) out[id].spaceErrors = v; else spaceErr[id] = v; }
		__device__ __forceinline__ DevNode full(uint32_t id) const
		{
			if constexpr (sizeof(NodeT) == sizeof(DevNode)) return out[id];
			else
			{
				const NodeT g = out[id];
				DevNode nn;
				nn.form = g.form; nn.startPos = g.startPos; nn.endPos = g.endPos; nn.prev = g.prev; nn.sibling = g.sibling; nn.uformOff = g.uformOff; nn.uformLen = g.uformLen;
				nn.spaceErrors = spaceErr[id]; nn.nflags = 0; nn.nPrev = 0; nn.packOff = 0; nn.candCnt = 0; nn.fflags = 0; nn.flen = 0; nn.ownFeat = 0; nn.pad = 0;
				return nn;
			}
		}
	};
	using LatticeCtx = LatticeCtxT<DevNode>;

	// `qual`: the node counts for hasFormAlready (zero typo cost and unknown-or-has-a-full-morpheme); `lenKey`: its length there
	template<class LC>
	__device__ __forceinline__ bool latAppend(LC& L, uint32_t s, uint32_t e, uint32_t form, uint32_t uOff, uint32_t uLen, uint32_t nMap, bool qual = false, uint32_t lenKey = 0, uint8_t zbits = 0)
	{
		const uint32_t ms = L.endPosMap[s];
		if ((ms & 0xFFFF) == (ms >> 16)) return false;
		if (L.nOut >= L.cap) { L.overflow = true; return false; }
		const uint32_t id = L.nOut++;
		typename LC::Node nn;
		nn.form = form; nn.startPos = (uint16_t)s; nn.endPos = (uint16_t)e; nn.prev = (uint16_t)(id - (ms & 0xFFFF)); nn.sibling = 0;
		nn.uformOff = (uint16_t)uOff; nn.uformLen = (uint16_t)uLen;
		if constexpr (sizeof(typename LC::Node) == sizeof(DevNode)) { nn.spaceErrors = 0; nn.nflags = 0; nn.nPrev = 0; nn.packOff = 0; nn.candCnt = 0; nn.fflags = 0; nn.flen = 0; nn.ownFeat = 0; nn.pad = 0; }
		else L.spaceErr[id] = 0;
		L.out[id] = nn;
		L.lastEnd = e;
		if (e >= nMap) return true;
		if (qual && lenKey >= 1 && lenKey <= 64) L.fullMask[e] |= 1ull << (lenKey - 1);
		if (zbits) L.zAt[e] |= zbits;
		const uint32_t me = L.endPosMap[e];
		if ((me & 0xFFFF) == (me >> 16)) L.endPosMap[e] = id | ((id + 1) << 16);
		else
		{
			const uint32_t last = (me >> 16) - 1;
			L.out[last].sibling = (uint16_t)(id - last);
			L.endPosMap[e] = (me & 0xFFFF) | ((id + 1) << 16);
		}
		return true;
	}

	// latAppend for the run of candidates that all END at the same position e < nMap (flushCandidates): that position's index entry, length mask and
	// flag bits are carried in registers by the caller (loaded before the run, written back after it) instead of being read, modified and written
	// in LDS / HBM for every candidate -- the compiler cannot keep them itself (it cannot prove that the node writes do not alias them)
	template<class LC>
	__device__ __forceinline__ bool latAppendAt(LC& L, uint32_t s, uint32_t e, uint32_t form, bool qual, uint32_t lenKey, uint8_t zbits, uint32_t& epmE, uint64_t& fmE, uint8_t& zE)
	{
		const uint32_t ms = L.endPosMap[s];
		if ((ms & 0xFFFF) == (ms >> 16)) return false;
		if (L.nOut >= L.cap) { L.overflow = true; return false; }
		const uint32_t id = L.nOut++;
		typename LC::Node nn;
		nn.form = form; nn.startPos = (uint16_t)s; nn.endPos = (uint16_t)e; nn.prev = (uint16_t)(id - (ms & 0xFFFF)); nn.sibling = 0;
		nn.uformOff = 0; nn.uformLen = 0;
		if constexpr (sizeof(typename LC::Node) == sizeof(DevNode)) { nn.spaceErrors = 0; nn.nflags = 0; nn.nPrev = 0; nn.packOff = 0; nn.candCnt = 0; nn.fflags = 0; nn.flen = 0; nn.ownFeat = 0; nn.pad = 0; }
		else L.spaceErr[id] = 0;
		L.out[id] = nn;
		L.lastEnd = e;
		if (qual && lenKey >= 1 && lenKey <= 64) fmE |= 1ull << (lenKey - 1);
		zE |= zbits;
		if ((epmE & 0xFFFF) == (epmE >> 16)) epmE = id | ((id + 1) << 16);
		else
		{
			const uint32_t last = (epmE >> 16) - 1;
			L.out[last].sibling = (uint16_t)(id - last);
			epmE = (epmE & 0xFFFF) | ((id + 1) << 16);
		}
		return true;
	}

	template<class LC>
	__device__ __forceinline__ uint32_t latNodeLen(const LC& L, const typename LC::Node& g)
	{
		if (g.uformLen) return g.uformLen;
		const FormRec f = L.M->forms[g.form];
		return f.len - f.numSpaces;
	}

	template<class LC>
	__device__ bool latHasForm(const LC& L, uint32_t s, uint32_t e)   // Splitter::hasFormAlready (KTrie.cpp:897-905)
	{
		// nodes are indexed by (end, length) in a 64-bit mask per end position; only longer spans need the scan
		if (e - s <= 64) return (L.fullMask[e] >> (e - s - 1)) & 1;
		const uint32_t me = L.endPosMap[e];
		uint32_t a = me & 0xFFFF; const uint32_t b = me >> 16;
		if (a == b) return false;
		if (a < 1) a = 1;
		for (uint32_t i = a; i < b; ++i)
		{
			const typename LC::Node g = L.out[i];
			if (g.endPos == e && g.endPos - latNodeLen(L, g) == s && (g.form == NOFORM || (L.M->forms[g.form].flags & FF_HAS_ANY_FULL))) return true;
		}
		return false;
	}

	template<class LC>
	__device__ __forceinline__ void latTrim(const LC& L, uint32_t off, uint32_t len, uint32_t& o, uint32_t& l)
	{
		// (NOT the dictionary scan's non-space flag: that one also counts the unit after a lone high surrogate as part of a pair)
		while (len && isSpace(L.str[off + len - 1])) --len;
		o = off; l = len;
	}

	template<class LC>
	__device__ void latInsertUnk(LC& L, uint32_t s, uint32_t e, bool hasJ, uint32_t nMap)   // Splitter::insertUnkForm (KTrie.cpp:921-953)
	{
		if (s >= e || latHasForm(L, s, e)) return;
		uint32_t lastPos = L.lastEnd;
		if (lastPos < e)
		{
			if (lastPos && isHangulCoda(L.str[L.nsToPos[lastPos]])) lastPos--;
			if (lastPos != s && !latHasForm(L, lastPos, e))
			{
				uint32_t o, l; latTrim(L, L.nsToPos[lastPos], L.nsToPos[e - 1] + 1 - L.nsToPos[lastPos], o, l);
				latAppend(L, lastPos, e, NOFORM, o, l, nMap, true, l);
			}
		}
		const uint32_t limit = hasJ ? L.P->maxUnkJ : L.P->maxUnk;
		if (e - s <= limit)
		{
			uint32_t o, l; latTrim(L, L.nsToPos[s], L.nsToPos[e - 1] + 1 - L.nsToPos[s], o, l);
			latAppend(L, s, e, NOFORM, o, l, nMap, true, l);
		}
	}

	template<class LC>
	__device__ __forceinline__ void latUnkPair(LC& L, uint32_t boundary, uint32_t unkStart, uint32_t e, bool hasJ, uint32_t nMap)
	{
		if (boundary < unkStart) latInsertUnk(L, boundary, e, hasJ, nMap);
		latInsertUnk(L, unkStart, e, hasJ, nMap);
	}

	// working arrays of one chunk's lattice build: HBM (thread-per-chunk variant) or the wave's LDS (wave-per-chunk variant)
	struct LatticeMem
	{
		const uint16_t* str; const uint8_t* cls; const uint8_t* script; const uint8_t* cflag;
		const uint64_t* mask; const uint32_t* moff; const uint32_t* mforms; const uint2* mfrec;   // mfrec: {start | space errors << 16, form flags | valid << 8} of every packed match, or null
		uint16_t* queue; uint16_t* connOrd;
	};

	// Splitter::splitByTrie replayed over the packed match lists (KTrie.cpp:1040-1137, 921-996): strictly sequential, one lane.
	template<class LC>
	__device__ __forceinline__ void latticeSerialBuild(const ModelView& M, const BatchView& B, const SearchParams& P, LC& L, const LatticeMem& Q,
		uint32_t chunk, uint32_t n, uint32_t nNs, uint32_t nMap)
	{
		const uint16_t* str = Q.str; const uint8_t* cls = Q.cls; const uint8_t* script = Q.script; const uint8_t* cflag = Q.cflag;
		const uint64_t* mask = Q.mask; const uint32_t* moff = Q.moff; const uint32_t* mforms = Q.mforms; const uint2* mfrec = Q.mfrec;
		const DevPattern* pat = B.patterns + B.patOff[chunk];
		const DevPattern* patEnd = B.patterns + B.patOff[chunk + 1];
		// pretokenized spans of the chunk: the entries behind its patterns (device_types.hpp kSpanTag), in text order
		const DevPattern* spanEnd = patEnd;
		while (patEnd != pat && (patEnd[-1].tag & kSpanTag)) --patEnd;
		const DevPattern* span = patEnd;

		uint8_t lastType = T_UNKNOWN, lastScript = 0;
		uint32_t specialStart = 0, unkStart = 0, boundary = 0;
		uint32_t resetNs = 0;   // dictionary matches starting before this ns position are void (see k_dict_scan step 1)
		bool staleZ = false; uint32_t staleZform = 0;      // a z-coda / saisiot candidate raised at a span's first unit: the reference flushes it with the NEXT flush (its list is not cleared)
		const uint8_t scriptVS = 98;
		for (uint32_t j = 0; j < n; ++j)
		{
			const uint16_t ch = str[j];
			const bool pair = isHighSurrogate(ch) && j + 1 < n;
			const uint32_t c32 = pair ? mergeSurrogate(ch, str[j + 1]) : ch;
			const bool inPattern = pat != patEnd && j >= pat->end - pat->length;
			uint8_t type = cls[j] & 0x3F, sct = script[j];
			bool overridden = false;
			if (lastType == T_SW && (c32 == 0x200d || (0x1f3fb <= c32 && c32 <= 0x1f3ff) || sct == scriptVS)) { overridden = type == T_UNKNOWN; type = lastType; sct = lastScript; }
			const uint8_t curT = inPattern ? (uint8_t)T_UNKNOWN : type;
			const bool symL = lastType == T_SL || lastType == T_SH || lastType == T_SW;
			const bool symC = curT == T_SL || curT == T_SH || curT == T_SW;
			const bool discont = (symL && symC) ? (lastScript != sct) : (lastType != curT);
			if (discont || lastType == T_SSO || lastType == T_SSC)
			{
				if (lastType != T_MAX && lastType != T_UNKNOWN && lastType != T_SS)
				{
					const bool sj = T_SF <= lastType && lastType <= T_SW;
					latUnkPair(L, boundary, unkStart, specialStart, sj, nMap);
					uint32_t o, l; latTrim(L, L.nsToPos[specialStart], j - L.nsToPos[specialStart], o, l);
					latAppend(L, specialStart, L.posToNs[j], lastType - 1u, o, l, nMap);
				}
				unkStart = specialStart;
				specialStart = L.posToNs[j];
				if (T_SF <= lastType && lastType <= T_SW) boundary = specialStart;
			}
			else if (type == T_MAX) unkStart = specialStart;
			lastType = curT; lastScript = sct;

			bool zcand = false; uint32_t zform = 0;
			if (!pair)
			{
				if (type == T_UNKNOWN)
				{
					latUnkPair(L, boundary, unkStart, L.posToNs[j + 1], true, nMap);
					boundary = specialStart = unkStart = L.posToNs[j + 1];
					continue;
				}
				// a space-class unit promoted to a symbol was fed to the trie by the reference and reset the walk
				if (overridden && (cflag[j] & 1)) resetNs = L.posToNs[j] + 1;
				bool zc = false, zs = false;
				const uint32_t p = L.posToNs[j];
				if (p < nNs) { const uint8_t zb = L.zAt[p]; zc = zb & FF_ZCODA_APPENDABLE; zs = zb & FF_ZSIOT_APPENDABLE; }
				if ((P.match & M_Z_CODA) && zc && isHangulCoda(ch) && (j + 1 >= n || !isHangulSyllable(str[j + 1]))) { zcand = true; zform = kDefaultTagSize + (ch - 0x11A8) - 1; }
				else if ((P.match & (M_SPLIT_SAISIOT | M_MERGE_SAISIOT)) && zs && ch == 0x11BA && j + 1 < n && isHangulSyllable(str[j + 1])) { zcand = true; zform = kDefaultTagSize + (0x11BA - 0x11A8) - 1; }
			}
			if (pat != patEnd)
			{
				const uint32_t curEnd = j + (pair ? 2 : 1);
				while (pat != patEnd && pat->end == curEnd)
				{
					const uint32_t ms = pat->end - pat->length;
					const bool wj = T_W_URL <= pat->tag && pat->tag <= T_W_EMOJI;
					latUnkPair(L, boundary, unkStart, L.posToNs[ms], wj, nMap);
					latAppend(L, L.posToNs[ms], L.posToNs[pat->end], pat->tag - 1u, ms, pat->length, nMap);
					++pat;
				}
			}
			// a pretokenized span begins here (KTrie.cpp:1177-1210): the pending unknown-form spans are closed, ONE node with the span's form is appended -- a
			// fallback form takes the text as its own string --, the dictionary walk restarts behind the span (matches that start before its end are void)
			if (span != spanEnd && span->end - span->length == j)
			{
				const uint32_t sb = j, se_ = span->end, sform = span->tag & kSpanFormMask; const bool fb = (span->tag & kSpanFallback) != 0;
				latUnkPair(L, boundary, unkStart, L.posToNs[sb], false, nMap);
				const FormRec sf = M.forms[sform];
				latAppend(L, L.posToNs[sb], L.posToNs[se_], sform, fb ? sb : 0u, fb ? se_ - sb : 0u, nMap, (sf.flags & FF_HAS_ANY_FULL) != 0, fb ? se_ - sb : (uint32_t)(sf.len - sf.numSpaces), sf.flags & 3);
				j += (se_ - sb) - 1;
				++span;
				lastType = T_UNKNOWN;
				resetNs = L.posToNs[j + 1];
				specialStart = unkStart = boundary = L.posToNs[j + 1];
				if (zcand && !staleZ) { staleZ = true; staleZform = zform; }
				continue;
			}
			if (pair) { ++j; continue; }

			// flushCandidates (KTrie.cpp:955-996) over [z-coda shortcut] + the packed dictionary matches ending here
			const uint32_t endNs = L.posToNs[j + 1];
			const uint32_t m0 = moff[endNs], m1 = m0 + __popcll(mask[endNs]);
			const uint32_t nZ = (staleZ ? 1u : 0u) + (zcand ? 1u : 0u);
			const uint32_t kFirst = m0 - nZ;
			if (kFirst == m1) continue;
			// every candidate of this run ends at endNs: that position's index entry / length mask / flag bits stay in registers (latAppendAt)
			uint32_t epmE = L.endPosMap[endNs]; uint64_t fmE = L.fullMask[endNs]; uint8_t zE = L.zAt[endNs];
			const uint32_t epm0 = epmE; const uint64_t fm0 = fmE; const uint8_t z0 = zE;
			for (uint32_t k = kFirst; k != m1; ++k)
			{
				const bool isZ = k - kFirst < nZ;      // (unsigned: the z entries sit in front of the matches)
				if (!isZ && mfrec)
				{
					// wave-per-chunk variant: the match's facts were computed by the staging pass
					const uint2 r = mfrec[k]; const uint32_t fiK = mforms[k];      // (two independent reads: one round trip)
					if (!(r.y & 0x100)) continue;
					const uint32_t nb = r.x & 0xFFFF, ne = endNs, se = r.x >> 16; const uint8_t fl = (uint8_t)r.y;
					if (nb < resetNs) continue;
					if (!(fl & FF_FIRST_IS_CODA))
					{
						// insertUnkForm(s, nb) returns at once when s >= nb or a form already covers [s, nb): both tests of the two calls read the same
						// length mask -- one read decides the common case (nothing to insert); a call that does run may append, so the second is then made in full
						const uint64_t fmNb = L.fullMask[nb];
						const bool skipB = !(boundary < nb) || (nb - boundary <= 64 && ((fmNb >> (nb - boundary - 1)) & 1));
						const bool skipU = unkStart >= nb || (nb - unkStart <= 64 && ((fmNb >> (nb - unkStart - 1)) & 1));
						if (!skipB || !skipU)
						{
							const bool hj = (fl & FF_HAS_JCLASS) || (fl & FF_IS_STAG);
							if (!skipB) latInsertUnk(L, boundary, nb, hj, nMap);
							if (!skipB || !skipU) latInsertUnk(L, unkStart, nb, hj, nMap);
						}
					}
					if (se <= P.spaceTol)
					{
						if (latAppendAt(L, nb, ne, fiK, (fl & FF_HAS_ANY_FULL) != 0, ne - nb, fl & 3, epmE, fmE, zE)) L.setSpaceErrors(L.nOut - 1, (uint8_t)(se > 255 ? 255 : se));
					}
					continue;
				}
				const uint32_t fi = isZ ? ((staleZ && k == kFirst) ? staleZform : zform) : mforms[k];
				const FormRec f = M.forms[fi];
				const uint32_t flen = f.len - f.numSpaces;
				if (flen > endNs) continue;
				const uint32_t nb = endNs - flen, ne = endNs;
				if (!isZ && nb < resetNs) continue;
				if (!(f.flags & FF_FIRST_IS_CODA))
				{
					const bool hj = (f.flags & FF_HAS_JCLASS) || (f.flags & FF_IS_STAG);
					if (boundary < nb) latInsertUnk(L, boundary, nb, hj, nMap);
					latInsertUnk(L, unkStart, nb, hj, nMap);
				}
				// countSpaceErrors (KTrie.cpp:316-328)
				uint32_t se = 0, off = 0;
				if (!f.numSpaces)
				{
					// a form without spaces: every gap inside the span is an error; the form's characters are not needed
					for (uint32_t i = 1; i < ne - nb; ++i) se += (L.nsToPos[nb + i] - L.nsToPos[nb + i - 1] > 1) ? 1u : 0u;
				}
				else
				{
					const uint16_t* fs = M.formChars + f.charOff;
					for (uint32_t i = 1; i < ne - nb; ++i)
					{
						const bool hasSpace = L.nsToPos[nb + i] - L.nsToPos[nb + i - 1] > 1;
						const uint16_t fc = (i + off < f.len) ? fs[i + off] : 0;
						if (hasSpace && fc != u' ') ++se;
						if (fc == u' ') ++off;
					}
				}
				if (se <= P.spaceTol)
				{
					if (latAppendAt(L, nb, ne, fi, (f.flags & FF_HAS_ANY_FULL) != 0, flen, f.flags & 3, epmE, fmE, zE)) L.setSpaceErrors(L.nOut - 1, (uint8_t)(se > 255 ? 255 : se));
				}
			}
			if (epmE != epm0) L.endPosMap[endNs] = epmE;
			if (fmE != fm0) L.fullMask[endNs] = fmE;
			if (zE != z0) L.zAt[endNs] = zE;
			staleZ = false;
		}
		if (lastType != T_MAX && lastType != T_UNKNOWN && lastType != T_SS)
		{
			const bool sj = T_SF <= lastType && lastType <= T_SW;
			latUnkPair(L, boundary, unkStart, specialStart, sj, nMap);
			uint32_t o, l; latTrim(L, L.nsToPos[specialStart], n - L.nsToPos[specialStart], o, l);
			latAppend(L, specialStart, L.posToNs[n], lastType - 1u, o, l, nMap);
			unkStart = specialStart;
			if (sj) boundary = L.posToNs[n];
		}
		if (nNs && n == (uint32_t)L.nsToPos[nNs - 1] + 1) latUnkPair(L, boundary, unkStart, L.posToNs[n], true, nMap);
		latAppend(L, nNs, nNs + 1, NOFORM, 0, 0, nMap);
		L.out[L.nOut - 1].endPos = (uint16_t)nNs;
	}

	// removeUnconnected, part 1 (KTrie.cpp:240-299): backward BFS from the end node, then the new index of every connected node.
	// Returns the number of connected nodes; inv (= Q.queue) holds old -> new (0xFFFF: dropped); endPosMap[e] := connected nodes ending at e.
	template<class LC>
	__device__ __forceinline__ uint32_t latticeConnect(LC& L, const LatticeMem& Q, uint32_t cap, uint32_t nNs)
	{
		uint16_t* queue = Q.queue; uint16_t* connOrd = Q.connOrd;
		const uint32_t G = L.nOut;
		for (uint32_t i = 0; i < G; ++i) connOrd[i] = 0;
		uint32_t qh = 0, qt = 0;
		queue[qt++] = (uint16_t)(G - 1); connOrd[G - 1] = 1;
		while (qh < qt)
		{
			const uint32_t id = queue[qh++];
			const uint32_t sp = L.out[id].startPos;
			const uint32_t me = L.endPosMap[sp];
			for (uint32_t i = me & 0xFFFF; i < (me >> 16); ++i)
			{
				if (L.out[i].endPos != sp || connOrd[i]) continue;
				connOrd[i] = 1; queue[qt++] = (uint16_t)i;
			}
		}
		// new index of each connected node: nodes grouped by end position ascending, original order inside a group.
		// The sibling chain of endPosMap[e] enumerates exactly the nodes ending at e in index order; the end-of-input
		// node is not on any chain and sorts last among nodes ending at nNs.
		uint16_t* inv = queue;
		uint32_t nConn = 0;
		for (uint32_t e = 0; e <= nNs; ++e)
		{
			const uint32_t me = L.endPosMap[e];
			uint32_t chainConn = 0;
			if ((me & 0xFFFF) != (me >> 16))
			{
				for (uint32_t i = me & 0xFFFF;;)
				{
					if (i != G - 1)
					{
						if (connOrd[i]) { inv[i] = (uint16_t)nConn++; ++chainConn; }
						else inv[i] = (uint16_t)0xFFFF;
					}
					const uint32_t sib = L.out[i].sibling;
					if (!sib) break;
					i += sib;
				}
			}
			L.endPosMap[e] = chainConn;   // from here on: number of connected nodes ending at e
		}
		inv[G - 1] = (uint16_t)nConn++;
		return nConn;
	}

	// removeUnconnected, part 2: old node idx -> final record at its new index (predecessor-dependent facts the search kernel
	// needs once per node included).  Returns the node's candidate count, or 0xFFFFFFFF for a dropped node.  Nodes are independent.
	template<class LC>
	__device__ __forceinline__ uint32_t latticeEmitNode(const ModelView& M, const LC& L, const uint16_t* str, const uint8_t* cls, const uint16_t* inv,
		DevNode* fin, uint32_t idx, uint32_t n, uint32_t nConn, uint32_t textOff)
	{
		const uint32_t ni = inv[idx];
		if (ni == 0xFFFF) return 0xFFFFFFFFu;
		DevNode nn = L.full(idx);
		const uint32_t startNs = nn.startPos;
		uint8_t nf = 0;
		if (ni >= 1)
		{
			// predecessor-dependent facts the search kernel needs once per node
			const typename LC::Node pn = L.out[idx - nn.prev];
			const uint32_t startStr = (ni + 1 == nConn) ? n : (uint32_t)L.nsToPos[startNs];
			const bool pnBos = (idx - nn.prev) == 0;
			const uint32_t pnEndStr = pnBos ? 0 : (uint32_t)L.nsToPos[pn.endPos - 1] + 1;
			// the reference compares absolute text offsets; the start node's end is 0 (PathEvaluator.hpp:24-31, 436, 568)
			const bool spaceBefore = pnBos ? (textOff + startStr > 0) : (pnEndStr < startStr);
			bool lb = pnBos || spaceBefore;
			if (!lb && pn.uformLen)
			{
				const uint32_t lp = pn.uformOff + pn.uformLen - 1;
				const uint16_t c = str[lp];
				const uint8_t tag = (isLowSurrogate(c) || isHighSurrogate(c)) ? (uint8_t)T_SH : (uint8_t)(cls[lp] & 0x3F);
				if (tag == T_SSC || c == u'"' || c == u'\'') lb = false;
				else if (T_SF <= tag && tag <= T_SB) lb = true;
			}
			if (spaceBefore) nf |= NF_SPACE_BEFORE;
			if (lb) nf |= NF_LEFT_BOUNDARY;
			if (nn.uformLen && str[nn.uformOff + nn.uformLen - 1] == u'.') nf |= NF_UFORM_ENDS_POINT;
			nn.nPrev = (uint16_t)L.endPosMap[startNs];
		}
		if (nn.form != NOFORM)
		{
			const FormRec f = M.forms[nn.form];
			nn.candCnt = f.candCnt; nn.fflags = f.flags; nn.flen = f.len;
			if (f.flags2 & FF2_ALL_PARTIAL) nf |= NF_ALL_PARTIAL;
		}
		if (nn.uformLen)
		{
			uint16_t of = featMask(str + nn.uformOff, nn.uformLen) & 0x1FFF;
			const uint32_t lp = nn.uformOff + nn.uformLen - 1;
			const uint16_t c = str[lp];
			const uint8_t tag = (isLowSurrogate(c) || isHighSurrogate(c)) ? (uint8_t)T_SH : (uint8_t)(cls[lp] & 0x3F);
			if (tag == T_SSC) of |= LF_STR_SSC;
			nn.ownFeat = of;
		}
		nn.nflags = nf;
		if (nn.prev) nn.prev = (uint16_t)(ni - inv[idx - nn.prev]);
		if (nn.sibling)
		{
			const uint32_t ns = inv[idx + nn.sibling];
			nn.sibling = ns == 0xFFFF ? 0 : (uint16_t)(ns - ni);
		}
		if (ni >= 1 && ni + 1 < nConn)
		{
			nn.startPos = L.nsToPos[nn.startPos];
			nn.endPos = (uint16_t)(L.nsToPos[nn.endPos - 1] + 1);
		}
		else if (ni + 1 == nConn) nn.startPos = nn.endPos = (uint16_t)n;
		nn.packOff = 0;
		fin[ni] = nn;
		return nn.candCnt;
	}

	// dynamic LDS of the wave-per-chunk variant
	extern __shared__ __align__(16) uint8_t lSmem[];

	// One lane group of GW lanes per chunk (64: one chunk per wavefront; 16: four).  The build itself is sequential (the group's lane 0), but every
	// access of it is an LDS access instead of a dependent HBM round trip: the chunk's text, index maps, packed matches (+ their form records) are
	// staged into LDS by all lanes first, the node list grows in LDS, and the final reorder / per-node fact computation runs one node per lane.
	// Chunks whose working set exceeds ldsBytes are left to k_build_lattice_big.
	// GW = 16 (EXPERIMENT, KAMD_LATTICE_GROUP=16): the replay is branchy one-lane code -- 14 k scalar and 6 k vector instructions per 40-jamo chunk --
	// and four chunks replayed by lanes 0 / 16 / 32 / 48 of one wavefront would share that wherever their control flow agrees.  Measured on the
	// MI355X it is 1.5x SLOWER than one chunk per wavefront (engine.hip has the numbers): the replays rarely agree.  ldsBytes = ONE group's region.
	template<int GW>
	__global__ void __launch_bounds__(64) k_build_lattice(ModelView M, BatchView B, WorkView W, SearchParams P, const uint32_t* chunkList, uint32_t chunkCount, uint32_t ldsBytes)
	{
		constexpr uint32_t NG = 64 / GW;
		const uint32_t lane = threadIdx.x % GW, grp = threadIdx.x / GW;
		const uint32_t wi = blockIdx.x * NG + grp;
		const uint32_t dbgStop = ldsBytes >> 24; ldsBytes &= 0xFFFFFFu;      // EXPERIMENT (KAMD_LATTICE_STOP): leave after phase 1 / 2 / 3, results void
		if (wi >= chunkCount) return;
		const uint32_t chunk = chunkList[wi];      // one launch per LDS size class: a slice of the longest-first work order
		uint8_t* const lS = lSmem + grp * ldsBytes;      // this group's region
		if (W.results[chunk].status >= 16) return;
		const uint32_t cOff = B.charOff[chunk], n = B.charOff[chunk + 1] - cOff;
		const uint32_t nNs = W.nNs[chunk];
		const uint32_t nBase = W.nodeBase[chunk], cap = W.nodeBase[chunk + 1] - nBase;
		const uint32_t mBase = W.matchBase[chunk], mCap = W.matchBase[chunk + 1] - mBase;
		const LatticeLds lay = latticeLdsLayout(n, cap, mCap);
		if (lay.total > ldsBytes) return;                      // k_build_lattice_big takes it
		const uint32_t nMap = nNs + 1;
		if (nNs > 0xFFF0 || cap > 0xFFF0 || cap < 4) { if (lane == 0) W.results[chunk].status = CS_ERR_TOO_LONG; return; }

		uint16_t* str = reinterpret_cast<uint16_t*>(lS + lay.str);
		uint8_t* cls = lS + lay.cls; uint8_t* script = lS + lay.script; uint8_t* cflag = lS + lay.cflag;
		uint16_t* nsToPos = reinterpret_cast<uint16_t*>(lS + lay.nsToPos); uint16_t* posToNs = reinterpret_cast<uint16_t*>(lS + lay.posToNs);
		uint64_t* mask = reinterpret_cast<uint64_t*>(lS + lay.mask); uint32_t* moff = reinterpret_cast<uint32_t*>(lS + lay.moff);
		uint32_t* mforms = reinterpret_cast<uint32_t*>(lS + lay.mforms); uint2* mfrec = reinterpret_cast<uint2*>(lS + lay.mfrec);
		// ---- stage the chunk into LDS (all lanes, coalesced) ----
		{
			const uint16_t* gstr = B.chars + cOff; const uint8_t* gcls = B.cls + cOff; const uint8_t* gscript = B.script + cOff; const uint8_t* gcflag = W.cflag + cOff;
			const uint16_t* gn2p = W.nsToPos + cOff + chunk; const uint16_t* gp2n = W.posToNs + cOff + chunk;
			const uint64_t* gmask = W.matchMask + cOff + chunk; const uint32_t* gmoff = W.matchOff + cOff + chunk;
			for (uint32_t i = lane; i < n; i += GW) { str[i] = gstr[i]; cls[i] = gcls[i]; script[i] = gscript[i]; cflag[i] = gcflag[i]; }
			for (uint32_t i = lane; i <= n; i += GW) { posToNs[i] = gp2n[i]; if (i < nNs) nsToPos[i] = gn2p[i]; }
			for (uint32_t i = lane; i <= nNs; i += GW) { mask[i] = gmask[i]; moff[i] = gmoff[i]; }
			const uint32_t mTot = gmoff[nNs] + __popcll(gmask[nNs]);
			const uint32_t* gforms = W.matchForm + mBase;
			if (mTot > latticeLdsCap(n, mCap)) { if (lane == 0) W.nNodes[chunk] = kLatticeNeedsBig; return; }    // group-uniform
			waveSync();
			// one packed match per lane: everything the replay needs of it that does not depend on the lattice built so far -- start position,
			// space errors of the span (countSpaceErrors, KTrie.cpp:316-328), form flags -- so that the serial loop reads one 8-byte record
			for (uint32_t k = lane; k < mTot; k += GW)
			{
				const uint32_t fi = gforms[k]; const FormRec f = M.forms[fi];
				mforms[k] = fi;
				uint32_t lo = 0, hi = nNs;      // end position of match k: the last e with moff[e] <= k
				while (lo < hi) { const uint32_t mid = (lo + hi + 1) >> 1; if (moff[mid] <= k) lo = mid; else hi = mid - 1; }
				const uint32_t endNs = lo, flen = f.len - f.numSpaces;
				uint32_t nb = 0, se = 0, valid = 0;
				if (flen <= endNs)
				{
					valid = 1; nb = endNs - flen;
					uint32_t off = 0;
					if (!f.numSpaces) { for (uint32_t i = 1; i < flen; ++i) se += (nsToPos[nb + i] - nsToPos[nb + i - 1] > 1) ? 1u : 0u; }
					else
					{
						const uint16_t* fs = M.formChars + f.charOff;
						for (uint32_t i = 1; i < flen; ++i)
						{
							const bool hasSpace = nsToPos[nb + i] - nsToPos[nb + i - 1] > 1;
							const uint16_t fc = (i + off < f.len) ? fs[i + off] : 0;
							if (hasSpace && fc != u' ') ++se;
							if (fc == u' ') ++off;
						}
					}
				}
				mfrec[k] = make_uint2(nb | ((se > 0xFFFFu ? 0xFFFFu : se) << 16), (uint32_t)f.flags | (valid << 8));
			}
		}
		LatticeCtxT<BuildNode16> L;
		L.M = &M; L.P = &P; L.str = str; L.nsToPos = nsToPos; L.posToNs = posToNs;
		L.spaceErr = lS + lay.spaceErr;
		L.out = reinterpret_cast<BuildNode16*>(lS + lay.out); L.endPosMap = reinterpret_cast<uint32_t*>(lS + lay.endPosMap);
		L.fullMask = reinterpret_cast<uint64_t*>(lS + lay.fullMask); L.zAt = lS + lay.zAt; L.nOut = 0; L.cap = latticeLdsCap(n, cap); L.overflow = false;
		const uint32_t ldsCap = L.cap;
		for (uint32_t i = lane; i < nMap; i += GW) { L.endPosMap[i] = 0; L.fullMask[i] = 0; L.zAt[i] = 0; }    // first == second : empty
		waveSync();
		LatticeMem Q{ str, cls, script, cflag, mask, moff, mforms, mfrec, reinterpret_cast<uint16_t*>(lS + lay.queue), reinterpret_cast<uint16_t*>(lS + lay.queue) + ldsCap };
		uint32_t nConn = 0, G = 0, err = 0;
		if (dbgStop == 1) { if (lane == 0) W.results[chunk].status = CS_NO_LATTICE; return; }
		if (lane == 0)
		{
			L.endPosMap[0] = 0 | (1u << 16);
			BuildNode16 bos; bos.form = NOFORM; bos.startPos = bos.endPos = 0; bos.prev = bos.sibling = 0; bos.uformOff = bos.uformLen = 0;
			L.out[0] = bos; L.spaceErr[0] = 0; L.nOut = 1;
			latticeSerialBuild(M, B, P, L, Q, chunk, n, nNs, nMap);
			if (L.overflow || L.nOut + 1 >= ldsCap) err = ldsCap < cap ? 0xFFFFu : (uint32_t)CS_ERR_NODE_OVERFLOW;   // 0xFFFF: outgrew the LDS copy only
			else G = L.nOut;
		}
		waveSync();
		err = __shfl(err, 0, GW); G = __shfl(G, 0, GW);
		if (err) { if (lane == 0) { if (err == 0xFFFFu) W.nNodes[chunk] = kLatticeNeedsBig; else W.results[chunk].status = err; } return; }
		if (dbgStop == 2) { if (lane == 0) W.results[chunk].status = CS_NO_LATTICE; return; }
		nConn = latticeConnectWave<GW>(L.out, L.endPosMap, Q.queue, Q.connOrd, reinterpret_cast<uint32_t*>(L.fullMask), G, nNs + 1, lane);
		if (dbgStop == 3) { if (lane == 0) W.results[chunk].status = CS_NO_LATTICE; return; }

		// ---- final records, one node per lane; candidate-record offsets by a wave scan over the new order ----
		DevNode* fin = W.nodes + nBase;
		const uint16_t* inv = Q.queue;
		uint16_t* cc = Q.connOrd;                       // candidate count per NEW index (the connected flags are no longer needed)
		const uint32_t textOff = B.textOffset[chunk];
		waveSync();
		for (uint32_t base = 0; base < G; base += GW)
		{
			const uint32_t idx = base + lane;
			uint32_t cnt = 0xFFFFFFFFu;
			if (idx < G) cnt = latticeEmitNode(M, L, str, cls, inv, fin, idx, n, nConn, textOff);
			if (cnt != 0xFFFFFFFFu) cc[inv[idx]] = (uint16_t)cnt;
		}
		waveSync();
		uint32_t packTop = 0;
		for (uint32_t base = 0; base < nConn; base += GW)
		{
			const uint32_t i = base + lane;
			const uint32_t c = i < nConn ? (uint32_t)cc[i] : 0u;
			uint32_t incl = c;
			for (uint32_t d = 1; d < GW; d <<= 1) { const uint32_t v = __shfl_up(incl, d, GW); if (lane >= d) incl += v; }
			if (i < nConn) fin[i].packOff = packTop + incl - c;
			packTop += __shfl(incl, GW - 1, GW);
		}
		if (lane == 0)
		{
			const uint32_t packCap = W.packBase[chunk + 1] - W.packBase[chunk];
			if (packTop > packCap) W.results[chunk].status = CS_ERR_NODE_OVERFLOW;
			else { W.nNodes[chunk] = nConn; if (nConn <= 2) W.results[chunk].status = CS_NO_LATTICE; }
		}
	}

	template __global__ void k_build_lattice<64>(ModelView, BatchView, WorkView, SearchParams, const uint32_t*, uint32_t, uint32_t);
	template __global__ void k_build_lattice<16>(ModelView, BatchView, WorkView, SearchParams, const uint32_t*, uint32_t, uint32_t);

	// One THREAD per chunk, all working arrays in HBM: for chunks whose working set does not fit the LDS budget of k_build_lattice.
	// waveLayout != 0: the chunks within ldsBytes were built by k_lattice_wave (lattice_wave.hip), whose LDS need is latticeWaveLayout's at that match ratio
	__global__ void __launch_bounds__(64) k_build_lattice_big(ModelView M, BatchView B, WorkView W, SearchParams P, uint32_t chunkBegin, uint32_t chunkCount, uint32_t ldsBytes, uint32_t waveLayout)
	{
		const uint32_t local = blockIdx.x * blockDim.x + threadIdx.x;
		if (local >= chunkCount) return;
		const uint32_t chunk = chunkBegin + local;
		if (W.results[chunk].status >= 16) return;
		const uint32_t cOff = B.charOff[chunk], n = B.charOff[chunk + 1] - cOff;
		const uint32_t nNs = W.nNs[chunk];
		const uint32_t nBase = W.nodeBase[chunk], cap = W.nodeBase[chunk + 1] - nBase;
		const uint32_t mCapB = W.matchBase[chunk + 1] - W.matchBase[chunk];
		const uint32_t needB = waveLayout ? latticeWaveLayout(n, cap, mCapB, waveLayout).total : latticeLdsLayout(n, cap, mCapB).total;
		if (needB <= ldsBytes && W.nNodes[chunk] != kLatticeNeedsBig && W.nNodes[chunk] != kLatticeNeedsWide) return;   // done by the wave-per-chunk kernel
		const uint16_t* str = B.chars + cOff;
		const uint8_t* cls = B.cls + cOff;
		LatticeCtx L;
		L.spaceErr = nullptr;
		L.M = &M; L.P = &P; L.str = str; L.nsToPos = W.nsToPos + cOff + chunk; L.posToNs = W.posToNs + cOff + chunk;
		L.out = W.tmpNodes + nBase; L.endPosMap = W.endPosMap + cOff + chunk; L.fullMask = W.fullMask + cOff + chunk; L.zAt = W.zAt + cOff + chunk; L.nOut = 0; L.cap = cap; L.overflow = false;
		const uint32_t nMap = nNs + 1;
		if (nNs > 0xFFF0 || cap > 0xFFF0 || cap < 4) { W.results[chunk].status = CS_ERR_TOO_LONG; return; }
		for (uint32_t i = 0; i < nMap; ++i) { L.endPosMap[i] = 0; L.fullMask[i] = 0; L.zAt[i] = 0; }    // first == second : empty
		L.endPosMap[0] = 0 | (1u << 16);
		{
			DevNode bos; bos.form = NOFORM; bos.startPos = bos.endPos = 0; bos.prev = bos.sibling = 0; bos.uformOff = bos.uformLen = 0; bos.spaceErrors = 0; bos.nflags = 0; bos.nPrev = 0; bos.packOff = 0; bos.candCnt = 0; bos.fflags = 0; bos.flen = 0; bos.ownFeat = 0; bos.pad = 0;
			L.out[0] = bos; L.nOut = 1;
		}
		uint16_t* queue = W.tmpIdx + 2ull * nBase;
		LatticeMem Q{ str, cls, B.script + cOff, W.cflag + cOff, W.matchMask + cOff + chunk, W.matchOff + cOff + chunk, W.matchForm + W.matchBase[chunk], nullptr, queue, queue + cap };
		latticeSerialBuild(M, B, P, L, Q, chunk, n, nNs, nMap);
		if (L.overflow || L.nOut + 1 >= cap) { W.results[chunk].status = CS_ERR_NODE_OVERFLOW; return; }
		const uint32_t G = L.nOut;
		const uint32_t nConn = latticeConnect(L, Q, cap, nNs);
		DevNode* fin = W.nodes + nBase;
		const uint32_t textOff = B.textOffset[chunk];
		for (uint32_t idx = 0; idx < G; ++idx) latticeEmitNode(M, L, str, cls, queue, fin, idx, n, nConn, textOff);
		{
			uint32_t packTop = 0;
			const uint32_t packCap = W.packBase[chunk + 1] - W.packBase[chunk];
			for (uint32_t i = 0; i < nConn; ++i) { fin[i].packOff = packTop; packTop += fin[i].candCnt; }
			if (packTop > packCap) { W.results[chunk].status = CS_ERR_NODE_OVERFLOW; return; }
		}
		W.nNodes[chunk] = nConn;
		if (nConn <= 2) W.results[chunk].status = CS_NO_LATTICE;
	}

	// Static candidate records per lattice node (one block per chunk, one thread per node): resolves
	// form -> candidate list -> morpheme record -> first LM id once, off the search kernel's dependent-load chain.
	// transposedOrder (CoNgram models): a node's records in the order the reference's transposed evaluator takes the candidates -- z-coda and
	// z-siot shortcuts first, then the regular candidates, then the left halves of split stems, then the right halves (src/PathEvaluator.hpp:
	// 884-915 + MorphemeEvaluator<CoNgramState>, src/CoNgramModel.cpp:86-135), each class in dictionary order
	// distMask (global CoNgram model, else null): regular candidates whose first word is a valid distant token come after the other regular ones
	// (MorphemeEvaluator<CoNgramState<7>>, src/CoNgramModel.cpp:105-124)
	__global__ void __launch_bounds__(64) k_expand_cands(ModelView M, BatchView B, WorkView W, uint32_t chunkBegin, uint32_t chunkCount, uint32_t transposedOrder, const uint8_t* distMask)
	{
		if (blockIdx.x >= chunkCount) return;
		const uint32_t chunk = chunkBegin + blockIdx.x;
		if (W.results[chunk].status != CS_OK || W.expanded[chunk]) return;      // (expanded: k_lattice_wave wrote the records itself)
		const uint32_t nBase = W.nodeBase[chunk], G = W.nNodes[chunk];
		CandStatic* packs = W.packs + W.packBase[chunk];
		const uint32_t* blk = W.blockBits;
		auto blocked = [&](uint32_t m2) -> bool { return blk && ((blk[m2 >> 5] >> (m2 & 31)) & 1); };
		for (uint32_t i = 1 + threadIdx.x; i + 1 < G; i += blockDim.x)
		{
			const DevNode nd = W.nodes[nBase + i];
			if (nd.form == NOFORM) continue;
			const uint32_t candOff = M.forms[nd.form].candOff;
			uint32_t kept = 0;
			for (uint32_t k = 0; k < nd.candCnt; ++k)
			{
				const uint32_t mid = M.formCand[candOff + k];
				// a candidate on the blocklist is not a candidate (src/PathEvaluator.hpp:385, 892): it gets no record, the node's list shrinks
				if (blocked(mid)) continue;
				++kept;
				const CandStatic o = candStaticOf(M, mid);
				uint32_t at = 0;
				if (transposedOrder)
				{
					auto cls = [&](uint32_t m2) -> uint32_t
					{
						const MorphRec r = M.morphs[m2];
						uint32_t c2 = 2u * candClassOf(r.tag, r.socket, r.flags);
						if (distMask && c2 == 4u) { const uint32_t fw = (r.flags & MF_SINGLE) ? r.lmId : M.chunkLm[r.chunkOff]; c2 += (distMask[fw >> 3] >> (fw & 7)) & 1u; }
						return c2;
					};
					const uint32_t mine = cls(mid);
					for (uint32_t j = 0; j < nd.candCnt; ++j)
					{
						if (j == k) continue;
						const uint32_t mj = M.formCand[candOff + j];
						if (blocked(mj)) continue;
						const uint32_t other = cls(mj);
						if (other < mine || (other == mine && j < k)) ++at;
					}
				}
				else if (blk) { for (uint32_t j = 0; j < k; ++j) if (!blocked(M.formCand[candOff + j])) ++at; }
				else at = k;
				packs[nd.packOff + at] = o;
			}
			if (kept != nd.candCnt) W.nodes[nBase + i].candCnt = (uint16_t)kept;
		}
	}

	// ------------------------------------------------------------------------------------------------
	// Position program of the position-step search kernel (k_pos_path, viterbi_pos.inc; device_types.hpp PosRec / PosDesc).
	// The search sweeps the lattice one END POSITION at a time; what a step does is, apart from the incoming paths, a function of the lattice:
	// which candidates are evaluated at all (PathEvaluator.hpp:385-446: complex morphemes under splitComplex, z-siot without the saisiot options,
	// the "ha" contraction after a space are skipped), their node-level score terms (whitespace / typo discount, unknown-form score, left-boundary
	// tag score: PathEvaluator.hpp:366-383, 1224-1318), the rule scorer's node-side inputs (:88-109), the own-form facts of the states they create.
	// One wave per chunk, three passes of one node per lane, after k_expand_cands (and k_unk_chr): A counts records and positions, B completes the
	// position table, C writes the records.  Memory-bound, off the search kernel's dependent chain.
	__global__ void __launch_bounds__(64) k_expand_pos(ModelView M, BatchView B, WorkView W, SearchParams P, uint32_t chunkBegin, uint32_t chunkCount, const float* nodeTypoAll, uint32_t useChr)      // useChr: bit 0 = unknown forms are scored by the character model, bit 1 = a CoNgram model
	{
		if (blockIdx.x >= chunkCount) return;
		const uint32_t lane = threadIdx.x;
		const uint32_t chunk = chunkBegin + blockIdx.x;
		if (W.expanded[chunk]) return;      // (k_lattice_wave wrote the position program itself)
		const uint32_t nBase = W.nodeBase[chunk];
		PosDesc* desc = W.posDesc + nBase;
		if (lane == 0) { desc[0].firstNode = 0; desc[0].nNodes = 0; desc[0].flags = 0; desc[0].firstRec = 0; desc[0].nRec = 0; }      // no positions until the table is complete
		if (W.results[chunk].status != CS_OK) return;
		const uint32_t G = W.nNodes[chunk];
		const uint32_t nUniq = B.spOff[chunk + 1] - B.spOff[chunk];
		if (G <= 2 || G > 0xFFF0u || nUniq + 1 > 0x1Fu) return;
		const DevNode* nodes = W.nodes + nBase;
		const CandStatic* packs = W.packs + W.packBase[chunk];
		const CandStatic* unkPacks = reinterpret_cast<const CandStatic*>(M.unkPacks);
		PosRec* recs = W.posRecs + W.packBase[chunk];
		const uint32_t recCap = W.packBase[chunk + 1] - W.packBase[chunk];
		uint32_t* prev = W.posPrev + nBase; uint32_t* nodeRec = W.posNodeRec + nBase;
		const uint8_t* cls = B.cls + B.charOff[chunk];

		// ---- A: records per node, position of every node ----
		uint32_t recTop = 0, posTop = 0;
		for (uint32_t base = 1; base + 1 < G; base += 64)
		{
			const uint32_t i = base + lane;
			const bool act = i + 1 < G;
			uint32_t cnt = 0; bool slow = false, isFirst = false;
			if (act)
			{
				const DevNode nd = nodes[i];
				if (nd.form != NOFORM)
				{
					const bool spaceBefore = nd.nflags & NF_SPACE_BEFORE;
					for (uint32_t k = 0; k < nd.candCnt; ++k)
					{
						const uint4 m1 = reinterpret_cast<const uint4*>(packs + nd.packOff + k)[1];
						const uint32_t kind = posCandKind(P, m1.y & 0xFFFF, (uint8_t)m1.z, spaceBefore);
						if (kind) ++cnt;      // a regular candidate, or the z-coda / z-siot shortcut (one item per incoming path)
					}
					if (!cnt) slow = true;      // nothing to evaluate: the reference then retries without conditions and falls back (PathEvaluator.hpp:468-473, 1286-1299)
					if (nd.nflags & NF_ALL_PARTIAL) ++cnt;
				}
				else cnt = 2;
				// (a step = the nodes of equal end: the text end, and for a lattice over a typo graph the multiplied end the build left in DevNode::pad -- 0 everywhere else)
				isFirst = i == 1 || nodes[i - 1].endPos != nd.endPos || nodes[i - 1].pad != nd.pad;
				const uint32_t firstPrev = i - nd.prev;
				prev[i] = firstPrev;      // (completed in pass B': | (predecessor count - 1) << 16 | position of the predecessors << 24)
				// the nodes of one step must not feed each other: true for spans of the text (a predecessor ends where the node starts, before its end);
				// a lattice over a typo graph may hold nodes of equal text end that do -- such a position is left to the general kernel
				uint32_t j0 = i;
				while (j0 > 1 && nodes[j0 - 1].endPos == nd.endPos && nodes[j0 - 1].pad == nd.pad) --j0;
				if (firstPrev + nd.nPrev > j0) slow = true;
			}
			uint32_t incl = cnt;
			for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d); if (lane >= d) incl += v; }
			const uint64_t fb = __ballot(isFirst);
			const uint32_t p = posTop + (uint32_t)__popcll(fb & ((2ull << lane) - 1));      // the node's position (numbered from 1; the start node is position 0)
			// (record offset: 18 bits, position: 13 bits -- a chunk beyond either is left to the general kernel below)
			if (act) nodeRec[i] = ((recTop + incl - cnt) & 0x3FFFFu) | ((p & 0x1FFFu) << 18) | (slow ? 0x80000000u : 0u);
			if (isFirst) { desc[p].firstNode = (uint16_t)i; desc[p].firstRec = recTop + incl - cnt; }
			recTop += __shfl(incl, 63);
			posTop += (uint32_t)__popcll(fb);
		}
		if (recTop > recCap || recTop >= 0x3FFFFu || posTop >= 0x1FFFu) return;      // (wave-uniform) the records do not fit the chunk's region / the packed fields: the general kernel takes the chunk
		const uint32_t nPos = posTop;
		if (lane == 0) { nodeRec[G - 1] = recTop; desc[nPos + 1].firstNode = (uint16_t)(G - 1); desc[nPos + 1].firstRec = recTop; nodeRec[0] = 0; }
		waveSync();
		// position of a node's predecessors (they all end where it starts): what the reachability propagation of the disconnected-lattice test runs over
		auto startPosOf = [&](uint32_t node) -> uint32_t { const uint32_t fp = node - nodes[node].prev; return fp ? (nodeRec[fp] >> 18) & 0x1FFFu : 0u; };

		// ---- B: extent of every position ----
		for (uint32_t p = 1 + lane; p <= nPos; p += 64)
		{
			const uint32_t n0 = desc[p].firstNode, n1 = desc[p + 1].firstNode, r0 = desc[p].firstRec, r1 = desc[p + 1].firstRec;
			const uint32_t maxRec = nodeTypoAll ? 32u : 16u;      // (the typo compilations of the position steps take a position's records in two passes of 16)
			bool slow = n1 - n0 > 16 || r1 - r0 > maxRec || r1 == r0;
			for (uint32_t j = n0; j < n1 && !slow; ++j) slow = (nodeRec[j] >> 31) != 0 || nodes[j].nPrev > 256;
			// (developer statistics, KAMD_POS_STATS: why a position is left to the general search -- counters 20 .. 23 of the batch: more than 16 nodes, more than 16
			// records, no record, a node that feeds its step / has nothing to evaluate / has more than 256 predecessors)
			if (slow) atomicAdd(W.outCounters + (n1 - n0 > 16 ? 20 : r1 - r0 > maxRec ? 21 : r1 == r0 ? 22 : 23), 1u);
			// the distinct start positions of the position's nodes, four bytes (morphemes of a handful of lengths end at one place); more than four, or a
			// chunk of more than 255 positions: no propagation for this chunk (header flag), the general kernel does its tests
			// (a lattice over a typo graph -- nodeTypoAll is bound -- gets EIGHT: the alternatives of a typo end at one place with many lengths, 8 of c5's 8 192 chunks had a
			// position with more than four; the second word lies one whole batch of nodes behind the first, W.posMask[total nodes + ...], read by the typo compilations)
			const uint32_t maxSt = nodeTypoAll ? 8u : 4u;
			uint32_t st4[8] = { 0, 0, 0, 0, 0, 0, 0, 0 }, nSt = 0; bool over = nPos > 255;
			for (uint32_t j = n0; j < n1 && !over; ++j)
			{
				const uint32_t sp = startPosOf(j);
				bool seen = false;
				for (uint32_t t = 0; t < nSt; ++t) seen = seen || st4[t] == sp;
				if (seen) continue;
				if (nSt == maxSt) { over = true; break; }
				st4[nSt++] = sp;
			}
			for (uint32_t t = nSt; t < 8; ++t) st4[t] = st4[0];
			W.posMask[nBase + p] = over ? 0xFFFFFFFFu : (st4[0] | (st4[1] << 8) | (st4[2] << 16) | (st4[3] << 24));
			if (nodeTypoAll) W.posMask[W.nodeBase[B.nChunks] + nBase + p] = over ? 0xFFFFFFFFu : (st4[4] | (st4[5] << 8) | (st4[6] << 16) | (st4[7] << 24));
			if (over) desc[0].nNodes = 1;
			uint32_t formless = 0;      // bit j: node n0 + j has no dictionary form (the reference never sets its `reachable` flag itself, PathEvaluator.hpp:1300-1318)
			for (uint32_t j = n0; j < n1 && j < n0 + 16; ++j) if (nodes[j].form == NOFORM) formless |= 1u << (j - n0);
			desc[p].pad = (uint16_t)formless;
			uint32_t pass1 = 0;      // bit j: node n0 + j also gets the unknown proper-noun reading (a second evaluation with its own ignore-conditions retry)
			for (uint32_t j = n0; j < n1 && j < n0 + 16; ++j) if (nodes[j].form != NOFORM && (nodes[j].nflags & NF_ALL_PARTIAL)) pass1 |= 1u << (j - n0);
			desc[p].pad2 = pass1;
			desc[p].nNodes = (uint8_t)(n1 - n0 > 255 ? 255 : n1 - n0); desc[p].flags = slow ? (uint8_t)POSF_SLOW : (uint8_t)0; desc[p].nRec = (uint16_t)(r1 - r0 > 0xFFFF ? 0xFFFF : r1 - r0);
		}

		for (uint32_t i = 1 + lane; i + 1 < G; i += 64)
		{
			const uint32_t sp = startPosOf(i), np1 = (uint32_t)nodes[i].nPrev - 1u;
			prev[i] = (prev[i] & 0xFFFFu) | ((np1 > 255u ? 255u : np1) << 16) | ((sp > 255u ? 255u : sp) << 24);
		}
		if (lane == 0) desc[0].pad2 = startPosOf(G - 1);      // where the end node's predecessors end: reachable <=> the lattice is connected

		// ---- C: the records ----
		for (uint32_t i = 1 + lane; i + 1 < G; i += 64)
		{
			const DevNode nd = nodes[i];
			const uint32_t nr = nodeRec[i];
			const uint32_t nl = i - desc[(nr >> 18) & 0x1FFFu].firstNode;      // the node's index inside its position (pass A numbered the positions)
			if (nl >= 16) continue;      // (its position is marked slow)
			PosRec* out = recs + (nr & 0x3FFFFu);
			float ws = 0;
			if (!nd.uformLen && nd.form != NOFORM && nd.flen && nd.spaceErrors) ws = -P.spacePenalty * (float)nd.spaceErrors;
			const float tc = nodeTypoAll ? nodeTypoAll[nBase + i] : 0.f;
			const float baseDiscount = ws + (-tc * P.typoCostWeight);      // whitespaceDiscount + typoDiscount (PathEvaluator.hpp:366-371)
			const bool spaceBefore = nd.nflags & NF_SPACE_BEFORE;
			const uint8_t ownKind0 = nd.uformLen ? 1 : 0;
			auto emit = [&](const CandStatic* cs, float disc, uint8_t ownKind, uint16_t ownFeat, uint32_t extra)
			{
				const uint4* q = reinterpret_cast<const uint4*>(cs);
				const uint4 m0 = q[0], m1 = q[1], mx = q[2];
				const uint8_t tag = (uint8_t)m1.z, special = (uint8_t)(m1.w >> 24);
				const uint32_t sbType = mx.z;
				const bool quote = special == 0 || special == 1 || special == 3 || special == 4;
				const uint32_t R = ((sbType || quote) && nUniq > 1) ? nUniq : 1u;
				PosRec r;
				r.firstWid = mx.y; r.secondWid = mx.w; r.chunkOff = m0.z; r.lastSeqId = m0.y;
				r.morph = mx.x; r.flagsFeat = m1.y; r.tagw = m1.z; r.cntw = m1.w;
				r.additional = __uint_as_float(m0.w) + disc + leftBoundaryScore(((nd.nflags & NF_LEFT_BOUNDARY) ? T_MAX : 0) + clearIrregular(tag)) * 5.f;
				const uint32_t ruleBits = ((isEClass(tag) && (nd.fflags & FF_STARTS_WITH_A)) ? 1u : 0u) | ((tag == T_SN && (nd.nflags & NF_UFORM_ENDS_POINT)) ? 2u : 0u)
					| ((M.morphDialect && M.morphDialect[r.morph]) ? 4u : 0u);      // (RB_DIALECT of the search kernels)
				r.nodeOwn = i | ((uint32_t)ownFeat << 16);
				r.bits = (sbType & 0xFF) | (ruleBits << 8) | ((uint32_t)ownKind << 16) | ((uint32_t)nd.nflags << 24);
				r.rq = R | (nl << 8) | extra;
				*out++ = r;
			};
			// CoNgram: do the regular candidates of one evaluation share their first word?  (decides which of the reference's kernels rounds their scores)
			auto sharedFirstWord = [&](const CandStatic* cl, uint32_t n) -> uint32_t
			{
				if (!(useChr & 2u)) return 0u;      // (only the CoNgram kernels read the flag)
				uint32_t nReg = 0, ref = 0; bool one = true;
				for (uint32_t k = 0; k < n; ++k)
				{
					const uint4* q = reinterpret_cast<const uint4*>(cl + k);
					const uint4 m1 = q[1], mx = q[2];
					const uint32_t flags = m1.y & 0xFFFF; const uint8_t tag = (uint8_t)m1.z, sock = (uint8_t)(m1.z >> 24);
					if (posCandKind(P, flags, tag, spaceBefore) != 1 || sock || (flags & MF_FIRST_WID_IS_P)) continue;
					if (!nReg) ref = mx.y; else if (mx.y != ref) one = false;
					++nReg;
				}
				return (nReg && one) ? (uint32_t)PR_OUT_FIRST : 0u;
			};
			if (nd.form != NOFORM)
			{
				const CandStatic* cl = packs + nd.packOff;
				const uint32_t of0 = sharedFirstWord(cl, nd.candCnt);
				for (uint32_t k = 0; k < nd.candCnt; ++k)
				{
					const uint4 m1 = reinterpret_cast<const uint4*>(cl + k)[1];
					const uint32_t kind = posCandKind(P, m1.y & 0xFFFF, (uint8_t)m1.z, spaceBefore);
					if (kind == 1) emit(cl + k, baseDiscount + 0.f, ownKind0, nd.ownFeat, of0);
					else if (kind == 2)
					{
						// z-coda / z-siot shortcut (PathEvaluator.hpp:389-432): qualifying incoming paths are copied with the shortcut's morpheme put on; the
						// record carries that morpheme (cm.lmId), the shortcut's tag and score, and what a path ending in the new morpheme exposes
						const MorphRec cm = M.morphs[reinterpret_cast<const uint4*>(cl + k)[2].x];
						const MorphRec nm = M.morphs[cm.lmId];
						PosRec r;
						r.firstWid = cm.lmId; r.secondWid = cm.tag; r.chunkOff = (uint32_t)nm.feat | ((uint32_t)nm.prevFlags << 16) | (nm.socket ? 1u << 24 : 0u); r.lastSeqId = 0;
						r.morph = cm.lmId; r.flagsFeat = 0; r.tagw = 0; r.cntw = 0;
						r.additional = cm.userScore;
						r.nodeOwn = i; r.bits = (uint32_t)nd.nflags << 24; r.rq = 1u | (nl << 8) | (uint32_t)PR_Z;
						*out++ = r;
					}
				}
				if (nd.nflags & NF_ALL_PARTIAL)
				{
					// the form read as an unknown proper noun (PathEvaluator.hpp:1277-1287): own form = the dictionary form's string
					const FormRec f = M.forms[nd.form];
					uint16_t of = featMask(M.formChars + f.charOff, f.len) & 0x1FFF;
					if (f.flags & FF_ENDS_WITH_SSC) of |= LF_STR_SSC;
					float disc;
					if (useChr & 1u) disc = baseDiscount + ((W.unkChrForm ? W.unkChrForm[nBase + i] : M.formUnkChr[nd.form]) - P.oovChrBias);
					else disc = baseDiscount + -((float)f.len * P.oovRuleScale + P.oovRuleBias);
					emit(unkPacks + 1, disc, 2, of, (uint32_t)PR_PASS1 | sharedFirstWord(unkPacks + 1, 1));
				}
			}
			else
			{
				// unknown form: the two unknown-noun candidates (PathEvaluator.hpp:1204-1206, 1300-1318), scored by UnkFormScorer (src/UnkFormScorer.h:40-58)
				const float emo = (cls[nd.uformOff] & 0x80) ? -10.f : 0.f;
				float disc;
				if (useChr & 1u) disc = baseDiscount + (W.unkChr[nBase + i] - P.oovChrBias);
				else disc = baseDiscount + (emo - ((float)nd.uformLen * P.oovRuleScale + P.oovRuleBias));
				const uint32_t of0 = sharedFirstWord(unkPacks, 2);
				emit(unkPacks, disc, ownKind0, nd.ownFeat, of0);
				emit(unkPacks + 1, disc, ownKind0, nd.ownFeat, of0);
			}
		}
		waveSync();
		if (lane == 0) desc[0].firstRec = nPos;
	}

	// Match::oovChrModel (UnkFormScorer::chrBasedScore, src/UnkFormScorer.cpp:53-66): the character model's score of every
	// lattice node's unknown form, once per node and off the search kernel's dependent chain -- a formless node's own string, else the node's
	// text span (what the search uses when the node left the lattice disconnected).  One block per chunk, one node per thread: the local
	// quantised CoNgram step (flat_model.hpp chrProgress: byte-keyed context trie + int8 dot product) per UTF-16 unit, then </s>; fp32 sum in
	// that order.  hiTok / loTok: the tokens of a lone high / low surrogate unit (ChrTokenizer::encodeOne sees units, not code points).
	__global__ void __launch_bounds__(64) k_unk_chr(ModelView M, BatchView B, WorkView W, ChrView C, uint32_t chunkBegin, uint32_t chunkCount, uint32_t hiTok, uint32_t loTok)
	{
		if (blockIdx.x >= chunkCount) return;
		const uint32_t chunk = chunkBegin + blockIdx.x;
		if (W.results[chunk].status != CS_OK) return;
		const uint32_t nBase = W.nodeBase[chunk], G = W.nNodes[chunk];
		const uint32_t cOff = B.charOff[chunk];
		const uint16_t* str = B.chars + cOff; const uint8_t* cls = B.cls + cOff;
		for (uint32_t i = 1 + threadIdx.x; i + 1 < G; i += blockDim.x)
		{
			const DevNode nd = W.nodes[nBase + i];
			const uint32_t off = nd.form == NOFORM ? nd.uformOff : nd.startPos, len = nd.form == NOFORM ? nd.uformLen : (uint32_t)(nd.endPos - nd.startPos);
			int32_t node = C.bosNode; uint32_t ctx = C.bosCtx;
			float score = 0;
			for (uint32_t k = 0; k < len; ++k)
			{
				const uint16_t c = str[off + k];
				const uint32_t tok = isHighSurrogate(c) ? hiTok : isLowSurrogate(c) ? loTok : chrToken(c, cls[off + k] & 0x7F);
				score += chrProgress(C, node, ctx, tok);
			}
			score += chrProgress(C, node, ctx, 0);
			W.unkChr[nBase + i] = score;
		}
	}
	// Match::oovChrFreqModel / oovChrFreqBranchModel (row f4; UnkFormScorer::chrFreqBasedScore, src/UnkFormScorer.cpp:68-121; chr_freq.hpp): k_unk_chr with the
	// substring frequencies of the text under analysis mixed in.  One block per chunk, one node per thread: the thread counts how often every prefix (<= 32 units)
	// of its node's string occurs in the FILTERED text the chunk belongs to (one pass over that text; 16-bit counters in LDS, one column per thread), then walks
	// the string through the character model.  W.unkChr[node]: a formless node's own string, else the node's text span -- as in k_unk_chr, but FINAL (bias
	// subtracted; the early exit of :101 leaves without it).  W.unkChrForm[node]: the same for the own string of a node's dictionary form, which the
	// frequency-free mode reads from a per-form table (ModelView::formUnkChr) and which here depends on the text; when that string is the text span the span's
	// score is reused.  HBM-bound integer / fp32 work off the search kernel's dependent chain; no MFMA.
	__global__ void __launch_bounds__(64) k_unk_chr_freq(ModelView M, BatchView B, WorkView W, ChrView C, ChrFreqParams Q, float chrBias, uint32_t chunkBegin, uint32_t chunkCount, uint32_t hiTok, uint32_t loTok)
	{
		__shared__ uint16_t cnt[kSubstrMaxLen * 64];
		if (blockIdx.x >= chunkCount) return;
		const uint32_t chunk = chunkBegin + blockIdx.x;
		if (W.results[chunk].status != CS_OK) return;
		const uint32_t nBase = W.nodeBase[chunk], G = W.nNodes[chunk];
		const uint32_t cOff = B.charOff[chunk];
		const uint16_t* str = B.chars + cOff; const uint8_t* cls = B.cls + cOff;
		const uint16_t* text = B.filtChars + B.filtOff[chunk]; const uint32_t textLen = B.filtLen[chunk];
		uint16_t* myCnt = cnt + threadIdx.x;
		for (uint32_t i = 1 + threadIdx.x; i + 1 < G; i += blockDim.x)
		{
			const DevNode nd = W.nodes[nBase + i];
			const uint32_t off = nd.form == NOFORM ? nd.uformOff : nd.startPos, len = nd.form == NOFORM ? nd.uformLen : (uint32_t)(nd.endPos - nd.startPos);
			bool biased;
			substringCounts(text, textLen, str + off, len, myCnt, 64);
			float score = chrFreqScore(C, Q, len, [&](uint32_t k) -> uint32_t
			{
				const uint16_t c = str[off + k];
				return isHighSurrogate(c) ? hiTok : isLowSurrogate(c) ? loTok : chrToken(c, cls[off + k] & 0x7F);
			}, myCnt, 64, biased);
			if (biased) score -= chrBias;
			W.unkChr[nBase + i] = score;
			if (nd.form == NOFORM) continue;
			const FormRec f = M.forms[nd.form];
			const uint16_t* fs = M.formChars + f.charOff;
			bool same = f.len == len;
			for (uint32_t k = 0; same && k < len; ++k) same = fs[k] == str[off + k];
			if (!same)
			{
				const uint16_t* ft = M.formChrTok + f.charOff;
				substringCounts(text, textLen, fs, f.len, myCnt, 64);
				score = chrFreqScore(C, Q, f.len, [&](uint32_t k) -> uint32_t { return ft[k]; }, myCnt, 64, biased);
				if (biased) score -= chrBias;
			}
			W.unkChrForm[nBase + i] = score;
		}
	}
}
