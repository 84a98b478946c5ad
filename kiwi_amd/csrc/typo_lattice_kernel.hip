// HIP kernel: the morpheme lattice of a chunk built OVER ITS TYPO GRAPH -- Splitter::search /
// progressNode / flushCandidates / insertUnkForm / hasFormAlready / removeUnconnected / writeResult of the reference
// (/root/reference/src/KTrie.cpp:897-996, 998-1464, 240-299) in their general form: one search state per (typo-graph node, way of
// reaching it), positions multiplied by 2^posMultiplierBit for the halves of continual typos.
//
// Two kernels over one build context: k_build_lattice_typo_lds -- one WAVEFRONT per chunk, text / index maps / node list / end-position index in LDS, the
// replay itself on lane 0 (the default; DESIGN.md section 4 "Typo correction on the device") -- and k_build_lattice_typo -- one thread per chunk over
// HBM arrays, for chunks that outgrow the LDS copy.  Two outputs: the dump records of the parity hook (Engine::dumpTypoLattices / kamd_typo_lattices), or,
// in engine mode, the search kernel's DevNode records plus a typo cost per node.  The typo graphs come from k_typo_graph (typo_graph_kernel.hip).
// GPU-green since round 2 (tests/test_gpu_typo.py: lattices and analyses bit-exact against the oracle = the real reference).
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdlib>
#include "device_types.hpp"
#include "feature.hpp"
#include "typo.hpp"
#include "typo_lattice_kernel.hpp"
#include "lattice_connect.hpp"

namespace kamd
{
	namespace
	{
		constexpr uint32_t NPOS = 0xFFFFFFFFu;
		constexpr uint32_t MAXCAND = kTypoMaxCand;

		// Every method is force-inlined: a context that is passed to a real call lives in scratch memory (every member access a scratch load) and its
		// LDS pointers degrade to flat accesses; inlined, it stays in registers and the LDS arrays are addressed with ds_read / ds_write.
		// LDS = false: every working array in HBM (thread per chunk).  LDS = true: the wave-per-chunk kernel -- text, index maps, node list
		// (TypoLdsNode), end-position index (first | (last + 1) << 16 per multiplied position) in LDS; hasFormAlready and the z-coda / z-siot
		// look-ups answered from per-end-position summaries (fullMask: lengths of the qualifying nodes ending there, zAt: their form flags)
		// instead of scans over the node list, as in lattice_kernels.hip.
		template<bool LDS>
		struct CtxT
		{
			const ModelView& M; const TypoLatView& V; TypoLatChunk& C;
			const uint16_t* str; const uint8_t* cls; const uint8_t* script; uint32_t n, nNs, pmb;
			const uint16_t* nsToPos; const uint16_t* posToNs;
			uint2* endPosMap; TypoLatNode* out;                                                     // HBM variant
			uint32_t* epm; TypoLdsNode* lout; uint64_t* fullMask; uint8_t* zAt; uint32_t* candBuf;   // LDS variant
			uint32_t ldsCap, lastEnd; bool outgrown;
			uint32_t mapLen;      // (TypoLatChunk::mapLen, read once: the chunk record is in HBM and every append asked for it)
			uint32_t nOut;
			const DevPattern* pat; const DevPattern* patEnd;
			bool overflow;

			// fp: the form's record when the caller holds it (null: read here if needed)
			__device__ __forceinline__ bool append(uint32_t s, uint32_t e, uint32_t form, uint32_t uOff, uint32_t uLen, float typoCost = 0.f, const FormRec* fp = nullptr)
			{
				if constexpr (LDS)
				{
					const uint32_t ms = epm[s];
					if ((ms & 0xFFFF) == (ms >> 16)) return false;
					if (nOut >= ldsCap) { outgrown = true; return false; }
					const uint32_t id = nOut++;
					TypoLdsNode nn; nn.form = form; nn.startPos = (uint16_t)s; nn.endPos = (uint16_t)e; nn.prev = (uint16_t)(id - (ms & 0xFFFF)); nn.sibling = 0;
					nn.uformOff = (uint16_t)uOff; nn.uformLen = (uint16_t)uLen; nn.typoCost = typoCost; nn.spaceErrors = 0; nn.pad[0] = nn.pad[1] = nn.pad[2] = 0;
					lout[id] = nn;
					lastEnd = e;
					if (e >= mapLen) return true;
					if ((e & ((1u << pmb) - 1)) == 0)      // only whole positions are ever asked about (insertUnk, the z-coda test)
					{
						uint8_t fl = 0; uint32_t flen = 0;
						if (form != NOFORM)
						{
							if (fp) { fl = fp->flags; flen = fp->len - fp->numSpaces; }
							else { const FormRec f = M.forms[form]; fl = f.flags; flen = f.len - f.numSpaces; }
							zAt[e >> pmb] |= fl & 3;
						}
						const uint32_t lenKey = uLen ? uLen : flen;
						if (typoCost == 0 && (form == NOFORM || (fl & FF_HAS_ANY_FULL)) && lenKey >= 1 && lenKey <= 64) fullMask[e >> pmb] |= 1ull << (lenKey - 1);
					}
					const uint32_t me = epm[e];
					if ((me & 0xFFFF) == (me >> 16)) epm[e] = id | ((id + 1) << 16);
					else
					{
						const uint32_t last = (me >> 16) - 1;
						lout[last].sibling = (uint16_t)(id - last);
						epm[e] = (me & 0xFFFF) | ((id + 1) << 16);
					}
					return true;
				}
				else
				{
					if (endPosMap[s].x == endPosMap[s].y) return false;
					if (nOut >= C.nodeCap) { overflow = true; return false; }
					const uint32_t id = nOut++;
					TypoLatNode nn; nn.startPos = s; nn.endPos = e; nn.prev = id - endPosMap[s].x; nn.sibling = 0; nn.form = (int32_t)form; nn.uformLen = uLen; nn.uformOff = uOff; nn.spaceErrors = 0; nn.typoCost = typoCost;
					out[id] = nn;
					if (e >= mapLen) return true;
					uint2 m = endPosMap[e];
					if (m.x == m.y) { m.x = id; m.y = id + 1; }
					else { out[m.y - 1].sibling = id - (m.y - 1); m.y = id + 1; }
					endPosMap[e] = m;
					return true;
				}
			}
			__device__ __forceinline__ void setLastSpaceErrors(uint32_t se) { if constexpr (LDS) lout[nOut - 1].spaceErrors = (uint8_t)se; else out[nOut - 1].spaceErrors = se; }
			__device__ __forceinline__ uint32_t lastNodeEnd() const { if constexpr (LDS) return lastEnd; else return out[nOut - 1].endPos; }
			__device__ __forceinline__ uint32_t nodeLen(const TypoLatNode& g) const
			{
				if (g.uformLen) return g.uformLen;
				const FormRec f = M.forms[g.form];
				return f.len - f.numSpaces;
			}
			__device__ __forceinline__ bool hasForm(uint32_t ms, uint32_t me) const      // both whole positions, multiplied
			{
				if constexpr (LDS)
				{
					const uint32_t len = (me - ms) >> pmb;
					if (len <= 64) return (fullMask[me >> pmb] >> (len - 1)) & 1;
					const uint32_t m = epm[me];
					uint32_t a = m & 0xFFFF; const uint32_t b = m >> 16;
					if (a < 1) a = 1;
					for (uint32_t i = a; i < b; ++i)
					{
						const TypoLdsNode g = lout[i];
						if (g.endPos != me || g.typoCost != 0) continue;
						uint32_t l = g.uformLen; bool full = g.form == NOFORM;
						if (!l || !full) { const FormRec f = M.forms[g.form]; if (!l) l = f.len - f.numSpaces; full = full || (f.flags & FF_HAS_ANY_FULL); }
						if (full && g.endPos - (l << pmb) == ms) return true;
					}
					return false;
				}
				else
				{
					const uint2 m = endPosMap[me];
					if (m.x == NPOS) return false;
					for (uint32_t i = m.x < 1 ? 1 : m.x; i < m.y; ++i)
					{
						const TypoLatNode g = out[i];
						if (g.endPos == me && g.endPos - (nodeLen(g) << pmb) == ms && g.typoCost == 0 && (g.form < 0 || (M.forms[g.form].flags & FF_HAS_ANY_FULL))) return true;
					}
					return false;
				}
			}
			__device__ __forceinline__ void trimmed(uint32_t off, uint32_t len, uint32_t& o, uint32_t& l) const
			{
				while (len && isSpace(str[off + len - 1])) --len;
				o = off; l = len;
			}
			__device__ __forceinline__ void insertUnk(uint32_t s, uint32_t e, bool hasJ)
			{
				if (s >= e || hasForm(s << pmb, e << pmb)) return;
				uint32_t lastPos = lastNodeEnd();      // (a multiplied position against plain ones: as in the reference)
				if (lastPos < e)
				{
					if (lastPos && isHangulCoda(str[nsToPos[lastPos]])) lastPos--;
					if (lastPos != s && !hasForm(lastPos << pmb, e << pmb))
					{
						uint32_t o, l; trimmed(nsToPos[lastPos], nsToPos[e - 1] + 1 - nsToPos[lastPos], o, l);
						append(lastPos << pmb, e << pmb, NOFORM, o, l);
					}
				}
				const uint32_t limit = hasJ ? V.maxUnkJ : V.maxUnk;
				if (e - s <= limit)
				{
					uint32_t o, l; trimmed(nsToPos[s], nsToPos[e - 1] + 1 - nsToPos[s], o, l);
					append(s << pmb, e << pmb, NOFORM, o, l);
				}
			}
			__device__ __forceinline__ void unkPair(uint32_t boundary, uint32_t unkStart, uint32_t e, bool hasJ)
			{
				if (boundary < unkStart) insertUnk(boundary, e, hasJ);
				insertUnk(unkStart, e, hasJ);
			}
			__device__ __forceinline__ uint32_t spaceErrors(const FormRec& f, uint32_t b, uint32_t e) const
			{
				uint32_t nErr = 0, off = 0;
				if (!f.numSpaces)      // a form without spaces: every gap inside the span is an error; the form's characters are not needed
				{
					for (uint32_t i = 1; i < e - b; ++i) nErr += (nsToPos[b + i] - nsToPos[b + i - 1] > 1) ? 1u : 0u;
					return nErr;
				}
				const uint16_t* fs = M.formChars + f.charOff;
				for (uint32_t i = 1; i < e - b; ++i)
				{
					const bool hasSpace = nsToPos[b + i] - nsToPos[b + i - 1] > 1;
					const uint16_t fc = (i + off < f.len) ? fs[i + off] : 0;
					if (hasSpace && fc != u' ') ++nErr;
					if (fc == u' ') ++off;
				}
				return nErr;
			}
			// value: the child's TrieNodeRec::value where the edge slot carries it (every step below the root); kValueUnknown below the root's table
			static constexpr int32_t kValueUnknown = -3;
			__device__ __forceinline__ int32_t trieNext(uint32_t node, uint16_t c, int32_t& value) const
			{
				value = kValueUnknown;
				if (node == 0) { const uint32_t r = M.trieRoot[c]; return r ? (int32_t)r : -1; }
				// (the edge hash: one dependent load per probe instead of record + binary search + child, flat_model.hpp TrieEdgeSlot)
				uint32_t h = trieEdgeHash(node, c) & M.trieEdgeMask;
				for (;;)
				{
					const uint4 s = reinterpret_cast<const uint4*>(M.trieEdges)[h];
					if (s.x == node && s.y == (uint32_t)c) { value = (int32_t)s.w; return (int32_t)s.z; }
					if (s.x == TRIE_EDGE_EMPTY) return -1;
					h = (h + 1) & M.trieEdgeMask;
				}
			}
			__device__ __forceinline__ int32_t trieNext(uint32_t node, uint16_t c) const { int32_t v; return trieNext(node, c, v); }
			__device__ __forceinline__ uint16_t formChar(const TypoGraphNode& g, uint32_t j) const
			{
				return (g.formOff & TYPO_FORM_IN_POOL) ? V.pool[(g.formOff & ~TYPO_FORM_IN_POOL) + j] : str[g.formOff + j];
			}

			// candidates: form id in the low 24 bits, syllables skipped as lengthening in the high 8
			__device__ __forceinline__ void flush(uint32_t* cands, uint32_t& nCands, uint32_t endNs, int32_t startPosOffset, uint32_t unkStart, uint32_t boundary, float typoCost, uint32_t startCti, uint32_t endCti)
			{
				for (uint32_t k = 0; k < nCands; ++k)
				{
					const uint32_t fi = cands[k] & 0xFFFFFFu, lengthened = cands[k] >> 24;
					const FormRec f = M.forms[fi];
					const uint32_t nb = (uint32_t)((int32_t)endNs - (int32_t)(f.len - f.numSpaces) - (int32_t)lengthened + startPosOffset), ne = endNs;
					if (startCti == 0 && !(f.flags & FF_FIRST_IS_CODA))
					{
						const bool hj = (f.flags & FF_HAS_JCLASS) || (f.flags & FF_IS_STAG);
						if (boundary < nb) insertUnk(boundary, nb, hj);
						insertUnk(unkStart, nb, hj);
					}
					const uint32_t se = spaceErrors(f, nb, ne);
					if (se <= V.spaceTol)
					{
						const uint32_t b2 = startCti ? (nb << pmb) + startCti : nb << pmb;
						const uint32_t e2 = endCti ? ((ne - 1) << pmb) + endCti : ne << pmb;
						if (append(b2, e2, fi, 0, 0, typoCost + (lengthened ? V.lengtheningCost * (float)(3 + lengthened) : 0.f), &f)) setLastSpaceErrors(se);
					}
				}
				nCands = 0;
			}

			// (a state's fixed part: its lengthening lists are read entry by entry, nL of them)
			struct Head { int32_t node; float cost; uint32_t minFormLen; int32_t startPosOffset; uint32_t specialStart, unkStart, boundary; uint8_t lastType, lastScript, hasLast, pad; uint16_t startCti, pad2; uint32_t lastChr, nL; };
			static_assert(sizeof(Head) == offsetof(TypoState, lsize) && sizeof(Head) == 4 * kTypoStateHeadWords, "TypoState head");

			// progressNode: state `st` of graph node `prevT` continued through graph node `tn`; new states go to cur[0..nCur).  stHead: the state's head where the
			// caller holds it (the LDS ring of recent states), null: read from stRef; ring: where the head of a NEW state is mirrored (slot = absolute state index
			// modulo kTypoLdsRing; null: nowhere), ringTop: absolute index of cur[0]
			__device__ __forceinline__ static Head headOf(const TypoState& s) { return *reinterpret_cast<const Head*>(&s); }
			// (the eleven words of a head, field by field: a head whose address is taken lives in scratch memory)
			__device__ __forceinline__ static void headWords(const Head& h, uint32_t (&w)[kTypoStateHeadWords])
			{
				w[0] = (uint32_t)h.node; w[1] = __float_as_uint(h.cost); w[2] = h.minFormLen; w[3] = (uint32_t)h.startPosOffset; w[4] = h.specialStart; w[5] = h.unkStart; w[6] = h.boundary;
				w[7] = (uint32_t)h.lastType | ((uint32_t)h.lastScript << 8) | ((uint32_t)h.hasLast << 16) | ((uint32_t)h.pad << 24); w[8] = (uint32_t)h.startCti | ((uint32_t)h.pad2 << 16); w[9] = h.lastChr; w[10] = h.nL;
			}
			__device__ __forceinline__ static Head headFromWords(const uint32_t (&w)[kTypoStateHeadWords])
			{
				Head h; h.node = (int32_t)w[0]; h.cost = __uint_as_float(w[1]); h.minFormLen = w[2]; h.startPosOffset = (int32_t)w[3]; h.specialStart = w[4]; h.unkStart = w[5]; h.boundary = w[6];
				h.lastType = (uint8_t)w[7]; h.lastScript = (uint8_t)(w[7] >> 8); h.hasLast = (uint8_t)(w[7] >> 16); h.pad = (uint8_t)(w[7] >> 24); h.startCti = (uint16_t)w[8]; h.pad2 = (uint16_t)(w[8] >> 16); h.lastChr = w[9]; h.nL = w[10];
				return h;
			}
			__device__ __forceinline__ void progress(const TypoGraphNode& prevT, const TypoGraphNode& tn, uint32_t tnIdx, const TypoState& stRef, const Head st, TypoState* cur, uint32_t& nCur, uint32_t curCap,
				uint32_t* ring = nullptr, uint32_t ringTop = 0, int32_t lastPair = -1)
			{
				float typoCost = st.cost + tn.typoCost;
				if (typoCost > V.threshold) return;
				uint8_t lastType = st.hasLast ? st.lastType : (uint8_t)T_UNKNOWN;
				uint8_t lastScript = st.hasLast ? st.lastScript : (uint8_t)0;
				uint8_t outType = st.lastType, outScript = st.lastScript, outHas = st.hasLast;      // what the last character of this node leaves behind
				uint32_t specialStart = st.specialStart, unkStart = st.unkStart, boundary = st.boundary;
				uint32_t minFormLen = st.minFormLen;
				int32_t startPosOffset = st.startPosOffset;
				const uint32_t fsz = tn.formLen;
				if (tn.typoCost > 0) startPosOffset += (int32_t)fsz - (int32_t)(tn.endPos - prevT.endPos);
				// (type and script of the node's last character: from the caller where it holds them -- lastPair = type | script << 8, the LDS copy --, else TypoLatView::graphLast)
				if (fsz)
				{
					if (lastPair >= 0) { outType = (uint8_t)(lastPair & 0xFF); outScript = (uint8_t)(lastPair >> 8); }
					else { outType = V.graphLast[2 * tnIdx]; outScript = V.graphLast[2 * tnIdx + 1]; }
					outHas = outType != 0xFF;
				}
				int32_t curNode = st.node;
				const uint8_t scriptVS = 98;
				uint32_t candsLocal[LDS ? 1 : MAXCAND]; uint32_t* cands; uint32_t nCands = 0;
				if constexpr (LDS) cands = candBuf; else cands = candsLocal;      // (no select between address spaces: the LDS buffer stays an LDS pointer)
				auto push = [&](uint32_t f) { if (nCands < MAXCAND) cands[nCands++] = f; else overflow = true; };      // (form ids stay below 2^24: checked by the engine)
				const bool lengthening = V.lengtheningCost < INFINITY;
				uint32_t prevChr = st.lastChr;
				uint8_t lsz[kTypoLengthNodes]; int32_t lnd[kTypoLengthNodes]; uint32_t nL = st.nL;
				for (uint32_t k = 0; k < nL; ++k) { lsz[k] = stRef.lsize[k]; lnd[k] = stRef.lnode[k]; }
				for (uint32_t j = 0; j < fsz; ++j)
				{
					const uint16_t ch = formChar(tn, j);
					uint32_t c32 = ch;
					if (isHighSurrogate(c32) && j + 1 < fsz) c32 = mergeSurrogate(c32, formChar(tn, j + 1));
					const uint32_t pos = tn.endPos + j - fsz;
					if (typoCost == 0)
					{
						const bool inPattern = pat != patEnd && pos >= pat->end - pat->length;
						uint8_t type = cls[pos] & 0x3F, sct = script[pos];
						if (lastType == T_SW && (c32 == 0x200d || (0x1f3fb <= c32 && c32 <= 0x1f3ff) || sct == scriptVS)) { type = lastType; sct = lastScript; }
						const uint8_t curT = inPattern ? (uint8_t)T_UNKNOWN : type;
						const bool symA = lastType == T_SL || lastType == T_SH || lastType == T_SW, symB = curT == T_SL || curT == T_SH || curT == T_SW;
						const bool discont = (symA && symB) ? (lastScript != sct) : (lastType != curT);
						if (discont || lastType == T_SSO || lastType == T_SSC)
						{
							if (lastType != T_MAX && lastType != T_UNKNOWN && lastType != T_SS)
							{
								const bool sj = T_SF <= lastType && lastType <= T_SW;
								unkPair(boundary, unkStart, specialStart, sj);
								uint32_t o, l; trimmed(nsToPos[specialStart], pos - nsToPos[specialStart], o, l);
								append(specialStart << pmb, (uint32_t)posToNs[pos] << pmb, lastType - 1u, o, l);
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
							const uint32_t p = posToNs[pos];
							if (p < nNs)
							{
								if constexpr (LDS) { const uint8_t zb = zAt[p]; zc = zb & FF_ZCODA_APPENDABLE; zs = zb & FF_ZSIOT_APPENDABLE; }
								else
								{
									const uint2 m = endPosMap[p << pmb];
									if (m.x != NPOS) for (uint32_t i = m.x; i < m.y; ++i)
									{
										const TypoLatNode g = out[i];
										if (g.endPos != (p << pmb) || g.form < 0) continue;
										const uint8_t ff = M.forms[g.form].flags;
										zc = zc || (ff & FF_ZCODA_APPENDABLE); zs = zs || (ff & FF_ZSIOT_APPENDABLE);
									}
								}
							}
							if ((V.match & M_Z_CODA) && zc && isHangulCoda(ch) && (pos + 1 >= n || !isHangulSyllable(str[pos + 1]))) push(kDefaultTagSize + (ch - 0x11A8) - 1u);
							else if ((V.match & (M_SPLIT_SAISIOT | M_MERGE_SAISIOT)) && zs && ch == 0x11BA && pos + 1 < n && isHangulSyllable(str[pos + 1])) push(kDefaultTagSize + (0x11BA - 0x11A8) - 1u);
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
							append((uint32_t)posToNs[ms] << pmb, (uint32_t)posToNs[pat->end] << pmb, pat->tag - 1u, ms, pat->length);
							++pat;
						}
					}
					if (c32 >= 0x10000) { ++j; prevChr = c32; continue; }
					if (lengthening)      // KTrie.cpp:1215-1270
					{
						const uint8_t lengtheningVowel[21] = { 0, 1, 0, 1, 4, 5, 4, 5, 8, 0, 1, 1, 8, 13, 4, 5, 20, 13, 18, 20, 20 };
						const uint32_t prevSize = nL;
						if (prevChr && prevChr < 0x10000 && isHangulSyllable((uint16_t)prevChr) && (0xC544 <= ch && ch < 0xC790)
							&& lengtheningVowel[((prevChr - 0xAC00) / 28) % 21] == ((ch - 0xAC00) / 28) % 21)
						{
							if (nL < kTypoLengthNodes) { lsz[nL] = 1; lnd[nL] = curNode; ++nL; } else overflow = true;
							for (uint32_t k = 0; k < prevSize; ++k) if (lsz[k] < 8) { if (nL < kTypoLengthNodes) { lsz[nL] = (uint8_t)(lsz[k] + 1); lnd[nL] = lnd[k]; ++nL; } else overflow = true; }
						}
						uint32_t outIdx = 0;
						for (uint32_t k = 0; k < nL; ++k)
						{
							uint8_t sz = lsz[k]; int32_t nd = lnd[k];
							if (k < prevSize) { nd = trieNext((uint32_t)nd, ch); lnd[k] = nd; if (nd < 0) continue; }
							bool dup = false;
							for (uint32_t q = 0; q < outIdx; ++q) dup = dup || (lsz[q] == sz && lnd[q] == nd);
							if (dup) continue;
							lsz[outIdx] = sz; lnd[outIdx] = nd; ++outIdx;
						}
						nL = outIdx;
					}
					prevChr = c32;

					if (minFormLen > 0 || tn.typoCost > 0) ++minFormLen;
					int32_t nxValue;
					int32_t nx = trieNext((uint32_t)curNode, ch, nxValue);
					while (nx < 0)
					{
						curNode = M.trie[curNode].fail;
						if (curNode < 0) break;
						nx = trieNext((uint32_t)curNode, ch, nxValue);
					}
					if (nx >= 0)
					{
						curNode = nx;
						if (tn.typoCost == 0 || j == fsz - 1)
						{
							// (no form and no submatch ends at this node -- the edge slot said so: the chain below would stop at its first record, which is not read)
							if (nxValue == TRIE_NONE) {}
							else if (typoCost > 0 && M.trie[curNode].depth < minFormLen) {}      // early pruning
							else for (int32_t sm = curNode; sm >= 0; sm = M.trie[sm].fail)
							{
								const int32_t v = M.trie[sm].value;
								if (v == TRIE_NONE) break;
								if (v != TRIE_SUBMATCH)
								{
									if (M.forms[v].len < minFormLen) break;
									push((uint32_t)v);
								}
							}
							for (uint32_t k = 0; k < nL; ++k)
							{
								const int32_t v = M.trie[lnd[k]].value;
								if (v >= 0 && M.forms[v].len >= minFormLen) push((uint32_t)v | ((uint32_t)lsz[k] << 24));
							}
						}
					}
					else
					{
						nL = 0;
						if (typoCost == 0) curNode = 0;
						else return;
					}
					flush(cands, nCands, posToNs[tn.endPos + j + 1 - fsz], startPosOffset, unkStart, boundary, typoCost, st.startCti, tn.continualTypoIdx);
				}
				if (typoCost == 0 && lastType != T_MAX && lastType != T_UNKNOWN && lastType != T_SS)
				{
					const bool sj = T_SF <= lastType && lastType <= T_SW;
					unkPair(boundary, unkStart, specialStart, sj);
					uint32_t o, l; trimmed(nsToPos[specialStart], tn.endPos - nsToPos[specialStart], o, l);
					append(specialStart << pmb, (uint32_t)posToNs[tn.endPos] << pmb, lastType - 1u, o, l);
					unkStart = specialStart;
					if (sj) boundary = posToNs[tn.endPos];
				}
				if (curNode >= 0)
				{
					if (tn.continualTypoIdx)
					{
						curNode = 0; typoCost = 0; minFormLen = 0; startPosOffset = -1;
						if (nCur) return;
						nL = 0;
					}
					if (typoCost > 0 && M.trie[curNode].depth < minFormLen && nL == 0) return;      // early pruning
					if (nCur >= curCap) { overflow = true; return; }
					Head ns; ns.node = curNode; ns.cost = typoCost; ns.minFormLen = minFormLen; ns.startPosOffset = startPosOffset;
					ns.specialStart = specialStart; ns.unkStart = unkStart; ns.boundary = boundary;
					ns.lastType = outType; ns.lastScript = outScript; ns.hasLast = outHas; ns.pad = 0;
					ns.startCti = tn.continualTypoIdx ? tn.continualTypoIdx : st.startCti; ns.pad2 = 0;
					ns.lastChr = prevChr; ns.nL = nL;
					uint32_t nw[kTypoStateHeadWords];
					headWords(ns, nw);
					if (ring)
					{
						uint32_t* slot = ring + ((ringTop + nCur) % kTypoLdsRing) * kTypoStateHeadWords;
						for (uint32_t q = 0; q < kTypoStateHeadWords; ++q) slot[q] = nw[q];
					}
					TypoState& dst = cur[nCur++];
					{ uint32_t* dw = reinterpret_cast<uint32_t*>(&dst); for (uint32_t q = 0; q < kTypoStateHeadWords; ++q) dw[q] = nw[q]; }
					for (uint32_t k = 0; k < nL; ++k) { dst.lsize[k] = lsz[k]; dst.lnode[k] = lnd[k]; }
				}
			}
		};
	}

	// One THREAD per chunk, `stride` lanes apart: a strictly serial, branch-heavy replay runs at the speed of the SUM of its lanes' paths when 64
	// chunks share a wavefront (lanes diverge at every branch).  With few active lanes per wave the chunks spread over all SIMDs instead
	// (8192 chunks: 8192 one-lane waves resident at once) and a wave's time is one chunk's time.
	__global__ void __launch_bounds__(64) k_build_lattice_typo(ModelView M, TypoLatView V, uint32_t nChunks, uint32_t stride, uint32_t ldsBudget)
	{
		if (threadIdx.x % stride) return;
		const uint32_t c = blockIdx.x * (64 / stride) + threadIdx.x / stride;
		if (c >= nChunks) return;
		TypoLatChunk& C = V.chunks[c];
		if (ldsBudget && C.ldsNeed <= ldsBudget && C.status != kTypoLdsNeedsBig) return;      // done by the wave-per-chunk kernel
		C.pad = C.status == kTypoLdsNeedsBig ? 1u : 0u;      // (developer statistics: this chunk outgrew its LDS copy)
		CtxT<false> X{ M, V, C };
		X.mapLen = C.mapLen;
		X.str = V.chars + C.charOff; X.cls = V.cls + C.charOff; X.script = V.script + C.charOff; X.n = C.nChars; X.pmb = C.pmb;
		uint16_t* nsToPos = V.nsToPos + C.nsOff; uint16_t* posToNs = V.posToNs + C.nsOff;
		uint32_t nNs = 0;
		for (uint32_t i = 0; i < X.n; ++i)
		{
			posToNs[i] = (uint16_t)nNs;
			if (!isSpace(X.str[i]))
			{
				nsToPos[nNs++] = (uint16_t)i;
				if (isHighSurrogate(X.str[i]) && i + 1 < X.n) { posToNs[i + 1] = (uint16_t)nNs; nsToPos[nNs++] = (uint16_t)(i + 1); ++i; }
			}
		}
		posToNs[X.n] = (uint16_t)nNs;
		X.nNs = nNs; X.nsToPos = nsToPos; X.posToNs = posToNs;
		X.endPosMap = V.endPosMap + C.mapOff; X.out = V.nodes + C.nodeOff; X.nOut = 0; X.overflow = false;
		X.pat = V.patterns + C.patOff; X.patEnd = X.pat + C.patCnt;
		auto fail = [&](uint32_t code) { C.status = code; if (V.results) V.results[C.chunkId].status = code; };
		if (V.results && V.results[C.chunkId].status >= 16) { C.status = V.results[C.chunkId].status; return; }
		if (((nNs << X.pmb) + 1) > C.mapLen || nNs != C.nNs) { fail(CS_ERR_TOO_LONG); return; }
		for (uint32_t i = 0; i < C.mapLen; ++i) X.endPosMap[i] = make_uint2(NPOS, NPOS);
		X.endPosMap[0] = make_uint2(0, 1);
		{ TypoLatNode z{}; z.form = -1; X.out[X.nOut++] = z; }

		// search (KTrie.cpp:1414-1452): states of graph node i are the contiguous run stateIdx[2i] .. +stateIdx[2i+1] of the chunk's arena
		const TypoGraphNode* graph = V.graph + C.graphOff;
		TypoState* states = V.states + C.stateOff;
		uint32_t* sIdx = V.stateIdx + 2 * C.graphOff;
		const uint32_t totEnd = nNs ? (uint32_t)nsToPos[nNs - 1] + 1 : 0;
		uint32_t top = 0;
		{ TypoState s0{}; states[top] = s0; sIdx[0] = 0; sIdx[1] = 1; ++top; }
		for (uint32_t i = 1; i < C.graphCnt; ++i)
		{
			const TypoGraphNode tn = graph[i];
			const uint32_t curBeg = top; uint32_t nCur = 0;
			for (uint32_t p = tn.prevOffset ? i - tn.prevOffset : NPOS; p != NPOS; p = graph[p].siblingOffset ? p + graph[p].siblingOffset : NPOS)
			{
				const TypoGraphNode pt = graph[p];
				for (uint32_t k = 0; k < sIdx[2 * p + 1]; ++k)
					{ const TypoState& sr = states[sIdx[2 * p] + k]; X.progress(pt, tn, C.graphOff + i, sr, CtxT<false>::headOf(sr), states + curBeg, nCur, C.stateCap - curBeg); }
			}
			sIdx[2 * i] = curBeg; sIdx[2 * i + 1] = nCur; top = curBeg + nCur;
			if (tn.typoCost == 0 && tn.endPos == totEnd)
				for (uint32_t k = 0; k < nCur; ++k) X.unkPair(states[curBeg + k].boundary, states[curBeg + k].unkStart, posToNs[totEnd], true);
		}
		X.append(nNs << X.pmb, (nNs << X.pmb) + 1, NOFORM, 0, 0);
		X.out[X.nOut - 1].endPos = nNs << X.pmb;
		if (X.overflow || X.nOut + 1 >= C.nodeCap) { fail(CS_ERR_NODE_OVERFLOW); return; }

		// removeUnconnected (KTrie.cpp:240-299): reachable from the end node backwards; stable order by (connected, end position)
		const uint32_t G = X.nOut;
		uint32_t* conn = V.scratch + 3ull * C.nodeOff; uint32_t* sorted = conn + C.nodeCap; uint32_t* inv = sorted + C.nodeCap;
		for (uint32_t i = 0; i < G; ++i) conn[i] = 0;
		uint32_t qh = 0, qt = 0;
		sorted[qt++] = G - 1; conn[G - 1] = 1;      // (`sorted` doubles as the BFS queue first)
		while (qh < qt)
		{
			const uint32_t id = sorted[qh++];
			const uint2 m = X.endPosMap[X.out[id].startPos];
			if (m.x == NPOS) continue;
			for (uint32_t i = m.x; i < m.y; ++i)
			{
				if (X.out[i].endPos != X.out[id].startPos || conn[i]) continue;
				conn[i] = 1; sorted[qt++] = i;
			}
		}
		uint32_t nConn = 0;
		for (uint32_t i = 0; i < G; ++i) nConn += conn[i];
		// stable sort of the connected nodes by end position: insertion sort over the original order (lattices are a few hundred nodes)
		uint32_t k = 0;
		for (uint32_t i = 0; i < G; ++i)
		{
			if (!conn[i]) continue;
			uint32_t j = k++;
			const uint32_t e = X.out[i].endPos;
			while (j > 0 && X.out[sorted[j - 1]].endPos > e) { sorted[j] = sorted[j - 1]; --j; }
			sorted[j] = i;
		}
		for (uint32_t i = 0; i < G; ++i) inv[i] = NPOS;
		for (uint32_t i = 0; i < nConn; ++i) inv[sorted[i]] = i;
		if (V.devNodes)
		{
			// engine mode: the record the search kernel reads, with the predecessor-dependent facts of lattice_kernels.hip latticeEmitNode
			DevNode* dn = V.devNodes + C.nodeOff; float* tc = V.nodeTypo + C.nodeOff;
			if (nNs > 0xFFF0 || C.nodeCap > 0xFFF0 || nConn > 0xFFF0) { fail(CS_ERR_TOO_LONG); return; }
			uint32_t packTop = 0;
			for (uint32_t i = 0; i < nConn; ++i)
			{
				const uint32_t idx = sorted[i];
				const TypoLatNode g = X.out[idx];
				DevNode nn;
				nn.form = g.form < 0 ? NOFORM : (uint32_t)g.form; nn.uformOff = (uint16_t)g.uformOff; nn.uformLen = (uint16_t)g.uformLen; nn.spaceErrors = (uint8_t)g.spaceErrors;
				nn.nPrev = 0; nn.candCnt = 0; nn.fflags = 0; nn.flen = 0; nn.ownFeat = 0; nn.pad = 0; nn.prev = 0; nn.sibling = 0;
				uint8_t nf = 0;
				if (i >= 1)
				{
					const uint32_t pidx = idx - g.prev;
					const TypoLatNode pn = X.out[pidx];
					const uint32_t startStr = (i + 1 == nConn) ? X.n : (uint32_t)nsToPos[g.startPos >> X.pmb];
					const bool pnBos = pidx == 0;
					const uint32_t pnEndStr = pnBos ? 0 : (uint32_t)nsToPos[((pn.endPos + (1u << X.pmb) - 1) >> X.pmb) - 1] + 1;
					const bool spaceBefore = pnBos ? (C.textOffset + startStr > 0) : (pnEndStr < startStr);
					bool lb = pnBos || spaceBefore;
					if (!lb && pn.uformLen)
					{
						const uint32_t lp = pn.uformOff + pn.uformLen - 1;
						const uint16_t ch = X.str[lp];
						const uint8_t tag = (isLowSurrogate(ch) || isHighSurrogate(ch)) ? (uint8_t)T_SH : (uint8_t)(X.cls[lp] & 0x3F);
						if (tag == T_SSC || ch == u'"' || ch == u'\'') lb = false;
						else if (T_SF <= tag && tag <= T_SB) lb = true;
					}
					if (spaceBefore) nf |= NF_SPACE_BEFORE;
					if (lb) nf |= NF_LEFT_BOUNDARY;
					if (g.uformLen && X.str[g.uformOff + g.uformLen - 1] == u'.') nf |= NF_UFORM_ENDS_POINT;
					uint32_t np = 0;      // connected nodes ending where this one starts
					const uint2 m = X.endPosMap[g.startPos];
					if (m.x != NPOS) for (uint32_t j = m.x; j < m.y; ++j) if (X.out[j].endPos == g.startPos && conn[j]) ++np;
					nn.nPrev = (uint16_t)np;
				}
				if (g.form >= 0)
				{
					const FormRec f = M.forms[g.form];
					nn.candCnt = f.candCnt; nn.fflags = f.flags; nn.flen = f.len;
					if ((f.flags2 & FF2_ALL_PARTIAL) && g.typoCost == 0) nf |= NF_ALL_PARTIAL;      // (PathEvaluator.hpp:1275: only for nodes without a typo)
				}
				if (g.uformLen)
				{
					uint16_t of = featMask(X.str + g.uformOff, g.uformLen) & 0x1FFF;
					const uint32_t lp = g.uformOff + g.uformLen - 1;
					const uint16_t ch = X.str[lp];
					const uint8_t tag = (isLowSurrogate(ch) || isHighSurrogate(ch)) ? (uint8_t)T_SH : (uint8_t)(X.cls[lp] & 0x3F);
					if (tag == T_SSC) of |= LF_STR_SSC;
					nn.ownFeat = of;
				}
				nn.nflags = nf;
				if (g.prev) nn.prev = (uint16_t)(i - inv[idx - g.prev]);
				if (g.sibling) { const uint32_t ns = inv[idx + g.sibling]; nn.sibling = ns == NPOS ? 0 : (uint16_t)(ns - i); }
				if (i >= 1 && i + 1 < nConn)
				{
					nn.startPos = nsToPos[g.startPos >> X.pmb];
					nn.endPos = (uint16_t)(nsToPos[((g.endPos + (1u << X.pmb) - 1) >> X.pmb) - 1] + 1);
				}
				else if (i + 1 == nConn) nn.startPos = nn.endPos = (uint16_t)X.n;
				else nn.startPos = nn.endPos = 0;
				nn.pad = (uint16_t)g.endPos;      // (the MULTIPLIED end position: nodes of equal text end but different halves of a continual typo are different steps of the position-step search, k_expand_pos)
				nn.packOff = packTop; packTop += nn.candCnt;
				dn[i] = nn; tc[i] = g.typoCost;
			}
			if (packTop > C.packCap) { fail(CS_ERR_NODE_OVERFLOW); return; }
			V.nNodes[C.chunkId] = nConn;
			C.nOutFinal = nConn; C.status = CS_OK;
			if (nConn <= 2) V.results[C.chunkId].status = CS_NO_LATTICE;
			return;
		}
		TypoLatNode* fin = V.nodesFinal + C.nodeOff;
		for (uint32_t i = 0; i < nConn; ++i)
		{
			const uint32_t idx = sorted[i];
			TypoLatNode nn = X.out[idx];
			if (nn.prev) nn.prev = i - inv[idx - nn.prev];
			if (nn.sibling)
			{
				const uint32_t ns = inv[idx + nn.sibling];
				nn.sibling = ns == NPOS ? 0 : ns - i;
			}
			if (i >= 1 && i + 1 < nConn)      // writeResult (KTrie.cpp:1454-1464)
			{
				nn.startPos = (uint32_t)nsToPos[nn.startPos >> X.pmb] + C.textOffset;
				nn.endPos = (uint32_t)nsToPos[((nn.endPos + (1u << X.pmb) - 1) >> X.pmb) - 1] + 1 + C.textOffset;
				if (nn.uformLen) nn.uformOff += C.textOffset;
			}
			else if (i + 1 == nConn) { nn.startPos = nn.endPos = C.textOffset + X.n; }
			fin[i] = nn;
		}
		C.nOutFinal = nConn;
		C.status = CS_OK;
	}

	// dynamic LDS of the wave-per-chunk variant
	extern __shared__ __align__(16) uint8_t tSmem[];

	// One WAVE per chunk (engine mode).  The search over the typo graph is the same strictly sequential replay (lane 0), but every access to
	// the chunk's own data is an LDS access instead of a dependent HBM round trip; search states and the typo graph stay in HBM (read once per
	// step).  The final reorder and the per-node facts run one node per lane.  A chunk whose node list outgrows its LDS copy is handed to
	// the thread-per-chunk kernel (TypoLatChunk::status = kTypoLdsNeedsBig).
	#ifndef KAMD_TYPO_LDS_WPS
#define KAMD_TYPO_LDS_WPS 5      // (79 VGPRs, five wavefronts per SIMD = the 20 per CU its 8 KB of LDS allow: c5 lattice 2.16 -> 2.01 ms, profiles/r06_g; 8: no gain)
#endif
	__global__ void __launch_bounds__(64, KAMD_TYPO_LDS_WPS) k_build_lattice_typo_lds(ModelView M, TypoLatView V, const uint32_t* chunkList, uint32_t chunkCount, uint32_t ldsBytes)
	{
		if (blockIdx.x >= chunkCount) return;
		const uint32_t lane = threadIdx.x;
		TypoLatChunk& C = V.chunks[chunkList[blockIdx.x]];
		if (V.results[C.chunkId].status >= 16) { if (lane == 0) C.status = V.results[C.chunkId].status; return; }
		const uint32_t n = C.nChars, pmb = C.pmb;
		const TypoLds lay = typoLdsLayout(n, C.nNs, pmb, C.ldsCap, C.ldsGraphCap);
		if (lay.total > ldsBytes || C.mapLen > 0xFFF0 || C.nodeCap > 0xFFF0 || C.mapLen != (C.nNs << pmb) + 1 || C.mapLen > 64 * (C.nNs + 1)) { if (lane == 0) C.status = kTypoLdsNeedsBig; return; }
		uint16_t* str = reinterpret_cast<uint16_t*>(tSmem + lay.str); uint8_t* cls = tSmem + lay.cls; uint8_t* script = tSmem + lay.script;
		uint16_t* nsToPos = reinterpret_cast<uint16_t*>(tSmem + lay.nsToPos); uint16_t* posToNs = reinterpret_cast<uint16_t*>(tSmem + lay.posToNs);
		uint32_t* epm = reinterpret_cast<uint32_t*>(tSmem + lay.epm); uint64_t* fullMask = reinterpret_cast<uint64_t*>(tSmem + lay.fullMask); uint8_t* zAt = tSmem + lay.zAt;
		TypoLdsNode* lout = reinterpret_cast<TypoLdsNode*>(tSmem + lay.nodes);
		{
			const uint16_t* gstr = V.chars + C.charOff; const uint8_t* gcls = V.cls + C.charOff; const uint8_t* gscript = V.script + C.charOff;
			for (uint32_t i = lane; i < n; i += 64) { str[i] = gstr[i]; cls[i] = gcls[i]; script[i] = gscript[i]; }
			for (uint32_t i = lane; i < C.mapLen; i += 64) epm[i] = 0;      // first == last + 1 - 1 ... : lo == hi, empty
			for (uint32_t i = lane; i <= C.nNs; i += 64) { fullMask[i] = 0; zAt[i] = 0; }
		}
		// the chunk's typo graph beside them (28 bytes per node, read word by word: coalesced), the state ranges and the ring of state heads follow
		const uint32_t gCap = lay.graphCap;
		TypoLdsGraphNode* gL = reinterpret_cast<TypoLdsGraphNode*>(tSmem + lay.graph); uint32_t* sL = reinterpret_cast<uint32_t*>(tSmem + lay.sidx); uint32_t* ringL = gCap ? reinterpret_cast<uint32_t*>(tSmem + lay.ring) : nullptr;
		uint16_t* glL = reinterpret_cast<uint16_t*>(tSmem + lay.glast);
		for (uint32_t i = lane; i < gCap; i += 64)
		{
			glL[i] = reinterpret_cast<const uint16_t*>(V.graphLast)[C.graphOff + i];      // (type | script << 8)
			const TypoGraphNode g = V.graph[C.graphOff + i];
			TypoLdsGraphNode c; c.formOff = g.formOff; c.typoCost = g.typoCost; c.formLen = (uint16_t)g.formLen; c.endPos = (uint16_t)g.endPos;
			c.prevOffset = (uint8_t)g.prevOffset; c.siblingOffset = (uint8_t)g.siblingOffset; c.continualTypoIdx = g.continualTypoIdx; c.pad = 0;
			gL[i] = c;
		}
		waveSync();
		CtxT<true> X{ M, V, C };
		X.mapLen = C.mapLen;
		X.str = str; X.cls = cls; X.script = script; X.n = n; X.pmb = pmb; X.nsToPos = nsToPos; X.posToNs = posToNs;
		X.endPosMap = nullptr; X.out = nullptr; X.epm = epm; X.lout = lout; X.fullMask = fullMask; X.zAt = zAt; X.candBuf = reinterpret_cast<uint32_t*>(tSmem + lay.cands);
		X.ldsCap = lay.nodeCap; X.lastEnd = 0; X.outgrown = false; X.nOut = 0; X.overflow = false;
		X.pat = V.patterns + C.patOff; X.patEnd = X.pat + C.patCnt;
		uint16_t* inv = reinterpret_cast<uint16_t*>(tSmem + lay.queue);      // BFS queue first, then old -> new index
		uint16_t* conn = reinterpret_cast<uint16_t*>(tSmem + lay.conn);      // connected flags, then candidate counts by new index
		uint32_t err = 0, G = 0, nConn = 0;
		if (lane == 0)
		{
			uint32_t nNs = 0;
			for (uint32_t i = 0; i < n; ++i)
			{
				posToNs[i] = (uint16_t)nNs;
				if (!isSpace(str[i]))
				{
					nsToPos[nNs++] = (uint16_t)i;
					if (isHighSurrogate(str[i]) && i + 1 < n) { posToNs[i + 1] = (uint16_t)nNs; nsToPos[nNs++] = (uint16_t)(i + 1); ++i; }
				}
			}
			posToNs[n] = (uint16_t)nNs;
			X.nNs = nNs;
			if (nNs != C.nNs) err = CS_ERR_TOO_LONG;
			else
			{
				epm[0] = 0 | (1u << 16);
				{ TypoLdsNode z{}; z.form = NOFORM; lout[X.nOut++] = z; }
				// search (KTrie.cpp:1414-1452): states of graph node i are the contiguous run stateIdx[2i] .. +stateIdx[2i+1] of the chunk's arena
				const TypoGraphNode* graph = V.graph + C.graphOff;
				TypoState* states = V.states + C.stateOff;
				uint32_t* sIdx = V.stateIdx + 2 * C.graphOff;
				// (graph records / state ranges from the LDS copies when the chunk has them, gCap = graphCnt, else from HBM)
				auto graphAt = [&](uint32_t i) -> TypoGraphNode
				{
					if (i < gCap)
					{
						const TypoLdsGraphNode c = gL[i];
						TypoGraphNode r; r.formOff = c.formOff; r.formLen = c.formLen; r.endPos = c.endPos; r.typoCost = c.typoCost; r.prevOffset = c.prevOffset; r.siblingOffset = c.siblingOffset;
						r.continualTypoIdx = c.continualTypoIdx; r.pad = 0; r.dialect = 0;      // (the dialect of a replacement was applied when the graph was made: the lattice build does not read it)
						return r;
					}
					return graph[i];
				};
				auto setRange = [&](uint32_t i, uint32_t beg, uint32_t cnt) { if (i < gCap) sL[i] = beg | (cnt << 16); else { sIdx[2 * i] = beg; sIdx[2 * i + 1] = cnt; } };      // (beg < stateCap <= 65535 where the tables exist; a node has far fewer states)
				const uint32_t totEnd = nNs ? (uint32_t)nsToPos[nNs - 1] + 1 : 0;
				uint32_t top = 0;
				{
					TypoState s0{}; states[top] = s0; setRange(0, 0, 1);
					if (ringL) for (uint32_t q = 0; q < kTypoStateHeadWords; ++q) ringL[q] = 0;
					++top;
				}
#if defined(KAMD_HIPEMU) && defined(KAMD_TYPOSTATS)
				// developer statistics of a host (lane-emulated) build: automaton steps of the replay as it is (one (node, predecessor, state) after the other), with the
				// triples of a NODE side by side, and with the triples of a LEVEL (nodes none of whose predecessors is in the level) side by side
				unsigned long long wSeq = 0, wNode = 0, wLevel = 0, calls = 0; uint32_t levelStart = 1, levelMax = 0;
#endif
				const uint32_t graphCnt = C.graphCnt, graphOff = C.graphOff, stateCap = C.stateCap;      // (the chunk record is in HBM: read once)
				for (uint32_t i = 1; i < graphCnt; ++i)
				{
					const TypoGraphNode tn = graphAt(i);
					const int32_t lastPair = i < gCap ? (int32_t)glL[i] : -1;
					const uint32_t curBeg = top; uint32_t nCur = 0;
#if defined(KAMD_HIPEMU) && defined(KAMD_TYPOSTATS)
					{
						uint32_t nCalls = 0, maxPred = 0;
						for (uint32_t p = tn.prevOffset ? i - tn.prevOffset : NPOS; p != NPOS; ) { const TypoGraphNode pt = graphAt(p); nCalls += p < gCap ? (sL[p] >> 16) : sIdx[2 * p + 1]; if (p > maxPred) maxPred = p; p = pt.siblingOffset ? p + pt.siblingOffset : NPOS; }
						if (maxPred >= levelStart) { wLevel += levelMax; levelStart = i; levelMax = 0; }
						if (nCalls) { wSeq += (unsigned long long)nCalls * tn.formLen; wNode += tn.formLen; if (tn.formLen > levelMax) levelMax = tn.formLen; calls += nCalls; }
						if (i + 1 == graphCnt) { wLevel += levelMax; fprintf(stderr, "[typostats] graph %u nodes, %llu calls, automaton steps: sequential %llu, per node %llu, per level %llu\n", graphCnt, calls, wSeq, wNode, wLevel); }
					}
#endif
					for (uint32_t p = tn.prevOffset ? i - tn.prevOffset : NPOS; p != NPOS; )
					{
						const TypoGraphNode pt = graphAt(p);
						const uint32_t pr = p < gCap ? sL[p] : 0u;
						const uint32_t pBeg = p < gCap ? (pr & 0xFFFFu) : sIdx[2 * p], pCnt = p < gCap ? (pr >> 16) : sIdx[2 * p + 1];
						for (uint32_t k = 0; k < pCnt; ++k)
						{
							const uint32_t a = pBeg + k;      // absolute index of the state: its head is in the ring while it is one of the last kTypoLdsRing
							CtxT<true>::Head hd;
							if (ringL && curBeg + nCur - a <= kTypoLdsRing)
							{
								uint32_t w[kTypoStateHeadWords];
								const uint32_t* slot = ringL + (a % kTypoLdsRing) * kTypoStateHeadWords;
								for (uint32_t q = 0; q < kTypoStateHeadWords; ++q) w[q] = slot[q];
								hd = CtxT<true>::headFromWords(w);
							}
							else hd = CtxT<true>::headOf(states[a]);
							X.progress(pt, tn, graphOff + i, states[a], hd, states + curBeg, nCur, stateCap - curBeg, ringL, curBeg, lastPair);
						}
						p = pt.siblingOffset ? p + pt.siblingOffset : NPOS;
					}
					setRange(i, curBeg, nCur); top = curBeg + nCur;
					if (tn.typoCost == 0 && tn.endPos == totEnd)
						for (uint32_t k = 0; k < nCur; ++k) X.unkPair(states[curBeg + k].boundary, states[curBeg + k].unkStart, posToNs[totEnd], true);
				}
				X.append(nNs << pmb, (nNs << pmb) + 1, NOFORM, 0, 0);
				lout[X.nOut - 1].endPos = (uint16_t)(nNs << pmb);
				if (X.outgrown || (X.nOut + 1 >= lay.nodeCap && lay.nodeCap < C.nodeCap)) err = kTypoLdsNeedsBig;
				else if (X.overflow || X.nOut + 1 >= C.nodeCap) err = CS_ERR_NODE_OVERFLOW;
				else { G = X.nOut; C.pad = G << 1; }      // (developer statistics: nodes built; bit 0 = outgrew its LDS copy -- set by the thread-per-chunk kernel)
			}
		}
		waveSync();
		err = __shfl(err, 0); G = __shfl(G, 0);
		if (!err)
		{
			// removeUnconnected (KTrie.cpp:240-299), all lanes: reachability by a downward sweep over the positions, new index = nodes grouped by end
			// position ascending, original order inside a group (lattice_connect.hpp); the length masks of the build serve as its flag bits
			nConn = latticeConnectWave(lout, epm, inv, conn, reinterpret_cast<uint32_t*>(fullMask), G, C.mapLen, lane);
			if (C.nNs > 0xFFF0 || nConn > 0xFFF0) err = CS_ERR_TOO_LONG;
		}
		if (err)
		{
			if (lane == 0) { C.status = err; if (err != kTypoLdsNeedsBig) V.results[C.chunkId].status = err; }
			return;
		}
		// ---- final records, one node per lane (the facts of lattice_kernels.hip latticeEmitNode); candidate-record offsets by a wave scan ----
		DevNode* dn = V.devNodes + C.nodeOff; float* tc = V.nodeTypo + C.nodeOff;
		uint16_t* cc = conn;
		waveSync();
		for (uint32_t base = 0; base < G; base += 64)
		{
			const uint32_t idx = base + lane;
			uint32_t cnt = 0xFFFFFFFFu, ni = 0xFFFF;
			if (idx < G) ni = inv[idx];
			if (ni != 0xFFFF)
			{
				const TypoLdsNode g = lout[idx];
				DevNode nn;
				nn.form = g.form; nn.uformOff = g.uformOff; nn.uformLen = g.uformLen; nn.spaceErrors = g.spaceErrors;
				nn.nPrev = 0; nn.candCnt = 0; nn.fflags = 0; nn.flen = 0; nn.ownFeat = 0; nn.pad = 0; nn.prev = 0; nn.sibling = 0;
				uint8_t nf = 0;
				if (ni >= 1)
				{
					const uint32_t pidx = idx - g.prev;
					const TypoLdsNode pn = lout[pidx];
					const uint32_t startStr = (ni + 1 == nConn) ? n : (uint32_t)nsToPos[g.startPos >> pmb];
					const bool pnBos = pidx == 0;
					const uint32_t pnEndStr = pnBos ? 0 : (uint32_t)nsToPos[((pn.endPos + (1u << pmb) - 1) >> pmb) - 1] + 1;
					const bool spaceBefore = pnBos ? (C.textOffset + startStr > 0) : (pnEndStr < startStr);
					bool lb = pnBos || spaceBefore;
					if (!lb && pn.uformLen)
					{
						const uint32_t lp = pn.uformOff + pn.uformLen - 1;
						const uint16_t ch = str[lp];
						const uint8_t tag = (isLowSurrogate(ch) || isHighSurrogate(ch)) ? (uint8_t)T_SH : (uint8_t)(cls[lp] & 0x3F);
						if (tag == T_SSC || ch == u'"' || ch == u'\'') lb = false;
						else if (T_SF <= tag && tag <= T_SB) lb = true;
					}
					if (spaceBefore) nf |= NF_SPACE_BEFORE;
					if (lb) nf |= NF_LEFT_BOUNDARY;
					if (g.uformLen && str[g.uformOff + g.uformLen - 1] == u'.') nf |= NF_UFORM_ENDS_POINT;
					nn.nPrev = (uint16_t)epm[g.startPos];
				}
				if (g.form != NOFORM)
				{
					const FormRec f = M.forms[g.form];
					nn.candCnt = f.candCnt; nn.fflags = f.flags; nn.flen = f.len;
					if ((f.flags2 & FF2_ALL_PARTIAL) && g.typoCost == 0) nf |= NF_ALL_PARTIAL;      // (PathEvaluator.hpp:1275: only for nodes without a typo)
				}
				if (g.uformLen)
				{
					uint16_t of = featMask(str + g.uformOff, g.uformLen) & 0x1FFF;
					const uint32_t lp = g.uformOff + g.uformLen - 1;
					const uint16_t ch = str[lp];
					const uint8_t tag = (isLowSurrogate(ch) || isHighSurrogate(ch)) ? (uint8_t)T_SH : (uint8_t)(cls[lp] & 0x3F);
					if (tag == T_SSC) of |= LF_STR_SSC;
					nn.ownFeat = of;
				}
				nn.nflags = nf;
				if (g.prev) nn.prev = (uint16_t)(ni - inv[idx - g.prev]);
				if (g.sibling) { const uint32_t ns = inv[idx + g.sibling]; nn.sibling = ns == 0xFFFF ? 0 : (uint16_t)(ns - ni); }
				if (ni >= 1 && ni + 1 < nConn)
				{
					nn.startPos = nsToPos[g.startPos >> pmb];
					nn.endPos = (uint16_t)(nsToPos[((g.endPos + (1u << pmb) - 1) >> pmb) - 1] + 1);
				}
				else if (ni + 1 == nConn) nn.startPos = nn.endPos = (uint16_t)n;
				else nn.startPos = nn.endPos = 0;
				nn.pad = g.endPos;      // (the multiplied end position, see the thread-per-chunk kernel)
				nn.packOff = 0;
				dn[ni] = nn; tc[ni] = g.typoCost;
				cnt = nn.candCnt;
			}
			// (cc aliases the connected flags: all lanes of this round have read theirs above through inv)
			if (cnt != 0xFFFFFFFFu) cc[ni] = (uint16_t)cnt;
		}
		waveSync();
		uint32_t packTop = 0;
		for (uint32_t base = 0; base < nConn; base += 64)
		{
			const uint32_t i = base + lane;
			const uint32_t c = i < nConn ? (uint32_t)cc[i] : 0u;
			uint32_t incl = c;
			for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d); if (lane >= d) incl += v; }
			if (i < nConn) dn[i].packOff = packTop + incl - c;
			packTop += __shfl(incl, 63);
		}
		if (lane == 0)
		{
			if (packTop > C.packCap) { C.status = CS_ERR_NODE_OVERFLOW; V.results[C.chunkId].status = CS_ERR_NODE_OVERFLOW; }
			else
			{
				V.nNodes[C.chunkId] = nConn;
				C.nOutFinal = nConn; C.status = CS_OK;
				if (nConn <= 2) V.results[C.chunkId].status = CS_NO_LATTICE;
			}
		}
	}

	static uint32_t strideFor(uint32_t nChunks)
	{
		// active lanes per wave: 1 up to 16k chunks, 4 up to 64k, 16 beyond (the machine holds 8192 waves)
		return nChunks <= 16384 ? 64u : nChunks <= 65536 ? 16u : 4u;
	}
	void launchTypoLattice(const ModelView& M, const TypoLatView& V, uint32_t nChunks, hipStream_t stream)
	{
		const uint32_t stride = strideFor(nChunks), perWave = 64 / stride;
		hipLaunchKernelGGL(k_build_lattice_typo, dim3((nChunks + perWave - 1) / perWave), dim3(64), 0, stream, M, V, nChunks, stride, 0u);
	}
	void launchTypoLatticeRest(const ModelView& M, const TypoLatView& V, uint32_t nChunks, uint32_t ldsBudget, hipStream_t stream)
	{
		const uint32_t stride = strideFor(nChunks), perWave = 64 / stride;
		hipLaunchKernelGGL(k_build_lattice_typo, dim3((nChunks + perWave - 1) / perWave), dim3(64), 0, stream, M, V, nChunks, stride, ldsBudget);
	}
	void launchTypoLatticeLds(const ModelView& M, const TypoLatView& V, const uint32_t* chunkList, uint32_t chunkCount, uint32_t ldsBytes, hipStream_t stream)
	{
		hipLaunchKernelGGL(k_build_lattice_typo_lds, dim3(chunkCount), dim3(64), ldsBytes, stream, M, V, chunkList, chunkCount, ldsBytes);
	}
}
