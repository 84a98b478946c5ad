// k_lattice_wave: the lattice of one chunk built by ALL 64 lanes of its wavefront (gfx950 / MI355X).
//
// The reference builds a chunk's lattice by a strictly sequential replay (Splitter::progressNode / flushCandidates / insertUnkForm /
// appendNewNode, /root/reference/src/KTrie.cpp:15-43, 897-996, 1040-1137, 1350-1380, 1434-1450): every dictionary candidate, special-character
// run, pattern span and space is one "op" that may insert unknown-form nodes in front of its start position and then append its own node,
// and whether it does depends on what was appended before it.  k_build_lattice replays that on lane 0 (20 k instructions per 40-jamo chunk,
// scalar-issue bound).  Here the ops of a chunk are laid out in the reference's order ("time") and DECIDED TOGETHER:
//
//   * what an op does depends on earlier ops only through four things -- is a position reachable (does a node end there yet), which
//     (end, length) pairs already have a node (hasFormAlready), where the most recently appended node ended (insertUnkForm's extra bridge
//     node), and the z-coda flags of the forms ending at a position;
//   * unknown-form nodes ending at position q are only ever inserted by ops STARTING at q: lane q walks that short list in time order with the
//     position's length mask in registers ("by-start pass");
//   * a node an op appends for itself ends at the op's end position, and ops are in end-position order: lane T handles op T, publishes
//     what it appends into per-position tables with LDS atomics and finds the end of the most recently appended node with a ballot ("by-time pass");
//   * the two passes are iterated until nothing changes.  Every decision depends only on decisions of strictly earlier ops, so the fixpoint
//     is unique and equals the sequential replay (by induction over time); a regular chunk converges in three rounds;
//   * removeUnconnected (KTrie.cpp:240-299) keeps exactly the nodes that end at a position from which the end node is reachable and orders
//     them by end position, insertion order inside one: the final index of a node is a prefix sum over positions plus its rank at its
//     position -- no node list in build order, no renumbering pass.
//
//   * the tail of the kernel writes what the search reads (the candidate records of every node and the per-position program: what
//     k_expand_cands / k_expand_pos wrote in two more launches) while the lattice is still in LDS; W.expanded[chunk] tells those kernels to skip.
//
// Hand-over: a chunk whose ops or nodes outgrow the LDS room of this launch is flagged kLatticeNeedsWide and listed in W.wideList; the same
// kernel runs over that list once more with the wide layout (kLatticeWideBit).  What no layout covers (spans longer than 64 positions, an
// end node that cannot be appended, more rounds than kMaxRounds) is flagged kLatticeNeedsBig and replayed by k_build_lattice_big.
// Memory-bound integer work: no MFMA.
#include <hip/hip_runtime.h>
#include "device_types.hpp"
#include "feature.hpp"
#include "lattice_expand.hpp"

namespace kamd
{
	extern __shared__ __align__(16) uint8_t lSmem[];

	namespace lw
	{
		enum OpFlag : uint16_t
		{
			OF_HASUNK = 1, OF_CONDBU = 2,      // unknown-form attempts in front of the op; the boundary attempt is made when boundary < unkStart (else: boundary < start)
			OF_LIMJ = 4, OF_HASAPP = 8, OF_SEOK = 16, OF_QUAL = 32, OF_ZBITS = 64 | 128, OF_VALID = 256, OF_ZSEL = 512 | 1024, OF_END = 2048,
		};
		constexpr uint32_t kMaxRounds = 16;

		__device__ __forceinline__ uint32_t topBit(uint64_t m) { return 63u - (uint32_t)__builtin_clzll(m); }      // m != 0
		__device__ __forceinline__ uint32_t trimmedLen(const uint16_t* str, uint32_t off, uint32_t len)
		{
			while (len && isSpace(str[off + len - 1])) --len;
			return len;
		}
	}

	// LW_PROFILE build (make lwprof): cycles per phase, one record of 16 words per chunk in WorkView::beacon
#ifdef LW_PROFILE
#define LW_T0() unsigned long long lwT = clock64(); uint32_t lwAcc[13] = {};
#define LW_MARK(k) { const unsigned long long lwN = clock64(); lwAcc[(k)] += (uint32_t)(lwN - lwT); lwT = lwN; if ((k) == 12 && lane == 0 && W.beacon) for (int lwI = 0; lwI < 13; ++lwI) W.beacon[16 * (size_t)chunk + lwI] = lwAcc[lwI]; }
#else
#define LW_T0()
#define LW_MARK(k)
#endif
	// hands the chunk to k_build_lattice_big; outCounters[4 + reason] counts (developer statistics, KAMD_LATTICE_STATS)
#define LW_HAND_OVER(reason) { if (lane == 0) { const bool toWide = !wide && ((reason) == 0 || (reason) == 1 || (reason) == 6); W.nNodes[chunk] = toWide ? kLatticeNeedsWide : kLatticeNeedsBig; \
		if (toWide) W.wideList[atomicAdd(&W.outCounters[3], 1u)] = chunk; atomicAdd(&W.outCounters[(reason) == 6 ? 4 : 4 + (reason)], 1u); } return; }
	__global__ void __launch_bounds__(64) k_lattice_wave(ModelView M, BatchView B, WorkView W, SearchParams P, const uint32_t* chunkList, uint32_t chunkCount, uint32_t ldsBytes, uint32_t matchRatio16, uint32_t expandMode)
	{
		using namespace lw;
		const uint32_t lane = threadIdx.x;
		// the second, `wide` launch: only the chunks the first one could not hold, from the list it left (one block per entry; chunkCount = the most
		// entries this launch takes -- whatever is beyond stays flagged for k_build_lattice_big)
		const bool wide = (matchRatio16 & kLatticeWideBit) != 0;
		if (wide) chunkCount = W.outCounters[3] < chunkCount ? W.outCounters[3] : chunkCount;
		if (blockIdx.x >= chunkCount) return;
		LW_T0()
		const uint32_t chunk = wide ? W.wideList[blockIdx.x] : chunkList[blockIdx.x];
		if (W.results[chunk].status >= 16) return;
		const uint32_t dbgStop = ldsBytes >> 24; ldsBytes &= 0xFFFFFFu;      // EXPERIMENT (KAMD_LATTICE_STOP): leave after phase 1 .. 6, results void
#define LW_STOP(k) if (dbgStop == (k)) { if (lane == 0) W.results[chunk].status = CS_NO_LATTICE; return; }
		const uint32_t cOff = B.charOff[chunk], n = B.charOff[chunk + 1] - cOff;
		const uint32_t nNs = W.nNs[chunk];
		const uint32_t nBase = W.nodeBase[chunk], cap = W.nodeBase[chunk + 1] - nBase;
		const uint32_t mBase = W.matchBase[chunk], mCap = W.matchBase[chunk + 1] - mBase;
		if (wide && W.nNodes[chunk] != kLatticeNeedsWide) return;
		const LwLds lay = latticeWaveLayout(n, cap, mCap, matchRatio16);
		if (lay.total > ldsBytes) { if (wide && lane == 0) W.nNodes[chunk] = kLatticeNeedsBig; return; }      // the first launch leaves it to the wide one, that one to k_build_lattice_big
		if (nNs > 0xFFF0 || cap > 0xFFF0 || cap < 4) { if (lane == 0) W.results[chunk].status = CS_ERR_TOO_LONG; return; }
		uint8_t* const lS = lSmem;

		uint16_t* str = reinterpret_cast<uint16_t*>(lS + lay.str);
		uint8_t* cls = lS + lay.cls; uint8_t* script = lS + lay.script; uint8_t* cflag = lS + lay.cflag;
		uint16_t* nsToPos = reinterpret_cast<uint16_t*>(lS + lay.nsToPos); uint16_t* posToNs = reinterpret_cast<uint16_t*>(lS + lay.posToNs);
		uint64_t* mask = reinterpret_cast<uint64_t*>(lS + lay.mask); uint32_t* moff = reinterpret_cast<uint32_t*>(lS + lay.moff);
		uint32_t* mforms = reinterpret_cast<uint32_t*>(lS + lay.mforms); uint8_t* mse = lS + lay.mse; uint32_t* mfc = reinterpret_cast<uint32_t*>(lS + lay.mfc);
		uint32_t* ctlBU = reinterpret_cast<uint32_t*>(lS + lay.ctlBU); uint16_t* ctlT = reinterpret_cast<uint16_t*>(lS + lay.ctlT); uint16_t* ctlRs = reinterpret_cast<uint16_t*>(lS + lay.ctlRs);
		uint32_t* opNE = reinterpret_cast<uint32_t*>(lS + lay.opNE); uint32_t* opBU = reinterpret_cast<uint32_t*>(lS + lay.opBU);
		uint16_t* opFl = reinterpret_cast<uint16_t*>(lS + lay.opFl); uint16_t* opSrc = reinterpret_cast<uint16_t*>(lS + lay.opSrc);
		uint16_t* decS = reinterpret_cast<uint16_t*>(lS + lay.decS); uint32_t* decT = reinterpret_cast<uint32_t*>(lS + lay.decT);
		uint16_t* grpList = reinterpret_cast<uint16_t*>(lS + lay.grpList);
		uint32_t* miscForm = reinterpret_cast<uint32_t*>(lS + lay.miscForm); uint32_t* miscU = reinterpret_cast<uint32_t*>(lS + lay.miscU); uint32_t* miscFc = reinterpret_cast<uint32_t*>(lS + lay.miscFc);
		uint32_t* grpOff = reinterpret_cast<uint32_t*>(lS + lay.grpOff); uint32_t* posA = reinterpret_cast<uint32_t*>(lS + lay.posA); uint32_t* posZ = reinterpret_cast<uint32_t*>(lS + lay.posZ);
		uint64_t* fd = reinterpret_cast<uint64_t*>(lS + lay.fd); uint32_t* fdw = reinterpret_cast<uint32_t*>(lS + lay.fd);
		uint16_t* unkMinT = reinterpret_cast<uint16_t*>(lS + lay.unkMinT); uint16_t* cntU = reinterpret_cast<uint16_t*>(lS + lay.cntU); uint32_t* cntA = reinterpret_cast<uint32_t*>(lS + lay.cntA);
		uint64_t* succ = reinterpret_cast<uint64_t*>(lS + lay.succ); uint32_t* succw = reinterpret_cast<uint32_t*>(lS + lay.succ);
		uint16_t* base = reinterpret_cast<uint16_t*>(lS + lay.base); uint32_t* firstU = reinterpret_cast<uint32_t*>(lS + lay.firstU);
		uint32_t* cc = reinterpret_cast<uint32_t*>(lS + lay.cc); uint32_t* recOff = reinterpret_cast<uint32_t*>(lS + lay.recOff); uint32_t* scal = reinterpret_cast<uint32_t*>(lS + lay.scal);
		const uint32_t nPosAll = nNs + 2;      // positions 0 .. nNs, and nNs + 1 for the end node

		// ---- 0. stage the chunk (all lanes, coalesced) and digest the packed matches, one per lane (as k_build_lattice) ----
		uint32_t mTot;
		{
			const uint16_t* gstr = B.chars + cOff; const uint8_t* gcls = B.cls + cOff; const uint8_t* gscript = B.script + cOff; const uint8_t* gcflag = W.cflag + cOff;
			const uint16_t* gn2p = W.nsToPos + cOff + chunk; const uint16_t* gp2n = W.posToNs + cOff + chunk;
			const uint64_t* gmask = W.matchMask + cOff + chunk; const uint32_t* gmoff = W.matchOff + cOff + chunk;
			for (uint32_t i = lane; i < n; i += 64) { str[i] = gstr[i]; cls[i] = gcls[i]; script[i] = gscript[i]; cflag[i] = gcflag[i]; }
			for (uint32_t i = lane; i <= n; i += 64) { posToNs[i] = gp2n[i]; if (i < nNs) nsToPos[i] = gn2p[i]; }
			for (uint32_t i = lane; i <= nNs; i += 64) { mask[i] = gmask[i]; moff[i] = gmoff[i]; }
			for (uint32_t i = lane; i < nPosAll; i += 64) { ctlT[i] = 0; ctlRs[i] = 0; ctlBU[i] = 0; unkMinT[i] = 0xFFFF; cntU[i] = 0; grpOff[i] = 0; posZ[i] = 0; base[i] = 0xFFFF; }
			if (lane == 0) { grpOff[nPosAll] = 0; scal[0] = 0; scal[1] = 0; }
			mTot = gmoff[nNs] + __popcll(gmask[nNs]);
			const uint32_t* gforms = W.matchForm + mBase;
			if (mTot > lay.matchCap) LW_HAND_OVER(0)
			waveSync();
			// end position of every packed match (its list is contiguous), and the number of gaps (spaces skipped by the non-space index) up to every position
			uint16_t* endOf = grpList; uint16_t* gapPre = decS;      // (both arrays are free until the ops exist)
			for (uint32_t e = lane; e <= nNs; e += 64) { const uint32_t m0 = moff[e], c = (uint32_t)__popcll(mask[e]); for (uint32_t i = 0; i < c; ++i) endOf[m0 + i] = (uint16_t)e; }
			uint32_t run = 0;
			for (uint32_t b0 = 0; b0 < nNs; b0 += 64)
			{
				const uint32_t i = b0 + lane;
				const uint32_t g = (i > 0 && i < nNs && nsToPos[i] - nsToPos[i - 1] > 1) ? 1u : 0u;
				uint32_t incl = g;
				for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d); if (lane >= d) incl += v; }
				if (i < nNs) gapPre[i] = (uint16_t)(run + incl);
				run += __shfl(incl, 63);
			}
		}
		waveSync();
		LW_MARK(0)
		LW_STOP(1)

		// ---- 1. the character-type state machine of progressNode (KTrie.cpp:1040-1137, 1350-1380): a function of the text alone.  It decides the
		// boundary / unknown-form start every op is made under and emits the ops that are not dictionary candidates, each at its place in time.
		// Text without patterns, surrogate pairs and emoji modifiers (the state then depends on the previous unit only): one unit per lane, the three
		// running values -- start of the special run, unknown-form start, last boundary -- are "value at the most recent event", found with a ballot.
		bool plain = B.patOff[chunk] == B.patOff[chunk + 1];
		for (uint32_t b0 = 0; b0 < n; b0 += 64)
		{
			const uint32_t j = b0 + lane;
			bool odd = false;
			if (j < n) { const uint16_t c = str[j]; odd = isHighSurrogate(c) || isLowSurrogate(c) || c == 0x200d || script[j] == 98; }
			if (__ballot(odd)) plain = false;
		}
		if (plain)
		{
			uint32_t tC = T_UNKNOWN, sC = 0, ssC = 0, bC = 0, uC = 0, Tbase = 1, miscBase = 0;
			bool over = false;
			auto lastEvent = [&](uint64_t ev, uint32_t val, uint32_t carry) -> uint32_t      // the value of the most recent event at or before this lane
			{
				const uint64_t m = ev & (((1ull << lane) - 1ull) | (1ull << lane));
				const uint32_t got = __shfl(val, m ? (int)topBit(m) : (int)lane);
				return m ? got : carry;
			};
			for (uint32_t b0 = 0; b0 < n; b0 += 64)
			{
				const uint32_t j = b0 + lane; const bool act = j < n;
				const uint32_t t = act ? (uint32_t)(cls[j] & 0x3F) : (uint32_t)T_UNKNOWN, sc = act ? (uint32_t)script[j] : 0u;
				const uint32_t tUp = __shfl_up(t, 1), sUp = __shfl_up(sc, 1);
				const uint32_t tp = lane ? tUp : tC, sp = lane ? sUp : sC;
				const bool symL = tp == T_SL || tp == T_SH || tp == T_SW, symC = t == T_SL || t == T_SH || t == T_SW;
				const bool disc = act && (((symL && symC) ? (sp != sc) : (tp != t)) || tp == T_SSO || tp == T_SSC);
				const bool spec = disc && tp != T_MAX && tp != T_UNKNOWN && tp != T_SS;
				const bool sjp = T_SF <= tp && tp <= T_SW;
				const bool isM = act && !disc && t == T_MAX, isS = act && t == T_UNKNOWN;
				const uint32_t pj = act ? (uint32_t)posToNs[j] : 0u, pj1 = act ? (uint32_t)posToNs[j + 1] : 0u;
				const uint32_t ss = lastEvent(__ballot(disc || isS), isS ? pj1 : pj, ssC);
				const uint32_t ssUp = __shfl_up(ss, 1); const uint32_t ssPrev = lane ? ssUp : ssC;
				const uint32_t bb = lastEvent(__ballot(isS || (disc && sjp)), isS ? pj1 : pj, bC);
				const uint32_t bUp = __shfl_up(bb, 1); const uint32_t bPrev = lane ? bUp : bC;
				const uint32_t uu = lastEvent(__ballot(isS || disc || isM), isS ? pj1 : ssPrev, uC);
				const uint32_t uUp = __shfl_up(uu, 1); const uint32_t uPrev = lane ? uUp : uC;
				// the z-coda / saisiot shortcut (KTrie.cpp:1126-1135), text side; the form's own facts come from the dictionary
				uint32_t zsel = 0, zform = 0, zfl = 0;
				const bool flush = act && !isS;
				if (flush && pj < nNs)
				{
					const uint16_t ch = str[j];
					if ((P.match & M_Z_CODA) && isHangulCoda(ch) && (j + 1 >= n || !isHangulSyllable(str[j + 1]))) { zsel = 1; zform = kDefaultTagSize + (ch - 0x11A8) - 1; }
					else if ((P.match & (M_SPLIT_SAISIOT | M_MERGE_SAISIOT)) && ch == 0x11BA && j + 1 < n && isHangulSyllable(str[j + 1])) { zsel = 2; zform = kDefaultTagSize + (0x11BA - 0x11A8) - 1; }
					if (zsel)
					{
						const FormRec f = M.forms[zform];
						const uint32_t flen = f.len - f.numSpaces;
						if (flen > pj1) zsel = 0;
						else if (flen != 1) { over = true; zsel = 0; }
						else
						{
							const bool hj = (f.flags & FF_HAS_JCLASS) || (f.flags & FF_IS_STAG);
							zfl = OF_HASAPP | OF_SEOK | OF_VALID | (zsel << 9) | ((uint32_t)(f.flags & 3) << 6);
							if (!(f.flags & FF_FIRST_IS_CODA)) zfl |= OF_HASUNK | (hj ? OF_LIMJ : 0);
							if (f.flags & FF_HAS_ANY_FULL) zfl |= OF_QUAL;
						}
					}
				}
				const uint32_t nMiscMine = (spec ? 1u : 0u) + (isS ? 1u : 0u) + (zsel ? 1u : 0u);
				const uint32_t nOpsMine = nMiscMine + (flush ? (uint32_t)__popcll(mask[pj1]) : 0u);
				uint32_t inclT = nOpsMine, inclM = nMiscMine;
				for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(inclT, d), w = __shfl_up(inclM, d); if (lane >= d) { inclT += v; inclM += w; } }
				uint32_t T = Tbase + inclT - nOpsMine, mi = miscBase + inclM - nMiscMine;
				const uint32_t totT = __shfl(inclT, 63), totM = __shfl(inclM, 63);
				if (Tbase + totT >= lay.opCap || miscBase + totM >= lay.miscCap) over = true;      // (uniform)
				if (!over)
				{
					auto put = [&](uint32_t nb, uint32_t e, uint32_t bnd, uint32_t unk, uint32_t fl, uint32_t form, uint32_t uOff, uint32_t uLen)
					{
						opNE[T] = nb | (e << 16); opBU[T] = bnd | (unk << 16); opFl[T] = (uint16_t)fl; opSrc[T] = (uint16_t)(0x8000u | mi);
						miscForm[mi] = form; miscU[mi] = uOff | (uLen << 16);
						++T; ++mi;
					};
					if (spec)
					{
						const uint32_t o = nsToPos[ssPrev], l = trimmedLen(str, o, j - o);
						put(ssPrev, pj, bPrev, uPrev, OF_HASUNK | OF_CONDBU | (sjp ? OF_LIMJ : 0) | OF_HASAPP | OF_SEOK | OF_VALID, tp - 1u, o, l);
					}
					// (the values a space's own attempts are made under: after the type change of this unit, before the space resets them)
					if (isS) put(pj1, pj1, (disc && sjp) ? pj : bPrev, (disc || isM) ? ssPrev : uPrev, OF_HASUNK | OF_CONDBU | OF_LIMJ | OF_VALID, NOFORM, 0, 0);
					if (flush)
					{
						if (zsel) put(pj1 - 1, pj1, bb, uu, zfl, zform, 0, 0);
						ctlBU[pj1] = bb | (uu << 16); ctlT[pj1] = (uint16_t)T; ctlRs[pj1] = 0;
					}
				}
				Tbase += totT; miscBase += totM;
				tC = __shfl(t, 63); sC = __shfl(sc, 63); ssC = __shfl(ss, 63); bC = __shfl(bb, 63); uC = __shfl(uu, 63);
				if (b0 + 64 > n) { const uint32_t last = (n - 1) & 63u; tC = __shfl(t, last); sC = __shfl(sc, last); }      // (the running values beyond the text repeat the last unit's: no events there)
			}
			over = __ballot(over) != 0;
			if (lane == 0)
			{
				// after the last unit (KTrie.cpp:1350-1380, 1434-1450): the special run still open, the whole-tail unknown form, the end node
				uint32_t T = Tbase, mi = miscBase, boundary = bC, unkStart = uC; const uint32_t specialStart = ssC, lastType = tC;
				auto put = [&](uint32_t nb, uint32_t e, uint32_t fl, uint32_t form, uint32_t uOff, uint32_t uLen)
				{
					if (T >= lay.opCap || mi >= lay.miscCap) { over = true; return; }
					opNE[T] = nb | (e << 16); opBU[T] = boundary | (unkStart << 16); opFl[T] = (uint16_t)fl; opSrc[T] = (uint16_t)(0x8000u | mi);
					miscForm[mi] = form; miscU[mi] = uOff | (uLen << 16);
					++T; ++mi;
				};
				if (lastType != T_MAX && lastType != T_UNKNOWN && lastType != T_SS)
				{
					const bool sj = T_SF <= lastType && lastType <= T_SW;
					const uint32_t o = nsToPos[specialStart], l = trimmedLen(str, o, n - o);
					put(specialStart, posToNs[n], OF_HASUNK | OF_CONDBU | (sj ? OF_LIMJ : 0) | OF_HASAPP | OF_SEOK | OF_VALID, lastType - 1u, o, l);
					unkStart = specialStart;
					if (sj) boundary = posToNs[n];
				}
				if (nNs && n == (uint32_t)nsToPos[nNs - 1] + 1) put(posToNs[n], posToNs[n], OF_HASUNK | OF_CONDBU | OF_LIMJ | OF_VALID, NOFORM, 0, 0);
				put(nNs, nNs + 1, OF_HASAPP | OF_SEOK | OF_VALID | OF_END, NOFORM, 0, 0);
				scal[0] = over ? 1u : 0u; scal[1] = T - 1;
			}
		}
		// any other text: the same machine unit by unit on one lane (~40 instructions per unit)
		else if (lane == 0)
		{
			const DevPattern* pat = B.patterns + B.patOff[chunk];
			const DevPattern* patEnd = B.patterns + B.patOff[chunk + 1];
			uint8_t lastType = T_UNKNOWN, lastScript = 0;
			uint32_t specialStart = 0, unkStart = 0, boundary = 0, resetNs = 0;
			uint32_t T = 1, nMisc = 0; bool over = false;
			const uint8_t scriptVS = 98;
			auto emit = [&](uint32_t nb, uint32_t e, uint32_t fl, uint32_t form, uint32_t uOff, uint32_t uLen)
			{
				if (T >= lay.opCap || nMisc >= lay.miscCap) { over = true; return; }
				opNE[T] = nb | (e << 16); opBU[T] = boundary | (unkStart << 16); opFl[T] = (uint16_t)fl; opSrc[T] = (uint16_t)(0x8000u | nMisc);
				miscForm[nMisc] = form; miscU[nMisc] = uOff | (uLen << 16);
				++nMisc; ++T;
			};
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
						const uint32_t o = nsToPos[specialStart], l = trimmedLen(str, o, j - o);
						emit(specialStart, posToNs[j], OF_HASUNK | OF_CONDBU | (sj ? OF_LIMJ : 0) | OF_HASAPP | OF_SEOK | OF_VALID, lastType - 1u, o, l);
					}
					unkStart = specialStart;
					specialStart = posToNs[j];
					if (T_SF <= lastType && lastType <= T_SW) boundary = specialStart;
				}
				else if (type == T_MAX) unkStart = specialStart;
				lastType = curT; lastScript = sct;

				uint32_t zsel = 0, zform = 0;
				if (!pair)
				{
					if (type == T_UNKNOWN)
					{
						emit(posToNs[j + 1], posToNs[j + 1], OF_HASUNK | OF_CONDBU | OF_LIMJ | OF_VALID, NOFORM, 0, 0);
						boundary = specialStart = unkStart = posToNs[j + 1];
						continue;
					}
					// a space-class unit promoted to a symbol was fed to the trie by the reference and reset the walk
					if (overridden && (cflag[j] & 1)) resetNs = posToNs[j] + 1u;
					// z-coda / saisiot shortcut (KTrie.cpp:1126-1135): the text side of the condition; the other side -- a form ending here allows it -- is the op's dynamic validity
					if (posToNs[j] < nNs)
					{
						if ((P.match & M_Z_CODA) && isHangulCoda(ch) && (j + 1 >= n || !isHangulSyllable(str[j + 1]))) { zsel = 1; zform = kDefaultTagSize + (ch - 0x11A8) - 1; }
						else if ((P.match & (M_SPLIT_SAISIOT | M_MERGE_SAISIOT)) && ch == 0x11BA && j + 1 < n && isHangulSyllable(str[j + 1])) { zsel = 2; zform = kDefaultTagSize + (0x11BA - 0x11A8) - 1; }
					}
				}
				if (pat != patEnd)
				{
					const uint32_t curEnd = j + (pair ? 2 : 1);
					while (pat != patEnd && pat->end == curEnd)
					{
						const uint32_t ms = pat->end - pat->length;
						const bool wj = T_W_URL <= pat->tag && pat->tag <= T_W_EMOJI;
						emit(posToNs[ms], posToNs[pat->end], OF_HASUNK | OF_CONDBU | (wj ? OF_LIMJ : 0) | OF_HASAPP | OF_SEOK | OF_VALID, pat->tag - 1u, ms, pat->length);
						++pat;
					}
				}
				if (pair) { ++j; continue; }
				const uint32_t endNs = posToNs[j + 1];
				if (zsel)
				{
					const FormRec f = M.forms[zform];
					const uint32_t flen = f.len - f.numSpaces;
					if (flen == 1 && flen <= endNs)
					{
						const bool hj = (f.flags & FF_HAS_JCLASS) || (f.flags & FF_IS_STAG);
						uint32_t fl = OF_HASAPP | OF_SEOK | OF_VALID | (zsel << 9) | ((uint32_t)(f.flags & 3) << 6);
						if (!(f.flags & FF_FIRST_IS_CODA)) fl |= OF_HASUNK | (hj ? OF_LIMJ : 0);
						if (f.flags & FF_HAS_ANY_FULL) fl |= OF_QUAL;
						emit(endNs - 1, endNs, fl, zform, 0, 0);
					}
					else if (flen <= endNs) over = true;      // (a z form of more than one unit: left to the replay)
				}
				ctlBU[endNs] = boundary | (unkStart << 16); ctlT[endNs] = (uint16_t)T; ctlRs[endNs] = (uint16_t)resetNs;
				T += (uint32_t)__popcll(mask[endNs]);
			}
			if (lastType != T_MAX && lastType != T_UNKNOWN && lastType != T_SS)
			{
				const bool sj = T_SF <= lastType && lastType <= T_SW;
				const uint32_t o = nsToPos[specialStart], l = trimmedLen(str, o, n - o);
				emit(specialStart, posToNs[n], OF_HASUNK | OF_CONDBU | (sj ? OF_LIMJ : 0) | OF_HASAPP | OF_SEOK | OF_VALID, lastType - 1u, o, l);
				unkStart = specialStart;
				if (sj) boundary = posToNs[n];
			}
			if (nNs && n == (uint32_t)nsToPos[nNs - 1] + 1) emit(posToNs[n], posToNs[n], OF_HASUNK | OF_CONDBU | OF_LIMJ | OF_VALID, NOFORM, 0, 0);
			emit(nNs, nNs + 1, OF_HASAPP | OF_SEOK | OF_VALID | OF_END, NOFORM, 0, 0);
			if (T > lay.opCap) over = true;
			scal[0] = over ? 1u : 0u; scal[1] = T - 1;
		}
		waveSync();
		if (scal[0]) LW_HAND_OVER(1)
		const uint32_t K = scal[1];      // ops 1 .. K
		LW_MARK(1)
		LW_STOP(2)

		// ---- 2. the dictionary candidates as ops, one per lane: time = the time of their end position's first candidate + rank in its list;
		// start position, space errors of the span (countSpaceErrors, KTrie.cpp:316-328) and form flags decide everything static about the op ----
		uint32_t hazardEarly = 0;
		{
			const uint32_t* gforms = W.matchForm + mBase;
			const uint16_t* endOf = grpList; const uint16_t* gapPre = decS;
			for (uint32_t k = lane; k < mTot; k += 64)
			{
				const uint32_t fi = gforms[k]; const FormRec f = M.forms[fi];
				const uint32_t e = endOf[k], flen = f.len - f.numSpaces;
				const uint32_t T = (uint32_t)ctlT[e] + (k - moff[e]);
				uint32_t nb = 0, se = 0; bool valid = false;
				if (flen <= e)
				{
					valid = true; nb = e - flen;
					if (!f.numSpaces) { if (flen > 1) se = (uint32_t)gapPre[e - 1] - (uint32_t)gapPre[nb]; }      // every gap inside the span is an error
					else
					{
						const uint16_t* fs = M.formChars + f.charOff;
						uint32_t off = 0;
						for (uint32_t i = 1; i < flen; ++i)
						{
							const bool hasSpace = nsToPos[nb + i] - nsToPos[nb + i - 1] > 1;
							const uint16_t fc = (i + off < f.len) ? fs[i + off] : 0;
							if (hasSpace && fc != u' ') ++se;
							if (fc == u' ') ++off;
						}
					}
				}
				const uint8_t fl = f.flags;
				const bool hj = (fl & FF_HAS_JCLASS) || (fl & FF_IS_STAG);
				uint32_t of = OF_HASAPP | ((uint32_t)(fl & 3) << 6);
				if (!(fl & FF_FIRST_IS_CODA)) of |= OF_HASUNK | (hj ? OF_LIMJ : 0);
				if (se <= P.spaceTol) of |= OF_SEOK;
				if (fl & FF_HAS_ANY_FULL) of |= OF_QUAL;
				if (valid && nb >= ctlRs[e]) of |= OF_VALID;
				opNE[T] = nb | (e << 16); opBU[T] = ctlBU[e]; opFl[T] = (uint16_t)of; opSrc[T] = (uint16_t)k;
				mforms[k] = fi; mse[k] = (uint8_t)(se > 255 ? 255 : se);
				mfc[k] = (uint32_t)(f.candCnt & 0x7FFFu) | ((f.flags2 & FF2_ALL_PARTIAL) ? 0x8000u : 0u) | ((uint32_t)f.flags << 16) | ((uint32_t)f.len << 24);
				if (f.candCnt > 0x7FFFu) hazardEarly = 1;
				if ((of & OF_VALID) && (of & OF_SEOK) && (fl & 3)) atomicOr(&posZ[e], (uint32_t)(fl & 3));      // first guess of the z-coda flags: every candidate appended
			}
		}
		waveSync();

		if (__ballot(hazardEarly != 0)) LW_HAND_OVER(0)
		LW_MARK(2)
		// ---- 3. ops grouped by START position (counting sort; the few ops of a position then ordered by time) ----
		for (uint32_t T = 1 + lane; T <= K; T += 64) atomicAdd(&grpOff[(opNE[T] & 0xFFFF) + 1], 1u);
		waveSync();
		{
			uint32_t run = 0;
			for (uint32_t b0 = 0; b0 <= nPosAll; b0 += 64)
			{
				const uint32_t q = b0 + lane;
				const uint32_t c = q <= nPosAll ? grpOff[q] : 0u;
				uint32_t incl = c;
				for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d); if (lane >= d) incl += v; }
				if (q <= nPosAll) grpOff[q] = run + incl;      // grpOff[q + 1] held the count of q: inclusive sum = first slot of position q
				if (q < nPosAll) posA[q] = 0;                   // (scatter cursors)
				run += __shfl(incl, 63);
			}
		}
		waveSync();
		for (uint32_t T = 1 + lane; T <= K; T += 64)
		{
			const uint32_t q = opNE[T] & 0xFFFF;
			grpList[grpOff[q] + atomicAdd(&posA[q], 1u)] = (uint16_t)T;
			const uint32_t fl0 = opFl[T], zs0 = (fl0 >> 9) & 3u;
			decS[T] = (uint16_t)((1u << 4) | (((fl0 & OF_VALID) && (!zs0 || (posZ[q] & zs0))) ? (1u << 5) : 0u));      // first guess: reachable; a z shortcut is valid if any candidate ending at its start allows it
			decT[T] = 0xFFFFu << 16;
		}
		waveSync();
		for (uint32_t q = lane; q < nPosAll; q += 64)
		{
			const uint32_t g0 = grpOff[q], g1 = grpOff[q + 1];
			for (uint32_t i = g0 + 1; i < g1; ++i)
			{
				const uint16_t v = grpList[i]; uint32_t j = i;
				while (j > g0 && grpList[j - 1] > v) { grpList[j] = grpList[j - 1]; --j; }
				grpList[j] = v;
			}
		}
		waveSync();

		LW_MARK(3)
		LW_STOP(3)
		// ---- 4. the fixpoint ----
		auto reachAt = [&](uint32_t x, uint32_t T) -> bool { return posA[x] < T || (uint32_t)unkMinT[x] < T; };      // a node ends at x before time T
		uint32_t hazard = 0, nRounds = 0;
		for (uint32_t round = 0;; ++round)
		{
			if (round >= kMaxRounds) { hazard = 1; break; }
			nRounds = round + 1;
			bool chg = false;
			for (uint32_t q = lane; q < nPosAll; q += 64) { posA[q] = q ? 0xFFFFFFFFu : 0u; posZ[q] = 0; fd[q] = 0; }      // (the start node ends at position 0, at time 0)
			waveSync();
			LW_MARK(4)
			// by-time pass: lane = op.  Its own node (needs its start reachable), the tables the other pass reads, the end of the most recently appended node
			{
				uint32_t carry = 0;
				for (uint32_t b0 = 1; b0 <= K; b0 += 64)
				{
					const uint32_t T = b0 + lane; const bool act = T <= K;
					uint32_t fl = 0, nb = 0, e = 0, s = 0;
					if (act) { fl = opFl[T]; const uint32_t ne = opNE[T]; nb = ne & 0xFFFF; e = ne >> 16; s = decS[T]; }
					const bool app = act && (fl & OF_HASAPP) && (fl & OF_SEOK) && ((s >> 5) & 1) && ((s >> 4) & 1);
					const bool any = app || (s & 15u);
					const uint32_t lastAfter = app ? e : nb;
					const uint64_t m = __ballot(any);
					const uint64_t lower = m & ((1ull << lane) - 1ull);
					const uint32_t got = __shfl(lastAfter, lower ? (int)topBit(lower) : (int)lane);
					const uint32_t lb = lower ? got : carry;
					const uint32_t top = __shfl(lastAfter, m ? (int)topBit(m) : 0);
					if (m) carry = top;
					if (act)
					{
						const uint32_t nd = (app ? 1u : 0u) | (lb << 16);
						decT[T] = nd;
						if (app)
						{
							atomicMin(&posA[e], T);
							const uint32_t len = e - nb;
							if ((fl & OF_QUAL) && len >= 1 && len <= 64) atomicOr(&fdw[2 * e + ((len - 1) >> 5)], 1u << ((len - 1) & 31));
							const uint32_t zb = (fl >> 6) & 3u;
							if (zb) atomicOr(&posZ[e], zb);
						}
					}
				}
			}
			waveSync();
			LW_MARK(5)
			// by-start pass: lane = op, in (start position, time) order.  The unknown-form nodes in front of it (insertUnkForm, KTrie.cpp:921-953).  What the
			// ops of one start position share -- the position's length mask as it grows, the count of its unknown-form nodes -- are segmented scans over
			// adjacent lanes: an attempt on a span is made by the first op that is allowed to, whichever op that is, so the OR of what every earlier op WOULD
			// insert on the dictionary's mask alone equals the OR of what they did insert; only the bridge nodes (made when the last node ended before
			// the position) depend on the order, and those are taken from the previous round
			{
				uint32_t carryQ = 0xFFFFFFFFu, carryCnt = 0; uint64_t carryF = 0;
				for (uint32_t b0 = 0; b0 < K; b0 += 64)
				{
					const uint32_t g = b0 + lane; const bool act = g < K;
					uint32_t T = 0, fl = 0, q = 0xFFFFFFFEu, b = 0, u = 0, lastB = 0xFFFF, prev = 0;
					if (act) { T = grpList[g]; fl = opFl[T]; const uint32_t bu = opBU[T]; b = bu & 0xFFFF; u = bu >> 16; q = opNE[T] & 0xFFFF; lastB = decT[T] >> 16; prev = decS[T]; }
					uint64_t Fdict = 0; uint32_t rmT = 0xFFFFFFFFu, zq = 0;
					if (act) { Fdict = fd[q]; rmT = posA[q]; zq = posZ[q]; }
					bool valid = (fl & OF_VALID) != 0;
					const uint32_t zsel = (fl >> 9) & 3u;
					if (zsel) valid = valid && (zq & zsel);
					const bool tries = act && valid && (fl & OF_HASUNK);
					const uint32_t lim = (fl & OF_LIMJ) ? P.maxUnkJ : P.maxUnk;
					const bool doB = (fl & OF_CONDBU) ? (b < u) : (b < q);
					auto lenKeyOf = [&](uint32_t s0) -> uint32_t { const uint32_t o = nsToPos[s0], len = nsToPos[q - 1] + 1u - o; return plain ? len : trimmedLen(str, o, len); };
					// the bridge start: the end of the last node, not a lone coda
					uint32_t lp = lastB;
					if (tries && lastB < q && lp && isHangulCoda(str[nsToPos[lp]])) --lp;
					// what this op contributes to its position's mask, whatever the earlier ops of the position did
					uint64_t contrib = 0;
					if (tries)
					{
						for (uint32_t a = doB ? 0u : 1u; a < 2; ++a)
						{
							const uint32_t s0 = a ? u : b;
							if (s0 >= q) continue;
							const uint32_t L = q - s0;
							if (L > 64) { hazard = 1; continue; }
							if (!((Fdict >> (L - 1)) & 1) && L <= lim && reachAt(s0, T)) { const uint32_t l = lenKeyOf(s0); if (l > 64) hazard = 1; else if (l) contrib |= 1ull << (l - 1); }
						}
						if ((prev & 5u) && lastB < q) { const uint32_t l = lenKeyOf(lp); if (l > 64) hazard = 1; else if (l) contrib |= 1ull << (l - 1); }
					}
					// segmented exclusive OR over the lanes of the same start position
					const uint32_t qUp = __shfl_up(q, 1);
					const bool segStart = lane == 0 ? (q != carryQ) : (q != qUp);
					const uint64_t sb = __ballot(segStart);
					const uint64_t mineSeg = sb & (((1ull << lane) - 1ull) | (1ull << lane));
					const uint32_t start = mineSeg ? topBit(mineSeg) : 0u;      // first lane of this lane's segment (0: it began in an earlier block)
					uint32_t lo = (uint32_t)contrib, hi = (uint32_t)(contrib >> 32);
					for (uint32_t d = 1; d < 64; d <<= 1)
					{
						const uint32_t vl = __shfl_up(lo, d), vh = __shfl_up(hi, d);
						if (lane >= d && lane - d >= start) { lo |= vl; hi |= vh; }
					}
					const uint32_t exLo = __shfl_up(lo, 1), exHi = __shfl_up(hi, 1);
					uint64_t F = Fdict;
					if (lane > start) F |= ((uint64_t)exHi << 32) | exLo;
					if (!mineSeg) F |= carryF;
					// the op itself, exactly as the reference makes its attempts, on the mask as it stands when its turn comes
					uint32_t um = 0, own = 0;
					if (tries)
					{
						uint32_t le = lastB;
						for (uint32_t a = doB ? 0u : 1u; a < 2; ++a)
						{
							const uint32_t s0 = a ? u : b;
							if (s0 >= q) continue;
							const uint32_t L = q - s0;
							if (L > 64) continue;
							if ((F >> (L - 1)) & 1) continue;
							if (le < q && lp != s0)
							{
								const uint32_t L2 = q - lp;
								if (L2 > 64) hazard = 1;
								else if (!((F >> (L2 - 1)) & 1) && reachAt(lp, T))
								{
									const uint32_t l = lenKeyOf(lp);
									if (l > 64) hazard = 1; else if (l) F |= 1ull << (l - 1);
									um |= 1u << (2 * a); ++own; le = q;
								}
							}
							if (L <= lim && reachAt(s0, T))
							{
								const uint32_t l = lenKeyOf(s0);
								if (l > 64) hazard = 1; else if (l) F |= 1ull << (l - 1);
								um |= 2u << (2 * a); ++own; le = q;
							}
						}
					}
					// rank of its nodes among the position's unknown-form nodes: segmented exclusive sum
					uint32_t incl = own;
					for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d); if (lane >= d && lane - d >= start) incl += v; }
					uint32_t rank = incl - own;
					if (!mineSeg) rank += carryCnt;
					if (act)
					{
						if (rank > 250) hazard = 1;
						const bool rq = rmT < T || rank + own > 0;
						const uint32_t nd = um | (rq ? 16u : 0u) | (valid ? 32u : 0u) | ((rank & 0xFF) << 8);
#ifdef LW_DEBUG_ROUNDS
						if (prev != nd && round >= 1) printf("round %u q %u T %u fl %x old %x new %x lastB %u rmT %u nb|e %x\n", round, q, T, fl, prev, nd, lastB, rmT, opNE[T]);
#endif
						if (prev != nd) { chg = true; decS[T] = (uint16_t)nd; }
						if (own && rank == 0) base[q] = (uint16_t)T;      // the position's first unknown-form node (published after the pass: other lanes still read this round's unkMinT)
					}
					// the last lane of a segment knows the position's totals
					bool segEnd = act && lane < 63 && ((sb >> (lane + 1)) & 1);      // (the lanes beyond the last op form a segment of their own)
					if (act && lane == 63) segEnd = g + 1 >= K || (opNE[grpList[g + 1]] & 0xFFFFu) != q;
					if (segEnd) { cntU[q] = (uint16_t)(rank + own); if (rank + own == 0) base[q] = 0xFFFF; }
					// carry into the next block: the segment of lane 63
					const uint64_t fullF = ((uint64_t)hi << 32) | lo;
					const uint32_t cLo = __shfl((uint32_t)fullF, 63), cHi = __shfl((uint32_t)(fullF >> 32), 63), cCnt = __shfl(rank + own, 63), cQ = __shfl(q, 63);
					const bool contSeg = sb == 0;      // no segment began in this block: lane 63 still continues the carried one
					carryF = (contSeg ? carryF : 0ull) | ((uint64_t)cHi << 32) | cLo;
					carryCnt = cCnt;      // (rank + own already includes the carried count when the segment continued)
					carryQ = cQ;
				}
			}
			waveSync();
			LW_MARK(6)
#ifdef LW_DEBUG_ROUNDS
			for (uint32_t q = lane; q < nPosAll; q += 64) if (unkMinT[q] != base[q] && round >= 1) printf("round %u q %u unkMinT %u -> %u\n", round, q, (unsigned)unkMinT[q], (unsigned)base[q]);
#endif
			for (uint32_t q = lane; q < nPosAll; q += 64) if (unkMinT[q] != base[q]) { chg = true; unkMinT[q] = base[q]; }
			waveSync();
			LW_MARK(7)
			if (__ballot(hazard != 0)) { hazard = 1; break; }
			if (!__ballot(chg)) break;
		}
		if (__ballot(hazard != 0)) LW_HAND_OVER(2)
		// the end node must exist (else the reference renames whatever node came last: left to the replay)
		if (!(decT[K] & 1u)) LW_HAND_OVER(3)
		if (lane == 0)
		{
			// what the engine sizes the next batch's LDS arrays by: the most matches per text unit seen (an atomic only when it grows)
			const uint32_t r100 = (mTot * 100u) / (n ? n : 1u);
			if (r100 > W.outCounters[13]) atomicMax(&W.outCounters[13], r100);
			if (matchRatio16 & 0x4000u) { atomicAdd(&W.outCounters[10], 1u); atomicAdd(&W.outCounters[11], nRounds); atomicMax(&W.outCounters[12], (K * 100u) / (n ? n : 1u)); atomicAdd(&W.outCounters[14], K); atomicAdd(&W.outCounters[15], n); }      // developer statistics
		}

		LW_MARK(8)
		LW_STOP(4)
		// ---- 5. rank of every appended node at its end position; successor masks; the first node of every position ----
		uint64_t* pred = fd; uint32_t* predw = fdw;      // (the dictionary masks are no longer needed: lengths of the nodes ENDING at a position, for the position program)
		for (uint32_t i = lane; i < nPosAll; i += 64) { cntA[i] = 0; succ[i] = 0; firstU[i] = 0; pred[i] = 0; }      // (these arrays take over the bytes of the scan's masks and the control words)
		waveSync();
		{
			uint32_t carryE = 0xFFFFFFFFu, carryCnt = 0;
			for (uint32_t b0 = 1; b0 <= K; b0 += 64)
			{
				const uint32_t T = b0 + lane; const bool act = T <= K;
				uint32_t nb = 0, e = 0xFFFFFFFEu; bool app = false;
				if (act) { const uint32_t ne = opNE[T]; nb = ne & 0xFFFF; e = ne >> 16; app = decT[T] & 1u; }
				const uint32_t ePrev = __shfl_up(e, 1);
				const bool segStart = lane == 0 ? (e != carryE) : (e != ePrev);
				const uint64_t sb = __ballot(segStart), am = __ballot(app);
				const uint64_t below = (1ull << lane) - 1ull;
				const uint64_t mine = sb & (below | (1ull << lane));      // segment starts at or below this lane
				const uint32_t start = mine ? topBit(mine) : 0u;
				uint32_t rank = (uint32_t)__popcll(am & below & ~((1ull << start) - 1ull));
				if (!mine) rank += carryCnt;      // the segment began in an earlier block of 64 ops
				if (act && app)
				{
					decT[T] = (decT[T] & 0xFFFF0001u) | (rank << 1);
					if (!(opFl[T] & OF_END))
					{
						const uint32_t len = e - nb;
						if (len < 1 || len > 64) hazard = 1; else { atomicOr(&succw[2 * nb + ((len - 1) >> 5)], 1u << ((len - 1) & 31)); atomicOr(&predw[2 * e + ((len - 1) >> 5)], 1u << ((len - 1) & 31)); }
						if (rank == 0) { const uint32_t src = opSrc[T]; firstU[e] = (src & 0x8000u) ? miscU[src & 0x7FFFu] : 0u; }
					}
					atomicAdd(&cntA[e], 1u);
				}
				// carry: the segment of the last lane
				const uint32_t eLast = __shfl(e, 63);
				const uint32_t startLast = sb ? topBit(sb) : 0u;
				const uint32_t inLast = (uint32_t)__popcll(am & ~((1ull << startLast) - 1ull));
				carryCnt = sb ? inLast : carryCnt + inLast;
				carryE = eLast;
			}
		}
		waveSync();
		// unknown-form nodes: successor masks, and the first node of a position that no op ends at
		for (uint32_t q = lane; q < nPosAll; q += 64)
		{
			const uint32_t g0 = grpOff[q], g1 = grpOff[q + 1];
			bool first = cntA[q] == 0 && q != 0;
			for (uint32_t g = g0; g < g1; ++g)
			{
				const uint32_t T = grpList[g]; const uint32_t s4 = decS[T] & 15u;
				if (!s4) continue;
				const uint32_t bu = opBU[T];
				uint32_t lp = decT[T] >> 16;      // (a bridge node was made: the last node ended before q)
				if ((s4 & 5u) && lp && isHangulCoda(str[nsToPos[lp]])) --lp;
				for (uint32_t i = 0; i < 4; ++i)
				{
					if (!((s4 >> i) & 1)) continue;
					const uint32_t s = (i & 1) ? ((i & 2) ? (bu >> 16) : (bu & 0xFFFF)) : lp;
					const uint32_t len = q - s;
					atomicOr(&succw[2 * s + ((len - 1) >> 5)], 1u << ((len - 1) & 31)); atomicOr(&predw[2 * q + ((len - 1) >> 5)], 1u << ((len - 1) & 31));
					if (first) { const uint32_t o = nsToPos[s]; firstU[q] = o | (trimmedLen(str, o, nsToPos[q - 1] + 1u - o) << 16); first = false; }
				}
			}
		}
		waveSync();
		if (__ballot(hazard != 0)) LW_HAND_OVER(4)

		LW_MARK(9)
		LW_STOP(5)
		// ---- 6. removeUnconnected, part 1: from which positions is the end node reachable (sweep from the end; window of the next 64 positions) ----
		uint16_t* keep = unkMinT;      // (no longer needed)
		{
			// lane 0: position by position, downwards; bit d - 1 of win = position q + d is kept
			if (lane == 0)
			{
				uint64_t win = 0;      // (position nNs is kept by definition: the end node starts there and exists)
				keep[nNs + 1] = 1;
				for (uint32_t q = nNs + 1; q-- > 0;)
				{
					const bool k = q == nNs ? true : (succ[q] & win) != 0;
					keep[q] = k ? 1 : 0;
					win = (win << 1) | (k ? 1ull : 0ull);
				}
			}
		}
		waveSync();
		if (!keep[0]) LW_HAND_OVER(5)      // (a lattice that does not reach back to the start node: left to the replay)
		// part 2: first final index of every position
		uint32_t nConn = 0;
		for (uint32_t b0 = 0; b0 < nPosAll; b0 += 64)
		{
			const uint32_t q = b0 + lane;
			uint32_t c = 0;
			if (q < nPosAll && keep[q]) c = q == 0 ? 1u : cntA[q] + cntU[q];
			uint32_t incl = c;
			for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d); if (lane >= d) incl += v; }
			if (q < nPosAll) base[q] = (uint16_t)(nConn + incl - c);
			nConn += __shfl(incl, 63);
		}
		if (nConn + 1 >= cap) { if (lane == 0) W.results[chunk].status = CS_ERR_NODE_OVERFLOW; return; }
		if (nConn > lay.nodeCap) LW_HAND_OVER(6)
		waveSync();

		LW_MARK(10)
		LW_STOP(6)
		// ---- 7. the final records (removeUnconnected part 2 + the per-node facts the search kernel needs).  First the candidate count of every final
		// node (the forms' facts were kept when the matches were digested), their prefix sum = the nodes' candidate-record offsets; then one op per
		// lane builds its nodes whole and stores each once ----
		DevNode* fin = W.nodes + nBase;
		const uint32_t textOff = B.textOffset[chunk];
		auto factsOf = [&](uint32_t form) -> uint32_t { const FormRec f = M.forms[form]; return (uint32_t)(f.candCnt & 0x7FFFu) | ((f.flags2 & FF2_ALL_PARTIAL) ? 0x8000u : 0u) | ((uint32_t)f.flags << 16) | ((uint32_t)f.len << 24); };
		if (lane == 0) cc[0] = 0;
		for (uint32_t T = 1 + lane; T <= K; T += 64)
		{
			const uint32_t ne = opNE[T], nb = ne & 0xFFFF, e = ne >> 16, ds = decS[T], s4 = ds & 15u, dt = decT[T];
			if (s4 && keep[nb]) { uint32_t r = base[nb] + cntA[nb] + ((ds >> 8) & 0xFFu); for (uint32_t i = 0; i < 4; ++i) if ((s4 >> i) & 1) cc[r++] = 0; }
			if ((dt & 1u) && keep[e])
			{
				const uint32_t src = opSrc[T];
				uint32_t fc = 0;
				if (!(src & 0x8000u)) fc = mfc[src];
				else { const uint32_t form = miscForm[src & 0x7FFFu]; if (form != NOFORM) fc = factsOf(form); miscFc[src & 0x7FFFu] = fc; }
				cc[base[e] + ((dt >> 1) & 0x7FFFu)] = fc & 0x7FFFu;
			}
		}
		waveSync();
		LW_MARK(11)
		uint32_t packTop = 0;
		for (uint32_t b0 = 0; b0 < nConn; b0 += 64)
		{
			const uint32_t i = b0 + lane;
			const uint32_t c = i < nConn ? cc[i] : 0u;
			uint32_t incl = c;
			for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d); if (lane >= d) incl += v; }
			if (i < nConn) cc[i] = packTop + incl - c;
			packTop += __shfl(incl, 63);
		}
		waveSync();
		// (uniform) the candidate records would not fit the chunk's pack region: report the overflow BEFORE anything of phase 8 is written -- the expansion
		// below stores packs[cc[i] + ...] and would run into the neighbour chunk's region (or past the end of the batch's); kamd_run re-runs the chunk with 4 x capacities
		if (packTop > W.packBase[chunk + 1] - W.packBase[chunk])
		{
			if (lane == 0) W.results[chunk].status = CS_ERR_NODE_OVERFLOW;
			return;
		}
		// ---- 8. (expandMode bit 0) the candidate records and the position program of the position-step search, straight from the lattice in LDS
		// -- what k_expand_cands and k_expand_pos (lattice_kernels.hip) make of the stored lattice in two more launches.  First every node's
		// candidates: static records (reference order; CoNgram models: the transposed evaluator's class order) and how many of them are evaluated
		// at all; the prefix sum is the nodes' record offsets; then the emission below writes a node's records with the node still in registers ----
		const bool doExpand = (expandMode & 1u) != 0, transposed = (expandMode & 2u) != 0;
		const uint32_t G = nConn;
		const uint32_t nUniq = B.spOff[chunk + 1] - B.spOff[chunk];
		CandStatic* packs = W.packs + W.packBase[chunk];
		const CandStatic* unkPacks = reinterpret_cast<const CandStatic*>(M.unkPacks);
		PosRec* recs = doExpand ? W.posRecs + W.packBase[chunk] : nullptr;
		PosDesc* desc = doExpand ? W.posDesc + nBase : nullptr;
		uint32_t* posSlow = grpOff; uint32_t* posPass1 = posA; uint32_t* posIdx = posZ;      // (free since the fixpoint ended)
		bool posOk = doExpand && G > 2 && G <= 0xFFF0u && nUniq + 1 <= 0x1Fu;
		uint32_t nPos = 0, recTop = 0;
		auto spaceBeforeOf = [&](uint32_t s0, bool isEnd) -> bool
		{
			const uint32_t startStr = isEnd ? n : (uint32_t)nsToPos[s0];
			return s0 == 0 ? (textOff + startStr > 0) : ((uint32_t)nsToPos[s0 - 1] + 1u < startStr);
		};
		if (doExpand)
		{
			if (lane == 0) { desc[0].firstNode = 0; desc[0].nNodes = 0; desc[0].flags = 0; desc[0].firstRec = 0; desc[0].nRec = 0; }      // no positions until the table is complete
			for (uint32_t q = lane; q < nPosAll; q += 64) { posSlow[q] = 0; posPass1[q] = 0; }
			if (lane == 0) recOff[0] = 0;
			waveSync();
			for (uint32_t T = 1 + lane; T <= K; T += 64)
			{
				const uint32_t ne = opNE[T], nb = ne & 0xFFFF, e = ne >> 16, fl = opFl[T], ds = decS[T], s4 = ds & 15u, dt = decT[T];
				if (s4 && keep[nb])
				{
					// unknown forms: their two unknown-noun candidates each
					uint32_t r = base[nb] + cntA[nb] + ((ds >> 8) & 0xFFu);
					for (uint32_t i = 0; i < 4; ++i) if ((s4 >> i) & 1) recOff[r++] = 2;
					const uint32_t bu = opBU[T];
					uint32_t lp = dt >> 16;
					if ((s4 & 5u) && lp && isHangulCoda(str[nsToPos[lp]])) --lp;
					for (uint32_t i = 0; i < 4; ++i)
					{
						if (!((s4 >> i) & 1)) continue;
						const uint32_t s0 = (i & 1) ? ((i & 2) ? (bu >> 16) : (bu & 0xFFFF)) : lp;
						if (s0 && cntA[s0] + cntU[s0] > 256) atomicOr(&posSlow[nb], 1u);      // (more than 256 predecessors: the position is left to the general kernel)
					}
				}
				if ((dt & 1u) && keep[e])
				{
					const uint32_t rank = (dt >> 1) & 0x7FFFu, src = opSrc[T], ni = base[e] + rank;
					if (fl & OF_END) { recOff[ni] = 0; continue; }
					uint32_t form, fc;
					if (src & 0x8000u) { form = miscForm[src & 0x7FFFu]; fc = miscFc[src & 0x7FFFu]; } else { form = mforms[src]; fc = mfc[src]; }
					if (nb && cntA[nb] + cntU[nb] > 256) atomicOr(&posSlow[e], 1u);
					if (form == NOFORM) { recOff[ni] = 2; continue; }
					const uint32_t candCnt = fc & 0x7FFFu, candOff = M.forms[form].candOff, pk = cc[ni];
					const bool spaceBefore = spaceBeforeOf(nb, false);
					uint32_t cnt = 0, classCnt[5] = { 0, 0, 0, 0, 0 };
					if (transposed)
						for (uint32_t k = 0; k < candCnt; ++k) { const MorphRec r = M.morphs[M.formCand[candOff + k]]; ++classCnt[candClassOf(r.tag, r.socket, r.flags)]; }
					uint32_t classAt[5] = { 0, classCnt[0], classCnt[0] + classCnt[1], classCnt[0] + classCnt[1] + classCnt[2], classCnt[0] + classCnt[1] + classCnt[2] + classCnt[3] };
					for (uint32_t k = 0; k < candCnt; ++k)
					{
						const CandStatic o = candStaticOf(M, M.formCand[candOff + k]);
						const uint32_t flags = o.m1.y & 0xFFFF; const uint8_t tag = (uint8_t)o.m1.z;
						const uint32_t at = transposed ? classAt[candClassOf(tag, o.m1.z >> 24, flags)]++ : k;
						packs[pk + at] = o;
						if (posCandKind(P, flags, tag, spaceBefore)) ++cnt;
					}
					if (cnt == 0) atomicOr(&posSlow[e], 1u);      // nothing to evaluate: the reference then retries without conditions and falls back (PathEvaluator.hpp:468-473, 1286-1299)
					if ((fc & 0x8000u) && rank < 16) atomicOr(&posPass1[e], 1u << rank);
					recOff[ni] = cnt + ((fc & 0x8000u) ? 1u : 0u);
				}
			}
			waveSync();
			for (uint32_t b0 = 0; b0 < G; b0 += 64)
			{
				const uint32_t i = b0 + lane;
				const uint32_t c = i < G ? recOff[i] : 0u;
				uint32_t incl = c;
				for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d); if (lane >= d) incl += v; }
				if (i < G) recOff[i] = recTop + incl - c;
				recTop += __shfl(incl, 63);
			}
			// the end positions that hold nodes, numbered from 1 (the start node is position 0; the end node is none)
			for (uint32_t b0 = 0; b0 < nPosAll; b0 += 64)
			{
				const uint32_t q = b0 + lane;
				const bool has = q >= 1 && q <= nNs && keep[q] && cntA[q] + cntU[q] > 0;
				const uint64_t hb = __ballot(has);
				if (q < nPosAll) posIdx[q] = has ? nPos + 1u + (uint32_t)__popcll(hb & ((1ull << lane) - 1ull)) : 0u;
				nPos += (uint32_t)__popcll(hb);
			}
			const uint32_t recCap = W.packBase[chunk + 1] - W.packBase[chunk];
			if (recTop > recCap || recTop >= 0x3FFFFu || nPos >= 0x1FFFu) posOk = false;      // (uniform) the records do not fit the chunk's region / the packed fields: the general kernel takes the chunk
			waveSync();
			if (posOk)
			{
				uint32_t* posMask = W.posMask + nBase;
				bool overAny = false;
				for (uint32_t q = 1 + lane; q <= nNs; q += 64)
				{
					const uint32_t p = posIdx[q];
					if (!p) continue;
					const uint32_t n0 = base[q], nN = cntA[q] + cntU[q], r0 = recOff[n0], r1 = n0 + nN < G ? recOff[n0 + nN] : recTop;
					const bool slow = nN > 16 || r1 - r0 > 16 || r1 == r0 || posSlow[q] != 0;
					// the distinct start positions of the position's nodes, four bytes; more than four, or a chunk of more than 255 positions: no propagation for this chunk
					uint32_t st4[4] = { 0, 0, 0, 0 }, nSt = 0; bool over = nPos > 255;
					for (uint64_t m = pred[q]; m && !over; m &= m - 1)
					{
						const uint32_t len = (uint32_t)__ffsll((unsigned long long)m), sp = posIdx[q - len];
						if (nSt == 4) { over = true; break; }
						st4[nSt++] = sp;
					}
					for (uint32_t t = nSt; t < 4; ++t) st4[t] = st4[0];
					posMask[p] = over ? 0xFFFFFFFFu : (st4[0] | (st4[1] << 8) | (st4[2] << 16) | (st4[3] << 24));
					overAny = overAny || over;
					// bit j: node j of the position has no dictionary form -- its unknown-form nodes, which follow the others
					const uint32_t a = cntA[q], tot = nN < 16 ? nN : 16u;
					const uint32_t formless = a < 16 ? (((1u << tot) - 1u) & ~((1u << a) - 1u)) : 0u;
					PosDesc d;
					d.firstNode = (uint16_t)n0; d.nNodes = (uint8_t)(nN > 255 ? 255 : nN); d.flags = slow ? (uint8_t)POSF_SLOW : (uint8_t)0;
					d.firstRec = r0; d.nRec = (uint16_t)(r1 - r0 > 0xFFFF ? 0xFFFF : r1 - r0); d.pad = (uint16_t)formless; d.pad2 = posPass1[q];
					desc[p] = d;
				}
				if (__ballot(overAny) && lane == 0) desc[0].nNodes = 1;
				if (lane == 0)
				{
					PosDesc d; d.firstNode = (uint16_t)(G - 1); d.nNodes = 0; d.flags = 0; d.firstRec = recTop; d.nRec = 0; d.pad = 0; d.pad2 = 0;
					desc[nPos + 1] = d;
					desc[0].pad2 = posIdx[nNs];      // where the end node's predecessors end: reachable <=> the lattice is connected
				}
			}
		}
		auto emitNode = [&](uint32_t ni, uint32_t s, uint32_t t, uint32_t form, uint32_t fc, uint32_t uOff, uint32_t uLen, uint32_t se, uint32_t rankAt, bool isEnd)
		{
			DevNode nn;
			nn.form = form; nn.uformOff = (uint16_t)uOff; nn.uformLen = (uint16_t)uLen; nn.spaceErrors = (uint8_t)(se > 255 ? 255 : se);
			nn.ownFeat = 0; nn.pad = 0; nn.packOff = cc[ni];
			const bool pnBos = s == 0;
			// the reference compares absolute text offsets; the start node's end is 0 (PathEvaluator.hpp:24-31, 436, 568)
			const bool spaceBefore = spaceBeforeOf(s, isEnd);
			bool lb = pnBos || spaceBefore;
			const uint32_t pu = firstU[s];
			if (!lb && (pu >> 16))
			{
				const uint32_t lp = (pu & 0xFFFF) + (pu >> 16) - 1;
				const uint16_t c = str[lp];
				const uint8_t tag = (isLowSurrogate(c) || isHighSurrogate(c)) ? (uint8_t)T_SH : (uint8_t)(cls[lp] & 0x3F);
				if (tag == T_SSC || c == u'"' || c == u'\'') lb = false;
				else if (T_SF <= tag && tag <= T_SB) lb = true;
			}
			uint8_t nf = 0;
			if (spaceBefore) nf |= NF_SPACE_BEFORE;
			if (lb) nf |= NF_LEFT_BOUNDARY;
			if (uLen && str[uOff + uLen - 1] == u'.') nf |= NF_UFORM_ENDS_POINT;
			if (fc & 0x8000u) nf |= NF_ALL_PARTIAL;
			nn.nPrev = (uint16_t)(pnBos ? 1u : cntA[s] + cntU[s]);
			nn.candCnt = (uint16_t)(fc & 0x7FFFu); nn.fflags = (uint8_t)(fc >> 16); nn.flen = (uint8_t)(fc >> 24);
			if (uLen)
			{
				uint16_t of = featMaskFast(str + uOff, uLen) & 0x1FFF;
				const uint32_t lp = uOff + uLen - 1;
				const uint16_t c = str[lp];
				const uint8_t tag = (isLowSurrogate(c) || isHighSurrogate(c)) ? (uint8_t)T_SH : (uint8_t)(cls[lp] & 0x3F);
				if (tag == T_SSC) of |= LF_STR_SSC;
				nn.ownFeat = of;
			}
			nn.nflags = nf;
			nn.prev = (uint16_t)(ni - base[s]);
			nn.sibling = (!isEnd && rankAt + 1 < cntA[t] + cntU[t]) ? 1 : 0;
			if (isEnd) nn.startPos = nn.endPos = (uint16_t)n;
			else { nn.startPos = nsToPos[s]; nn.endPos = (uint16_t)(nsToPos[t - 1] + 1); }
			fin[ni] = nn;
			if (!posOk || isEnd) return;
			// the node's entry of the per-node predecessor table and its records (k_expand_pos pass B' and C; PathEvaluator.hpp:366-383, 1204-1318)
			{
				const uint32_t np1 = (uint32_t)nn.nPrev - 1u, sp = posIdx[s];
				(W.posPrev + nBase)[ni] = base[s] | ((np1 > 255u ? 255u : np1) << 16) | ((sp > 255u ? 255u : sp) << 24);
			}
			const uint32_t nl = rankAt;      // the node's index inside its position
			if (nl >= 16) return;          // (its position is marked slow)
			PosRec* out = recs + recOff[ni];
			float ws = 0;
			if (!nn.uformLen && nn.form != NOFORM && nn.flen && nn.spaceErrors) ws = -P.spacePenalty * (float)nn.spaceErrors;
			const float baseDiscount = ws + (-0.f * P.typoCostWeight);      // whitespaceDiscount + typoDiscount (no typo costs on this path)
			const uint8_t ownKind0 = nn.uformLen ? 1 : 0;
			auto emit = [&](const CandStatic* cs, float disc, uint8_t ownKind, uint16_t ownFeat, uint32_t extra)
			{
				const uint4* q4 = reinterpret_cast<const uint4*>(cs);
				const uint4 m0 = q4[0], m1 = q4[1], mx = q4[2];
				const uint8_t tag = (uint8_t)m1.z, special = (uint8_t)(m1.w >> 24);
				const uint32_t sbType = mx.z;
				const bool quote = special == 0 || special == 1 || special == 3 || special == 4;
				const uint32_t R = ((sbType || quote) && nUniq > 1) ? nUniq : 1u;
				PosRec r;
				r.firstWid = mx.y; r.secondWid = mx.w; r.chunkOff = m0.z; r.lastSeqId = m0.y;
				r.morph = mx.x; r.flagsFeat = m1.y; r.tagw = m1.z; r.cntw = m1.w;
				r.additional = __uint_as_float(m0.w) + disc + leftBoundaryScore(((nn.nflags & NF_LEFT_BOUNDARY) ? T_MAX : 0) + clearIrregular(tag)) * 5.f;
				const uint32_t ruleBits = ((isEClass(tag) && (nn.fflags & FF_STARTS_WITH_A)) ? 1u : 0u) | ((tag == T_SN && (nn.nflags & NF_UFORM_ENDS_POINT)) ? 2u : 0u)
					| ((M.morphDialect && M.morphDialect[r.morph]) ? 4u : 0u);      // (RB_DIALECT of the search kernels)
				r.nodeOwn = ni | ((uint32_t)ownFeat << 16);
				r.bits = (sbType & 0xFF) | (ruleBits << 8) | ((uint32_t)ownKind << 16) | ((uint32_t)nn.nflags << 24);
				r.rq = R | (nl << 8) | extra;
				*out++ = r;
			};
			// CoNgram: do the regular candidates of one evaluation share their first word?  (decides which of the reference's kernels rounds their scores)
			auto sharedFirstWord = [&](const CandStatic* cl, uint32_t nc) -> uint32_t
			{
				if (!transposed) return 0u;      // (only the CoNgram kernels read the flag)
				uint32_t nReg = 0, ref = 0; bool one = true;
				for (uint32_t k = 0; k < nc; ++k)
				{
					const uint4* q4 = reinterpret_cast<const uint4*>(cl + k);
					const uint4 m1 = q4[1], mx = q4[2];
					const uint32_t flags = m1.y & 0xFFFF; const uint8_t tag = (uint8_t)m1.z, sock = (uint8_t)(m1.z >> 24);
					if (posCandKind(P, flags, tag, spaceBefore) != 1 || sock || (flags & MF_FIRST_WID_IS_P)) continue;
					if (!nReg) ref = mx.y; else if (mx.y != ref) one = false;
					++nReg;
				}
				return (nReg && one) ? (uint32_t)PR_OUT_FIRST : 0u;
			};
			if (nn.form != NOFORM)
			{
				const CandStatic* cl = packs + nn.packOff;      // (this lane's own stores of a moment ago)
				const uint32_t of0 = sharedFirstWord(cl, nn.candCnt);
				for (uint32_t k = 0; k < nn.candCnt; ++k)
				{
					const uint4 m1 = reinterpret_cast<const uint4*>(cl + k)[1];
					const uint32_t kind = posCandKind(P, m1.y & 0xFFFF, (uint8_t)m1.z, spaceBefore);
					if (kind == 1) emit(cl + k, baseDiscount + 0.f, ownKind0, nn.ownFeat, of0);
					else if (kind == 2)
					{
						// z-coda / z-siot shortcut (PathEvaluator.hpp:389-432): the record carries the morpheme put on, the shortcut's tag and score, and what a path ending in the new morpheme exposes
						const MorphRec cm = M.morphs[reinterpret_cast<const uint4*>(cl + k)[2].x];
						const MorphRec nm = M.morphs[cm.lmId];
						PosRec r;
						r.firstWid = cm.lmId; r.secondWid = cm.tag; r.chunkOff = (uint32_t)nm.feat | ((uint32_t)nm.prevFlags << 16) | (nm.socket ? 1u << 24 : 0u); r.lastSeqId = 0;
						r.morph = cm.lmId; r.flagsFeat = 0; r.tagw = 0; r.cntw = 0;
						r.additional = cm.userScore;
						r.nodeOwn = ni; r.bits = (uint32_t)nn.nflags << 24; r.rq = 1u | (nl << 8) | (uint32_t)PR_Z;
						*out++ = r;
					}
				}
				if (nn.nflags & NF_ALL_PARTIAL)
				{
					// the form read as an unknown proper noun (PathEvaluator.hpp:1277-1287): own form = the dictionary form's string
					const FormRec f = M.forms[nn.form];
					uint16_t of = featMaskFast(M.formChars + f.charOff, f.len) & 0x1FFF;
					if (f.flags & FF_ENDS_WITH_SSC) of |= LF_STR_SSC;
					const float disc = baseDiscount + -((float)f.len * P.oovRuleScale + P.oovRuleBias);
					emit(unkPacks + 1, disc, 2, of, (uint32_t)PR_PASS1 | sharedFirstWord(unkPacks + 1, 1));
				}
			}
			else
			{
				// unknown form: the two unknown-noun candidates (PathEvaluator.hpp:1204-1206, 1300-1318), scored by UnkFormScorer (src/UnkFormScorer.h:40-58)
				const float emo = (cls[nn.uformOff] & 0x80) ? -10.f : 0.f;
				const float disc = baseDiscount + (emo - ((float)nn.uformLen * P.oovRuleScale + P.oovRuleBias));
				const uint32_t of0 = sharedFirstWord(unkPacks, 2);
				emit(unkPacks, disc, ownKind0, nn.ownFeat, of0);
				emit(unkPacks + 1, disc, ownKind0, nn.ownFeat, of0);
			}
		};
		if (lane == 0)
		{
			DevNode bos; bos.form = NOFORM; bos.startPos = bos.endPos = 0; bos.prev = bos.sibling = 0; bos.uformOff = bos.uformLen = 0; bos.spaceErrors = 0; bos.nflags = 0; bos.nPrev = 0; bos.packOff = 0; bos.candCnt = 0; bos.fflags = 0; bos.flen = 0; bos.ownFeat = 0; bos.pad = 0;
			fin[0] = bos;
		}
		for (uint32_t T = 1 + lane; T <= K; T += 64)
		{
			const uint32_t ne = opNE[T], nb = ne & 0xFFFF, e = ne >> 16, fl = opFl[T], s4 = decS[T] & 15u, dt = decT[T];
			if (s4 && keep[nb])
			{
				const uint32_t bu = opBU[T];
				uint32_t lp = dt >> 16;
				if ((s4 & 5u) && lp && isHangulCoda(str[nsToPos[lp]])) --lp;
				uint32_t r = cntA[nb] + ((decS[T] >> 8) & 0xFFu);
				for (uint32_t i = 0; i < 4; ++i)
				{
					if (!((s4 >> i) & 1)) continue;
					const uint32_t s0 = (i & 1) ? ((i & 2) ? (bu >> 16) : (bu & 0xFFFF)) : lp;
					const uint32_t o = nsToPos[s0], len = nsToPos[nb - 1] + 1u - o, l = plain ? len : trimmedLen(str, o, len);
					emitNode(base[nb] + r, s0, nb, NOFORM, 0, o, l, 0, r, false);
					++r;
				}
			}
			if ((dt & 1u) && keep[e])
			{
				const uint32_t rank = (dt >> 1) & 0x7FFFu, src = opSrc[T];
				uint32_t form, fc, uOff = 0, uLen = 0, se = 0;
				if (src & 0x8000u) { form = miscForm[src & 0x7FFFu]; fc = miscFc[src & 0x7FFFu]; const uint32_t mu = miscU[src & 0x7FFFu]; uOff = mu & 0xFFFF; uLen = mu >> 16; }
				else { form = mforms[src]; fc = mfc[src]; se = mse[src]; }
				emitNode(base[e] + rank, nb, e, form, fc, uOff, uLen, se, rank, (fl & OF_END) != 0);
			}
		}
		if (doExpand)
		{
			waveSync();
			if (lane == 0) { if (posOk) desc[0].firstRec = nPos; W.expanded[chunk] = 1; }
		}
		LW_MARK(12)
		if (lane == 0)
		{
			const uint32_t packCap = W.packBase[chunk + 1] - W.packBase[chunk];
			if (packTop > packCap) W.results[chunk].status = CS_ERR_NODE_OVERFLOW;
			else { W.nNodes[chunk] = nConn; if (nConn <= 2) W.results[chunk].status = CS_NO_LATTICE; }
		}
	}
}
