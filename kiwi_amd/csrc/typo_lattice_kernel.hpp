// Launch-side declarations of the typo-lattice kernel (typo_lattice_kernel.hip): plain device pointers, one record per chunk.
#pragma once
#include <hip/hip_runtime.h>
#include "device_types.hpp"
#include "typo.hpp"

namespace kamd
{
	struct TypoLatChunk
	{
		uint32_t charOff, nChars;      // chunk text inside chars / cls / script
		uint32_t patOff, patCnt;       // pattern spans of the chunk (chunk-relative)
		uint32_t graphOff, graphCnt;   // its typo graph
		uint32_t pmb;                  // posMultiplierBit = ceil(log2(max continual-typo index)) (KTrie.cpp:860-891)
		uint32_t nNs;                  // non-space positions (the kernel recounts and compares)
		uint32_t nodeOff, nodeCap;     // build-order and final node regions (same offsets), 3 x nodeCap words of scratch
		uint32_t mapOff, mapLen;       // endPosMap: ((nNs << pmb) + 1) entries
		uint32_t nsOff;                // nsToPos / posToNs: nChars + 2 entries each
		uint32_t stateOff, stateCap;   // search-state arena
		uint32_t textOffset;           // offset of the chunk in the normalised text (added to final positions of the dump records)
		uint32_t chunkId, packCap;     // engine mode: index of the chunk in the batch, capacity of its candidate-pack region
		uint32_t nOutFinal, status;    // out: number of connected nodes, ChunkStatus
		uint32_t ldsGraphCap;          // engine mode: graph nodes whose records / state ranges are kept in LDS (graphCnt, or 0: in HBM)
		uint32_t ldsCap;               // engine mode: nodes the chunk's LDS copy holds (typoLdsNodeCap, host-computed: the kernel lays its LDS out from it)
		uint32_t ldsNeed, pad;         // engine mode: dynamic LDS the wave-per-chunk kernel needs for this chunk (typoLdsLayout().total)
	};
	// lattice node in the layout of the parity dumps (kamd_dump_lattices; the reference bridge writes the same): positions are text offsets when final
	struct TypoLatNode { uint32_t startPos, endPos, prev, sibling; int32_t form; uint32_t uformLen, uformOff, spaceErrors; float typoCost; };
	// SearchState (KTrie.cpp:671-707); of the last character its type / script travel along (what the next node derives from it; the host
	// computes them per graph node) and, for lengthening typos, the character itself; lsize / lnode = LengtheningTypoNodes<true>
	constexpr uint32_t kTypoLengthNodes = 64;
	struct TypoState
	{
		int32_t node; float cost; uint32_t minFormLen; int32_t startPosOffset;
		uint32_t specialStart, unkStart, boundary;
		uint8_t lastType, lastScript, hasLast, pad; uint16_t startCti, pad2;
		uint32_t lastChr, nL;
		uint8_t lsize[kTypoLengthNodes]; int32_t lnode[kTypoLengthNodes];
	};
	struct TypoLatView
	{
		const uint16_t* chars; const uint8_t* cls; const uint8_t* script; const DevPattern* patterns;
		const TypoGraphNode* graph;
		const uint8_t* graphLast;      // per graph node: {type, script} of its last character (type 0xFF: the character is NUL = "none"); host-computed
		const uint16_t* pool;          // replacement strings of the prepared transformer
		TypoLatChunk* chunks;
		TypoLatNode* nodes; TypoLatNode* nodesFinal;
		uint2* endPosMap; uint16_t* nsToPos; uint16_t* posToNs;
		TypoState* states; uint32_t* stateIdx;      // per graph node: {first state, count}
		uint32_t* scratch;
		// engine mode (all null in the parity hook): the search kernel's node records (chunk-relative positions, per-node facts as
		// latticeEmitNode computes them), each node's typo cost beside them, node counts and chunk statuses of the batch
		DevNode* devNodes; float* nodeTypo; uint32_t* nNodes; DevChunkResult* results;
		float threshold; uint32_t maxUnk, maxUnkJ, spaceTol; uint64_t match;
		float lengtheningCost;         // INFINITY: no lengthening typos (PreparedTypoTransformer::getLengtheningTypoCost)
	};
	void launchTypoLattice(const ModelView& M, const TypoLatView& V, uint32_t nChunks, hipStream_t stream);

	// ---- wave-per-chunk variant (engine mode): the chunk's text, index maps, node list and end-position index live in the wave's LDS ----
	constexpr uint32_t kTypoLdsNeedsBig = 0xFFFFu;      // TypoLatChunk::status: left to the thread-per-chunk kernel
	constexpr uint32_t kTypoMaxCand = 96;               // dictionary candidates of one input position
	// node of the build in LDS (24 bytes; positions are multiplied positions < 65536)
	struct TypoLdsNode { uint32_t form; uint16_t startPos, endPos, prev, sibling, uformOff, uformLen; float typoCost; uint8_t spaceErrors, pad[3]; };
	// graph / sidx / ring (round 6): the chunk's typo graph, its per-node state ranges and the heads of the most recent search states, which the replay read from
	// HBM once per (node, predecessor, state) -- over half of its 2 500 dependent loads per chunk; graphCap = 0: left in HBM (graphs of more than kTypoLdsGraphMax nodes, KAMD_TYPO_LDS_TABLES=0)
	constexpr uint32_t kTypoLdsGraphMax = 255, kTypoLdsRing = 24, kTypoStateHeadWords = 11;
	// a graph node as the LDS copy keeps it (16 bytes: offsets inside a graph of at most kTypoLdsGraphMax nodes fit a byte, positions of a chunk sixteen bits)
	struct TypoLdsGraphNode { uint32_t formOff; float typoCost; uint16_t formLen, endPos; uint8_t prevOffset, siblingOffset, continualTypoIdx, pad; };
	static_assert(sizeof(TypoLdsGraphNode) == 16, "TypoLdsGraphNode");
	struct TypoLds { uint32_t str, cls, script, nsToPos, posToNs, epm, fullMask, zAt, cands, nodes, queue, conn, nodeCap, graph, sidx, ring, glast, graphCap, total; };
	// nodes a chunk's LDS copy holds: mul4 / 4 per text unit + add (the defaults cover every chunk of the bench corpora; a chunk that outgrows its copy is
	// built again by the thread-per-chunk kernel).  Host side only -- the kernel reads TypoLatChunk::ldsCap.
#ifdef KAMD_TEST_SMALL_CAPS
	// test build (make smallcaps): most chunks outgrow their LDS node list and are handed to the thread-per-chunk kernel
	inline uint32_t typoLdsNodeCap(uint32_t nChars, uint32_t nodeCap, uint32_t = 0, uint32_t = 0) { const uint32_t c = nChars / 2 + 4; return c < nodeCap ? c : nodeCap; }
#else
	inline uint32_t typoLdsNodeCap(uint32_t nChars, uint32_t nodeCap, uint32_t mul4 = 14, uint32_t add = 40) { const uint32_t c = mul4 * nChars / 4 + add; return c < nodeCap ? c : nodeCap; }
#endif
	__host__ __device__ inline TypoLds typoLdsLayout(uint32_t nChars, uint32_t nNs, uint32_t pmb, uint32_t ldsNodeCap, uint32_t ldsGraphCap)
	{
		TypoLds l; uint32_t top = 0;
		auto take = [&](uint32_t bytes, uint32_t align) { top = (top + align - 1) / align * align; const uint32_t o = top; top += bytes; return o; };
		const uint32_t mapLen = (nNs << pmb) + 1;
		l.nodeCap = ldsNodeCap;
		l.fullMask = take(8 * (nNs + 1), 8);
		l.nodes = take(24 * l.nodeCap, 8);
		l.epm = take(4 * mapLen, 4);
		l.cands = take(4 * kTypoMaxCand, 4);
		l.str = take(2 * nChars, 2);
		l.nsToPos = take(2 * (nChars + 2), 2);
		l.posToNs = take(2 * (nChars + 2), 2);
		l.cls = take(nChars, 1);
		l.script = take(nChars, 1);
		l.zAt = take(nNs + 1, 1);
		// the tables of the SEARCH (graph copy, state ranges, ring of state heads) and the arrays of what follows it (the queue and the flags of removeUnconnected)
		// share their bytes: the search has ended when the sweep starts
		l.graphCap = ldsGraphCap;
		const uint32_t shared = take(0, 16);
		l.graph = take(16 * ldsGraphCap, 16);
		l.sidx = take(4 * ldsGraphCap, 4);
		l.ring = take(ldsGraphCap ? 4 * kTypoStateHeadWords * kTypoLdsRing : 0, 4);
		l.glast = take(2 * ldsGraphCap, 2);      // TypoLatView::graphLast of the chunk's nodes
		const uint32_t searchEnd = top;
		top = shared;
		l.queue = take(2 * l.nodeCap, 2);
		l.conn = take(2 * l.nodeCap, 2);
		if (top < searchEnd) top = searchEnd;
		l.total = (top + 15) / 16 * 16;
		return l;
	}
	// chunkList: chunk indices into V.chunks, all with ldsNeed <= ldsBytes (the launch's dynamic LDS size)
	void launchTypoLatticeLds(const ModelView& M, const TypoLatView& V, const uint32_t* chunkList, uint32_t chunkCount, uint32_t ldsBytes, hipStream_t stream);
	// the thread-per-chunk kernel over all chunks; ldsBudget > 0: chunks with ldsNeed <= ldsBudget that the LDS kernel finished are skipped
	void launchTypoLatticeRest(const ModelView& M, const TypoLatView& V, uint32_t nChunks, uint32_t ldsBudget, hipStream_t stream);
}
