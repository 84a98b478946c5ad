// removeUnconnected, part 1 (reference src/KTrie.cpp:240-299) for the wave-per-chunk lattice kernels (lattice_kernels.hip, typo_lattice_kernel.hip):
// which nodes reach the end node, and the new index of every such node -- nodes grouped by end position ascending, original order inside a group.
#pragma once
#include <hip/hip_runtime.h>
#include "device_types.hpp"

namespace kamd
{
	// All W lanes of the chunk's lane group call it (W = 64: one chunk per wavefront; 16: four).  `out`: the build-order node list (fields startPos, sibling; node 0 = start node, node G - 1 = end node, which is on no
	// chain); endPosMap[p] = first | (last + 1) << 16 of the nodes ending at position p (0 .. nPos - 1), whose sibling links chain exactly those nodes in
	// index order.  Returns the number of connected nodes; inv[i] = new index (0xFFFF: dropped); endPosMap[p] := connected nodes ending at p.
	// Instead of a backward BFS with a queue and range scans (the ids between the first and the last node ending at a position include every node
	// appended in between -- OOV-span nodes are appended long after the position they end at; measured 31 % of k_build_lattice on mixed-length chunks):
	//   1. all lanes clear the flags;  2. lane 0 sweeps the positions from the end downwards -- a node is connected iff a connected node starts where it
	//   ends, and every edge leads to a strictly smaller position;  3. one position per lane: connected nodes on its chain, wave scan over the positions
	//   (the first new index of a position is parked in inv[first node of its chain]);  4. one position per lane: new indices.
	// flagBits: nPos bits of scratch (an array of the build that is no longer needed).
	template<int W = 64, class NodeT>
	__device__ __forceinline__ uint32_t latticeConnectWave(NodeT* out, uint32_t* endPosMap, uint16_t* inv, uint16_t* conn, uint32_t* flagBits, uint32_t G, uint32_t nPos, uint32_t lane)
	{
		for (uint32_t i = lane; i < G; i += W) conn[i] = 0;
		for (uint32_t w = lane; w < (nPos + 31) / 32; w += W) flagBits[w] = 0;
		waveSync();
		if (lane == 0)
		{
			conn[G - 1] = 1;
			{ const uint32_t sp = out[G - 1].startPos; flagBits[sp >> 5] |= 1u << (sp & 31); }
			for (uint32_t p = nPos; p-- > 0;)
			{
				if (!((flagBits[p >> 5] >> (p & 31)) & 1)) continue;
				const uint32_t me = endPosMap[p];
				if ((me & 0xFFFF) == (me >> 16)) continue;
				for (uint32_t i = me & 0xFFFF;;)
				{
					const NodeT g = out[i];
					if (i != G - 1) { conn[i] = 1; const uint32_t sp = g.startPos; flagBits[sp >> 5] |= 1u << (sp & 31); }
					if (!g.sibling) break;
					i += g.sibling;
				}
			}
		}
		waveSync();
		uint32_t total = 0;
		for (uint32_t base = 0; base < nPos; base += W)
		{
			const uint32_t e = base + lane;
			uint32_t c = 0, first = 0xFFFFFFFFu;
			if (e < nPos)
			{
				const uint32_t me = endPosMap[e];
				if ((me & 0xFFFF) != (me >> 16))
				{
					first = me & 0xFFFF;
					for (uint32_t i = first;;)
					{
						if (i != G - 1 && conn[i]) ++c;
						const uint32_t sib = out[i].sibling;
						if (!sib) break;
						i += sib;
					}
				}
			}
			uint32_t incl = c;
			for (uint32_t d = 1; d < W; d <<= 1) { const uint32_t v = __shfl_up(incl, d, W); if (lane >= d) incl += v; }
			if (first != 0xFFFFFFFFu) inv[first] = (uint16_t)(total + incl - c);      // first new index of the nodes ending at e
			total += __shfl(incl, W - 1, W);
		}
		waveSync();
		for (uint32_t base = 0; base < nPos; base += W)
		{
			const uint32_t e = base + lane;
			if (e >= nPos) continue;
			const uint32_t me = endPosMap[e];
			uint32_t c = 0;
			if ((me & 0xFFFF) != (me >> 16))
			{
				uint32_t next = inv[me & 0xFFFF];
				for (uint32_t i = me & 0xFFFF;;)
				{
					const uint32_t sib = out[i].sibling;      // (read before inv[i] is written: inv[first] still holds the position's base above)
					if (i != G - 1)
					{
						if (conn[i]) { inv[i] = (uint16_t)next++; ++c; }
						else inv[i] = (uint16_t)0xFFFF;
					}
					if (!sib) break;
					i += sib;
				}
			}
			endPosMap[e] = c;   // from here on: number of connected nodes ending at e
		}
		if (lane == 0) inv[G - 1] = (uint16_t)total;
		waveSync();
		return total + 1;
	}
}
