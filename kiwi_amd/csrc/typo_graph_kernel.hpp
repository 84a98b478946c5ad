// Launch-side declarations of the typo-graph kernel (typo_graph_kernel.hip): the typo graph of every chunk of a
// batch generated on the device from the prepared transformer's flat tables (typo.hpp) -- PreparedTypoTransformer::generateGraph
// (/root/reference/src/TypoTransformer.cpp:594-628, 811-1049).  Plain device pointers, one record per chunk.
#pragma once
#include <hip/hip_runtime.h>
#include "typo.hpp"

namespace kamd
{
	struct TypoGraphTables
	{
		const PreparedTypo::TrieNode* trie; const uint16_t* keys; const uint32_t* children;
		const PreparedTypo::Pattern* pats; const PreparedTypo::Repl* repls;
		const uint8_t* replLast;       // 6 bytes per replacement (PreparedTypo::replLast)
		const uint16_t* pool;          // replacement strings + one NUL unit
		float continualCost; int32_t continualOn;      // continualOn = isfinite(continualCost)
		int32_t entryNode;             // the automaton's state before the first character (the NUL edge of the root)
		uint8_t hiType, hiScript, loType, loScript;    // identifySpecialChr / chr2ScriptType of a lone high / low surrogate unit
	};
	// graphOff / graphCap: the chunk's region of graph / graphLast / temp (write pass: exactly the count the count pass reported);
	// scrOff / scrCap: its region of the small working arrays, PreparedTypo::scratchCapFor(nChars, graphCap) entries
	struct TypoGraphChunk { uint32_t charOff, nChars, graphOff, graphCap, scrOff, scrCap; };
	// status: 0 ok, 1 a region was too small (the bounds make that impossible), 2 more than kTypoMaxContinual continual first characters in one
	// pattern, 3 a pattern longer than the text before its end (a rule with an explicit NUL): the engine refuses the batch loudly
	struct TypoGraphOut { uint32_t graphCnt, maxCti, status; };
	constexpr uint32_t kTypoMaxContinual = 32;
	struct TypoGraphView
	{
		const uint16_t* chars; const uint8_t* cls; const uint8_t* script;      // the batch's text block (chunk c: + chunks[c].charOff)
		const TypoGraphChunk* chunks; TypoGraphOut* out;
		TypoGraphNode* graph; uint8_t* graphLast;      // out, at graphOff: the graph in end-position order, {type, script} of every node's last character
		TypoGraphNode* temp;                            // write pass: build-order nodes, at graphOff
		uint2* matches; uint32_t* bp; uint2* epm;      // at scrOff: the pending cluster's matches / break points, the end-position index of the cluster
		uint32_t* rev; uint32_t* cnt;                  // write pass, at scrOff: new index of every node, counting-sort buckets (first: last-character facts)
		uint32_t allowedDialect;
	};
	// countOnly: out[c].graphCnt = the number of nodes the chunk's graph has, nothing else is written
	void launchTypoGraph(const TypoGraphTables& T, const TypoGraphView& V, uint32_t nChunks, bool countOnly, hipStream_t stream);
}
