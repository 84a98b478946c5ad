// Pretokenized spans of Kiwi::analyze (the `pretokenized` argument of kiwi_analyze{,_w}, include/kiwi/capi.h:1351-1407): what the caller's spans become
// before the lattice is built -- makePretokenizedSpanGroup, /root/reference/src/Kiwi.cpp:785-946.
//   * a span without tokens points at the dictionary form spelled by its text, or at the default form of tag NNP with the text as the node's own string;
//   * a span of ONE token points at the dictionary form of that spelling if it has exactly one candidate of the token's tag; otherwise it gets a TEMPORARY form
//     whose candidates are the entry's morphemes of that tag (at most two), or one temporary morpheme with the tag's default LM id;
//   * a span of several tokens gets a temporary form with ONE temporary morpheme whose chunks are the tokens (dictionary morphemes of exactly that spelling and
//     tag, else temporary ones), each with its range inside the span.
// Temporary forms / morphemes live behind the model's own (TempEntries, flat_model.hpp); the engine uploads their derived records per batch (TempOverlay).
#pragma once
#include <cstdint>
#include <string>
#include <vector>
#include "flat_model.hpp"

namespace kamd
{
	struct PtToken { std::u16string form; uint32_t begin, end; uint8_t tag; bool inferRegularity; };      // BasicToken (include/kiwi/Types.h:393-403); begin / end relative to the span
	struct PtSpan { uint32_t begin, end; std::vector<PtToken> tokens; };                                      // PretokenizedSpan (:405-413), offsets into the RAW text
	struct PretokGroup
	{
		// the spans in offsets of the NORMALISED text, ascending, with the form their lattice node carries; fallback: a default tag form, the node takes the text as
		// its own string (KTrie.cpp:1197-1200)
		struct Span { uint32_t begin, end, form; bool fallback; };
		std::vector<Span> spans;
		TempEntries temps;
		TempOverlay overlay;
		std::vector<MorphRec> devMorphs;      // overlay.morphs as the device keeps morphemes: feat / prevFlags are what a PATH ending in the morpheme exposes (morphPath)
		// the tables the result assembly reads (post.cpp) with the temporaries appended -- only filled when there are temporaries
		FlatModel hostModel;
		bool hasTemps() const { return !overlay.empty(); }
	};
	// throws std::invalid_argument for spans outside the text, empty or overlapping ones (the reference's own message for the latter)
	void makePretokGroup(const FlatModel& m, const char16_t* text, size_t len, uint64_t match, const std::vector<PtSpan>& spans, PretokGroup& out);
}
