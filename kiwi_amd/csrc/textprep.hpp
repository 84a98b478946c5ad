// Host-side text preparation for the batched lattice kernels: normalise, type every character, run the
// pattern recognisers and cut the text into independently analysable chunks.
//   Kiwi::analyze prologue  /root/reference/src/Kiwi.cpp:1028-1056
//   Splitter::preparePattern /root/reference/src/KTrie.cpp:766-858
//   matchPattern            /root/reference/src/PatternMatcher.cpp:366-384
#pragma once
#include <utility>
#include <cstdint>
#include <string>
#include <vector>
#include "hostutil.hpp"

namespace kamd
{
	uint8_t chr2ScriptType(uint32_t c);           // src/ScriptType.cpp:5-333 (range table, see unicode_tables.inc)
	int isEmoji(uint32_t c0, uint32_t c1);        // src/ScriptType.cpp:569-752
	const char* scriptName(uint8_t script);
	// returns (matched length, tag) ; tag T_UNKNOWN when nothing matched
	std::pair<size_t, uint8_t> matchPattern(char16_t left, const char16_t* first, const char16_t* last, uint64_t matchOptions);

	struct PatternSpan { uint32_t end, length; uint8_t tag; };   // chunk-relative, sorted by (end, length, tag)

	// One chunk = the unit one lattice is built for (the reference's per-call splitByTrie range).
	struct ChunkDesc
	{
		uint32_t textId;
		uint32_t startOffset;   // offset of the chunk in the normalised text
		uint32_t nChars;        // chunk length in UTF-16 units (stop position)
		uint32_t nextOffset;    // where the next chunk starts (splitEnd)
		uint32_t patBegin, patEnd;
		bool empty;             // no non-space character: yields no lattice (Kiwi.cpp:1119)
	};

	struct PreparedText
	{
		U16 norm;                         // normalised string
		std::vector<uint32_t> position;   // norm index -> raw index table (size raw+1), StrUtils.h:494-521
		std::vector<uint8_t> cls;         // per norm unit: low 6 bits character type (POSTag), bit 7 = starts an emoji
		std::vector<uint8_t> script;      // per norm unit: ScriptType id
		std::vector<ChunkDesc> chunks;
		std::vector<PatternSpan> patterns;
	};

	void normalizeWithPosition(const char16_t* s, size_t n, U16& out, std::vector<uint32_t>& pos);
	void normalizeCoda(U16& s);
	void normalizeCoda(char16_t* s, size_t n);
	// Fills everything in `out` for one raw text.
	// spans: pretokenized spans as [begin, end) offsets into the NORMALISED text, ascending (Kiwi::analyze maps the caller's through the position table,
	// src/Kiwi.cpp:817-818): the chunk cut steps over them (KTrie.cpp:782-790) -- no pattern starts inside one, no chunk ends inside one
	void prepareText(PreparedText& out, const char16_t* raw, size_t n, uint64_t matchOptions, uint32_t textId, const std::pair<uint32_t, uint32_t>* spans = nullptr, size_t nSpans = 0);

	// The batch path prepares runs of consecutive texts into ONE set of flat arrays (a block is filled by one host worker): six
	// allocations per block instead of six per text, which is what the preparation of an 8k-sentence batch otherwise spends its time on.
	template<class T> struct Span
	{
		const T* p = nullptr; size_t n = 0;
		const T* data() const { return p; }
		size_t size() const { return n; }
		const T& operator[](size_t i) const { return p[i]; }
		const T* begin() const { return p; }
		const T* end() const { return p + n; }
	};
	struct PreparedView   // what PreparedText holds, as views into a PrepBlock
	{
		Span<char16_t> norm; Span<uint32_t> position; Span<uint8_t> cls, script; Span<ChunkDesc> chunks; Span<PatternSpan> patterns;
		U16 normSubstr(size_t off, size_t len) const { return U16{ norm.p + off, len }; }
	};
	struct PrepBlock
	{
		U16 norm; std::vector<uint32_t> position; std::vector<uint8_t> cls, script; std::vector<ChunkDesc> chunks; std::vector<PatternSpan> patterns;
		struct Idx { size_t normOff, normLen, posOff, posLen, chunkOff, nChunks, patOff, nPat; };
		std::vector<Idx> idx;
		void append(const char16_t* raw, size_t n, uint64_t matchOptions, uint32_t textId, const std::pair<uint32_t, uint32_t>* spans = nullptr, size_t nSpans = 0);
		PreparedView view(size_t k) const
		{
			const Idx& x = idx[k]; PreparedView v;
			v.norm = { norm.data() + x.normOff, x.normLen }; v.position = { position.data() + x.posOff, x.posLen };
			v.cls = { cls.data() + x.normOff, x.normLen }; v.script = { script.data() + x.normOff, x.normLen };
			v.chunks = { chunks.data() + x.chunkOff, x.nChunks }; v.patterns = { patterns.data() + x.patOff, x.nPat };
			return v;
		}
	};
}
