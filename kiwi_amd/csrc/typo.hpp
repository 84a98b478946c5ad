// Typo correction, host side: the rule container, its preparation into a pattern automaton, and the typo
// graph of a normalised chunk -- the input of the typo-aware lattice build.  Reference behaviour reproduced:
//   TypoTransformer::addTypo / update / scaleCost        /root/reference/src/TypoTransformer.cpp:224-373
//   PreparedTypoTransformer (prepare)                    src/TypoTransformer.cpp:375-486
//   PreparedTypoTransformer::generateGraph               src/TypoTransformer.cpp:594-628, 811-1049
// The lattice over such a graph is built by typo_lattice_kernel.hip; tests: tests/test_typo_product.py (host), tests/test_gpu_typo.py (device).
//
// Layout for the device (flat arrays, uploaded as they are): the pattern automaton is a CSR trie like the form trie (node = edge range,
// failure link, pattern id or "a shorter pattern ends here" mark; sorted u16 keys; children), patterns index a replacement table, every
// replacement names a span of one UTF-16 pool.  Graph nodes are 28-byte records whose forms are spans of the chunk text or of that pool.
#pragma once
#include <cmath>
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

namespace kamd
{
	enum TypoCond : uint8_t { TC_NONE, TC_ANY, TC_VOWEL, TC_VOCALIC, TC_VOCALIC_H, TC_NON_VOWEL, TC_NON_VOCALIC, TC_NON_VOCALIC_H, TC_APPLOSIVE, TC_CONTINUAL, TC_BOUNDARY };   // CondVowel

	class TypoTransformer
	{
	public:
		struct Key
		{
			std::u16string orig, error; uint8_t cond; uint16_t dialect;
			bool operator==(const Key& o) const { return orig == o.orig && error == o.error && cond == o.cond && dialect == o.dialect; }
		};
		// Hash<std::tuple<KString, KString, CondVowel, Dialect>> of the reference (include/kiwi/Types.h:499-518).  It has to be THIS hash over THIS
		// container: prepare() walks the rules in the map's iteration order, which decides the order of a pattern's replacements.
		struct KeyHash { size_t operator()(const Key& k) const; };
		using Map = std::unordered_map<Key, float, KeyHash>;

		void add(const std::u16string& orig, const std::u16string& error, float cost = 1.f, uint8_t cond = TC_NONE, uint16_t dialect = 0);   // addTypo: throws std::invalid_argument
		void addEntry(const std::u16string& orig, const std::u16string& error, float cost, uint8_t cond, uint16_t dialect);                    // one entry of update()
		void update(const TypoTransformer& o);
		void scaleCost(float scale);
		TypoTransformer withDialect(uint16_t dialect) const;      // copyWithDialectOverriding (src/TypoTransformer.cpp:325-336)
		void setContinualCost(float c) { continualCost_ = c; }
		void setLengtheningCost(float c) { lengtheningCost_ = c; }
		float continualCost() const { return continualCost_; }
		float lengtheningCost() const { return lengtheningCost_; }
		const Map& rules() const { return typos_; }
		bool empty() const { return typos_.empty() && !std::isfinite(continualCost_) && !std::isfinite(lengtheningCost_); }

	private:
		void addWithCond(const std::u16string& orig, const std::u16string& error, float cost, uint8_t cond, uint16_t dialect);
		void addNormalized(const std::u16string& orig, const std::u16string& error, float cost, uint8_t cond, uint16_t dialect);
		Map typos_;
		float continualCost_ = INFINITY, lengtheningCost_ = INFINITY;
	};

	// Kiwi's built-in typo sets (reference getDefaultTypoSet, src/TypoTransformer.cpp:1058-1254; DefaultTypoSet ids 0..6 = capi.h:485-491):
	// assembled once from the rule tables of typo_sets.inc; throws std::invalid_argument for any other id.  The objects live for the process.
	const TypoTransformer& defaultTypoSet(int set);
	class PreparedTypo;
	// getDefaultPreparedTypoSet(DefaultTypoSet::dialect) (src/TypoTransformer.cpp:1263, 1278): what an analysis with allowed dialects and no transformer of
	// its own is corrected with (src/Kiwi.cpp:1037-1041); prepared once per process
	const PreparedTypo& defaultDialectTypo();

	struct TypoGraphNode      // TypoGraphNode of the reference (include/kiwi/TypoTransformer.h:131-157), form as a span
	{
		uint32_t formOff;      // bit 31 set: offset into the prepared transformer's pool, else into the chunk text
		uint32_t formLen;
		uint32_t endPos;
		float typoCost;
		uint32_t prevOffset, siblingOffset;
		uint8_t continualTypoIdx; uint8_t pad; uint16_t dialect;
	};
	static_assert(sizeof(TypoGraphNode) == 28, "TypoGraphNode");
	constexpr uint32_t TYPO_FORM_IN_POOL = 0x80000000u;

	class PreparedTypo
	{
	public:
		struct TrieNode { uint32_t edgeOff; uint16_t numNexts, depth; int32_t fail; int32_t pattern; };   // pattern: id, -1 none, -2 a shorter pattern ends here
		struct Pattern { uint32_t replOff, replCnt, patLength; };
		struct Repl { uint32_t strOff, strLen; float cost; uint8_t cond; uint8_t pad; uint16_t dialect; };

		PreparedTypo(const TypoTransformer& tt, bool inverse);
		float continualCost() const { return continualCost_; }
		float lengtheningCost() const { return lengtheningCost_; }
		bool ready() const { return !repls_.empty(); }

		// generateGraph over the normalised text str[0..n); returns max(continual typo index) + 1 over the match clusters (>= 0; 0 = none seen)
		size_t graph(const char16_t* str, size_t n, uint16_t allowedDialect, std::vector<TypoGraphNode>& out) const;
		std::u16string formOf(const TypoGraphNode& g, const char16_t* str) const;

		// {type, script} of the last character of a graph node's form as the lattice build wants them (type 0xFF: empty form or NUL = "none"):
		// identifySpecialChr / chr2ScriptType of the last code point, surrogate pairs merged inside the form
		void lastOf(const TypoGraphNode& g, const char16_t* str, uint8_t out[2]) const;

		// bounds the device graph kernel (typo_graph_kernel.hip) sizes its working regions with: patterns that can end at ONE text position
		// (the longest chain of pattern-bearing suffixes in the automaton), and max(continual typo index) + 1 over any cluster
		uint32_t matchesPerEndBound() const { return matchesPerEnd_; }
		uint32_t maxCtiBound() const { return maxCtiBound_; }
		// entries of the small per-chunk working arrays: a cluster's matches + break points, the end-position index (nChars + 2), and in the
		// write pass one slot per graph node
		uint32_t scratchCapFor(uint32_t nChars, uint32_t graphCnt) const { const uint32_t a = nChars * matchesPerEnd_ + 4, b = nChars + 4; const uint32_t m = a > b ? a : b; return m > graphCnt ? m : graphCnt; }
		int32_t entryNode() const { return step(0, 0); }
		// per replacement 6 bytes: {type, script} of the last character of the whole string, of its first unit alone, of the string without its first unit
		const std::vector<uint8_t>& replLast() const { return replLast_; }

		// flat tables (device upload)
		const std::vector<TrieNode>& trie() const { return trie_; }
		const std::vector<uint16_t>& trieKeys() const { return keys_; }
		const std::vector<uint32_t>& trieChildren() const { return children_; }
		const std::vector<Pattern>& patterns() const { return pats_; }
		const std::vector<Repl>& replacements() const { return repls_; }
		const std::u16string& pool() const { return pool_; }

	private:
		int32_t step(int32_t node, char16_t c) const;
		std::vector<TrieNode> trie_; std::vector<uint16_t> keys_; std::vector<uint32_t> children_;
		std::vector<Pattern> pats_; std::vector<Repl> repls_; std::u16string pool_;
		std::vector<uint8_t> replLast_; uint32_t matchesPerEnd_ = 0, maxCtiBound_ = 1;
		float continualCost_ = INFINITY, lengtheningCost_ = INFINITY;
	};

	// FeatureTestor::isMatched(begin, end, CondVowel) (src/FeatureTestor.cpp:6-60) on the prefix [0, n) of s -- host and device
#if defined(__HIPCC__) || defined(__HIP__)
#define KAMD_TYPO_HD __host__ __device__ inline
#else
#define KAMD_TYPO_HD inline
#endif
	KAMD_TYPO_HD bool typoLeftCondMatched(const uint16_t* s, size_t n, uint8_t cond)
	{
		if (cond == TC_NONE) return true;
		if (n == 0) return false;
		if (cond == TC_ANY) return true;
		const uint16_t l = s[n - 1];
		if (cond == TC_APPLOSIVE)
		{
			switch (l) { case 0x11A8: case 0x11A9: case 0x11AA: case 0x11AE: case 0x11B8: case 0x11B9: case 0x11BA: case 0x11BB: case 0x11BD: case 0x11BE: case 0x11BF: case 0x11C0: case 0x11C1: return true; default: return false; }
		}
		if (!(0xAC00 <= l && l <= 0xD7A4) && !(0x11A8 <= l && l <= 0x11C2)) return true;
		const bool coda = 0x11A8 <= l && l <= 0x11C2;
		switch (cond)
		{
		case TC_VOCALIC_H: if (l == 0x11C2) return true; [[fallthrough]];
		case TC_VOCALIC: if (l == 0x11AF) return true; [[fallthrough]];
		case TC_VOWEL: return !coda;
		case TC_NON_VOCALIC_H: if (l == 0x11C2) return false; [[fallthrough]];
		case TC_NON_VOCALIC: if (l == 0x11AF) return false; [[fallthrough]];
		case TC_NON_VOWEL: return !(0xAC00 <= l && l <= 0xD7A4);
		default: return false;
		}
	}
}
