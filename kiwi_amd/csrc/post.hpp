// Host-side result assembly: per-chunk best paths -> token lists of a text.
//   insertPathIntoResults  /root/reference/src/Kiwi.cpp:615-783
//   joinAffixTokens        /root/reference/src/Kiwi.cpp:495-588
//   fillPairedTokenInfo    /root/reference/src/Kiwi.cpp:98-143
//   fillSentLineInfo       /root/reference/src/Kiwi.cpp:325-415 (+ SentenceParser :145-312)
//   getWordPositions / allNewLinePositions  /root/reference/src/Kiwi.cpp:465-487, 70-96
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>
#include "flat_model.hpp"
#include "hostutil.hpp"

namespace kamd
{
	struct Token   // mirrors kiwi::TokenInfo (include/kiwi/Types.h:344-384); morph is an id (-1: none)
	{
		U16 str;
		uint32_t position = 0, wordPosition = 0, sentPosition = 0, lineNumber = 0;
		uint16_t length = 0;
		uint8_t tag = 0;
		uint8_t senseId = 0;   // aliases `script` for SL/SH/SW/W_EMOJI tokens, as in the reference's union
		float score = 0, typoCost = 0;
		uint32_t typoFormId = 0, pairedToken = (uint32_t)-1, subSentPosition = 0;
		uint16_t dialect = 0;
		int32_t morph = -1;
		uint32_t endPos() const { return position + length; }
	};
	using TokenResult = std::pair<std::vector<Token>, float>;

	struct PathTok   // mirrors kiwi::PathNode (src/PathEvaluator.h:33-68)
	{
		uint32_t morph = 0;
		U16 str;               // own (out-of-dictionary) surface form, empty when the morpheme's form is used
		uint32_t begin = 0, end = 0;   // offsets in the normalised text
		float wordScore = 0, typoCost = 0;
		uint32_t typoFormId = 0, nodeId = 0;
	};
	struct PathResult { std::vector<PathTok> path; float score = 0; uint8_t prevState = 0, curState = 0; };

	// Packed token record as the C ABI hands it out (include/kiwi_amd.h kamd_token_t; its first 44 bytes are the reference's
	// kiwi_token_info_t, include/kiwi/capi.h:43-61, so that kiwi_res_token_info can return a pointer into it).
	struct FlatToken
	{
		uint32_t position, wordPosition, sentPosition, lineNumber;
		uint16_t length; uint8_t tag; uint8_t senseOrScript;
		float score, typoCost;
		uint32_t typoFormId, pairedToken, subSentPosition;
		uint16_t dialect; uint16_t formLen;
		int32_t morph;
		uint64_t formOff;      // offset of the NUL-terminated UTF-16 form in the segment's `forms`
	};
	static_assert(sizeof(FlatToken) == 56, "FlatToken");

	// Results of a run of consecutive texts, flat: analyses of text t are textAna[t] .. textAna[t+1], tokens of analysis a are
	// anaTok[a] .. anaTok[a+1].  One segment is filled by one host worker, without per-token allocations.
	struct ResultSegment
	{
		std::vector<uint32_t> textAna{ 0 }, anaTok{ 0 };
		std::vector<float> anaScore;
		std::vector<FlatToken> toks;
		std::vector<char16_t> forms;
		size_t texts() const { return textAna.size() - 1; }
		void appendText(const std::vector<TokenResult>& analyses);
	};

	class ResultBuilder
	{
		const FlatModel& mdl;
		std::vector<TokenResult> ret;
		std::vector<uint8_t> spStatesByRet;
		const uint32_t* positionTable = nullptr; size_t positionLen = 0;
		std::vector<uint16_t> wordPositions;
		std::vector<size_t> parentMap;      // (scratch of insertPaths: a builder is reused for the texts of a task, its vectors keep their capacity)
		size_t topN; uint64_t match; bool integrateAllomorph;
	public:
		ResultBuilder(const FlatModel& m, size_t _topN, uint64_t _match, bool _integrateAllomorph)
			: mdl(m), topN(_topN), match(_match), integrateAllomorph(_integrateAllomorph) {}
		void begin(const char16_t* raw, size_t n, const uint32_t* positionTable, size_t positionLen);
		const std::vector<uint8_t>& spStates() const { return spStatesByRet; }
		void insertPaths(const std::vector<PathResult>& paths);
		std::vector<TokenResult> finish(const char16_t* raw, size_t n);
	};
}
