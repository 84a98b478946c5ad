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

	class ResultBuilder
	{
		const FlatModel& mdl;
		std::vector<TokenResult> ret;
		std::vector<uint8_t> spStatesByRet;
		const std::vector<uint32_t>* positionTable = nullptr;
		std::vector<uint16_t> wordPositions;
		size_t topN; uint64_t match; bool integrateAllomorph;
	public:
		ResultBuilder(const FlatModel& m, size_t _topN, uint64_t _match, bool _integrateAllomorph)
			: mdl(m), topN(_topN), match(_match), integrateAllomorph(_integrateAllomorph) {}
		void begin(const char16_t* raw, size_t n, const std::vector<uint32_t>& positionTable);
		const std::vector<uint8_t>& spStates() const { return spStatesByRet; }
		void insertPaths(const std::vector<PathResult>& paths);
		std::vector<TokenResult> finish(const char16_t* raw, size_t n);
	};
}
