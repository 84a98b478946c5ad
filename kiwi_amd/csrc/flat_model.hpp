// Flat ("baked") model: every pointer of the reference's in-memory Kiwi object replaced by an index,
// laid out as plain arrays that are uploaded verbatim to HBM.
//   forms/morphemes : /root/reference/include/kiwi/Form.h:142-257 (Morpheme 40 B, Form 56 B, pointer-rich)
//   form trie       : /root/reference/include/kiwi/FrozenTrie.h:69-158 (Aho-Corasick automaton)
//   Knlm            : /root/reference/include/kiwi/Knlm.h:17-24 + src/Knlm.hpp:1003-1167
// The same POD views (DeviceModel) are used by the HIP kernels and, with host pointers, by host code.
#pragma once
#include <cstdint>
#include <cmath>
#include <string>
#include <vector>
#include "kchars.hpp"

namespace kamd
{
	// ---- form record (16 B) ---------------------------------------------------------------------
	enum FormFlag : uint8_t
	{
		FF_ZCODA_APPENDABLE = 1, FF_ZSIOT_APPENDABLE = 2, FF_HAS_JCLASS = 4, FF_HAS_ANY_FULL = 8,
		FF_FIRST_IS_CODA = 16,   // isHangulCoda(form[0])            (KTrie.cpp:969)
		FF_IS_STAG = 32,         // single special char sf..sw        (KTrie.cpp:971-972)
		FF_STARTS_WITH_A = 64,   // form[0] == U+C544 '아'            (PathEvaluator.hpp:104)
		FF_ENDS_WITH_SSC = 128,  // identifySpecialChr(form.back()) == ssc (PathEvaluator.hpp:289, own-form case)
	};
	struct FormRec
	{
		uint32_t candOff;    // into formCand[]
		uint32_t charOff;    // into formChars[] (form string incl. spaces)
		uint16_t candCnt;
		uint8_t len;         // form.size()
		uint8_t numSpaces;
		uint8_t flags;       // FormFlag
		uint8_t flags2;      // FormFlag2
		uint8_t vowelPolar;  // CondVowel | CondPolarity << 4 (kept for the dictionary dump; the new splitter does not test them)
		uint8_t formHash;
	};
	enum FormFlag2 : uint8_t
	{
		// every candidate is a partial morpheme (split stem / chunked) and the form is not a lone chunked UNKNOWN-tag
		// candidate: such nodes also get an unknown proper-noun reading (PathEvaluator.hpp:1258-1287)
		FF2_ALL_PARTIAL = 1,
	};
	static_assert(sizeof(FormRec) == 16, "FormRec");

	// ---- morpheme record (32 B) -----------------------------------------------------------------
	enum MorphFlag : uint16_t
	{
		MF_COMPLEX = 1, MF_SAISIOT = 2, MF_SINGLE = 4,      // Form.h:163-174
		MF_HAS_COMPLEX = 8,                                 // Morpheme::hasComplex (Form.h:176-185)
		MF_KFORM_EMPTY = 16,
		// current-morpheme side of RuleBasedScorer (PathEvaluator.hpp:88-109)
		MF_VOWEL_E = 32, MF_INF_J = 64, MF_BAD_PAIR_OF_L = 128, MF_CONTRACTABLE_E = 256,
		// "하다/하게/하지" contraction guard (PathEvaluator.hpp:436-446)
		MF_HA_CONTRACTION = 512,
		MF_ENDS_WITH_SSC = 1024,                            // identifySpecialChr(kform.back()) == ssc (PathEvaluator.hpp:289)
		MF_FIRST_WID_IS_P = 2048,                           // morphBase[firstWid].tag == P (PathEvaluator.hpp:604)
		MF_ANY_REST_WID_IS_P = 4096,                        // any chunk i>=1 has tag P (PathEvaluator.hpp:616)
		MF_IN_VOCAB_LAST = 8192,                            // lastMorph index < langVocabSize (PathEvaluator.hpp:551-558)
	};
	enum PrevFlag : uint8_t   // previous-morpheme side of RuleBasedScorer (PathEvaluator.hpp:111-181)
	{
		PF_IRREGULAR = 1, PF_INFLECTENDA_NP = 2, PF_VERB_L = 4, PF_POSITIVE_VERB = 8, PF_VERB_VOWEL = 16,
		PF_VA_OR_XSA = 32, PF_E_NOT_EF = 64, PF_UNK_EF_SF = 128,
	};
	struct MorphRec
	{
		uint32_t lmId;        // lmMorphemeId; the first LM id fed is lmId (single) or chunkLm[chunkOff] (PathEvaluator.hpp:537-548)
		uint32_t lastSeqId;   // wid recorded on the path (PathEvaluator.hpp:550-558)
		uint32_t chunkOff;    // into chunkMorph[] / chunkLm[] / chunkPos[]
		float userScore;
		int32_t combinedId;   // absolute id of getCombined()
		uint16_t flags;       // MorphFlag
		uint16_t feat;        // left-feature mask of kform: bit v = isMatched(kform, CondVowel v), bit 9+p = CondPolarity p
		uint8_t tag, vowel, polar, socket;
		uint8_t nChunks, senseId, prevFlags, special;   // special: SpecialMorph (0..5) or 6 ; sbType in sbInfo[]
	};
	static_assert(sizeof(MorphRec) == 32, "MorphRec");

	struct TrieNodeRec
	{
		uint32_t edgeOff;
		uint16_t numNexts, depth;
		int32_t fail;        // absolute node index, -1 for root
		int32_t value;       // form id, TRIE_NONE, or TRIE_SUBMATCH
	};
	static_assert(sizeof(TrieNodeRec) == 16, "TrieNodeRec");
	// Edge (node, key) -> child of the form trie as ONE memory round trip: an open-addressing table of 16-byte slots (linear probing, at most a quarter full)
	// over every edge below the root (the root has its direct table).  The sorted keys per node stay (host walks, the bake's fail links): a device walk through
	// them costs 2 + log2(fan-out) DEPENDENT loads per character -- the record, the halving steps, the child --, which is what bounds the dictionary scan and
	// the typo lattice's automaton steps.
	struct TrieEdgeSlot { uint32_t node, key, child; int32_t value; };      // value: the child's TrieNodeRec::value (a walk that only asks "does a form end here" needs no second load)
	static_assert(sizeof(TrieEdgeSlot) == 16, "TrieEdgeSlot");
	constexpr uint32_t TRIE_EDGE_EMPTY = 0xFFFFFFFFu;
	KAMD_HD uint32_t trieEdgeHash(uint32_t node, uint32_t key) { uint32_t h = node * 0x9E3779B1u + key * 0x85EBCA6Bu; h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 13; return h; }
	constexpr int32_t TRIE_NONE = -1, TRIE_SUBMATCH = -2;
	// bits 13..15 of a left-feature mask (bits 0..12: feature.hpp featMask)
	constexpr uint16_t LF_STR_SSC = 1u << 13, LF_PREV_ZSIOT = 1u << 14, LF_TAG_SSC = 1u << 15;

	// Knlm edge hash (device lookup structure): bucket = 4 slots = 64 B; a lookup reads one bucket and, only when it
	// is full, the next one.  Slot of edge (node, wid): value as in lmValues plus the log-likelihood the edge yields
	// (child node's ll for value > 0, the leaf ll otherwise), so a hit needs no second load.
	struct LmSlot { uint32_t node, wid; int32_t value; float ll; };
	static_assert(sizeof(LmSlot) == 16, "LmSlot");
	constexpr uint32_t LM_SLOT_EMPTY = 0xFFFFFFFFu;
	struct LmRootRec { int32_t value; float ll; };      // root direct table entry: value 0 = unseen word
	struct LmBackoff { int32_t lower; float gamma; };   // per node: what a miss needs
	KAMD_HD uint32_t lmHashOf(uint32_t node, uint32_t wid) { uint32_t h = node * 0x9E3779B1u ^ (wid * 0x85EBCA77u); h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 13; return h; }

	struct LmNodeRec
	{
		uint32_t nextOff, numNexts;
		int32_t lower;       // relative, as in the reference
		float ll, gamma;
	};
	static_assert(sizeof(LmNodeRec) == 20, "LmNodeRec");

	struct ModelHeader
	{
		uint32_t nForms, nMorphs, vocabSize, nTrieNodes, nTrieEdges, nLmNodes, nLmEdges;
		int32_t bosNode;
		float unkLl;
		uint32_t specialMorph[6];
		uint32_t maxFormLen;
		uint32_t lmOrder;
		uint32_t lmKeyBytes;   // key width of the reference's in-memory Knlm (2 or 4): only used by ALG_BYTES accounting
	};

	// Pointers into one arena (host or device).
	struct ModelView
	{
		ModelHeader h;
		const FormRec* forms;
		const uint16_t* formChars;
		const uint32_t* formCand;
		const MorphRec* morphs;
		const uint32_t* chunkMorph;   // chunk morpheme ids
		const uint32_t* chunkLm;      // chunks[i]->lmMorphemeId
		const uint8_t* chunkPos;      // (begin,end) pairs
		const uint8_t* sbInfo;        // per morph: sbType (0 for non-SB), see getSBType (src/Utils.cpp:264-298)
		const uint32_t* morphPath;    // per morph: what a path ending in it exposes to its successor: leftFeat (low 16) | prevFlags << 16
		const TrieNodeRec* trie;
		const uint16_t* trieKeys;     // sorted per node
		const uint32_t* trieChild;    // absolute
		const uint32_t* trieRoot;     // direct table [65536]: child of the root for each UTF-16 unit, 0 if none
		const LmNodeRec* lmNodes;
		const uint32_t* lmKeys;       // sorted per node
		const int32_t* lmValues;      // >0 relative child offset, <=0 leaf ll float bits
		const int32_t* lmRoot;        // all_value_data: direct table [vocab]
		const LmSlot* lmHash; uint32_t lmHashMask;   // bucket index mask (buckets of 4 slots)
		const LmRootRec* lmRoot2;     // [vocab]
		const LmBackoff* lmBackoff;   // [nLmNodes]
		const void* unkPacks;         // device only: CandStatic[2] for the unknown-noun candidates NNG, NNP (PathEvaluator.hpp:1204-1206)
		// history-transformed Knlm (KnLangModelHeader::htx_offset; the reference's builder writes one by default): per word the root's child for its
		// TRANSFORMED id (a node index, 0 = none) -- where a walk lands when no context continues with the word (Knlm.cpp:61-70, 116-126); null = plain model
		const int32_t* lmHtxNode;
		// character model present: per form the score of its own string (what a dictionary node's unknown proper-noun reading costs under Match::oovChrModel)
		const float* formUnkChr;
		// ... and the character model's token of every unit of formChars (Match::oovChrFreqModel walks a dictionary form's string per node: chr_freq.hpp)
		const uint16_t* formChrTok;
		// Dialect bits per form / per morpheme (null: the model has no dialect morphemes).  morphDialect: a morpheme whose dialect is neither standard nor allowed
		// is no candidate (PathEvaluator.hpp:386, 893: blockBits per batch), an allowed one costs dialectCost.  formDialect is a BAKE-time fact only -- a form of
		// dialects that are not enabled stays out of the trie (KiwiBuilder.cpp:2501-2504); the splitter this library restates (flushCandidates, KTrie.cpp:955-996)
		// does not test a form's dialect when it builds the lattice (the test of KTrie.cpp:207-229 belongs to insertCandidates, the legacy splitter behind
		// useOldSplitter, which is refused): the pointer stays null on the device
		const uint16_t* formDialect; const uint16_t* morphDialect;
		// device only (Knlm): per LM node the next two nodes of its back-off chain as absolute ids {node + lower, that node's lower node}; 0 = the root,
		// where every chain ends.  A search state that carries this pair can probe all three contexts of its next transition at once (viterbi_pos.inc)
		const uint32_t* lmChain;
		// form-trie edges as a hash (TrieEdgeSlot): slot (trieEdgeHash(node, key) + i) & trieEdgeMask, i = 0, 1, ... until the edge or an empty slot
		const TrieEdgeSlot* trieEdges; uint32_t trieEdgeMask;
	};

	// SkipBigram tables (reference src/SkipBigramModel.hpp:40-105), kept apart from ModelView: only the CPU restatement uses
	// them so far (the device scoring of this model type is a later row).
	struct SbgView
	{
		uint32_t vocabSize = 0, windowSize = 0;
		const uint32_t* ptrs = nullptr;        // [vocab + 1] into keys / comps
		const uint32_t* keys = nullptr;        // history word ids, sorted per `next` word
		const float* comps = nullptr;          // compensation per key
		const float* discnts = nullptr;        // [vocab]
		const uint8_t* valid = nullptr;        // [vocab]
		float logWindowSize = 0;
		bool present() const { return vocabSize != 0; }
	};

	// CoNgram model, local (window 0) and quantised (reference src/CoNgramModel.cpp: ModelType::cong): a context trie -- same shape as the Knlm
	// trie; an edge leads to a child context node (value > 0: relative offset) or to a leaf (value < 0: minus its context id); every node
	// names the context id its history maps to -- plus two int8 embedding tables.  score(context c, word w) =
	// float(sum_k ctx[c][k] * out[w][k]) * ctxScale[c] * outScale[w] + ctxBias[c]   (CoNgramModel.cpp:886-894; the reference stores the context
	// row biased by +128 and subtracts 128 * sum(out[w]) again: the same integer).  Rows: dim x s8, then f32 scale, f32 bias (context) /
	// f32 scale, 4 unused bytes (output): stride dim + 8.
	struct CongNodeRec { uint32_t nextOff, numNexts; int32_t lower; uint32_t value; };
	struct CongView
	{
		uint32_t dim = 0, stride = 0, nCtx = 0, vocabSize = 0;
		const uint8_t* ctxEmb = nullptr; const uint8_t* outEmb = nullptr;
		// host-side walk (oracle, bake); the device walks the edge hash that is uploaded in place of the Knlm one
		const CongNodeRec* nodes = nullptr; const uint32_t* keys = nullptr; const int32_t* values = nullptr; const int32_t* root = nullptr;
		uint32_t rootSize = 0;
		// variable-length keys (cong.mdl keySize 3): a word id >= vlTMax is spelt as the two trie keys vlTMax + (r >> vlBits) and
		// vlTMax + (1 << vlBits) + (r & mask), r = id - vlTMax (CoNgramModel::progressContextNode, src/CoNgramModel.hpp:271-300); 0xFFFFFFFF = no such ids
		uint32_t vlTMax = 0xFFFFFFFFu, vlBits = 0;
		// the global model (ModelType::congGlobal; reference window 7): the sections a cong.mdl with windowSize > 0 carries on top.  A word whose bit is
		// set in distMask ("valid distant token") is scored as a mixture over the context and the last `window` such words of the path
		// (CoNgramModel::progress, src/CoNgramModel.cpp:802-868); window == 0: no such sections / local scoring asked for.
		uint32_t window = 0;
		uint32_t keyBytes = 4;                 // sizeof(KeyType) of the reference's instantiation (keySize 2 -> 2, else 4): Hash<CoNgramState> reads 8 bytes of history
		const float* ctxConf = nullptr;        // [nCtx][2]: confidence, valid-token sum
		const uint8_t* distEmb = nullptr;      // [vocab] rows like ctxEmb: dim x s8, f32 scale, f32 bias
		const float* distConf = nullptr;       // [vocab]
		const float* posConf = nullptr;        // [window + 1], [0] = 0
		const uint8_t* distMask = nullptr;     // (vocab + 7) / 8 bytes
		bool present() const { return dim != 0; }
		KAMD_HD bool distant(uint32_t w) const { return window && (distMask[w >> 3] >> (w & 7) & 1); }
	};
	// the score of word `w` in context `c` (one fp32 conversion, two multiplications, one addition, in this order: CoNgramModel.cpp:886-894)
	KAMD_HD float congScore(const CongView& C, uint32_t c, uint32_t w)
	{
		const int8_t* a = reinterpret_cast<const int8_t*>(C.ctxEmb + (size_t)c * C.stride);
		const int8_t* b = reinterpret_cast<const int8_t*>(C.outEmb + (size_t)w * C.stride);
		int32_t acc = 0;
		for (uint32_t k = 0; k < C.dim; ++k) acc += (int32_t)a[k] * (int32_t)b[k];
		float cs, os, bias;
		__builtin_memcpy(&cs, a + C.dim, 4); __builtin_memcpy(&bias, a + C.dim + 4, 4); __builtin_memcpy(&os, b + C.dim, 4);
		return (float)acc * cs * os + bias;
	}

	// the same with the output scale multiplied in first: the rounding of the reference's batched SSE4.1 kernel (src/archImpl/sse4_1.cpp:116)
	KAMD_HD float congScoreOutputFirst(const CongView& C, uint32_t c, uint32_t w)
	{
		const int8_t* a = reinterpret_cast<const int8_t*>(C.ctxEmb + (size_t)c * C.stride);
		const int8_t* b = reinterpret_cast<const int8_t*>(C.outEmb + (size_t)w * C.stride);
		int32_t acc = 0;
		for (uint32_t k = 0; k < C.dim; ++k) acc += (int32_t)a[k] * (int32_t)b[k];
		float cs, os, bias;
		__builtin_memcpy(&cs, a + C.dim, 4); __builtin_memcpy(&bias, a + C.dim + 4, 4); __builtin_memcpy(&os, b + C.dim, 4);
		return (float)acc * os * cs + bias;
	}

	// Character-level CoNgram model for unknown-form scoring (reference nounchr.mdl; Match::oovChrModel, src/UnkFormScorer.cpp:53-66): the local quantised
	// CoNgram step over a trie with BYTE keys -- a (reordered) token id >= 192 is spelt as two bytes, 192 + (r >> 5) and 224 + (r & 31)
	// (CoNgramModel::progressContextNode, src/CoNgramModel.hpp:271-300, VlKeyType = uint8_t) --, an output bias per token, and the token reordering.
	// One source for the host (bake: per-form scores), the oracle and the device kernel (k_unk_chr).
	struct ChrView
	{
		uint32_t dim = 0, stride = 0, nCtx = 0, vocab = 0;
		const uint8_t* ctxEmb = nullptr;      // nCtx rows: dim x s8, f32 scale, f32 bias
		const uint8_t* outEmb = nullptr;      // vocab rows: dim x s8, f32 scale, f32 output bias
		const CongNodeRec* nodes = nullptr; const uint8_t* keys = nullptr; const int32_t* values = nullptr;
		const int32_t* root = nullptr;        // [256]
		const uint16_t* inv = nullptr;        // [vocab]: token -> key space (hasReorderedVocab), or null
		int32_t bosNode = 0; uint32_t bosCtx = 0;      // the state after <s> (UnkFormScorer's constructor)
		// Match::oovChrFreqModel (chr_freq.hpp): the trie values as the walk returns them carry a quantised frequency in their top byte
		// (header flag hasTrieFrequency; CoNgramModel::getContextFrequency -> dequantizeFrequencyScale, src/CoNgramModel.hpp:34-39, 76-86: freqTab);
		// depth[node] = tokens on the path from the root (CoNgramModel::getNodeDepth; assigned breadth first, src/CoNgramModel.cpp:551-572:
		// the second byte of a two-byte spelling does not count)
		const uint16_t* depth = nullptr; const float* freqTab = nullptr; uint32_t bosCtxPacked = 0; bool hasFreq = false;
		bool present() const { return dim != 0; }
	};
	KAMD_HD bool chrSearch(const ChrView& C, const CongNodeRec& nd, uint32_t key, int32_t& v)
	{
		uint32_t lo = 0, hi = nd.numNexts;
		const uint8_t* k = C.keys + nd.nextOff;
		while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (k[mid] < key) lo = mid + 1; else hi = mid; }
		if (lo == nd.numNexts || k[lo] != key) return false;
		v = C.values[nd.nextOff + lo];
		return v != 0;
	}
	// progressContextNodeVl (src/CoNgramModel.hpp:306-385) with byte keys
	KAMD_HD uint32_t chrContextVl(const ChrView& C, int32_t& nodeIdx, uint32_t key)
	{
		for (;;)
		{
			int32_t v;
			const CongNodeRec* node = &C.nodes[nodeIdx];
			if (nodeIdx != 0)
			{
				if (!chrSearch(C, *node, key, v))
				{
					if (!node->lower) return 0;
					nodeIdx += node->lower;
					continue;
				}
			}
			else
			{
				v = C.root[key & 255];
				if (v == 0) return 0;
			}
			if (v > 0) { nodeIdx += v; return C.nodes[nodeIdx].value; }
			while (node->lower)
			{
				node += node->lower;
				int32_t lv;
				if (node != C.nodes)
				{
					if (chrSearch(C, *node, key, lv) && lv > 0) { nodeIdx = (int32_t)(node + lv - C.nodes); return (uint32_t)-v; }
				}
				else
				{
					lv = C.root[key & 255];
					if (lv > 0) { nodeIdx = lv; return (uint32_t)-v; }
				}
			}
			nodeIdx = 0;
			return (uint32_t)-v;
		}
	}
	// CoNgramModel::progressOneStep -> progress(), window 0, quantised (src/CoNgramModel.cpp:869-908): score of `tok` in the current context (one
	// fp32 conversion, two multiplications, the context bias, then the output bias), then the context moves on.  chrProgressPacked keeps the context id as the
	// trie holds it (a frequency in the top byte where the file has them); chrProgress: the UNPACKED id (CoNgramModel::unpackContextId)
	KAMD_HD float chrProgressPacked(const ChrView& C, int32_t& node, uint32_t& packedCtx, uint32_t tok)
	{
		const uint32_t ctx = packedCtx & 0x00FFFFFFu;
		const int8_t* a = reinterpret_cast<const int8_t*>(C.ctxEmb + (size_t)ctx * C.stride);
		const int8_t* b = reinterpret_cast<const int8_t*>(C.outEmb + (size_t)tok * C.stride);
		int32_t acc = 0;
		for (uint32_t k = 0; k < C.dim; ++k) acc += (int32_t)a[k] * (int32_t)b[k];
		float cs, os, bias, obias;
		__builtin_memcpy(&cs, a + C.dim, 4); __builtin_memcpy(&bias, a + C.dim + 4, 4); __builtin_memcpy(&os, b + C.dim, 4); __builtin_memcpy(&obias, b + C.dim + 4, 4);
		float ll = (float)acc * cs * os + bias;
		ll += obias;
		uint32_t key = C.inv ? C.inv[tok] : tok;
		uint32_t c;
		if (key < 192) c = chrContextVl(C, node, key);
		else { const uint32_t r = key - 192; chrContextVl(C, node, 192 + (r >> 5)); c = chrContextVl(C, node, 224 + (r & 31)); }
		packedCtx = c;
		return ll;
	}
	KAMD_HD float chrProgress(const ChrView& C, int32_t& node, uint32_t& ctx, uint32_t tok)
	{
		uint32_t packed = ctx;
		const float ll = chrProgressPacked(C, node, packed, tok);
		ctx = packed & 0x00FFFFFFu;
		return ll;
	}
	// ChrTokenizer::encodeOne (src/Dataset.cpp:805-847) of one UTF-16 unit whose identifySpecialChr type is `type`
	KAMD_HD uint32_t chrToken(uint32_t c, uint8_t type)
	{
		if (0xAC00 <= c && c < 0xD7A4) return 10 + (c - 0xAC00) / 28;
		if (0x11A8 <= c && c <= 0x11C2) return 10 + 399 + (c - 0x11A8);
		if (0x21 <= c && c < 0x7F) return 10 + 399 + 27 + (c - 0x21);
		switch (type)
		{
		case T_SF: return 1; case T_SP: return 2; case T_SS: return 3; case T_SSO: return 4; case T_SSC: return 5; case T_SE: return 6; case T_SO: return 7; case T_SH: return 9;
		default: return 8;
		}
	}

	// Host-side owner.
	struct FlatModel
	{
		ModelHeader h{};
		std::vector<FormRec> forms;
		std::vector<uint16_t> formChars;
		std::vector<uint32_t> formCand;
		std::vector<MorphRec> morphs;
		std::vector<uint32_t> chunkMorph, chunkLm;
		std::vector<uint8_t> chunkPos;
		std::vector<uint8_t> sbInfo;
		std::vector<uint32_t> morphPath;
		std::vector<uint32_t> morphKform;    // form id of each morpheme's kform (host-side result building)
		std::vector<uint8_t> morphSenseDialect;
		// Dialect bits (include/kiwi/Types.h:320-335; 0 = standard) per morpheme / per form, EMPTY when the model has no dialect morpheme at all
		std::vector<uint16_t> morphDialect, formDialect;
		std::vector<TrieNodeRec> trie;
		std::vector<uint16_t> trieKeys;
		std::vector<uint32_t> trieChild;
		std::vector<uint32_t> trieRoot;
		std::vector<TrieEdgeSlot> trieEdges; uint32_t trieEdgeMask = 0;      // ModelView::trieEdges (model.cpp buildTrieEdges)
		std::vector<LmNodeRec> lmNodes;
		std::vector<uint32_t> lmKeys;
		std::vector<int32_t> lmValues;
		std::vector<int32_t> lmRoot;
		std::vector<LmSlot> lmHash; uint32_t lmHashMask = 0;
		std::vector<LmRootRec> lmRoot2;
		std::vector<LmBackoff> lmBackoff;
		std::vector<uint32_t> lmHtx; std::vector<int32_t> lmHtxNode;      // history transformer [vocab] and what the walks need of it (ModelView::lmHtxNode)
		std::vector<uint32_t> sbgPtrs, sbgKeys; std::vector<float> sbgComps, sbgDiscnts; std::vector<uint8_t> sbgValid; uint32_t sbgWindow = 0;
		// CoNgram (see CongView): host trie, the device lookup structures in the Knlm shapes (edge hash with the child's context id in the slot's
		// `ll` bits, root table, per-node suffix link), embeddings
		std::vector<CongNodeRec> congNodes; std::vector<uint32_t> congKeys; std::vector<int32_t> congValues, congRoot;
		std::vector<LmSlot> congHash; uint32_t congHashMask = 0; std::vector<LmRootRec> congRoot2; std::vector<LmBackoff> congBackoff;
		// character model of Match::oovChrModel (ChrView) and, per form, the sum of its steps over the form's own string incl. </s> (what a dictionary
		// node's unknown-noun reading is scored with; the analyze-time bias is subtracted where it is used)
		std::vector<CongNodeRec> chrNodes; std::vector<uint8_t> chrKeys; std::vector<int32_t> chrValues, chrRoot; std::vector<uint16_t> chrInv;
		std::vector<uint8_t> chrCtxEmb, chrOutEmb; uint32_t chrDim = 0, chrCtx = 0, chrVocab = 0; int32_t chrBosNode = 0; uint32_t chrBosCtx = 0;
		std::vector<float> formUnkChr;
		// Match::oovChrFreqModel: ChrView::depth / freqTab, and the character model's token of every unit of formChars (the frequency-based score of a
		// dictionary form depends on the text, so the device walks the form's string itself)
		std::vector<uint16_t> chrDepth, formChrTok; std::vector<float> chrFreqTab; uint32_t chrBosCtxPacked = 0; bool chrHasFreq = false;
		std::vector<uint8_t> congCtxEmb, congOutEmb; uint32_t congDim = 0, congCtx = 0, congVocab = 0, congVlTMax = 0xFFFFFFFFu, congVlBits = 0;
		// sections of the global model (CongView::window ...); congGlobal: score with them (ModelType::congGlobal) -- set by whoever opens the model
		std::vector<float> congCtxConf, congDistConf, congPosConf; std::vector<uint8_t> congDistEmb, congDistMask; uint32_t congWindow = 0, congKeyBytes = 4;
		bool congGlobal = false;

		CongView congView() const
		{
			CongView v;
			if (!congDim) return v;
			v.dim = congDim; v.stride = congDim + 8; v.nCtx = congCtx; v.vocabSize = congVocab; v.rootSize = (uint32_t)congRoot.size(); v.vlTMax = congVlTMax; v.vlBits = congVlBits;
			v.ctxEmb = congCtxEmb.data(); v.outEmb = congOutEmb.data();
			v.nodes = congNodes.data(); v.keys = congKeys.data(); v.values = congValues.data(); v.root = congRoot.data();
			v.keyBytes = congKeyBytes;
			if (congGlobal && congWindow)
			{
				v.window = congWindow; v.ctxConf = congCtxConf.data(); v.distEmb = congDistEmb.data(); v.distConf = congDistConf.data();
				v.posConf = congPosConf.data(); v.distMask = congDistMask.data();
			}
			return v;
		}

		ChrView chrView() const
		{
			ChrView v;
			if (!chrDim) return v;
			v.dim = chrDim; v.stride = chrDim + 8; v.nCtx = chrCtx; v.vocab = chrVocab;
			v.ctxEmb = chrCtxEmb.data(); v.outEmb = chrOutEmb.data(); v.nodes = chrNodes.data(); v.keys = chrKeys.data(); v.values = chrValues.data(); v.root = chrRoot.data();
			v.inv = chrInv.empty() ? nullptr : chrInv.data(); v.bosNode = chrBosNode; v.bosCtx = chrBosCtx;
			v.depth = chrDepth.empty() ? nullptr : chrDepth.data(); v.freqTab = chrFreqTab.empty() ? nullptr : chrFreqTab.data(); v.bosCtxPacked = chrBosCtxPacked; v.hasFreq = chrHasFreq;
			return v;
		}

		SbgView sbgView() const
		{
			SbgView v;
			if (sbgPtrs.empty()) return v;
			v.vocabSize = (uint32_t)sbgDiscnts.size(); v.windowSize = sbgWindow;
			v.ptrs = sbgPtrs.data(); v.keys = sbgKeys.data(); v.comps = sbgComps.data(); v.discnts = sbgDiscnts.data(); v.valid = sbgValid.data();
			v.logWindowSize = std::log((float)sbgWindow);
			return v;
		}

		ModelView view() const
		{
			ModelView v;
			v.h = h;
			v.forms = forms.data(); v.formChars = formChars.data(); v.formCand = formCand.data();
			v.morphs = morphs.data(); v.chunkMorph = chunkMorph.data(); v.chunkLm = chunkLm.data(); v.chunkPos = chunkPos.data();
			v.sbInfo = sbInfo.data(); v.morphPath = morphPath.data();
			v.trie = trie.data(); v.trieKeys = trieKeys.data(); v.trieChild = trieChild.data(); v.trieRoot = trieRoot.data();
			v.lmNodes = lmNodes.data(); v.lmKeys = lmKeys.data(); v.lmValues = lmValues.data(); v.lmRoot = lmRoot.data();
			v.lmHash = lmHash.data(); v.lmHashMask = lmHashMask; v.lmRoot2 = lmRoot2.data(); v.lmBackoff = lmBackoff.data();
			v.lmHtxNode = lmHtxNode.empty() ? nullptr : lmHtxNode.data();
			v.formUnkChr = formUnkChr.empty() ? nullptr : formUnkChr.data();
			v.formChrTok = formChrTok.empty() ? nullptr : formChrTok.data();
			v.lmChain = nullptr;
			v.trieEdges = trieEdges.empty() ? nullptr : trieEdges.data(); v.trieEdgeMask = trieEdgeMask;
			v.formDialect = formDialect.empty() ? nullptr : formDialect.data(); v.morphDialect = morphDialect.empty() ? nullptr : morphDialect.data();
			return v;
		}

		std::u16string formStr(uint32_t f) const
		{
			return std::u16string{ (const char16_t*)formChars.data() + forms[f].charOff, forms[f].len };
		}
	};

	// model.cpp
	// enabledDialects: KiwiBuilder's enabledDialects (kiwi_init's last argument; Dialect bits, 0 = standard only)
	void bakeModel(FlatModel& out, const std::string& rawModelPath, uint32_t enabledDialects = 0);
	void buildTrieEdges(FlatModel& m);      // FlatModel::trieEdges from trie / trieKeys / trieChild (the end of every bake)
	// ... with temporary forms and morphemes behind the model's own (pretokenized spans, src/Kiwi.cpp:785-946; model.cpp).  Form j gets id nForms + j, morpheme k
	// id nMorphs + k; `cands` / `chunks[].morph` are morpheme ids of the model or of these temporaries
	struct TempEntries
	{
		struct Form { std::u16string str; std::vector<uint32_t> cands; };
		struct Chunk { uint32_t morph; uint8_t begin, end; };
		struct Morph { uint32_t tempForm; uint8_t tag; uint32_t lmId; std::vector<Chunk> chunks; };
		std::vector<Form> forms; std::vector<Morph> morphs;
	};
	void bakeModelWithTemps(FlatModel& out, const std::string& rawModelPath, uint32_t enabledDialects, const TempEntries& temps);
	// The same temporaries as an OVERLAY behind a baked model: exactly what bakeModelWithTemps appends to the tables a lattice node, the search and the result
	// assembly index by form / morpheme id -- computed from the baked model alone (no second bake, no model file), so that an analysis with pretokenized spans
	// costs the temporaries and not the dictionary.  The engine uploads it per batch behind the device copies of those tables (ids >= the model's counts).
	// tests/test_pretok_overlay.py: equal to the tail of bakeModelWithTemps' tables on every golden span case.
	struct TempOverlay
	{
		uint32_t nBaseForms = 0, nBaseMorphs = 0;
		std::vector<FormRec> forms;          // the temporary forms, then the closing sentinel: forms[nBaseForms ...]
		std::vector<uint16_t> formChars;     // formChars[base.formChars.size() - 1 ...]: over the base's closing 0, with its own
		std::vector<uint32_t> formCand;      // appended
		std::vector<MorphRec> morphs;        // appended (host values; the device copy takes feat / prevFlags from morphPath as the model's do)
		std::vector<uint32_t> chunkMorph, chunkLm; std::vector<uint8_t> chunkPos;      // appended
		std::vector<uint8_t> sbInfo; std::vector<uint32_t> morphPath, morphKform;      // per temporary morpheme
		std::vector<float> formUnkChr; std::vector<uint16_t> formChrTok;                // a model with the character model: per temporary form / per unit of formChars
		bool empty() const { return forms.empty(); }
	};
	void bakeTempsOverlay(const FlatModel& base, const TempEntries& temps, TempOverlay& out);
	// serialises the baked dictionary in the layout of oracle/ref_bridge.cpp:kref_dump_dict (tests compare both)
	std::vector<uint8_t> dumpDict(const FlatModel& m);
	// Kiwi::findMorphemes (src/Kiwi.cpp:1281-1297, findForm src/KTrie.cpp:2172-2192): the morphemes of the dictionary form spelled `s` (raw text: it is
	// normalised like the reference's normalizeHangul) whose tag, irregularity aside, is `tag` (0 = any tag); halves of split stems are not returned
	std::vector<uint32_t> findMorphemes(const FlatModel& m, const char16_t* s, size_t n, uint8_t tag);
	// findForm (src/KTrie.cpp:2172-2192): the dictionary form whose string, spaces aside, is exactly the NORMALISED string `nrm`; -1 if none ends there
	int32_t formIdOfString(const FlatModel& m, const std::u16string& nrm);
	// Morpheme::hasMorpheme over a set of morpheme ids (include/kiwi/Form.h:187-196), for every morpheme at once: bit m is set iff the set holds
	// m's combined morpheme or one of its chunks -- what the candidate loops test a blocklist with (src/PathEvaluator.hpp:385, 892)
	std::vector<uint32_t> blockBitsOf(const FlatModel& m, const std::vector<uint32_t>& ids);
	// UnkFormScorer::chrBasedScore without the final bias (src/UnkFormScorer.cpp:53-66) on the host: the bake's per-form table, the oracle
	float chrScoreHost(const ChrView& C, const uint16_t* s, size_t n);
}
