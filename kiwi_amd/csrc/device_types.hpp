// Device-side data layout of one analyse batch (all buffers live in HBM; plain pointers, no torch types).
// A "chunk" is the unit one lattice is built for (the reference's per-call splitByTrie range,
// /root/reference/src/KTrie.cpp:766-858); chunks of a batch are independent.
#pragma once
#include <cstdint>
#include "flat_model.hpp"

namespace kamd
{
	struct DevPattern { uint32_t end, length; uint32_t tag; };   // chunk-relative (textprep.hpp PatternSpan)
	// A pretokenized span of the chunk (pretok.hpp; KTrie.cpp:1177-1210) travels as an entry BEHIND the chunk's patterns: {end, length} chunk-relative, tag =
	// kSpanTag | kSpanFallback (the node takes the text as its own string) | the form id of its lattice node.  Only the replay of the reference's splitter
	// (latticeSerialBuild, lattice_kernels.hip) reads them: the engine sends a batch with spans to that kernel.
	constexpr uint32_t kSpanTag = 0x80000000u, kSpanFallback = 0x40000000u, kSpanFormMask = 0x00FFFFFFu;

	// lattice node, 32 B (reference: KGraphNode 56 B, src/KTrie.h:57-77)
	struct alignas(16) DevNode
	{
		uint32_t form;                 // form id or NOFORM
		uint16_t startPos, endPos;     // ns positions while building, chunk-relative string offsets when final
		uint16_t prev, sibling;        // relative links, as in the reference
		uint16_t uformOff, uformLen;   // chunk-relative substring for OOV / special / pattern nodes
		uint8_t spaceErrors;
		uint8_t nflags;                // NF_* bits, filled by k_build_lattice for the search kernel
		uint16_t nPrev;                // number of predecessor nodes (length of the prev/sibling chain)
		uint32_t packOff;              // first candidate record of this node in the chunk's candidate-pack region
		uint16_t candCnt;              // candidate morphemes of `form`
		uint8_t fflags;                // FormRec::flags
		uint8_t flen;                  // FormRec::len
		uint16_t ownFeat;              // left-feature mask (+ LF_STR_SSC) of the node's own surface string (uform)
		uint16_t pad;
	};
	enum NodeFlag : uint8_t { NF_SPACE_BEFORE = 1, NF_LEFT_BOUNDARY = 2, NF_UFORM_ENDS_POINT = 4, NF_ALL_PARTIAL = 8 };
	static_assert(sizeof(DevNode) == 32, "DevNode");
	// static part of a candidate record (k_expand_cands): MorphRec dwords 0..7, then {morpheme id, first LM id, sbType, 0}
	struct alignas(16) Quad { uint32_t x, y, z, w; };
	struct CandStatic { Quad m0, m1, x; };
	constexpr uint32_t NOFORM = 0xFFFFFFFFu;

	// ---- position program of the position-step search kernel (k_pos_path, viterbi_pos.inc), written once per batch by k_expand_pos ----
	// Lattice nodes that END at the same position have all their predecessors complete and do not feed each other, so the search advances one
	// END POSITION per step instead of one node.  Everything about such a step that is a function of the lattice alone is precomputed:
	// one PosRec per (node, candidate that will be evaluated) -- statically skipped candidates are gone, the unknown-noun candidates of formless
	// nodes and the extra proper-noun reading of all-partial forms (PathEvaluator.hpp:1277-1287) are ordinary records -- with every node-level
	// term (discounts, left-boundary score, rule-scorer inputs, own-form facts) already folded in.
	struct alignas(16) PosRec
	{
		uint32_t firstWid, secondWid, chunkOff, lastSeqId;      // LM ids fed first / second, chunk table offset, word id recorded on the path
		uint32_t morph, flagsFeat, tagw, cntw;                  // morpheme id; MorphRec dwords 5..7 (device copy: path-side feat / prevFlags)
		float additional;                                       // userScore + node-level discount + left-boundary tag score (PathEvaluator.hpp:366-383)
		uint32_t nodeOwn;                                       // node index | own-form feature mask << 16
		uint32_t bits;                                          // sbType | ruleBits << 8 | ownKind << 16 | node flags (NF_*) << 24
		uint32_t rq;                                            // R (start states a quote / sentence-break candidate is tried under) | node's index inside its position << 8 | PR_* << 16
	};
	static_assert(sizeof(PosRec) == 48, "PosRec");
	enum PosRecFlag : uint32_t { PR_PASS1 = 1u << 16,          // the all-partial form's extra unknown-noun reading (a second evaluation of the node)
		PR_OUT_FIRST = 1u << 17,
		PR_Z = 1u << 18 };                                      // z-coda / z-siot shortcut: firstWid = the morpheme put on, secondWid = the shortcut's tag, chunkOff = path-side feat | prevFlags << 16 | socket << 24, additional = its score                              // CoNgram: the evaluation's regular candidates share one first word (qgemm dispatch)
	// one END position: nodes [firstNode, firstNode + nNodes), records [firstRec, firstRec + nRec) of the chunk.  Entry 0 of a chunk's table is
	// its header: firstRec = number of positions (0: the chunk is left to the general kernel), nNodes != 0: no reachability propagation for this chunk,
	// pad2 = the position the end node's predecessors end at.
	struct alignas(16) PosDesc { uint16_t firstNode; uint8_t nNodes; uint8_t flags; uint32_t firstRec; uint16_t nRec; uint16_t pad; uint32_t pad2; };      // pad: bit j = node j of the position has no dictionary form; pad2 bit j: it has the extra unknown-noun reading (PR_PASS1 record)
	static_assert(sizeof(PosDesc) == 16, "PosDesc");
	constexpr uint32_t kPosChunkDone = 0xFFFFFFu;      // DevChunkResult::pad of a chunk k_pos_path searched to the end
	enum PosFlag : uint8_t { POSF_SLOW = 1 };                     // something the position-step kernel does not do (z-coda / z-siot shortcut, > 16 records or nodes): hand over

	// search state, 48 B = three 16-byte quads (reference: WordLL<KnLMState> 48 B, src/BestPathContainer.hpp:21-67).
	// Quad 0 is everything a successor transition reads ("hot": one 16-byte load per work item); quads 1-2 are only
	// needed when a state is created from its parent and by the back-trace.
	struct alignas(16) DevState
	{
		int32_t lmNode;
		float accScore;
		uint16_t leftFeat;             // feature mask of the left string as seen by the next morpheme (+ LF_* bits)
		uint8_t rootId, spState;
		uint8_t socket, prevFlags;
		uint8_t dead;                  // pruned (PathEvaluator.hpp:503-511); dead states stay in place and are skipped
		uint8_t ownKind;               // 0 none, 1 node.uform, 2 node.form string, 3 text[node.start,node.end)
		float accTypoCost;
		uint32_t wid;
		uint32_t parent;               // chunk-relative state index
		uint32_t morph;
		float firstChunkScore;
		uint16_t nodeId;
		uint16_t ownNode;              // node whose own form this path carries
		uint32_t pad0, pad1;
	};
	static_assert(sizeof(DevState) == 48, "DevState");
	constexpr uint8_t COMMON_ROOT = 0xFF;

	// output token, 24 B (reference: PathNode 72 B, src/PathEvaluator.h:33-68)
	struct DevToken
	{
		uint32_t morph;
		uint16_t begin, end;           // chunk-relative offsets in the normalised text
		float wordScore, typoCost;
		uint32_t ownA;                 // ownKind 1/3: chunk-relative offset of the own form; 2: form id
		uint16_t ownLen;
		uint8_t ownKind, pad;
	};
	static_assert(sizeof(DevToken) == 24, "DevToken");

	// The end stage writes what the host needs -- path headers and token records -- COMPACTLY into two batch-wide output arrays
	// (WorkView::outPaths / outTokens, ranges handed out by wave-aggregated atomics on WorkView::outCounters), so that the D2H copy is
	// sum(tokens) x 24 B instead of the token arenas at capacity, and a chunk may return any number of paths.
	struct DevPathHeader { float score; uint32_t tokOff; uint16_t nTokens; uint8_t prevState, curState; };   // tokOff: relative to the chunk's first output token
	// nEnd/endOff: end-node candidates of the chunk, left by k_best_path in the unused tail of the chunk's state arena
	// (entry index endOff, 24-byte records) for k_finish_paths; pathOff/tokOff: the chunk's ranges in the output arrays
	struct DevChunkResult { uint32_t nPaths, status, nEnd, endOff, pathOff, tokOff, nTok, pad; };
	static_assert(sizeof(DevChunkResult) == 32, "DevChunkResult");

	enum ChunkStatus : uint32_t
	{
		CS_OK = 0, CS_NO_LATTICE = 1,   // <= 2 nodes: contributes nothing (Kiwi.cpp:1119)
		CS_ERR_MATCH_OVERFLOW = 16, CS_ERR_NODE_OVERFLOW = 17, CS_ERR_STATE_OVERFLOW = 18, CS_ERR_TOKEN_OVERFLOW = 19,
		CS_ERR_PATH_OVERFLOW = 20, CS_ERR_TOO_LONG = 21, CS_ERR_PAIR_OVERFLOW = 22,
	};

	struct SearchParams   // KiwiConfig + per-call options (include/kiwi/Kiwi.h:150-167, 69-134)
	{
		uint64_t match;
		float cutOff, spacePenalty, typoCostWeight, oovRuleScale, oovRuleBias;
		uint32_t maxUnk, maxUnkJ, spaceTol;
		uint32_t splitComplex, splitSaisiot, mergeSaisiot;
		uint32_t smallMax, mediumMax, bucketCap;   // container selection by incoming paths (128, 512) and per-bucket key cap (128): BestPathContainer.hpp:275-277
		uint32_t topN;                 // paths kept per (candidate, key): 1..kMaxTopN (BestPathContainer.hpp:151-222 for N > 1)
		float oovChrBias;              // KiwiConfig::oovChrBias: subtracted from the character model's score of an unknown form (Match::oovChrModel)
		// Match::oovChrFreqModel: KiwiConfig::oovGlobalWeight / oovLocalWeight / oovGlobalMinFreq, and the bias k_unk_chr_freq applies itself (oovChrBias is 0 then)
		float oovGlobalWeight, oovLocalWeight, oovGlobalMinFreq, oovChrFreqBias;
		// AnalyzeOption::allowedDialects (Dialect bits) / dialectCost: a dictionary form or a morpheme of another dialect is skipped, one of an allowed
		// dialect costs dialectCost (KTrie.cpp:207-229, PathEvaluator.hpp:231, 386, 893); only read when the model has dialect morphemes
		uint32_t allowedDialect; float dialectCost;
	};
	constexpr uint32_t kMaxTopN = 16;

	struct BatchView
	{
		uint32_t nChunks;
		const uint16_t* chars;         // normalised text of all chunks, concatenated
		const uint8_t* cls;            // per unit: character type (low 6 bits) | 0x80 emoji start
		const uint8_t* script;         // per unit: script id
		const uint32_t* charOff;       // [nChunks+1]
		const uint32_t* patOff;        // [nChunks+1]
		const DevPattern* patterns;
		const uint32_t* spOff;         // [nChunks+1] into spStates: sorted unique previous SpecialStates of the chunk
		const uint8_t* spStates;
		const uint8_t* chunkFlags;     // bit0: openEnding applies to this chunk; bit1: the only chunk of its text (a top-1 analysis needs its best path only: k_finish_paths)
		const uint32_t* textOffset;    // [nChunks] offset of the chunk inside its normalised text (Kiwi.cpp:1095-1117 `splitEnd`)
		// Match::oovChrFreqModel only (null otherwise): the FILTERED normalised text a chunk belongs to (Kiwi.cpp:1058-1086: special characters and spaces
		// blanked) -- the whole text, not the chunk: substring frequencies are counted over it (chr_freq.hpp)
		const uint16_t* filtChars; const uint32_t* filtOff; const uint32_t* filtLen;      // [nChunks] each
	};

	// Scratch + outputs.  All per-chunk regions are laid out by the host from the chunk lengths
	// (capacities are linear in the chunk length; a chunk that outgrows one reports CS_ERR_* and is re-run
	// by the host with larger capacities).  "+c" slots: per-character arrays have one extra slot per chunk.
	struct WorkView
	{
		uint16_t* nsToPos;             // [charOff[c] + c + i]
		uint16_t* posToNs;             // [charOff[c] + c + i], i in [0, nChars]
		uint8_t* cflag;                // [charOff[c] + i]  bit0 non-space, bit1 skipped by the dictionary scan
		uint64_t* matchMask;           // [charOff[c] + c + e] per ns end position e: bit d-1 set <=> a form of d units ends at e
		uint32_t* matchOff;            // [charOff[c] + c + e] offset of e's forms inside the chunk's matchForm region
		uint32_t* nNs;                 // [c]
		const uint32_t* matchBase;     // [nChunks+1]
		uint32_t* matchForm;
		const uint32_t* nodeBase;      // [nChunks+1]
		DevNode* nodes;                // final lattice of chunk c at nodeBase[c]
		DevNode* tmpNodes;             // build-order nodes, same offsets
		uint32_t* endPosMap;           // [charOff[c] + c + p] : first | second<<16
		uint64_t* fullMask;            // [charOff[c] + c + e] bit L-1: a zero-cost node of length L that is unknown or has a full morpheme ends at e
		uint8_t* zAt;                  // [charOff[c] + c + e] bit0/1: a form ending at e allows a trailing z-coda / saisiot
		uint16_t* tmpIdx;              // per node scratch (inverse permutation / BFS queue), 2 per node
		uint32_t* nNodes;              // [c]
		const uint64_t* stateBase;     // [nChunks+1]
		DevState* states;
		// A chunk whose arena fills up carries on in one twice as large from the batch's pool -- the states [poolBase, poolBase + poolCap) behind the chunks' own arenas,
		// handed out append-only through one counter -- so that the arenas are sized for the typical chunk, not for the worst one.  stateAt[c] = where chunk c's
		// states lie now (stateBase[c] until it grows); state indices are arena-relative.  poolTop null: no pool (the arenas hold the worst case or the chunk is re-run).
		// slotCap != 0 (models whose states carry histories: SkipBigram, global CoNgram): the arenas belong to the search kernel's lane groups, not to the chunks --
		// group g of block b searches every chunk it takes in states [(b * groups + g) * slotCap, + slotCap), and runs the chunk's end stage itself (finishPathsSolo)
		// before it takes the next chunk: the batch's state memory is what the chunks IN FLIGHT need, whatever the batch size
		uint32_t slotCap;
		uint64_t* slotTable;           // [2 * slots], zeroed before a run: a group that grew keeps the larger arena for its later chunks -- {first state, capacity}, capacity 0 = its own arena
		uint64_t* stateAt;             // [nChunks]
		unsigned long long* poolTop;   // states of the pool handed out so far
		uint64_t poolBase, poolCap;
		uint32_t* nodeStateOff;        // per node (same offsets as nodes): chunk-relative first state
		uint32_t* nodeStateCnt;
		uint8_t* reach;                // per node: the reference's `reachable` flags (PathEvaluator.hpp:1159-1176, 1286)
		const uint32_t* packBase;      // [nChunks+1] candidate-pack regions
		CandStatic* packs;
		const uint64_t* tokenBase;     // [nChunks+1]
		DevToken* tokens;
		DevChunkResult* results;       // [c]
		DevPathHeader* outPaths;       // compact output of the end stage: path headers of all chunks ...
		DevToken* outTokens;           // ... and their token records (D2H copies exactly what was produced)
		uint32_t* outCounters;         // [0] path headers handed out, [1] token records handed out, [2] chunks that ended in a scratch overflow, [3] entries of wideList, [4..15] k_lattice_wave's counters
		uint32_t* wideList;            // [nChunks] chunks k_lattice_wave's first launch left to its wide launch
		uint8_t* expanded;             // [nChunks] 1: k_lattice_wave wrote the chunk's candidate records and position program itself -- k_expand_cands / k_expand_pos skip it
		// Match::oovChrModel (null: unknown forms are scored by the length rule): per node, same offsets as nodes, the character model's score of
		// the node's unknown form -- its own string of a formless node, else its text span (k_unk_chr; UnkFormScorer::chrBasedScore before the bias)
		float* unkChr;
		// Match::oovChrFreqModel (null otherwise): unkChr holds the FINAL scores (bias subtracted: the consumers see a bias of 0), and per node with a dictionary form
		// the frequency-based score of that form's own string (replaces ModelView::formUnkChr, which cannot know the text)
		float* unkChrForm;
		const uint32_t* blockBits;     // AnalyzeOption::blocklist as one bit per morpheme id (null: none): k_expand_cands drops those candidates
		uint32_t outPathCap, outTokCap;
		// position program (k_expand_pos -> k_pos_path): records at the chunk's packBase offset (same capacity as its candidate packs), position
		// table, per-node predecessor ranges (first | (count - 1) << 16 | position of the predecessors << 24) and per-position start positions (four bytes) at its nodeBase offset; null = the position-step kernel is not used
		PosRec* posRecs; PosDesc* posDesc; uint32_t* posPrev; uint32_t* posNodeRec; uint32_t* posMask;
		uint8_t* posBig;               // k_pos_path: staging of the items of a record with more than 16 of them, 64 x 20 bytes per lane group (4 per block)
		uint8_t* posScratch; uint32_t* posContCounter; uint32_t posContSlots;      // k_pos_path carrying a chunk on in the general search itself: item scratch slots (GroupScratch each), slots taken, number of slots
		uint32_t* posHandOver;         // set by k_pos_path when it hands a chunk over (DevChunkResult::pad = the node to resume at, kPosChunkDone = nothing left): k_best_path returns at once while it is 0
		uint8_t* bigScratch;           // fallback scratch for nodes with > 128 incoming (path, root) pairs
		uint32_t bigScratchBytes;      // per wave
		uint32_t* beacon;              // developer aid (KAMD_TIMELINE builds): per-chunk timeline records, else null
	};

	// SkipBigram model on the device (reference src/SkipBigramModel.hpp:40-105) plus the history storage of the search:
	// only the SkipBigram instantiation of the search kernel takes this view (Knlm-only models never see it).
	struct SbgDev
	{
		const uint32_t* ptrs;          // [vocab + 1] into keys / comps
		const uint32_t* keys;          // partner (history) word ids, sorted per `next` word
		const float* comps;            // compensation per key
		const float* discnts;          // [vocab]
		const uint8_t* valid;          // [vocab]
		uint32_t vocabSize; float logWindowSize;
		uint32_t* hist;                // [state][8]: history ring of every search state, parallel to WorkView::states (ring position: DevState::pad0)
		uint8_t* itemScratch;          // per lane group: SbgScratch (viterbi_kernel.hpp), rings of the work items of one batch
	};

	// CoNgram model on the device (flat_model.hpp CongView): embedding rows of dim x s8 + f32 scale + f32 bias (context) / f32 scale + 4 unused bytes (output)
	struct CongDev { const uint8_t* ctxEmb; const uint8_t* outEmb; uint32_t dim, stride; uint32_t vlTMax, vlBits; };      // vlTMax / vlBits: CongView (variable-length trie keys; 0xFFFFFFFF = none)
	// ... and what the GLOBAL model (ModelType::congGlobal, window 7: flat_model.hpp CongView, cong_global.hpp) adds: the window sections of the file, and the
	// history storage of the search -- seven distant words + the newest slot per search state, parallel to WorkView::states like the SkipBigram rings.
	// Only the congGlobal instantiations of the search kernel take this view (viterbi_kernel_congg.hip).
	struct CongGDev
	{
		const float* ctxConf; const uint8_t* distEmb; const float* distConf; const float* posConf; const uint8_t* distMask;
		uint32_t window, keyBytes;     // keyBytes: sizeof(KeyType) of the reference's instantiation -- its state hash reads 8 BYTES of the history
		uint32_t* hist;                // [state][8]
		uint8_t* itemScratch;          // per lane group: SbgScratch (viterbi_kernel.hpp): histories of the work items of one batch
	};

	// LDS layout of the wave-per-chunk lattice build (byte offsets): n text units, node capacity, packed-match capacity
	struct LatticeLds { uint32_t str, cls, script, cflag, nsToPos, posToNs, mask, moff, endPosMap, fullMask, zAt, mforms, mfrec, out, spaceErr, queue, total; };
	// LDS-side capacities are the typical need (3 per text unit), not the worst-case HBM capacities: a chunk that outgrows
	// them at run time is handed to the thread-per-chunk kernel (flag kLatticeNeedsBig in nNodes[chunk])
	constexpr uint32_t kLatticeNeedsBig = 0xFFFFFFFEu;
	constexpr uint32_t kLatticeNeedsWide = 0xFFFFFFFDu;      // k_lattice_wave: the chunk's ops outgrew the LDS arrays of the first launch -- the wide launch takes it
	// (3 per unit up to 128 units, 2 per unit beyond: long chunks are the ones whose LDS copies limit the resident wavefronts -- MI355X, c4-cong: lattice
	// stage 19.0 -> 17.x ms -- and they average fewer nodes per unit; short chunks gain nothing from smaller copies, c2-64k 1.99 ms either way)
	KAMD_HD uint32_t latticeLdsCap(uint32_t n, uint32_t hbmCap) { const uint32_t c = n < 128 ? 3 * n + 32 : 2 * n + 160; return c < hbmCap ? c : hbmCap; }
	KAMD_HD LatticeLds latticeLdsLayout(uint32_t n, uint32_t nodeCapHbm, uint32_t matchCapHbm)
	{
		const uint32_t nodeCap = latticeLdsCap(n, nodeCapHbm), matchCap = latticeLdsCap(n, matchCapHbm);
		LatticeLds l; uint32_t o = 0;
		auto take = [&](uint32_t bytes) { const uint32_t at = o; o = (o + bytes + 15u) & ~15u; return at; };
		l.str = take(2 * n); l.cls = take(n); l.script = take(n); l.cflag = take(n);
		l.nsToPos = take(2 * (n + 2)); l.posToNs = take(2 * (n + 2));
		l.mask = take(8 * (n + 2)); l.moff = take(4 * (n + 2));
		l.endPosMap = take(4 * (n + 2)); l.fullMask = take(8 * (n + 2)); l.zAt = take(n + 2);
		l.mforms = take(4 * matchCap); l.mfrec = take(8 * matchCap);
		l.out = take(16 * nodeCap); l.spaceErr = take(nodeCap); l.queue = take(4 * nodeCap);     // 16-byte build nodes (lattice_kernels.hip BuildNode16)
		l.total = o;
		return l;
	}

	// ---- LDS layout of k_lattice_wave (lattice_wave.hip): the lattice build in which all 64 lanes work -- the chunk's appends ("ops", in the
	// reference's order) are decided together by a fixpoint over the few order-dependent quantities instead of being replayed one by one.
	// n text units, P = n + 2 positions, Mc packed matches, Kc ops (matches + special / space / pattern / tail / end ops), Nc final nodes.
	struct LwLds
	{
		uint32_t str, cls, script, cflag, nsToPos, posToNs, mask, moff, mforms, mse, mfc;   // staged inputs; form id, space errors and form facts of every packed match
		uint32_t ctlBU, ctlT, ctlRs;                                                        // per end position: boundary | unkStart << 16, time of its first match op, resetNs
		uint32_t opNE, opBU, opFl, opSrc, decS, decT, grpList, miscForm, miscU, miscFc;     // per op (time order, 1-based)
		uint32_t grpOff, posA, posZ, fd, unkMinT, cntU, cntA, succ, base, firstU, cc, recOff, scal; // per position (cc: per final node; scal: a few wave-wide words)
		uint32_t total, matchCap, opCap, miscCap, nodeCap;
	};
	// matchRatio16: LDS room for the packed matches, in sixteenths per text unit (the engine follows what its model's dictionary produces: a chunk that
	// needs more is built by the second, `wide` launch -- 3 matches per unit and a miscellaneous op per unit).  Arrays that are only needed while the ops
	// are made (the scan's masks, the per-position control words, script / cflag / posToNs) share their bytes with the arrays of the final phases.
	constexpr uint32_t kLatticeWideRatio16 = 48, kLatticeWideBit = 0x8000u;      // (the bit marks the wide launch in the kernel's ratio argument)
	KAMD_HD LwLds latticeWaveLayout(uint32_t n, uint32_t nodeCapHbm, uint32_t matchCapHbm, uint32_t matchRatio16)
	{
		LwLds l; uint32_t o = 0;
		auto take = [&](uint32_t bytes) { const uint32_t at = o; o = (o + bytes + 15u) & ~15u; return at; };
		const uint32_t P = n + 2;
		const bool wide = (matchRatio16 & kLatticeWideBit) != 0;
		const uint32_t wantM = n * (matchRatio16 & 0x3FFFu) / 16 + 16;      // (bits 14 / 15 of the kernel's argument are flags)
		l.matchCap = wantM < matchCapHbm ? wantM : matchCapHbm; l.miscCap = (wide ? n : n / 2) + 12; l.opCap = l.matchCap + l.miscCap + 2;
		const uint32_t wantN = wide ? 2 * l.opCap : l.opCap + 16;
		l.nodeCap = wantN < nodeCapHbm ? wantN : nodeCapHbm;
		l.str = take(2 * n); l.cls = take(n); l.nsToPos = take(2 * P);
		l.mforms = take(4 * l.matchCap); l.mse = take(l.matchCap); l.mfc = take(4 * l.matchCap);
		l.opNE = take(4 * l.opCap); l.opBU = take(4 * l.opCap); l.opFl = take(2 * l.opCap); l.opSrc = take(2 * l.opCap);
		l.decS = take(2 * l.opCap); l.decT = take(4 * l.opCap); l.grpList = take(2 * l.opCap);
		l.miscForm = take(4 * l.miscCap); l.miscU = take(4 * l.miscCap); l.miscFc = take(4 * l.miscCap);
		l.grpOff = take(4 * (P + 1)); l.posA = take(4 * P); l.posZ = take(4 * P); l.fd = take(8 * P); l.unkMinT = take(2 * P); l.cntU = take(2 * P); l.base = take(2 * P);
		l.scal = take(16);
		const uint32_t shared = o;
		l.script = take(n); l.cflag = take(n); l.posToNs = take(2 * P); l.mask = take(8 * P); l.moff = take(4 * P); l.ctlBU = take(4 * P); l.ctlT = take(2 * P); l.ctlRs = take(2 * P);
		const uint32_t endEarly = o;
		o = shared;
		l.cntA = take(4 * P); l.succ = take(8 * P); l.firstU = take(4 * P); l.cc = take(4 * l.nodeCap); l.recOff = take(4 * l.nodeCap);
		l.total = o > endEarly ? o : endEarly;
		return l;
	}

#ifdef __HIPCC__
	// Lanes of one wavefront exchange data through LDS / HBM between phases.  A memory fence alone orders one lane's own
	// accesses; it does not stop the compiler from letting lanes that left a divergent loop early run ahead into the next
	// phase (plain loads and stores may be duplicated into loop exits).  The wave barrier is a convergent no-op: every lane
	// that reaches this point in the source reaches it together in the generated code, so the phases stay separated.
	__device__ inline __attribute__((always_inline)) void waveSync()
	{
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
	}
	// The same separation of phases for data exchanged through LDS only: waits for the wave's LDS operations, not for its outstanding global loads /
	// stores (a store's acknowledgement takes ~1 us; the position-step kernel exchanges everything a following step reads through LDS, and what it
	// does re-read from HBM are earlier stores of the same wave, which its loads follow in order through the same L1)
	__device__ inline __attribute__((always_inline)) void waveSyncLds()
	{
#if defined(__HIP_DEVICE_COMPILE__)
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront", "local");
#else
		waveSync();
#endif
	}
#endif
}
