// HIP kernel for the lattice best-path search (Viterbi over morpheme candidates, Knlm scoring) on gfx950.
//
// Work decomposition.  A lattice node has on average < 2 candidate morphemes and < 5 incoming paths, so a
// whole wavefront per chunk leaves > 90 % of its lanes idle and the search is bound by the latency of its
// dependent loads (Knlm back-off walk, state and morpheme records), not by bandwidth.  The kernel therefore
// splits every 64-wide wavefront into NG = 64/G independent *lane groups* of G lanes; each group owns one
// chunk at a time (pulled longest-first from an atomic counter, so the groups of a wave see similar work) and
// sweeps its lattice in stored order (nodes sorted by end position = topological order).  Within a group the
// work items of a node -- (candidate morpheme, incoming path[, root]) tuples, flattened over consecutive
// candidates -- are scored G at a time, one per lane: side terms + the Knlm walk.  That puts 64 independent
// dependent-load chains in flight per wave instead of ~5.
//
// Dependent-load levels per node are kept small: node facts come precomputed from k_build_lattice and are
// prefetched one node ahead, per-node state ranges live in an LDS ring, the Knlm walk uses a bucketed edge hash
// (one 64-byte bucket + the node's back-off record per level, the edge's log-likelihood stored in the slot),
// and pruning never re-reads states: scores are staged in LDS and losers are only *marked* dead in HBM.
//
// De-duplication per (candidate, LM state, root, special state) -- the reference's per-morpheme hash
// container (src/BestPathContainer.hpp:279-483): an item is the representative of its key iff no earlier item
// of the same candidate carries the key; the winner of a key is the first item with the maximal score ("first
// inserted wins on ties").  Common case (a batch fits the group, small container; G = 8 or 16, where a group is
// half / all of a 16-lane DPP row): keys and scores stay in registers and 15 branch-free row_ror steps decide
// representative, winner and (top-N) rank; pruning is a 4-step DPP max or the same 15-step count.  Otherwise the
// scored items are staged in LDS (<= QCAP) or per-group HBM scratch and scanned.  Group ballots (slices of the
// wave ballot) give container order.  States stream to a per-chunk arena in HBM (48 B each: a 16-B hot quad read
// by successors + 32 B for the back-trace); losers of the pruning are only *marked* dead.  The chunk ends with the
// end-of-sentence transition of the surviving paths (finishChunk: a real function call that gets COPIES of the
// context, so that the context itself and the kernel arguments stay in registers / the kernarg segment); sorting
// them, the selection and the back-trace are k_finish_paths, one thread per chunk.
//
// Lanes of a group exchange data between phases through LDS / HBM; phases are separated by waveSync()
// (wavefront fence + wave_barrier, device_types.hpp).  Developer aids: `make timeline` (per-chunk / per-phase
// stamps, tools/timeline.py), KAMD_HANGDUMP=1 (engine.hip: reads per-chunk progress back while the kernel hangs),
// `make smallcaps` (tiny LDS capacities: every fallback path in the parity suite).
//
// Reference behaviour reproduced: BestPathFinder::findBestPath (src/PathEvaluator.hpp:1178-1419),
// PathEvaluator::operator()/evalSingleMorpheme (:347-635), RuleBasedScorer/insertToPathContainer/
// FormEvaluator (:88-311), generateTokenList (:1038-1157), KnLangModel::progress (src/Knlm.cpp:44-130).
// top-N (1 < N <= kMaxTopN): the N best paths per key are kept (each with its own values), pruning uses the N-th best score
// of a root, the end stage hands on ceil(2N / groups) candidates per group; kept paths are handed on in item order (top-N).
#include <hip/hip_runtime.h>
#if defined(KAMD_LMSTATS) || defined(KAMD_POSSTATS)
#include <atomic>
#include <cstdio>
#endif
#include "device_types.hpp"
#include "feature.hpp"
#include "viterbi_kernel.hpp"
// SkipBigram models (Knlm + skip-bigram mixture): this file is compiled a second time by viterbi_kernel_sbg.hip with KAMD_SBG
// defined, into namespace kamd::sbgk.  Everything that exists only there is spelled SBG_ONLY(...) or sits under #ifdef KAMD_SBG,
// so that without the macro the translation unit is, token for token, the Knlm kernel that was measured and profiled.
#ifdef KAMD_SBG
#include "sbg_eval.hpp"
#define SBG_ONLY(...) __VA_ARGS__
#else
#define SBG_ONLY(...)
#endif
// Typo correction: a third compilation (viterbi_kernel_typo.hip, KAMD_TYPO, namespace kamd::typok) in which every lattice node carries a
// typo cost (an array beside the node records): it discounts the node's candidates and accumulates on the paths (PathEvaluator.hpp:235, 371).
#ifdef KAMD_TYPO
#define TYPO_ONLY(...) __VA_ARGS__
#else
#define TYPO_ONLY(...)
#endif
// CoNgram models (local, quantised; reference src/CoNgramModel.cpp): a fourth compilation (viterbi_kernel_cong.hip, KAMD_CONG, namespace kamd::congk).
// The LM state is the context-trie node plus the context id of the history (DevState::pad0); a transition is scored by an int8 dot product of
// two embedding rows; candidates are evaluated in the order of the reference's transposed evaluator (MorphemeEvaluator<CoNgramState>).
#ifdef KAMD_CONG
#define CONG_ONLY(...) __VA_ARGS__
#else
#define CONG_ONLY(...)
#endif
// The GLOBAL CoNgram model (ModelType::congGlobal, window 7; viterbi_kernel_congg.hip: KAMD_CONG + KAMD_CONGG, namespace kamd::congk::gk): a state also
// carries the last seven valid distant words of its path and the newest slot (cong_global.hpp).  The history travels exactly as the SkipBigram ring does
// -- an arena parallel to the states, a copy per work item in the lane group's item scratch, a digest in the de-duplication keys --, so that plumbing is
// spelled HIST_ONLY(...) / KAMD_HIST and serves both compilations; what differs is spelled CONGG_ONLY / SBG_ONLY.
#ifdef KAMD_CONGG
#include "cong_global.hpp"
#define CONGG_ONLY(...) __VA_ARGS__
#else
#define CONGG_ONLY(...)
#endif
#if defined(KAMD_SBG) || defined(KAMD_CONGG)
#define KAMD_HIST 1
#define HIST_ONLY(...) __VA_ARGS__
#else
#define HIST_ONLY(...)
#endif
#if defined(KAMD_SBG) || defined(KAMD_CONG)
#define STATE_EXTRA(...) __VA_ARGS__      // the state's spare dword carries part of the LM state (ring position / context id)
#else
#define STATE_EXTRA(...)
#endif
#if defined(KAMD_SBG) || defined(KAMD_TYPO) || defined(KAMD_CONG)
#define KAMD_VARIANT 1      // not the Knlm translation unit: the pieces that exist once (end-stage kernel, LDS size helper) are left out
#endif

namespace kamd
{
#ifdef KAMD_TYPO
namespace typok
{
#endif
#ifdef KAMD_CONG
namespace congk
{
#ifdef KAMD_CONGG
namespace gk
{
	// (with histories in the keys a node gathers many more paths: the SkipBigram kernel's staging capacity)
	constexpr uint32_t BIGQ = BIGQ_SBG;
	using GroupScratch = GroupScratchCong<BIGQ_SBG>;
#else
	// (the big queue of this namespace also carries the context id of every work item)
	using GroupScratch = GroupScratchCong<BIGQ>;
#endif
#endif
#ifdef KAMD_SBG
namespace sbgk
{
	// (shadow the Knlm kernel's staging capacity and scratch record inside this namespace)
	constexpr uint32_t BIGQ = BIGQ_SBG;
	using GroupScratch = GroupScratchT<BIGQ_SBG>;
#endif
	constexpr uint64_t KINVALID = ~0ull;
// Inlining level of the node loop's stages (3 = everything inlined into the kernel: the lane-group context then lives in
// registers).  The end-candidate stage (finishChunk) always stays a real function: inlined as well, the hipcc of ROCm 7.2
// generated gfx950 code in which the end-stage loop of EVERY chunk never terminated (nested divergent loops, > 270 SGPRs
// of lane masks spilled to VGPR lanes); tools/quick_gpu.py with KAMD_HANGDUMP=1 shows where a chunk stops.
#ifndef KAMD_INL
#define KAMD_INL 3
#endif
#if KAMD_INL >= 1
#define INL1 __forceinline__
#else
#define INL1 __noinline__
#endif
#if KAMD_INL >= 2
#define INL2 __forceinline__
#else
#define INL2 __noinline__
#endif
#if KAMD_INL >= 3
#define INL3 __forceinline__
#else
#define INL3
#endif
#ifdef KAMD_TIMELINE
#define TLMARK(X, k) { if ((X).gl == 0) { LDS_AS unsigned long long* a_ = ldsPtr<unsigned long long>((X).lds + Lay<G>::TLACC); const unsigned long long t_ = wall_clock64(); a_[k] += t_ - a_[12]; a_[12] = t_; } }
#else
#define TLMARK(X, k)
#endif
#ifdef KAMD_TEST_SMALL_CAPS
	constexpr uint32_t SCAP = 4, RING = 4;
#elif defined(KAMD_TYPO) && !defined(KAMD_HIST)
	constexpr uint32_t SCAP = 32;
	// lattices over typo graphs hold 2.3 nodes per (multiplied) position: a form of fourteen positions has its predecessors more than 32 nodes back, and the
	// position-step search handed 8.5 % of c5's chunks over for that alone (profiles/r06_qq_*).  kTypoRingExtra (viterbi_kernel.hpp) is what the larger ring
	// adds to a lane group's LDS; the engine adds it to the launches of these compilations
	constexpr uint32_t RING = 64;
	static_assert(12 * (RING - 32) == kTypoRingExtra, "kTypoRingExtra");
#else
	constexpr uint32_t SCAP = 32;    // new states of one node whose scores are staged in LDS for pruning
	constexpr uint32_t RING = 32;    // most recent nodes whose state ranges are kept in LDS
#endif
	enum StageBits : uint8_t { SB_SLOT_MASK = 0x1F, SB_DEAD = 0x20, SB_MORPH_SOCKET = 0x40, SB_STATE_SOCKET = 0x80 };
	enum { RB_POSITIVE_E = 1, RB_SN_POINT = 2, RB_DIALECT = 4 };      // (RB_DIALECT: a morpheme of an allowed dialect: SearchParams::dialectCost comes off its score, PathEvaluator.hpp:231-236)

	// All LDS of the kernel is one dynamic array; per-group slices are addressed by byte offsets so that every access
	// keeps its address space (ds_* instructions) even inside non-inlined helpers.
	extern __shared__ __align__(16) uint8_t kSmem[];
	// Every LDS access goes through an address_space(3) pointer, so it is a ds_* instruction by construction: a generic
	// pointer that the optimiser cannot trace back to kSmem (e.g. merged with an HBM pointer in a select) would become a
	// flat_* access, which is slower and waits on both memory counters.
#define LDS_AS __attribute__((address_space(3)))
	typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
	template<class T> __device__ __forceinline__ LDS_AS T* ldsPtr(uint32_t off) { return (LDS_AS T*)((LDS_AS uint8_t*)kSmem + off); }
	__device__ __forceinline__ uint4 ldsLoad4(uint32_t off) { const u32x4_t v = *ldsPtr<u32x4_t>(off); return make_uint4(v.x, v.y, v.z, v.w); }
	__device__ __forceinline__ void ldsStore4(uint32_t off, const uint4 a) { u32x4_t v; v.x = a.x; v.y = a.y; v.z = a.z; v.w = a.w; *ldsPtr<u32x4_t>(off) = v; }
	// DPP rotation inside a row of 16 lanes (= one lane group when G == 16): a register-to-register cross-lane move, no LDS
	// round trip.  Which neighbour a lane receives is read off by rotating the lane index along with the data.
	template<int N> __device__ __forceinline__ uint32_t rowRor(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x120 + N, 0xF, 0xF, false); }
	// `old` is what a lane receives when its source lane is switched off (an 8-lane group whose row neighbour took another path)
	template<int N> __device__ __forceinline__ uint32_t rowRorOld(uint32_t v, uint32_t old) { return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, 0x120 + N, 0xF, 0xF, false); }
	template<int N> __device__ __forceinline__ float rowRorF(float v) { return __uint_as_float(rowRor<N>(__float_as_uint(v))); }
	__device__ __forceinline__ float rowMax16(float v)
	{
		v = fmaxf(v, rowRorF<8>(v)); v = fmaxf(v, rowRorF<4>(v)); v = fmaxf(v, rowRorF<2>(v)); v = fmaxf(v, rowRorF<1>(v));
		return v;
	}

	template<int G>
	struct Lay
	{
		static constexpr uint32_t MAXC = G < 8 ? G : 8;            // candidates per batch
		static constexpr uint32_t KEY = 0;                          // u64[QCAP]
		static constexpr uint32_t SCORE = KEY + 8 * QCAP;           // f32[QCAP]
		static constexpr uint32_t FCS = SCORE + 4 * QCAP;
#ifdef KAMD_CONG
		static constexpr uint32_t CTXQ = FCS + 4 * QCAP;           // u32[QCAP]: context id of every staged work item
		static constexpr uint32_t CAND = CTXQ + 4 * QCAP;          // 64-byte packed candidate records [MAXC]
#else
		static constexpr uint32_t CAND = FCS + 4 * QCAP;           // 64-byte packed candidate records [MAXC]
#endif
		static constexpr uint32_t STSCORE = CAND + 64 * MAXC;       // f32[SCAP]
		static constexpr uint32_t STBITS = STSCORE + 4 * SCAP;      // u8[SCAP]
		static constexpr uint32_t RBEG = STBITS + SCAP;             // u32[RING]
		static constexpr uint32_t REND = RBEG + 4 * RING;
		static constexpr uint32_t RLIVE = REND + 4 * RING;          // u32[RING]: live paths of nodes 0..j (running total)
		// G == 64 (one chunk per wave): the chunk's lattice, candidate records and path heads are kept LDS-resident, so
		// that inside the node loop only the Knlm walk reads HBM
#ifdef KAMD_HIST
		// (the history compilations run three waves per SIMD, KAMD_HIST_WPS below: 13 KB of LDS per wave)
		static constexpr uint32_t HCAP = G == 64 ? 192 : 0;
		static constexpr uint32_t NCAP = G == 64 ? 64 : 0;
		static constexpr uint32_t PCAP = G == 64 ? 64 : 0;
#else
		static constexpr uint32_t HCAP = G == 64 ? 256 : 0;         // hot state quads (+ typo cost) cached
		static constexpr uint32_t NCAP = G == 64 ? 96 : 0;          // lattice nodes cached
		static constexpr uint32_t PCAP = G == 64 ? 160 : 0;         // static candidate records cached
#endif
		static constexpr uint32_t HOT = (RLIVE + 4 * RING + 15) & ~15u;   // uint4[HCAP]
		static constexpr uint32_t HTYPO = HOT + 16 * HCAP;          // f32[HCAP]
		static constexpr uint32_t NODES = HTYPO + 4 * HCAP;         // 32 B x NCAP
		static constexpr uint32_t PACKS = NODES + 32 * NCAP;        // 48 B x (PCAP + 2): last two = the unknown-noun candidates
#ifdef KAMD_TIMELINE
		static constexpr uint32_t TLACC = (PACKS + 48 * (G == 64 ? PCAP + 2 : 0) + 15) & ~15u;   // u64[13]: 12 phase sums + last stamp
		static constexpr uint32_t SIZE = (TLACC + 112 + 15) & ~15u;
#else
		static constexpr uint32_t SIZE = (PACKS + 48 * (G == 64 ? PCAP + 2 : 0) + 15) & ~15u;
#endif
		static constexpr uint32_t LB = (64 / G) * SIZE;              // f32[2*T_MAX+1], shared by the groups
		static constexpr uint32_t TOTAL = LB + 4 * (2 * T_MAX + 1);
	};
#ifdef KAMD_HIST
	uint32_t histKernelLdsBytes(int G) { return G == 16 ? Lay<16>::TOTAL : Lay<64>::TOTAL; }      // (this compilation's own layout: its LDS caches of the chunk differ from the Knlm kernel's)
#endif
#ifndef KAMD_VARIANT
	uint32_t searchKernelLdsBytes(int G)
	{
		switch (G) { case 4: return Lay<4>::TOTAL; case 8: return Lay<8>::TOTAL; case 16: return Lay<16>::TOTAL; case 32: return Lay<32>::TOTAL; default: return Lay<64>::TOTAL; }
	}
#endif

	// candidate record as the scoring lanes see it (registers; 64 bytes in LDS)
	struct Cand
	{
		uint32_t lmId, lastSeqId, chunkOff; float userScore;     // MorphRec dwords 0..3
		int32_t combinedId; uint32_t flagsFeat, tagw, cntw;       // MorphRec dwords 4..7
		uint32_t morph, qOff, R; float additional;
		uint32_t sbw;        // sbType | firstWid of a chunked candidate is in firstWid
		uint32_t ruleBits;
		uint32_t firstWid;   // first LM id fed (lmId, or the first chunk's for chunked candidates)
		uint32_t secondWid;  // LM id of the second chunk (chunked candidates with >= 2 chunks)
		__device__ __forceinline__ uint32_t flags() const { return flagsFeat & 0xFFFF; }
		__device__ __forceinline__ uint8_t tag() const { return (uint8_t)tagw; }
		__device__ __forceinline__ uint8_t vowel() const { return (uint8_t)(tagw >> 8); }
		__device__ __forceinline__ uint8_t polar() const { return (uint8_t)(tagw >> 16); }
		__device__ __forceinline__ uint8_t socket() const { return (uint8_t)(tagw >> 24); }
		__device__ __forceinline__ uint32_t nChunks() const { return cntw & 0xFF; }
		__device__ __forceinline__ uint8_t senseId() const { return (uint8_t)(cntw >> 8); }
		__device__ __forceinline__ uint8_t special() const { return (uint8_t)(cntw >> 24); }
		// the device copy of the morpheme table carries, in `feat` / `prevFlags`, what a PATH ending in the morpheme exposes
		// to its successor (FlatModel::morphPath), not the morpheme's own form features
		__device__ __forceinline__ uint16_t leftFeat() const { return (uint16_t)(flagsFeat >> 16); }
		__device__ __forceinline__ uint8_t prevFlags() const { return (uint8_t)(cntw >> 16); }
		__device__ __forceinline__ uint8_t sbType() const { return (uint8_t)sbw; }
		__device__ __forceinline__ bool single() const { return flags() & MF_SINGLE; }
		__device__ __forceinline__ bool quoteOrBullet() const { const uint8_t s = special(); return sbType() || s == 0 || s == 1 || s == 3 || s == 4; }
	};
	__device__ __forceinline__ Cand loadCand(uint32_t ldsOff)
	{
		const uint4 a = ldsLoad4(ldsOff), b = ldsLoad4(ldsOff + 16), c = ldsLoad4(ldsOff + 32), d = ldsLoad4(ldsOff + 48);
		Cand o;
		o.lmId = a.x; o.lastSeqId = a.y; o.chunkOff = a.z; o.userScore = __uint_as_float(a.w);
		o.combinedId = (int32_t)b.x; o.flagsFeat = b.y; o.tagw = b.z; o.cntw = b.w;
		o.morph = c.x; o.qOff = c.y; o.R = c.z; o.additional = __uint_as_float(c.w);
		o.sbw = d.x; o.ruleBits = d.y; o.firstWid = d.z; o.secondWid = d.w;
		return o;
	}

	// hot quad of a state (first 16 bytes of DevState) as registers
	struct Hot
	{
		int32_t lmNode; float accScore; uint32_t w2, w3;
		__device__ __forceinline__ uint16_t leftFeat() const { return (uint16_t)w2; }
		__device__ __forceinline__ uint8_t rootId() const { return (uint8_t)(w2 >> 16); }
		__device__ __forceinline__ uint8_t spState() const { return (uint8_t)(w2 >> 24); }
		__device__ __forceinline__ uint8_t socket() const { return (uint8_t)w3; }
		__device__ __forceinline__ uint8_t prevFlags() const { return (uint8_t)(w3 >> 8); }
		__device__ __forceinline__ bool dead() const { return (w3 >> 16) & 0xFF; }
		__device__ __forceinline__ uint8_t ownKind() const { return (uint8_t)(w3 >> 24); }
	};
	__device__ __forceinline__ Hot loadHot(const DevState* st, uint32_t i)
	{
		const uint4 a = *reinterpret_cast<const uint4*>(st + i);
		Hot h; h.lmNode = (int32_t)a.x; h.accScore = __uint_as_float(a.y); h.w2 = a.z; h.w3 = a.w;
		return h;
	}
	__device__ __forceinline__ void storeState(DevState* st, uint32_t i, int32_t lmNode, float acc, float typo, uint32_t wid, uint16_t leftFeat, uint8_t rootId, uint8_t sp,
		uint8_t socket, uint8_t prevFlags, uint8_t ownKind, uint32_t parent, uint32_t morph, float fcs, uint16_t nodeId, uint16_t ownNode STATE_EXTRA(, uint32_t histPos = 0))
	{
		uint4* p = reinterpret_cast<uint4*>(st + i);
		p[0] = make_uint4((uint32_t)lmNode, __float_as_uint(acc), (uint32_t)leftFeat | ((uint32_t)rootId << 16) | ((uint32_t)sp << 24),
			(uint32_t)socket | ((uint32_t)prevFlags << 8) | ((uint32_t)ownKind << 24));
		p[1] = make_uint4(__float_as_uint(typo), wid, parent, morph);
#if defined(KAMD_SBG) || defined(KAMD_CONG)
		p[2] = make_uint4(__float_as_uint(fcs), (uint32_t)nodeId | ((uint32_t)ownNode << 16), histPos, 0);   // DevState::pad0 = ring position / context id
#else
		p[2] = make_uint4(__float_as_uint(fcs), (uint32_t)nodeId | ((uint32_t)ownNode << 16), 0, 0);
#endif
	}

	// edge (node, wid) of the Knlm trie through the bucketed hash built at load time (flat_model.hpp LmSlot)
	__device__ __forceinline__ bool lmLookup(const ModelView& M, uint32_t node, uint32_t wid, int32_t& v, float& ll)
	{
		uint32_t b = lmHashOf(node, wid) & M.lmHashMask;
		for (;;)
		{
			const uint4* p = reinterpret_cast<const uint4*>(M.lmHash + (size_t)b * 4);
			const uint4 s0 = p[0], s1 = p[1], s2 = p[2], s3 = p[3];
			const bool h0 = (s0.x == node) & (s0.y == wid), h1 = (s1.x == node) & (s1.y == wid), h2 = (s2.x == node) & (s2.y == wid), h3 = (s3.x == node) & (s3.y == wid);
			if (h0 | h1 | h2 | h3)
			{
				const uint32_t vz = h0 ? s0.z : h1 ? s1.z : h2 ? s2.z : s3.z, vw = h0 ? s0.w : h1 ? s1.w : h2 ? s2.w : s3.w;
				v = (int32_t)vz; ll = __uint_as_float(vw);
				return true;
			}
			if (s3.x == LM_SLOT_EMPTY) return false;   // slots fill front to back: a free last slot means the bucket never overflowed
			b = (b + 1) & M.lmHashMask;
		}
	}

	// KnLangModel::progress (src/Knlm.cpp:44-130); float additions in the same order
	__device__ INL3 float lmProgress(const ModelView& M, int32_t& node, uint32_t next, float acc0 = 0.f)
	{
		float acc = acc0;      // (back-off weights already collected by lmProgressChain, which hands a walk over half way; 0 otherwise)
		// the unigram record of `next` does not depend on the walk: fetched together with the first bucket probe, because
		// most walks (a context miss, then the root) would otherwise pay a second dependent round trip for it
		const LmRootRec rootRec = M.lmRoot2[next];
		for (;;)
		{
			int32_t v; float ll;
			if (node == 0)
			{
				const LmRootRec r = rootRec;
				if (r.value == 0) { if (M.lmHtxNode) node = M.lmHtxNode[next]; return acc + M.h.unkLl; }      // (history-transformed model, Knlm.cpp:61-70)
				v = r.value; ll = r.ll;
			}
			else
			{
				const LmBackoff bo = M.lmBackoff[node];      // independent of the bucket load: both are in flight together
				if (!lmLookup(M, (uint32_t)node, next, v, ll)) { acc += bo.gamma; node += bo.lower; continue; }
			}
			if (v > 0) { node += v; return acc + ll; }
			// leaf: the new state is the longest suffix context that continues with `next` (Knlm.cpp:96-128)
			int32_t cur = node;
			for (;;)
			{
				const int32_t lower = M.lmBackoff[cur].lower;
				if (!lower) break;
				cur += lower;
				int32_t lv; float l2;
				if (cur == 0) { lv = rootRec.value; if (lv > 0) { node = lv; return acc + ll; } }
				else if (lmLookup(M, (uint32_t)cur, next, lv, l2) && lv > 0) { node = cur + lv; return acc + ll; }
			}
			node = M.lmHtxNode ? M.lmHtxNode[next] : 0;      // (history-transformed model: the root's child for the transformed id, Knlm.cpp:116-126)
			return acc + ll;
		}
	}

	// developer statistics of a host (lane-emulated) build with -DKAMD_LMSTATS: which way the calls of lmProgressChain go, printed at exit
#ifdef KAMD_LMSTATS
	struct LmStats { std::atomic<unsigned long long> c[16]; ~LmStats() { fprintf(stderr, "[lmstats] calls %llu root-start %llu hit0-child %llu hit0-leaf %llu ovf0 %llu hit1-child %llu hit1-leaf %llu ovf1 %llu n2 %llu root-unk %llu root-child %llu root-leaf %llu walk %llu\n",
		c[0].load(), c[1].load(), c[2].load(), c[3].load(), c[4].load(), c[5].load(), c[6].load(), c[7].load(), c[8].load(), c[9].load(), c[10].load(), c[11].load(), c[12].load()); } };
	static LmStats gLmStats;
#define LMSTAT(k) gLmStats.c[k]++;
#else
#define LMSTAT(k) (void)0;
#endif
#ifndef KAMD_CONG
	// KnLangModel::progress once more, for a state that carries the next two nodes of its back-off chain (ModelView::lmChain: n1, n2; 0 = root): the
	// edge (node, next), the edge (n1, next), the unigram record and both back-off weights are requested TOGETHER, so the usual walk -- a miss in the
	// context, then the shorter context, then the root -- costs one memory round trip instead of three.  What this cannot settle (a bucket that
	// overflowed into its neighbour, a chain longer than three contexts, a leaf edge) goes on in lmProgress' own loop: same result, same additions in
	// the same order.
	__device__ INL3 float lmProgressChain(const ModelView& M, int32_t& node, uint32_t n1, uint32_t n2, uint32_t next)
	{
		const LmRootRec rootRec = M.lmRoot2[next];
		const uint32_t n0 = (uint32_t)node;
		float acc = 0, result = 0;
		bool walk = false; LMSTAT(0)      // hand the rest to lmProgress' loop from `node` with `acc`
		uint4 s0 = make_uint4(0, 0, 0, 0), s1 = s0, s2 = s0, s3 = s0, t0 = s0, t1 = s0, t2 = s0, t3 = s0; float g0 = 0, g1 = 0;
		if (n0)
		{
			const uint4* b0 = reinterpret_cast<const uint4*>(M.lmHash + (size_t)(lmHashOf(n0, next) & M.lmHashMask) * 4);
			const uint4* b1 = reinterpret_cast<const uint4*>(M.lmHash + (size_t)(lmHashOf(n1, next) & M.lmHashMask) * 4);
			s0 = b0[0]; s1 = b0[1]; s2 = b0[2]; s3 = b0[3];
			g0 = M.lmBackoff[n0].gamma;
			if (n1) { t0 = b1[0]; t1 = b1[1]; t2 = b1[2]; t3 = b1[3]; g1 = M.lmBackoff[n1].gamma; }
		}
		else { LMSTAT(1) }
		do
		{
			// the edge (n1, next), if the shorter context has it (n1 == 0: no bucket was loaded, nothing matches)
			const bool u0 = (t0.x == n1) & (t0.y == next), u1 = (t1.x == n1) & (t1.y == next), u2 = (t2.x == n1) & (t2.y == next), u3 = (t3.x == n1) & (t3.y == next);
			const bool hit1 = (n1 != 0) & (u0 | u1 | u2 | u3);
			const int32_t v1 = (int32_t)(u0 ? t0.z : u1 ? t1.z : u2 ? t2.z : t3.z);
			const bool full1 = (n1 != 0) & (t3.x != LM_SLOT_EMPTY);      // (the bucket overflowed: the edge may sit in the next one)
			if (n0)
			{
				// context n0
				const bool h0 = (s0.x == n0) & (s0.y == next), h1 = (s1.x == n0) & (s1.y == next), h2 = (s2.x == n0) & (s2.y == next), h3 = (s3.x == n0) & (s3.y == next);
				if (h0 | h1 | h2 | h3)
				{
					const int32_t v = (int32_t)(h0 ? s0.z : h1 ? s1.z : h2 ? s2.z : s3.z);
					result = acc + __uint_as_float(h0 ? s0.w : h1 ? s1.w : h2 ? s2.w : s3.w);
					if (v > 0) { LMSTAT(2) node = (int32_t)n0 + v; break; }
					// a leaf edge: the new state is the longest suffix context that continues with `next` (Knlm.cpp:96-128) -- n1, then n2, then the root.  The
					// bucket of (n1, next) is here already; only a chain with a third context to ask goes through the general walk (which finds the edge again)
					LMSTAT(3)
					if (hit1 && v1 > 0) { node = (int32_t)n1 + v1; break; }
					if ((n1 != 0) & ((!hit1 & full1) | (n2 != 0))) { walk = true; result = 0; break; }
					if (rootRec.value > 0) node = rootRec.value;
					else node = M.lmHtxNode ? M.lmHtxNode[next] : 0;
					break;
				}
				if (s3.x != LM_SLOT_EMPTY) { LMSTAT(4) walk = true; break; }      // (the bucket overflowed: the edge may sit in the next one)
				acc += g0;
				// context n1
				if (n1)
				{
					if (hit1)
					{
						if (v1 > 0) { LMSTAT(5) node = (int32_t)n1 + v1; result = acc + __uint_as_float(u0 ? t0.w : u1 ? t1.w : u2 ? t2.w : t3.w); }
						else { LMSTAT(6) node = (int32_t)n1; walk = true; }
						break;
					}
					if (full1) { LMSTAT(7) node = (int32_t)n1; walk = true; break; }
					acc += g1;
					if (n2) { LMSTAT(8) node = (int32_t)n2; walk = true; break; }      // (a fourth context: carry on from it)
				}
			}
			// the root (a state at the root starts here: the unigram record is the whole walk)
			if (rootRec.value == 0) { LMSTAT(9) node = M.lmHtxNode ? M.lmHtxNode[next] : 0; result = acc + M.h.unkLl; }
			else if (rootRec.value > 0) { LMSTAT(10) node = rootRec.value; result = acc + rootRec.ll; }
			else { LMSTAT(11) node = 0; walk = true; }      // (a leaf unigram: the general walk's last lines)
		} while (0);
		if (walk) { LMSTAT(12) result = lmProgress(M, node, next, acc); }
		return result;
	}
#endif

#ifdef KAMD_CONG
	__device__ __forceinline__ int32_t dot4s8(uint32_t a, uint32_t b, int32_t acc)
	{
#if defined(__HIP_DEVICE_COMPILE__)
		return __builtin_amdgcn_sdot4((int)a, (int)b, acc, false);      // v_dot4_i32_i8
#else
		for (int k = 0; k < 4; ++k) acc += (int32_t)(int8_t)(a >> (8 * k)) * (int32_t)(int8_t)(b >> (8 * k));
		return acc;
#endif
	}
	// CoNgramModel::progress, window 0, quantised (src/CoNgramModel.cpp:869-908): the score of `next` in the CURRENT context, then the context
	// moves on (progressContextNodeVl, src/CoNgramModel.hpp:306-385, over the edge hash: slot.value > 0 child offset with the child's context id in
	// the slot's ll bits, < 0 minus the context id of a leaf).  outputFirst: the rounding of the reference's batched SSE4.1 kernel
	// (src/archImpl/sse4_1.cpp:116: ((x * outputScale) * contextScale) + bias) instead of progress()'s ((x * contextScale) * outputScale) + bias.
	// progressContextNodeVl (src/CoNgramModel.hpp:306-385) over the edge hash: the context id `key` leads to from `node` (a miss at the root: context 0,
	// stay at the root; a leaf names its own context and the walk re-anchors at the longest suffix that continues with `key`); rootRec = M.lmRoot2[key]
	__device__ __forceinline__ uint32_t congWalk(const ModelView& M, int32_t& node, uint32_t key, const LmRootRec rootRec)
	{
		for (;;)
		{
			int32_t v; float cbits;
			if (node == 0)
			{
				v = rootRec.value; cbits = rootRec.ll;
				if (v == 0) return 0;
			}
			else
			{
				const LmBackoff bo = M.lmBackoff[node];
				if (!lmLookup(M, (uint32_t)node, key, v, cbits))
				{
					if (!bo.lower) return 0;
					node += bo.lower;
					continue;
				}
			}
			if (v > 0) { node += v; return __float_as_uint(cbits); }
			// leaf: its own context id; the new node is the longest suffix context that continues with `key`
			int32_t cur = node;
			for (;;)
			{
				const int32_t lower = M.lmBackoff[cur].lower;
				if (!lower) break;
				cur += lower;
				int32_t lv; float l2;
				if (cur == 0) { lv = rootRec.value; if (lv > 0) { node = lv; return (uint32_t)-v; } }
				else if (lmLookup(M, (uint32_t)cur, key, lv, l2) && lv > 0) { node = cur + lv; return (uint32_t)-v; }
			}
			node = 0;
			return (uint32_t)-v;
		}
	}
	__device__ INL3 float congStep(const ModelView& M, const CongDev& CG, int32_t& node, uint32_t& ctx, uint32_t next, bool outputFirst)
	{
		const uint32_t* a = reinterpret_cast<const uint32_t*>(CG.ctxEmb + (size_t)ctx * CG.stride);
		const uint32_t* b = reinterpret_cast<const uint32_t*>(CG.outEmb + (size_t)next * CG.stride);
		const LmRootRec rootRec = M.lmRoot2[next];      // (fetched with the rows; a two-key id below reads its own)
		const uint32_t nw = CG.dim >> 2;
		int32_t acc = 0;
		for (uint32_t k = 0; k < nw; ++k) acc = dot4s8(a[k], b[k], acc);
		const float cs = __uint_as_float(a[nw]), bias = __uint_as_float(a[nw + 1]), os = __uint_as_float(b[nw]);
		const float x = (float)acc;
		const float ll = outputFirst ? x * os * cs + bias : x * cs * os + bias;
		if (next < CG.vlTMax) ctx = congWalk(M, node, next, rootRec);
		else
		{
			// variable-length keys (cong.mdl keySize 3; CoNgramModel::progressContextNode, src/CoNgramModel.hpp:271-300): two steps, the first context id is dropped
			const uint32_t r = next - CG.vlTMax, k1 = CG.vlTMax + (r >> CG.vlBits), k2 = CG.vlTMax + (1u << CG.vlBits) + (r & ((1u << CG.vlBits) - 1));
			congWalk(M, node, k1, M.lmRoot2[k1]);
			ctx = congWalk(M, node, k2, M.lmRoot2[k2]);
		}
		return ll;
	}
#endif

#ifdef KAMD_HIST
	// ---- SkipBigram LM state beyond the Knlm node: ring of the last 8 valid word ids + write position (SbgState,
	// src/SkipBigramModel.hpp:141-182).  Rings are only ever indexed with compile-time constants or select chains, so they
	// stay in registers.
	// Global CoNgram model: the same eight words are CoNgramState<7>::history (h[0..6] the last seven valid distant words, h[7] the newest slot;
	// cong_global.hpp pushHistory); pos is unused (0).
	struct Ring { uint32_t h[8]; uint32_t pos; };
	__device__ __forceinline__ Ring loadRing(const uint32_t* base, uint32_t pos)
	{
		const uint4 a = reinterpret_cast<const uint4*>(base)[0], b = reinterpret_cast<const uint4*>(base)[1];
		Ring r; r.h[0] = a.x; r.h[1] = a.y; r.h[2] = a.z; r.h[3] = a.w; r.h[4] = b.x; r.h[5] = b.y; r.h[6] = b.z; r.h[7] = b.w; r.pos = pos;
		return r;
	}
	__device__ __forceinline__ void storeRing(uint32_t* base, const Ring& r)
	{
		reinterpret_cast<uint4*>(base)[0] = make_uint4(r.h[0], r.h[1], r.h[2], r.h[3]);
		reinterpret_cast<uint4*>(base)[1] = make_uint4(r.h[4], r.h[5], r.h[6], r.h[7]);
	}
	__device__ __forceinline__ uint32_t ringAt(const Ring& r, uint32_t i)
	{
		uint32_t v = r.h[0];
#pragma unroll
		for (uint32_t k = 1; k < 8; ++k) v = i == k ? r.h[k] : v;
		return v;
	}
	// the k-th of the last four words fed, oldest first (SbgState::getLastHistory, SkipBigramModel.hpp:161-167)
	__device__ __forceinline__ uint32_t ringLast4(const Ring& r, uint32_t k) { return ringAt(r, (r.pos + 4u + k) & 7u); }
	// LM-state equality as the path containers see it: the whole ring and its position for top-1 (SbgState::operator==,
	// SkipBigramModel.hpp:156-159); the last four words only for top-N (PathHash<SbgState>, src/SkipBigramModel.cpp:8-35)
	__device__ __forceinline__ bool sameRing(const Ring& a, const Ring& b, bool last4)
	{
		bool eq = true;
#ifdef KAMD_CONGG
		// CoNgramState<7>::operator== (src/CoNgramModel.hpp:452-461): history[3..6] -- not the newest word, not the three oldest; top-N compares the same (PathHash)
		(void)last4;
		eq = (a.h[3] == b.h[3]) & (a.h[4] == b.h[4]) & (a.h[5] == b.h[5]) & (a.h[6] == b.h[6]);
		return eq;
#endif
		if (last4)
		{
#pragma unroll
			for (uint32_t k = 0; k < 4; ++k) eq = eq & (ringLast4(a, k) == ringLast4(b, k));
		}
		else
		{
			eq = a.pos == b.pos;
#pragma unroll
			for (uint32_t k = 0; k < 8; ++k) eq = eq & (a.h[k] == b.h[k]);
		}
		return eq;
	}
	// 32-bit digest over exactly what sameRing compares: unequal digests prove unequal rings, equal digests are re-checked on the rings
	__device__ __forceinline__ uint32_t ringDigest(const Ring& r, bool last4)
	{
#ifdef KAMD_TEST_WEAK_DIGEST
		return 0;      // test build: every pair of items with equal keys reaches the exact comparison / the collision hand-over
#else
#ifdef KAMD_CONGG
		{ uint32_t dg = 0x51ED270Bu; (void)last4;
#pragma unroll
		  for (uint32_t k = 3; k < 7; ++k) { dg = (dg ^ r.h[k]) * 0x9E3779B1u; dg ^= dg >> 15; }
		  return dg; }
#endif
		uint32_t d = last4 ? 0x51ED270Bu : r.pos;
		if (last4)
		{
#pragma unroll
			for (uint32_t k = 0; k < 4; ++k) { d = (d ^ ringLast4(r, k)) * 0x9E3779B1u; d ^= d >> 15; }
		}
		else
		{
#pragma unroll
			for (uint32_t k = 0; k < 8; ++k) { d = (d ^ r.h[k]) * 0x9E3779B1u; d ^= d >> 15; }
		}
		return d;
#endif
	}
#endif

#ifdef KAMD_CONGG
	// CoNgramModel::progress with distant tokens (src/CoNgramModel.cpp:802-868) / one entry of progressMatrixWSort / WOSort (:1037-1466): a valid distant
	// word is scored as a mixture over the context and the state's seven history words (cong_global.hpp: the arithmetic, written once for oracle and
	// device), any other word as the local model scores it; then the context moves on and the word (or 0) enters the history.
	__device__ INL3 float congStepG(const ModelView& M, const CongDev& CG, const CongGDev& GG, int32_t& node, uint32_t& ctx, Ring& ring, uint32_t next, bool outputFirst, bool matrix)
	{
		CongView C;
		C.dim = CG.dim; C.stride = CG.stride; C.ctxEmb = CG.ctxEmb; C.outEmb = CG.outEmb; C.window = GG.window; C.keyBytes = GG.keyBytes;
		C.ctxConf = GG.ctxConf; C.distEmb = GG.distEmb; C.distConf = GG.distConf; C.posConf = GG.posConf; C.distMask = GG.distMask;
		float ll;
		if (C.distant(next)) ll = matrix ? congg::scoreMatrix(C, ctx, ring.h, next, outputFirst) : congg::scoreSingle(C, ctx, ring.h, next);
		else
		{
			const uint32_t* a = reinterpret_cast<const uint32_t*>(CG.ctxEmb + (size_t)ctx * CG.stride);
			const uint32_t* b = reinterpret_cast<const uint32_t*>(CG.outEmb + (size_t)next * CG.stride);
			const uint32_t nw = CG.dim >> 2;
			int32_t acc = 0;
			for (uint32_t k = 0; k < nw; ++k) acc = dot4s8(a[k], b[k], acc);
			const float cs = __uint_as_float(a[nw]), bias = __uint_as_float(a[nw + 1]), os = __uint_as_float(b[nw]);
			const float x = (float)acc;
			ll = outputFirst ? x * os * cs + bias : x * cs * os + bias;
		}
		const LmRootRec rootRec = M.lmRoot2[next];
		if (next < CG.vlTMax) ctx = congWalk(M, node, next, rootRec);
		else
		{
			const uint32_t r = next - CG.vlTMax, k1 = CG.vlTMax + (r >> CG.vlBits), k2 = CG.vlTMax + (1u << CG.vlBits) + (r & ((1u << CG.vlBits) - 1));
			congWalk(M, node, k1, M.lmRoot2[k1]);
			ctx = congWalk(M, node, k2, M.lmRoot2[k2]);
		}
		congg::pushHistory(C, ring.h, next);
		return ll;
	}
#endif

	__device__ __forceinline__ uint8_t hashSb(uint32_t type, uint32_t order)   // PathEvaluator.hpp:83-86
	{
		type &= 0xFF; order &= 0xFF;
		return (uint8_t)((((int)type << 1) ^ (int)(type >> 7) ^ (int)order) % 63 + 1);
	}

	// RuleBasedScorer::operator() (PathEvaluator.hpp:111-181)
	__device__ __forceinline__ float ruleScore(const Cand& c, uint8_t prevFlags, uint8_t sp)
	{
		float a = 0;
		const uint32_t fl = c.flags();
		if ((fl & MF_VOWEL_E) && (prevFlags & PF_IRREGULAR)) a -= 10;
		if ((fl & MF_INF_J) && (prevFlags & PF_INFLECTENDA_NP)) a -= 5;
		if ((fl & MF_BAD_PAIR_OF_L) && (prevFlags & PF_VERB_L)) a -= 7;
		if ((c.ruleBits & RB_POSITIVE_E) && !(prevFlags & PF_POSITIVE_VERB)) a -= 100;
		if ((fl & MF_CONTRACTABLE_E) && (prevFlags & PF_VERB_VOWEL)) a -= 3;
		if (c.polar() == CP_NON_ADJ && (prevFlags & PF_VA_OR_XSA)) a -= 10;
		const uint8_t special = c.special();
		if (special <= 2) { if (special != (sp & 1)) a -= 2; }
		else if (special <= 5) { if ((uint8_t)(special - 3) != ((sp >> 1) & 1)) a -= 2; }
		const uint8_t sb = c.sbType();
		if (sb == 5) a -= 5;
		if (sb && (prevFlags & PF_E_NOT_EF)) a -= 10;
		if (sb && (sp >> 2) == hashSb(sb, c.senseId())) a += 3;
		if ((c.ruleBits & RB_SN_POINT) && (prevFlags & PF_UNK_EF_SF)) a -= 5;
		return a;
	}

	__device__ __forceinline__ uint8_t nextSpState(const Cand& c, uint8_t sp)   // PathEvaluator.hpp:222-231
	{
		const uint8_t special = c.special();
		if (special == 0) sp |= 1; else if (special == 1) sp &= ~1; else if (special == 3) sp |= 2; else if (special == 4) sp &= ~2;
		if (c.sbType()) sp = (uint8_t)((sp & 3) | (hashSb(c.sbType(), (uint32_t)c.senseId() + 1) << 2));
		return sp;
	}

	// ---------------------------------------------------------------------------------------------------
	template<int G>
	struct GroupCtx
	{
		static constexpr uint64_t GMASK = G == 64 ? ~0ull : ((1ull << G) - 1);
		const ModelView& M; const SearchParams& P;
		__device__ __forceinline__ GroupCtx(const ModelView& m, const SearchParams& p) : M(m), P(p) {}
		// same lane-group state, other copies of the model / parameter views
		__device__ __forceinline__ GroupCtx(const GroupCtx& o, const ModelView& m, const SearchParams& p) : M(m), P(p)
		{
			gl = o.gl; gshift = o.gshift; lds = o.lds; nodes = o.nodes; Gn = o.Gn; str = o.str; cls = o.cls; st = o.st; stCap = o.stCap; stTop = o.stTop;
			nodeStOff = o.nodeStOff; nodeStCnt = o.nodeStCnt; nodeLive = o.nodeLive; uniq = o.uniq; nUniq = o.nUniq;
			overflow = o.overflow; pairOverflow = o.pairOverflow; stageOverflow = o.stageOverflow; scratch = o.scratch; tl = o.tl; compact = o.compact; nPE = o.nPE;
			SBG_ONLY(S = o.S;) HIST_ONLY(hist = o.hist; sscr = o.sscr;)
			TYPO_ONLY(typoAll = o.typoAll; nodeTypo = o.nodeTypo;)
			CONG_ONLY(CG = o.CG; outFirst = o.outFirst;)
			CONGG_ONLY(GG = o.GG; matrix = o.matrix;)
		}
		// SkipBigram: the model view, the chunk's state rings (parallel to st) and the lane group's item scratch
		SBG_ONLY(const SbgDev* S;) HIST_ONLY(uint32_t* hist; SbgScratch* sscr;)
		// global CoNgram model: the window sections; `matrix`: the regular candidates of this evaluation are scored by progressMatrix* (more than one path or
		// candidate), not by state.next() -- other roundings (cong_global.hpp scoreMatrix / scoreSingle)
		CONGG_ONLY(const CongGDev* GG; bool matrix;)
		TYPO_ONLY(const float* typoAll; const float* nodeTypo;)      // typo cost of every node of the batch / of the chunk's nodes
		CONG_ONLY(const CongDev* CG; bool outFirst;)                 // embedding tables; which kernel of the reference rounds the regular candidates' scores at this node
		CONG_ONLY(__device__ __forceinline__ LDS_AS uint32_t* qCtx() const { return ldsPtr<uint32_t>(lds + Lay<G>::CTXQ); })
		uint32_t gl, gshift, lds;       // lane in group, group's first lane, byte offset of the group's LDS slice
		const DevNode* nodes; uint32_t Gn;
		const uint16_t* str; const uint8_t* cls;
		DevState* st; uint32_t stCap, stTop;
		uint32_t* nodeStOff; uint32_t* nodeStCnt; uint16_t* nodeLive;   // HBM copies (far-back lookups, end node)
		const uint8_t* uniq; uint32_t nUniq;
		bool overflow, pairOverflow, stageOverflow;
		bool compact; uint32_t nPE;      // the node's items are formed over its LIVE incoming paths (scratch->live, nPE of them) instead of all E.nP (evaluateNode)
		GroupScratch* scratch;
		unsigned long long* tl;     // per-chunk timeline record (KAMD_TIMELINE builds), else unused

		__device__ __forceinline__ uint64_t ballot(bool p) const { return (__ballot(p) >> gshift) & GMASK; }
		__device__ __forceinline__ bool any(bool p) const { return ballot(p) != 0; }
		__device__ __forceinline__ uint32_t prefix(uint64_t b) const { return __popcll(b & ((1ull << gl) - 1)); }
		template<class T> __device__ __forceinline__ T bcast(T v, int srcLane) const { return __shfl(v, srcLane, G); }
		// LDS accessors (address space preserved)
		__device__ __forceinline__ LDS_AS uint64_t* qKey() const { return ldsPtr<uint64_t>(lds + Lay<G>::KEY); }
		__device__ __forceinline__ LDS_AS float* qScore() const { return ldsPtr<float>(lds + Lay<G>::SCORE); }
		__device__ __forceinline__ LDS_AS float* qFcs() const { return ldsPtr<float>(lds + Lay<G>::FCS); }
		__device__ __forceinline__ uint32_t candOff(uint32_t k) const { return lds + Lay<G>::CAND + 64 * k; }
		__device__ __forceinline__ uint32_t candQOff(uint32_t k) const { return *ldsPtr<uint32_t>(candOff(k) + 36); }
		__device__ __forceinline__ LDS_AS float* stScore() const { return ldsPtr<float>(lds + Lay<G>::STSCORE); }
		__device__ __forceinline__ LDS_AS uint8_t* stBits() const { return ldsPtr<uint8_t>(lds + Lay<G>::STBITS); }
		__device__ __forceinline__ LDS_AS uint32_t* ringBeg() const { return ldsPtr<uint32_t>(lds + Lay<G>::RBEG); }
		__device__ __forceinline__ LDS_AS uint32_t* ringEnd() const { return ldsPtr<uint32_t>(lds + Lay<G>::REND); }
		__device__ __forceinline__ LDS_AS uint32_t* ringCum() const { return ldsPtr<uint32_t>(lds + Lay<G>::RLIVE); }
		__device__ __forceinline__ const LDS_AS float* lb() const { return ldsPtr<float>(Lay<G>::LB); }
	};

	struct NodeEnv { uint32_t pBeg, nP, nLive; uint32_t nodeIdx, nodeStart; uint8_t nflags, fflags; TYPO_ONLY(float typoCost;) };

	template<int G>
	__device__ __forceinline__ Hot getHot(const GroupCtx<G>& X, uint32_t i)
	{
		if constexpr (Lay<G>::HCAP != 0)
		{
			if (i < Lay<G>::HCAP)
			{
				const uint4 a = ldsLoad4(X.lds + Lay<G>::HOT + 16 * i);
				Hot h; h.lmNode = (int32_t)a.x; h.accScore = __uint_as_float(a.y); h.w2 = a.z; h.w3 = a.w;
				return h;
			}
		}
		return loadHot(X.st, i);
	}
	template<int G>
	__device__ __forceinline__ float getTypo(const GroupCtx<G>& X, uint32_t i)
	{
		if constexpr (Lay<G>::HCAP != 0) { if (i < Lay<G>::HCAP) return ldsPtr<float>(X.lds + Lay<G>::HTYPO)[i]; }
		return X.st[i].accTypoCost;
	}
	template<int G>
	__device__ __forceinline__ void putState(GroupCtx<G>& X, uint32_t i, int32_t lmNode, float acc, float typo, uint32_t wid, uint16_t leftFeat, uint8_t rootId, uint8_t sp,
		uint8_t socket, uint8_t prevFlags, uint8_t ownKind, uint32_t parent, uint32_t morph, float fcs, uint16_t nodeId, uint16_t ownNode STATE_EXTRA(, uint32_t histPos = 0))
	{
#ifdef KAMD_POS_TRACE
		fprintf(stderr, "gput pos %u node %u parent %u score %.9g typo %g morph %u lm %d\n", i, (unsigned)nodeId, parent, (double)acc, (double)typo, morph, lmNode);
#endif
		storeState(X.st, i, lmNode, acc, typo, wid, leftFeat, rootId, sp, socket, prevFlags, ownKind, parent, morph, fcs, nodeId, ownNode STATE_EXTRA(, histPos));
		if constexpr (Lay<G>::HCAP != 0)
		{
			if (i < Lay<G>::HCAP)
			{
				ldsStore4(X.lds + Lay<G>::HOT + 16 * i, make_uint4((uint32_t)lmNode, __float_as_uint(acc),
					(uint32_t)leftFeat | ((uint32_t)rootId << 16) | ((uint32_t)sp << 24), (uint32_t)socket | ((uint32_t)prevFlags << 8) | ((uint32_t)ownKind << 24)));
				ldsPtr<float>(X.lds + Lay<G>::HTYPO)[i] = typo;
			}
		}
	}
	template<int G>
	__device__ __forceinline__ void markDead(GroupCtx<G>& X, uint32_t i)
	{
		X.st[i].dead = 1;
		if constexpr (Lay<G>::HCAP != 0) { if (i < Lay<G>::HCAP) ldsPtr<uint32_t>(X.lds + Lay<G>::HOT)[4 * i + 3] |= 1u << 16; }
	}
	template<int G>
	__device__ __forceinline__ DevNode getNode(const GroupCtx<G>& X, uint32_t i)
	{
		if constexpr (Lay<G>::NCAP != 0)
		{
			if (i < Lay<G>::NCAP)
			{
				DevNode nd;
				const uint4 a[2] = { ldsLoad4(X.lds + Lay<G>::NODES + 32 * i), ldsLoad4(X.lds + Lay<G>::NODES + 32 * i + 16) };
				__builtin_memcpy(&nd, a, sizeof(nd));
				return nd;
			}
		}
		return X.nodes[i];
	}


	// records a freshly written state of the current node for pruning / reachability (LDS; falls back to HBM when a node
	// produces more than SCAP states)
	template<int G>
	__device__ __forceinline__ void stageState(GroupCtx<G>& X, uint32_t rel, float score, uint8_t rootId, bool morphSocket, bool stateSocket)
	{
		if (rel < SCAP)
		{
			X.stScore()[rel] = score;
			X.stBits()[rel] = (uint8_t)((rootId == COMMON_ROOT ? 0 : rootId + 1u) | (morphSocket ? SB_MORPH_SOCKET : 0) | (stateSocket ? SB_STATE_SOCKET : 0));
		}
		else X.stageOverflow = true;
	}

	// One batch of regular candidates (packed records in LDS) against the incoming paths [pBeg, pBeg+nP): Qtot work items.
	// mode: 0 small container, 1 medium (4 hash buckets), 2 large (PathEvaluator.hpp:447-466).
	template<int G>
	__device__ INL1 void evalBatch(GroupCtx<G>& X, uint32_t nC, uint32_t Qtot, const NodeEnv& E, float ignoreCondScore, uint8_t ownKind, uint16_t ownFeat, int mode)
	{
		const ModelView& M = X.M;
		const bool big = Qtot > QCAP;
		const uint32_t pBeg = E.pBeg;
		const bool spaceBefore = E.nflags & NF_SPACE_BEFORE;
		TLMARK(X, 1)

		// Common case (G == 16): the whole batch fits the group once and the node uses the small container -- scores stay in
		// registers and the per-key de-duplication runs on DPP rotations instead of LDS scans.
#ifdef KAMD_NO_FAST8
		const bool fast = (G == 16) && Qtot <= (uint32_t)G && mode == 0;
#else
		const bool fast = (G == 16 || G == 8) && Qtot <= (uint32_t)G && mode == 0;
#endif
		uint64_t rKey = KINVALID; float rScore = 0, rFcs = 0, rTypo = 0;
		CONG_ONLY(uint32_t rCtx = 0;)      // context id of the item's new LM state
#ifdef KAMD_HIST
		uint32_t rDigest = 0;      // digest of the item's history ring (the ring itself goes to X.sscr)
		const bool last4 = X.P.topN > 1;   // what of the ring belongs to the container key (sameRing)
#endif

#ifndef KAMD_CONG
		// A right half of a split stem (socket, several chunks) takes the combined word id of the LATEST matching left half among the paths up to its own
		// (PathEvaluator.hpp:578-591: `firstWid` is assigned inside the loop over the paths and never reset).  Looked for path by path from every item, that is
		// quadratic in the node's paths -- a SkipBigram node of a thousand paths spent a third of its time there (profiles/r05_s_*) --: with more items than the
		// LDS queues hold, one pass over the paths per such candidate leaves every item the index of that path (0xFFFFFFFF: none) where its first-chunk score
		// will go (GroupScratch::fcs, written by the item itself at the end of its scoring).
		bool rightHalves = false;
		if (big)
		{
			for (uint32_t k = X.gl; k < nC; k += G) { const Cand c = loadCand(X.candOff(k)); rightHalves |= c.socket() && !c.single(); }
			rightHalves = X.any(rightHalves);
		}
		if (rightHalves)
		{
			const bool never = spaceBefore && !(X.P.spaceTol > 0);
			for (uint32_t k = 0; k < nC; ++k)
			{
				const Cand c = loadCand(X.candOff(k));
				if (!(c.socket() && !c.single())) continue;      // (uniform)
				const uint8_t ctag = c.tag(), csock = c.socket();
				uint32_t carry = 0xFFFFFFFFu;
				for (uint32_t jb = 0; jb < X.nPE; jb += G)
				{
					const uint32_t j = jb + X.gl;      // the j-th path the items are formed over
					uint32_t p = j;
					bool m = false;
					if (j < X.nPE)
					{
						if (X.compact) p = X.scratch->live[j];
						if (!never)
						{
							const Hot qs = getHot<G>(X, pBeg + p);
							m = !qs.dead() && qs.socket() && qs.socket() == csock && !((qs.leftFeat() & LF_PREV_ZSIOT) && (!isNNClass(ctag) || spaceBefore));
						}
					}
					const uint64_t bal = X.ballot(m);
					const uint64_t upTo = bal & (X.gl >= 63u ? ~0ull : ((2ull << X.gl) - 1ull));
					const uint32_t lastP = X.bcast(p, upTo ? 63 - (int)__builtin_clzll((unsigned long long)upTo) : 0);      // (the PATH of the latest match up to this lane)
					const uint32_t last = upTo ? lastP : carry;
					if (j < X.nPE) for (uint32_t r = 0; r < c.R; ++r) reinterpret_cast<uint32_t*>(X.scratch->fcs)[c.qOff + j * c.R + r] = last;
					const uint32_t carryP = X.bcast(p, bal ? 63 - (int)__builtin_clzll((unsigned long long)bal) : 0);
					if (bal) carry = carryP;
				}
			}
			waveSync();
		}
#endif
		// ---- scoring pass: one work item per lane -------------------------------------------------------
		for (uint32_t qb = 0; qb < Qtot; qb += G)
		{
			const uint32_t q = qb + X.gl;
			bool valid = q < Qtot;
			uint32_t k = 0;
			if (valid) { while (k + 1 < nC && q >= X.candQOff(k + 1)) ++k; }
			float cand = 0, firstChunk = 0; int32_t lmNode = 0; uint8_t rootKey = 0, sp = 0;
			HIST_ONLY(Ring ring{};)
			CONG_ONLY(uint32_t ctx = 0; float icDeferred = 0;)
			if (valid)
			{
				const Cand c = loadCand(X.candOff(k));
				const uint32_t local = q - c.qOff;
				uint32_t p = local, r = 0;
				if (c.R != 1) { p = local / c.R; r = local % c.R; }      // R > 1 only for quote / sentence-break candidates under several start states
				if (X.compact) p = X.scratch->live[p];
				const Hot ps = getHot<G>(X, pBeg + p);
				if (fast) rTypo = getTypo<G>(X, pBeg + p);      // rides along with the hot quad: the winner's is picked up by a lane read later
				const bool single = c.single();
				const uint8_t ctag = c.tag(), csock = c.socket();
				uint32_t firstWid = c.firstWid; bool widReplaced = false;
				do
				{
					if (ps.dead()) { valid = false; break; }
#ifdef KAMD_CONG
					// the candidate evaluator of the transposed search tests the sai-siot rule for regular candidates only (CoNgramModel.cpp:170-176)
					if (!csock && (ps.leftFeat() & LF_PREV_ZSIOT) && (!isNNClass(ctag) || spaceBefore)) { valid = false; break; }
					// ... pairs the right half of a split stem with paths that end in a left half ONLY, each with the combined word of ITS path
					// (CoNgramModel.cpp:262-283): no carried-over word id here
					if (csock && !single && !ps.socket()) { valid = false; break; }
					// ... and pairs everything else with socket-free paths only
					if (!(csock && !single) && ps.socket()) { valid = false; break; }
#else
					if ((ps.leftFeat() & LF_PREV_ZSIOT) && (!isNNClass(ctag) || spaceBefore)) { valid = false; break; }
#endif
					cand = ps.accScore + c.additional;
					firstChunk = c.additional;
					if (ps.socket())
					{
						if (ps.socket() != csock || single) { valid = false; break; }
						if (spaceBefore)
						{
							if (X.P.spaceTol > 0) cand -= X.P.spacePenalty; else { valid = false; break; }
						}
					}
#ifdef KAMD_CONG
					if (csock && !single) { firstWid = M.morphs[M.morphs[X.st[pBeg + p].wid].combinedId].lmId; widReplaced = true; }
					if (false)
#else
					if (csock && !single && rightHalves)
					{
						const uint32_t pp = reinterpret_cast<const uint32_t*>(X.scratch->fcs)[q];      // (the pass over the paths above)
						if (pp != 0xFFFFFFFFu) { firstWid = M.morphs[M.morphs[X.st[pBeg + pp].wid].combinedId].lmId; widReplaced = true; }
					}
					else if (csock && !single)
#endif
					{
						// the reference keeps the combined word id of the latest matching split stem for all later predecessors
						// (PathEvaluator.hpp:578-591: `firstWid` is assigned inside the loop and never reset)
						for (int32_t pp = (int32_t)p; pp >= 0; --pp)
						{
							const Hot qs = getHot<G>(X, pBeg + pp);
							if (qs.dead() || !qs.socket() || qs.socket() != csock) continue;
							if ((qs.leftFeat() & LF_PREV_ZSIOT) && (!isNNClass(ctag) || spaceBefore)) continue;
							if (spaceBefore && !(X.P.spaceTol > 0)) continue;
							firstWid = M.morphs[M.morphs[X.st[pBeg + pp].wid].combinedId].lmId; widReplaced = true;
							break;
						}
					}
					// FormEvaluator (PathEvaluator.hpp:293-310)
					if (!(ps.leftFeat() & (LF_STR_SSC | LF_TAG_SSC)))
					{
						const bool ok = featTest(ps.leftFeat() & 0x1FFF, c.vowel(), c.polar());
#ifdef KAMD_CONG
						// a regular candidate gets the penalty AFTER its first LM score: ((acc + morphScore) + ll) + ignoreCondScore (CoNgramModel.cpp:163-168)
						if (ignoreCondScore != 0) { if (!csock) icDeferred = ok ? 0 : ignoreCondScore; else cand += ok ? 0 : ignoreCondScore; }
#else
						if (ignoreCondScore != 0) cand += ok ? 0 : ignoreCondScore;
#endif
						else if (!ok) { valid = false; break; }
					}
					lmNode = ps.lmNode;
					SBG_ONLY(ring = loadRing(X.hist + 8ull * (pBeg + p), X.st[pBeg + p].pad0);)
					CONGG_ONLY(ring = loadRing(X.hist + 8ull * (pBeg + p), 0u);)
					CONG_ONLY(ctx = X.st[pBeg + p].pad0;)
					if (!(csock && single))
					{
						// prohibit <v> without <chunk> (PathEvaluator.hpp:604-608): static per candidate unless the word id was replaced above
						if (widReplaced ? (M.morphs[firstWid].tag == T_P) : ((c.flags() & MF_FIRST_WID_IS_P) != 0)) { valid = false; break; }
#if defined(KAMD_CONGG)
						// (a right half is scored by state.next(): progress(); regular candidates by the evaluation's matrix kernel or, one path and one candidate, next())
						float ll = congStepG(M, *X.CG, *X.GG, lmNode, ctx, ring, firstWid, X.outFirst && !csock, X.matrix && !csock);
						cand += ll; firstChunk += ll;
						cand += icDeferred;
#elif defined(KAMD_CONG)
						float ll = congStep(M, *X.CG, lmNode, ctx, firstWid, X.outFirst && !csock);
						cand += ll; firstChunk += ll;
						cand += icDeferred;
#else
						TLMARK(X, 7)      // (timeline builds: lane 0's item -- what came before the first LM step of the pass)
						float ll = lmProgress(M, lmNode, firstWid);
						TLMARK(X, 10)     // (... the Knlm step; the SkipBigram mixture and the rest of the pass go to phase 2)
						SBG_ONLY(ll = sbgNext(*X.S, ring.h, ring.pos, firstWid, ll);)
						cand += ll; firstChunk += ll;
#endif
						if (!single)
						{
							const uint32_t nCh = c.nChunks();
							for (uint32_t ch = 1; ch < nCh; ++ch)
							{
								const uint32_t wid = ch == 1 ? c.secondWid : M.chunkLm[c.chunkOff + ch];
								if ((c.flags() & MF_ANY_REST_WID_IS_P) && M.morphs[wid].tag == T_P) { valid = false; break; }
#if defined(KAMD_CONGG)
								ll = congStepG(M, *X.CG, *X.GG, lmNode, ctx, ring, wid, false, false);
#elif defined(KAMD_CONG)
								ll = congStep(M, *X.CG, lmNode, ctx, wid, false);
#else
								ll = lmProgress(M, lmNode, wid);
								SBG_ONLY(ll = sbgNext(*X.S, ring.h, ring.pos, wid, ll);)
#endif
								cand += ll;
							}
							if (!valid) break;
						}
					}
					// insertToPathContainer (PathEvaluator.hpp:193-251)
					sp = ps.spState();
					rootKey = ps.rootId();
					if (c.quoteOrBullet())
					{
						if (rootKey == COMMON_ROOT) sp = X.uniq[r];
						else if (r != 0) { valid = false; break; }   // a path already bound to a root is inserted once
					}
					const float rs = ruleScore(c, ps.prevFlags(), sp);
					cand = cand + rs; firstChunk = firstChunk + rs;
					if (c.ruleBits & RB_DIALECT) { cand = cand - X.P.dialectCost; firstChunk = firstChunk - X.P.dialectCost; }
					sp = nextSpState(c, sp);
				} while (0);
			}
			if (q < Qtot)
			{
				// key: LM node | new special state | previous root | candidate ; r is recoverable from q
				const uint64_t key = valid ? ((uint64_t)(uint32_t)lmNode | ((uint64_t)sp << 32) | ((uint64_t)rootKey << 40) | ((uint64_t)k << 48)) : KINVALID;
				rKey = key; rScore = cand; rFcs = firstChunk;
				CONG_ONLY(rCtx = ctx; if (!fast HIST_ONLY(|| true)) { if (big) X.scratch->ctx[q] = ctx; else X.qCtx()[q] = ctx; })      // (history compilations: the register path may hand the batch to the scanning path)
#ifdef KAMD_HIST
				// the LM state of the item beyond the Knlm node; the queues are filled on the register path too, which hands a
				// batch over to the scanning path when two digests collide
				rDigest = ringDigest(ring, last4);
				if (valid) { storeRing(X.sscr->hist[q], ring); X.sscr->pos[q] = ring.pos; }
				X.sscr->hash[q] = rDigest;
				if (false) {}
#else
				if (fast) {}
#endif
				else if (big) { X.scratch->key[q] = key; X.scratch->score[q] = cand; X.scratch->fcs[q] = firstChunk; }
				else { X.qKey()[q] = key; X.qScore()[q] = cand; X.qFcs()[q] = firstChunk; }
			}
		}
		waveSync();
		TLMARK(X, 2)

		// ---- emission pass: representatives in container iteration order, each carrying its key's winner ----
		// writes the state of key `wkey` (winner item qw of candidate k) at arena slot pos
		auto emitState = [&](uint32_t k, uint32_t qw, uint64_t wkey, float wscore, float wfcs, uint32_t pos, bool haveTypo, float parentTypo CONG_ONLY(, uint32_t wctx))
		{
			const Cand c = loadCand(X.candOff(k));
			const uint32_t local = qw - c.qOff;
			uint32_t pl = local, r = 0;
			if (c.R != 1) { pl = local / c.R; r = local % c.R; }
			if (X.compact) pl = X.scratch->live[pl];
			const uint32_t parent = pBeg + pl;
#ifdef KAMD_TYPO
			const float wtypo = (haveTypo ? parentTypo : getTypo<G>(X, parent)) + E.typoCost;      // accTypoCost + node->typoCost (PathEvaluator.hpp:235)
#else
			const float wtypo = (haveTypo ? parentTypo : getTypo<G>(X, parent)) + 0.f;
#endif
			const bool single = c.single();
			const uint8_t rootKey = (uint8_t)(wkey >> 40);
			const uint8_t newRoot = (c.quoteOrBullet() && rootKey == COMMON_ROOT) ? (uint8_t)r : rootKey;
			const uint8_t stSocket = single ? c.socket() : 0;
			const bool own = single && ownKind;
			const uint16_t lf = own ? (uint16_t)(ownFeat | (c.leftFeat() & (LF_TAG_SSC | LF_PREV_ZSIOT))) : c.leftFeat();
#ifdef KAMD_HIST
			// the ring of item qw (top-1 SkipBigram: the key's winner has, by key equality, the ring of every item of the key; global CoNgram: the key holds
			// history[3..6] only -- the WINNER's whole history is the state's, as the reference's container replaces the entry)
			const Ring er = loadRing(X.sscr->hist[qw], X.sscr->pos[qw]);
			storeRing(X.hist + 8ull * pos, er);
#endif
			putState<G>(X, pos, (int32_t)(uint32_t)wkey, wscore, wtypo, c.lastSeqId, lf, newRoot, (uint8_t)(wkey >> 32), stSocket, c.prevFlags(),
				own ? ownKind : 0, parent, c.morph, wfcs, (uint16_t)E.nodeIdx, own ? (uint16_t)E.nodeIdx : 0 SBG_ONLY(, er.pos) CONG_ONLY(, wctx));
			stageState<G>(X, pos - E.nodeStart, wscore, newRoot, c.socket() != 0, stSocket != 0);
		};
#ifdef KAMD_HIST
#ifdef KAMD_SBG
		// top-N: the container key leaves the previous root out (PathHash<SbgState>::operator==, src/SkipBigramModel.cpp:26-29)
		const uint64_t keyMask = last4 ? ~(0xFFull << 40) : ~0ull;
#else
		const uint64_t keyMask = ~0ull;      // (the generic PathHash compares the root as well, BestPathContainer.hpp:105-110)
#endif
		bool scan = !fast;      // the register path hands its batch to the scanning path when ring digests collide
#endif
		if (fast)
		{
			if constexpr (G == 16 || G == 8)
			{
				// a DPP row is 16 lanes: one group (G == 16) or two (G == 8); rl = lane in the row, partners of another group are ignored
				const uint32_t q = X.gl, rl = threadIdx.x & 15u;
				const uint32_t keyLo = (uint32_t)rKey, keyHi = (uint32_t)(rKey >> 32);
				bool rep = rKey != KINVALID;
				float best = rScore; uint32_t qw = q;
				uint32_t beaten = 0;      // top-N: items of the same key that beat this one (higher score, or equal and earlier)
#ifdef KAMD_HIST
				uint32_t firstSame = q;   // earliest item whose key and ring digest equal this one's
				const uint32_t hiMask = (uint32_t)(keyMask >> 32);
#define KAMD_SAME_ITEM(N) ((((orl ^ rl) & (16u - G)) == 0)) & (ol == keyLo) & ((((oh ^ keyHi) & hiMask) == 0)) & (rowRor<N>(rDigest) == rDigest)
#define KAMD_TRACK_FIRST firstSame = (same & (oi < firstSame)) ? oi : firstSame;
#else
#define KAMD_SAME_ITEM(N) ((((orl ^ rl) & (16u - G)) == 0)) & (ol == keyLo) & (oh == keyHi)
#define KAMD_TRACK_FIRST
#endif
				// branch-free on purpose (bitwise logic on predicates + selects): 15 short dependent steps instead of 45 exec-mask branches
#define KAMD_ROT_STEP(N) { const uint32_t orl = rowRor<N>(rl), oi = orl & (G - 1), ol = rowRorOld<N>(keyLo, 0xFFFFFFFFu), oh = rowRorOld<N>(keyHi, 0xFFFFFFFFu); const float os = rowRorF<N>(rScore); \
				const bool same = KAMD_SAME_ITEM(N); KAMD_TRACK_FIRST \
				rep = rep & !(same & (oi < q)); \
				const bool better = same & ((os > best) | ((os == best) & (oi < qw))); \
				beaten += (same & ((os > rScore) | ((os == rScore) & (oi < q)))) ? 1u : 0u; \
				best = better ? os : best; qw = better ? oi : qw; }
				KAMD_ROT_STEP(1) KAMD_ROT_STEP(2) KAMD_ROT_STEP(3) KAMD_ROT_STEP(4) KAMD_ROT_STEP(5) KAMD_ROT_STEP(6) KAMD_ROT_STEP(7) KAMD_ROT_STEP(8)
				KAMD_ROT_STEP(9) KAMD_ROT_STEP(10) KAMD_ROT_STEP(11) KAMD_ROT_STEP(12) KAMD_ROT_STEP(13) KAMD_ROT_STEP(14) KAMD_ROT_STEP(15)
#undef KAMD_ROT_STEP
#undef KAMD_SAME_ITEM
#undef KAMD_TRACK_FIRST
				TLMARK(X, 7)
#ifdef KAMD_HIST
				// the rotations compared digests: every item checks its ring against the earliest item of its digest class.
				// All checks passing makes the classes exact (equality is transitive); a single mismatch sends the whole
				// batch through the scanning path, which compares rings.
				bool clash = false;
				if (rKey != KINVALID && firstSame != q)
					clash = !sameRing(loadRing(X.sscr->hist[q], X.sscr->pos[q]), loadRing(X.sscr->hist[firstSame], X.sscr->pos[firstSame]), last4);
				scan = X.any(clash);
				if (!scan)
				{
#endif
				if (X.P.topN > 1)
				{
					// keep the N best of every key, each with its own values, in item order (top-N)
					rep = rKey != KINVALID && beaten < X.P.topN;
					qw = q; best = rScore;
				}
				const float wfcs = X.bcast(rFcs, (int)qw);
				const float wtyp = X.bcast(rTypo, (int)qw);
				CONG_ONLY(const uint32_t wctx = X.bcast(rCtx, (int)qw);)
				const uint64_t kbal = X.ballot(rep);
				if (rep)
				{
					const uint32_t pos = X.stTop + X.prefix(kbal);
					if (pos < X.stCap) emitState((uint32_t)(rKey >> 48), qw, rKey, best, wfcs, pos, true, wtyp CONG_ONLY(, wctx));
					else X.overflow = true;
				}
				X.stTop += __popcll(kbal);
				HIST_ONLY(})
			}
		}
#ifdef KAMD_HIST
		if (scan)
#else
		else
#endif
		{
			const int nBuckets = mode == 1 ? 4 : 1;
#ifdef KAMD_HIST
			// top-1: representative and winner of every container key through a hash table in the group's HBM scratch (one slot
			// per key, claimed by compare-and-swap, settled by atomic max) -- with rings in the keys a node gathers thousands of
			// items per candidate, far too many for every item to scan its candidate's list.  top-N keeps the scan (its keys hold
			// only the last four ring words, and the N-th-best pruning keeps those lists short).
			const bool hashed = X.P.topN == 1;
			// top-N: the items of a container key linked into a list through the same table (SbgScratch::next): the count of better items below walks the
			// key's own list.  Same slots, same cleaning; the key leaves the previous root out and compares the last four ring words (keyMask / last4).
			const bool listed = X.P.topN > 1 && !fast;
			// the table's live part is sized by the node: four slots per item (a power of two, at least 256).  The whole table is 1 MB per lane group -- 2 GB over
			// the groups of a launch, every atomic an L2 miss --; the part a typical node needs stays in the L2 from node to node (MI355X: 39 % of the kernel's L2
			// requests missed, 549 M atomics per 8192 sentences, profiles/r05_o_*)
			uint32_t TMASK = 255u;
			while (TMASK + 1u < 4u * Qtot && TMASK < 2u * BIGQ_SBG - 1u) TMASK = 2u * TMASK + 1u;
			if (listed)
			{
				for (uint32_t qb = 0; qb < Qtot; qb += G)
				{
					const uint32_t q = qb + X.gl;
					if (q >= Qtot) continue;
					const uint64_t key = big ? X.scratch->key[q] : X.qKey()[q];
					if (key == KINVALID) continue;
					const uint64_t mkey = key & keyMask;
					const Ring myRing = loadRing(X.sscr->hist[q], X.sscr->pos[q]); const uint32_t myDigest = X.sscr->hash[q];
					uint32_t h = (uint32_t)mkey * 0x9E3779B1u ^ (uint32_t)(mkey >> 32) * 0x85EBCA77u ^ myDigest; h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 13;
					SbgSlot* e;
					for (h &= TMASK; ; h = (h + 1) & TMASK)
					{
						e = &X.sscr->table[h];
						uint32_t o = atomicCAS(&e->owner, 0u, q + 1u);
						if (o == 0) break;                      // claimed a free slot: a new key
						o -= 1;
						if (o == q) break;
						const uint64_t ko = big ? X.scratch->key[o] : X.qKey()[o];
						if (((ko ^ key) & keyMask) == 0 && X.sscr->hash[o] == myDigest && sameRing(loadRing(X.sscr->hist[o], X.sscr->pos[o]), myRing, last4)) break;
					}
					X.sscr->next[q] = atomicExch(&e->firstInv, q + 1u);      // (push front: the order of the list does not matter, the count below is symmetric)
					X.sscr->slot[q] = h;
				}
				waveSync();
			}
			if (hashed)
			{
				for (uint32_t qb = 0; qb < Qtot; qb += G)
				{
					const uint32_t q = qb + X.gl;
					if (q >= Qtot) continue;
					const uint64_t key = big ? X.scratch->key[q] : X.qKey()[q];
					if (key == KINVALID) continue;
					const Ring myRing = loadRing(X.sscr->hist[q], X.sscr->pos[q]); const uint32_t myDigest = X.sscr->hash[q];
					uint32_t h = (uint32_t)key * 0x9E3779B1u ^ (uint32_t)(key >> 32) * 0x85EBCA77u ^ myDigest; h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 13;
					SbgSlot* e;
					for (h &= TMASK; ; h = (h + 1) & TMASK)
					{
						e = &X.sscr->table[h];
						uint32_t o = atomicCAS(&e->owner, 0u, q + 1u);
						if (o == 0) break;                      // claimed a free slot: a new key
						o -= 1;
						if (o == q) break;
						const uint64_t ko = big ? X.scratch->key[o] : X.qKey()[o];
						if (ko == key && X.sscr->hash[o] == myDigest && sameRing(loadRing(X.sscr->hist[o], X.sscr->pos[o]), myRing, false)) break;
					}
					const uint32_t sb = __float_as_uint(big ? X.scratch->score[q] : X.qScore()[q]);
					const uint32_t ord = (sb & 0x80000000u) ? ~sb : (sb | 0x80000000u);      // order-preserving image of the fp32 score
					atomicMax(&e->firstInv, 0xFFFFFFFFu - q);
					atomicMax(&e->best, ((unsigned long long)ord << 32) | (unsigned long long)(0xFFFFFFFFu - q));
					X.sscr->slot[q] = h;
				}
				waveSync();
			}
#endif
#ifdef KAMD_CONGG
			// Hash<WordLL<CoNgramState<7>>> of an item (BestPathContainer.hpp:80-85 over src/CoNgramModel.hpp:520-532): Hash<uint32_t>(node), then the last 8 BYTES of
			// history[0..6] read as one word -- four 16-bit or two 32-bit ids --, then root and special state
			auto wordHash = [&](uint64_t key, const Ring& ring) -> uint64_t
			{
				const uint64_t nv = (uint64_t)(uint32_t)key;
				uint64_t lmv = (nv * 2305843009213693951ull) ^ ((nv << 33) | (nv >> 31));
				uint64_t hw = X.GG->keyBytes == 2
					? ((uint64_t)(uint16_t)ring.h[3] | ((uint64_t)(uint16_t)ring.h[4] << 16) | ((uint64_t)(uint16_t)ring.h[5] << 32) | ((uint64_t)(uint16_t)ring.h[6] << 48))
					: ((uint64_t)ring.h[5] | ((uint64_t)ring.h[6] << 32));
				hw = (hw * 2305843009213693951ull) ^ ((hw << 31) | (hw >> 33));
				lmv = hw ^ ((lmv << 3) | (lmv >> 61));
				return (uint64_t)(((key >> 40) & 0xFF) | (((key >> 32) & 0xFF) << 8)) ^ ((lmv << 3) | (lmv >> 61));
			};
			// The reference's container past 64 entries (BucketedHashContainer::insertOptimized of the SIMD builds, BestPathContainer.hpp:316-384; the oracle's
			// contInsert has the two defects spelled out): entries 0..63 are never found again, a later item is looked up among the entries 64.. by hash BYTE and
			// compared with the entry of the FIRST half at the same offset -- duplicates of an equal state stay alive, and with the global model their other
			// history words change later scores.  A container is one candidate's (and in the medium mode one of its four buckets'); only a candidate with more
			// than 64 items can get there, and for those the insertions are replayed in item order, one item per step, the lanes holding the entries.
			bool anyBig = false;
			if (hashed && mode != 2)
			{
				for (uint32_t k = X.gl; k < nC; k += G) anyBig |= ((k + 1 < nC) ? X.candQOff(k + 1) : Qtot) - X.candQOff(k) > 64u;
				anyBig = X.any(anyBig);
			}
			const uint32_t contCap = mode == 1 ? X.P.bucketCap : 128u;
			// entries of the candidate's container in order -> sscr->hash[lo ..] (the digests are not read again once the table is built); returns their number
			auto replay = [&](uint32_t lo, uint32_t hi, int b) -> uint32_t
			{
				constexpr int R = 64 / G;
				uint32_t eSlot[R], eItem[R], sItem[R], sHb[R]; float eScore[R], sScore[R];
#pragma unroll
				for (int r = 0; r < R; ++r) { eSlot[r] = eItem[r] = sItem[r] = sHb[r] = 0; eScore[r] = sScore[r] = 0.f; }
				uint32_t size = 0;
				for (uint32_t qb = lo; qb < hi; qb += G)
				{
					const uint32_t q = qb + X.gl;
					bool act = false; uint32_t slot = 0, hb = 0; float sc = 0.f;
					if (q < hi)
					{
						const uint64_t key = big ? X.scratch->key[q] : X.qKey()[q];
						if (key != KINVALID)
						{
							const uint64_t h = wordHash(key, loadRing(X.sscr->hist[q], X.sscr->pos[q]));
							if (mode != 1 || (int)((h >> 8) & 3) == b) { act = true; hb = (uint32_t)(h & 0xFF); slot = X.sscr->slot[q]; sc = big ? X.scratch->score[q] : X.qScore()[q]; }
						}
					}
					for (uint64_t m = X.ballot(act); m; m &= m - 1)
					{
						const int l = __ffsll((unsigned long long)m) - 1;
						const uint32_t js = X.bcast(slot, l), jh = X.bcast(hb, l), j = qb + (uint32_t)l; const float jsc = X.bcast(sc, l);
						if (size < 64u)
						{
							bool found = false;
#pragma unroll
							for (int r = 0; r < R; ++r)
							{
								const bool mine = (uint32_t)r * G + X.gl < size && eSlot[r] == js;
								if (mine && jsc > eScore[r]) { eScore[r] = jsc; eItem[r] = j; }
								found |= mine;
							}
							if (!X.any(found))
							{
#pragma unroll
								for (int r = 0; r < R; ++r) if ((uint32_t)r * G + X.gl == size) { eSlot[r] = js; eScore[r] = jsc; eItem[r] = j; }
								++size;
							}
							continue;
						}
						const uint32_t n2 = size - 64u;
						bool done = false;
						if (n2 < 64u)
						{
#pragma unroll
							for (int r = 0; r < R; ++r)
							{
								if (done) continue;      // (uniform)
								const uint64_t hm = X.ballot((uint32_t)r * G + X.gl < n2 && sHb[r] == jh && eSlot[r] == js);
								if (!hm) continue;
								if ((int)X.gl == __ffsll((unsigned long long)hm) - 1 && jsc > sScore[r]) { sScore[r] = jsc; sItem[r] = j; }
								done = true;
							}
						}
						if (done || size >= contCap) continue;
						if (n2 < 64u)
						{
#pragma unroll
							for (int r = 0; r < R; ++r) if ((uint32_t)r * G + X.gl == n2) { sHb[r] = jh; sScore[r] = jsc; sItem[r] = j; }
						}
						else if (X.gl == 0) X.sscr->hash[lo + size] = j;      // (a capacity above 128 is a test knob: entries past 128 are never looked at again)
						++size;
					}
				}
#pragma unroll
				for (int r = 0; r < R; ++r)
				{
					const uint32_t i = (uint32_t)r * G + X.gl;
					if (i < size && i < 64u) X.sscr->hash[lo + i] = eItem[r];
					if (size > 64u && i < size - 64u && i < 64u) X.sscr->hash[lo + 64u + i] = sItem[r];
				}
				return size;
			};
			// (uniform) the next candidate at or after k0 whose container -- of bucket b -- reaches 64 entries; repsUpTo = sscr->next, see below
			auto nextReplayed = [&](uint32_t k0) -> uint32_t
			{
				for (uint32_t k = k0; k < nC; ++k)
				{
					const uint32_t lo = X.candQOff(k), hi = (k + 1 < nC) ? X.candQOff(k + 1) : Qtot;
					if (hi - lo > 64u && X.sscr->next[hi - 1] - (lo ? X.sscr->next[lo - 1] : 0u) >= 64u) return k;
				}
				return nC;
			};
#endif
			TLMARK(X, 9)      // (timeline builds: the key table is built -- phase 9 otherwise holds the batch formation, a fraction of a microsecond per node)
			for (int b = 0; b < nBuckets; ++b)
			{
				uint32_t emittedInBucket = 0;
#ifdef KAMD_CONGG
				uint32_t dk = nC;      // the next candidate whose container is replayed (nC: none)
				if (anyBig)
				{
					// number of container keys of bucket b among the items 0..q -> sscr->next[q] (free while the keys are not listed)
					uint32_t run = 0;
					for (uint32_t qb = 0; qb < Qtot; qb += G)
					{
						const uint32_t q = qb + X.gl;
						bool first = false;
						if (q < Qtot)
						{
							const uint64_t key = big ? X.scratch->key[q] : X.qKey()[q];
							if (key != KINVALID)
							{
								first = (0xFFFFFFFFu - atomicMax(&X.sscr->table[X.sscr->slot[q]].firstInv, 0u)) == q;
								if (first && mode == 1) first = (int)((wordHash(key, loadRing(X.sscr->hist[q], X.sscr->pos[q])) >> 8) & 3) == b;
							}
						}
						const uint64_t fb = X.ballot(first);
						if (q < Qtot) X.sscr->next[q] = run + X.prefix(fb) + (first ? 1u : 0u);
						run += __popcll(fb);
					}
					waveSync();
					for (uint32_t k = nextReplayed(0); k < nC; k = nextReplayed(k + 1))
					{
						const uint32_t lo = X.candQOff(k), hi = (k + 1 < nC) ? X.candQOff(k + 1) : Qtot;
						const uint32_t n = replay(lo, hi, b);
						if (X.gl == 0) X.sscr->next[lo] = n;      // (no count that is read again: lo is neither the last item of this candidate nor of the one before)
					}
					waveSync();
					dk = nextReplayed(0);
				}
#endif
				for (uint32_t qb = 0; qb < Qtot; qb += G)
				{
					const uint32_t q = qb + X.gl;
					bool rep = false; uint32_t qw = q; uint64_t key = KINVALID; uint32_t k = 0;
					if (q < Qtot) key = big ? X.scratch->key[q] : X.qKey()[q];
					if (key != KINVALID)
					{
						k = (uint32_t)(key >> 48);
						const uint32_t lo = X.candQOff(k), hi = (k + 1 < nC) ? X.candQOff(k + 1) : Qtot;
						rep = true;
						float best = -INFINITY; bool haveBest = false;
#ifdef KAMD_HIST
						// does item j (key kj) belong to another container key than this item?  Packed key, then digest, then the rings
						const Ring myRing = loadRing(X.sscr->hist[q], X.sscr->pos[q]); const uint32_t myDigest = X.sscr->hash[q];
						auto otherKey = [&](uint32_t j, uint64_t kj) -> bool
						{
							if (((kj ^ key) & keyMask) != 0 || X.sscr->hash[j] != myDigest) return true;
							return j != q && !sameRing(loadRing(X.sscr->hist[j], X.sscr->pos[j]), myRing, last4);
						};
#define KAMD_OTHER_KEY(j, kj) otherKey(j, kj)
#else
#define KAMD_OTHER_KEY(j, kj) kj != key
#endif
#ifdef KAMD_HIST
						if (listed)
						{
							const float sq = big ? X.scratch->score[q] : X.qScore()[q];
							uint32_t beaten = 0;
							// (the head was set by atomics of other lanes: read it the same way)
							for (uint32_t jn = atomicOr(&X.sscr->table[X.sscr->slot[q]].firstInv, 0u); jn; )
							{
								const uint32_t j = jn - 1u;
								jn = X.sscr->next[j];
								if (j == q) continue;
								const float sj = big ? X.scratch->score[j] : X.qScore()[j];
								if (sj > sq || (sj == sq && j < q)) ++beaten;
							}
							rep = beaten < X.P.topN;
						}
						else
#endif
						if (X.P.topN > 1)
						{
							const float sq = big ? X.scratch->score[q] : X.qScore()[q];
							uint32_t beaten = 0;
							for (uint32_t j = lo; j < hi; ++j)
							{
								const uint64_t kj = big ? X.scratch->key[j] : X.qKey()[j];
								if (KAMD_OTHER_KEY(j, kj) || j == q) continue;
								const float sj = big ? X.scratch->score[j] : X.qScore()[j];
								if (sj > sq || (sj == sq && j < q)) ++beaten;
							}
							rep = beaten < X.P.topN;      // qw stays q: every kept item is written with its own values
						}
#ifdef KAMD_HIST
						else if (hashed)
						{
							// (atomic reads: the slot was settled by atomics of other lanes, a plain load could be served from a stale cache line)
							SbgSlot* e = &X.sscr->table[X.sscr->slot[q]];
							rep = (0xFFFFFFFFu - atomicMax(&e->firstInv, 0u)) == q;
							qw = 0xFFFFFFFFu - (uint32_t)atomicMax(&e->best, 0ull);
						}
#endif
						else
						{
						for (uint32_t j = lo; j < hi; ++j)
						{
							const uint64_t kj = big ? X.scratch->key[j] : X.qKey()[j];
							if (KAMD_OTHER_KEY(j, kj)) continue;
							if (j < q) { rep = false; break; }
							const float sj = big ? X.scratch->score[j] : X.qScore()[j];
							if (!haveBest || sj > best) { best = sj; qw = j; haveBest = true; }
						}
						}
#undef KAMD_OTHER_KEY
						if (rep && mode == 1)
						{
							// bucket = (h >> 8) & 3 of Hash<WordLL> (BestPathContainer.hpp:80-85, 323)
#if defined(KAMD_SBG)
							// Hash<SbgState> (src/SkipBigramModel.hpp:186-201): the ring words chained onto the Knlm node
							uint64_t lmv = (uint64_t)(int64_t)(int32_t)(uint32_t)key;
#pragma unroll
							for (int w = 0; w < 8; ++w) lmv = (uint64_t)myRing.h[w] ^ ((lmv << 3) | (lmv >> 61));
#elif defined(KAMD_CONGG)
							const uint64_t hh = wordHash(key, myRing);
#elif defined(KAMD_CONG)
							// Hash<CoNgramState<0>> = Hash<uint32_t>(node) (src/CoNgramModel.hpp:505-541)
							const uint64_t nv = (uint64_t)(uint32_t)key;
							const uint64_t lmv = (nv * 2305843009213693951ull) ^ ((nv << 33) | (nv >> 31));
#else
							const uint64_t lmv = (uint64_t)(int64_t)(int32_t)(uint32_t)key;
#endif
#ifndef KAMD_CONGG
							const uint64_t hh = (uint64_t)(((key >> 40) & 0xFF) | (((key >> 32) & 0xFF) << 8)) ^ ((lmv << 3) | (lmv >> 61));
#endif
							rep = (int)((hh >> 8) & 3) == b;
						}
#ifdef KAMD_CONGG
						// a replayed container writes its own entries (below, where its first item lies)
						if (anyBig && hi - lo > 64u && X.sscr->next[hi - 1] - (lo ? X.sscr->next[lo - 1] : 0u) >= 64u) rep = false;
#endif
					}
					const uint64_t bal = X.ballot(rep);
					// mode 1 runs exactly one candidate per batch, so the per-bucket rank is the container's per-bucket fill
					const uint32_t rank = emittedInBucket + X.prefix(bal);
					const bool keep = rep && (mode == 2 || X.P.topN > 1 || rank < (mode == 1 ? X.P.bucketCap : 128u));   // a full bucket drops later keys (BestPathContainer.hpp:363-367)
					const uint64_t kbal = X.ballot(keep);
					uint32_t extra = 0, blocks = 0;
#ifdef KAMD_CONGG
					for (; dk < nC && X.candQOff(dk) < qb + G; dk = nextReplayed(dk + 1))
					{
						// the entries of a replayed container, between the states of the items before its first item and of those after its last
						const uint32_t lo = X.candQOff(dk), a = lo - qb, n = X.sscr->next[lo];
						const uint32_t base = X.stTop + (uint32_t)__popcll(kbal & ((1ull << a) - 1ull)) + blocks;
						for (uint32_t i = X.gl; i < n; i += G)
						{
							const uint32_t ew = X.sscr->hash[lo + i];
							if (base + i < X.stCap)
								emitState(dk, ew, big ? X.scratch->key[ew] : X.qKey()[ew], big ? X.scratch->score[ew] : X.qScore()[ew], big ? X.scratch->fcs[ew] : X.qFcs()[ew], base + i, false, 0.f,
									big ? X.scratch->ctx[ew] : X.qCtx()[ew]);
							else X.overflow = true;
						}
						if (X.gl >= a) extra += n;
						blocks += n;
					}
#endif
					if (keep)
					{
						const uint32_t pos = X.stTop + X.prefix(kbal) + extra;
						if (pos < X.stCap)
						{
							const uint64_t wkey = big ? X.scratch->key[qw] : X.qKey()[qw];
							const float wscore = big ? X.scratch->score[qw] : X.qScore()[qw];
							const float wfcs = big ? X.scratch->fcs[qw] : X.qFcs()[qw];
							CONG_ONLY(const uint32_t wctx = big ? X.scratch->ctx[qw] : X.qCtx()[qw];)
							emitState(k, qw, wkey, wscore, wfcs, pos, false, 0.f CONG_ONLY(, wctx));
						}
						else X.overflow = true;
					}
					X.stTop += __popcll(kbal) + blocks;
					emittedInBucket += __popcll(bal);
				}
			}
#ifdef KAMD_HIST
			if (hashed || listed)
			{
				// leave the table as it was found: every item frees the slot of its key
				waveSync();
				for (uint32_t qb = 0; qb < Qtot; qb += G)
				{
					const uint32_t q = qb + X.gl;
					if (q >= Qtot) continue;
					const uint64_t key = big ? X.scratch->key[q] : X.qKey()[q];
					if (key == KINVALID) continue;
					SbgSlot* e = &X.sscr->table[X.sscr->slot[q]];
					atomicExch(&e->owner, 0u); atomicExch(&e->firstInv, 0u); atomicExch(&e->best, 0ull);
				}
			}
#endif
		}
		X.overflow = X.any(X.overflow);
		X.stageOverflow = X.any(X.stageOverflow);
		if (X.stTop > X.stCap) X.stTop = X.stCap;
		waveSync();
		TLMARK(X, 3)
	}

	// z_coda / z_siot shortcut (PathEvaluator.hpp:389-432): copies of the qualifying incoming paths
	template<int G>
	__device__ INL1 void evalZShortcut(GroupCtx<G>& X, uint32_t zMorph, const NodeEnv& E)
	{
		const ModelView& M = X.M;
		const MorphRec cm = M.morphs[zMorph];
		const uint32_t newMorph = cm.lmId;
		const MorphRec nm = M.morphs[newMorph];
		const bool newMorphSocket = nm.socket != 0;
		const uint16_t lfMorph = nm.feat;   // path-side value in the device table
		for (uint32_t pb = 0; pb < E.nP; pb += G)
		{
			const uint32_t p = pb + X.gl;
			bool keep = false; DevState ns{};
			if (p < E.nP)
			{
				ns = X.st[E.pBeg + p];
				const uint8_t lastTag = M.morphs[ns.wid].tag;
				keep = !ns.dead && (cm.tag == T_Z_CODA ? (isJClass(lastTag) || isEClass(lastTag)) : isNNClass(lastTag));
			}
			const uint64_t bal = X.ballot(keep);
			if (keep)
			{
				const uint32_t pos = X.stTop + X.prefix(bal);
				if (pos < X.stCap)
				{
					ns.accScore += cm.userScore * X.P.typoCostWeight;
					ns.accTypoCost -= cm.userScore;
					ns.parent = E.pBeg + p; ns.morph = newMorph; ns.wid = newMorph; ns.nodeId = (uint16_t)E.nodeIdx;
					ns.leftFeat = ns.ownKind ? (uint16_t)((ns.leftFeat & (0x1FFF | LF_STR_SSC)) | (lfMorph & (LF_TAG_SSC | LF_PREV_ZSIOT))) : lfMorph;
					ns.prevFlags = nm.prevFlags;
					SBG_ONLY(storeRing(X.hist + 8ull * pos, loadRing(X.hist + 8ull * (E.pBeg + p), ns.pad0));)   // the LM state is handed on unchanged
					CONGG_ONLY(storeRing(X.hist + 8ull * pos, loadRing(X.hist + 8ull * (E.pBeg + p), 0u));)
					putState<G>(X, pos, ns.lmNode, ns.accScore, ns.accTypoCost, ns.wid, ns.leftFeat, ns.rootId, ns.spState, ns.socket, ns.prevFlags, ns.ownKind,
						ns.parent, ns.morph, ns.firstChunkScore, ns.nodeId, ns.ownNode STATE_EXTRA(, ns.pad0));
					stageState<G>(X, pos - E.nodeStart, ns.accScore, ns.rootId, newMorphSocket, ns.socket != 0);
				}
				else X.overflow = true;
			}
			X.stTop += __popcll(bal);
		}
		X.overflow = X.any(X.overflow);
		X.stageOverflow = X.any(X.stageOverflow);
		if (X.stTop > X.stCap) X.stTop = X.stCap;
		waveSync();
	}

	// PathEvaluator::operator() (PathEvaluator.hpp:347-512) for one candidate list
	template<int G>
	__device__ INL2 void evaluateNode(GroupCtx<G>& X, const NodeEnv& E, const CandStatic* cands, uint32_t ldsPack, uint32_t nCands, uint8_t ownKind, uint16_t ownFeat, float nodeLevelDiscount)
	{
		const ModelView& M = X.M;
		const SearchParams& P = X.P;
		// top-N (> 1) uses one container for every size (PathEvaluator.hpp:450-453): batched like the small one, without its 128-key cap
		const bool topn = P.topN > 1;
		const int mode = topn ? 0 : E.nLive <= P.smallMax ? 0 : E.nLive <= P.mediumMax ? 1 : 2;
		// Pruned paths keep their slots (nothing is moved, state indices stay valid), so a node's incoming range holds dead paths -- two thirds of it in a SkipBigram
		// top-3 search -- and an item formed over a dead path costs its pass a lane.  A node with more paths than the group has lanes, a quarter or more of
		// them dead, forms its items over the list of its live paths instead (path order kept: the order of the items of a key, and with it every tie, is the same)
		X.compact = false; X.nPE = E.nP;
		if (E.nP > (uint32_t)G && E.nP <= sizeof(X.scratch->live) / 4 && (uint64_t)E.nLive * 4u < (uint64_t)E.nP * 3u)
		{
			uint32_t n = 0;
			for (uint32_t pb = 0; pb < E.nP; pb += G)
			{
				const uint32_t p = pb + X.gl;
				const bool alive = p < E.nP && !getHot<G>(X, E.pBeg + p).dead();
				const uint64_t bal = X.ballot(alive);
				if (alive) X.scratch->live[n + X.prefix(bal)] = p;
				n += (uint32_t)__popcll(bal);
			}
			waveSync();
			X.compact = true; X.nPE = n;
		}
		const bool spaceBefore = E.nflags & NF_SPACE_BEFORE;
		enum { K_NONE = 0, K_SKIP = 1, K_Z = 2, K_REG = 3 };
		constexpr int MAXC = Lay<G>::MAXC;
#ifdef KAMD_CONG
		// Which kernel of the reference scores the (socket-free incoming path x regular candidate) matrix of this evaluation decides the rounding of
		// its entries (congStep).  One path and one candidate: progress().  Otherwise progressMatrixNoWindow over the m UNIQUE context ids and the n
		// UNIQUE first word ids -> qgemm::scatteredGEMMOpt<sse4_1> (src/qgemm.hpp:157-205): the specialised scatteredGEMV (output scale first) iff
		// n == 1, m >= 4 and m != 8; the baseline kernel (context scale first, like progress()) in every other case.
		{
			X.outFirst = false;
			uint32_t nReg = 0, refWid = 0xFFFFFFFFu; bool oneWid = true;
			for (uint32_t cb = 0; cb < nCands; cb += G)
			{
				const uint32_t idx = cb + X.gl;
				bool reg = false; uint32_t fw = 0;
				if (idx < nCands)
				{
					const uint4* cs = reinterpret_cast<const uint4*>(cands + idx);
					const uint4 m1 = cs[1], mx = cs[2];
					const uint32_t flags = m1.y & 0xFFFF; const uint8_t tag = (uint8_t)m1.z, sock = (uint8_t)(m1.z >> 24);
					const bool skip = (P.splitComplex && (flags & MF_HAS_COMPLEX)) || tag == T_Z_CODA || tag == T_Z_SIOT
						|| (!(flags & MF_SINGLE) && (flags & MF_HA_CONTRACTION) && E.nodeIdx && spaceBefore);
					reg = !skip && !sock && !(flags & MF_FIRST_WID_IS_P);
					fw = mx.y;
				}
				const uint64_t bal = X.ballot(reg);
				if (bal)
				{
					const uint32_t f0 = X.bcast(fw, __ffsll((unsigned long long)bal) - 1);
					if (refWid == 0xFFFFFFFFu) refWid = f0;
					if (X.any(reg && fw != refWid)) oneWid = false;
					nReg += __popcll(bal);
				}
			}
#ifdef KAMD_CONGG
			// Global model: the (path x candidate) matrix is scored by progressMatrixWOSort when it has at most 16 paths and 16 candidates (src/CoNgramModel.cpp:
			// 1470-1480): m = paths + EVERY non-empty history slot of theirs, n = candidates; by progressMatrixWSort otherwise: m = unique contexts + unique
			// history words, n = unique first words.  One path and one candidate: state.next().  (What decides is whether n == 1 and m >= 4, m != 8.)
			X.matrix = false;
			{
				uint32_t nPrevReg = 0, slots = 0;
				for (uint32_t pb = 0; pb < E.nP; pb += G)
				{
					const uint32_t p = pb + X.gl;
					bool live = false; uint32_t ns = 0;
					if (p < E.nP)
					{
						const Hot h = getHot<G>(X, E.pBeg + p);
						live = !h.dead() && !h.socket();
						if (live) { const Ring r = loadRing(X.hist + 8ull * (E.pBeg + p), 0u); for (uint32_t k = 0; k < congg::WINDOW; ++k) ns += r.h[k] ? 1u : 0u; }
					}
					nPrevReg += (uint32_t)__popcll(X.ballot(live));
					for (int d = G / 2; d; d >>= 1) ns += __shfl_xor(ns, d, G);
					slots += ns;
				}
				X.matrix = nReg && nPrevReg && !(nPrevReg == 1 && nReg == 1);
				if (X.matrix && nPrevReg <= 16 && nReg <= 16)
				{
					const uint32_t m = nPrevReg + slots;
					X.outFirst = nReg == 1 && m >= 4 && m != 8;
				}
				else if (X.matrix && oneWid)
				{
					// unique contexts + unique history words of the live socket-free paths, through the (free) key table of the item scratch: a slot is claimed
					// per distinct value (contexts and words are numbered apart), the claims counted, the slots freed again
					uint32_t TM = 255u;      // (the live part of the table follows the node: at most eight values per path, two slots per value)
					while (TM + 1u < 16u * E.nP && TM < 2u * BIGQ_SBG - 1u) TM = 2u * TM + 1u;
					uint32_t m = 0, maxProbe = 0;      // (maxProbe: the longest probe sequence of the claims; the freeing pass walks that far past freed slots)
					for (int pass = 0; pass < 2; ++pass)      // pass 0: claim and count; pass 1: free
					{
						for (uint32_t pb = 0; pb < E.nP; pb += G)
						{
							const uint32_t p = pb + X.gl;
							uint32_t mine = 0;
							if (p < E.nP)
							{
								const Hot h = getHot<G>(X, E.pBeg + p);
								if (!h.dead() && !h.socket())
								{
									const Ring r = loadRing(X.hist + 8ull * (E.pBeg + p), 0u);
									for (uint32_t k = 0; k <= congg::WINDOW; ++k)
									{
										// k < 7: history word k (0 = empty); k == 7: the path's context id
										const uint32_t v = k < congg::WINDOW ? r.h[k] : X.st[E.pBeg + p].pad0;
										if (k < congg::WINDOW && !v) continue;
										const uint32_t tagged = (v << 1 | (k < congg::WINDOW ? 1u : 0u)) + 1u;
										uint32_t probes = 0;
										for (uint32_t hsh = (tagged * 0x9E3779B1u) >> 7 & TM; ; hsh = (hsh + 1) & TM, ++probes)
										{
											SbgSlot* e = &X.sscr->table[hsh];
											if (pass == 0)
											{
												const uint32_t o = atomicCAS(&e->owner, 0u, tagged);
												if (o == 0) { ++mine; break; }
												if (o == tagged) break;
											}
											else
											{
												if (atomicCAS(&e->owner, tagged, 0u) == tagged || probes >= maxProbe) break;      // (another path's lane may have freed the value's slot already)
											}
										}
										if (pass == 0) maxProbe = probes > maxProbe ? probes : maxProbe;
									}
								}
							}
							if (pass == 0) { for (int d = G / 2; d; d >>= 1) mine += __shfl_xor(mine, d, G); m += mine; }
						}
						if (pass == 0) for (int d = G / 2; d; d >>= 1) { const uint32_t o = __shfl_xor(maxProbe, d, G); maxProbe = o > maxProbe ? o : maxProbe; }
						waveSync();
					}
					X.outFirst = m >= 4 && m != 8;
				}
			}
			if (false)
#else
			if (nReg && oneWid && E.nP >= 4)
#endif
			{
				uint32_t m = 0;
				for (uint32_t pb = 0; pb < E.nP; pb += G)
				{
					const uint32_t p = pb + X.gl;
					bool isNew = false;
					if (p < E.nP)
					{
						const Hot h = getHot<G>(X, E.pBeg + p);
						if (!h.dead() && !h.socket())
						{
							const uint32_t cx = X.st[E.pBeg + p].pad0;
							isNew = true;
							for (uint32_t j = 0; j < p; ++j)
							{
								const Hot hj = getHot<G>(X, E.pBeg + j);
								if (hj.dead() || hj.socket()) continue;
								if (X.st[E.pBeg + j].pad0 == cx) { isNew = false; break; }
							}
						}
					}
					m += __popcll(X.ballot(isNew));
				}
				X.outFirst = m >= 4 && m != 8;
			}
		}
#endif

		for (int ignoreCond = 0; ignoreCond < 2; ++ignoreCond)
		{
			uint32_t c = 0;
			while (c < nCands)
			{
				// ---- lane j classifies candidate c+j; then the group agrees on the next batch ----------------
				const uint32_t idx = c + X.gl;
				uint32_t kind = K_NONE, Q = 0, R = 1, mid = 0, sbType = 0;
				uint4 m0 = make_uint4(0, 0, 0, 0), m1 = make_uint4(0, 0, 0, 0);
				uint4 mx = make_uint4(0, 0, 0, 0);
				if (idx < nCands && X.gl < (uint32_t)MAXC)
				{
					// static record: from the LDS-resident copy when the chunk's records fit, else one dependent HBM level
					bool fromLds = false;
					if constexpr (Lay<G>::PCAP != 0)
					{
						if (ldsPack + idx < Lay<G>::PCAP + 2)
						{
							const uint32_t po = X.lds + Lay<G>::PACKS + 48 * (ldsPack + idx);
							m0 = ldsLoad4(po); m1 = ldsLoad4(po + 16); mx = ldsLoad4(po + 32);
							fromLds = true;
						}
					}
					if (!fromLds)
					{
						const uint4* cs = reinterpret_cast<const uint4*>(cands + idx);
						m0 = cs[0]; m1 = cs[1]; mx = cs[2];
					}
					mid = mx.x; sbType = mx.z;
					const uint32_t flags = m1.y & 0xFFFF; const uint8_t tag = (uint8_t)m1.z; const uint8_t special = (uint8_t)(m1.w >> 24);
					if (P.splitComplex && (flags & MF_HAS_COMPLEX)) kind = K_SKIP;
					else if (tag == T_Z_CODA || tag == T_Z_SIOT) kind = (tag == T_Z_SIOT && !(P.splitSaisiot || P.mergeSaisiot)) ? K_SKIP : K_Z;
					else if (!(flags & MF_SINGLE) && (flags & MF_HA_CONTRACTION) && E.nodeIdx && spaceBefore) kind = K_SKIP;
					else
					{
						kind = K_REG;
						const bool quote = special == 0 || special == 1 || special == 3 || special == 4;
						R = ((sbType || quote) && X.nUniq > 1) ? X.nUniq : 1;
						Q = X.nPE * R;
					}
				}
				TLMARK(X, 8)
				uint32_t nTake = 0, nC = 0, Qtot = 0, myK = 0xFFFFFFFFu, myOff = 0, zMorph = 0;
				bool zShortcut = false;
				for (int j = 0; j < MAXC; ++j)
				{
					const uint32_t kj = X.bcast(kind, j);
					const uint32_t Qj = X.bcast(Q, j);
					const uint32_t midj = X.bcast(mid, j);
					if (kj == K_NONE) break;
					if (kj == K_SKIP) { ++nTake; continue; }
					if (kj == K_Z)
					{
						if (nC) break;                       // flush the regular batch first: container results keep candidate order
						zShortcut = true; zMorph = midj; ++nTake;
						break;
					}
					if (nC && (mode != 0 || Qtot + Qj > QCAP)) break;   // only small-container nodes share a batch
					if (Qj > BIGQ) { X.pairOverflow = true; ++nTake; continue; }
					if ((uint32_t)j == X.gl) { myK = nC; myOff = Qtot; }
					++nC; Qtot += Qj; ++nTake;
					if (mode != 0 || Qtot > QCAP) break;
				}
				TLMARK(X, 9)
				if (myK != 0xFFFFFFFFu)
				{
					const uint8_t tag = (uint8_t)m1.z;
					const float additional = __uint_as_float(m0.w) + nodeLevelDiscount + X.lb()[((E.nflags & NF_LEFT_BOUNDARY) ? T_MAX : 0) + clearIrregular(tag)] * 5.f;
					const uint32_t ruleBits = ((isEClass(tag) && (E.fflags & FF_STARTS_WITH_A)) ? RB_POSITIVE_E : 0) | ((tag == T_SN && (E.nflags & NF_UFORM_ENDS_POINT)) ? RB_SN_POINT : 0)
						| ((M.morphDialect && M.morphDialect[mid]) ? RB_DIALECT : 0);
					const uint32_t o = X.candOff(myK);
					ldsStore4(o, m0); ldsStore4(o + 16, m1);
					ldsStore4(o + 32, make_uint4(mid, myOff, R, __float_as_uint(additional)));
					ldsStore4(o + 48, make_uint4(sbType, ruleBits, mx.y, mx.w));
				}
				waveSync();
				c += nTake;
				if (nC) evalBatch<G>(X, nC, Qtot, E, ignoreCond ? -10.f : 0.f, ownKind, ownFeat, mode);
				if (zShortcut) evalZShortcut<G>(X, zMorph, E);
			}
			if (X.stTop > E.nodeStart) break;
		}

		// ---- pruning (PathEvaluator.hpp:475-511): paths further than cutOff below the best of their root die.
		// Nothing is moved: dead paths keep their slot (marked in LDS and HBM) and are skipped by every consumer.
		TLMARK(X, 1)
		const uint32_t cnt = X.stTop - E.nodeStart;
#ifdef KAMD_POS_TRACE
		if (X.gl == 0) fprintf(stderr, "  general node %u nodeStart %u cnt %u stageOverflow %d nP %u pBeg %u\n", E.nodeIdx, E.nodeStart, cnt, (int)X.stageOverflow, E.nP, E.pBeg);
#endif
		if (!cnt) return;
		const bool staged = !X.stageOverflow;
		const uint32_t nRootSlots = 1 + X.nUniq;
		if constexpr (G == 16 || G == 8)
		{
			if (staged && cnt <= (uint32_t)G)
			{
				// common case: the node's new paths fit the group once -- one LDS read per lane, maxima per root by DPP
				const bool in = X.gl < cnt;
				uint8_t bits = 0; float sc = 0;
				if (in) { bits = X.stBits()[X.gl]; sc = X.stScore()[X.gl]; }
				const uint32_t slot = bits & SB_SLOT_MASK;
				const bool alive = in && !(bits & SB_DEAD);
				bool kill = false;
				if (G == 16 && !topn)
				{
					for (uint32_t rs = 0; rs < nRootSlots; ++rs)
					{
						const bool mine = alive && slot == rs;
						const float mx = rowMax16((mine && !(bits & SB_MORPH_SOCKET)) ? sc : -INFINITY);
						if (mine && sc + P.cutOff < mx) kill = true;
					}
				}
				else
				{
					// threshold = the N-th best score of the root (PathEvaluator.hpp:488-503): a path dies iff at least N
					// socket-free paths of its root lie more than cutOff above it
					// (top-1 is N = 1; the group id is part of the compared word, so the other group of an 8-lane pair never matches)
					const float lim = sc + P.cutOff;
					const uint32_t grp = (threadIdx.x & 15u) / G;
					const uint32_t mineKey = alive ? (slot | (grp << 8)) : 0xFFFFu;
					const uint32_t qual = (alive && !(bits & SB_MORPH_SOCKET)) ? (slot | (grp << 8)) : 0xFFFEu;
					uint32_t above = 0;
#define KAMD_PRUNE_STEP(N) { const uint32_t oq = rowRorOld<N>(qual, 0xFFFEu); const float os = rowRorF<N>(sc); above += ((oq == mineKey) & (lim < os)) ? 1u : 0u; }
					KAMD_PRUNE_STEP(1) KAMD_PRUNE_STEP(2) KAMD_PRUNE_STEP(3) KAMD_PRUNE_STEP(4) KAMD_PRUNE_STEP(5) KAMD_PRUNE_STEP(6) KAMD_PRUNE_STEP(7) KAMD_PRUNE_STEP(8)
					KAMD_PRUNE_STEP(9) KAMD_PRUNE_STEP(10) KAMD_PRUNE_STEP(11) KAMD_PRUNE_STEP(12) KAMD_PRUNE_STEP(13) KAMD_PRUNE_STEP(14) KAMD_PRUNE_STEP(15)
#undef KAMD_PRUNE_STEP
					kill = alive && above >= (topn ? P.topN : 1u);
				}
				if (kill) { X.stBits()[X.gl] = bits | SB_DEAD; markDead<G>(X, E.nodeStart + X.gl); }
				waveSync();
				TLMARK(X, 4)
				return;
			}
		}
		if (topn)
		{
			// general sizes.  A path dies iff at least N socket-free paths of its root lie more than cutOff above it, i.e. iff the N-th best socket-free
			// score of the root (counted with multiplicity) exceeds its limit.  That threshold is found per root in at most N passes over the node's new paths
			// ("the best score below the previous one" + how often it occurs), each a strided loop with a wave reduction: O(N x paths / G) -- where every path
			// counting the paths above it was O(paths^2 / G), which for BASELINE config 3's nodes with thousands of new paths (SkipBigram keys) was 87 % of
			// the slowest chunks' time (profiles/r05_b_timeline_c3_sbg_8k.txt).  What the passes read -- score, root slot, dead / socket bits -- is first
			// compacted into the group's item scratch (free between two evaluations) when the paths are not staged in LDS: one dependent read of
			// DevState + MorphRec per path instead of one per path and pass.
			const bool compact = !staged && cnt <= BIGQ;
			if (compact)
			{
				for (uint32_t b = 0; b < cnt; b += G)
				{
					const uint32_t i = b + X.gl;
					if (i >= cnt) continue;
					const DevState* t = &X.st[E.nodeStart + i];
					X.scratch->score[i] = t->accScore;
					X.scratch->key[i] = (uint64_t)((t->rootId == COMMON_ROOT ? 0u : t->rootId + 1u) | (t->dead ? 0x100u : 0u) | (M.morphs[t->morph].socket != 0 ? 0x200u : 0u));
				}
				waveSync();
			}
			// (score, root slot, dead, morpheme socket) of new path i
			auto pathOf = [&](uint32_t i, float& sc, uint32_t& slot, bool& dead, bool& msock)
			{
				if (staged) { const uint8_t bits = X.stBits()[i]; sc = X.stScore()[i]; slot = bits & SB_SLOT_MASK; dead = bits & SB_DEAD; msock = bits & SB_MORPH_SOCKET; }
				else if (compact) { const uint32_t w = (uint32_t)X.scratch->key[i]; sc = X.scratch->score[i]; slot = w & 0xFFu; dead = (w & 0x100u) != 0; msock = (w & 0x200u) != 0; }
				else { const DevState* t = &X.st[E.nodeStart + i]; sc = t->accScore; slot = t->rootId == COMMON_ROOT ? 0 : t->rootId + 1u; dead = t->dead; msock = M.morphs[t->morph].socket != 0; }
			};
			for (uint32_t rs = 0; rs < nRootSlots; ++rs)
			{
				float bound = INFINITY, thr = -INFINITY; uint32_t have = 0; bool first = true;
				for (uint32_t round = 0; round < P.topN; ++round)
				{
					float mx = -INFINITY; uint32_t c = 0;
					for (uint32_t b = 0; b < cnt; b += G)
					{
						const uint32_t i = b + X.gl;
						if (i >= cnt) continue;
						float sc; uint32_t slot; bool dead, msock;
						pathOf(i, sc, slot, dead, msock);
						if (dead || msock || slot != rs || !(first || sc < bound)) continue;
						if (sc > mx) { mx = sc; c = 1; } else if (sc == mx) ++c;
					}
					float all = mx;
					for (int d = G / 2; d; d >>= 1) all = fmaxf(all, __shfl_xor(all, d, G));
					uint32_t call = (mx == all) ? c : 0u;
					for (int d = G / 2; d; d >>= 1) call += __shfl_xor(call, d, G);
					if (!call) break;                 // fewer than N socket-free paths of this root: nothing of it dies
					have += call;
					if (have >= P.topN) { thr = all; break; }
					bound = all; first = false;
				}
				if (thr == -INFINITY) continue;
				for (uint32_t b = 0; b < cnt; b += G)
				{
					const uint32_t i = b + X.gl;
					if (i >= cnt) continue;
					float sc; uint32_t slot; bool dead, msock;
					pathOf(i, sc, slot, dead, msock);
					if (dead || slot != rs || !(sc + P.cutOff < thr)) continue;
					if (staged) X.stBits()[i] = X.stBits()[i] | SB_DEAD;
					markDead<G>(X, E.nodeStart + i);
				}
			}
			waveSync();
			TLMARK(X, 4)
			return;
		}
		for (uint32_t rs = 0; rs < nRootSlots; ++rs)
		{
			float mx = -INFINITY; bool anyOfRoot = false;
			for (uint32_t b = 0; b < cnt; b += G)
			{
				const uint32_t i = b + X.gl;
				if (i < cnt)
				{
					float sc; uint32_t slot; bool dead, msock;
					if (staged) { const uint8_t bits = X.stBits()[i]; sc = X.stScore()[i]; slot = bits & SB_SLOT_MASK; dead = bits & SB_DEAD; msock = bits & SB_MORPH_SOCKET; }
					else { const DevState* s = &X.st[E.nodeStart + i]; sc = s->accScore; slot = s->rootId == COMMON_ROOT ? 0 : s->rootId + 1u; dead = s->dead; msock = M.morphs[s->morph].socket != 0; }
					if (!dead && slot == rs) { anyOfRoot = true; if (!msock) mx = fmaxf(mx, sc); }
				}
			}
			for (int d = G / 2; d; d >>= 1) mx = fmaxf(mx, __shfl_xor(mx, d, G));
			if (!X.any(anyOfRoot)) continue;
			for (uint32_t b = 0; b < cnt; b += G)
			{
				const uint32_t i = b + X.gl;
				if (i < cnt)
				{
					if (staged)
					{
						const uint8_t bits = X.stBits()[i];
						if (!(bits & SB_DEAD) && (uint32_t)(bits & SB_SLOT_MASK) == rs && X.stScore()[i] + P.cutOff < mx) { X.stBits()[i] = bits | SB_DEAD; markDead<G>(X, E.nodeStart + i); }
					}
					else
					{
						DevState* s = &X.st[E.nodeStart + i];
						const uint32_t slot = s->rootId == COMMON_ROOT ? 0 : s->rootId + 1u;
						if (!s->dead && slot == rs && s->accScore + P.cutOff < mx) markDead<G>(X, E.nodeStart + i);   // (also in the LDS copy of the hot quad, G == 64)
					}
				}
			}
		}
		waveSync();
		TLMARK(X, 4)
	}

#if !defined(KAMD_VARIANT) || defined(KAMD_HIST)    // (the end stage after the EOS transition does not depend on the LM type: one copy, in the Knlm translation unit -- and one in each of the compilations that run it inside the search kernel, finishPathsSolo)
	// libstdc++'s std::sort restated for the end-node candidate list (the reference sorts it with an unstable
	// std::sort, PathEvaluator.hpp:1359-1368; equal keys must land where introsort puts them).  Runs on one lane.
	__device__ __forceinline__ bool endLess(const EndCand& a, const EndCand& b)
	{
		if (a.rootId < b.rootId) return true;
		if (a.rootId > b.rootId) return false;
		if (a.sp < b.sp) return true;
		if (a.sp > b.sp) return false;
		return a.score > b.score;
	}
	__device__ __forceinline__ void insertionSortEnd(EndCand* v, int lo, int hi, bool guarded)
	{
		for (int i = lo; i < hi; ++i)
		{
			const EndCand val = v[i];
			if (guarded && endLess(val, v[0])) { for (int j = i; j > 0; --j) v[j] = v[j - 1]; v[0] = val; }
			else { int j = i; while (endLess(val, v[j - 1])) { v[j] = v[j - 1]; --j; } v[j] = val; }
		}
	}
	__device__ __forceinline__ void sortEndCands(EndCand* v, int n)
	{
		if (n <= 16) { insertionSortEnd(v, 1, n, true); return; }
		// introsort: median-of-three quick partitions down to 16-element runs (heap sort when the depth budget runs out),
		// then the final insertion sort -- std::__introsort_loop / std::__final_insertion_sort
		int stLo[40], stHi[40], stDepth[40]; int sp = 0;
		int depth = 0; for (int t = n; t > 1; t >>= 1) ++depth; depth *= 2;
		stLo[sp] = 0; stHi[sp] = n; stDepth[sp] = depth; ++sp;
		while (sp)
		{
			--sp;
			int lo = stLo[sp], hi = stHi[sp], dl = stDepth[sp];
			while (hi - lo > 16)
			{
				if (dl == 0)
				{
					// std::__partial_sort(first, last, last): make_heap + sort_heap with std::__adjust_heap
					const int len = hi - lo; EndCand* a = v + lo;
					auto adjust = [&](int hole, int length, EndCand val)
					{
						const int top = hole;
						int child = hole;
						while (child < (length - 1) / 2)
						{
							child = 2 * (child + 1);
							if (endLess(a[child], a[child - 1])) --child;
							a[hole] = a[child]; hole = child;
						}
						if ((length & 1) == 0 && child == (length - 2) / 2) { child = 2 * (child + 1); a[hole] = a[child - 1]; hole = child - 1; }
						int parent = (hole - 1) / 2;
						while (hole > top && endLess(a[parent], val)) { a[hole] = a[parent]; hole = parent; parent = (hole - 1) / 2; }
						a[hole] = val;
					};
					for (int parent = (len - 2) / 2; parent >= 0; --parent) adjust(parent, len, a[parent]);
					for (int last = len - 1; last > 0; --last) { const EndCand val = a[last]; a[last] = a[0]; adjust(0, last, val); }
					break;
				}
				--dl;
				const int first = lo, mid = lo + (hi - lo) / 2, ia = first + 1, ic = hi - 1;
				int med;
				if (endLess(v[ia], v[mid])) { if (endLess(v[mid], v[ic])) med = mid; else if (endLess(v[ia], v[ic])) med = ic; else med = ia; }
				else if (endLess(v[ia], v[ic])) med = ia; else if (endLess(v[mid], v[ic])) med = ic; else med = mid;
				{ const EndCand t = v[first]; v[first] = v[med]; v[med] = t; }
				int i = first + 1, j = hi;
				for (;;)
				{
					while (endLess(v[i], v[first])) ++i;
					--j;
					while (endLess(v[first], v[j])) --j;
					if (!(i < j)) break;
					const EndCand t = v[i]; v[i] = v[j]; v[j] = t;
					++i;
				}
				stLo[sp] = i; stHi[sp] = hi; stDepth[sp] = dl; ++sp;
				hi = i;
			}
		}
		insertionSortEnd(v, 1, 16, true);
		insertionSortEnd(v, 16, n, false);
	}

	__device__ __forceinline__ uint32_t unifyMorpheme(const ModelView& M, uint32_t m)   // PathEvaluator.hpp:1054-1058
	{
		if (m >= M.h.vocabSize || M.morphs[m].combinedId != (int32_t)m) return m;
		return M.morphs[m].lmId;
	}

	// generateTokenList (PathEvaluator.hpp:1038-1157) for one end candidate; single lane. Returns the token count or < 0.
	__device__ INL3 int backTrace(const ModelView& M, const SearchParams& P, const DevNode* nodes, const DevState* st, uint32_t endParent, DevToken* out, uint32_t cap, uint32_t* chain, uint32_t chainCap)
	{
		// walk the parent chain once (newest first), then emit oldest first
		uint32_t nSteps = 0;
		for (uint32_t s = endParent; ; )
		{
			const uint32_t par = st[s].parent;
			if (par == 0xFFFFFFFFu) break;
			if (nSteps >= chainCap) return -3;
			chain[nSteps++] = s;
			s = par;
		}
		if (!nSteps) return 0;
		int nTok = 0;
		uint32_t prevIdx = st[chain[nSteps - 1]].parent;
		for (uint32_t step = nSteps; step-- > 0;)
		{
			const uint32_t s = chain[step];
			const DevState cur = st[s];
			const DevState prev = st[prevIdx];
			const DevNode g = nodes[cur.nodeId];
			const MorphRec mm = M.morphs[cur.morph];
			const float scoreDiff = cur.accScore - prev.accScore;
			float typoDiff = cur.accTypoCost - prev.accTypoCost;
			const bool single = mm.flags & MF_SINGLE;
			const bool saisiotSplit = P.splitSaisiot && (mm.flags & MF_SAISIOT);
			const uint32_t numNew = (saisiotSplit || !single) ? mm.nChunks : 1;
			const float firstScore = cur.firstChunkScore + typoDiff * P.typoCostWeight;
			const float restScores = numNew > 1 ? (scoreDiff - cur.firstChunkScore) / (float)(numNew - 1) : 0.f;
			typoDiff /= (float)numNew;
			auto emit = [&](uint32_t morph, uint32_t b, uint32_t e, float sc, uint8_t ownKind, uint32_t ownA, uint32_t ownLen) -> bool
			{
				if ((uint32_t)nTok >= cap) return false;
				DevToken t; t.morph = morph; t.begin = (uint16_t)b; t.end = (uint16_t)e; t.wordScore = sc; t.typoCost = typoDiff;
				t.ownKind = ownKind; t.ownA = ownA; t.ownLen = (uint16_t)ownLen; t.pad = 0;
				out[nTok++] = t;
				return true;
			};
			auto chunkTok = [&](uint32_t c, float sc) -> bool
			{
				return emit(unifyMorpheme(M, M.chunkMorph[mm.chunkOff + c]), g.startPos + M.chunkPos[2 * (mm.chunkOff + c)], g.startPos + M.chunkPos[2 * (mm.chunkOff + c) + 1], sc, 0, 0, 0);
			};
			bool ok = true;
			if (saisiotSplit || (!single && !mm.socket))
			{
				for (uint32_t c = 0; c < numNew && ok; ++c) ok = chunkTok(c, c == 0 ? firstScore : restScores);
				if (ok && nTok) out[nTok - 1].end = g.endPos;
			}
			else if (single)
			{
				const uint8_t ok2 = cur.ownKind; uint32_t oa = 0, ol = 0;
				if (ok2)
				{
					const DevNode on = nodes[cur.ownNode];
					if (ok2 == 1) { oa = on.uformOff; ol = on.uformLen; }
					else if (ok2 == 2) { oa = on.form; ol = M.forms[on.form].len; }
					else { oa = on.startPos; ol = on.endPos - on.startPos; }
				}
				ok = emit(unifyMorpheme(M, cur.morph), g.startPos, g.endPos, firstScore, ok2, oa, ol);
			}
			else
			{
				if (!nTok) return -2;
				DevToken& b = out[nTok - 1];
				b.morph = (uint32_t)M.morphs[b.morph].combinedId;
				b.end = (uint16_t)(g.startPos + M.chunkPos[2 * mm.chunkOff + 1]);
				b.wordScore = firstScore; b.typoCost = typoDiff;
				for (uint32_t c = 1; c < numNew && ok; ++c) ok = chunkTok(c, restScores);
				if (ok) out[nTok - 1].end = g.endPos;
			}
			if (!ok) return -1;
			prevIdx = s;
		}
		return nTok;
	}

#endif
	// End node, first half (PathEvaluator.hpp:1320-1358): EOS transition of every surviving path -> end-candidate list for k_finish_paths.
	template<int G>
	__device__ __noinline__ bool finishChunk(GroupCtx<G>& X, uint32_t chunk, bool openEnding, DevChunkResult* res)      // (false: the end candidates did not fit into the arena's tail)
	{
		const ModelView& M = X.M;
		const uint32_t Gn = X.Gn;
		const DevNode en = getNode<G>(X, Gn - 1);
		const uint32_t firstPrev = Gn - 1 - en.prev;
		const uint32_t pBeg = X.nodeStOff[firstPrev];
		const uint32_t nP = (en.prev && en.nPrev) ? X.nodeStOff[firstPrev + en.nPrev - 1] + X.nodeStCnt[firstPrev + en.nPrev - 1] - pBeg : 0;
		// candidates go to the unused tail of the chunk's state arena, followed by room for the back-trace chain (<= Gn steps)
		EndCand* endBuf = reinterpret_cast<EndCand*>(X.st + X.stTop);
		const uint64_t freeBytes = (uint64_t)(X.stCap - X.stTop) * sizeof(DevState);
		const uint32_t endCap = freeBytes > (uint64_t)Gn * 4 ? (uint32_t)((freeBytes - (uint64_t)Gn * 4) / sizeof(EndCand)) : 0u;
		uint32_t nEnd = 0; bool endOverflow = false;
		for (uint32_t pb = 0; pb < nP; pb += G)
		{
			const uint32_t p = pb + X.gl;
			bool ok = false; DevState ps{}; float c = 0, first = 0;
			if (p < nP)
			{
				ps = X.st[pBeg + p];
				const MorphRec pm = M.morphs[ps.morph];
				ok = !ps.socket && !ps.dead;
				if (ok && !(pm.flags & MF_SINGLE) && pm.nChunks <= (pm.socket ? 2u : 1u) && pm.vowel != CV_NONE) ok = false;   // isMatched(nullptr, vowel)
				if (ok && pm.tag == T_Z_SIOT) ok = false;
				if (ok)
				{
					c = ps.accScore;
					if (!openEnding)
					{
						int32_t ln = ps.lmNode;
#if defined(KAMD_CONGG)
						uint32_t ectx = ps.pad0;
						{ Ring er = loadRing(X.hist + 8ull * (pBeg + p), 0u); first = congStepG(M, *X.CG, *X.GG, ln, ectx, er, 1u, false, false); }      // state.next(eos)
#elif defined(KAMD_CONG)
						uint32_t ectx = ps.pad0;
						first = congStep(M, *X.CG, ln, ectx, 1u, false);
#else
						first = lmProgress(M, ln, 1);
#endif
						SBG_ONLY({ Ring er = loadRing(X.hist + 8ull * (pBeg + p), ps.pad0); first = sbgNext(*X.S, er.h, er.pos, 1u, first); })
						c += first;
						if (ps.spState & 1) c -= 2;
						if (ps.spState & 2) c -= 2;
					}
				}
			}
			const uint32_t mult = (ok && ps.rootId == COMMON_ROOT) ? X.nUniq : (ok ? 1u : 0u);
			uint32_t incl = mult;
			for (int d = 1; d < G; d <<= 1) { const uint32_t v = __shfl_up(incl, d, G); if ((int)X.gl >= d) incl += v; }
			const uint32_t base = nEnd + incl - mult;
			for (uint32_t r = 0; r < mult; ++r)
			{
				if (base + r < endCap)
				{
					EndCand e; e.score = c; e.fcs = first; e.typo = ps.accTypoCost; e.parent = pBeg + p; e.pad = 0;
					if (ps.rootId == COMMON_ROOT) { e.rootId = (uint8_t)r; e.sp = X.uniq[r]; } else { e.rootId = ps.rootId; e.sp = ps.spState; }
					endBuf[base + r] = e;
				}
				else endOverflow = true;
			}
			nEnd += X.bcast(incl, G - 1);
		}
		endOverflow = X.any(endOverflow);
		if (X.gl == 0)
		{
			res->nEnd = nEnd; res->endOff = X.stTop; res->nPaths = 0;
			res->status = endOverflow ? CS_ERR_STATE_OVERFLOW : CS_OK;
		}
		return !endOverflow;
	}

#if !defined(KAMD_VARIANT) || defined(KAMD_HIST)
	// sort + selection + back-trace of one chunk (PathEvaluator.hpp:1359-1418), the work of ONE thread, in three steps with the output ranges handed out between them:
	// by wave prefix sums + one atomic per wave in k_finish_paths (one thread per chunk, a kernel of its own), by the chunk's own atomics in finishPathsSolo
	// (called by the search kernel for a chunk it has just searched, so that the chunk's state arena is free for the group's next chunk).
	struct FinishSel { uint32_t nEnd = 0, nBase = 0, Gn = 0, perGroup = 0, nSel = 0, bestOnly = 0xFFFFFFFFu; DevState* st = nullptr; EndCand* endBuf = nullptr; uint32_t* chain = nullptr; const uint8_t* uniq = nullptr; };
	__device__ __forceinline__ void finishSelect(const BatchView& B, const WorkView& W, const SearchParams& P, uint32_t chunk, const DevChunkResult* res, DevState* st, FinishSel& F)
	{
		F.nEnd = res->nEnd; F.nBase = W.nodeBase[chunk]; F.Gn = W.nNodes[chunk];
		F.st = st;
		F.endBuf = reinterpret_cast<EndCand*>(st + res->endOff);
		F.chain = reinterpret_cast<uint32_t*>(F.endBuf + F.nEnd);
		F.uniq = B.spStates + B.spOff[chunk];
		const uint32_t nEnd = F.nEnd; EndCand* endBuf = F.endBuf;
		sortEndCands(endBuf, (int)nEnd);
		// distinct (root, state) groups: the sort above made every group contiguous (its key starts with root and state), so they are counted at
		// their boundaries -- one pass instead of comparing every candidate with all earlier ones
		uint32_t numUniq = 0;
		for (uint32_t a = 0; a < nEnd; ++a) if (!a || endBuf[a].rootId != endBuf[a - 1].rootId || endBuf[a].sp != endBuf[a - 1].sp) ++numUniq;
		F.perGroup = numUniq ? (2 * P.topN + numUniq - 1) / numUniq : 0;   // ceil(topN*2 / numUniq)
		// paths this chunk hands on: the first perGroup candidates of every (root, state) group
		for (uint32_t a = 0, startIdx = 0; a < nEnd; ++a)
		{
			if (a && (endBuf[a].rootId != endBuf[a - 1].rootId || endBuf[a].sp != endBuf[a - 1].sp)) startIdx = a;
			if (a - startIdx < F.perGroup) ++F.nSel;
		}
		// The only chunk of its text under top-1: of the 2 N paths the reference hands on (PathEvaluator.hpp:1359-1418) only the best one can become the
		// analysis -- Kiwi::analyze sorts what insertPathIntoResults kept by score and cuts to N (Kiwi.cpp:1143-1158), and with no second chunk nothing
		// is combined with the rest.  That is the first of the selected paths in the host's order: highest score, the earlier one of equal scores (its sort
		// of these few paths is an insertion sort).  The groups' first candidates are their best ones, so it is the best of those.  One back-trace, one
		// path and its tokens over PCIe instead of two.
		if (P.topN == 1 && (B.chunkFlags[chunk] & 2) && F.nSel > 1)
		{
			for (uint32_t a = 0; a < nEnd; ++a)
			{
				if (a && endBuf[a].rootId == endBuf[a - 1].rootId && endBuf[a].sp == endBuf[a - 1].sp) continue;
				if (F.bestOnly == 0xFFFFFFFFu || endBuf[a].score > endBuf[F.bestOnly].score) F.bestOnly = a;
			}
			F.nSel = 1;
		}
	}
	// the back-traces of the selected candidates: path headers at outPaths[pathOff ..], token records into the chunk's own token region (tokTop of them)
	__device__ __forceinline__ void finishTrace(const ModelView& M, const WorkView& W, const SearchParams& P, uint32_t chunk, const FinishSel& F, uint32_t pathOff, uint32_t& status, DevToken*& tok, uint32_t& tokTop, uint32_t& nPaths)
	{
		tok = W.tokens + W.tokenBase[chunk];
		const uint32_t tokCap = (uint32_t)(W.tokenBase[chunk + 1] - W.tokenBase[chunk]);
		for (uint32_t a = 0, startIdx = 0; a < F.nEnd; ++a)
		{
			if (a && (F.endBuf[a].rootId != F.endBuf[a - 1].rootId || F.endBuf[a].sp != F.endBuf[a - 1].sp)) startIdx = a;
			if (a - startIdx >= F.perGroup) continue;
			if (F.bestOnly != 0xFFFFFFFFu && a != F.bestOnly) continue;
			const int nt = backTrace(M, P, W.nodes + F.nBase, F.st, F.endBuf[a].parent, tok + tokTop, tokCap - tokTop, F.chain, F.Gn);
			if (nt < 0) { status = CS_ERR_TOKEN_OVERFLOW; break; }
			DevPathHeader ph;
			ph.score = F.endBuf[a].score; ph.tokOff = tokTop; ph.nTokens = (uint16_t)nt;
			ph.prevState = F.uniq[F.endBuf[a].rootId]; ph.curState = F.endBuf[a].sp;
			W.outPaths[pathOff + nPaths++] = ph;
			tokTop += (uint32_t)nt;
		}
		if (status != CS_OK) tokTop = 0;
	}
	__device__ __forceinline__ void finishCopyTokens(const WorkView& W, const DevToken* tok, uint32_t tokTop, uint32_t tokOff)
	{
		const uint2* src = reinterpret_cast<const uint2*>(tok);            // 24-byte records, 8-byte aligned
		uint2* dst = reinterpret_cast<uint2*>(W.outTokens + tokOff);
		for (uint32_t i = 0; i < 3 * tokTop; ++i) dst[i] = src[i];
	}
	// the whole stage for one chunk by the calling thread alone (no cross-lane operation: lane groups of a wavefront get here at different times)
	__device__ __noinline__ void finishPathsSolo(const ModelView& M, const BatchView& B, const WorkView& W, const SearchParams& P, uint32_t chunk, DevState* st)
	{
		DevChunkResult* res = &W.results[chunk];
		if (res->status != CS_OK) { atomicAdd(&W.outCounters[2], 1u); return; }
		FinishSel F;
		finishSelect(B, W, P, chunk, res, st, F);
		uint32_t status = CS_OK;
		const uint32_t pathOff = F.nSel ? atomicAdd(&W.outCounters[0], F.nSel) : 0u;
		if (pathOff + F.nSel > W.outPathCap) status = CS_ERR_PATH_OVERFLOW;
		DevToken* tok = nullptr; uint32_t tokTop = 0, nPaths = 0;
		if (status == CS_OK) finishTrace(M, W, P, chunk, F, pathOff, status, tok, tokTop, nPaths);
		const uint32_t tokOff = tokTop ? atomicAdd(&W.outCounters[1], tokTop) : 0u;
		if (status == CS_OK && tokOff + tokTop > W.outTokCap) status = CS_ERR_TOKEN_OVERFLOW;
		if (status == CS_OK) finishCopyTokens(W, tok, tokTop, tokOff);
		res->status = status; res->nPaths = status == CS_OK ? nPaths : 0;
		res->pathOff = pathOff; res->tokOff = tokOff; res->nTok = status == CS_OK ? tokTop : 0;
		if (status >= 16) atomicAdd(&W.outCounters[2], 1u);
	}
#endif

#ifndef KAMD_VARIANT
	// `stride`: lanes between two active threads of a wave (64 = one chunk per wave): the stage is serial and branchy per chunk, so chunks that
	// share a wavefront run at the sum of their paths -- few active lanes per wave spread them over the SIMDs instead
	__global__ void __launch_bounds__(64) k_finish_paths(ModelView M, BatchView B, WorkView W, SearchParams P, uint32_t chunkBegin, uint32_t chunkCount, uint32_t stride)
	{
		// (no early return: every lane of the wave takes part in the prefix sums that hand out the output ranges)
		const uint32_t lane = threadIdx.x & 63u;
		const bool mine = (threadIdx.x % stride) == 0;
		const uint32_t t = mine ? blockIdx.x * (64 / stride) + threadIdx.x / stride : 0xFFFFFFFFu;
		const uint32_t chunk = chunkBegin + (t < chunkCount ? t : 0u);
		DevChunkResult* res = &W.results[chunk];
		const bool active = t < chunkCount && res->status == CS_OK;
		FinishSel F;
		if (active) finishSelect(B, W, P, chunk, res, W.states + (W.stateAt ? W.stateAt[chunk] : W.stateBase[chunk]), F);      // (stateAt: where the chunk's arena lies after it grew)
		const uint32_t nSel = F.nSel;
		// output range of the path headers: wave prefix sum + one atomic per wave
		uint32_t status = CS_OK;
		uint32_t pathOff;
		{
			uint32_t incl = nSel;
			for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d, 64); if (lane >= d) incl += v; }
			const uint32_t total = __shfl(incl, 63, 64);
			uint32_t base = 0;
			if (lane == 0 && total) base = atomicAdd(&W.outCounters[0], total);
			base = __shfl(base, 0, 64);
			pathOff = base + incl - nSel;
			if (active && pathOff + nSel > W.outPathCap) status = CS_ERR_PATH_OVERFLOW;
		}
		DevToken* tok = nullptr; uint32_t tokTop = 0, nPaths = 0;
		if (active && status == CS_OK) finishTrace(M, W, P, chunk, F, pathOff, status, tok, tokTop, nPaths);
		// output range of the token records, then the copy out of the chunk's arena
		uint32_t tokOff;
		{
			uint32_t incl = tokTop;
			for (uint32_t d = 1; d < 64; d <<= 1) { const uint32_t v = __shfl_up(incl, d, 64); if (lane >= d) incl += v; }
			const uint32_t total = __shfl(incl, 63, 64);
			uint32_t base = 0;
			if (lane == 0 && total) base = atomicAdd(&W.outCounters[1], total);
			base = __shfl(base, 0, 64);
			tokOff = base + incl - tokTop;
			if (active && status == CS_OK && tokOff + tokTop > W.outTokCap) status = CS_ERR_TOKEN_OVERFLOW;
		}
		if (active && status == CS_OK) finishCopyTokens(W, tok, tokTop, tokOff);
		if (t < chunkCount && res->status == CS_OK)
		{
			res->status = status; res->nPaths = status == CS_OK ? nPaths : 0;
			res->pathOff = pathOff; res->tokOff = tokOff; res->nTok = status == CS_OK ? tokTop : 0;
		}
		// chunks of the batch that ended in a scratch overflow (the host searches them again with larger capacities): one counter, so that
		// a run without any costs the host four bytes of D2H instead of the status array
		{
			const unsigned long long bad = __ballot(mine && t < chunkCount && res->status >= 16);
			if (lane == 0 && bad) atomicAdd(&W.outCounters[2], (uint32_t)__popcll(bad));
		}
	}
#endif

	// The chunk's state arena is full: carry on in one twice as large taken from the batch's pool (WorkView::poolTop; one atomic add per growth -- the pool is
	// append-only, an arena that was left is not handed out again).  The first `keep` states and their history words move; state indices are arena-relative and
	// stay what they are.  false: no pool, or the pool is used up (the chunk then ends with an overflow status and the host re-runs it with larger arenas).
	template<int G>
	__device__ __noinline__ bool growArena(GroupCtx<G>& X, const WorkView& W, uint32_t chunk, uint32_t keep)
	{
		if (!W.poolTop) return false;
		const uint64_t newCap = 2ull * X.stCap;
		if (newCap > 0x7FFFFFFFull) return false;
		uint32_t atLo = 0, atHi = 0;
		if (X.gl == 0) { const unsigned long long a = atomicAdd(W.poolTop, (unsigned long long)newCap); atLo = (uint32_t)a; atHi = (uint32_t)(a >> 32); }
		const uint64_t at = (uint64_t)X.bcast(atLo, 0) | ((uint64_t)X.bcast(atHi, 0) << 32);
		if (at + newCap > W.poolCap) return false;
		DevState* ns = W.states + (W.poolBase + at);
		{
			const uint4* s4 = reinterpret_cast<const uint4*>(X.st); uint4* d4 = reinterpret_cast<uint4*>(ns);
			static_assert(sizeof(DevState) == 48, "three 16-byte words per state");
			for (uint64_t k = X.gl; k < 3ull * keep; k += G) d4[k] = s4[k];
		}
#ifdef KAMD_HIST
		{
			SBG_ONLY(uint32_t* nh = X.S->hist + 8ull * (W.poolBase + at);)
			CONGG_ONLY(uint32_t* nh = X.GG->hist + 8ull * (W.poolBase + at);)
			const uint4* s4 = reinterpret_cast<const uint4*>(X.hist); uint4* d4 = reinterpret_cast<uint4*>(nh);
			for (uint64_t k = X.gl; k < 2ull * keep; k += G) d4[k] = s4[k];
			X.hist = nh;
		}
#endif
		X.st = ns; X.stCap = (uint32_t)newCap;
		if (X.gl == 0)
		{
			if (!W.slotCap) W.stateAt[chunk] = W.poolBase + at;
			else { const size_t slot = (size_t)blockIdx.x * (64 / G) + X.gshift / G; W.slotTable[2 * slot] = W.poolBase + at; W.slotTable[2 * slot + 1] = newCap; }
		}
		waveSync();
		return true;
	}

	template<int G>
	__device__ INL3 void searchChunk(GroupCtx<G>& X, const BatchView& B, const WorkView& W, uint32_t chunk, uint32_t resumeAt = 0xFFFFFFFFu)
	{
		const ModelView& M = X.M;
		const SearchParams& P = X.P;
		DevChunkResult* res = &W.results[chunk];
		// (slot mode: no k_finish_paths pass counts the chunks that ended with an error status -- every exit does it itself)
		if (res->status != CS_OK) { if (X.gl == 0) { res->nPaths = 0; HIST_ONLY(if (W.slotCap) atomicAdd(&W.outCounters[2], 1u);) } return; }
		const uint32_t cOff = B.charOff[chunk];
		const uint32_t nBase = W.nodeBase[chunk];
		X.nodes = W.nodes + nBase; X.Gn = W.nNodes[chunk];
		X.str = B.chars + cOff; X.cls = B.cls + cOff;
		X.st = W.states + W.stateBase[chunk]; X.stCap = (uint32_t)(W.stateBase[chunk + 1] - W.stateBase[chunk]); X.stTop = 0;
		SBG_ONLY(X.hist = X.S->hist + 8ull * W.stateBase[chunk];)
		CONGG_ONLY(X.hist = X.GG->hist + 8ull * W.stateBase[chunk];)
#ifdef KAMD_HIST
		if (W.slotCap)
		{
			// the lane group's own arena (WorkView::slotCap), used by one chunk after the other
			const size_t slot = (size_t)blockIdx.x * (64 / G) + X.gshift / G;
			uint64_t at = (uint64_t)slot * W.slotCap; uint32_t cap = W.slotCap;
			if (W.slotTable[2 * slot + 1]) { at = W.slotTable[2 * slot]; cap = (uint32_t)W.slotTable[2 * slot + 1]; }      // (an earlier chunk of this group grew into the pool: the group keeps that arena)
			X.st = W.states + at; X.stCap = cap;
			SBG_ONLY(X.hist = X.S->hist + 8ull * at;)
			CONGG_ONLY(X.hist = X.GG->hist + 8ull * at;)
		}
#endif
		TYPO_ONLY(X.nodeTypo = X.typoAll + nBase;)
		X.nodeStOff = W.nodeStateOff + nBase; X.nodeStCnt = W.nodeStateCnt + nBase; X.nodeLive = W.tmpIdx + 2ull * nBase;
		X.uniq = B.spStates + B.spOff[chunk]; X.nUniq = B.spOff[chunk + 1] - B.spOff[chunk];
		X.overflow = false; X.pairOverflow = false;
		const uint32_t Gn = X.Gn;
		const bool openEnding = B.chunkFlags[chunk] & 1;
		uint8_t* reach = W.reach + nBase;
		if (X.nUniq + 1 > SB_SLOT_MASK) { if (X.gl == 0) { res->status = CS_ERR_PATH_OVERFLOW; res->nPaths = 0; HIST_ONLY(if (W.slotCap) atomicAdd(&W.outCounters[2], 1u);) } return; }
		// a chunk the position-step kernel (viterbi_pos.inc) worked on before: nodes [0, resume) are done -- their states, state ranges, live counts and
		// reachable flags are in HBM -- and this kernel carries on at node `resume` (Gn - 1: only the end stage is left)
		// (resumeAt: k_pos_path carrying on by itself; otherwise the node is in DevChunkResult::pad, bits 24..31 = why it was handed over, developer statistics)
		const uint32_t resume = resumeAt != 0xFFFFFFFFu ? resumeAt : (res->pad & 0xFFFFFFu);
		if (resume == kPosChunkDone) return;
#ifdef KAMD_TIMELINE
		unsigned long long* tl = W.beacon ? reinterpret_cast<unsigned long long*>(W.beacon) + 16ull * chunk : nullptr;
		const unsigned long long tlClk0 = clock64();
		if (tl && X.gl == 0) { tl[0] = wall_clock64(); tl[3] = ((unsigned long long)blockIdx.x << 32) | X.Gn; }
		if (X.gl == 0) { LDS_AS unsigned long long* a_ = ldsPtr<unsigned long long>(X.lds + Lay<G>::TLACC); for (int k = 0; k < 12; ++k) a_[k] = 0; a_[12] = wall_clock64(); }
#endif

		if constexpr (Lay<G>::NCAP != 0)
		{
			// chunk prologue: lattice nodes and static candidate records become LDS-resident (coalesced 16-byte loads)
			const uint32_t nN = Gn < Lay<G>::NCAP ? Gn : Lay<G>::NCAP;
			const uint4* src = reinterpret_cast<const uint4*>(X.nodes);
			for (uint32_t k = X.gl; k < 2 * nN; k += G) ldsStore4(X.lds + Lay<G>::NODES + 16 * k, src[k]);
			const DevNode lastN = X.nodes[Gn - 1];
			uint32_t nPk = lastN.packOff + lastN.candCnt;
			if (nPk > Lay<G>::PCAP) nPk = Lay<G>::PCAP;
			const uint4* psrc = reinterpret_cast<const uint4*>(W.packs + W.packBase[chunk]);
			for (uint32_t k = X.gl; k < 3 * nPk; k += G) ldsStore4(X.lds + Lay<G>::PACKS + 16 * k, psrc[k]);
			const uint4* usrc = reinterpret_cast<const uint4*>(M.unkPacks);
			for (uint32_t k = X.gl; k < 6; k += G) ldsStore4(X.lds + Lay<G>::PACKS + 16 * (3 * Lay<G>::PCAP + k), usrc[k]);
			waveSync();
		}
		uint32_t cumLive = 1;    // live paths of nodes 0..i-1 (group-uniform)
		if (!resume)
		{
		// start node (PathEvaluator.hpp:1224-1226)
		if (X.gl == 0)
		{
			const MorphRec m0 = M.morphs[0];
			putState<G>(X, 0, M.h.bosNode, 0.f, 0.f, 0, m0.feat, COMMON_ROOT, 0, 0, m0.prevFlags, 0, 0xFFFFFFFFu, 0, 0.f, 0, 0);
			HIST_ONLY({ const Ring z{}; storeRing(X.hist, z); })   // SbgState() / CoNgramState(): empty history, position 0
			X.nodeStOff[0] = 0; X.nodeStCnt[0] = 1; X.nodeLive[0] = 1;
			X.ringBeg()[0] = 0; X.ringEnd()[0] = 1; X.ringCum()[0] = 1;
		}
		X.stTop = 1;
		for (uint32_t k = X.gl; k < Gn; k += G) reach[k] = k == 0 ? 1 : 0;
		}
		else if (resume + 1 < Gn)
		{
			// the LDS ring of the last RING nodes, rebuilt from the HBM tables (running live totals from an arbitrary base: only differences inside the
			// window are read, and a window that starts at node 0 starts at 0)
			const uint32_t j0 = resume > RING ? resume - RING : 0u;
			uint32_t cum = 0;
			for (uint32_t j = j0; j < resume; ++j)
			{
				cum += X.nodeLive[j];
				if (X.gl == 0) { const uint32_t o = X.nodeStOff[j]; X.ringBeg()[j & (RING - 1)] = o; X.ringEnd()[j & (RING - 1)] = o + X.nodeStCnt[j]; X.ringCum()[j & (RING - 1)] = cum; }
			}
			cumLive = cum;
			X.stTop = X.nodeStOff[resume - 1] + X.nodeStCnt[resume - 1];      // (before the hot-quad copy below, which is sized by it)
			if constexpr (Lay<G>::HCAP != 0)
			{
				// (one chunk per wave: the hot quads of the first HCAP states are read from their LDS copies)
				const uint32_t nH = X.stTop < Lay<G>::HCAP ? X.stTop : Lay<G>::HCAP;
				for (uint32_t k = X.gl; k < nH; k += G)
				{
					ldsStore4(X.lds + Lay<G>::HOT + 16 * k, *reinterpret_cast<const uint4*>(X.st + k));
					ldsPtr<float>(X.lds + Lay<G>::HTYPO)[k] = X.st[k].accTypoCost;
				}
			}
		}
		if (resume) X.stTop = X.nodeStOff[resume - 1] + X.nodeStCnt[resume - 1];
#ifdef KAMD_POS_TRACE
		if (X.gl == 0) fprintf(stderr, "general: chunk %u resume %u stTop %u Gn %u\n", chunk, resume, X.stTop, Gn);
#endif
		waveSync();

		const CandStatic* unkPacks = reinterpret_cast<const CandStatic*>(M.unkPacks);
		const CandStatic* packs = W.packs + W.packBase[chunk];
		for (uint32_t i = resume ? resume : 1u; i + 1 < Gn; ++i)
		{
			const DevNode node = getNode<G>(X, i);      // (prefetching it one node ahead cost 8 VGPRs and was slower at every batch size)
			NodeEnv E;
			const uint32_t firstPrev = i - node.prev, lastPrev = firstPrev + node.nPrev - 1;
			if (i - firstPrev + 1 < RING)
			{
				// state ranges and running live totals of the last RING nodes are in LDS: three independent reads
				E.pBeg = X.ringBeg()[firstPrev & (RING - 1)];
				E.nP = X.ringEnd()[lastPrev & (RING - 1)] - E.pBeg;
				E.nLive = X.ringCum()[lastPrev & (RING - 1)] - (firstPrev ? X.ringCum()[(firstPrev - 1) & (RING - 1)] : 0u);
			}
			else
			{
				E.pBeg = X.nodeStOff[firstPrev];
				E.nP = X.nodeStOff[lastPrev] + X.nodeStCnt[lastPrev] - E.pBeg;
				E.nLive = 0;
				for (uint32_t j = firstPrev; j <= lastPrev; ++j) E.nLive += X.nodeLive[j];
			}
			E.nflags = node.nflags; E.fflags = node.fflags; E.nodeIdx = i;
			const uint32_t nodeStart = X.stTop;
			E.nodeStart = nodeStart;
			X.stageOverflow = false;
			float ws = 0;
			if (!node.uformLen && node.form != NOFORM && node.flen && node.spaceErrors) ws = -P.spacePenalty * (float)node.spaceErrors;
#ifdef KAMD_TYPO
			E.typoCost = X.nodeTypo[i];
			const float baseDiscount = ws + (-E.typoCost * P.typoCostWeight);
#else
			const float baseDiscount = ws + (-0.f * P.typoCostWeight);   // whitespaceDiscount + typoDiscount (PathEvaluator.hpp:366-371)
#endif

			TLMARK(X, 0)
			const uint8_t ownKind = node.uformLen ? 1 : 0; const uint16_t ownFeat = node.ownFeat;
			// up to three candidate lists per node, evaluated through ONE inlined copy of evaluateNode:
			//   pass 0  the form's candidates, or the two unknown-word candidates of a formless node
			//   pass 1  forms whose candidates are all partial morphemes also get an unknown proper-noun reading (PathEvaluator.hpp:1277-1287)
			//   pass 2  a node that left the lattice disconnected gets the unknown-word candidates (PathEvaluator.hpp:1286-1299)
			for (int pass = 0; pass < 3; ++pass)
			{
				const CandStatic* cl; uint32_t clLds, clN; uint8_t ok; uint16_t of; float disc;
				if (pass == 0)
				{
					if (node.form != NOFORM)
					{
						cl = packs + node.packOff; clLds = (node.packOff + node.candCnt <= Lay<G>::PCAP) ? node.packOff : 0xFFFF0000u; clN = node.candCnt;
						ok = ownKind; of = ownFeat; disc = baseDiscount + 0.f;
					}
					else
					{
						const float emo = (X.cls[node.uformOff] & 0x80) ? -10.f : 0.f;
						cl = unkPacks; clLds = Lay<G>::PCAP; clN = 2; ok = ownKind; of = ownFeat;
						// UnkFormScorer::operator() (src/UnkFormScorer.h:40-58): the character model's score of the form (k_unk_chr) under Match::oovChrModel, else the length rule
						// (CoNgram compilations only: the reference loads the character model quantised next to a CoNgram model, and so does the loader here --
						// the Knlm / SkipBigram / typo kernels keep their code)
#ifdef KAMD_CONG
						if (W.unkChr) disc = baseDiscount + (W.unkChr[(uint32_t)(X.nodes - W.nodes) + i] - P.oovChrBias);
						else
#endif
						disc = baseDiscount + (emo - ((float)node.uformLen * P.oovRuleScale + P.oovRuleBias));
					}
				}
				else if (pass == 1)
				{
					if (node.form == NOFORM) break;
					if (!(node.nflags & NF_ALL_PARTIAL)) continue;
					const FormRec f = M.forms[node.form];
					const uint16_t* fs = M.formChars + f.charOff;
					of = featMask(fs, f.len) & 0x1FFF;
					if (f.flags & FF_ENDS_WITH_SSC) of |= LF_STR_SSC;
					cl = unkPacks + 1; clLds = Lay<G>::PCAP + 1; clN = 1; ok = 2;
#ifdef KAMD_CONG
					if (W.unkChr) disc = baseDiscount + ((W.unkChrForm ? W.unkChrForm[(uint32_t)(X.nodes - W.nodes) + i] : M.formUnkChr[node.form]) - P.oovChrBias);
					else
#endif
					disc = baseDiscount + -((float)f.len * P.oovRuleScale + P.oovRuleBias);
				}
				else
				{
					// reachable[i] and the forward re-scan of the persistent flags (PathEvaluator.hpp:1159-1176, 1286-1299)
					const uint32_t cntNow = X.stTop - nodeStart;
					bool anyFree = false;
					for (uint32_t b = 0; b < cntNow; b += G)
					{
						const uint32_t k = b + X.gl;
						if (k < cntNow)
						{
							if (!X.stageOverflow) { const uint8_t bits = X.stBits()[k]; if (!(bits & (SB_DEAD | SB_STATE_SOCKET))) anyFree = true; }
							else { const DevState* s = &X.st[nodeStart + k]; if (!s->dead && !s->socket) anyFree = true; }
						}
					}
					anyFree = X.any(anyFree);
					if (X.gl == 0) reach[i] = anyFree ? 1 : 0;
					if (anyFree) break;
					uint32_t dc = 0;
					if (X.gl == 0)
					{
						for (uint32_t k = i + 1; k < Gn; ++k)
						{
							const DevNode nk = getNode<G>(X, k);
							uint8_t r = 0;
							if (nk.prev) for (uint32_t pj = k - nk.prev, e = pj + nk.nPrev; pj < e; ++pj) if (reach[pj]) { r = 1; break; }
							reach[k] = r;
						}
						dc = reach[Gn - 1] ? 0 : 1;
					}
					dc = X.bcast(dc, 0);
					if (!dc) break;
					const uint32_t len = node.endPos - node.startPos;
					of = featMask(X.str + node.startPos, len) & 0x1FFF;
					if (len)
					{
						const uint16_t c = X.str[node.endPos - 1];
						const uint8_t tag = (isLowSurrogate(c) || isHighSurrogate(c)) ? (uint8_t)T_SH : (uint8_t)(X.cls[node.endPos - 1] & 0x3F);
						if (tag == T_SSC) of |= LF_STR_SSC;
					}
					const float emo = (X.cls[node.startPos] & 0x80) ? -10.f : 0.f;
					cl = unkPacks; clLds = Lay<G>::PCAP; clN = 2; ok = 3;
#ifdef KAMD_CONG
					if (W.unkChr) disc = baseDiscount + (W.unkChr[(uint32_t)(X.nodes - W.nodes) + i] - P.oovChrBias);
					else
#endif
					disc = baseDiscount + (emo - ((float)len * P.oovRuleScale + P.oovRuleBias));
				}
				evaluateNode<G>(X, E, cl, clLds, clN, ok, of, disc);
			}
			TLMARK(X, 6)
			if (X.overflow && !X.pairOverflow && W.poolTop)
			{
				// the arena is full: a larger one from the batch's pool, and the node is evaluated again from its first state (nothing of it is recorded yet).
				// (a real function call on COPIES, like the end stage below: no address of X or of a kernel argument may escape from the node loop)
				const ModelView Mc = X.M; const SearchParams Pc = X.P; const WorkView Wc = W;
				GroupCtx<G> Y(X, Mc, Pc);
				SBG_ONLY(const SbgDev Sc = *X.S; Y.S = &Sc;)
				CONGG_ONLY(const CongGDev Gc = *X.GG; Y.GG = &Gc;)
				if (growArena<G>(Y, Wc, chunk, nodeStart))
				{
					X.st = Y.st; X.stCap = Y.stCap; HIST_ONLY(X.hist = Y.hist;)
					X.overflow = false; X.stTop = nodeStart;
					--i; continue;
				}
			}
			// node bookkeeping: state range + live count (LDS ring and HBM)
			{
				const uint32_t cntAll = X.stTop - nodeStart;
				uint32_t live = 0;
				for (uint32_t b = 0; b < cntAll; b += G)
				{
					const uint32_t k = b + X.gl;
					bool alive = false;
					if (k < cntAll) alive = !X.stageOverflow ? !(X.stBits()[k] & SB_DEAD) : !X.st[nodeStart + k].dead;
					live += __popcll(X.ballot(alive));
				}
				if (X.gl == 0)
				{
					X.ringBeg()[i & (RING - 1)] = nodeStart; X.ringEnd()[i & (RING - 1)] = X.stTop; X.ringCum()[i & (RING - 1)] = cumLive + live;
					X.nodeStOff[i] = nodeStart; X.nodeStCnt[i] = cntAll; X.nodeLive[i] = (uint16_t)(live > 0xFFFF ? 0xFFFF : live);
				}
				cumLive += live;
			}
			waveSync();
			TLMARK(X, 5)
			if (X.overflow || X.pairOverflow) break;
		}
		if (X.overflow || X.pairOverflow)
		{
			if (X.gl == 0) { res->status = X.overflow ? CS_ERR_STATE_OVERFLOW : CS_ERR_PAIR_OVERFLOW; res->nPaths = 0; HIST_ONLY(if (W.slotCap) atomicAdd(&W.outCounters[2], 1u);) }
			return;
		}
#ifdef KAMD_TIMELINE
		if (tl && X.gl == 0) { const unsigned long long clkNow_ = clock64(); tl[1] = wall_clock64(); LDS_AS unsigned long long* a_ = ldsPtr<unsigned long long>(X.lds + Lay<G>::TLACC); for (int k = 0; k < 11; ++k) tl[4 + k] = a_[k]; tl[15] = clkNow_ - tlClk0; }
#endif
		for (;;)
		{
			// the end stage is a real function call; it gets COPIES of the context and of the views, so that no address of X or of
			// a kernel argument escapes and all of them stay in registers / the kernarg segment throughout the node loop (with the
			// originals passed by reference, every X.field and M.pointer access in the node loop became a scratch load)
			const ModelView Mc = X.M; const SearchParams Pc = X.P;
			GroupCtx<G> Y(X, Mc, Pc);
			SBG_ONLY(const SbgDev Sc = *X.S; Y.S = &Sc;)
			CONG_ONLY(const CongDev Cc = *X.CG; Y.CG = &Cc;)
			CONGG_ONLY(const CongGDev Gc = *X.GG; Y.GG = &Gc;)
			if (finishChunk<G>(Y, chunk, openEnding, res)) break;
			// (its candidates go behind the states: a full arena grows here too, and the stage runs again)
			const WorkView Wc = W;
			if (!growArena<G>(Y, Wc, chunk, Y.stTop)) break;
			X.st = Y.st; X.stCap = Y.stCap; HIST_ONLY(X.hist = Y.hist;)
		}
#ifdef KAMD_HIST
		if (W.slotCap)
		{
			// the arena is this group's, not the chunk's: sort, selection and back-traces now (one lane; a thousandth of the chunk's search), then it is free
			waveSync();
			if (X.gl == 0) { const ModelView Mc = X.M; const SearchParams Pc = X.P; const WorkView Wc = W; const BatchView Bc = B; finishPathsSolo(Mc, Bc, Wc, Pc, chunk, X.st); }
			waveSync();
		}
#endif
#ifdef KAMD_TIMELINE
		if (tl && X.gl == 0) tl[2] = wall_clock64();
#endif
		TLMARK(X, 6)
	}

#include "viterbi_pos.inc"

	// WPS = waves per SIMD the kernel is compiled for (register budget 512 / WPS): 2 is fastest when a batch is small enough
	// to be latency-bound (c2: 8192 chunks), 3 (with a few spills) when there are chunks to fill the extra wave slots
	template<int G, int WPS>
#ifdef KAMD_HIST
	// The history compilations (SkipBigram, global CoNgram) are built for THREE waves per SIMD whatever WPS says: their chunks take milliseconds each, a launch is
	// bound by how many are in flight, and twelve one-wave blocks per CU measured 12 - 16 % faster than eight (MI355X: c3-sbg 159 -> 138 ms per 16 384 sentences,
	// c4-cong-global 346 -> 308 ms per 32 768; four per SIMD: no further gain) although the 168-VGPR build spills more (profiles/r05_p_*)
#define KAMD_KERNEL_WPS 3
#else
#define KAMD_KERNEL_WPS WPS
#endif
	__global__ void __launch_bounds__(64, KAMD_KERNEL_WPS) k_best_path(ModelView M, BatchView B, WorkView W, SearchParams P, uint32_t* chunkCounter, const uint32_t* chunkOrder, uint32_t nWork SBG_ONLY(, SbgDev S) TYPO_ONLY(, const float* nodeTypoAll) CONG_ONLY(, CongDev CGv) CONGG_ONLY(, CongGDev GGv))
	{
		constexpr int NG = 64 / G;
		if (W.posHandOver && *W.posHandOver == 0) return;      // the position-step kernel ran before and left nothing to do
		const uint32_t lane = threadIdx.x;
		// TagSequenceScorer tables (src/TagUtils.cpp:49-62): [0..T_MAX) without, [T_MAX..2*T_MAX) with a left boundary.
		// Tag PA (== T_MAX) indexes one past a row in the reference (include/kiwi/TagUtils.h:10-18): row 0 spills into
		// row 1, row 1 spills into the `weight` member (5.0) -- reproduced by the flat layout plus one extra slot.
		LDS_AS float* lb = ldsPtr<float>(Lay<G>::LB);
		for (uint32_t t = lane; t < 2 * T_MAX + 1; t += 64)
		{
			float v = 0;
			if (t == 2 * T_MAX) v = 5.f;
			else if (t < T_MAX) { if (t == T_NNP || t == T_NP || t == T_IC) v = -1; else if (t == T_SB) v = -3; }
			else { const uint8_t r = (uint8_t)(t - T_MAX); v = (isEClass(r) || isJClass(r) || isSuffixTag(r) || r == T_VCP) ? -1.f : 0.f; }
			lb[t] = v;
		}
		__syncthreads();

		const uint32_t gid = lane / G;
		GroupCtx<G> X(M, P);
		X.gl = lane % G; X.gshift = gid * G; X.lds = gid * Lay<G>::SIZE;
		X.scratch = reinterpret_cast<GroupScratch*>(W.bigScratch) + ((size_t)blockIdx.x * NG + gid);
		X.tl = nullptr;
		SBG_ONLY(X.S = &S; X.hist = nullptr; X.sscr = reinterpret_cast<SbgScratch*>(S.itemScratch) + ((size_t)blockIdx.x * NG + gid);)
		CONGG_ONLY(X.GG = &GGv; X.matrix = false; X.hist = nullptr; X.sscr = reinterpret_cast<SbgScratch*>(GGv.itemScratch) + ((size_t)blockIdx.x * NG + gid);)
		TYPO_ONLY(X.typoAll = nodeTypoAll; X.nodeTypo = nullptr;)
		CONG_ONLY(X.CG = &CGv; X.outFirst = false;)

		for (;;)
		{
			uint32_t ci = 0;
			if (X.gl == 0) ci = atomicAdd(chunkCounter, 1u);
			ci = X.bcast(ci, 0);
			if (ci >= nWork) break;
			searchChunk<G>(X, B, W, chunkOrder[ci]);
		}
	}

#if defined(KAMD_TYPO) && defined(KAMD_CONGG)
	template __global__ void k_best_path<16, 2>(ModelView, BatchView, WorkView, SearchParams, uint32_t*, const uint32_t*, uint32_t, const float*, CongDev, CongGDev);
	template __global__ void k_best_path<64, 2>(ModelView, BatchView, WorkView, SearchParams, uint32_t*, const uint32_t*, uint32_t, const float*, CongDev, CongGDev);
}
}
}
#elif defined(KAMD_CONGG)
	template __global__ void k_best_path<16, 2>(ModelView, BatchView, WorkView, SearchParams, uint32_t*, const uint32_t*, uint32_t, CongDev, CongGDev);
	template __global__ void k_best_path<64, 2>(ModelView, BatchView, WorkView, SearchParams, uint32_t*, const uint32_t*, uint32_t, CongDev, CongGDev);
}
}
#elif defined(KAMD_TYPO) && defined(KAMD_CONG)
	template __global__ void k_best_path<16, 2>(ModelView, BatchView, WorkView, SearchParams, uint32_t*, const uint32_t*, uint32_t, const float*, CongDev);
	template __global__ void k_best_path<64, 2>(ModelView, BatchView, WorkView, SearchParams, uint32_t*, const uint32_t*, uint32_t, const float*, CongDev);
	template __global__ void k_pos_path<16, 2>(ModelView, BatchView, WorkView, SearchParams, const uint32_t*, uint32_t, const float*, CongDev);
	template __global__ void k_pos_path<16, 3>(ModelView, BatchView, WorkView, SearchParams, const uint32_t*, uint32_t, const float*, CongDev);
	template __global__ void k_pos_path<8, 2>(ModelView, BatchView, WorkView, SearchParams, const uint32_t*, uint32_t, const float*, CongDev);
	template __global__ void k_pos_path<8, 3>(ModelView, BatchView, WorkView, SearchParams, const uint32_t*, uint32_t, const float*, CongDev);
}
}
#elif defined(KAMD_TYPO) && defined(KAMD_SBG)
	template __global__ void k_best_path<16, 2>(ModelView, BatchView, WorkView, SearchParams, uint32_t*, const uint32_t*, uint32_t, SbgDev, const float*);
	template __global__ void k_best_path<64, 2>(ModelView, BatchView, WorkView, SearchParams, uint32_t*, const uint32_t*, uint32_t, SbgDev, const float*);
}
}
#elif defined(KAMD_SBG)
	template __global__ void k_best_path<16, 2>(ModelView, BatchView, WorkView, SearchParams, uint32_t*, const uint32_t*, uint32_t, SbgDev);
	template __global__ void k_best_path<64, 2>(ModelView, BatchView, WorkView, SearchParams, uint32_t*, const uint32_t*, uint32_t, SbgDev);
}
#elif defined(KAMD_TYPO)
	template __global__ void k_best_path<16, 2>(ModelView, BatchView, WorkView, SearchParams, uint32_t*, const uint32_t*, uint32_t, const float*);
	template __global__ void k_best_path<64, 2>(ModelView, BatchView, WorkView, SearchParams, uint32_t*, const uint32_t*, uint32_t, const float*);
	template __global__ void k_pos_path<16, 2>(ModelView, BatchView, WorkView, SearchParams, const uint32_t*, uint32_t, const float*);
	template __global__ void k_pos_path<16, 3>(ModelView, BatchView, WorkView, SearchParams, const uint32_t*, uint32_t, const float*);
	template __global__ void k_pos_path<8, 2>(ModelView, BatchView, WorkView, SearchParams, const uint32_t*, uint32_t, const float*);
	template __global__ void k_pos_path<8, 3>(ModelView, BatchView, WorkView, SearchParams, const uint32_t*, uint32_t, const float*);
}
#elif defined(KAMD_CONG)
	template __global__ void k_best_path<16, 2>(ModelView, BatchView, WorkView, SearchParams, uint32_t*, const uint32_t*, uint32_t, CongDev);
	template __global__ void k_best_path<64, 2>(ModelView, BatchView, WorkView, SearchParams, uint32_t*, const uint32_t*, uint32_t, CongDev);
	template __global__ void k_pos_path<16, 2>(ModelView, BatchView, WorkView, SearchParams, const uint32_t*, uint32_t, CongDev);
	template __global__ void k_pos_path<16, 3>(ModelView, BatchView, WorkView, SearchParams, const uint32_t*, uint32_t, CongDev);
	template __global__ void k_pos_path<8, 2>(ModelView, BatchView, WorkView, SearchParams, const uint32_t*, uint32_t, CongDev);
	template __global__ void k_pos_path<8, 3>(ModelView, BatchView, WorkView, SearchParams, const uint32_t*, uint32_t, CongDev);
}
#else
	template __global__ void k_best_path<4, 2>(ModelView, BatchView, WorkView, SearchParams, uint32_t*, const uint32_t*, uint32_t);
	template __global__ void k_best_path<8, 2>(ModelView, BatchView, WorkView, SearchParams, uint32_t*, const uint32_t*, uint32_t);
	template __global__ void k_best_path<16, 2>(ModelView, BatchView, WorkView, SearchParams, uint32_t*, const uint32_t*, uint32_t);
	template __global__ void k_best_path<32, 2>(ModelView, BatchView, WorkView, SearchParams, uint32_t*, const uint32_t*, uint32_t);
	template __global__ void k_best_path<64, 2>(ModelView, BatchView, WorkView, SearchParams, uint32_t*, const uint32_t*, uint32_t);
	template __global__ void k_best_path<8, 3>(ModelView, BatchView, WorkView, SearchParams, uint32_t*, const uint32_t*, uint32_t);
	template __global__ void k_best_path<16, 3>(ModelView, BatchView, WorkView, SearchParams, uint32_t*, const uint32_t*, uint32_t);
	template __global__ void k_pos_path<16, 2>(ModelView, BatchView, WorkView, SearchParams, const uint32_t*, uint32_t);
	template __global__ void k_pos_path<16, 3>(ModelView, BatchView, WorkView, SearchParams, const uint32_t*, uint32_t);
	template __global__ void k_pos_path<8, 2>(ModelView, BatchView, WorkView, SearchParams, const uint32_t*, uint32_t);
	template __global__ void k_pos_path<8, 3>(ModelView, BatchView, WorkView, SearchParams, const uint32_t*, uint32_t);
#endif
}
