// Batch driver (see engine.hpp).  Device memory is plain hipMalloc'd arenas sized from the batch's chunk
// lengths; one stream; three kernel launches per round.
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <cstdio>
#include <cstring>
#include <ctime>
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <numeric>
#include <stdexcept>
#include <thread>
#include "engine.hpp"
#include "pretok.hpp"
#include "hostpool.hpp"
#include "post_fast.hpp"
#include "viterbi_kernel.hpp"
#include "exact_math.hpp"
#include "cong_global.hpp"
#include "chr_freq.hpp"
#include "typo_lattice_kernel.hpp"
#include "typo_graph_kernel.hpp"

namespace kamd
{
	__global__ void k_dict_scan(ModelView M, BatchView B, WorkView W, uint32_t chunkBegin, uint32_t chunkCount);
	template<int GW> __global__ void k_build_lattice(ModelView M, BatchView B, WorkView W, SearchParams P, const uint32_t* chunkList, uint32_t chunkCount, uint32_t ldsBytes);
	__global__ void k_build_lattice_big(ModelView M, BatchView B, WorkView W, SearchParams P, uint32_t chunkBegin, uint32_t chunkCount, uint32_t ldsBytes, uint32_t waveLayout);
	__global__ void k_lattice_wave(ModelView M, BatchView B, WorkView W, SearchParams P, const uint32_t* chunkList, uint32_t chunkCount, uint32_t ldsBytes, uint32_t matchRatio16, uint32_t expandMode);
	__global__ void k_expand_cands(ModelView M, BatchView B, WorkView W, uint32_t chunkBegin, uint32_t chunkCount, uint32_t transposedOrder, const uint8_t* distMask);
	__global__ void k_expand_pos(ModelView M, BatchView B, WorkView W, SearchParams P, uint32_t chunkBegin, uint32_t chunkCount, const float* nodeTypoAll, uint32_t useChr);
	__global__ void k_unk_chr(ModelView M, BatchView B, WorkView W, ChrView C, uint32_t chunkBegin, uint32_t chunkCount, uint32_t hiTok, uint32_t loTok);
	__global__ void k_unk_chr_freq(ModelView M, BatchView B, WorkView W, ChrView C, ChrFreqParams Q, float chrBias, uint32_t chunkBegin, uint32_t chunkCount, uint32_t hiTok, uint32_t loTok);

	namespace
	{
		void hipCheck(hipError_t e, const char* what)
		{
			if (e != hipSuccess) throw std::runtime_error{ std::string{ "HIP error in " } + what + ": " + hipGetErrorString(e) };
		}
#define HIPCHECK(x) hipCheck((x), #x)

		// Device blocks released by a finished batch are kept for the next one: a batch needs ~40 regions (one of them ~1 GB
		// for 8192 sentences), and hipMalloc/hipFree of those per batch cost more than the kernels.  Bounded, per process.
		struct DevBlockCache
		{
			struct Block { void* p; size_t cap; int device; };      // device -1: pinned host memory
			std::mutex mu; std::vector<Block> blocks; size_t bytes = 0;
			static constexpr size_t kMaxBlocks = 256, kMaxBytes = 64ull << 30;
			static void release(const Block& b) { if (b.device < 0) (void)hipHostFree(b.p); else (void)hipFree(b.p); }   // (hipFree takes a pointer of any device)
			void* take(size_t n, size_t& capOut, int device)
			{
				std::lock_guard<std::mutex> g{ mu };
				size_t best = blocks.size();
				for (size_t i = 0; i < blocks.size(); ++i)
					if (blocks[i].device == device && blocks[i].cap >= n && blocks[i].cap <= 2 * n + (1u << 20) && (best == blocks.size() || blocks[i].cap < blocks[best].cap)) best = i;
				if (best == blocks.size()) return nullptr;
				void* p = blocks[best].p; capOut = blocks[best].cap; bytes -= capOut;
				blocks.erase(blocks.begin() + best);
				return p;
			}
			void give(void* p, size_t cap, int device)
			{
				{
					std::lock_guard<std::mutex> g{ mu };
					if (blocks.size() < kMaxBlocks && bytes + cap <= kMaxBytes) { blocks.push_back(Block{ p, cap, device }); bytes += cap; return; }
				}
				release(Block{ p, cap, device });
			}
			void trim()
			{
				std::lock_guard<std::mutex> g{ mu };
				for (auto& b : blocks) release(b);
				blocks.clear(); bytes = 0;
			}
		};
		DevBlockCache& devCache() { static DevBlockCache c; return c; }
		int currentDevice() { int d = 0; (void)hipGetDevice(&d); return d; }

		struct DevBuf
		{
			void* p = nullptr; size_t cap = 0; int device = 0;
			DevBuf() = default;
			DevBuf(const DevBuf&) = delete;
			DevBuf& operator=(const DevBuf&) = delete;
			~DevBuf() { release(); }
			void release() { if (p) devCache().give(p, cap, device); p = nullptr; cap = 0; }
			void ensure(size_t n)
			{
				if (n <= cap) return;
				release();
				device = currentDevice();      // (the engine binds its device to the calling thread before any allocation)
				const size_t want = n + n / 8 + 256;
				p = devCache().take(want, cap, device);
				if (!p)
				{
					hipError_t e = hipMalloc(&p, want);
					if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); devCache().trim(); e = hipMalloc(&p, want); }      // recycled blocks of other sizes are in the way
					if (e != hipSuccess) { p = nullptr; (void)hipGetLastError(); throw std::runtime_error{ std::string{ "HIP error in hipMalloc(" } + std::to_string(want) + " bytes): " + hipGetErrorString(e) }; }
					cap = want;
				}
				// developer aid: KAMD_POISON=1 fills every (re)acquired block with 0xCD, so that a read of memory no kernel has written
				// yet shows up the same way on every run (fresh and recycled blocks otherwise hold arbitrary bytes)
				static const bool poison = std::getenv("KAMD_POISON") != nullptr;
				if (poison) { HIPCHECK(hipMemset(p, 0xCD, cap)); HIPCHECK(hipDeviceSynchronize()); }   // (the engine's streams do not wait for the null stream)
			}
			template<class T> T* as() const { return reinterpret_cast<T*>(p); }
		};

		// pinned host staging memory (H2D of a batch's inputs, D2H of its compact outputs), recycled like the device blocks
		struct PinBuf
		{
			void* p = nullptr; size_t cap = 0;
			PinBuf() = default;
			PinBuf(const PinBuf&) = delete;
			PinBuf& operator=(const PinBuf&) = delete;
			~PinBuf() { if (p) devCache().give(p, cap, -1); }
			void ensure(size_t n)
			{
				if (n <= cap) return;
				if (p) devCache().give(p, cap, -1);
				p = nullptr; cap = 0;
				const size_t want = n + n / 4 + 4096;
				p = devCache().take(want, cap, -1);
				if (!p) { HIPCHECK(hipHostMalloc(&p, want, 0)); cap = want; }
			}
			template<class T> T* as() const { return reinterpret_cast<T*>(p); }
		};

		template<class T> void upload(DevBuf& b, const std::vector<T>& v, hipStream_t s)
		{
			b.ensure(std::max<size_t>(v.size() * sizeof(T), 16));
			if (!v.empty()) HIPCHECK(hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
		}

		// the prepared typo transformer's flat tables on the device (typo_graph_kernel.hpp); the sources are members of the PreparedTypo, which
		// the caller keeps alive for the batch
		struct TypoGraphDev
		{
			DevBuf trie, keys, children, pats, repls, replLast, pool;
			TypoGraphTables tables{};
		};
		void uploadTypoTables(TypoGraphDev& d, const PreparedTypo& T, hipStream_t s)
		{
			upload(d.trie, T.trie(), s); upload(d.keys, T.trieKeys(), s); upload(d.children, T.trieChildren(), s);
			upload(d.pats, T.patterns(), s); upload(d.repls, T.replacements(), s); upload(d.replLast, T.replLast(), s);
			const std::u16string& pool = T.pool();
			d.pool.ensure(2 * (pool.size() + 1) + 16);
			HIPCHECK(hipMemcpyAsync(d.pool.p, pool.c_str(), 2 * (pool.size() + 1), hipMemcpyHostToDevice, s));      // (with the terminating NUL: an empty replacement at the end of the pool reads it)
			TypoGraphTables& t = d.tables;
			t.trie = d.trie.as<PreparedTypo::TrieNode>(); t.keys = d.keys.as<uint16_t>(); t.children = d.children.as<uint32_t>();
			t.pats = d.pats.as<PreparedTypo::Pattern>(); t.repls = d.repls.as<PreparedTypo::Repl>(); t.replLast = d.replLast.as<uint8_t>(); t.pool = d.pool.as<uint16_t>();
			t.continualCost = T.continualCost(); t.continualOn = std::isfinite(T.continualCost()) ? 1 : 0;
			t.entryNode = T.entryNode();
			t.hiType = identifySpecialChr(0xD800); t.hiScript = chr2ScriptType(0xD800); t.loType = identifySpecialChr(0xDC00); t.loScript = chr2ScriptType(0xDC00);
		}

	}

	// the sorted, distinct special states a chunk is searched under: almost always {0}.  Kept inline up to 14 of them -- one heap block per chunk was
	// 65 536 allocations when a batch is staged and as many frees when it is released, on the calling thread (3 + 3 ms of a 29 ms end-to-end batch)
	struct SpSet
	{
		uint8_t n = 0; uint8_t v[14] = {}; std::vector<uint8_t> big;
		SpSet() = default;
		SpSet(std::initializer_list<uint8_t> l) { assign(l.begin(), l.size()); }
		void assign(const uint8_t* p, size_t k) { big.clear(); if (k <= sizeof(v)) { n = (uint8_t)k; if (k) std::memcpy(v, p, k); } else { n = 0xFF; big.assign(p, p + k); } }
		SpSet& operator=(const std::vector<uint8_t>& o) { assign(o.data(), o.size()); return *this; }
		size_t size() const { return n == 0xFF ? big.size() : n; }
		bool empty() const { return size() == 0; }
		const uint8_t* data() const { return n == 0xFF ? big.data() : v; }
		bool operator==(const std::vector<uint8_t>& o) const { return o.size() == size() && (o.empty() || std::memcmp(o.data(), data(), o.size()) == 0); }
		bool operator!=(const std::vector<uint8_t>& o) const { return !(*this == o); }
	};
	struct ChunkRef { uint32_t text, chunk; SpSet sp; bool openEnding; bool onlyChunk = false; };      // onlyChunk: the text has no other chunk to analyse

	struct StagedBatch
	{
		U16 rawFlat; std::vector<uint64_t> rawOff;   // the raw texts (result assembly reads them: word positions, line breaks)
		std::vector<PrepBlock> prepBlocks;            // flat storage of the prepared texts, kPrepBlock consecutive texts per block
		std::vector<PreparedView> prep;               // per text: views into its block
		static constexpr size_t kPrepBlock = 64;
		int hostThreads = 0;
		std::vector<ChunkRef> refs;
		uint64_t match = 0;
		uint32_t capScale = 1;
		// Pretokenized spans (Kiwi::analyze's `pretokenized`, pretok.hpp): the spans of text 0 in normalised offsets with the forms of their lattice nodes and the
		// temporary forms / morphemes behind the model's tables (uploaded before the batch's first kernel); a batch of runRefs shares its parent's
		std::shared_ptr<const PretokGroup> pretok;
		std::vector<uint32_t> blockBitsHost;          // a model with dialect morphemes: the blocklist united with the morphemes of dialects the analysis does not allow
		bool isRerun = false;                         // a batch of runRefs (chunks searched again): the engine's adaptive capacities do not learn from it
		uint64_t units = 0, devBytes = 0;
		// host layout
		std::vector<uint32_t> charOff, patOff, spOff, matchBase, nodeBase, packBase;
		std::vector<uint64_t> stateBase, tokenBase;
		// device
		PinBuf hIn; DevBuf dIn;   // the batch's input block (layoutAndUpload)
		DevBuf dFullMask, dZAt, dNsToPos, dPosToNs, dCflag, dMask, dMoff, dNNs, dMatchForm, dNodes, dTmpNodes, dEndPosMap, dTmpIdx, dNNodes, dWideList, dExpanded;
		// LDS size classes of the lattice kernel per sub-batch {first, end, bytes}, first = the chunks beyond the budget; made once per (batch, match ratio, kernel)
		struct LatClass { uint32_t i, j, need, stream; };
		std::vector<std::vector<LatClass>> latClasses; std::vector<uint32_t> latSkip; uint32_t latClassesKey = 0xFFFFFFFFu;
		DevBuf dHist;   // SkipBigram models: history ring of every search state (8 x u32), parallel to dStates
		uint32_t slotCap = 0; uint64_t ownStates = 0;   // slotCap != 0: the arenas are the search kernel's lane groups' (WorkView::slotCap); ownStates: states in the chunks' / groups' own arenas
		DevBuf dStateAt, dSlotTable; uint64_t poolStates = 0, poolUsed = 0;   // where each chunk's arena lies now; the pool behind the arenas (WorkView::poolTop) and what the last run took of it
		// typo correction: the transformer the batch is analysed with, the typo graph of every chunk and the
		// working arrays of k_build_lattice_typo, the typo cost of every lattice node beside dNodes
		TypoOption typo;
		TypoGraphDev typoDev;
		DevBuf dTypoGraph, dTypoLast, dTypoPool, dTypoChunks, dTypoTmp, dTypoMap, dTypoNs, dTypoPs, dTypoStates, dTypoSIdx, dTypoScratch, dNodeTypo, dTypoOrder, dBlockBits, dUnkChr, dUnkChrForm;
		TypoLatView tv{};
		DevBuf dPacks, dStates, dNodeStOff, dNodeStCnt, dReach, dTokens, dResults, dOrder;
		DevBuf dPosRecs, dPosDesc, dPosPrev, dPosNodeRec, dPosMask, dPosBig;   // position program of k_pos_path (k_expand_pos)
		DevBuf dOutPaths, dOutTokens, dOutCounters; uint32_t outPathCap = 0, outTokCap = 0;   // compact outputs of the end stage
		PinBuf hOut, hOut2;       // D2H landing zones: counters + chunk results; then the path headers and token records that were produced
		uint64_t outBytes = 0;    // bytes the last download copied
		BatchView bv{}; WorkView wv{};
		const DevChunkResult* hResults = nullptr; const DevPathHeader* hPaths = nullptr; const DevToken* hTokens = nullptr; size_t hTokCount = 0;   // inside hOut
		bool ran = false;
		hipEvent_t evDone = nullptr; bool launched = false; uint32_t launchS = 1;      // recorded behind the batch's last kernel (Engine::launch); Engine::finish waits for it
		~StagedBatch()
		{
			if (evDone) (void)hipEventDestroy(evDone);
			// the prepared texts go back to the allocator on the workers (a thousand blocks of six vectors each: 1.4 ms per 65 536 texts on one thread)
			if (prepBlocks.size() >= 64)
			{
				try { HostPool::instance().run(prepBlocks.size(), 16, hostThreads, [&](size_t i0, size_t i1, int) { for (size_t i = i0; i < i1; ++i) { PrepBlock dead = std::move(prepBlocks[i]); } }); }
				catch (...) {}
			}
		}
		uint32_t subBatches = 0;
		uint32_t topN = 1;            // the search of the last run() kept this many paths per key
		// chunks of the last run that overflowed and were searched again with larger capacities: index into overPaths per chunk (SIZE_MAX: none)
		std::vector<size_t> overIdx; std::vector<std::vector<PathResult>> overPaths; uint32_t rerunChunks = 0; float rerunMs = 0;
		std::vector<uint32_t> typoNeed, typoOrder;   // typo lattices: LDS need of every chunk, chunks by descending need inside each sub-batch (= dTypoOrder)
		std::vector<uint32_t> order;   // host copy of dOrder (work order: longest chunk first inside each sub-batch)
	};

	struct Engine::Impl
	{
		std::shared_ptr<FlatModel> modelOwner;   // the baked model is shared by the engines of one handle (one replica of the DEVICE tables per GPU)
		FlatModel& model;
		explicit Impl(std::shared_ptr<FlatModel> m) : modelOwner(std::move(m)), model(*modelOwner) {}
		ModelView dview{};
		std::vector<std::unique_ptr<DevBuf>> modelBufs;
		hipStream_t stream = nullptr, stream2 = nullptr;   // lattice stages / search stage (sub-batches overlap)
		hipStream_t latStream[3] = { nullptr, nullptr, nullptr };   // the lattice kernel's LDS size classes are launched round-robin over `stream` and these: their tails overlap
		// Batches in flight (Engine::launch / finish: the host prepares the next part of a large batch while the kernels of the previous one run): uploads and
		// downloads go over `streamCopy`, so that waiting for them never waits for kernels of another batch queued on `stream` / `stream2`; the kernels of two
		// batches do NOT overlap -- the first launch of a batch waits for `lastDone`, the end of the batch launched before it (the engine's scratch
		// blocks, work counters and timing events are shared) -- what overlaps is host work with device work.
		hipStream_t streamCopy = nullptr;
		hipEvent_t lastDone = nullptr, joinEv = nullptr; bool haveLast = false;
		hipEvent_t latFork = nullptr, latJoin[3] = { nullptr, nullptr, nullptr };
		std::vector<hipEvent_t> evs;
		int subBatches = 0;   // 0 = automatic
		int device = 0;
		uint32_t persistBlocks = 0;
		int latticeGroupForced = 0;      // KAMD_LATTICE_GROUP=16 / 64: lanes per chunk of k_build_lattice
		bool latticeWave = true;         // k_lattice_wave (all lanes build the lattice); KAMD_LATTICE_WAVE=0: k_build_lattice's one-lane replay
		// LDS room of k_lattice_wave for packed matches, sixteenths per text unit: follows what the model's dictionary produced in the batches so far
		// (read back with every batch's counters); the first batch assumes 3 per unit, a chunk beyond the room goes to the wide launch (KAMD_LATTICE_RATIO fixes it)
		uint32_t latticeRatio16 = kLatticeWideRatio16; bool latticeRatioForced = false;
		// state arenas: sixteenths of the worst-case capacity (48 states per text unit + 256; SkipBigram models x 8) a chunk's region gets.  Follows what the
		// chunks of the batches so far needed (x 2, read from the downloaded per-chunk results); a batch in which a chunk ran out goes back to the full
		// capacity (the capacity ladder inside run() has searched that chunk again meanwhile).  KAMD_STATE_SCALE=<sixteenths> fixes it
		// State arenas: a chunk gets stateScale64 / 64 of its worst-case capacity -- what 90 % of the chunks of the batches so far needed (SkipBigram / global CoNgram;
		// otherwise twice what 99.9 % needed) -- and one that needs more grows
		// into the batch's pool (WorkView::poolTop: twice the size each time, append-only); the pool holds poolFrac64 / 64 of the arenas' total, twice what the last batch took.
		// [needed by top-1 batches, by top-N batches]; a region gets the larger of the two (a batch is laid out before its top-N is known)
		uint32_t stateScale64[2] = { 0, 0 }; bool stateScaleForced = false;      // (0: no batch of that kind seen yet -- the other kind's scale serves, the pool covers the difference; neither: the whole worst case)
		uint32_t poolFrac64 = 64; bool poolForced = false;
		uint32_t latticeLdsBudget = 64 * 1024;   // dynamic LDS one lattice-build wave may ask for (KAMD_LATTICE_LDS; 0 = HBM kernel only)
		uint32_t latticeWaveBudget = 128 * 1024; // ... and k_lattice_wave, which is allowed beyond the default 64 KB limit (a 400-unit chunk needs ~70 KB; the CU has 160 KB)
		bool groupLanesForced = false; int wpsForced = 0;   // KAMD_GROUP_LANES / KAMD_WPS given
		bool posPath = true;  // the position-step search kernel runs first, the general one on what it hands over (KAMD_POS_PATH=0: general kernel only)
		bool wantCongGlobal = false;     // LmMode::CongGlobal: score with the distant-token (window) sections of the CoNgram file
		bool posPathForced = false;      // KAMD_POS_PATH=2 (kept for the parity suites: the position steps wherever a compilation has them)
		bool posPathTypo = true;         // ... also over typo lattices (KAMD_POS_PATH=3: not there, the behaviour of rounds 3 - 5)
		int posGroupForced = 0;          // KAMD_POS_G=8 / 16: lane-group width of the position-step kernel (0: by batch size)
		int groupLanes = 16;  // lanes per chunk in the search kernel (KAMD_GROUP_LANES = 4 | 8 | 16 | 32 | 64); 16 measured best
		DevBuf bigScratch, counter, posScratch;
		uint32_t posContSlots = 256;     // chunks per launch that k_pos_path carries on in the general search itself (KAMD_POS_CONT=0: none, all left to k_best_path)
		ChrView chr{};      // character model of Match::oovChrModel on the device (absent: dim 0)
		CongDev cong{}; bool hasCong = false;   // CoNgram model: the context trie is uploaded where the Knlm tables would be (ModelView::lmHash / lmRoot2 / lmBackoff)
		CongGDev congG{}; bool hasCongG = false;   // global CoNgram model (LmMode::CongGlobal): the window sections on the device; its search shares the SkipBigram kernel's history plumbing
		bool histStates() const { return hasSbg || hasCongG; }      // search states carry eight history words beside DevState; item histories in sbgScratch
		// ... and their state arenas belong to the search kernel's lane groups (WorkView::slotCap): so many arenas a launch can use -- 8 persistent one-wave blocks
		// per CU, one group per wave unless 16-lane groups are forced (four)
		uint32_t histBlocksPerCu() const { static const int v = std::getenv("KAMD_HIST_BLOCKS") ? std::atoi(std::getenv("KAMD_HIST_BLOCKS")) : 12; return (uint32_t)std::min(16, std::max(1, v)); }      // (12 = three waves per SIMD, what those kernels are built for; the knob is a developer's)
		uint32_t histSlots() const { return persistBlocks / 12 * histBlocksPerCu() * ((groupLanesForced && groupLanes == 16) ? 4u : 1u); }
		SbgDev sbg{}; bool hasSbg = false; DevBuf sbgScratch;   // SkipBigram tables on the device + per-lane-group item scratch of its search kernel
		// the engine owns ONE pair of streams, one work counter and one scratch arena: device work of concurrent callers (the C API
		// is callable from many threads, reference capi threading contract) is serialised per engine; host preparation is not
		std::recursive_mutex deviceMu;
		TokenTemplates tokTmpl;      // (post_fast.hpp; built with the first fetch)

		// room behind the model's form / morpheme tables for the temporary entries of a batch with pretokenized spans (TempOverlay): elements per table
		static constexpr size_t kTempForms = 4096, kTempMorphs = 8192, kTempChars = 1u << 17, kTempCand = 16384, kTempChunks = 16384;
		template<class T> const T* up(const std::vector<T>& v, size_t slack = 0)
		{
			modelBufs.emplace_back(new DevBuf);
			DevBuf& b = *modelBufs.back();
			b.ensure(std::max<size_t>((v.size() + slack) * sizeof(T), 16));
			if (!v.empty()) HIPCHECK(hipMemcpy(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
			return b.as<T>();
		}
	};

	int Engine::deviceIndex() const { return impl->device; }
	void Engine::bindThread() const { (void)hipSetDevice(impl->device); }
	int Engine::visibleDevices()
	{
		int n = 0;
		if (hipGetDeviceCount(&n) != hipSuccess) return 0;
		return n;
	}

	Engine::Engine(const std::string& path, int device, LmMode lm, uint32_t enabledDialects) : impl(new Impl(std::make_shared<FlatModel>()))
	{
		bakeModel(impl->model, path, enabledDialects);
		if (lm == LmMode::Sbg && impl->model.sbgPtrs.empty()) throw std::runtime_error{ "Cannot open required files for skipbigram model" };   // KiwiBuilder.cpp:1008-1013
		if ((lm == LmMode::Cong || lm == LmMode::CongGlobal) && !impl->model.congDim) throw std::runtime_error{ "Cannot open ConG model file 'cong.mdl'" };      // KiwiBuilder.cpp:1018-1023
		impl->wantCongGlobal = lm == LmMode::CongGlobal;
		// a model without a Knlm blob (the layout of models/cong/base: sj.morph + cong.mdl) cannot serve the Knlm / SkipBigram types: the reference
		// fails to open sj.knlm there (KiwiBuilder.cpp:985-1001); searching with empty LM tables would read out of bounds
		if ((lm == LmMode::Knlm || lm == LmMode::Sbg) && impl->model.lmNodes.empty()) throw std::runtime_error{ "Cannot open required file 'sj.knlm' for the requested model type" };
		if (lm == LmMode::Knlm || lm == LmMode::Sbg)
		{
			impl->model.congDim = 0;
			// ... and the character model is run quantised only next to a CoNgram model (the reference uses its fp32 scorer otherwise): not with these types
			impl->model.chrDim = 0; impl->model.formUnkChr.clear(); impl->model.formChrTok.clear();
		}
		if (impl->model.congDim) { impl->model.sbgPtrs.clear(); impl->model.sbgKeys.clear(); impl->model.sbgComps.clear(); impl->model.sbgDiscnts.clear(); impl->model.sbgValid.clear(); }
		if (lm == LmMode::Knlm) { impl->model.sbgPtrs.clear(); impl->model.sbgKeys.clear(); impl->model.sbgComps.clear(); impl->model.sbgDiscnts.clear(); impl->model.sbgValid.clear(); }
		if (!impl->model.sbgPtrs.empty() && impl->model.sbgWindow != 8)
			throw std::runtime_error{ "kiwi_amd: SkipBigram window size must be 8 (the reference instantiates SbgState<8> only, src/SkipBigramModel.cpp)" };
		openDevice(device);
	}

	// a replica on another GPU: same baked model on the host, its own device tables, streams and scratch
	Engine::Engine(const Engine& other, int device) : impl(new Impl(other.impl->modelOwner)), config(other.config) { impl->wantCongGlobal = other.impl->wantCongGlobal; openDevice(device); }

	void Engine::openDevice(int device)
	{
		int nDev = 0;
		if (hipGetDeviceCount(&nDev) != hipSuccess || nDev == 0)
			throw std::runtime_error{ "kiwi_amd: no HIP device visible -- the analyze path has no CPU fallback" };
		if (device < 0 && hipGetDevice(&device) != hipSuccess) device = 0;      // -1: the calling thread's current device (device 0 unless the caller chose another)
		impl->device = device;
		HIPCHECK(hipSetDevice(device));
		HIPCHECK(hipStreamCreateWithFlags(&impl->stream, hipStreamNonBlocking));
		HIPCHECK(hipStreamCreateWithFlags(&impl->stream2, hipStreamNonBlocking));
		for (auto& ls : impl->latStream) HIPCHECK(hipStreamCreateWithFlags(&ls, hipStreamNonBlocking));
		// (KAMD_COPY_PRIORITY=1: the copy stream at the highest priority.  Streams share a handful of hardware queues, and a copy that lands in the queue of a compute stream
		// waits for every kernel enqueued there before it: on c4-cong the 64-byte counter read-back of a batch's FIRST part took 32 ms, until the kernels of all four parts had
		// run.  With its own queue the parts are collected under the later parts' kernels -- and the batch takes as long as before, 67 ms: what bounds c4-cong end to end is
		// that the kernels of a PART have a floor (14.6 / 17.9 / 27.0 ms for 32 768 / 65 536 / 131 072 sentences: the chain of the longest chunks), so four parts cost 58 ms of
		// kernels where one batch costs 27; c2-64k is no faster either (its kernels are hidden): profiles/r06_c4_*.  Left off.)
		{
			int least = 0, greatest = 0;
			const bool plain = [] { const char* e = std::getenv("KAMD_COPY_PRIORITY"); return !(e && std::atoi(e) == 1); }();
			if (plain || hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess || least == greatest) HIPCHECK(hipStreamCreateWithFlags(&impl->streamCopy, hipStreamNonBlocking));
			else HIPCHECK(hipStreamCreateWithPriority(&impl->streamCopy, hipStreamNonBlocking, greatest));
		}
		HIPCHECK(hipEventCreateWithFlags(&impl->lastDone, hipEventDisableTiming)); HIPCHECK(hipEventCreateWithFlags(&impl->joinEv, hipEventDisableTiming));
		HIPCHECK(hipEventCreate(&impl->latFork)); for (auto& ev : impl->latJoin) HIPCHECK(hipEventCreate(&ev));

		if (const char* sb = std::getenv("KAMD_SUBBATCHES")) impl->subBatches = std::atoi(sb);
		const FlatModel& m = impl->model;
		ModelView& v = impl->dview;
		v.h = m.h;
		using I_ = Engine::Impl;
		v.forms = impl->up(m.forms, I_::kTempForms); v.formChars = impl->up(m.formChars, I_::kTempChars); v.formCand = impl->up(m.formCand, I_::kTempCand);
		{
			// device copy of the morpheme table: `feat` / `prevFlags` are replaced by the path-side values (FlatModel::morphPath)
			std::vector<MorphRec> dm = m.morphs;
			for (size_t i = 0; i < dm.size(); ++i) { dm[i].feat = (uint16_t)m.morphPath[i]; dm[i].prevFlags = (uint8_t)(m.morphPath[i] >> 16); }
			v.morphs = impl->up(dm, I_::kTempMorphs);
			std::vector<CandStatic> unk(2);
			for (int k = 0; k < 2; ++k)
			{
				const uint32_t mid = (k == 0 ? T_NNG : T_NNP) + 1u;
				std::memcpy(&unk[k].m0, &dm[mid], 32);
				const uint32_t firstWid = (dm[mid].flags & MF_SINGLE) ? dm[mid].lmId : m.chunkLm[dm[mid].chunkOff];
				unk[k].x = Quad{ mid, firstWid, 0, 0 };
			}
			v.unkPacks = impl->up(unk);
		}
		v.chunkMorph = impl->up(m.chunkMorph, I_::kTempChunks); v.chunkLm = impl->up(m.chunkLm, I_::kTempChunks); v.chunkPos = impl->up(m.chunkPos, 2 * I_::kTempChunks);
		v.sbInfo = impl->up(m.sbInfo, I_::kTempMorphs); v.morphPath = impl->up(m.morphPath, I_::kTempMorphs);
		v.trie = impl->up(m.trie); v.trieKeys = impl->up(m.trieKeys); v.trieChild = impl->up(m.trieChild); v.trieRoot = impl->up(m.trieRoot);
		v.trieEdges = impl->up(m.trieEdges); v.trieEdgeMask = m.trieEdgeMask;
		v.lmNodes = impl->up(m.lmNodes); v.lmKeys = impl->up(m.lmKeys); v.lmValues = impl->up(m.lmValues); v.lmRoot = impl->up(m.lmRoot);
		if (m.congDim)
		{
			// CoNgram: the search walks the CONTEXT trie through the same lookup structures (edge hash with the child's context id in the slot,
			// root table, suffix links); the initial state is the root with context 0 (CoNgramState())
			v.lmHash = impl->up(m.congHash); v.lmHashMask = m.congHashMask; v.lmRoot2 = impl->up(m.congRoot2); v.lmBackoff = impl->up(m.congBackoff);
			v.h.bosNode = 0;
			impl->cong = CongDev{ impl->up(m.congCtxEmb), impl->up(m.congOutEmb), m.congDim, m.congDim + 8, m.congVlTMax, m.congVlBits };
			impl->hasCong = true;
			if (impl->wantCongGlobal)
			{
				if (!m.congWindow) throw std::runtime_error{ "Cannot open ConG model with distant tokens: the file has no window sections" };
				if (m.congWindow != 7) throw std::runtime_error{ "kiwi_amd: CoNgram window size must be 7 (the reference instantiates CoNgramState<7> only)" };
				CongGDev& g = impl->congG;
				g.ctxConf = impl->up(m.congCtxConf); g.distEmb = impl->up(m.congDistEmb); g.distConf = impl->up(m.congDistConf); g.posConf = impl->up(m.congPosConf);
				g.distMask = impl->up(m.congDistMask); g.window = m.congWindow; g.keyBytes = m.congKeyBytes; g.hist = nullptr; g.itemScratch = nullptr;
				impl->hasCongG = true;
			}
		}
		else { v.lmHash = impl->up(m.lmHash); v.lmHashMask = m.lmHashMask; v.lmRoot2 = impl->up(m.lmRoot2); v.lmBackoff = impl->up(m.lmBackoff); }
		v.lmHtxNode = (m.congDim || m.lmHtxNode.empty()) ? nullptr : impl->up(m.lmHtxNode);      // history-transformed Knlm only
		// (dialect bits: the room behind them is zero -- a temporary entry belongs to no dialect)
		v.formDialect = nullptr;      // (a bake-time fact: flat_model.hpp ModelView::formDialect)
		v.morphDialect = m.morphDialect.empty() ? nullptr : impl->up(m.morphDialect, I_::kTempMorphs);
		if (v.morphDialect) HIPCHECK(hipMemset(const_cast<uint16_t*>(v.morphDialect) + m.morphDialect.size(), 0, 2 * I_::kTempMorphs));
		v.formUnkChr = nullptr; v.formChrTok = nullptr;
		v.lmChain = nullptr;
		if (!m.congDim && !m.lmBackoff.empty())
		{
			std::vector<uint32_t> chain(2 * m.lmBackoff.size());
			for (size_t n = 0; n < m.lmBackoff.size(); ++n)
			{
				const uint32_t l1 = n ? (uint32_t)((int64_t)n + m.lmBackoff[n].lower) : 0u;
				const uint32_t l2 = l1 ? (uint32_t)((int64_t)l1 + m.lmBackoff[l1].lower) : 0u;
				chain[2 * n] = l1; chain[2 * n + 1] = l2;
			}
			v.lmChain = impl->up(chain);
		}
		if (m.chrDim)
		{
			ChrView c = m.chrView();
			c.ctxEmb = impl->up(m.chrCtxEmb); c.outEmb = impl->up(m.chrOutEmb); c.nodes = impl->up(m.chrNodes); c.keys = impl->up(m.chrKeys); c.values = impl->up(m.chrValues);
			c.root = impl->up(m.chrRoot); c.inv = m.chrInv.empty() ? nullptr : impl->up(m.chrInv);
			c.depth = impl->up(m.chrDepth); c.freqTab = impl->up(m.chrFreqTab);
			impl->chr = c;
			v.formUnkChr = impl->up(m.formUnkChr, I_::kTempForms); v.formChrTok = impl->up(m.formChrTok, I_::kTempChars);
		}
		if (!m.sbgPtrs.empty())
		{
			const SbgView sv = m.sbgView();
			SbgDev& d = impl->sbg;
			d.ptrs = impl->up(m.sbgPtrs); d.keys = impl->up(m.sbgKeys); d.comps = impl->up(m.sbgComps); d.discnts = impl->up(m.sbgDiscnts); d.valid = impl->up(m.sbgValid);
			d.vocabSize = sv.vocabSize; d.logWindowSize = sv.logWindowSize; d.hist = nullptr; d.itemScratch = nullptr;
			impl->hasSbg = true;
		}
		hipDeviceProp_t prop;
		HIPCHECK(hipGetDeviceProperties(&prop, device));
		impl->persistBlocks = (uint32_t)prop.multiProcessorCount * 12;   // one-wave persistent blocks: 3 waves per SIMD is the most the search kernel is built for
		if (const char* g = std::getenv("KAMD_GROUP_LANES"))
		{
			const int v = std::atoi(g);
			if (v == 4 || v == 8 || v == 16 || v == 32 || v == 64) { impl->groupLanes = v; impl->groupLanesForced = true; }
			else throw std::runtime_error{ "KAMD_GROUP_LANES must be 4, 8, 16, 32 or 64" };
		}
		if (const char* w = std::getenv("KAMD_WPS")) { const int v = std::atoi(w); if (v == 2 || v == 3) impl->wpsForced = v; else throw std::runtime_error{ "KAMD_WPS must be 2 or 3" }; }
		if (const char* pp = std::getenv("KAMD_POS_PATH")) { impl->posPath = std::atoi(pp) != 0; impl->posPathForced = std::atoi(pp) == 2; impl->posPathTypo = std::atoi(pp) != 3; }
		if (const char* pg = std::getenv("KAMD_POS_G")) { const int v = std::atoi(pg); if (v == 8 || v == 16) impl->posGroupForced = v; else throw std::runtime_error{ "KAMD_POS_G must be 8 or 16" }; }
		if (const char* pc = std::getenv("KAMD_POS_CONT")) impl->posContSlots = (uint32_t)std::max(0, std::min(4096, std::atoi(pc)));
		if (const char* lg = std::getenv("KAMD_LATTICE_GROUP")) { const int v = std::atoi(lg); if (v == 16 || v == 64) impl->latticeGroupForced = v; }
		if (const char* lw = std::getenv("KAMD_LATTICE_WAVE")) impl->latticeWave = std::atoi(lw) != 0;
		if (const char* ss = std::getenv("KAMD_STATE_SCALE")) { impl->stateScale64[0] = impl->stateScale64[1] = (uint32_t)std::min(64, std::max(1, std::atoi(ss))); impl->stateScaleForced = true; }      // (64ths of the worst case)
		if (const char* ps = std::getenv("KAMD_STATE_POOL")) { impl->poolFrac64 = (uint32_t)std::min(4096, std::max(0, std::atoi(ps))); impl->poolForced = true; }      // (64ths of the arenas' total; 0: no pool)
		if (const char* lr = std::getenv("KAMD_LATTICE_RATIO")) { impl->latticeRatio16 = (uint32_t)std::min(4096, std::max(4, std::atoi(lr))); impl->latticeRatioForced = true; }
		if (const char* l = std::getenv("KAMD_LATTICE_LDS")) { impl->latticeLdsBudget = (uint32_t)std::min(64 * 1024, std::max(0, std::atoi(l))); impl->latticeWaveBudget = (uint32_t)std::min(128 * 1024, std::max(0, std::atoi(l))); }
		if (impl->latticeWaveBudget > 64 * 1024) HIPCHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lattice_wave), hipFuncAttributeMaxDynamicSharedMemorySize, (int)impl->latticeWaveBudget));
		impl->counter.ensure(256);
	}

	Engine::~Engine()
	{
		if (impl)
		{
			(void)hipSetDevice(impl->device);
			(void)hipDeviceSynchronize();
			for (auto& e : impl->evs) if (e) (void)hipEventDestroy(e);
			// the engine's own blocks go back to the cache first, then the cache is emptied: nothing stays allocated after the last engine
			impl->modelBufs.clear();
			impl->bigScratch.release(); impl->posScratch.release(); impl->counter.release(); impl->sbgScratch.release();
			devCache().trim();
			if (impl->stream) (void)hipStreamDestroy(impl->stream);
			if (impl->stream2) (void)hipStreamDestroy(impl->stream2);
			for (auto ls : impl->latStream) if (ls) (void)hipStreamDestroy(ls);
			if (impl->streamCopy) (void)hipStreamDestroy(impl->streamCopy);
			if (impl->lastDone) (void)hipEventDestroy(impl->lastDone);
			if (impl->joinEv) (void)hipEventDestroy(impl->joinEv);
			if (impl->latFork) (void)hipEventDestroy(impl->latFork);
			for (auto ev : impl->latJoin) if (ev) (void)hipEventDestroy(ev);
		}
	}

	const FlatModel& Engine::model() const { return impl->model; }
	bool Engine::usesCong() const { return impl->hasCong; }
	bool Engine::usesSbg() const { return impl->hasSbg; }
	bool Engine::usesCongGlobal() const { return impl->hasCongG; }
	uint32_t Engine::congWindow() const { return impl->hasCong ? impl->model.congWindow : 0u; }

	namespace
	{
		// developer aid: KAMD_HOST_TIMING=1 prints where the host side of stage() / fetch() spends its time
		struct HostTimer
		{
			bool on = std::getenv("KAMD_HOST_TIMING") != nullptr; const char* what; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
			double cpu0 = on ? cpuNow() : 0.0;
			explicit HostTimer(const char* w) : what(w) {}
			// CPU time of the whole process (every worker): what a CFS quota meters
			static double cpuNow() { timespec ts; clock_gettime(CLOCK_PROCESS_CPUTIME_ID, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
			void lap(const char* name)
			{
				if (!on) return;
				const auto t1 = std::chrono::steady_clock::now(); const double c1 = cpuNow();
				fprintf(stderr, "[host] %s: %s %.2f ms (process CPU %.1f ms)\n", what, name, std::chrono::duration<double, std::milli>(t1 - t0).count(), c1 - cpu0);
				t0 = t1; cpu0 = c1;
			}
		};
	}

	// Lays a set of chunks out in HBM.  Everything the kernels read about the batch -- text, character classes, scripts, patterns, special
	// states, flags and the region offsets -- is assembled in ONE pinned host block and uploaded with ONE copy.
	// The pretokenized spans that begin inside chunk `d` of text `text` (the chunk cut never ends inside one), as entries BEHIND the chunk's pattern list:
	// {end, length} chunk-relative, tag = kSpanTag | fallback << 30 | form id (device_types.hpp; the lattice replay splits the list at the first such tag)
	template<class F> static void forSpansOfChunk(const StagedBatch& b, uint32_t text, const ChunkDesc& d, F&& f)
	{
		if (!b.pretok || text != 0) return;
		for (const auto& sn : b.pretok->spans)
			if (sn.begin >= d.startOffset && sn.begin < d.startOffset + d.nChars)
				f(DevPattern{ sn.end - d.startOffset, sn.end - sn.begin, kSpanTag | (sn.fallback ? kSpanFallback : 0u) | sn.form });
	}

	static void layoutAndUpload(Engine::Impl& I, StagedBatch& b, const SearchParams&)
	{
		HostTimer tmL{ "layout" };
		const size_t nC = b.refs.size();
		b.charOff.assign(nC + 1, 0); b.patOff.assign(nC + 1, 0); b.spOff.assign(nC + 1, 0);
		b.matchBase.assign(nC + 1, 0); b.nodeBase.assign(nC + 1, 0); b.packBase.assign(nC + 1, 0); b.stateBase.assign(nC + 1, 0); b.tokenBase.assign(nC + 1, 0);
		const uint64_t sc = b.capScale;
		const bool tinyArenas = std::getenv("KAMD_TEST_TINY_ARENAS") != nullptr;
		static const bool noSlots = std::getenv("KAMD_NO_STATE_SLOTS") != nullptr;      // (developer switch: per-chunk arenas for the history models too)
		const bool slotMode = I.histStates() && !noSlots;
		uint64_t slotCap = 0;
		// what chunk c takes of each region (the offsets are running sums of these)
		struct ChunkSizes { uint64_t n, pat, sp, mcap, ncap, scap, tcap, slot; };
		const uint32_t scale64 = std::max(I.stateScale64[0], I.stateScale64[1]) ? std::max(I.stateScale64[0], I.stateScale64[1]) : 64u;
		auto sizesOf = [&](size_t c)
		{
			ChunkSizes z{};
			const auto& r = b.refs[c];
			const ChunkDesc& d = b.prep[r.text].chunks[r.chunk];
			const uint64_t n = d.nChars;
			z.n = n;
			z.pat = d.patEnd - d.patBegin;
			forSpansOfChunk(b, r.text, d, [&](const DevPattern&) { ++z.pat; });
			z.sp = r.sp.size();
			uint64_t mcap = (6 * n + 64) * sc, ncap = std::min<uint64_t>((4 * n + 32) * sc, 0xFFE0), scap = (48 * n + 256) * sc, tcap = (4 * n + 32) * sc;
			// SkipBigram states carry their history ring in the container key: far fewer paths merge, a node keeps hundreds to thousands of them
			if (I.histStates()) scap *= 8;
			if (sc == 1 && !tinyArenas && !slotMode) scap = std::max<uint64_t>(scap * scale64 / 64, 64);      // (what the batches so far needed; a re-run at a higher rung takes the whole capacity)
			if (b.typo.typo) ncap = std::min<uint64_t>(2 * ncap, 0xFFE0);      // lattices over typo graphs come out about twice as large
			if (tinyArenas)   // test hook (KAMD_TEST_TINY_ARENAS): regions far too small at scale 1, so that the overflow -> re-run ladder is exercised
			{
				mcap = (n / 2 + 8) * sc; ncap = std::min<uint64_t>((n / 2 + 8) * sc, 0xFFE0); scap = (n + 16) * sc; tcap = (n / 4 + 4) * sc;
			}
			if (slotMode && sc == 1 && !tinyArenas && I.stateScaleForced) scap = std::max<uint64_t>(scap * I.stateScale64[0] / 64, 64);      // (KAMD_STATE_SCALE: tests of the growth into the pool)
			// (half the worst case: two chunks in a thousand need more -- c3-sbg: the 99.9th percentile used 35/64 of it -- and grow into the pool)
			if (slotMode && sc == 1 && !tinyArenas && !I.stateScaleForced) scap = std::max<uint64_t>(scap / 2, 64);
			if (slotMode) { z.slot = scap; scap = 0; }      // (the arenas are the search kernel's lane groups', each large enough for the batch's longest chunk)
			z.mcap = mcap; z.ncap = ncap; z.scap = scap; z.tcap = tcap;
			return z;
		};
		// running sums in two passes over blocks of chunks on the workers (one loop on the calling thread was 0.9 ms per 65 536 chunks): the blocks' totals,
		// their prefix on this thread, then every block writes its chunks' offsets from its own base
		constexpr size_t kOffBlock = 2048;
		const size_t nBlk = (nC + kOffBlock - 1) / kOffBlock;
		std::vector<ChunkSizes> blkBase(nBlk + 1, ChunkSizes{});
		HostPool::instance().run(nBlk, 1, b.hostThreads, [&](size_t k0, size_t k1, int)
		{
			for (size_t k = k0; k < k1; ++k)
			{
				ChunkSizes t{};
				for (size_t c = k * kOffBlock; c < std::min(nC, (k + 1) * kOffBlock); ++c)
				{
					const ChunkSizes z = sizesOf(c);
					t.n += z.n; t.pat += z.pat; t.sp += z.sp; t.mcap += z.mcap; t.ncap += z.ncap; t.scap += z.scap; t.tcap += z.tcap; t.slot = std::max(t.slot, z.slot);
				}
				blkBase[k + 1] = t;
			}
		});
		for (size_t k = 0; k < nBlk; ++k)
		{
			ChunkSizes& t = blkBase[k + 1]; const ChunkSizes& p = blkBase[k];
			slotCap = std::max(slotCap, t.slot);
			t.n += p.n; t.pat += p.pat; t.sp += p.sp; t.mcap += p.mcap; t.ncap += p.ncap; t.scap += p.scap; t.tcap += p.tcap;
		}
		{
			const ChunkSizes& t = blkBase[nBlk];
			if (t.mcap > 0xFFFFFFFFull || t.ncap > 0xFFFFFFFFull || 3 * t.ncap > 0xFFFFFFFFull || t.n > 0xFFFFFFFFull || t.pat > 0xFFFFFFFFull || t.sp > 0xFFFFFFFFull)
				throw std::runtime_error{ "batch too large for 32-bit scratch offsets: split the batch" };
		}
		HostPool::instance().run(nBlk, 1, b.hostThreads, [&](size_t k0, size_t k1, int)
		{
			for (size_t k = k0; k < k1; ++k)
			{
				ChunkSizes t = blkBase[k];
				for (size_t c = k * kOffBlock; c < std::min(nC, (k + 1) * kOffBlock); ++c)
				{
					const ChunkSizes z = sizesOf(c);
					t.n += z.n; t.pat += z.pat; t.sp += z.sp; t.mcap += z.mcap; t.ncap += z.ncap; t.scap += z.scap; t.tcap += z.tcap;
					b.charOff[c + 1] = (uint32_t)t.n; b.patOff[c + 1] = (uint32_t)t.pat; b.spOff[c + 1] = (uint32_t)t.sp;
					b.matchBase[c + 1] = (uint32_t)t.mcap; b.nodeBase[c + 1] = (uint32_t)t.ncap; b.packBase[c + 1] = (uint32_t)(3 * t.ncap);
					b.stateBase[c + 1] = t.scap; b.tokenBase[c + 1] = t.tcap;
				}
			}
		});
		tmL.lap("offsets per chunk");
		const size_t totChars = b.charOff[nC];
		// the input block: sections at 256-byte boundaries, same layout on the host (pinned) and on the device
		size_t top = 0;
		auto take = [&](size_t bytes) { const size_t at = top; top = (top + std::max<size_t>(bytes, 16) + 255) & ~(size_t)255; return at; };
		const size_t oChars = take(2 * totChars), oCls = take(totChars), oScript = take(totChars);
		const size_t oCharOff = take(4 * (nC + 1)), oPatOff = take(4 * (nC + 1)), oPats = take(sizeof(DevPattern) * (size_t)b.patOff[nC]);
		const size_t oSpOff = take(4 * (nC + 1)), oSp = take(b.spOff[nC]), oFlags = take(nC), oTextOff = take(4 * nC);
		// Match::oovChrFreqModel: the filtered normalised texts of the batch, once per text, and where each chunk's text lies
		const bool chrFreq = ((b.match >> 8) & 3) > 1;
		// (only the texts this batch's chunks come from: a re-run of one chunk borrows the parent batch's whole `prep` -- ADVICE r04)
		std::vector<uint32_t> filtAt; std::vector<uint32_t> filtTexts;
		size_t totFilt = 0;
		if (chrFreq)
		{
			constexpr uint32_t kUnused = 0xFFFFFFFFu;
			filtAt.assign(b.prep.size(), kUnused);
			for (const auto& r : b.refs) if (filtAt[r.text] == kUnused) { filtAt[r.text] = 0; filtTexts.push_back((uint32_t)r.text); }
			std::sort(filtTexts.begin(), filtTexts.end());
			for (const uint32_t t : filtTexts) { if (totFilt > 0xFFFFFFF0ull) break; filtAt[t] = (uint32_t)totFilt; totFilt += b.prep[t].norm.size(); }
			if (totFilt > 0xFFFFFFF0ull) throw std::runtime_error{ "batch too large for 32-bit text offsets: split the batch" };
		}
		const size_t oFilt = take(2 * totFilt), oFiltOff = take(chrFreq ? 4 * nC : 0), oFiltLen = take(chrFreq ? 4 * nC : 0);
		const size_t oMatchBase = take(4 * (nC + 1)), oNodeBase = take(4 * (nC + 1)), oPackBase = take(4 * (nC + 1)), oStateBase = take(8 * (nC + 1)), oTokenBase = take(8 * (nC + 1));
		b.hIn.ensure(top); b.dIn.ensure(top);
		uint8_t* H = b.hIn.as<uint8_t>();
		std::memcpy(H + oCharOff, b.charOff.data(), 4 * (nC + 1)); std::memcpy(H + oPatOff, b.patOff.data(), 4 * (nC + 1)); std::memcpy(H + oSpOff, b.spOff.data(), 4 * (nC + 1));
		std::memcpy(H + oMatchBase, b.matchBase.data(), 4 * (nC + 1)); std::memcpy(H + oNodeBase, b.nodeBase.data(), 4 * (nC + 1)); std::memcpy(H + oPackBase, b.packBase.data(), 4 * (nC + 1));
		std::memcpy(H + oStateBase, b.stateBase.data(), 8 * (nC + 1)); std::memcpy(H + oTokenBase, b.tokenBase.data(), 8 * (nC + 1));
		std::atomic<uint64_t> units{ 0 };
		HostPool::instance().run(nC, 512, b.hostThreads, [&](size_t c0, size_t c1, int)
		{
			uint64_t u = 0;
			for (size_t c = c0; c < c1; ++c)
			{
				const auto& r = b.refs[c];
				const PreparedView& pt = b.prep[r.text];
				const ChunkDesc& d = pt.chunks[r.chunk];
				std::memcpy(H + oChars + 2 * (size_t)b.charOff[c], pt.norm.data() + d.startOffset, 2 * (size_t)d.nChars);
				std::memcpy(H + oCls + b.charOff[c], pt.cls.data() + d.startOffset, d.nChars);
				std::memcpy(H + oScript + b.charOff[c], pt.script.data() + d.startOffset, d.nChars);
				DevPattern* pats = reinterpret_cast<DevPattern*>(H + oPats) + b.patOff[c];
				for (uint32_t k = d.patBegin; k < d.patEnd; ++k) pats[k - d.patBegin] = DevPattern{ pt.patterns[k].end, pt.patterns[k].length, pt.patterns[k].tag };
				{ uint32_t at = d.patEnd - d.patBegin; forSpansOfChunk(b, r.text, d, [&](const DevPattern& sp) { pats[at++] = sp; }); }
				for (uint32_t k = 0; k < d.nChars; ++k) if (!isSpace(pt.norm[d.startOffset + k])) ++u;
				if (!r.sp.empty()) std::memcpy(H + oSp + b.spOff[c], r.sp.data(), r.sp.size());
				H[oFlags + c] = (r.openEnding ? 1 : 0) | (r.onlyChunk ? 2 : 0);
				reinterpret_cast<uint32_t*>(H + oTextOff)[c] = d.startOffset;
				if (chrFreq) { reinterpret_cast<uint32_t*>(H + oFiltOff)[c] = filtAt[r.text]; reinterpret_cast<uint32_t*>(H + oFiltLen)[c] = (uint32_t)pt.norm.size(); }
			}
			units += u;
		});
		if (chrFreq)
		{
			// Kiwi.cpp:1064-1084: identifySpecialChr of every UTF-16 UNIT (a surrogate on its own, unlike `cls`, which types a pair at its first unit)
			const uint8_t hiType = identifySpecialChr(0xD800), loType = identifySpecialChr(0xDC00);
			HostPool::instance().run(filtTexts.size(), 256, b.hostThreads, [&](size_t t0, size_t t1, int)
			{
				for (size_t ti = t0; ti < t1; ++ti)
				{
					const size_t t = filtTexts[ti];
					const PreparedView& pt = b.prep[t];
					uint16_t* o = reinterpret_cast<uint16_t*>(H + oFilt) + filtAt[t];
					for (size_t k = 0; k < pt.norm.size(); ++k)
					{
						const uint16_t c = (uint16_t)pt.norm[k];
						const uint8_t type = isHighSurrogate(c) ? hiType : isLowSurrogate(c) ? loType : (uint8_t)(pt.cls[k] & 0x7F);
						o[k] = chrFreqFiltered(type) ? (uint16_t)u' ' : c;
					}
				}
			});
		}
		b.units = units;
		tmL.lap("input block (workers)");
		hipStream_t s = I.streamCopy;      // (uploads, and the typo graph kernels over this batch's own buffers: never behind another batch's kernels)
		if (top) HIPCHECK(hipMemcpyAsync(b.dIn.p, H, top, hipMemcpyHostToDevice, s));
		uint8_t* D = b.dIn.as<uint8_t>();
		const size_t perChar = totChars + nC + 16;
		const size_t totNodes = b.nodeBase[nC], totMatch = b.matchBase[nC];
		// the pool behind the chunks' own arenas (none under KAMD_TEST_TINY_ARENAS: that hook is there to exercise the re-run ladder)
		if (slotCap > 0x7FFFFFFFull) throw std::runtime_error{ "chunk too long for a state arena" };
		b.slotCap = (uint32_t)slotCap;
		// (slot mode: as many arenas as a launch can use -- lane group blockIdx * groups + g of min(persistent blocks, ceil(chunks / groups)) blocks --, not the whole
		// machine's whatever the batch holds: a re-run of one chunk at a high rung of the capacity ladder took gigabytes otherwise)
		const uint64_t ownStates = slotMode ? std::min<uint64_t>(I.histSlots(), (uint64_t)nC + 4) * slotCap : b.stateBase[nC];
		b.poolStates = tinyArenas ? 0 : slotMode ? (I.poolForced ? ownStates * I.poolFrac64 / 64 : ownStates / 4)
			: std::max<uint64_t>(ownStates * I.poolFrac64 / 64, I.poolFrac64 ? std::min<uint64_t>(ownStates, 1u << 16) : 0);
		b.ownStates = ownStates;
		const uint64_t totStates = ownStates + b.poolStates, totTokens = b.tokenBase[nC];
		b.dNsToPos.ensure(perChar * 2); b.dPosToNs.ensure(perChar * 2); b.dCflag.ensure(perChar); b.dMask.ensure(perChar * 8); b.dMoff.ensure(perChar * 4);
		b.dNNs.ensure(nC * 4 + 16); b.dMatchForm.ensure(totMatch * 4 + 16);
		b.dNodes.ensure(totNodes * sizeof(DevNode) + 16); b.dTmpNodes.ensure(totNodes * sizeof(DevNode) + 16);
		b.dEndPosMap.ensure(perChar * 4); b.dFullMask.ensure(perChar * 8); b.dZAt.ensure(perChar); b.dTmpIdx.ensure(totNodes * 4 + 16); b.dNNodes.ensure(nC * 4 + 16); b.dWideList.ensure(nC * 4 + 16); b.dExpanded.ensure(nC + 16);
		b.dPacks.ensure((size_t)b.packBase[nC] * sizeof(CandStatic) + 16);
		b.dStates.ensure(totStates * sizeof(DevState) + 16); b.dNodeStOff.ensure(totNodes * 4 + 16); b.dNodeStCnt.ensure(totNodes * 4 + 16); b.dReach.ensure(totNodes + 16);
		b.dTokens.ensure(totTokens * sizeof(DevToken) + 16); b.dResults.ensure(nC * sizeof(DevChunkResult) + 16);
		const bool posPath = I.posPath && !I.histStates();
		if (posPath)
		{
			b.dPosRecs.ensure((size_t)b.packBase[nC] * sizeof(PosRec) + 16); b.dPosDesc.ensure(totNodes * sizeof(PosDesc) + 16);
			b.dPosPrev.ensure(totNodes * 4 + 16); b.dPosNodeRec.ensure(totNodes * 4 + 16); b.dPosMask.ensure(totNodes * (b.typo.typo ? 8 : 4) + 16) /* typo lattices: a second word per position, one batch of nodes behind the first */; b.dPosBig.ensure(((nC + 7) / 8 * 8) * (size_t)64 * 20 + 16);
		}
		if (I.histStates()) b.dHist.ensure(totStates * 32 + 32);
		// compact outputs of the end stage: as many token records as the arenas could hold, 16 path headers per chunk (x capacity scale)
		b.outTokCap = (uint32_t)std::min<uint64_t>(totTokens, 0xFFFFFFF0ull); b.outPathCap = (uint32_t)std::min<uint64_t>((uint64_t)nC * 16 * sc, 0xFFFFFFF0ull);
		b.dOutTokens.ensure((size_t)b.outTokCap * sizeof(DevToken) + 16); b.dOutPaths.ensure((size_t)b.outPathCap * sizeof(DevPathHeader) + 16); b.dOutCounters.ensure(128);
		b.dStateAt.ensure(nC * 8 + 16); if (slotMode) b.dSlotTable.ensure((size_t)I.histSlots() * 16 + 16);
		b.devBytes = 0;
		for (const DevBuf* d : { &b.dIn, &b.dOutTokens, &b.dOutPaths, &b.dNsToPos, &b.dPosToNs, &b.dCflag, &b.dMask, &b.dMoff, &b.dMatchForm, &b.dNodes, &b.dTmpNodes, &b.dEndPosMap, &b.dTmpIdx,
			&b.dPacks, &b.dStates, &b.dHist, &b.dStateAt, &b.dSlotTable, &b.dNodeStOff, &b.dNodeStCnt, &b.dReach, &b.dTokens, &b.dResults, &b.dPosRecs, &b.dPosDesc, &b.dPosPrev, &b.dPosNodeRec, &b.dPosMask, &b.dPosBig }) b.devBytes += d->cap;

		BatchView& bv = b.bv;
		bv.nChunks = (uint32_t)nC; bv.chars = (const uint16_t*)(D + oChars); bv.cls = D + oCls; bv.script = D + oScript;
		bv.charOff = (const uint32_t*)(D + oCharOff); bv.patOff = (const uint32_t*)(D + oPatOff); bv.patterns = (const DevPattern*)(D + oPats);
		bv.spOff = (const uint32_t*)(D + oSpOff); bv.spStates = D + oSp; bv.chunkFlags = D + oFlags; bv.textOffset = (const uint32_t*)(D + oTextOff);
		bv.filtChars = chrFreq ? (const uint16_t*)(D + oFilt) : nullptr; bv.filtOff = chrFreq ? (const uint32_t*)(D + oFiltOff) : nullptr; bv.filtLen = chrFreq ? (const uint32_t*)(D + oFiltLen) : nullptr;
		WorkView& w = b.wv;
		w.nsToPos = b.dNsToPos.as<uint16_t>(); w.posToNs = b.dPosToNs.as<uint16_t>(); w.cflag = b.dCflag.as<uint8_t>();
		w.matchMask = b.dMask.as<uint64_t>(); w.matchOff = b.dMoff.as<uint32_t>(); w.nNs = b.dNNs.as<uint32_t>();
		w.matchBase = (const uint32_t*)(D + oMatchBase); w.matchForm = b.dMatchForm.as<uint32_t>();
		w.nodeBase = (const uint32_t*)(D + oNodeBase); w.nodes = b.dNodes.as<DevNode>(); w.tmpNodes = b.dTmpNodes.as<DevNode>();
		w.endPosMap = b.dEndPosMap.as<uint32_t>(); w.fullMask = b.dFullMask.as<uint64_t>(); w.zAt = b.dZAt.as<uint8_t>(); w.tmpIdx = b.dTmpIdx.as<uint16_t>(); w.nNodes = b.dNNodes.as<uint32_t>(); w.wideList = b.dWideList.as<uint32_t>(); w.expanded = b.dExpanded.as<uint8_t>();
		w.packBase = (const uint32_t*)(D + oPackBase); w.packs = b.dPacks.as<CandStatic>();
		w.stateBase = (const uint64_t*)(D + oStateBase); w.states = b.dStates.as<DevState>();
		w.stateAt = b.dStateAt.as<uint64_t>(); w.poolTop = b.poolStates ? reinterpret_cast<unsigned long long*>(b.dOutCounters.as<uint32_t>() + 16) : nullptr;      // (counters[16..17]: cleared with the others before every run)
		w.poolBase = b.ownStates; w.poolCap = b.poolStates; w.slotCap = b.slotCap; w.slotTable = b.slotCap ? b.dSlotTable.as<uint64_t>() : nullptr;
		w.nodeStateOff = b.dNodeStOff.as<uint32_t>(); w.nodeStateCnt = b.dNodeStCnt.as<uint32_t>(); w.reach = b.dReach.as<uint8_t>();
		w.tokenBase = (const uint64_t*)(D + oTokenBase); w.tokens = b.dTokens.as<DevToken>(); w.results = b.dResults.as<DevChunkResult>();
		w.outTokens = b.dOutTokens.as<DevToken>(); w.outPaths = b.dOutPaths.as<DevPathHeader>(); w.outCounters = b.dOutCounters.as<uint32_t>();
		w.outTokCap = b.outTokCap; w.outPathCap = b.outPathCap;
		w.posRecs = posPath ? b.dPosRecs.as<PosRec>() : nullptr; w.posDesc = posPath ? b.dPosDesc.as<PosDesc>() : nullptr;
		w.posPrev = posPath ? b.dPosPrev.as<uint32_t>() : nullptr; w.posNodeRec = posPath ? b.dPosNodeRec.as<uint32_t>() : nullptr; w.posMask = posPath ? b.dPosMask.as<uint32_t>() : nullptr; w.posBig = posPath ? b.dPosBig.as<uint8_t>() : nullptr;
		w.posHandOver = nullptr; w.posScratch = nullptr; w.posContCounter = nullptr; w.posContSlots = 0;
		w.blockBits = nullptr;
		w.unkChr = nullptr;
		w.unkChrForm = nullptr;
		if ((b.match >> 8) & 3) { b.dUnkChr.ensure(totNodes * 4 + 16); w.unkChr = b.dUnkChr.as<float>(); }      // Match::oovChrModel (checked in stage())
		if (((b.match >> 8) & 3) > 1) { b.dUnkChrForm.ensure(totNodes * 4 + 16); w.unkChrForm = b.dUnkChrForm.as<float>(); }      // Match::oovChrFreqModel / oovChrFreqBranchModel
		if (b.typo.blocked && !b.typo.blocked->empty() && b.typo.blocked->size() != (I.model.morphs.size() + 31) / 32) throw std::invalid_argument{ "kiwi_amd: blocklist bit set does not belong to this model" };
		const size_t nTempMorphs = b.pretok ? b.pretok->temps.morphs.size() : 0;      // (a temporary morpheme is never blocked; its bit has to exist)
		if (!I.model.morphDialect.empty() || (nTempMorphs && b.typo.blocked && !b.typo.blocked->empty()))
		{
			// a model with dialect morphemes: those of a dialect this analysis does not allow are skipped by the candidate loops exactly like blocked ones
			// (PathEvaluator.hpp:386, 893 beside the blocklist test) -- one bit set per batch, the blocklist's united with them
			std::vector<uint32_t>& bits = b.blockBitsHost;      // (kept with the batch: the upload is asynchronous)
			bits.assign((I.model.morphs.size() + 31) / 32, 0u);
			if (b.typo.blocked && !b.typo.blocked->empty()) bits = *b.typo.blocked;
			bool any = b.typo.blocked && !b.typo.blocked->empty();
			for (size_t i = 0; i < I.model.morphDialect.size(); ++i) { const uint32_t d = I.model.morphDialect[i]; if (d && !(d & b.typo.allowedDialect)) { bits[i >> 5] |= 1u << (i & 31); any = true; } }
			bits.resize((I.model.morphs.size() + nTempMorphs + 31) / 32, 0u);
			// (nothing blocked -- every dialect allowed, no blocklist: no bit set is bound, and the lattice kernel keeps writing the search's records itself)
			if (any) { upload(b.dBlockBits, bits, s); w.blockBits = b.dBlockBits.as<uint32_t>(); }
		}
		else if (b.typo.blocked && !b.typo.blocked->empty())
		{
			upload(b.dBlockBits, *b.typo.blocked, s);
			w.blockBits = b.dBlockBits.as<uint32_t>();
		}
		w.bigScratch = nullptr; w.bigScratchBytes = 0;   // bound at launch
		b.subBatches = 0;
		tmL.lap("device buffers");
		if (b.typo.typo)
		{
			// typo graphs on the DEVICE (typo_graph_kernel.hip; row f3): a count pass over the text block that is already on its way, the counts
			// come back (the state arenas and the LDS size classes of the lattice build are laid out from them), then the write pass into exactly
			// sized regions.  The host keeps PreparedTypo::graph for the parity hooks only.
			const PreparedTypo& T = *b.typo.typo;
			if (T.maxCtiBound() > kTypoMaxContinual + 1) throw std::runtime_error{ "kiwi_amd: a typo pattern with more than 32 distinct continual replacements is not supported" };
			std::vector<TypoLatChunk> tch(nC); std::vector<TypoGraphChunk> gch(nC);
			HostPool::instance().run(nC, 256, b.hostThreads, [&](size_t c0, size_t c1, int)
			{
				for (size_t c = c0; c < c1; ++c)
				{
					const auto& r = b.refs[c];
					const PreparedView& pt = b.prep[r.text];
					const ChunkDesc& d = pt.chunks[r.chunk];
					const char16_t* str = (const char16_t*)pt.norm.data() + d.startOffset;
					TypoLatChunk& t = tch[c];
					t = TypoLatChunk{};
					t.charOff = b.charOff[c]; t.nChars = d.nChars; t.textOffset = d.startOffset; t.chunkId = (uint32_t)c;
					t.patOff = b.patOff[c]; t.patCnt = b.patOff[c + 1] - b.patOff[c];
					for (uint32_t i = 0; i < d.nChars; ++i) if (!isSpace(str[i])) { ++t.nNs; if (isHighSurrogate(str[i]) && i + 1 < d.nChars) { ++t.nNs; ++i; } }
					t.nodeOff = b.nodeBase[c]; t.nodeCap = b.nodeBase[c + 1] - b.nodeBase[c]; t.packCap = b.packBase[c + 1] - b.packBase[c];
				}
			});
			TypoGraphDev& gd = b.typoDev;      // (the lattice build reads replacement strings from the pool uploaded here: it lives as long as the batch)
			uploadTypoTables(gd, T, s);
			uint64_t scrTop = 0;
			for (size_t c = 0; c < nC; ++c)
			{
				gch[c] = TypoGraphChunk{ tch[c].charOff, tch[c].nChars, 0, 0, (uint32_t)scrTop, T.scratchCapFor(tch[c].nChars, 0) };
				scrTop += gch[c].scrCap;
				if (scrTop > 0xFFFFFFF0ull) throw std::runtime_error{ "batch too large for 32-bit typo scratch offsets: split the batch" };
			}
			std::vector<TypoGraphOut> gout(nC);
			DevBuf dGch, dGout, dMatches, dBp, dEpm, dRev, dCnt, dTemp;
			TypoGraphView gv{};
			gv.chars = bv.chars; gv.cls = bv.cls; gv.script = bv.script; gv.allowedDialect = b.typo.allowedDialect;
			upload(dGch, gch, s); dGout.ensure(nC * sizeof(TypoGraphOut) + 16);
			dMatches.ensure(scrTop * 8 + 16); dBp.ensure(scrTop * 4 + 16); dEpm.ensure(scrTop * 8 + 16);
			gv.chunks = dGch.as<TypoGraphChunk>(); gv.out = dGout.as<TypoGraphOut>(); gv.matches = dMatches.as<uint2>(); gv.bp = dBp.as<uint32_t>(); gv.epm = dEpm.as<uint2>();
			launchTypoGraph(gd.tables, gv, (uint32_t)nC, true, s);
			HIPCHECK(hipGetLastError());
			HIPCHECK(hipMemcpyAsync(gout.data(), dGout.p, nC * sizeof(TypoGraphOut), hipMemcpyDeviceToHost, s));
			HIPCHECK(hipStreamSynchronize(s));
			uint64_t mapTop = 0, nsTop = 0, stateTop = 0, graphTop = 0;
			scrTop = 0;
			// (developer knob KAMD_TYPO_LDS_CAP="<mul4>x<add>": nodes per text unit x 4 and the constant of typoLdsNodeCap)
			uint32_t typoCapMul4 = 14, typoCapAdd = 40;      // (3.5 nodes per text unit + 40: the chunks of c5 build 2.2 at the median, 3.5 at most -- profiles/r06_p)
			const bool typoLdsTables = [] { const char* e = std::getenv("KAMD_TYPO_LDS_TABLES"); return !e || std::atoi(e) != 0; }();      // (developer switch: 0 = graph / state ranges / state heads in HBM as in rounds 2 - 5)
			if (const char* e = std::getenv("KAMD_TYPO_LDS_CAP")) { unsigned a = 0, c2 = 0; if (std::sscanf(e, "%ux%u", &a, &c2) == 2 && a) { typoCapMul4 = a; typoCapAdd = c2; } }
			for (size_t c = 0; c < nC; ++c)
			{
				TypoLatChunk& t = tch[c];
				if (gout[c].status) throw std::runtime_error{ "kiwi_amd: typo graph kernel status " + std::to_string(gout[c].status) + (gout[c].status == 3 ? " (a typo pattern with an explicit NUL)" : "") };
				t.graphCnt = gout[c].graphCnt;
				if (gout[c].maxCti > 1) { size_t v = gout[c].maxCti - 1; while (v > 0) { v >>= 1; ++t.pmb; } }
				t.mapLen = (t.nNs << t.pmb) + 1;
				t.ldsCap = typoLdsNodeCap(t.nChars, t.nodeCap, typoCapMul4, typoCapAdd);
				t.ldsGraphCap = (typoLdsTables && t.graphCnt <= kTypoLdsGraphMax && t.stateCap <= 0xFFFFu && t.nChars <= 0xFFF0u) ? t.graphCnt : 0u;      // (what the 16-byte graph records and the packed state ranges can hold)
				t.ldsNeed = typoLdsLayout(t.nChars, t.nNs, t.pmb, t.ldsCap, t.ldsGraphCap).total;
				t.graphOff = (uint32_t)graphTop; graphTop += t.graphCnt;
				t.mapOff = (uint32_t)mapTop; mapTop += t.mapLen;
				t.nsOff = (uint32_t)nsTop; nsTop += t.nChars + 2;
				t.stateOff = (uint32_t)stateTop; t.stateCap = (uint32_t)std::min<uint64_t>(((uint64_t)t.graphCnt * 16 + 64) * sc, 0x7FFFFFFF); stateTop += t.stateCap;
				gch[c].graphOff = t.graphOff; gch[c].graphCap = t.graphCnt; gch[c].scrOff = (uint32_t)scrTop; gch[c].scrCap = T.scratchCapFor(t.nChars, t.graphCnt);
				scrTop += gch[c].scrCap;
				if (mapTop > 0xFFFFFFF0ull || stateTop > 0xFFFFFFF0ull || graphTop > 0xFFFFFFF0ull || scrTop > 0xFFFFFFF0ull) throw std::runtime_error{ "batch too large for 32-bit typo scratch offsets: split the batch" };
			}
			b.typoNeed.resize(nC); for (size_t c = 0; c < nC; ++c) b.typoNeed[c] = tch[c].ldsNeed;
			const size_t graphSize = std::max<uint64_t>(graphTop, 1);
			b.dTypoGraph.ensure(graphSize * sizeof(TypoGraphNode) + 64); b.dTypoLast.ensure(graphSize * 2 + 64);
			dTemp.ensure(graphSize * sizeof(TypoGraphNode) + 64);
			dMatches.ensure(scrTop * 8 + 16); dBp.ensure(scrTop * 4 + 16); dEpm.ensure(scrTop * 8 + 16); dRev.ensure(scrTop * 4 + 16); dCnt.ensure(scrTop * 4 + 16);
			upload(dGch, gch, s);
			gv.chunks = dGch.as<TypoGraphChunk>(); gv.matches = dMatches.as<uint2>(); gv.bp = dBp.as<uint32_t>(); gv.epm = dEpm.as<uint2>(); gv.rev = dRev.as<uint32_t>(); gv.cnt = dCnt.as<uint32_t>();
			gv.graph = b.dTypoGraph.as<TypoGraphNode>(); gv.graphLast = b.dTypoLast.as<uint8_t>(); gv.temp = dTemp.as<TypoGraphNode>();
			launchTypoGraph(gd.tables, gv, (uint32_t)nC, false, s);
			HIPCHECK(hipGetLastError());
			upload(b.dTypoChunks, tch, s);
			b.dTypoTmp.ensure(totNodes * sizeof(TypoLatNode) + 64); b.dTypoMap.ensure(mapTop * 8 + 64); b.dTypoNs.ensure(nsTop * 2 + 64); b.dTypoPs.ensure(nsTop * 2 + 64);
			b.dTypoStates.ensure(stateTop * sizeof(TypoState) + 64); b.dTypoSIdx.ensure(graphSize * 8 + 64); b.dTypoScratch.ensure(totNodes * 12 + 64);
			b.dNodeTypo.ensure(totNodes * 4 + 64);
			TypoLatView& v = b.tv;
			v = TypoLatView{};
			v.chars = bv.chars; v.cls = bv.cls; v.script = bv.script; v.patterns = bv.patterns;
			v.graph = b.dTypoGraph.as<TypoGraphNode>(); v.graphLast = b.dTypoLast.as<uint8_t>(); v.pool = gd.tables.pool; v.chunks = b.dTypoChunks.as<TypoLatChunk>();
			v.nodes = b.dTypoTmp.as<TypoLatNode>(); v.nodesFinal = nullptr; v.endPosMap = b.dTypoMap.as<uint2>(); v.nsToPos = b.dTypoNs.as<uint16_t>(); v.posToNs = b.dTypoPs.as<uint16_t>();
			v.states = b.dTypoStates.as<TypoState>(); v.stateIdx = b.dTypoSIdx.as<uint32_t>(); v.scratch = b.dTypoScratch.as<uint32_t>();
			v.devNodes = w.nodes; v.nodeTypo = b.dNodeTypo.as<float>(); v.nNodes = w.nNodes; v.results = w.results;
			HIPCHECK(hipStreamSynchronize(s));      // the host vectors above are the sources of asynchronous copies: they must outlive them
		}
		HIPCHECK(hipStreamSynchronize(s));
		tmL.lap("waiting for the upload");
		b.ran = false;
	}

	// Probe of csrc/exact_math.hpp on the device (tests/test_exact_math.py): e[i] = expf_glibc(x[i]), l[i] = logf_glibc(x[i]), t[i] = tanhf_glibc(x[i]) (each where asked for)
	__global__ void k_exact_math_probe(const float* x, float* e, float* l, float* t, uint32_t n)
	{
		const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
		if (i >= n) return;
		if (e) e[i] = exact::expf_glibc(x[i]);
		if (l) l[i] = exact::logf_glibc(x[i]);
		if (t) t[i] = exact::tanhf_glibc(x[i]);
	}
	void exactMathProbe(const float* x, float* e, float* l, float* t, uint32_t n)
	{
		DevBuf dx, de, dl, dt;
		dx.ensure((size_t)n * 4 + 16); de.ensure((size_t)n * 4 + 16); dl.ensure((size_t)n * 4 + 16); dt.ensure((size_t)n * 4 + 16);
		HIPCHECK(hipMemcpy(dx.p, x, (size_t)n * 4, hipMemcpyHostToDevice));
		hipLaunchKernelGGL(k_exact_math_probe, dim3((n + 255) / 256), dim3(256), 0, 0, dx.as<float>(), e ? de.as<float>() : (float*)nullptr, l ? dl.as<float>() : (float*)nullptr, t ? dt.as<float>() : (float*)nullptr, n);
		HIPCHECK(hipGetLastError());
		if (e) HIPCHECK(hipMemcpy(e, de.p, (size_t)n * 4, hipMemcpyDeviceToHost));
		if (l) HIPCHECK(hipMemcpy(l, dl.p, (size_t)n * 4, hipMemcpyDeviceToHost));
		if (t) HIPCHECK(hipMemcpy(t, dt.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	}

	// Probe of csrc/cong_global.hpp on the device (tests/test_cong_global.py): the score of `next[i]` after context ctx[i] and the seven history words hist[i][0..6]
	// under the GLOBAL CoNgram model -- flags[i] bit 0: the progressMatrix* entry instead of state.next(), bit 1: output scale first -- and the local score for a
	// word that is no valid distant token.  The first device piece of that model type (the search kernel does not use it yet): the arithmetic and the layout of
	// the window sections in HBM, checked bit for bit against the host evaluation of the same header that the oracle pins to the real reference.
	__global__ void k_congg_probe(CongView C, const uint32_t* ctx, const uint32_t* hist, const uint32_t* next, const uint8_t* flags, float* out, uint32_t n)
	{
		const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
		if (i >= n) return;
		uint32_t h[7];
		for (int k = 0; k < 7; ++k) h[k] = hist[7ull * i + k];
		const bool matrix = flags[i] & 1, outFirst = flags[i] & 2;
		if (C.distant(next[i])) out[i] = matrix ? congg::scoreMatrix(C, ctx[i], h, next[i], outFirst) : congg::scoreSingle(C, ctx[i], h, next[i]);
		else out[i] = outFirst ? congScoreOutputFirst(C, ctx[i], next[i]) : congScore(C, ctx[i], next[i]);
	}
	void conggProbe(const FlatModel& m, const uint32_t* ctx, const uint32_t* hist, const uint32_t* next, const uint8_t* flags, float* out, uint32_t n)
	{
		if (!m.congDim || !m.congWindow) throw std::runtime_error{ "the model has no CoNgram window sections" };
		DevBuf dCtxEmb, dOutEmb, dDistEmb, dCtxConf, dDistConf, dPosConf, dMask, dCtx, dHist, dNext, dFlags, dOut;
		auto up = [](DevBuf& b, const void* p, size_t bytes) { b.ensure(bytes + 16); HIPCHECK(hipMemcpy(b.p, p, bytes, hipMemcpyHostToDevice)); };
		up(dCtxEmb, m.congCtxEmb.data(), m.congCtxEmb.size()); up(dOutEmb, m.congOutEmb.data(), m.congOutEmb.size()); up(dDistEmb, m.congDistEmb.data(), m.congDistEmb.size());
		up(dCtxConf, m.congCtxConf.data(), m.congCtxConf.size() * 4); up(dDistConf, m.congDistConf.data(), m.congDistConf.size() * 4);
		up(dPosConf, m.congPosConf.data(), m.congPosConf.size() * 4); up(dMask, m.congDistMask.data(), m.congDistMask.size());
		up(dCtx, ctx, (size_t)n * 4); up(dHist, hist, (size_t)n * 28); up(dNext, next, (size_t)n * 4); up(dFlags, flags, n);
		dOut.ensure((size_t)n * 4 + 16);
		CongView C;
		C.dim = m.congDim; C.stride = m.congDim + 8; C.nCtx = m.congCtx; C.vocabSize = m.congVocab; C.window = m.congWindow; C.keyBytes = m.congKeyBytes;
		C.ctxEmb = dCtxEmb.as<uint8_t>(); C.outEmb = dOutEmb.as<uint8_t>(); C.distEmb = dDistEmb.as<uint8_t>();
		C.ctxConf = dCtxConf.as<float>(); C.distConf = dDistConf.as<float>(); C.posConf = dPosConf.as<float>(); C.distMask = dMask.as<uint8_t>();
		hipLaunchKernelGGL(k_congg_probe, dim3((n + 63) / 64), dim3(64), 0, 0, C, dCtx.as<uint32_t>(), dHist.as<uint32_t>(), dNext.as<uint32_t>(), dFlags.as<uint8_t>(), dOut.as<float>(), n);
		HIPCHECK(hipGetLastError());
		HIPCHECK(hipMemcpy(out, dOut.p, (size_t)n * 4, hipMemcpyDeviceToHost));
	}

	static SearchParams makeParams(const EngineConfig& c, uint64_t match, uint32_t topN = 1, const TypoOption* opt = nullptr)
	{
		SearchParams p{};
		p.allowedDialect = opt ? opt->allowedDialect : 0u; p.dialectCost = opt ? opt->dialectCost : 3.f;
		p.match = match; p.cutOff = c.cutOffThreshold; p.spacePenalty = c.spacePenalty; p.typoCostWeight = c.typoCostWeight;
		p.oovRuleScale = c.oovRuleScale; p.oovRuleBias = c.oovRuleBias;
		p.oovGlobalWeight = c.oovGlobalWeight; p.oovLocalWeight = c.oovLocalWeight; p.oovGlobalMinFreq = c.oovGlobalMinFreq; p.oovChrFreqBias = c.oovChrBias;
		p.oovChrBias = ((match >> 8) & 3) > 1 ? 0.f : c.oovChrBias;      // (the frequency-based scores arrive with the bias applied: WorkView::unkChrForm)
		p.maxUnk = c.maxUnkFormSize; p.maxUnkJ = c.maxUnkFormSizeFollowedByJClass; p.spaceTol = c.spaceTolerance;
		p.splitComplex = (match & M_SPLIT_COMPLEX) ? 1 : 0; p.splitSaisiot = (match & M_SPLIT_SAISIOT) ? 1 : 0; p.mergeSaisiot = (match & M_MERGE_SAISIOT) ? 1 : 0;
		p.topN = topN;
		p.smallMax = 128; p.mediumMax = 512; p.bucketCap = 128;
		if (const char* l = std::getenv("KAMD_CONTAINER_LIMITS"))   // test hook: "small,medium,bucket"
		{
			unsigned a = 0, b = 0, c = 0;
			if (std::sscanf(l, "%u,%u,%u", &a, &b, &c) == 3 && a <= b && c >= 1) { p.smallMax = a; p.mediumMax = b; p.bucketCap = c; }
			else throw std::runtime_error{ "KAMD_CONTAINER_LIMITS must be small,medium,bucket" };
		}
		return p;
	}

#ifdef KAMD_TIMELINE
	static void* gTimeline = nullptr;
#endif
	static DevBuf lwProf;         // developer aid (make lwprof + KAMD_LATTICE_PROFILE=1): phase cycle sums of k_lattice_wave
	static DevBuf posBeacon;      // developer aid (KAMD_POS_DEBUG builds): progress beacons / phase timers of k_pos_path, 256 bytes per chunk (KAMD_POS_BEACON=1)
	// wait = false: returns when everything is enqueued (Engine::launch); afterLaunch() is then the caller's business once the batch's `evDone` has fired
	static void afterLaunch(Engine::Impl& I, StagedBatch& b, KernelTimes& t, bool timed);
	static KernelTimes launchAll(Engine::Impl& I, StagedBatch& b, const SearchParams& sp, bool wait = true)
	{
		KernelTimes t;
		const uint32_t nC = (uint32_t)b.refs.size();
		if (!nC) return t;
		HostTimer tm{ "launch" };
		hipStream_t sA = I.stream, sB = I.stream2;
		// Sub-batches: the lattice stages of sub-batch k+1 (few, long-running waves) overlap the search of sub-batch k
		// (many latency-bound waves) on a second stream.
		// Measured on MI355X (c2): both stages are bound by the serial latency of ONE chunk, not by the chunk count, so
		// splitting only adds launches (S=1: 4.5 ms, S=2: 6.4 ms, S=4: 9.9 ms per 8192 chunks).  Kept as an option only.
		// Round 3, 65 536 chunks with the position-step kernel (throughput-bound stages): still slower -- c2-64k 7.04 / 7.59 / 7.74 / 8.47 ms per step
		// for S = 1 / 2 / 4 / 8, c4-cong 46.3 / 49.8 / 69.3 ms: the overlapped kernels contend for the same wave slots and LDS.
		uint32_t S = I.subBatches > 0 ? (uint32_t)I.subBatches : 1u;
		S = std::min(S, std::min(nC, 16u));
		if (b.subBatches != S)
		{
			// work order of the search kernel: longest chunks first inside each sub-batch (a stable counting sort by length: this runs once per batch on the
			// calling thread, between the upload and the first launch)
			std::vector<uint32_t> order(nC);
			{
				uint32_t maxLen = 0;
				for (uint32_t c = 0; c < nC; ++c) maxLen = std::max(maxLen, b.charOff[c + 1] - b.charOff[c]);
				std::vector<uint32_t> at(maxLen + 2);
				for (uint32_t k = 0; k < S; ++k)
				{
					const uint32_t c0 = (uint32_t)((uint64_t)nC * k / S), c1 = (uint32_t)((uint64_t)nC * (k + 1) / S);
					std::fill(at.begin(), at.end(), 0u);
					for (uint32_t c = c0; c < c1; ++c) ++at[maxLen - (b.charOff[c + 1] - b.charOff[c]) + 1];      // bucket = maxLen - length: longest first
					for (uint32_t l = 0; l <= maxLen; ++l) at[l + 1] += at[l];
					for (uint32_t c = c0; c < c1; ++c) order[c0 + at[maxLen - (b.charOff[c + 1] - b.charOff[c])]++] = c;
				}
			}
			upload(b.dOrder, order, I.streamCopy);      // (not on the kernels' stream: behind a batch in flight a copy from pageable memory would hold the caller until that batch has finished)
			b.order = order;
			b.subBatches = S;
			if (b.typo.typo)
			{
				// typo lattices, wave-per-chunk kernel: largest LDS need first inside each sub-batch (one launch per size class)
				b.typoOrder.resize(nC);
				std::iota(b.typoOrder.begin(), b.typoOrder.end(), 0u);
				for (uint32_t k = 0; k < S; ++k)
				{
					const uint32_t c0 = (uint32_t)((uint64_t)nC * k / S), c1 = (uint32_t)((uint64_t)nC * (k + 1) / S);
					std::stable_sort(b.typoOrder.begin() + c0, b.typoOrder.begin() + c1, [&](uint32_t a, uint32_t c) { return b.typoNeed[a] > b.typoNeed[c]; });
				}
				upload(b.dTypoOrder, b.typoOrder, I.streamCopy);
			}
			HIPCHECK(hipStreamSynchronize(I.streamCopy));
		}
		const size_t nEv = 6 * (size_t)S + 2;
		while (I.evs.size() < nEv) { hipEvent_t e; HIPCHECK(hipEventCreate(&e)); I.evs.push_back(e); }
		if (I.haveLast) HIPCHECK(hipStreamWaitEvent(sA, I.lastDone, 0));      // the kernels of two batches do not overlap (shared scratch, counters, events)
		if (b.pretok && b.pretok->hasTemps())
		{
			// the batch's temporary forms / morphemes go behind the model's tables (ids >= the model's counts): after the previous batch's last kernel, before
			// this batch's first -- no other batch's kernels run in between, and a batch without spans never refers to an id up there
			const TempOverlay& o = b.pretok->overlay; const FlatModel& m = I.model; const ModelView& v = I.dview;
			using I_ = Engine::Impl;
			if (o.forms.size() > I_::kTempForms || o.morphs.size() > I_::kTempMorphs || o.formChars.size() > I_::kTempChars || o.formCand.size() > I_::kTempCand || o.chunkMorph.size() > I_::kTempChunks)
				throw std::invalid_argument{ "kiwi_amd: too many temporary forms / morphemes in the pretokenized spans of one call" };
			auto put = [&](const auto* devBase, size_t at, const auto& vec)
			{
				using T = std::remove_cv_t<std::remove_pointer_t<decltype(devBase)>>;
				if (!vec.empty()) HIPCHECK(hipMemcpyAsync(const_cast<T*>(devBase) + at, vec.data(), vec.size() * sizeof(T), hipMemcpyHostToDevice, sA));
			};
			put(v.forms, o.nBaseForms, o.forms); put(v.formChars, m.formChars.size() - 1, o.formChars); put(v.formCand, m.formCand.size(), o.formCand);
			put(v.morphs, o.nBaseMorphs, b.pretok->devMorphs); put(v.chunkMorph, m.chunkMorph.size(), o.chunkMorph); put(v.chunkLm, m.chunkLm.size(), o.chunkLm);
			put(v.chunkPos, m.chunkPos.size(), o.chunkPos); put(v.sbInfo, o.nBaseMorphs, o.sbInfo); put(v.morphPath, o.nBaseMorphs, o.morphPath);
			if (v.formUnkChr) { put(v.formUnkChr, o.nBaseForms, o.formUnkChr); put(v.formChrTok, m.formChars.size() - 1, o.formChrTok); }
		}
		HIPCHECK(hipMemsetAsync(b.dResults.p, 0, nC * sizeof(DevChunkResult), sA));
		HIPCHECK(hipMemsetAsync(b.dOutCounters.p, 0, 128, sA));
		HIPCHECK(hipMemcpyAsync(b.dStateAt.p, b.wv.stateBase, (size_t)nC * 8, hipMemcpyDeviceToDevice, sA));      // every chunk starts in its own arena
		if (b.slotCap) HIPCHECK(hipMemsetAsync(b.dSlotTable.p, 0, (size_t)I.histSlots() * 16, sA));      // ... every lane group in its own
		HIPCHECK(hipMemsetAsync(I.counter.p, 0, 256, sA));
		HIPCHECK(hipMemsetAsync(b.dNNodes.p, 0, (size_t)nC * 4, sA));   // also clears the lattice kernels' hand-over flag
		HIPCHECK(hipMemsetAsync(b.dExpanded.p, 0, nC, sA));
		if (getenv("KAMD_HANGDUMP")) HIPCHECK(hipMemsetAsync(b.dNodeStCnt.p, 0xFF, (size_t)b.nodeBase[nC] * 4, sA));
		// SkipBigram: one chunk per wave unless 16-lane groups are forced -- with history rings in the container keys a lattice node gathers
		// thousands of work items, so a chunk's serial chain is items / lanes (MI355X, small model: 480 texts 0.7 s with 64 lanes, 7 s with 16)
		const bool variant64 = I.histStates() ? !(I.groupLanesForced && I.groupLanes == 16) : (I.groupLanesForced && I.groupLanes == 64);
		const uint32_t nGroups = (I.histStates() || b.typo.typo || I.hasCong) ? (variant64 ? 1u : 4u)
			: 64u / (uint32_t)(I.groupLanesForced ? I.groupLanes : 8);   // most groups per wave a launch below may use
		const uint32_t maxWork = (nC + S - 1) / S + 1;
		// (the history kernels are built for 3 waves per SIMD and carry 1.8 MB of item scratch per lane group: 12 persistent blocks per CU)
		const uint32_t persistBlocks = I.histStates() ? I.persistBlocks / 12 * I.histBlocksPerCu() : I.persistBlocks;
		const uint32_t maxBlocks = std::min(persistBlocks, (maxWork + nGroups - 1) / nGroups);
		const size_t groupScratchBytes = I.hasSbg ? sizeof(GroupScratchT<BIGQ_SBG>) : I.hasCongG ? sizeof(GroupScratchCong<BIGQ_SBG>) : I.hasCong ? sizeof(GroupScratchCong<BIGQ>) : sizeof(GroupScratch);
		I.bigScratch.ensure((size_t)maxBlocks * nGroups * groupScratchBytes * std::min(S, 2u));
		b.wv.bigScratch = I.bigScratch.as<uint8_t>(); b.wv.bigScratchBytes = (uint32_t)groupScratchBytes;
		if (I.histStates())
		{
			// the kernel's key hash tables (SbgScratch::table) are handed over all-zero and left all-zero by every batch
			const void* before = I.sbgScratch.p;
			I.sbgScratch.ensure((size_t)maxBlocks * nGroups * sizeof(SbgScratch) * std::min(S, 2u));
			if (I.sbgScratch.p != before) HIPCHECK(hipMemsetAsync(I.sbgScratch.p, 0, I.sbgScratch.cap, sA));
		}
		const uint32_t ldsBytes = searchKernelLdsBytes(I.groupLanes);
		if (!b.typo.typo)
		{
			// the lattice kernel's LDS size classes: the work order (longest chunk first = largest LDS need first) of every sub-batch cut into classes of
			// <= 25 % unused LDS, one launch each.  Made before anything is enqueued (between two launches it left the device idle for 0.45 ms at 65 536
			// chunks) and kept with the batch while the match ratio stays what it was
			const bool wave = I.latticeWave && I.latticeGroupForced == 0;
			const uint32_t ratioKey = wave ? (I.latticeRatio16 & 0x3FFFu) : 0x10000u;
			// (a batch with pretokenized spans: every chunk to the replay of the reference's splitter, k_build_lattice_big -- the only lattice kernel that reads spans)
			const uint32_t budget = b.pretok ? 0u : wave ? I.latticeWaveBudget : I.latticeLdsBudget;
			const uint32_t classesKey = ratioKey * 31u + budget / 16u;
			if (b.latClassesKey != classesKey || b.latClasses.size() != S)
			{
				// (the LDS need of a chunk is a function of its length -- its HBM capacities are -- : worked out once per length, not per chunk; a batch of
				// 65 536 chunks asked this 200 000 times on the calling thread, 7 ms of a 29 ms end-to-end batch)
				std::vector<uint32_t> needByLen;
				auto needOf = [&](uint32_t c)
				{
					const uint32_t nCh = b.charOff[c + 1] - b.charOff[c], nodeCap = b.nodeBase[c + 1] - b.nodeBase[c], matchCap = b.matchBase[c + 1] - b.matchBase[c];
					if (nCh >= needByLen.size()) needByLen.resize(nCh + 1, 0u);
					if (!needByLen[nCh]) needByLen[nCh] = wave ? latticeWaveLayout(nCh, nodeCap, matchCap, ratioKey).total : latticeLdsLayout(nCh, nodeCap, matchCap).total;
					return needByLen[nCh];
				};
				b.latClasses.assign(S, {});
				for (uint32_t k = 0; k < S; ++k)
				{
					const uint32_t c0 = (uint32_t)((uint64_t)nC * k / S), c1 = (uint32_t)((uint64_t)nC * (k + 1) / S);
					uint32_t i = c0;
					while (i < c1 && needOf(b.order[i]) > budget) ++i;
					while (i < c1)
					{
						const uint32_t need = needOf(b.order[i]);
						uint32_t j = i + 1;
						while (j < c1 && (uint64_t)needOf(b.order[j]) * 4 >= (uint64_t)need * 3) ++j;      // <= 25 % of a class's LDS unused
						b.latClasses[k].push_back({ i, j, need, 0u });
						i = j;
					}
					// which of the four streams a class is launched on: largest first onto the least loaded (cost ~ chunks x LDS bytes: the classes are LDS-occupancy bound)
					std::vector<size_t> byCost(b.latClasses[k].size());
					std::iota(byCost.begin(), byCost.end(), (size_t)0);
					auto cost = [&](size_t t) { const auto& c = b.latClasses[k][t]; return (uint64_t)(c.j - c.i) * c.need; };
					std::stable_sort(byCost.begin(), byCost.end(), [&](size_t x, size_t y) { return cost(x) > cost(y); });
					uint64_t load[4] = { 0, 0, 0, 0 };
					for (size_t t : byCost) { const uint32_t st = (uint32_t)(std::min_element(load, load + 4) - load); b.latClasses[k][t].stream = st; load[st] += cost(t); }
				}
				b.latClassesKey = classesKey;
			}
		}
		tm.lap("work order, size classes, scratch");
		for (uint32_t k = 0; k < S; ++k)
		{
			const uint32_t c0 = (uint32_t)((uint64_t)nC * k / S), c1 = (uint32_t)((uint64_t)nC * (k + 1) / S), cn = c1 - c0;
			hipEvent_t* e = &I.evs[6 * (size_t)k];
			HIPCHECK(hipEventRecord(e[0], sA));
			if (b.typo.typo)
			{
				// typo correction: the lattice of every chunk over its typo graph (thread per chunk; the dictionary scan happens inside, per search state)
				TypoLatView tv = b.tv;
				tv.lengtheningCost = b.typo.typo->lengtheningCost();
				tv.threshold = b.typo.threshold; tv.maxUnk = sp.maxUnk; tv.maxUnkJ = sp.maxUnkJ; tv.spaceTol = sp.spaceTol; tv.match = sp.match;
				HIPCHECK(hipEventRecord(e[1], sA));
				// wave-per-chunk kernel with the chunk's working set in LDS, one launch per LDS size class (as for the plain lattice below); what
				// is over the budget or outgrows its LDS copy is left to the thread-per-chunk kernel (returns at once otherwise)
				{
					uint32_t i = c0;
					while (i < c1 && b.typoNeed[b.typoOrder[i]] > I.latticeLdsBudget) ++i;
					while (i < c1)
					{
						const uint32_t need = b.typoNeed[b.typoOrder[i]];
						uint32_t j = i + 1;
						while (j < c1 && (uint64_t)b.typoNeed[b.typoOrder[j]] * 4 >= (uint64_t)need * 3) ++j;
						launchTypoLatticeLds(I.dview, tv, b.dTypoOrder.as<uint32_t>() + i, j - i, need, sA);
						i = j;
					}
				}
				tv.chunks += c0;
				launchTypoLatticeRest(I.dview, tv, cn, I.latticeLdsBudget, sA);
			}
			else
			{
			hipLaunchKernelGGL(k_dict_scan, dim3((cn + 3) / 4), dim3(256), 0, sA, I.dview, b.bv, b.wv, c0, cn);
			HIPCHECK(hipEventRecord(e[1], sA));
			// wave-per-chunk build with the chunk's working set in LDS.  The dynamic LDS size of a launch is uniform, so the
			// work order (longest chunk first = largest LDS need first) is cut into size classes, one launch each: a batch of
			// mixed lengths does not run at the occupancy its longest chunk allows.  Chunks beyond the budget, and chunks that
			// outgrow their LDS copy at run time, are picked up by the thread-per-chunk kernel (returns at once otherwise).
			{
				const bool wave = I.latticeWave && I.latticeGroupForced == 0;
				const uint32_t ratio16 = I.latticeRatio16 | (getenv("KAMD_LATTICE_STATS") ? 0x4000u : 0u);      // (bit 14: the kernel also counts developer statistics)
				// the candidate records and the position program written by the lattice kernel itself (no k_expand_cands / k_expand_pos pass over these chunks):
				// for the position-step search without a blocklist and without character-model scores of unknown forms (k_unk_chr sits between the two);
				// bit 1: a CoNgram model (records in the transposed evaluator's order).  KAMD_LATTICE_EXPAND=0: the two kernels do it
				static const bool fuseExpand = !(getenv("KAMD_LATTICE_EXPAND") && std::atoi(getenv("KAMD_LATTICE_EXPAND")) == 0);
				const bool posEarly = b.wv.posRecs && sp.topN == 1 && !I.groupLanesForced && S <= 8 && (!b.typo.typo || I.posPathTypo);
				const uint32_t expandMode = (fuseExpand && posEarly && !b.wv.unkChr && !b.wv.blockBits) ? (1u | (I.hasCong ? 2u : 0u)) : 0u;
				if (getenv("KAMD_LATTICE_PROFILE")) { lwProf.ensure((size_t)nC * 64); HIPCHECK(hipMemsetAsync(lwProf.p, 0, (size_t)nC * 64, sA)); b.wv.beacon = lwProf.as<uint32_t>(); }
				const uint32_t budget = b.pretok ? 0u : wave ? I.latticeWaveBudget : I.latticeLdsBudget;
				// the size classes of k_lattice_wave go over four streams (forked from and joined back into sA; which one: decided with the classes): a class ends when its slowest
				// wavefront does, and the next class's wavefronts fill the machine meanwhile
				uint32_t nClass = 0;
				// (the list of chunks for the wide launch starts empty for every sub-batch: entries of an earlier sub-batch, long built, would use its slots up -- ADVICE r04)
				if (wave && k) HIPCHECK(hipMemsetAsync(b.dOutCounters.as<uint32_t>() + 3, 0, 4, sA));
				if (wave) { HIPCHECK(hipEventRecord(I.latFork, sA)); for (auto ls : I.latStream) HIPCHECK(hipStreamWaitEvent(ls, I.latFork, 0)); }
				for (const auto& lc : b.latClasses[k])
				{
					const uint32_t i = lc.i, j = lc.j, need = lc.need;
					static const uint32_t dbgStop = std::getenv("KAMD_LATTICE_STOP") ? (uint32_t)std::atoi(std::getenv("KAMD_LATTICE_STOP")) : 0u;      // EXPERIMENT
					// one chunk per wavefront.  KAMD_LATTICE_GROUP=16 (EXPERIMENT) packs four: measured slower on the MI355X -- c2-64k 3.04 ms against
					// 1.96 ms, c2 0.57 against 0.42 ms (profiles/r03_o_*): the four replays diverge, and a block with four working sets leaves a
					// quarter of the wavefronts to hide their LDS chains
					const uint32_t need16 = (need + 15u) & ~15u;
					const bool four = I.latticeGroupForced == 16 && need16 * 4 <= 64 * 1024;
					hipStream_t sL = (wave && lc.stream) ? I.latStream[lc.stream - 1] : sA;
					++nClass;
					if (wave) hipLaunchKernelGGL(k_lattice_wave, dim3(j - i), dim3(64), need, sL, I.dview, b.bv, b.wv, sp, b.dOrder.as<uint32_t>() + i, j - i, need | (dbgStop << 24), ratio16, expandMode);
					else if (four) hipLaunchKernelGGL(k_build_lattice<16>, dim3((j - i + 3) / 4), dim3(64), need16 * 4, sA, I.dview, b.bv, b.wv, sp, b.dOrder.as<uint32_t>() + i, j - i, need16 | (dbgStop << 24));
					else hipLaunchKernelGGL(k_build_lattice<64>, dim3(j - i), dim3(64), need, sA, I.dview, b.bv, b.wv, sp, b.dOrder.as<uint32_t>() + i, j - i, need | (dbgStop << 24));
				}
				if (wave) for (int t = 0; t < 3; ++t) { HIPCHECK(hipEventRecord(I.latJoin[t], I.latStream[t])); HIPCHECK(hipStreamWaitEvent(sA, I.latJoin[t], 0)); }
				// what outgrew the first launch's LDS arrays: the same kernel with room for 3 matches and one other op per text unit, over the list the first launch left
				if (wave && budget) hipLaunchKernelGGL(k_lattice_wave, dim3(std::min(cn, 16384u)), dim3(64), budget, sA, I.dview, b.bv, b.wv, sp, b.dOrder.as<uint32_t>() + c0, std::min(cn, 16384u), budget, kLatticeWideRatio16 | kLatticeWideBit | (ratio16 & 0x4000u), expandMode);
				hipLaunchKernelGGL(k_build_lattice_big, dim3((cn + 63) / 64), dim3(64), 0, sA, I.dview, b.bv, b.wv, sp, c0, cn, budget, wave ? (ratio16 & 0x3FFFu) : 0u);
			}
			}
			hipLaunchKernelGGL(k_expand_cands, dim3(cn), dim3(64), 0, sA, I.dview, b.bv, b.wv, c0, cn, I.hasCong ? 1u : 0u, I.hasCongG ? I.congG.distMask : (const uint8_t*)nullptr);
			if (b.wv.unkChrForm)      // Match::oovChrFreqModel / oovChrFreqBranchModel: ... mixed with the substring frequencies of the text (chr_freq.hpp)
				hipLaunchKernelGGL(k_unk_chr_freq, dim3(cn), dim3(64), 0, sA, I.dview, b.bv, b.wv, I.chr, ChrFreqParams{ sp.oovGlobalWeight, sp.oovLocalWeight, sp.oovGlobalMinFreq }, sp.oovChrFreqBias,
					c0, cn, chrToken(0xD800, identifySpecialChr(0xD800)), chrToken(0xDC00, identifySpecialChr(0xDC00)));
			else if (b.wv.unkChr)      // Match::oovChrModel: every node's unknown form scored by the character model, once, before the search
				hipLaunchKernelGGL(k_unk_chr, dim3(cn), dim3(64), 0, sA, I.dview, b.bv, b.wv, I.chr, c0, cn, chrToken(0xD800, identifySpecialChr(0xD800)), chrToken(0xDC00, identifySpecialChr(0xDC00)));
			// position-step search first (viterbi_pos.inc): top-1, 16-lane groups, not for SkipBigram models; what it cannot finish is resumed by the general kernel below
			// (typo correction: until round 6 the lattices over typo graphs held positions the step kernel left to the general one -- nodes of equal TEXT end that feed
			// each other, the halves of a continual typo: 86 % of c5's chunks were handed over, search 3.97 ms with the position steps against 2.25 without,
			// profiles/r04_j_c5_pos_path.txt.  A step is now the nodes of equal MULTIPLIED end (DevNode::pad, k_expand_pos) and the typo compilations keep the
			// state ranges of 64 nodes in LDS: 16 of 8 192 chunks leave the steps, search 2.44 -> 1.95 ms, profiles/r06_rr_*; KAMD_POS_PATH=3 is the old behaviour)
			const bool usePos = b.wv.posRecs && sp.topN == 1 && !I.groupLanesForced && S <= 8 && (!b.typo.typo || I.posPathTypo);      // (per-sub-batch counters: 8 of each)
			if (usePos)
				hipLaunchKernelGGL(k_expand_pos, dim3(cn), dim3(64), 0, sA, I.dview, b.bv, b.wv, sp, c0, cn, b.typo.typo ? b.dNodeTypo.as<float>() : (const float*)nullptr, ((I.hasCong && b.wv.unkChr) ? 1u : 0u) | (I.hasCong ? 2u : 0u));      // (bit 0: unknown forms scored by the character model; bit 1: a CoNgram model)
			HIPCHECK(hipEventRecord(e[2], sA));
			HIPCHECK(hipStreamWaitEvent(sB, e[2], 0));
			HIPCHECK(hipEventRecord(e[3], sB));
			const uint32_t blocks = std::min(I.persistBlocks, (cn + nGroups - 1) / nGroups);   // (upper bound; used by the developer dumps)
			// consecutive searches may overlap at their tails: alternate between two scratch halves
			WorkView wv = b.wv;
			wv.beacon = nullptr;
			wv.posHandOver = usePos ? I.counter.as<uint32_t>() + 48 + k : nullptr;      // (zeroed with the work counters above)
			if (usePos && I.posContSlots)
			{
				// (consecutive sub-batches' searches may overlap at their tails: like bigScratch, the continuation slots alternate between two halves)
				I.posScratch.ensure((size_t)I.posContSlots * groupScratchBytes * std::min(S, 2u));
				wv.posScratch = I.posScratch.as<uint8_t>() + (size_t)(k & 1) * ((S > 1) ? (size_t)I.posContSlots * groupScratchBytes : 0);
				wv.posContCounter = I.counter.as<uint32_t>() + 56 + (k & 7); wv.posContSlots = I.posContSlots;
			}
#ifdef KAMD_TIMELINE
			static DevBuf tlBuf;
			tlBuf.ensure((size_t)nC * 128);
			HIPCHECK(hipMemsetAsync(tlBuf.p, 0, (size_t)nC * 128, sB));
			wv.beacon = tlBuf.as<uint32_t>(); gTimeline = tlBuf.p;
#endif
#ifndef KAMD_TIMELINE
			static const bool envPosBeacon = getenv("KAMD_POS_BEACON") != nullptr;      // (read once: this is the per-batch hot path)
			if (envPosBeacon)
			{
				posBeacon.ensure((size_t)nC * 256);
				HIPCHECK(hipMemsetAsync(posBeacon.p, 0, (size_t)nC * 256, sB));
				wv.beacon = posBeacon.as<uint32_t>();
			}
#endif
			wv.bigScratch = I.bigScratch.as<uint8_t>() + (size_t)(k & 1) * ((S > 1) ? (size_t)maxBlocks * nGroups * groupScratchBytes : 0);
			uint32_t* counter = I.counter.as<uint32_t>() + k;
			const uint32_t* order = b.dOrder.as<uint32_t>() + c0;
			// lane-group width / register budget: with few chunks the step is bound by the dependent chain of one chunk (16-lane
			// groups, 2 waves per SIMD measured best on 8192 x 40 jamo); with many chunks it is a throughput problem and narrower
			// groups + a third wave per SIMD win (65536 x 40 jamo: 6.9 vs 9.3 ms).  KAMD_GROUP_LANES / KAMD_WPS override.
			const bool many = cn >= 32768 && !usePos;      // (after the position-step kernel only a handful of chunks are left: what counts is one chunk's chain, not throughput)
			const int gl = (I.histStates() || b.typo.typo || I.hasCong) ? (variant64 ? 64 : 16) : I.groupLanesForced ? I.groupLanes : (many ? 8 : 16);
			const int wps = b.typo.typo ? 2 : I.wpsForced ? I.wpsForced : ((many && (gl == 8 || gl == 16)) ? 3 : 2);
			const uint32_t nGroupsK = 64u / (uint32_t)gl;
			const uint32_t blocksK = std::min(persistBlocks, (cn + nGroupsK - 1) / nGroupsK);
			// (the history compilations have their own LDS layout: smaller caches of the chunk, so that three waves per SIMD fit)
			const uint32_t ldsK = I.hasSbg ? (b.typo.typo ? typok::sbgk::histKernelLdsBytes(gl) : sbgk::histKernelLdsBytes(gl))
				: I.hasCongG ? (b.typo.typo ? typok::congk::gk::histKernelLdsBytes(gl) : congk::gk::histKernelLdsBytes(gl))
				: searchKernelLdsBytes(gl) + (b.typo.typo ? nGroupsK * kTypoRingExtra : 0u);      // (typok:: / typok::congk::: the state ranges of 64 nodes per lane group)
			if (usePos)
			{
				const bool wide = I.wpsForced ? I.wpsForced == 3 : cn >= 16384;      // many chunks: three waves per SIMD (what the kernel's LDS allows; a 168-VGPR build); few: the latency-bound regime (KAMD_WPS overrides)
				// lane-group width of the position steps: four chunks per wavefront (16-lane groups).  The eight-chunk build (8-lane groups, KAMD_POS_G=8) issues
				// 20 % fewer vector instructions per position -- the bookkeeping of a step serves twice the positions, the rounds of 64 item lanes are fuller --
				// and is NOT faster on the MI355X: c2-64k 3.24 against 3.19 ms, c2 0.96 against 0.63, c4-cong 20.9 against 17.7 (profiles/r05_d_*): a wavefront's
				// step takes as much longer as it serves more positions (2.7 rounds of items per step against 1.6, each a chain of dependent LDS / memory round
				// trips), and what bounds the kernel is that chain per resident wavefront, not the instruction issue
				const bool narrow = I.posGroupForced == 8;
				const uint32_t perBlock = narrow ? 8u : 4u;
				const uint32_t blocksP = (cn + perBlock - 1) / perBlock;      // one chunk per lane group, one wavefront per block, no persistent loop (viterbi_pos.inc)
				const float* nodeTypoP = b.typo.typo ? b.dNodeTypo.as<float>() : nullptr;
				const uint32_t posLds16 = kPosKernelLdsBytes + (b.typo.typo ? 4 * kTypoRingExtra : 0u), posLds8 = kPosKernelLdsBytes8 + (b.typo.typo ? 8 * kTypoRingExtra : 0u);
#define KAMD_POS_LAUNCH(NS, ...) { if (narrow && wide) hipLaunchKernelGGL((NS k_pos_path<8, 3>), dim3(blocksP), dim3(64), posLds8, sB, I.dview, b.bv, wv, sp, order, cn, ##__VA_ARGS__); \
				else if (narrow) hipLaunchKernelGGL((NS k_pos_path<8, 2>), dim3(blocksP), dim3(64), posLds8, sB, I.dview, b.bv, wv, sp, order, cn, ##__VA_ARGS__); \
				else if (wide) hipLaunchKernelGGL((NS k_pos_path<16, 3>), dim3(blocksP), dim3(64), posLds16, sB, I.dview, b.bv, wv, sp, order, cn, ##__VA_ARGS__); \
				else hipLaunchKernelGGL((NS k_pos_path<16, 2>), dim3(blocksP), dim3(64), posLds16, sB, I.dview, b.bv, wv, sp, order, cn, ##__VA_ARGS__); }
				if (I.hasCong && b.typo.typo) KAMD_POS_LAUNCH(typok::congk::, nodeTypoP, I.cong)
				else if (I.hasCong) KAMD_POS_LAUNCH(congk::, I.cong)
				else if (b.typo.typo) KAMD_POS_LAUNCH(typok::, nodeTypoP)
				else KAMD_POS_LAUNCH(kamd::)
#undef KAMD_POS_LAUNCH
			}
#define KAMD_LAUNCH(GG, WW) hipLaunchKernelGGL((k_best_path<GG, WW>), dim3(blocksK), dim3(64), ldsK, sB, I.dview, b.bv, wv, sp, counter, order, cn)
			if (I.hasSbg)
			{
				// SkipBigram model: the search kernel with history rings (16-lane groups, or one chunk per wave when 64 is forced)
				SbgDev sd = I.sbg;
				sd.hist = b.dHist.as<uint32_t>();
				sd.itemScratch = I.sbgScratch.as<uint8_t>() + (size_t)(k & 1) * ((S > 1) ? (size_t)maxBlocks * nGroups * sizeof(SbgScratch) : 0);
				if (b.typo.typo)
				{
					// ... over lattices with typo costs: both additions (viterbi_kernel_sbg_typo.hip)
					const float* nodeTypo = b.dNodeTypo.as<float>();
					if (gl == 64) hipLaunchKernelGGL((typok::sbgk::k_best_path<64, 2>), dim3(blocksK), dim3(64), ldsK, sB, I.dview, b.bv, wv, sp, counter, order, cn, sd, nodeTypo);
					else hipLaunchKernelGGL((typok::sbgk::k_best_path<16, 2>), dim3(blocksK), dim3(64), ldsK, sB, I.dview, b.bv, wv, sp, counter, order, cn, sd, nodeTypo);
				}
				else if (gl == 64) hipLaunchKernelGGL((sbgk::k_best_path<64, 2>), dim3(blocksK), dim3(64), ldsK, sB, I.dview, b.bv, wv, sp, counter, order, cn, sd);
				else hipLaunchKernelGGL((sbgk::k_best_path<16, 2>), dim3(blocksK), dim3(64), ldsK, sB, I.dview, b.bv, wv, sp, counter, order, cn, sd);
			}
			else if (I.hasCongG)
			{
				// global CoNgram model (viterbi_kernel_congg.hip / _congg_typo.hip): CoNgram scoring with distant tokens + history words beside the states
				CongGDev gd = I.congG;
				gd.hist = b.dHist.as<uint32_t>();
				gd.itemScratch = I.sbgScratch.as<uint8_t>() + (size_t)(k & 1) * ((S > 1) ? (size_t)maxBlocks * nGroups * sizeof(SbgScratch) : 0);
				const uint32_t ldsC = ldsK + (64u / (uint32_t)gl) * 4u * QCAP;
				if (b.typo.typo)
				{
					const float* nodeTypo = b.dNodeTypo.as<float>();
					if (gl == 64) hipLaunchKernelGGL((typok::congk::gk::k_best_path<64, 2>), dim3(blocksK), dim3(64), ldsC, sB, I.dview, b.bv, wv, sp, counter, order, cn, nodeTypo, I.cong, gd);
					else hipLaunchKernelGGL((typok::congk::gk::k_best_path<16, 2>), dim3(blocksK), dim3(64), ldsC, sB, I.dview, b.bv, wv, sp, counter, order, cn, nodeTypo, I.cong, gd);
				}
				else if (gl == 64) hipLaunchKernelGGL((congk::gk::k_best_path<64, 2>), dim3(blocksK), dim3(64), ldsC, sB, I.dview, b.bv, wv, sp, counter, order, cn, I.cong, gd);
				else hipLaunchKernelGGL((congk::gk::k_best_path<16, 2>), dim3(blocksK), dim3(64), ldsC, sB, I.dview, b.bv, wv, sp, counter, order, cn, I.cong, gd);
			}
			else if (I.hasCong && b.typo.typo)
			{
				// typo correction with a CoNgram model: both additions (viterbi_kernel_cong_typo.hip)
				const float* nodeTypo = b.dNodeTypo.as<float>();
				const uint32_t ldsC = ldsK + (64u / (uint32_t)gl) * 4u * QCAP;
				if (gl == 64) hipLaunchKernelGGL((typok::congk::k_best_path<64, 2>), dim3(blocksK), dim3(64), ldsC, sB, I.dview, b.bv, wv, sp, counter, order, cn, nodeTypo, I.cong);
				else hipLaunchKernelGGL((typok::congk::k_best_path<16, 2>), dim3(blocksK), dim3(64), ldsC, sB, I.dview, b.bv, wv, sp, counter, order, cn, nodeTypo, I.cong);
			}
			else if (I.hasCong)
			{
				// CoNgram model: the search kernel compiled with KAMD_CONG (16-lane groups, or one chunk per wave when 64 is forced); its LDS
				// slices also stage the context id of every work item
				const uint32_t ldsC = ldsK + (64u / (uint32_t)gl) * 4u * QCAP;
				if (gl == 64) hipLaunchKernelGGL((congk::k_best_path<64, 2>), dim3(blocksK), dim3(64), ldsC, sB, I.dview, b.bv, wv, sp, counter, order, cn, I.cong);
				else hipLaunchKernelGGL((congk::k_best_path<16, 2>), dim3(blocksK), dim3(64), ldsC, sB, I.dview, b.bv, wv, sp, counter, order, cn, I.cong);
			}
			else if (b.typo.typo)
			{
				// lattice nodes with typo costs: the search kernel compiled with KAMD_TYPO (16-lane groups, or one chunk per wave when 64 is forced)
				const float* nodeTypo = b.dNodeTypo.as<float>();
				if (gl == 64) hipLaunchKernelGGL((typok::k_best_path<64, 2>), dim3(blocksK), dim3(64), ldsK, sB, I.dview, b.bv, wv, sp, counter, order, cn, nodeTypo);
				else hipLaunchKernelGGL((typok::k_best_path<16, 2>), dim3(blocksK), dim3(64), ldsK, sB, I.dview, b.bv, wv, sp, counter, order, cn, nodeTypo);
			}
			else if (wps == 3 && gl == 8) KAMD_LAUNCH(8, 3);
			else if (wps == 3 && gl == 16) KAMD_LAUNCH(16, 3);
			else switch (gl)
			{
			case 4: KAMD_LAUNCH(4, 2); break;
			case 8: KAMD_LAUNCH(8, 2); break;
			case 16: KAMD_LAUNCH(16, 2); break;
			case 32: KAMD_LAUNCH(32, 2); break;
			default: KAMD_LAUNCH(64, 2); break;
			}
#undef KAMD_LAUNCH
			HIPCHECK(hipEventRecord(e[4], sB));
			{
				static const uint32_t strideEnv = std::getenv("KAMD_FINISH_STRIDE") ? (uint32_t)std::atoi(std::getenv("KAMD_FINISH_STRIDE")) : 0u;      // EXPERIMENT
				const uint32_t stride = (strideEnv == 1 || strideEnv == 2 || strideEnv == 4 || strideEnv == 8 || strideEnv == 16 || strideEnv == 32 || strideEnv == 64) ? strideEnv : cn <= 32768 ? 16u : 4u, perWave = 64 / stride;      // active lanes per wave: 4 up to 32k chunks, 16 beyond
				if (!wv.slotCap)      // (slot mode: the search kernel has run every chunk's end stage itself)
				hipLaunchKernelGGL(k_finish_paths, dim3((cn + perWave - 1) / perWave), dim3(64), 0, sB, I.dview, b.bv, wv, sp, c0, cn, stride);
			}
			HIPCHECK(hipEventRecord(e[5], sB));
			if (getenv("KAMD_HANGDUMP"))
			{
				// developer aid: if the search kernel does not finish in 6 s, read back (on a third stream, while it is still
				// running) how far every chunk got -- results, per-node state counts -- print the stragglers and exit
				for (int ms = 0; ms < 6000 && hipEventQuery(e[4]) == hipErrorNotReady; ++ms) usleep(1000);
				if (hipEventQuery(e[4]) == hipErrorNotReady)
				{
					hipStream_t sC; HIPCHECK(hipStreamCreateWithFlags(&sC, hipStreamNonBlocking));
					std::vector<DevChunkResult> res(nC); std::vector<uint32_t> nn(nC), cnt(b.nodeBase[nC]), cntr(64);
					HIPCHECK(hipMemcpyAsync(res.data(), b.dResults.p, nC * sizeof(DevChunkResult), hipMemcpyDeviceToHost, sC));
					HIPCHECK(hipMemcpyAsync(nn.data(), b.dNNodes.p, nC * 4, hipMemcpyDeviceToHost, sC));
					HIPCHECK(hipMemcpyAsync(cnt.data(), b.dNodeStCnt.p, cnt.size() * 4, hipMemcpyDeviceToHost, sC));
					HIPCHECK(hipMemcpyAsync(cntr.data(), I.counter.p, 256, hipMemcpyDeviceToHost, sC));
					HIPCHECK(hipStreamSynchronize(sC));
					fprintf(stderr, "[hangdump] search kernel still running; chunks %u, work counter %u, blocks %u, groups/wave %u\n", nC, cntr[k], blocks, nGroups);
					uint32_t shown = 0, done = 0;
					for (uint32_t c = 0; c < nC; ++c)
					{
						uint32_t reached = 0;
						for (uint32_t j = 0; j < nn[c]; ++j) if (cnt[b.nodeBase[c] + j] == 0xFFFFFFFFu) break; else reached = j + 1;
						const bool fin = res[c].status != CS_OK || res[c].nPaths != 0;
						done += fin;
						if (!fin && shown < 24) { ++shown; fprintf(stderr, "  chunk %u text %u: status %u nPaths %u nEnd %u endOff %u nodes %u, node records written up to %u\n", c, (uint32_t)b.refs[c].text, res[c].status, res[c].nPaths, res[c].nEnd, res[c].endOff, nn[c], reached); }
					}
					fprintf(stderr, "[hangdump] finished chunks: %u / %u\n", done, nC);
					if (wv.beacon)
					{
						std::vector<uint32_t> bc((size_t)nC * 64);
						HIPCHECK(hipMemcpyAsync(bc.data(), wv.beacon, bc.size() * 4, hipMemcpyDeviceToHost, sC));
						HIPCHECK(hipStreamSynchronize(sC));
						uint32_t hist[16] = {}, shownB = 0;
						for (uint32_t c = 0; c < nC; ++c)
						{
							const uint32_t* q = &bc[(size_t)c * 64];
							++hist[q[0] & 15];
							if (q[0] != 13 && q[0] != 11 && q[0] != 0 && shownB < 12)
							{
								++shownB;
								fprintf(stderr, "  [pos beacon] chunk %u: stage %u args %u %u 0x%x beats %u\n    Q:", c, q[0], q[1], q[2], q[3], q[4]);
								for (int l = 0; l < 16; ++l) fprintf(stderr, " %u", q[16 + l]);
								fprintf(stderr, "\n    S:"); for (int l = 0; l < 16; ++l) fprintf(stderr, " %u", q[32 + l]);
								fprintf(stderr, "\n    k|local<<8:"); for (int l = 0; l < 16; ++l) fprintf(stderr, " 0x%x", q[48 + l]);
								fprintf(stderr, "\n");
							}
						}
						fprintf(stderr, "[hangdump] pos beacon stages:"); for (int l = 0; l < 16; ++l) fprintf(stderr, " %d:%u", l, hist[l]); fprintf(stderr, "\n");
					}
					fflush(stderr);
					_exit(7);
				}
			}
		}
		HIPCHECK(hipGetLastError());
		// the end of the batch on the device: stream B behind everything stream A was given
		HIPCHECK(hipEventRecord(I.joinEv, sA)); HIPCHECK(hipStreamWaitEvent(sB, I.joinEv, 0));
		if (!b.evDone) HIPCHECK(hipEventCreateWithFlags(&b.evDone, hipEventDisableTiming | hipEventBlockingSync));      // (blocking: a spinning wait was 10 ms of CPU per 65 536-sentence batch, counted against the host workers' quota)
		HIPCHECK(hipEventRecord(b.evDone, sB)); HIPCHECK(hipEventRecord(I.lastDone, sB));
		I.haveLast = true; b.launched = true; b.launchS = S;
		tm.lap("enqueueing the launches");
		if (!wait) return t;
		HIPCHECK(hipStreamSynchronize(sA));
		HIPCHECK(hipStreamSynchronize(sB));
		tm.lap("waiting for the kernels");
		afterLaunch(I, b, t, true);
		return t;
	}

	// what follows the kernels of a batch on the host: developer read-outs, the lattice kernel's LDS room for the next batch, the stage timings
	static void afterLaunch(Engine::Impl& I, StagedBatch& b, KernelTimes& t, bool timed)
	{
		const uint32_t nC = (uint32_t)b.refs.size();
		const uint32_t S = b.launchS;
		if (getenv("KAMD_POS_BEACON") && getenv("KAMD_POS_PHASES") && posBeacon.p)
		{
			// developer aid (KAMD_POS_DEBUG build): cycles per phase of a position step, averaged over the chunks' steps
			std::vector<uint32_t> bc((size_t)nC * 64);
			HIPCHECK(hipMemcpy(bc.data(), posBeacon.p, bc.size() * 4, hipMemcpyDeviceToHost));
			double acc[14] = {}, steps = 0;
			for (uint32_t c = 0; c < nC; ++c) { for (int k2 = 0; k2 < 14; ++k2) acc[k2] += bc[(size_t)c * 64 + 16 + k2]; steps += bc[(size_t)c * 64 + 31]; }
			static const char* names[14] = { "loop/prefetch", "nodes+records", "round header+mapping", "record decode", "scoring after the LM step (rules, key)", "arg-max rotations", "emit", "end of rounds", "prune+counters", "retries+flags", "commit", "end stage", "wait for parent state + record", "LM step" };
			double tot = 0; for (int k2 = 0; k2 < 14; ++k2) tot += acc[k2];
			fprintf(stderr, "[pos phases] cycles per step (group-lane-0 clock; %0.f steps):", steps);
			for (int k2 = 0; k2 < 14; ++k2) fprintf(stderr, " %s %.0f (%.0f%%);", names[k2], steps ? acc[k2] / steps : 0.0, tot ? 100.0 * acc[k2] / tot : 0.0);
			fprintf(stderr, " total %.0f\n", steps ? tot / steps : 0.0);
		}
		if (I.latticeWave && !I.latticeRatioForced && !b.typo.typo && nC >= 256 && b.capScale == 1 && !b.isRerun)
		{
			// k_lattice_wave's LDS room for the next batch: 1.25 x the most matches per text unit any chunk of this batch had (+ 1/8), more at once
			// when chunks had to go to the wide launch for lack of room; never below 3/4 nor above 3 per unit
			uint32_t c16[16] = {};
			HIPCHECK(hipMemcpyAsync(c16, b.dOutCounters.p, 64, hipMemcpyDeviceToHost, I.streamCopy)); HIPCHECK(hipStreamSynchronize(I.streamCopy));
			uint32_t want = (c16[13] * 16u * 5u / 4u + 99u) / 100u + 2u;
			if (c16[4] + c16[5] > nC / 256) want = std::max(want, I.latticeRatio16 * 3u / 2u);
			// (a decaying maximum, not the last value: one batch of plain text does not send the next dictionary-dense one to the wide and big fall-backs -- ADVICE r04)
			want = std::max(want, I.latticeRatio16 - std::min(I.latticeRatio16, (I.latticeRatio16 + 7u) / 8u));
			I.latticeRatio16 = std::min(kLatticeWideRatio16, std::max(12u, want));
		}
		if (getenv("KAMD_LATTICE_PROFILE") && lwProf.p)
		{
			// developer aid (make lwprof): cycles per phase of k_lattice_wave, per wavefront
			std::vector<uint32_t> rec((size_t)nC * 16);
			HIPCHECK(hipMemcpy(rec.data(), lwProf.p, rec.size() * 4, hipMemcpyDeviceToHost));
			double pr[13] = {}; uint32_t seen = 0;
			for (uint32_t c = 0; c < nC; ++c) { if (!rec[(size_t)c * 16]) continue; ++seen; for (int k2 = 0; k2 < 13; ++k2) pr[k2] += rec[(size_t)c * 16 + k2]; }
			static const char* names[13] = { "stage", "type pass", "match ops", "groups", "clear", "by-time", "by-start", "publish", "checks", "ranks", "sweep+prefix", "emit", "pack offsets" };
			double tot = 0; for (int k2 = 0; k2 < 13; ++k2) tot += pr[k2];
			fprintf(stderr, "[lattice profile] clock ticks per wavefront (%u chunks):", seen);
			for (int k2 = 0; k2 < 13; ++k2) fprintf(stderr, " %s %.0f (%.0f%%);", names[k2], seen ? pr[k2] / seen : 0.0, tot ? 100.0 * pr[k2] / tot : 0.0);
			fprintf(stderr, " total %.0f\n", seen ? tot / seen : 0.0);
		}
		if (getenv("KAMD_LATTICE_STATS"))
		{
			// developer aid: chunks k_lattice_wave built / handed over to the replay (by reason), fixpoint rounds per chunk
			uint32_t c16[16] = {};
			HIPCHECK(hipMemcpy(c16, b.dOutCounters.p, 64, hipMemcpyDeviceToHost));
			std::vector<uint8_t> ex(nC); std::vector<PosDesc> pd;
			HIPCHECK(hipMemcpy(ex.data(), b.dExpanded.p, nC, hipMemcpyDeviceToHost));
			uint32_t nEx = 0, nProg = 0;
			for (uint32_t c = 0; c < nC; ++c) if (ex[c]) { ++nEx; if (b.wv.posDesc) { PosDesc d; HIPCHECK(hipMemcpy(&d, b.wv.posDesc + b.nodeBase[c], sizeof(d), hipMemcpyDeviceToHost)); nProg += d.firstRec != 0; } }
			fprintf(stderr, "[lattice wave] records written by the lattice kernel: %u chunks (%u with a position program)\n", nEx, nProg);
			fprintf(stderr, "[lattice wave] chunks %u: built %u (%.2f rounds each); handed over: matches %u, ops %u, long span / rounds %u, no end node %u, long node %u, no start %u; ops per text unit: mean %.2f, max %.2f (matches: max %.2f)\n",
				(uint32_t)nC, c16[10], c16[10] ? (double)c16[11] / c16[10] : 0.0, c16[4], c16[5], c16[6], c16[7], c16[8], c16[9], c16[15] ? (double)c16[14] / c16[15] : 0.0, c16[12] / 100.0, c16[13] / 100.0);
		}
		if (b.wv.posRecs && getenv("KAMD_POS_STATS"))
		{
			// developer aid: how many chunks the position-step kernel handed over before the end node, how far into their lattices it got, and why
			std::vector<DevChunkResult> res(nC); std::vector<uint32_t> nn(nC);
			HIPCHECK(hipMemcpy(res.data(), b.dResults.p, nC * sizeof(DevChunkResult), hipMemcpyDeviceToHost));
			HIPCHECK(hipMemcpy(nn.data(), b.dNNodes.p, nC * 4, hipMemcpyDeviceToHost));
			double frac = 0; uint32_t early = 0, carried = 0, atStart = 0, seen = 0, why[16] = {};
			for (uint32_t c = 0; c < nC; ++c)
			{
				const uint32_t at = res[c].pad & 0xFFFFFFu;
				if (!at || !nn[c]) continue;
				++seen;
				if (at == kPosChunkDone && (res[c].pad >> 24)) { ++carried; ++why[(res[c].pad >> 24) & 15]; if (carried <= 8) fprintf(stderr, "[pos] carried on: chunk %u of %u, reason %u\n", c, (uint32_t)nC, (res[c].pad >> 24) & 15); continue; }
				if (at == kPosChunkDone || at + 1 >= nn[c]) continue;
				++early; frac += (double)at / nn[c]; atStart += at <= 1; ++why[(res[c].pad >> 24) & 15];
			}
			{ uint32_t c32[32]; HIPCHECK(hipMemcpy(c32, b.dOutCounters.p, 128, hipMemcpyDeviceToHost)); fprintf(stderr, "[pos] positions left to the general search by k_expand_pos: more than 16 nodes %u, more than 16 records %u, no record %u, node-level (feeds its step / nothing to evaluate / > 256 predecessors) %u\n", c32[20], c32[21], c32[22], c32[23]); }
			fprintf(stderr, "[pos] chunks %u (searched %u), carried on in the general search by the kernel itself %u, left to k_best_path %u (%.2f %%), of those from the start %u, mean hand-over point %.2f of the lattice; reasons: static %u, ring %u, container %u, record size %u, staging %u, retry %u, disconnected %u, no program %u, test cannot run %u\n",
				nC, seen, carried, early, seen ? 100.0 * early / seen : 0.0, atStart, early ? frac / early : 0.0, why[1], why[2], why[3], why[4], why[5], why[6], why[7], why[8], why[9]);
		}
#ifdef KAMD_TIMELINE
		if (getenv("KAMD_TIMELINE_PRINT") && gTimeline)
		{
			// developer aid: per-chunk stamps of the search kernel (constant 100 MHz clock) -> where a chunk's time goes
			std::vector<unsigned long long> tl((size_t)nC * 16);
			HIPCHECK(hipMemcpy(tl.data(), gTimeline, tl.size() * 8, hipMemcpyDeviceToHost));
			unsigned long long t0 = ~0ull, tEnd = 0;
			for (uint32_t c = 0; c < nC; ++c) if (tl[16ull * c]) { t0 = std::min(t0, tl[16ull * c]); tEnd = std::max(tEnd, tl[16ull * c + 2]); }
			std::vector<double> start, nodes, fin, perNode;
			for (uint32_t c = 0; c < nC; ++c)
			{
				if (!tl[16ull * c] || !tl[16ull * c + 2]) continue;
				start.push_back((tl[16ull * c] - t0) * 0.01); nodes.push_back((tl[16ull * c + 1] - tl[16ull * c]) * 0.01); fin.push_back((tl[16ull * c + 2] - tl[16ull * c + 1]) * 0.01);
				perNode.push_back(nodes.back() / std::max<double>(1.0, (double)(uint32_t)tl[16ull * c + 3]));
			}
			auto stat = [](std::vector<double> v, const char* name)
			{
				if (v.empty()) return;
				std::sort(v.begin(), v.end());
				double sum = 0; for (double x : v) sum += x;
				fprintf(stderr, "[timeline] %-26s mean %9.1f  p50 %9.1f  p90 %9.1f  p99 %9.1f  max %9.1f  (us, n=%zu)\n", name, sum / v.size(), v[v.size() / 2], v[v.size() * 9 / 10], v[v.size() * 99 / 100], v.back(), v.size());
			};
			fprintf(stderr, "[timeline] first chunk start -> last chunk end: %.1f us\n", (tEnd - t0) * 0.01);
			static const char* phName[12] = { "ph0 node setup", "ph1 cand record + misc", "ph2 scoring (+Knlm)", "ph3 emission: write states", "ph4 prune", "ph5 bookkeeping", "ph6 passes/reach", "ph7 scoring: before the LM step", "ph8 classify (pack load)", "ph9 batch formation + key table build", "ph10 scoring: Knlm step", "ph11" };
			for (int k = 0; k < 10; ++k)
			{
				std::vector<double> v;
				for (uint32_t c = 0; c < nC; ++c) if (tl[16ull * c] && tl[16ull * c + 2]) v.push_back(tl[16ull * c + 4 + k] * 0.01 / std::max<double>(1.0, (double)(uint32_t)tl[16ull * c + 3]));
				stat(v, phName[k]);
			}
			{
				std::vector<double> mhz;
				for (uint32_t c = 0; c < nC; ++c) if (tl[16ull * c] && tl[16ull * c + 1] > tl[16ull * c]) mhz.push_back((double)tl[16ull * c + 15] / ((tl[16ull * c + 1] - tl[16ull * c]) * 0.01));
				stat(mhz, "shader clock (MHz)");
			}
			stat(start, "chunk start offset"); stat(nodes, "node loop"); stat(fin, "end-candidate stage"); stat(perNode, "node loop / node");
			{
				// the slowest chunks, phase by phase (a batch bound by its heaviest chunks: BASELINE config 3)
				std::vector<uint32_t> ord;
				for (uint32_t c = 0; c < nC; ++c) if (tl[16ull * c] && tl[16ull * c + 2]) ord.push_back(c);
				std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b2) { return tl[16ull * a + 2] - tl[16ull * a] > tl[16ull * b2 + 2] - tl[16ull * b2]; });
				for (size_t r = 0; r < ord.size() && r < 16; ++r)
				{
					const uint32_t c = ord[r];
					fprintf(stderr, "[timeline] slow chunk %u: start %.0f us, total %.0f us, nodes %u; phases (us):", c, (tl[16ull * c] - t0) * 0.01, (tl[16ull * c + 2] - tl[16ull * c]) * 0.01, (uint32_t)tl[16ull * c + 3]);
					for (int k = 0; k < 10; ++k) fprintf(stderr, " %d:%.0f", k, tl[16ull * c + 4 + k] * 0.01);
					fprintf(stderr, "\n");
				}
			}
		}
#endif
		for (uint32_t k = 0; timed && k < S; ++k)      // (the timing events are the engine's: only a batch that was waited for in launchAll reads them)
		{
			hipEvent_t* e = &I.evs[6 * (size_t)k];
			float a = 0, l = 0, r = 0, f = 0;
			HIPCHECK(hipEventElapsedTime(&a, e[0], e[1]));
			HIPCHECK(hipEventElapsedTime(&l, e[1], e[2]));
			HIPCHECK(hipEventElapsedTime(&r, e[3], e[4]));
			HIPCHECK(hipEventElapsedTime(&f, e[4], e[5]));
			t.scanMs += a; t.latticeMs += l; t.searchMs += r; t.finishMs += f;
		}
		t.searchLaunches = S;
		b.ran = true;
		if (b.typo.typo && std::getenv("KAMD_HOST_TIMING"))      // developer aid: how the typo lattices were built
		{
			std::vector<TypoLatChunk> tch(nC);
			HIPCHECK(hipMemcpy(tch.data(), b.dTypoChunks.p, nC * sizeof(TypoLatChunk), hipMemcpyDeviceToHost));
			uint64_t need = 0, big = 0, over = 0, nodes = 0, cap = 0; uint32_t maxNeed = 0;
			for (auto& c : tch) { need += c.ldsNeed; maxNeed = std::max(maxNeed, c.ldsNeed); over += c.ldsNeed > I.latticeLdsBudget; nodes += c.nOutFinal; cap += c.ldsCap; }
			for (auto& c : tch) big += c.pad & 1;
			{
				// nodes BUILT (unconnected ones included: what the LDS copy has to hold) per text unit, x 100
				std::vector<uint32_t> ratio, built;
				for (auto& c : tch) if (!(c.pad & 1) && c.nChars) { built.push_back(c.pad >> 1); ratio.push_back((c.pad >> 1) * 100 / c.nChars); }
				std::sort(ratio.begin(), ratio.end()); std::sort(built.begin(), built.end());
				auto q = [](const std::vector<uint32_t>& v, double f) { return v.empty() ? 0u : v[std::min(v.size() - 1, (size_t)(f * v.size()))]; };
				fprintf(stderr, "[host] typo lattices: nodes built p50 %u p99 %u p99.9 %u max %u; per text unit x 100: p50 %u p99 %u p99.9 %u max %u\n",
					q(built, 0.5), q(built, 0.99), q(built, 0.999), q(built, 1.0), q(ratio, 0.5), q(ratio, 0.99), q(ratio, 0.999), q(ratio, 1.0));
			}
			fprintf(stderr, "[host] typo lattices: LDS need avg %.0f max %u B, %llu chunks over the budget, %llu outgrew their LDS copy; connected nodes avg %.1f of LDS capacity avg %.1f\n",
				(double)need / nC, maxNeed, (unsigned long long)over, (unsigned long long)big, (double)nodes / nC, (double)cap / nC);
		}
	}

	// D2H of what the end stage produced: 32 B per chunk and the two counters first, then exactly the path headers and token
	// records that were written (pinned landing zone; the token arenas at capacity stay on the device).
	static void download(Engine::Impl& I, StagedBatch& b)
	{
		const size_t nC = b.refs.size();
		b.hResults = nullptr; b.hPaths = nullptr; b.hTokens = nullptr;
		if (!nC) return;
		const size_t oRes = 64, resBytes = (nC * sizeof(DevChunkResult) + 255) & ~(size_t)255;
		b.hOut.ensure(oRes + resBytes);
		uint8_t* H = b.hOut.as<uint8_t>();
		HIPCHECK(hipMemcpyAsync(H, b.dOutCounters.p, 8, hipMemcpyDeviceToHost, I.streamCopy));
		HIPCHECK(hipMemcpyAsync(H + 16, b.dOutCounters.as<uint8_t>() + 64, 8, hipMemcpyDeviceToHost, I.streamCopy));      // (states of the pool handed out)
		HIPCHECK(hipMemcpyAsync(H + oRes, b.dResults.p, nC * sizeof(DevChunkResult), hipMemcpyDeviceToHost, I.streamCopy));
		HIPCHECK(hipStreamSynchronize(I.streamCopy));
		b.poolUsed = *reinterpret_cast<const uint64_t*>(H + 16);
		const uint32_t nPaths = std::min(reinterpret_cast<const uint32_t*>(H)[0], b.outPathCap), nTok = std::min(reinterpret_cast<const uint32_t*>(H)[1], b.outTokCap);
		const size_t oTok = ((size_t)nPaths * sizeof(DevPathHeader) + 255) & ~(size_t)255;
		b.hOut2.ensure(oTok + (size_t)nTok * sizeof(DevToken) + 16);
		uint8_t* H2 = b.hOut2.as<uint8_t>();
		if (nPaths) HIPCHECK(hipMemcpyAsync(H2, b.dOutPaths.p, (size_t)nPaths * sizeof(DevPathHeader), hipMemcpyDeviceToHost, I.streamCopy));
		if (nTok) HIPCHECK(hipMemcpyAsync(H2 + oTok, b.dOutTokens.p, (size_t)nTok * sizeof(DevToken), hipMemcpyDeviceToHost, I.streamCopy));
		HIPCHECK(hipStreamSynchronize(I.streamCopy));
		b.hResults = reinterpret_cast<const DevChunkResult*>(H + oRes);
		b.hPaths = reinterpret_cast<const DevPathHeader*>(H2);
		b.hTokens = reinterpret_cast<const DevToken*>(H2 + oTok); b.hTokCount = nTok;
		b.outBytes = 8 + nC * sizeof(DevChunkResult) + (size_t)nPaths * sizeof(DevPathHeader) + (size_t)nTok * sizeof(DevToken);
	}

	// the tables the result assembly reads: the model's, or the batch's copy with its temporary entries behind them
	static const FlatModel& hostModelOf(const Engine::Impl& I, const StagedBatch& b) { return (b.pretok && b.pretok->hasTemps()) ? b.pretok->hostModel : I.model; }

	static void chunkPaths(std::vector<PathResult>& out, const FlatModel& m, const StagedBatch& b, size_t c)
	{
		const DevChunkResult& r = b.hResults[c];
		out.resize(r.nPaths);      // (not clear(): the paths' token vectors keep their capacity from text to text)
		const auto& ref = b.refs[c];
		const PreparedView& pt = b.prep[ref.text];
		const uint32_t so = pt.chunks[ref.chunk].startOffset;
		for (uint32_t p = 0; p < r.nPaths; ++p)
		{
			PathResult& pr = out[p];
			pr.path.clear();
			const DevPathHeader& ph = b.hPaths[r.pathOff + p];
			pr.score = ph.score; pr.prevState = ph.prevState; pr.curState = ph.curState;
			const DevToken* tk = b.hTokens + r.tokOff + ph.tokOff;
			pr.path.reserve(ph.nTokens);
			for (uint32_t k = 0; k < ph.nTokens; ++k)
			{
				PathTok t;
				t.morph = tk[k].morph; t.begin = tk[k].begin + so; t.end = tk[k].end + so; t.wordScore = tk[k].wordScore; t.typoCost = tk[k].typoCost;
				if (tk[k].ownKind == 2) t.str = m.formStr(tk[k].ownA);
				else if (tk[k].ownKind) t.str = pt.normSubstr(so + tk[k].ownA, tk[k].ownLen);
				pr.path.push_back(std::move(t));
			}
		}
		if (b.pretok && ref.text == 0)
		{
			// findPretokenizedGroupOfNode (src/Kiwi.cpp:949-969) + Kiwi.cpp:745-750: a token of a lattice node inside span i of the CHUNK reports i + 1 as its
			// typoFormId (no node straddles a span: the tokens of a node inside one lie inside it)
			const ChunkDesc& d = pt.chunks[ref.chunk];
			uint32_t idx = 0;
			for (const auto& sn : b.pretok->spans)
			{
				if (!(sn.begin >= d.startOffset && sn.begin < d.startOffset + d.nChars)) continue;
				++idx;
				for (auto& pr : out) for (auto& t : pr.path) if (t.begin >= sn.begin && t.end <= sn.end && t.begin < t.end) t.typoFormId = idx;
			}
		}
		std::sort(out.begin(), out.end(), [](const PathResult& a, const PathResult& b2) { return a.score > b2.score; });   // PathEvaluator.hpp:1414-1417
	}

	std::shared_ptr<StagedBatch> Engine::stage(const std::vector<std::pair<const char16_t*, size_t>>& texts, uint64_t match, bool openEnding, int hostThreads, TypoOption typo)
	{
		return stagePretok(texts, match, openEnding, hostThreads, typo, nullptr);
	}

	std::shared_ptr<StagedBatch> Engine::stagePretok(const std::vector<std::pair<const char16_t*, size_t>>& texts, uint64_t match, bool openEnding, int hostThreads, TypoOption typo,
		std::shared_ptr<const PretokGroup> pretok)
	{
		if (pretok && pretok->spans.empty()) pretok.reset();
		// (the reference steps over a span in the typo graph as well, KTrie.cpp:882 -- not restated: there is no pin for it)
		if (pretok && typo.typo) throw std::invalid_argument{ "kiwi_amd: pretokenized spans together with a typo transformer are not supported" };
		std::vector<std::pair<uint32_t, uint32_t>> spanCut;      // the spans of text 0 in normalised offsets: the chunk cut and the pattern recognisers step over them
		if (pretok) for (const auto& sn : pretok->spans) spanCut.emplace_back(sn.begin, sn.end);
		if (typo.typo)
		{
			if (impl->model.forms.size() >= (1u << 24)) throw std::runtime_error{ "kiwi_amd: typo correction supports up to 2^24 forms" };
		}
		if ((match >> 8) & 3)
		{
			// Match::oovMask (include/kiwi/PatternMatcher.h:20-24): 1 = unknown forms scored by the character model; 2 / 3 add substring frequencies
			if (!impl->chr.present()) throw std::invalid_argument{ "`oovChrModel` option is set but the character-level noun model is not loaded." };      // Kiwi.cpp:1032-1035
		}
		HostTimer tm{ "stage" };
		auto b = std::make_shared<StagedBatch>();
		b->match = match;
		b->hostThreads = hostThreads;
		b->prep.resize(texts.size()); b->rawOff.assign(texts.size() + 1, 0);
		b->prepBlocks.resize((texts.size() + StagedBatch::kPrepBlock - 1) / StagedBatch::kPrepBlock);
		for (size_t i = 0; i < texts.size(); ++i) b->rawOff[i + 1] = b->rawOff[i] + texts[i].second;
		b->rawFlat.resize(b->rawOff.back());
		HostPool::instance().run(texts.size(), StagedBatch::kPrepBlock, hostThreads, [&](size_t i0, size_t i1, int)
		{
			PrepBlock& blk = b->prepBlocks[i0 / StagedBatch::kPrepBlock];
			size_t units = 0;
			for (size_t i = i0; i < i1; ++i) units += texts[i].second;
			blk.norm.reserve(2 * units + 16); blk.position.reserve(units + (i1 - i0) + 16); blk.cls.reserve(2 * units + 16); blk.script.reserve(2 * units + 16);
			blk.chunks.reserve(2 * (i1 - i0)); blk.idx.reserve(i1 - i0);
			for (size_t i = i0; i < i1; ++i)
			{
				if (texts[i].second) std::memcpy(&b->rawFlat[b->rawOff[i]], texts[i].first, 2 * texts[i].second);
				if (i == 0 && !spanCut.empty()) blk.append(texts[i].first, texts[i].second, match, (uint32_t)i, spanCut.data(), spanCut.size());
				else blk.append(texts[i].first, texts[i].second, match, (uint32_t)i);
			}
			for (size_t i = i0; i < i1; ++i) b->prep[i] = blk.view(i - i0);
		});
		tm.lap("text preparation (workers)");
		// the chunk list: the non-empty chunks of the texts in text order -- counted per preparation block, placed by a running sum over the blocks, written on the
		// workers (one loop on the calling thread was 1.5 ms per 65 536 texts, and the caller's chain is what the pipelined parts wait for)
		{
			const size_t nBlk = b->prepBlocks.size();
			// (a block holds the texts ONE call of the preparation appended to it: kPrepBlock of them when the pool cut the batch, all of them when it ran on one thread)
			std::vector<size_t> blkAt(nBlk + 1, 0), textAt(nBlk + 1, 0);
			for (size_t k = 0; k < nBlk; ++k)
			{
				size_t live = 0;
				for (const ChunkDesc& d : b->prepBlocks[k].chunks) live += d.empty ? 0 : 1;
				blkAt[k + 1] = blkAt[k] + live;
				textAt[k + 1] = textAt[k] + b->prepBlocks[k].idx.size();
			}
			if (textAt[nBlk] != texts.size()) throw std::logic_error{ "kiwi_amd: prepared texts and blocks disagree" };
			b->refs.resize(blkAt[nBlk]);
			HostPool::instance().run(nBlk, 16, hostThreads, [&](size_t k0, size_t k1, int)
			{
				for (size_t k = k0; k < k1; ++k)
				{
					size_t at = blkAt[k];
					for (size_t i = textAt[k]; i < textAt[k + 1]; ++i)
					{
						const auto& pt = b->prep[i];
						size_t live = 0;
						for (size_t c = 0; c < pt.chunks.size(); ++c) live += pt.chunks[c].empty ? 0 : 1;
						for (size_t c = 0; c < pt.chunks.size(); ++c)
						{
							if (pt.chunks[c].empty) continue;
							if (pt.chunks[c].nChars > 0xFFF0) throw std::runtime_error{ "chunk longer than 65520 units" };
							b->refs[at++] = ChunkRef{ (uint32_t)i, (uint32_t)c, { 0 }, openEnding && pt.chunks[c].nextOffset == pt.norm.size(), live == 1 };
						}
					}
				}
			});
		}
		tm.lap("chunk list");
		std::lock_guard<std::recursive_mutex> devLock{ impl->deviceMu };
		HIPCHECK(hipSetDevice(impl->device));      // the device is bound per thread: callers come from any thread
		b->typo = typo;
		b->pretok = std::move(pretok);
		layoutAndUpload(*impl, *b, makeParams(config, match, 1, &b->typo));
		tm.lap("layout + device buffers + upload");
		return b;
	}

	static void runRefs(Engine& E, Engine::Impl& I, StagedBatch& parent, std::vector<ChunkRef> refs, uint32_t capScale, std::vector<std::vector<PathResult>>& out);

	// One pass of all kernels over the batch, then -- if any chunk outgrew its scratch regions -- those chunks again, together, with larger
	// capacities (the ladder of runRefs): after run() every chunk of the batch has been searched to the end.
	KernelTimes Engine::run(StagedBatch& b)
	{
		std::lock_guard<std::recursive_mutex> devLock{ impl->deviceMu };
		HIPCHECK(hipSetDevice(impl->device));
		HostTimer tm{ "run" };
		KernelTimes t = launchAll(*impl, b, makeParams(config, b.match, b.topN, &b.typo));
		tm.lap("work order + launches + kernels");
		rerunOverflows(b, t);
		return t;
	}

	// The same in two halves, for a caller that has host work to do while the kernels run: launch() returns when everything is enqueued, finish() waits for
	// the batch (not for batches launched after it) and searches again what outgrew its scratch regions.
	void Engine::launch(StagedBatch& b)
	{
		std::lock_guard<std::recursive_mutex> devLock{ impl->deviceMu };
		HIPCHECK(hipSetDevice(impl->device));
		launchAll(*impl, b, makeParams(config, b.match, b.topN, &b.typo), false);
	}
	void Engine::finish(StagedBatch& b)
	{
		HostTimer tm{ "finish" };
		HIPCHECK(hipSetDevice(impl->device));
		if (b.evDone) HIPCHECK(hipEventSynchronize(b.evDone));      // (not under the device lock: another thread may be staging / launching the next part)
		tm.lap("waiting for the batch's kernels");
		std::lock_guard<std::recursive_mutex> devLock{ impl->deviceMu };
		tm.lap("waiting for the device lock");
		KernelTimes t;
		afterLaunch(*impl, b, t, false);
		tm.lap("counters of the launch");
		rerunOverflows(b, t);
		tm.lap("chunk statuses, re-runs");
	}

	void Engine::rerunOverflows(StagedBatch& b, KernelTimes& t)
	{
		b.overIdx.clear(); b.overPaths.clear(); b.rerunChunks = 0; b.rerunMs = 0;
		const size_t nC = b.refs.size();
		uint32_t nOver = 0;
		if (nC) { HIPCHECK(hipMemcpyAsync(&nOver, b.dOutCounters.as<uint32_t>() + 2, 4, hipMemcpyDeviceToHost, impl->streamCopy)); HIPCHECK(hipStreamSynchronize(impl->streamCopy)); }      // (the batch's kernels are done; the copy stream never waits for another batch's)
		if (nOver)
		{
			const auto t0 = std::chrono::steady_clock::now();
			std::vector<DevChunkResult> res(nC);
			HIPCHECK(hipMemcpy(res.data(), b.dResults.p, nC * sizeof(DevChunkResult), hipMemcpyDeviceToHost));
			if (std::getenv("KAMD_HOST_TIMING"))   // developer aid: which overflow
			{
				std::map<uint32_t, uint32_t> hist;
				for (auto& r : res) hist[r.status]++;
				for (auto& h : hist) fprintf(stderr, "[host] chunk status %u: %u chunks\n", h.first, h.second);
			}
			b.overIdx.assign(nC, SIZE_MAX);
			std::vector<ChunkRef> over;
			for (size_t c = 0; c < nC; ++c) if (res[c].status >= 16) { b.overIdx[c] = over.size(); over.push_back(b.refs[c]); }
			b.rerunChunks = (uint32_t)over.size();
			if (!over.empty()) runRefs(*this, *impl, b, std::move(over), b.capScale * 4, b.overPaths);
			b.rerunMs = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
		}
		t.rerunChunks = b.rerunChunks; t.rerunMs = b.rerunMs;
	}
	size_t Engine::stagedChunks(const StagedBatch& b) { return b.refs.size(); }
	uint64_t Engine::stagedUnits(const StagedBatch& b) { return b.units; }
	uint64_t Engine::stagedDeviceBytes(const StagedBatch& b) { return b.devBytes; }
	void Engine::stagedPool(const StagedBatch& b, uint64_t* out3) { out3[0] = b.ownStates; out3[1] = b.poolStates; out3[2] = b.poolUsed; }

	// Runs an explicit list of chunks (re-runs with larger capacities or non-default special states).  Chunks that overflow again go
	// up the capacity ladder TOGETHER (one launch per rung, not one per chunk); a rung is cut into slices of bounded device memory.
	static void runRefs(Engine& E, Engine::Impl& I, StagedBatch& parent, std::vector<ChunkRef> refs, uint32_t capScale,
		std::vector<std::vector<PathResult>>& out)
	{
		out.clear(); out.resize(refs.size());
		const SearchParams sp = makeParams(E.config, parent.match, parent.topN, &parent.typo);
		// ~ (48 n + 256) states of ~ 100 B per chunk and capacity step: 12 GB per slice
		const uint64_t sliceBudget = 120000000ull;
		size_t r0 = 0;
		while (r0 < refs.size())
		{
			uint64_t w = 0; size_t r1 = r0;
			while (r1 < refs.size())
			{
				const uint64_t cw = (48ull * parent.prep[refs[r1].text].chunks[refs[r1].chunk].nChars + 256) * capScale * (I.histStates() ? 8 : 1);
				if (r1 > r0 && w + cw > sliceBudget) break;
				w += cw; ++r1;
			}
			StagedBatch b;
			b.match = parent.match; b.capScale = capScale; b.topN = parent.topN; b.typo = parent.typo; b.hostThreads = 1; b.isRerun = true; b.pretok = parent.pretok;
			b.refs.assign(refs.begin() + r0, refs.begin() + r1);
			std::vector<size_t> failing;
			b.prep.swap(parent.prep);   // borrow the prepared texts for the duration of the launch
			try
			{
				layoutAndUpload(I, b, sp);
				launchAll(I, b, sp);
				download(I, b);
				for (size_t c = 0; c < b.refs.size(); ++c)
				{
					if (b.hResults[c].status >= 16)
					{
						if (capScale >= 64) throw std::runtime_error{ "analyze: device scratch overflow (status " + std::to_string(b.hResults[c].status) + ") even at 64x capacity" };
						failing.push_back(c);
					}
					else chunkPaths(out[r0 + c], hostModelOf(I, b), b, c);
				}
			}
			catch (...) { b.prep.swap(parent.prep); throw; }
			b.prep.swap(parent.prep);
			if (!failing.empty())
			{
				std::vector<ChunkRef> again; again.reserve(failing.size());
				for (size_t c : failing) again.push_back(b.refs[c]);
				std::vector<std::vector<PathResult>> sub;
				runRefs(E, I, parent, std::move(again), capScale * 4, sub);
				for (size_t k = 0; k < failing.size(); ++k) out[r0 + failing[k]] = std::move(sub[k]);
			}
			r0 = r1;
		}
	}

	uint32_t Engine::rerunChunks(const StagedBatch& b, float* ms) { if (ms) *ms = b.rerunMs; return b.rerunChunks; }

	BatchResults Engine::fetch(StagedBatch& b, size_t topN)
	{
		if (topN < 1 || topN > kMaxTopN) throw std::invalid_argument{ "kiwi_amd: top_n must be 1.." + std::to_string(kMaxTopN) + " on the device path" };
		// (the device -- and the engine's adaptive sizes below -- are held for the download only: the assembly of the results is host work, and a caller
		// that pipelines batches, analyzeBatch below, stages and launches the next part meanwhile; the lock is taken again for chunks that must be searched again)
		std::unique_lock<std::recursive_mutex> devLock{ impl->deviceMu };
		HIPCHECK(hipSetDevice(impl->device));
		if (!b.ran || b.topN != (uint32_t)topN) { b.topN = (uint32_t)topN; run(b); }
		HostTimer tm{ "fetch" };
		download(*impl, b);
		tm.lap("download");
		if (b.capScale == 1 && b.refs.size() >= 64 && !std::getenv("KAMD_TEST_TINY_ARENAS") && !b.slotCap)
		{
			// how much of the worst-case state capacity the chunks of this batch used (states + the end candidates and the back-trace chain behind them), in 64ths:
			// the next batches' arenas hold what 90 % of these chunks needed -- the others grow into the pool, which gets twice what this batch took of it
			// (nothing grew? then the pool keeps a sixteenth of the arenas' total).  A batch that still had chunks re-run doubles both.
			uint32_t hist[66] = {}; size_t counted = 0;
			for (size_t c = 0; c < b.refs.size(); ++c)
			{
				const DevChunkResult& r = b.hResults[c];
				if (r.status >= 16) continue;
				const uint64_t full = (48ull * (b.charOff[c + 1] - b.charOff[c]) + 256) * (impl->histStates() ? 8 : 1);
				const uint64_t used = (uint64_t)r.endOff + r.nEnd / 2 + (b.nodeBase[c + 1] - b.nodeBase[c]) / 12 + 16;
				++hist[std::min<uint64_t>(65, (used * 64 + full - 1) / full)]; ++counted;
			}
			uint64_t need64 = 1; size_t seen = 0;
			// (models without histories in their states: the 99.9th percentile, doubled -- their arenas are small, and a chunk that fills its arena under the
			// position-step kernel is handed to the slower general kernel, the one that can grow)
			const bool lean = impl->histStates() && b.poolStates;
			for (uint32_t k = 0; k < 66; ++k) { seen += hist[k]; need64 = std::max<uint64_t>(1, k); if (lean ? seen * 10 >= counted * 9 : seen * 1000 >= counted * 999) break; }
			const int which = b.topN > 1 ? 1 : 0;
			const uint64_t arenas = b.stateBase[b.refs.size()];
			if (!impl->stateScaleForced)
				impl->stateScale64[which] = b.rerunChunks ? std::min(64u, std::max(impl->stateScale64[which], 8u) * 2) : (uint32_t)std::min<uint64_t>(64, lean ? need64 : std::max<uint64_t>(8, need64 * 2));
			if (!impl->poolForced && arenas)
				impl->poolFrac64 = b.rerunChunks ? std::min(4096u, impl->poolFrac64 * 2) : (uint32_t)std::min<uint64_t>(4096, std::max<uint64_t>(4, (b.poolUsed * 2 * 64 + arenas - 1) / arenas));
			if (std::getenv("KAMD_LATTICE_STATS"))
			{
				fprintf(stderr, "[state arenas] top-%u batch of %zu chunks: 90 %% (models without state histories: 99.9 %%) of them used <= %llu/64 of the worst-case capacity, pool %llu of %llu states (arenas %llu), %u re-run -> scale %u/64 (top-1) %u/64 (top-N), pool %u/64; chunks by 64ths used:",
					b.topN, b.refs.size(), (unsigned long long)need64, (unsigned long long)b.poolUsed, (unsigned long long)b.poolStates, (unsigned long long)arenas, b.rerunChunks, impl->stateScale64[0], impl->stateScale64[1], impl->poolFrac64);
				for (uint32_t k = 0; k < 66; ++k) if (hist[k]) fprintf(stderr, " %u:%u", k, hist[k]);
				fprintf(stderr, "\n");
			}
		}
		const bool fastAssembly = FastAssembly::applies(topN, b.match, (bool)b.pretok);
		if (fastAssembly && !impl->tokTmpl.built) impl->tokTmpl.build(impl->model);
		devLock.unlock();
		const size_t nT = b.prep.size();
		BatchResults ret;
		ret.nTexts = nT; ret.d2hBytes = b.outBytes;
		ret.segs.resize((nT + BatchResults::kSegTexts - 1) / BatchResults::kSegTexts);
		// chunk index of each text inside refs
		std::vector<size_t> firstRef(nT + 1, 0);
		for (auto& r : b.refs) firstRef[r.text + 1]++;
		for (size_t i = 0; i < nT; ++i) firstRef[i + 1] += firstRef[i];
		// (chunks whose scratch overflowed were searched again inside run(): b.overIdx / b.overPaths)
		const std::vector<size_t>& overIdx = b.overIdx; const std::vector<std::vector<PathResult>>& overPaths = b.overPaths;
		// texts are independent: post-process them on the host workers, one segment of consecutive texts per task; a text whose chunk
		// must be searched again (other start states than the speculative {0}, or a scratch overflow) needs the device and is finished
		// afterwards, one by one
		static const std::vector<uint8_t> kOnlyZero{ 0 };
		// developer aid (KAMD_HOST_TIMING=1): the assembly's time by stage, summed over the workers
		std::atomic<uint64_t> stageNs[5] = {};
		const bool stageTiming = tm.on;
		auto nowNs = [] { return (uint64_t)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
		auto doText = [&](size_t i, bool mayRerun, ResultSegment& seg, std::vector<PathResult>& paths, ResultBuilder& rb, FastAssembly& fa) -> bool
		{
			uint64_t tLast = stageTiming ? nowNs() : 0;
			auto lapNs = [&](int k) { if (stageTiming) { const uint64_t t = nowNs(); stageNs[k] += t - tLast; tLast = t; } };
			const char16_t* raw = b.rawFlat.data() + b.rawOff[i]; const size_t rawLen = (size_t)(b.rawOff[i + 1] - b.rawOff[i]);
			if (fastAssembly && firstRef[i + 1] - firstRef[i] == 1)
			{
				// one chunk searched from state 0 that came back with one path: packed records straight from the device's token records (post_fast.hpp)
				const size_t c = firstRef[i];
				const DevChunkResult& r = b.hResults[c];
				if (r.status == CS_OK && r.nPaths == 1 && b.refs[c].sp.size() == 1 && b.refs[c].sp.data()[0] == 0)
				{
					const DevPathHeader& ph = b.hPaths[r.pathOff];
					const PreparedView& pt = b.prep[i];
					fa.text(impl->model, impl->tokTmpl, b.match, config.integrateAllomorph, raw, rawLen, pt, pt.chunks[b.refs[c].chunk].startOffset,
						b.hTokens + r.tokOff + ph.tokOff, ph.nTokens, b.hTokens + b.hTokCount, ph.score, seg);
					lapNs(4);
					return true;
				}
			}
			rb.begin(raw, rawLen, b.prep[i].position.data(), b.prep[i].position.size());
			lapNs(0);
			std::vector<uint8_t> uniqBuf;
			for (size_t c = firstRef[i]; c < firstRef[i + 1]; ++c)
			{
				// special states actually carried into this chunk (Kiwi.cpp:1122-1140) vs. the ones it was searched with
				// (a text's first chunk, and every chunk behind analyses that all ended in state 0: the set {0}, no vector built)
				const std::vector<uint8_t>& carried = rb.spStates();
				bool onlyZero = true;
				for (uint8_t v : carried) if (v) { onlyZero = false; break; }
				if (!onlyZero)
				{
					uniqBuf = carried;
					std::sort(uniqBuf.begin(), uniqBuf.end());
					uniqBuf.erase(std::unique(uniqBuf.begin(), uniqBuf.end()), uniqBuf.end());
				}
				const std::vector<uint8_t>& uniq = onlyZero ? kOnlyZero : uniqBuf;
				const uint32_t st = b.hResults[c].status;
				if (st >= 16 && b.refs[c].sp == uniq && c < overIdx.size() && overIdx[c] != SIZE_MAX)
				{
					if (!overPaths[overIdx[c]].empty()) rb.insertPaths(overPaths[overIdx[c]]);
					continue;
				}
				if (st >= 16 || b.refs[c].sp != uniq)
				{
					if (!mayRerun) return false;
					std::vector<std::vector<PathResult>> one;
					ChunkRef r = b.refs[c]; r.sp = uniq;
					runRefs(*this, *impl, b, { r }, st >= 16 ? b.capScale * 4 : b.capScale, one);
					if (!one[0].empty()) rb.insertPaths(one[0]);
					continue;
				}
				if (st == CS_NO_LATTICE) continue;
				chunkPaths(paths, hostModelOf(*impl, b), b, c);
				lapNs(1);
				rb.insertPaths(paths);
				lapNs(2);
			}
			if (b.pretok && i == 0 && b.pretok->hasTemps())
			{
				// token.morph = nullptr for the span group's own morphemes (src/Kiwi.cpp:733): they do not outlive the call
				auto res = rb.finish(raw, rawLen);
				const int32_t nOwn = (int32_t)b.pretok->overlay.nBaseMorphs;
				for (auto& r : res) for (auto& tk : r.first) if (tk.morph >= nOwn) tk.morph = -1;
				seg.appendText(res);
			}
			else
			{
				auto res = rb.finish(raw, rawLen);
				lapNs(3);
				seg.appendText(res);
				lapNs(4);
			}
			return true;
		};
		std::vector<uint8_t> again(nT, 0);
		const int postThreads = std::getenv("KAMD_POST_THREADS") ? std::atoi(std::getenv("KAMD_POST_THREADS")) : b.hostThreads;
		HostPool::instance().run(ret.segs.size(), 1, postThreads, [&](size_t s0, size_t s1, int)
		{
			std::vector<PathResult> paths;
			FastAssembly fa;
			ResultBuilder rb{ hostModelOf(*impl, b), topN, b.match, config.integrateAllomorph };      // (one builder per task: begin() starts a text)
			for (size_t sIdx = s0; sIdx < s1; ++sIdx)
			{
				ResultSegment& seg = ret.segs[sIdx];
				const size_t t0 = sIdx * BatchResults::kSegTexts, t1 = std::min(nT, t0 + BatchResults::kSegTexts);
				// (room for what the device reports for these texts' chunks -- 32 tokens per text reserved twice what c2 needs, and every reserved page is a fault)
				size_t nTokSeg = 0;
				for (size_t c = firstRef[t0]; c < firstRef[t1]; ++c) if (b.hResults[c].status < 16) nTokSeg += b.hResults[c].nTok;
				seg.toks.reserve(nTokSeg + 8); seg.forms.reserve(4 * nTokSeg + 64);
				seg.textAna.reserve(t1 - t0 + 1); seg.anaTok.reserve(t1 - t0 + 1); seg.anaScore.reserve(t1 - t0);
				for (size_t i = t0; i < t1; ++i)
					if (!doText(i, false, seg, paths, rb, fa)) { again[i] = 1; seg.appendText({}); }
			}
		});
		// (the nested run above must not be re-entered: runRefs uses the device and the pool from this thread only)
		std::vector<PathResult> paths;
		ResultBuilder rbAgain{ hostModelOf(*impl, b), topN, b.match, config.integrateAllomorph };
		FastAssembly faAgain;
		for (size_t i = 0; i < nT; ++i) if (again[i])
		{
			if (!devLock.owns_lock()) { devLock.lock(); HIPCHECK(hipSetDevice(impl->device)); }
			ret.overrides.emplace_back(i, ResultSegment{});
			doText(i, true, ret.overrides.back().second, paths, rbAgain, faAgain);
		}
		tm.lap("post-processing");
		if (stageTiming && nT) fprintf(stderr, "[host] fetch: per text, summed over the workers: begin %.0f ns, chunk paths %.0f, insert paths %.0f, finish %.0f, append %.0f\n",
			(double)stageNs[0] / nT, (double)stageNs[1] / nT, (double)stageNs[2] / nT, (double)stageNs[3] / nT, (double)stageNs[4] / nT);
		return ret;
	}

	BatchResults Engine::analyzeBatch(const std::vector<std::pair<const char16_t*, size_t>>& texts,
		size_t topN, uint64_t match, bool openEnding, int hostThreads, TypoOption typo, const PartSink* onPart)
	{
		if (topN < 1 || topN > kMaxTopN) throw std::invalid_argument{ "kiwi_amd: top_n must be 1.." + std::to_string(kMaxTopN) + " on the device path" };
		// A large batch goes through in PARTS whose host stages overlap the kernels of their neighbours: while part k is searched on the device the host
		// prepares and uploads part k + 1, and while part k + 1 is searched it downloads and assembles the results of part k (MI355X box with a 16-CPU
		// quota, 65 536 sentences: text preparation 2.8 + upload 1.3 + kernels 5.3 + download 0.5 + result assembly 3.3 + release 1.0 ms one after the
		// other = 14.2 ms; the kernels of the parts still run one part after the other).  Parts are cut at multiples of the result segment size, so the
		// parts' segments concatenate into the batch's.  KAMD_BATCH_PARTS fixes the number (1: one piece).
		const int forcedParts = [] { const char* e = std::getenv("KAMD_BATCH_PARTS"); return e ? std::atoi(e) : 0; }();
		const size_t seg = BatchResults::kSegTexts;
		size_t parts = forcedParts > 0 ? (size_t)forcedParts : std::min<size_t>(4, texts.size() / 16384);
		parts = std::max<size_t>(1, std::min(parts, (texts.size() + seg - 1) / seg));
		if (parts == 1)
		{
			auto b = stage(texts, match, openEnding, hostThreads, typo);
			b->topN = (uint32_t)topN;
			run(*b);
			BatchResults r = fetch(*b, topN);
			HostTimer tm{ "batch" };
			b.reset();
			tm.lap("release of the staged batch");
			if (onPart) { (*onPart)(0, std::move(r)); return BatchResults{}; }
			return r;
		}
		std::vector<size_t> cut(parts + 1, texts.size());
		for (size_t k = 0; k < parts; ++k) cut[k] = std::min(texts.size(), (texts.size() * k / parts + seg - 1) / seg * seg);
		std::vector<std::shared_ptr<StagedBatch>> staged(parts);
		BatchResults all;
		auto collect = [&](size_t k)
		{
			finish(*staged[k]);
			BatchResults r = fetch(*staged[k], topN);
			if (onPart) { staged[k].reset(); (*onPart)(cut[k], std::move(r)); return; }
			for (auto& sg : r.segs) all.segs.push_back(std::move(sg));
			for (auto& o : r.overrides) all.overrides.emplace_back(o.first + cut[k], std::move(o.second));
			all.nTexts += r.nTexts; all.d2hBytes += r.d2hBytes;
			HostTimer tmR{ "collect" };
			staged[k].reset();
			tmR.lap("release of the part");
		};
		// Two host threads: the caller prepares, uploads and launches part after part; a collector waits for each part's kernels, downloads, assembles
		// and releases it.  The stages that run on ONE thread (layout + upload 1.9 ms, work order + launches 0.5, download 0.5, release of the part's buffers
		// 1.4 per 65 536 sentences on the MI355X box) then overlap the other thread's pooled stages (text preparation 4.4 ms, result assembly 5.3) instead
		// of leaving the pool idle: profiles/r06_h_*.  KAMD_BATCH_PIPELINE=0: one thread, as in rounds 2 - 5.
		const bool pipelined = [] { const char* e = std::getenv("KAMD_BATCH_PIPELINE"); return !e || std::atoi(e) != 0; }();
		auto waitInFlight = [&]
		{
			// parts still in flight read their buffers: wait before anything is released
			for (auto& sb : staged) if (sb && sb->launched && sb->evDone) (void)hipEventSynchronize(sb->evDone);
		};
		if (!pipelined)
		{
			size_t collected = 0;
			try
			{
				for (size_t k = 0; k < parts; ++k)
				{
					std::vector<std::pair<const char16_t*, size_t>> part(texts.begin() + cut[k], texts.begin() + cut[k + 1]);
					staged[k] = stage(part, match, openEnding, hostThreads, typo);
					staged[k]->topN = (uint32_t)topN;
					launch(*staged[k]);
					if (k >= 1) { collect(collected); ++collected; }      // (part k - 1: its kernels ran while part k was prepared)
				}
				for (; collected < parts; ++collected) collect(collected);
			}
			catch (...) { waitInFlight(); throw; }
			return all;
		}
		std::mutex qmu; std::condition_variable qcv;
		size_t launchedParts = 0; bool stop = false;
		std::exception_ptr collectorError;
		std::thread collector{ [&]
		{
			try
			{
				for (size_t k = 0; k < parts; ++k)
				{
					{
						std::unique_lock<std::mutex> lk{ qmu };
						qcv.wait(lk, [&] { return launchedParts > k || stop; });
						if (launchedParts <= k) return;
					}
					collect(k);
				}
			}
			catch (...) { collectorError = std::current_exception(); }
		} };
		try
		{
			for (size_t k = 0; k < parts; ++k)
			{
				std::vector<std::pair<const char16_t*, size_t>> part(texts.begin() + cut[k], texts.begin() + cut[k + 1]);
				std::shared_ptr<StagedBatch> sb = stage(part, match, openEnding, hostThreads, typo);
				sb->topN = (uint32_t)topN;
				{ std::lock_guard<std::mutex> g{ qmu }; staged[k] = sb; }      // (from here on the part is waited for if anything throws)
				launch(*sb);
				{ std::lock_guard<std::mutex> g{ qmu }; ++launchedParts; }
				qcv.notify_one();
			}
		}
		catch (...)
		{
			{ std::lock_guard<std::mutex> g{ qmu }; stop = true; }
			qcv.notify_one();
			collector.join();
			waitInFlight();
			throw;
		}
		HostTimer tmJ{ "batch" };
		{ std::lock_guard<std::mutex> g{ qmu }; stop = true; }
		qcv.notify_one();
		collector.join();
		tmJ.lap("caller waiting for the collector after its last launch");
		if (collectorError) { waitInFlight(); std::rethrow_exception(collectorError); }
		return all;
	}

	// Kiwi::analyze(text, option, pretokenized): one text whose spans are fixed by the caller (pretok.hpp) -- the spans' forms (temporary ones behind the model's
	// tables for this batch), the chunk cut and the recognisers stepping over them, ONE forced lattice node per span, typoFormId of the tokens inside
	BatchResults Engine::analyzePretokenized(const char16_t* text, size_t n, const std::vector<PtSpan>& spans, size_t topN, uint64_t match, bool openEnding, int hostThreads, TypoOption typo)
	{
		if (topN < 1 || topN > kMaxTopN) throw std::invalid_argument{ "kiwi_amd: top_n must be 1.." + std::to_string(kMaxTopN) + " on the device path" };
		auto g = std::make_shared<PretokGroup>();
		makePretokGroup(impl->model, text, n, match, spans, *g);
		std::vector<std::pair<const char16_t*, size_t>> texts{ { text, n } };
		auto b = stagePretok(texts, match, openEnding, hostThreads, typo, g);
		b->topN = (uint32_t)topN;
		run(*b);
		return fetch(*b, topN);
	}

	std::vector<uint8_t> Engine::dumpLattices(const char16_t* text, size_t n, uint64_t match)
	{
		std::vector<std::pair<const char16_t*, size_t>> texts{ { text, n } };
		std::lock_guard<std::recursive_mutex> devLock{ impl->deviceMu };
		auto b = stage(texts, match, false, 1);
		run(*b);
		const size_t nC = b->refs.size();
		std::vector<uint32_t> nNodes(nC);
		std::vector<DevNode> nodes(b->nodeBase[nC]);
		std::vector<DevChunkResult> res(nC);
		if (nC)
		{
			HIPCHECK(hipMemcpy(nNodes.data(), b->dNNodes.p, nC * 4, hipMemcpyDeviceToHost));
			HIPCHECK(hipMemcpy(nodes.data(), b->dNodes.p, nodes.size() * sizeof(DevNode), hipMemcpyDeviceToHost));
			HIPCHECK(hipMemcpy(res.data(), b->dResults.p, nC * sizeof(DevChunkResult), hipMemcpyDeviceToHost));
		}
		std::vector<uint8_t> out;
		auto put32 = [&](uint32_t v) { out.insert(out.end(), (uint8_t*)&v, (uint8_t*)&v + 4); };
		const PreparedView& pt = b->prep[0];
		put32((uint32_t)pt.chunks.size());
		size_t ri = 0;
		for (size_t c = 0; c < pt.chunks.size(); ++c)
		{
			const ChunkDesc& d = pt.chunks[c];
			if (d.empty)
			{
				put32(2); put32(d.nextOffset);
				for (int k = 0; k < 2; ++k) { put32(0); put32(0); put32(0); put32(0); put32(0xFFFFFFFFu); put32(0); put32(0); put32(0); put32(0); }
				continue;
			}
			if (res[ri].status >= 16) throw std::runtime_error{ "dumpLattices: device status " + std::to_string(res[ri].status) };
			const uint32_t G = nNodes[ri];
			put32(G); put32(d.nextOffset);
			for (uint32_t k = 0; k < G; ++k)
			{
				const DevNode& nd = nodes[b->nodeBase[ri] + k];
				const bool inner = k >= 1 && k + 1 < G;
				put32(nd.startPos + ((inner || k + 1 == G) ? d.startOffset : 0)); put32(nd.endPos + ((inner || k + 1 == G) ? d.startOffset : 0));
				put32(nd.prev); put32(nd.sibling);
				put32(nd.form == NOFORM ? 0xFFFFFFFFu : nd.form);
				put32(nd.uformLen); put32(nd.uformLen ? nd.uformOff + d.startOffset : 0);
				put32(nd.spaceErrors); put32(0);
			}
			++ri;
		}
		return out;
	}

	// Parity hook of typo_graph_kernel.hip: the typo graph of a whole text (normalised, taken as one chunk) in the layout of kamd_typo_graph, followed
	// by two bytes per node ({type, script} of the last character of its form); useDevice: from the kernel's two passes, else from the host module.
	std::vector<uint8_t> Engine::dumpTypoGraph(const PreparedTypo& typo, uint16_t allowedDialect, const char16_t* text, size_t n, bool normCoda, bool useDevice)
	{
		PreparedText pt;
		prepareText(pt, text, n, normCoda ? (uint64_t)M_NORMALIZE_CODA : 0, 0);
		const uint32_t L = (uint32_t)pt.norm.size();
		std::vector<TypoGraphNode> g; std::vector<uint8_t> last; size_t maxIdx = 0;
		if (!useDevice)
		{
			maxIdx = typo.graph((const char16_t*)pt.norm.data(), L, allowedDialect, g);
			last.resize(2 * g.size());
			for (size_t i = 0; i < g.size(); ++i) typo.lastOf(g[i], (const char16_t*)pt.norm.data(), &last[2 * i]);
		}
		else
		{
			std::lock_guard<std::recursive_mutex> devLock{ impl->deviceMu };
			HIPCHECK(hipSetDevice(impl->device));
			hipStream_t s = impl->stream;
			std::vector<uint16_t> chars(pt.norm.begin(), pt.norm.end());
			DevBuf dChars, dCls, dScript, dGch, dGout, dMatches, dBp, dEpm, dRev, dCnt, dTemp, dGraph, dLast;
			TypoGraphDev gd;
			uploadTypoTables(gd, typo, s);
			upload(dChars, chars, s); upload(dCls, pt.cls, s); upload(dScript, pt.script, s);
			std::vector<TypoGraphChunk> gch(1, TypoGraphChunk{ 0, L, 0, 0, 0, typo.scratchCapFor(L, 0) });
			upload(dGch, gch, s); dGout.ensure(64);
			dMatches.ensure((size_t)gch[0].scrCap * 8 + 16); dBp.ensure((size_t)gch[0].scrCap * 4 + 16); dEpm.ensure((size_t)gch[0].scrCap * 8 + 16);
			TypoGraphView gv{};
			gv.chars = dChars.as<uint16_t>(); gv.cls = dCls.as<uint8_t>(); gv.script = dScript.as<uint8_t>(); gv.allowedDialect = allowedDialect;
			gv.chunks = dGch.as<TypoGraphChunk>(); gv.out = dGout.as<TypoGraphOut>(); gv.matches = dMatches.as<uint2>(); gv.bp = dBp.as<uint32_t>(); gv.epm = dEpm.as<uint2>();
			launchTypoGraph(gd.tables, gv, 1, true, s);
			HIPCHECK(hipGetLastError());
			TypoGraphOut out{};
			HIPCHECK(hipMemcpyAsync(&out, dGout.p, sizeof(out), hipMemcpyDeviceToHost, s));
			HIPCHECK(hipStreamSynchronize(s));
			if (out.status) throw std::runtime_error{ "typo graph kernel (count pass): status " + std::to_string(out.status) };
			const uint32_t cnt = out.graphCnt;
			gch[0].graphCap = cnt; gch[0].scrCap = typo.scratchCapFor(L, cnt);
			upload(dGch, gch, s);
			const size_t sc = gch[0].scrCap;
			dMatches.ensure(sc * 8 + 16); dBp.ensure(sc * 4 + 16); dEpm.ensure(sc * 8 + 16); dRev.ensure(sc * 4 + 16); dCnt.ensure(sc * 4 + 16);
			dTemp.ensure((size_t)cnt * sizeof(TypoGraphNode) + 64); dGraph.ensure((size_t)cnt * sizeof(TypoGraphNode) + 64); dLast.ensure((size_t)cnt * 2 + 64);
			gv.chunks = dGch.as<TypoGraphChunk>(); gv.matches = dMatches.as<uint2>(); gv.bp = dBp.as<uint32_t>(); gv.epm = dEpm.as<uint2>(); gv.rev = dRev.as<uint32_t>(); gv.cnt = dCnt.as<uint32_t>();
			gv.temp = dTemp.as<TypoGraphNode>(); gv.graph = dGraph.as<TypoGraphNode>(); gv.graphLast = dLast.as<uint8_t>();
			launchTypoGraph(gd.tables, gv, 1, false, s);
			HIPCHECK(hipGetLastError());
			g.resize(cnt); last.resize(2 * (size_t)cnt);
			TypoGraphOut out2{};
			HIPCHECK(hipMemcpyAsync(&out2, dGout.p, sizeof(out2), hipMemcpyDeviceToHost, s));
			if (cnt) { HIPCHECK(hipMemcpyAsync(g.data(), dGraph.p, (size_t)cnt * sizeof(TypoGraphNode), hipMemcpyDeviceToHost, s)); HIPCHECK(hipMemcpyAsync(last.data(), dLast.p, 2 * (size_t)cnt, hipMemcpyDeviceToHost, s)); }
			HIPCHECK(hipStreamSynchronize(s));
			if (out2.status || out2.graphCnt != cnt || out2.maxCti != out.maxCti) throw std::runtime_error{ "typo graph kernel (write pass): status " + std::to_string(out2.status) + ", the two passes disagree" };
			maxIdx = out.maxCti;
		}
		std::vector<uint8_t> o;
		auto put = [&](const void* v, size_t nb) { const uint8_t* q = (const uint8_t*)v; o.insert(o.end(), q, q + nb); };
		auto put32 = [&](uint32_t v) { put(&v, 4); };
		put32(L); put(pt.norm.data(), 2 * (size_t)L);
		put32((uint32_t)g.size());
		for (auto& nd : g)
		{
			const std::u16string f = typo.formOf(nd, (const char16_t*)pt.norm.data());
			put32((uint32_t)f.size()); put(f.data(), 2 * f.size());
			put32(nd.endPos); put(&nd.typoCost, 4); put32(nd.prevOffset); put32(nd.siblingOffset); put(&nd.continualTypoIdx, 1); put(&nd.dialect, 2);
		}
		put32((uint32_t)maxIdx);
		put(last.data(), last.size());
		return o;
	}

	// Parity hook: typo graphs on the host (typo.cpp), the lattice over each of them by k_build_lattice_typo, dumped like dumpLattices.
	std::vector<uint8_t> Engine::dumpTypoLattices(const PreparedTypo& typo, float threshold, uint16_t allowedDialect, const char16_t* text, size_t n, uint64_t match)
	{
		PreparedText pt;
		prepareText(pt, text, n, match, 0);
		std::vector<TypoLatChunk> chunks; std::vector<uint32_t> chunkOf;
		std::vector<TypoGraphNode> graph; std::vector<uint8_t> graphLast; std::vector<DevPattern> pats;
		uint32_t nodeTop = 0, mapTop = 0, nsTop = 0, stateTop = 0;
		std::vector<TypoGraphNode> g;
		for (size_t ci = 0; ci < pt.chunks.size(); ++ci)
		{
			const ChunkDesc& d = pt.chunks[ci];
			if (d.empty) continue;
			const char16_t* str = (const char16_t*)pt.norm.data() + d.startOffset;
			if (d.nChars > 20000) throw std::runtime_error{ "typo lattices: chunk too long" };
			const size_t maxCti = typo.graph(str, d.nChars, allowedDialect, g);
			TypoLatChunk c{};
			c.charOff = d.startOffset; c.nChars = d.nChars; c.textOffset = d.startOffset;
			c.patOff = (uint32_t)pats.size(); c.patCnt = d.patEnd - d.patBegin;
			for (uint32_t k = d.patBegin; k < d.patEnd; ++k) pats.push_back(DevPattern{ pt.patterns[k].end, pt.patterns[k].length, pt.patterns[k].tag });
			c.graphOff = (uint32_t)graph.size(); c.graphCnt = (uint32_t)g.size();
			for (auto& gn : g)
			{
				// type / script of the node's last character as progressNode leaves it in prevChr (surrogate pairs merged; NUL = none)
				uint32_t lastC = 0; bool any = false;
				const std::u16string f = typo.formOf(gn, str);
				for (size_t j = 0; j < f.size(); ++j)
				{
					uint32_t c32 = f[j];
					if (isHighSurrogate(c32) && j + 1 < f.size()) { c32 = mergeSurrogate(c32, f[j + 1]); ++j; }
					lastC = c32; any = true;
				}
				graphLast.push_back((any && lastC) ? identifySpecialChr(lastC) : (uint8_t)0xFF);
				graphLast.push_back((any && lastC) ? chr2ScriptType(lastC) : (uint8_t)0);
				graph.push_back(gn);
			}
			c.pmb = 0;
			if (maxCti > 1) { size_t v = maxCti - 1; while (v > 0) { v >>= 1; ++c.pmb; } }
			uint32_t nNs = 0;
			for (uint32_t i = 0; i < d.nChars; ++i) if (!isSpace(str[i])) { ++nNs; if (isHighSurrogate(str[i]) && i + 1 < d.nChars) { ++nNs; ++i; } }
			c.nNs = nNs;
			c.nodeOff = nodeTop; c.nodeCap = 16 * d.nChars + 64; nodeTop += c.nodeCap;
			c.mapOff = mapTop; c.mapLen = (nNs << c.pmb) + 1; mapTop += c.mapLen;
			c.nsOff = nsTop; nsTop += d.nChars + 2;
			c.stateOff = stateTop; c.stateCap = (uint32_t)g.size() * 16 + 64; stateTop += c.stateCap;
			chunkOf.push_back((uint32_t)ci);
			chunks.push_back(c);
		}
		std::vector<TypoLatNode> fin(nodeTop);
		if (!chunks.empty())
		{
			std::lock_guard<std::recursive_mutex> devLock{ impl->deviceMu };
			HIPCHECK(hipSetDevice(impl->device));
			hipStream_t s = impl->stream;
			DevBuf dChars, dCls, dScript, dPats, dGraph, dGraphLast, dPool, dChunks, dNodes, dFinal, dMap, dNs, dPs, dStates, dSIdx, dScratch;
			std::vector<uint16_t> chars(pt.norm.begin(), pt.norm.end()), pool(typo.pool().begin(), typo.pool().end());
			if (pats.empty()) pats.push_back(DevPattern{ 0, 0, 0 });
			if (pool.empty()) pool.push_back(0);
			upload(dChars, chars, s); upload(dCls, pt.cls, s); upload(dScript, pt.script, s); upload(dPats, pats, s);
			upload(dGraph, graph, s); upload(dGraphLast, graphLast, s); upload(dPool, pool, s); upload(dChunks, chunks, s);
			dNodes.ensure((size_t)nodeTop * sizeof(TypoLatNode)); dFinal.ensure((size_t)nodeTop * sizeof(TypoLatNode)); dMap.ensure((size_t)mapTop * 8 + 16);
			dNs.ensure((size_t)nsTop * 2 + 16); dPs.ensure((size_t)nsTop * 2 + 16); dStates.ensure((size_t)stateTop * sizeof(TypoState)); dSIdx.ensure(graph.size() * 8 + 16);
			dScratch.ensure((size_t)nodeTop * 12 + 16);
			TypoLatView v{};
			v.chars = dChars.as<uint16_t>(); v.cls = dCls.as<uint8_t>(); v.script = dScript.as<uint8_t>(); v.patterns = dPats.as<DevPattern>();
			v.graph = dGraph.as<TypoGraphNode>(); v.graphLast = dGraphLast.as<uint8_t>(); v.pool = dPool.as<uint16_t>(); v.chunks = dChunks.as<TypoLatChunk>();
			v.nodes = dNodes.as<TypoLatNode>(); v.nodesFinal = dFinal.as<TypoLatNode>(); v.endPosMap = dMap.as<uint2>(); v.nsToPos = dNs.as<uint16_t>(); v.posToNs = dPs.as<uint16_t>();
			v.states = dStates.as<TypoState>(); v.stateIdx = dSIdx.as<uint32_t>(); v.scratch = dScratch.as<uint32_t>();
			v.lengtheningCost = typo.lengtheningCost();
			v.threshold = threshold; v.maxUnk = config.maxUnkFormSize; v.maxUnkJ = config.maxUnkFormSizeFollowedByJClass; v.spaceTol = config.spaceTolerance; v.match = match;
			launchTypoLattice(impl->dview, v, (uint32_t)chunks.size(), s);
			HIPCHECK(hipGetLastError());
			HIPCHECK(hipMemcpyAsync(chunks.data(), dChunks.p, chunks.size() * sizeof(TypoLatChunk), hipMemcpyDeviceToHost, s));
			HIPCHECK(hipMemcpyAsync(fin.data(), dFinal.p, fin.size() * sizeof(TypoLatNode), hipMemcpyDeviceToHost, s));
			HIPCHECK(hipStreamSynchronize(s));
		}
		std::vector<uint8_t> out;
		auto put32 = [&](uint32_t v) { out.insert(out.end(), (uint8_t*)&v, (uint8_t*)&v + 4); };
		put32((uint32_t)pt.chunks.size());
		size_t ri = 0;
		for (size_t c = 0; c < pt.chunks.size(); ++c)
		{
			const ChunkDesc& d = pt.chunks[c];
			if (d.empty)
			{
				put32(2); put32(d.nextOffset);
				for (int k = 0; k < 2; ++k) { put32(0); put32(0); put32(0); put32(0); put32(0xFFFFFFFFu); put32(0); put32(0); put32(0); put32(0); }
				continue;
			}
			const TypoLatChunk& tc = chunks[ri++];
			if (tc.status != CS_OK) throw std::runtime_error{ "typo lattices: device status " + std::to_string(tc.status) };
			put32(tc.nOutFinal); put32(d.nextOffset);
			for (uint32_t k = 0; k < tc.nOutFinal; ++k)
			{
				const TypoLatNode& nd = fin[tc.nodeOff + k];
				put32(nd.startPos); put32(nd.endPos); put32(nd.prev); put32(nd.sibling); put32((uint32_t)nd.form);
				put32(nd.uformLen); put32(nd.uformLen ? nd.uformOff : 0); put32(nd.spaceErrors);
				uint32_t cost; std::memcpy(&cost, &nd.typoCost, 4); put32(cost);
			}
		}
		return out;
	}
}
