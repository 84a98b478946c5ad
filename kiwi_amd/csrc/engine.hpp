// Batch driver around the HIP kernels: the MI355X counterpart of the reference's per-sentence loop in
// Kiwi::analyze (/root/reference/src/Kiwi.cpp:1095-1141) and of its thread-pool batch driver
// (/root/reference/include/kiwi/Kiwi.h:402-454).  Host code prepares chunks (textprep), the kernels build
// lattices and search them, host code stitches chunk results into token lists (post).
#pragma once
#include <algorithm>
#include <cstdint>
#include <functional>
#include <memory>
#include <string>
#include <vector>
#include "device_types.hpp"
#include "post.hpp"
#include "pretok.hpp"
#include "textprep.hpp"

namespace kamd
{
	struct EngineConfig   // KiwiConfig (include/kiwi/Kiwi.h:150-167)
	{
		bool integrateAllomorph = true;
		float cutOffThreshold = 8, oovRuleScale = 4, oovRuleBias = 4, oovChrBias = 0, spacePenalty = 7, typoCostWeight = 6;
		uint32_t maxUnkFormSize = 6, maxUnkFormSizeFollowedByJClass = 0xFFFFFFFFu, spaceTolerance = 0;
		float oovGlobalWeight = 35, oovLocalWeight = 3, oovGlobalMinFreq = 4;      // Match::oovChrFreqModel (include/kiwi/Kiwi.h:157-159)
	};

	struct KernelTimes { float scanMs = 0, latticeMs = 0, searchMs = 0, finishMs = 0; uint32_t searchLaunches = 1;   // sums over the sub-batches of one run
		uint32_t rerunChunks = 0; float rerunMs = 0; };   // chunks that overflowed their scratch and were searched again inside the run (wall time of that)

	struct StagedBatch;   // chunks of one round resident in HBM

	// Results of a batch: flat segments of kSegTexts consecutive texts each (post.hpp ResultSegment), filled in parallel by the host
	// workers; a text whose chunks had to be searched again on the device (other start states than the speculative ones, scratch
	// overflow) is finished afterwards and kept as a one-text override.
	struct BatchResults
	{
		static constexpr size_t kSegTexts = 128;
		size_t nTexts = 0;
		std::vector<ResultSegment> segs;
		std::vector<std::pair<size_t, ResultSegment>> overrides;     // sorted by text
		// segment and local text index of text t
		const ResultSegment& locate(size_t t, size_t& local) const
		{
			if (!overrides.empty())
			{
				auto it = std::lower_bound(overrides.begin(), overrides.end(), t, [](const std::pair<size_t, ResultSegment>& a, size_t v) { return a.first < v; });
				if (it != overrides.end() && it->first == t) { local = 0; return it->second; }
			}
			local = t % kSegTexts;
			return segs[t / kSegTexts];
		}
		uint64_t d2hBytes = 0;     // bytes the device -> host copy of this batch moved
	};
	class PreparedTypo;
	// a prepared typo transformer applied to an analysis (AnalyzeOption::typoTransformer / typoThreshold / allowedDialects of the reference)
	struct TypoOption { const PreparedTypo* typo = nullptr; float threshold = 2.5f; uint16_t allowedDialect = 0;
		float dialectCost = 3.f;      // AnalyzeOption::dialectCost: what a morpheme of an allowed dialect costs (src/PathEvaluator.hpp:231)
		// AnalyzeOption::blocklist: one bit per morpheme id (flat_model.hpp blockBitsOf), null = none; must outlive the batch
		const std::vector<uint32_t>* blocked = nullptr; };

	class Engine
	{
	public:
		struct Impl;
	private:
		std::unique_ptr<Impl> impl;
		void openDevice(int device);
	public:
		EngineConfig config;
		// which language model of the container scores the search (reference ModelType, include/kiwi/Types.h:292-335): Auto = SkipBigram when the
		// container carries its tables, else Knlm; Knlm = Knlm even then; Sbg = SkipBigram or an error
		enum class LmMode { Auto, Knlm, Sbg, Cong, CongGlobal };      // Auto: CoNgram (local) when the container has a blob, else SkipBigram when it has tables, else Knlm; CongGlobal: ModelType::congGlobal (window 7)
		// enabledDialects: KiwiBuilder's enabledDialects (kiwi_init's last argument): forms of other dialects stay out of the dictionary trie
		explicit Engine(const std::string& rawModelPath, int device = -1, LmMode lm = LmMode::Auto, uint32_t enabledDialects = 0);
		Engine(const Engine& other, int device);      // replica of `other` on another GPU (shares the baked host model)
		static int visibleDevices();
		int deviceIndex() const;      // the HIP device this engine's tables and streams live on
		void bindThread() const;      // makes that device the calling thread's current one
		bool usesCong() const; bool usesSbg() const;      // which language model scores the search
		bool usesCongGlobal() const;                      // the distant-token (window) sections are scored
		uint32_t congWindow() const;                      // window size of the CoNgram model's distant-token sections (0: none, or not a CoNgram model)
		~Engine();
		const FlatModel& model() const;

		// Full path: prepare -> kernels -> results, for a batch of raw UTF-16 texts.  Results are per text.
		// onPart (optional): the results are handed over PART BY PART, in text order, as the parts of a large batch complete -- onPart(first text of the part, its
		// results), called from the thread that assembled them while later parts are still on the device; the call then returns an empty BatchResults
		using PartSink = std::function<void(size_t, BatchResults&&)>;
		BatchResults analyzeBatch(const std::vector<std::pair<const char16_t*, size_t>>& texts,
			size_t topN, uint64_t match, bool openEnding, int hostThreads = 0, TypoOption typo = {}, const PartSink* onPart = nullptr);

		// Staged path (benchmarks): stage() does host preparation + upload of every chunk of the texts (one round,
		// the common case where no quote/bullet state crosses chunk boundaries); run() launches the three kernels on
		// the resident batch and returns their event-timed durations; fetch() downloads and assembles results.
		std::shared_ptr<StagedBatch> stage(const std::vector<std::pair<const char16_t*, size_t>>& texts, uint64_t match, bool openEnding, int hostThreads = 0, TypoOption typo = {});
		// ... with the pretokenized spans of texts[0] (pretok.hpp; null / no spans: stage())
		std::shared_ptr<StagedBatch> stagePretok(const std::vector<std::pair<const char16_t*, size_t>>& texts, uint64_t match, bool openEnding, int hostThreads, TypoOption typo,
			std::shared_ptr<const PretokGroup> pretok);
		// Kiwi::analyze(text, option, pretokenized) for one text (reference src/Kiwi.cpp:1014-1158 with :1043-1051)
		BatchResults analyzePretokenized(const char16_t* text, size_t n, const std::vector<PtSpan>& spans, size_t topN, uint64_t match, bool openEnding, int hostThreads = 0, TypoOption typo = {});
		KernelTimes run(StagedBatch& b);
		// run() in two halves: launch() enqueues the batch's kernels and returns, finish() waits for THIS batch and re-runs what overflowed -- the host
		// prepares the next batch in between (analyzeBatch cuts a large batch into parts that way)
		void launch(StagedBatch& b);
		void finish(StagedBatch& b);
		void rerunOverflows(StagedBatch& b, KernelTimes& t);      // (internal: the capacity ladder behind run() / finish())
		BatchResults fetch(StagedBatch& b, size_t topN);
		static size_t stagedChunks(const StagedBatch& b);
		static uint64_t stagedUnits(const StagedBatch& b);     // non-space normalised units ("jamo")
		static uint64_t stagedDeviceBytes(const StagedBatch& b);
		static void stagedPool(const StagedBatch& b, uint64_t* out3);      // states in the chunks' own arenas, states of the pool behind them, states of the pool the last fetched run asked for
		// chunks of the last run() that overflowed their scratch in the first pass and were searched again inside run(), and the wall time of that
		static uint32_t rerunChunks(const StagedBatch& b, float* ms);

		// debugging / parity hooks: lattice of every chunk of one text in the layout of oracle's korc_split
		std::vector<uint8_t> dumpLattices(const char16_t* text, size_t n, uint64_t match);
		// ... and the lattices built over the typo graphs a prepared transformer gives for the chunks (typo_lattice_kernel.hip); same layout.
		// Parity hook of a building block: analyze does not take typo transformers yet.
		std::vector<uint8_t> dumpTypoLattices(const PreparedTypo& typo, float threshold, uint16_t allowedDialect, const char16_t* text, size_t n, uint64_t match);
		// ... and the typo graph itself, from the device kernel the analyze path uses (typo_graph_kernel.hip) or from the host module: layout of
		// kamd_typo_graph + two bytes per node (type and script of the last character of its form)
		std::vector<uint8_t> dumpTypoGraph(const PreparedTypo& typo, uint16_t allowedDialect, const char16_t* text, size_t n, bool normCoda, bool useDevice);
	};
}
