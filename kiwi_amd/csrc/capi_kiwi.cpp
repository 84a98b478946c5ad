// Kiwi-compatible C API (include/kiwi_capi.h) on top of kamd::Engine: the drop-in boundary for
// Kiwi::analyze.  Conventions follow /root/reference/src/capi/kiwi_c.cpp: handles are heap objects owned by the
// caller, nothing throws across the boundary, failures are recorded in a thread-local slot read by kiwi_error().
#include <chrono>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <future>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <sys/stat.h>
#include <thread>
#include <cstdlib>
#include "../../include/kiwi_capi.h"
#include "engine.hpp"
#include "hostpool.hpp"
#include "typo.hpp"

using namespace kamd;

struct kiwi_s
{
	std::unique_ptr<Engine> engine;                      // on the caller's current device at kiwi_init: single-text calls, configuration
	// replicas of the device tables on the other visible GPUs (KAMD_DEVICES=n limits it; 1 = single-GPU behaviour): created by the first kiwi_analyze_m /
	// _mw call, which spreads a batch over all of them -- a process that only calls kiwi_analyze, or one rank of a one-process-per-GPU job that did not
	// narrow HIP_VISIBLE_DEVICES, allocates nothing on the other GPUs
	std::vector<std::unique_ptr<Engine>> replicas;
	std::mutex replicaMu; bool replicasMade = false;
	int numThreads = 0;
	int batchSize = 65536;
	Engine& device(size_t d) { return d == 0 ? *engine : *replicas[d - 1]; }
	size_t devices() const { return 1 + replicas.size(); }
	void makeReplicas()
	{
		std::lock_guard<std::mutex> g{ replicaMu };
		if (replicasMade) return;
		int nDev = Engine::visibleDevices();
		if (const char* e = std::getenv("KAMD_DEVICES")) nDev = std::max(1, std::min(nDev, std::atoi(e)));
		const int own = engine->deviceIndex();
		// built aside and swapped in whole: a replica that fails to open (out of memory on one GPU) leaves the handle as it was -- no half set that the
		// next call would extend with duplicates -- and the thread back on the handle's own device
		std::vector<std::unique_ptr<Engine>> made;
		try
		{
			for (int d = 0; d < Engine::visibleDevices() && (int)made.size() + 1 < nDev; ++d) if (d != own) made.emplace_back(new Engine(*engine, d));
		}
		catch (...) { engine->bindThread(); throw; }
		engine->bindThread();      // (opening a replica made its device the thread's current one)
		replicas.swap(made);
		replicasMade = true;
	}
};

struct kiwi_typo { kamd::TypoTransformer tt; };                       // capi.h:35
struct kiwi_morphset { kiwi_h owner; std::vector<uint32_t> ids, bits; };   // capi.h:36 (src/capi/kiwi_c.cpp: a set of Morpheme pointers; here ids + the bit set the kernels test)
struct kiwi_prepared_typo { kamd::PreparedTypo p; };                  // capi.h:38

// One text's analyses, flat (its slice of a batch's ResultSegment): token records whose first 44 bytes are kiwi_token_info_t -- the
// reference hands out a pointer into its TokenInfo the same way (kiwi_c.cpp:1097) -- and one pool of NUL-terminated UTF-16 forms.
// UTF-8 forms and UTF-16 tag names are materialised on first request, as the reference's ResultBuffer does (kiwi_c.cpp:20-23).
// A result handed to the caller: a VIEW of one text's analyses inside the batch's flat result segments (post.hpp), which every result of the batch
// keeps alive together -- no per-text copies of token records or form strings (the reference hands out a vector of its own per text,
// src/capi/kiwi_c.cpp:45-60; a batch of 65 536 texts made 5 allocations and 4 copies per text here before).
// Results of kiwi_analyze_m are handed to the receiver one handle per line; the handles of a delivered part live side by side in one block (ResSlab)
// that goes when the last of them is closed -- a heap block per line and its release were 40 % of what delivering a line costs the calling thread.
struct ResSlab { std::unique_ptr<unsigned char[]> mem; };
struct kiwi_res
{
	std::shared_ptr<ResSlab> slab;      // set: this handle lies in the slab's block (kiwi_res_close destroys it in place)
	std::shared_ptr<const BatchResults> batch;
	const ResultSegment* seg = nullptr;
	uint32_t a0 = 0, nAna = 0;      // analyses [a0, a0 + nAna) of the segment
	std::map<std::pair<int, int>, std::string> formBuf;
	std::map<std::pair<int, int>, std::u16string> tagBufW;
	size_t size() const { return nAna; }
	size_t tokens(int index) const { return seg->anaTok[a0 + index + 1] - seg->anaTok[a0 + index]; }
	const FlatToken& tok(int index, int num) const { return seg->toks[seg->anaTok[a0 + index] + num]; }
	float score(int index) const { return seg->anaScore[a0 + index]; }
	const char16_t* form(int index, int num) const { return seg->forms.data() + tok(index, num).formOff; }      // NUL-terminated
};
static_assert(offsetof(FlatToken, dialect) == offsetof(kiwi_token_info_t, dialect) && offsetof(FlatToken, score) == offsetof(kiwi_token_info_t, score)
	&& offsetof(FlatToken, subSentPosition) == offsetof(kiwi_token_info_t, sub_sent_position) && offsetof(FlatToken, tag) == offsetof(kiwi_token_info_t, tag), "FlatToken starts with kiwi_token_info_t");

namespace
{
	thread_local std::string currentError;
	thread_local bool hasError = false;
	void setError(const std::exception& e) { currentError = e.what(); hasError = true; }

	const char* tagNames[] = { "UN", "NNG", "NNP", "NNB", "VV", "VA", "MAG", "NR", "NP", "VX", "MM", "MAJ", "IC", "XPN", "XSN", "XSV", "XSA", "XSM", "XR", "VCP", "VCN",
		"SF", "SP", "SS", "SSO", "SSC", "SE", "SO", "SW", "SB", "SL", "SH", "SN", "W_URL", "W_EMAIL", "W_MENTION", "W_HASHTAG", "W_SERIAL", "W_EMOJI",
		"JKS", "JKC", "JKG", "JKO", "JKB", "JKV", "JKQ", "JX", "JC", "EP", "EF", "EC", "ETN", "ETM", "Z_CODA", "Z_SIOT", "USER0", "USER1", "USER2", "USER3", "USER4", "@", "@" };
	const char* tagToString(uint8_t t)   // src/Utils.cpp tagToString: irregular variants get an -I suffix
	{
		if (t & 0x80)
		{
			switch (t & 0x7F) { case T_VV: return "VV-I"; case T_VA: return "VA-I"; case T_VX: return "VX-I"; case T_XSA: return "XSA-I"; default: return "@"; }
		}
		return t < sizeof(tagNames) / sizeof(tagNames[0]) ? tagNames[t] : "@";
	}

	uint8_t parseTag(const char* pos)   // parse_tag (src/capi/kiwi_c.cpp:86-93) + toPOSTag (src/StrUtils.h:552-633)
	{
		std::string u;
		for (const char* p = pos; *p; ++p) u.push_back((char)std::toupper((unsigned char)*p));
		for (uint8_t t = 1; t < T_P; ++t) if (u == tagNames[t]) return t;
		if (u == "NF" || u == "NV" || u == "NA" || u == "UNK" || u == "UN" || u == "^") return T_UNKNOWN;
		if (u == "V" || u == "A") return T_P;
		if (u == "VV-I") return T_VV | 0x80; if (u == "VA-I") return T_VA | 0x80; if (u == "VX-I") return T_VX | 0x80; if (u == "XSA-I") return T_XSA | 0x80;
		if (u == "VV-R") return T_VV; if (u == "VA-R") return T_VA; if (u == "VX-R") return T_VX; if (u == "XSA-R") return T_XSA;
		throw std::invalid_argument{ std::string{ "Unknown POSTag : " } + pos };
	}

	// UTF-8 -> UTF-16 into `out` (room for n units: a code point never takes more UTF-16 units than it took UTF-8 bytes); src/StrUtils.h:228-303 (strict
	// decoder, throws on malformed input).  Returns the number of units written.
	size_t utf8To16Into(const char* s, size_t n, char16_t* out)
	{
		size_t k = 0;
		for (size_t i = 0; i < n; ++i)
		{
			uint32_t code, b = (uint8_t)s[i];
			auto cont = [&]() -> uint32_t
			{
				if (++i == n) throw std::runtime_error{ "unexpected ending" };
				const uint32_t c = (uint8_t)s[i];
				if ((c & 0xC0) != 0x80) throw std::runtime_error{ "unexpected trailing byte" };
				return c & 0x3F;
			};
			if ((b & 0xF8) == 0xF0) { code = (b & 7) << 18; code |= cont() << 12; code |= cont() << 6; code |= cont(); }
			else if ((b & 0xF0) == 0xE0) { code = (b & 0xF) << 12; code |= cont() << 6; code |= cont(); }
			else if ((b & 0xE0) == 0xC0) { code = (b & 0x1F) << 6; code |= cont(); }
			else if ((b & 0x80) == 0) code = b;
			else throw std::runtime_error{ "unicode error" };
			if (code < 0x10000) out[k++] = (char16_t)code;
			else if (code < 0x10FFFF) { code -= 0x10000; out[k++] = (char16_t)(0xD800 | (code >> 10)); out[k++] = (char16_t)(0xDC00 | (code & 0x3FF)); }
			else throw std::runtime_error{ "unicode error" };
		}
		return k;
	}
	std::u16string utf8To16(const char* s, size_t n)
	{
		std::u16string ret(n, u'\0');
		ret.resize(n ? utf8To16Into(s, n, &ret[0]) : 0);
		return ret;
	}

	std::string utf16To8(const std::u16string& s)
	{
		std::string ret;
		for (size_t i = 0; i < s.size(); ++i)
		{
			uint32_t c = s[i];
			if (isHighSurrogate(c) && i + 1 < s.size() && isLowSurrogate(s[i + 1])) c = mergeSurrogate(c, s[++i]);
			if (c <= 0x7F) ret.push_back((char)c);
			else if (c <= 0x7FF) { ret.push_back((char)(0xC0 | (c >> 6))); ret.push_back((char)(0x80 | (c & 0x3F))); }
			else if (c <= 0xFFFF) { ret.push_back((char)(0xE0 | (c >> 12))); ret.push_back((char)(0x80 | ((c >> 6) & 0x3F))); ret.push_back((char)(0x80 | (c & 0x3F))); }
			else { ret.push_back((char)(0xF0 | (c >> 18))); ret.push_back((char)(0x80 | ((c >> 12) & 0x3F))); ret.push_back((char)(0x80 | ((c >> 6) & 0x3F))); ret.push_back((char)(0x80 | (c & 0x3F))); }
		}
		return ret;
	}

}
// kiwi_pretokenized (src/capi/kiwi_c.cpp: a std::vector<PretokenizedSpan>), built through the reference's entry points; kiwi_analyze{,_w} hand its spans to
// Engine::analyzePretokenized (pretok.hpp: makePretokenizedSpanGroup restated, temporary forms / morphemes as a per-batch overlay on the device)
struct kiwi_pretokenized
{
	struct Token { std::u16string form; std::string tag; int begin, end; };
	struct Span { int begin, end; std::vector<Token> tokens; };
	std::vector<Span> spans;
};
namespace
{
	void checkOption(const kiwi_analyze_option_t& o, kiwi_pretokenized_h pt)
	{
		// allowed_dialects / dialect_cost: the candidate loops skip a morpheme whose dialect is neither standard nor allowed and charge dialect_cost for an
		// allowed one (src/PathEvaluator.hpp:231, 386, 893): typoOf below hands them to the engine (round 5; a model without dialect morphemes: no effect,
		// exactly as in the reference -- except that an analysis with a dialect allowed and no transformer is corrected with the built-in `dialect` set)
		(void)pt;      // (a pretokenized object without spans is no constraint; with spans: spansOf below)
		// Match::oovChrModel (bits 8-9): the engine checks that the model carries the character model (nounchr.mdl next to a CoNgram model) and refuses
		// with the reference's own message otherwise; the two frequency-based modes are not built
		if ((uint32_t)o.match_options & (1u << 30)) throw std::invalid_argument{ "kiwi_amd: useOldSplitter is not supported" };
	}

	// AnalyzeOption::typoTransformer / typoThreshold
	TypoOption typoOf(const kiwi_analyze_option_t& o);

	// the caller's spans as the engine takes them: offsets into the UTF-16 text.  bytePos (kiwi_analyze: UTF-8 text): the byte offset of every UTF-16 unit --
	// Kiwi::mapPretokenizedSpansToU16 (src/Kiwi.cpp:34-44) maps a span's begin / end through it; the tokens' offsets, relative to their span, are taken as they
	// are (the reference does not map them either)
	std::vector<PtSpan> spansOf(const kiwi_pretokenized& pt, const std::vector<size_t>* bytePos)
	{
		std::vector<PtSpan> ret;
		for (const auto& s : pt.spans)
		{
			if (s.begin < 0 || s.end < 0) throw std::invalid_argument{ "pretokenized span with a negative offset" };
			PtSpan o;
			if (bytePos)
			{
				o.begin = (uint32_t)(std::upper_bound(bytePos->begin(), bytePos->end(), (size_t)s.begin) - bytePos->begin() - 1);
				o.end = (uint32_t)(std::lower_bound(bytePos->begin(), bytePos->end(), (size_t)s.end) - bytePos->begin());
			}
			else { o.begin = (uint32_t)s.begin; o.end = (uint32_t)s.end; }
			for (const auto& tk : s.tokens) o.tokens.push_back(PtToken{ tk.form, (uint32_t)tk.begin, (uint32_t)tk.end, parseTag(tk.tag.c_str()), true });      // (BasicToken::inferRegularity defaults to 1; the C API cannot change it)
			ret.push_back(std::move(o));
		}
		return ret;
	}

	void fillRes(kiwi_res& res, const std::shared_ptr<const BatchResults>& br, size_t text)
	{
		size_t local;
		const ResultSegment& seg = br->locate(text, local);
		res.batch = br; res.seg = &seg;
		res.a0 = seg.textAna[local]; res.nAna = seg.textAna[local + 1] - seg.textAna[local];
	}
	kiwi_res* makeRes(const std::shared_ptr<const BatchResults>& br, size_t text)
	{
		auto res = std::make_unique<kiwi_res>();
		fillRes(*res, br, text);
		return res.release();
	}
	// the handles of texts [first, first + count) of `br`, constructed side by side in one block; handle i is at slabRes(slab, i)
	std::shared_ptr<ResSlab> makeResSlab(size_t count)
	{
		auto slab = std::make_shared<ResSlab>();
		slab->mem.reset(new unsigned char[count * sizeof(kiwi_res) + alignof(kiwi_res)]);
		return slab;
	}
	kiwi_res* slabRes(const std::shared_ptr<ResSlab>& slab, size_t i, const std::shared_ptr<const BatchResults>& br, size_t text)
	{
		unsigned char* base = slab->mem.get();
		base += (alignof(kiwi_res) - reinterpret_cast<uintptr_t>(base) % alignof(kiwi_res)) % alignof(kiwi_res);
		kiwi_res* r = new (base + i * sizeof(kiwi_res)) kiwi_res{};
		r->slab = slab;
		fillRes(*r, br, text);
		return r;
	}

	// kiwi_analyze_m / _mw (src/capi/kiwi_c.cpp:914-960 over Kiwi::analyze(reader, receiver), include/kiwi/Kiwi.h:402-454): reader and receiver are the
	// caller's functions and are called from the calling thread, in input order, like the reference does -- while the batch read before is on the
	// device(s): batch k is analysed by a worker thread while the caller's thread delivers batch k - 1 and reads batch k + 1.
	template<class ReadFn>
	int analyzeMany(kiwi_h h, ReadFn&& readNext, kiwi_receiver_t receiver, void* ud, int topN, const kiwi_analyze_option_t& opt)
	{
		checkOption(opt, nullptr);
		h->makeReplicas();
		// The lines of a batch back to back in ONE buffer (a string per line was two heap blocks per line on the calling thread, whose reading and
		// delivering is what bounds kiwi_analyze_m): `raw8` as the UTF-8 reader delivered them, `w` as UTF-16 -- read directly (_mw), or converted on
		// the worker pool into the slot at the line's byte offset (a line never has more UTF-16 units than UTF-8 bytes).
		struct Job
		{
			std::string raw8; std::u16string w;
			std::vector<size_t> off, len;      // line i: units [off[i], off[i] + len[i]) of w (and bytes [off[i], off[i + 1]) of raw8); off has one more entry
			size_t count() const { return len.size(); }
			std::vector<size_t> cut;
			// results, in text order, as they complete: a part of a large batch (one device: Engine::analyzeBatch hands its parts over one by one while the
			// later ones are still on the device), or a device's whole share (several devices); `done`: nothing more will come
			struct Part { size_t count; std::shared_ptr<const BatchResults> br; };
			std::mutex mu; std::condition_variable cv; std::deque<Part> ready; bool done = false;
			void push(size_t count, BatchResults&& r) { { std::lock_guard<std::mutex> g{ mu }; ready.push_back(Part{ count, std::make_shared<const BatchResults>(std::move(r)) }); } cv.notify_one(); }
			void finish() { { std::lock_guard<std::mutex> g{ mu }; done = true; } cv.notify_one(); }
		};
		auto analyse = [h, topN, &opt](Job& job)
		{
			if (!job.raw8.empty())
			{
				job.w.resize(job.raw8.size());
				HostPool::instance().run(job.count(), 512, h->numThreads, [&](size_t a, size_t b, int)
				{
					for (size_t i = a; i < b; ++i) job.len[i] = utf8To16Into(job.raw8.data() + job.off[i], job.off[i + 1] - job.off[i], &job.w[job.off[i]]);
				});
				std::string().swap(job.raw8);      // (the batch keeps one copy of its texts, not two, while it is on the device)
			}
			std::vector<std::pair<const char16_t*, size_t>> views;
			views.reserve(job.count());
			for (size_t i = 0; i < job.count(); ++i) views.emplace_back(job.w.data() + job.off[i], job.len[i]);
			// One host process drives every GPU (the reference's driver keeps a thread pool busy, include/kiwi/Kiwi.h:402-454): the batch is cut into
			// contiguous parts of about equal text volume, part d is analysed by the engine of device d on its own thread, results are delivered in
			// input order.  (Texts are independent: no exchange between the devices; each holds a replica of the model tables.)
			const size_t nDev = std::min(h->devices(), std::max<size_t>(1, job.count() / 64));
			job.cut.assign(nDev + 1, 0);
			{
				size_t total = 0; for (size_t i = 0; i < job.count(); ++i) total += job.len[i] + 8;
				size_t acc = 0, d = 1;
				for (size_t i = 0; i < job.count() && d < nDev; ++i) { acc += job.len[i] + 8; if (acc * nDev >= total * d) job.cut[d++] = i + 1; }
				for (; d <= nDev; ++d) job.cut[d] = job.count();
			}
			struct Finish { Job& j; ~Finish() { j.finish(); } } finishGuard{ job };      // (also when the analysis throws: the calling thread must not wait for parts that never come)
			if (nDev == 1)
			{
				const Engine::PartSink sink = [&job](size_t, BatchResults&& r) { const size_t n = r.nTexts; job.push(n, std::move(r)); };
				h->device(0).analyzeBatch(views, (size_t)topN, (uint64_t)(uint32_t)opt.match_options, !!opt.open_ending, h->numThreads, typoOf(opt), &sink);
				return;
			}
			std::vector<BatchResults> shares(nDev);
			std::vector<std::exception_ptr> errs(nDev);
			auto work = [&](size_t d)
			{
				try
				{
					std::vector<std::pair<const char16_t*, size_t>> v(views.begin() + job.cut[d], views.begin() + job.cut[d + 1]);
					shares[d] = h->device(d).analyzeBatch(v, (size_t)topN, (uint64_t)(uint32_t)opt.match_options, !!opt.open_ending, h->numThreads, typoOf(opt));
				}
				catch (...) { errs[d] = std::current_exception(); }
			};
			std::vector<std::thread> workers;
			for (size_t d = 1; d < nDev; ++d) workers.emplace_back(work, d);
			work(0);
			for (auto& w : workers) w.join();
			for (auto& e : errs) if (e) std::rethrow_exception(e);
			for (size_t d = 0; d < nDev; ++d) if (job.cut[d + 1] > job.cut[d]) job.push(job.cut[d + 1] - job.cut[d], std::move(shares[d]));
		};
		int readerIdx = 0, receiverIdx = 0;
		// Delivers the parts of `job` that are ready, in order (handles are constructed one at a time, right before their delivery: a receiver that throws leaves no
		// constructed-but-undelivered ones behind).  untilAnalysed: waits for parts and returns as soon as the analysis has ended -- what is still queued then is
		// delivered by the next call, after the following batch has been started; otherwise: everything that is left.
		auto deliver = [&](Job& job, bool untilAnalysed)
		{
			for (;;)
			{
				typename Job::Part part;
				{
					std::unique_lock<std::mutex> lk{ job.mu };
					if (untilAnalysed) { job.cv.wait(lk, [&] { return job.done || !job.ready.empty(); }); if (job.done) return; }
					else if (job.ready.empty()) return;
					part = std::move(job.ready.front()); job.ready.pop_front();
				}
				if (!part.count) continue;
				const auto slab = makeResSlab(part.count);
				for (size_t i = 0; i < part.count; ++i) (*receiver)(receiverIdx++, slabRes(slab, i, part.br, i), ud);   // in input order; the receiver owns the result
			}
		};
		// A batch is handed over when it is full -- or, from 8 192 lines on, as soon as the device side has nothing to do: the first batch of a call starts
		// after a fraction of the reading instead of after all of it, and from then on a batch holds whatever was read while its predecessor was analysed (a
		// reader as fast as memory fills the second batch at once -- two batches, where the round-5 ramp of 8 192, 16 384, 32 768, ... lines cut 65 536 lines
		// into four small ones whose host stages did not overlap: 16.5 of a pass' 20.9 ms waiting for them, profiles/r06_m_*; a slow reader gets the short
		// batches the ramp was made for).
		std::unique_ptr<Job> running, finished;
		std::future<void> pending;
		constexpr size_t kMinBatch = 8192;
		constexpr double kMinReadSeconds = 1.5e-3;      // (a reader that delivers a whole batch in less fills it: short batches cost the device side more than they start it earlier)
		auto deviceIdle = [&] { return !running || pending.wait_for(std::chrono::seconds(0)) == std::future_status::ready; };
		auto readBatch = [&](Job& job)
		{
			if (job.off.empty()) job.off.push_back(0);
			const auto started = std::chrono::steady_clock::now();
			while ((long long)job.count() < (long long)h->batchSize)
			{
				if (!readNext(readerIdx, job.raw8, job.w, job.off, job.len)) return false;
				++readerIdx;
				if (job.count() >= kMinBatch && (job.count() & 1023) == 0 && deviceIdle()
					&& std::chrono::duration<double>(std::chrono::steady_clock::now() - started).count() >= kMinReadSeconds) break;
			}
			return true;
		};
		bool more = true;
		// developer aid (KAMD_CAPI_TIMING=1): where the calling thread's time goes
		static const bool timing = std::getenv("KAMD_CAPI_TIMING") != nullptr;
		double tRead = 0, tWait = 0, tDeliver = 0;
		auto clk = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
		try
		{
			while (more || running)
			{
				std::unique_ptr<Job> next;
				const double t0 = clk();
				if (more)
				{
					next = std::make_unique<Job>();
					more = readBatch(*next);      // (overlaps the batch on the device)
					if (!next->count()) next.reset();
				}
				const double t1 = clk();
				// (the running batch's parts are delivered as they complete -- its later parts are still on the device --, what is left when its analysis ends after the
				// next batch has been started)
				if (running) { deliver(*running, true); pending.get(); finished = std::move(running); }
				const double t2 = clk();
				if (next) { running = std::move(next); Job* j = running.get(); pending = std::async(std::launch::async, [&analyse, j] { analyse(*j); }); }
				if (finished) { deliver(*finished, false); finished.reset(); }      // (overlaps the next batch on the device)
				tRead += t1 - t0; tWait += t2 - t1; tDeliver += clk() - t2;
			}
		}
		catch (...)
		{
			if (pending.valid()) { try { pending.get(); } catch (...) {} }
			throw;
		}
		if (timing) fprintf(stderr, "[kiwi_analyze_m] %d texts: reading %.1f ms, waiting for the device side + delivering its parts as they complete %.1f ms, delivering the rest %.1f ms\n", readerIdx, 1e3 * tRead, 1e3 * tWait, 1e3 * tDeliver);
		return readerIdx;
	}

	bool validIdx(kiwi_res_h r, int index, int num)
	{
		return index >= 0 && (size_t)index < r->size() && num >= 0 && (size_t)num < r->tokens(index);
	}
}

namespace
{
	TypoOption typoOf(const kiwi_analyze_option_t& o)
	{
		TypoOption t;
		t.allowedDialect = (uint16_t)o.allowed_dialects; t.dialectCost = o.dialect_cost;
		if (o.typo_transformer) { t.typo = &o.typo_transformer->p; t.threshold = o.typo_threshold; }
		else if (o.allowed_dialects) { t.typo = &kamd::defaultDialectTypo(); t.threshold = 2.5f; }      // src/Kiwi.cpp:1037-1041
		if (o.blocklist && !o.blocklist->ids.empty()) t.blocked = &o.blocklist->bits;      // AnalyzeOption::blocklist (src/capi/kiwi_c.cpp:870)
		return t;
	}
}

extern "C"
{
	// ---- morpheme sets: the blocklist of kiwi_analyze_option_t (capi.h:655-664, 1243-1263; src/capi/kiwi_c.cpp:851-864, 1779-1826)
	kiwi_morphset_h kiwi_new_morphset(kiwi_h h)
	{
		if (!h) return nullptr;
		try { auto* m = new kiwi_morphset; m->owner = h; return m; }
		catch (const std::exception& e) { setError(e); return nullptr; }
	}
	static int morphsetAdd(kiwi_morphset_h m, const std::u16string& form, const char* tag)
	{
		const uint8_t t = tag ? parseTag(tag) : (uint8_t)T_UNKNOWN;
		const auto found = kamd::findMorphemes(m->owner->engine->model(), form.data(), form.size(), t);
		m->ids.insert(m->ids.end(), found.begin(), found.end());
		m->bits = kamd::blockBitsOf(m->owner->engine->model(), m->ids);
		return (int)found.size();
	}
	int kiwi_morphset_add(kiwi_morphset_h m, const char* form, const char* tag)
	{
		if (!m) return KIWIERR_INVALID_HANDLE;
		try { return morphsetAdd(m, utf8To16(form, std::strlen(form)), tag); }
		catch (const std::exception& e) { setError(e); return KIWIERR_FAIL; }
	}
	int kiwi_morphset_add_w(kiwi_morphset_h m, const kchar16_t* form, const char* tag)
	{
		if (!m) return KIWIERR_INVALID_HANDLE;
		try { size_t n = 0; while (form[n]) ++n; return morphsetAdd(m, std::u16string{ (const char16_t*)form, n }, tag); }
		catch (const std::exception& e) { setError(e); return KIWIERR_FAIL; }
	}
	kiwi_pretokenized_h kiwi_pt_init() { return new (std::nothrow) kiwi_pretokenized; }      // capi.h:1351
	int kiwi_pt_add_span(kiwi_pretokenized_h h, int begin, int end)                            // capi.h:1367: the id of the new span
	{
		if (!h) return KIWIERR_INVALID_HANDLE;
		try { h->spans.push_back(kiwi_pretokenized::Span{ begin, end, {} }); return (int)h->spans.size() - 1; }
		catch (const std::exception& e) { setError(e); return KIWIERR_FAIL; }
	}
	int kiwi_pt_add_token_to_span_w(kiwi_pretokenized_h h, int span_id, const kchar16_t* form, const char* tag, int begin, int end)      // capi.h:1397
	{
		if (!h) return KIWIERR_INVALID_HANDLE;
		try
		{
			if (begin < 0 || end < 0 || span_id < 0 || (size_t)span_id >= h->spans.size()) return KIWIERR_INVALID_INDEX;      // src/capi/kiwi_c.cpp:1994
			(void)parseTag(tag ? tag : "");      // (parse_tag: an unknown tag is refused here, as the reference does, not at the analysis)
			size_t n = 0; while (form[n]) ++n;
			h->spans[(size_t)span_id].tokens.push_back(kiwi_pretokenized::Token{ std::u16string{ (const char16_t*)form, n }, tag ? tag : "", begin, end });
			return 0;
		}
		catch (const std::exception& e) { setError(e); return KIWIERR_FAIL; }
	}
	int kiwi_pt_add_token_to_span(kiwi_pretokenized_h h, int span_id, const char* form, const char* tag, int begin, int end)               // capi.h:1382
	{
		if (!h) return KIWIERR_INVALID_HANDLE;
		try { const std::u16string u = utf8To16(form, std::strlen(form)); return kiwi_pt_add_token_to_span_w(h, span_id, (const kchar16_t*)u.c_str(), tag, begin, end); }
		catch (const std::exception& e) { setError(e); return KIWIERR_FAIL; }
	}
	int kiwi_pt_close(kiwi_pretokenized_h h) { if (!h) return KIWIERR_INVALID_HANDLE; delete h; return 0; }      // capi.h:1405

	int kiwi_morphset_close(kiwi_morphset_h m)
	{
		if (!m) return KIWIERR_INVALID_HANDLE;
		delete m;
		return 0;
	}

	// ---- typo transformers (capi.h:459-588; src/capi/kiwi_c.cpp:540-715).
	kiwi_typo_h kiwi_typo_init() { try { return new kiwi_typo; } catch (const std::exception& e) { setError(e); return nullptr; } }
	kiwi_typo_h kiwi_typo_get_basic() { return kiwi_typo_get_default(1); }
	kiwi_typo_h kiwi_typo_get_default(int set)
	{
		// the handle of a built-in set lives for the process and must not be closed (capi.h:494-501; the reference returns the address of a static)
		try
		{
			static kiwi_typo* sets[7] = {};
			static std::mutex mu;
			const TypoTransformer& tt = kamd::defaultTypoSet(set);      // throws for ids outside 0..6
			std::lock_guard<std::mutex> g{ mu };
			if (!sets[set]) sets[set] = new kiwi_typo{ tt };
			return sets[set];
		}
		catch (const std::exception& e) { setError(e); return nullptr; }
	}
	int kiwi_typo_add(kiwi_typo_h h, const char** orig, int orig_size, const char** error, int error_size, float cost, int condition)
	{
		if (!h) return KIWIERR_INVALID_HANDLE;
		try
		{
			std::vector<std::u16string> origs, errors;
			for (int i = 0; i < orig_size; ++i) origs.push_back(utf8To16(orig[i], std::strlen(orig[i])));
			for (int i = 0; i < error_size; ++i) errors.push_back(utf8To16(error[i], std::strlen(error[i])));
			for (auto& o : origs) for (auto& e : errors) h->tt.add(o, e, cost, (uint8_t)condition, 0);
			return 0;
		}
		catch (const std::exception& e) { setError(e); return -1; }
	}
	kiwi_typo_h kiwi_typo_copy(kiwi_typo_h h) { if (!h) return nullptr; try { return new kiwi_typo{ *h }; } catch (const std::exception& e) { setError(e); return nullptr; } }
	int kiwi_typo_update(kiwi_typo_h h, kiwi_typo_h src)
	{
		if (!h || !src) return KIWIERR_INVALID_HANDLE;
		try { h->tt.update(src->tt); return 0; } catch (const std::exception& e) { setError(e); return -1; }
	}
	int kiwi_typo_scale_cost(kiwi_typo_h h, float scale)
	{
		if (!h) return KIWIERR_INVALID_HANDLE;
		try { h->tt.scaleCost(scale); return 0; } catch (const std::exception& e) { setError(e); return -1; }
	}
	int kiwi_typo_set_continual_typo_cost(kiwi_typo_h h, float threshold) { if (!h) return KIWIERR_INVALID_HANDLE; h->tt.setContinualCost(threshold); return 0; }
	int kiwi_typo_set_lengthening_typo_cost(kiwi_typo_h h, float threshold) { if (!h) return KIWIERR_INVALID_HANDLE; h->tt.setLengtheningCost(threshold); return 0; }
	int kiwi_typo_close(kiwi_typo_h h) { if (!h) return KIWIERR_INVALID_HANDLE; delete h; return 0; }
	kiwi_prepared_typo_h kiwi_typo_prepare(kiwi_typo_h h)
	{
		if (!h) return nullptr;
		try { return new kiwi_prepared_typo{ kamd::PreparedTypo{ h->tt, true } }; } catch (const std::exception& e) { setError(e); return nullptr; }
	}
	int kiwi_prepared_typo_close(kiwi_prepared_typo_h h) { if (!h) return KIWIERR_INVALID_HANDLE; delete h; return 0; }

	const char* kiwi_version(void) { return "0.23.1+kiwi_amd"; }
	const char* kiwi_error(void) { return hasError ? currentError.c_str() : nullptr; }
	void kiwi_clear_error(void) { hasError = false; currentError.clear(); }

	kiwi_h kiwi_init(const char* model_path, int num_threads, int options, int enabled_dialects)
	{
		try
		{
			// options (capi.h:158-171; kiwi_c.cpp:717-736): bit 0 = integrateAllomorph (KiwiBuilder.cpp:2413); bits 1-3 ask for dictionaries that a raw
			// container already has baked in (or not) -- they cannot change anything here; 0x0F00 = model type
			if (options & ~0x0F0F) throw std::invalid_argument{ "kiwi_amd: unknown build option bits" };
			Engine::LmMode lm; bool knlmUnlessCong = false, largest = false;
			switch (options & 0x0F00)
			{
			// default / largest: the reference looks for cong.mdl first (-> cong / congGlobal), then skipbigram.mdl (-> knlm / sbg), then sj.knlm
			// (KiwiBuilder.cpp:939-961); here: the container's CoNgram blob when it has one -- scored locally by default, with its distant-token
			// (window 7) sections for `largest` --, else Knlm by default and SkipBigram for `largest`
			case 0x0000: lm = Engine::LmMode::Auto; knlmUnlessCong = true; break;
			case 0x0100: lm = Engine::LmMode::Auto; largest = true; break;
			case 0x0200: lm = Engine::LmMode::Knlm; break;
			case 0x0300: lm = Engine::LmMode::Sbg; break;
			case 0x0400: lm = Engine::LmMode::Cong; break;
			case 0x0500: lm = Engine::LmMode::CongGlobal; break;      // (a file without window sections is refused by the engine; the reference reads past the file's sections there)
			default: throw std::invalid_argument{ "kiwi_amd: unknown model type" };
			}
			// enabled_dialects (KIWI_DIALECT_* bits; KiwiBuilder.cpp:963-967): forms of dialects that are not enabled stay out of the dictionary trie (:2500-2504)
			if ((uint32_t)enabled_dialects > 1023u) throw std::invalid_argument{ "kiwi_amd: unknown dialect bits in enabled_dialects" };
			const std::string path = model_path ? model_path : "";      // a directory with sj.morph + sj.knlm (+ skipbigram.mdl) or kiwi_amd.raw, or a raw container file
			auto h = std::make_unique<kiwi_s>();
			h->engine.reset(new Engine(path, -1, lm, (uint32_t)enabled_dialects));      // (-1: the caller's current device)
			if (knlmUnlessCong && !h->engine->usesCong() && h->engine->usesSbg()) h->engine.reset(new Engine(path, -1, Engine::LmMode::Knlm, (uint32_t)enabled_dialects));
			// LARGEST on a cong.mdl is the reference's congGlobal (KiwiBuilder.cpp:939-946); a blob without window sections can only be scored locally
			if (largest && h->engine->usesCong() && h->engine->congWindow()) h->engine.reset(new Engine(path, -1, Engine::LmMode::CongGlobal, (uint32_t)enabled_dialects));
			h->engine->config.integrateAllomorph = !!(options & 1);
			h->numThreads = num_threads < 0 ? 0 : (num_threads == 0 ? 1 : num_threads);
			if (const char* bs = std::getenv("KAMD_CAPI_BATCH")) { const int v = std::atoi(bs); if (v > 0) h->batchSize = v; }      // (developer knob; kiwi_set_option(KIWI_GPU_BATCH_SIZE) is the API)
			return h.release();
		}
		catch (const std::exception& e) { setError(e); return nullptr; }
	}

	int kiwi_close(kiwi_h handle)
	{
		if (!handle) return KIWIERR_INVALID_HANDLE;
		delete handle;
		return 0;
	}

	void kiwi_set_global_config(kiwi_h h, kiwi_config_t c)
	{
		if (!h) return;
		auto& g = h->engine->config;
		g.integrateAllomorph = !!c.integrate_allomorph; g.cutOffThreshold = c.cut_off_threshold; g.oovRuleScale = c.oov_rule_scale; g.oovRuleBias = c.oov_rule_bias; g.oovChrBias = c.oov_chr_bias;
		g.oovGlobalWeight = c.oov_global_weight; g.oovLocalWeight = c.oov_local_weight; g.oovGlobalMinFreq = c.oov_global_min_freq;
		g.spacePenalty = c.space_penalty; g.typoCostWeight = c.typo_cost_weight; g.maxUnkFormSize = c.max_unk_form_size;
		g.maxUnkFormSizeFollowedByJClass = c.max_unk_form_size_followed_by_j_class; g.spaceTolerance = c.space_tolerance;
		for (auto& r : h->replicas) r->config = g;
	}

	kiwi_config_t kiwi_get_global_config(kiwi_h h)
	{
		kiwi_config_t c{};
		if (!h) return c;
		const auto& g = h->engine->config;
		c.integrate_allomorph = g.integrateAllomorph; c.cut_off_threshold = g.cutOffThreshold; c.oov_rule_scale = g.oovRuleScale; c.oov_rule_bias = g.oovRuleBias; c.oov_chr_bias = g.oovChrBias;
		c.oov_global_weight = g.oovGlobalWeight; c.oov_local_weight = g.oovLocalWeight; c.oov_global_min_freq = g.oovGlobalMinFreq;
		c.space_penalty = g.spacePenalty; c.typo_cost_weight = g.typoCostWeight; c.max_unk_form_size = g.maxUnkFormSize;
		c.max_unk_form_size_followed_by_j_class = g.maxUnkFormSizeFollowedByJClass; c.space_tolerance = g.spaceTolerance;
		return c;
	}

	void kiwi_set_option(kiwi_h h, int option, int value)
	{
		if (!h) return;
		if (option == KIWI_NUM_THREADS) h->numThreads = value;
		else if (option == KIWI_GPU_BATCH_SIZE && value > 0) h->batchSize = value;
		else { currentError = "Invalid option value: " + std::to_string(option); hasError = true; }
	}

	// capi.h:644-652; kiwi_c.cpp:826-849: no float option exists any more (deprecated in favour of the global config) -- every id is invalid
	void kiwi_set_option_f(kiwi_h h, int option, float)
	{
		if (!h) return;
		currentError = "Invalid option value: " + std::to_string(option); hasError = true;
	}
	float kiwi_get_option_f(kiwi_h h, int option)
	{
		if (!h) return KIWIERR_INVALID_HANDLE;
		currentError = "Invalid option value: " + std::to_string(option); hasError = true;
		return KIWIERR_INVALID_INDEX;
	}

	int kiwi_get_option(kiwi_h h, int option)
	{
		if (!h) return KIWIERR_INVALID_HANDLE;
		if (option == KIWI_NUM_THREADS) return h->numThreads;
		if (option == KIWI_GPU_BATCH_SIZE) return h->batchSize;
		currentError = "Invalid option value: " + std::to_string(option); hasError = true;
		return KIWIERR_FAIL;
	}

	kiwi_res_h kiwi_analyze_w(kiwi_h h, const kchar16_t* text, int top_n, kiwi_analyze_option_t opt, kiwi_pretokenized_h pt)
	{
		if (!h) return nullptr;
		try
		{
			checkOption(opt, pt);
			size_t n = 0; while (text[n]) ++n;
			if (pt && !pt->spans.empty())
			{
				auto res = h->engine->analyzePretokenized((const char16_t*)text, n, spansOf(*pt, nullptr), (size_t)top_n, (uint64_t)(uint32_t)opt.match_options, !!opt.open_ending, 1, typoOf(opt));
				return makeRes(std::make_shared<const BatchResults>(std::move(res)), 0);
			}
			std::vector<std::pair<const char16_t*, size_t>> v{ { (const char16_t*)text, n } };
			auto res = h->engine->analyzeBatch(v, (size_t)top_n, (uint64_t)(uint32_t)opt.match_options, !!opt.open_ending, 1, typoOf(opt));
			return makeRes(std::make_shared<const BatchResults>(std::move(res)), 0);
		}
		catch (const std::exception& e) { setError(e); return nullptr; }
	}

	kiwi_res_h kiwi_analyze(kiwi_h h, const char* text, int top_n, kiwi_analyze_option_t opt, kiwi_pretokenized_h pt)
	{
		if (!h) return nullptr;
		try
		{
			checkOption(opt, pt);
			const std::u16string u = utf8To16(text, std::strlen(text));
			if (pt && !pt->spans.empty())
			{
				// byte offset of every UTF-16 unit (utf8To16(str, bytePositions), src/StrUtils.h:236-300: both units of a surrogate pair carry the pair's offset)
				std::vector<size_t> bytePos; bytePos.reserve(u.size());
				for (size_t i = 0, n8 = std::strlen(text); i < n8;)
				{
					const uint8_t c = (uint8_t)text[i];
					const size_t len = c < 0x80 ? 1 : (c & 0xE0) == 0xC0 ? 2 : (c & 0xF0) == 0xE0 ? 3 : 4;
					bytePos.push_back(i); if (len == 4) bytePos.push_back(i);
					i += len;
				}
				auto res = h->engine->analyzePretokenized(u.data(), u.size(), spansOf(*pt, &bytePos), (size_t)top_n, (uint64_t)(uint32_t)opt.match_options, !!opt.open_ending, 1, typoOf(opt));
				return makeRes(std::make_shared<const BatchResults>(std::move(res)), 0);
			}
			std::vector<std::pair<const char16_t*, size_t>> v{ { u.data(), u.size() } };
			auto res = h->engine->analyzeBatch(v, (size_t)top_n, (uint64_t)(uint32_t)opt.match_options, !!opt.open_ending, 1, typoOf(opt));
			return makeRes(std::make_shared<const BatchResults>(std::move(res)), 0);
		}
		catch (const std::exception& e) { setError(e); return nullptr; }
	}

	int kiwi_analyze_mw(kiwi_h h, kiwi_reader_w_t reader, kiwi_receiver_t receiver, void* ud, int top_n, kiwi_analyze_option_t opt)
	{
		if (!h) return KIWIERR_INVALID_HANDLE;
		try
		{
			return analyzeMany(h, [&](int idx, std::string&, std::u16string& w, std::vector<size_t>& off, std::vector<size_t>& len)
			{
				const size_t n = (size_t)(*reader)(idx, nullptr, ud), at = w.size();
				if (!n) return false;
				w.resize(at + n);
				(*reader)(idx, (kchar16_t*)&w[at], ud);
				off.back() = at; off.push_back(at + n); len.push_back(n);
				return true;
			}, receiver, ud, top_n, opt);
		}
		catch (const std::exception& e) { setError(e); return KIWIERR_FAIL; }
	}

	int kiwi_analyze_m(kiwi_h h, kiwi_reader_t reader, kiwi_receiver_t receiver, void* ud, int top_n, kiwi_analyze_option_t opt)
	{
		if (!h) return KIWIERR_INVALID_HANDLE;
		try
		{
			return analyzeMany(h, [&](int idx, std::string& raw8, std::u16string&, std::vector<size_t>& off, std::vector<size_t>& len)
			{
				const size_t n = (size_t)(*reader)(idx, nullptr, ud), at = raw8.size();
				if (!n) return false;
				raw8.resize(at + n);
				(*reader)(idx, &raw8[at], ud);      // (UTF-8 -> UTF-16 happens with the batch, on the worker pool)
				off.back() = at; off.push_back(at + n); len.push_back(0);
				return true;
			}, receiver, ud, top_n, opt);
		}
		catch (const std::exception& e) { setError(e); return KIWIERR_FAIL; }
	}

	const char* kiwi_tag_to_string(kiwi_h, uint8_t tag) { return tagToString(tag); }
	const char* kiwi_get_script_name(uint8_t script) { return scriptName(script); }

	int kiwi_res_size(kiwi_res_h r) { return r ? (int)r->size() : KIWIERR_INVALID_HANDLE; }
	float kiwi_res_prob(kiwi_res_h r, int index) { return (r && index >= 0 && (size_t)index < r->size()) ? r->score(index) : 0.f; }
	int kiwi_res_word_num(kiwi_res_h r, int index)
	{
		if (!r) return KIWIERR_INVALID_HANDLE;
		if (index < 0 || (size_t)index >= r->size()) return KIWIERR_INVALID_INDEX;
		return (int)r->tokens(index);
	}
	const kiwi_token_info_t* kiwi_res_token_info(kiwi_res_h r, int index, int num) { return (r && validIdx(r, index, num)) ? reinterpret_cast<const kiwi_token_info_t*>(&r->tok(index, num)) : nullptr; }
	int kiwi_res_morpheme_id(kiwi_res_h r, int index, int num, kiwi_h h)
	{
		if (!r || !h) return KIWIERR_INVALID_HANDLE;
		if (!validIdx(r, index, num)) return KIWIERR_INVALID_INDEX;
		return r->tok(index, num).morph;
	}
	const kchar16_t* kiwi_res_form_w(kiwi_res_h r, int index, int num) { return (r && validIdx(r, index, num)) ? (const kchar16_t*)r->form(index, num) : nullptr; }
	const kchar16_t* kiwi_res_tag_w(kiwi_res_h r, int index, int num)
	{
		if (!r || !validIdx(r, index, num)) return nullptr;
		auto& s = r->tagBufW[{ index, num }];
		if (s.empty()) for (const char* p = tagToString(r->tok(index, num).tag); *p; ++p) s.push_back((char16_t)*p);
		return (const kchar16_t*)s.c_str();
	}
	const char* kiwi_res_form(kiwi_res_h r, int index, int num)
	{
		if (!r || !validIdx(r, index, num)) return nullptr;
		auto it = r->formBuf.find({ index, num });
		if (it == r->formBuf.end()) it = r->formBuf.emplace(std::make_pair(index, num), utf16To8(std::u16string{ r->form(index, num), r->tok(index, num).formLen })).first;
		return it->second.c_str();
	}
	const char* kiwi_res_tag(kiwi_res_h r, int index, int num) { return (r && validIdx(r, index, num)) ? tagToString(r->tok(index, num).tag) : nullptr; }
	int kiwi_res_position(kiwi_res_h r, int index, int num) { return !r ? KIWIERR_INVALID_HANDLE : !validIdx(r, index, num) ? KIWIERR_INVALID_INDEX : (int)r->tok(index, num).position; }
	int kiwi_res_length(kiwi_res_h r, int index, int num) { return !r ? KIWIERR_INVALID_HANDLE : !validIdx(r, index, num) ? KIWIERR_INVALID_INDEX : (int)r->tok(index, num).length; }
	int kiwi_res_word_position(kiwi_res_h r, int index, int num) { return !r ? KIWIERR_INVALID_HANDLE : !validIdx(r, index, num) ? KIWIERR_INVALID_INDEX : (int)r->tok(index, num).wordPosition; }
	int kiwi_res_sent_position(kiwi_res_h r, int index, int num) { return !r ? KIWIERR_INVALID_HANDLE : !validIdx(r, index, num) ? KIWIERR_INVALID_INDEX : (int)r->tok(index, num).sentPosition; }
	float kiwi_res_score(kiwi_res_h r, int index, int num) { return (r && validIdx(r, index, num)) ? r->tok(index, num).score : 0.f; }
	float kiwi_res_typo_cost(kiwi_res_h r, int index, int num) { return (r && validIdx(r, index, num)) ? r->tok(index, num).typoCost : 0.f; }
	int kiwi_res_close(kiwi_res_h r)
	{
		if (!r) return KIWIERR_INVALID_HANDLE;
		if (r->slab)
		{
			const std::shared_ptr<ResSlab> keep = std::move(r->slab);      // (the block outlives the destructor call; it goes with the last handle)
			r->~kiwi_res();
		}
		else delete r;
		return 0;
	}
}
