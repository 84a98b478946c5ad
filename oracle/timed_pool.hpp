// TEST INFRASTRUCTURE (CPU baseline of bench.py): a sound multi-thread timing of "analyse this corpus".
// The worker threads exist BEFORE the clock starts (their thread_local scratch -- the reference's path containers, allocator pools -- is warm after
// one untimed pass over the corpus), the timed region is whole passes over the corpus, repeated until at least `minSeconds` of wall time have
// elapsed, texts are handed out through one atomic counter (the reference's own pool hands out one text per task and delivers in input order,
// include/kiwi/Kiwi.h:402-454; the results are dropped here, which only helps the baseline).
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace timedpool
{
	// fn(threadIndex, textIndex) analyses one text.  Returns wall seconds of the timed passes; *passesOut = how many.
	template<class Fn>
	double run(int threads, uint32_t nTexts, double minSeconds, uint32_t* passesOut, Fn&& fn)
	{
		if (threads < 1) threads = 1;
		std::mutex mu; std::condition_variable cvGo, cvDone;
		uint64_t generation = 0; int running = 0; bool quit = false;
		std::atomic<uint32_t> next{ 0 };
		auto worker = [&](int tid)
		{
			uint64_t seen = 0;
			for (;;)
			{
				{
					std::unique_lock<std::mutex> lk(mu);
					cvGo.wait(lk, [&] { return quit || generation != seen; });
					if (quit) return;
					seen = generation;
				}
				for (;;) { const uint32_t i = next.fetch_add(1, std::memory_order_relaxed); if (i >= nTexts) break; fn(tid, i); }
				{
					std::lock_guard<std::mutex> lk(mu);
					if (--running == 0) cvDone.notify_one();
				}
			}
		};
		std::vector<std::thread> ts;
		for (int t = 0; t < threads; ++t) ts.emplace_back(worker, t);
		auto onePass = [&]()
		{
			std::unique_lock<std::mutex> lk(mu);
			next.store(0); running = threads; ++generation;
			cvGo.notify_all();
			cvDone.wait(lk, [&] { return running == 0; });
		};
		onePass();      // untimed: warms every thread
		uint32_t passes = 0;
		const auto t0 = std::chrono::steady_clock::now();
		double el = 0;
		do { onePass(); ++passes; el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); } while (el < minSeconds);
		{ std::lock_guard<std::mutex> lk(mu); quit = true; }
		cvGo.notify_all();
		for (auto& t : ts) t.join();
		if (passesOut) *passesOut = passes;
		return el;
	}
}
