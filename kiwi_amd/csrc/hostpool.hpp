// Persistent host worker pool for the per-text stages around the kernels (text preparation before, result assembly after).
// The reference fans sentences out over a ThreadPool created once per Kiwi object (/root/reference/include/kiwi/ThreadPool.h:22-92,
// used at include/kiwi/Kiwi.h:402-454); here the per-text host work of a whole batch is cut into blocks that the workers pull
// from an atomic counter.  The pool is created once per process and kept: threads spawned per call cost more than the work of
// an 8k-sentence batch (fresh malloc arenas and first-touch page faults on every call).
#pragma once
#include <atomic>
#include <cstdlib>
#include <unistd.h>
#include <condition_variable>
#include <exception>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace kamd
{
	class HostPool
	{
		std::vector<std::thread> threads;
		std::mutex mu, runMu;
		std::condition_variable cvWork, cvDone;
		const std::function<void(size_t, size_t, int)>* job = nullptr;   // (begin, end, worker id)
		std::atomic<size_t> next{ 0 };
		size_t total = 0, grain = 1;
		uint64_t generation = 0;
		int wanted = 0, running = 0;
		bool stopping = false;
		const pid_t owner = ::getpid();
		std::exception_ptr error;
		std::atomic<bool> failed{ false };

		void drain(int id)
		{
			try
			{
				for (;;)
				{
					const size_t i = next.fetch_add(grain);
					if (i >= total || failed.load(std::memory_order_relaxed)) break;
					(*job)(i, std::min(total, i + grain), id);
				}
			}
			catch (...)
			{
				if (!failed.exchange(true)) { std::lock_guard<std::mutex> g{ mu }; error = std::current_exception(); }
			}
		}

		void workerMain(int id)
		{
			uint64_t seen = 0;
			std::unique_lock<std::mutex> lk{ mu };
			for (;;)
			{
				cvWork.wait(lk, [&] { return stopping || (generation != seen && id < wanted); });
				if (stopping) return;
				seen = generation;
				lk.unlock();
				drain(id + 1);
				lk.lock();
				if (--running == 0) cvDone.notify_all();
			}
		}

	public:
		explicit HostPool(int n)
		{
			for (int i = 0; i < n; ++i) threads.emplace_back([this, i] { workerMain(i); });
		}
		~HostPool()
		{
			if (::getpid() != owner) { for (auto& t : threads) t.detach(); return; }      // (a forked child: the workers never existed here)
			{ std::lock_guard<std::mutex> g{ mu }; stopping = true; }
			cvWork.notify_all();
			for (auto& t : threads) t.join();
		}
		int size() const { return (int)threads.size() + 1; }

		// Runs fn(begin, end, worker) over [0, n) in blocks of `block` items on up to `maxThreads` threads (the caller is one of
		// them; worker ids are 0 .. size()-1).  Calls from several threads are serialised; an exception of any block is rethrown.
		void run(size_t n, size_t block, int maxThreads, const std::function<void(size_t, size_t, int)>& fn)
		{
			if (!n) return;
			if (block < 1) block = 1;
			const size_t blocks = (n + block - 1) / block;
			int helpers = (int)std::min<size_t>(threads.size(), blocks - 1);
			if (maxThreads > 0) helpers = std::min(helpers, maxThreads - 1);
			if (helpers <= 0 || ::getpid() != owner) { fn(0, n, 0); return; }
			std::lock_guard<std::mutex> serial{ runMu };
			{
				std::lock_guard<std::mutex> g{ mu };
				job = &fn; total = n; grain = block; next = 0; failed = false; error = nullptr;
				wanted = helpers; running = helpers; ++generation;
			}
			cvWork.notify_all();
			drain(0);
			std::unique_lock<std::mutex> lk{ mu };
			cvDone.wait(lk, [&] { return running == 0; });
			job = nullptr;
			if (error) { auto e = error; error = nullptr; std::rethrow_exception(e); }
		}

		// the process-wide pool: one thread per hardware thread (KAMD_HOST_THREADS overrides; the batch stages take their own `maxThreads` on top).
		// A child process after fork() has none of the parent's threads: it runs its stages on the calling thread alone.
		static HostPool& instance()
		{
			static HostPool pool{ defaultThreads() - 1 };
			return pool;
		}
		static int defaultThreads()
		{
			unsigned n = std::max(1u, std::thread::hardware_concurrency());
			if (const char* e = std::getenv("KAMD_HOST_THREADS")) { const long v = std::atol(e); if (v > 0) n = (unsigned)v; }
			return (int)std::min(1024u, n);
		}
	};
}
