// Persistent host worker pool for the per-text stages around the kernels (text preparation before, result assembly after).
// The reference fans sentences out over a ThreadPool created once per Kiwi object (/root/reference/include/kiwi/ThreadPool.h:22-92,
// used at include/kiwi/Kiwi.h:402-454); here the per-text host work of a whole batch is cut into blocks that the workers pull
// from an atomic counter.  The pool is created once per process and kept: threads spawned per call cost more than the work of
// an 8k-sentence batch (fresh malloc arenas and first-touch page faults on every call).
// Several callers may run at once -- one per GPU when kiwi_analyze_m drives every visible device from one process -- and SHARE the
// workers: every call registers a job, a worker takes blocks from whichever registered job has some left (and room under its
// thread limit), the caller works on its own job meanwhile and returns when the last of its blocks is done.
#pragma once
#include <algorithm>
#include <atomic>
#include <cstdio>
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
		struct Job
		{
			const std::function<void(size_t, size_t, int)>* fn;      // (begin, end, worker id)
			size_t total, grain, blocks;
			int maxHelpers;                      // pool workers that may work on it at once (the caller is not counted)
			std::atomic<size_t> next{ 0 };       // first item of the next block to hand out
			std::atomic<size_t> done{ 0 };       // blocks finished
			std::atomic<int> helpers{ 0 };
			std::atomic<bool> failed{ false };
			std::exception_ptr error;
		};
		std::vector<std::thread> threads;
		std::mutex mu;
		std::condition_variable cvWork, cvDone;
		std::vector<Job*> jobs;      // registered jobs (under mu)
		size_t turn = 0;
		bool stopping = false;
		const pid_t owner = ::getpid();

		// blocks of `j` until none is left; true if this call finished the job's last block
		bool drain(Job& j, int id)
		{
			bool last = false;
			for (;;)
			{
				const size_t i = j.next.fetch_add(j.grain);
				if (i >= j.total) break;
				if (!j.failed.load(std::memory_order_relaxed))
				{
					try { (*j.fn)(i, std::min(j.total, i + j.grain), id); }
					catch (...) { if (!j.failed.exchange(true)) { std::lock_guard<std::mutex> g{ mu }; j.error = std::current_exception(); } }
				}
				if (j.done.fetch_add(1) + 1 == j.blocks) last = true;
			}
			return last;
		}

		void workerMain(int id)
		{
			std::unique_lock<std::mutex> lk{ mu };
			for (;;)
			{
				Job* pick = nullptr;
				for (size_t t = 0; t < jobs.size() && !pick; ++t)      // (round robin over the registered jobs: concurrent callers share the workers evenly)
				{
					Job* j = jobs[(turn + t) % jobs.size()];
					if (j->next.load(std::memory_order_relaxed) < j->total && j->helpers.load(std::memory_order_relaxed) < j->maxHelpers) pick = j;
				}
				++turn;
				if (!pick)
				{
					if (stopping) return;
					cvWork.wait(lk);
					continue;
				}
				pick->helpers.fetch_add(1);
				lk.unlock();
				const bool last = drain(*pick, id + 1);
				lk.lock();
				pick->helpers.fetch_sub(1);      // (under mu: the submitter waits for helpers == 0 before its Job goes out of scope)
				if (last || pick->helpers.load() == 0) cvDone.notify_all();
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
		// them; worker ids are 0 .. size()-1, 0 = the caller).  Calls from several threads proceed side by side; an exception of any block is rethrown.
		void run(size_t n, size_t block, int maxThreads, const std::function<void(size_t, size_t, int)>& fn)
		{
			if (!n) return;
			if (block < 1) block = 1;
			const size_t blocks = (n + block - 1) / block;
			int helpers = (int)std::min<size_t>(threads.size(), blocks - 1);
			if (maxThreads > 0) helpers = std::min(helpers, maxThreads - 1);
			if (helpers <= 0 || ::getpid() != owner) { fn(0, n, 0); return; }
			Job j;
			j.fn = &fn; j.total = n; j.grain = block; j.blocks = blocks; j.maxHelpers = helpers;
			{ std::lock_guard<std::mutex> g{ mu }; jobs.push_back(&j); }
			cvWork.notify_all();
			drain(j, 0);
			std::unique_lock<std::mutex> lk{ mu };
			cvDone.wait(lk, [&] { return j.done.load() == j.blocks && j.helpers.load() == 0; });
			jobs.erase(std::find(jobs.begin(), jobs.end(), &j));
			lk.unlock();
			if (j.error) std::rethrow_exception(j.error);
		}

		// the process-wide pool: one thread per hardware thread, at most two per CPU of a CFS quota (KAMD_HOST_THREADS overrides; the batch stages take their own `maxThreads` on top).
		// A child process after fork() has none of the parent's threads: it runs its stages on the calling thread alone.
		static HostPool& instance()
		{
			static HostPool pool{ defaultThreads() - 1 };
			return pool;
		}
		static int defaultThreads()
		{
			unsigned n = std::max(1u, std::thread::hardware_concurrency());
			// A container can see every logical CPU of its host and be scheduled on a fraction of them (CFS quota).  A worker per visible CPU then costs
			// more than it brings, and what the workers burn beyond the work itself counts against the quota: once that is used up the whole process
			// is stopped until the next period (cpu.stat nr_throttled: stages of 1 - 3 ms stretched to 20 - 40 ms a few times per second).  MI355X box,
			// 256 logical CPUs under a quota of 16, a 65 536-sentence batch end to end: 22.3 ms with 256 workers, 16.6 - 19.5 with 64, 15.4 - 16.9 with 32,
			// 22.1 with 16 (profiles/r04_r_*, r04_t_*): two workers per CPU of the quota.
			const double quota = cpuQuota();
			if (quota > 0) n = std::min(n, (unsigned)std::max(1.0, 2.0 * quota + 0.5));
			if (const char* e = std::getenv("KAMD_HOST_THREADS")) { const long v = std::atol(e); if (v > 0) n = (unsigned)v; }
			return (int)std::min(1024u, n);
		}
		// the content of a cgroup v2 cpu.max file -- "<quota> <period>" in microseconds, or "max <period>" for no limit -- as CPUs; 0 = no limit / not understood
		static double parseCpuMax(const char* text)
		{
			long long q = 0, p = 0;
			if (std::sscanf(text, "%lld %lld", &q, &p) == 2 && q > 0 && p > 0) return (double)q / (double)p;
			return 0;
		}
		// CPUs' worth of run time per period this process' control group may use (cgroup v2 cpu.max, v1 cpu.cfs_quota_us / cpu.cfs_period_us); 0 = no limit known
		static double cpuQuota()
		{
			auto readAll = [](const char* path, char* buf, size_t cap) -> bool
			{
				FILE* f = std::fopen(path, "r");
				if (!f) return false;
				const size_t k = std::fread(buf, 1, cap - 1, f);
				std::fclose(f);
				buf[k] = 0;
				return k > 0;
			};
			char buf[128];
			if (readAll("/sys/fs/cgroup/cpu.max", buf, sizeof(buf))) return parseCpuMax(buf);
			long long q = 0, p = 0;
			if (readAll("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", buf, sizeof(buf))) q = std::atoll(buf);
			if (readAll("/sys/fs/cgroup/cpu/cpu.cfs_period_us", buf, sizeof(buf))) p = std::atoll(buf);
			return (q > 0 && p > 0) ? (double)q / (double)p : 0;
		}
	};
}
