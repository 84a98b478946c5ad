// tests/test_hostpool.py: the worker pool's reading of the container's CPU quota, and the pool itself (blocks handed out once each, exceptions rethrown,
// concurrent callers sharing the workers).
#include <atomic>
#include <cstdio>
#include <stdexcept>
#include <thread>
#include <vector>
#include "hostpool.hpp"
using kamd::HostPool;
int main()
{
	int bad = 0;
	auto near = [](double a, double b) { return a > b - 1e-9 && a < b + 1e-9; };
	bad += !near(HostPool::parseCpuMax("1600000 100000\n"), 16.0);
	bad += !near(HostPool::parseCpuMax("50000 100000"), 0.5);
	bad += !near(HostPool::parseCpuMax("max 100000\n"), 0.0);
	bad += !near(HostPool::parseCpuMax(""), 0.0);
	bad += !near(HostPool::parseCpuMax("-1 100000"), 0.0);
	bad += !near(HostPool::parseCpuMax("garbage"), 0.0);
	bad += HostPool::defaultThreads() < 1;
	HostPool pool{ 7 };
	// every item exactly once, whatever the block size and the thread limit
	for (size_t n : { (size_t)1, (size_t)63, (size_t)1000, (size_t)65536 })
		for (size_t block : { (size_t)1, (size_t)17, (size_t)512 })
			for (int maxThreads : { 0, 1, 3 })
			{
				std::vector<std::atomic<int>> hit(n);
				for (auto& h : hit) h = 0;
				pool.run(n, block, maxThreads, [&](size_t a, size_t b, int) { for (size_t i = a; i < b; ++i) hit[i]++; });
				for (auto& h : hit) bad += h != 1;
			}
	// an exception of one block reaches the caller; the pool stays usable
	bool thrown = false;
	try { pool.run(1000, 10, 0, [&](size_t a, size_t, int) { if (a == 500) throw std::runtime_error{ "x" }; }); }
	catch (const std::runtime_error&) { thrown = true; }
	bad += !thrown;
	// four callers at once share the workers
	std::atomic<long> sum{ 0 };
	std::vector<std::thread> callers;
	for (int c = 0; c < 4; ++c) callers.emplace_back([&] { for (int r = 0; r < 20; ++r) pool.run(5000, 64, 0, [&](size_t a, size_t b, int) { sum += (long)(b - a); }); });
	for (auto& t : callers) t.join();
	bad += sum != 4L * 20 * 5000;
	printf("%d\n", bad);
	return bad != 0;
}
