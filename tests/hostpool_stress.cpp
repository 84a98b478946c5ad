// CPU test program (tests/test_hostpool.py): eight threads call HostPool::run at once, with and without a thread limit, some jobs throwing --
// every sum must come out right and every exception must reach its own caller.  Built with and without ThreadSanitizer.
#include "hostpool.hpp"
#include <cstdio>
#include <numeric>
int main()
{
	using namespace kamd;
	HostPool& p = HostPool::instance();
	// many concurrent callers, each summing a range through the pool; checks totals, exceptions, nested sizes
	std::atomic<long long> bad{ 0 };
	std::vector<std::thread> callers;
	for (int c = 0; c < 8; ++c) callers.emplace_back([&, c]
	{
		for (int rep = 0; rep < 200; ++rep)
		{
			const size_t n = 1000 + 37 * c + rep;
			std::atomic<long long> sum{ 0 };
			p.run(n, 16, (c % 3 == 0) ? 2 : 0, [&](size_t a, size_t b, int) { long long s = 0; for (size_t i = a; i < b; ++i) s += (long long)i; sum += s; });
			if (sum.load() != (long long)n * (n - 1) / 2) ++bad;
			if (rep % 50 == 0)
			{
				bool thrown = false;
				try { p.run(500, 8, 0, [&](size_t a, size_t, int) { if (a == 248) throw std::runtime_error("x"); }); } catch (const std::runtime_error&) { thrown = true; }
				if (!thrown) ++bad;
			}
		}
	});
	for (auto& t : callers) t.join();
	printf("bad %lld\n", bad.load());
	return bad.load() != 0;
}
