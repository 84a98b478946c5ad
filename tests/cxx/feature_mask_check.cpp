// TEST INFRASTRUCTURE: featMaskFast (the kernels' single-pass left-feature mask, kiwi_amd/csrc/feature.hpp) against featMask (the thirteen
// predicates of /root/reference/src/FeatureTestor.cpp:6-78 it folds) -- every single unit of the Hangul blocks, random strings of them.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../kiwi_amd/csrc/feature.hpp"
using namespace kamd;
int main()
{
	std::vector<uint16_t> units;
	for (uint32_t c = 0x1100; c < 0x1200; ++c) units.push_back((uint16_t)c);
	for (uint32_t c = 0xAC00; c < 0xD7B0; ++c) units.push_back((uint16_t)c);
	for (uint32_t c : { 0x20u, 0x41u, 0x30u, 0x2Eu, 0x3131u, 0xFFFFu, 0xD800u, 0xDC00u, 0x119Eu, 0xD7A4u, 0xD7A3u, 0xABFFu, 0x11A7u, 0x11C3u }) units.push_back((uint16_t)c);
	size_t bad = 0, tot = 0;
	uint16_t buf[8];
	if (featMask(buf, 0) != featMaskFast(buf, 0)) ++bad;
	for (uint16_t c : units) { buf[0] = c; ++tot; if (featMask(buf, 1) != featMaskFast(buf, 1)) ++bad; }
	srand(7);
	for (int it = 0; it < 2000000; ++it)
	{
		const uint32_t n = 1 + rand() % 6;
		for (uint32_t i = 0; i < n; ++i)
		{
			const int r = rand() % 10;
			buf[i] = r < 6 ? units[rand() % units.size()] : r < 8 ? (uint16_t)(0x11A8 + rand() % 27) : (uint16_t)(0xAC00 + rand() % 11172);
		}
		++tot;
		if (featMask(buf, n) != featMaskFast(buf, n)) ++bad;
	}
	printf("checked %zu bad %zu\n", tot, bad);
	return bad != 0;
}
