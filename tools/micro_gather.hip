// Micro-benchmark (round 5): how a wavefront should fetch 64-byte hash buckets at random addresses -- every lane its own bucket with four 16-byte loads
// (what lmProgressChain does), or four adjacent lanes one bucket with one 16-byte load each (the lanes of a quad then touch one 64-byte line).
// hipcc --offload-arch=gfx950 -O3 gather.hip -o gather && ./gather
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__device__ __forceinline__ uint32_t mix(uint32_t h) { h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 13; h *= 0x9E3779B1u; h ^= h >> 16; return h; }
// A: lane = probe; 4 x 16 B per lane
__global__ void __launch_bounds__(64) k_lane(const uint4* tab, uint32_t mask, uint32_t rounds, uint32_t* out)
{
	uint32_t h = mix(blockIdx.x * 64u + threadIdx.x + 1u), acc = 0;
	for (uint32_t r = 0; r < rounds; ++r)
	{
		const uint4* b = tab + (size_t)(h & mask) * 4;
		const uint4 s0 = b[0], s1 = b[1], s2 = b[2], s3 = b[3];
		acc += (s0.x == h) + (s1.x == h) + (s2.x == h) + (s3.x == h) + s0.y;
		h = mix(h + acc);      // (the next address depends on the loaded data: one round trip per round, as in the search)
	}
	out[blockIdx.x * 64u + threadIdx.x] = acc;
}
// B: quad = probe; the 64 probes of a round are fetched in 4 instructions of 16 probes each; results handed back through __shfl
__global__ void __launch_bounds__(64) k_quad(const uint4* tab, uint32_t mask, uint32_t rounds, uint32_t* out)
{
	const uint32_t lane = threadIdx.x;
	uint32_t h = mix(blockIdx.x * 64u + lane + 1u), acc = 0;
	for (uint32_t r = 0; r < rounds; ++r)
	{
		uint32_t got = 0, y0 = 0;
		uint4 v[4];
#pragma unroll
		for (uint32_t it = 0; it < 4; ++it)
		{
			const uint32_t src = it * 16u + (lane >> 2);            // the probe this lane helps with
			const uint32_t hs = __shfl(h, (int)src, 64);
			v[it] = tab[(size_t)(hs & mask) * 4 + (lane & 3u)];
		}
#pragma unroll
		for (uint32_t it = 0; it < 4; ++it)
		{
			const uint32_t src = it * 16u + (lane >> 2);
			const uint32_t hs = __shfl(h, (int)src, 64);
			uint32_t m = (v[it].x == hs) ? 1u : 0u;
			m += __shfl_xor(m, 1, 64); m += __shfl_xor(m, 2, 64);   // matches of the quad
			const uint32_t y = __shfl(v[it].y, (int)(lane & ~3u), 64);
			// hand back to the probe's lane: lane `src` takes what quad (src % 16) of iteration (src / 16) found
			const uint32_t mm = __shfl(m, (int)((lane & 15u) * 4u), 64), yy = __shfl(y, (int)((lane & 15u) * 4u), 64);
			if ((lane >> 4) == it) { got = mm; y0 = yy; }
		}
		acc += got + y0;
		h = mix(h + acc);
	}
	out[blockIdx.x * 64u + lane] = acc;
}
int main()
{
	for (int big = 0; big < 2; ++big)
	{
		const size_t nBuckets = big ? (size_t(1) << 23) : (size_t(1) << 15);      // 512 MB (HBM) / 2 MB (L2)
		std::vector<uint4> h(nBuckets * 4);
		for (size_t i = 0; i < h.size(); ++i) h[i] = make_uint4((uint32_t)(i * 2654435761u), (uint32_t)(i & 3), 0, 0);
		uint4* d; uint32_t* out; CK(hipMalloc(&d, h.size() * 16)); CK(hipMalloc(&out, 4u * 64 * 65536));
		CK(hipMemcpy(d, h.data(), h.size() * 16, hipMemcpyHostToDevice));
		hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
		for (int blocks : { 3072, 16384 })
		{
			const uint32_t rounds = 200;
			for (int variant = 0; variant < 2; ++variant)
			{
				float best = 1e9f;
				for (int rep = 0; rep < 4; ++rep)
				{
					CK(hipEventRecord(a));
					if (variant == 0) hipLaunchKernelGGL(k_lane, dim3(blocks), dim3(64), 0, 0, d, (uint32_t)(nBuckets - 1), rounds, out);
					else hipLaunchKernelGGL(k_quad, dim3(blocks), dim3(64), 0, 0, d, (uint32_t)(nBuckets - 1), rounds, out);
					CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
					float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
				}
				const double probes = (double)blocks * 64 * rounds;
				printf("table %s, %5d one-wave blocks, %s: %.3f ms, %.1f G probes/s, %.0f GB/s of buckets\n", big ? "512 MB" : "2 MB", blocks, variant ? "quad per bucket (1 x 16 B per lane)" : "lane per bucket (4 x 16 B per lane)", best, probes / best * 1e-6, probes * 64 / best * 1e-6);
			}
		}
		CK(hipFree(d)); CK(hipFree(out));
	}
	return 0;
}
