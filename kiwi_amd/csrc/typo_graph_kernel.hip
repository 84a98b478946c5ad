// HIP kernel: the typo graph of every chunk of a batch, generated on the device --
// PreparedTypoTransformer::generateGraph of the reference (/root/reference/src/TypoTransformer.cpp:811-1049; appendNewNode :594-628),
// without pretokenized spans: Aho-Corasick scan of the chunk over the pattern automaton, clusters of overlapping matches, per cluster the
// unchanged segments between the break points plus one node per admissible replacement (two for the halves of a continual typo), then the
// nodes in end-position order.  The host module (typo.cpp: PreparedTypo::graph, byte-identical to the reference's graphs) is the same
// algorithm over std containers; this is its restatement over fixed per-chunk regions, checked against it node for node
// (the CPU and GPU suites, through kamd_typo_graph_device).
//
// Shape: the build is sequential per chunk (every append asks what already ends at its start position; node ids are handed out in order), so
// it runs on ONE lane per chunk with few active lanes per wave (`stride`, as k_finish_paths / k_build_lattice_typo): chunks spread over all
// SIMDs and a wave's time is one chunk's time.  Two passes of the same code: COUNT = true runs the build without storing nodes and reports how
// many there are (the host sizes the graph, state and index regions of the lattice build from that -- it needs the counts anyway), COUNT =
// false writes them into exactly sized regions.  Working arrays are small per-chunk HBM regions (L2-resident).  Why on the device at all: the graphs were the largest host-side cost of a typo-correcting batch
// (generation on the worker pool, concatenation, one more upload), and host cores are what eight GPUs share.
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdlib>
#include "kchars.hpp"
#include "device_types.hpp"
#include "typo_graph_kernel.hpp"

namespace kamd
{
	namespace
	{
		constexpr uint32_t GNPOS = 0xFFFFFFFFu;

		template<bool COUNT>
		struct GraphCtx
		{
			const TypoGraphTables& T;
			const uint16_t* str; const uint8_t* cls; const uint8_t* script; uint32_t n;
			TypoGraphNode* temp; uint16_t* tlast; uint2* epm; uint32_t cap;
			uint32_t nTemp, last, epmSize; bool overflow;

			// {type, script} of the last character of the text span [off, off + len): the host's forward scan merges a high surrogate with whatever
			// follows it, so the units before the last one decide whether it stands alone -- a run of k high surrogates before it pairs up from
			// its start: k odd = the last unit is the second half of a pair that starts one unit earlier.  The text block's class / script arrays
			// hold the typing of every code point at its first unit (textprep.cpp), NUL = "none".
			__device__ __forceinline__ uint16_t lastOfText(uint32_t off, uint32_t len) const
			{
				if (!len) return 0x00FF;
				const uint32_t j = off + len - 1;
				uint32_t k = 0;
				while (k < len - 1 && isHighSurrogate(str[j - 1 - k])) ++k;
				if (k & 1) return (uint16_t)((cls[j - 1] & 0x7F) | ((uint16_t)script[j - 1] << 8));
				const uint16_t c = str[j];
				if (!c) return 0x00FF;
				if (isHighSurrogate(c)) return (uint16_t)(T.hiType | ((uint16_t)T.hiScript << 8));      // a high surrogate that ends the form stands alone there
				return (uint16_t)((cls[j] & 0x7F) | ((uint16_t)script[j] << 8));
			}

			// appendNewNode (TypoTransformer.cpp:594-628); startPos / endPos GNPOS = "none" (the halves of a continual typo)
			__device__ __forceinline__ bool append(uint32_t formOff, uint32_t formLen, uint32_t startPos, uint32_t endPos, float cost, uint16_t lastInfo)
			{
				if (startPos != GNPOS)
				{
					if (startPos < last || startPos - last >= epmSize) return false;
					if (epm[startPos - last].x == GNPOS) return false;
				}
				const uint32_t newId = nTemp;
				if constexpr (!COUNT)
				{
					if (nTemp >= cap) { overflow = true; return false; }
					TypoGraphNode nn;
					nn.formOff = formOff; nn.formLen = formLen; nn.endPos = endPos; nn.typoCost = cost; nn.siblingOffset = 0;
					nn.continualTypoIdx = 0; nn.pad = 0; nn.dialect = 0;
					nn.prevOffset = startPos == GNPOS ? newId - 1 : epm[startPos - last].x;
					temp[newId] = nn; tlast[newId] = lastInfo;
				}
				nTemp = newId + 1;
				if ((uint64_t)endPos >= (uint64_t)epmSize + last) return true;
				uint2 slot = epm[endPos - last];
				if (slot.x == GNPOS) slot.x = newId;
				else if constexpr (!COUNT) temp[slot.y].siblingOffset = newId;
				slot.y = newId;
				epm[endPos - last] = slot;
				return true;
			}
		};

		__device__ __forceinline__ int32_t tgStep(const TypoGraphTables& T, int32_t node, uint16_t c)
		{
			const PreparedTypo::TrieNode t = T.trie[node];
			const uint16_t* kb = T.keys + t.edgeOff;
			uint32_t lo = 0, hi = t.numNexts;
			while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (kb[mid] < c) lo = mid + 1; else hi = mid; }
			if (lo == t.numNexts || kb[lo] != c) return -1;
			return (int32_t)T.children[t.edgeOff + lo];
		}

		// std::sort of the cluster's matches by pattern start (libstdc++ introsort: the reference sorts with an unstable std::sort, equal starts
		// must land where it puts them -- the order of the replacement nodes depends on it); v[i] = {end, pattern}
		struct MatchLess
		{
			const PreparedTypo::Pattern* pats;
			__device__ __forceinline__ bool operator()(const uint2& a, const uint2& b) const { return a.x - pats[a.y].patLength < b.x - pats[b.y].patLength; }
		};
		__device__ __forceinline__ void insertionSortM(uint2* v, int lo, int hi, bool guarded, const MatchLess& less)
		{
			for (int i = lo; i < hi; ++i)
			{
				const uint2 val = v[i];
				if (guarded && less(val, v[0])) { for (int j = i; j > 0; --j) v[j] = v[j - 1]; v[0] = val; }
				else { int j = i; while (less(val, v[j - 1])) { v[j] = v[j - 1]; --j; } v[j] = val; }
			}
		}
		__device__ void sortMatches(uint2* v, int n, const MatchLess& less)
		{
			if (n <= 16) { insertionSortM(v, 1, n, true, less); return; }
			int stLo[48], stHi[48], stDepth[48]; int sp = 0;
			int depth = 0; for (int t = n; t > 1; t >>= 1) ++depth; depth *= 2;
			stLo[sp] = 0; stHi[sp] = n; stDepth[sp] = depth; ++sp;
			while (sp)
			{
				--sp;
				int lo = stLo[sp], hi = stHi[sp], dl = stDepth[sp];
				while (hi - lo > 16)
				{
					if (dl == 0)
					{
						// std::__partial_sort(first, last, last): make_heap + sort_heap with std::__adjust_heap
						const int len = hi - lo; uint2* a = v + lo;
						auto adjust = [&](int hole, int length, uint2 val)
						{
							const int top = hole;
							int child = hole;
							while (child < (length - 1) / 2)
							{
								child = 2 * (child + 1);
								if (less(a[child], a[child - 1])) --child;
								a[hole] = a[child]; hole = child;
							}
							if ((length & 1) == 0 && child == (length - 2) / 2) { child = 2 * (child + 1); a[hole] = a[child - 1]; hole = child - 1; }
							int parent = (hole - 1) / 2;
							while (hole > top && less(a[parent], val)) { a[hole] = a[parent]; hole = parent; parent = (hole - 1) / 2; }
							a[hole] = val;
						};
						for (int parent = (len - 2) / 2; parent >= 0; --parent) adjust(parent, len, a[parent]);
						for (int lastI = len - 1; lastI > 0; --lastI) { const uint2 val = a[lastI]; a[lastI] = a[0]; adjust(0, lastI, val); }
						break;
					}
					--dl;
					const int first = lo, mid = lo + (hi - lo) / 2, ia = first + 1, ic = hi - 1;
					int med;
					if (less(v[ia], v[mid])) { if (less(v[mid], v[ic])) med = mid; else if (less(v[ia], v[ic])) med = ic; else med = ia; }
					else if (less(v[ia], v[ic])) med = ia; else if (less(v[mid], v[ic])) med = ic; else med = mid;
					{ const uint2 t = v[first]; v[first] = v[med]; v[med] = t; }
					int i = first + 1, j = hi;
					for (;;)
					{
						while (less(v[i], v[first])) ++i;
						--j;
						while (less(v[first], v[j])) --j;
						if (!(i < j)) break;
						const uint2 t = v[i]; v[i] = v[j]; v[j] = t;
						++i;
					}
					stLo[sp] = i; stHi[sp] = hi; stDepth[sp] = dl; ++sp;
					hi = i;
				}
			}
			insertionSortM(v, 1, 16, true, less);
			insertionSortM(v, 16, n, false, less);
		}

		__device__ __forceinline__ bool isSyllableU(uint16_t c) { return 0xAC00 <= c && c < 0xD7A4; }

		// insertBranch (TypoTransformer.cpp:870-1010): the cluster of matches collected so far becomes nodes
		template<bool COUNT>
		__device__ __forceinline__ uint32_t insertBranch(GraphCtx<COUNT>& G, uint2* matches, uint32_t nM, uint32_t* bp, uint32_t allowedDialect, uint32_t maxCti, uint32_t& status)
		{
			const TypoGraphTables& T = G.T;
			const uint32_t totStart = matches[0].x - T.pats[matches[0].y].patLength, totEnd = matches[nM - 1].x;
			const uint2 carry = G.epm[G.epmSize - 1];
			G.epmSize = (totEnd - G.last) + 1;
			for (uint32_t i = 0; i < G.epmSize; ++i) G.epm[i] = make_uint2(GNPOS, GNPOS);
			G.epm[0] = carry;
			// break points: the cluster's start and every distinct match end -- the ends arrive in ascending order and lie beyond the start
			uint32_t nB = 0;
			bp[nB++] = totStart;
			for (uint32_t i = 0; i < nM; ++i) if (matches[i].x != bp[nB - 1]) bp[nB++] = matches[i].x;
			const MatchLess less{ T.pats };
			sortMatches(matches, (int)nM, less);

			if (G.last < totStart) G.append(G.last, totStart - G.last, G.last, totStart, 0.f, COUNT ? (uint16_t)0 : G.lastOfText(G.last, totStart - G.last));
			for (uint32_t i = 1; i < nB; ++i) G.append(bp[i - 1], bp[i] - bp[i - 1], bp[i - 1], bp[i], 0.f, COUNT ? (uint16_t)0 : G.lastOfText(bp[i - 1], bp[i] - bp[i - 1]));

			for (uint32_t mi = 0; mi < nM; ++mi)
			{
				const uint32_t e = matches[mi].x;
				const PreparedTypo::Pattern P = T.pats[matches[mi].y];
				const uint32_t s = e - P.patLength;
				// first replacement character -> (continual index, node of the first half): std::unordered_map of the reference; entries only ever
				// leave it right after they were inserted
				uint16_t ckey[kTypoMaxContinual]; uint32_t cnode[kTypoMaxContinual], cidx[kTypoMaxContinual]; uint32_t nC = 0;
				for (uint32_t ri = 0; ri < P.replCnt; ++ri)
				{
					const PreparedTypo::Repl r = T.repls[P.replOff + ri];
					const uint32_t rOff = r.strOff | TYPO_FORM_IN_POOL;
					const uint8_t* rl = T.replLast + 6 * (size_t)(P.replOff + ri);
					const uint16_t lWhole = (uint16_t)(rl[0] | (rl[1] << 8)), lFirst = (uint16_t)(rl[2] | (rl[3] << 8)), lRest = (uint16_t)(rl[4] | (rl[5] << 8));
					if (r.dialect != 0 && !(allowedDialect & r.dialect)) continue;
					if (r.cond == TC_VOWEL) { if (s == 0 || !isSyllableU(G.str[s - 1])) continue; }
					else if (r.cond == TC_ANY) { if (s == 0) continue; }
					else if (r.cond == TC_CONTINUAL || r.cond == TC_BOUNDARY)
					{
						if (r.cond == TC_CONTINUAL && (s == 0 || !isSyllableU(G.str[s - 1]))) continue;
						if (r.cond == TC_CONTINUAL && !T.continualOn) continue;
						const float scale = r.cond == TC_CONTINUAL ? T.continualCost : 1.f;
						const uint16_t key = T.pool[r.strOff];
						uint32_t k = 0;
						while (k < nC && ckey[k] != key) ++k;
						if (k == nC)
						{
							if (nC >= kTypoMaxContinual) { status = 2; continue; }
							ckey[nC] = key; cidx[nC] = nC + 1; cnode[nC] = 0; ++nC;
							if (G.append(rOff, 1, s, GNPOS, r.cost * scale / 2, lFirst))
							{
								if constexpr (!COUNT) { TypoGraphNode& b = G.temp[G.nTemp - 1]; b.endPos = e; b.continualTypoIdx = (uint8_t)cidx[k]; b.dialect = r.dialect; }
								cnode[k] = G.nTemp - 1;
								if (G.append(rOff + 1, r.strLen - 1, GNPOS, e, r.cost * scale / 2, lRest)) { if constexpr (!COUNT) { TypoGraphNode& h = G.temp[G.nTemp - 1]; h.prevOffset = cnode[k]; h.dialect = r.dialect; } }
							}
							else --nC;
						}
						else if (G.append(rOff + 1, r.strLen - 1, GNPOS, e, r.cost * scale / 2, lRest)) { if constexpr (!COUNT) { TypoGraphNode& h = G.temp[G.nTemp - 1]; h.prevOffset = cnode[k]; h.dialect = r.dialect; } }
						continue;
					}
					else if (!typoLeftCondMatched(G.str, s, r.cond)) continue;
					if (G.append(rOff, r.strLen, s, e, r.cost, lWhole)) { if constexpr (!COUNT) G.temp[G.nTemp - 1].dialect = r.dialect; }
				}
				if (nC + 1 > maxCti) maxCti = nC + 1;
			}
			G.last = totEnd;
			return maxCti;
		}

		template<bool COUNT>
		__global__ void __launch_bounds__(64) k_typo_graph(TypoGraphTables T, TypoGraphView V, uint32_t nChunks, uint32_t stride)
		{
			// write pass with one chunk per wave (stride 64): the build itself runs on lane 0, the final ordering and the output copy on all 64 lanes
			const bool wave = !COUNT && stride == 64;
			const uint32_t lane = threadIdx.x;
			const bool mine = (lane % stride) == 0;
			const uint32_t c = wave ? blockIdx.x : mine ? blockIdx.x * (64 / stride) + lane / stride : 0xFFFFFFFFu;
			if (c >= nChunks) return;      // (wave mode: the whole wave; else: the idle lanes)
			const TypoGraphChunk C = V.chunks[c];
			const uint32_t n = C.nChars;
			uint32_t status = 0, maxCti = 0, nT = 0;
			if (mine) do
			{
				if (C.scrCap < n + 2 || (!COUNT && !C.graphCap)) { status = 1; break; }
				uint16_t* tlast = COUNT ? nullptr : reinterpret_cast<uint16_t*>(V.cnt + C.scrOff);      // (cnt is only needed by the final ordering: until then it holds the last-character facts)
				GraphCtx<COUNT> G{ T, V.chars + C.charOff, V.cls + C.charOff, V.script + C.charOff, n, COUNT ? nullptr : V.temp + C.graphOff, tlast, V.epm + C.scrOff, C.graphCap, 0, 0, 1, false };
				uint2* matches = V.matches + C.scrOff; uint32_t* bp = V.bp + C.scrOff;
				uint32_t nM = 0;
				G.epm[0] = make_uint2(0, 0);
				if constexpr (!COUNT)
				{
					TypoGraphNode first{}; first.formOff = 0; first.formLen = 0; first.endPos = 0; first.typoCost = 0.f; first.prevOffset = 0; first.siblingOffset = 0;
					first.continualTypoIdx = 0; first.pad = 0; first.dialect = 0;
					G.temp[0] = first; G.tlast[0] = 0x00FF;
				}
				G.nTemp = 1;
				int32_t node = T.entryNode;
				if (node < 0) node = 0;
				for (uint32_t i = 0; i <= n; ++i)      // (i == n: the text is over -- the pending cluster is closed through the same call site)
				{
					bool flush = false;
					if (i < n)
					{
						const uint16_t ch = G.str[i];
						int32_t nx = tgStep(T, node, ch);
						while (nx < 0)
						{
							node = T.trie[node].fail;
							if (node >= 0) nx = tgStep(T, node, ch);
							else { node = 0; break; }
						}
						if (nx < 0) continue;
						node = nx;
						const int32_t pat = T.trie[node].pattern;
						if (pat == -1) continue;
						// a node that only carries the "a shorter pattern ends here" mark starts, in the reference's arithmetic, far beyond the text: it closes the pending cluster
						if (nM)
						{
							if (pat < 0) flush = true;
							else
							{
								const uint32_t pl = T.pats[pat].patLength;
								if (pl > i + 1) { status = 3; flush = true; }
								else flush = matches[nM - 1].x < i + 1 - pl;
							}
						}
					}
					else flush = nM != 0;
					if (flush) { maxCti = insertBranch(G, matches, nM, bp, V.allowedDialect, maxCti, status); nM = 0; }
					if (i == n) break;
					const uint32_t endPos = i + 1;
					for (int32_t sub = node; sub >= 0; sub = T.trie[sub].fail)
					{
						const int32_t sp = T.trie[sub].pattern;
						if (sp == -1) break;
						if (sp < 0) continue;
						if (T.pats[sp].patLength > endPos) { status = 3; continue; }
						if (nM + 2 >= C.scrCap) { G.overflow = true; break; }      // (the break points of the cluster: at most its matches + 1)
						matches[nM++] = make_uint2(endPos, (uint32_t)sp);
					}
				}
				{
					const uint2 carry = G.epm[G.epmSize - 1];
					G.epm[0] = carry; G.epmSize = 1;
					G.append(G.last, n - G.last, G.last, n + 1, 0.f, COUNT ? (uint16_t)0 : G.lastOfText(G.last, n - G.last));
					if constexpr (!COUNT) G.temp[G.nTemp - 1].endPos = n;      // (also when the append was refused: the reference sets the end of whatever node is last)
				}
				nT = G.nTemp;
				if (G.overflow) status = 1;
			} while (false);
			if constexpr (COUNT) { if (mine) V.out[c] = TypoGraphOut{ nT, maxCti, status }; return; }
			else
			{
				if (wave) { waveSync(); nT = __shfl(nT, 0); status = __shfl(status, 0); maxCti = __shfl(maxCti, 0); }
				if (status == 1 || nT > C.scrCap) { if (mine) V.out[c] = TypoGraphOut{ nT, maxCti, 1 }; return; }
				// the nodes in end-position order (std::stable_sort by endPos), links re-based to the new indices
				TypoGraphNode* temp = V.temp + C.graphOff; TypoGraphNode* out = V.graph + C.graphOff;
				uint32_t* rev = V.rev + C.scrOff; uint32_t* cnt = V.cnt + C.scrOff; uint32_t* bp = V.bp + C.scrOff;
				const uint16_t* tlast = reinterpret_cast<const uint16_t*>(cnt);
				uint16_t* glast = reinterpret_cast<uint16_t*>(V.graphLast) + C.graphOff;
				if (!wave)
				{
					// one lane: counting sort
					for (uint32_t i = 0; i < nT; ++i) rev[i] = tlast[i];      // (tlast shares cnt's region: out of the way first)
					for (uint32_t i = 0; i <= n + 1; ++i) cnt[i] = 0;
					for (uint32_t i = 0; i < nT; ++i) { const uint32_t e = temp[i].endPos; cnt[(e <= n ? e : n) + 1]++; }
					for (uint32_t i = 1; i <= n + 1; ++i) cnt[i] += cnt[i - 1];
					for (uint32_t i = 0; i < nT; ++i) bp[i] = rev[i];
					for (uint32_t i = 0; i < nT; ++i) { const uint32_t e = temp[i].endPos; rev[i] = cnt[e <= n ? e : n]++; }
					for (uint32_t i = 0; i < nT; ++i)
					{
						TypoGraphNode g = temp[i];
						const uint32_t ni = rev[i];
						g.prevOffset = ni - rev[g.prevOffset];
						if (g.siblingOffset != 0) g.siblingOffset = rev[g.siblingOffset] - ni;
						out[ni] = g;
						glast[ni] = (uint16_t)bp[i];
					}
				}
				else
				{
					// 64 lanes: bp[i] = last-character facts, rev[i] = end position, cnt = histogram -> bucket starts, newIdx[i] = bucket start + number of
					// earlier nodes with the same end (the stable order), then every lane copies its nodes
					uint32_t* newIdx = reinterpret_cast<uint32_t*>(V.matches + C.scrOff);      // (the cluster's matches are history)
					for (uint32_t i = lane; i < nT; i += 64) bp[i] = tlast[i];
					waveSync();
					for (uint32_t i = lane; i <= n + 1; i += 64) cnt[i] = 0;
					waveSync();
					for (uint32_t i = lane; i < nT; i += 64) { uint32_t e = temp[i].endPos; if (e > n) e = n; rev[i] = e; atomicAdd(&cnt[e + 1], 1u); }
					waveSync();
					if (lane == 0) for (uint32_t i = 1; i <= n + 1; ++i) cnt[i] += cnt[i - 1];
					waveSync();
					for (uint32_t i = lane; i < nT; i += 64)
					{
						const uint32_t e = rev[i];
						uint32_t r = cnt[e];
						for (uint32_t j = 0; j < i; ++j) r += rev[j] == e ? 1u : 0u;
						newIdx[i] = r;
					}
					waveSync();
					for (uint32_t i = lane; i < nT; i += 64)
					{
						TypoGraphNode g = temp[i];
						const uint32_t ni = newIdx[i];
						g.prevOffset = ni - newIdx[g.prevOffset];
						if (g.siblingOffset != 0) g.siblingOffset = newIdx[g.siblingOffset] - ni;
						out[ni] = g;
						glast[ni] = (uint16_t)bp[i];
					}
				}
				if (mine) V.out[c] = TypoGraphOut{ nT, maxCti, status };
			}
		}
	}

	void launchTypoGraph(const TypoGraphTables& T, const TypoGraphView& V, uint32_t nChunks, bool countOnly, hipStream_t stream)
	{
		if (!nChunks) return;
		// active lanes per wave: 1 up to 16k chunks, 4 up to 64k, 16 beyond (the machine holds 8192 waves)
		uint32_t stride = nChunks <= 16384 ? 64u : nChunks <= 65536 ? 16u : 4u;
		if (const char* f = std::getenv("KAMD_TYPO_GRAPH_STRIDE")) { const int v = std::atoi(f); if (v == 4 || v == 16 || v == 64) stride = (uint32_t)v; }      // test hook: the several-chunks-per-wave path on small batches
		const uint32_t perWave = 64 / stride;
		if (countOnly) hipLaunchKernelGGL(k_typo_graph<true>, dim3((nChunks + perWave - 1) / perWave), dim3(64), 0, stream, T, V, nChunks, stride);
		else hipLaunchKernelGGL(k_typo_graph<false>, dim3((nChunks + perWave - 1) / perWave), dim3(64), 0, stream, T, V, nChunks, stride);
	}
}
