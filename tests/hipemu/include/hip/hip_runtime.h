// TEST INFRASTRUCTURE -- a stand-in for <hip/hip_runtime.h> that lets the product's HIP sources (kiwi_amd/csrc/*.hip) be
// compiled as plain C++ for the host and their kernels be EXECUTED on the CPU, lane by lane, for the parity tests that run
// without a GPU (tests/test_hipemu.py).  It is not a CPU fallback of the product: the product library (libkiwi_hip.so) is
// built from the same sources with hipcc and refuses to open without a HIP device; nothing under kiwi_amd/ knows this
// directory exists; the emulated library (tests/hipemu/_build/libkiwi_hipemu.so) is only ever loaded by tests that name it.
//
// What is emulated is the *execution model the kernels rely on*, not the hardware:
//   * every lane of a block is a fiber (ucontext) running the kernel function; blocks run one after another;
//   * cross-lane operations (__ballot, __shfl*, DPP row rotations, wave barrier, __syncthreads) are rendezvous points: a lane
//     blocks until every still-running lane of its *convergence domain* has arrived at the same kind of operation, then all
//     of them see each other's operands.  The domain is the group of lanes the kernel keeps in uniform control flow: the
//     lane-group width G of k_best_path<G, .> (its 64/G groups work on different chunks and diverge freely), the whole
//     wavefront (64) for every other kernel; __syncthreads spans the block;
//   * a lane outside the domain is what an inactive lane is to the hardware: absent from ballots, and a DPP read from it
//     returns the `old` operand;
//   * if no lane of a block can make progress (lanes of one domain waiting at different operations, i.e. a cross-lane
//     operation in divergent control flow) the emulator aborts and names the lanes and the operations they wait at.
// Memory is the host's, sequentially consistent: a data race between lanes that the hardware would expose only through
// timing is NOT detected, except that lanes really do run far apart between rendezvous points, so a missing barrier shows.
#pragma once
#define __HIPCC__ 1
#define KAMD_HIPEMU 1      // (kernels that need a whole-wavefront operation beside their lane-group convergence width ask for hipemu_ballot64 / hipemu_shfl64)
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <algorithm>
#include <functional>

#define __host__
#define __device__
#define __global__
#define __shared__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __align__(n) __attribute__((aligned(n)))

struct alignas(16) uint4 { uint32_t x, y, z, w; };
struct alignas(8) uint2 { uint32_t x, y; };
inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{ x, y, z, w }; }
inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{ x, y }; }
struct dim3 { uint32_t x, y, z; dim3(uint32_t x_ = 1, uint32_t y_ = 1, uint32_t z_ = 1) : x(x_), y(y_), z(z_) {} };

namespace hipemu
{
	struct LaneCtx { dim3 tid, bid, bdim, gdim; };
	LaneCtx& ctx();                         // of the running lane
	enum Op : int { OP_BALLOT = 1, OP_SHFL, OP_SHFL_UP, OP_SHFL_XOR, OP_DPP, OP_WAVE_BARRIER, OP_SYNCTHREADS };
	// Rendezvous of the running lane with the other running lanes of its domain (`domain` lanes, aligned; 0 = the kernel's
	// convergence width).  Returns the payloads of all lanes of the block, indexed by lane-in-block (valid until the lane's
	// next rendezvous); *active gets the mask of participating lanes relative to the domain's first lane.
	const uint64_t* exchange(Op op, uint32_t domain, uint64_t mine, uint64_t* active, uint32_t* domainBase);
	uint32_t width();                        // convergence width of the running kernel
	void launch(const char* name, dim3 grid, dim3 block, size_t ldsBytes, const std::function<void()>& laneBody);
	template<class T> inline uint64_t pack(T v) { static_assert(sizeof(T) <= 8, "payload"); uint64_t u = 0; std::memcpy(&u, &v, sizeof(T)); return u; }
	template<class T> inline T unpack(uint64_t u) { T v; std::memcpy(&v, &u, sizeof(T)); return v; }
}

#define threadIdx (hipemu::ctx().tid)
#define blockIdx (hipemu::ctx().bid)
#define blockDim (hipemu::ctx().bdim)
#define gridDim (hipemu::ctx().gdim)

// ---- cross-lane operations -------------------------------------------------------------------------------------
inline unsigned long long __ballot(int pred)
{
	uint64_t active; uint32_t base;
	const uint64_t* v = hipemu::exchange(hipemu::OP_BALLOT, 0, pred ? 1 : 0, &active, &base);
	const uint32_t w = hipemu::width(), wave = (threadIdx.x & ~63u);
	unsigned long long r = 0;
	for (uint32_t i = 0; i < w; ++i) if (((active >> i) & 1) && v[base + i]) r |= 1ull << ((base + i) - wave);
	return r;
}
template<class T> inline T __shfl(T var, int srcLane, int width = 64)
{
	uint64_t active; uint32_t base;
	const uint64_t* v = hipemu::exchange(hipemu::OP_SHFL, 0, hipemu::pack(var), &active, &base);
	const uint32_t self = threadIdx.x, seg = self & ~(uint32_t)(width - 1), src = seg + ((uint32_t)srcLane & (uint32_t)(width - 1));
	if (src < base || src >= base + hipemu::width() || !((active >> (src - base)) & 1)) return var;   // (hardware: undefined; kernels never read an inactive lane)
	return hipemu::unpack<T>(v[src]);
}
template<class T> inline T __shfl_up(T var, unsigned delta, int width = 64)
{
	uint64_t active; uint32_t base;
	const uint64_t* v = hipemu::exchange(hipemu::OP_SHFL_UP, 0, hipemu::pack(var), &active, &base);
	const uint32_t self = threadIdx.x, pos = self & (uint32_t)(width - 1);
	if (pos < delta) return var;
	const uint32_t src = self - delta;
	if (src < base || !((active >> (src - base)) & 1)) return var;
	return hipemu::unpack<T>(v[src]);
}
template<class T> inline T __shfl_xor(T var, int laneMask, int width = 64)
{
	uint64_t active; uint32_t base;
	const uint64_t* v = hipemu::exchange(hipemu::OP_SHFL_XOR, 0, hipemu::pack(var), &active, &base);
	const uint32_t self = threadIdx.x, src = self ^ (uint32_t)laneMask;
	if ((src & ~(uint32_t)(width - 1)) != (self & ~(uint32_t)(width - 1))) return var;
	if (src < base || src >= base + hipemu::width() || !((active >> (src - base)) & 1)) return var;
	return hipemu::unpack<T>(v[src]);
}
// DPP: only the row rotations (ctrl 0x121..0x12F, full row / bank masks) the kernels use
inline int hipemu_update_dpp(int old, int src, int ctrl, int rowMask, int bankMask, bool boundCtrl)
{
	if (ctrl < 0x121 || ctrl > 0x12F || rowMask != 0xF || bankMask != 0xF || boundCtrl) { std::abort(); }
	uint64_t active; uint32_t base;
	const uint64_t* v = hipemu::exchange(hipemu::OP_DPP, 0, hipemu::pack(src), &active, &base);
	const uint32_t self = threadIdx.x, n = (uint32_t)ctrl - 0x120u, from = (self & ~15u) | ((self - n) & 15u);   // row_ror:n -- lane i reads lane (i - n) mod 16 of its row
	if (from < base || from >= base + hipemu::width() || !((active >> (from - base)) & 1)) return old;
	return hipemu::unpack<int>(v[from]);
}
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) hipemu_update_dpp((old), (src), (ctrl), (rm), (bm), (bc))
// ds_permute_b32 (forward / "push" permute): every lane sends `data` to lane addr / 4 of its wavefront; a lane nobody writes to receives 0, of
// several writers the highest lane wins
inline int hipemu_ds_permute(int addr, int data)
{
	uint64_t active; uint32_t base;
	const uint64_t* v = hipemu::exchange(hipemu::OP_SHFL, 0, ((uint64_t)(uint32_t)addr << 32) | (uint32_t)data, &active, &base);
	const uint32_t self = threadIdx.x, wave = self & ~63u, w = hipemu::width();
	int r = 0;
	for (uint32_t i = 0; i < w; ++i)
	{
		if (!((active >> i) & 1)) continue;
		const uint64_t pv = v[base + i];
		if (wave + (((uint32_t)(pv >> 32) >> 2) & 63u) == self) r = (int)(uint32_t)pv;
	}
	return r;
}
#define __builtin_amdgcn_ds_permute(addr, data) hipemu_ds_permute((addr), (data))
// whole-wavefront ballot / shuffle for kernels whose convergence width is a lane GROUP but which also have phases all 64 lanes run together
// (k_pos_path: items of four chunks packed into one wavefront): a rendezvous of the block (one wavefront per block in those kernels)
inline unsigned long long hipemu_ballot64(int pred)
{
	uint64_t active; uint32_t base;
	const uint64_t* v = hipemu::exchange(hipemu::OP_BALLOT, blockDim.x, pred ? 1 : 0, &active, &base);
	unsigned long long r = 0;
	for (uint32_t i = 0; i < blockDim.x && i < 64; ++i) if (((active >> i) & 1) && v[base + i]) r |= 1ull << i;
	return r;
}
template<class T> inline T hipemu_shfl64(T var, int srcLane)
{
	uint64_t active; uint32_t base;
	const uint64_t* v = hipemu::exchange(hipemu::OP_SHFL, blockDim.x, hipemu::pack(var), &active, &base);
	const uint32_t src = (uint32_t)srcLane & 63u;
	if (src >= blockDim.x || !((active >> src) & 1)) return var;
	return hipemu::unpack<T>(v[base + src]);
}
// row_ror:n over the 16-lane rows of the whole wavefront (a kernel whose convergence width is narrower than a DPP row, in a phase all 64 lanes run together:
// k_pos_path<8, .> packs the items of its eight lane groups into the four rows); an absent lane reads as 0
inline int hipemu_row_ror64(int src, int n)
{
	uint64_t active; uint32_t base;
	const uint64_t* v = hipemu::exchange(hipemu::OP_DPP, blockDim.x, hipemu::pack(src), &active, &base);
	const uint32_t self = threadIdx.x, from = (self & ~15u) | ((self - (uint32_t)n) & 15u);
	if (from >= blockDim.x || !((active >> from) & 1)) return 0;
	return hipemu::unpack<int>(v[base + from]);
}
// the convergence width of the running kernel changes (all lanes of the block call this together): k_pos_path<8, .> carries chunks on in the general
// search, whose lane groups are 16 wide
namespace hipemu { void setWidth(uint32_t w); }
inline void hipemu_set_width(uint32_t w) { uint64_t a; uint32_t b; (void)hipemu::exchange(hipemu::OP_SYNCTHREADS, blockDim.x, 0, &a, &b); hipemu::setWidth(w); (void)hipemu::exchange(hipemu::OP_SYNCTHREADS, blockDim.x, 0, &a, &b); }
#define __builtin_amdgcn_fence(order, scope) ((void)0)
// HIPEMU_TEST_DROP_WAVE_BARRIER=<kernel name> turns the wave barrier of that kernel into nothing: the self-test of the race detector
// (the ThreadSanitizer build must then report the races the barrier exists to prevent; tests/test_hipemu.py)
namespace hipemu { bool dropWaveBarrier(); }
inline void hipemu_wave_barrier() { if (hipemu::dropWaveBarrier()) return; uint64_t a; uint32_t b; (void)hipemu::exchange(hipemu::OP_WAVE_BARRIER, 0, 0, &a, &b); }
#define __builtin_amdgcn_wave_barrier() hipemu_wave_barrier()
inline void __syncthreads() { uint64_t a; uint32_t b; (void)hipemu::exchange(hipemu::OP_SYNCTHREADS, blockDim.x, 0, &a, &b); }

// ---- scalar device functions ---------------------------------------------------------------------------------------
inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
inline int __popc(unsigned v) { return __builtin_popcount(v); }
inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
inline int __ffs(int v) { return __builtin_ffs(v); }
inline int __clz(unsigned v) { return v ? __builtin_clz(v) : 32; }
inline uint32_t __float_as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
inline float __uint_as_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
// device atomics: relaxed (as on the GPU they order nothing else), real atomics so that the ThreadSanitizer build sees them as such
template<class T> inline T atomicAdd(T* p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
template<class T> inline T atomicOr(T* p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_RELAXED); }
template<class T> inline T atomicCAS(T* p, T cmp, T v) { __atomic_compare_exchange_n(p, &cmp, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED); return cmp; }
template<class T> inline T atomicExch(T* p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
template<class T> inline T atomicMax(T* p, T v) { T o = __atomic_load_n(p, __ATOMIC_RELAXED); while (v > o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }
template<class T> inline T atomicMin(T* p, T v) { T o = __atomic_load_n(p, __ATOMIC_RELAXED); while (v < o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {} return o; }
inline unsigned long long wall_clock64() { return 0; }
inline unsigned long long clock64() { return 0; }
using std::min;
using std::max;

// ---- runtime API (synchronous: every launch has finished when the call returns) ------------------------------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorNotReady = 600, hipErrorInvalidValue = 1, hipErrorOutOfMemory = 2 };
typedef struct hipemuStream* hipStream_t;
typedef struct hipemuEvent* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipStreamNonBlocking = 1 };
struct hipDeviceProp_t { char name[256]; int multiProcessorCount; size_t totalGlobalMem; };
inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { const char* e = std::getenv("HIPEMU_DEVICES"); *n = e ? std::max(1, std::atoi(e)) : 1; return hipSuccess; }      // (device ids are ignored: every "device" is the host)
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { std::memset(p, 0, sizeof(*p)); std::strcpy(p->name, "hipemu (CPU lanes)"); p->multiProcessorCount = 1; p->totalGlobalMem = 1ull << 34; return hipSuccess; }
template<class T> inline hipError_t hipMalloc(T** p, size_t n) { *p = (T*)std::aligned_alloc(256, (n + 255) & ~(size_t)255); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
template<class T> inline hipError_t hipHostMalloc(T** p, size_t n, unsigned = 0) { *p = (T*)std::aligned_alloc(256, (n + 255) & ~(size_t)255); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }      // (the emulated LDS is the hardware's 160 KB for every kernel)
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t = nullptr) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (hipStream_t)std::malloc(8); return hipSuccess; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (hipStream_t)std::malloc(8); return hipSuccess; }
inline hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 0; *greatest = -1; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t s) { std::free(s); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
inline double hipemu_now_ms() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = (hipEvent_t)std::malloc(8); *(double*)*e = 0; return hipSuccess; }
constexpr unsigned hipEventDisableTiming = 2;
constexpr unsigned hipEventBlockingSync = 1;
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = (hipEvent_t)std::malloc(8); *(double*)*e = 0; return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { std::free(e); return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { *(double*)e = hipemu_now_ms(); return hipSuccess; }      // (everything before it has finished)
inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(*(double*)b - *(double*)a); return hipSuccess; }      // host wall clock of the emulated work

// the kernel expression is evaluated by every lane (a call through the function the product names); arguments are read-only views
#define hipLaunchKernelGGL(kernel, grid, block, lds, stream, ...) \
	hipemu::launch(#kernel, dim3(grid), dim3(block), (size_t)(lds), [&]() { (kernel)(__VA_ARGS__); })
