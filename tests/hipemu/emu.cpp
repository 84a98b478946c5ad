// TEST INFRASTRUCTURE -- the lane scheduler behind tests/hipemu/include/hip/hip_runtime.h (read its header first).
// One block at a time; every lane of the block is a ucontext fiber; a cross-lane operation is a rendezvous of the running lanes of
// a convergence domain.  Also defines the dynamic-LDS arrays the kernels declare `extern __shared__`.
#include <hip/hip_runtime.h>
#include <ucontext.h>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

// Sanitizer builds (`make asan`): AddressSanitizer has to be told about every stack switch
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define HIPEMU_ASAN 1
extern "C" void __sanitizer_start_switch_fiber(void** fakeStackSave, const void* bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void* fakeStackSave, const void** bottomOld, size_t* sizeOld);
extern "C" void __asan_poison_memory_region(void const volatile* addr, size_t size);
extern "C" void __asan_unpoison_memory_region(void const volatile* addr, size_t size);
#endif
#endif

// ThreadSanitizer builds (`make tsan`; this file itself is compiled WITHOUT the sanitizer, the kernels with it): every lane is a TSan
// fiber, switched without implied synchronisation; a rendezvous is release (on arrival) + acquire (on leaving) on its domain; a kernel
// launch is ordered after everything the launching thread did before and before everything it does afterwards.  Two lanes touching the
// same LDS / HBM location with no rendezvous in between, at least one of them writing non-atomically, is then a reported data race --
// the missing-barrier bug that on the GPU shows up as a rare wrong answer.
#ifdef HIPEMU_TSAN
extern "C" void* __tsan_get_current_fiber(void);
extern "C" void* __tsan_create_fiber(unsigned flags);
extern "C" void __tsan_destroy_fiber(void* fiber);
extern "C" void __tsan_switch_to_fiber(void* fiber, unsigned flags);
extern "C" void __tsan_acquire(void* addr);
extern "C" void __tsan_release(void* addr);
#define TSAN_ONLY(...) __VA_ARGS__
#else
#define TSAN_ONLY(...)
#endif

namespace kamd
{
	// dynamic LDS of the product's kernels (lattice_kernels.hip, viterbi_kernel.hip and its SkipBigram compilation): blocks run one
	// after another, so one buffer of the hardware's size per declaration is enough
	alignas(16) uint8_t lSmem[160 * 1024];
	alignas(16) uint8_t kSmem[160 * 1024];
	alignas(16) uint8_t tSmem[160 * 1024];      // typo_lattice_kernel.hip
	namespace sbgk { alignas(16) uint8_t kSmem[160 * 1024]; }
	namespace typok { alignas(16) uint8_t kSmem[160 * 1024]; namespace congk { alignas(16) uint8_t kSmem[160 * 1024]; } namespace sbgk { alignas(16) uint8_t kSmem[160 * 1024]; } }
	namespace congk { alignas(16) uint8_t kSmem[160 * 1024]; namespace gk { alignas(16) uint8_t kSmem[160 * 1024]; } }
	namespace typok { namespace congk { namespace gk { alignas(16) uint8_t kSmem[160 * 1024]; } } }
}

namespace hipemu
{
	namespace
	{
		constexpr uint32_t MAXT = 1024;
		constexpr size_t STACK = 512 * 1024;
		struct Domain { uint32_t arrived = 0; uint64_t gen = 0; Op op = (Op)0; uint64_t active[2] = { 0, 0 }; };
		struct Lane
		{
			ucontext_t uc; bool finished = true; LaneCtx ctx;
			void* tsanFiber = nullptr;
			bool waiting = false; Op waitOp = (Op)0; uint32_t waitBase = 0, waitSize = 0; uint64_t waitGen = 0;
		};
		struct Block
		{
			std::vector<Lane> lanes; uint32_t n = 0, width = 64;
			Domain groupDom[MAXT];      // convergence domains of the kernel's width, by first lane / width (no allocation inside fibers)
			Domain blockDom;            // __syncthreads
			uint64_t vals[2][MAXT];
			uint64_t valsBlock[2][MAXT];      // payloads of block-wide rendezvous: a kernel may interleave them with group-wide ones, whose generations are independent
			uint64_t progress = 0;
			const std::function<void()>* body = nullptr;
			ucontext_t sched; int cur = -1;
			const char* kernel = ""; bool dropBarrier = false;
			const void* schedStack = nullptr; size_t schedStackSize = 0;     // (AddressSanitizer builds: the launching thread's stack)
			void* schedFiber = nullptr;                                       // (ThreadSanitizer builds)
		};
		char streamOrder;      // ThreadSanitizer builds: the sync object that stands for stream order (host -> kernel -> host)
		Block* B = nullptr;
		std::vector<char*> stacks;
		std::mutex launchMu;

		// lane -> scheduler
		void yieldToScheduler(Lane& l, bool dying)
		{
#ifdef HIPEMU_ASAN
			void* fake = nullptr;
			__sanitizer_start_switch_fiber(dying ? nullptr : &fake, B->schedStack, B->schedStackSize);
#endif
			TSAN_ONLY(if (dying) __tsan_release(&streamOrder); __tsan_switch_to_fiber(B->schedFiber, 1 /* no implied synchronisation */);)
			swapcontext(&l.uc, &B->sched);
#ifdef HIPEMU_ASAN
			__sanitizer_finish_switch_fiber(fake, &B->schedStack, &B->schedStackSize);
#endif
			(void)dying;
		}
		void trampoline()
		{
#ifdef HIPEMU_ASAN
			__sanitizer_finish_switch_fiber(nullptr, &B->schedStack, &B->schedStackSize);
#endif
			TSAN_ONLY(__tsan_acquire(&streamOrder);)
			(*B->body)();
			Lane& l = B->lanes[B->cur];
			l.finished = true; ++B->progress;
			yieldToScheduler(l, true);
		}
		uint32_t running(uint32_t base, uint32_t size, uint64_t* mask)
		{
			uint32_t c = 0; uint64_t m = 0;
			for (uint32_t i = 0; i < size && base + i < B->n; ++i) if (!B->lanes[base + i].finished) { ++c; if (i < 64) m |= 1ull << i; }
			if (mask) *mask = m;
			return c;
		}
		const char* opName(Op o)
		{
			switch (o) { case OP_BALLOT: return "__ballot"; case OP_SHFL: return "__shfl"; case OP_SHFL_UP: return "__shfl_up"; case OP_SHFL_XOR: return "__shfl_xor";
			case OP_DPP: return "dpp row_ror"; case OP_WAVE_BARRIER: return "wave_barrier"; case OP_SYNCTHREADS: return "__syncthreads"; default: return "?"; }
		}
		[[noreturn]] void die(const char* why)
		{
			std::fprintf(stderr, "hipemu: %s in kernel %s, block %u\n", why, B->kernel, B->lanes[0].ctx.bid.x);
			for (uint32_t i = 0; i < B->n; ++i)
			{
				const Lane& l = B->lanes[i];
				if (l.finished) continue;
				std::fprintf(stderr, "  lane %3u: %s", i, l.waiting ? "waits at " : "runnable");
				if (l.waiting) std::fprintf(stderr, "%s (domain %u+%u, generation %llu)", opName(l.waitOp), l.waitBase, l.waitSize, (unsigned long long)l.waitGen);
				std::fprintf(stderr, "\n");
			}
			std::abort();
		}
	}

	LaneCtx& ctx() { return B->lanes[B->cur].ctx; }
	uint32_t width() { return B->width; }
	void setWidth(uint32_t w) { if (w == 0 || w > 64 || (w & (w - 1))) die("unsupported convergence width"); B->width = w; }      // (between two block-wide rendezvous: no lane is inside a group-wide one)
	bool dropWaveBarrier() { return B->dropBarrier; }

	const uint64_t* exchange(Op op, uint32_t domain, uint64_t mine, uint64_t* active, uint32_t* domainBase)
	{
		const uint32_t self = (uint32_t)B->cur;
		const uint32_t size = domain ? domain : B->width;
		const uint32_t base = self / size * size;
		if (size != B->width && size != B->n) die("unsupported rendezvous size");
		Domain& d = (domain == 0) ? B->groupDom[base / size] : B->blockDom;
		Lane& l = B->lanes[self];
		if (d.arrived == 0) d.op = op;
		else if (d.op != op)
		{
			l.waiting = true; l.waitOp = op; l.waitBase = base; l.waitSize = size; l.waitGen = d.gen;
			die("lanes of one convergence domain arrived at different cross-lane operations (divergent control flow around a cross-lane operation)");
		}
		const uint64_t g = d.gen;
		TSAN_ONLY(__tsan_release(&d);)
		(domain == 0 ? B->vals : B->valsBlock)[g & 1][self] = mine;
		++d.arrived; ++B->progress;
		l.waiting = true; l.waitOp = op; l.waitBase = base; l.waitSize = size; l.waitGen = g;
		for (;;)
		{
			if (d.gen != g) break;                                   // completed by another lane
			uint64_t mask;
			if (d.arrived == running(base, size, &mask)) { d.active[g & 1] = mask; d.arrived = 0; ++d.gen; ++B->progress; break; }
			yieldToScheduler(l, false);
		}
		l.waiting = false;
		TSAN_ONLY(__tsan_acquire(&d);)
		*active = d.active[g & 1]; *domainBase = base;
		return (domain == 0 ? B->vals : B->valsBlock)[g & 1];
	}

	void launch(const char* name, dim3 grid, dim3 block, size_t ldsBytes, const std::function<void()>& laneBody)
	{
		std::lock_guard<std::mutex> lk{ launchMu };
		if (block.x > MAXT || block.y != 1 || block.z != 1 || grid.y != 1 || grid.z != 1 || ldsBytes > 160 * 1024) { std::fprintf(stderr, "hipemu: unsupported launch shape of %s\n", name); std::abort(); }
		// convergence width: the lane-group width of k_best_path<G, ...> / k_pos_path<G, ...> / k_build_lattice<G>, else the wavefront
		uint32_t width = 64;
		const std::string nm{ name };
		const size_t at = nm.find("k_best_path<"), atPos = nm.find("k_pos_path<"), atLat = nm.find("k_build_lattice<");
		if (at != std::string::npos) width = (uint32_t)std::atoi(nm.c_str() + at + 12);
		else if (atPos != std::string::npos) width = (uint32_t)std::atoi(nm.c_str() + atPos + 11);
		else if (atLat != std::string::npos) width = (uint32_t)std::atoi(nm.c_str() + atLat + 16);
		if (width == 0 || width > 64 || (width & (width - 1))) { std::fprintf(stderr, "hipemu: cannot read the lane-group width out of '%s'\n", name); std::abort(); }
		Block blk; blk.n = block.x; blk.width = width; blk.body = &laneBody; blk.kernel = name;
		if (const char* drop = std::getenv("HIPEMU_TEST_DROP_WAVE_BARRIER")) blk.dropBarrier = nm.find(drop) != std::string::npos;
		blk.lanes.resize(block.x);
		while (stacks.size() < block.x) stacks.push_back((char*)std::malloc(STACK));
#ifdef HIPEMU_ASAN
		// dynamic LDS beyond what the launch asked for does not exist
		for (uint8_t* a : { kamd::lSmem, kamd::kSmem, kamd::tSmem, kamd::sbgk::kSmem, kamd::typok::kSmem, kamd::congk::kSmem, kamd::typok::congk::kSmem, kamd::typok::sbgk::kSmem, kamd::congk::gk::kSmem, kamd::typok::congk::gk::kSmem })
		{
			__asan_unpoison_memory_region(a, 160 * 1024);
			__asan_poison_memory_region(a + ((ldsBytes + 7) & ~(size_t)7), 160 * 1024 - ((ldsBytes + 7) & ~(size_t)7));
		}
#endif
		TSAN_ONLY(blk.schedFiber = __tsan_get_current_fiber(); __tsan_release(&streamOrder);)
		Block* outer = B;      // (a kernel never launches a kernel; kept for symmetry)
		for (uint32_t b = 0; b < grid.x; ++b)
		{
			B = &blk;
			for (auto& d : blk.groupDom) d = Domain{};
			blk.blockDom = Domain{}; blk.progress = 0;
			for (uint32_t t = 0; t < block.x; ++t)
			{
				Lane& l = blk.lanes[t];
				l.finished = false; l.waiting = false;
				l.ctx.tid = dim3(t); l.ctx.bid = dim3(b); l.ctx.bdim = block; l.ctx.gdim = grid;
				TSAN_ONLY(l.tsanFiber = __tsan_create_fiber(0);)
				getcontext(&l.uc);
				l.uc.uc_stack.ss_sp = stacks[t]; l.uc.uc_stack.ss_size = STACK; l.uc.uc_link = nullptr;
				makecontext(&l.uc, trampoline, 0);
			}
			for (;;)
			{
				const uint64_t before = blk.progress;
				bool any = false;
				for (uint32_t t = 0; t < block.x; ++t)
				{
					if (blk.lanes[t].finished) continue;
					any = true; blk.cur = (int)t;
#ifdef HIPEMU_ASAN
					void* fake = nullptr;
					__sanitizer_start_switch_fiber(&fake, stacks[t], STACK);
#endif
					TSAN_ONLY(__tsan_switch_to_fiber(blk.lanes[t].tsanFiber, 1);)
					swapcontext(&blk.sched, &blk.lanes[t].uc);
#ifdef HIPEMU_ASAN
					__sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#endif
				}
				if (!any) break;
				if (blk.progress == before) die("no lane can make progress (a cross-lane operation some lanes of the domain never reach)");
			}
			TSAN_ONLY(for (uint32_t t = 0; t < block.x; ++t) __tsan_destroy_fiber(blk.lanes[t].tsanFiber);)
		}
		B = outer;
		TSAN_ONLY(__tsan_acquire(&streamOrder);)
	}
}
