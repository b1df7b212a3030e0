// tools/flagbench.hip — what does a round boundary cost on an MI355X? (round 5; measurement tool, not part of the product)
//
// The small grid (1024x256: one round of 256 blocks per launch) pays ~15 % of a launch for the kernel boundary. The proposed alternative:
// PERSISTENT blocks (cooperative launch: co-residency guaranteed or the launch fails) that stay on their tile for several launches' worth
// of levels and meet their eight neighbours through device-scope flags in global memory. This tool prices that boundary with no work
// attached, next to the kernel boundary it would replace:
//   flags    every block publishes "round r done" (release) and waits for its eight neighbours' words (acquire), `rounds` times:
//            mode 0 relaxed words only; mode 1 with the agent-scope release / acquire fences a real exchange needs (L2 write-back and
//            invalidate on gfx950: the L2s of the eight XCDs are not coherent with each other); mode 2 like 1 plus `bytes` of stores
//            per block and round in front of the release (the tile a block would hand over); mode 3 like 1 with ONE wave per block
//            executing the acquire (the cheapest correct form)
//   kernels  the same number of empty launches of the same grid back to back on one stream
// Every spin is BOUNDED (s_memrealtime, 100 MHz): a block that waits longer than `limit_us` writes its id to an error word, publishes a
// POISON value so that its neighbours leave too, and exits; `--withhold B R` makes block B publish nothing from round R on to show it.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++20 -o tools/flagbench tools/flagbench.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int POISON = 0x7fffffff;

__device__ __forceinline__ unsigned long long realtime() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

struct FlagArgs {
    int* flags;             // one word per block: epoch0 + rounds completed
    int* err;               // first block that ran into the bound (+1), 0 = none
    double* sink;           // mode 2: where the stand-in stores go (bytes per block and round)
    int nbx, nby, rounds, epoch0, mode, bytes;
    unsigned long long limit_ticks;
    int withhold_block, withhold_round;
};

__global__ void __launch_bounds__(1024) k_flag_rounds(const FlagArgs p) {
    const int b = (int)blockIdx.x, by = b / p.nbx, bx = b - by * p.nbx;
    __shared__ int leave;
    if (threadIdx.x == 0) leave = 0;
    __syncthreads();
    for (int r = 0; r < p.rounds; ++r) {
        if (p.mode == 2) {
            const int n = p.bytes / 8;
            double* q = p.sink + (size_t)b * n;
            for (int k = threadIdx.x; k < n; k += blockDim.x) q[k] = (double)(r + k);
        }
        __syncthreads();                                              // every thread's stores are issued (and, with the barrier's fence, performed)
        if (threadIdx.x == 0 && !(b == p.withhold_block && r >= p.withhold_round)) {      // (--withhold B R: block B publishes nothing from round R on)
            if (p.mode >= 1) __atomic_store_n(&p.flags[b], p.epoch0 + r + 1, __ATOMIC_RELEASE);        // agent-scope release: write-back of this XCD's L2
            else __hip_atomic_store(&p.flags[b], p.epoch0 + r + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (threadIdx.x < 8) {
            static constexpr int dx[8] = {-1, 0, 1, -1, 1, -1, 0, 1}, dy[8] = {-1, -1, -1, 0, 0, 1, 1, 1};
            const int nx_ = bx + dx[threadIdx.x], ny_ = by + dy[threadIdx.x];
            if (nx_ >= 0 && nx_ < p.nbx && ny_ >= 0 && ny_ < p.nby) {
                const int n = ny_ * p.nbx + nx_, want = p.epoch0 + r + 1;
                const unsigned long long t0 = realtime();
                for (;;) {
                    const int v = __hip_atomic_load(&p.flags[n], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (v == POISON) { leave = 1; break; }
                    if (v >= want) break;
                    if (realtime() - t0 > p.limit_ticks) { atomicCAS(p.err, 0, b + 1); leave = 1; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
        }
        __syncthreads();
        if (leave) {                                                  // block-uniform: leave, and tell the neighbours
            if (threadIdx.x == 0) __hip_atomic_store(&p.flags[b], POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        if (p.mode == 3) { if (threadIdx.x < 64) __atomic_thread_fence(__ATOMIC_ACQUIRE); __syncthreads(); }      // ONE wave invalidates for the block
        else if (p.mode >= 1) __atomic_thread_fence(__ATOMIC_ACQUIRE);     // agent-scope acquire: what the neighbours wrote is re-read from memory
    }
}

__global__ void __launch_bounds__(1024) k_empty(int* p) { if (p && threadIdx.x == 4096) *p = 1; }

int main(int argc, char** argv) {
    int nbx = 32, nby = 8, rounds = 200, reps = 5, bytes = 73728, withhold_block = -1, withhold_round = -1;
    double limit_us = 200000.0;
    for (int i = 1; i < argc; ++i) {
        std::string k = argv[i];
        if (k == "--nbx") nbx = atoi(argv[++i]);
        else if (k == "--nby") nby = atoi(argv[++i]);
        else if (k == "--rounds") rounds = atoi(argv[++i]);
        else if (k == "--reps") reps = atoi(argv[++i]);
        else if (k == "--bytes") bytes = atoi(argv[++i]);
        else if (k == "--limit-us") limit_us = atof(argv[++i]);
        else if (k == "--withhold") { withhold_block = atoi(argv[++i]); withhold_round = atoi(argv[++i]); }
    }
    const int nb = nbx * nby;
    int *flags, *err;
    double* sink;
    CK(hipMalloc(&flags, nb * sizeof(int)));
    CK(hipMalloc(&err, sizeof(int)));
    CK(hipMalloc(&sink, (size_t)nb * bytes));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int max_blocks = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&max_blocks, k_flag_rounds, 1024, 0));
    hipDeviceProp_t pr;
    CK(hipGetDeviceProperties(&pr, 0));
    printf("%s: %d CUs, %d resident blocks of 1024 threads per CU, cooperative launch %s; grid %d x %d = %d blocks\n", pr.name, pr.multiProcessorCount, max_blocks,
           pr.cooperativeLaunch ? "supported" : "NOT supported", nbx, nby, nb);
    for (int mode = 0; mode <= 3; ++mode) {
        std::vector<double> us;
        int errv = 0;
        for (int rep = 0; rep < reps + 1; ++rep) {
            CK(hipMemsetAsync(flags, 0, nb * sizeof(int), s));
            CK(hipMemsetAsync(err, 0, sizeof(int), s));
            FlagArgs a{flags, err, sink, nbx, nby, rounds, 0, mode, bytes, (unsigned long long)(limit_us * 100.0), withhold_block, withhold_round};
            void* args[] = {&a};
            CK(hipEventRecord(e0, s));
            const hipError_t le = hipLaunchCooperativeKernel((const void*)k_flag_rounds, dim3(nb), dim3(1024), args, 0, s);
            if (le != hipSuccess) { printf("mode %d: cooperative launch refused: %s\n", mode, hipGetErrorString(le)); (void)hipGetLastError(); break; }
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            CK(hipMemcpy(&errv, err, sizeof(int), hipMemcpyDeviceToHost));
            if (rep) us.push_back(ms * 1e3 / rounds);
            if (errv) { printf("mode %d: block %d ran into the spin bound of %.0f us; the grid drained in %.1f us\n", mode, errv - 1, limit_us, ms * 1e3); break; }
        }
        if (!us.empty() && !errv) {
            std::sort(us.begin(), us.end());
            printf("flags mode %d (%s): %.2f us per round (median of %d; best %.2f)\n", mode,
                   mode == 0 ? "relaxed words" : mode == 1 ? "release / acquire fences, every wave" : mode == 2 ? "fences + stores" : "release by one thread, acquire by one wave", us[us.size() / 2], (int)us.size(), us[0]);
        }
        if (errv) break;
    }
    {
        std::vector<double> us;
        for (int rep = 0; rep < reps + 1; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(k_empty, dim3(nb), dim3(1024), 0, s, (int*)nullptr);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) us.push_back(ms * 1e3 / rounds);
        }
        std::sort(us.begin(), us.end());
        printf("kernel boundaries: %.2f us per empty launch of %d x 1024 threads back to back (median; best %.2f)\n", us[us.size() / 2], nb, us[0]);
    }
    return 0;
}
