// LDS-DMA fill-rate probe (round 4): how many bytes per clock and CU does `buffer_load_dwordx4 ... lds` deliver, as a function of
//   ROWB   : contiguous bytes a row contributes to one 1 KiB piece (64 = the BK = 32 layout of linear_pw v1-v3, 128 = full cache lines)
//   source : all CUs read one small region (L2-hot, shared: the W operand), each CU re-reads its own 160 KB region (L2-hot, private),
//            every CU streams fresh lines (HBM: the A operand)
//   depth  : pieces a wave keeps in flight (counted vmcnt)
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 -o /tmp/dma_rate tools/probe/dma_rate.hip && /tmp/dma_rate
// Output: one line per configuration, GB/s aggregate and B/clk/CU at the measured wall time (2.4 GHz nominal is NOT assumed: the
// kernel reads s_memtime for its own cycle count).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// Each workgroup (8 waves) performs `iters` rounds; in a round every wave issues PER pieces of 1 KiB.  A piece covers 1024 / ROWB rows
// of ROWB contiguous bytes; consecutive rows are `pitch` bytes apart (a [rows][K] matrix slice), like the operand tiles of a GEMM.
template <int ROWB, int DEPTH>
__global__ __launch_bounds__(512) void dma_kernel(const char* base, unsigned bytes, unsigned region_bytes, unsigned pitch, int iters, int mode,
                                                  unsigned long long* cycles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(base), 0, bytes, 0x00020000);
    constexpr int LPR = ROWB / 16;             // lanes per row
    constexpr int RPP = 64 / LPR;              // rows per piece
    constexpr int PER = 4;                     // pieces per wave and round
    // region of this workgroup
    unsigned rbase = 0;
    if (mode == 1) rbase = blockIdx.x * region_bytes;              // private, re-read
    if (mode == 2) rbase = blockIdx.x * region_bytes;              // streaming: region_bytes = everything this workgroup reads
    const unsigned rows_in_region = region_bytes / pitch;          // rows of `pitch` bytes
    const unsigned long long t0 = __builtin_readcyclecounter();
    unsigned row = wave * RPP * PER, koff = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const unsigned rr = (row + q * RPP + lane / LPR) % rows_in_region;
            const unsigned off = rbase + rr * pitch + koff + (lane % LPR) * 16;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + ((it & 3) * 8 + wave) * 4096 + q * 1024), 16, off, 0, 0, 0);
        }
        row += 8 * RPP * PER;
        if (row >= rows_in_region) { row -= rows_in_region; koff += ROWB; if (koff >= pitch) koff = 0; }
        if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else if (DEPTH == 4) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    if (smem[threadIdx.x] == 123 && iters < 0) cycles[0] = 1;     // keep LDS live
}

template <int ROWB, int DEPTH> void run(const char* name, const char* buf, size_t bytes, unsigned region, unsigned pitch, int mode, int iters,
                                        unsigned long long* dcyc) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(dma_kernel<ROWB, DEPTH>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    for (int rep = 0; rep < 2; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((dma_kernel<ROWB, DEPTH>), dim3(256), dim3(512), 131072, 0, buf, (unsigned)bytes, region, pitch, iters, mode, dcyc);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
    }
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned long long> c(256);
    CHECK(hipMemcpy(c.data(), dcyc, 256 * 8, hipMemcpyDeviceToHost));
    double avg = 0;
    for (auto v : c) avg += (double)v;
    avg /= 256;
    const double total = 256.0 * iters * 8 * 4 * 1024;
    printf("%-28s rowB %3d depth %d : %7.1f us  %6.2f TB/s  %5.1f B/clk/CU (cycle counter %.0f per WG: %5.1f B/cyc)\n", name, ROWB, DEPTH, ms * 1e3,
           total / ms / 1e9, total / 256 / (ms * 1e-3 * 2.1e9), avg, (double)iters * 32768 / avg);
}

int main() {
    const size_t bytes = (size_t)3 << 30;   // 3 GB (32-bit buffer offsets)
    char* buf;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMemset(buf, 1, bytes));
    unsigned long long* dcyc;
    CHECK(hipMalloc(&dcyc, 256 * 8));
    const int iters = 2000;                 // 2000 x 32 KB = 64 MB per workgroup, 16 GB in all (streaming wraps inside its region)
    // W-like: one 640 KB region (1024 rows x 640 B) shared by all workgroups
    run<128, 4>("shared 640KB (W)", buf, bytes, 1024 * 640, 640, 0, iters, dcyc);
    run<64, 4>("shared 640KB (W)", buf, bytes, 1024 * 640, 640, 0, iters, dcyc);
    run<128, 8>("shared 640KB (W)", buf, bytes, 1024 * 640, 640, 0, iters, dcyc);
    run<64, 8>("shared 640KB (W)", buf, bytes, 1024 * 640, 640, 0, iters, dcyc);
    run<128, 2>("shared 640KB (W)", buf, bytes, 1024 * 640, 640, 0, iters, dcyc);
    run<128, 1>("shared 640KB (W)", buf, bytes, 1024 * 640, 640, 0, iters, dcyc);
    // private L2-hot: each workgroup re-reads its own 160 KB (256 rows x 640 B): 41 MB in all = beyond the 32 MB of L2, inside the MALL
    run<128, 4>("private 160KB (L2/MALL)", buf, bytes, 256 * 640, 640, 1, iters, dcyc);
    run<64, 4>("private 160KB (L2/MALL)", buf, bytes, 256 * 640, 640, 1, iters, dcyc);
    // private 80 KB: 20 MB in all, inside L2
    run<128, 4>("private 80KB (L2)", buf, bytes, 128 * 640, 640, 1, iters, dcyc);
    run<64, 4>("private 80KB (L2)", buf, bytes, 128 * 640, 640, 1, iters, dcyc);
    run<128, 8>("private 80KB (L2)", buf, bytes, 128 * 640, 640, 1, iters, dcyc);
    // streaming: each workgroup walks 10 MB of its own (2.6 GB in all) -> HBM
    run<128, 4>("stream 10MB/WG (HBM)", buf, bytes, 10 << 20, 640, 2, iters, dcyc);
    run<64, 4>("stream 10MB/WG (HBM)", buf, bytes, 10 << 20, 640, 2, iters, dcyc);
    run<128, 8>("stream 10MB/WG (HBM)", buf, bytes, 10 << 20, 640, 2, iters, dcyc);
    run<64, 8>("stream 10MB/WG (HBM)", buf, bytes, 10 << 20, 640, 2, iters, dcyc);
    // full-row pitch = ROWB (dense): a [rows][64 or 128 B] matrix
    run<128, 4>("stream dense rows", buf, bytes, 10 << 20, 128, 2, iters, dcyc);
    run<64, 4>("stream dense rows", buf, bytes, 10 << 20, 64, 2, iters, dcyc);
    return 0;
}
