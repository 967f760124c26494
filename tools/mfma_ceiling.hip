// What a hipcc-built stream of v_mfma_f32_16x16x32_bf16 sustains on this chip: the number the
// MFMA kernels' roofline fractions should be read against next to the nominal 2.5 PFLOP/s. Stand-alone (no torch):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_ceiling.hip -o tools/build/mfma_ceiling && tools/build/mfma_ceiling
// Variants (each on all 256 CUs, random bf16 operands, >= 1 s of back-to-back launches before the timed ones):
//   bare<NACC>      NACC accumulator tiles per wave (16 = the shipped 8-wave 3x3 kernel's wave tile, 32 = the 4-wave
//                   prototype's), operands in registers, nothing but MFMAs and the loop;
//   lds<NACC>       the same with every operand fragment re-read from LDS each iteration at the shipped kernel's ratio
//                   (0.375 ds_read_b128 per MFMA), reads fenced ahead of the MFMAs they feed (one fragment set ahead);
//   1 or 2 waves per SIMD (256- or 512-thread blocks, one block per CU).
// Per variant: cycles per MFMA (s_memtime, median over blocks), the clock the chip holds (s_memtime / s_memrealtime),
// wall time by HIP events -> TFLOP/s and the fraction of 2.5 PFLOP/s.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

typedef __bf16 x8 __attribute__((ext_vector_type(8)));
typedef float acc_t __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NACC, bool LDS, int TB>
__global__ void __launch_bounds__(TB) mfma_loop(const x8* __restrict__ in, float* __restrict__ out, int iters,
                                                  long long* __restrict__ stamps) {
    constexpr int NA = 4, NB = NACC / 4;                  // acc[j] += A[j % 4] x B[j / 4]: 12 fragments per 32 MFMAs = 0.375
    __shared__ __attribute__((aligned(16))) unsigned char sm[LDS ? 2 * (NA + NB) * 1024 : 16];   // two fragment sets, shared by the block's waves
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    x8 a[NA], b[NB];
    for (int i = 0; i < NA; ++i) a[i] = in[(i * 64 + lane) % 4096];
    for (int i = 0; i < NB; ++i) b[i] = in[((NA + i) * 64 + lane + 17 * wv) % 4096];
    if (LDS) {
        // two fragment sets [set][frag][lane] (16 B per lane, conflict-free ds_read_b128), every wave reads the same bytes
        x8* mine = reinterpret_cast<x8*>(sm);
        if (wv == 0) for (int s = 0; s < 2; ++s) {
            for (int i = 0; i < NA; ++i) mine[(s * (NA + NB) + i) * 64 + lane] = in[((s * 31 + i) * 64 + lane) % 4096];
            for (int i = 0; i < NB; ++i) mine[(s * (NA + NB) + NA + i) * 64 + lane] = in[((s * 29 + NA + i) * 64 + lane + 17 * wv) % 4096];
        }
        __syncthreads();
    }
    acc_t acc[NACC];
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = (acc_t){0.f, 0.f, 0.f, 0.f};
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (!LDS) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j % NA], b[j / NA], acc[j], 0, 0, 0);
        }
    } else {
        const x8* mine = reinterpret_cast<const x8*>(sm) + lane;
        x8 a2[NA], b2[NB];
        for (int it = 0; it < iters; it += 2) {
            // read set 1 while set 0 multiplies, then the other way round
#pragma unroll
            for (int i = 0; i < NA; ++i) a2[i] = mine[((NA + NB) + i) * 64];
#pragma unroll
            for (int i = 0; i < NB; ++i) b2[i] = mine[((NA + NB) + NA + i) * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j % NA], b[j / NA], acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NA; ++i) a[i] = mine[i * 64];
#pragma unroll
            for (int i = 0; i < NB; ++i) b[i] = mine[(NA + i) * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2[j % NA], b2[j / NA], acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    acc_t s = (acc_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NACC; ++j) s += acc[j];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (lane == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + wv;
        stamps[2 * w] = t1 - t0;
        stamps[2 * w + 1] = r1 - r0;
    }
}

template <int NACC, bool LDS, int TB>
static void run(const char* name, int waves_per_simd, const x8* in, float* out, long long* stamps, int cus, float warm_ms = 1000.f) {
    const int threads = 256 * waves_per_simd, blocks = cus;
    const int iters = 20000;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int nw = blocks * threads / 64;
    // warm: >= 1 s of back-to-back launches on random data so the clock settles where a real run holds it
    float ms = 0.f;
    int reps = 0;
    for (float total = 0.f; total < warm_ms; ++reps) {
        CK(hipEventRecord(e0));
        mfma_loop<NACC, LDS, TB><<<blocks, threads>>>(in, out, iters, stamps);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        total += ms;
    }
    std::vector<float> t;
    for (int r = 0; r < 10; ++r) {
        CK(hipEventRecord(e0));
        mfma_loop<NACC, LDS, TB><<<blocks, threads>>>(in, out, iters, stamps);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    std::vector<long long> h(2 * nw);
    CK(hipMemcpy(h.data(), stamps, sizeof(long long) * 2 * nw, hipMemcpyDeviceToHost));
    std::vector<double> cyc(nw), clk(nw);
    for (int w = 0; w < nw; ++w) {
        cyc[w] = (double)h[2 * w] / ((double)iters * NACC);
        clk[w] = (double)h[2 * w] / (double)h[2 * w + 1] * 0.1;     // GHz (s_memrealtime ticks at 100 MHz)
    }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double flop = 2.0 * 16 * 16 * 32 * (double)NACC * iters * nw;
    const double tf = flop / (t[t.size() / 2] * 1e-3) / 1e12;
    // cycles per MFMA as the SIMD sees them: a wave's cycles per its own MFMA, divided by the waves sharing the pipe
    printf("{\"variant\": \"%s\", \"acc_tiles\": %d, \"waves_per_simd\": %d, \"wave_cycles_per_mfma\": %.2f, \"simd_cycles_per_mfma\": %.2f, "
           "\"clock_GHz\": %.3f, \"ms\": %.3f, \"TFLOPs\": %.1f, \"frac_of_2500\": %.3f, \"launches_before\": %d}\n",
           name, NACC, waves_per_simd, cyc[nw / 2], cyc[nw / 2] / waves_per_simd, clk[nw / 2], t[t.size() / 2], tf, tf / 2500.0, reps);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const bool quick = argc > 1 && std::string(argv[1]) == "--quick";      // bench.py: two variants, 0.4 s of warm-up each
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("# %s, %d CUs\n", p.name, cus);
    std::vector<unsigned short> h(4096 * 8);
    srand(1);
    for (auto& v : h) {                      // uniform [-1, 1) as bf16 (full-range random operands: DVFS-honest)
        float f = (float)(rand() & 0xFFFFFF) / 16777216.f * 2.f - 1.f;
        unsigned u; memcpy(&u, &f, 4);
        v = (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
    }
    x8* in; float* out; long long* stamps;
    CK(hipMalloc(&in, h.size() * 2)); CK(hipMalloc(&out, sizeof(float) * cus * 512)); CK(hipMalloc(&stamps, sizeof(long long) * 2 * cus * 8));
    CK(hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    if (quick) {
        run<16, false, 512>("bare", 2, in, out, stamps, cus, 400.f);
        run<16, true, 512>("lds_0.375_reads_per_mfma", 2, in, out, stamps, cus, 400.f);
        return 0;
    }
    // TB = 512: at most 256 registers per lane (two waves per SIMD fit); TB = 256: up to 512 (accumulators may sit in AGPRs)
    run<16, false, 512>("bare", 1, in, out, stamps, cus);
    run<16, false, 512>("bare", 2, in, out, stamps, cus);
    run<32, false, 256>("bare", 1, in, out, stamps, cus);
    run<16, true, 512>("lds_0.375_reads_per_mfma", 1, in, out, stamps, cus);
    run<16, true, 512>("lds_0.375_reads_per_mfma", 2, in, out, stamps, cus);
    run<32, true, 256>("lds_0.375_reads_per_mfma", 1, in, out, stamps, cus);
    run<32, true, 512>("lds_0.375_reads_per_mfma_256regs", 2, in, out, stamps, cus);
    return 0;
}
