// What the 3x3 kernel's STAGE costs beyond its MFMAs (round 4, after tools/mfma_ceiling.hip): a stand-alone model of one stage of
// conv3x3_kernel - 8 waves, one block per CU; per stage and wave 48 v_mfma_f32_16x16x32_bf16 on 16 accumulator tiles, 18
// ds_read_b128 (6 "halo" fragments + 3 x 4 "weight" fragments), 3 LDS-DMA pieces of 1 KB into the other weight buffer, a wait for
// them and a block barrier - with the pieces switched off or issued in different forms:
//   none            no weight stream, no barrier                      (= mfma_ceiling's lds variant at this kernel's read count)
//   barrier         no weight stream, one barrier per stage
//   global          global_load_lds_dwordx4, 64-bit per-lane address  (what the kernel ships)
//   buffer_voff     buffer_load_dwordx4 ... offen lds, per-lane 32-bit offset recomputed per piece
//   buffer_soff     buffer_load_dwordx4 ... offen lds, per-lane offset fixed (lane * 16), piece offset in an SGPR
// Build / run:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/stage_ceiling.hip -o tools/build/stage_ceiling && tools/build/stage_ceiling
// Output per variant: cycles per stage and SIMD (s_memtime, median over waves; two waves per SIMD), the clock, TFLOP/s by HIP events.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 x8 __attribute__((ext_vector_type(8)));
typedef float acc_t __attribute__((ext_vector_type(4)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kStage = 3 * 128 * 64;           // 24 576 bytes: [ky 3][co 128][64 B]
constexpr int kHalo = 324 * 160;               // 51 840
constexpr int kWeights = 12 * kStage;          // one 128 -> 128 tile's weights: 294 912 bytes, L2-resident

enum { NONE = 0, BARRIER = 1, GLOBAL = 2, BUF_VOFF = 3, BUF_SOFF = 4 };

template <int MODE>
__global__ void __launch_bounds__(512, 1) stage_loop(const unsigned char* __restrict__ w, const x8* __restrict__ in, float* __restrict__ out,
                                                       int stages, long long* __restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;                  // one halo image
    unsigned char* Bs = smem + kHalo;          // [2][kStage]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l15 = lane & 15, lq = lane >> 4;
    for (int i = tid; i < (kHalo + 2 * kStage) / 16; i += 512) reinterpret_cast<x8*>(smem)[i] = in[i % 4096];
    __syncthreads();
    const unsigned char* abase = As + ((4 * wm) * 18 + l15) * 160 + lq * 16;
    const unsigned char* bbase = Bs + (wn * 64 + l15) * 64 + ((lq ^ ((l15 >> 1) & 3)) << 4);
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)w, 0, kWeights, 0x00020000);
    acc_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (acc_t){0.f, 0.f, 0.f, 0.f};
    auto issue = [&](int stage, int buf) {
        unsigned char* dst = Bs + buf * kStage + wave * 1024;
        if (MODE == GLOBAL) {
            const unsigned char* src = w + (size_t)stage * kStage + (size_t)tid * 16;
#pragma unroll
            for (int j = 0; j < 3; ++j)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j * 8192),
                                                 (__attribute__((address_space(3))) void*)(dst + j * 8192), 16, 0, 0);
        } else if (MODE == BUF_VOFF) {
#pragma unroll
            for (int j = 0; j < 3; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + j * 8192), 16,
                                                         stage * kStage + tid * 16 + j * 8192, 0, 0, 0);
        } else if (MODE == BUF_SOFF) {
            const int so = __builtin_amdgcn_readfirstlane(stage * kStage);
#pragma unroll
            for (int j = 0; j < 3; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + j * 8192), 16,
                                                         tid * 16, so + j * 8192, 0, 0);
        }
    };
    if (MODE >= GLOBAL) { issue(0, 0); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int st12 = 0;
#pragma unroll 1
    for (int ss = 0; ss < stages; ++ss) {
        const unsigned char* bb = bbase + (ss & 1) * kStage;
        const int a_off = (st12 >> 2) * 160 + (st12 & 1) * 64;       // a wave-uniform, stage-dependent halo offset
        st12 = st12 == 11 ? 0 : st12 + 1;
        if (MODE >= GLOBAL) issue(st12, (ss + 1) & 1);
        x8 a[6], b0[4], b1[4];
#pragma unroll
        for (int r = 0; r < 6; ++r) a[r] = *reinterpret_cast<const x8*>(abase + a_off + r * (18 * 160));
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) b0[nt] = *reinterpret_cast<const x8*>(bb + nt * 1024);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) b1[nt] = *reinterpret_cast<const x8*>(bb + 8192 + nt * 1024);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[nt], a[mt], acc[mt][nt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) b0[nt] = *reinterpret_cast<const x8*>(bb + 16384 + nt * 1024);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b1[nt], a[mt + 1], acc[mt][nt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b0[nt], a[mt + 2], acc[mt][nt], 0, 0, 0);
        if (MODE >= GLOBAL) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE >= BARRIER) __syncthreads();
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    acc_t s = (acc_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) s += acc[i][j];
    out[(size_t)blockIdx.x * 512 + tid] = s[0] + s[1] + s[2] + s[3];
    if (lane == 0) {
        const int wv = blockIdx.x * 8 + wave;
        stamps[2 * wv] = t1 - t0;
        stamps[2 * wv + 1] = r1 - r0;
    }
}

template <int MODE>
static void run(const char* name, const unsigned char* w, const x8* in, float* out, long long* stamps, int cus) {
    const int stages = 12 * 400;
    const int smem = kHalo + 2 * kStage;
    CK(hipFuncSetAttribute((const void*)stage_loop<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0.f;
    for (float total = 0.f; total < 600.f;) {
        CK(hipEventRecord(e0));
        stage_loop<MODE><<<cus, 512, smem>>>(w, in, out, stages, stamps);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        total += ms;
    }
    std::vector<float> t;
    for (int r = 0; r < 7; ++r) {
        CK(hipEventRecord(e0));
        stage_loop<MODE><<<cus, 512, smem>>>(w, in, out, stages, stamps);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    const int nw = cus * 8;
    std::vector<long long> h(2 * nw);
    CK(hipMemcpy(h.data(), stamps, sizeof(long long) * 2 * nw, hipMemcpyDeviceToHost));
    std::vector<double> cyc(nw), clk(nw);
    for (int i = 0; i < nw; ++i) {
        cyc[i] = (double)h[2 * i] / stages;
        clk[i] = (double)h[2 * i] / (double)h[2 * i + 1] * 0.1;
    }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double flop = 2.0 * 16 * 16 * 32 * 48.0 * stages * nw;
    const double tf = flop / (t[t.size() / 2] * 1e-3) / 1e12;
    printf("{\"variant\": \"%s\", \"cycles_per_stage\": %.0f, \"ideal_cycles_per_stage\": 1536, \"clock_GHz\": %.3f, \"ms\": %.3f, \"TFLOPs\": %.1f, \"frac_of_2500\": %.3f}\n",
           name, cyc[nw / 2], clk[nw / 2], t[t.size() / 2], tf, tf / 2500.0);
    fflush(stdout);
}

int main() {
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    printf("# %s, %d CUs\n", p.name, cus);
    std::vector<unsigned short> h(4096 * 8);
    srand(1);
    for (auto& v : h) {
        float f = (float)(rand() & 0xFFFFFF) / 16777216.f * 2.f - 1.f;
        unsigned u; memcpy(&u, &f, 4);
        v = (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
    }
    std::vector<unsigned short> hw(kWeights / 2);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = h[i % h.size()];
    x8* in; float* out; long long* stamps; unsigned char* w;
    CK(hipMalloc(&in, h.size() * 2)); CK(hipMalloc(&out, sizeof(float) * cus * 512)); CK(hipMalloc(&stamps, sizeof(long long) * 2 * cus * 8));
    CK(hipMalloc(&w, kWeights));
    CK(hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(w, hw.data(), kWeights, hipMemcpyHostToDevice));
    run<NONE>("none", w, in, out, stamps, cus);
    run<BARRIER>("barrier", w, in, out, stamps, cus);
    run<GLOBAL>("global_load_lds", w, in, out, stamps, cus);
    run<BUF_VOFF>("buffer_load_lds_voffset", w, in, out, stamps, cus);
    run<BUF_SOFF>("buffer_load_lds_soffset", w, in, out, stamps, cus);
    return 0;
}
