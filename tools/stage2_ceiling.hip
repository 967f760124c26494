// Round 5: a stand-alone model of the stage of a 3x3 kernel whose eight waves split the OUTPUT CHANNELS (16 each) and all read the
// whole 16 x 16-pixel halo image: a wave's weight fragments are its own (3 per stage: straight from L2 into registers, no LDS
// image, no LDS-DMA), so the block barrier is needed once per 64-channel chunk (the halo image) instead of once per stage.
// Per stage and wave, as in conv3x3_kernel: 48 v_mfma_f32_16x16x32_bf16 on 16 accumulator tiles and 18 ds_read_b128 - here all 18
// are halo fragments (rows h = 0..17; row h feeds output rows h, h - 1, h - 2 at ky = 0, 1, 2), streamed through a short ring.
// Variants:
//   bare        no weight loads, no halo work, no barrier
//   weights     + the 3 global_load_dwordx4 of the NEXT stage's fragments per stage
//   chunk       + one barrier per 6 stages
//   halo        + per chunk and thread six 16-byte global loads, an affine + ReLU on them, six ds_write_b128 into the other image
// Build / run:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/stage2_ceiling.hip -o tools/build/stage2_ceiling && tools/build/stage2_ceiling
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 x8 __attribute__((ext_vector_type(8)));
typedef float acc_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kStage = 3 * 128 * 64;           // 24 576 bytes: [ky 3][co 128][64 B]
constexpr int kRS = 160, kHW = 18;
constexpr int kHalo = 324 * kRS;               // 51 840
constexpr int kWeights = 12 * kStage;

enum { BARE = 0, WEIGHTS = 1, CHUNK = 2, HALO = 3 };

__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    typedef __bf16 b2_t __attribute__((ext_vector_type(2)));
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2_t));
}

template <int H, int RING, int VALU> __device__ __forceinline__ void stage_order() {
    if constexpr (H < 18) {
        constexpr int nm = (H < 2 ? H + 1 : (H > 15 ? 18 - H : 3));
        __builtin_amdgcn_sched_group_barrier(0x008, nm, 0);
        if constexpr (H + RING < 18) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        if constexpr (VALU > 0) {
            __builtin_amdgcn_sched_group_barrier(0x002, VALU, 0);
            if constexpr (H == 8 || H == 17) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
        }
        stage_order<H + 1, RING, VALU>();
    }
}

template <int MODE, int RING, int VPS = 4>
__global__ void __launch_bounds__(512, 1) stage_loop(const unsigned char* __restrict__ w, const x8* __restrict__ in, float* __restrict__ out,
                                                       int chunks, long long* __restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;                  // two halo images
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, lq = lane >> 4;
    for (int i = tid; i < 2 * kHalo / 16; i += 512) reinterpret_cast<x8*>(smem)[i] = in[i % 4096];
    __syncthreads();
    const unsigned char* abase = As + l15 * kRS + lq * 16;
    const unsigned char* wlane = w + wave * 1024 + lane * 16;
    acc_t acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (acc_t){0.f, 0.f, 0.f, 0.f};
    x8 b[2][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) b[0][ky] = *reinterpret_cast<const x8*>(wlane + ky * 8192);
    const int q54 = tid >> 3, slot = tid & 7;
    unsigned char* cdst = As + (q54 < 54 ? q54 : 324 * 2) * kRS + slot * 16;     // (past both images: a dummy row)
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    int st12 = 0;
#pragma unroll 1
    for (int cc = 0; cc < chunks; ++cc) {
        const unsigned char* ab = abase + (cc & 1) * kHalo;
        uint4 av[6];
#pragma unroll
        for (int sl = 0; sl < 6; ++sl) {
            const int a_off = (sl >> 1) * kRS + (sl & 1) * 64;
            st12 = st12 == 11 ? 0 : st12 + 1;
            if (MODE >= WEIGHTS) {
                const unsigned char* ws = wlane + (size_t)st12 * kStage;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) b[(sl + 1) & 1][ky] = *reinterpret_cast<const x8*>(ws + ky * 8192);
            }
            if (MODE >= HALO && sl == 0) {
#pragma unroll
                for (int i = 0; i < 6; ++i) av[i] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned char*>(in) + ((q54 * 8 + slot + i * 432 + cc * 64) & 4095) * 16);
            }
            x8 a[18];
#pragma unroll
            for (int h = 0; h < RING; ++h) a[h] = *reinterpret_cast<const x8*>(ab + a_off + h * (kHW * kRS));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < 18; ++h) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int r = h - ky;
                    if (r >= 0 && r < 16) acc[r] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[sl & 1][ky], a[h], acc[r], 0, 0, 0);
                }
                if (h + RING < 18) a[h + RING] = *reinterpret_cast<const x8*>(ab + a_off + (h + RING) * (kHW * kRS));
                const bool commit = MODE >= HALO && sl >= 1 && sl <= 3 && (h == 0 || h == 9);
                if (commit) {
                    // commit one piece: affine + relu + pack + ds_write_b128 into the other image (threads past the image: a dummy row)
                    const int i = (sl - 1) * 2 + (h == 9);
                    const unsigned u[4] = {av[i].x, av[i].y, av[i].z, av[i].w};
                    unsigned o[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        f32x2_t f = {__uint_as_float(u[j] << 16), __uint_as_float(u[j] & 0xffff0000u)};
                        f = f * (f32x2_t){1.01f, 0.99f} + (f32x2_t){0.01f, -0.01f};
                        o[j] = pack_bf16x2(fmaxf(f[0], 0.f), fmaxf(f[1], 0.f));
                    }
                    *reinterpret_cast<uint4*>(cdst + ((cc + 1) & 1) * kHalo + 54 * i * kRS) = make_uint4(o[0], o[1], o[2], o[3]);
                }
            }
            // the order of the stage: the MFMAs of halo row h, then the read of row h + RING
            if (MODE >= HALO && sl >= 1 && sl <= 3) stage_order<0, RING, VPS>(); else stage_order<0, RING, 0>();
            __builtin_amdgcn_sched_barrier(0);
        }
        if (MODE >= CHUNK) __syncthreads();
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    acc_t s = (acc_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[(size_t)blockIdx.x * 512 + tid] = s[0] + s[1] + s[2] + s[3];
    if (lane == 0) {
        const int wv = blockIdx.x * 8 + wave;
        stamps[2 * wv] = t1 - t0;
        stamps[2 * wv + 1] = r1 - r0;
    }
}

template <int MODE, int RING, int VPS = 4>
static void run(const char* name, const unsigned char* w, const x8* in, float* out, long long* stamps, int cus) {
    const int chunks = 2 * 400;
    const int smem = 2 * kHalo + 6 * 54 * kRS;
    CK(hipFuncSetAttribute((const void*)stage_loop<MODE, RING, VPS>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms = 0.f;
    for (float total = 0.f; total < 600.f;) {
        CK(hipEventRecord(e0));
        stage_loop<MODE, RING, VPS><<<cus, 512, smem>>>(w, in, out, chunks, stamps);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        total += ms;
    }
    std::vector<float> t;
    for (int r = 0; r < 7; ++r) {
        CK(hipEventRecord(e0));
        stage_loop<MODE, RING, VPS><<<cus, 512, smem>>>(w, in, out, chunks, stamps);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        t.push_back(ms);
    }
    std::sort(t.begin(), t.end());
    const int nw = cus * 8;
    std::vector<long long> h(2 * nw);
    CK(hipMemcpy(h.data(), stamps, sizeof(long long) * 2 * nw, hipMemcpyDeviceToHost));
    const int stages = chunks * 6;
    std::vector<double> cyc(nw), clk(nw);
    for (int i = 0; i < nw; ++i) {
        cyc[i] = (double)h[2 * i] / stages;
        clk[i] = (double)h[2 * i] / (double)h[2 * i + 1] * 0.1;
    }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    const double flop = 2.0 * 16 * 16 * 32 * 48.0 * stages * nw;
    const double tf = flop / (t[t.size() / 2] * 1e-3) / 1e12;
    printf("{\"variant\": \"%s\", \"ring\": %d, \"cycles_per_stage\": %.0f, \"ideal_cycles_per_stage\": 1536, \"clock_GHz\": %.3f, \"ms\": %.3f, \"TFLOPs\": %.1f, \"frac_of_2500\": %.3f}\n",
           name, RING, VPS, cyc[nw / 2], clk[nw / 2], t[t.size() / 2], tf, tf / 2500.0);
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
    run<BARE, 6>("bare", w, in, out, stamps, cus);
    run<WEIGHTS, 6>("weights", w, in, out, stamps, cus);
    run<CHUNK, 6>("chunk", w, in, out, stamps, cus);
    run<HALO, 6>("halo", w, in, out, stamps, cus);
    run<HALO, 6, 2>("halo", w, in, out, stamps, cus);
    run<HALO, 6, 3>("halo", w, in, out, stamps, cus);
    run<HALO, 6, 6>("halo", w, in, out, stamps, cus);
    run<HALO, 9, 4>("halo", w, in, out, stamps, cus);
    return 0;
}
