// What bounds the depthwise sliding-window walk when its tensors are cold: a stand-alone model of the
// kernel's MEMORY shape only - a thread owns VEC bytes of channels of one column pair and walks down a strip of rows, per
// output row it loads the 4 window pieces of the new input row and stores 2 output pieces - with the arithmetic reduced to a
// few adds. Knobs: P = input rows in flight per thread (the shipped kernel: 3), VEC = 8 or 16 bytes per piece, and the
// number of rows per strip. Buffers rotate over sets far larger than the 256 MB memory-side cache.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/walk_ceiling.hip -o tools/build/walk_ceiling && tools/build/walk_ceiling
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int VEC> struct Piece;
template <> struct Piece<8> { uint2 v; __device__ void add(const Piece& o) { v.x += o.v.x; v.y ^= o.v.y; } };
template <> struct Piece<16> { uint4 v; __device__ void add(const Piece& o) { v.x += o.v.x; v.y ^= o.v.y; v.z += o.v.z; v.w ^= o.v.w; } };

// x, y: [N][H][W][C] 16-bit elements. Block = 256 threads = ncg channel groups x cols column pairs; grid = N * yblocks * xblocks * cblocks
template <int P, int VEC>
__global__ __launch_bounds__(256) void walk(const unsigned char* __restrict__ x, unsigned char* __restrict__ y, int H, int W, int C,
                                             int rows, int ncg, int cols, int xblocks, int yblocks, int cblocks) {
    int b = blockIdx.x;
    const int xb = b % xblocks; b /= xblocks;
    const int yb = b % yblocks; b /= yblocks;
    const int cgb = b % cblocks;
    const int img = b / cblocks;
    const int cgl = threadIdx.x % ncg, col = threadIdx.x / ncg;
    const long long cbyte = (long long)(cgb * ncg + cgl) * VEC;
    const int ox = (xb * cols + col) * 2;
    if (ox >= W || col >= cols) return;
    const long long rowb = (long long)W * C * 2, pixb = (long long)C * 2;
    const unsigned char* xi = x + (long long)img * H * rowb + cbyte;
    unsigned char* yo = y + (long long)img * H * rowb + cbyte + (long long)ox * pixb;
    long long xoff[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) xoff[k] = (long long)std::min(std::max(ox - 1 + k, 0), W - 1) * pixb;
    const int oy0 = yb * rows, oy1 = std::min(oy0 + rows, H);
    Piece<VEC> buf[P][4];
    auto load = [&](Piece<VEC> (&r)[4], int iy) {
        const unsigned char* rp = xi + (long long)std::min(std::max(iy, 0), H - 1) * rowb;
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = *reinterpret_cast<const Piece<VEC>*>(rp + xoff[k]);
    };
#pragma unroll
    for (int j = 0; j < P; ++j) load(buf[j], oy0 - 1 + j);
    Piece<VEC> w0 = buf[0][0], w1 = buf[0][3];          // stand-in for the window carried across rows
    for (int oy = oy0; oy < oy1; oy += P) {
#pragma unroll
        for (int j = 0; j < P; ++j) {
            if (oy + j < oy1) {
                Piece<VEC> a = buf[j][0], c = buf[j][2];
                a.add(buf[j][1]); a.add(w0);
                c.add(buf[j][3]); c.add(w1);
                w0 = buf[j][1]; w1 = buf[j][2];
                load(buf[j], oy + j + P - 1);
                unsigned char* yp = yo + (long long)(oy + j) * rowb;
                *reinterpret_cast<Piece<VEC>*>(yp) = a;
                if (ox + 1 < W) *reinterpret_cast<Piece<VEC>*>(yp + pixb) = c;
            }
        }
    }
}

// Mixed piece sizes: which side's request size matters? A thread owns 16 bytes of channels of a column pair; LV / SV = bytes per load /
// store instruction (8: two instructions per piece, 512-byte wave requests; 16: one, 1-KB requests).
template <int LV, int SV>
__global__ __launch_bounds__(256) void walk_mixed(const unsigned char* __restrict__ x, unsigned char* __restrict__ y, int H, int W, int C,
                                                   int rows, int ncg, int cols, int xblocks, int yblocks, int cblocks) {
    int b = blockIdx.x;
    const int xb = b % xblocks; b /= xblocks;
    const int yb = b % yblocks; b /= yblocks;
    const int cgb = b % cblocks;
    const int img = b / cblocks;
    const int cgl = threadIdx.x % ncg, col = threadIdx.x / ncg;
    const long long cbyte = (long long)(cgb * ncg + cgl) * 16;
    const int ox = (xb * cols + col) * 2;
    if (ox >= W || col >= cols) return;
    const long long rowb = (long long)W * C * 2, pixb = (long long)C * 2;
    const unsigned char* xi = x + (long long)img * H * rowb + cbyte;
    unsigned char* yo = y + (long long)img * H * rowb + cbyte + (long long)ox * pixb;
    long long xoff[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) xoff[k] = (long long)std::min(std::max(ox - 1 + k, 0), W - 1) * pixb;
    const int oy0 = yb * rows, oy1 = std::min(oy0 + rows, H);
    uint4 buf[2][4];
    auto load = [&](uint4 (&r)[4], int iy) {
        const unsigned char* rp = xi + (long long)std::min(std::max(iy, 0), H - 1) * rowb;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (LV == 16) r[k] = *reinterpret_cast<const uint4*>(rp + xoff[k]);
            else {
                const uint2 a = *reinterpret_cast<const uint2*>(rp + xoff[k]), c = *reinterpret_cast<const uint2*>(rp + xoff[k] + 8);
                r[k] = make_uint4(a.x, a.y, c.x, c.y);
            }
        }
    };
    load(buf[0], oy0 - 1); load(buf[1], oy0);
    uint4 w0 = buf[0][0], w1 = buf[0][3];
    for (int oy = oy0; oy < oy1; oy += 2) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (oy + j < oy1) {
                uint4 a = buf[j][0], c = buf[j][2];
                a.x += buf[j][1].x + w0.x; a.y ^= buf[j][1].y ^ w0.y; a.z += buf[j][1].z + w0.z; a.w ^= buf[j][1].w ^ w0.w;
                c.x += buf[j][3].x + w1.x; c.y ^= buf[j][3].y ^ w1.y; c.z += buf[j][3].z + w1.z; c.w ^= buf[j][3].w ^ w1.w;
                w0 = buf[j][1]; w1 = buf[j][2];
                load(buf[j], oy + j + 1);
                unsigned char* yp = yo + (long long)(oy + j) * rowb;
                if (SV == 16) {
                    *reinterpret_cast<uint4*>(yp) = a;
                    if (ox + 1 < W) *reinterpret_cast<uint4*>(yp + pixb) = c;
                } else {
                    *reinterpret_cast<uint2*>(yp) = make_uint2(a.x, a.y); *reinterpret_cast<uint2*>(yp + 8) = make_uint2(a.z, a.w);
                    if (ox + 1 < W) { *reinterpret_cast<uint2*>(yp + pixb) = make_uint2(c.x, c.y); *reinterpret_cast<uint2*>(yp + pixb + 8) = make_uint2(c.z, c.w); }
                }
            }
        }
    }
}
template <int LV, int SV>
static double run_mixed(int N, int H, int W, int C, int rows, std::vector<unsigned char*>& xs, std::vector<unsigned char*>& ys, int iters) {
    int ncg = C / 8; if (ncg > 16) ncg = 16;
    const int cblocks = (C / 8) / ncg, cols = 256 / ncg;
    const int xblocks = (W / 2 + cols - 1) / cols, yblocks = (H + rows - 1) / rows;
    const int grid = N * yblocks * xblocks * cblocks;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int ns = (int)xs.size();
    for (int i = 0; i < ns; ++i) walk_mixed<LV, SV><<<grid, 256>>>(xs[i], ys[i], H, W, C, rows, ncg, cols, xblocks, yblocks, cblocks);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) walk_mixed<LV, SV><<<grid, 256>>>(xs[i % ns], ys[i % ns], H, W, C, rows, ncg, cols, xblocks, yblocks, cblocks);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e-3 / iters;
}

// The same walk with the input rows staged by LDS-DMA into a WAVE-PRIVATE ring of D rows (no block barriers: a wave waits
// for its own DMA with a counted vmcnt and reads only what it fetched): thread = 8 bytes of channels of one column pair as in
// the shipped kernel, a wave = 64 / ncg column pairs, its ring row = (2 * pairs + 2) pixels x C x 2 bytes fetched by
// 1-KB DMA instructions (lane -> 16-byte slot of a pixel; lanes past the row clamp to its last slot).
template <int D>
__global__ __launch_bounds__(256) void walk_ring(const unsigned char* __restrict__ x, unsigned char* __restrict__ y, int H, int W, int C,
                                                  int rows, int ncg, int cols, int xblocks, int yblocks, int cblocks) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];
    int b = blockIdx.x;
    const int xb = b % xblocks; b /= xblocks;
    const int yb = b % yblocks; b /= yblocks;
    const int cgb = b % cblocks;
    const int img = b / cblocks;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cgl = threadIdx.x % ncg, col = threadIdx.x / ncg;
    const int ppw = 64 / ncg;                         // column pairs per wave
    const int wcol0 = (xb * cols + wave * ppw) * 2;   // first output column of this wave
    const int npx = 2 * ppw + 2;                      // pixels per ring row (with the two halo columns)
    const int cb = ncg * 8;                           // bytes of channels per pixel inside the block
    const int rowbytes = npx * cb;
    const int ninst = (rowbytes + 1023) / 1024;
    unsigned char* myring = ring + wave * D * ninst * 1024;
    const long long rowb = (long long)W * C * 2, pixb = (long long)C * 2;
    const unsigned char* xi = x + (long long)img * H * rowb + (long long)cgb * cb;
    const int ox = (xb * cols + col) * 2;
    unsigned char* yo = y + (long long)img * H * rowb + (long long)(cgb * ncg + cgl) * 8 + (long long)ox * pixb;
    // DMA lane map: byte l*16 of the ring row -> pixel (l*16)/cb, byte offset (l*16)%cb
    long long soff[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        int byte = (k * 64 + lane) * 16;
        if (byte >= rowbytes) byte = rowbytes - 16;
        const int px = byte / cb, within = byte % cb;
        const int ix = std::min(std::max(wcol0 - 1 + px, 0), W - 1);
        soff[k] = (long long)ix * pixb + within;
    }
    const int oy0 = yb * rows, oy1 = std::min(oy0 + rows, H);
    auto dma = [&](int iy, int slot) {
        const unsigned char* rp = xi + (long long)std::min(std::max(iy, 0), H - 1) * rowb;
        unsigned char* dst = myring + slot * ninst * 1024;
        for (int k = 0; k < ninst; ++k)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rp + soff[k]),
                                             (__attribute__((address_space(3))) void*)(dst + k * 1024), 16, 0, 0);
    };
    if (ox >= W || col >= cols) {
        // (idle lanes still take part in the wave's DMA: no early return)
    }
    for (int j = 0; j < D; ++j) dma(oy0 - 1 + j, j);
    const int mypx = (col - wave * ppw) * 2;          // first of this thread's 4 window pixels inside the ring row
    uint2 w0 = {0u, 0u}, w1 = {0u, 0u};
    int slot = 0;
    for (int oy = oy0; oy < oy1; ++oy) {
        // rows oy-1 .. oy-1+D-1 are in flight / landed; the oldest is the one to read: everything issued after it may stay in flight
        if (ninst == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((D - 1) * 1 + 2 * (D - 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((D - 1) * 2 + 2 * (D - 1)) : "memory");
        const unsigned char* r = myring + slot * ninst * 1024 + mypx * cb + cgl * 8;
        const uint2 p0 = *reinterpret_cast<const uint2*>(r), p1 = *reinterpret_cast<const uint2*>(r + cb);
        const uint2 p2 = *reinterpret_cast<const uint2*>(r + 2 * cb), p3 = *reinterpret_cast<const uint2*>(r + 3 * cb);
        uint2 a = {p0.x + p1.x + w0.x, p0.y ^ p1.y ^ w0.y}, c = {p2.x + p3.x + w1.x, p2.y ^ p3.y ^ w1.y};
        w0 = p1; w1 = p2;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the slot's reads are done before it is refilled
        dma(oy - 1 + D, slot);
        if (ox < W && col < cols) {
            unsigned char* yp = yo + (long long)oy * rowb;
            *reinterpret_cast<uint2*>(yp) = a;
            if (ox + 1 < W) *reinterpret_cast<uint2*>(yp + pixb) = c;
        }
        slot = slot + 1 == D ? 0 : slot + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int D>
static double run_ring(int N, int H, int W, int C, int rows, std::vector<unsigned char*>& xs, std::vector<unsigned char*>& ys, int iters) {
    int ncg = C / 4; if (ncg > 32) ncg = 32;
    const int cblocks = (C / 4) / ncg, cols = 256 / ncg;
    const int xblocks = (W / 2 + cols - 1) / cols, yblocks = (H + rows - 1) / rows;
    const int grid = N * yblocks * xblocks * cblocks;
    const int ppw = 64 / ncg, rowbytes = (2 * ppw + 2) * ncg * 8, ninst = (rowbytes + 1023) / 1024;
    if (ninst > 2) return -1.0;
    const size_t smem = (size_t)4 * D * ninst * 1024;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int ns = (int)xs.size();
    CK(hipFuncSetAttribute((const void*)walk_ring<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
    for (int i = 0; i < ns; ++i) walk_ring<D><<<grid, 256, smem>>>(xs[i], ys[i], H, W, C, rows, ncg, cols, xblocks, yblocks, cblocks);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) walk_ring<D><<<grid, 256, smem>>>(xs[i % ns], ys[i % ns], H, W, C, rows, ncg, cols, xblocks, yblocks, cblocks);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e-3 / iters;
}

template <int P, int VEC>
static double run(int N, int H, int W, int C, int rows, std::vector<unsigned char*>& xs, std::vector<unsigned char*>& ys, int iters) {
    int ncg = C * 2 / VEC; if (ncg > 32 * 8 / VEC) ncg = 32 * 8 / VEC;      // blocks of at most 128 channels
    const int cblocks = (C * 2 / VEC) / ncg, cols = 256 / ncg;
    const int xblocks = (W / 2 + cols - 1) / cols, yblocks = (H + rows - 1) / rows;
    const int grid = N * yblocks * xblocks * cblocks;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int ns = (int)xs.size();
    for (int i = 0; i < ns; ++i) walk<P, VEC><<<grid, 256>>>(xs[i], ys[i], H, W, C, rows, ncg, cols, xblocks, yblocks, cblocks);
    CK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) walk<P, VEC><<<grid, 256>>>(xs[i % ns], ys[i % ns], H, W, C, rows, ncg, cols, xblocks, yblocks, cblocks);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e-3 / iters;
}

__global__ void copy16(const uint4* __restrict__ a, uint4* __restrict__ b, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) b[i] = a[i];
}

int main() {
    const int N = 32;
    struct L { int H, C; } layers[] = {{256, 32}, {128, 128}, {64, 256}};
    for (auto l : layers) {
        const int H = l.H, W = l.H, C = l.C;
        const size_t bytes = (size_t)N * H * W * C * 2;
        const int ns = (int)std::max<size_t>(2, std::min<size_t>(12, (size_t)1.5e9 / (2 * bytes)));
        std::vector<unsigned char*> xs(ns), ys(ns);
        for (int i = 0; i < ns; ++i) { CK(hipMalloc(&xs[i], bytes)); CK(hipMalloc(&ys[i], bytes)); CK(hipMemset(xs[i], i + 1, bytes)); }
        const int iters = 3 * ns;
        const double tb = 2.0 * bytes;
        printf("[%d,%d,%d,%d] bf16, %d sets of %.0f MB, cold:\n", N, H, W, C, ns, 2 * bytes / 1e6);
        {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            for (int i = 0; i < ns; ++i) copy16<<<256 * 16, 256>>>((const uint4*)xs[i], (uint4*)ys[i], bytes / 16);
            CK(hipEventRecord(e0));
            for (int i = 0; i < iters; ++i) copy16<<<256 * 16, 256>>>((const uint4*)xs[i % ns], (uint4*)ys[i % ns], bytes / 16);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("  grid-stride 16-byte copy: %.1f us  %.3f of 8 TB/s\n", ms * 1e3 / iters, tb / (ms * 1e-3 / iters) / 8e12);
        }
        for (int rows : {32, 64}) {
            if (rows > H) continue;
#define RUN(P, V) { const double t = run<P, V>(N, H, W, C, rows, xs, ys, iters); \
            printf("  rows/strip %2d  P=%d rows in flight  %2d-byte pieces: %6.1f us  %.3f of 8 TB/s\n", rows, P, V, t * 1e6, tb / t / 8e12); fflush(stdout); }
            RUN(2, 8) RUN(3, 8) RUN(4, 8) RUN(6, 8) RUN(8, 8)
            RUN(2, 16) RUN(3, 16) RUN(4, 16) RUN(6, 16)
#define RUNR(D) { const double t = run_ring<D>(N, H, W, C, rows, xs, ys, iters); \
            if (t > 0) printf("  rows/strip %2d  LDS-DMA ring of %d rows per wave, 8-byte window reads: %6.1f us  %.3f of 8 TB/s\n", rows, D, t * 1e6, tb / t / 8e12); fflush(stdout); }
            RUNR(3) RUNR(4) RUNR(6) RUNR(8)
#define RUNM(L, S) { const double t = run_mixed<L, S>(N, H, W, C, rows, xs, ys, iters); \
            printf("  rows/strip %2d  16-byte lanes, %2d-byte loads, %2d-byte stores: %6.1f us  %.3f of 8 TB/s\n", rows, L, S, t * 1e6, tb / t / 8e12); fflush(stdout); }
            if (rows == 32) { RUNM(16, 16) RUNM(8, 16) RUNM(16, 8) RUNM(8, 8) }
        }
        for (int i = 0; i < ns; ++i) { CK(hipFree(xs[i])); CK(hipFree(ys[i])); }
    }
    return 0;
}
