// 3x3 convolution (forward and data gradient), 16-bit storage, 128 output channels per tile: the CHANNEL-SPLIT kernel (round 5).
//
// Same tile as conv3x3.hip - one persistent 8-wave block per CU, 16 x 16 pixels x 128 output channels, 64-channel chunks of an
// 18 x 18 halo image in LDS - but the eight waves split the OUTPUT CHANNELS (16 each) and every wave multiplies all 256 pixels:
//   * a wave's weight fragments are its own: three 16 x 32 fragments per stage (the taps ky = 0..2 of one kernel column for one
//     32-channel k-step) come STRAIGHT FROM L2 INTO REGISTERS (one coalesced 1 KB load each, requested a stage ahead) - no LDS
//     image of the weights, no LDS-DMA, and therefore no block barrier per stage: the block meets once per 64-channel chunk (the
//     halo image changes) instead of six times. conv3x3.hip's stamps showed where its time went: eight waves leave
//     every stage barrier together, read, wait and multiply at the same times, and every vector instruction of the halo commit and
//     the epilogue added its issue time to a stage that 96 MFMAs already fill by half. Here the waves drift apart between two chunk
//     barriers, so one wave's commit or wait runs beside its SIMD partner's MFMAs.
//   * the LDS traffic per MFMA is unchanged: halo row h (16 pixels x 32 channels, one ds_read_b128 per lane) feeds output rows
//     h, h - 1, h - 2 at ky = 0, 1, 2 - 18 fragment reads per 48 MFMAs, streamed through a ring of six - and the weight image in
//     global memory is the one conv3x3.hip streams by LDS-DMA (its 64-byte rows of one output channel: a wave's fragment (ky, its
//     16 channels) is one contiguous KB of it), so mpn_conv_pack_weights is unchanged.
//   * the tile leaves through ONE block-wide bf16 image [256 px][128 co] OF ITS OWN (nothing aliases: both halo images are unpadded
//     128-byte pixel rows with XOR-swizzled 16-byte slots, so two of them, the image and the tables fit the CU's 160 KB), written
//     from the accumulators by each wave as it finishes its last stage - no barrier in front, the chunk's barrier behind - read
//     back by transposing reads for the batch-norm statistics (ones x F, F^T x F on the matrix unit: a wave sums ITS 16 channels
//     over all 256 pixels in the accumulator of one MFMA chain) and as whole 256-byte pixel rows for the stores: every store
//     instruction writes four complete pixels.
// A stand-alone model of this stage (tools/stage2_ceiling.hip, profiles/r05_stage2_ceiling.txt) runs at 0.65 of the nominal MFMA
// peak WITH the halo staging work, where the model of conv3x3.hip's stage reaches 0.66 without it.
// LDS: [halo 0: 41 472][halo 1: 41 472][image 65 536][statistics 8 960][scale / shift table 4 096] = 161 536 bytes.
#include "conv3x3.h"

namespace {
using namespace mpn_c3;

// every LDS access of this kernel goes through an explicit address-space-3 pointer: inside the kernel's inlined lambdas hipcc lost
// track of generic pointers into the dynamic LDS array and emitted flat_load + spilled 64-bit addresses for the fragment reads
#define LDS __attribute__((address_space(3)))
typedef LDS unsigned char* lds_p;
typedef LDS float* lds_f;
template <typename V> __device__ __forceinline__ V lds_ld(lds_p p) { return *(const LDS V*)p; }
template <typename V> __device__ __forceinline__ void lds_st(lds_p p, const V& v) { *(LDS V*)p = v; }

constexpr int kHW = 18;                      // halo width = height
constexpr int kNPix = kHW * kHW;             // 324
// Halo image: 128 bytes per pixel (the chunk's 64 channels), NO padding; the eight 16-byte slots of pixel (hy, hx) are stored at slot ^ (hx & 7).
// A fragment read (16 pixels of one halo row x 32 channels: lane (l15, lq) reads slot 4 ks + lq of pixel l15 + kx) then puts each of
// ds_read_b128's 16-lane groups ({0-3, 12-15, 20-27}, ...: 8 pixels distinct mod 8 at lq, the other 8 at lq ^ 1) on 16 distinct slots
// of the 256-byte bank row: conflict-free without the 32 bytes of padding per pixel the first version carried.
constexpr int kRS = 128;
constexpr int kABytes = kNPix * kRS;         // 41 472
// Output image: 256 bytes per pixel, the sixteen 16-byte slots of pixel px stored at slot ^ (px & 15)
constexpr int kRSO = 256;
constexpr int kImg = 256 * kRSO;             // 65 536
constexpr int kImgOff = 2 * kABytes;
constexpr int kHaloArea = 2 * kABytes + kImg;     // [halo 0][halo 1][image]
constexpr int kRedBytes = (8 * 2 * 128 + 192) * (int)sizeof(float);   // BNR: [8 waves][2][128]; statistics: [2 tiles][2][128]; + 192 dummy words (lanes off the Gram diagonal)
constexpr int kMaxCin = 512;
constexpr int kTabBytes = 2 * kMaxCin * (int)sizeof(float);
#ifdef MPN_DIAG
constexpr int kDiagBytes = 8 * 32 * 8;       // per-wave stamps (below)
#else
constexpr int kDiagBytes = 0;
#endif
constexpr int kLds = kHaloArea + kRedBytes + kTabBytes + kDiagBytes;
constexpr int kAVec = 6;                     // 16-byte pieces of a halo image per thread
constexpr int kRing = 6;                     // halo fragments in flight (4 and 8 measured equal or slower: profiles/r05_c3cs_ab.txt)
static_assert(kLds <= 160 * 1024, "LDS budget");

__device__ __forceinline__ int halo_off(int buf) { return buf ? kABytes : 0; }

// diagnostic build (tools/stamp_c3cs.py; never in the shipped library): per WAVE, s_memtime of the block's THIRD tile at the loop top
// (0), behind each chunk's barrier (1..8), behind the image's barrier (9); 10 = behind the barrier that ends the PREVIOUS tile's
// epilogue (in this tile's first chunk); 12 = s_memrealtime at kernel entry, 13 at its end. Kept in LDS, copied out at the end:
// dbg[(block * 8 + wave) * 32 + k]; 16 + 6 c + s = the end of stage s of chunk c (c < 2)
#ifdef MPN_DIAG
#define CS_STAMP(k) do { if (g.job[0].dbg && titer == 2 && (threadIdx.x & 63) == 0) wst[(threadIdx.x >> 6) * 32 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define CS_STAMP(k) do { } while (0)
#endif

// a / b for 0 <= a < 2^22, b > 0 (tile counts): one v_rcp_f32 and a correction instead of the ~25 scalar instructions of an integer
// division - three divisions per tile stood in front of every tile's first MFMA, in all eight waves at once
__device__ __forceinline__ int qdiv(int a, int b) {
    int q = (int)((float)a * __builtin_amdgcn_rcpf((float)b));
    const int r = a - q * b;
    q += (r >= b) ? 1 : 0;
    q -= (r < 0) ? 1 : 0;
    return q;
}
template <bool N64>
__device__ __forceinline__ Tile tile_fast(const Group& g, int w) {
    Tile t;
    t.job = 0;
#pragma unroll
    for (int k = 1; k < kMaxJobs; ++k)
        if (k < g.njobs && w >= g.begin[k]) t.job = k;
    const Job& p = g.job[t.job];
    int b = w - g.begin[t.job];
    const int n_tiles = N64 ? (p.Cout >> 6) : (p.Cout >> 7), tiles_x = (p.W + 15) >> 4, tiles_y = (p.H + 15) >> 4;
    int q = qdiv(b, n_tiles); t.ntile = b - q * n_tiles; b = q;
    q = qdiv(b, tiles_x); t.tx = b - q * tiles_x; b = q;
    q = qdiv(b, tiles_y); t.ty = b - q * tiles_y;
    t.img = q;
    t.oy0 = t.ty * 16; t.ox0 = t.tx * 16;
    return t;
}

// ACT: 0 = no producer affine, 1 = affine + (ReLU or none, by in_act), 2 = affine + ReLU6
// MODE: 0 = plain, 1 = batch-norm statistics of the output, 2 = data gradient that also reduces for the batch-norm it feeds (BNR)
// N64: 64 output channels per tile (Cout an odd multiple of 64: final_conv3x3 512 -> 64, the detector's 64 -> 64 towers): the eight
//   waves are FOUR channel groups of 16 x TWO pixel halves (output rows 8 ph .. 8 ph + 7): 24 MFMAs per wave and stage on 10 halo
//   fragments and the same three weight fragments; the packed weight image is conv3x3.hip's 64-channel one
//   ([chunk][kx 3][ky 3][k-step 2][co 64][64 bytes]), the output image 128 bytes per pixel.
template <typename T, int ACT, int MODE, bool N64 = false>
__global__ __launch_bounds__(kThreads, 1) void conv3x3_cs_kernel(const Group g) {
    constexpr bool AFFINE = ACT != 0, STATS = MODE == 1, BNR = MODE == 2;
    constexpr int MT = N64 ? 8 : 16;            // output rows (m-tiles) of a wave
    constexpr int HR = MT + 2;                  // halo rows (fragments) a stage reads
    constexpr int CT = N64 ? 64 : 128;          // output channels of a tile
    constexpr int kPx = CT * 2;                 // output image: bytes per pixel (= kRSO without N64)
    constexpr int kSw = N64 ? 7 : 15;           // its slot swizzle: slot ^ (px & kSw)
    // the fused-reduction variant finishes a tile BEHIND its last chunk instead of under the next tile's first stages: its epilogue
    // (the raw tensor of the fed batch-norm, the masks, sixteen running sums) does not fit beside the accumulators - 43 spilled registers
    // (round 6: on 64-channel tiles the overlapped form fits - 212 registers, no spill - and is SLOWER in the detector's step, 5.61-5.63 against
    //  5.55-5.60 ms, profiles/r06_bnr64_pipe.txt: a one-chunk tile has six stages to hide it under; on 128-channel tiles it spills 36-48 registers
    //  whatever the fragment ring's depth)
    constexpr bool PIPE = !BNR;
    static_assert(!BNR || ACT == 0, "the fused batch-norm backward reduction rides on a data gradient (no producer affine)");
    using H = H16<T>;
    using X8 = typename H::x8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const lds_p L = (lds_p)smem;
    const lds_f red = (lds_f)(L + kHaloArea);                 // [8 waves][2][128] statistics of the tile that has just finished
    const lds_f tab = (lds_f)(L + kHaloArea + kRedBytes);     // [2][kMaxCin] scale, shift of the job in `tab_job`

#ifdef MPN_DIAG
    LDS unsigned long long* wst = (LDS unsigned long long*)(L + kHaloArea + kRedBytes + kTabBytes);   // (kDiagBytes)
    if (threadIdx.x < 256) wst[threadIdx.x] = 0;
    int titer = 0;
#endif
    const int total = g.begin[g.njobs];
    // XCD-aware walk (conv3x3.hip): the blocks of one XCD take consecutive tiles of every round
    int w = blockIdx.x;
    if ((gridDim.x & 7) == 0) w = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (w >= total) return;
    const int w_first = w;
#ifdef MPN_DIAG
    const unsigned long long rt_begin = __builtin_amdgcn_s_memrealtime();
#endif

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = N64 ? (wave & 3) : wave, ph = N64 ? (wave >> 2) : 0;      // channel group of 16, pixel half
    const int l15 = lane & 15, lq = lane >> 4;
    const int Cin = g.job[0].Cin, Cout = g.job[0].Cout;    // (shared by the jobs of a group)
    const int nchunk = Cin >> 6;
    const long long wtile = 9ll * Cin * CT * 2;

    // per-lane fragment bases; everything added later is a compile-time or wave-uniform offset
    // (byte offsets inside a halo buffer, one per kernel column kx: pixel c = l15 + kx of a halo row, slot (4 ks + lq) ^ (c & 7); the
    //  k-step toggles bit 6: base ^ 64)
    int abase_kx[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
        const int c = l15 + kx;
        abase_kx[kx] = c * kRS + ((lq ^ (c & 3)) << 4) + (((c >> 2) & 1) << 6);
    }
    // a wave's weight fragment (ky, channels 16 wave .. + 15) inside a stage of the packed image [ky 3][co 128][64 bytes]: rows of
    // 64 bytes = 32 input channels, their four 16-byte slots swizzled with swz(co) (the LDS image of conv3x3.hip)
    const unsigned lane_w = (unsigned)(cg * 1024 + l15 * 64 + ((lq ^ swz(l15)) << 4));
    // buffer loads: the per-lane offset is a constant of the thread, the stage and tap offsets are scalar (no 64-bit vector address
    // arithmetic per load - plain pointer arithmetic cost two VALU per load and kept address pairs live across the loop, which
    // spilled). The descriptor is built AT the load from the readfirstlane'd halves of the tile's weight pointer: carried across
    // the loop (or picked by a select) it lives in vector registers and every load gets a waterfall loop.
    auto b_load = [&](X8 (&dst)[3], const unsigned char* wp, int soff) __attribute__((always_inline)) {
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        const unsigned long long a = (unsigned long long)wp;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, (int)wtile, 0x00020000);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
            dst[ky] = __builtin_bit_cast(X8, (u32x4_t)__builtin_amdgcn_raw_buffer_load_b128(rs, lane_w, soff + ky * 8192, 0));
    };

    // byte offset of stage `st` (0..5 of a chunk, continuing into the next chunk / tile) inside a tile's packed weights
    // (128-channel tiles: [chunk][kx][k-step][ky 3][co 128][64 B], stages of 24 576 B; 64-channel tiles: [chunk][kx][ky 3][k-step 2][co 64][64 B]:
    //  a stage pair of 24 576 B, the k-step 4 096 B inside it; the tap ky 8 192 B in both)
    auto w_off = [&](int chunk, int st) __attribute__((always_inline)) -> int {
        if constexpr (N64) return (chunk * 3 + (st >> 1)) * kStageBytes + (st & 1) * 4096;
        else return (chunk * 6 + st) * kStageBytes;
    };

    // ---- halo staging (conv3x3.hip): thread -> 16-byte slot tid % 8 of halo pixels q + 54 i, q < 54, i = 0..5 (three halo rows per
    // step: the column hx = q % 18 is fixed, rows hy = q / 18 + 3 i). Threads 432..511 REPEAT the work of threads 352..431 (the same
    // values to the same addresses): no predicate, no branch inside a stage.
    const int slot = tid & 7, q64 = tid >> 3;
    const int q54 = q64 < 54 ? q64 : q64 - 10;
    const int qy = (int)(__umul24((unsigned)q54, 3641u) >> 16), qx = q54 - ((qy << 4) + (qy << 1));
    static_assert(kHW == 18 && kNPix == 6 * 54, "six steps of three halo rows");
    const unsigned act_lo2 = (AFFINE && g.job[0].in_act != MPN_ACT_NONE) ? 0u : 0x80008000u;     // (in_act is shared by the jobs)
    int tab_job = -1;
    auto tab_load = [&](int job) {
        if constexpr (AFFINE) {
            for (int i = tid; i < Cin; i += kThreads) { tab[i] = g.job[job].in_scale[i]; tab[kMaxCin + i] = g.job[job].in_shift[i]; }
        }
        if constexpr (BNR) {   // (no producer affine in a data gradient: the table holds the fed batch-norm's scale / shift)
            for (int i = tid; i < Cout; i += kThreads) { tab[i] = g.job[job].bnr_scale[i]; tab[kMaxCin + i] = g.job[job].bnr_shift[i]; }
        }
        tab_job = job;
    };
    unsigned okmask = 0;        // of the image fetched last: piece i of this thread lies inside the image
    Vec16<T> av[kAVec];
    auto a_load = [&](const Tile& t, int chunk) __attribute__((always_inline)) {
        const Job& p = g.job[t.job];
        // the image base is scalar, row and column BYTE offsets are 24-bit multiplies added as an unsigned 32-bit offset
        // (launch() checks 2 W xs < 2^24 and the tensor below 2^31 elements)
        const int wxb = p.W * p.xs * 2;
        const unsigned char* xb = reinterpret_cast<const unsigned char*>(p.x) + ((long long)t.img * p.H * (p.W * p.xs) + chunk * 64) * 2;
        int hx = qx, hy = qy;
        asm volatile("" : "+v"(hx), "+v"(hy));   // opaque: keeps the offset arithmetic here instead of hoisted (and spilled) across the tile loop
        const int ix = t.ox0 + hx - 1;
        const bool okx = (unsigned)ix < (unsigned)p.W;
        const unsigned col = __umul24((unsigned)min(max(ix, 0), p.W - 1), (unsigned)(p.xs * 2)) + slot * 16;
        okmask = 0;
#pragma unroll
        for (int i = 0; i < kAVec; ++i) {
            const int iy = t.oy0 + (3 * i - 1) + hy;
            const bool ok = okx & ((unsigned)iy < (unsigned)p.H);
            okmask |= (ok ? 1u : 0u) << i;
            av[i].load(reinterpret_cast<const T*>(xb + (__umul24((unsigned)min(max(iy, 0), p.H - 1), (unsigned)wxb) + col)));
        }
    };
    // The commit of piece i of the fetched image into halo buffer `buf`, in five steps that a stage spreads over its MFMA groups:
    // j = 0..3: dword j (two channels: affine + activation + zero padding), j = 4: the 16-byte LDS store. The scale / shift of the
    // chunk's channels at this thread's slot are read from the table first (tab_read).
    f32x2_t sc[4], sh[4];
    auto tab_read = [&](int chunk) __attribute__((always_inline)) {
        if constexpr (AFFINE) {
            const lds_p ts = (lds_p)(tab + chunk * 64 + slot * 8);
            const f32x4_t s0 = lds_ld<f32x4_t>(ts), s1 = lds_ld<f32x4_t>(ts + 16);
            const f32x4_t h0 = lds_ld<f32x4_t>(ts + kMaxCin * 4), h1 = lds_ld<f32x4_t>(ts + kMaxCin * 4 + 16);
            sc[0] = (f32x2_t){s0[0], s0[1]}; sc[1] = (f32x2_t){s0[2], s0[3]}; sc[2] = (f32x2_t){s1[0], s1[1]}; sc[3] = (f32x2_t){s1[2], s1[3]};
            sh[0] = (f32x2_t){h0[0], h0[1]}; sh[1] = (f32x2_t){h0[2], h0[3]}; sh[2] = (f32x2_t){h1[0], h1[1]}; sh[3] = (f32x2_t){h1[2], h1[3]};
        }
    };
    auto commit_step = [&](const int i, const int j, int buf) __attribute__((always_inline)) {
        if (j < 4) {
            unsigned u = j == 0 ? av[i].raw.x : (j == 1 ? av[i].raw.y : (j == 2 ? av[i].raw.z : av[i].raw.w));
            if constexpr (AFFINE) {
                f32x2_t f;
                if constexpr (std::is_same<T, bf16_t>::value) {
                    f = (f32x2_t){__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
                } else {
                    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
                    const h2_t hh = __builtin_bit_cast(h2_t, u);
                    f = (f32x2_t){(float)hh[0], (float)hh[1]};
                }
                f = __builtin_elementwise_fma(f, sc[j], sh[j]);
                if constexpr (std::is_same<T, bf16_t>::value) {
                    u = pack_bf16x2(f[0], f[1]);
                } else {
                    typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
                    const h2_t hh = {(_Float16)f[0], (_Float16)f[1]};
                    u = __builtin_bit_cast(unsigned, hh);
                }
                u = pk_max_i16(u, act_lo2);
                if constexpr (ACT == 2) u = pk_min_i16(u, six_pair<T>());
            }
            if (!((okmask >> i) & 1u)) u = 0u;
            if (j == 0) av[i].raw.x = u; else if (j == 1) av[i].raw.y = u; else if (j == 2) av[i].raw.z = u; else av[i].raw.w = u;
        } else {
            typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
            // (pixels q54 + 54 i share the column qx: the swizzle is a constant of the thread)
            lds_st<u32x4_t>(L + halo_off(buf) + q54 * kRS + ((slot ^ (qx & 7)) << 4) + i * (54 * kRS), (u32x4_t){av[i].raw.x, av[i].raw.y, av[i].raw.z, av[i].raw.w});
        }
    };

    // Statistics: a block SUMS the rows of its tiles of one job (threads 0..255: one of the two sums, one channel; f32, in the order of
    // its walk) and writes them once, behind its last tile of that job: one slab row per block (mpn_conv_stats_rows).
    float* st_dst = nullptr;
    int e_par = 0;             // which half of `red` the waiting tile's sums go to (0 / 256 words, alternating: a block reads the sums of
                               // tile t behind a barrier while faster waves may already write those of tile t + 1)
    float st_acc = 0.f;
    bool st_pending = false, st_store = false;
    const int st_which = (tid >> 7) & 1, st_c = tid & 127;
    auto stats_flush = [&]() {
        if (st_pending) {
            if constexpr (BNR) {
                const lds_f r = red + st_which * 128 + st_c;    // red [8 waves][2][128], fixed order
                st_acc += ((r[0] + r[256]) + (r[512] + r[768])) + ((r[1024] + r[1280]) + (r[1536] + r[1792]));
            } else {
                if constexpr (N64) st_acc += red[e_par + st_which * 64 + st_c] + red[e_par + 128 + st_which * 64 + st_c];   // [2 tiles][2 halves][2][64]
                else st_acc += red[e_par + st_which * 128 + st_c];   // red [2 tiles][2][128]: the tile's sums over its 256 pixels
            }
            if (st_store) { *st_dst = st_acc; st_acc = 0.f; }
        }
        st_pending = false;
    };

    // ---- the epilogue of a tile runs UNDER THE NEXT TILE'S FIRST THREE STAGES (its image lies over the halo buffer that the next
    // tile's first chunk does not read, and is released - one block barrier - before the commits of that chunk's stages 3..5 write
    // there): the state of the tile whose image is waiting
    // (a block's FIRST tile runs the same stages over an image that does not exist yet: prev_real = false masks its stores)
    bool prev_real = false;
    // everything its stores need, as scalars fixed when the image was written (a job field read inside a stage is an s_load whose
    // wait - lgkmcnt(0) - also drains the fragment reads in flight: 250-300 cycles per store in the first version):
    // the tile's first output element, bytes per image row / per pixel, rows and columns of the tile inside the image
    unsigned char* e_y = nullptr;
    int e_rowb = 0, e_pxb = 0, e_h = 0, e_w = 0;
    const unsigned char* e_bx = nullptr;       // BNR: the same for the raw tensor of the fed batch-norm
    int e_bx_rowb = 0, e_bx_pxb = 0, e_bx_hmax = 0, e_bx_wmax = 0, e_ntile = 0;
    float e_blo = 0.f, e_bhi = 0.f;
    // Statistics: wave `wave` sums ITS 16 channels over all 256 pixels, 32 pixels per MFMA pair.
    // transposing reads (ds_read_b64_tr_b16): lane 4q+pp of a 16-lane group supplies the address of block row q, channels 4pp..4pp+3;
    // lane group lq covers pixels 2 (4 lq + q) + {0, 1} of a 32-pixel group (the k order of a sum is free; the eight pixels of a
    // 32-lane half differ in bits 1..3: with the image's swizzle their 8-byte pieces cover the 256-byte bank row once)
    // (64-channel tiles: 128-byte pixels; a wave sums its 16 channels over ITS 128 pixels - four groups of 32 -, lane group lq the
    //  pixels 4 lq + q and + 16: four even and four odd pixels per 32-lane half, each on its own pair of slots)
    const int tP = N64 ? lq * 4 + (l15 >> 2) : 2 * (lq * 4 + (l15 >> 2));
    const int tQ = N64 ? tP + 16 : tP + 1;
    const int t_lo = kImgOff + (ph * 128 + tP) * kPx + (((cg * 2 + ((l15 & 3) >> 1)) ^ (tP & kSw)) << 4) + (l15 & 1) * 8;
    const int t_hi = kImgOff + (ph * 128 + tQ) * kPx + (((cg * 2 + ((l15 & 3) >> 1)) ^ (tQ & kSw)) << 4) + (l15 & 1) * 8;
    // Copy-out: wave `wave` stores pixels 32 wave .. + 31 (output rows 2 wave, 2 wave + 1), all channels of the tile: lane = 16-byte piece
    // `piece` of pixels prow + 4 k, k = 0..7; its slot piece ^ (px & 15) = (piece ^ prow) ^ 4 (k & 3): bits 6, 7 of the byte offset
    // (64-channel tiles: eight pieces per pixel, pixels prow + 8 k, k = 0..3: the slot piece ^ prow is the same for every k)
    const int piece = N64 ? (lane & 7) : (lane & 15), prow = N64 ? (lane >> 3) : (lane >> 4);
    const int c_off = kImgOff + (wave * 32 + prow) * kPx + ((piece ^ prow) << 4);
    // (+ e_par; 128-channel tiles: [2][128]; 64-channel tiles: [2 pixel halves][2][64]; the dummy words: 2048..2111)
    const int red_s = N64 ? ph * 128 + cg * 16 + l15 : wave * 16 + l15;
    const int red_q = (lq == (l15 >> 2)) ? (N64 ? ph * 128 + 64 + cg * 16 + l15 : 128 + wave * 16 + l15) : 2048 + lane;
    f32x4_t st_sa = {0.f, 0.f, 0.f, 0.f}, st_ga = {0.f, 0.f, 0.f, 0.f};
    X8 ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = 1.0f;
    float dsel[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) dsel[j] = (l15 & 3) == j ? 1.f : 0.f;
    auto tr_read_lds = [&](lds_p p) __attribute__((always_inline)) -> typename H::x4 {
        if constexpr (std::is_same<T, bf16_t>::value) return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS typename H::x4*)p);
        else {
            typedef __fp16 fp4_t __attribute__((ext_vector_type(4)));
            return __builtin_bit_cast(typename H::x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((LDS fp4_t*)p));
        }
    };
    typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
    auto pack4 = [&](const f32x4_t& v) __attribute__((always_inline)) -> u32x2_t {
        if constexpr (std::is_same<T, bf16_t>::value) return (u32x2_t){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        else {
            typedef _Float16 h4_t __attribute__((ext_vector_type(4)));
            const h4_t hh = {(_Float16)v[0], (_Float16)v[1], (_Float16)v[2], (_Float16)v[3]};
            return __builtin_bit_cast(u32x2_t, hh);
        }
    };
    typename H::x4 tlo, thi;
    uint4 co[2], bx[BNR ? (N64 ? 4 : 8) : 1];      // (128-channel tiles: all eight pieces of a lane in flight from the tile's last stage on)
    f32x2_t bs[4], bq[4];
    auto bnr_ld = [&](const int k) __attribute__((always_inline)) -> uint4 {
        // (rows / columns past the image: clamped - their dy is 0)
        const int oy = min(2 * wave + (N64 ? (k >> 1) : (k >> 2)), e_bx_hmax), ox = min(prow + (N64 ? 8 * (k & 1) : 4 * (k & 3)), e_bx_wmax);
        return *reinterpret_cast<const uint4*>(e_bx + (unsigned)(oy * e_bx_rowb + ox * e_bx_pxb + piece * 16));
    };
    // step (sl, h) of the waiting epilogue: sl = 0: statistics (two transposing reads at even h, the two MFMAs + the sums' rows at
    // odd h; BNR: the raw tensor of the fed batch-norm at the first copy-out positions); sl = 1, 2: the copy-out of pixel rows
    // prow + 4 k, k = 4 (sl - 1) .. + 3 (LDS reads at h = 0..3, stores at h = 6, 8, 10, 12); BNR: the sums at the end of sl = 2
    auto e2_step = [&](const int sl, const int h) __attribute__((always_inline)) {
        if constexpr (STATS) {
            constexpr int NPG = N64 ? 4 : 8;      // groups of 32 pixels a wave sums
            if (sl == 0 && h < 2 * NPG) {
                const int pg = h >> 1;       // pixels 32 pg .. + 31 (of the wave's half)
                if ((h & 1) == 0) {
                    tlo = tr_read_lds(L + t_lo + pg * (32 * kPx));
                    thi = tr_read_lds(L + t_hi + pg * (32 * kPx));
                } else {
                    const X8 f = __builtin_shufflevector(tlo, thi, 0, 1, 2, 3, 4, 5, 6, 7);
                    st_sa = H::mfma(ones, f, pg == 0 ? (f32x4_t){0.f, 0.f, 0.f, 0.f} : st_sa);
                    st_ga = H::mfma(f, f, pg == 0 ? (f32x4_t){0.f, 0.f, 0.f, 0.f} : st_ga);
                }
            }
            if (sl == 0 && h == 2 * NPG + 1) {
                // the Gram diagonal sits in element l15 & 3 of the lanes lq == l15 >> 2: picked by a 0 / 1 weight per element (exact:
                // the others add + 0; written as a select chain hipcc made three branches of it)
                const float q = __builtin_fmaf(st_ga[3], dsel[3], __builtin_fmaf(st_ga[2], dsel[2], __builtin_fmaf(st_ga[1], dsel[1], st_ga[0] * dsel[0])));
                red[e_par + red_s] = st_sa[0];       // (every row of ones x F is the column sum: the four lane groups write the same value)
                red[(lq == (l15 >> 2) ? e_par : 0) + red_q] = q;     // (the Gram diagonal; the other lanes into the dummy words)
            }
        }
        if constexpr (BNR) {
            if (sl == 0 && h == 10) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { bs[j] = (f32x2_t){0.f, 0.f}; bq[j] = (f32x2_t){0.f, 0.f}; }
            }
            // (the raw tensor's first four pieces were requested under the tile's last stage: chunk_body)
        }
        if (sl == 1 || (sl == 2 && !N64)) {
            // pixel rows prow + 4 k, k = 4 (sl - 1) + kk: the LDS read of row kk at h = 3 kk + 1, its store at h = 3 kk + 6 (two rows in
            // flight); BNR: the mask and the sums of its dwords 0, 1 at h = 3 kk + 4, of dwords 2, 3 at h = 3 kk + 5
            const int half = sl - 1;
            const int kk = h >= 6 ? (h - 6) / 3 : -1, ph = h >= 4 ? (h - 4) % 3 : -1;     // (of the row whose math / store falls on this h)
            if (h >= 1 && h <= 10 && (h - 1) % 3 == 0) {
                typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
                const int kr = (h - 1) / 3;
                const u32x4_t v = N64 ? lds_ld<u32x4_t>(L + c_off + (8 * kr) * kPx) : lds_ld<u32x4_t>(L + (c_off ^ (kr * 64)) + (4 * (half * 4 + kr)) * kPx);
                co[kr & 1] = make_uint4(v[0], v[1], v[2], v[3]);
            }
            if constexpr (BNR) {
                // g = dy where the fed batch-norm's activation passes (lo < x * scale + shift < hi, the test of bn_bwd_reduce /
                // bn_bwd_apply, fused multiply-add), else 0; sums of g and g * x (pixels outside the image hold dy = 0).
                // The batch-norm's scale / shift of this lane's channels come from the table at every use (held in registers
                // they spill; the opaque offset keeps hipcc from hoisting the reads back out).
                if (h >= 4 && h <= 14 && (ph == 0 || ph == 1)) {
                    const int km = (h - 4) / 3;
                    int toff = (e_ntile * CT + piece * 8 + ph * 4) * 4;
                    asm volatile("" : "+v"(toff));
                    const f32x4_t s4 = lds_ld<f32x4_t>((lds_p)tab + toff), h4 = lds_ld<f32x4_t>((lds_p)tab + toff + kMaxCin * 4);
                    const uint4 xv = bx[(N64 ? 0 : half * 4) + km];
                    unsigned du[2] = {ph == 0 ? co[km & 1].x : co[km & 1].z, ph == 0 ? co[km & 1].y : co[km & 1].w};
                    const unsigned xu[2] = {ph == 0 ? xv.x : xv.z, ph == 0 ? xv.y : xv.w};
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        const int j = ph * 2 + jj;
                        const f32x2_t bsc = {s4[2 * jj], s4[2 * jj + 1]}, bsh = {h4[2 * jj], h4[2 * jj + 1]};
                        const f32x2_t xf = {to_f32(__builtin_bit_cast(T, (unsigned short)(xu[jj] & 0xffffu))),
                                            to_f32(__builtin_bit_cast(T, (unsigned short)(xu[jj] >> 16)))};
                        const f32x2_t pre = xf * bsc + bsh;
                        const unsigned m = ((pre[0] > e_blo && pre[0] < e_bhi) ? 0x0000ffffu : 0u) | ((pre[1] > e_blo && pre[1] < e_bhi) ? 0xffff0000u : 0u);
                        du[jj] &= m;
                        const f32x2_t gf = {to_f32(__builtin_bit_cast(T, (unsigned short)(du[jj] & 0xffffu))),
                                            to_f32(__builtin_bit_cast(T, (unsigned short)(du[jj] >> 16)))};
                        bs[j] += gf;
                        bq[j] += gf * xf;
                    }
                    if (ph == 0) { co[km & 1].x = du[0]; co[km & 1].y = du[1]; } else { co[km & 1].z = du[0]; co[km & 1].w = du[1]; }
                }
            }
            if (h >= 6 && h <= 15 && (h - 6) % 3 == 0) {
                // whole 256-byte pixel rows to HBM (four complete pixels per instruction): scalar base + a 32-bit lane offset.
                // Edge tiles: pixels outside the image are not stored (one compare per store, every lane passes on full tiles);
                // a block's first tile has no image yet (prev_real).
                const int k = half * 4 + kk;
                const uint4 o = co[kk & 1];
                const int row = 2 * wave + (N64 ? (k >> 1) : (k >> 2)), col = N64 ? 8 * (k & 1) : 4 * (k & 3);
                unsigned char* yb = e_y + ((long long)row * e_rowb + col * e_pxb);       // (scalar)
                const unsigned lane_off = (unsigned)(prow * e_pxb + piece * 16);
                if (prev_real && row < e_h && col + prow < e_w) *reinterpret_cast<uint4*>(yb + lane_off) = o;
            }
        }
        if constexpr (BNR) {
            if (sl == 2 && h >= 14 && h < 18) {
                // the four lanes of one 16-byte piece (lane bits 4, 5): fixed butterfly, one channel pair per step (all four at once
                // spill), then all four lane groups hold the wave's sums of their channels and write the same values to the same words
                const int j = h - 14;
#pragma unroll
                for (int o = N64 ? 8 : 16; o < 64; o <<= 1) {
                    bs[j][0] += __shfl_xor(bs[j][0], o, 64); bs[j][1] += __shfl_xor(bs[j][1], o, 64);
                    bq[j][0] += __shfl_xor(bq[j][0], o, 64); bq[j][1] += __shfl_xor(bq[j][1], o, 64);
                }
                const int cl = piece * 8 + 2 * j;
                lds_st<f32x2_t>((lds_p)(red + wave * 256 + cl), bs[j]);
                lds_st<f32x2_t>((lds_p)(red + wave * 256 + 128 + cl), bq[j]);
            }
        }
    };
    // behind the barrier that ends an epilogue: its sums into the block's running sums, another job's batch-norm table
    auto e2_done = [&](int next_job) __attribute__((always_inline)) {
        stats_flush();
        if constexpr (BNR) { if (next_job != tab_job) tab_load(next_job); }
    };

    Tile cur = tile_of<N64>(g, w);
    if constexpr (BNR) { e_bx = reinterpret_cast<const unsigned char*>(g.job[cur.job].bnr_x); }     // (a block's first tile: loads that nothing uses, from a valid address)
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(g.job[cur.job].wp) + cur.ntile * wtile;
    // weight fragments requested this many stages ahead (BD + 1 register sets): on 128-channel tiles two stages ahead measured equal
    // (profiles/r05_c3cs_ab.txt); a 64-channel tile's stage is half as long (24 MFMAs per wave): two ahead, 1.5-2.5 % faster on the
    // detector's 64 -> 64 towers (profiles/r06_c3_n64_ab.txt)
    constexpr int BD = N64 ? 2 : 1;
    X8 b[BD + 1][3];
    int cc = 0;                // running chunk counter: the chunk reads halo buffer cc & 1
    b_load(b[0], wsrc, 0);
    if constexpr (BD == 2) b_load(b[1], wsrc, w_off(0, 1));
    a_load(cur, 0);
    if constexpr (AFFINE || BNR) tab_load(cur.job);
    __syncthreads();           // table visible
    tab_read(0);
#pragma unroll
    for (int i = 0; i < kAVec; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) commit_step(i, j, 0);
    __syncthreads();

    Tile nxt_c = tile_fast<N64>(g, w + (int)gridDim.x < total ? w + (int)gridDim.x : w);
    const unsigned char* wsrc_next_c = reinterpret_cast<const unsigned char*>(g.job[nxt_c.job].wp) + nxt_c.ntile * wtile;
    for (;;) {
        CS_STAMP(0);
        // (the NEXT tile's coordinates and weight pointer were computed under the previous tile's stages - their scalar loads and
        //  divisions stood in front of every tile's first MFMA in all eight waves at once; the tile after it is computed under this one)
        const int wnext = w + (int)gridDim.x;
        const bool has_next = wnext < total;
        const Tile nxt = nxt_c;
        const unsigned char* wsrc_next = wsrc_next_c;

        f32x4_t acc[MT];     // (not zeroed: a tile's first MFMA into each tile takes a literal 0 as its C operand - 64 v_mov per tile less)

        // ONE chunk: six stages = (kx, k-step); per stage 48 MFMAs in 18 groups (halo row h feeds output rows h, h - 1, h - 2), each
        // followed by the read of row h + kRing and by whatever else the stage carries (sched_barrier pins every group: the order
        // below IS the instruction order): stage 0 the fetch of the next halo image, stages 3..5 its commit (two pieces per stage),
        // stages 0..2 of a tile's FIRST chunk the previous tile's epilogue. (The two variants run one after the other, never as the
        // two sides of a branch: hipcc hoists what both sides share in front of the branch - 74 more live registers.)
        // (the weight pointers by value: captured by reference they stayed in scratch memory, one flat load + vmcnt(0) per stage)
        auto chunk_body = [&](const int chunk, const unsigned char* const ws, const unsigned char* const ws_next, auto epi_tag, auto first_tag) __attribute__((always_inline)) {
            constexpr bool EPI = decltype(epi_tag)::value, FIRST = decltype(first_tag)::value;     // FIRST: the tile's first chunk
            const lds_p hb = L + halo_off(cc & 1) + ph * (8 * kHW * kRS);       // (this wave's first halo row)
            const bool last_chunk = chunk + 1 == nchunk;
            // the image to prepare under this chunk: the tile's next chunk, or the first chunk of the next tile (without a next tile
            // this tile's first chunk once more, never read: the staging stays unconditional)
            const Tile& st_tile = last_chunk ? nxt : cur;
            const int st_chunk = last_chunk ? 0 : chunk + 1;
            if constexpr (BNR) {
                // the fused reduction's epilogue runs BEHIND the tile (not under the next one): where it reads the fed batch-norm's raw tensor
                // is known now, and its first four pieces are requested under the tile's last stage instead of in front of their use
                // (round 6: ~2 k cycles of exposed load latency per tile)
                if (last_chunk) {
                    const Job& p = g.job[cur.job];
                    const int wbs = p.W * p.bnr_xs;
                    e_bx = reinterpret_cast<const unsigned char*>(reinterpret_cast<const T*>(p.bnr_x) + ((long long)cur.img * p.H + cur.oy0) * wbs + (long long)cur.ox0 * p.bnr_xs + cur.ntile * CT);
                    e_bx_rowb = wbs * 2; e_bx_pxb = p.bnr_xs * 2; e_bx_hmax = p.H - 1 - cur.oy0; e_bx_wmax = p.W - 1 - cur.ox0;
                }
            }
            if constexpr (AFFINE) {
                // another job's table (a few times per launch): no wave commits between a chunk's barrier and its stage 3
                if (last_chunk && nxt.job != tab_job) {
                    tab_load(nxt.job);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
            }
#pragma unroll
            for (int sl = 0; sl < 6; ++sl) {
                const lds_p ab = hb + (abase_kx[sl >> 1] ^ ((sl & 1) * 64));
                // the next stage's weight fragments (of this tile, or the first ones of the next tile) into the other register set
                if (sl + BD < 6) b_load(b[(sl + BD) % (BD + 1)], ws, w_off(chunk, sl + BD));
                else b_load(b[(sl + BD) % (BD + 1)], last_chunk ? ws_next : ws, last_chunk ? w_off(0, sl + BD - 6) : w_off(chunk, sl + BD));
                if (sl == 0) a_load(st_tile, st_chunk);
                X8 a[HR];
#pragma unroll
                for (int h = 0; h < kRing; ++h) a[h] = lds_ld<X8>(ab + h * (kHW * kRS));
                if (sl == 3) tab_read(st_chunk);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int h = 0; h < HR; ++h) {
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const int r = h - ky;
                        if (r >= 0 && r < MT)       // D^T = W^T x A^T
                            acc[r] = H::mfma(b[sl % (BD + 1)][ky], a[h], (FIRST && sl == 0 && ky == 0) ? (f32x4_t){0.f, 0.f, 0.f, 0.f} : acc[r]);
                    }
                    if (h + kRing < HR) a[h + kRing] = lds_ld<X8>(ab + (h + kRing) * (kHW * kRS));
                    if (sl >= 3) {
                        // pieces 2 (sl - 3) at h = 1..5 and 2 (sl - 3) + 1 at h = 9..13 (64-channel tiles, ten groups: h = 0..4, 5..9)
                        constexpr int c0 = N64 ? 0 : 1, c1 = N64 ? 5 : 9;
                        if (h >= c0 && h < c0 + 5) commit_step(2 * (sl - 3), h - c0, (cc + 1) & 1);
                        if (h >= c1 && h < c1 + 5) commit_step(2 * (sl - 3) + 1, h - c1, (cc + 1) & 1);
                    }
                    if constexpr (BNR) {
                        if (last_chunk && sl == 5 && h >= 2 && h < 6) bx[h - 2] = bnr_ld(h - 2);
                        if (!N64 && last_chunk && sl == 5 && h >= 10 && h < 14) bx[h - 6] = bnr_ld(h - 6);
                    }
                    if (EPI && sl < 3) {
                        if constexpr (N64) {       // (the epilogue's 18 steps on ten groups: an odd step - MFMAs, stores - with the NEXT even one's reads)
                            if (h >= 1) e2_step(sl, 2 * h - 1);
                            if (h < 9) e2_step(sl, 2 * h);
                        }
                        else e2_step(sl, h);
                    }
                    if (FIRST && sl == 1 && h == 2) {
                        // the tile after the next (or, at the end of the walk, a tile this block already owns: never used)
                        const int w2 = wnext + (int)gridDim.x;
                        nxt_c = tile_fast<N64>(g, w2 < total ? w2 : w);
                        wsrc_next_c = reinterpret_cast<const unsigned char*>(g.job[nxt_c.job].wp) + nxt_c.ntile * wtile;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                CS_STAMP(16 + (chunk < 1 ? chunk : 1) * 6 + sl);
                if (EPI && sl == 2 && last_chunk) {
                    // (a one-chunk tile writes ITS image at the end of this chunk: every wave must have read the waiting one first)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    CS_STAMP(10);
                    e2_done(cur.job);
                }
            }
            if (last_chunk) {
                // ================= tile `cur` into its image, by every wave as it finishes - no barrier in front (the image is nobody's
                // alias; the tile before it was read under this tile's FIRST chunk, a barrier ago), the chunk's barrier behind.
                // Output row r, pixel l15, this wave's channels lq * 4 .. + 3; slot (2 wave + lq / 2) ^ l15 of the pixel's sixteen.
                // Pixels outside the image must not count in the statistics: zeroed on edge tiles.
                const Job& p = g.job[cur.job];
                const bool full_tile = cur.oy0 + 16 <= p.H && cur.ox0 + 16 <= p.W;
                // (64-channel tiles: rows 8 ph + r, r < 8; 128-byte pixels, slot (2 cg + lq / 2) ^ (l15 & 7))
                const lds_p iw = L + kImgOff + (ph * 128 + l15) * kPx + (((cg * 2 + (lq >> 1)) ^ (l15 & kSw)) << 4) + (lq & 1) * 8;
                if (full_tile) {
#pragma unroll
                    for (int r = 0; r < MT; ++r) lds_st<u32x2_t>(iw + r * (16 * kPx), pack4(acc[r]));
                } else {
                    const bool okx = cur.ox0 + l15 < p.W;
#pragma unroll
                    for (int r = 0; r < MT; ++r) {
                        const bool ok = okx && cur.oy0 + ph * 8 + r < p.H;
                        lds_st<u32x2_t>(iw + r * (16 * kPx), pack4(ok ? acc[r] : (f32x4_t){0.f, 0.f, 0.f, 0.f}));
                    }
                }
                prev_real = true;
                e_par ^= 256;
                {
                    const int wys = p.W * p.ys;
                    e_y = reinterpret_cast<unsigned char*>(reinterpret_cast<T*>(p.y) + ((long long)cur.img * p.H + cur.oy0) * wys + (long long)cur.ox0 * p.ys + cur.ntile * CT);
                    e_rowb = wys * 2; e_pxb = p.ys * 2; e_h = p.H - cur.oy0; e_w = p.W - cur.ox0; e_ntile = cur.ntile;
                    if constexpr (BNR) {      // (e_bx and its strides: at the top of the chunk)
                        e_blo = p.bnr_act != MPN_ACT_NONE ? 0.f : -INFINITY;
                        e_bhi = p.bnr_act == MPN_ACT_RELU6 ? 6.f : INFINITY;
                    }
                }
                if (STATS || BNR) {
                    // the block's row of this job's slab: row = the position of the block's FIRST tile of the job among the job's first
                    // gridDim.x tiles (mpn_conv_stats_rows; the grid is a multiple of n_tiles, so a block keeps its channel tile within a job)
                    const int n_tiles = N64 ? (Cout >> 6) : (Cout >> 7), grid = (int)gridDim.x;
                    int off = (w_first - g.begin[cur.job]) % grid;
                    if (off < 0) off += grid;
                    st_pending = tid < 256 && (!N64 || st_c < 64);
                    st_store = !has_next || nxt.job != cur.job;
                    st_dst = p.stats_part + ((long long)(off / n_tiles) * 2 + st_which) * Cout + cur.ntile * CT + st_c;
                }
                CS_STAMP(9);
            }
            // the chunk's barrier: every wave has read this halo image for the last time and committed its pieces of the next one;
            // behind a tile's last chunk its image is complete
            // (raw: only LDS traffic is ordered - the weight fragments in flight and the tile's output stores stay in flight)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            CS_STAMP(1 + (chunk < 7 ? chunk : 7));
            ++cc;
            if (EPI && !last_chunk) {
                // every wave has read the waiting image and written its sums
                CS_STAMP(10);
                e2_done(cur.job);
            }
        };
        if constexpr (PIPE) chunk_body(0, wsrc, wsrc_next, std::true_type{}, std::true_type{});
        else chunk_body(0, wsrc, wsrc_next, std::false_type{}, std::true_type{});
        for (int chunk = 1; chunk < nchunk; ++chunk) chunk_body(chunk, wsrc, wsrc_next, std::false_type{}, std::false_type{});

        {
            if constexpr (!PIPE) {
#pragma unroll
                for (int sl = 0; sl < 3; ++sl)
#pragma unroll
                    for (int h = 0; h < 18; ++h) { e2_step(sl, h); __builtin_amdgcn_sched_barrier(0); }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                e2_done(nxt.job);
            }
        }
#ifdef MPN_DIAG
        ++titer;
#endif
        if (!has_next) break;
        cur = nxt;
        wsrc = wsrc_next;
        w = wnext;
    }
    // the last tile's epilogue, on its own
    if constexpr (PIPE) {
#pragma unroll
        for (int sl = 0; sl < 3; ++sl)
#pragma unroll
            for (int h = 0; h < 18; ++h) { e2_step(sl, h); __builtin_amdgcn_sched_barrier(0); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        stats_flush();
    }
#ifdef MPN_DIAG
    if (g.job[0].dbg) {
        if ((threadIdx.x & 63) == 0) { wst[(threadIdx.x >> 6) * 32 + 12] = rt_begin; wst[(threadIdx.x >> 6) * 32 + 13] = __builtin_amdgcn_s_memrealtime(); }
        __syncthreads();
        if (threadIdx.x < 256) g.job[0].dbg[(size_t)blockIdx.x * 256 + threadIdx.x] = wst[threadIdx.x];
    }
#endif
}

template <typename T, int ACT, int MODE, bool N64 = false>
int launch_t(const Group& g, int blocks, hipStream_t st) {
    static mpn_attr_mask_t attr_mask{0};
    MPN_HIP(mpn_ensure_dynamic_lds((const void*)conv3x3_cs_kernel<T, ACT, MODE, N64>, kLds, &attr_mask));
    conv3x3_cs_kernel<T, ACT, MODE, N64><<<dim3((unsigned)blocks), dim3(kThreads), kLds, st>>>(g);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}
template <typename T, int ACT, bool N64>
int launch_m(const Group& g, int blocks, bool stats, hipStream_t st) {
    return stats ? launch_t<T, ACT, 1, N64>(g, blocks, st) : launch_t<T, ACT, 0, N64>(g, blocks, st);
}
template <typename T, bool N64>
int launch_v(const Group& g, int blocks, bool affine, bool bnr, hipStream_t st) {
    if (bnr) return launch_t<T, 0, 2, N64>(g, blocks, st);
    const bool stats = g.job[0].stats_part != nullptr;
    if (!affine) return launch_m<T, 0, N64>(g, blocks, stats, st);
    return g.job[0].in_act == MPN_ACT_RELU6 ? launch_m<T, 2, N64>(g, blocks, stats, st) : launch_m<T, 1, N64>(g, blocks, stats, st);
}

}  // namespace

namespace mpn_c3 {

int launch_cs(const Group& g, int blocks, int dtype, bool affine, bool bnr, bool n64, hipStream_t st) {
    // (the table holds the producer's affine per INPUT channel, or - fused reduction - the fed batch-norm's per OUTPUT channel; nothing else
    //  is sized by the channel counts)
    MPN_REQUIRE(g.job[0].Cin <= kMaxCin, MPN_ERR_BAD_SHAPE, "conv3x3: at most %d input channels", kMaxCin);
    MPN_REQUIRE(!bnr || g.job[0].Cout <= kMaxCin, MPN_ERR_BAD_SHAPE, "conv3x3: the fused reduction takes at most %d output channels", kMaxCin);
    for (int j = 1; j < g.njobs; ++j)
        MPN_REQUIRE((g.job[j].stats_part != nullptr) == (g.job[0].stats_part != nullptr), MPN_ERR_BAD_ARG,
                    "conv3x3: the jobs of a group share the statistics / no statistics variant");
    if (dtype == MPN_BF16) return n64 ? launch_v<bf16_t, true>(g, blocks, affine, bnr, st) : launch_v<bf16_t, false>(g, blocks, affine, bnr, st);
    if (dtype == MPN_F16) return n64 ? launch_v<half_t, true>(g, blocks, affine, bnr, st) : launch_v<half_t, false>(g, blocks, affine, bnr, st);
    MPN_FAIL(MPN_ERR_BAD_DTYPE, "conv3x3: 16-bit storage only");
}

}  // namespace mpn_c3
