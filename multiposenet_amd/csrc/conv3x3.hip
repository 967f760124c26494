// 3x3 convolution (forward and data gradient), 16-bit storage, on v_mfma_f32_16x16x32_{bf16,f16}: ONE 8-wave block per CU owns a
// 16 x 16 pixel tile x 64 output channels - the kernel of the 64-CHANNEL TILES THAT ARE DEEPER THAN ONE 64-CHANNEL CHUNK (Cout an odd
// multiple of 64, Cin > 64: final_conv3x3's forward 512 -> 64, keypoint_subnet.py:38). Rounds 2-4 ran every 3x3 shape here; since
// round 5 the 128-channel tiles and the one-chunk 64-channel tiles (the detector's 64 -> 64 towers, box_predictor.py:101-103) run on
// the channel-split kernel of conv3x3_cs.hip, and round 6 removed this file's 128-channel form (its history: docs/DESIGN_history_r1-r5.md
// 4f-4i). What is left is the variant that still wins its shape (profiles/r06_c3_n64_ab.txt): it shares a stage's weights through LDS
// among the eight waves, where the channel-split kernel's 64-channel tile pulls them from L2 at twice its 128-channel tile's rate.
//   * a weight stage holds the three taps of one kernel COLUMN (ky = 0..2 at fixed kx) for ALL 64 input channels of the chunk
//     ([ky 3][k-step 2][co 64][64 bytes] = 24 576 bytes by LDS-DMA, two buffers): output row r at tap ky reads the same halo row r + ky
//     as output row r + 1 at tap ky - 1, so a wave loads the 6 halo rows of its 4 output rows ONCE per stage and uses each fragment for
//     up to three taps: 18 fragment reads per 48 MFMAs; 3 stages per chunk;
//   * waves = 4 row groups (4 output rows each) x 2 K halves: wave (wm, wk) multiplies k-step wk of every stage; at the end of a tile the
//     pair adds its two partial accumulators through LDS (each wave hands over the half of the rows it will not finish: 8 KB per wave,
//     one block barrier) and finishes 32 pixels x 64 channels each;
//   * the 18 x 18 halo image of the next 64-channel chunk is fetched into registers and committed (producer's batch-norm affine +
//     activation + zero padding) into the other half of a double-buffered LDS image while the current chunk is multiplied.
// LDS: 2 x 51 840 (halo images, 160-byte rows: conflict-free ds_read_b128) + 2 x 24 576 (weight stages) = 152 832 B.
// Epilogue: wave-local through a bf16 LDS image (whole 128-byte pixel rows to HBM), the batch-norm partial sums from the matrix unit
// (ones x F and F^T x F over transposed reads of that image), summed over a block's tiles of a layer: one row of the partial slab
// per block (mpn_conv_stats_rows).
#include "conv3x3.h"
#include <type_traits>

namespace {
using namespace mpn_c3;

constexpr int kHW = 18;                      // halo width = height
constexpr int kNPix = kHW * kHW;             // 324
constexpr int kRS = 160;                     // LDS bytes per halo pixel: 128 bytes of K + 32 of padding
constexpr int kABytes = kNPix * kRS;         // 51 840
constexpr int kSmem = 2 * kABytes + 2 * kStageBytes;
constexpr int kAVec = (kNPix * 8 + kThreads - 1) / kThreads;   // 16-byte pieces of a halo image per thread: 6
constexpr int kRedBytes = 4 * 2 * 128 * (int)sizeof(float);
constexpr int kMaxCin = 512;                 // scale / shift table of the producer's batch-norm: [2][kMaxCin] floats
constexpr int kTabBytes = 2 * kMaxCin * (int)sizeof(float);
#ifdef MPN_DIAG
constexpr int kDiagBytes = 8 * 13 * 3 * 8;   // per-wave stage stamps (below)
#else
constexpr int kDiagBytes = 0;
#endif
constexpr int kLds = kSmem + kRedBytes + kTabBytes + kDiagBytes;
static_assert(8 * 32 * (64 * 2 + 8) <= kABytes, "the wave-private epilogue images fit in one halo buffer");
static_assert(kLds <= 160 * 1024, "LDS budget");
static_assert(kAVec == 6, "the counted wait of a chunk's first stage assumes six halo loads per thread");

// diagnostic build (tools/stamp_c3.py): s_memtime of this block's THIRD tile at the loop top (0), after each chunk (1, 2, ...)
// and after the epilogue (7)
// knock-outs (tools/ko_c3.sh; never in the shipped library): MPN_KO & 1 = no epilogue, & 2 = no halo staging inside the tile loop,
// & 8 = every halo load from image 0 / tile 0 (cache-hot), & 16 = epilogue without the global stores
#ifndef MPN_KO
#define MPN_KO 0
#endif
// and, per WAVE, three stamps around the end of each of that tile's first 12 stages (before the counted wait, after it, after the
// barrier) plus the start and the end of its epilogue: kept in LDS (a global store inside the loop would change the vector-memory
// counter the waits count), copied out behind the block's slots at the end: dbg[256 * 8 + ((block * 8 + wave) * 13 + stage) * 3 + k]
#ifdef MPN_DIAG
#define C3_STAMP(k) do { if (g.job[0].dbg && threadIdx.x == 0 && titer == 2) g.job[0].dbg[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define C3_WSTAMP(st, k) do { if (g.job[0].dbg && titer == 2 && (st) < 13 && (threadIdx.x & 63) == 0) \
    wst[((threadIdx.x >> 6) * 13 + (st)) * 3 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define C3_STAMP(k) do { } while (0)
#define C3_WSTAMP(st, k) do { } while (0)
#endif

// A 16-byte LDS store the compiler does not see as one: in front of a C++ LDS store hipcc waits for every LDS-DMA in
// flight (vmcnt(0): it cannot tell that the weight ring and the halo image do not overlap) - here that exposed the whole
// latency of the weight stage requested at the top of the same stage, once per chunk.
template <int OFF>
__device__ __forceinline__ void lds_store16_raw(unsigned char* p, const uint4& v) {
    static_assert(OFF >= 0 && OFF < 65536, "16-bit immediate offset");
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    const u32x4_t d = {v.x, v.y, v.z, v.w};
    const unsigned a = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)p;
    asm volatile("ds_write_b128 %0, %1 offset:%2" :: "v"(a), "v"(d), "n"(OFF) : "memory");
}

// PERSISTENT: block b walks tiles b, b + gridDim.x, ... of the group. Across tiles nothing drains: the weight stream and
// the halo double buffer run on (running stage / chunk counters pick the buffers), the next tile's first halo image is
// staged under the current tile's last chunk, and the epilogue is wave-local (no block barrier): each wave converts its
// 64 pixels x 64 channels, stores them straight from the registers (8 bytes per lane: the four 16-channel pieces of a
// pixel's 128 bytes come from four consecutive stores of one wave and merge in the L2) and - when statistics are asked for -
// takes them from a wave-private LDS image of 32 pixels at a time in the halo buffer that has just been released.
template <typename T, bool AFFINE, bool BNR = false>
__global__ __launch_bounds__(kThreads, 1) void conv3x3_kernel(const Group g) {
    static_assert(!BNR || !AFFINE, "the fused batch-norm backward reduction rides on a data gradient (no producer affine)");
    constexpr int NS = 3;     // weight stages per 64-channel chunk (one per kernel column, both k-steps inside)
    constexpr int BN = 64;    // output channels per tile
    using H = H16<T>;
    using X8 = typename H::x8;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;                      // [2][324 halo pixels][160]
    unsigned char* Bs = smem + 2 * kABytes;        // [2][ky 3][128 co][64]
    float* red = reinterpret_cast<float*>(smem + kSmem);   // [4 wm][2][128] statistics of the tile that has just finished
    float* tab = reinterpret_cast<float*>(smem + kSmem + kRedBytes);   // [2][kMaxCin] scale, shift of the job in `tab_job`
#ifdef MPN_DIAG
    unsigned long long* wst = reinterpret_cast<unsigned long long*>(smem + kSmem + kRedBytes + kTabBytes);
    for (int i = threadIdx.x; i < 8 * 13 * 3; i += kThreads) wst[i] = 0;
#endif

    const int total = g.begin[g.njobs];
    // XCD-aware walk: the dispatcher deals consecutive block ids round-robin over the 8 XCDs (one L2 each); the blocks of one XCD
    // take CONSECUTIVE tiles of every round, so that the halo rows and columns neighbouring tiles share are read through one L2
    // instead of from the memory side (measured: 174 MB fetched per launch for 134 MB of input with the plain walk)
    int w = blockIdx.x;
    if ((gridDim.x & 7) == 0) w = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (w >= total) return;
    const int w_first = w;
#ifdef MPN_DIAG
    if (g.job[0].dbg && threadIdx.x == 0) g.job[0].dbg[(size_t)blockIdx.x * 8 + 4] = __builtin_amdgcn_s_memrealtime();
#endif

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;       // 4 waves along the tile's rows (4 rows each) x 2 along the two k-steps of a stage
    const int l15 = lane & 15, lq = lane >> 4;
    const int Cin = g.job[0].Cin, Cout = g.job[0].Cout;    // (shared by the jobs of a group)
    const int nchunk = Cin >> 6;
    const long long wtile = 9ll * Cin * BN * 2;

    // per-lane fragment bases; everything added later is a compile-time or wave-uniform offset
    const unsigned char* abase = As + ((4 * wm) * kHW + l15) * kRS + lq * 16;
    const unsigned char* bbase = Bs + (wn * 64 + l15) * 64 + ((lq ^ swz(l15)) << 4);

    // ---- weights: LDS-DMA, one stage = 3 pieces of 8 KB (one per ky) = one 16-byte vector per thread and piece
    auto b_issue = [&](const unsigned char* wsrc, int stage, int buf) {
        const unsigned char* src = wsrc + (size_t)stage * kStageBytes + (size_t)tid * 16;
        unsigned char* dst = Bs + buf * kStageBytes + wave * 1024;            // (+ lane * 16 by the hardware)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + j * 8192),
                                             (__attribute__((address_space(3))) void*)(dst + j * 8192), 16, 0, 0);
    };

    // ---- halo staging: thread -> 16-byte slot tid % 8 of halo pixels q + 54 i, q = tid / 8 < 54, i = 0..5: three halo rows per
    // step, so a thread's column hx = q % 18 is FIXED and its rows are hy = q / 18 + 3 i - the column terms of the address and of the
    // range test are formed once per image, each load adds one row term (the first version walked pixels q + 64 i: a division by
    // 18, two clamps and two multiplies per load, 111 vector instructions per image; now ~45). Threads with q >= 54 (two lanes of
    // wave 6, all of wave 7) load clamped addresses and store nothing. Loads are unpredicated from clamped coordinates (a
    // predicated load is waited for on the spot); pixels outside the image are zeroed at commit time.
    const int slot = tid & 7, q54 = tid >> 3;
    const int qy = (int)(__umul24((unsigned)q54, 3641u) >> 16), qx = q54 - ((qy << 4) + (qy << 1));
    static_assert(kHW == 18 && kNPix == 6 * 54, "six steps of three halo rows");
    const unsigned act_lo2 = (AFFINE && g.job[0].in_act != MPN_ACT_NONE) ? 0u : 0x80008000u;     // (in_act is shared by the jobs)
    const bool act_relu6 = AFFINE && g.job[0].in_act == MPN_ACT_RELU6;
    int tab_job = -1;
    // the producer's scale / shift of a job, kept in LDS (16 registers per thread otherwise, live across two stages). Rewritten
    // only where no commit that reads the old table can follow before the next barrier.
    auto tab_load = [&](int job) {
        if constexpr (AFFINE) {
            if (job != tab_job) {
                for (int i = tid; i < Cin; i += kThreads) { tab[i] = g.job[job].in_scale[i]; tab[kMaxCin + i] = g.job[job].in_shift[i]; }
                tab_job = job;
            }
        }
        if constexpr (BNR) {   // (no producer affine in a data gradient: the table holds the fed batch-norm's scale / shift)
            if (job != tab_job) {
                for (int i = tid; i < Cout; i += kThreads) { tab[i] = g.job[job].bnr_scale[i]; tab[kMaxCin + i] = g.job[job].bnr_shift[i]; }
                tab_job = job;
            }
        }
    };
    unsigned okmask = 0;        // of the tile whose halo was fetched last: pixel i of this thread lies inside the image
    Vec16<T> av[kAVec];
    // pieces I0 .. I0 + NP - 1 of the halo image of (tile, chunk) into av[]; a_load = all six
    // (i0, np: constants once the stage loop is unrolled - the guards below fold away)
    auto a_load_part = [&](const int i0, const int np, const Tile& t, int chunk) __attribute__((always_inline)) {
        const Job& p = g.job[t.job];
        // Offsets in full-rate arithmetic: the image base is scalar, row and column BYTE offsets are 24-bit multiplies added as an
        // unsigned 32-bit offset to it (the load's scalar-base + vector-offset form: no 64-bit address arithmetic; launch() checks
        // 2 W xs < 2^24 and the tensor below 2^31 elements)
        const int wxb = p.W * p.xs * 2;
#if (MPN_KO & 8)
        const unsigned char* xb = reinterpret_cast<const unsigned char*>(p.x) + (long long)chunk * 128;
        const int ox0 = 0, oy0 = 0;
#else
        const unsigned char* xb = reinterpret_cast<const unsigned char*>(p.x) + ((long long)t.img * p.H * (p.W * p.xs) + chunk * 64) * 2;
        const int ox0 = t.ox0, oy0 = t.oy0;
#endif
        int hx = qx, hy = qy;
        asm volatile("" : "+v"(hx), "+v"(hy));   // opaque: keeps the offset arithmetic here instead of hoisted (and spilled) across the tile loop
        const int ix = ox0 + hx - 1;
        const bool okx = (unsigned)ix < (unsigned)p.W;
        const unsigned col = __umul24((unsigned)min(max(ix, 0), p.W - 1), (unsigned)(p.xs * 2)) + slot * 16;
        if (i0 == 0) okmask = 0;
#pragma unroll
        for (int i = 0; i < kAVec; ++i) {
            if (i < i0 || i >= i0 + np) continue;
            const int iy = oy0 + (3 * i - 1) + hy;
            // (bitwise, unsigned compares: the short-circuit form compiles to exec-mask branches around every test)
            const bool ok = okx & ((unsigned)iy < (unsigned)p.H);
            okmask |= (ok ? 1u : 0u) << i;
            av[i].load(reinterpret_cast<const T*>(xb + (__umul24((unsigned)min(max(iy, 0), p.H - 1), (unsigned)wxb) + col)));
        }
    };
    auto a_commit_t = [&](auto relu6, const int i0, const int np, int buf, int chunk) __attribute__((always_inline)) {
        unsigned char* dst = As + buf * kABytes + q54 * kRS + slot * 16;
        f32x2_t sc[4], sh[4];
        if constexpr (AFFINE) {
            const float* ts = tab + chunk * 64 + slot * 8;
            const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(ts), s1 = *reinterpret_cast<const f32x4_t*>(ts + 4);
            const f32x4_t h0 = *reinterpret_cast<const f32x4_t*>(ts + kMaxCin), h1 = *reinterpret_cast<const f32x4_t*>(ts + kMaxCin + 4);
            sc[0] = (f32x2_t){s0[0], s0[1]}; sc[1] = (f32x2_t){s0[2], s0[3]}; sc[2] = (f32x2_t){s1[0], s1[1]}; sc[3] = (f32x2_t){s1[2], s1[3]};
            sh[0] = (f32x2_t){h0[0], h0[1]}; sh[1] = (f32x2_t){h0[2], h0[3]}; sh[2] = (f32x2_t){h1[0], h1[1]}; sh[3] = (f32x2_t){h1[2], h1[3]};
        }
        if (q54 < 54) {
#pragma unroll
            for (int i = 0; i < kAVec; ++i) {
                if (i < i0 || i >= i0 + np) continue;
                if constexpr (AFFINE) affine_act<T, decltype(relu6)::value>(av[i], sc, sh, act_lo2);
                if (!((okmask >> i) & 1u)) av[i].zero();
                if (i == 0) lds_store16_raw<0>(dst, av[i].raw);
                if (i == 1) lds_store16_raw<1 * 54 * kRS>(dst, av[i].raw);
                if (i == 2) lds_store16_raw<2 * 54 * kRS>(dst, av[i].raw);
                if (i == 3) lds_store16_raw<3 * 54 * kRS>(dst, av[i].raw);
                if (i == 4) lds_store16_raw<4 * 54 * kRS>(dst, av[i].raw);
                if (i == 5) lds_store16_raw<5 * 54 * kRS>(dst, av[i].raw);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the compiler does not track the raw stores: complete before the barrier)
    };
    auto a_commit_part = [&](const int i0, const int np, int buf, int chunk) __attribute__((always_inline)) {
        // (a wave-uniform branch around the WHOLE commit: inside it hipcc would if-convert the clamp back into a select per element)
        if (act_relu6) a_commit_t(std::true_type{}, i0, np, buf, chunk); else a_commit_t(std::false_type{}, i0, np, buf, chunk);
    };
    // Statistics: a block SUMS the rows of its tiles of one job (threads 0..255: one of the two sums, one channel; f32, in the order of
    // its walk) and writes them once, behind its last tile of that job: one slab row per block instead of two per tile (the four
    // pyramid levels at batch 32: 256 rows for the finalize instead of 5 440). The sums of the tile that finished last are taken from
    // `red` after the next block barrier (complete by then).
    float* st_dst = nullptr;
    float st_acc = 0.f;
    bool st_pending = false, st_store = false;
    const int st_which = (tid >> 7) & 1, st_c = tid & 127;
    auto stats_flush = [&]() {
        if (st_pending) {
            // red [8 waves][2][64], fixed order: the four waves of the upper half, then of the lower half
            const float* r = red + st_which * 64 + st_c;
            st_acc += ((r[0] + r[128]) + (r[256] + r[384])) + ((r[512] + r[640]) + (r[768] + r[896]));
            if (st_store) { *st_dst = st_acc; st_acc = 0.f; }
        }
        st_pending = false;
    };

    // BNR: the raw tensor of the fed batch-norm at this wave's 64 pixels x 64 channels, in the copy-out layout of the epilogue
    // (lane = 16-byte piece lane % 8 of image rows lane / 8 + 8 k; a wave finishes rows 2 wn, 2 wn + 1 of its row group, all 64 channels):
    // four 16-byte loads per lane, requested under the tile's LAST weight stage and consumed by the epilogue
    uint4 bx[BNR ? 4 : 1];
    auto bnr_load = [&](const Tile& t, int hp) {
        if constexpr (BNR) {
            const Job& p = g.job[t.job];
            const int wbs = p.W * p.bnr_xs;          // (scalar image base + 24-bit offsets: see a_load)
            const T* xb = reinterpret_cast<const T*>(p.bnr_x) + (long long)t.img * p.H * wbs + t.ntile * BN + (lane & 7) * 8;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = (lane >> 3) + 8 * k;
                const int oy = min(t.oy0 + 4 * wm + hp * 2 + (row >> 4), p.H - 1), ox = min(t.ox0 + (row & 15), p.W - 1);
                bx[k] = *reinterpret_cast<const uint4*>(xb + (int)(__umul24((unsigned)oy, (unsigned)wbs) + __umul24((unsigned)ox, (unsigned)p.bnr_xs)));
            }
        }
    };

    Tile cur = tile_of<true>(g, w);
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(g.job[cur.job].wp) + cur.ntile * wtile;
    int cc = 0, ss = 0;        // running chunk / stage counters: halo buffer cc & 1, weight buffer ss & 1
    b_issue(wsrc, 0, 0);
    a_load_part(0, kAVec, cur, 0);
    tab_load(cur.job);
    __syncthreads();           // table visible
    a_commit_part(0, kAVec, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#ifdef MPN_DIAG
    int titer = 0;
#endif
    for (;;) {
        C3_STAMP(0);
        const int wnext = w + (int)gridDim.x;
        const bool has_next = wnext < total;
        const Tile nxt = tile_of<true>(g, has_next ? wnext : w);
        const unsigned char* wsrc_next = reinterpret_cast<const unsigned char*>(g.job[nxt.job].wp) + nxt.ntile * wtile;

        f32x4_t acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

        for (int chunk = 0; chunk < nchunk; ++chunk, ++cc) {
            const unsigned char* ab = abase + (cc & 1) * kABytes;
            const bool last_chunk = chunk + 1 == nchunk;
            const bool stage_more = !last_chunk || has_next;       // a halo image to prepare under this chunk
#pragma unroll
            for (int sl = 0; sl < NS; ++sl, ++ss) {
                // stage sl = kx: taps (ky, kx) for ky = 0..2, both k-steps of the chunk; this wave multiplies k-step wn (input channels
                // chunk * 64 + wn * 32 .. + 31)
                const int a_off = sl * kRS + wn * 64;
                const unsigned char* bb = bbase + (ss & 1) * kStageBytes;
                // The stage's requests (next weight stage, in a chunk's first stage the next halo image): BEHIND the stage's first fragment
                // reads - in front of them every wave of the block spent its first ~200 cycles behind the barrier issuing LDS-DMA pieces
                // while the matrix pipe had nothing to do (same box: 135.3 -> 132.9 us with affine + statistics). The fused-reduction
                // variants keep them in front: with ten fragments live across the requests they spill.
                auto requests = [&]() __attribute__((always_inline)) {
                    // the next weight stage (of this tile, or the first one of the next tile) into the buffer that stage ss - 1 read
                    if (sl < NS - 1 || !last_chunk) b_issue(wsrc, chunk * NS + sl + 1, (ss + 1) & 1);
                    else if (has_next) b_issue(wsrc_next, 0, (ss + 1) & 1);
                    if constexpr (BNR) { if (last_chunk && sl == NS - 1) bnr_load(cur, wn); }
#if !(MPN_KO & 2)
                    if (sl == 0) {
                        // UNCONDITIONAL (without a following image the same chunk is fetched again and dropped): behind a condition
                        // the six registers become PHIs, hipcc copies them right behind the loads - and waits for the loads there.
                        // (Spreading the six loads over the chunk's first three stages, two per stage behind that stage's weight pieces,
                        //  with the commits two stages later, changes nothing: the counted waits are short - profiles/r04_c3_stage_stamps.txt)
                        if (last_chunk) a_load_part(0, kAVec, nxt, 0); else a_load_part(0, kAVec, cur, chunk + 1);
                        // (last chunk: this tile's commits are all behind a barrier - the table may change for the next tile's job.
                        //  AFFINE only: the fused-reduction table is read by THIS tile's epilogue - it changes behind it, below. Round 4
                        //  switched it here too, and a block that walked from one job of a group into the next masked its last tile of
                        //  the first job with the second job's batch-norm: found in round 5: tests/test_ops_bwd_gpu.py::test_grouped_fused_reduction_when_a_block_walks_from_one_job_into_the_next)
                        if constexpr (AFFINE) { if (last_chunk && has_next) tab_load(nxt.job); }
                    }
#endif
                };
                if constexpr (BNR) requests();
                X8 a[6], b0[4], b1[4];
#pragma unroll
                for (int r = 0; r < 6; ++r) a[r] = *reinterpret_cast<const X8*>(ab + a_off + r * (kHW * kRS));
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) b0[nt] = *reinterpret_cast<const X8*>(bb + nt * 1024);
                if constexpr (!BNR) requests();
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) b1[nt] = *reinterpret_cast<const X8*>(bb + 8192 + nt * 1024);
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = H::mfma(b0[nt], a[mt], acc[mt][nt]);       // D^T = W^T x A^T
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < 4; ++nt) b0[nt] = *reinterpret_cast<const X8*>(bb + 16384 + nt * 1024);   // (ky = 2 into the ky = 0 registers)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = H::mfma(b1[nt], a[mt + 1], acc[mt][nt]);
                __builtin_amdgcn_sched_barrier(0);
                // the halo image prepared under this chunk: its loads have had two stages to land, and the buffer was last read
                // in the previous chunk (the wave-private epilogue images in it: before the barrier of this chunk's first stage)
                // (sl == 2 is the chunk's last stage - the commit still completes in front of its barrier)
#if !(MPN_KO & 2)
                if (sl == 2 && stage_more) a_commit_part(0, kAVec, (cc + 1) & 1, last_chunk ? 0 : chunk + 1);
#endif
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = H::mfma(b0[nt], a[mt + 2], acc[mt][nt]);
                // this wave's pieces of the next weight stage have landed. In a chunk's first stage the six halo loads issued BEHIND
                // them stay in flight (the counter retires in order; they are committed two stages later) - behind a raw barrier:
                // __syncthreads() would wait for them too
                C3_WSTAMP(chunk * NS + sl, 0);
                if (sl == 0 && !(MPN_KO & 2)) {
                    asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
                    C3_WSTAMP(chunk * NS + sl, 1);
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    C3_WSTAMP(chunk * NS + sl, 1);
                    __syncthreads();
                }
                C3_WSTAMP(chunk * NS + sl, 2);
                if (sl == 0 && chunk == 0) stats_flush();          // (every wave's `red` rows of the previous tile are visible)
            }
            C3_STAMP(1 + (chunk < 5 ? chunk : 5));
        }

#if (MPN_KO & 1)
        {
            f32x4_t s4 = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) s4 += acc[i][j];
            if (s4[0] + s4[1] + s4[2] + s4[3] == 12345.678f) reinterpret_cast<float*>(g.job[cur.job].y)[tid] = s4[0];
        }
#else
        // ================= epilogue of tile `cur`, wave-local. Halo buffer (cc - 1) & 1 is free: every wave is past the last
        // stage's barrier, and the next halo image to be committed into it comes two barriers from now.
        {
            C3_WSTAMP(12, 0);
            const Job& p = g.job[cur.job];
            T* __restrict__ y = reinterpret_cast<T*>(p.y);
            const int n0 = cur.ntile * BN;
            constexpr int RSW = 64 * 2 + 8;                                   // wave-private image: 32 pixels x (128 + 8) bytes
            unsigned char* Ow = As + ((cc - 1) & 1) * kABytes + wave * (32 * RSW);
            {
                // the pair (wm, 0), (wm, 1) holds two partial sums of the same 64 pixels x 64 channels. Each wave hands over the
                // two image rows it will NOT finish (wave wn keeps rows 2 wn, 2 wn + 1) as f32: 8 KB per wave, slots 0..5 in the
                // released halo buffer, 6 and 7 in the released weight buffer ((ss + 1) & 1: the last stage read ss - 1... the
                // one in flight is ss & 1). One block barrier; afterwards a wave's private image lives in the slot it has read.
                auto slot_of = [&](int wv) -> unsigned char* {
                    return wv < 6 ? As + ((cc - 1) & 1) * kABytes + wv * 8192 : Bs + ((ss + 1) & 1) * kStageBytes + (wv - 6) * 8192;
                };
                unsigned char* mine = slot_of(wave);
                // (wave-uniform branches with constant register indices: a select between two accumulator rows becomes a
                //  dynamically indexed array, i.e. scratch)
                if (wn == 0) {
#pragma unroll
                    for (int ml = 0; ml < 2; ++ml)
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) *reinterpret_cast<f32x4_t*>(mine + ((ml * 4 + nt) * 64 + lane) * 16) = acc[2 + ml][nt];
                } else {
#pragma unroll
                    for (int ml = 0; ml < 2; ++ml)
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) *reinterpret_cast<f32x4_t*>(mine + ((ml * 4 + nt) * 64 + lane) * 16) = acc[ml][nt];
                }
                // (raw barriers: only LDS traffic is ordered here - __syncthreads() would also drain the vector-memory counter)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                const unsigned char* theirs = slot_of(wave ^ 1);
                f32x4_t o[2][4];
#pragma unroll
                for (int ml = 0; ml < 2; ++ml)
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) o[ml][nt] = *reinterpret_cast<const f32x4_t*>(theirs + ((ml * 4 + nt) * 64 + lane) * 16);
                // (fixed order: k-step 0's partial sum + k-step 1's)
                if (wn == 0) {
#pragma unroll
                    for (int ml = 0; ml < 2; ++ml)
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) acc[ml][nt] = acc[ml][nt] + o[ml][nt];
                } else {
#pragma unroll
                    for (int ml = 0; ml < 2; ++ml)
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) acc[ml][nt] = o[ml][nt] + acc[2 + ml][nt];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                Ow = const_cast<unsigned char*>(theirs);
            }
            const bool stats = p.stats_part != nullptr;
            // BNR: per-lane scale / shift of its 8 channels, and the running sums of g and g * x over the lane's 8 rows
            f32x2_t bsc[4], bsh[4], bs[4], bq[4];
            float blo = -INFINITY, bhi = INFINITY;
            if constexpr (BNR) {
                const float* ts = tab + cur.ntile * BN + (lane & 7) * 8;
                const f32x4_t s0 = *reinterpret_cast<const f32x4_t*>(ts), s1 = *reinterpret_cast<const f32x4_t*>(ts + 4);
                const f32x4_t h0 = *reinterpret_cast<const f32x4_t*>(ts + kMaxCin), h1 = *reinterpret_cast<const f32x4_t*>(ts + kMaxCin + 4);
                bsc[0] = (f32x2_t){s0[0], s0[1]}; bsc[1] = (f32x2_t){s0[2], s0[3]}; bsc[2] = (f32x2_t){s1[0], s1[1]}; bsc[3] = (f32x2_t){s1[2], s1[3]};
                bsh[0] = (f32x2_t){h0[0], h0[1]}; bsh[1] = (f32x2_t){h0[2], h0[3]}; bsh[2] = (f32x2_t){h1[0], h1[1]}; bsh[3] = (f32x2_t){h1[2], h1[3]};
#pragma unroll
                for (int j = 0; j < 4; ++j) { bs[j] = (f32x2_t){0.f, 0.f}; bq[j] = (f32x2_t){0.f, 0.f}; }
                blo = p.bnr_act != MPN_ACT_NONE ? 0.f : -INFINITY;
                bhi = p.bnr_act == MPN_ACT_RELU6 ? 6.f : INFINITY;
            }
            f32x4_t sa[4], ga[4];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) { sa[nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; ga[nt] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; }
            X8 ones;
#pragma unroll
            for (int j = 0; j < 8; ++j) ones[j] = 1.0f;
            // copy-out lanes: lane j moves 16-byte piece j % 8 of image rows j / 8 + 8 k (8 lanes = the wave's 128 bytes of a pixel)
            const int cpiece = lane & 7, crow = lane >> 3;
            const int wys = p.W * p.ys;
            T* __restrict__ yimg = y + (long long)cur.img * p.H * wys + n0;   // (wave-uniform)
            // Full tiles (block-uniform) skip the zeroing of pixels outside the image (64 selects per wave) and the store predicates:
            // a branch around the whole pass - inside it hipcc if-converts the test back into the selects.
            const bool full_tile = cur.oy0 + 16 <= p.H && cur.ox0 + 16 <= p.W;
            // a lane's store offset inside its wave's 32 pixels is a constant (row = crow + 8 k -> image row k / 2, column crow + 8 (k & 1));
            // the rest of the address is scalar: the stores take a scalar base + this 32-bit byte offset, no vector address arithmetic
            const unsigned lane_off = (unsigned)(crow * p.ys + cpiece * 8) * 2u;
            auto ep_pass = [&](auto edge) __attribute__((always_inline)) {
                constexpr bool EDGE = decltype(edge)::value;
                const int hp = wn;                                            // (the rows this wave finishes; their sums sit in acc[0..1])
#pragma unroll
                for (int ml = 0; ml < 2; ++ml) {
                    const int mt = hp * 2 + ml;
                    const bool ok = !EDGE || ((cur.oy0 + 4 * wm + mt) < p.H && (cur.ox0 + l15) < p.W);
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        f32x4_t v = acc[ml][nt];
                        if (EDGE && !ok) v = (f32x4_t){0.f, 0.f, 0.f, 0.f};   // pixels outside the image must not count in the statistics
                        store4(reinterpret_cast<T*>(Ow + (ml * 16 + l15) * RSW + (nt * 16 + lq * 4) * 2), v);
                    }
                }
                if (stats && !BNR) {
                    // the 32 pixels x 64 channels just written, read back transposed (ds_read_b64_tr_b16): lane 4q+pp of a 16-lane
                    // group supplies the address of block row q, channels 4pp..4pp+3; the k order of a sum is free
                    const unsigned char* tb = Ow + ((lq >> 1) * 2 + 4 * ((lq & 1) * 4 + (l15 >> 2))) * RSW + (4 * (l15 & 3)) * 2;
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) {
                        const typename H::x4 lo = H::tr_read(tb + nt * 32);
                        const typename H::x4 hi = H::tr_read(tb + RSW + nt * 32);
                        const X8 f = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                        sa[nt] = H::mfma(ones, f, sa[nt]);
                        ga[nt] = H::mfma(f, f, ga[nt]);
                    }
                }
                // whole 128-byte pixel rows (the tile's 64 channels) to HBM:
                // all eight LDS reads first (unconditional), then the stores - one LDS round trip per 32 pixels
                uint2 ca[4], cb[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    ca[k] = *reinterpret_cast<const uint2*>(Ow + (crow + 8 * k) * RSW + cpiece * 16);
                    cb[k] = *reinterpret_cast<const uint2*>(Ow + (crow + 8 * k) * RSW + cpiece * 16 + 8);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = crow + 8 * k;                              // image row = (ml, l15)
                    const int oyb = cur.oy0 + 4 * wm + hp * 2 + (k >> 1), oxb = cur.ox0 + 8 * (k & 1);     // (scalar)
                    uint4 o = make_uint4(ca[k].x, ca[k].y, cb[k].x, cb[k].y);
                    if constexpr (BNR) {
                        // g = dy where the fed batch-norm's activation passes (lo < x * scale + shift < hi, the test of
                        // bn_bwd_reduce / bn_bwd_apply, fused multiply-add), else 0; sums of g and g * x (pixels outside the image
                        // hold dy = 0). 16-bit storage: element pairs per dword.
                        const uint4 xv = bx[k];
                        const unsigned xu[4] = {xv.x, xv.y, xv.z, xv.w};
                        unsigned du[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const f32x2_t xf = {to_f32(__builtin_bit_cast(T, (unsigned short)(xu[j] & 0xffffu))),
                                                to_f32(__builtin_bit_cast(T, (unsigned short)(xu[j] >> 16)))};
                            const f32x2_t pre = xf * bsc[j] + bsh[j];
                            const unsigned m = ((pre[0] > blo && pre[0] < bhi) ? 0x0000ffffu : 0u) | ((pre[1] > blo && pre[1] < bhi) ? 0xffff0000u : 0u);
                            du[j] &= m;
                            const f32x2_t gf = {to_f32(__builtin_bit_cast(T, (unsigned short)(du[j] & 0xffffu))),
                                                to_f32(__builtin_bit_cast(T, (unsigned short)(du[j] >> 16)))};
                            bs[j] += gf;
                            bq[j] += gf * xf;
                        }
                        o = make_uint4(du[0], du[1], du[2], du[3]);
                    }
                    // (launch() checks the output tensor below 2^31 elements)
                    unsigned char* ybase = reinterpret_cast<unsigned char*>(yimg + ((long long)oyb * wys + (long long)oxb * p.ys));
                    const bool inside = !EDGE || (oyb < p.H && cur.ox0 + (row & 15) < p.W);
                    if (inside && (!(MPN_KO & 16) || o.x == 0x12345678u))
                        *reinterpret_cast<uint4*>(ybase + lane_off) = o;
                }
            };
            if (full_tile) ep_pass(std::false_type{}); else ep_pass(std::true_type{});
            if constexpr (BNR) {
                // the lanes of one 16-byte piece (lane bits 3..5 = the 8 row lanes): fixed butterfly, then lanes 0..7 hold the wave's
                // sums of their 8 channels
#pragma unroll
                for (int o = 8; o < 64; o <<= 1)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bs[j][0] += __shfl_xor(bs[j][0], o, 64); bs[j][1] += __shfl_xor(bs[j][1], o, 64);
                        bq[j][0] += __shfl_xor(bq[j][0], o, 64); bq[j][1] += __shfl_xor(bq[j][1], o, 64);
                    }
                if (lane < 8) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int cl = lane * 8 + 2 * j;      // red [8 waves][2][64]
                        red[(wave * 2 + 0) * 64 + cl] = bs[j][0]; red[(wave * 2 + 0) * 64 + cl + 1] = bs[j][1];
                        red[(wave * 2 + 1) * 64 + cl] = bq[j][0]; red[(wave * 2 + 1) * 64 + cl + 1] = bq[j][1];
                    }
                }
            }
            if (stats) {
                const int r = l15 & 3;
#pragma unroll
                for (int nt = 0; nt < 4 && !BNR; ++nt) {
                    const float q = r == 0 ? ga[nt][0] : (r == 1 ? ga[nt][1] : (r == 2 ? ga[nt][2] : ga[nt][3]));
                    if (lq == (l15 >> 2)) {
                        const int cl = nt * 16 + l15;
                        red[(wave * 2 + 0) * 64 + cl] = sa[nt][0];
                        red[(wave * 2 + 1) * 64 + cl] = q;
                    }
                }
                // the block's row of this job's slab (pixels outside the image count as zeros): row = the position of the block's
                // FIRST tile of the job among the job's first gridDim.x tiles, i.e. rows 0 .. min(grid, tiles) / n_tiles - 1
                // (mpn_conv_stats_rows; the grid is a multiple of n_tiles, so a block keeps its channel tile within a job)
                {
                    const int n_tiles = Cout >> 6, grid = (int)gridDim.x;
                    int off = (w_first - g.begin[cur.job]) % grid;
                    if (off < 0) off += grid;
                    st_pending = tid < 256 && st_c < 64;
                    st_store = !has_next || nxt.job != cur.job;
                    st_dst = p.stats_part + ((long long)(off / n_tiles) * 2 + st_which) * Cout + n0 + st_c;
                }
            }
            // the next tile's first weight stage is requested into the weight buffer that held slots 6 and 7
            // (a raw barrier: the tile's output stores stay in flight)
            if (has_next) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
#endif
        C3_WSTAMP(12, 1);
        C3_STAMP(7);
#ifdef MPN_DIAG
        ++titer;
#endif
        if (!has_next) break;
        if constexpr (BNR) {
            // (block-uniform) the next job's batch-norm, once every wave has finished this tile's epilogue (the barrier that ends it, above)
            if (nxt.job != tab_job) tab_load(nxt.job);
        }
        cur = nxt;
        wsrc = wsrc_next;
        w = wnext;
    }
    __syncthreads();
    stats_flush();
#ifdef MPN_DIAG
    if (g.job[0].dbg && threadIdx.x == 0) g.job[0].dbg[(size_t)blockIdx.x * 8 + 5] = __builtin_amdgcn_s_memrealtime();
    if (g.job[0].dbg) {
        __syncthreads();
        for (int i = threadIdx.x; i < 8 * 13 * 3; i += kThreads) g.job[0].dbg[(size_t)gridDim.x * 8 + (size_t)blockIdx.x * (8 * 13 * 3) + i] = wst[i];
    }
#endif
}

template <typename T, bool AFFINE, bool BNR = false>
int launch_t(const Group& g, int blocks, hipStream_t st) {
    static mpn_attr_mask_t attr_mask{0};
    MPN_HIP(mpn_ensure_dynamic_lds((const void*)conv3x3_kernel<T, AFFINE, BNR>, kLds, &attr_mask));
    conv3x3_kernel<T, AFFINE, BNR><<<dim3((unsigned)blocks), dim3(kThreads), kLds, st>>>(g);
    MPN_LAUNCH_CHECK();
    return MPN_OK;
}
template <typename T>
int launch_v(const Group& g, int blocks, bool affine, bool bnr, hipStream_t st) {
    if (bnr) return launch_t<T, false, true>(g, blocks, st);
    return affine ? launch_t<T, true>(g, blocks, st) : launch_t<T, false>(g, blocks, st);
}

}  // namespace

#ifdef MPN_DIAG
static int g_diag_block_cap = 0;
extern "C" void mpn_diag_set_c3_blocks(int n) { g_diag_block_cap = n; }     // (tools/stamp_c3cs.py; never in the shipped library)
#endif

namespace mpn_c3 {

int launch(const Job* jobs, int njobs, int dtype, hipStream_t st) {
    MPN_REQUIRE(njobs >= 1 && njobs <= kMaxJobs, MPN_ERR_BAD_ARG, "conv3x3: %d jobs", njobs);
    Group g = {};
    int begin = 0;
    const bool affine = jobs[0].in_scale != nullptr;
    MPN_REQUIRE(jobs[0].Cin <= kMaxCin, MPN_ERR_BAD_SHAPE, "conv3x3: at most %d input channels", kMaxCin);
    for (int j = 0; j < njobs; ++j) {
        MPN_REQUIRE((jobs[j].in_scale != nullptr) == affine, MPN_ERR_BAD_ARG, "conv3x3: the jobs of a group share the affine / no affine variant");
        MPN_REQUIRE((long long)jobs[j].N * jobs[j].H * jobs[j].W * jobs[j].xs < (1ll << 31), MPN_ERR_BAD_SHAPE,
                    "conv3x3: the input tensor must span fewer than 2^31 elements");
        MPN_REQUIRE((long long)jobs[j].W * jobs[j].xs < (1ll << 23) && (long long)jobs[j].W * jobs[j].ys < (1ll << 24) && jobs[j].H < (1 << 24),
                    MPN_ERR_BAD_SHAPE, "conv3x3: a pixel row must span fewer than 2^23 input / 2^24 output elements (24-bit offset arithmetic)");
        MPN_REQUIRE((long long)jobs[j].N * jobs[j].H * jobs[j].W * jobs[j].ys < (1ll << 31), MPN_ERR_BAD_SHAPE,
                    "conv3x3: the output tensor must span fewer than 2^31 elements");
        g.job[j] = jobs[j];
        g.begin[j] = begin;
        begin += blocks_of(jobs[j]);
    }
    for (int j = njobs; j <= kMaxJobs; ++j) g.begin[j] = begin;
    g.njobs = njobs;
    // one persistent block per CU (LDS allows one); fewer when the group has fewer tiles
    int dev = 0, cus = 0;
    MPN_HIP(hipGetDevice(&dev));
    MPN_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const bool n64 = (jobs[0].Cout & 127) != 0;     // (the jobs of a group share Cin and Cout)
    {   // `begin` is the grid size from here on: a multiple of the channel tiles per pixel tile (see the statistics rows)
        const int n_tiles = n64 ? jobs[0].Cout / 64 : jobs[0].Cout / 128;
        if (begin > cus) begin = cus - cus % n_tiles;
        MPN_REQUIRE(begin > 0, MPN_ERR_BAD_SHAPE, "conv3x3: %d channel tiles on %d compute units", n_tiles, cus);
    }
#ifdef MPN_DIAG
    // diagnostic builds only: fewer persistent blocks than compute units (what the clock does when part of the chip multiplies)
    if (g_diag_block_cap > 0 && g_diag_block_cap < begin) begin = g_diag_block_cap;
#endif
    const bool bnr = jobs[0].bnr_x != nullptr;
    for (int j = 0; j < njobs; ++j) {
        MPN_REQUIRE((jobs[j].bnr_x != nullptr) == bnr, MPN_ERR_BAD_ARG, "conv3x3: the jobs of a group share the fused-reduction variant");
        MPN_REQUIRE(!bnr || ((long long)jobs[j].W * jobs[j].bnr_xs < (1ll << 24) && (long long)jobs[j].N * jobs[j].H * jobs[j].W * jobs[j].bnr_xs < (1ll << 31)),
                    MPN_ERR_BAD_SHAPE, "conv3x3: the fed batch-norm's raw tensor must span fewer than 2^31 elements, a pixel row fewer than 2^24");
        MPN_REQUIRE(!bnr || (!affine && jobs[j].stats_part && jobs[j].bnr_scale && jobs[j].bnr_shift && jobs[j].bnr_xs >= jobs[j].Cout &&
                             jobs[j].bnr_xs % 8 == 0 && jobs[j].Cout <= kMaxCin && mpn_aligned16(jobs[j].bnr_x)),
                    MPN_ERR_BAD_ARG, "conv3x3: the fused batch-norm reduction needs a data gradient (no producer affine), a partial slab and the layer's affine");
    }
    // 128-channel tiles and one-chunk 64-channel tiles (the detector's 64 -> 64 towers) run on the channel-split kernel (conv3x3_cs.hip).
    // (64-channel tiles of MORE than one 64-channel chunk - final_conv3x3's forward, 512 -> 64 - stay on this file's kernel: measured
    //  again in round 6 with the weights two stages ahead, 255.8 / 285.6 us (plain / affine + statistics) here against 264.8 / 287.7 there:
    //  a 64-channel tile pulls its weight fragments from L2 at twice the 128-channel tile's rate per MFMA, this kernel shares them through
    //  LDS; one-chunk tiles are 6-11 % faster on the channel-split one: profiles/r05_c3cs_n64.txt, r06_c3_n64_ab.txt)
    if (!n64 || jobs[0].Cin == 64) return launch_cs(g, begin, dtype, affine, bnr, n64, st);
    if (dtype == MPN_BF16) return launch_v<bf16_t>(g, begin, affine, bnr, st);
    if (dtype == MPN_F16) return launch_v<half_t>(g, begin, affine, bnr, st);
    MPN_FAIL(MPN_ERR_BAD_DTYPE, "conv3x3: 16-bit storage only");
}

/* rows of a job's statistics slab: one per block that has a tile of the job (see the kernel) */
int stats_rows(int N, int H, int W, int Cout) {
    const bool n64 = (Cout & 127) != 0;
    const int n_tiles = n64 ? Cout / 64 : Cout / 128;
    const long long tiles = (long long)N * ((H + 15) / 16) * ((W + 15) / 16) * n_tiles;
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return -1;
    const long long grid = cus - cus % n_tiles;
    if (grid <= 0) return -1;
    return (int)((tiles < grid ? tiles : grid) / n_tiles);
}

}  // namespace mpn_c3
