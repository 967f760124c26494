// Batch-norm finalize fused into the kernel that produces the partial sums ("last block finishes").
// Every stats-producing block writes its partial row [2][C] (or its BN-channel slice of it), then calls bn_tail():
//   level 1: rows are grouped (<= 128 groups); the block that completes a group (agent-scope ticket) sums the group's
//            rows in f64 into row2[group];
//   level 2: the block that completes the last group sums row2 in f64, in group order, and does what mpn_bn_finalize /
//            mpn_bn_bwd_finalize do (scale/shift + moving statistics, or dgamma/dbeta/k1/k2).
// Fixed summation order at both levels -> deterministic. Tickets reset themselves; the workspace is zero-filled once.
// Saves 80 launches per training step that sat at the ~5 us launch floor (0.58 ms of 12.0).
#pragma once
#include "common.h"

struct BnTailDev {            // device-side view of mpn_bn_tail_t (+ geometry filled by the launcher)
    int mode;                 // 0 = off, 1 = forward statistics, 2 = backward sums
    int nparts, group, ngroups;
    double count;
    float momentum, eps;
    const float* gamma; const float* beta;
    float* moving_mean; float* moving_var; float* scale; float* shift; float* save_mean; float* save_invstd;
    float* dgamma; float* dbeta; float* k1; float* k2;
    unsigned* counters;       // [slices][1 + ngroups]
    double* row2;             // [slices][ngroups][2][cw]
};

constexpr int kBnTailMaxGroups = 128;

int bn_tail_check(const mpn_bn_tail_t* t, int C, const char* who);   // defined in bn.hip (argument validation)

inline size_t bn_tail_workspace_bytes_for(int C) {
    // worst case: 8 channel slices (Cout 1024 / BN 128) x (1 + 128 tickets), row2 [128 groups][2][C] doubles
    return (size_t)16 * 1024 + (size_t)kBnTailMaxGroups * 2 * (size_t)C * sizeof(double);
}

// host: fill the geometry for `nparts` partial rows; returns false if the tail is off
inline bool bn_tail_prepare(const mpn_bn_tail_t* t, int nparts, int C, BnTailDev* d) {
    d->mode = 0;
    if (t == nullptr || t->mode == 0) return false;
    d->mode = t->mode;
    d->nparts = nparts;
    d->group = (nparts + kBnTailMaxGroups - 1) / kBnTailMaxGroups;
    d->ngroups = (nparts + d->group - 1) / d->group;
    d->count = (double)t->count;
    d->momentum = t->momentum; d->eps = t->eps;
    d->gamma = t->gamma; d->beta = t->beta;
    d->moving_mean = t->moving_mean; d->moving_var = t->moving_var;
    d->scale = t->scale; d->shift = t->shift; d->save_mean = t->save_mean; d->save_invstd = t->save_invstd;
    d->dgamma = t->dgamma; d->dbeta = t->dbeta; d->k1 = t->k1; d->k2 = t->k2;
    d->counters = reinterpret_cast<unsigned*>(t->workspace);
    d->row2 = reinterpret_cast<double*>(reinterpret_cast<unsigned char*>(t->workspace) + 16 * 1024);
    return true;
}

// Hand-off without fences (MI355X_MICROARCH.md, "Hand-offs measured with sc1 loads in place of the acquire"): every
// handed-off byte is written by an `sc1` store and read by an `sc1` load (agent-scope relaxed atomics compile to exactly
// those), every storing wave waits `vmcnt(0)`, a workgroup barrier, then ONE lane's agent-scope atomic add; the block
// whose add came last reads after a barrier. An agent-scope release/acquire fence here (`__threadfence()`) writes back and
// invalidates the XCD's L2 per block and made the whole training step 3.3x slower.
__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_sc1(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_sc1(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), __builtin_bit_cast(unsigned long long, v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_sc1(const double* p) {
    return __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                        __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_sc1(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Called by ALL threads of a block (uniformly) after the block's partial row `row` has been written by this block.
//   part   [nparts][2][C] f32 written with st_sc1, this block covers channels [c_begin, c_begin + cw) of row `row`
//   slice  index of the channel slice (blocks with the same slice share tickets; 0 when a row is written whole)
// `sh_flag` is one int of shared memory.
__device__ __forceinline__ void bn_tail(const BnTailDev& f, const float* __restrict__ part, int C, int row, int c_begin,
                                        int cw, int slice, int tid, int nthreads, int* sh_flag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's sc1 stores of the partial row have landed
    __syncthreads();
    const int g = row / f.group;
    unsigned* cnt = f.counters + (size_t)slice * (1 + kBnTailMaxGroups);
    double* row2 = f.row2;   // slices use disjoint channel ranges of the same [ngroups][2][C] table
    if (tid == 0) {
        const int rows_in_group = min(f.group, f.nparts - g * f.group);
        const unsigned t = __hip_atomic_fetch_add(&cnt[1 + g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *sh_flag = (t == (unsigned)rows_in_group - 1u);
    }
    __syncthreads();
    if (!*sh_flag) return;
    {
        const int r0 = g * f.group, r1 = min(r0 + f.group, f.nparts);
        for (int col = tid; col < 2 * cw; col += nthreads) {
            const int which = col / cw, c = c_begin + (col - which * cw);
            if (c < C) {
                // 8 independent sc1 loads in flight (a rolled `s += load` loop pays one memory round trip per row)
                double s = 0.0;
                for (int r = r0; r < r1; r += 8) {
                    float v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = ld_sc1(part + ((long long)min(r + u, r1 - 1) * 2 + which) * C + c);
#pragma unroll
                    for (int u = 0; u < 8; ++u) s += (r + u < r1) ? (double)v[u] : 0.0;
                }
                st_sc1(row2 + ((size_t)g * 2 + which) * C + c, s);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        st_sc1(&cnt[1 + g], 0u);                          // ticket ready for the next launch
        const unsigned t2 = __hip_atomic_fetch_add(&cnt[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *sh_flag = (t2 == (unsigned)f.ngroups - 1u);
    }
    __syncthreads();
    if (!*sh_flag) return;
    for (int cc = tid; cc < cw; cc += nthreads) {
        const int c = c_begin + cc;
        if (c >= C) continue;
        double s = 0.0, q = 0.0;
        for (int gg = 0; gg < f.ngroups; gg += 4) {
            double a[4], b[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int gi = min(gg + u, f.ngroups - 1);
                a[u] = ld_sc1(row2 + ((size_t)gi * 2 + 0) * C + c);
                b[u] = ld_sc1(row2 + ((size_t)gi * 2 + 1) * C + c);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (gg + u < f.ngroups) { s += a[u]; q += b[u]; }
            }
        }
        if (f.mode == 1) {   // == bn_finalize_kernel
            const double mean = s / f.count;
            double var = q / f.count - mean * mean;
            if (var < 0.0) var = 0.0;
            const float invstd = (float)(1.0 / sqrt(var + (double)f.eps));
            const float sc = f.gamma[c] * invstd;
            f.scale[c] = sc;
            f.shift[c] = f.beta[c] - (float)mean * sc;
            if (f.save_mean) f.save_mean[c] = (float)mean;
            if (f.save_invstd) f.save_invstd[c] = invstd;
            if (f.moving_mean) {
                const double unbiased = f.count > 1.0 ? var * (f.count / (f.count - 1.0)) : var;
                f.moving_mean[c] = f.moving_mean[c] * f.momentum + (float)mean * (1.f - f.momentum);
                f.moving_var[c] = f.moving_var[c] * f.momentum + (float)unbiased * (1.f - f.momentum);
            }
        } else {             // == bn_bwd_finalize_kernel
            f.dbeta[c] = (float)s;
            f.dgamma[c] = (float)q;
            f.k1[c] = (float)(s / f.count);
            f.k2[c] = (float)(q / f.count);
        }
    }
    if (tid == 0) st_sc1(&cnt[0], 0u);
}
