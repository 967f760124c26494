// Round 6, VERDICT r5 item 4(b): what a captured {hipMemsetAsync(counter) -> appending kernel -> reading kernel} chain looks like as a
// hipGraph, and whether a replay ever lets the appends see a counter that the memset node has not cleared yet.
//
// It is the shape of mpn_retina_nms as round 3-5 shipped it (csrc/retina.hip before 8a071c8: a 4-byte hipMemsetAsync of the candidate
// counters, then retina_candidates_kernel's wave-aggregated atomicAdd appends, then retina_nms_kernel reading the count), with the
// appends made HARMLESS: the kernel never stores through the slot it draws, it only records the largest base any wave drew. A base
// >= total in a replay means the counter did not start at zero. Nothing here can write out of bounds.
//
//   hipcc --offload-arch=gfx950 -O2 tools/graph_memset_order.hip -o tools/build/graph_memset_order && tools/build/graph_memset_order
//
// Prints: the node list of the captured graph (type per node), its edges, and per variant the number of replays whose appends saw a
// non-zero counter. Variants: memset node vs reset kernel; synchronised after every replay vs back to back; the capture made in
// relaxed mode on a non-blocking stream (what torch.cuda.graph does).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

__global__ void reset_kernel(int* counter) { if (threadIdx.x == 0 && blockIdx.x == 0) *counter = 0; }

// every lane is "live": one wave-aggregated add of 64 per wave, as retina_candidates_kernel with every anchor a candidate
__global__ void append_kernel(int* counter, int* max_base) {
    const int lane = threadIdx.x & 63;
    int base = 0;
    if (lane == 0) base = atomicAdd(counter, 64);
    base = __shfl(base, 0, 64);
    if (lane == 0) atomicMax(max_base, base);
}

__global__ void read_kernel(const int* counter, int* seen, int* max_base, int* max_seen) {
    if (threadIdx.x == 0) { *seen = *counter; *max_seen = *max_base; *max_base = 0; }
}

static const char* type_name(hipGraphNodeType t) {
    switch (t) {
        case hipGraphNodeTypeKernel: return "kernel";
        case hipGraphNodeTypeMemset: return "memset";
        case hipGraphNodeTypeMemcpy: return "memcpy";
        case hipGraphNodeTypeEmpty: return "empty";
        default: return "other";
    }
}

int main() {
    const int blocks = 616 * 2, threads = 256;           // (157 542 anchors / 256 threads x 2 images: the grid of cfg4's candidates kernel at B = 2)
    const int total = blocks * threads;
    int *counter, *max_base, *seen, *max_seen;
    CK(hipMalloc(&counter, 16)); CK(hipMalloc(&max_base, 4)); CK(hipMalloc(&seen, 4)); CK(hipMalloc(&max_seen, 4));
    CK(hipMemset(counter, 0, 16)); CK(hipMemset(max_base, 0, 4));
    int rt = 0; CK(hipRuntimeGetVersion(&rt));
    printf("hip runtime version %d; grid %d x %d, total appended per call %d\n", rt, blocks, threads, total);
    for (int use_memset = 1; use_memset >= 0; --use_memset) {
        hipStream_t s;
        CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        // an eager call first (as Detector._replay does): the counter is left at `total`
        if (use_memset) CK(hipMemsetAsync(counter, 0, 4, s)); else reset_kernel<<<1, 64, 0, s>>>(counter);
        append_kernel<<<blocks, threads, 0, s>>>(counter, max_base);
        read_kernel<<<1, 64, 0, s>>>(counter, seen, max_base, max_seen);
        CK(hipStreamSynchronize(s));
        hipGraph_t g;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
        if (use_memset) CK(hipMemsetAsync(counter, 0, 4, s)); else reset_kernel<<<1, 64, 0, s>>>(counter);
        append_kernel<<<blocks, threads, 0, s>>>(counter, max_base);
        read_kernel<<<1, 64, 0, s>>>(counter, seen, max_base, max_seen);
        CK(hipStreamEndCapture(s, &g));
        size_t nn = 0, ne = 0;
        CK(hipGraphGetNodes(g, nullptr, &nn));
        std::vector<hipGraphNode_t> nodes(nn);
        CK(hipGraphGetNodes(g, nodes.data(), &nn));
        CK(hipGraphGetEdges(g, nullptr, nullptr, &ne));
        std::vector<hipGraphNode_t> from(ne), to(ne);
        if (ne) CK(hipGraphGetEdges(g, from.data(), to.data(), &ne));
        printf("\n== %s: %zu nodes, %zu edges\n", use_memset ? "counter cleared by hipMemsetAsync (rounds 3-5)" : "counter cleared by a kernel (since 8a071c8)", nn, ne);
        auto idx = [&](hipGraphNode_t n) { for (size_t i = 0; i < nn; ++i) if (nodes[i] == n) return (int)i; return -1; };
        for (size_t i = 0; i < nn; ++i) {
            hipGraphNodeType t; CK(hipGraphNodeGetType(nodes[i], &t));
            size_t nd = 0; CK(hipGraphNodeGetDependencies(nodes[i], nullptr, &nd));
            std::vector<hipGraphNode_t> deps(nd);
            if (nd) CK(hipGraphNodeGetDependencies(nodes[i], deps.data(), &nd));
            printf("  node %zu: %-6s depends on:", i, type_name(t));
            for (size_t d = 0; d < nd; ++d) printf(" %d", idx(deps[d]));
            if (t == hipGraphNodeTypeMemset) {
                hipMemsetParams mp; CK(hipGraphMemsetNodeGetParams(nodes[i], &mp));
                printf("   [memset: elementSize %u, width %zu, height %zu, value %u]", mp.elementSize, mp.width, mp.height, mp.value);
            }
            printf("\n");
        }
        for (size_t e = 0; e < ne; ++e) printf("  edge %d -> %d\n", idx(from[e]), idx(to[e]));
        hipGraphExec_t ex;
        CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
        for (int sync_each = 1; sync_each >= 0; --sync_each) {
            const int replays = 2000;
            int bad = 0, bad_final = 0, worst = 0;
            for (int r = 0; r < replays; ++r) {
                CK(hipGraphLaunch(ex, s));
                if (sync_each || r + 1 == replays || (r & 63) == 63) {
                    CK(hipStreamSynchronize(s));
                    int hs = 0, hm = 0;
                    CK(hipMemcpy(&hs, seen, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(&hm, max_seen, 4, hipMemcpyDeviceToHost));
                    if (hm >= total) { ++bad; if (hm > worst) worst = hm; }
                    if (hs != total) ++bad_final;
                }
            }
            printf("  %d replays, %s: appends that saw a counter the reset had not cleared: %d checks (largest base %d); final count != %d: %d checks\n",
                   replays, sync_each ? "synchronised after each" : "back to back (checked every 64th)", bad, worst, total, bad_final);
        }
        CK(hipGraphExecDestroy(ex)); CK(hipGraphDestroy(g)); CK(hipStreamDestroy(s));
    }
    return 0;
}
