// Diagnostics only (no reference counterpart): a stream-ordered WATCHPOINT on a small device region.
//
// hpcs_debug_snapshot copies the watched bytes into a shadow buffer; hpcs_debug_watch, enqueued after every later
// launch (hpcs_amd/_lib.py, HPCS_WATCH), compares the region with its shadow and records the sequence number of the FIRST
// launch after which they differ.  Both are ordinary kernels, so they are captured into a hipGraph with the step and
// bracket every captured node; neither allocates, so the memory layout of the run under test is unchanged.  This is how
// the stray store of round 2 (DESIGN.md "The stray store") was attributed to its kernel.
#include <stdlib.h>
#include <string.h>

#include "common.h"

namespace {

__global__ void debug_snapshot_kernel(const unsigned* __restrict__ src, unsigned* __restrict__ dst, int nwords) {
    for (int i = threadIdx.x; i < nwords; i += blockDim.x) dst[i] = src[i];
}

// status[0] first sequence number with a mismatch (0: none yet), [1] word offset of the first differing word,
// [2] value found, [3] value expected, [4] number of differing words at that moment, [5] the word after it (an 8-byte
// store shows as two), [6] launches checked so far (all replays), [7] value of [6] when the mismatch was first seen
__global__ void debug_watch_kernel(const unsigned* __restrict__ watched, const unsigned* __restrict__ shadow, int nwords, int seq,
                                   int* __restrict__ status) {
    __shared__ int first, count;
    if (threadIdx.x == 0) { first = 0x7fffffff; count = 0; }
    __syncthreads();
    for (int i = threadIdx.x; i < nwords; i += blockDim.x)
        if (watched[i] != shadow[i]) { atomicMin(&first, i); atomicAdd(&count, 1); }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int n = ++status[6];
        if (count > 0 && status[0] == 0) {
            status[0] = seq;
            status[1] = first;
            status[2] = (int)watched[first];
            status[3] = (int)shadow[first];
            status[4] = count;
            status[5] = first + 1 < nwords ? (int)watched[first + 1] : 0;
            status[7] = n;
        }
    }
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int hpcs_debug_snapshot(const void* src, void* dst, int nbytes, hpcs_stream_t stream) {
    if (!src || !dst) return HPCS_ERR_NULL;
    if (nbytes <= 0 || nbytes % 4) return HPCS_ERR_SHAPE;
    hipLaunchKernelGGL(debug_snapshot_kernel, dim3(1), dim3(256), 0, ST, (const unsigned*)src, (unsigned*)dst, nbytes / 4);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

extern "C" int hpcs_debug_watch(const void* watched, const void* shadow, int nbytes, int seq, int32_t* status,
                                hpcs_stream_t stream) {
    if (!watched || !shadow || !status) return HPCS_ERR_NULL;
    if (nbytes <= 0 || nbytes % 4) return HPCS_ERR_SHAPE;
    hipLaunchKernelGGL(debug_watch_kernel, dim3(1), dim3(256), 0, ST, (const unsigned*)watched, (const unsigned*)shadow,
                       nbytes / 4, seq, (int*)status);
    HPCS_CHECK_LAUNCH();
    return HPCS_OK;
}

// Zero-fill of up to four buffers as ONE launch (csrc/common.h hpcs_zero_async): the step driver clears its flat gradient
// bucket and its fp64 accumulator arena with it.  Sizes in bytes, multiples of 4; NULL pointers are skipped.
extern "C" int hpcs_zero_buffers(void* p0, size_t n0, void* p1, size_t n1, void* p2, size_t n2, void* p3, size_t n3,
                                 hpcs_stream_t stream) {
    return hpcs_zero_async((hipStream_t)stream, p0, n0, p1, n1, p2, n2, p3, n3);
}

// Node census of a captured hipGraph: counts[t] = number of nodes of hipGraphNodeType t (0 kernel, 1 memcpy,
// 2 memset, 3 host, 4 child graph, 5 empty, ...; types >= 15 are counted in counts[15]), child graphs walked recursively.
// The step driver refuses a capture that holds a memset node: on this runtime those lose their arguments once a second
// graph is instantiated (DESIGN.md "The stray store"); torch ops inside the captured step could still emit them.
// Host-only, no launch.  Returns the total node count or a negative status.
static int graph_census(hipGraph_t graph, int64_t* counts, int depth) {
    if (depth > 8) return HPCS_ERR_SHAPE;
    size_t n = 0;
    hipError_t e = hipGraphGetNodes(graph, nullptr, &n);
    if (e != hipSuccess) return HPCS_ERR_HIP - (int)e;
    if (n == 0) return 0;
    hipGraphNode_t* nodes = (hipGraphNode_t*)malloc(n * sizeof(hipGraphNode_t));
    if (!nodes) return HPCS_ERR_SHAPE;
    e = hipGraphGetNodes(graph, nodes, &n);
    int total = 0;
    for (size_t i = 0; e == hipSuccess && i < n; ++i) {
        hipGraphNodeType t;
        e = hipGraphNodeGetType(nodes[i], &t);
        if (e != hipSuccess) break;
        counts[(int)t < 15 ? (int)t : 15] += 1;
        ++total;
        if (t == hipGraphNodeTypeGraph) {
            hipGraph_t child = nullptr;
            e = hipGraphChildGraphNodeGetGraph(nodes[i], &child);
            if (e != hipSuccess) break;
            const int sub = graph_census(child, counts, depth + 1);
            if (sub < 0) { free(nodes); return sub; }
            total += sub;
        }
    }
    free(nodes);
    return e == hipSuccess ? total : HPCS_ERR_HIP - (int)e;
}

extern "C" int hpcs_debug_graph_census(void* graph, int64_t* counts16) {
    if (!graph || !counts16) return HPCS_ERR_NULL;
    for (int i = 0; i < 16; ++i) counts16[i] = 0;
    return graph_census((hipGraph_t)graph, counts16, 0);
}

// ---- tuning table (common.h HpcsTuning): read once, then immutable -------------------------------------------------------
static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return (e && *e) ? atoi(e) : dflt;
}
static HpcsTuning read_tuning() {
    HpcsTuning t;
    t.nt_wg_target = env_int("HPCS_NT_WG_TARGET", 0);
    t.nt_big_min = env_int("HPCS_NT_BIG_MIN", 0);        // 0: 128 with the bf16x3 kernels, 300 with the fma-chain ones (gemm.hip)
    t.nt_big_wg = env_int("HPCS_NT_BIG_WG", 512);
    t.gemm_vecstore = env_int("HPCS_GEMM_VECSTORE", 1) != 0;
    t.gemm_tn_guarded = env_int("HPCS_GEMM_TN_GUARDED", 0) != 0;
    t.gemm_x3 = env_int("HPCS_GEMM_X3", 1);
    const char* k = getenv("HPCS_KNN_KERNEL");
    t.knn_kernel = !k || !*k ? 0 : (k[0] == 't' ? 1 : (k[0] == 'b' ? (k[1] == 'q' && k[2] == '2' ? 5 : 4) : 2));
    t.sampled_v = env_int("HPCS_SAMPLED_V", 4) == 8 ? 8 : 4;
    t.sampled_seg = env_int("HPCS_SAMPLED_SEG", 0);
    t.sampled_lds_kb = env_int("HPCS_SAMPLED_LDS_KB", 156);
    t.edge_bwd_atomic = getenv("HPCS_EDGE_BWD_ATOMIC") != nullptr;
    t.x3_nt_lds_kb = env_int("HPCS_X3_NT_LDS_KB", 56);
    return t;
}
static HpcsTuning& tuning_storage() {
    static HpcsTuning t = read_tuning();           // thread-safe one-time initialisation
    return t;
}
extern "C" const HpcsTuning* hpcs_tuning_table(void) { return &tuning_storage(); }

// Tests only: replace one value of the table (names = the environment variables' names).  Returns 0, or HPCS_ERR_SHAPE for an
// unknown name.
extern "C" int hpcs_debug_set_tuning(const char* name, int value) {
    if (!name) return HPCS_ERR_NULL;
    HpcsTuning& t = tuning_storage();
    struct { const char* n; int* p; } tab[] = {
        {"HPCS_NT_WG_TARGET", &t.nt_wg_target}, {"HPCS_NT_BIG_MIN", &t.nt_big_min}, {"HPCS_NT_BIG_WG", &t.nt_big_wg},
        {"HPCS_GEMM_VECSTORE", &t.gemm_vecstore}, {"HPCS_GEMM_TN_GUARDED", &t.gemm_tn_guarded}, {"HPCS_GEMM_X3", &t.gemm_x3}, {"HPCS_KNN_KERNEL", &t.knn_kernel},
        {"HPCS_SAMPLED_V", &t.sampled_v}, {"HPCS_SAMPLED_SEG", &t.sampled_seg}, {"HPCS_SAMPLED_LDS_KB", &t.sampled_lds_kb},
        {"HPCS_EDGE_BWD_ATOMIC", &t.edge_bwd_atomic}, {"HPCS_X3_NT_LDS_KB", &t.x3_nt_lds_kb}};
    for (auto& e : tab)
        if (strcmp(e.n, name) == 0) { *e.p = value; return HPCS_OK; }
    return HPCS_ERR_SHAPE;
}
