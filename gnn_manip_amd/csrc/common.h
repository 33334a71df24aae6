// Internal helpers shared by the HIP translation units of libgnnmanip_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <mutex>
#include "../../include/gnn_manip_hip.h"

namespace gm {

void set_error(const char* fmt, ...);

#define GM_HIP_CHECK(expr)                                                                  \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            gm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,  \
                          __LINE__);                                                        \
            return GM_ERR_HIP;                                                              \
        }                                                                                   \
    } while (0)

#define GM_REQUIRE(cond, code, ...)          \
    do {                                     \
        if (!(cond)) {                       \
            gm::set_error(__VA_ARGS__);      \
            return (code);                   \
        }                                    \
    } while (0)

#define GM_LAUNCH_CHECK() GM_HIP_CHECK(hipGetLastError())

// Makes the device that owns `p` current for the duration of an entry point (the caller may have tensors on cuda:N while
// another device is current: launches, hipMalloc and the dynamic-LDS attributes all go by the current device).
struct DevGuard {
    int prev = -1, target = -1;
    explicit DevGuard(const void* p) {
        if (!p) return;
        hipPointerAttribute_t a;
        if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return; }
        if (a.type != hipMemoryTypeDevice) return;
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess || cur == a.device) return;
        if (hipSetDevice(a.device) == hipSuccess) { prev = cur; target = a.device; }
    }
    ~DevGuard() { if (prev >= 0 && target >= 0) (void)hipSetDevice(prev); }
    DevGuard(const DevGuard&) = delete;
    DevGuard& operator=(const DevGuard&) = delete;
};

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: runs `setup` once per device and call site, under a lock
// (entry points may be called from several host threads), and remembers the device only when the setup succeeded -- a
// failed attempt is retried by the next launch instead of leaving the attribute unset behind an opaque launch error.
struct PerDeviceOnce {
    std::mutex mu;
    bool done[64] = {};
    template <class F>
    int run(F&& setup) {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return setup();
        std::lock_guard<std::mutex> lock(mu);
        if (done[d]) return GM_OK;
        const int rc = setup();
        if (rc == GM_OK) done[d] = true;
        return rc;
    }
};

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Carves a caller-owned workspace into aligned arrays.
struct Carver {
    char* base;
    size_t off;
    explicit Carver(void* p) : base(static_cast<char*>(p)), off(0) {}
    template <typename T>
    T* take(size_t count) {
        off = align_up(off, 256);
        T* r = reinterpret_cast<T*>(base + off);
        off += count * sizeof(T);
        return r;
    }
    size_t used() const { return align_up(off, 256); }
};

// ---- error flags written by device code into workspace headers
enum : int { ERRF_NONFINITE_POS = 1, ERRF_BAD_EDGE_INDEX = 2, ERRF_CAPACITY = 4, ERRF_SPLIT_RANGE = 8 };

// ---- graph workspace (graph.hip) ---------------------------------------------------------
struct GraphHeader {  // lives at the start of the graph workspace (device memory)
    int n_edges;      // E of the last build
    int error_flags;
    int ncells;
    int dims[3];
    unsigned bbox_min[3];  // order-preserving uint encoding of float
    unsigned bbox_max[3];
    double origin[3];
    double inv_h;
    int n_per_graph;   // nodes per graph of a batch of equal-sized graphs (= n for a single graph)
    int ncells_local;  // cells of one graph's grid; graph b owns cells [b*ncells_local, (b+1)*ncells_local)
    int ticket;        // blocks of the bounding-box kernel that have finished (the last one derives the grid)
    int order_skip;    // cell_order(): a cell held more rows than its ranking loop takes -- the order is the identity this time
};

struct GraphWs {
    GraphHeader* hdr;
    int* cell_of;      // [N]
    int* cell_start;   // [max_cells + 1]  counts, then exclusive offsets
    int* cell_cursor;  // [max_cells]
    float4* sorted;    // [N] (x, y, z, index bits), grouped by cell
    int* cnt;          // [N + 1] neighbours kept per query (cnt[N] = 0)
    int* out_ptr;      // [N + 1] exclusive scan of cnt; out_ptr[N] = E
    int* nbr;          // [N * max_nb] neighbour lists, distance-sorted
    int* scan_cells;   // state of the single-pass scan over the cell counts (scan_state_ints)
    int* scan_cnt;     // ... over cnt
    int max_cells;
    size_t bytes;
};
int max_cells_for(int64_t n);
GraphWs carve_graph(void* ws, int64_t n, int max_nb);

// ---- destination-sorted edge structure (graph.hip) ----------------------------------------
struct CsrHeader {
    int n_edges;
    int error_flags;
    int flow;      // 0: dst = edge_index[1] (receiver), src = edge_index[0] (sender); 1: the other way round
    int pad;
};
struct CsrWs {
    CsrHeader* hdr;
    int* in_ptr;   // [N + 1] exclusive scan of in-degree
    int* cursor;   // [N]
    int* dst;      // [cap] aggregation node of sorted position p
    int* src;      // [cap]
    int* eid;      // [cap] original edge id (row in the caller's edge order)
    int* scan_tmp;  // block sums of the three-kernel scan (csr_from_edge_index)
    int* scan_in;   // state of the single-pass scan over the in-degrees (scan_state_ints)
    int* sort_tmp;  // [2 * cap] copies of long segments during the destination sort
    int64_t cap;
    int* blocks;   // block / chunk tables for the systolic edge kernel (hedge.h: carve_edge_blocks)
    size_t bytes;
};
CsrWs carve_csr(void* ws, int64_t n, int64_t cap);

// ---- optional per-kernel timing with HIP events on the launch stream, owned by a model handle (gm_model_profile)
// PROF_REST: everything else of a rollout step -- state update + node features, destination sort + block tables + edge features,
// the forward's clears, integration + window shift -- so that the kinds of a step add up to the step
enum ProfKind : int { PROF_EDGE = 0, PROF_NODE = 1, PROF_GRAPH = 2, PROF_ENC = 3, PROF_REST = 4, PROF_KINDS = 5 };
constexpr int PROF_MAX = 4096;
struct ProfState {
    int mask = 0;  // bit k: record kind k
    int count[PROF_KINDS] = {};
    int dropped[PROF_KINDS] = {};   // scopes that found the kind's PROF_MAX event pairs used up (gm_model_profile_query reports them)
    hipEvent_t* start[PROF_KINDS] = {};
    hipEvent_t* stop[PROF_KINDS] = {};
    ~ProfState();
};
struct ProfScope {
    ProfState* p;
    int idx;
    int kind;
    hipStream_t s;
    ProfScope(ProfState* state, int kind, hipStream_t s);   // state may be null: nothing is recorded
    ~ProfScope();
};

struct StepClear;
// fused kernels of the rollout step (features.hip)
int rollout_pre_features(float* obs, int64_t n, const gm_feature_desc* d, const int* rank, const float* target, float* out,
                         hipStream_t s, const StepClear* clear);   // clear (or nullptr): the step's resets, done by the same launch
int rollout_integrate_post(float* obs, int64_t n, const gm_feature_desc* d, const float* pred, const int* rank,
                           const float* target, float* pred_out, hipStream_t s);
// renumbered rollout (features.hip): out[t][j] = in[t][src(j)] over the k frames, src = perm (or the identity when the order was
// skipped: GraphHeader::order_skip of graph_ws); total_out[j] = total_in ? total_in[src] : src; rank_out[j] = rank_caller[total_out[j]]
int renumber_gather(const float* in, float* out, int k, int64_t n, int D, const int* perm, const void* graph_ws, const int* total_in,
                    int* total_out, const int* rank_caller, int* rank_out, hipStream_t s);
// out[t][total[j]] = in[t][j] over `frames` frames
int renumber_scatter(const float* in, float* out, int frames, int64_t n, int D, const int* total, hipStream_t s);
// destination sort of the radius graph with the edge features computed in the same pass (graph.hip)
int csr_from_graph_with_features(const void* graph_ws, int64_t n, int K, void* csr_ws, size_t csr_ws_bytes, const float* pos,
                                 int64_t pos_stride, float conn_r, float* edge_attr, int flow, hipStream_t s);
// destination-sorted structure of an edge_index; with_blocks = false leaves the inference kernels' block tables out (training)
int csr_from_edge_index(const int64_t* ei, int64_t n, int64_t e, int flow, void* csr_ws, size_t csr_ws_bytes, bool with_blocks, hipStream_t stream);

// Order of the rows by (grid cell of the radius-graph grid over `pos`, row index) -- graphs of a batch stay apart, in order: the
// deterministic form of the build's own cell sort (graph.hip).  perm[j] = row that comes j-th.  Uses graph_ws as scratch (the
// next build clears it); hdr->order_skip tells the consumers (launched after this) that perm is to be read as the identity.
int cell_order(const float* pos, int64_t pos_stride, int64_t n, int64_t n_per, double conn_r, int K, void* graph_ws, size_t graph_ws_bytes,
               int* perm, hipStream_t s);
int exclusive_scan_i32(const int* in, int* out, int64_t n_max, const int* n_dev, int* tmp, hipStream_t s, int* total_out);
size_t scan_tmp_ints(int64_t n_max);

// Single-pass exclusive scan (decoupled look-back: a tile publishes its sum, then its inclusive prefix, in one 64-bit status word;
// tiles are handed out by a ticket so that a tile only ever waits for tiles that are already running): ONE launch per scan, and two
// independent scans can share it.  `state` (scan_state_ints(n) ints, 8-byte aligned) must be ZERO when the launch starts: the
// callers' clear pass (StepClear) does that.  out may alias in; total_out (optional) receives the sum of all n items.
size_t scan_state_ints(int64_t n_max);
struct ScanJob {
    const int* in;
    int* out;
    long long n;
    int* total_out;
    int* state;
};
int scan_lookback(const ScanJob* jobs, int n_jobs, hipStream_t s);

// Everything a build has to reset before its first kernel, as ONE pass that can ride in another launch (the rollout step's first
// kernel) or be a launch of its own (the stand-alone entry points): value fills + the two workspace headers.
struct ClearJob {
    int* p;
    long long n;
    int v;
};
constexpr int kClearJobsMax = 12;
struct GraphHeader;
struct StepClear {
    int n_jobs = 0;
    ClearJob job[kClearJobsMax] = {};
    GraphHeader* gh = nullptr;   // reset for a build
    CsrHeader* ch = nullptr;     // reset for a destination sort (flow = `flow`)
    int flow = 0;
    void add(int* p, long long n, int v) { if (p && n > 0 && n_jobs < kClearJobsMax) job[n_jobs++] = ClearJob{p, n, v}; }
};
int launch_step_clear(const StepClear& c, hipStream_t s);
#if defined(__HIPCC__)
// thread t of nt: its share of every job (16-byte stores where the array allows), thread 0 the headers
__device__ inline void step_clear_run(const StepClear& c, long long t, long long nt) {
    for (int q = 0; q < c.n_jobs; ++q) {
        int* p = c.job[q].p;
        const long long n = c.job[q].n;
        const int v = c.job[q].v;
        if ((reinterpret_cast<uintptr_t>(p) & 15) == 0 && n >= 4) {
            const long long n4 = n >> 2;
            int4* p4 = reinterpret_cast<int4*>(p);
            for (long long i = t; i < n4; i += nt) p4[i] = make_int4(v, v, v, v);
            for (long long i = (n4 << 2) + t; i < n; i += nt) p[i] = v;
        } else {
            for (long long i = t; i < n; i += nt) p[i] = v;
        }
    }
    if (t == 0) {
        if (GraphHeader* hdr = c.gh) {
            hdr->n_edges = 0;
            hdr->error_flags = 0;
            hdr->ncells = 1;
            for (int a = 0; a < 3; ++a) {
                hdr->bbox_min[a] = 0xffffffffu;
                hdr->bbox_max[a] = 0u;
                hdr->dims[a] = 1;
                hdr->origin[a] = 0.0;
            }
            hdr->inv_h = 1.0;
            hdr->n_per_graph = 1;
            hdr->ncells_local = 1;
            hdr->ticket = 0;
            hdr->order_skip = 0;
        }
        if (CsrHeader* ch = c.ch) {
            ch->n_edges = 0;
            ch->error_flags = 0;
            ch->flow = c.flow;
            ch->pad = 0;
        }
    }
}
#endif
struct GraphWs;
struct CsrWs;
void graph_clear_jobs(StepClear& c, const GraphWs& g, int64_t n);             // what gm_radius_graph_build resets
void csr_clear_jobs(StepClear& c, const CsrWs& w, int64_t n, int flow);        // what the destination sort of a radius graph resets

// The rollout step's graph path with its resets already done (StepClear in the step's first launch) and the in-degree count
// riding in the neighbour search: radius graph -> destination sort + edge features + block tables in 9 launches.
int radius_graph_build_fused(const float* pos, int64_t pos_stride, int64_t n, int64_t n_per, double conn_r, int K, void* graph_ws,
                             size_t graph_ws_bytes, void* csr_ws, size_t csr_ws_bytes, int flow, hipStream_t s);
int csr_from_graph_fused(const void* graph_ws, int64_t n, int K, void* csr_ws, size_t csr_ws_bytes, const float* pos, int64_t pos_stride,
                         float conn_r, float* edge_attr, int flow, hipStream_t s);

}  // namespace gm
