// Radius graph (uniform cell list, float64 distance test, per-query top-k by (d2, index)) and
// the destination-sorted edge structure.  HBM-light integer work: coalesced streams, LDS-held
// per-query candidate lists, no host synchronisation.
//
// Replaces get_connectivity, gnn_manip/utils/utils.py:64-93 (sklearn KDTree.query_radius:
// float64 d2 = sum_j (x_j - y_j)^2 accumulated j = 0..2, kept when d2 <= r*r, ordered by
// ascending distance, first max_neighbours kept).
#include <stdarg.h>
#include "common.h"
#include "hedge.h"
#include "blocks_dev.h"

namespace gm {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
const char* last_error() { return g_err; }

// ------------------------------------------------------------------------------------------
// profiling hooks
// ------------------------------------------------------------------------------------------
ProfState::~ProfState() {
    for (int k = 0; k < PROF_KINDS; ++k) {
        if (!start[k]) continue;
        for (int i = 0; i < PROF_MAX; ++i) { (void)hipEventDestroy(start[k][i]); (void)hipEventDestroy(stop[k][i]); }
        delete[] start[k];
        delete[] stop[k];
    }
}

ProfScope::ProfScope(ProfState* state, int k, hipStream_t st) : p(state), idx(-1), kind(k), s(st) {
    if (!p || !((p->mask >> k) & 1)) return;
    if (p->count[k] >= PROF_MAX) { ++p->dropped[k]; return; }
    if (!p->start[k]) {
        hipEvent_t* a = new hipEvent_t[PROF_MAX];
        hipEvent_t* b = new hipEvent_t[PROF_MAX];
        int made = 0;
        bool ok = true;
        for (; made < PROF_MAX && ok; ++made) {
            if (hipEventCreate(&a[made]) != hipSuccess) { ok = false; break; }
            if (hipEventCreate(&b[made]) != hipSuccess) { (void)hipEventDestroy(a[made]); ok = false; break; }
        }
        if (!ok) {   // all or nothing: a partly created set is released and this kind stays unrecorded
            for (int i = 0; i < made; ++i) { (void)hipEventDestroy(a[i]); (void)hipEventDestroy(b[i]); }
            delete[] a;
            delete[] b;
            (void)hipGetLastError();
            return;
        }
        p->start[k] = a;
        p->stop[k] = b;
    }
    idx = p->count[k]++;
    (void)hipEventRecord(p->start[k][idx], s);
}
ProfScope::~ProfScope() {
    if (idx >= 0) (void)hipEventRecord(p->stop[kind][idx], s);
}

// ------------------------------------------------------------------------------------------
// exclusive scan of int32, n known on the host or read from device memory
// ------------------------------------------------------------------------------------------
constexpr int SCAN_BLOCK = 256;
constexpr int SCAN_ITEMS = 4;
constexpr int SCAN_TILE = SCAN_BLOCK * SCAN_ITEMS;

size_t scan_tmp_ints(int64_t n_max) { return (size_t)cdiv(n_max, SCAN_TILE) + 1; }

__device__ __forceinline__ int wave_incl_scan(int v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// block-wide exclusive scan of one int per thread (blockDim.x == SCAN_BLOCK); returns exclusive
// prefix, *total = block sum.  sm: >= 4 ints of LDS.
__device__ __forceinline__ int block_excl_scan(int v, int* sm, int* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int incl = wave_incl_scan(v, lane);
    if (lane == 63) sm[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SCAN_BLOCK / 64; ++w) {
        int s = sm[w];
        if (w < wave) base += s;
        tot += s;
    }
    __syncthreads();
    *total = tot;
    return base + incl - v;
}

__global__ void __launch_bounds__(SCAN_BLOCK) scan_reduce_kernel(const int* __restrict__ in, int64_t n_host,
                                                                  const int* __restrict__ n_dev,
                                                                  int* __restrict__ tmp) {
    __shared__ int sm[4];
    const int64_t n = n_dev ? (int64_t)(*n_dev) : n_host;
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
    if (base >= n) return;
    int v = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        int64_t i = base + (int64_t)threadIdx.x * SCAN_ITEMS + k;
        if (i < n) v += in[i];
    }
    int tot;
    block_excl_scan(v, sm, &tot);
    if (threadIdx.x == 0) tmp[blockIdx.x] = tot;
}

// second pass: every block adds up the tile sums in front of its tile itself (a few hundred values at most: cheaper than a
// launch of its own for a one-block scan of them), scans its tile and -- the block that holds the last item -- writes the total
__global__ void __launch_bounds__(SCAN_BLOCK) scan_apply_kernel(const int* in, int64_t n_host,
                                                                 const int* __restrict__ n_dev,
                                                                 const int* __restrict__ tmp, int* out, int* total_out) {
    __shared__ int sm[4];
    const int64_t n = n_dev ? (int64_t)(*n_dev) : n_host;
    const int64_t base = (int64_t)blockIdx.x * SCAN_TILE;
    if (base >= n) return;
    int item[SCAN_ITEMS];
    int v = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        int64_t i = base + (int64_t)threadIdx.x * SCAN_ITEMS + k;
        item[k] = i < n ? in[i] : 0;
        v += item[k];
    }
    int before = 0;
    for (int i = threadIdx.x; i < (int)blockIdx.x; i += SCAN_BLOCK) before += tmp[i];
    int prefix;
    (void)block_excl_scan(before, sm, &prefix);   // prefix = sum of the tile sums of the blocks in front
    int tot;
    int ex = block_excl_scan(v, sm, &tot) + prefix;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) {
        int64_t i = base + (int64_t)threadIdx.x * SCAN_ITEMS + k;
        if (i < n) out[i] = ex;
        if (total_out && i == n - 1) *total_out = ex;   // callers that ask for the total append a zero item: the last output is the sum
        ex += item[k];
    }
}

// small arrays: one 1024-thread block walks the array in chunks with a running carry (one launch instead
// of three; at N = 5k every scan of the graph build is this small)
constexpr int SCAN1_THREADS = 1024;
constexpr int64_t SCAN1_MAX = 40000;  // one workgroup up to here (~10 us at 32k); above, the three-kernel scan (~15 us) wins
__global__ void __launch_bounds__(SCAN1_THREADS) scan_single_kernel(const int* in, int64_t n, int* out, int* total_out) {
    __shared__ int wsum[SCAN1_THREADS / 64];
    __shared__ int carry_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int64_t b = 0; b < n; b += SCAN1_THREADS * SCAN_ITEMS) {
        int item[SCAN_ITEMS];
        int v = 0;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) {
            const int64_t i = b + (int64_t)threadIdx.x * SCAN_ITEMS + k;
            item[k] = i < n ? in[i] : 0;
            v += item[k];
        }
        const int incl = wave_incl_scan(v, lane);
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int base = carry_s;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        int ex = base + incl - v;
#pragma unroll
        for (int k = 0; k < SCAN_ITEMS; ++k) {
            const int64_t i = b + (int64_t)threadIdx.x * SCAN_ITEMS + k;
            if (i < n) out[i] = ex;
            ex += item[k];
        }
        __syncthreads();
        if (threadIdx.x == SCAN1_THREADS - 1) carry_s = base + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry_s;
}

// out may alias in.  Scans n items (n = *n_dev when n_dev != nullptr, bounded by n_max).  total_out (optional)
// receives the sum; callers that ask for it append a zero item, so the sum is also the last output.
int exclusive_scan_i32(const int* in, int* out, int64_t n_max, const int* n_dev, int* tmp, hipStream_t s, int* total_out) {
    if (n_max <= 0) return GM_OK;
    if (!n_dev && n_max <= SCAN1_MAX) {
        hipLaunchKernelGGL(scan_single_kernel, dim3(1), dim3(SCAN1_THREADS), 0, s, in, n_max, out, total_out);
        GM_LAUNCH_CHECK();
        return GM_OK;
    }
    const int nb = (int)cdiv(n_max, SCAN_TILE);
    GM_REQUIRE(!total_out || !n_dev, GM_ERR_INVALID_ARGUMENT, "exclusive_scan_i32: total_out needs a host-side length");
    hipLaunchKernelGGL(scan_reduce_kernel, dim3(nb), dim3(SCAN_BLOCK), 0, s, in, n_max, n_dev, tmp);
    hipLaunchKernelGGL(scan_apply_kernel, dim3(nb), dim3(SCAN_BLOCK), 0, s, in, n_max, n_dev, tmp, out, total_out);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

// ------------------------------------------------------------------------------------------
// single-pass scan (decoupled look-back), up to two independent scans per launch
// ------------------------------------------------------------------------------------------
constexpr int SCAN2_ITEMS = 8;
constexpr int SCAN2_TILE = SCAN_BLOCK * SCAN2_ITEMS;
constexpr int SCAN2_JOBS = 2;
size_t scan_state_ints(int64_t n_max) { return 2 * (size_t)cdiv(n_max > 0 ? n_max : 1, SCAN2_TILE) + 4; }   // [ticket, pad | status u64 per tile]
struct ScanJobs {
    ScanJob j[SCAN2_JOBS];
};
typedef unsigned long long u64;
__device__ __forceinline__ u64 status_load(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void status_store(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__global__ void __launch_bounds__(SCAN_BLOCK) scan_lookback_kernel(ScanJobs J) {
    __shared__ int sm[4];
    __shared__ int s_tile, s_prefix;
    const ScanJob& job = J.j[blockIdx.y];
    const long long n = job.n;
    const int tiles = (int)((n + SCAN2_TILE - 1) / SCAN2_TILE);
    if ((int)blockIdx.x >= tiles) return;   // exactly `tiles` workgroups take a ticket
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // tiles in ticket order: a tile only waits for tiles with a smaller ticket, and those have started
    if (tid == 0) s_tile = atomicAdd(&job.state[0], 1);
    __syncthreads();
    const int tile = s_tile;
    const long long base = (long long)tile * SCAN2_TILE + (long long)tid * SCAN2_ITEMS;
    int item[SCAN2_ITEMS];
    int v = 0;
#pragma unroll
    for (int k = 0; k < SCAN2_ITEMS; ++k) {
        item[k] = base + k < n ? job.in[base + k] : 0;
        v += item[k];
    }
    int tot;
    int ex = block_excl_scan(v, sm, &tot);
    if (wave == 0) {
        u64* st = reinterpret_cast<u64*>(job.state + 2);
        int prefix = 0;
        if (tile == 0) {
            if (lane == 0) status_store(st, (2ull << 32) | (unsigned)tot);
        } else {
            if (lane == 0) status_store(st + tile, (1ull << 32) | (unsigned)tot);    // this tile's sum
            int back = tile - 1;
            while (true) {   // 64 predecessors per look: sums up to the nearest tile that already knows its inclusive prefix
                const int p = back - lane;
                u64 sv = 2ull << 32;   // in front of tile 0: inclusive prefix 0
                if (p >= 0) {
                    do { sv = status_load(st + p); } while ((sv >> 32) == 0);
                }
                const u64 inc = __ballot((sv >> 32) == 2);
                int val = (int)(unsigned)sv;
                if (inc) val = lane <= (int)__builtin_ctzll(inc) ? val : 0;
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) val += __shfl_xor(val, d, 64);
                prefix += val;
                if (inc) break;
                back -= 64;
            }
            if (lane == 0) status_store(st + tile, (2ull << 32) | (unsigned)(prefix + tot));
        }
        if (lane == 0) s_prefix = prefix;
    }
    __syncthreads();
    ex += s_prefix;
#pragma unroll
    for (int k = 0; k < SCAN2_ITEMS; ++k) {
        if (base + k < n) job.out[base + k] = ex;
        ex += item[k];
    }
    if (job.total_out && tile == tiles - 1 && tid == 0) *job.total_out = s_prefix + tot;
}

int scan_lookback(const ScanJob* jobs, int n_jobs, hipStream_t s) {
    GM_REQUIRE(n_jobs >= 1 && n_jobs <= SCAN2_JOBS, GM_ERR_INVALID_ARGUMENT, "scan_lookback: 1 or 2 scans per launch");
    ScanJobs J{};
    long long nmax = 0;
    for (int i = 0; i < n_jobs; ++i) {
        J.j[i] = jobs[i];
        GM_REQUIRE(jobs[i].in && jobs[i].out && jobs[i].state && jobs[i].n >= 0, GM_ERR_INVALID_ARGUMENT, "scan_lookback: bad job");
        nmax = jobs[i].n > nmax ? jobs[i].n : nmax;
    }
    if (nmax <= 0) return GM_OK;
    hipLaunchKernelGGL(scan_lookback_kernel, dim3((unsigned)cdiv(nmax, SCAN2_TILE), (unsigned)n_jobs), dim3(SCAN_BLOCK), 0, s, J);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

// ------------------------------------------------------------------------------------------
// resets of a build (StepClear, common.h)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) step_clear_kernel(StepClear c) {
    step_clear_run(c, (long long)blockIdx.x * blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x);
}
int launch_step_clear(const StepClear& c, hipStream_t s) {
    long long work = 1;
    for (int q = 0; q < c.n_jobs; ++q) work = c.job[q].n > work ? c.job[q].n : work;
    long long nb = cdiv(work, 256 * 4);
    nb = nb < 1 ? 1 : (nb > 1024 ? 1024 : nb);
    hipLaunchKernelGGL(step_clear_kernel, dim3((unsigned)nb), dim3(256), 0, s, c);
    GM_LAUNCH_CHECK();
    return GM_OK;
}
void graph_clear_jobs(StepClear& c, const GraphWs& g, int64_t n) {
    c.gh = g.hdr;
    c.add(g.cell_start, (long long)g.max_cells + 1, 0);   // counts -> offsets
    c.add(g.cell_cursor, g.max_cells, 0);
    c.add(g.cnt + n, 1, 0);                                // cnt[0 .. n) is written by the neighbour search; cnt[n] closes the scan
    c.add(g.scan_cells, (long long)scan_state_ints((int64_t)g.max_cells + 1), 0);
    c.add(g.scan_cnt, (long long)scan_state_ints(n + 1), 0);
}
void csr_clear_jobs(StepClear& c, const CsrWs& w, int64_t n, int flow) {
    c.ch = w.hdr;
    c.flow = flow;
    c.add(w.in_ptr, n + 1, 0);
    c.add(w.scan_in, (long long)scan_state_ints(n + 1), 0);
    c.add(carve_edge_blocks(w.blocks, n, w.cap).stitch, n, -1);
}

// ------------------------------------------------------------------------------------------
// workspace carving
// ------------------------------------------------------------------------------------------
int max_cells_for(int64_t n) {
    int64_t c = 2 * n;
    if (c < 32768) c = 32768;
    if (c > (1 << 22)) c = (1 << 22);
    return (int)c;
}

GraphWs carve_graph(void* ws, int64_t n, int max_nb) {
    GraphWs g;
    Carver c(ws);
    g.max_cells = max_cells_for(n);
    g.hdr = c.take<GraphHeader>(1);
    g.cell_of = c.take<int>(n);
    g.cell_start = c.take<int>((size_t)g.max_cells + 1);
    g.cell_cursor = c.take<int>(g.max_cells);
    g.sorted = c.take<float4>(n);
    g.cnt = c.take<int>(n + 1);
    g.out_ptr = c.take<int>(n + 1);
    g.nbr = c.take<int>((size_t)n * max_nb);
    g.scan_cells = c.take<int>(scan_state_ints((int64_t)g.max_cells + 1));
    g.scan_cnt = c.take<int>(scan_state_ints(n + 1));
    g.bytes = c.used();
    return g;
}

CsrWs carve_csr(void* ws, int64_t n, int64_t cap) {
    CsrWs w;
    Carver c(ws);
    w.hdr = c.take<CsrHeader>(1);
    w.in_ptr = c.take<int>(n + 1);
    w.cursor = c.take<int>(n + 1);
    w.dst = c.take<int>(cap);
    w.src = c.take<int>(cap);
    w.eid = c.take<int>(cap);
    w.scan_tmp = c.take<int>(scan_tmp_ints(n + 1));
    w.scan_in = c.take<int>(scan_state_ints(n + 1));
    w.sort_tmp = c.take<int>(2 * (size_t)cap);   // copies of long segments (in-degree > 96) while they are rank-sorted
    w.cap = cap;
    w.blocks = c.take<int>(edge_blocks_ints(n, cap));
    w.bytes = c.used();
    return w;
}

// ------------------------------------------------------------------------------------------
// cell list
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned f2ord(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__device__ void grid_params(GraphHeader* hdr, double r, int max_cells, int64_t n, int64_t n_per);

// bounding box; the block that finishes last derives the grid from it (no separate launch)
__global__ void __launch_bounds__(256) bbox_kernel(const float* __restrict__ pos, int64_t stride, int64_t n,
                                                    GraphHeader* hdr, double r, int max_cells, int64_t n_per) {
    unsigned mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
    bool bad = false;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float v = pos[i * stride + a];
            if (!isfinite(v)) { bad = true; continue; }
            unsigned o = f2ord(v);
            mn[a] = min(mn[a], o);
            mx[a] = max(mx[a], o);
        }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            mn[a] = min(mn[a], (unsigned)__shfl_xor((int)mn[a], d, 64));
            mx[a] = max(mx[a], (unsigned)__shfl_xor((int)mx[a], d, 64));
        }
    }
    // one set of atomics per workgroup (the six addresses are shared by the whole grid: per-wave atomics serialised
    // to ~100 us at N = 100k)
    __shared__ unsigned wmn[4][3], wmx[4][3];
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            wmn[threadIdx.x >> 6][a] = mn[a];
            wmx[threadIdx.x >> 6][a] = mx[a];
        }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        atomicMin(&hdr->bbox_min[a], min(min(wmn[0][a], wmn[1][a]), min(wmn[2][a], wmn[3][a])));
        atomicMax(&hdr->bbox_max[a], max(max(wmx[0][a], wmx[1][a]), max(wmx[2][a], wmx[3][a])));
    }
    if (bad) atomicOr(&hdr->error_flags, ERRF_NONFINITE_POS);
    // last block done -> grid parameters.  Every block's atomics are ordered before its ticket by the
    // release fence; the last arriver acquires before reading the box.
    __shared__ int last;
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        last = atomicAdd(&hdr->ticket, 1) == (int)gridDim.x - 1;
    }
    __syncthreads();
    if (last && threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        // read the box through device-scope atomics (the values were produced by atomics in L2)
        for (int a = 0; a < 3; ++a) {
            hdr->bbox_min[a] = atomicMin(&hdr->bbox_min[a], 0xffffffffu);
            hdr->bbox_max[a] = atomicMax(&hdr->bbox_max[a], 0u);
        }
        grid_params(hdr, r, max_cells, n, n_per);
    }
}

// One thread: grid origin, cell edge h >= r*(1+2^-10) (so |x_i-x_j| <= r implies cell coordinates
// differ by at most 1 on every axis despite rounding), enlarged until the grid fits max_cells.
__device__ void grid_params(GraphHeader* hdr, double r, int max_cells, int64_t n, int64_t n_per) {
    // a batch of equal-sized graphs shares one grid geometry; every graph gets its own block of cells, so
    // no edge can cross graphs (the batch offset rule of collate_utils.py:76)
    const int64_t n_graphs = n_per > 0 && n > 0 ? (n + n_per - 1) / n_per : 1;
    max_cells = (int)(max_cells / n_graphs) > 0 ? (int)(max_cells / n_graphs) : 1;
    double lo[3], ext[3];
    for (int a = 0; a < 3; ++a) {
        double mn = n > 0 && hdr->bbox_min[a] <= hdr->bbox_max[a] ? (double)ord2f(hdr->bbox_min[a]) : 0.0;
        double mx = n > 0 && hdr->bbox_min[a] <= hdr->bbox_max[a] ? (double)ord2f(hdr->bbox_max[a]) : 0.0;
        lo[a] = mn;
        ext[a] = mx - mn;
    }
    double h = r * (1.0 + 1.0 / 1024.0);
    if (!(h > 0.0)) h = 1.0;
    int d[3];
    for (int it = 0; it < 256; ++it) {
        double prod = 1.0;
        for (int a = 0; a < 3; ++a) {
            double c = floor(ext[a] / h) + 1.0;
            if (c > 2.0e9) c = 2.0e9;
            d[a] = (int)c;
            prod *= c;
        }
        if (prod <= (double)max_cells) break;
        h *= 1.2599210498948732;  // doubles the cell volume
    }
    for (int a = 0; a < 3; ++a) {
        hdr->dims[a] = d[a];
        hdr->origin[a] = lo[a];
    }
    hdr->inv_h = 1.0 / h;
    hdr->ncells_local = d[0] * d[1] * d[2];
    hdr->ncells = hdr->ncells_local * (int)n_graphs;
    hdr->n_per_graph = (int)(n_per > 0 ? n_per : (n > 0 ? n : 1));
}

__device__ __forceinline__ int cell_coord(float v, double origin, double inv_h, int dim) {
    double c = floor(((double)v - origin) * inv_h);
    int ci = (c > 0.0) ? (c < (double)(dim - 1) ? (int)c : dim - 1) : 0;
    return ci;
}

__global__ void __launch_bounds__(256) cell_assign_kernel(const float* __restrict__ pos, int64_t stride, int64_t n,
                                                           const GraphHeader* __restrict__ hdr,
                                                           int* __restrict__ cell_of, int* __restrict__ cell_count) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int dx = hdr->dims[0], dy = hdr->dims[1], dz = hdr->dims[2];
    float x = pos[i * stride], y = pos[i * stride + 1], z = pos[i * stride + 2];
    if (!isfinite(x)) x = 0.f;
    if (!isfinite(y)) y = 0.f;
    if (!isfinite(z)) z = 0.f;
    int cx = cell_coord(x, hdr->origin[0], hdr->inv_h, dx);
    int cy = cell_coord(y, hdr->origin[1], hdr->inv_h, dy);
    int cz = cell_coord(z, hdr->origin[2], hdr->inv_h, dz);
    int c = (int)(i / hdr->n_per_graph) * hdr->ncells_local + (cz * dy + cy) * dx + cx;
    cell_of[i] = c;
    atomicAdd(&cell_count[c], 1);
}

__global__ void __launch_bounds__(256) cell_fill_kernel(const float* __restrict__ pos, int64_t stride, int64_t n,
                                                         const int* __restrict__ cell_of,
                                                         const int* __restrict__ cell_start,
                                                         int* __restrict__ cell_cursor, float4* __restrict__ sorted) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int c = cell_of[i];
    int p = cell_start[c] + atomicAdd(&cell_cursor[c], 1);
    sorted[p] = make_float4(pos[i * stride], pos[i * stride + 1], pos[i * stride + 2], __int_as_float((int)i));
}

// Rows in (cell, index) order: slot p of `sorted` holds row i of cell c; it comes (rows of c with a smaller index) places after the
// cell's first slot.  cell_fill_kernel places rows with an atomic cursor, so the order inside a cell differs from run to run; this
// ranking does not.  Cells of a dense scene hold ~5 rows; a cell beyond kCellOrderCap rows (a degenerate scene) would make the
// loop quadratic, so it raises order_skip instead and the caller keeps the rows where they are (the order only buys locality).
constexpr int kCellOrderCap = 1024;
__global__ void __launch_bounds__(256) cell_order_kernel(const float4* __restrict__ sorted, const int* __restrict__ cell_of,
                                                          const int* __restrict__ cell_start, GraphHeader* hdr, int64_t n,
                                                          int* __restrict__ perm) {
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const int i = __float_as_int(sorted[p].w);
    const int c = cell_of[i];
    const int lo = cell_start[c], hi = cell_start[c + 1];
    if (hi - lo > kCellOrderCap) {
        if (p == lo) atomicOr(&hdr->order_skip, 1);
        return;
    }
    int before = 0;
    for (int q = lo; q < hi; ++q) before += __float_as_int(sorted[q].w) < i ? 1 : 0;
    perm[lo + before] = i;
}

// ------------------------------------------------------------------------------------------
// neighbour search, fast path: NB_LPQ (16) lanes per query.  The lanes scan the 27-cell candidate ranges
// together, append in-radius candidates (d2, index) to the query's list in LDS through a ballot
// prefix, then rank the list (rank = number of smaller (d2, index) keys: no dependent chains, the
// order (d2, index) is total so the result is order-independent) and emit the first K.  Queries
// with more than NB_CAP in-radius candidates are marked (cnt = -1) for the general kernel below.
// ------------------------------------------------------------------------------------------
#ifndef NB_LPQ
#define NB_LPQ 16                      // lanes per query.  Round 6, same box: 4 lanes graph scope 77 us at C2 / 238 us at the target, 8 lanes
#endif                                 // (rounds 2 - 5) 55 / 170, 16 lanes 40 / 133: an x-run of three cells holds ~14 candidates, one step of 16 lanes
static_assert(NB_LPQ == 4 || NB_LPQ == 8 || NB_LPQ == 16, "lanes per query: a power of two that divides a wave, below 32 (the ballot mask)");
constexpr int NB_CAP = 96;
constexpr int NB_STRIDE = NB_CAP + 1;  // doubles per list: odd stride keeps the groups of a wave on distinct banks
constexpr int NB_QPB = 256 / NB_LPQ;   // queries per 256-thread block

// indeg / arrival (optional: the rollout step's fused path): the counting pass of the destination sort rides along -- every kept
// neighbour takes its arrival number within its aggregation node's segment (flow 0: the neighbour, 1: the query) from an atomic on
// indeg[], so that the fill pass places the edge with plain stores (see indeg_graph_kernel, the stand-alone form of the same count).
__global__ void __launch_bounds__(256) neighbor_fast_kernel(const float4* __restrict__ sorted,
                                                             const int* __restrict__ cell_start,
                                                             const GraphHeader* __restrict__ hdr, int64_t n, double r2,
                                                             int K, int* __restrict__ cnt, int* __restrict__ nbr,
                                                             int* __restrict__ indeg, int* __restrict__ arrival, int flow) {
    __shared__ double sd2[NB_QPB * NB_STRIDE];
    __shared__ int sj[NB_QPB * NB_STRIDE];
    const int tid = threadIdx.x, sub = tid & (NB_LPQ - 1), ql = tid / NB_LPQ, lane = tid & 63;
    const int grp_shift = (lane / NB_LPQ) * NB_LPQ;  // position of this group's bits in the wave ballot
    const int64_t slot = (int64_t)blockIdx.x * NB_QPB + ql;
    const bool active = slot < n;
    const float4 q = sorted[active ? slot : 0];
    const int qi = __float_as_int(q.w);
    const int dx = hdr->dims[0], dy = hdr->dims[1], dz = hdr->dims[2];
    const double inv_h = hdr->inv_h;
    const int cx = cell_coord(q.x, hdr->origin[0], inv_h, dx);
    const int cy = cell_coord(q.y, hdr->origin[1], inv_h, dy);
    const int cz = cell_coord(q.z, hdr->origin[2], inv_h, dz);
    const double qx = (double)q.x, qy = (double)q.y, qz = (double)q.z;
    const int cbase = (qi / hdr->n_per_graph) * hdr->ncells_local;  // this graph's block of cells
    double* ld = sd2 + ql * NB_STRIDE;
    int* lj = sj + ql * NB_STRIDE;
    int count = 0;  // group-uniform
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, dx - 1);
    if (active) {
        for (int z = max(cz - 1, 0); z <= min(cz + 1, dz - 1); ++z) {
            for (int y = max(cy - 1, 0); y <= min(cy + 1, dy - 1); ++y) {
                const int row = cbase + (z * dy + y) * dx;
                const int b = cell_start[row + x0], e = cell_start[row + x1 + 1];
                for (int c0 = b; c0 < e; c0 += NB_LPQ) {
                    const int c = c0 + sub;
                    bool in = false;
                    double d2 = 0.0;
                    int j = 0;
                    if (c < e) {
                        const float4 p = sorted[c];
                        double d = qx - (double)p.x;
                        d2 = __dmul_rn(d, d);
                        d = qy - (double)p.y;
                        d2 = __dadd_rn(d2, __dmul_rn(d, d));
                        d = qz - (double)p.z;
                        d2 = __dadd_rn(d2, __dmul_rn(d, d));
                        in = d2 <= r2;
                        j = __float_as_int(p.w);
                    }
                    const unsigned m = (unsigned)(__ballot(in) >> grp_shift) & ((1u << NB_LPQ) - 1u);
                    const int pos = count + __popc(m & ((1u << sub) - 1u));
                    if (in && pos < NB_CAP) {
                        ld[pos] = d2;
                        lj[pos] = j;
                    }
                    count += __popc(m);
                }
            }
        }
    }
    __syncthreads();
    if (!active) return;
    if (count > NB_CAP) {
        if (sub == 0) cnt[qi] = -1;  // general kernel takes this query
        return;
    }
    // rank = number of smaller (d2, index) keys.  A lane keeps four of its elements in registers and walks the list once for all
    // of them (the list reads are the same address for the 8 lanes of the group: one broadcast per element)
    for (int a0 = sub; a0 < count; a0 += 4 * NB_LPQ) {
        double da[4];
        int ja[4], rank[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int a = a0 + NB_LPQ * u;
            da[u] = a < count ? ld[a] : -1.0;   // d2 >= 0: nothing ranks below a filler, its rank is never used
            ja[u] = a < count ? lj[a] : 0;
            rank[u] = 0;
        }
        for (int f = 0; f < count; ++f) {
            const double df = ld[f];
            const int jf = lj[f];
#pragma unroll
            for (int u = 0; u < 4; ++u) rank[u] += (df < da[u] || (df == da[u] && jf < ja[u])) ? 1 : 0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (a0 + NB_LPQ * u < count && rank[u] < K) {
                nbr[(int64_t)qi * K + rank[u]] = ja[u];
                if (indeg) arrival[(int64_t)qi * K + rank[u]] = atomicAdd(&indeg[flow ? qi : ja[u]], 1);
            }
    }
    if (sub == 0) cnt[qi] = count < K ? count : K;
}

// ------------------------------------------------------------------------------------------
// neighbour search: thread t owns the query in sorted slot t; candidate lists live in LDS as
// [slot][thread] (conflict-free), kept sorted by (d2, index).
// ------------------------------------------------------------------------------------------
template <int BS>
__global__ void __launch_bounds__(BS) neighbor_kernel(const float4* __restrict__ sorted,
                                                       const int* __restrict__ cell_start,
                                                       const GraphHeader* __restrict__ hdr, int64_t n, double r2,
                                                       int K, int* __restrict__ cnt, int* __restrict__ nbr,
                                                       int* __restrict__ indeg, int* __restrict__ arrival, int flow) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double* kd = reinterpret_cast<double*>(smem);                   // [K][BS]
    int* ki = reinterpret_cast<int*>(smem + (size_t)K * BS * 8);    // [K][BS]
    const int t = threadIdx.x;
    const int64_t slot = (int64_t)blockIdx.x * BS + t;
    if (slot >= n) return;
    const float4 q = sorted[slot];
    const int qi = __float_as_int(q.w);
    if (cnt[qi] != -1) return;  // done by neighbor_fast_kernel
    const int cbase = (qi / hdr->n_per_graph) * hdr->ncells_local;
    const int dx = hdr->dims[0], dy = hdr->dims[1], dz = hdr->dims[2];
    const double inv_h = hdr->inv_h;
    const int cx = cell_coord(q.x, hdr->origin[0], inv_h, dx);
    const int cy = cell_coord(q.y, hdr->origin[1], inv_h, dy);
    const int cz = cell_coord(q.z, hdr->origin[2], inv_h, dz);
    const double qx = (double)q.x, qy = (double)q.y, qz = (double)q.z;
    int kept = 0;
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, dx - 1);
    for (int z = max(cz - 1, 0); z <= min(cz + 1, dz - 1); ++z) {
        for (int y = max(cy - 1, 0); y <= min(cy + 1, dy - 1); ++y) {
            const int row = cbase + (z * dy + y) * dx;
            const int b = cell_start[row + x0], e = cell_start[row + x1 + 1];
            for (int c = b; c < e; ++c) {
                const float4 p = sorted[c];
                // float64, accumulated x -> y -> z (utils.py:76-78 via sklearn rdist); every
                // product of float32-origin differences is exact in float64.
                double d = qx - (double)p.x;
                double d2 = __dmul_rn(d, d);
                d = qy - (double)p.y;
                d2 = __dadd_rn(d2, __dmul_rn(d, d));
                d = qz - (double)p.z;
                d2 = __dadd_rn(d2, __dmul_rn(d, d));
                if (!(d2 <= r2)) continue;
                const int j = __float_as_int(p.w);
                int pos;
                if (kept < K) {
                    pos = kept++;
                } else {
                    const double ld = kd[(K - 1) * BS + t];
                    const int lj = ki[(K - 1) * BS + t];
                    if (!(d2 < ld || (d2 == ld && j < lj))) continue;
                    pos = K - 1;
                }
                while (pos > 0) {
                    const double pd = kd[(pos - 1) * BS + t];
                    const int pj = ki[(pos - 1) * BS + t];
                    if (!(d2 < pd || (d2 == pd && j < pj))) break;
                    kd[pos * BS + t] = pd;
                    ki[pos * BS + t] = pj;
                    --pos;
                }
                kd[pos * BS + t] = d2;
                ki[pos * BS + t] = j;
            }
        }
    }
    cnt[qi] = kept;
    for (int s = 0; s < kept; ++s) {
        const int j = ki[s * BS + t];
        nbr[(int64_t)qi * K + s] = j;
        if (indeg) arrival[(int64_t)qi * K + s] = atomicAdd(&indeg[flow ? qi : j], 1);
    }
}

__global__ void __launch_bounds__(256) emit_edges_kernel(const int* __restrict__ cnt, const int* __restrict__ out_ptr,
                                                          const int* __restrict__ nbr, int64_t n, int K,
                                                          int64_t capacity, int64_t* __restrict__ senders,
                                                          int64_t* __restrict__ receivers) {
    int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n * K) return;
    int64_t i = id / K;
    int s = (int)(id - i * K);
    if (s >= cnt[i]) return;
    int64_t e = (int64_t)out_ptr[i] + s;
    if (e >= capacity) return;
    senders[e] = i;
    receivers[e] = nbr[id];
}

// ------------------------------------------------------------------------------------------
// destination-sorted structure
// ------------------------------------------------------------------------------------------
// mode 0: slots of the radius graph (source = query i, destination = neighbour)
// flow 0: the aggregation node of edge (query i -> neighbour) is the neighbour (edge_index[1]); 1: the query (edge_index[0])
// The counting pass hands every edge its arrival number within its destination's segment (`slot`), so the fill pass places the
// edge with plain stores: one atomic per edge instead of two.  (The arrival order is arbitrary; segment_sort_kernel fixes the order.)
__global__ void __launch_bounds__(256) indeg_graph_kernel(const int* __restrict__ cnt, const int* __restrict__ nbr,
                                                           int64_t n, int K, int flow, int* __restrict__ indeg, int* __restrict__ slot) {
    int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n * K) return;
    int64_t i = id / K;
    if ((int)(id - i * K) >= cnt[i]) return;
    slot[id] = atomicAdd(&indeg[flow ? (int)i : nbr[id]], 1);
}

// One thread of the launch also lays out the 32-edge block tables' plan (per-graph block counts: it needs in_ptr only, which the scan
// in front has finished) -- the tables themselves are filled by the launch that follows (segment_sort_kernel).
__global__ void __launch_bounds__(256) fill_graph_kernel(const int* __restrict__ cnt, const int* __restrict__ out_ptr,
                                                          const int* __restrict__ nbr, int64_t n, int K,
                                                          const int* __restrict__ in_ptr, const int* __restrict__ slot,
                                                          int64_t cap, int flow, int* __restrict__ dst, int* __restrict__ src,
                                                          int* __restrict__ eid, CsrHeader* hdr, const int* n_per_dev,
                                                          EdgeBlockHeader* tab, int* gblk) {
    int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id == 0 && tab) edge_blocks_plan(in_ptr, (int)n, n_per_dev, 0, tab, gblk, (int)n + 1);
    if (id >= n * K) return;
    int64_t i = id / K;
    int s = (int)(id - i * K);
    if (s >= cnt[i]) return;
    const int d = flow ? (int)i : nbr[id];
    int p = in_ptr[d] + slot[id];
    if (p >= cap) { atomicOr(&hdr->error_flags, ERRF_CAPACITY); return; }
    dst[p] = d;
    src[p] = flow ? nbr[id] : (int)i;
    eid[p] = out_ptr[i] + s;
}

// mode 1: caller-supplied edge_index [2, E] (int64)
__global__ void __launch_bounds__(256) indeg_ei_kernel(const int64_t* __restrict__ ei, int64_t n, int64_t e, int flow,
                                                        int* __restrict__ indeg, CsrHeader* hdr) {
    int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= e) return;
    int64_t s = ei[flow ? e + id : id], d = ei[flow ? id : e + id];
    if (s < 0 || s >= n || d < 0 || d >= n) { atomicOr(&hdr->error_flags, ERRF_BAD_EDGE_INDEX); return; }
    atomicAdd(&indeg[d], 1);
}

__global__ void __launch_bounds__(256) fill_ei_kernel(const int64_t* __restrict__ ei, int64_t n, int64_t e, int flow,
                                                       const int* __restrict__ in_ptr, int* __restrict__ cursor,
                                                       int* __restrict__ dst, int* __restrict__ src,
                                                       int* __restrict__ eid, CsrHeader* hdr) {
    int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= e) return;
    int64_t s = ei[flow ? e + id : id], d = ei[flow ? id : e + id];
    if (s < 0 || s >= n || d < 0 || d >= n) {
        // An edge with an index out of range is left out (flagged by the counting pass).  Its place is one of the slots past the
        // E valid edges: it gets VALID indices there (node 0, its own edge id), so that a caller that walks all `e` slots with its
        // host-side count -- the training kernels -- stays inside every array; what it computes for a flagged forward is not used.
        const int p = (int)e - 1 - atomicAdd(&hdr->pad, 1);
        dst[p] = 0;
        src[p] = 0;
        eid[p] = (int)id;
        return;
    }
    int p = in_ptr[d] + atomicAdd(&cursor[d], 1);
    dst[p] = (int)d;
    src[p] = (int)s;
    eid[p] = (int)id;
}

// Make the order inside each destination segment deterministic: ascending original edge id.
// 8 lanes per segment: the segment is staged in LDS, every element is ranked against the others
// (ids are unique) and written back in place.  Segments longer than SEG_CAP (in-degree > 96) are rank-sorted by the
// whole workgroup through a copy in the workspace.
#ifndef SEG_LPS
#define SEG_LPS 8                  // lanes per segment (A/B builds)
#endif
constexpr int SEG_PER_WG = 256 / SEG_LPS;   // segments per workgroup
constexpr int SEG_CAP = 96;
constexpr int SEG_STRIDE = SEG_CAP + 1;
// The first n_tab workgroups of the grid fill the 32-edge block tables of the same structure (they read in_ptr and dst, which the
// sort does not touch; the plan was laid out by the launch in front; a thread walks a block's 32 destinations serially, so these
// workgroups go FIRST and the sort's run beside them), the other cdiv(n, 32) sort.  n_tab == 0: no tables (csr_from_edge_index).
struct BlockTabArgs {
    const int* dst;
    EdgeBlockHeader* hdr;
    const int* gblk;
    int2* blk;
    int2* seg;
    int* head;
    int* stitch;
    int* stitch_list;
    int n_tab;   // workgroups of the launch that fill the tables
};
__global__ void __launch_bounds__(256) segment_sort_kernel(const int* __restrict__ in_ptr, int64_t n,
                                                            int* __restrict__ src, int* __restrict__ eid,
                                                            CsrHeader* hdr, const float* __restrict__ pos,
                                                            int64_t pos_stride, float conn_r, float* __restrict__ edge_attr, int flow,
                                                            int* __restrict__ tmp_eid, int* __restrict__ tmp_src, BlockTabArgs bt) {
    __shared__ int se[SEG_PER_WG * SEG_STRIDE];
    __shared__ int ss[SEG_PER_WG * SEG_STRIDE];
    __shared__ int s_long[SEG_PER_WG];
    if ((int)blockIdx.x < bt.n_tab) {
        edge_blocks_fill(in_ptr, bt.dst, (int)n, bt.hdr, bt.gblk, bt.blk, bt.seg, bt.head, bt.stitch, bt.stitch_list,
                         (int)blockIdx.x * 256 + (int)threadIdx.x, bt.n_tab * 256);
        return;
    }
    const int wg = (int)blockIdx.x - bt.n_tab;   // this workgroup's SEG_PER_WG segments
    const int tid = threadIdx.x, sub = tid & (SEG_LPS - 1), sl = tid / SEG_LPS;
    const int64_t i = (int64_t)wg * SEG_PER_WG + sl;
    if (i == 0 && sub == 0) hdr->n_edges = in_ptr[n];
    const bool active = i < n;
    const int b = active ? in_ptr[i] : 0, e = active ? in_ptr[i + 1] : 0;
    const int len = e - b;
    int* le = se + sl * SEG_STRIDE;
    int* ls = ss + sl * SEG_STRIDE;
    const bool small = len <= SEG_CAP;
    if (small)
        for (int a = sub; a < len; a += SEG_LPS) {
            le[a] = eid[b + a];
            ls[a] = src[b + a];
        }
    if (sub == 0) s_long[sl] = (active && !small) ? 1 : 0;
    __syncthreads();
    auto write_attr = [&](int64_t node, int srcv, int64_t at) {  // [(p_s - p_r)/r, |.|] of the edge at sorted position `at` (utils.py:43-61)
        float d[3];
#pragma unroll
        for (int c = 0; c < 3; ++c)
            d[c] = flow ? __fdiv_rn(__fsub_rn(pos[node * pos_stride + c], pos[(int64_t)srcv * pos_stride + c]), conn_r)
                        : __fdiv_rn(__fsub_rn(pos[(int64_t)srcv * pos_stride + c], pos[node * pos_stride + c]), conn_r);
        float q = __fmul_rn(d[0], d[0]);
        q = __fadd_rn(q, __fmul_rn(d[1], d[1]));
        q = __fadd_rn(q, __fmul_rn(d[2], d[2]));
        *reinterpret_cast<float4*>(edge_attr + at * 4) = make_float4(d[0], d[1], d[2], __fsqrt_rn(q));
    };
    if (active && small) {
        for (int a = sub; a < len; a += SEG_LPS) {
            const int ka = le[a];
            int rank = 0;
            for (int f = 0; f < len; ++f) rank += le[f] < ka ? 1 : 0;
            eid[b + rank] = ka;
            src[b + rank] = ls[a];
            if (edge_attr) write_attr(i, ls[a], (int64_t)b + rank);
        }
    }
    // long segments (hub nodes of a caller's edge_index): the whole workgroup rank-sorts one at a time through a copy in
    // the workspace, len^2 / 256 comparisons per thread instead of a single lane's insertion sort
    for (int q = 0; q < SEG_PER_WG; ++q) {
        if (!s_long[q]) continue;   // uniform
        const int64_t node = (int64_t)wg * SEG_PER_WG + q;
        const int lb = in_ptr[node], ll = in_ptr[node + 1] - lb;
        for (int a = tid; a < ll; a += 256) { tmp_eid[lb + a] = eid[lb + a]; tmp_src[lb + a] = src[lb + a]; }
        __syncthreads();   // the copy is complete (this workgroup is the only writer of the segment)
        for (int a = tid; a < ll; a += 256) {
            const int ka = tmp_eid[lb + a];
            int rank = 0;
            for (int f = 0; f < ll; ++f) rank += tmp_eid[lb + f] < ka ? 1 : 0;
            eid[lb + rank] = ka;
            src[lb + rank] = tmp_src[lb + a];
            if (edge_attr) write_attr(node, tmp_src[lb + a], (int64_t)lb + rank);
        }
        __syncthreads();
    }
}

int cell_order(const float* pos, int64_t pos_stride, int64_t n, int64_t n_per, double conn_r, int K, void* graph_ws, size_t graph_ws_bytes,
               int* perm, hipStream_t s) {
    GM_REQUIRE(pos && graph_ws && perm && n > 0 && pos_stride >= 3 && conn_r > 0.0, GM_ERR_INVALID_ARGUMENT, "cell_order: bad argument");
    GraphWs g = carve_graph(graph_ws, n, K);
    GM_REQUIRE(graph_ws_bytes >= g.bytes, GM_ERR_WORKSPACE, "cell_order: workspace %zu < %zu", graph_ws_bytes, g.bytes);
    const int nb = (int)cdiv(n, 256);
    StepClear clr;
    graph_clear_jobs(clr, g, n);
    int rc = launch_step_clear(clr, s);
    if (rc != GM_OK) return rc;
    hipLaunchKernelGGL(bbox_kernel, dim3(nb < 64 ? nb : 64), dim3(256), 0, s, pos, pos_stride, n, g.hdr, conn_r, g.max_cells, n_per);
    hipLaunchKernelGGL(cell_assign_kernel, dim3(nb), dim3(256), 0, s, pos, pos_stride, n, g.hdr, g.cell_of, g.cell_start);
    const ScanJob sj{g.cell_start, g.cell_start, (long long)g.max_cells + 1, nullptr, g.scan_cells};
    rc = scan_lookback(&sj, 1, s);
    if (rc != GM_OK) return rc;
    hipLaunchKernelGGL(cell_fill_kernel, dim3(nb), dim3(256), 0, s, pos, pos_stride, n, g.cell_of, g.cell_start, g.cell_cursor, g.sorted);
    hipLaunchKernelGGL(cell_order_kernel, dim3(nb), dim3(256), 0, s, g.sorted, g.cell_of, g.cell_start, g.hdr, n, perm);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

}  // namespace gm

using namespace gm;

extern "C" {

const char* gm_last_error(void) { return gm::last_error(); }

int gm_abi_version(void) { return 7; }

size_t gm_graph_workspace_bytes(int64_t n_nodes, int max_neighbours) {
    if (n_nodes < 0 || max_neighbours < 1) return 0;
    return carve_graph(nullptr, n_nodes, max_neighbours).bytes;
}

int gm_radius_graph_build(const float* pos, int64_t pos_stride, int64_t n, double conn_r, int K, void* ws,
                          size_t ws_bytes, void* stream) {
    return gm_radius_graph_build_batched(pos, pos_stride, n, n, conn_r, K, ws, ws_bytes, stream);
}

}  // extern "C"

namespace {
int check_build_args(const float* pos, int64_t pos_stride, int64_t n, int64_t n_per, double conn_r, int K, const void* ws) {
    GM_REQUIRE(n_per >= 0 && (n_per == 0 || n % n_per == 0 || n_per >= n), GM_ERR_INVALID_ARGUMENT,
               "gm_radius_graph_build_batched: n_nodes=%lld is not a multiple of nodes_per_graph=%lld", (long long)n, (long long)n_per);
    GM_REQUIRE(n >= 0 && n < (int64_t)1 << 30, GM_ERR_INVALID_ARGUMENT, "gm_radius_graph_build: n_nodes=%lld out of range", (long long)n);
    GM_REQUIRE(K >= 1 && K <= 160, GM_ERR_UNSUPPORTED, "gm_radius_graph_build: max_neighbours=%d unsupported (1..160)", K);
    GM_REQUIRE(n * (int64_t)K < (int64_t)1 << 31, GM_ERR_UNSUPPORTED, "gm_radius_graph_build: n*max_neighbours overflows int32");
    GM_REQUIRE(conn_r > 0.0 && conn_r == conn_r, GM_ERR_INVALID_ARGUMENT, "gm_radius_graph_build: conn_r must be > 0");
    GM_REQUIRE(pos_stride >= 3, GM_ERR_INVALID_ARGUMENT, "gm_radius_graph_build: pos_stride < 3");
    GM_REQUIRE(ws != nullptr && (pos != nullptr || n == 0), GM_ERR_INVALID_ARGUMENT, "gm_radius_graph_build: null pointer");
    return GM_OK;
}

// The build behind its resets (StepClear: a launch of its own in the stand-alone entry point, part of the step's first launch in the
// rollout): bounding box + grid, cell assignment, ONE-launch scan of the cell counts, cell fill, the two neighbour kernels.  The scan
// of cnt (-> out_ptr, E) is the caller's: the rollout path runs it together with the in-degree scan of the destination sort.
// indeg / slot: the destination sort's counting pass rides in the neighbour kernels (nullptr: it does not).
int radius_graph_build_core(const float* pos, int64_t pos_stride, int64_t n, int64_t n_per, double conn_r, int K, const GraphWs& g,
                            int* indeg, int* slot, int flow, hipStream_t s) {
    if (n <= 0) return GM_OK;
    int nb = (int)cdiv(n, 256);
    hipLaunchKernelGGL(bbox_kernel, dim3(nb < 64 ? nb : 64), dim3(256), 0, s, pos, pos_stride, n, g.hdr, conn_r,
                       g.max_cells, n_per);
    hipLaunchKernelGGL(cell_assign_kernel, dim3(nb), dim3(256), 0, s, pos, pos_stride, n, g.hdr, g.cell_of, g.cell_start);
    const ScanJob sj{g.cell_start, g.cell_start, (long long)g.max_cells + 1, nullptr, g.scan_cells};
    int rc = scan_lookback(&sj, 1, s);
    if (rc != GM_OK) return rc;
    hipLaunchKernelGGL(cell_fill_kernel, dim3(nb), dim3(256), 0, s, pos, pos_stride, n, g.cell_of, g.cell_start,
                       g.cell_cursor, g.sorted);
    const double r2 = conn_r * conn_r;  // KDTree compares rdist with r*r in float64
    hipLaunchKernelGGL(neighbor_fast_kernel, dim3((unsigned)cdiv(n, NB_QPB)), dim3(256), 0, s, g.sorted, g.cell_start,
                       g.hdr, n, r2, K, g.cnt, g.nbr, indeg, slot, flow);
    if (K <= 64) {
        constexpr int BS = 128;
        size_t lds = (size_t)K * BS * 12;
        if (lds > 64 * 1024) {
            static PerDeviceOnce big_lds;
            const int rc_attr = big_lds.run([]() -> int {
                GM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(neighbor_kernel<BS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                return GM_OK;
            });
            if (rc_attr != GM_OK) return rc_attr;
        }
        hipLaunchKernelGGL(neighbor_kernel<BS>, dim3((int)cdiv(n, BS)), dim3(BS), lds, s, g.sorted, g.cell_start,
                           g.hdr, n, r2, K, g.cnt, g.nbr, indeg, slot, flow);
    } else {
        constexpr int BS = 64;
        size_t lds = (size_t)K * BS * 12;
        GM_REQUIRE(lds <= 160 * 1024, GM_ERR_UNSUPPORTED, "gm_radius_graph_build: max_neighbours=%d needs %zu bytes of LDS", K, lds);
        if (lds > 64 * 1024) {
            static PerDeviceOnce big_lds64;
            const int rc_attr = big_lds64.run([]() -> int {
                GM_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(neighbor_kernel<BS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                return GM_OK;
            });
            if (rc_attr != GM_OK) return rc_attr;
        }
        hipLaunchKernelGGL(neighbor_kernel<BS>, dim3((int)cdiv(n, BS)), dim3(BS), lds, s, g.sorted, g.cell_start,
                           g.hdr, n, r2, K, g.cnt, g.nbr, indeg, slot, flow);
    }
    GM_LAUNCH_CHECK();
    return GM_OK;
}
}  // namespace

extern "C" {

int gm_radius_graph_build_batched(const float* pos, int64_t pos_stride, int64_t n, int64_t n_per, double conn_r, int K,
                                  void* ws, size_t ws_bytes, void* stream) {
    gm::DevGuard dev_guard(pos);
    int rc = check_build_args(pos, pos_stride, n, n_per, conn_r, K, ws);
    if (rc != GM_OK) return rc;
    GraphWs g = carve_graph(ws, n, K);
    GM_REQUIRE(ws_bytes >= g.bytes, GM_ERR_WORKSPACE, "gm_radius_graph_build: workspace %zu < %zu", ws_bytes, g.bytes);
    hipStream_t s = (hipStream_t)stream;
    StepClear clr;
    graph_clear_jobs(clr, g, n);
    rc = launch_step_clear(clr, s);
    if (rc == GM_OK) rc = radius_graph_build_core(pos, pos_stride, n, n_per, conn_r, K, g, nullptr, nullptr, 0, s);
    if (rc != GM_OK) return rc;
    const ScanJob sj{g.cnt, g.out_ptr, (long long)n + 1, &g.hdr->n_edges, g.scan_cnt};
    return scan_lookback(&sj, 1, s);
}

int gm_radius_graph_num_edges(const void* ws, int64_t* n_edges_host, void* stream) {
    gm::DevGuard dev_guard(ws);
    GM_REQUIRE(ws && n_edges_host, GM_ERR_INVALID_ARGUMENT, "gm_radius_graph_num_edges: null pointer");
    GraphHeader h;
    GM_HIP_CHECK(hipMemcpyAsync(&h, ws, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream));
    GM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    *n_edges_host = h.n_edges;
    GM_REQUIRE(!(h.error_flags & ERRF_NONFINITE_POS), GM_ERR_DATA, "radius graph: non-finite position in input");
    return GM_OK;
}

int gm_radius_graph_edges(const void* ws, int64_t n, int K, int64_t* senders, int64_t* receivers, int64_t capacity,
                          void* stream) {
    gm::DevGuard dev_guard(ws);
    GM_REQUIRE(ws && (capacity == 0 || (senders && receivers)), GM_ERR_INVALID_ARGUMENT, "gm_radius_graph_edges: null pointer");
    if (n == 0 || capacity == 0) return GM_OK;
    GraphWs g = carve_graph(const_cast<void*>(ws), n, K);
    hipLaunchKernelGGL(emit_edges_kernel, dim3((unsigned)cdiv(n * K, 256)), dim3(256), 0, (hipStream_t)stream, g.cnt,
                       g.out_ptr, g.nbr, n, K, capacity, senders, receivers);
    GM_LAUNCH_CHECK();
    return GM_OK;
}

size_t gm_csr_workspace_bytes(int64_t n_nodes, int64_t edge_capacity) {
    if (n_nodes < 0 || edge_capacity < 0) return 0;
    return carve_csr(nullptr, n_nodes, edge_capacity).bytes;
}

int gm_csr_from_graph(const void* graph_ws, int64_t n, int K, void* csr_ws, size_t csr_ws_bytes, void* stream) {
    return gm_csr_from_graph_flow(graph_ws, n, K, 0, csr_ws, csr_ws_bytes, stream);
}

int gm_csr_from_graph_flow(const void* graph_ws, int64_t n, int K, int flow, void* csr_ws, size_t csr_ws_bytes, void* stream) {
    gm::DevGuard dev_guard(graph_ws);
    return gm::csr_from_graph_with_features(graph_ws, n, K, csr_ws, csr_ws_bytes, nullptr, 3, 1.f, nullptr, flow, (hipStream_t)stream);
}

}  // extern "C"

namespace gm {
namespace {
// scan(s) -> fill (+ block-table plan) -> segment sort + edge features + block tables.  with_cnt_scan: the radius graph's own scan
// of cnt (-> out_ptr, E) shares the scan launch (the fused rollout path; the stand-alone build has already run it).
int csr_tail(const GraphWs& g, int64_t n, int K, const CsrWs& c, const float* pos, int64_t pos_stride, float conn_r, float* edge_attr,
             int flow, bool with_cnt_scan, hipStream_t s) {
    const int64_t cap = n * K;
    const unsigned nb = (unsigned)cdiv(cap, 256);
    int* slot = c.sort_tmp;   // [cap]: free until segment_sort_kernel (which runs after the fill) needs it for long segments
    ScanJob sj[2] = {{c.in_ptr, c.in_ptr, (long long)n + 1, nullptr, c.scan_in},
                     {g.cnt, g.out_ptr, (long long)n + 1, &g.hdr->n_edges, g.scan_cnt}};
    int rc = scan_lookback(sj, with_cnt_scan ? 2 : 1, s);
    if (rc != GM_OK) return rc;
    const EdgeBlocks t = carve_edge_blocks(c.blocks, n, cap);
    hipLaunchKernelGGL(fill_graph_kernel, dim3(nb), dim3(256), 0, s, g.cnt, g.out_ptr, g.nbr, n, K, c.in_ptr,
                       slot, cap, flow, c.dst, c.src, c.eid, c.hdr, &g.hdr->n_per_graph, t.hdr, t.gblk);
    // 32-edge blocks aligned to every graph's first edge (the systolic processor edge kernel walks them): extra workgroups of the sort
    // a thread per block of the list's capacity (cap / 32 + the per-graph padding blocks), at most 512 workgroups: they loop
    int gb = (int)cdiv(cdiv(cap, kBlockEdges) + 8, 256);
    gb = gb < 1 ? 1 : (gb > 512 ? 512 : gb);
    const BlockTabArgs bt{c.dst, t.hdr, t.gblk, t.blk, t.seg, t.head, t.stitch, t.stitch_list, gb};
    hipLaunchKernelGGL(segment_sort_kernel, dim3((unsigned)cdiv(n, SEG_PER_WG) + (unsigned)gb), dim3(256), 0, s, c.in_ptr, n, c.src, c.eid, c.hdr,
                       pos, pos_stride, conn_r, edge_attr, flow, c.sort_tmp, c.sort_tmp + c.cap, bt);
    GM_LAUNCH_CHECK();
    return GM_OK;
}
}  // namespace

int csr_from_graph_with_features(const void* graph_ws, int64_t n, int K, void* csr_ws, size_t csr_ws_bytes, const float* pos,
                                 int64_t pos_stride, float conn_r, float* edge_attr, int flow, hipStream_t stream) {
    GM_REQUIRE(graph_ws && csr_ws, GM_ERR_INVALID_ARGUMENT, "gm_csr_from_graph: null pointer");
    GM_REQUIRE(flow == 0 || flow == 1, GM_ERR_INVALID_ARGUMENT, "gm_csr_from_graph: flow must be 0 or 1");
    const int64_t cap = n * K;
    GraphWs g = carve_graph(const_cast<void*>(graph_ws), n, K);
    CsrWs c = carve_csr(csr_ws, n, cap);
    GM_REQUIRE(csr_ws_bytes >= c.bytes, GM_ERR_WORKSPACE, "gm_csr_from_graph: workspace %zu < %zu", csr_ws_bytes, c.bytes);
    hipStream_t s = (hipStream_t)stream;
    StepClear clr;
    csr_clear_jobs(clr, c, n, flow);
    int rc = launch_step_clear(clr, s);
    if (rc != GM_OK || n <= 0) return rc;
    hipLaunchKernelGGL(indeg_graph_kernel, dim3((unsigned)cdiv(cap, 256)), dim3(256), 0, s, g.cnt, g.nbr, n, K, flow, c.in_ptr, c.sort_tmp);
    return csr_tail(g, n, K, c, pos, pos_stride, conn_r, edge_attr, flow, false, s);
}

// The rollout step's graph path (resets done by the caller's StepClear, see graph_clear_jobs / csr_clear_jobs): 9 launches from
// positions to the destination-sorted structure with edge features and block tables.
int radius_graph_build_fused(const float* pos, int64_t pos_stride, int64_t n, int64_t n_per, double conn_r, int K, void* graph_ws,
                             size_t graph_ws_bytes, void* csr_ws, size_t csr_ws_bytes, int flow, hipStream_t s) {
    int rc = check_build_args(pos, pos_stride, n, n_per, conn_r, K, graph_ws);
    if (rc != GM_OK) return rc;
    GM_REQUIRE(csr_ws && (flow == 0 || flow == 1), GM_ERR_INVALID_ARGUMENT, "radius_graph_build_fused: bad csr workspace / flow");
    GraphWs g = carve_graph(graph_ws, n, K);
    CsrWs c = carve_csr(csr_ws, n, n * K);
    GM_REQUIRE(graph_ws_bytes >= g.bytes && csr_ws_bytes >= c.bytes, GM_ERR_WORKSPACE, "radius_graph_build_fused: workspace too small");
    return radius_graph_build_core(pos, pos_stride, n, n_per, conn_r, K, g, c.in_ptr, c.sort_tmp, flow, s);
}
int csr_from_graph_fused(const void* graph_ws, int64_t n, int K, void* csr_ws, size_t csr_ws_bytes, const float* pos, int64_t pos_stride,
                         float conn_r, float* edge_attr, int flow, hipStream_t s) {
    if (n <= 0) return GM_OK;
    GraphWs g = carve_graph(const_cast<void*>(graph_ws), n, K);
    CsrWs c = carve_csr(csr_ws, n, n * K);
    GM_REQUIRE(csr_ws_bytes >= c.bytes, GM_ERR_WORKSPACE, "csr_from_graph_fused: workspace %zu < %zu", csr_ws_bytes, c.bytes);
    return csr_tail(g, n, K, c, pos, pos_stride, conn_r, edge_attr, flow, true, s);
}
}  // namespace gm

extern "C" {

int gm_csr_from_edge_index(const int64_t* ei, int64_t n, int64_t e, void* csr_ws, size_t csr_ws_bytes, void* stream) {
    return gm_csr_from_edge_index_flow(ei, n, e, 0, csr_ws, csr_ws_bytes, stream);
}

int gm_csr_from_edge_index_flow(const int64_t* ei, int64_t n, int64_t e, int flow, void* csr_ws, size_t csr_ws_bytes, void* stream) {
    return gm::csr_from_edge_index(ei, n, e, flow, csr_ws, csr_ws_bytes, true, (hipStream_t)stream);
}

}  // extern "C"

namespace gm {
// with_blocks = false: the sorted lists and in_ptr only, without the 32-edge block tables of the inference edge kernels (the training
// kernels walk rows: five launches fewer per structure)
int csr_from_edge_index(const int64_t* ei, int64_t n, int64_t e, int flow, void* csr_ws, size_t csr_ws_bytes, bool with_blocks, hipStream_t stream) {
    gm::DevGuard dev_guard(csr_ws);
    GM_REQUIRE(flow == 0 || flow == 1, GM_ERR_INVALID_ARGUMENT, "gm_csr_from_edge_index: flow must be 0 or 1");
    GM_REQUIRE(csr_ws && (ei || e == 0), GM_ERR_INVALID_ARGUMENT, "gm_csr_from_edge_index: null pointer");
    GM_REQUIRE(n >= 0 && e >= 0 && e < ((int64_t)1 << 31) && n < ((int64_t)1 << 31), GM_ERR_INVALID_ARGUMENT,
               "gm_csr_from_edge_index: sizes out of range");
    CsrWs c = carve_csr(csr_ws, n, e);
    GM_REQUIRE(csr_ws_bytes >= c.bytes, GM_ERR_WORKSPACE, "gm_csr_from_edge_index: workspace %zu < %zu", csr_ws_bytes, c.bytes);
    hipStream_t s = (hipStream_t)stream;
    {
        StepClear clr;
        csr_clear_jobs(clr, c, n, flow);
        clr.add(c.cursor, n + 1, 0);
        const int rc = launch_step_clear(clr, s);
        if (rc != GM_OK) return rc;
    }
    if (e > 0) {
        unsigned nb = (unsigned)cdiv(e, 256);
        hipLaunchKernelGGL(indeg_ei_kernel, dim3(nb), dim3(256), 0, s, ei, n, e, flow, c.in_ptr, c.hdr);
        const ScanJob sj{c.in_ptr, c.in_ptr, (long long)n + 1, nullptr, c.scan_in};
        int rc = scan_lookback(&sj, 1, s);
        if (rc != GM_OK) return rc;
        hipLaunchKernelGGL(fill_ei_kernel, dim3(nb), dim3(256), 0, s, ei, n, e, flow, c.in_ptr, c.cursor, c.dst, c.src, c.eid, c.hdr);
    }
    if (n > 0)
        hipLaunchKernelGGL(segment_sort_kernel, dim3((unsigned)cdiv(n, SEG_PER_WG)), dim3(256), 0, s, c.in_ptr, n, c.src, c.eid, c.hdr, nullptr, 3, 1.f, nullptr, 0, c.sort_tmp, c.sort_tmp + c.cap, BlockTabArgs{});
    GM_LAUNCH_CHECK();
    if (!with_blocks) return GM_OK;
    return build_edge_blocks(c.in_ptr, c.dst, n, e, nullptr, (int)n, carve_edge_blocks(c.blocks, n, e), s);
}
}  // namespace gm

extern "C" {

int gm_csr_header_status(const int32_t* header_host, int64_t* n_edges_host) {
    GM_REQUIRE(header_host, GM_ERR_INVALID_ARGUMENT, "gm_csr_header_status: null pointer");
    CsrHeader h;
    memcpy(&h, header_host, sizeof(h));
    if (n_edges_host) *n_edges_host = h.n_edges;
    GM_REQUIRE(!(h.error_flags & ERRF_BAD_EDGE_INDEX), GM_ERR_DATA, "edge_index entry out of range [0, n_nodes)");
    GM_REQUIRE(!(h.error_flags & ERRF_CAPACITY), GM_ERR_DATA, "edge capacity exceeded");
    GM_REQUIRE(!(h.error_flags & ERRF_SPLIT_RANGE), GM_ERR_DATA,
               "fp16 split range exceeded: a feature, latent or hidden activation of the last forward over this edge structure reached "
               "|x| >= 65504 in a matrix-pipe operand image (include/gnn_manip_hip.h, numeric domain); its results are not valid");
    return GM_OK;
}

int gm_csr_num_edges(const void* csr_ws, int64_t* n_edges_host, void* stream) {
    gm::DevGuard dev_guard(csr_ws);
    GM_REQUIRE(csr_ws && n_edges_host, GM_ERR_INVALID_ARGUMENT, "gm_csr_num_edges: null pointer");
    CsrHeader h;
    GM_HIP_CHECK(hipMemcpyAsync(&h, csr_ws, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream));
    GM_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
    return gm_csr_header_status(reinterpret_cast<const int32_t*>(&h), n_edges_host);
}

}  // extern "C"
