// fnn_hip.hip -- gfx950 kernels and the C ABI of libfastnn_hip.so.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared
// (-ffp-contract=off on host AND device: every fp64 operation of the path is
// rounded once, in the reference's source order; no FMA anywhere).
//
// Kernel inventory (one agglomeration event = the fixed launch sequence of
// HipBackend::launch_event; all control state lives in device memory so the host
// never has to wait for a decision):
//   k_scan      all-pairs Q-criterion argmin over the lower triangle of the live
//               m x m block (NeighborNetCanonical.java:151-178).  HBM-bound: reads
//               each live matrix entry once, 16 B per lane, 1 KiB per wave-load.
//   k_pick      reduce the per-block records, form Cx/Cy (NetMakerOriginal.java:376-380)
//   k_rx_fill   ComputeRx terms in reference position order (:549-561)
//   k_decide    the <=4 sequential Rx sums + candidate choice + merge plan (:413-488)
//   k_subtract  subtractClusterDistance x2 per node (:455-461, 681-696)
//   k_op        one micro-op of the plan: agg3way row/column rewrite (:653-656),
//               slot swap or slot move (layout maintenance)
//   k_add       updateClusterDistances per-node part (:520-531)
//   k_finalize  sequential u.Sx sum (:532), event log, loop condition (:339)
#include <hip/hip_runtime.h>

#include <cmath>
#include <new>

#include "fnn_engine.h"

namespace fnn {

constexpr int SCAN_TW = 512;  // columns per scan tile (256 threads x 2)
constexpr int SCAN_TH = 32;   // rows per scan tile
constexpr int SCAN_THREADS = 256;

// ------------------------------------------------------------------ reductions
__device__ __forceinline__ Cand wave_reduce(Cand c) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        Cand o;
        o.q = __shfl_down(c.q, off, 64);
        o.key = (uint64_t)__shfl_down((unsigned long long)c.key, off, 64);
        if (cand_better(o, c)) c = o;
    }
    return c;
}

template <int NWAVES>
__device__ __forceinline__ Cand block_reduce(Cand c, Cand* sh) {
    c = wave_reduce(c);
    int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) sh[w] = c;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 1; k < NWAVES; k++)
            if (cand_better(sh[k], c)) c = sh[k];
    }
    return c;  // valid in thread 0
}

// ------------------------------------------------------------------ k_scan
__global__ __launch_bounds__(SCAN_THREADS) void k_scan(Dev d) {
    __shared__ Cand sh[SCAN_THREADS / 64];
    const State* st = d.st;
    Cand best;
    best.q = inf_f64();
    best.key = ~0ULL;
    const int rt = blockIdx.y, ct = blockIdx.x;
    const int m = st->m;
    const int rbase = rt * SCAN_TH;
    if (!st->done && rbase < m && ct * SCAN_TW <= rbase + SCAN_TH - 1) {
        const int twoP = 2 * st->P;
        const double cm2 = (double)st->c - 2.0;
        const int c0 = ct * SCAN_TW + 2 * (int)threadIdx.x;
        if (c0 < m && c0 <= rbase + SCAN_TH - 2) {
            const double2 sxc = *reinterpret_cast<const double2*>(d.Sx + c0);
            const int2 pc = *reinterpret_cast<const int2*>(d.spos + c0);
            const double* colbase = d.D + c0;
#pragma unroll 1
            for (int half = 0; half < SCAN_TH / 16; half++) {
                const int rb = rbase + 16 * half;
                if (rb >= m) break;
                double2 a[8], b[8];
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int r0 = rb + 2 * k;  // r0 + 1 < nrows (padded), c0 + 1 < ld: always in bounds
                    a[k] = *reinterpret_cast<const double2*>(colbase + (int64_t)r0 * d.ld);
                    b[k] = *reinterpret_cast<const double2*>(colbase + (int64_t)(r0 + 1) * d.ld);
                }
#pragma unroll
                for (int k = 0; k < 8; k++) {
                    const int r0 = rb + 2 * k;
                    if (c0 <= r0) {
                        const double2 sxr = *reinterpret_cast<const double2*>(d.Sx + r0);
                        const int2 pr = *reinterpret_cast<const int2*>(d.spos + r0);
                        scan_micro(r0, c0, m, twoP, cm2, a[k].x, a[k].y, b[k].x, b[k].y,
                                   sxr.x, sxr.y, pr.x, pr.y, sxc.x, sxc.y, pc.x, pc.y, best);
                    }
                }
            }
        }
    }
    best = block_reduce<SCAN_THREADS / 64>(best, sh);
    if (threadIdx.x == 0) d.recs[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = best;
}

// ------------------------------------------------------------------ k_pick
__global__ __launch_bounds__(1024) void k_pick(Dev d, int nrecs) {
    __shared__ Cand sh[16];
    Cand best;
    best.q = inf_f64();
    best.key = ~0ULL;
    if (!d.st->done) {
        for (int i = threadIdx.x; i < nrecs; i += 1024) {
            Cand c = d.recs[i];
            if (cand_better(c, best)) best = c;
        }
    }
    best = block_reduce<16>(best, sh);
    if (threadIdx.x == 0) pick(d, best);
}

// ------------------------------------------------------------------ k_rx_fill
__global__ __launch_bounds__(256) void k_rx_fill(Dev d) {
    const State* st = d.st;
    if (!st->ev_active || st->ev_finish || !st->need_rx) return;
    rx_fill_thread(d, blockIdx.x * 256 + threadIdx.x);
}

// Sequential sum of buf[0..m) in index order, exactly as a scalar loop would do it.
// One wave: the lanes stage 512 values at a time in LDS, then every lane runs the
// same dependent add chain over the staged values (LDS broadcast reads).
constexpr int CHAIN_CHUNK = 512;
__device__ __forceinline__ double wave_chain_sum(const double* buf, int m, double* lds, bool active) {
    double s = 0.0;
    const int lane = threadIdx.x & 63;
    for (int base = 0; base < m; base += CHAIN_CHUNK) {
        __syncthreads();
        if (active) {
#pragma unroll
            for (int i = lane; i < CHAIN_CHUNK; i += 64) lds[i] = (base + i < m) ? buf[base + i] : 0.0;
        }
        __syncthreads();
        if (active) {
#pragma unroll 16
            for (int i = 0; i < CHAIN_CHUNK; i++) s += lds[i];  // + 0.0 padding changes no bit
        }
    }
    return s;
}

// ------------------------------------------------------------------ k_decide
__global__ __launch_bounds__(256) void k_decide(Dev d) {
    __shared__ double lds[4][CHAIN_CHUNK];
    __shared__ double rx[4];
    State* st = d.st;
    if (!st->ev_active || st->ev_finish) return;
    const int w = threadIdx.x >> 6;
    if (threadIdx.x < 4) rx[threadIdx.x] = 0.0;
    if (st->need_rx) {
        int z = (w == 0) ? st->sa : (w == 1) ? st->sap : (w == 2) ? st->sb : st->sbp;
        double s = wave_chain_sum(d.chain + (size_t)w * d.n, st->m_old, lds[w], z >= 0);
        __syncthreads();
        if ((threadIdx.x & 63) == 0 && z >= 0) rx[w] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double r[4] = {rx[0], rx[1], rx[2], rx[3]};
        decide(d, r);
    }
}

// ------------------------------------------------------------------ k_subtract / k_op / k_add
__global__ __launch_bounds__(256) void k_subtract(Dev d) {
    const State* st = d.st;
    if (!st->ev_active || st->ev_finish) return;
    subtract_thread(d, blockIdx.x * 256 + threadIdx.x);
}

__global__ __launch_bounds__(256) void k_op(Dev d, int idx) {
    const State* st = d.st;
    if (!st->ev_active || idx >= st->nops) return;
    Op op = st->ops[idx];
    op_thread(d, op, blockIdx.x * 256 + threadIdx.x);
}

__global__ __launch_bounds__(256) void k_add(Dev d) {
    const State* st = d.st;
    if (!st->ev_active || st->ev_finish) return;
    add_thread(d, blockIdx.x * 256 + threadIdx.x);
}

// ------------------------------------------------------------------ k_finalize
__global__ __launch_bounds__(64) void k_finalize(Dev d) {
    __shared__ double lds[CHAIN_CHUNK];
    State* st = d.st;
    if (!st->ev_active) return;
    double usx = 0.0;
    if (!st->ev_finish) usx = wave_chain_sum(d.chain, st->m, lds, true);
    if (threadIdx.x == 0) finalize(d, usx);
}

// ------------------------------------------------------------------ setup kernels
__global__ __launch_bounds__(256) void k_init(Dev d) { init_thread(d, blockIdx.x * 256 + threadIdx.x); }

__global__ __launch_bounds__(256) void k_synth(Dev d, uint64_t seed, int dist) {
    int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t r = blockIdx.y;
    if (c >= d.n) return;
    double v = 0.0;
    if (r < c) v = synth_entry(d.n, r, c, seed, dist);
    else if (c < r) v = synth_entry(d.n, c, r, seed, dist);
    d.D[r * d.ld + c] = v;
}

// symmetric (bitwise), finite, zero diagonal: 32x32 tiles, transposed partner through LDS
__global__ __launch_bounds__(256) void k_validate(Dev d, int* bad) {
    __shared__ uint64_t tile[32][33];
    const int bx = blockIdx.x, by = blockIdx.y;
    if (bx > by) return;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    const uint64_t* D = reinterpret_cast<const uint64_t*>(d.D);
    for (int k = ty; k < 32; k += 8) {
        int r = bx * 32 + k, c = by * 32 + tx;  // the transposed tile
        tile[k][tx] = (r < d.n && c < d.n) ? D[(int64_t)r * d.ld + c] : 0;
    }
    __syncthreads();
    int flag = 0;
    for (int k = ty; k < 32; k += 8) {
        int r = by * 32 + k, c = bx * 32 + tx;
        if (r < d.n && c < d.n) {
            uint64_t v = D[(int64_t)r * d.ld + c];
            uint64_t t = tile[tx][k];
            if (v != t) flag = 1;
            if (((v >> 52) & 0x7FF) == 0x7FF) flag = 1;
            if (r == c && v != 0) flag = 1;
        }
    }
    if (flag) atomicOr(bad, 1);
}

__global__ __launch_bounds__(256) void k_stream(const double2* p, int64_t n16, double* sink) {
    double acc = 0.0;
    int64_t stride = (int64_t)gridDim.x * 256 * 4;
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x); i + 3 * (int64_t)gridDim.x * 256 < n16; i += stride) {
        double2 a = p[i], b = p[i + (int64_t)gridDim.x * 256], c = p[i + 2 * (int64_t)gridDim.x * 256],
                e = p[i + 3 * (int64_t)gridDim.x * 256];
        acc += (a.x + a.y) + (b.x + b.y) + (c.x + c.y) + (e.x + e.y);
    }
    if (acc == 123.456) *sink = acc;  // keep the loads alive
}

// ------------------------------------------------------------------ backend
#define HIPOK(x) ((last = (x)) == hipSuccess)

struct HipBackend {
    static constexpr int64_t kRowPad = SCAN_TH;
    static constexpr int64_t kColPad = SCAN_TW;
    hipError_t last = hipSuccess;
    hipStream_t stream = nullptr;
    int device = 0;
    bool opened = false;
    // scan timing
    bool timing = false;
    std::vector<hipEvent_t> ev_pool;
    size_t ev_used = 0;
    double scan_ms = 0.0;
    int64_t scan_launches = 0;
    int* d_bad = nullptr;

    std::string err() const { return std::string(hipGetErrorString(last)); }

    int32_t open(int32_t dev) {
        int cnt = 0;
        if (!HIPOK(hipGetDeviceCount(&cnt)) || cnt <= 0)
            return fail(FNN_EHIP, "no HIP device available (" + err() + ")");
        if (dev < 0 || dev >= cnt) return fail(FNN_EINVAL, "device ordinal out of range");
        device = dev;
        if (!HIPOK(hipSetDevice(device))) return fail(FNN_EHIP, "hipSetDevice failed (" + err() + ")");
        if (!HIPOK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking)))
            return fail(FNN_EHIP, "hipStreamCreate failed (" + err() + ")");
        opened = true;
        return FNN_OK;
    }
    void close() {
        if (!opened) return;
        (void)hipSetDevice(device);
        for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
        ev_pool.clear();
        if (d_bad) (void)hipFree(d_bad);
        d_bad = nullptr;
        if (stream) (void)hipStreamDestroy(stream);
        stream = nullptr;
        opened = false;
    }
    void* alloc(size_t b) {
        void* p = nullptr;
        (void)hipSetDevice(device);
        if (!HIPOK(hipMalloc(&p, b ? b : 8))) return nullptr;
        return p;
    }
    void free(void* p) {
        if (p) (void)hipFree(p);
    }
    size_t max_records(int32_t n) {
        size_t nn = (size_t)(n > 0 ? n : 1);
        return ((nn + SCAN_TW - 1) / SCAN_TW) * ((nn + SCAN_TH - 1) / SCAN_TH);
    }
    int32_t memset(void* p, int v, size_t b) {
        return HIPOK(hipMemsetAsync(p, v, b, stream)) && HIPOK(hipStreamSynchronize(stream)) ? FNN_OK : FNN_EHIP;
    }
    int32_t h2d(void* d, const void* s, size_t b) {
        return HIPOK(hipMemcpyAsync(d, s, b, hipMemcpyHostToDevice, stream)) && HIPOK(hipStreamSynchronize(stream)) ? FNN_OK : FNN_EHIP;
    }
    int32_t d2h(void* d, const void* s, size_t b) {
        return HIPOK(hipMemcpyAsync(d, s, b, hipMemcpyDeviceToHost, stream)) && HIPOK(hipStreamSynchronize(stream)) ? FNN_OK : FNN_EHIP;
    }
    int32_t copy2d(double* d, int64_t ldd, const double* s, int64_t lds, int64_t w, int64_t h, hipMemcpyKind kind) {
        return HIPOK(hipMemcpy2DAsync(d, (size_t)ldd * 8, s, (size_t)lds * 8, (size_t)w * 8, (size_t)h, kind, stream)) &&
                       HIPOK(hipStreamSynchronize(stream))
                   ? FNN_OK
                   : FNN_EHIP;
    }
    int32_t h2d_2d(double* d, int64_t ldd, const double* s, int64_t lds, int64_t w, int64_t h) {
        return copy2d(d, ldd, s, lds, w, h, hipMemcpyHostToDevice);
    }
    int32_t d2d_2d(double* d, int64_t ldd, const double* s, int64_t lds, int64_t w, int64_t h) {
        return copy2d(d, ldd, s, lds, w, h, hipMemcpyDeviceToDevice);
    }
    int32_t d2h_2d(double* d, int64_t ldd, const double* s, int64_t lds, int64_t w, int64_t h) {
        return copy2d(d, ldd, s, lds, w, h, hipMemcpyDeviceToHost);
    }

    void drain_timing() {
        for (size_t i = 0; i + 1 < ev_used; i += 2) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, ev_pool[i], ev_pool[i + 1]) == hipSuccess) scan_ms += ms;
        }
        ev_used = 0;
    }
    int32_t sync() {
        if (!HIPOK(hipStreamSynchronize(stream))) return FNN_EHIP;
        if (timing) drain_timing();
        return FNN_OK;
    }
    void collect_timing(fnn_stats& s) {
        s.t_scan_s = scan_ms * 1e-3;
        s.scan_launches = scan_launches;
    }
    void reset_timing() { scan_ms = 0.0; scan_launches = 0; ev_used = 0; }

    hipEvent_t next_event() {
        if (ev_used == ev_pool.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return nullptr;
            ev_pool.push_back(e);
        }
        return ev_pool[ev_used++];
    }

    static dim3 grid1(int32_t cnt) { return dim3((unsigned)((cnt + 255) / 256 > 0 ? (cnt + 255) / 256 : 1)); }

    int32_t launch_synth(const Dev& d, uint64_t seed, int32_t dist) {
        dim3 g((unsigned)((d.n + 255) / 256), (unsigned)d.n);
        hipLaunchKernelGGL(k_synth, g, dim3(256), 0, stream, d, seed, dist);
        return HIPOK(hipGetLastError()) && HIPOK(hipStreamSynchronize(stream)) ? FNN_OK : FNN_EHIP;
    }
    int32_t launch_validate(const Dev& d, int32_t* bad) {
        if (!d_bad && !HIPOK(hipMalloc((void**)&d_bad, sizeof(int)))) return FNN_EHIP;
        if (!HIPOK(hipMemsetAsync(d_bad, 0, sizeof(int), stream))) return FNN_EHIP;
        unsigned t = (unsigned)((d.n + 31) / 32);
        hipLaunchKernelGGL(k_validate, dim3(t, t), dim3(256), 0, stream, d, d_bad);
        if (!HIPOK(hipGetLastError())) return FNN_EHIP;
        int hb = 0;
        if (!HIPOK(hipMemcpyAsync(&hb, d_bad, sizeof(int), hipMemcpyDeviceToHost, stream)) ||
            !HIPOK(hipStreamSynchronize(stream)))
            return FNN_EHIP;
        *bad = hb;
        return FNN_OK;
    }
    int32_t launch_init(const Dev& d) {
        reset_timing();
        hipLaunchKernelGGL(k_init, grid1(d.n), dim3(256), 0, stream, d);
        return HIPOK(hipGetLastError()) ? FNN_OK : FNN_EHIP;
    }

    int32_t launch_event(const Dev& d, int32_t m_bound) {
        if (m_bound < 1) m_bound = 1;
        dim3 gs((unsigned)((m_bound + SCAN_TW - 1) / SCAN_TW), (unsigned)((m_bound + SCAN_TH - 1) / SCAN_TH));
        dim3 g1 = grid1(m_bound);
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (timing) { e0 = next_event(); e1 = next_event(); }
        if (e0) (void)hipEventRecord(e0, stream);
        hipLaunchKernelGGL(k_scan, gs, dim3(SCAN_THREADS), 0, stream, d);
        if (e1) (void)hipEventRecord(e1, stream);
        scan_launches++;
        hipLaunchKernelGGL(k_pick, dim3(1), dim3(1024), 0, stream, d, (int)(gs.x * gs.y));
        hipLaunchKernelGGL(k_rx_fill, g1, dim3(256), 0, stream, d);
        hipLaunchKernelGGL(k_decide, dim3(1), dim3(256), 0, stream, d);
        hipLaunchKernelGGL(k_subtract, g1, dim3(256), 0, stream, d);
        for (int i = 0; i < MAX_OPS; i++) hipLaunchKernelGGL(k_op, g1, dim3(256), 0, stream, d, i);
        hipLaunchKernelGGL(k_add, g1, dim3(256), 0, stream, d);
        hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, stream, d);
        return HIPOK(hipGetLastError()) ? FNN_OK : FNN_EHIP;
    }
};

using HipEngine = Engine<HipBackend>;

}  // namespace fnn

// ---------------------------------------------------------------------- C ABI
struct fnn_handle {
    fnn::HipEngine eng;
};

#define FNN_TRY(body)                                                     \
    try {                                                                 \
        body                                                              \
    } catch (const std::bad_alloc&) {                                     \
        return fnn::fail(FNN_ENOMEM, "out of host memory");               \
    } catch (const std::exception& e) {                                   \
        return fnn::fail(FNN_ESTATE, std::string("exception: ") + e.what()); \
    }

extern "C" {

int32_t fnn_abi_version(void) { return FASTNN_ABI_VERSION; }
const char* fnn_last_error(void) { return fnn::g_last_error.c_str(); }

int32_t fnn_device_count(void) {
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess) return fnn::fail(FNN_EHIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    return cnt;
}

int32_t fnn_create(int32_t n, const fnn_opts* opts, fnn_handle** out) {
    if (!out) return fnn::fail(FNN_EINVAL, "fnn_create: out is NULL");
    *out = nullptr;
    FNN_TRY(
        fnn_handle* h = new fnn_handle();
        int32_t rc = h->eng.create(n, opts);
        if (rc != FNN_OK) { h->eng.destroy(); delete h; return rc; }
        *out = h;
        return FNN_OK;
    )
}

int32_t fnn_destroy(fnn_handle* h) {
    if (!h) return FNN_OK;
    h->eng.destroy();
    delete h;
    return FNN_OK;
}

#define FNN_NEED(h) if (!(h)) return fnn::fail(FNN_EINVAL, "NULL handle")

int32_t fnn_set_rows(fnn_handle* h, int32_t row0, int32_t nrows, const double* rows, int64_t ld_in) {
    FNN_NEED(h);
    FNN_TRY(return h->eng.set_rows(row0, nrows, rows, ld_in);)
}
int32_t fnn_set_matrix_device(fnn_handle* h, const double* d_matrix, int64_t ld_in) {
    FNN_NEED(h);
    FNN_TRY(return h->eng.set_matrix_device(d_matrix, ld_in);)
}
int32_t fnn_synth(fnn_handle* h, uint64_t seed, int32_t dist) {
    FNN_NEED(h);
    FNN_TRY(return h->eng.synth(seed, dist);)
}
int32_t fnn_run(fnn_handle* h, int32_t* order_out, fnn_stats* stats) {
    FNN_NEED(h);
    if (!order_out) return fnn::fail(FNN_EINVAL, "fnn_run: order_out is NULL");
    FNN_TRY(return h->eng.run(order_out, stats);)
}
int32_t fnn_begin(fnn_handle* h) {
    FNN_NEED(h);
    FNN_TRY(return h->eng.begin();)
}
int32_t fnn_step(fnn_handle* h, fnn_event* ev) {
    FNN_NEED(h);
    FNN_TRY(return h->eng.step(ev);)
}
int32_t fnn_finish(fnn_handle* h, int32_t* order_out) {
    FNN_NEED(h);
    FNN_TRY(return h->eng.finish(order_out);)
}
int64_t fnn_get_events(fnn_handle* h, fnn_event* out, int64_t max_events) {
    if (!h) return fnn::fail(FNN_EINVAL, "NULL handle");
    return h->eng.get_events(out, max_events);
}
int32_t fnn_get_counts(fnn_handle* h, int32_t* num_active, int32_t* num_clusters, int32_t* num_nodes) {
    FNN_NEED(h);
    FNN_TRY(
        if (!h->eng.begun) return fnn::fail(FNN_ESTATE, "fnn_get_counts: call fnn_begin first");
        int32_t rc = h->eng.pull_state();
        if (rc != FNN_OK) return rc;
        if (num_active) *num_active = h->eng.hst.m;
        if (num_clusters) *num_clusters = h->eng.hst.c;
        if (num_nodes) *num_nodes = h->eng.hst.num_nodes;
        return FNN_OK;
    )
}
int32_t fnn_get_nodes(fnn_handle* h, int32_t* id, int32_t* nbr_id, double* Sx) {
    FNN_NEED(h);
    FNN_TRY(return h->eng.get_nodes(id, nbr_id, Sx);)
}
int32_t fnn_get_live_matrix(fnn_handle* h, double* out) {
    FNN_NEED(h);
    if (!out) return fnn::fail(FNN_EINVAL, "fnn_get_live_matrix: out is NULL");
    FNN_TRY(return h->eng.get_live_matrix(out);)
}
int32_t fnn_set_scan_timing(fnn_handle* h, int32_t enable) {
    FNN_NEED(h);
    h->eng.be.timing = enable != 0;
    return FNN_OK;
}

int32_t fnn_canonical_order_f64(const double* D, int32_t n, int64_t ld, const fnn_opts* opts,
                                int32_t* order_out, fnn_stats* stats) {
    if (n < 0 || !order_out || (n > 0 && !D)) return fnn::fail(FNN_EINVAL, "fnn_canonical_order_f64: bad arguments");
    if (n <= 3) {  // NetMakerOriginal.java:133-140, no device needed
        for (int32_t i = 0; i <= n; i++) order_out[i] = i;
        if (stats) *stats = fnn_stats{};
        return FNN_OK;
    }
    fnn_handle* h = nullptr;
    int32_t rc = fnn_create(n, opts, &h);
    if (rc != FNN_OK) return rc;
    rc = fnn_set_rows(h, 0, n, D, ld);
    if (rc == FNN_OK) rc = fnn_run(h, order_out, stats);
    std::string keep = fnn::g_last_error;
    fnn_destroy(h);
    fnn::g_last_error = keep;
    return rc;
}

int32_t fnn_stream_probe(int32_t device, int64_t bytes, int32_t reps, double* gbps_out) {
    if (bytes < (1 << 20) || reps < 1 || !gbps_out) return fnn::fail(FNN_EINVAL, "fnn_stream_probe: bad arguments");
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return fnn::fail(FNN_EHIP, hipGetErrorString(e));
    void* p = nullptr;
    double* sink = nullptr;
    if ((e = hipMalloc(&p, (size_t)bytes)) != hipSuccess) return fnn::fail(FNN_ENOMEM, hipGetErrorString(e));
    if ((e = hipMalloc((void**)&sink, 8)) != hipSuccess) { (void)hipFree(p); return fnn::fail(FNN_ENOMEM, hipGetErrorString(e)); }
    (void)hipMemset(p, 0, (size_t)bytes);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    int64_t n16 = bytes / 16;
    dim3 g(256 * 8);
    hipLaunchKernelGGL(fnn::k_stream, g, dim3(256), 0, 0, (const double2*)p, n16, sink);  // warm-up
    (void)hipEventRecord(e0, 0);
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL(fnn::k_stream, g, dim3(256), 0, 0, (const double2*)p, n16, sink);
    (void)hipEventRecord(e1, 0);
    e = hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    (void)hipFree(p);
    (void)hipFree(sink);
    if (e != hipSuccess || ms <= 0.f) return fnn::fail(FNN_EHIP, "fnn_stream_probe: timing failed");
    *gbps_out = (double)bytes * reps / (ms * 1e-3) / 1e9;
    return FNN_OK;
}

}  // extern "C"
