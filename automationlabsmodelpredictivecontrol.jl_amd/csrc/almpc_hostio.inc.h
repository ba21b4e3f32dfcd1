// almpc_hostio.inc.h -- the host-facing side of a step (included by almpc_api.hip after the handle and its entry points).
//
// The reference's per-step contract is host in / host out: update_initialization!(C, x0) takes a host vector
// (src/main/computation_mpc.jl:17-29), calculate!(C) leaves x, e_x, u, e_u in host matrices (src/main/computation_mpc.jl:50-53),
// and what a receding-horizon caller applies is u[:,1].  Through almpc_update_initialization / almpc_get_results that is a
// synchronous pageable upload and seven synchronous pageable read-backs per step (32 MB at the benchmark shape).  Here:
//   * pinned staging owned by the handle, rings of IO_DEPTH slots;
//   * x0 is not copied at all: the step's kernels read it from the pinned slot over the link (393 KB), so an update is a host
//     memcpy (or none: almpc_x0_staging) and a pointer flip -- no HIP call.  The price is +9 us on the step (the 393 KB cross the link
//     at its start, every workgroup waits for its tile); the alternative, a copy-engine upload into a device slot on a stream of its
//     own (ALMPC_X0_UPLOAD=1), removes that but costs three HIP calls and a DMA latency per step and came out slower (11.7 against
//     12.8 k batch-steps/s pipelined, 8.3 against 10.4 k serial);
//   * the small results (first inputs, status, iteration counts: 147 KB) are written by one pack kernel straight into the pinned
//     slot, followed by one event: the next step starts right behind it;
//   * the large arrays go out on a copy-out stream tied to the compute stream by events, so their read-back of step k runs under
//     the kernel of step k+1 only as far as the result buffers allow (the next step waits for it before overwriting them);
//   * the first input of every instance (m doubles: 131 KB at the benchmark shape instead of 32 MB) as a result of its own;
//   * zero-copy views of the pinned slots for callers that can read them in place (Julia: unsafe_wrap).
// And one process driving several GPUs (almpc_group_*): one handle per device on its contiguous shard of the batch, every entry
// point fans out over the handles without a cross-device synchronisation on the step path (SURVEY.md section 8b proposal:
// almpc_create(..., n_devices, device_ids, ...); section 8e: instances never interact).

namespace {

// u0[i][a] = u[i][0][a] and copies of the three per-instance words of the step: everything small a host caller wants after a step,
// packed into the slot's own buffers so that the next step (which rewrites u / status / iters) can start while they travel
inline __global__ __launch_bounds__(256) void k_pack_step_summary(int batch, int m, int N, const double* u, const int32_t* status,
                                                           const int32_t* iters, const int32_t* piters, double* u0, int32_t* ints) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < (long)batch * m) u0[t] = u[(t / m) * (long)m * N + (t % m)];
    if (t < batch) {
        ints[t] = status[t];
        ints[(long)batch + t] = iters ? iters[t] : 0;
        ints[2L * batch + t] = piters[t];
    }
}

template <typename T>
hipError_t pinned(T** p, size_t count) {
    return hipHostMalloc(reinterpret_cast<void**>(p), count * sizeof(T), hipHostMallocDefault);
}

int io_init_body(almpc_handle* h) {
    almpc_handle::Io& io = h->io;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, hipStreamCreateWithFlags(&io.s_out, hipStreamNonBlocking));
    HIP_TRY(h, hipStreamCreateWithFlags(&io.s_in, hipStreamNonBlocking));
    { const char* up = getenv("ALMPC_X0_UPLOAD"); io.upload = up && up[0] == '1'; }
    const size_t b = (size_t)h->batch;
    for (int s = 0; s < almpc_handle::IO_DEPTH; ++s) {
        HIP_TRY(h, pinned(&io.hX0[s], b * h->n));
        HIP_TRY(h, pinned(&io.hU0[s], b * h->m));
        HIP_TRY(h, pinned(&io.hInts[s], 3 * b));
        // the device's view of the pinned slots (kernels read x0 from / write the step summary to host memory directly)
        HIP_TRY(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&io.dX0[s]), io.hX0[s], 0));
        HIP_TRY(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&io.dU0[s]), io.hU0[s], 0));
        HIP_TRY(h, hipHostGetDevicePointer(reinterpret_cast<void**>(&io.dInts[s]), io.hInts[s], 0));
        HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&io.dX0dev[s]), b * h->n * sizeof(double)));
        for (hipEvent_t* e : {&io.ev_used[s], &io.ev_packed[s], &io.ev_done[s], &io.ev_in[s]})
            HIP_TRY(h, hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    HIP_TRY(h, hipEventCreateWithFlags(&io.ev_big, hipEventDisableTiming));
    io.ready = true;
    return ALMPC_OK;
}

void io_free(almpc_handle* h);

int io_init(almpc_handle* h) {
    if (h->io.ready) return ALMPC_OK;
    const int rc = io_init_body(h);
    if (rc != ALMPC_OK) {   // a failure half-way: release what was made (the message of the failed call survives), so that a retry starts clean
        const std::string msg = h->err;
        io_free(h);
        h->err = msg;
    }
    return rc;
}

void io_free(almpc_handle* h) {
    almpc_handle::Io& io = h->io;
    if (io.dX0_own) h->dX0 = io.dX0_own;
    if (io.s_out) (void)hipStreamSynchronize(io.s_out);
    if (io.s_in) (void)hipStreamSynchronize(io.s_in);
    for (int s = 0; s < almpc_handle::IO_DEPTH; ++s) {
        if (io.dX0dev[s]) (void)hipFree(io.dX0dev[s]);
        if (io.ev_in[s]) (void)hipEventDestroy(io.ev_in[s]);
        for (void* p : {(void*)io.hX0[s], (void*)io.hX[s], (void*)io.hEx[s], (void*)io.hU[s], (void*)io.hEu[s], (void*)io.hU0[s], (void*)io.hInts[s]})
            if (p) (void)hipHostFree(p);
        for (hipEvent_t e : {io.ev_used[s], io.ev_packed[s], io.ev_done[s]})
            if (e) (void)hipEventDestroy(e);
    }
    if (io.ev_big) (void)hipEventDestroy(io.ev_big);
    if (io.s_out) (void)hipStreamDestroy(io.s_out);
    if (io.s_in) (void)hipStreamDestroy(io.s_in);
    io = almpc_handle::Io();
}

// Wait for an event the way almpc_synchronize waits for the stream: poll first (a blocking wake-up costs milliseconds on this pool)
hipError_t event_wait(hipEvent_t e) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipEventQuery(e);
        if (q != hipErrorNotReady) return q;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) break;
    }
    return hipEventSynchronize(e);
}

}  // namespace

extern "C" {

// Zero-copy input: the pinned slot the NEXT almpc_update_initialization_async will hand to the device.  The caller writes its states
// there and passes the same pointer on: no staging copy.
int almpc_x0_staging(almpc_handle* h, double** x0_slot) {
    if (!h || !x0_slot) return h ? fail(h, ALMPC_ERR_INVALID, "x0_staging: null pointer") : ALMPC_ERR_INVALID;
    { const int rc = io_init(h); if (rc != ALMPC_OK) return rc; }
    almpc_handle::Io& io = h->io;
    const int s = (int)(io.x0_count % almpc_handle::IO_DEPTH);
    if (io.used_pending[s]) {   // the slot is written by the caller from here on: the last step that read it must have finished
        HIP_TRY(h, hipSetDevice(h->device));
        HIP_TRY(h, event_wait(io.ev_used[s]));
        io.used_pending[s] = false;
    }
    *x0_slot = io.hX0[s];
    return ALMPC_OK;
}

int almpc_update_initialization_async(almpc_handle* h, const double* x0) {
    if (!h || !x0) return h ? fail(h, ALMPC_ERR_INVALID, "update_initialization_async: null x0") : ALMPC_ERR_INVALID;
    { const int rc = io_init(h); if (rc != ALMPC_OK) return rc; }
    almpc_handle::Io& io = h->io;
    const int s = (int)(io.x0_count % almpc_handle::IO_DEPTH);
    // the slot is free once the last step that read it has finished (normally long ago: IO_DEPTH - 1 steps may be in flight)
    if (io.used_pending[s]) {
        HIP_TRY(h, hipSetDevice(h->device));
        HIP_TRY(h, event_wait(io.ev_used[s]));
        io.used_pending[s] = false;
    }
    if (x0 != io.hX0[s]) std::memcpy(io.hX0[s], x0, (size_t)h->batch * h->n * sizeof(double));   // (almpc_x0_staging: already in place)
    if (io.x0_slot < 0) io.dX0_own = h->dX0;   // (the handle's own buffer is kept and freed with the handle)
    if (io.upload) {
        // pinned slot -> device slot on the copy-in stream (under the step that is running); everything enqueued on the compute stream
        // from here on comes after it
        HIP_TRY(h, hipSetDevice(h->device));
        HIP_TRY(h, hipMemcpyAsync(io.dX0dev[s], io.hX0[s], (size_t)h->batch * h->n * sizeof(double), hipMemcpyHostToDevice, io.s_in));
        HIP_TRY(h, hipEventRecord(io.ev_in[s], io.s_in));
        HIP_TRY(h, hipStreamWaitEvent(h->stream, io.ev_in[s], 0));
        h->dX0 = io.dX0dev[s];
    } else {
        h->dX0 = io.dX0[s];                    // kernels enqueued from here on read this slot, in place, over the link
    }
    io.x0_slot = s;
    io.x0_count += 1;
    return ALMPC_OK;
}

int almpc_get_results_async(almpc_handle* h, uint32_t want) {
    if (!h) return ALMPC_ERR_INVALID;
    if (!h->designed) return fail(h, ALMPC_ERR_NOT_DESIGNED, "get_results_async before design");
    if (want == 0 || (want & ~(uint32_t)ALMPC_WANT_ALL)) return fail(h, ALMPC_ERR_INVALID, "get_results_async: want must be a non-empty mask of ALMPC_WANT_*");
    { const int rc = io_init(h); if (rc != ALMPC_OK) return rc; }
    almpc_handle::Io& io = h->io;
    HIP_TRY(h, hipSetDevice(h->device));
    // "solution or verdict" on the ticket path too: what the step's finish left undecided is redone on the stream before the results
    // are packed (gated launches: nothing runs unless the step left something)
    { const int rc = enqueue_gated_redo(h); if (rc != ALMPC_OK) return rc; }
    const long t = io.next_ticket;
    const int s = (int)(t % almpc_handle::IO_DEPTH);
    const size_t b = (size_t)h->batch, xs = b * h->n * (h->N + 1), us = b * h->nz;
    // (the pinned slot of ticket t - IO_DEPTH is overwritten from here on: its views are valid until this call, as the header says)
    const bool big = (want & (ALMPC_WANT_X | ALMPC_WANT_E_X | ALMPC_WANT_U | ALMPC_WANT_E_U)) != 0;
    if (io.big_copy_pending && big) {   // two read-backs from the same result buffers in a row (no step between them): keep order simple
        HIP_TRY(h, hipStreamWaitEvent(h->stream, io.ev_big, 0));
        io.big_copy_pending = false;
    }
    if ((want & ALMPC_WANT_X) && !io.hX[s]) HIP_TRY(h, pinned(&io.hX[s], xs));
    if ((want & ALMPC_WANT_E_X) && !io.hEx[s]) HIP_TRY(h, pinned(&io.hEx[s], xs));
    if ((want & ALMPC_WANT_U) && !io.hU[s]) HIP_TRY(h, pinned(&io.hU[s], us));
    if ((want & ALMPC_WANT_E_U) && !io.hEu[s]) HIP_TRY(h, pinned(&io.hEu[s], us));
    const bool small = (want & (ALMPC_WANT_FIRST_INPUT | ALMPC_WANT_STATUS | ALMPC_WANT_ITERS | ALMPC_WANT_POLISH_ITERS)) != 0;
    if (small) {   // straight into the pinned slot: no copy, the next step starts right behind this kernel
        const long cnt = std::max((long)b * h->m, (long)b);
        hipLaunchKernelGGL(k_pack_step_summary, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->stream, h->batch, h->m, h->N,
                           (const double*)h->dU, (const int32_t*)h->dStatus, (const int32_t*)h->dIters, (const int32_t*)h->dPiters,
                           io.dU0[s], io.dInts[s]);
        HIP_TRY(h, hipGetLastError());
    }
    if (big) {
        HIP_TRY(h, hipEventRecord(io.ev_packed[s], h->stream));
        HIP_TRY(h, hipStreamWaitEvent(io.s_out, io.ev_packed[s], 0));
        if (want & ALMPC_WANT_U) HIP_TRY(h, hipMemcpyAsync(io.hU[s], h->dU, us * sizeof(double), hipMemcpyDeviceToHost, io.s_out));
        if (want & ALMPC_WANT_E_U) HIP_TRY(h, hipMemcpyAsync(io.hEu[s], h->dEu, us * sizeof(double), hipMemcpyDeviceToHost, io.s_out));
        if (want & ALMPC_WANT_X) HIP_TRY(h, hipMemcpyAsync(io.hX[s], h->dX, xs * sizeof(double), hipMemcpyDeviceToHost, io.s_out));
        if (want & ALMPC_WANT_E_X) HIP_TRY(h, hipMemcpyAsync(io.hEx[s], h->dEx, xs * sizeof(double), hipMemcpyDeviceToHost, io.s_out));
        HIP_TRY(h, hipEventRecord(io.ev_done[s], io.s_out));   // (after the pack kernel as well: s_out waited for the compute stream)
        // the next step waits for THIS read-back: an event of its own, which later small requests landing in the same slot (they
        // re-record ev_done on the compute stream) cannot replace
        HIP_TRY(h, hipEventRecord(io.ev_big, io.s_out));
    } else {
        HIP_TRY(h, hipEventRecord(io.ev_done[s], h->stream));
    }
    if (big) { io.big_copy_pending = true; io.big_copy_slot = s; }
    io.want[s] = want;
    io.ticket[s] = t;
    io.next_ticket += 1;
    return (int)(t & 0x3fffffff);
}

static int io_slot_of(almpc_handle* h, int ticket, const char* who) {
    almpc_handle::Io& io = h->io;
    if (!io.ready || ticket < 0) { fail(h, ALMPC_ERR_INVALID, std::string(who) + ": no such ticket"); return -1; }
    for (int s = 0; s < almpc_handle::IO_DEPTH; ++s)
        if (io.ticket[s] >= 0 && (int)(io.ticket[s] & 0x3fffffff) == ticket) return s;
    fail(h, ALMPC_ERR_INVALID, std::string(who) + ": ticket is not outstanding (only the last IO_DEPTH requests are kept)");
    return -1;
}

int almpc_get_results_wait(almpc_handle* h, int ticket, double* x, double* e_x, double* u, double* e_u, double* u0, int32_t* status,
                           int32_t* iters, int32_t* polish_iters) {
    if (!h) return ALMPC_ERR_INVALID;
    const int s = io_slot_of(h, ticket, "get_results_wait");
    if (s < 0) return ALMPC_ERR_INVALID;
    almpc_handle::Io& io = h->io;
    HIP_TRY(h, hipSetDevice(h->device));
    HIP_TRY(h, event_wait(io.ev_done[s]));
    const uint32_t w = io.want[s];
    const size_t b = (size_t)h->batch, xs = b * h->n * (h->N + 1), us = b * h->nz;
    auto need = [&](const void* p, uint32_t bit) { return p != nullptr && !(w & bit); };
    if (need(x, ALMPC_WANT_X) || need(e_x, ALMPC_WANT_E_X) || need(u, ALMPC_WANT_U) || need(e_u, ALMPC_WANT_E_U) ||
        need(u0, ALMPC_WANT_FIRST_INPUT) || need(status, ALMPC_WANT_STATUS) || need(iters, ALMPC_WANT_ITERS) ||
        need(polish_iters, ALMPC_WANT_POLISH_ITERS))
        return fail(h, ALMPC_ERR_INVALID, "get_results_wait: a result was asked for that the request of this ticket did not include");
    if (x) std::memcpy(x, io.hX[s], xs * sizeof(double));
    if (e_x) std::memcpy(e_x, io.hEx[s], xs * sizeof(double));
    if (u) std::memcpy(u, io.hU[s], us * sizeof(double));
    if (e_u) std::memcpy(e_u, io.hEu[s], us * sizeof(double));
    if (u0) std::memcpy(u0, io.hU0[s], b * h->m * sizeof(double));
    if (status) std::memcpy(status, io.hInts[s], b * sizeof(int32_t));
    if (iters) std::memcpy(iters, io.hInts[s] + b, b * sizeof(int32_t));
    if (polish_iters) std::memcpy(polish_iters, io.hInts[s] + 2 * b, b * sizeof(int32_t));
    return ALMPC_OK;
}

int almpc_host_results(almpc_handle* h, int ticket, const double** x, const double** e_x, const double** u, const double** e_u,
                       const double** u0, const int32_t** status, const int32_t** iters, const int32_t** polish_iters) {
    if (!h) return ALMPC_ERR_INVALID;
    const int s = io_slot_of(h, ticket, "host_results");
    if (s < 0) return ALMPC_ERR_INVALID;
    almpc_handle::Io& io = h->io;
    const uint32_t w = io.want[s];
    const size_t b = (size_t)h->batch;
    if (x) *x = (w & ALMPC_WANT_X) ? io.hX[s] : nullptr;
    if (e_x) *e_x = (w & ALMPC_WANT_E_X) ? io.hEx[s] : nullptr;
    if (u) *u = (w & ALMPC_WANT_U) ? io.hU[s] : nullptr;
    if (e_u) *e_u = (w & ALMPC_WANT_E_U) ? io.hEu[s] : nullptr;
    if (u0) *u0 = (w & ALMPC_WANT_FIRST_INPUT) ? io.hU0[s] : nullptr;
    if (status) *status = (w & ALMPC_WANT_STATUS) ? io.hInts[s] : nullptr;
    if (iters) *iters = (w & ALMPC_WANT_ITERS) ? io.hInts[s] + b : nullptr;
    if (polish_iters) *polish_iters = (w & ALMPC_WANT_POLISH_ITERS) ? io.hInts[s] + 2 * b : nullptr;
    return ALMPC_OK;
}

int almpc_get_first_input(almpc_handle* h, double* u0) {
    if (!h || !u0) return h ? fail(h, ALMPC_ERR_INVALID, "get_first_input: null u0") : ALMPC_ERR_INVALID;
    if (h->lazy_pending) {   // (synchronous getter: settle a lazily deferred redo before the first inputs are packed)
        HIP_TRY(h, hipSetDevice(h->device));
        const int rc_ = wait_and_settle(h, true);
        if (rc_ != ALMPC_OK) return rc_;
    }
    const int t = almpc_get_results_async(h, ALMPC_WANT_FIRST_INPUT);
    if (t < 0) return t;
    return almpc_get_results_wait(h, t, nullptr, nullptr, nullptr, nullptr, u0, nullptr, nullptr, nullptr);
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------
// One process, several GPUs
// ------------------------------------------------------------------------------------------------
struct almpc_group {
    int n = 0, m = 0, N = 0, batch = 0;
    std::vector<almpc_handle*> hs;
    std::vector<int> first, count;   // shard of handle i: instances [first, first + count)
    std::vector<int> tickets;
    // almpc_group_get_results_async: group ticket t -> the handles' own tickets (ring of two, as the handles keep theirs)
    std::vector<int> gt[2];
    long gticket[2] = {-1, -1};
    long next_gticket = 0;
    std::string err;
};

namespace {
int gfail(almpc_group* g, int code, int i) {
    if (g) g->err = "handle " + std::to_string(i) + ": " + (g->hs[i] ? g->hs[i]->err : std::string("null"));
    return code;
}

// Every handle's call on a host thread of its own (the design entry points are synchronous: uploads, design kernels, a host-side DARE):
// all devices design at the same time instead of one after the other.  Returns the first failure.
template <class F>
int group_fanout(almpc_group* g, F&& call) {
    const size_t k = g->hs.size();
    std::vector<int> rc(k, ALMPC_OK);
    if (k == 1) rc[0] = call((int)0);
    else {
        std::vector<std::thread> th;
        for (size_t i = 0; i < k; ++i) th.emplace_back([&, i]() { rc[i] = call((int)i); });
        for (auto& q : th) q.join();
    }
    for (size_t i = 0; i < k; ++i)
        if (rc[i] != ALMPC_OK) return gfail(g, rc[i], (int)i);
    return ALMPC_OK;
}

}  // namespace

extern "C" {

int almpc_group_create(almpc_group** out, int n, int m, int N, int batch, int n_devices, const int* device_ids, uint32_t flags) {
    if (!out) return ALMPC_ERR_INVALID;
    *out = nullptr;
    if (n_devices < 1 || !device_ids || batch < n_devices) return ALMPC_ERR_INVALID;
    almpc_group* g = new almpc_group();
    g->n = n; g->m = m; g->N = N; g->batch = batch;
    for (int i = 0; i < n_devices; ++i) {   // contiguous shards, sizes differing by at most one (the rule of sharding.shard_range)
        const int base = batch / n_devices, rem = batch % n_devices;
        const int cnt = base + (i < rem ? 1 : 0), fst = i * base + std::min(i, rem);
        almpc_handle* h = nullptr;
        const int rc = almpc_create(&h, n, m, N, cnt, device_ids[i], flags);
        if (rc != ALMPC_OK) {
            for (almpc_handle* q : g->hs) almpc_destroy(q);
            delete g;
            return rc;
        }
        g->hs.push_back(h); g->first.push_back(fst); g->count.push_back(cnt);
    }
    g->tickets.assign(n_devices, -1);
    *out = g;
    return ALMPC_OK;
}

void almpc_group_destroy(almpc_group* g) {
    if (!g) return;
    for (almpc_handle* h : g->hs) almpc_destroy(h);
    delete g;
}

const char* almpc_group_last_error(const almpc_group* g) { return g ? g->err.c_str() : "null group"; }
int almpc_group_size(const almpc_group* g) { return g ? (int)g->hs.size() : 0; }
almpc_handle* almpc_group_handle(almpc_group* g, int i) { return (g && i >= 0 && i < (int)g->hs.size()) ? g->hs[i] : nullptr; }

int almpc_group_shard(const almpc_group* g, int i, int* first, int* count) {
    if (!g || i < 0 || i >= (int)g->hs.size()) return ALMPC_ERR_INVALID;
    if (first) *first = g->first[i];
    if (count) *count = g->count[i];
    return ALMPC_OK;
}

int almpc_group_design_shared(almpc_group* g, const double* A, const double* B, const double* Q, const double* R, const double* S,
                              const double* P, const double* umin, const double* umax, const double* xmin, const double* xmax,
                              double rho, double sigma) {
    if (!g) return ALMPC_ERR_INVALID;
    return group_fanout(g, [&](int i) { return almpc_design_shared(g->hs[i], A, B, Q, R, S, P, umin, umax, xmin, xmax, rho, sigma); });
}

// per-handle options, fanned out (each takes effect at the next design, as the single-handle calls say)
int almpc_group_set_terminal_equality(almpc_group* g, int on) {
    if (!g) return ALMPC_ERR_INVALID;
    for (size_t i = 0; i < g->hs.size(); ++i) { const int rc = almpc_set_terminal_equality(g->hs[i], on); if (rc != ALMPC_OK) return gfail(g, rc, (int)i); }
    return ALMPC_OK;
}
int almpc_group_set_rho_profile(almpc_group* g, int mode) {
    if (!g) return ALMPC_ERR_INVALID;
    for (size_t i = 0; i < g->hs.size(); ++i) { const int rc = almpc_set_rho_profile(g->hs[i], mode); if (rc != ALMPC_OK) return gfail(g, rc, (int)i); }
    return ALMPC_OK;
}
int almpc_group_set_structured_fallback(almpc_group* g, int on) {
    if (!g) return ALMPC_ERR_INVALID;
    for (size_t i = 0; i < g->hs.size(); ++i) { const int rc = almpc_set_structured_fallback(g->hs[i], on); if (rc != ALMPC_OK) return gfail(g, rc, (int)i); }
    return ALMPC_OK;
}
int almpc_group_set_state_box(almpc_group* g, const double* xmin, const double* xmax) {
    if (!g) return ALMPC_ERR_INVALID;
    for (size_t i = 0; i < g->hs.size(); ++i) { const int rc = almpc_set_state_box(g->hs[i], xmin, xmax); if (rc != ALMPC_OK) return gfail(g, rc, (int)i); }
    return ALMPC_OK;
}

// one model per instance: A_batch [batch][n*n], B_batch [batch][n*m], P (NULL | n*n | [batch][n*n]) cut along the shards
int almpc_group_design_batched(almpc_group* g, const double* A_batch, const double* B_batch, const double* Q, const double* R, const double* S,
                               const double* P, int P_per_instance, const double* umin, const double* umax, double rho, double sigma) {
    if (!g || !A_batch || !B_batch) return ALMPC_ERR_INVALID;
    const size_t nn = (size_t)g->n * g->n, nm = (size_t)g->n * g->m;
    return group_fanout(g, [&](int i) {
        const size_t f = (size_t)g->first[i];
        return almpc_design_batched(g->hs[i], A_batch + f * nn, B_batch + f * nm, Q, R, S, (P && P_per_instance) ? P + f * nn : P, P_per_instance,
                                    umin, umax, rho, sigma);
    });
}

// the re-linearisation pipeline of a black-box Fnn model (BASELINE configs[3]) on every device: shared network and references
int almpc_group_relin_fnn_setup(almpc_group* g, int H, int L, int activation, const double* W_in, const double* W_h, const double* b_h,
                                const double* W_out, const double* xref, const double* uref, const double* Q, const double* R,
                                const double* S, const double* P, const double* umin, const double* umax, double rho, double sigma) {
    if (!g) return ALMPC_ERR_INVALID;
    return group_fanout(g, [&](int i) {
        return almpc_relin_fnn_setup(g->hs[i], H, L, activation, W_in, W_h, b_h, W_out, xref, uref, Q, R, S, P, umin, umax, rho, sigma);
    });
}
int almpc_group_relin_fnn_step_async(almpc_group* g, const almpc_opts* opts) {
    if (!g) return ALMPC_ERR_INVALID;
    for (size_t i = 0; i < g->hs.size(); ++i) { const int rc = almpc_relin_fnn_step_async(g->hs[i], opts); if (rc != ALMPC_OK) return gfail(g, rc, (int)i); }
    return ALMPC_OK;
}
int almpc_group_relin_fnn_step(almpc_group* g, const almpc_opts* opts) {
    const int rc = almpc_group_relin_fnn_step_async(g, opts);
    return rc != ALMPC_OK ? rc : almpc_group_synchronize(g);
}
int almpc_group_relin_fnn_advance(almpc_group* g) {
    if (!g) return ALMPC_ERR_INVALID;
    for (size_t i = 0; i < g->hs.size(); ++i) { const int rc = almpc_relin_fnn_advance(g->hs[i]); if (rc != ALMPC_OK) return gfail(g, rc, (int)i); }
    return ALMPC_OK;
}
int almpc_group_advance_plant(almpc_group* g) {
    if (!g) return ALMPC_ERR_INVALID;
    for (size_t i = 0; i < g->hs.size(); ++i) { const int rc = almpc_advance_plant(g->hs[i]); if (rc != ALMPC_OK) return gfail(g, rc, (int)i); }
    return ALMPC_OK;
}

// the SQP loop (BASELINE configs[4]) on every device; P n*n (P_per_instance = 0) or [batch][n*n] cut along the shards
int almpc_group_sqp_fnn_set_structured(almpc_group* g, int on) {
    if (!g) return ALMPC_ERR_INVALID;
    for (size_t i = 0; i < g->hs.size(); ++i) { const int rc = almpc_sqp_fnn_set_structured(g->hs[i], on); if (rc != ALMPC_OK) return gfail(g, rc, (int)i); }
    return ALMPC_OK;
}
int almpc_group_sqp_fnn_set_step_rule(almpc_group* g, int rule) {
    if (!g) return ALMPC_ERR_INVALID;
    for (size_t i = 0; i < g->hs.size(); ++i) { const int rc = almpc_sqp_fnn_set_step_rule(g->hs[i], rule); if (rc != ALMPC_OK) return gfail(g, rc, (int)i); }
    return ALMPC_OK;
}
int almpc_group_sqp_fnn_setup(almpc_group* g, int H, int L, int activation, const double* W_in, const double* W_h, const double* b_h,
                              const double* W_out, const double* xref, const double* uref, const double* Q, const double* R,
                              const double* S, const double* P, int P_per_instance, const double* umin, const double* umax, double rho,
                              double sigma) {
    if (!g || !P) return ALMPC_ERR_INVALID;
    const size_t nn = (size_t)g->n * g->n;
    return group_fanout(g, [&](int i) {
        return almpc_sqp_fnn_setup(g->hs[i], H, L, activation, W_in, W_h, b_h, W_out, xref, uref, Q, R, S,
                                   P_per_instance ? P + (size_t)g->first[i] * nn : P, P_per_instance, umin, umax, rho, sigma);
    });
}
int almpc_group_sqp_fnn_start(almpc_group* g, const double* x0, const double* u_guess) {
    if (!g || !x0) return ALMPC_ERR_INVALID;
    return group_fanout(g, [&](int i) {
        const size_t f = (size_t)g->first[i];
        return almpc_sqp_fnn_start(g->hs[i], x0 + f * g->n, u_guess ? u_guess + f * g->m * g->N : nullptr);
    });
}
// `iters` iterations on every device at the same time (each handle's loop synchronises its own stream once at the end);
// step_inf / defect_inf: maxima over the whole batch.  A device that had to skip an instance returns ALMPC_ERR_NUMERIC as the
// single-handle call does (the others have finished their iterations): almpc_group_sqp_fnn_skipped tells which instances.
int almpc_group_sqp_fnn_iterate(almpc_group* g, int iters, double step_scale, const almpc_opts* opts, double* step_inf, double* defect_inf) {
    if (!g || iters < 1) return ALMPC_ERR_INVALID;
    const size_t k = g->hs.size();
    std::vector<std::vector<double>> st(k, std::vector<double>((size_t)iters, 0.0)), de(k, std::vector<double>((size_t)iters, 0.0));
    const int rc = group_fanout(g, [&](int i) { return almpc_sqp_fnn_iterate(g->hs[i], iters, step_scale, opts, st[i].data(), de[i].data()); });
    for (int it = 0; it < iters; ++it) {
        double a = 0.0, b = 0.0;
        for (size_t i = 0; i < k; ++i) { a = std::max(a, st[i][it]); b = std::max(b, de[i][it]); }
        if (step_inf) step_inf[it] = a;
        if (defect_inf) defect_inf[it] = b;
    }
    return rc;
}
int almpc_group_sqp_fnn_skipped(almpc_group* g, int32_t* skipped) {
    if (!g || !skipped) return ALMPC_ERR_INVALID;
    for (size_t i = 0; i < g->hs.size(); ++i) { const int rc = almpc_sqp_fnn_skipped(g->hs[i], skipped + g->first[i]); if (rc != ALMPC_OK) return gfail(g, rc, (int)i); }
    return ALMPC_OK;
}

// zero-copy input of the next almpc_group_update_initialization: slots[i] = the pinned buffer of handle i ([count_i][n]); write the
// shard's states there and pass the group call a NULL x0 ... (see almpc_x0_staging): here the pointers only
int almpc_group_x0_staging(almpc_group* g, double** slots) {
    if (!g || !slots) return ALMPC_ERR_INVALID;
    for (size_t i = 0; i < g->hs.size(); ++i) { const int rc = almpc_x0_staging(g->hs[i], &slots[i]); if (rc != ALMPC_OK) return gfail(g, rc, (int)i); }
    return ALMPC_OK;
}
// the states were written into the slots of almpc_group_x0_staging: hand every device its slot (no copy)
int almpc_group_update_initialization_staged(almpc_group* g, double* const* slots) {
    if (!g || !slots) return ALMPC_ERR_INVALID;
    for (size_t i = 0; i < g->hs.size(); ++i) { const int rc = almpc_update_initialization_async(g->hs[i], slots[i]); if (rc != ALMPC_OK) return gfail(g, rc, (int)i); }
    return ALMPC_OK;
}

int almpc_group_set_reference(almpc_group* g, const double* xref, const double* uref, int per_instance) {
    if (!g || !xref || !uref) return ALMPC_ERR_INVALID;
    for (size_t i = 0; i < g->hs.size(); ++i) {
        const size_t ox = per_instance ? (size_t)g->first[i] * g->n * (g->N + 1) : 0, ou = per_instance ? (size_t)g->first[i] * g->m * g->N : 0;
        const int rc = almpc_set_reference(g->hs[i], xref + ox, uref + ou, per_instance);
        if (rc != ALMPC_OK) return gfail(g, rc, (int)i);
    }
    return ALMPC_OK;
}

// x0 [batch][n], host: every device's upload is enqueued (pinned staging, copy-in streams) before any of them is waited for
int almpc_group_update_initialization(almpc_group* g, const double* x0) {
    if (!g || !x0) return ALMPC_ERR_INVALID;
    for (size_t i = 0; i < g->hs.size(); ++i) {
        const int rc = almpc_update_initialization_async(g->hs[i], x0 + (size_t)g->first[i] * g->n);
        if (rc != ALMPC_OK) return gfail(g, rc, (int)i);
    }
    return ALMPC_OK;
}

int almpc_group_calculate_async(almpc_group* g, const almpc_opts* opts) {
    if (!g) return ALMPC_ERR_INVALID;
    for (size_t i = 0; i < g->hs.size(); ++i) {
        const int rc = almpc_calculate_async(g->hs[i], opts);
        if (rc != ALMPC_OK) return gfail(g, rc, (int)i);
    }
    return ALMPC_OK;
}

int almpc_group_synchronize(almpc_group* g) {
    if (!g) return ALMPC_ERR_INVALID;
    for (size_t i = 0; i < g->hs.size(); ++i) {
        const int rc = almpc_synchronize(g->hs[i]);
        if (rc != ALMPC_OK) return gfail(g, rc, (int)i);
    }
    return ALMPC_OK;
}

int almpc_group_calculate(almpc_group* g, const almpc_opts* opts) {
    const int rc = almpc_group_calculate_async(g, opts);
    return rc != ALMPC_OK ? rc : almpc_group_synchronize(g);
}

// Results of the whole batch into the caller's arrays (layouts of almpc_get_results; u0 [batch][m]; any pointer may be NULL): the
// read-backs of all devices are in flight together, each into its slice.
int almpc_group_get_results(almpc_group* g, double* x, double* e_x, double* u, double* e_u, double* u0, int32_t* status, int32_t* iters,
                            int32_t* polish_iters) {
    if (!g) return ALMPC_ERR_INVALID;
    const uint32_t want = (x ? ALMPC_WANT_X : 0) | (e_x ? ALMPC_WANT_E_X : 0) | (u ? ALMPC_WANT_U : 0) | (e_u ? ALMPC_WANT_E_U : 0) |
                          (u0 ? ALMPC_WANT_FIRST_INPUT : 0) | (status ? ALMPC_WANT_STATUS : 0) | (iters ? ALMPC_WANT_ITERS : 0) |
                          (polish_iters ? ALMPC_WANT_POLISH_ITERS : 0);
    if (!want) return ALMPC_OK;
    for (size_t i = 0; i < g->hs.size(); ++i)   // (a synchronous look at the results: a lazily deferred redo is settled first, per device)
        if (g->hs[i]->lazy_pending) { const int rc = almpc_synchronize(g->hs[i]); if (rc != ALMPC_OK) return gfail(g, rc, (int)i); }
    for (size_t i = 0; i < g->hs.size(); ++i) {
        g->tickets[i] = almpc_get_results_async(g->hs[i], want);
        if (g->tickets[i] < 0) return gfail(g, g->tickets[i], (int)i);
    }
    for (size_t i = 0; i < g->hs.size(); ++i) {
        const size_t f = (size_t)g->first[i], xs = (size_t)g->n * (g->N + 1), us = (size_t)g->m * g->N;
        const int rc = almpc_get_results_wait(g->hs[i], g->tickets[i], x ? x + f * xs : nullptr, e_x ? e_x + f * xs : nullptr,
                                              u ? u + f * us : nullptr, e_u ? e_u + f * us : nullptr, u0 ? u0 + f * g->m : nullptr,
                                              status ? status + f : nullptr, iters ? iters + f : nullptr,
                                              polish_iters ? polish_iters + f : nullptr);
        if (rc != ALMPC_OK) return gfail(g, rc, (int)i);
    }
    return ALMPC_OK;
}

// Asynchronous read-back of the whole batch: the request goes to every device (pack kernels / copy streams), the ticket is the group's;
// almpc_group_get_results_wait gathers into the caller's arrays (layouts of almpc_group_get_results).  The last two tickets are kept.
int almpc_group_get_results_async(almpc_group* g, uint32_t want) {
    if (!g) return ALMPC_ERR_INVALID;
    const long t = g->next_gticket;
    const int s = (int)(t % 2);
    g->gt[s].assign(g->hs.size(), -1);
    for (size_t i = 0; i < g->hs.size(); ++i) {
        g->gt[s][i] = almpc_get_results_async(g->hs[i], want);
        if (g->gt[s][i] < 0) return gfail(g, g->gt[s][i], (int)i);
    }
    g->gticket[s] = t;
    g->next_gticket += 1;
    return (int)(t & 0x3fffffff);
}
int almpc_group_get_results_wait(almpc_group* g, int ticket, double* x, double* e_x, double* u, double* e_u, double* u0, int32_t* status,
                                 int32_t* iters, int32_t* polish_iters) {
    if (!g || ticket < 0) return ALMPC_ERR_INVALID;
    int s = -1;
    for (int q = 0; q < 2; ++q)
        if (g->gticket[q] >= 0 && (int)(g->gticket[q] & 0x3fffffff) == ticket) s = q;
    if (s < 0) { g->err = "get_results_wait: ticket is not outstanding (only the last two requests are kept)"; return ALMPC_ERR_INVALID; }
    for (size_t i = 0; i < g->hs.size(); ++i) {
        const size_t f = (size_t)g->first[i], xs = (size_t)g->n * (g->N + 1), us = (size_t)g->m * g->N;
        const int rc = almpc_get_results_wait(g->hs[i], g->gt[s][i], x ? x + f * xs : nullptr, e_x ? e_x + f * xs : nullptr,
                                              u ? u + f * us : nullptr, e_u ? e_u + f * us : nullptr, u0 ? u0 + f * g->m : nullptr,
                                              status ? status + f : nullptr, iters ? iters + f : nullptr,
                                              polish_iters ? polish_iters + f : nullptr);
        if (rc != ALMPC_OK) return gfail(g, rc, (int)i);
    }
    return ALMPC_OK;
}

}  // extern "C"
