// Context, memory helpers, scalar read-back and per-class device timers.
#include "common.h"

#define MADQP_MAX_BLOCKS 1024

extern "C" int32_t madqp_version(void) { return 100; }

extern "C" int32_t madqp_ctx_create(int32_t device, void* stream, madqp_ctx** out) {
    if (!out) return MADQP_ERR_ARG;
    *out = nullptr;
    madqp_ctx* ctx = new (std::nothrow) madqp_ctx();
    if (!ctx) return MADQP_ERR_ALLOC;
    ctx->device = device;
    ctx->stream = (hipStream_t)stream;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
            ctx->gemm_slots = 2 * (int64_t)cus;
    }
    if (e == hipSuccess) e = hipMalloc(&ctx->d_res, MADQP_RESULT_SLOTS * sizeof(double));
    if (e == hipSuccess) e = hipMemset(ctx->d_res, 0, MADQP_RESULT_SLOTS * sizeof(double));
    if (e == hipSuccess)
        e = hipHostMalloc((void**)&ctx->h_res, MADQP_RESULT_SLOTS * sizeof(double), hipHostMallocDefault);
    if (e == hipSuccess)
        e = hipMalloc(&ctx->d_part, (size_t)MADQP_MAX_BLOCKS * MADQP_RESULT_SLOTS * sizeof(double));
    if (e != hipSuccess) {
        fprintf(stderr, "madqp_ctx_create: %s\n", hipGetErrorString(e));
        delete ctx;
        return MADQP_ERR_HIP;
    }
    *out = ctx;
    return MADQP_OK;
}

extern "C" int32_t madqp_ctx_destroy(madqp_ctx* ctx) {
    if (!ctx) return MADQP_OK;
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& p : ctx->pending) {
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    for (auto& e : ctx->pool) (void)hipEventDestroy(e);
    madqp_gemm_release_tables(ctx);
    if (ctx->d_res) (void)hipFree(ctx->d_res);
    if (ctx->h_res) (void)hipHostFree(ctx->h_res);
    if (ctx->d_part) (void)hipFree(ctx->d_part);
    if (ctx->d_work) (void)hipFree(ctx->d_work);
    if (ctx->d_scaled) (void)hipFree(ctx->d_scaled);
    if (ctx->d_tickets) (void)hipFree(ctx->d_tickets);
    delete ctx;
    return MADQP_OK;
}

extern "C" const char* madqp_last_error(madqp_ctx* ctx) { return ctx ? ctx->err : "null context"; }

extern "C" int32_t madqp_ctx_sync(madqp_ctx* ctx) {
    ARG_TRY(ctx, ctx != nullptr);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MADQP_OK;
}

extern "C" int32_t madqp_malloc(madqp_ctx* ctx, size_t bytes, void** out) {
    ARG_TRY(ctx, ctx && out);
    *out = nullptr;
    if (bytes == 0) return MADQP_OK;
    hipError_t e = hipMalloc(out, bytes);
    if (e != hipSuccess)
        return madqp_fail(ctx, MADQP_ERR_ALLOC, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return MADQP_OK;
}

extern "C" int32_t madqp_free(madqp_ctx* ctx, void* ptr) {
    ARG_TRY(ctx, ctx != nullptr);
    if (ptr) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipFree(ptr));
    }
    return MADQP_OK;
}

extern "C" int32_t madqp_memcpy_h2d(madqp_ctx* ctx, void* dst, const void* src, size_t bytes) {
    ARG_TRY(ctx, ctx && (bytes == 0 || (dst && src)));
    if (bytes) {
        HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return MADQP_OK;
}

extern "C" int32_t madqp_memcpy_d2h(madqp_ctx* ctx, void* dst, const void* src, size_t bytes) {
    ARG_TRY(ctx, ctx && (bytes == 0 || (dst && src)));
    if (bytes) {
        HIP_TRY(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return MADQP_OK;
}

int32_t madqp_work_reserve(madqp_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->work_bytes) return MADQP_OK;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_work) HIP_TRY(ctx, hipFree(ctx->d_work));
    ctx->d_work = nullptr;
    ctx->work_bytes = 0;
    hipError_t e = hipMalloc(&ctx->d_work, bytes);
    if (e != hipSuccess)
        return madqp_fail(ctx, MADQP_ERR_ALLOC, "workspace hipMalloc(%zu): %s", bytes,
                          hipGetErrorString(e));
    ctx->work_bytes = bytes;
    return MADQP_OK;
}

// Copies the first `count` slots of the device result block to the host (one sync).  The whole block (512 B)
// travels, so the fault word in its last slot is seen by every scalar read-back: a sweep whose producer block never
// published (chol.hip, sweep_poll_block) is reported as MADQP_ERR_HIP here instead of surfacing as quiet NaNs that
// the loop would take for a numerical failure (src/linear_solver.jl:41-43).
int32_t madqp_read_results(madqp_ctx* ctx, int count, double* out_host) {
    ARG_TRY(ctx, count >= 0 && count < MADQP_FAULT_SLOT && out_host);
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_res, ctx->d_res, MADQP_RESULT_SLOTS * sizeof(double), hipMemcpyDeviceToHost,
                                ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(out_host, ctx->h_res, count * sizeof(double));
    if (ctx->h_res[MADQP_FAULT_SLOT] != 0.0) {
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_res + MADQP_FAULT_SLOT, 0, sizeof(double), ctx->stream));
        return madqp_fail(ctx, MADQP_ERR_HIP,
                          "triangular sweep hand-off timed out: a producer block never published its solution "
                          "(device fault word set; results of the last solve are invalid)");
    }
    return MADQP_OK;
}

// The same in two halves: the copy and an event behind it are queued, kernels queued after them run on while the host
// waits for the event only (mpc.hip).  h_dst: pinned, MADQP_RESULT_SLOTS doubles.
int32_t madqp_results_post(madqp_ctx* ctx, double* h_dst, hipEvent_t ev) {
    ARG_TRY(ctx, h_dst && ev);
    HIP_TRY(ctx, hipMemcpyAsync(h_dst, ctx->d_res, MADQP_RESULT_SLOTS * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipEventRecord(ev, ctx->stream));
    return MADQP_OK;
}
int32_t madqp_results_wait(madqp_ctx* ctx, const double* h_src, hipEvent_t ev, int count, double* out_host) {
    ARG_TRY(ctx, count >= 0 && count < MADQP_FAULT_SLOT && out_host && h_src && ev);
    HIP_TRY(ctx, hipEventSynchronize(ev));
    memcpy(out_host, h_src, count * sizeof(double));
    if (h_src[MADQP_FAULT_SLOT] != 0.0) {
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_res + MADQP_FAULT_SLOT, 0, sizeof(double), ctx->stream));
        return madqp_fail(ctx, MADQP_ERR_HIP,
                          "triangular sweep hand-off timed out: a producer block never published its solution "
                          "(device fault word set; results of the last solve are invalid)");
    }
    return MADQP_OK;
}

// Test hook: sets the device fault word as a timed-out sweep would.
extern "C" int32_t madqp_debug_inject_fault(madqp_ctx* ctx) {
    ARG_TRY(ctx, ctx != nullptr);
    const double one = 1.0;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_res + MADQP_FAULT_SLOT, &one, sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MADQP_OK;
}

// ---------------------------------------------------------------- profiling
static hipEvent_t prof_event(madqp_ctx* ctx) {
    if (!ctx->pool.empty()) {
        hipEvent_t e = ctx->pool.back();
        ctx->pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

void madqp_prof_begin(madqp_ctx* ctx, int cls) {
    ProfEvent p;
    p.cls = cls;
    p.a = prof_event(ctx);
    p.b = prof_event(ctx);
    (void)hipEventRecord(p.a, ctx->stream);
    ctx->pending.push_back(p);
}

void madqp_prof_end(madqp_ctx* ctx) {
    if (ctx->pending.empty()) return;
    (void)hipEventRecord(ctx->pending.back().b, ctx->stream);
}

static int32_t prof_drain(madqp_ctx* ctx) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (auto& p : ctx->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            ctx->prof_ms[p.cls] += ms;
            ctx->prof_n[p.cls] += 1;
        }
        ctx->pool.push_back(p.a);
        ctx->pool.push_back(p.b);
    }
    ctx->pending.clear();
    return MADQP_OK;
}

extern "C" int32_t madqp_prof_enable(madqp_ctx* ctx, int32_t mask) {
    ARG_TRY(ctx, ctx != nullptr);
    if (!mask && ctx->prof) {
        int32_t r = prof_drain(ctx);
        if (r) return r;
    }
    ctx->prof = (uint32_t)mask;
    return MADQP_OK;
}

extern "C" int32_t madqp_prof_reset(madqp_ctx* ctx) {
    ARG_TRY(ctx, ctx != nullptr);
    int32_t r = prof_drain(ctx);
    if (r) return r;
    for (int i = 0; i < MADQP_PROF_COUNT; ++i) {
        ctx->prof_ms[i] = 0;
        ctx->prof_n[i] = 0;
    }
    return MADQP_OK;
}

extern "C" int32_t madqp_prof_get(madqp_ctx* ctx, int32_t cls, double* ms, int64_t* launches) {
    ARG_TRY(ctx, ctx && cls >= 0 && cls < MADQP_PROF_COUNT && ms && launches);
    int32_t r = prof_drain(ctx);
    if (r) return r;
    *ms = ctx->prof_ms[cls];
    *launches = ctx->prof_n[cls];
    return MADQP_OK;
}
