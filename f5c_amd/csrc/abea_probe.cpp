/* abea_probe.cpp — abea_link_probe: what the host<->device link of this context's GPU delivers, measured the way the pipelines
 * use it.  The raw-signal entries (abea_chain.cpp) move 2 bytes per sample up and the 24-byte event_t tables down: whether
 * event_db (src/f5c.c:682-734) is bound by the link, by the detector's kernels or by the host loops can only be said against the
 * link's own ceiling on THIS box, not against a data-sheet number.  Pinned host memory, device memory = the context's arena,
 * HIP events on the streams the copies run on.  Diagnostic only: nothing in the library depends on the result. */
#include "abea_internal.h"

namespace {

struct probe_res {
    void* h_up = nullptr; void* h_dn = nullptr;
    hipStream_t s_up = nullptr, s_dn = nullptr;
    hipEvent_t e[4] = {nullptr, nullptr, nullptr, nullptr};
    ~probe_res() {
        for (hipEvent_t x : e) if (x) hipEventDestroy(x);
        if (s_up) hipStreamDestroy(s_up);
        if (s_dn) hipStreamDestroy(s_dn);
        if (h_up) hipHostFree(h_up);
        if (h_dn) hipHostFree(h_dn);
    }
};

}  // namespace

extern "C" int abea_link_probe(abea_ctx* c, uint64_t bytes, int32_t reps, double* out, int32_t n_out) {
    if (!c || !out || n_out < 7) return abea_fail(ABEA_EINVAL, "abea_link_probe: null argument or fewer than 7 outputs");
    if (!c->children.empty()) c = c->children[0];
    ABEA_API_ENTER(c, "abea_link_probe");
    HIP_TRY(hipSetDevice(c->device));
    if (bytes == 0) bytes = (uint64_t)256 << 20;
    if (reps <= 0) reps = 4;
    bytes = bytes / 4096 * 4096;
    if (bytes < 4096 || 2 * bytes + 4096 > c->arena_bytes) return abea_fail(ABEA_EINVAL, "abea_link_probe: %llu bytes do not fit the arena twice", (unsigned long long)bytes);
    probe_res R;
    HIP_TRY(hipHostMalloc(&R.h_up, bytes, hipHostMallocDefault));
    HIP_TRY(hipHostMalloc(&R.h_dn, bytes, hipHostMallocDefault));
    memset(R.h_up, 1, bytes); memset(R.h_dn, 0, bytes);                 /* touch every page before anything is timed */
    HIP_TRY(hipStreamCreateWithFlags(&R.s_up, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&R.s_dn, hipStreamNonBlocking));
    for (hipEvent_t& x : R.e) HIP_TRY(hipEventCreate(&x));
    uint8_t* d_up = c->arena;
    uint8_t* d_dn = c->arena + bytes;
    const size_t n16 = bytes / 16;
    const dim3 grid((unsigned)std::min<size_t>((size_t)abea_copy_kernel_blocks(), (n16 + 255) / 256));
    auto up = [&]() { return hipMemcpyAsync(d_up, R.h_up, bytes, hipMemcpyHostToDevice, R.s_up); };
    auto dn_sdma = [&]() { return hipMemcpyAsync(R.h_dn, d_dn, bytes, hipMemcpyDeviceToHost, R.s_dn); };
    auto up_kernel = [&]() {                             /* the same kernel, loading from pinned host memory */
        hipLaunchKernelGGL(abea_copy_out_kernel, grid, dim3(256), 0, R.s_up, (const uint4*)R.h_up, (uint4*)d_up, n16);
        return hipGetLastError();
    };
    auto dn_kernel = [&]() {
        hipLaunchKernelGGL(abea_copy_out_kernel, grid, dim3(256), 0, R.s_dn, (const uint4*)d_dn, (uint4*)R.h_dn, n16);
        return hipGetLastError();
    };
    /* one direction at a time, then both at once; rate = bytes x reps / the stream's own event span */
    auto run = [&](int mode_up, int mode_dn, double* r_up, double* r_dn) -> int {
        for (int warm = 0; warm < 2; ++warm) {
            if (mode_up) HIP_TRY(hipEventRecord(R.e[0], R.s_up));
            if (mode_dn) HIP_TRY(hipEventRecord(R.e[2], R.s_dn));
            for (int r = 0; r < (warm ? reps : 1); ++r) {
                if (mode_up == 1) HIP_TRY(up());
                if (mode_up == 2) HIP_TRY(up_kernel());
                if (mode_dn == 1) HIP_TRY(dn_sdma());
                if (mode_dn == 2) HIP_TRY(dn_kernel());
            }
            if (mode_up) HIP_TRY(hipEventRecord(R.e[1], R.s_up));
            if (mode_dn) HIP_TRY(hipEventRecord(R.e[3], R.s_dn));
            HIP_TRY(hipStreamSynchronize(R.s_up));
            HIP_TRY(hipStreamSynchronize(R.s_dn));
        }
        float ms = 0;
        if (mode_up) { HIP_TRY(hipEventElapsedTime(&ms, R.e[0], R.e[1])); *r_up = (double)bytes * reps / (ms * 1e-3) / 1e9; }
        if (mode_dn) { HIP_TRY(hipEventElapsedTime(&ms, R.e[2], R.e[3])); *r_dn = (double)bytes * reps / (ms * 1e-3) / 1e9; }
        return ABEA_OK;
    };
    double dummy = 0;
    for (int i = 0; i < n_out; ++i) out[i] = 0;
    int rc = run(1, 0, &out[0], &dummy);                 /* h2d, hipMemcpyAsync (SDMA engine) */
    if (!rc) rc = run(0, 1, &dummy, &out[1]);            /* d2h, hipMemcpyAsync */
    if (!rc) rc = run(0, 2, &dummy, &out[2]);            /* d2h, abea_copy_out_kernel (stores straight into pinned host memory) */
    if (!rc) rc = run(1, 2, &out[3], &out[4]);           /* both at once: h2d by copy engine, d2h by kernel (what the pipelines do) */
    if (!rc) rc = run(1, 1, &out[5], &out[6]);           /* both at once, both by copy engine */
    if (!rc && n_out >= 10) {
        rc = run(2, 0, &out[7], &dummy);                 /* h2d by kernel (loads from pinned host memory) */
        if (!rc) rc = run(2, 1, &out[8], &out[9]);       /* both at once: h2d by kernel, d2h by copy engine */
    }
    return rc;
}
