// Host engine: context / workspace management and the C-ABI entry points of the atomic-contact hot path
// (include/arpeggia_amd.h).  Compiled with hipcc for the HIP runtime API; all device code lives in kernels.hip.
// There is NO CPU compute path here: without a gfx950 device every compute call returns ARP_ERR_NO_DEVICE.
#include <cctype>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <memory>
#include <mutex>
#include <new>
#include <numeric>
#include <string>
#include <sys/mman.h>
#include <system_error>
#include <thread>
#include <unordered_map>
#include <vector>

#include "arp_internal.h"
#include "host_common.h"
#include "table_dev.h"

namespace arp {

thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) {                                                                    \
            set_error("HIP error %d (%s) at %s:%d: %s", (int)e_, hipGetErrorString(e_), __FILE__, __LINE__, #expr); \
            return (e_ == hipErrorOutOfMemory) ? ARP_ERR_OOM : ARP_ERR_HIP;                        \
        }                                                                                          \
    } while (0)

// ---- exact squared-distance decision bounds (see DevParams) ------------------------------------------------
// min{ s >= 0 : sqrt(s) >= T }  so that  (sqrt(s) < T)  <=>  (s < bound_lt(T)).  Host sqrt is correctly rounded.
double bound_lt(double T) {
    if (!(T > 0.0)) return 0.0;  // T <= 0 or NaN: d < T never holds for d >= 0
    if (std::isinf(T)) return INFINITY;
    double s = T * T;
    if (!std::isfinite(s)) s = DBL_MAX;
    while (s > 0.0 && std::sqrt(s) >= T) s = std::nextafter(s, 0.0);
    while (std::sqrt(s) < T) s = std::nextafter(s, INFINITY);
    return s;
}
// min{ s >= 0 : sqrt(s) > T }  so that  (sqrt(s) <= T)  <=>  (s < bound_le(T)).
double bound_le(double T) {
    if (!(T >= 0.0)) return 0.0;
    if (std::isinf(T)) return INFINITY;
    double s = T * T;
    if (!std::isfinite(s)) s = DBL_MAX;
    while (s > 0.0 && std::sqrt(s) > T) s = std::nextafter(s, 0.0);
    while (std::sqrt(s) <= T) s = std::nextafter(s, INFINITY);
    return s;
}

void make_dev_params(const arp_params &p, DevParams *d) {
    memset(d, 0, sizeof *d);
    const double c = p.vdw_comp;
    d->r2 = d->r2_call = p.dist_cutoff * p.dist_cutoff;  // complex.rs:191 (r2 itself is rewritten by every call's grid sizing)
    d->s_ion = bound_le(4.0);
    d->s_polar = bound_le(3.5);
    d->s_hphob = bound_le(4.5);
    for (int a = 0; a < 16; a++) {
        for (int b = 0; b < 16; b++) {
            double sum_cov = p.cov_radius[a] + p.cov_radius[b];  // vdw.rs:27
            double sum_vdw = p.vdw_radius[a] + p.vdw_radius[b];  // vdw.rs:28
            d->s_clash[a * 16 + b] = bound_lt(sum_cov - c);
            d->s_cov[a * 16 + b] = bound_lt(sum_cov + c);
            d->s_vdw[a * 16 + b] = bound_lt(sum_vdw + c);
            // vdw.rs:32-43 is a first-match chain (clash, covalent, vdW); k_emit counts how many of the three bounds a distance is below,
            // which is the same thing only for NESTED bounds.  A negative vdw_comp (cov - c > cov + c) or caller radii with vdw < cov break the
            // nesting; raising each bound to its predecessor restores it without changing any first-match outcome: below the clash bound
            // the pair is a clash whatever the others say, and a distance at or above it is below max(cov, clash) exactly when it is below cov.
            d->s_cov[a * 16 + b] = std::max(d->s_cov[a * 16 + b], d->s_clash[a * 16 + b]);
            d->s_vdw[a * 16 + b] = std::max(d->s_vdw[a * 16 + b], d->s_cov[a * 16 + b]);
        }
        d->s_hacc[a] = bound_le(p.h_vdw_radius + p.vdw_radius[a] + c);  // hbond.rs:54
    }
    d->s_cov_max = 0.0;
    for (int k = 0; k < 256; k++) d->s_cov_max = std::max(d->s_cov_max, d->s_cov[k]);  // (s_cov >= s_clash after the clamp above)
    d->r2f = (float)d->r2;
    d->strip_force = g_debug.strip_rows > 0 ? (uint32_t)g_debug.strip_rows : 0u;
    d->flags = p.flags;
}

}  // namespace arp

using namespace arp;

// ---- context -------------------------------------------------------------------------------------------------
struct arp_context {
    int device = 0;
    hipStream_t own_stream = nullptr, stream = nullptr;
    Workspace ws{};
    std::vector<void *> ws_allocs;
    // device staging of host inputs
    // host inputs travel as ONE block: the twelve arrays are packed into a pinned buffer and cross PCIe in a single copy
    // (twelve small pageable copies cost ~100 us of launch overhead on a PDB-sized structure)
    struct Staged {
        char *dev = nullptr, *pinned = nullptr;
        uint64_t bytes = 0;
    } st;
    arp_pair *out_buf = nullptr;            // reusable device output of the host-output path (grow-only)
    char *bounce[2] = {nullptr, nullptr};   // pinned staging of large device -> host copies
    hipEvent_t bounce_ev[2] = {nullptr, nullptr};
    uint64_t out_cap = 0;
    arp_pair *grp_buf = nullptr;            // batch path: the pack's pair list grouped by member (device); it lands in a SharedBlock on the host
    uint64_t grp_cap = 0;
    unsigned long long *h_offsets = nullptr;  // pinned: per-member offsets into the grouped list (+ the pack status word)
    uint64_t h_offsets_cap = 0;
    DevParams *h_params = nullptr;         // pinned
    unsigned long long *h_result = nullptr;  // pinned [kResultWords]: pairs, status flags, emit allocator head, deferred-list chunks, residue-run sample
    // Residue-rule memo: the last input's residues were runs of atoms (k_place's sample, result[4]) -- the next call's launcher then picks the
    // kernels that apply the reference's residue rule before the gathers (k_emit<.., RES>).  Same result either way; ARP_FLAG_RESIDUE_RUNS /
    // ARP_FLAG_NO_RESIDUE_RUNS overrule the memo.
    bool res_hint = false;
    bool rkey_valid = false;               // the workspace's residue words (Sorted::rkey) belong to the cell list that was built last
    // Deferred-pass memo: the arrays (address + length) of the last single-pass call that deferred NOTHING to the probe pass (no hydrogens,
    // no CYS SG pair in the covalent band -- every X-ray structure without hydrogens).  The next call on the same arrays does not launch
    // k_pairs_deferred; should it defer after all (the caller rewrote the arrays), k_fixup raises status bit 128 and the call is repeated
    // with the pass.  A guess that is checked on the device, never a correctness assumption.
    const double *nodefer_x = nullptr; uint64_t nodefer_n = 0;
    bool last_skip = false;
    bool last_direct = false;  // the enqueued call ran the hole-free sequence of small inputs (launch_emit): finish_result derives what k_fixup would have published
    arp_params last_params{};
    bool have_params = false;
    DevParams *params_on_device = nullptr;  // the workspace block that holds the current parameters (upload_params); reset with the workspace
    hipStream_t params_stream = nullptr;    // ... uploaded on this stream (a caller who swaps streams gets a fresh upload, ordered on the new one)
    hipEvent_t params_ev = nullptr;         // recorded behind the last upload of h_params: the block is rewritten only after that copy has run
    bool params_ev_armed = false;
    uint64_t last_capacity = 0;
    bool pending = false;
    char *scr_dev[2] = {nullptr, nullptr}, *scr_pin[2] = {nullptr, nullptr};  // table path: two grow-only scratch blocks (device / pinned)
    uint64_t scr_dev_cap[2] = {0, 0}, scr_pin_cap[2] = {0, 0};
    arp_context *peer = nullptr;           // batch path: the second context of this device (own stream + workspace), kept across calls
    uint32_t defer_scale = 1;              // the deferred-probe list is sized defer_scale x the default; grown on overflow
    const double *grid_x = nullptr; uint64_t grid_n = 0;  // the arrays the workspace's cell list was last built from (context_grid)
    arp_atoms last_atoms{};                // the enqueued call, kept so that arp_contacts_atomic_result can re-run it after growing a list
    arp_pair *last_out = nullptr;
    Profiler prof;
};

constexpr size_t kResultWords = 5;
constexpr unsigned long long kResRunsMin = 64;  // of the 255 atoms k_place samples

// Builds the cell list of a call.  Decides whether the pair pass will run the residue-rule kernels (then k_place also writes the residue words).
static void grid_for_call(arp_context *ctx, const DevAtoms &d, const arp_params *params, Profiler *prof, bool ordered) {
    bool res = !ordered && emit_takes_res_filter(d);
    if (res) res = (params->flags & ARP_FLAG_RESIDUE_RUNS) ? true : ((params->flags & ARP_FLAG_NO_RESIDUE_RUNS) ? false : ctx->res_hint);
    launch_grid(d, ctx->ws, ctx->stream, prof, params->dist_cutoff, ordered, res);
    ctx->rkey_valid = res;
}
static void note_residue_runs(arp_context *ctx) { ctx->res_hint = ctx->h_result[4] >= kResRunsMin; }

static arp_status check_device(arp_context *ctx) {
    if (!ctx) { set_error("null context"); return ARP_ERR_BAD_INPUT; }
    HIP_TRY(hipSetDevice(ctx->device));
    return ARP_OK;
}

template <typename T>
static arp_status dev_alloc(arp_context *ctx, T **p, size_t count) {
    void *q = nullptr;
    HIP_TRY(hipMalloc(&q, std::max<size_t>(count, 1) * sizeof(T)));
    ctx->ws_allocs.push_back(q);
    *p = (T *)q;
    return ARP_OK;
}

static void free_workspace(arp_context *ctx) {
    ctx->grid_x = nullptr; ctx->grid_n = 0; ctx->params_on_device = nullptr;
    for (void *p : ctx->ws_allocs) (void)hipFree(p);
    ctx->ws_allocs.clear();
    ctx->ws = Workspace{};
}

static arp_status ensure_workspace(arp_context *ctx, uint64_t n) {
    Workspace &w = ctx->ws;
    if (w.n_cap >= n && w.grid) return ARP_OK;
    if (n >= 0xFFFFFFF0ull) { set_error("too many atoms for 32-bit indices"); return ARP_ERR_BAD_INPUT; }
    (void)hipStreamSynchronize(ctx->stream);
    free_workspace(ctx);
    uint64_t cap = std::max<uint64_t>(n + n / 8, 1024);
    uint64_t ccap = std::min<uint64_t>(8 * cap + 65536, 0xFFFFFFF0ull);
    arp_status s;
#define A(ptr, cnt) if ((s = dev_alloc(ctx, &(ptr), (cnt))) != ARP_OK) { free_workspace(ctx); return s; }
    A(w.partials, 1024 * 8); A(w.tickets, 4); A(w.grid, 1); A(w.params, 1);
    A(w.cell_of_atom, cap); A(w.rank_of_atom, cap); A(w.cell_count, ccap + 1); A(w.cell_start, ccap + 1);
    A(w.perm, cap); A(w.slot_cell, cap);
    A(w.sorted.rec, cap + 64); A(w.sorted.fat, cap + 64); A(w.sorted.rkey, cap + 64);
    A(w.task_count, cap / 64 + 2); A(w.task_base, cap / 64 + 2);
    A(w.scan_tmp, 1024 + 1); A(w.scan_tmp64, 1024 + 1); A(w.result, 32 + 1024);  // scan_tmp*: >= kScanBlocks + 1; result: 32 words + the kScanBlocks chunk totals of k_scan_single
    A(w.hole_list, 2048); A(w.task_ctr, kTaskCtrWords); w.scratch_cap = emit_scratch_records(); A(w.scratch, w.scratch_cap);
    A(w.model_box, 65536u * 6u); A(w.model_org, 65536u * 6u);
    w.defer_cap = (uint64_t)ctx->defer_scale * std::max<uint64_t>(16 * cap, 1u << 20) + (1u << 20);  // + one partly used 512-entry chunk per block
    if (g_debug.defer_entries > 0) w.defer_cap = (uint64_t)ctx->defer_scale * (uint64_t)g_debug.defer_entries;  // tests: a tiny list, so that the grow-and-repeat path runs
    A(w.defer_list, w.defer_cap);
#undef A
    w.n_cap = (uint32_t)cap;
    w.ncells_cap = (uint32_t)ccap;
    {   // Every pointer a kernel may dereference must exist before the first launch.  (Round 1 recorded one GPU fault "on address
        // (nil)": an intermediate build launched k_bounds with the then-new `partials` member not yet allocated here.  A member
        // added to Workspace without its allocation now fails this check on the host instead of faulting on the device.)
        const void *members[] = {w.partials, w.tickets, w.grid, w.params, w.cell_of_atom, w.rank_of_atom, w.cell_count, w.cell_start, w.perm, w.slot_cell,
                                 w.sorted.rec, w.sorted.fat, w.sorted.rkey, w.task_count, w.task_base, w.scan_tmp, w.scan_tmp64, w.result, w.hole_list, w.scratch,
                                 w.task_ctr, w.defer_list, w.model_box, w.model_org};
        for (const void *m : members)
            if (!m) { free_workspace(ctx); set_error("internal error: a workspace member was not allocated"); return ARP_ERR_HIP; }
    }
    // self-cleaning state: the kernels leave these zeroed for the next call
    HIP_TRY(hipMemsetAsync(w.cell_count, 0, (ccap + 1) * sizeof(uint32_t), ctx->stream));
    HIP_TRY(hipMemsetAsync(w.tickets, 0, 4 * sizeof(uint32_t), ctx->stream));
    return ARP_OK;
}

extern "C" arp_status arp_context_create(int32_t device, arp_context **out) try {
    if (!out) { set_error("null out"); return ARP_ERR_BAD_INPUT; }
    *out = nullptr;
    int cnt = arp_device_count();
    if (cnt <= 0) { set_error("no gfx950 (MI355X) device visible; this engine has no CPU fallback"); return ARP_ERR_NO_DEVICE; }
    if (device < 0 || device >= cnt) { set_error("device %d out of range (0..%d)", device, cnt - 1); return ARP_ERR_NO_DEVICE; }
    arp_context *ctx = new arp_context();
    ctx->device = device;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipHostMalloc((void **)&ctx->h_params, sizeof(DevParams), hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void **)&ctx->h_result, 8 * sizeof(unsigned long long), hipHostMallocDefault);
    if (e != hipSuccess) {
        set_error("HIP error %d (%s) creating the context", (int)e, hipGetErrorString(e));
        arp_context_destroy(ctx);
        return ARP_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    *out = ctx;
    return ARP_OK;
} ARP_ABI_CATCH

static void free_staged(arp_context *ctx) {
    auto &s = ctx->st;
    if (ctx->nodefer_x == (const double *)s.dev) { ctx->nodefer_x = nullptr; ctx->nodefer_n = 0; }  // (a memo must not outlive the buffer it names)
    if (s.dev) (void)hipFree(s.dev);
    if (s.pinned) (void)hipHostFree(s.pinned);
    s = arp_context::Staged{};
}

extern "C" void arp_context_destroy(arp_context *ctx) {
    if (!ctx) return;
    if (ctx->peer) { arp_context_destroy(ctx->peer); ctx->peer = nullptr; }
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    free_workspace(ctx);
    free_staged(ctx);
    if (ctx->out_buf) (void)hipFree(ctx->out_buf);
    for (int k = 0; k < 2; k++) { if (ctx->scr_dev[k]) (void)hipFree(ctx->scr_dev[k]); if (ctx->scr_pin[k]) (void)hipHostFree(ctx->scr_pin[k]); }
    if (ctx->grp_buf) (void)hipFree(ctx->grp_buf);
    if (ctx->h_offsets) (void)hipHostFree(ctx->h_offsets);
    for (int k = 0; k < 2; k++) { if (ctx->bounce[k]) (void)hipHostFree(ctx->bounce[k]); if (ctx->bounce_ev[k]) (void)hipEventDestroy(ctx->bounce_ev[k]); }
    if (ctx->params_ev) (void)hipEventDestroy(ctx->params_ev);
    if (ctx->h_params) (void)hipHostFree(ctx->h_params);
    if (ctx->h_result) (void)hipHostFree(ctx->h_result);
    if (ctx->prof.created) for (int k = 0; k < Profiler::kMax; k++) { (void)hipEventDestroy(ctx->prof.ev0[k]); (void)hipEventDestroy(ctx->prof.ev1[k]); }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

extern "C" arp_status arp_context_set_stream(arp_context *ctx, void *hip_stream) {
    arp_status s = check_device(ctx);
    if (s != ARP_OK) return s;
    (void)hipStreamSynchronize(ctx->stream);
    ctx->stream = (hipStream_t)hip_stream;
    return ARP_OK;
}

extern "C" arp_status arp_context_synchronize(arp_context *ctx) {
    arp_status s = check_device(ctx);
    if (s != ARP_OK) return s;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return ARP_OK;
}

// ---- input handling -------------------------------------------------------------------------------------------
static arp_status validate(const arp_atoms *a, const arp_params *p) {
    if (!a || !p) { set_error("null atoms/params"); return ARP_ERR_BAD_INPUT; }
    if (a->n >= 0xFFFFFFF0ull) { set_error("too many atoms"); return ARP_ERR_BAD_INPUT; }
    if (a->n && (!a->x || !a->y || !a->z || !a->attr || !a->res_ord || !a->chain_rank || !a->model)) { set_error("null atom array"); return ARP_ERR_BAD_INPUT; }
    if (a->n_res && (!a->res_id || !a->res_h_ptr || !a->res_cb || !a->res_sg)) { set_error("null residue table"); return ARP_ERR_BAD_INPUT; }
    if (a->n_res >= 0xFFFFFFF0ull) { set_error("too many residues"); return ARP_ERR_BAD_INPUT; }
    if (a->location != ARP_MEM_HOST && a->location != ARP_MEM_DEVICE) { set_error("bad location"); return ARP_ERR_BAD_INPUT; }
    if (std::isnan(p->dist_cutoff) || std::isnan(p->vdw_comp)) { set_error("NaN parameter"); return ARP_ERR_BAD_INPUT; }
    return ARP_OK;
}

static arp_status stage_inputs(arp_context *ctx, const arp_atoms *a, DevAtoms *d) {
    d->n = (uint32_t)a->n;
    d->n_res = (uint32_t)a->n_res;
    if (a->location == ARP_MEM_DEVICE) {
        d->x = a->x; d->y = a->y; d->z = a->z; d->attr = a->attr; d->res_ord = a->res_ord; d->chain_rank = a->chain_rank; d->model = a->model;
        d->res_id = a->res_id; d->res_h_ptr = a->res_h_ptr; d->res_h_idx = a->res_h_idx; d->res_cb = a->res_cb; d->res_sg = a->res_sg;
        return ARP_OK;
    }
    auto &s = ctx->st;
    uint64_t nh = 0;
    if (a->n_res) nh = a->res_h_ptr[a->n_res];
    if (nh && !a->res_h_idx) { set_error("null res_h_idx"); return ARP_ERR_BAD_INPUT; }
    // layout of the block: 256-byte aligned segments
    struct Seg { const void *src; uint64_t bytes, off; };
    const uint64_t n = a->n, nr = a->n_res;
    Seg seg[12] = {{a->x, n * 8, 0}, {a->y, n * 8, 0}, {a->z, n * 8, 0}, {a->attr, n * 4, 0}, {a->res_ord, n * 4, 0}, {a->chain_rank, n * 4, 0},
                   {a->model, n * 4, 0}, {nr ? a->res_id : nullptr, nr ? n * 4 : 0, 0}, {nr ? a->res_h_ptr : nullptr, nr ? (nr + 1) * 4 : 0, 0},
                   {nr ? a->res_cb : nullptr, nr * 4, 0}, {nr ? a->res_sg : nullptr, nr * 4, 0}, {nh ? a->res_h_idx : nullptr, nh * 4, 0}};
    uint64_t total = 0;
    for (Seg &g : seg) { g.off = total; total += (g.bytes + 255u) & ~255ull; }
    total = std::max<uint64_t>(total, 256);
    if (s.bytes < total) {
        (void)hipStreamSynchronize(ctx->stream);
        free_staged(ctx);
        const uint64_t cap = total + total / 8;
        HIP_TRY(hipMalloc((void **)&s.dev, cap));
        HIP_TRY(hipHostMalloc((void **)&s.pinned, cap, hipHostMallocDefault));
        s.bytes = cap;
    }
    for (const Seg &g : seg) if (g.bytes) memcpy(s.pinned + g.off, g.src, g.bytes);
    HIP_TRY(hipMemcpyAsync(s.dev, s.pinned, total, hipMemcpyHostToDevice, ctx->stream));
    // The deferred-pass memo identifies an input by (x pointer, n): for host inputs that pointer is THIS staging buffer, whatever structure it
    // carries -- every host call of the same size would hit, and a stale hit costs a whole second pass.  Host inputs never use the memo.
    ctx->nodefer_x = nullptr; ctx->nodefer_n = 0;
    auto at = [&](int k) -> const void * { return s.dev + seg[k].off; };
    d->x = (const double *)at(0); d->y = (const double *)at(1); d->z = (const double *)at(2);
    d->attr = (const uint32_t *)at(3); d->res_ord = (const uint32_t *)at(4);
    d->chain_rank = (const uint32_t *)at(5); d->model = (const uint32_t *)at(6);
    d->res_id = (const uint32_t *)at(7); d->res_h_ptr = (const uint32_t *)at(8); d->res_cb = (const uint32_t *)at(9); d->res_sg = (const uint32_t *)at(10);
    d->res_h_idx = (const uint32_t *)at(11);
    return ARP_OK;
}

static arp_status upload_params(arp_context *ctx, const arp_params *p) {
    // The device copy is uploaded when the parameters (or the workspace it lives in) change, not per call: the three fields a call derives
    // from its input (DevParams::r2, r2f, s_cov_max) are rewritten by every call's grid sizing from fields that nothing on the device writes.
    if (!ctx->have_params || memcmp(&ctx->last_params, p, sizeof *p) != 0) {
        // the pinned block may still be in flight from the previous upload: wait for THAT copy (an event behind it), not for the stream -- a batch
        // whose flags differ from the previous call's used to drain the 52 MB input copy it had just queued (~1 ms per pack, profiles/r05_experiments.txt)
        if (ctx->params_ev_armed) HIP_TRY(hipEventSynchronize(ctx->params_ev));
        make_dev_params(*p, ctx->h_params);
        ctx->last_params = *p;
        ctx->have_params = true;
        ctx->params_on_device = nullptr;
    }
    if (ctx->params_on_device != ctx->ws.params || ctx->params_stream != ctx->stream) {
        HIP_TRY(hipMemcpyAsync(ctx->ws.params, ctx->h_params, sizeof(DevParams), hipMemcpyHostToDevice, ctx->stream));
        if (!ctx->params_ev) HIP_TRY(hipEventCreateWithFlags(&ctx->params_ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(ctx->params_ev, ctx->stream));
        ctx->params_ev_armed = true;
        ctx->params_on_device = ctx->ws.params; ctx->params_stream = ctx->stream;
    }
    return ARP_OK;
}

// After the four result words of a single-pass call have arrived.  The hole-free sequence of small inputs (launch_emit returned true) has no
// fix-up kernel to publish the pair count and the flags that depend on it: the records lie back to back from position 0 and result[2] counts
// them; its probes run inline, so result[3] (the chunks of a deferred list) stays 0 (the input-error flags were set by the grid sizing).
static void finish_result(arp_context *ctx, bool direct, bool skipped, unsigned long long capacity) {
    if (!direct) return;
    unsigned long long *r = ctx->h_result;
    r[0] = r[2];
    if (r[0] > capacity) r[1] |= 1ull;                // the list did not fit (k_fixup: P > capacity)
    if (skipped && r[3] != 0ull) r[1] |= 128ull;      // the probe pass was skipped on a memo that no longer holds
}

constexpr arp_status kRetryDefer = -1;      // internal: never crosses the C ABI
constexpr arp_status kRetryDeferPass = -2;  // internal: the deferred pass was skipped on a memo that no longer holds
static bool skip_deferred_pass(arp_context *ctx, const DevAtoms &d) { return d.x != nullptr && ctx->nodefer_x == d.x && ctx->nodefer_n == d.n; }
// after the results of a single-pass (emit) call have been read into h_result
static void note_deferred(arp_context *ctx, const DevAtoms &d, bool skipped) {
    if (ctx->st.dev && d.x == (const double *)ctx->st.dev) return;  // host input staged by the context: no memo (stage_inputs)
    if (skipped) { if (ctx->h_result[1] & 128ull) { ctx->nodefer_x = nullptr; ctx->nodefer_n = 0; } return; }
    if (ctx->h_result[3] == 0ull && !(ctx->h_result[1] & ~1ull)) { ctx->nodefer_x = d.x; ctx->nodefer_n = d.n; }
    else if (ctx->nodefer_x == d.x) { ctx->nodefer_x = nullptr; ctx->nodefer_n = 0; }
}
static arp_status flags_to_status(unsigned long long flags) {
    if (flags & 128ull) return kRetryDeferPass;
    if (flags & 4ull) { set_error("non-finite atom coordinate"); return ARP_ERR_BAD_INPUT; }
    if (flags & 64ull) { set_error("model ordinals must be dense: the largest model id exceeds what the workspace of this input holds (model ids count 0, 1, 2, ...)"); return ARP_ERR_BAD_INPUT; }
    if (flags & 16ull) { set_error("internal error: inconsistent hole plan in k_fixup"); return ARP_ERR_HIP; }
    if (flags & 2ull) { set_error("CYS SG..SG covalent pair whose residue has no CB (the reference panics in is_disulfide, vdw.rs:58)"); return ARP_ERR_BAD_INPUT; }
    if (flags & 8ull) return kRetryDefer;  // the deferred-probe list overflowed: the caller grows it and repeats the pass
    return ARP_OK;
}

// ---- device -> host of a large pair list ------------------------------------------------------------------------------
// hipMemcpy into freshly malloc'd pageable memory runs at ~2 GB/s (page faults + the runtime's staging); a 460 MB list took
// 230 ms.  Here the pages are populated by helper threads (MADV_POPULATE_WRITE, huge pages where the kernel grants them)
// while the list streams through two pinned 16 MB bounce buffers.
static arp_pair *download_pairs(arp_context *ctx, const arp_pair *dev, unsigned long long total, arp_status *status) {
    const size_t bytes = (size_t)total * sizeof(arp_pair);
    *status = ARP_OK;
    constexpr size_t kBounce = 16u << 20;
    if (bytes < 4 * kBounce) {
        arp_pair *host = (arp_pair *)malloc(bytes);
        if (!host) { set_error("out of host memory"); *status = ARP_ERR_OOM; return nullptr; }
        hipError_t e = hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { free(host); set_error("HIP error %d copying pairs to the host", (int)e); *status = ARP_ERR_HIP; return nullptr; }
        return host;
    }
    const size_t kHuge = 2u << 20;
    char *host = (char *)aligned_alloc(kHuge, (bytes + kHuge - 1) / kHuge * kHuge);
    if (!host) { set_error("out of host memory"); *status = ARP_ERR_OOM; return nullptr; }
#ifdef MADV_HUGEPAGE
    (void)madvise(host, bytes, MADV_HUGEPAGE);
#endif
    const int n_pop = 4;
    std::vector<std::thread> pop;
    for (int t = 0; t < n_pop; t++) try {
        pop.emplace_back([=]() {
            const size_t lo = bytes / n_pop * t / 4096 * 4096, hi = (t + 1 == n_pop) ? bytes : bytes / n_pop * (t + 1) / 4096 * 4096;
#ifdef MADV_POPULATE_WRITE
            if (madvise(host + lo, hi - lo, MADV_POPULATE_WRITE) == 0) return;
#endif
            (void)lo; (void)hi;  // older kernels: the copy below faults the pages in
        });
    } catch (const std::system_error &) { break; }  // no helper thread: the copy faults the pages in itself
    bool have_bounce = ctx->bounce[0] && ctx->bounce[1] && ctx->bounce_ev[0] && ctx->bounce_ev[1];
    if (!have_bounce) {  // all four handles or none: a half-built set would hand null buffers to later calls
        have_bounce = true;
        for (int k = 0; k < 2 && have_bounce; k++)
            have_bounce = hipHostMalloc((void **)&ctx->bounce[k], kBounce, hipHostMallocDefault) == hipSuccess &&
                          hipEventCreateWithFlags(&ctx->bounce_ev[k], hipEventDisableTiming) == hipSuccess;
        if (!have_bounce) {
            (void)hipGetLastError();
            for (int k = 0; k < 2; k++) {
                if (ctx->bounce[k]) (void)hipHostFree(ctx->bounce[k]);
                if (ctx->bounce_ev[k]) (void)hipEventDestroy(ctx->bounce_ev[k]);
                ctx->bounce[k] = nullptr; ctx->bounce_ev[k] = nullptr;
            }
        }
    }
    if (!have_bounce) {  // no pinned staging: the plain copy still works, only slower
        const hipError_t e = hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost);
        for (auto &t : pop) t.join();
        if (e != hipSuccess) { free(host); set_error("HIP error %d copying pairs to the host", (int)e); *status = ARP_ERR_HIP; return nullptr; }
        return (arp_pair *)host;
    }
    hipError_t e = hipSuccess;
    const size_t n_chunks = (bytes + kBounce - 1) / kBounce;
    for (size_t c = 0; c <= n_chunks && e == hipSuccess; c++) {
        if (c < n_chunks) {
            const size_t off = c * kBounce, len = std::min(kBounce, bytes - off);
            e = hipMemcpyAsync(ctx->bounce[c & 1], (const char *)dev + off, len, hipMemcpyDeviceToHost, ctx->stream);
            if (e == hipSuccess) e = hipEventRecord(ctx->bounce_ev[c & 1], ctx->stream);
        }
        if (c > 0 && e == hipSuccess) {  // drain the previous chunk while this one is in flight
            const size_t off = (c - 1) * kBounce, len = std::min(kBounce, bytes - off);
            e = hipEventSynchronize(ctx->bounce_ev[(c - 1) & 1]);
            if (e == hipSuccess) memcpy(host + off, ctx->bounce[(c - 1) & 1], len);
        }
    }
    for (auto &t : pop) t.join();
    if (e != hipSuccess) { (void)hipStreamSynchronize(ctx->stream); free(host); set_error("HIP error %d copying pairs to the host", (int)e); *status = ARP_ERR_HIP; return nullptr; }
    return (arp_pair *)host;
}

// The deferred-probe list (candidates whose rules need a hydrogen / disulfide probe) is sized for 16 candidates per atom; a
// denser input overflows it (status bit 8).  Like the pair buffer, it is then grown and the pass repeated.
static arp_status grow_defer_list(arp_context *ctx, uint64_t n) {
    if (ctx->defer_scale >= 64) { set_error("deferred-probe list overflow after growing it 64-fold"); return ARP_ERR_CAPACITY; }
    ctx->defer_scale *= 4;
    (void)hipStreamSynchronize(ctx->stream);
    free_workspace(ctx);
    return ensure_workspace(ctx, n);
}

// ---- the hot path ----------------------------------------------------------------------------------------------
extern "C" arp_status arp_contacts_atomic_enqueue(arp_context *ctx, const arp_atoms *atoms, const arp_params *params, arp_pair *out,
                                                  uint64_t capacity) try {
    arp_status s = check_device(ctx);
    if (s != ARP_OK) return s;
    if ((s = validate(atoms, params)) != ARP_OK) return s;
    if (atoms->location != ARP_MEM_DEVICE) { set_error("arp_contacts_atomic_enqueue needs device-resident inputs"); return ARP_ERR_BAD_INPUT; }
    if (!out && capacity) { set_error("null output buffer"); return ARP_ERR_BAD_INPUT; }
    if ((s = ensure_workspace(ctx, atoms->n)) != ARP_OK) return s;
    DevAtoms d{};
    if ((s = stage_inputs(ctx, atoms, &d)) != ARP_OK) return s;
    if ((s = upload_params(ctx, params)) != ARP_OK) return s;
    Profiler *prof = ctx->prof.enabled ? &ctx->prof : nullptr;
    const bool ordered = (params->flags & ARP_FLAG_DETERMINISTIC) != 0, only = (params->flags & ARP_FLAG_CONTACTS_ONLY) != 0;
    grid_for_call(ctx, d, params, prof, ordered); ctx->grid_x = d.x; ctx->grid_n = d.n;
    bool direct = false;
    if (!out || capacity == 0) {
        launch_count(d, ctx->ws, ctx->stream, prof, 0, true, only);  // size query: reports ARP_ERR_CAPACITY + the count
    } else if (params->flags & ARP_FLAG_DETERMINISTIC) {
        launch_count(d, ctx->ws, ctx->stream, prof, capacity, true, only);
        launch_fill_ordered(d, ctx->ws, out, capacity, ctx->stream, prof, only);
    } else {
        ctx->last_skip = !(params->flags & ARP_FLAG_NO_SPECULATION) && skip_deferred_pass(ctx, d);
        direct = launch_emit(d, ctx->ws, out, capacity, ctx->stream, prof, only, ctx->last_skip, ctx->rkey_valid);
    }
    ctx->last_direct = direct;
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(ctx->h_result, ctx->ws.result, kResultWords * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    ctx->last_capacity = capacity;
    ctx->last_atoms = *atoms; ctx->last_out = out;  // (device pointers: the caller keeps them alive until arp_contacts_atomic_result)
    ctx->pending = true;
    return ARP_OK;
} ARP_ABI_CATCH

extern "C" arp_status arp_contacts_atomic_result(arp_context *ctx, uint64_t *n_pairs) try {
    arp_status s = check_device(ctx);
    if (s != ARP_OK) return s;
    if (!ctx->pending) { set_error("no enqueued call"); return ARP_ERR_BAD_INPUT; }
    for (;;) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        ctx->pending = false;
        finish_result(ctx, ctx->last_direct, ctx->last_skip, ctx->last_capacity);
        note_residue_runs(ctx);
        if (n_pairs) *n_pairs = ctx->h_result[0];
        const arp_atoms again = ctx->last_atoms;
        const arp_params prm = ctx->last_params;
        if (ctx->last_out && ctx->last_capacity && !(prm.flags & ARP_FLAG_DETERMINISTIC)) {  // a single-pass call: keep the deferred-pass memo
            DevAtoms d{}; d.x = again.x; d.n = (uint32_t)again.n;
            note_deferred(ctx, d, ctx->last_skip);
        }
        s = flags_to_status(ctx->h_result[1]);
        if (s != kRetryDefer && s != kRetryDeferPass) break;
        // grow the deferred-probe list (or: run the probe pass after all) and run the enqueued call again (same inputs, same output buffer)
        if (s == kRetryDefer && (s = grow_defer_list(ctx, again.n)) != ARP_OK) return s;
        if ((s = arp_contacts_atomic_enqueue(ctx, &again, &prm, ctx->last_out, ctx->last_capacity)) != ARP_OK) return s;
    }
    if (s != ARP_OK) return s;
    if (ctx->h_result[0] > ctx->last_capacity) {
        set_error("pair buffer too small: %llu pairs needed, capacity %llu", ctx->h_result[0], (unsigned long long)ctx->last_capacity);
        return ARP_ERR_CAPACITY;
    }
    return ARP_OK;
} ARP_ABI_CATCH

// Single-pass emitter into the context's reusable device buffer: ONE pass -- no count pass, no hipMalloc/hipFree per call (together
// ~half of the latency of a PDB-sized structure).  A buffer that turns out too small only makes the pass report the size (k_fixup);
// it is then grown and the pass repeated.  Leaves the list in ctx->out_buf[0 .. *total).
static arp_status single_pass_into_context_buffer(arp_context *ctx, uint64_t n_atoms, const DevAtoms &d, const arp_params *params, Profiler *prof,
                                                  unsigned long long *total_out) {
    arp_status s;
    // first guess: 64 records per atom (twice the all-pairs density of a protein), at most 2 GiB; a larger result costs one more pass
    uint64_t want = std::max<uint64_t>(ctx->out_cap, std::min<uint64_t>(std::max<uint64_t>(64 * n_atoms, 1u << 16), 1u << 27));
    unsigned long long total = 0;
    for (int attempt = 0;; attempt++) {
        if (ctx->out_cap < want) {
            HIP_TRY(hipStreamSynchronize(ctx->stream));
            if (ctx->out_buf) (void)hipFree(ctx->out_buf);
            ctx->out_buf = nullptr; ctx->out_cap = 0;
            HIP_TRY(hipMalloc((void **)&ctx->out_buf, want * sizeof(arp_pair)));
            ctx->out_cap = want;
        }
        grid_for_call(ctx, d, params, prof, false); ctx->grid_x = d.x; ctx->grid_n = d.n;
        const bool skip = skip_deferred_pass(ctx, d);
        const bool direct = launch_emit(d, ctx->ws, ctx->out_buf, ctx->out_cap, ctx->stream, prof, (params->flags & ARP_FLAG_CONTACTS_ONLY) != 0, skip, ctx->rkey_valid);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(ctx->h_result, ctx->ws.result, kResultWords * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        finish_result(ctx, direct, skip, ctx->out_cap);
        note_residue_runs(ctx);
        note_deferred(ctx, d, skip);
        if ((s = flags_to_status(ctx->h_result[1])) == kRetryDeferPass) { attempt--; continue; }  // the memo was stale: once more, with the probe pass
        if (s != ARP_OK) return s;
        total = ctx->h_result[0];
        if (total <= ctx->out_cap) break;
        if (attempt) { set_error("internal error: pair count changed between passes"); return ARP_ERR_HIP; }
        want = total + total / 8;
    }
    *total_out = total;
    return ARP_OK;
}

// The table path's pair pass (table.cpp get_contacts_device): device-resident inputs, the list stays in the context's buffer --
// *data is a VIEW, valid until the next call on this context.
arp_status arp::contacts_atomic_view(arp_context *ctx, const arp_atoms *atoms, const arp_params *params, const arp_pair **data, uint64_t *n) {
    *data = nullptr; *n = 0;
    if (!atoms || atoms->location != ARP_MEM_DEVICE || !params || (params->flags & ARP_FLAG_DETERMINISTIC)) { set_error("contacts_atomic_view: bad arguments"); return ARP_ERR_BAD_INPUT; }
    for (;;) {
        arp_status s = check_device(ctx);
        if (s != ARP_OK) return s;
        if ((s = validate(atoms, params)) != ARP_OK) return s;
        if ((s = ensure_workspace(ctx, atoms->n)) != ARP_OK) return s;
        DevAtoms d{};
        if ((s = stage_inputs(ctx, atoms, &d)) != ARP_OK) return s;
        if ((s = upload_params(ctx, params)) != ARP_OK) return s;
        unsigned long long total = 0;
        const bool timing = g_debug.timing != 0;
        const auto t0 = std::chrono::steady_clock::now();
        s = single_pass_into_context_buffer(ctx, atoms->n, d, params, ctx->prof.enabled ? &ctx->prof : nullptr, &total);
        if (timing) fprintf(stderr, "    pair pass (launches + sync)      %8.3f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        if (s == kRetryDefer) { if ((s = grow_defer_list(ctx, atoms->n)) != ARP_OK) return s; continue; }
        if (s != ARP_OK) return s;
        *data = ctx->out_buf; *n = total;
        return ARP_OK;
    }
}

static arp_status contacts_atomic_once(arp_context *ctx, const arp_atoms *atoms, const arp_params *params, int32_t out_location, arp_pairs *out);
extern "C" arp_status arp_contacts_atomic(arp_context *ctx, const arp_atoms *atoms, const arp_params *params, int32_t out_location,
                                          arp_pairs *out) try {
    for (;;) {
        arp_status s = contacts_atomic_once(ctx, atoms, params, out_location, out);
        if (s != kRetryDefer) return s;
        if ((s = grow_defer_list(ctx, atoms->n)) != ARP_OK) return s;  // like the pair buffer: grow, repeat the pass
    }
} ARP_ABI_CATCH
static arp_status contacts_atomic_once(arp_context *ctx, const arp_atoms *atoms, const arp_params *params, int32_t out_location, arp_pairs *out) {
    if (!out) { set_error("null out"); return ARP_ERR_BAD_INPUT; }
    out->n = 0; out->data = nullptr; out->location = out_location;
    arp_status s = check_device(ctx);
    if (s != ARP_OK) return s;
    if ((s = validate(atoms, params)) != ARP_OK) return s;
    if (out_location != ARP_MEM_HOST && out_location != ARP_MEM_DEVICE) { set_error("bad out_location"); return ARP_ERR_BAD_INPUT; }
    if ((s = ensure_workspace(ctx, atoms->n)) != ARP_OK) return s;
    DevAtoms d{};
    if ((s = stage_inputs(ctx, atoms, &d)) != ARP_OK) return s;
    if ((s = upload_params(ctx, params)) != ARP_OK) return s;
    Profiler *prof = ctx->prof.enabled ? &ctx->prof : nullptr;
    if (out_location == ARP_MEM_HOST && !(params->flags & ARP_FLAG_DETERMINISTIC)) {
        unsigned long long total = 0;
        if ((s = single_pass_into_context_buffer(ctx, atoms->n, d, params, prof, &total)) != ARP_OK) return s;
        if (total == 0) return ARP_OK;
        arp_pair *host = download_pairs(ctx, ctx->out_buf, total, &s);
        if (!host) return s;
        out->data = host; out->n = total;
        return ARP_OK;
    }
    // count pass -> output size -> ordered fill or single-pass emit.  With ARP_FLAG_CONTACTS_ONLY the single-pass emitter
    // sizes the device buffer by the (cheap) candidate count, an upper bound; the ordered one needs the exact filtered counts.
    const bool ordered = (params->flags & ARP_FLAG_DETERMINISTIC) != 0, only = (params->flags & ARP_FLAG_CONTACTS_ONLY) != 0;
    grid_for_call(ctx, d, params, prof, ordered); ctx->grid_x = d.x; ctx->grid_n = d.n;
    launch_count(d, ctx->ws, ctx->stream, prof, 0, false, only && ordered);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(ctx->h_result, ctx->ws.result, kResultWords * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if ((s = flags_to_status(ctx->h_result[1])) != ARP_OK) return s;
    unsigned long long total = ctx->h_result[0];
    if (total == 0) return ARP_OK;
    arp_pair *dev = nullptr;
    HIP_TRY(hipMalloc((void **)&dev, total * sizeof(arp_pair)));
    bool direct = false;
    if (params->flags & ARP_FLAG_DETERMINISTIC) launch_fill_ordered(d, ctx->ws, dev, total, ctx->stream, prof, only);
    else direct = launch_emit(d, ctx->ws, dev, total, ctx->stream, prof, only, false, ctx->rkey_valid);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(ctx->h_result, ctx->ws.result, kResultWords * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e == hipSuccess) { finish_result(ctx, direct, false, total); note_residue_runs(ctx); }
    if (e != hipSuccess) { (void)hipFree(dev); set_error("HIP error %d (%s) in the fill pass", (int)e, hipGetErrorString(e)); return ARP_ERR_HIP; }
    if ((s = flags_to_status(ctx->h_result[1])) != ARP_OK) { (void)hipFree(dev); return s; }
    total = std::min<unsigned long long>(total, ctx->h_result[0]);  // fewer than the candidates with ARP_FLAG_CONTACTS_ONLY
    if (total == 0) { (void)hipFree(dev); return ARP_OK; }
    if (out_location == ARP_MEM_DEVICE) {
        out->data = dev; out->n = total;
        return ARP_OK;
    }
    arp_pair *host = download_pairs(ctx, dev, total, &s);
    (void)hipFree(dev);
    if (!host) return s;
    out->data = host; out->n = total;
    return ARP_OK;
}

// ---- pair lists that share one pinned block (the batch path) --------------------------------------------------------------------
// A pack's pair list crosses PCIe once, into ONE pinned block, and every member's arp_pairs is a view into it: no per-member malloc, no
// second copy (~70 k records = 1.1 MB per 5k-atom structure with full candidate lists: crossing PCIe once is the floor of that path,
// ~20 us per structure).  The block is reference-counted through a registry keyed by the members' data pointers -- arp_pairs_free
// finds it there -- and an idle block goes back to a pool instead of to the driver: pinning memory costs far more than the copy it saves.
namespace {
struct SharedBlock { char *pinned = nullptr; size_t cap = 0; long refs = 0; };
std::mutex g_shared_mu;
std::unordered_map<const void *, SharedBlock *> g_shared_views;
std::vector<SharedBlock *> g_shared_pool;
size_t g_shared_pool_bytes = 0;
// idle pinned memory kept for the next batch / table: 4 GiB unless ARPEGGIA_AMD_HOST_POOL_MB says otherwise (0 = keep nothing).  (2 GiB was
// measured in round 4: a batch of 2048 five-thousand-atom structures with full candidate lists returns 2.3 GB of lists, the blocks beyond the
// limit were unpinned and pinned again on every call -- 24 -> 41 us per structure.)
const size_t kSharedPoolLimit = [] {
    const char *e = getenv("ARPEGGIA_AMD_HOST_POOL_MB");
    const long long mb = e ? atoll(e) : 4096;
    return (size_t)(mb < 0 ? 0 : mb) << 20;
}();

SharedBlock *shared_acquire(size_t bytes) {
    {
        std::lock_guard<std::mutex> lk(g_shared_mu);
        size_t best = g_shared_pool.size();
        for (size_t k = 0; k < g_shared_pool.size(); k++)
            // (the smallest pooled block that fits -- but not one more than twice the request + 1 MiB: a 1 KB table must not pin a multi-GB block
            // for as long as one of its views lives)
            if (g_shared_pool[k]->cap >= bytes && g_shared_pool[k]->cap <= 2 * bytes + (1u << 20) &&
                (best == g_shared_pool.size() || g_shared_pool[k]->cap < g_shared_pool[best]->cap)) best = k;
        if (best != g_shared_pool.size()) {
            SharedBlock *b = g_shared_pool[best];
            g_shared_pool.erase(g_shared_pool.begin() + (long)best);
            g_shared_pool_bytes -= b->cap;
            return b;
        }
    }
    SharedBlock *b = new (std::nothrow) SharedBlock();
    if (!b) return nullptr;
    const size_t cap = bytes + bytes / 8 + 4096;
    if (hipHostMalloc((void **)&b->pinned, cap, hipHostMallocPortable) != hipSuccess) { (void)hipGetLastError(); delete b; return nullptr; }
    b->cap = cap;
    return b;
}
void shared_release(SharedBlock *b) {  // (g_shared_mu held)
    if (g_shared_pool_bytes + b->cap <= kSharedPoolLimit) { g_shared_pool.push_back(b); g_shared_pool_bytes += b->cap; return; }
    (void)hipHostFree(b->pinned);
    delete b;
}
}  // namespace

std::shared_ptr<char> arp::pinned_block(size_t bytes) {
    SharedBlock *b = shared_acquire(bytes);
    if (!b) return nullptr;
    return std::shared_ptr<char>(b->pinned, [b](char *) { std::lock_guard<std::mutex> lk(g_shared_mu); shared_release(b); });
}

extern "C" uint64_t arp_release_host_pool(void) {
    std::vector<SharedBlock *> idle;
    {
        std::lock_guard<std::mutex> lk(g_shared_mu);
        idle.swap(g_shared_pool);
        g_shared_pool_bytes = 0;
    }
    uint64_t bytes = 0;
    for (SharedBlock *b : idle) { bytes += b->cap; (void)hipHostFree(b->pinned); delete b; }
    return bytes;
}

extern "C" void arp_pairs_free(arp_pairs *pairs) {
    if (!pairs || !pairs->data) return;
    if (pairs->location == ARP_MEM_DEVICE) (void)hipFree(pairs->data);
    else {
        bool shared = false;
        {
            std::lock_guard<std::mutex> lk(g_shared_mu);
            auto it = g_shared_views.find(pairs->data);
            if (it != g_shared_views.end()) {
                shared = true;
                SharedBlock *b = it->second;
                g_shared_views.erase(it);
                if (--b->refs == 0) shared_release(b);
            }
        }
        if (!shared) free(pairs->data);
    }
    pairs->data = nullptr; pairs->n = 0;
}

namespace {
constexpr uint64_t kPackAtoms = 1u << 20;    // atoms per pack: ~200 structures of 5k atoms; ~50 packs keep the pipeline full on a 10^4 batch
constexpr uint32_t kPackMembers = 32768;     // members per pack (the device also checks that the models fit 16 bits)

struct PackPlan {
    std::vector<int32_t> members;
    uint64_t n = 0, n_res = 0, n_h = 0;
    bool single = false;                     // not packable: goes through arp_contacts_atomic on its own
};

bool packable(const arp_atoms *a) {
    if (!a || a->location != ARP_MEM_HOST || a->n == 0 || a->n_res == 0 || a->n >= kPackAtoms) return false;
    if (!a->x || !a->y || !a->z || !a->attr || !a->res_ord || !a->chain_rank || !a->model || !a->res_id || !a->res_h_ptr || !a->res_cb || !a->res_sg) return false;
    if (a->res_h_ptr[a->n_res] && !a->res_h_idx) return false;
    return true;
}

// segments of a pack's block, 256-byte aligned: the twelve input arrays, the descriptor table, then device-only scratch
struct PackLayout {
    enum { X, Y, Z, ATTR, RES_ORD, CHAIN, MODEL, RES_ID, RES_H_PTR, RES_CB, RES_SG, RES_H_IDX, DESC, N_MODELS, STATUS, COUNT, OFFSET, CURSOR, N_SEG };
    uint64_t off[N_SEG], upload = 0, total = 0;
    PackLayout(uint64_t n, uint64_t nr, uint64_t nh, uint64_t K) {
        const uint64_t bytes[N_SEG] = {n * 8, n * 8, n * 8, n * 4, n * 4, n * 4, n * 4, n * 4, (nr + 1) * 4, nr * 4, nr * 4, nh * 4, (K + 1) * 16,
                                       K * 4, 256, K * 8, (K + 1) * 8, K * 8};
        uint64_t t = 0;
        for (int k = 0; k < N_SEG; k++) { off[k] = t; t += (bytes[k] + 255u) & ~255ull; if (k == DESC) upload = t; }
        total = std::max<uint64_t>(t, 256);
    }
};

template <class F>
void run_helpers(int helpers, size_t n, F &&fn) {  // fn(item) over [0, n) on up to `helpers` threads (dynamic: items differ in size)
    if (helpers <= 1 || n <= 1) { for (size_t k = 0; k < n; k++) fn(k); return; }
    std::atomic<size_t> next{0};
    std::exception_ptr first_error;  // (as parallel_for, host_common.h: no exception leaves a helper thread, none unwinds past a joinable one)
    std::mutex error_mu;
    auto work = [&]() noexcept {
        try { for (size_t k; (k = next.fetch_add(1, std::memory_order_relaxed)) < n;) fn(k); }
        catch (...) { next.store(n, std::memory_order_relaxed); std::lock_guard<std::mutex> lk(error_mu); if (!first_error) first_error = std::current_exception(); }
    };
    std::vector<std::thread> th;
    th.reserve((size_t)helpers);
    {
        struct JoinAll { std::vector<std::thread> &t; ~JoinAll() { for (auto &x : t) if (x.joinable()) x.join(); } } join_all{th};
        for (int t = 1; t < helpers; t++) try { th.emplace_back(work); } catch (const std::system_error &) { break; }
        work();
    }
    if (first_error) std::rethrow_exception(first_error);
}

struct BatchLap {  // arp_debug_set("timing", 1): where a pack's host time goes (stderr)
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void operator()(const char *what) {
        if (!g_debug.timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "    batch %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

struct BatchSlot {
    arp_context *ctx = nullptr;
    PackPlan plan;
    bool in_flight = false;
    std::vector<uint64_t> first_atom;        // per member (+ sentinel): offsets inside the pack
    PackArrays pa{};
    DevAtoms dev{};
    bool ordered = false;
};

arp_status ensure_pack_buffers(arp_context *ctx, uint64_t in_bytes, uint64_t out_records, uint64_t K) {
    auto &s = ctx->st;
    if (s.bytes < in_bytes) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (s.dev) (void)hipFree(s.dev);
        if (s.pinned) (void)hipHostFree(s.pinned);
        s = arp_context::Staged{};
        const uint64_t cap = in_bytes + in_bytes / 4;
        HIP_TRY(hipMalloc((void **)&s.dev, cap));
        HIP_TRY(hipHostMalloc((void **)&s.pinned, cap, hipHostMallocDefault));
        s.bytes = cap;
    }
    if (ctx->out_cap < out_records || ctx->grp_cap < out_records) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->out_buf) (void)hipFree(ctx->out_buf);
        if (ctx->grp_buf) (void)hipFree(ctx->grp_buf);
        ctx->out_buf = ctx->grp_buf = nullptr; ctx->out_cap = ctx->grp_cap = 0;
        HIP_TRY(hipMalloc((void **)&ctx->out_buf, out_records * sizeof(arp_pair)));
        ctx->out_cap = out_records;
        HIP_TRY(hipMalloc((void **)&ctx->grp_buf, out_records * sizeof(arp_pair)));
        ctx->grp_cap = out_records;
    }
    if (ctx->h_offsets_cap < K + 4) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->h_offsets) (void)hipHostFree(ctx->h_offsets);
        ctx->h_offsets = nullptr; ctx->h_offsets_cap = 0;
        HIP_TRY(hipHostMalloc((void **)&ctx->h_offsets, (K + 4 + K / 4) * sizeof(unsigned long long), hipHostMallocDefault));
        ctx->h_offsets_cap = K + 4 + K / 4;
    }
    return ARP_OK;
}

// steps 4-6 of a pack: grid + pair kernels + split + the small D2H of counts (everything asynchronous on the slot's stream)
arp_status enqueue_pack_kernels(BatchSlot &sl, const arp_params *params) {
    arp_context *ctx = sl.ctx;
    const bool only = (params->flags & ARP_FLAG_CONTACTS_ONLY) != 0;
    Profiler *prof = nullptr;
    grid_for_call(ctx, sl.dev, params, prof, sl.ordered); ctx->grid_x = nullptr; ctx->grid_n = 0;  // (a pack's grid: per-model origins)
    if (sl.ordered) {
        launch_count(sl.dev, ctx->ws, ctx->stream, prof, ctx->out_cap, true, only);
        launch_fill_ordered(sl.dev, ctx->ws, ctx->out_buf, ctx->out_cap, ctx->stream, prof, only);
    } else {
        // (a pack never takes the hole-free sequence of small inputs -- DevAtoms::per_model rules it out in launch_emit_e --: that sequence leaves
        // result[0] and the capacity flag for the HOST to derive (finish_result), and the split kernels below read result[0] on the device)
        if (launch_emit(sl.dev, ctx->ws, ctx->out_buf, ctx->out_cap, ctx->stream, prof, only, false, ctx->rkey_valid)) {
            set_error("internal error: a pack ran the hole-free emit sequence, whose pair count only exists on the host");
            return ARP_ERR_HIP;
        }
    }
    launch_pack_split(sl.pa, ctx->ws.result, ctx->out_buf, std::min(ctx->out_cap, ctx->grp_cap), ctx->grp_buf, sl.ordered, ctx->stream);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(ctx->h_result, ctx->ws.result, kResultWords * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->h_offsets, sl.pa.offset, (sl.pa.K + 1) * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->h_offsets + sl.pa.K + 1, sl.pa.status, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    return ARP_OK;
}

arp_status launch_pack(BatchSlot &sl, const arp_atoms *const *atoms, const arp_params *params, int helpers) {
    arp_context *ctx = sl.ctx;
    const PackPlan &pk = sl.plan;
    const uint64_t K = pk.members.size();
    arp_status s = check_device(ctx);
    if (s != ARP_OK) return s;
    if ((s = ensure_workspace(ctx, pk.n)) != ARP_OK) return s;
    const PackLayout lay(pk.n, pk.n_res, pk.n_h, K);
    // output guess: contacts-only lists hold ~1 record per atom, full candidate lists ~15-30; a pack that needs more is re-run (finalize)
    const bool only = (params->flags & ARP_FLAG_CONTACTS_ONLY) != 0;
    const uint64_t guess = std::max<uint64_t>((only ? 4u : 32u) * pk.n, 1u << 16);
    BatchLap lap;
    if ((s = ensure_pack_buffers(ctx, lay.total, std::max(guess, ctx->out_cap), K)) != ARP_OK) return s;
    lap("launch: buffers");
    char *pin = ctx->st.pinned, *dev = ctx->st.dev;
    // member offsets (serial: three running sums), then the copies -- the only per-atom host work of the batch path
    sl.first_atom.assign(K + 1, 0);
    PackDesc *desc = reinterpret_cast<PackDesc *>(pin + lay.off[PackLayout::DESC]);
    {
        uint64_t o = 0, ro = 0, ho = 0;
        for (uint64_t m = 0; m < K; m++) {
            const arp_atoms &a = *atoms[pk.members[m]];
            desc[m] = PackDesc{(uint32_t)o, (uint32_t)ro, (uint32_t)ho, 0u};
            sl.first_atom[m] = o;
            o += a.n; ro += a.n_res; ho += a.res_h_ptr[a.n_res];
        }
        desc[K] = PackDesc{(uint32_t)o, (uint32_t)ro, (uint32_t)ho, 0u};
        sl.first_atom[K] = o;
    }
    auto seg = [&](int k) { return pin + lay.off[k]; };
    run_helpers(helpers, (size_t)K, [&](size_t m) {
        const arp_atoms &a = *atoms[pk.members[m]];
        const PackDesc d = desc[m];
        const uint64_t nh = a.res_h_ptr[a.n_res];
        memcpy(seg(PackLayout::X) + 8ull * d.first_atom, a.x, a.n * 8); memcpy(seg(PackLayout::Y) + 8ull * d.first_atom, a.y, a.n * 8);
        memcpy(seg(PackLayout::Z) + 8ull * d.first_atom, a.z, a.n * 8);
        memcpy(seg(PackLayout::ATTR) + 4ull * d.first_atom, a.attr, a.n * 4); memcpy(seg(PackLayout::RES_ORD) + 4ull * d.first_atom, a.res_ord, a.n * 4);
        memcpy(seg(PackLayout::CHAIN) + 4ull * d.first_atom, a.chain_rank, a.n * 4); memcpy(seg(PackLayout::MODEL) + 4ull * d.first_atom, a.model, a.n * 4);
        memcpy(seg(PackLayout::RES_ID) + 4ull * d.first_atom, a.res_id, a.n * 4);
        memcpy(seg(PackLayout::RES_H_PTR) + 4ull * d.first_res, a.res_h_ptr, a.n_res * 4);
        memcpy(seg(PackLayout::RES_CB) + 4ull * d.first_res, a.res_cb, a.n_res * 4); memcpy(seg(PackLayout::RES_SG) + 4ull * d.first_res, a.res_sg, a.n_res * 4);
        if (nh) memcpy(seg(PackLayout::RES_H_IDX) + 4ull * d.first_h, a.res_h_idx, nh * 4);
    });
    lap("launch: assemble (host)");
    HIP_TRY(hipMemcpyAsync(dev, pin, lay.upload, hipMemcpyHostToDevice, ctx->stream));
    lap("launch:   H2D call");
    auto at = [&](int k) { return dev + lay.off[k]; };
    PackArrays &pa = sl.pa;
    pa.n = (uint32_t)pk.n; pa.n_res = (uint32_t)pk.n_res; pa.n_h = (uint32_t)pk.n_h; pa.K = (uint32_t)K;
    pa.desc = (PackDesc *)at(PackLayout::DESC); pa.model = (uint32_t *)at(PackLayout::MODEL); pa.res_id = (uint32_t *)at(PackLayout::RES_ID);
    pa.res_h_ptr = (uint32_t *)at(PackLayout::RES_H_PTR); pa.res_cb = (uint32_t *)at(PackLayout::RES_CB); pa.res_sg = (uint32_t *)at(PackLayout::RES_SG);
    pa.res_h_idx = (uint32_t *)at(PackLayout::RES_H_IDX); pa.n_models = (uint32_t *)at(PackLayout::N_MODELS); pa.status = (uint32_t *)at(PackLayout::STATUS);
    pa.count = (unsigned long long *)at(PackLayout::COUNT); pa.offset = (unsigned long long *)at(PackLayout::OFFSET); pa.cursor = (unsigned long long *)at(PackLayout::CURSOR);
    launch_pack_fix(pa, ctx->stream);
    lap("launch:   pack_fix calls");
    DevAtoms &d = sl.dev;
    d = DevAtoms{};
    d.n = pa.n; d.n_res = pa.n_res; d.per_model = 1u;
    d.x = (const double *)at(PackLayout::X); d.y = (const double *)at(PackLayout::Y); d.z = (const double *)at(PackLayout::Z);
    d.attr = (const uint32_t *)at(PackLayout::ATTR); d.res_ord = (const uint32_t *)at(PackLayout::RES_ORD);
    d.chain_rank = (const uint32_t *)at(PackLayout::CHAIN); d.model = pa.model;
    d.res_id = pa.res_id; d.res_h_ptr = pa.res_h_ptr; d.res_h_idx = pa.res_h_idx; d.res_cb = pa.res_cb; d.res_sg = pa.res_sg;
    sl.ordered = false;  // (ordered calls are never packed, see the plan)
    if ((s = upload_params(ctx, params)) != ARP_OK) return s;
    lap("launch:   params");
    if ((s = enqueue_pack_kernels(sl, params)) != ARP_OK) return s;
    lap("launch: enqueue (kernels)");
    sl.in_flight = true;
    return ARP_OK;
}

// wait for a pack, fetch the grouped list, hand the members their lists.  An input error inside the pack (or more models than 16
// bits hold) is re-run member by member so that the failing structure reports it.
arp_status finalize_pack(BatchSlot &sl, const arp_atoms *const *atoms, const arp_params *params, arp_pairs *outs, int helpers) {
    if (!sl.in_flight) return ARP_OK;
    sl.in_flight = false;
    arp_context *ctx = sl.ctx;
    const PackPlan &pk = sl.plan;
    const uint64_t K = pk.members.size();
    arp_status s = check_device(ctx);
    if (s != ARP_OK) return s;
    unsigned long long total = 0;
    BatchLap lap;
    for (int attempt = 0;; attempt++) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        lap("finalize: wait for kernels");
        const uint32_t pack_status = *reinterpret_cast<const uint32_t *>(ctx->h_offsets + K + 1);
        note_residue_runs(ctx);
        s = flags_to_status(ctx->h_result[1]);
        if (s == kRetryDefer && attempt < 4) {  // deferred-probe list too small: grow it, run the pack's kernels again
            if ((s = grow_defer_list(ctx, pk.n)) != ARP_OK) return s;
            if ((s = upload_params(ctx, params)) != ARP_OK || (s = enqueue_pack_kernels(sl, params)) != ARP_OK) return s;
            continue;
        }
        if (s != ARP_OK || pack_status != 0u) {
            for (int32_t k : pk.members)
                if ((s = arp_contacts_atomic(ctx, atoms[k], params, ARP_MEM_HOST, &outs[k])) != ARP_OK) return s;
            return ARP_OK;
        }
        total = ctx->h_result[0];
        if (total <= ctx->out_cap) break;
        if (attempt >= 4) { set_error("internal error: pair count changed between passes"); return ARP_ERR_HIP; }
        const PackLayout lay(pk.n, pk.n_res, pk.n_h, K);
        if ((s = ensure_pack_buffers(ctx, lay.total, total + total / 8, K)) != ARP_OK) return s;  // (the staged inputs stay where they are)
        if ((s = upload_params(ctx, params)) != ARP_OK || (s = enqueue_pack_kernels(sl, params)) != ARP_OK) return s;
    }
    SharedBlock *blk = nullptr;
    if (total) {
        if (!(blk = shared_acquire(total * sizeof(arp_pair)))) { set_error("out of pinned host memory for the batch's pair lists"); return ARP_ERR_OOM; }
        lap("finalize: pinned block");
        hipError_t e = hipMemcpyAsync(blk->pinned, ctx->grp_buf, total * sizeof(arp_pair), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            std::lock_guard<std::mutex> lk(g_shared_mu);
            shared_release(blk);
            set_error("HIP error %d (%s) copying the batch's pair lists to the host", (int)e, hipGetErrorString(e));
            return ARP_ERR_HIP;
        }
    }
    lap("finalize: D2H of the lists");
    const unsigned long long *off = ctx->h_offsets;
    std::lock_guard<std::mutex> lk(g_shared_mu);
    for (uint64_t m = 0; m < K; m++) {  // every member's list is a view into the pack's block (arp_pairs_free drops the reference)
        arp_pairs &out = outs[pk.members[m]];
        const unsigned long long cnt = off[m + 1] - off[m];
        out.n = cnt; out.location = ARP_MEM_HOST; out.data = nullptr;
        if (!cnt) continue;
        out.data = reinterpret_cast<arp_pair *>(blk->pinned) + off[m];
        g_shared_views.emplace(out.data, blk);
        blk->refs++;
    }
    if (blk && blk->refs == 0) shared_release(blk);
    (void)helpers;
    return ARP_OK;
}
}  // namespace

extern "C" arp_status arp_contacts_atomic_batch(arp_context *const *ctxs, int32_t n_ctx, const arp_atoms *const *atoms, int32_t n_structures,
                                                const arp_params *params, arp_pairs *outs) try {
    if (!ctxs || n_ctx <= 0 || !atoms || n_structures < 0 || !outs || !params) { set_error("bad batch arguments"); return ARP_ERR_BAD_INPUT; }
    for (int32_t k = 0; k < n_structures; k++) outs[k] = arp_pairs{0, nullptr, ARP_MEM_HOST, 0};
    for (int32_t k = 0; k < n_structures; k++)
        if (!atoms[k]) { set_error("null structure %d in the batch", k); return ARP_ERR_BAD_INPUT; }
    for (int d = 0; d < n_ctx; d++)
        if (!ctxs[d]) { set_error("null context %d in the batch", d); return ARP_ERR_BAD_INPUT; }
    // longest-processing-time-first deal over the devices (the same rule as arpeggia_amd/sharding.py)
    std::vector<int32_t> order(n_structures);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return atoms[a]->n > atoms[b]->n; });
    std::vector<std::vector<int32_t>> queue(n_ctx);
    std::vector<uint64_t> load(n_ctx, 0);
    for (int32_t k : order) {
        int best = (int)(std::min_element(load.begin(), load.end()) - load.begin());
        queue[best].push_back(k);
        load[best] += atoms[k]->n + 1;
    }
    const int hw = (int)std::max(1u, std::thread::hardware_concurrency());
    const int helpers = std::max(1, std::min(8, hw / std::max(1, n_ctx)));
    std::vector<arp_status> st(n_ctx, ARP_OK);
    std::vector<std::string> msg(n_ctx);
    auto device_worker_body = [&](int d) {
        auto fail = [&](arp_status s) { st[d] = s; msg[d] = arp_last_error(); };
        // plan: consecutive members of the device's share form packs; what cannot be packed runs alone
        std::vector<PackPlan> plans;
        {
            PackPlan cur;
            auto close = [&]() { if (!cur.members.empty()) plans.push_back(std::move(cur)); cur = PackPlan{}; };
            for (int32_t k : queue[d]) {
                const arp_atoms *a = atoms[k];
                // The ordered emitter's promise (output byte-identical run to run) cannot be kept through a pack: its records are laid
                // out task by task, and the task that straddles two members interleaves their records.  Ordered calls go one by one.
                if (!packable(a) || (params->flags & ARP_FLAG_DETERMINISTIC)) {
                    close();
                    PackPlan one; one.members.push_back(k); one.single = true; plans.push_back(std::move(one));
                    continue;
                }
                if (!cur.members.empty() && (cur.n + a->n > kPackAtoms || cur.members.size() >= kPackMembers)) close();
                cur.members.push_back(k);
                cur.n += a->n; cur.n_res += a->n_res; cur.n_h += a->res_h_ptr[a->n_res];
            }
            close();
            for (PackPlan &p : plans) if (!p.single && p.members.size() == 1) p.single = true;  // nothing to share a launch with
        }
        BatchSlot slot[2];
        slot[0].ctx = ctxs[d];
        arp_status s = ARP_OK;
        const size_t np = plans.size();
        for (size_t i = 0; i <= np && s == ARP_OK; i++) {
            if (i < np) {
                if (plans[i].single) {  // synchronous: drain the pipeline first (it uses both contexts)
                    for (int q = 0; q < 2 && s == ARP_OK; q++) s = finalize_pack(slot[(i + q) & 1], atoms, params, outs, helpers);
                    if (s == ARP_OK) s = arp_contacts_atomic(ctxs[d], atoms[plans[i].members[0]], params, ARP_MEM_HOST, &outs[plans[i].members[0]]);
                    continue;
                }
                BatchSlot &sl = slot[i & 1];
                if (!sl.ctx) {  // the second context of the device: same device, its own stream and workspace; lives with the first
                    if (!ctxs[d]->peer && (s = arp_context_create(ctxs[d]->device, &ctxs[d]->peer)) != ARP_OK) break;
                    sl.ctx = ctxs[d]->peer;
                }
                sl.plan = std::move(plans[i]);
                if ((s = launch_pack(sl, atoms, params, helpers)) != ARP_OK) break;
            }
            if (i >= 1 && !(i - 1 < np && plans[i - 1].single)) s = finalize_pack(slot[(i - 1) & 1], atoms, params, outs, helpers);
        }
        if (s != ARP_OK) {
            fail(s);
            for (int q = 0; q < 2; q++) if (slot[q].ctx) (void)hipStreamSynchronize(slot[q].ctx->stream);
        }
    };
    // ARP_ABI_CATCH only guards the calling thread: an exception that left a std::thread's function would terminate the host process.  Every
    // worker therefore turns its own exceptions into a status (msg[d] is a short constant: no allocation on the way out of bad_alloc).
    auto device_worker = [&](int d) noexcept {
        try { device_worker_body(d); }
        catch (const std::bad_alloc &) { st[d] = ARP_ERR_OOM; try { msg[d] = "out of host memory in a batch worker"; } catch (...) {} }
        catch (const std::exception &e) { st[d] = ARP_ERR_HIP; try { msg[d] = e.what(); } catch (...) {} }
        catch (...) { st[d] = ARP_ERR_HIP; }
        if (st[d] != ARP_OK) {  // whatever was launched on this device's streams must not outlive the buffers the caller is about to get back
            (void)hipStreamSynchronize(ctxs[d]->stream);
            if (ctxs[d]->peer) (void)hipStreamSynchronize(ctxs[d]->peer->stream);
        }
    };
    std::vector<std::thread> th;
    th.reserve((size_t)n_ctx);  // (no reallocation while joinable threads sit in the vector)
    struct JoinAll { std::vector<std::thread> &t; ~JoinAll() { for (auto &x : t) if (x.joinable()) x.join(); } } join_all{th};
    for (int d = 1; d < n_ctx; d++)
        try { th.emplace_back(device_worker, d); } catch (const std::system_error &) { device_worker(d); }  // no thread: this device's share runs here
    device_worker(0);
    for (auto &t : th) t.join();
    for (int d = 0; d < n_ctx; d++)
        if (st[d] != ARP_OK) {
            for (int32_t k = 0; k < n_structures; k++) arp_pairs_free(&outs[k]);
            set_error("%s", msg[d].c_str());
            return st[d];
        }
    return ARP_OK;
} ARP_ABI_CATCH

// ---- SAP neighbour sum (reference src/sap.rs:155-204) ---------------------------------------------------
extern "C" float arp_sap_weight(const char *resn, float sasa) {
    // hydrophobicity (Black & Mould minus glycine, sap.rs:41-64) x clamp(sasa / max side-chain SASA (sap.rs:77-101), 0, 1); 0 for residues
    // without a hydrophobicity value (sap.rs:198-209)
    static const struct { const char *n; float h, a; } T[] = {
        {"ALA", 0.616f - 0.501f, 15.395f}, {"ARG", 0.000f - 0.501f, 124.338f}, {"ASN", 0.236f - 0.501f, 90.303f}, {"ASP", 0.028f - 0.501f, 87.601f},
        {"CYS", 0.680f - 0.501f, 46.456f}, {"GLU", 0.043f - 0.501f, 95.534f}, {"GLN", 0.251f - 0.501f, 99.186f}, {"GLY", 0.000f, 3.229f},
        {"HIS", 0.165f - 0.501f, 96.532f}, {"ILE", 0.943f - 0.501f, 31.448f}, {"LEU", 0.943f - 0.501f, 30.271f}, {"LYS", 0.283f - 0.501f, 61.962f},
        {"MET", 0.738f - 0.501f, 65.233f}, {"PHE", 1.000f - 0.501f, 67.945f}, {"PRO", 0.711f - 0.501f, 17.812f}, {"SER", 0.359f - 0.501f, 39.355f},
        {"THR", 0.450f - 0.501f, 42.648f}, {"TRP", 0.878f - 0.501f, 101.491f}, {"TYR", 0.880f - 0.501f, 94.478f}, {"VAL", 0.825f - 0.501f, 26.702f}};
    if (!resn) return 0.0f;
    char up[8] = {0};
    for (int k = 0; k < 7 && resn[k]; k++) up[k] = (char)toupper((unsigned char)resn[k]);
    for (const auto &t : T)
        if (strcmp(t.n, up) == 0) return t.h * std::min(1.0f, std::max(0.0f, sasa / t.a));
    return 0.0f;
}

extern "C" arp_status arp_sap_neighbor_sum(arp_context *ctx, uint64_t n, const double *x, const double *y, const double *z, const uint8_t *sidechain,
                                           const float *weight, float sap_radius, float *out) try {
    arp_status s = check_device(ctx);
    if (s != ARP_OK) return s;
    if (n && (!x || !y || !z || !sidechain || !weight || !out)) { set_error("null argument"); return ARP_ERR_BAD_INPUT; }
    if (n >= 0x5000000ull) { set_error("too many atoms for one SAP neighbour sum (the kernel addresses the sorted records with 32-bit byte offsets: < 83886080 atoms)"); return ARP_ERR_BAD_INPUT; }
    if (!(sap_radius >= 0.0f)) { set_error("bad sap_radius"); return ARP_ERR_BAD_INPUT; }
    if (n == 0) return ARP_OK;
    // the grid machinery of the contact search, over the side-chain atoms only: everything else is kept out by the attribute bit that
    // keeps hydrogens out of the contact grid
    std::vector<uint32_t> attr(n), zero32(n, 0);
    for (uint64_t i = 0; i < n; i++) attr[i] = sidechain[i] ? (ARP_ATTR_LIGAND | ARP_ATTR_RECEPTOR) : ARP_ATTR_H;
    arp_atoms a{};
    a.n = n; a.x = x; a.y = y; a.z = z; a.attr = attr.data(); a.res_ord = zero32.data(); a.chain_rank = zero32.data(); a.model = zero32.data();
    a.n_res = 0; a.location = ARP_MEM_HOST;
    arp_params prm;
    arp_default_params(&prm);
    prm.dist_cutoff = (double)sap_radius;
    if ((s = ensure_workspace(ctx, n)) != ARP_OK) return s;
    DevAtoms d{};
    if ((s = stage_inputs(ctx, &a, &d)) != ARP_OK) return s;
    if ((s = upload_params(ctx, &prm)) != ARP_OK) return s;
    char *dev = nullptr, *pin = nullptr;
    if ((s = context_scratch(ctx, 0, 2 * ((n * 4 + 255u) & ~255ull), 2 * ((n * 4 + 255u) & ~255ull), &dev, &pin)) != ARP_OK) return s;
    float *d_w = (float *)dev, *d_out = (float *)(dev + ((n * 4 + 255u) & ~255ull));
    memcpy(pin, weight, n * 4);
    HIP_TRY(hipMemcpyAsync(d_w, pin, n * 4, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemsetAsync(d_out, 0, n * 4, ctx->stream));  // atoms outside the side-chain set keep 0
    const double r2 = (double)(sap_radius * sap_radius);  // sap.rs:183: the product is formed in f32
    launch_neighbor_sum(d, ctx->ws, (double)sap_radius, r2, d_w, d_out, ctx->stream, ctx->prof.enabled ? &ctx->prof : nullptr);
    HIP_TRY(hipGetLastError());
    float *h_out = (float *)(pin + ((n * 4 + 255u) & ~255ull));
    HIP_TRY(hipMemcpyAsync(h_out, d_out, n * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->h_result, ctx->ws.result, kResultWords * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->ws.grid) {  // non-finite coordinates are reported by the grid build through the fix-up kernel only; check here
        for (uint64_t i = 0; i < n; i++)
            if (sidechain[i] && !(std::isfinite(x[i]) && std::isfinite(y[i]) && std::isfinite(z[i]))) { set_error("non-finite atom coordinate"); return ARP_ERR_BAD_INPUT; }
    }
    memcpy(out, h_out, n * 4);
    return ARP_OK;
} ARP_ABI_CATCH

// ---- accessors for the table path (table_dev.hip) ---------------------------------------------------------------------------
namespace arp {
void *context_stream(arp_context *ctx) { return (void *)ctx->stream; }
int context_device(arp_context *ctx) { return ctx->device; }
bool context_grid(arp_context *ctx, const double *x, uint64_t n, const GridParams **grid, const uint32_t **cell_start, const Fat **fat) {
    if (!ctx || ctx->pending || !ctx->ws.grid || !x || ctx->grid_x != x || ctx->grid_n != n) return false;
    *grid = ctx->ws.grid; *cell_start = ctx->ws.cell_start; *fat = ctx->ws.sorted.fat;
    return true;
}
arp_status context_scratch(arp_context *ctx, int slot, uint64_t dev_bytes, uint64_t pinned_bytes, char **dev, char **pinned) {
    arp_status s = check_device(ctx);
    if (s != ARP_OK) return s;
    if (slot < 0 || slot > 1) { set_error("bad scratch slot"); return ARP_ERR_BAD_INPUT; }
    if (ctx->scr_dev_cap[slot] < dev_bytes) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->scr_dev[slot]) (void)hipFree(ctx->scr_dev[slot]);
        ctx->scr_dev[slot] = nullptr; ctx->scr_dev_cap[slot] = 0;
        const uint64_t cap = dev_bytes + dev_bytes / 4 + 4096;
        HIP_TRY(hipMalloc((void **)&ctx->scr_dev[slot], cap));
        ctx->scr_dev_cap[slot] = cap;
    }
    if (ctx->scr_pin_cap[slot] < pinned_bytes) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->scr_pin[slot]) (void)hipHostFree(ctx->scr_pin[slot]);
        ctx->scr_pin[slot] = nullptr; ctx->scr_pin_cap[slot] = 0;
        const uint64_t cap = pinned_bytes + pinned_bytes / 4 + 4096;
        HIP_TRY(hipHostMalloc((void **)&ctx->scr_pin[slot], cap, hipHostMallocDefault));
        ctx->scr_pin_cap[slot] = cap;
    }
    *dev = ctx->scr_dev[slot]; *pinned = ctx->scr_pin[slot];
    return ARP_OK;
}
}  // namespace arp

// ---- profiling ---------------------------------------------------------------------------------------------------
extern "C" arp_status arp_profile_enable(arp_context *ctx, int32_t on) {
    arp_status s = check_device(ctx);
    if (s != ARP_OK) return s;
    ctx->prof.enabled = on != 0;
    ctx->prof.n = 0;
    return ARP_OK;
}
extern "C" int32_t arp_profile_read(arp_context *ctx, const char **names, float *ms, int32_t cap) {
    if (check_device(ctx) != ARP_OK) return 0;
    (void)hipStreamSynchronize(ctx->stream);
    int n = std::min<int>(ctx->prof.n, cap);
    for (int k = 0; k < n; k++) {
        names[k] = ctx->prof.names[k];
        float t = 0.f;
        (void)hipEventElapsedTime(&t, ctx->prof.ev0[k], ctx->prof.ev1[k]);
        ms[k] = t;
    }
    return n;
}

// ---- library-level -----------------------------------------------------------------------------------------------
namespace arp { DebugKnobs g_debug{0, 0, 0, 0, 0}; }
extern "C" arp_status arp_debug_set(const char *key, int64_t value) {
    if (!key) { set_error("null key"); return ARP_ERR_BAD_INPUT; }
    const std::string k(key);
    if (k == "timing") g_debug.timing = value != 0;
    else if (k == "emit_kernel") g_debug.emit_kernel = (int)value;
    else if (k == "defer_entries") g_debug.defer_entries = (long)value;
    else if (k == "table_host") g_debug.table_host = value != 0;
    else if (k == "strip_rows") {
        if (value < 0 || value > 1024 || (value & (value - 1)) != 0) { set_error("arp_debug_set: strip_rows takes 0 or a power of two up to 1024"); return ARP_ERR_BAD_INPUT; }
        g_debug.strip_rows = (int)value;
    }
    else { set_error("arp_debug_set: unknown key '%s' (timing, emit_kernel, defer_entries, strip_rows, table_host)", key); return ARP_ERR_BAD_INPUT; }
    return ARP_OK;
}
extern "C" int32_t arp_api_version(void) { return ARP_API_VERSION; }
extern "C" arp_status arp_check_api_version(int32_t header_version) {
    if (header_version == ARP_API_VERSION) return ARP_OK;
    set_error("API version mismatch: the caller was compiled against version %d of arpeggia_amd.h, the library implements version %d "
              "(v2: arp_atoms.chain_rank and arp_atoms.model are uint32_t)", (int)header_version, (int)ARP_API_VERSION);
    return ARP_ERR_BAD_INPUT;
}
extern "C" const char *arp_last_error(void) { return g_err; }
extern "C" const char *arp_strerror(arp_status s) {
    switch (s) {
        case ARP_OK: return "ok";
        case ARP_ERR_BAD_GROUPS: return "Invalid chain groups format! Use '/' for all-to-all comparisons.";
        case ARP_ERR_EMPTY_GROUPS: return "Empty chain groups!";
        case ARP_ERR_NO_RINGS: return "Error building ring positions";
        case ARP_ERR_BAD_INPUT: return "bad input";
        case ARP_ERR_HIP: return "HIP runtime error";
        case ARP_ERR_OOM: return "out of memory";
        case ARP_ERR_NO_DEVICE: return "no gfx950 device (no CPU fallback)";
        case ARP_ERR_IO: return "I/O error";
        case ARP_ERR_CAPACITY: return "pair buffer too small";
        default: return "unknown status";
    }
}
extern "C" int32_t arp_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    int ok = 0;
    for (int d = 0; d < n; d++) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, d) != hipSuccess) continue;
        if (strncmp(p.gcnArchName, "gfx950", 6) == 0) ok++;
        else return ok;  // devices are addressed by ordinal: stop at the first foreign one
    }
    return ok;
}
static const char *k_interactions[ARP_N_INTERACTIONS] = {
    "StericClash", "CovalentBond", "Disulfide", "VanDerWaalsContact", "IonicBond", "HydrogenBond", "WeakHydrogenBond",
    "PolarContact", "WeakPolarContact", "IonicRepulsion", "SaltBridge", "PiDisplacedStacking", "PiTStacking",
    "PiSandwichStacking", "PiParallelInPlaneStacking", "PiTiltedStacking", "PiLStacking", "CationPi", "HydrophobicContact"};
extern "C" const char *arp_interaction_name(int32_t code) { return (code >= 0 && code < ARP_N_INTERACTIONS) ? k_interactions[code] : "?"; }
