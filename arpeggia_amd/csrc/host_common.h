// Host-side declarations shared by engine.cpp, structure.cpp and table.cpp.
#pragma once
#include <array>
#include <cstdarg>
#include <cstdint>
#include <exception>
#include <mutex>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/arpeggia_amd.h"
#include "debug_knobs.h"

namespace arp {

void set_error(const char *fmt, ...);

// No exception may cross the C ABI (SURVEY.md 8b "Errors": the reference panics, a C boundary returns a status).  Every status-returning entry
// point that allocates or starts threads is a function-try-block that ends in this: `extern "C" arp_status f(...) try { ... } ARP_ABI_CATCH`.
#define ARP_ABI_CATCH                                                                                                               \
    catch (const std::bad_alloc &) { arp::set_error("out of host memory"); return ARP_ERR_OOM; }                                   \
    catch (const std::exception &e_) { arp::set_error("internal error: %s", e_.what()); return ARP_ERR_HIP; }                       \
    catch (...) { arp::set_error("internal error: unknown exception"); return ARP_ERR_HIP; }

// Fixed-width, NUL-padded string column (n x W chars), the layout the C ABI hands to numpy / Rust.
template <int W>
struct StrCol {
    std::vector<char> buf;
    void resize(size_t n) { buf.assign(n * W, 0); }
    size_t size() const { return buf.size() / W; }
    const char *at(size_t i) const { return &buf[i * W]; }
    char *at(size_t i) { return &buf[i * W]; }
    void set(size_t i, const char *s) {
        char *d = at(i);
        int k = 0;
        for (; k < W - 1 && s[k]; k++) d[k] = s[k];
        for (; k < W; k++) d[k] = 0;
    }
    std::string str(size_t i) const { return std::string(at(i)); }
};

struct Plane {
    double c[3], n[3];
};

struct ChainInfo {
    uint32_t model_idx;
    int32_t model_serial;
    std::string id;
};

struct ResidueInfo {
    uint32_t chain;       // index into chains
    int32_t resi;
    std::string icode;
    std::string name;     // Residue::name(): common conformer name
    uint32_t ord;         // positional index inside the chain (complex.rs:411-440)
    std::vector<uint32_t> atoms;         // atom indices in hierarchy order (conformer ordinal, then input order)
    std::vector<std::string> altlocs;    // distinct conformer altlocs in order of appearance
};

}  // namespace arp

// The parsed, filtered model (what `load_model` returns in the reference, utils.rs:51-63), as SoA columns.
struct arp_structure {
    uint64_t n = 0;
    std::vector<double> x, y, z, occ;
    std::vector<int32_t> serial, resi, model_serial;
    arp::StrCol<8> name, resn /* conformer */, res_resn /* residue */, chain;
    arp::StrCol<4> altloc, icode, elem;
    std::vector<uint32_t> res_ord, res_id, base_attr, attr;
    std::vector<uint32_t> chain_rank, model;
    std::vector<uint32_t> atom_chain;  // index into chains
    std::vector<arp::ChainInfo> chains;
    std::vector<arp::ResidueInfo> residues;
    std::vector<std::string> chain_ids;  // distinct ids, byte-wise sorted: chain_rank indexes this
    std::vector<uint32_t> res_h_ptr, res_h_idx, res_cb, res_sg;
    std::string groups_applied;
    bool groups_valid = false;
    // table.cpp: derived per-structure tables of the table path + the device-resident copy (built on the first arp_get_contacts)
    void *table_cache = nullptr;
    std::mutex table_cache_mu;   // guards the creation of table_cache
    void (*table_cache_free)(void *) = nullptr;
    ~arp_structure() { if (table_cache && table_cache_free) table_cache_free(table_cache); }
};

namespace arp {
arp_status parse_groups(const std::vector<std::string> &all_chains, const char *groups, std::vector<std::string> *ligand,
                        std::vector<std::string> *receptor);
arp_status apply_groups(arp_structure *s, const char *groups);
bool fit_plane(const std::vector<std::array<double, 3>> &pts, Plane *out);

// Host worker threads for the table path (planes, rows, sort, columns) -- the counterpart of the reference's global rayon pool
// (utils.rs:8-30, python.rs num_threads).  1 = serial (the reference's default), 0 = all hardware threads.
int host_threads();
void set_host_threads(int n);
// Pins the worker count of the calling thread for the lifetime of the scope: n > 0 that many, n == 0 all hardware threads,
// n < 0 a snapshot of the process-wide default.  parallel_for() is only ever called from the thread that owns the scope.
struct HostThreadsScope {
    int prev;
    explicit HostThreadsScope(int n);
    ~HostThreadsScope();
    HostThreadsScope(const HostThreadsScope &) = delete;
    HostThreadsScope &operator=(const HostThreadsScope &) = delete;
};
// fn(begin, end, worker) over [0, n) in contiguous slices, one per worker; serial below `min_per_worker` items per worker.
template <class F>
void parallel_for(size_t n, size_t min_per_worker, F &&fn) {
    size_t workers = (size_t)host_threads();
    if (min_per_worker && n / min_per_worker < workers) workers = n / min_per_worker;
    if (workers <= 1) { fn((size_t)0, n, (size_t)0); return; }
    // An exception must not leave a std::thread's function (std::terminate) nor unwind past joinable threads: a worker parks its exception,
    // the guard joins on every path, and the caller rethrows the first one -- it then reaches the C ABI's catch like any exception of the
    // calling thread.
    std::vector<std::thread> th;
    th.reserve(workers);
    std::exception_ptr first_error;
    std::mutex error_mu;
    auto guarded = [&](size_t w) noexcept {
        try { fn(n * w / workers, n * (w + 1) / workers, w); }
        catch (...) { std::lock_guard<std::mutex> lk(error_mu); if (!first_error) first_error = std::current_exception(); }
    };
    {
        struct JoinAll { std::vector<std::thread> &t; ~JoinAll() { for (auto &x : t) if (x.joinable()) x.join(); } } join_all{th};
        size_t started = 0;  // slices [0, started) run on threads of their own; the caller takes the last one and any that could not get a thread
        for (; started + 1 < workers; started++) {
            const size_t w = started;
            try { th.emplace_back([&guarded, w]() { guarded(w); }); } catch (const std::system_error &) { break; }
        }
        for (size_t w = started; w < workers; w++) guarded(w);
    }
    if (first_error) std::rethrow_exception(first_error);
}
}  // namespace arp
