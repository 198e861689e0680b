// Table assembly: the product-side equivalent of `arpeggia::get_contacts` (src/contacts/mod.rs:61-137).
// Atom-atom rows come from the GPU pair list (arp_contacts_atomic); ring planes, the low-volume ring rows
// (complex.rs:301-405), side-chain plane statistics (complex.rs:137-174) and the 10-key sort (mod.rs:120-134) are
// assembled here on the host in round 1 (SURVEY.md 8f rows f1/f2 move them to the device next).
#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include <hip/hip_runtime.h>

#include "host_common.h"
#include "table_dev.h"

namespace arp {

#ifdef ARP_WITH_HOST_TABLE   // host-side plane maths: only the test-only host assembly (tests/hosttable/table_host.inl) fits planes on the CPU
// ---- plane maths (residues.rs:24-75, 270-298) ----------------------------------------------------------------------
// Least-squares plane: centroid + eigenvector of the smallest eigenvalue of the 3x3 scatter matrix (cyclic Jacobi).
// nalgebra's svd.u.column(2) is the same direction up to sign; every use folds the angle into [0, 90] degrees.
bool fit_plane(const std::vector<std::array<double, 3>> &pts, Plane *out) {
    const size_t n = pts.size();
    if (n < 3) return false;  // residues.rs:273
    double c[3] = {0, 0, 0};
    for (auto &p : pts) for (int k = 0; k < 3; k++) c[k] += p[k];
    for (int k = 0; k < 3; k++) c[k] /= (double)n;
    double A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (auto &p : pts) {
        double d[3] = {p[0] - c[0], p[1] - c[1], p[2] - c[2]};
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) A[i][j] += d[i] * d[j];
    }
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 64; sweep++) {
        double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
        double diag = fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]);
        if (off <= 1e-300 || off <= 1e-18 * diag) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                if (fabs(A[p][q]) <= 1e-300) continue;
                double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < 3; k++) { double akp = A[k][p], akq = A[k][q]; A[k][p] = cs * akp - sn * akq; A[k][q] = sn * akp + cs * akq; }
                for (int k = 0; k < 3; k++) { double apk = A[p][k], aqk = A[q][k]; A[p][k] = cs * apk - sn * aqk; A[q][k] = sn * apk + cs * aqk; }
                for (int k = 0; k < 3; k++) { double vkp = V[k][p], vkq = V[k][q]; V[k][p] = cs * vkp - sn * vkq; V[k][q] = sn * vkp + cs * vkq; }
            }
    }
    int best = 0;
    for (int k = 1; k < 3; k++) if (A[k][k] < A[best][best]) best = k;
    double nn = sqrt(V[0][best] * V[0][best] + V[1][best] * V[1][best] + V[2][best] * V[2][best]);
    for (int k = 0; k < 3; k++) { out->c[k] = c[k]; out->n[k] = V[k][best] / nn; }
    return true;
}
static const double kRad2Deg = 180.0 / 3.14159265358979323846264338327950288;
static double norm3(const double v[3]) { return sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
static double fold_deg(double rad) {  // residues.rs:50-53,70-73
    if (rad > 1.57079632679489661923) rad = 3.14159265358979323846 - rad;
    return rad * kRad2Deg;
}
static double point_dist(const Plane &p, const double q[3]) { double v[3] = {q[0] - p.c[0], q[1] - p.c[1], q[2] - p.c[2]}; return norm3(v); }
static double point_angle(const Plane &p, const double q[3]) {
    double v[3] = {q[0] - p.c[0], q[1] - p.c[1], q[2] - p.c[2]};
    double dot = p.n[0] * v[0] + p.n[1] * v[1] + p.n[2] * v[2];
    return fold_deg(acos(dot / (norm3(p.n) * norm3(v))));
}
static double plane_dihedral(const Plane &a, const Plane &b) {
    double dot = a.n[0] * b.n[0] + a.n[1] * b.n[1] + a.n[2] * b.n[2];
    return fold_deg(acos(dot / (norm3(a.n) * norm3(b.n))));
}

#endif

// ---- plane tables (complex.rs:442-514) ------------------------------------------------------------------------------
static bool in_words(const char *words, const char *w) {
    size_t L = strlen(w);
    for (const char *p = words; *p;) {
        const char *e = strchr(p, ' ');
        size_t len = e ? (size_t)(e - p) : strlen(p);
        if (len == L && strncmp(p, w, L) == 0) return true;
        p += len;
        while (*p == ' ') p++;
    }
    return false;
}
static const char *ring_atoms_of(const std::string &resn) {  // residues.rs:163-186
    if (resn == "HIS") return "CG ND1 CE1 NE2 CD2";
    if (resn == "PHE" || resn == "TYR") return "CG CD1 CD2 CE1 CE2 CZ";
    if (resn == "TRP") return "CG CD1 CD2 NE1 CE2 CE3 CZ2 CZ3 CH2";
    return nullptr;
}
static const char *sc_atoms_of(const std::string &resn) {  // residues.rs:188-268
    static const std::map<std::string, const char *> m = {
        {"ARG", "NE CZ NH1 NH2"}, {"ASN", "CB CG OD1 ND2"}, {"ASP", "CB CG OD1 OD2"}, {"CYS", "CA CB SG"}, {"GLU", "CG CD OE1 OE2"},
        {"GLN", "CG CD OE1 NE2"}, {"ILE", "CB CG1 CG2 CD1"}, {"LEU", "CB CG CD1 CD2"}, {"LYS", "CG CD CE NZ"}, {"MET", "CG SD CE"},
        {"PRO", "N CA CB CG CD"}, {"SER", "CA CB OG"}, {"THR", "CA CB OG1 CG2"}, {"VAL", "CA CB CG1 CG2"}};
    if (const char *r = ring_atoms_of(resn)) return r;
    auto it = m.find(resn);
    return it == m.end() ? nullptr : it->second;
}

struct PlaneEntry {
    int32_t model_serial; std::string chain; int32_t resi; std::string icode, altloc, resn;
    Plane plane;
    bool has_ord = false; uint32_t ord = 0;   // res2idx[(model, chain, resi, icode, altloc, resn)]
    uint32_t res = 0, alt_k = 0;              // the residue (and which of its conformer altlocs) the plane was last written from
    uint32_t chain_rank = 0; bool in_l = false, in_r = false;
};
// (model serial, chain id, resi, icode, altloc, resn) with the names packed into integers (ids are <= 7 / 3 characters, the
// widths of the string columns): hashing and comparing keys costs no allocation.
struct PlaneKey {
    int32_t model, resi;
    uint32_t icode, altloc;
    uint64_t chain, resn;
    bool operator==(const PlaneKey &o) const {
        return model == o.model && resi == o.resi && icode == o.icode && altloc == o.altloc && chain == o.chain && resn == o.resn;
    }
};
struct PlaneKeyHash {
    size_t operator()(const PlaneKey &k) const {
        uint64_t h = k.chain * 0x9E3779B97F4A7C15ull ^ k.resn;
        h = (h ^ (h >> 29)) * 0xBF58476D1CE4E5B9ull + (((uint64_t)(uint32_t)k.model << 32) | (uint32_t)k.resi);
        h = (h ^ (h >> 31)) * 0x94D049BB133111EBull + (((uint64_t)k.icode << 32) | k.altloc);
        return (size_t)(h ^ (h >> 32));
    }
};
using PlaneIndex = std::unordered_map<PlaneKey, size_t, PlaneKeyHash>;
static uint64_t pack_name(const char *s, size_t cap) {  // up to `cap` bytes, NUL padded
    uint64_t v = 0;
    size_t n = 0;
    while (n < cap && s[n]) n++;
    memcpy(&v, s, n);
    return v;
}
static PlaneKey plane_key(int32_t model, const char *chain, int32_t resi, const char *icode, const char *altloc, const char *resn) {
    return PlaneKey{model, resi, (uint32_t)pack_name(icode, 4), (uint32_t)pack_name(altloc, 4), pack_name(chain, 8), pack_name(resn, 8)};
}

// `first_of_res` (single-model structures only, else left empty): entry of (residue r, its altloc k) = first_of_res[r] + k, -1 if r has
// no plane -- one model holds one residue per (chain, resi, icode), so no key can be written twice and no keyed lookup is needed.
// has_plane[r]: the residue has >= 3 plane atoms (residues.rs:273: the only way a fit fails).  fitted == nullptr: the planes
// themselves are fitted on the device (table_dev.hip); the entries then only say WHICH residue's plane an entity uses (e.res).
static void build_planes(const arp_structure &s, bool rings, const std::vector<char> &has_plane, const std::vector<Plane> *fitted_in, std::vector<PlaneEntry> *out,
                         PlaneIndex *index, std::vector<int64_t> *first_of_res) {
    std::vector<int32_t> serials;
    for (const ChainInfo &c : s.chains) if (std::find(serials.begin(), serials.end(), c.model_serial) == serials.end()) serials.push_back(c.model_serial);
    // res2idx: (model serial, chain, resi, icode) -> residue
    // With ONE model serial a plane entry can only resolve to the residue it was fitted from (the hierarchy holds one residue
    // per (chain, resi, icode)); the keyed lookup below is needed for multi-model files only.
    const bool one_model = serials.size() <= 1;
    const bool direct = s.chains.empty() || s.chains.back().model_idx == 0;  // a single model in the file
    if (direct) first_of_res->assign(s.residues.size(), -1);
    std::unordered_map<PlaneKey, uint32_t, PlaneKeyHash> res_of;
    if (!one_model) {
        res_of.reserve(s.residues.size() * 2);
        for (uint32_t r = 0; r < s.residues.size(); r++) {
            const ResidueInfo &ri = s.residues[r];
            res_of[plane_key(s.chains[ri.chain].model_serial, s.chains[ri.chain].id.c_str(), ri.resi, ri.icode.c_str(), "", "")] = r;
        }
    }
    if (!direct) index->reserve(s.residues.size() * 2);
    // complex.rs:447-449 / 489-492: for EVERY model serial, ALL chains of ALL models are visited; later inserts overwrite
    // the plane of a residue does not depend on the model serial it is filed under: fit once, file per serial
    static const std::vector<Plane> no_planes;
    const std::vector<Plane> &fitted = fitted_in ? *fitted_in : no_planes;
    for (int32_t m : serials)
        for (uint32_t r = 0; r < s.residues.size(); r++) {
            const ResidueInfo &ri = s.residues[r];
            if (!has_plane[r]) continue;
            const Plane pl = fitted_in ? fitted[r] : Plane{};
            if (direct) (*first_of_res)[r] = (int64_t)out->size();
            for (uint32_t k = 0; k < ri.altlocs.size(); k++) {
                const std::string &alt = ri.altlocs[k];
                size_t slot;
                if (direct) { slot = out->size(); out->push_back(PlaneEntry{}); }
                else {
                    const PlaneKey key = plane_key(m, s.chains[ri.chain].id.c_str(), ri.resi, ri.icode.c_str(), alt.c_str(), ri.name.c_str());
                    auto it = index->find(key);
                    if (it == index->end()) { it = index->emplace(key, out->size()).first; out->push_back(PlaneEntry{}); }
                    slot = it->second;
                }
                PlaneEntry &e = (*out)[slot];
                e.model_serial = m; e.chain = s.chains[ri.chain].id; e.resi = ri.resi; e.icode = ri.icode; e.altloc = alt; e.resn = ri.name;
                e.plane = pl; e.res = r; e.alt_k = k;
                if (one_model) { e.has_ord = true; e.ord = ri.ord; }
            }
        }
    if (!one_model) for (PlaneEntry &e : *out) {
        auto it = res_of.find(plane_key(e.model_serial, e.chain.c_str(), e.resi, e.icode.c_str(), "", ""));
        if (it == res_of.end()) continue;
        const ResidueInfo &ri = s.residues[it->second];
        if (ri.name != e.resn || std::find(ri.altlocs.begin(), ri.altlocs.end(), e.altloc) == ri.altlocs.end()) continue;
        e.has_ord = true; e.ord = ri.ord;
    }
}

// plane atoms per residue: bit 1 = ring-plane atom (residues.rs:163-186), bit 2 = sc-plane atom (residues.rs:188-268)
static void plane_atom_bits(const arp_structure &s, std::vector<uint8_t> *bits, std::vector<char> *has_ring, std::vector<char> *has_sc) {
    bits->assign(s.n, 0);
    has_ring->assign(s.residues.size(), 0); has_sc->assign(s.residues.size(), 0);
    parallel_for(s.residues.size(), 512, [&](size_t r0, size_t r1, size_t) {
        for (size_t r = r0; r < r1; r++) {
            const ResidueInfo &ri = s.residues[r];
            const char *rn = ring_atoms_of(ri.name), *sn = sc_atoms_of(ri.name);
            if (!rn && !sn) continue;
            uint32_t nr = 0, ns = 0;
            for (uint32_t a : ri.atoms) {
                uint8_t b = 0;
                if (rn && in_words(rn, s.name.at(a))) { b |= 1u; nr++; }
                if (sn && in_words(sn, s.name.at(a))) { b |= 2u; ns++; }
                (*bits)[a] = b;
            }
            (*has_ring)[r] = nr >= 3; (*has_sc)[r] = ns >= 3;  // residues.rs:273
        }
    });
}
#ifdef ARP_WITH_HOST_TABLE
static void fit_planes_host(const arp_structure &s, const std::vector<uint8_t> &bits, uint8_t want, std::vector<Plane> *fitted) {
    fitted->assign(s.residues.size(), Plane{});
    parallel_for(s.residues.size(), 512, [&](size_t r0, size_t r1, size_t) {
        std::vector<std::array<double, 3>> pts;
        for (size_t r = r0; r < r1; r++) {
            pts.clear();
            for (uint32_t a : s.residues[r].atoms) if (bits[a] & want) pts.push_back({s.x[a], s.y[a], s.z[a]});
            if (pts.size() >= 3) fit_plane(pts, &(*fitted)[r]);
        }
    });
}

// should_compare_residues (complex.rs:94-131) on prepared keys
struct ResKey { int32_t model_serial; uint32_t chain_rank; uint32_t ord; bool in_l, in_r; };
static bool compare_residues(const ResKey &a, const ResKey &b, bool symmetric) {
    if (a.model_serial != b.model_serial) return false;
    if (!((a.in_l && b.in_r) || (b.in_l && a.in_r))) return false;
    if (a.chain_rank == b.chain_rank) {
        if (symmetric) return (b.ord > 1) && (a.ord < b.ord - 1);
        bool neigh = (a.ord == 0) ? (b.ord == a.ord || b.ord == a.ord + 1) : (b.ord == a.ord - 1 || b.ord == a.ord || b.ord == a.ord + 1);
        return !neigh;
    }
    return !(symmetric && a.in_r && b.in_r && a.in_l && b.in_l && a.chain_rank > b.chain_rank);
}

#endif

}  // namespace arp

using namespace arp;

// One entity of the output vocabulary (structs.rs:55-70): an atom, or a ring (atomn "Ring", atomi 0: complex.rs:334-342).
// Fixed NUL-padded names (the widths of the table's string columns).  48 bytes: a row touches two of these.
struct EntityRec {
    char chain[8], resn[8], atomn[8], insertion[4], altloc[4];
    int32_t resi, atomi, atom;   // atom: index into the structure's atoms, -1 for a ring
    uint32_t model;              // model serial as the table shows it (mod.rs:143)
};
static_assert(sizeof(EntityRec) == 48, "EntityRec layout");
struct EntityBook {              // every entity a table of this structure can mention; immutable, shared by the tables (which may outlive the structure)
    // (plain arrays, not vectors: 52 bytes per entity that every worker writes once -- a vector would zero them first, on one thread)
    std::unique_ptr<EntityRec[]> ent;  // atoms, then rings
    std::unique_ptr<uint32_t[]> lens;  // string lengths of an entity, 4 bits each: chain, resn, atomn, insertion, altloc
    size_t n = 0;
};

struct arp_table {
    uint64_t n = 0;
    // device path: the table as it comes back -- rows of {from entity, to entity, distance, interaction} + side-chain statistics -- and the
    // entity book; the fixed-width columns below are materialised on first access (arp_table_column), the Arrow export reads rows + book
    std::shared_ptr<const EntityBook> book;
    std::shared_ptr<char> rows_owner;          // the pooled pinned block (or heap block) rows and sc live in
    const TableRow *rows = nullptr;
    const TableSc *sc = nullptr;
    std::once_flag columns_once;
    std::vector<uint32_t> model;
    std::vector<int32_t> interaction, from_resi, from_atomi, to_resi, to_atomi, from_atom, to_atom;
    std::vector<float> distance, sc_dist, sc_dihedral, sc_angle;
    std::vector<uint8_t> sc_valid;
    StrCol<8> from_chain, from_resn, from_atomn, to_chain, to_resn, to_atomn;
    StrCol<4> from_insertion, from_altloc, to_insertion, to_altloc;
};

namespace {
struct Entity {  // structs.rs:55-70; fixed NUL-padded names (the widths of the table's string columns): rows stay POD
    char chain[8], resn[8], atomn[8], insertion[4], altloc[4];
    int32_t resi = 0, atomi = 0, atom = -1;
    uint32_t chain_rank = 0;
    int64_t sc_plane = -1;
};
struct Row {
    uint32_t model; int32_t interaction; double distance;
    Entity from, to;
};
}  // namespace


// ---- the device table path (SURVEY.md 8f rows f1 + f2) ------------------------------------------------------------------------
// Host: which entities exist and their names (group-independent parts are kept with the structure, so is the device-resident copy
// of its arrays).  Device (table_dev.hip): plane fits, ring rows, row expansion, the sort, the side-chain plane statistics.
namespace {
struct TableCache {
    std::vector<uint8_t> plane_bits;
    std::vector<char> has_ring, has_sc;
    std::vector<uint32_t> res_atom_ptr, res_atom_idx, atom_sc_src, model_rank;
    std::vector<int32_t> model_serial_of;
    std::vector<EntKey> atom_keys;
    std::vector<PlaneEntry> rings, scp;       // entities (no planes: those are fitted on the device)
    PlaneIndex ring_idx, sc_idx;
    std::vector<int64_t> ring_first, sc_first;
    bool direct = true;
    std::vector<uint32_t> ring_sc_src;        // per ring entity: residue whose sc plane applies, or ARP_NONE
    std::vector<EntKey> ring_keys;
    std::vector<uint32_t> ring_model_rank;
    std::shared_ptr<EntityBook> book;
    std::thread book_job;                     // fills `book` while the first call's device work runs; joined before a table gets the book
    std::exception_ptr book_error;            // what the job threw, if anything: rethrown on the calling thread by wait_for_book
    void wait_for_book() {
        if (book_job.joinable()) book_job.join();
        if (book_error) { std::exception_ptr e = book_error; book_error = nullptr; std::rethrow_exception(e); }
    }
    void join_book() noexcept { if (book_job.joinable()) book_job.join(); }
    std::mutex mu;                            // held for the whole of a device-table call on this structure
    std::string rings_groups; bool have_rings_dev = false;   // the ring entities as the device wants them, for this chain-group spec
    std::vector<RingEnt> rings_dev;
    DevStructure dev;
    ~TableCache() { join_book(); if (dev.block) { (void)hipSetDevice(dev.device); (void)hipFree(dev.block); if (dev.derived) (void)hipFree(dev.derived); if (dev.rings_block) (void)hipFree(dev.rings_block); } }
};
void free_table_cache(void *p) { delete (TableCache *)p; }
uint32_t be32(const char *p) { return ((uint32_t)(unsigned char)p[0] << 24) | ((uint32_t)(unsigned char)p[1] << 16) | ((uint32_t)(unsigned char)p[2] << 8) | (uint32_t)(unsigned char)p[3]; }
uint32_t be32s(const std::string &v) { char b[4] = {0, 0, 0, 0}; memcpy(b, v.data(), std::min<size_t>(3, v.size())); return be32(b); }

// (Calls on ONE structure are serialised: its cache and its resident device copy are per-structure state.  Different structures run in parallel.)
TableCache *table_cache_of(arp_structure *s) {
    std::lock_guard<std::mutex> guard(s->table_cache_mu);
    if (s->table_cache) return (TableCache *)s->table_cache;
    std::unique_ptr<TableCache> holder(new TableCache());  // (published at the end: an exception on the way leaks nothing, and its job is joined)
    TableCache *c = holder.get();
    const size_t n = s->n, nr = s->residues.size();
    const bool timing = g_debug.timing != 0;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "    table cache %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    plane_atom_bits(*s, &c->plane_bits, &c->has_ring, &c->has_sc);
    lap("plane bits");
    c->res_atom_ptr.assign(nr + 1, 0);
    for (size_t r = 0; r < nr; r++) c->res_atom_ptr[r + 1] = c->res_atom_ptr[r] + (uint32_t)s->residues[r].atoms.size();
    c->res_atom_idx.resize(c->res_atom_ptr[nr]);
    parallel_for(nr, 4096, [&](size_t r0, size_t r1, size_t) {
        for (size_t r = r0; r < r1; r++) std::copy(s->residues[r].atoms.begin(), s->residues[r].atoms.end(), c->res_atom_idx.begin() + c->res_atom_ptr[r]);
    });
    lap("residue CSR");
    build_planes(*s, true, c->has_ring, nullptr, &c->rings, &c->ring_idx, &c->ring_first);
    lap("ring entities");
    // A file with a single model holds one residue per (chain, resi, icode): an entity's side-chain plane can only be its own residue's,
    // and the keyed entries (one per residue and conformer) are needed for multi-model files only.
    c->direct = s->chains.empty() || s->chains.back().model_idx == 0;
    if (!c->direct) build_planes(*s, false, c->has_sc, nullptr, &c->scp, &c->sc_idx, &c->sc_first);
    lap("sc-plane entities");
    // model tables: ordinal -> serial, and the rank of the serial (sort key `model`, mod.rs:122)
    {
        const uint32_t nm = s->chains.empty() ? 1u : s->chains.back().model_idx + 1u;
        c->model_serial_of.assign(nm, 0);
        for (const ChainInfo &ch : s->chains) c->model_serial_of[ch.model_idx] = ch.model_serial;
        std::vector<uint32_t> serials;
        for (int32_t v : c->model_serial_of) serials.push_back((uint32_t)v);  // the column is u32 (mod.rs:143)
        std::sort(serials.begin(), serials.end());
        serials.erase(std::unique(serials.begin(), serials.end()), serials.end());
        c->model_rank.resize(nm);
        for (uint32_t m = 0; m < nm; m++) c->model_rank[m] = (uint32_t)(std::lower_bound(serials.begin(), serials.end(), (uint32_t)c->model_serial_of[m]) - serials.begin());
        c->ring_model_rank.resize(c->rings.size());
        for (size_t k = 0; k < c->rings.size(); k++)
            c->ring_model_rank[k] = (uint32_t)(std::lower_bound(serials.begin(), serials.end(), (uint32_t)c->rings[k].model_serial) - serials.begin());
    }
    // per-atom entity keys and the residue whose side-chain plane applies (the join key of mod.rs:100-110 plus resn, as in collect_sc_stats)
    c->atom_keys.resize(n);
    c->atom_sc_src.assign(n, ARP_NONE);
    parallel_for(n, 1u << 15, [&](size_t a0, size_t a1, size_t) {
        for (size_t a = a0; a < a1; a++) c->atom_keys[a] = EntKey{s->resi[a], be32(s->altloc.at(a)), s->serial[a], be32(s->icode.at(a))};
    });
    parallel_for(nr, 2048, [&](size_t r0, size_t r1, size_t) {
        std::vector<uint32_t> src_of_alt;
        for (size_t r = r0; r < r1; r++) {
            const ResidueInfo &ri = s->residues[r];
            if (ri.atoms.empty()) continue;
            const uint32_t a0 = ri.atoms[0];
            src_of_alt.assign(ri.altlocs.size(), ARP_NONE);
            for (size_t k = 0; k < ri.altlocs.size(); k++) {
                if (c->direct) { if (c->has_sc[r]) src_of_alt[k] = (uint32_t)r; continue; }
                auto f = c->sc_idx.find(plane_key(s->model_serial[a0], s->chain.at(a0), s->resi[a0], s->icode.at(a0), ri.altlocs[k].c_str(), s->res_resn.at(a0)));
                if (f != c->sc_idx.end()) src_of_alt[k] = c->scp[f->second].res;
            }
            for (uint32_t a : ri.atoms)
                for (size_t k = 0; k < ri.altlocs.size(); k++)
                    if (strcmp(s->altloc.at(a), ri.altlocs[k].c_str()) == 0) { c->atom_sc_src[a] = src_of_alt[k]; break; }
        }
    });
    c->ring_sc_src.assign(c->rings.size(), ARP_NONE);
    c->ring_keys.resize(c->rings.size());
    for (size_t k = 0; k < c->rings.size(); k++) {
        const PlaneEntry &r = c->rings[k];
        if (c->direct) { if (c->has_sc[r.res]) c->ring_sc_src[k] = r.res; }  // same residue, same conformer altloc
        else {
            auto f = c->sc_idx.find(plane_key(r.model_serial, r.chain.c_str(), r.resi, r.icode.c_str(), r.altloc.c_str(), r.resn.c_str()));
            if (f != c->sc_idx.end()) c->ring_sc_src[k] = c->scp[f->second].res;
        }
        c->ring_keys[k] = EntKey{r.resi, be32s(r.altloc), 0, be32s(r.icode)};  // complex.rs:334-342: atomi 0
    }
    lap("keys + sc sources");
    // The entity book: names and numbers of every atom and ring, gathered once.  Only the tables' string columns read it, so it is filled
    // by a job of its own while the first call's uploads and kernels run; get_contacts_device waits for it before it hands out a table.
    c->book = std::make_shared<EntityBook>();
    {
        EntityBook *bk = c->book.get();
        bk->n = n + c->rings.size();
        bk->ent.reset(new EntityRec[bk->n]);
        bk->lens.reset(new uint32_t[bk->n]);
        const int workers = host_threads();
        const arp_structure *sp = s;
        const TableCache *cp = c;
        auto fill = [bk, sp, cp, workers]() {
            HostThreadsScope scope(workers);
            const size_t n_atoms = sp->n;
            auto lens_of = [](const EntityRec &e) {
                auto len = [](const char *p, int w) { int k = 0; while (k < w && p[k]) k++; return (uint32_t)k; };
                return len(e.chain, 8) | (len(e.resn, 8) << 4) | (len(e.atomn, 8) << 8) | (len(e.insertion, 4) << 12) | (len(e.altloc, 4) << 16);
            };
            parallel_for(n_atoms, 1u << 14, [&](size_t a0, size_t a1, size_t) {
                for (size_t a = a0; a < a1; a++) {  // structs.rs:109-119
                    EntityRec &e = bk->ent[a];
                    memcpy(e.chain, sp->chain.at(a), 8); memcpy(e.resn, sp->res_resn.at(a), 8); memcpy(e.atomn, sp->name.at(a), 8);
                    memcpy(e.insertion, sp->icode.at(a), 4); memcpy(e.altloc, sp->altloc.at(a), 4);
                    e.resi = sp->resi[a]; e.atomi = sp->serial[a]; e.atom = (int32_t)a; e.model = (uint32_t)sp->model_serial[a];
                    bk->lens[a] = lens_of(e);
                }
            });
            for (size_t k = 0; k < cp->rings.size(); k++) {  // complex.rs:334-342
                const PlaneEntry &r = cp->rings[k];
                EntityRec &e = bk->ent[n_atoms + k];
                auto put = [](char *dst, size_t cap, const std::string &v) { memset(dst, 0, cap); memcpy(dst, v.data(), std::min(cap - 1, v.size())); };
                put(e.chain, 8, r.chain); put(e.resn, 8, r.resn); put(e.atomn, 8, "Ring"); put(e.insertion, 4, r.icode); put(e.altloc, 4, r.altloc);
                e.resi = r.resi; e.atomi = 0; e.atom = -1; e.model = (uint32_t)r.model_serial;
                bk->lens[n_atoms + k] = lens_of(e);
            }
        };
        // (an exception must not leave the thread's function: it is parked and rethrown by wait_for_book on the thread that asks for the book)
        TableCache *cw = c;
        auto job = [fill, cw]() noexcept { try { fill(); } catch (...) { cw->book_error = std::current_exception(); } };
        try { c->book_job = std::thread(job); } catch (const std::system_error &) { fill(); }  // (no thread to be had: filled here)
    }
    lap("entity book (job started)");
    s->table_cache = holder.release(); s->table_cache_free = free_table_cache;
    return c;
}

// the fixed-width columns of a device-path table, all at once, on first access
void materialize_columns(arp_table *t) {
    const size_t nrow = t->n;
    const EntityBook &bk = *t->book;
    t->model.resize(nrow); t->interaction.resize(nrow); t->from_resi.resize(nrow); t->from_atomi.resize(nrow); t->to_resi.resize(nrow); t->to_atomi.resize(nrow);
    t->from_atom.resize(nrow); t->to_atom.resize(nrow); t->distance.resize(nrow);
    t->sc_dist.resize(nrow); t->sc_dihedral.resize(nrow); t->sc_angle.resize(nrow); t->sc_valid.resize(nrow);
    t->from_chain.resize(nrow); t->from_resn.resize(nrow); t->from_atomn.resize(nrow); t->to_chain.resize(nrow); t->to_resn.resize(nrow); t->to_atomn.resize(nrow);
    t->from_insertion.resize(nrow); t->from_altloc.resize(nrow); t->to_insertion.resize(nrow); t->to_altloc.resize(nrow);
    parallel_for(nrow, 1u << 14, [&](size_t k0, size_t k1, size_t) {
        for (size_t k = k0; k < k1; k++) {
            const TableRow &r = t->rows[k];
            const TableSc &c = t->sc[k];
            const EntityRec &f = bk.ent[r.from_ent], &o = bk.ent[r.to_ent];
            t->model[k] = f.model; t->interaction[k] = r.interaction; t->distance[k] = r.distance;
            memcpy(t->from_chain.at(k), f.chain, 8); memcpy(t->from_resn.at(k), f.resn, 8); memcpy(t->from_atomn.at(k), f.atomn, 8);
            memcpy(t->from_insertion.at(k), f.insertion, 4); memcpy(t->from_altloc.at(k), f.altloc, 4);
            t->from_resi[k] = f.resi; t->from_atomi[k] = f.atomi; t->from_atom[k] = f.atom;
            memcpy(t->to_chain.at(k), o.chain, 8); memcpy(t->to_resn.at(k), o.resn, 8); memcpy(t->to_atomn.at(k), o.atomn, 8);
            memcpy(t->to_insertion.at(k), o.insertion, 4); memcpy(t->to_altloc.at(k), o.altloc, 4);
            t->to_resi[k] = o.resi; t->to_atomi[k] = o.atomi; t->to_atom[k] = o.atom;
            t->sc_valid[k] = c.valid != 0.f ? 1 : 0; t->sc_dist[k] = c.dist; t->sc_dihedral[k] = c.dihedral; t->sc_angle[k] = c.angle;
        }
    });
}

#define TBL_HIP(expr)                                                                                                       \
    do {                                                                                                                    \
        hipError_t e_ = (expr);                                                                                             \
        if (e_ != hipSuccess) {                                                                                             \
            set_error("HIP error %d (%s) at %s:%d: %s", (int)e_, hipGetErrorString(e_), __FILE__, __LINE__, #expr);         \
            return (e_ == hipErrorOutOfMemory) ? ARP_ERR_OOM : ARP_ERR_HIP;                                                 \
        }                                                                                                                   \
    } while (0)

// the structure's arrays on the context's device: uploaded once; only the attribute words depend on the chain groups
arp_status ensure_resident(arp_context *ctx, arp_structure *s, TableCache *c, const char *groups) {
    DevStructure &d = c->dev;
    const int device = context_device(ctx);
    hipStream_t st = (hipStream_t)context_stream(ctx);
    TBL_HIP(hipSetDevice(device));
    const uint64_t n = s->n, nr = s->residues.size(), nh = s->res_h_idx.size(), nm = c->model_rank.size();
    if (!d.block || d.device != device) {
        if (d.block) { (void)hipSetDevice(d.device); (void)hipFree(d.block); if (d.derived) (void)hipFree(d.derived); if (d.rings_block) (void)hipFree(d.rings_block); (void)hipSetDevice(device); d.block = nullptr; d.derived = nullptr; d.rings_block = nullptr; d.rings_cap = 0; d.rings_host.clear(); }
        struct Seg { const void *src; uint64_t bytes; void **dst; };
        Seg seg[] = {{s->x.data(), n * 8, (void **)&d.x}, {s->y.data(), n * 8, (void **)&d.y}, {s->z.data(), n * 8, (void **)&d.z},
                     {s->attr.data(), n * 4, (void **)&d.attr}, {s->res_ord.data(), n * 4, (void **)&d.res_ord}, {s->res_id.data(), n * 4, (void **)&d.res_id},
                     {s->res_h_ptr.data(), (nr + 1) * 4, (void **)&d.res_h_ptr}, {s->res_h_idx.data(), nh * 4, (void **)&d.res_h_idx},
                     {s->res_cb.data(), nr * 4, (void **)&d.res_cb}, {s->res_sg.data(), nr * 4, (void **)&d.res_sg},
                     {s->chain_rank.data(), n * 4, (void **)&d.chain_rank}, {s->model.data(), n * 4, (void **)&d.model},
                     {c->plane_bits.data(), n, (void **)&d.plane_bits}, {c->res_atom_ptr.data(), (nr + 1) * 4, (void **)&d.res_atom_ptr},
                     {c->res_atom_idx.data(), c->res_atom_idx.size() * 4, (void **)&d.res_atom_idx}, {c->atom_sc_src.data(), n * 4, (void **)&d.atom_sc_src},
                     {c->atom_keys.data(), n * sizeof(EntKey), (void **)&d.ent_key}, {c->model_rank.data(), nm * 4, (void **)&d.model_rank},
                     {c->model_serial_of.data(), nm * 4, (void **)&d.model_serial_of}};
        uint64_t total = 0;
        for (const Seg &g : seg) total += (g.bytes + 255u) & ~255ull;
        total = std::max<uint64_t>(total, 256);
        const bool timing = g_debug.timing != 0;
        auto t_prev = std::chrono::steady_clock::now();
        auto lap = [&](const char *what) {
            if (!timing) return;
            auto now = std::chrono::steady_clock::now();
            fprintf(stderr, "    resident %-25s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
            t_prev = now;
        };
        TBL_HIP(hipMalloc((void **)&d.block, total));
        lap("device block");
        d.device = device;
        // one pinned staging block, filled by the host workers, one copy across PCIe (pageable arrays cross at ~3 GB/s)
        char *dev_scr = nullptr, *pin = nullptr;
        arp_status stt = context_scratch(ctx, 1, 0, total, &dev_scr, &pin);
        if (stt != ARP_OK) return stt;
        lap("pinned staging block");
        uint64_t off = 0;
        std::vector<uint64_t> offs;
        for (const Seg &g : seg) { offs.push_back(off); *g.dst = d.block + off; off += (g.bytes + 255u) & ~255ull; }
        const size_t n_seg = sizeof seg / sizeof seg[0];
        parallel_for(n_seg * 8, 1, [&](size_t k0, size_t k1, size_t) {  // every segment in 8 slices
            for (size_t k = k0; k < k1; k++) {
                const Seg &g = seg[k / 8];
                const uint64_t lo = g.bytes * (k % 8) / 8, hi = g.bytes * (k % 8 + 1) / 8;
                if (hi > lo) memcpy(pin + offs[k / 8] + lo, (const char *)g.src + lo, hi - lo);
            }
        });
        lap("gather into it");
        TBL_HIP(hipMemcpyAsync(d.block, pin, total, hipMemcpyHostToDevice, st));
        TBL_HIP(hipStreamSynchronize(st));  // (the pinned block is scratch: reused by the table pass below)
        lap("H2D");
        d.n = n; d.n_res = nr; d.n_h = nh;
        d.n_chains = (uint32_t)s->chain_ids.size(); d.n_models = (uint32_t)nm;
        d.any_icode = false;
        for (uint64_t a = 1; a < n && !d.any_icode; a++) d.any_icode = c->atom_keys[a].icode != c->atom_keys[0].icode;  // (rings carry their residue's: the same values)
        d.attr_groups = groups;
        lap("insertion-code scan");
    } else if (d.attr_groups != groups) {
        TBL_HIP(hipMemcpyAsync(d.attr, s->attr.data(), n * 4, hipMemcpyHostToDevice, st));
        TBL_HIP(hipStreamSynchronize(st));
        d.attr_groups = groups;
    }
    return ARP_OK;
}

arp_status get_contacts_device(arp_context *ctx, arp_structure *s, const char *groups, double vdw_comp, double dist_cutoff, arp_table **out) {
    const bool timing = g_debug.timing != 0;
    auto t_prev = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "  get_contacts %-22s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    TableCache *c = table_cache_of(s);
    std::lock_guard<std::mutex> one_call_per_structure(c->mu);
    {   // InteractionComplex::new (complex.rs:36-68): the ligand / receptor bits of this chain-group spec (a no-op when unchanged)
        const arp_status sg = apply_groups(s, groups);
        if (sg != ARP_OK) return sg;
    }
    if (c->rings.empty()) { set_error("Error building ring positions"); return ARP_ERR_NO_RINGS; }  // complex.rs:50
    lap("entities (cached)");
    arp_status st = ensure_resident(ctx, s, c, groups);
    if (st != ARP_OK) return st;
    lap("resident copy");
    // chain sets of this call (utils.rs:71-115): chains without atoms still belong to them.  Kept per chain-group spec: on a structure
    // with thousands of chains this bookkeeping costs more than the GPU pair pass.
    if (!c->have_rings_dev || c->rings_groups != groups) {
        std::vector<char> chain_l(s->chain_ids.size(), 0), chain_r(s->chain_ids.size(), 0);
        std::unordered_map<std::string, uint32_t> rank;
        for (size_t k = 0; k < s->chain_ids.size(); k++) rank[s->chain_ids[k]] = (uint32_t)k;
        {
            std::vector<std::string> L, R;
            if ((st = parse_groups(s->chain_ids, groups, &L, &R)) != ARP_OK) return st;
            for (auto &ch : L) chain_l[rank[ch]] = 1;
            for (auto &ch : R) chain_r[rank[ch]] = 1;
        }
        c->rings_dev.resize(c->rings.size());
        for (size_t k = 0; k < c->rings_dev.size(); k++) {
            PlaneEntry &e = c->rings[k];
            e.chain_rank = rank[e.chain]; e.in_l = chain_l[e.chain_rank]; e.in_r = chain_r[e.chain_rank];
            c->rings_dev[k] = RingEnt{e.res, e.model_serial, c->ring_model_rank[k], e.chain_rank, (e.in_l ? 1u : 0u) | (e.in_r ? 2u : 0u) | (e.has_ord ? 4u : 0u), e.ord, c->ring_sc_src[k], 0u};
        }
        c->rings_groups = groups; c->have_rings_dev = true;
    }
    const std::vector<RingEnt> &rings = c->rings_dev;
    // get_atomic_contacts (complex.rs:189-299): the GPU hot path, on the resident arrays, list left on the device
    arp_params prm;
    arp_default_params(&prm);
    prm.vdw_comp = vdw_comp; prm.dist_cutoff = dist_cutoff;
    prm.flags |= ARP_FLAG_CONTACTS_ONLY;  // only pairs with an interaction become rows
    DevStructure &d = c->dev;
    arp_atoms dv{};
    dv.n = d.n; dv.x = d.x; dv.y = d.y; dv.z = d.z; dv.attr = d.attr; dv.res_ord = d.res_ord; dv.chain_rank = d.chain_rank; dv.model = d.model;
    dv.res_id = d.res_id; dv.n_res = d.n_res; dv.res_h_ptr = d.res_h_ptr; dv.res_h_idx = d.res_h_idx; dv.res_cb = d.res_cb; dv.res_sg = d.res_sg;
    dv.location = ARP_MEM_DEVICE;
    const arp_pair *pairs = nullptr;  // a view of the context's own buffer: one pass, no allocation
    uint64_t n_pairs = 0;
    if ((st = contacts_atomic_view(ctx, &dv, &prm, &pairs, &n_pairs)) != ARP_OK) return st;
    lap("atomic pairs (GPU)");
    TableRowsHost rows;
    st = device_table(ctx, d, rings, c->ring_keys, pairs, n_pairs, dist_cutoff, &rows);
    if (st != ARP_OK) return st;
    lap("device table");
    // the table keeps the rows as they came back and a reference to the structure's entity book: no per-row host work here
    arp_table *t = new arp_table();
    t->n = rows.n;
    t->rows_owner = std::move(rows.owner); t->rows = rows.rows; t->sc = rows.sc;
    c->wait_for_book();
    t->book = c->book;
    lap("table object");
    *out = t;
    return ARP_OK;
}
}  // namespace

#ifdef ARP_WITH_HOST_TABLE
#include "../../tests/hosttable/table_host.inl"  // test-only: not part of the product library (build.py: build_host_table_library)
#endif

extern "C" arp_status arp_get_contacts(arp_context *ctx, arp_structure *s, const char *groups, double vdw_comp, double dist_cutoff,
                                       arp_table **out) {
    return arp_get_contacts_mt(ctx, s, groups, vdw_comp, dist_cutoff, -1, out);
}

extern "C" arp_status arp_get_contacts_mt(arp_context *ctx, arp_structure *s, const char *groups, double vdw_comp, double dist_cutoff,
                                          int32_t num_threads, arp_table **out) try {
    if (!s || !out || !groups) { set_error("null argument"); return ARP_ERR_BAD_INPUT; }
    *out = nullptr;
    HostThreadsScope threads(num_threads);  // one worker count for every pass of this call
#ifdef ARP_WITH_HOST_TABLE   // test-only build (tests/hosttable): the round-1 host assembly as a cross-check of the device table
    if (g_debug.table_host) return get_contacts_host(ctx, s, groups, vdw_comp, dist_cutoff, out);
#endif
    return get_contacts_device(ctx, s, groups, vdw_comp, dist_cutoff, out);
} ARP_ABI_CATCH

// The planes the device fits (residues.rs:270-298), per residue of the filtered model in hierarchy order: 12 doubles
// {ring centre, ring normal, sc centre, sc normal}; valid[r] bit 1 = ring plane, bit 2 = side-chain plane.
extern "C" uint64_t arp_structure_n_residues(const arp_structure *s) { return s ? s->residues.size() : 0; }
extern "C" arp_status arp_structure_fit_planes(arp_context *ctx, arp_structure *s, double *planes, uint8_t *valid) try {
    if (!ctx || !s || !planes || !valid) { set_error("null argument"); return ARP_ERR_BAD_INPUT; }
    arp_atoms view;
    arp_status st = arp_structure_atoms(s, s->groups_valid ? s->groups_applied.c_str() : "/", &view);
    if (st != ARP_OK) return st;
    TableCache *c = table_cache_of(s);
    if ((st = ensure_resident(ctx, s, c, s->groups_applied.c_str())) != ARP_OK) return st;
    std::vector<double> p;
    std::vector<uint8_t> v;
    if ((st = device_planes(ctx, c->dev, &p, &v)) != ARP_OK) return st;
    memcpy(planes, p.data(), p.size() * sizeof(double));
    memcpy(valid, v.data(), v.size());
    return ARP_OK;
} ARP_ABI_CATCH

extern "C" void arp_table_free(arp_table *t) { delete t; }
extern "C" uint64_t arp_table_rows(const arp_table *t) { return t ? t->n : 0; }
extern "C" const void *arp_table_column(const arp_table *t_const, const char *name, int32_t *width) {
    if (!t_const || !name) return nullptr;
    arp_table *t = const_cast<arp_table *>(t_const);
    if (t->book) std::call_once(t->columns_once, [t]() { materialize_columns(t); });
    std::string c(name);
    auto num = [&](const void *p, int w) -> const void * { if (width) *width = w; return p; };
    if (c == "model") return num(t->model.data(), 4);
    if (c == "interaction") return num(t->interaction.data(), 4);
    if (c == "distance") return num(t->distance.data(), 4);
    if (c == "from_resi") return num(t->from_resi.data(), 4);
    if (c == "from_atomi") return num(t->from_atomi.data(), 4);
    if (c == "to_resi") return num(t->to_resi.data(), 4);
    if (c == "to_atomi") return num(t->to_atomi.data(), 4);
    if (c == "from_atom") return num(t->from_atom.data(), 4);
    if (c == "to_atom") return num(t->to_atom.data(), 4);
    if (c == "sc_centroid_dist") return num(t->sc_dist.data(), 4);
    if (c == "sc_dihedral") return num(t->sc_dihedral.data(), 4);
    if (c == "sc_centroid_angle") return num(t->sc_angle.data(), 4);
    if (c == "sc_valid") return num(t->sc_valid.data(), 1);
    if (c == "from_chain") return num(t->from_chain.buf.data(), 8);
    if (c == "from_resn") return num(t->from_resn.buf.data(), 8);
    if (c == "from_atomn") return num(t->from_atomn.buf.data(), 8);
    if (c == "to_chain") return num(t->to_chain.buf.data(), 8);
    if (c == "to_resn") return num(t->to_resn.buf.data(), 8);
    if (c == "to_atomn") return num(t->to_atomn.buf.data(), 8);
    if (c == "from_insertion") return num(t->from_insertion.buf.data(), 4);
    if (c == "from_altloc") return num(t->from_altloc.buf.data(), 4);
    if (c == "to_insertion") return num(t->to_insertion.buf.data(), 4);
    if (c == "to_altloc") return num(t->to_altloc.buf.data(), 4);
    return nullptr;
}

// ---- Arrow C Data Interface export (mod.rs:140-214: the DataFrame the reference returns) ---------------------------------
namespace {
struct ArrowCol {  // owns the buffers of one child array
    std::vector<uint8_t> validity;
    std::vector<int32_t> offsets;
    std::vector<char> bytes;
    const void *bufs[3] = {nullptr, nullptr, nullptr};
};
struct ArrowBatch {  // private data of the struct array
    std::vector<ArrowArray> kids;
    std::vector<ArrowArray *> kid_ptrs;
    const void *bufs[1] = {nullptr};
};
struct ArrowFields {  // private data of the struct schema
    std::vector<ArrowSchema> kids;
    std::vector<ArrowSchema *> kid_ptrs;
};
void release_col(ArrowArray *a) {
    delete (ArrowCol *)a->private_data;
    a->release = nullptr;
}
void release_batch(ArrowArray *a) {
    ArrowBatch *b = (ArrowBatch *)a->private_data;
    for (ArrowArray &k : b->kids)
        if (k.release) k.release(&k);
    delete b;
    a->release = nullptr;
}
void release_leaf_schema(ArrowSchema *s) { s->release = nullptr; }
void release_fields(ArrowSchema *s) {
    ArrowFields *f = (ArrowFields *)s->private_data;
    for (ArrowSchema &k : f->kids)
        if (k.release) k.release(&k);
    delete f;
    s->release = nullptr;
}
template <typename T>
ArrowCol *numeric_col(const std::vector<T> &v) {
    ArrowCol *c = new ArrowCol();
    c->bytes.resize(v.size() * sizeof(T) + 8);
    memcpy(c->bytes.data(), v.data(), v.size() * sizeof(T));
    c->bufs[1] = c->bytes.data();
    return c;
}
template <int W>
ArrowCol *utf8_col(const StrCol<W> &v) {
    ArrowCol *c = new ArrowCol();
    const size_t n = v.size();
    c->offsets.resize(n + 1);
    c->bytes.reserve(n * 3 + 8);
    int32_t o = 0;
    for (size_t i = 0; i < n; i++) {
        c->offsets[i] = o;
        const char *p = v.at(i);
        int len = 0;
        while (len < W && p[len]) len++;
        c->bytes.insert(c->bytes.end(), p, p + len);
        o += len;
    }
    c->offsets[n] = o;
    c->bytes.resize(c->bytes.size() + 8);  // never hand out a null data pointer
    c->bufs[1] = c->offsets.data(); c->bufs[2] = c->bytes.data();
    return c;
}
}  // namespace

namespace {
// Arrow export of a device-path table: every buffer of the 20 columns straight from the rows and the entity book, in two
// passes over row chunks on the host workers (string bytes per chunk, then the chunk's offsets / bytes / numbers / validity bits).
// No intermediate fixed-width columns, no zero fill, no per-row allocation.
struct RawCol {  // owns malloc'd buffers of one child array
    void *validity = nullptr, *b1 = nullptr, *b2 = nullptr;
    const void *bufs[3] = {nullptr, nullptr, nullptr};
    ~RawCol() { free(validity); free(b1); free(b2); }
};
void release_raw_col(ArrowArray *a) {
    delete (RawCol *)a->private_data;
    a->release = nullptr;
}
arp_status export_arrow_fast(const arp_table *t, ArrowArray *out_array, ArrowSchema *out_schema) {
    const size_t n = t->n;
    const EntityBook &bk = *t->book;
    constexpr size_t kChunkRows = 1u << 15;  // a multiple of 8: validity bytes are never shared between chunks
    const size_t n_chunks = (n + kChunkRows - 1) / kChunkRows;
    // string columns: 0-4 from_{chain, resn, insertion, altloc, atomn}, 5-9 to_..., 10 interaction
    constexpr int kStr = 11;
    static const int shift_of[5] = {0, 4, 12, 16, 8};  // EntityBook::lens nibbles in the order chain, resn, insertion, altloc, atomn
    uint32_t name_len[ARP_N_INTERACTIONS];
    for (int k = 0; k < ARP_N_INTERACTIONS; k++) name_len[k] = (uint32_t)strlen(arp_interaction_name(k));
    std::vector<uint64_t> bytes((n_chunks + 1) * kStr, 0);
    parallel_for(n_chunks, 1, [&](size_t c0, size_t c1, size_t) {
        for (size_t c = c0; c < c1; c++) {
            uint64_t acc[kStr] = {0};
            const size_t k1 = std::min(n, (c + 1) * kChunkRows);
            for (size_t k = c * kChunkRows; k < k1; k++) {
                const TableRow &r = t->rows[k];
                const uint32_t lf = bk.lens[r.from_ent], lt = bk.lens[r.to_ent];
                for (int q = 0; q < 5; q++) { acc[q] += (lf >> shift_of[q]) & 15u; acc[5 + q] += (lt >> shift_of[q]) & 15u; }
                acc[10] += name_len[r.interaction];
            }
            for (int q = 0; q < kStr; q++) bytes[(c + 1) * kStr + q] = acc[q];
        }
    });
    for (size_t c = 0; c < n_chunks; c++) for (int q = 0; q < kStr; q++) bytes[(c + 1) * kStr + q] += bytes[c * kStr + q];
    for (int q = 0; q < kStr; q++) if (bytes[n_chunks * kStr + q] > 0x7FFFFFF0ull) { set_error("table too large for 32-bit utf8 offsets"); return ARP_ERR_BAD_INPUT; }
    auto raw = [](size_t count, size_t elem) { return malloc(std::max<size_t>(count * elem, 8) + 8); };
    RawCol *str[kStr];
    for (int q = 0; q < kStr; q++) {
        str[q] = new RawCol();
        str[q]->b1 = raw(n + 1, 4); str[q]->b2 = raw(bytes[n_chunks * kStr + q], 1);
        str[q]->bufs[1] = str[q]->b1; str[q]->bufs[2] = str[q]->b2;
    }
    // numeric columns: model u32, distance f32, from_resi, from_atomi, to_resi, to_atomi i32, sc_* f32 with validity
    enum { MODEL, DIST, FRESI, FATOMI, TRESI, TATOMI, SCD, SCDIH, SCANG, N_NUM };
    RawCol *num[N_NUM];
    for (int q = 0; q < N_NUM; q++) {
        num[q] = new RawCol();
        num[q]->b1 = raw(n, 4); num[q]->bufs[1] = num[q]->b1;
        if (q >= SCD) { num[q]->validity = raw((n + 7) / 8, 1); num[q]->bufs[0] = num[q]->validity; }
    }
    std::vector<uint64_t> nulls(n_chunks, 0);
    parallel_for(n_chunks, 1, [&](size_t c0, size_t c1, size_t) {
        for (size_t c = c0; c < c1; c++) {
            uint64_t off[kStr];
            for (int q = 0; q < kStr; q++) off[q] = bytes[c * kStr + q];
            const size_t k1 = std::min(n, (c + 1) * kChunkRows);
            uint64_t chunk_nulls = 0;
            for (size_t k = c * kChunkRows; k < k1; k++) {
                const TableRow &r = t->rows[k];
                const TableSc &sc = t->sc[k];
                const EntityRec &f = bk.ent[r.from_ent], &o = bk.ent[r.to_ent];
                const uint32_t lf = bk.lens[r.from_ent], lt = bk.lens[r.to_ent];
                const char *fs[5] = {f.chain, f.resn, f.insertion, f.altloc, f.atomn}, *ts[5] = {o.chain, o.resn, o.insertion, o.altloc, o.atomn};
                for (int q = 0; q < 5; q++) {
                    const uint32_t a = (lf >> shift_of[q]) & 15u, b = (lt >> shift_of[q]) & 15u;
                    ((int32_t *)str[q]->b1)[k] = (int32_t)off[q]; memcpy((char *)str[q]->b2 + off[q], fs[q], a); off[q] += a;
                    ((int32_t *)str[5 + q]->b1)[k] = (int32_t)off[5 + q]; memcpy((char *)str[5 + q]->b2 + off[5 + q], ts[q], b); off[5 + q] += b;
                }
                ((int32_t *)str[10]->b1)[k] = (int32_t)off[10];
                memcpy((char *)str[10]->b2 + off[10], arp_interaction_name(r.interaction), name_len[r.interaction]); off[10] += name_len[r.interaction];
                ((uint32_t *)num[MODEL]->b1)[k] = f.model; ((float *)num[DIST]->b1)[k] = r.distance;
                ((int32_t *)num[FRESI]->b1)[k] = f.resi; ((int32_t *)num[FATOMI]->b1)[k] = f.atomi;
                ((int32_t *)num[TRESI]->b1)[k] = o.resi; ((int32_t *)num[TATOMI]->b1)[k] = o.atomi;
                ((float *)num[SCD]->b1)[k] = sc.dist; ((float *)num[SCDIH]->b1)[k] = sc.dihedral; ((float *)num[SCANG]->b1)[k] = sc.angle;
                chunk_nulls += sc.valid == 0.f;
            }
            for (size_t k = c * kChunkRows; k < k1; k += 8) {  // validity bits, one byte per eight rows (null where either residue has no sc plane)
                uint8_t v = 0;
                for (size_t b = 0; b < 8 && k + b < k1; b++) v |= (uint8_t)((t->sc[k + b].valid != 0.f ? 1u : 0u) << b);
                for (int q = SCD; q < N_NUM; q++) ((uint8_t *)num[q]->validity)[k >> 3] = v;
            }
            nulls[c] = chunk_nulls;
        }
    });
    for (int q = 0; q < kStr; q++) ((int32_t *)str[q]->b1)[n] = (int32_t)bytes[n_chunks * kStr + q];
    int64_t sc_nulls = 0;
    for (uint64_t v : nulls) sc_nulls += (int64_t)v;
    struct Field { const char *name, *format; RawCol *col; int64_t n_buffers; bool nullable; };
    const Field fields[20] = {
        {"model", "I", num[MODEL], 2, false}, {"interaction", "u", str[10], 3, false}, {"distance", "f", num[DIST], 2, false},
        {"from_chain", "u", str[0], 3, false}, {"from_resn", "u", str[1], 3, false}, {"from_resi", "i", num[FRESI], 2, false},
        {"from_insertion", "u", str[2], 3, false}, {"from_altloc", "u", str[3], 3, false}, {"from_atomn", "u", str[4], 3, false}, {"from_atomi", "i", num[FATOMI], 2, false},
        {"to_chain", "u", str[5], 3, false}, {"to_resn", "u", str[6], 3, false}, {"to_resi", "i", num[TRESI], 2, false},
        {"to_insertion", "u", str[7], 3, false}, {"to_altloc", "u", str[8], 3, false}, {"to_atomn", "u", str[9], 3, false}, {"to_atomi", "i", num[TATOMI], 2, false},
        {"sc_centroid_dist", "f", num[SCD], 2, true}, {"sc_dihedral", "f", num[SCDIH], 2, true}, {"sc_centroid_angle", "f", num[SCANG], 2, true}};
    ArrowBatch *b = new ArrowBatch();
    ArrowFields *f = new ArrowFields();
    b->kids.resize(20); f->kids.resize(20);
    for (size_t k = 0; k < 20; k++) {
        ArrowArray &a = b->kids[k];
        a = ArrowArray{};
        a.length = (int64_t)n; a.null_count = fields[k].nullable ? sc_nulls : 0; a.offset = 0;
        a.n_buffers = fields[k].n_buffers; a.n_children = 0;
        a.buffers = fields[k].col->bufs; a.children = nullptr; a.dictionary = nullptr;
        a.release = release_raw_col; a.private_data = fields[k].col;
        b->kid_ptrs.push_back(&a);
        ArrowSchema &sch = f->kids[k];
        sch = ArrowSchema{};
        sch.format = fields[k].format; sch.name = fields[k].name; sch.metadata = nullptr;
        sch.flags = fields[k].nullable ? 2 /* ARROW_FLAG_NULLABLE */ : 0;
        sch.n_children = 0; sch.children = nullptr; sch.dictionary = nullptr;
        sch.release = release_leaf_schema; sch.private_data = nullptr;
        f->kid_ptrs.push_back(&sch);
    }
    *out_array = ArrowArray{};
    out_array->length = (int64_t)n; out_array->null_count = 0; out_array->offset = 0;
    out_array->n_buffers = 1; out_array->buffers = b->bufs;
    out_array->n_children = 20; out_array->children = b->kid_ptrs.data(); out_array->dictionary = nullptr;
    out_array->release = release_batch; out_array->private_data = b;
    *out_schema = ArrowSchema{};
    out_schema->format = "+s"; out_schema->name = ""; out_schema->metadata = nullptr; out_schema->flags = 0;
    out_schema->n_children = 20; out_schema->children = f->kid_ptrs.data(); out_schema->dictionary = nullptr;
    out_schema->release = release_fields; out_schema->private_data = f;
    return ARP_OK;
}
}  // namespace

extern "C" arp_status arp_table_export_arrow(const arp_table *t, ArrowArray *out_array, ArrowSchema *out_schema) try {
    if (!t || !out_array || !out_schema) { set_error("null argument"); return ARP_ERR_BAD_INPUT; }
    if (t->n > 0x7FFFFFF0ull) { set_error("table too large for 32-bit utf8 offsets"); return ARP_ERR_BAD_INPUT; }
    if (t->book) return export_arrow_fast(t, out_array, out_schema);
    const int64_t n = (int64_t)t->n;
    struct Field { const char *name, *format; ArrowCol *col; int64_t n_buffers; bool nullable; };
    auto names_col = [&]() {  // interaction code -> the reference's Display string (structs.rs:6-51)
        ArrowCol *c = new ArrowCol();
        c->offsets.resize(n + 1);
        int32_t o = 0;
        for (int64_t i = 0; i < n; i++) {
            c->offsets[i] = o;
            const char *s = arp_interaction_name(t->interaction[i]);
            const size_t len = strlen(s);
            c->bytes.insert(c->bytes.end(), s, s + len);
            o += (int32_t)len;
        }
        c->offsets[n] = o;
        c->bytes.resize(c->bytes.size() + 8);
        c->bufs[1] = c->offsets.data(); c->bufs[2] = c->bytes.data();
        return c;
    };
    int64_t sc_nulls = 0;
    auto sc_col = [&](const std::vector<float> &v) {  // null where either residue has no side-chain plane (mod.rs:100-110 left join)
        ArrowCol *c = numeric_col(v);
        c->validity.assign((size_t)(n + 7) / 8 + 8, 0);
        sc_nulls = 0;
        for (int64_t i = 0; i < n; i++) {
            if (t->sc_valid[i]) c->validity[i >> 3] |= (uint8_t)(1u << (i & 7));
            else sc_nulls++;
        }
        c->bufs[0] = c->validity.data();
        return c;
    };
    std::vector<Field> fields = {
        {"model", "I", numeric_col(t->model), 2, false}, {"interaction", "u", names_col(), 3, false}, {"distance", "f", numeric_col(t->distance), 2, false},
        {"from_chain", "u", utf8_col(t->from_chain), 3, false}, {"from_resn", "u", utf8_col(t->from_resn), 3, false},
        {"from_resi", "i", numeric_col(t->from_resi), 2, false}, {"from_insertion", "u", utf8_col(t->from_insertion), 3, false},
        {"from_altloc", "u", utf8_col(t->from_altloc), 3, false}, {"from_atomn", "u", utf8_col(t->from_atomn), 3, false},
        {"from_atomi", "i", numeric_col(t->from_atomi), 2, false},
        {"to_chain", "u", utf8_col(t->to_chain), 3, false}, {"to_resn", "u", utf8_col(t->to_resn), 3, false},
        {"to_resi", "i", numeric_col(t->to_resi), 2, false}, {"to_insertion", "u", utf8_col(t->to_insertion), 3, false},
        {"to_altloc", "u", utf8_col(t->to_altloc), 3, false}, {"to_atomn", "u", utf8_col(t->to_atomn), 3, false},
        {"to_atomi", "i", numeric_col(t->to_atomi), 2, false},
        {"sc_centroid_dist", "f", sc_col(t->sc_dist), 2, true}, {"sc_dihedral", "f", sc_col(t->sc_dihedral), 2, true},
        {"sc_centroid_angle", "f", sc_col(t->sc_angle), 2, true},
    };
    ArrowBatch *b = new ArrowBatch();
    ArrowFields *f = new ArrowFields();
    b->kids.resize(fields.size()); f->kids.resize(fields.size());
    for (size_t k = 0; k < fields.size(); k++) {
        ArrowArray &a = b->kids[k];
        a = ArrowArray{};
        a.length = n; a.null_count = fields[k].nullable ? sc_nulls : 0; a.offset = 0;
        a.n_buffers = fields[k].n_buffers; a.n_children = 0;
        a.buffers = fields[k].col->bufs; a.children = nullptr; a.dictionary = nullptr;
        a.release = release_col; a.private_data = fields[k].col;
        b->kid_ptrs.push_back(&a);
        ArrowSchema &s = f->kids[k];
        s = ArrowSchema{};
        s.format = fields[k].format; s.name = fields[k].name; s.metadata = nullptr;
        s.flags = fields[k].nullable ? 2 /* ARROW_FLAG_NULLABLE */ : 0;
        s.n_children = 0; s.children = nullptr; s.dictionary = nullptr;
        s.release = release_leaf_schema; s.private_data = nullptr;
        f->kid_ptrs.push_back(&s);
    }
    *out_array = ArrowArray{};
    out_array->length = n; out_array->null_count = 0; out_array->offset = 0;
    out_array->n_buffers = 1; out_array->buffers = b->bufs;
    out_array->n_children = (int64_t)fields.size(); out_array->children = b->kid_ptrs.data(); out_array->dictionary = nullptr;
    out_array->release = release_batch; out_array->private_data = b;
    *out_schema = ArrowSchema{};
    out_schema->format = "+s"; out_schema->name = ""; out_schema->metadata = nullptr; out_schema->flags = 0;
    out_schema->n_children = (int64_t)fields.size(); out_schema->children = f->kid_ptrs.data(); out_schema->dictionary = nullptr;
    out_schema->release = release_fields; out_schema->private_data = f;
    return ARP_OK;
} ARP_ABI_CATCH
